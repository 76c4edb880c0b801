"""Direct per-method calls through the C ABI (ivp_options_t.has_settings): `DOPRI5 {..}.solve()`, `DOP853 {..}.solve()`,
`RK23 {..}.solve()` with non-default struct fields, and the validation XXX::solve() performs on them
(dopri5.rs:143-198, dop853.rs:135-193, rk23.rs:102-129).  The bit-exact parity cases with settings live in
tests/cases.py ("settings-*") and run in test_gpu_parity.py."""
import numpy as np
import pytest

import ivp_amd
from oracle import oracle as O

pytestmark = pytest.mark.gpu

Y0 = np.array([[1.0], [0.0]])


def _code(method, settings, **kw):
    with pytest.raises(ivp_amd.ConfigError) as e:
        ivp_amd.solve_ivp_batch(ivp_amd.SHO(), 0.0, 1.0, Y0, None, ivp_amd.Options(method=method, settings=settings, **kw))
    return e.value.code


@pytest.mark.parametrize("method", ["DOPRI5", "DOP853"])
def test_dopri_validation_codes_match_reference(method):
    assert _code(method, dict(uround=1e-36)) == -2          # OutOfRange { parameter: "uround" }
    assert _code(method, dict(uround=1.0)) == -2
    assert _code(method, dict(safety_factor=1.0)) == -2     # OutOfRange { parameter: "safety_factor" }
    assert _code(method, dict(safety_factor=1e-4)) == -2
    assert _code(method, dict(beta=0.21)) == -2             # OutOfRange { parameter: "beta" }
    assert _code(method, dict(stiff_test=0)) == -1          # MustBePositive { parameter: "stiff_test" }
    # first failing check wins, in the reference's order: uround before stiff_test
    assert _code(method, dict(uround=2.0, stiff_test=0)) == -2
    # the oracle returns the same codes
    for st, code in ((dict(uround=1e-36), -2), (dict(beta=0.21), -2), (dict(stiff_test=0), -1)):
        with pytest.raises(ValueError, match=str(code)):
            O.solve_ivp("sho", 0.0, 1.0, [1.0, 0.0], method=method, settings=st)


def test_rk23_validation_codes_match_reference():
    assert _code("RK23", dict(safety_factor=1.5)) == -2
    assert _code("RK23", dict(scale_min=0.0)) == -6         # InvalidScaleFactors
    assert _code("RK23", dict(scale_min=2.0, scale_max=2.0)) == -6
    # RK23 has no uround / beta / stiff_test fields: values that DOPRI5 would reject are not looked at
    r = ivp_amd.solve_ivp_batch(ivp_amd.SHO(), 0.0, 1.0, Y0, None, ivp_amd.Options(method="RK23", settings=dict(beta=0.5)))
    assert int(r.status[0]) == 0


def test_settings_rejected_for_rk4_and_bdf():
    for m in ("RK4", "BDF"):
        with pytest.raises(ivp_amd.ConfigError):
            ivp_amd.solve_ivp_batch(ivp_amd.SHO(), 0.0, 1.0, Y0, None, ivp_amd.Options(method=m, settings={}))


def test_empty_settings_equal_the_defaults_bit_for_bit():
    """has_settings with the struct defaults is the same computation as solve_ivp()'s path."""
    y0, p, t0, t1 = ivp_amd.workloads.cr3bp_batch(512)
    for m, tol in (("DOPRI5", (1e-6, 1e-9)), ("DOP853", (1e-8, 1e-10)), ("RK23", (1e-4, 1e-7))):
        a = ivp_amd.solve_ivp_batch(ivp_amd.CR3BP(), t0, t1, y0, p, ivp_amd.Options(method=m, rtol=tol[0], atol=tol[1]))
        b = ivp_amd.solve_ivp_batch(ivp_amd.CR3BP(), t0, t1, y0, p, ivp_amd.Options(method=m, rtol=tol[0], atol=tol[1], settings={},
                                                                                  max_steps=10 ** 9))
        assert np.array_equal(a.y_end, b.y_end) and np.array_equal(a.naccpt, b.naccpt) and np.array_equal(a.nfev, b.nfev)


def test_settings_on_the_large_n_path():
    rng = np.random.default_rng(1)
    y0 = rng.uniform(-1, 1, (100, 40))
    st = dict(safety_factor=0.8, beta=0.0, scale_max=5.0, stiff_test=3)
    ref = O.solve_batch("linear_decay100", y0, None, 0.0, 6.0, detpow=True, method="DOPRI5", rtol=1e-7, atol=1e-9, settings=st)
    got = ivp_amd.solve_ivp_batch(ivp_amd.LinearDecay100(), 0.0, 6.0, y0, None,
                                  ivp_amd.Options(method="DOPRI5", rtol=1e-7, atol=1e-9, settings=st))
    assert np.array_equal(got.y_end, ref["y_end"]) and np.array_equal(got.naccpt, ref["naccpt"])
    assert np.array_equal(got.nrejct, ref["nrejct"]) and np.array_equal(got.status, ref["status"])
