"""One-pass accepted-step log, CPU leg: the paged form of the device DefaultSolOut (so_push_log in rk_core.h with
IvpKArgs.log_pool) run on the host-compiled kernel bodies (tests/host_emul), laid out as a CSR log by a numpy rendering
of log_gather.hip, and compared record for record with the dense [max_log] log of the same bodies and with the oracle's
Solution.t / Solution.y (the reference: src/solve/solout.rs:387-428, src/solve/solve_ivp.rs:288-312).
The GPU leg (the real pool, the real gather kernel, the C ABI) is tests/test_gpu_one_pass_log.py."""
import numpy as np
import pytest

from ivp_amd import workloads as W
from oracle import oracle as O
from tests.common import emul_batch
from tests.host_emul import emul as E


def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


def _dense_as_csr(res):
    cnt = res["n_log"].astype(np.int64)
    t = np.concatenate([res["t_log"][:c, b] for b, c in enumerate(cnt)]) if cnt.sum() else np.zeros(0)
    y = np.concatenate([res["y_log"][:c, :, b] for b, c in enumerate(cnt)]) if cnt.sum() else np.zeros((0, res["y_log"].shape[1]))
    return t, y


CASES = [
    ("cr3bp/DOPRI5", "cr3bp", lambda: W.cr3bp_batch(24), dict(method="DOPRI5", rtol=1e-6, atol=1e-9)),
    ("vdp/DOP853", "vdp", lambda: W.vdp_batch(40), dict(method="DOP853", rtol=1e-8, atol=1e-10)),
    ("vdp/RK23", "vdp", lambda: (W.vdp_batch(9)[0], W.vdp_batch(9)[1], 0.0, 6.0), dict(method="RK23", rtol=1e-4, atol=1e-7)),
    ("sho/RK4", "sho", lambda: (np.array([[1.0, 0.5], [0.0, 0.2]]), None, 0.0, 3.0), dict(method="RK4", first_step=0.01)),
    ("vdp_eps/BDF", "vdp_eps", lambda: (np.array([[2.0, 1.5], [0.0, 0.1]]), np.array([[1e-2, 2e-2]]), 0.0, 2.0), dict(method="BDF", rtol=1e-4, atol=1e-7)),
    ("cr3bp/DOPRI5/first_step", "cr3bp", lambda: W.cr3bp_batch(6), dict(method="DOPRI5", rtol=1e-6, atol=1e-9, first_step=1e-3)),
]


@pytest.mark.parametrize("name,rhs,make,opts", CASES, ids=[c[0] for c in CASES])
@pytest.mark.parametrize("chunk", [1, 7, 64])
def test_paged_log_equals_dense_log_and_oracle(name, rhs, make, opts, chunk):
    y0, p, t0, t1 = make()
    n = y0.shape[0]
    dense = emul_batch(rhs, y0, p, t0, t1, max_log=2048, flavour_log_only=False, **opts)   # the whole device DefaultSolOut (flavour 1)
    assert int(dense["n_log"].max()) <= 2048
    lean = emul_batch(rhs, y0, p, t0, t1, max_log=2048, **opts)                               # the log-only flavour where it applies
    for k in ("t_log", "y_log", "y_end", "t_end", "h_next"):
        assert np.array_equal(_bits(lean[k]), _bits(dense[k])), k
    for k in ("n_log", "nfev", "naccpt", "nrejct", "status"):
        assert np.array_equal(lean[k], dense[k]), k
    paged = emul_batch(rhs, y0, p, t0, t1, paged_log=1 << 25, chunk=chunk, **opts)
    assert not paged["log_overflow"] and 0 < paged["log_used"] <= 1 << 25
    assert np.array_equal(paged["n_log"], dense["n_log"])
    for k in ("y_end", "t_end", "h_next"):
        assert np.array_equal(_bits(paged[k]), _bits(dense[k])), k
    off, t, y = E.gather_pages(paged, n)
    td, yd = _dense_as_csr(dense)
    assert np.array_equal(_bits(t), _bits(td)) and np.array_equal(_bits(y), _bits(yd))
    # ... and the reference's Solution.t / Solution.y as the oracle restates them, trajectory by trajectory
    t0a, t1a = np.broadcast_to(np.asarray(t0, dtype=np.float64), (y0.shape[1],)), np.broadcast_to(np.asarray(t1, dtype=np.float64), (y0.shape[1],))
    for b in range(y0.shape[1]):
        ref = O.solve_ivp(rhs, float(t0a[b]), float(t1a[b]), y0[:, b], params=() if p is None else p[:, b], detpow=True, **opts)
        assert off[b + 1] - off[b] == len(ref.t)
        assert np.array_equal(_bits(t[off[b]:off[b + 1]]), _bits(ref.t))
        assert np.array_equal(_bits(y[off[b]:off[b + 1]]), _bits(ref.y))


def test_terminal_event_record_goes_to_the_pages():
    """A terminal event appends its point to Solution.t / Solution.y (solout.rs:316-319): through the paged form too."""
    y0 = np.array([[1.0, 0.3], [0.0, 1.0]])
    kw = dict(method="DOPRI5", rtol=1e-8, atol=1e-10, event_direction=[0], event_terminal=[2])
    dense = emul_batch("sho_ev", y0, None, 0.0, 20.0, max_log=512, **kw)
    paged = emul_batch("sho_ev", y0, None, 0.0, 20.0, paged_log=1 << 22, **kw)
    assert (dense["status"] == 1).all() and np.array_equal(paged["n_log"], dense["n_log"]) and not paged["log_overflow"]
    _, t, y = E.gather_pages(paged, 2)
    td, yd = _dense_as_csr(dense)
    assert np.array_equal(_bits(t), _bits(td)) and np.array_equal(_bits(y), _bits(yd))


def test_a_pool_that_runs_dry_keeps_counting():
    """Pool exhaustion is not an error of the integration: the end states and the counts stay exact (the host then runs
    the counted fill pass), the overflow is reported, and nothing is written outside the pool."""
    y0, p, t0, t1 = W.cr3bp_batch(16)
    opts = dict(method="DOPRI5", rtol=1e-6, atol=1e-9)
    dense = emul_batch("cr3bp", y0, p, t0, t1, max_log=2048, **opts)
    paged = emul_batch("cr3bp", y0, p, t0, t1, paged_log=64 * 700, **opts)       # 700 doubles per sub-pool: not even one page of a launch
    assert paged["log_overflow"]                            # only 16 of the 64 sub-pools are used here, each far beyond its 700 doubles
    assert np.array_equal(paged["n_log"], dense["n_log"])
    assert np.array_equal(_bits(paged["y_end"]), _bits(dense["y_end"]))
    guard = paged["log_pool"][64 * 700:]                   # behind the pool the bodies were told about
    assert guard.size == 256 and np.isnan(guard).all()


def test_zero_length_interval_and_nan_interval_lanes():
    """solve_ivp.rs:110-145: a zero-length interval records its single point; a NaN interval records nothing."""
    y0 = np.array([[1.0, 2.0, 3.0], [0.0, 0.0, 0.0]])
    t0 = np.array([0.0, 1.0, 0.0])
    t1 = np.array([0.0, 2.0, np.nan])
    paged = emul_batch("sho", y0, None, t0, t1, method="DOPRI5", rtol=1e-6, atol=1e-9, paged_log=1 << 22)
    assert not paged["log_overflow"]
    dense = emul_batch("sho", y0, None, t0, t1, method="DOPRI5", rtol=1e-6, atol=1e-9, max_log=64)
    assert np.array_equal(paged["n_log"], dense["n_log"]) and paged["n_log"][0] == 1 and paged["n_log"][2] == 0
    off, t, y = E.gather_pages(paged, 2)
    td, yd = _dense_as_csr(dense)
    assert np.array_equal(_bits(t), _bits(td)) and np.array_equal(_bits(y), _bits(yd))
    assert paged["n_log"][2] == 0


@pytest.mark.parametrize("rhs,make,rtol", [("vdp", lambda: W.vdp_batch(30), 1e-8), ("cr3bp", lambda: W.cr3bp_batch(10), 1e-9)])
@pytest.mark.parametrize("chunk", [1, 64])
def test_deferred_t_eval_sampling_equals_sampling_in_the_stepping_bodies(rhs, make, rtol, chunk):
    """Kernel flavour 3 (DOP853 + t_eval): the stepping bodies only note the sampled steps, dop853_sample_body redoes each
    noted step from (x, h, y, k1) and evaluates dense stages + samples.  Same expressions, same order: the samples are the
    bits the in-line sampling (flavour 1) and the oracle produce -- unsorted grids, repeated points, points outside the span
    and backward integration included."""
    y0, p, t0, t1 = make()
    t1s = float(np.max(t1))
    te = np.concatenate([np.linspace(0.0, t1s, 41), [0.3 * t1s, 0.3 * t1s, -1.0, 2.0 * t1s, 0.05 * t1s]])
    kw = dict(method="DOP853", rtol=rtol, atol=1e-11, t_eval=te, chunk=chunk)
    inline = emul_batch(rhs, y0, p, t0, t1, defer_eval=False, **kw)
    deferred = emul_batch(rhs, y0, p, t0, t1, **kw)
    assert "def_rec" in deferred and "def_rec" not in inline
    for k in ("n_filled", "eval_idx", "status", "nfev", "naccpt"):
        assert np.array_equal(deferred[k], inline[k]), k
    for k in ("y_eval", "y_end", "t_end", "h_next"):
        assert np.array_equal(_bits(deferred[k]), _bits(inline[k])), k
    assert int(deferred["n_filled"].min()) > 10
    # backward in time, a grid running backward
    back = dict(method="DOP853", rtol=rtol, atol=1e-11, t_eval=np.linspace(t1s, 0.0, 23)[1:], chunk=chunk)
    a = emul_batch(rhs, y0, p, t1s, 0.0, defer_eval=False, **back)
    b = emul_batch(rhs, y0, p, t1s, 0.0, **back)
    assert np.array_equal(a["n_filled"], b["n_filled"]) and np.array_equal(_bits(a["y_eval"]), _bits(b["y_eval"])) and int(b["n_filled"].max()) == 22
