"""Lane-cooperative DOPRI5 kernel (rk_coop.h: eight lanes per trajectory, used for the latency-bound tail of a batch).

In strict mode it must be bit-identical to the oracle -- and hence to the thread-per-trajectory kernels -- whether it
runs the whole integration (variant = 3) or takes over at a launch boundary (variant = 0, the default policy)."""
import numpy as np
import pytest

import ivp_amd
from ivp_amd import workloads as W
from tests.cases import CASES
from tests.common import assert_bitexact, gpu_batch, oracle_batch

pytestmark = pytest.mark.gpu

COOP_CASES = [c for c in CASES if "DOPRI5" in c[0].upper() or "DOP853" in c[0].upper() or c[0] in ("C1-decay", "exp2-vector-rtol")]


@pytest.mark.parametrize("case", COOP_CASES, ids=[c[0] for c in COOP_CASES])
def test_coop_everywhere_bitexact_vs_oracle(case):
    name, rhs, build = case
    y0, p, t0, t1, o = build()
    ref = oracle_batch(rhs, y0, p, t0, t1, **o)
    for chunk in (0, 17):
        got = gpu_batch(rhs, y0, p, t0, t1, variant=3, chunk=chunk, **o)
        assert_bitexact(got, ref, f"{name} coop chunk={chunk}: ")
        if "t_eval" in o:
            assert np.array_equal(got["n_filled"], ref["n_filled"])
            for b in range(y0.shape[1]):
                m = int(ref["n_filled"][b])
                assert np.array_equal(got["y_eval"][:m, :, b], ref["y_eval"][:m, :, b])


@pytest.mark.parametrize("method", ["DOPRI5", "DOP853"])
def test_coop_full_outputs_match_thread_per_trajectory_kernels(method):
    """t_eval samples, step log and dense segments written by the cooperative kernels (variant 3) are the records the
    thread-per-trajectory kernels (variant 1) write, bit for bit."""
    y0, p, t0, t1 = W.vdp_batch(300)
    tol = dict(DOPRI5=(1e-6, 1e-9), DOP853=(1e-8, 1e-10))[method]
    te = np.linspace(0.0, 50.0, 41)
    a = gpu_batch("vdp", y0, p, t0, t1, method=method, rtol=tol[0], atol=tol[1], t_eval=te, variant=1)
    b = gpu_batch("vdp", y0, p, t0, t1, method=method, rtol=tol[0], atol=tol[1], t_eval=te, variant=3)
    assert_bitexact(b, a, "t_eval ")
    assert np.array_equal(a["n_filled"], b["n_filled"]) and np.array_equal(a["y_eval"], b["y_eval"]) and np.array_equal(a["eval_idx"], b["eval_idx"])
    a = gpu_batch("vdp", y0[:, :40], p[:, :40], t0, t1[:40], method=method, rtol=tol[0], atol=tol[1], max_log=900, dense_output=True, variant=1)
    b = gpu_batch("vdp", y0[:, :40], p[:, :40], t0, t1[:40], method=method, rtol=tol[0], atol=tol[1], max_log=900, dense_output=True, variant=3)
    assert_bitexact(b, a, "log ")
    for k in ("n_log", "t_log", "y_log", "n_seg", "seg_xold", "seg_h", "seg_cont"):
        assert np.array_equal(a[k], b[k]), k


def test_c2_tail_switch_is_invisible():
    """20k of the C2 trajectories: the default policy starts on the thread-per-trajectory kernels and hands the
    stragglers to the cooperative kernel; forcing either kind for the whole run gives the same bits."""
    y0, p, t0, t1 = W.cr3bp_batch(100000)          # the BASELINE C2 batch (slowest trajectory: 702 attempts) ...
    y0, p = np.ascontiguousarray(y0[:, 40000:60000]), np.ascontiguousarray(p[:, 40000:60000])   # ... a 20k slice of it
    o = dict(method="DOPRI5", rtol=1e-6, atol=1e-9)
    auto = gpu_batch("cr3bp", y0, p, t0, t1, device_arrays=True, **o)
    lean = gpu_batch("cr3bp", y0, p, t0, t1, device_arrays=True, variant=1, **o)
    coop = gpu_batch("cr3bp", y0, p, t0, t1, device_arrays=True, variant=3, **o)
    assert_bitexact(auto, lean, "auto vs lean ")
    assert_bitexact(coop, lean, "coop vs lean ")
    ref = oracle_batch("cr3bp", y0[:, :2048], p[:, :2048], t0, t1, **o)
    sub = {k: (v[..., :2048] if isinstance(v, np.ndarray) else v) for k, v in auto.items()}
    assert_bitexact(sub, ref, "auto vs oracle ")


def test_coop_handles_ragged_groups_and_all_rhs():
    """Batch sizes that leave a partly filled 8-lane group / wave, for every built-in problem without events."""
    rng = np.random.default_rng(4)
    for rhs, n, npar, pv in (("decay", 1, 1, [0.7]), ("sho", 2, 0, None), ("vdp", 2, 1, [1.5]), ("lorenz", 3, 3, [10.0, 28.0, 8 / 3]),
                             ("linear", 2, 0, None), ("robertson", 3, 0, None), ("exp2", 2, 0, None)):
        for B in (1, 7, 9, 65):
            y0 = 1.0 + 0.1 * rng.standard_normal((n, B))
            p = None if not npar else np.repeat(np.asarray(pv)[:, None], B, axis=1)
            t1 = rng.uniform(0.1, 1.0, B)
            o = dict(method="DOPRI5", rtol=1e-7, atol=1e-9)
            ref = oracle_batch(rhs, y0, p, 0.0, t1, **o)
            got = gpu_batch(rhs, y0, p, 0.0, t1, variant=3, **o)
            assert_bitexact(got, ref, f"{rhs} B={B}: ")


from tests.cases import EVENT_CASES, check_events_against_oracle  # noqa: E402

COOP_EVENT_CASES = [c for c in EVENT_CASES if c[0].endswith("DOPRI5") or c[0].endswith("DOP853")]


@pytest.mark.parametrize("case", COOP_EVENT_CASES, ids=[c[0] for c in COOP_EVENT_CASES])
def test_coop_event_detection_matches_oracle(case):
    """Event problems through the cooperative kernels (variant 3): the event functions are evaluated on the gathered
    state by every lane of a group; detections, terminal handling and outputs are the oracle's."""
    exact = not case[1].startswith("rational")
    check_events_against_oracle(lambda rhs, y0, p, t0, t1, **kw: gpu_batch(rhs, y0, p, t0, t1, chunk=5, variant=3, **kw), case, exact=exact)
    stats_probe = gpu_batch(case[1], np.asarray(case[4], float).reshape(-1, 1), None if not len(case[5]) else np.asarray(case[5], float).reshape(-1, 1),
                            case[2], case[3], variant=3, profile=1, max_log=64, **{k: v for k, v in case[6].items() if k != "t_eval"})
    assert stats_probe["stats"]["coop_launches"] == stats_probe["stats"]["launches"] > 0


def test_speculative_hand_over_declines_then_accepts():
    """Tight tolerances: after each bulk round the speculative cooperative launch finds the active set still too large
    and does nothing (several times) before it finally takes over.  Results equal the lean-only run bit for bit."""
    y0, p, t0, t1 = W.cr3bp_batch(100000)      # > 65536 trajectories: the first rounds are bulk rounds
    o = dict(method="DOPRI5", rtol=1e-9, atol=1e-12)
    auto = gpu_batch("cr3bp", y0, p, t0, t1, device_arrays=True, profile=1, **o)
    lean = gpu_batch("cr3bp", y0, p, t0, t1, device_arrays=True, variant=1, **o)
    assert_bitexact(auto, lean, "auto vs lean ")
    st = auto["stats"]
    assert st["coop_launches"] >= 1 and st["launches"] - st["coop_launches"] >= 6     # at least two bulk rounds => a declined launch
    ref = oracle_batch("cr3bp", y0[:, :512], p[:, :512], t0, t1, **o)
    sub = {k: (v[..., :512] if isinstance(v, np.ndarray) else v) for k, v in auto.items()}
    assert_bitexact(sub, ref, "auto vs oracle ")
    # the DOP853 instantiation of the same mechanism
    o = dict(method="DOP853", rtol=1e-8, atol=1e-10)
    auto = gpu_batch("cr3bp", y0, p, t0, t1, device_arrays=True, profile=1, **o)
    lean = gpu_batch("cr3bp", y0, p, t0, t1, device_arrays=True, variant=1, **o)
    assert_bitexact(auto, lean, "DOP853 auto vs lean ")
    assert auto["stats"]["coop_launches"] >= 1
