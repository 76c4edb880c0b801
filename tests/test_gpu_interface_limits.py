"""Interface limits the reference does not have, lifted (GPU):

  * every solve_ivp() call has its OWN Options.t_eval (src/solve/options.rs:75-123): `Options.t_eval_per_trajectory`
    (C ABI: ivp_options_t.t_eval_offsets) gives each trajectory of a batch its own ragged grid; the samples come back as
    time-major CSR records;
  * trait IVP::n_events is unbounded (src/ivp.rs:31-52): more than four event functions (configurations travel as device
    arrays: ivp_options_t.ev_direction_vec / ev_terminal_vec);
  * a `jac` override that fills only its non-zero entries (the reference hands f.jac() a zero-initialised Matrix,
    bdf.rs:152) -- the advisor's round-2 finding.
"""
import numpy as np
import pytest

import ivp_amd
from ivp_amd import workloads as W
from oracle import oracle as O

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("method,fast", [("DOPRI5", False), ("DOP853", False), ("RK23", True), ("BDF", False)])
def test_per_trajectory_t_eval_grids_match_one_oracle_call_per_trajectory(method, fast):
    import torch
    B = 37
    rng = np.random.default_rng(21)
    y0 = np.stack([np.cos(rng.uniform(0, 1, B)), np.sin(rng.uniform(0, 1, B))])
    t1 = rng.uniform(1.0, 4.0, B)
    grids = []
    for b in range(B):
        k = int(rng.integers(0, 9))                                   # ragged, some empty
        g = np.sort(rng.uniform(-0.2, t1[b] + 0.2, k))                # points outside the span are skipped like in the reference
        if b % 5 == 0 and k:
            g[0] = 0.0                                                # the start point itself
        grids.append(g)
    dev = torch.device("cuda:0")
    o = ivp_amd.Options(method=method, rtol=1e-6, atol=1e-9, t_eval_per_trajectory=grids,
                        fp_mode=ivp_amd.FpMode.FMA if fast else ivp_amd.FpMode.STRICT, chunk_attempts=7)
    r = ivp_amd.solve_ivp_batch(ivp_amd.SHO(), 0.0, torch.as_tensor(t1, device=dev), torch.as_tensor(y0, device=dev), None, o)
    assert (r.status.cpu().numpy() == 0).all()
    for b in range(B):
        s = O.solve_ivp("sho", 0.0, float(t1[b]), y0[:, b], detpow=True, fma=fast, method=method, rtol=1e-6, atol=1e-9, t_eval=grids[b])
        idx, y = r.eval_of(b)
        idx, y = idx.cpu().numpy(), y.cpu().numpy()
        assert len(idx) == len(s.t), (b, len(idx), len(s.t))
        assert np.array_equal(grids[b][idx], s.t) and np.array_equal(y, s.y), b
    # the end states are those of the plain solve
    plain = ivp_amd.solve_ivp_batch(ivp_amd.SHO(), 0.0, torch.as_tensor(t1, device=dev), torch.as_tensor(y0, device=dev), None,
                                    ivp_amd.Options(method=method, rtol=1e-6, atol=1e-9, fp_mode=o.fp_mode))
    assert torch.equal(plain.y_end, r.y_end) and torch.equal(plain.naccpt, r.naccpt)


SIX_EVENTS_SRC = r"""
__device__ void ode(double t, const double* y, double* d, const double* p) { d[0] = y[1]; d[1] = -y[0]; }
__device__ void events(double t, const double* y, double* g, const double* p)
{
    g[0] = y[0]; g[1] = y[1]; g[2] = y[0] - 0.5; g[3] = y[1] + 0.5; g[4] = y[0] + y[1]; g[5] = t - p[0];
}
"""


def test_six_event_functions_with_a_terminal_one_match_the_oracle():
    from ivp_amd import Direction, EventConfig
    cfgs = [EventConfig(), EventConfig(Direction.Positive), EventConfig(Direction.Negative), EventConfig(),
            EventConfig(Direction.Positive), EventConfig(Direction.All, 1)]             # the sixth is terminal (t = p0)
    f = ivp_amd.DeviceIVP(SIX_EVENTS_SRC, n=2, params=(5.5,), events=cfgs)
    assert f.n_events() == 6
    s = ivp_amd.solve_ivp(f, 0.0, 10.0, [1.0, 0.0], ivp_amd.Options(method="DOPRI5", rtol=1e-8, atol=1e-10))
    ode = lambda t, y, p: [y[1], -y[0]]
    ev = lambda t, y, p: [y[0], y[1], y[0] - 0.5, y[1] + 0.5, y[0] + y[1], t - p[0]]
    o = O.solve_ivp(ode, 0.0, 10.0, [1.0, 0.0], params=(5.5,), events=ev, n_events=6, event_direction=[0, 1, -1, 0, 1, 0],
                    event_terminal=[0, 0, 0, 0, 0, 1], method="DOPRI5", rtol=1e-8, atol=1e-10, detpow=True)
    assert int(s.status) == o.status == 1
    for i in range(6):
        assert np.array_equal(np.asarray(s.t_events[i]), np.asarray(o.t_events[i])), i
        assert np.array_equal(np.asarray(s.y_events[i]).reshape(-1), np.asarray(o.y_events[i]).reshape(-1)), i
    assert np.array_equal(s.t, o.t) and np.array_equal(s.y, o.y)
    np.testing.assert_allclose(s.t_events[0], [np.pi / 2, 3 * np.pi / 2], rtol=1e-7)
    np.testing.assert_allclose(s.t_events[5], [5.5], rtol=1e-9)


SPARSE_JAC_SRC = r"""
__device__ void ode(double t, const double* s, double* d, const double* p)
{
    const double x = s[0], y = s[1], z = s[2];
    d[0] = -0.04 * x + 1e4 * y * z;
    d[1] = 0.04 * x - 1e4 * y * z - 3e7 * y * y;
    d[2] = 3e7 * y * y;
}
// fills ONLY the structurally non-zero entries of the Robertson Jacobian: j[2][0] and j[2][2] are never written
__device__ void jac(double t, const double* s, double* j, const double* p)
{
    const double y = s[1], z = s[2];
    j[0] = -0.04; j[1] = 1e4 * z;             j[2] = 1e4 * y;
    j[3] = 0.04;  j[4] = -1e4 * z - 6e7 * y;  j[5] = -1e4 * y;
                  j[7] = 6e7 * y;
}
"""


def test_jac_override_that_fills_only_its_nonzero_entries():
    """bdf.rs:152 hands f.jac() a zero-initialised persistent Matrix, so entries the override never writes are 0."""
    f = ivp_amd.DeviceIVP(SPARSE_JAC_SRC, n=3, jac=True)
    o = ivp_amd.Options(method="BDF", rtol=1e-6, atol=1e-6)
    s = ivp_amd.solve_ivp(f, 0.0, 1e8, [1e4, 0.0, 0.0], o)
    ref = O.solve_ivp("robertson_jac", 0.0, 1e8, [1e4, 0.0, 0.0], method="BDF", rtol=1e-6, atol=1e-6, detpow=True)
    assert int(s.status) == 0 and s.njev == ref.njev and s.nfev == ref.nfev and s.nlu == ref.nlu
    assert np.array_equal(s.y[-1], ref.y[-1]) and np.array_equal(s.t, ref.t)


LARGE_JAC_SRC = r"""
// 24 coupled linear reactions: y_i' = -k_i y_i + 0.5 k_{i-1} y_{i-1} + 0.25 k_{i+1} y_{i+1}, k_i = s (1 + i)
__device__ double ode_comp(int i, double t, const double* y, const double* p)
{
    double d = -p[0] * (1.0 + i) * y[i];
    if (i > 0) d += 0.5 * p[0] * (double)i * y[i - 1];
    if (i < 23) d += 0.25 * p[0] * (2.0 + i) * y[i + 1];
    return d;
}
// column form of the analytic Jacobian: only the three structurally non-zero entries of a column are written
__device__ void jac_col(int col, double t, const double* y, double* column, const double* p)
{
    column[col] = -p[0] * (1.0 + col);
    if (col + 1 < 24) column[col + 1] = 0.5 * p[0] * (double)(col + 1);
    if (col > 0) column[col - 1] = 0.25 * p[0] * (2.0 + (col - 1));
}
"""


def test_column_form_jac_override_on_the_wave_per_trajectory_path():
    """`impl IVP { fn jac }` for n > 8 (src/ivp.rs:67-107): jac_col(col, ...) replaces the n + 1 right-hand-side
    evaluations of the forward-difference default; same bits as the oracle run with the same analytic Jacobian, and
    njev counts it like the default."""
    K = 24
    f = ivp_amd.DeviceIVP(LARGE_JAC_SRC, n=K, params=(40.0,), jac=True)
    rng = np.random.default_rng(6)
    y0 = 1.0 + 0.2 * rng.standard_normal((K, 5))
    scale = np.full((1, 5), 40.0) * (1.0 + 0.1 * np.arange(5))[None, :]
    o = ivp_amd.Options(method="BDF", rtol=1e-6, atol=1e-9)
    r = ivp_amd.solve_ivp_batch(f, 0.0, 0.5, y0, scale, o)

    def fun(t, y, p):
        d = [-p[0] * (1.0 + i) * y[i] for i in range(K)]
        for i in range(K):
            if i > 0:
                d[i] += 0.5 * p[0] * float(i) * y[i - 1]
            if i < K - 1:
                d[i] += 0.25 * p[0] * (2.0 + i) * y[i + 1]
        return d

    def jac(t, y, p):
        j = [[0.0] * K for _ in range(K)]
        for c in range(K):
            j[c][c] = -p[0] * (1.0 + c)
            if c + 1 < K:
                j[c + 1][c] = 0.5 * p[0] * float(c + 1)
            if c > 0:
                j[c - 1][c] = 0.25 * p[0] * (2.0 + (c - 1))
        return j

    for b in range(5):
        s = O.solve_ivp(fun, 0.0, 0.5, list(y0[:, b]), params=[float(scale[0, b])], jac=jac, method="BDF", rtol=1e-6, atol=1e-9, detpow=True)
        assert int(r.status[b]) == s.status == 0
        assert np.array_equal(r.y_end[:, b], s.y[-1]), b
        assert int(r.njev[b]) == s.njev and int(r.nlu[b]) == s.nlu and int(r.nfev[b]) == s.nfev
    # the forward-difference default (n + 1 right-hand-side evaluations per Jacobian, not counted in nfev: src/ivp.rs:67-107)
    # integrates the same system to the same answer within the tolerance
    fd = ivp_amd.solve_ivp_batch(ivp_amd.DeviceIVP(LARGE_JAC_SRC.split("// column form")[0], n=K, params=(40.0,)), 0.0, 0.5, y0, scale, o)
    np.testing.assert_allclose(fd.y_end, r.y_end, rtol=1e-4, atol=1e-8)


def test_per_trajectory_grids_through_the_host_pointer_and_the_multi_context_entry_points():
    """ivp_options_t.t_eval_offsets on ivp_batch_solve (host arrays in, host arrays out) and on ivp_batch_solve_multi (here:
    three contexts on one GPU, uneven shards): the library re-bases the offsets per shard and places every shard's CSR
    sample records in the batch-wide arrays -- same bits as the single-context device entry point."""
    import torch
    B = 41
    rng = np.random.default_rng(33)
    y0 = np.stack([np.cos(rng.uniform(0, 1, B)), np.sin(rng.uniform(0, 1, B))])
    t1 = rng.uniform(1.0, 4.0, B)
    grids = [np.sort(rng.uniform(-0.1, t1[b] + 0.1, int(rng.integers(0, 7)))) for b in range(B)]
    o = ivp_amd.Options(method="DOP853", rtol=1e-7, atol=1e-10, t_eval_per_trajectory=grids)
    dev = torch.device("cuda:0")
    ref = ivp_amd.solve_ivp_batch(ivp_amd.SHO(), 0.0, torch.as_tensor(t1, device=dev), torch.as_tensor(y0, device=dev), None, o)
    host = ivp_amd.solve_ivp_batch(ivp_amd.SHO(), 0.0, t1, y0, None, o)                       # numpy in, numpy out
    from ivp_amd.distributed import solve_ivp_batch_multi
    multi = solve_ivp_batch_multi(ivp_amd.SHO(), 0.0, t1, y0, None, o, devices=[0, 0, 0])
    total = int(ref.eval_offsets[-1])
    assert total == sum(len(g) for g in grids) and total > 0
    for other, to_np in ((host, lambda a: np.asarray(a)), (multi, lambda a: a.cpu().numpy())):
        assert np.array_equal(to_np(other.eval_offsets), ref.eval_offsets.cpu().numpy())
        assert np.array_equal(to_np(other.n_filled), ref.n_filled.cpu().numpy())
        assert np.array_equal(to_np(other.y_end).view(np.uint64), ref.y_end.cpu().numpy().view(np.uint64))
        for b in range(B):
            (ia, ya), (ib, yb) = other.eval_of(b), ref.eval_of(b)
            assert np.array_equal(to_np(ia), ib.cpu().numpy()) and np.array_equal(to_np(ya).view(np.uint64), yb.cpu().numpy().view(np.uint64)), b


def test_csr_step_log_through_the_host_pointer_entry_point():
    """Solution.t / Solution.y of every trajectory (every accepted step, src/solve/solout.rs:387-428) with host arrays in and
    out: a counting call (Options.count_log), the caller's exclusive scan, and the filling call with out.log_offsets /
    t_log / y_log as HOST arrays -- record for record what the device-pointer path (solve_ivp_batch_logged) returns."""
    B = 23
    y0, p, t0, t1 = W.cr3bp_batch(B)
    t1 = 3.0
    base = dict(method="DOPRI5", rtol=1e-7, atol=1e-10)
    f = ivp_amd.CR3BP()
    ref = ivp_amd.solve_ivp_batch_logged(f, t0, t1, y0, p, ivp_amd.Options(**base))
    cnt = ivp_amd.solve_ivp_batch(f, t0, t1, y0, p, ivp_amd.Options(**base, count_log=True))
    off = np.zeros(B + 1, dtype=np.uint64)
    off[1:] = np.cumsum(cnt.n_log.astype(np.uint64))
    total = int(off[-1])
    assert total == int(ref.log_offsets[-1]) and np.array_equal(off.astype(np.int64), ref.log_offsets.cpu().numpy())
    out = ivp_amd.BatchSolution(y_end=cnt.y_end, t_end=cnt.t_end, status=cnt.status, nfev=cnt.nfev, nstep=cnt.nstep, naccpt=cnt.naccpt,
                                nrejct=cnt.nrejct, h_next=cnt.h_next, njev=cnt.njev, nlu=cnt.nlu, n_log=cnt.n_log,
                                t_log=np.full(total, np.nan), y_log=np.full((total, 6), np.nan), log_offsets=off)
    res = ivp_amd.solve_ivp_batch(f, t0, t1, y0, p, ivp_amd.Options(**base), out=out)
    assert np.array_equal(res.t_log.view(np.uint64), ref.t_log.cpu().numpy().view(np.uint64))
    assert np.array_equal(res.y_log.view(np.uint64), ref.y_log.cpu().numpy().view(np.uint64))
    assert np.array_equal(res.y_end.view(np.uint64), ref.y_end.cpu().numpy().view(np.uint64))
    t, y = res.log_of(5)
    assert t[0] == t0 and t[-1] == t1 and y.shape == (len(t), 6)
