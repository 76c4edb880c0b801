"""Unbounded outputs: the reference's Solution.t / .y / .t_events / .y_events are Vecs that grow with every accepted
step and every event occurrence (/root/reference/src/solve/solout.rs:158-331 events, :387-428 step records).  The GPU
path writes into caller-sized buffers; these tests pin that nothing is silently truncated:

  * solve_ivp (one trajectory) reruns with larger buffers when the step log OR an event buffer overflowed;
  * solve_ivp_batch reports an event overflow (flag + RuntimeWarning) and n_event_hits keeps the true count;
  * solve_ivp_batch_logged returns every accepted step of every trajectory in CSR form (two passes: count, fill):
    memory is sum(n_log) records, not max_log x B.
"""
import warnings

import numpy as np
import pytest

import ivp_amd
from ivp_amd import workloads as W
from oracle import oracle as O

pytestmark = pytest.mark.gpu


def _sho_many_crossings():
    # tests/ivp.rs:151-221 problem (SHO, event y0 = 0, Direction::All) over 60 periods: 120 zero crossings > 64
    return ivp_amd.SHOZeroEvent(ivp_amd.EventConfig()), 0.0, 60 * 2 * np.pi, [1.0, 0.0]


def test_single_solve_reruns_until_every_event_fits():
    f, t0, t1, y0 = _sho_many_crossings()
    opts = dict(method="DOPRI5", rtol=1e-8, atol=1e-10)
    s = ivp_amd.solve_ivp(f, t0, t1, y0, ivp_amd.Options(max_events=8, **opts))      # 8 << 120: forces several reruns
    o = O.solve_ivp("sho_ev", t0, t1, y0, detpow=True, event_direction=[0], event_terminal=[0], **opts)
    assert len(o.t_events[0]) == 120
    assert len(s.t_events[0]) == 120 and np.array_equal(s.t_events[0], o.t_events[0]) and np.array_equal(s.y_events[0], o.y_events[0])
    assert np.array_equal(s.t, o.t) and np.array_equal(s.y, o.y) and s.nfev == o.nfev
    # the default capacity (64) is also exceeded: same result
    s2 = ivp_amd.solve_ivp(f, t0, t1, y0, ivp_amd.Options(**opts))
    assert np.array_equal(s2.t_events[0], o.t_events[0])


def test_terminal_event_with_minimal_capacity():
    """examples/bouncing_ball.rs:5-31: the reference's ball does not bounce -- its ground event is terminal -- so the
    > 64-occurrence case above is the SHO; here the event buffers are as small as they can be (one slot)."""
    f = ivp_amd.BouncingBall(9.81, 0.0, ivp_amd.EventConfig(ivp_amd.Direction.Negative, 1))
    s = ivp_amd.solve_ivp(f, 0.0, 10.0, [10.0, 5.0], ivp_amd.Options(method="DOPRI5", rtol=1e-8, atol=1e-10, max_events=1))
    o = O.solve_ivp("ball", 0.0, 10.0, [10.0, 5.0], params=(9.81, 0.0), detpow=True, method="DOPRI5", rtol=1e-8, atol=1e-10,
                    event_direction=[-1], event_terminal=[1])
    assert s.status == ivp_amd.Status.UserInterrupt and np.array_equal(s.t_events[0], o.t_events[0]) and np.array_equal(s.t, o.t)


def test_batch_event_overflow_is_reported_not_hidden():
    f, t0, t1, y0 = _sho_many_crossings()
    y0b = np.asarray(y0).reshape(2, 1).repeat(3, axis=1)
    opts = dict(method="DOPRI5", rtol=1e-8, atol=1e-10)
    with pytest.warns(RuntimeWarning, match="event buffers overflowed"):
        r = ivp_amd.solve_ivp_batch(f, t0, t1, y0b, None, ivp_amd.Options(max_events=16, max_log=4096, **opts))
    assert r.event_overflow and int(r.n_event_hits.max()) == 120 and r.t_events.shape[1] == 16
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        r2 = ivp_amd.solve_ivp_batch(f, t0, t1, y0b, None, ivp_amd.Options(max_events=128, max_log=4096, **opts))
    assert not r2.event_overflow and np.array_equal(r2.t_events[0, :16, :], r.t_events[0, :, :])


def test_csr_step_log_equals_the_dense_log_and_the_oracle():
    import torch
    B = 3000
    y0, p, t0, t1 = W.cr3bp_batch(B)
    opt = dict(method="DOPRI5", rtol=1e-6, atol=1e-9)
    dev = torch.device("cuda:0")
    y0d, pd = torch.as_tensor(y0, device=dev), torch.as_tensor(p, device=dev)
    dense = ivp_amd.solve_ivp_batch(ivp_amd.CR3BP(), t0, t1, y0d, pd, ivp_amd.Options(max_log=640, **opt))
    csr = ivp_amd.solve_ivp_batch_logged(ivp_amd.CR3BP(), t0, t1, y0d, pd, ivp_amd.Options(**opt))
    assert torch.equal(csr.n_log, dense.n_log) and int(dense.n_log.max()) <= 640
    assert torch.equal(csr.y_end, dense.y_end) and torch.equal(csr.naccpt, dense.naccpt)
    off = csr.log_offsets.cpu().numpy()
    assert off[0] == 0 and np.array_equal(np.diff(off), dense.n_log.cpu().numpy()) and csr.t_log.shape[0] == off[-1]
    tl, yl = dense.t_log.cpu().numpy(), dense.y_log.cpu().numpy()
    ct, cy = csr.t_log.cpu().numpy(), csr.y_log.cpu().numpy()
    for b in list(range(0, B, 97)) + [B - 1]:
        m = off[b + 1] - off[b]
        assert np.array_equal(ct[off[b]:off[b + 1]], tl[:m, b]) and np.array_equal(cy[off[b]:off[b + 1]], yl[:m, :, b].reshape(m, 6)), b
    for b in (0, 1234, B - 1):        # and the reference's Solution.t / Solution.y for that solve_ivp() call
        s = O.solve_ivp("cr3bp", t0, t1, y0[:, b], params=p[:, b], detpow=True, **opt)
        t, y = csr.log_of(b)
        assert np.array_equal(t.cpu().numpy(), s.t) and np.array_equal(y.cpu().numpy(), s.y)
    # numpy inputs, first_step output enforcement and backward integration go through the same two passes
    csr2 = ivp_amd.solve_ivp_batch_logged(ivp_amd.SHO(), 2 * np.pi, 0.0, np.array([[1.0, 0.5], [0.0, 0.2]]), None,
                                          ivp_amd.Options(method="RK23", rtol=1e-6, atol=1e-9, first_step=0.1))
    for b in range(2):
        s = O.solve_ivp("sho", 2 * np.pi, 0.0, [[1.0, 0.0], [0.5, 0.2]][b], detpow=True, method="RK23", rtol=1e-6, atol=1e-9, first_step=0.1)
        t, y = csr2.log_of(b)
        assert np.array_equal(t.cpu().numpy(), s.t) and np.array_equal(y.cpu().numpy(), s.y)


def test_c2_full_step_log_fits_in_sum_of_records():
    """BASELINE C2 (100k CR3BP): the full Solution.t / Solution.y of every trajectory takes sum(naccpt + 1) records of
    (n + 1) * 8 = 56 bytes -- 0.95 GB -- where the dense layout needs (longest log) x B records, about 3x as much."""
    import torch
    B = 100_000
    y0, p, t0, t1 = W.cr3bp_batch(B)
    dev = torch.device("cuda:0")
    r = ivp_amd.solve_ivp_batch_logged(ivp_amd.CR3BP(), t0, t1, torch.as_tensor(y0, device=dev), torch.as_tensor(p, device=dev),
                                       ivp_amd.Options(method="DOPRI5", rtol=1e-6, atol=1e-9))
    total = int(r.log_offsets[-1])
    assert total == int(r.n_log.sum())
    # one record per accepted step + the initial point, minus the few the 1e-12 dedupe drops (solout.rs:424)
    assert int(r.naccpt.sum()) + B - 1000 < total <= int(r.naccpt.sum()) + B
    assert r.t_log.numel() == total and r.y_log.shape == (total, 6)
    assert (r.t_log.numel() + r.y_log.numel()) * 8 == total * 56
    assert total * 56 < 0.5 * int(r.n_log.max()) * B * 56       # well below the dense max_log x B layout
    assert bool((r.status == 0).all())
    # every trajectory's log starts at t0 and ends at t1; records are strictly increasing in t
    off = r.log_offsets
    assert bool((r.t_log[off[:-1]] == t0).all()) and bool(((r.t_log[off[1:] - 1] - t1).abs() <= 1e-12).all())
    d = r.t_log[1:] - r.t_log[:-1]
    d[off[1:-1] - 1] = 1.0                                        # boundaries between trajectories
    assert bool((d > 0).all())
    for b in (0, 51234, B - 1):
        s = O.solve_ivp("cr3bp", t0, t1, y0[:, b], params=p[:, b], detpow=True, method="DOPRI5", rtol=1e-6, atol=1e-9)
        t, y = r.log_of(b)
        assert np.array_equal(t.cpu().numpy(), s.t) and np.array_equal(y.cpu().numpy(), s.y)


def test_csr_step_log_on_every_lane_mapping():
    """The CSR log goes through the same so_push_log in all three lane mappings: thread per trajectory (variant 1),
    eight lanes per trajectory (variant 3) and one wavefront per trajectory (n = 256)."""
    import torch
    y0, p, t0, t1 = W.cr3bp_batch(600)
    opt = dict(method="DOP853", rtol=1e-8, atol=1e-11)
    dev = torch.device("cuda:0")
    y0d, pd = torch.as_tensor(y0, device=dev), torch.as_tensor(p, device=dev)
    a = ivp_amd.solve_ivp_batch_logged(ivp_amd.CR3BP(), t0, 4.0, y0d, pd, ivp_amd.Options(variant=1, **opt))
    b = ivp_amd.solve_ivp_batch_logged(ivp_amd.CR3BP(), t0, 4.0, y0d, pd, ivp_amd.Options(variant=3, **opt))
    assert torch.equal(a.log_offsets, b.log_offsets) and torch.equal(a.t_log, b.t_log) and torch.equal(a.y_log, b.y_log)
    s = O.solve_ivp("cr3bp", t0, 4.0, y0[:, 7], params=p[:, 7], detpow=True, **opt)
    t, y = b.log_of(7)
    assert np.array_equal(t.cpu().numpy(), s.t) and np.array_equal(y.cpu().numpy(), s.y)
    rng = np.random.default_rng(11)
    x = np.arange(1, 257) / 257.0
    h0 = np.sin(np.pi * x[:, None] * np.array([1, 2, 3])[None, :]) + 0.05 * rng.standard_normal((256, 3))
    kappa = np.array([[30.0, 80.0, 200.0]])
    g = ivp_amd.solve_ivp_batch_logged(ivp_amd.Heat1D256(), 0.0, 0.2, h0, kappa, ivp_amd.Options(method="DOPRI5", rtol=1e-6, atol=1e-9))
    for bb in range(3):
        s = O.solve_ivp("heat1d256", 0.0, 0.2, h0[:, bb], params=kappa[:, bb], detpow=True, method="DOPRI5", rtol=1e-6, atol=1e-9)
        t, y = g.log_of(bb)
        assert np.array_equal(t.cpu().numpy(), s.t) and np.array_equal(y.cpu().numpy(), s.y), bb
