"""Multi-device drivers (reference partition spec: independent solve_ivp() calls,
/root/reference/src/solve/solve_ivp.rs:99-313).

  * ivp_batch_solve_multi (C ABI): N contexts driven by ONE host thread through submit/poll, shards gathered with
    peer copies.  A one-GPU box runs the degenerate case -- several contexts on device 0 -- which exercises the
    sharding, the concurrent state machines and the strided gather; the cross-device copy branch needs > 1 GPU.
  * the section-8e oracle for the collective: the peer-copy gather and the torch.distributed all-gather of the byte
    arena must deliver byte-identical buffers.
"""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

import ivp_amd
from ivp_amd import distributed as D
from ivp_amd import workloads as W

pytestmark = pytest.mark.gpu

FIELDS = ("y_end", "t_end", "h_next", "status", "nfev", "nstep", "naccpt", "nrejct")


def _single(y0, p, t0, t1, opt):
    import torch
    dev = torch.device("cuda:0")
    return ivp_amd.solve_ivp_batch(ivp_amd.CR3BP(), t0, t1, torch.as_tensor(y0, device=dev), torch.as_tensor(p, device=dev), opt)


@pytest.mark.parametrize("shards", [1, 2, 3])
def test_multi_context_solve_is_bit_identical_to_one_context(shards):
    import torch
    B = 4001                                        # odd: unequal shards
    y0, p, t0, t1 = W.cr3bp_batch(B, seed=77)
    perm = W.shard_permutation(B)
    opt = ivp_amd.Options(method="DOPRI5", rtol=1e-6, atol=1e-9)
    ref = _single(y0, p, t0, t1, opt)
    got = D.solve_ivp_batch_multi(ivp_amd.CR3BP(), t0, t1, y0, p, opt, devices=[0] * shards, permutation=perm)
    for k in FIELDS:
        assert torch.equal(getattr(got, k), getattr(ref, k)), k
    assert bool((got.status == 0).all())


def test_multi_context_per_trajectory_end_times_and_empty_shard():
    import torch
    y0, p, t0, t1 = W.vdp_batch(2, seed=5)          # two trajectories over three contexts: the last shard is empty
    opt = ivp_amd.Options(method="DOP853", rtol=1e-8, atol=1e-10)
    dev = torch.device("cuda:0")
    ref = ivp_amd.solve_ivp_batch(ivp_amd.VanDerPol(), t0, torch.as_tensor(t1, device=dev), torch.as_tensor(y0, device=dev),
                                  torch.as_tensor(p, device=dev), opt)
    got = D.solve_ivp_batch_multi(ivp_amd.VanDerPol(), t0, t1, y0, p, opt, devices=[0, 0, 0])
    for k in FIELDS:
        assert torch.equal(getattr(got, k), getattr(ref, k)), k


def test_multi_rejects_what_a_single_solve_rejects():
    y0, p, t0, t1 = W.cr3bp_batch(64)
    with pytest.raises(ivp_amd.ConfigError) as e:
        D.solve_ivp_batch_multi(ivp_amd.CR3BP(), t0, t1, y0, p, ivp_amd.Options(method="RADAU"), devices=[0, 0])
    assert e.value.code == -101
    ctx = ivp_amd.Context(0)
    with pytest.raises(ivp_amd.ConfigError):      # one context cannot run two shards at once
        D.solve_ivp_batch_multi(ivp_amd.CR3BP(), t0, t1, y0, p, ivp_amd.Options(), devices=[0, 0], contexts=[ctx, ctx])


_WORKER = r"""
import os, sys
sys.path.insert(0, {root!r})
import numpy as np, torch, torch.distributed as dist
from ivp_amd import workloads as W, distributed as D
import ivp_amd

rank = int(sys.argv[1])
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:{port}", rank=rank, world_size=2)
B = 4001
y0, p, t0, t1 = W.cr3bp_batch(B, seed=77)
perm = W.shard_permutation(B)
opt = ivp_amd.Options(method="DOPRI5", rtol=1e-6, atol=1e-9)
# collective path: one process per shard, byte arena, all-gather
coll = D.solve_ivp_sharded(ivp_amd.CR3BP(), t0, t1, y0, p, opt, permutation=perm, device="cuda:0")
if rank == 0:
    # peer-copy path: one process, two contexts, ivp_batch_solve_multi
    peer = D.solve_ivp_batch_multi(ivp_amd.CR3BP(), t0, t1, y0, p, opt, devices=[0, 0], permutation=perm)
    for k in ("y_end", "t_end", "h_next", "status", "nfev", "nstep", "naccpt", "nrejct"):
        a = np.ascontiguousarray(coll[k]); b = np.ascontiguousarray(getattr(peer, k).cpu().numpy())
        assert a.tobytes() == b.astype(a.dtype).tobytes(), k
dist.barrier()
dist.destroy_process_group()
print("rank", rank, "ok")
"""


def test_peer_copy_gather_and_collective_gather_are_byte_identical(tmp_path):
    """'hipMemcpyPeerAsync path must give byte-identical buffers' as the all-gather."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    script = tmp_path / "multi_worker.py"
    script.write_text(_WORKER.format(root=root, port=port))
    procs = [subprocess.Popen([sys.executable, str(script), str(r)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(2)]
    outs = [p.communicate(timeout=600)[0].decode() for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o[-2000:]


# ---- the whole Solution travels, not only the end states (BASELINE C4: "RCCL gather of sol.y") --------------------

@pytest.mark.parametrize("shards", [2, 3])
def test_multi_context_t_eval_samples_and_csr_log_match_one_context_and_the_oracle(shards):
    """solve_ivp_batch_multi with Options.t_eval gathers y_eval / eval_idx / n_filled; with log=True it gathers
    Solution.t / Solution.y of every trajectory as a CSR log (count pass, fill pass, offsets re-based per shard inside
    ivp_batch_solve_multi) -- all in the ORIGINAL trajectory order after the permutation, bit-identical to one context
    and to the oracle's records."""
    import torch
    from oracle import oracle as O
    B = 1001
    y0, p, t0, t1 = W.cr3bp_batch(B, seed=5)
    perm = W.shard_permutation(B)
    te = np.linspace(0.0, 2.0, 9)
    opt = ivp_amd.Options(method="DOPRI5", rtol=1e-6, atol=1e-9, t_eval=te)
    dev = torch.device("cuda:0")
    one = ivp_amd.solve_ivp_batch(ivp_amd.CR3BP(), t0, 2.0, torch.as_tensor(y0, device=dev), torch.as_tensor(p, device=dev), opt)
    got = D.solve_ivp_batch_multi(ivp_amd.CR3BP(), t0, 2.0, y0, p, opt, devices=[0] * shards, permutation=perm)
    for k in FIELDS + ("y_eval", "eval_idx", "n_filled"):
        assert torch.equal(getattr(got, k), getattr(one, k).to(getattr(got, k).dtype)), k
    # CSR log
    opt = ivp_amd.Options(method="DOPRI5", rtol=1e-6, atol=1e-9)
    one = ivp_amd.solve_ivp_batch_logged(ivp_amd.CR3BP(), t0, 2.0, y0, p, opt)
    got = D.solve_ivp_batch_multi(ivp_amd.CR3BP(), t0, 2.0, y0, p, opt, devices=[0] * shards, permutation=perm, log=True)
    assert torch.equal(got.log_offsets, one.log_offsets) and torch.equal(got.t_log, one.t_log) and torch.equal(got.y_log, one.y_log)
    assert torch.equal(got.n_log.to(torch.int64), one.n_log.to(torch.int64))
    for b in (0, 17, B - 1):
        s = O.solve_ivp("cr3bp", t0, 2.0, y0[:, b], params=p[:, b], detpow=True, method="DOPRI5", rtol=1e-6, atol=1e-9)
        t, y = got.log_of(b)
        assert np.array_equal(t.cpu().numpy(), s.t) and np.array_equal(y.cpu().numpy(), s.y)


_WORKER_SOL = r"""
import os, sys
sys.path.insert(0, {root!r})
import numpy as np, torch, torch.distributed as dist
from ivp_amd import workloads as W, distributed as D
import ivp_amd

rank = int(sys.argv[1])
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:{port}", rank=rank, world_size=2)
B = 2001
y0, p, t0, t1 = W.cr3bp_batch(B, seed=78)
perm = W.shard_permutation(B)
te = np.linspace(0.0, 3.0, 7)
opt_e = ivp_amd.Options(method="DOPRI5", rtol=1e-6, atol=1e-9, t_eval=te)
opt_l = ivp_amd.Options(method="DOPRI5", rtol=1e-6, atol=1e-9)
coll_e = D.solve_ivp_sharded(ivp_amd.CR3BP(), t0, 3.0, y0, p, opt_e, permutation=perm, device="cuda:0")
coll_l = D.solve_ivp_sharded(ivp_amd.CR3BP(), t0, 3.0, y0, p, opt_l, permutation=perm, device="cuda:0", log=True)
if rank == 0:
    peer_e = D.solve_ivp_batch_multi(ivp_amd.CR3BP(), t0, 3.0, y0, p, opt_e, devices=[0, 0], permutation=perm)
    for k in ("y_end", "t_end", "status", "naccpt", "y_eval", "eval_idx", "n_filled"):
        a = np.ascontiguousarray(coll_e[k]); b = np.ascontiguousarray(getattr(peer_e, k).cpu().numpy())
        assert a.tobytes() == b.astype(a.dtype).tobytes(), k
    peer_l = D.solve_ivp_batch_multi(ivp_amd.CR3BP(), t0, 3.0, y0, p, opt_l, devices=[0, 0], permutation=perm, log=True)
    for k in ("y_end", "n_log", "log_offsets", "t_log", "y_log"):
        a = np.ascontiguousarray(coll_l[k]); b = np.ascontiguousarray(getattr(peer_l, k).cpu().numpy())
        assert a.tobytes() == b.astype(a.dtype).tobytes(), k
    assert int(coll_l["log_offsets"][-1]) == coll_l["t_log"].shape[0] == int(coll_l["n_log"].sum())
dist.barrier()
dist.destroy_process_group()
print("rank", rank, "ok")
"""


def test_collective_and_peer_copy_gathers_of_the_whole_solution_are_byte_identical(tmp_path):
    """t_eval samples (one collective with the end states) and the CSR step log (one more collective of the padded
    record buffers): two processes sharing GPU 0 under gloo vs the two-context peer-copy path -- byte-identical."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    script = tmp_path / "multi_worker_sol.py"
    script.write_text(_WORKER_SOL.format(root=root, port=port))
    procs = [subprocess.Popen([sys.executable, str(script), str(r)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(2)]
    outs = [p.communicate(timeout=600)[0].decode() for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o[-2000:]
