"""CPU-only checks of the host side: the C-ABI library loads and exports every symbol include/ivp_hip.h
declares, option marshalling / enums mirror the reference, compute entry points fail loudly without a GPU
(there is no CPU fallback), and the multi-rank sharding + gather logic is correct under gloo (world_size 2)."""
import ctypes as C
import os
import re
import subprocess
import sys

import numpy as np
import pytest

import ivp_amd
from ivp_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    _lib.build()
    return _lib.load()


def test_library_exports_every_declared_symbol(lib):
    hdr = open(os.path.join(ROOT, "include", "ivp_hip.h")).read()
    declared = set(re.findall(r"^\s*(?:int|void|const char \*)\s*\*?\s*(ivp_\w+)\s*\(", hdr, flags=re.M))
    assert declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.ivp_abi_version() == 5


def test_struct_layouts_match_the_header(lib):
    # ivp_options_default must fill the struct exactly as Options::builder().build() does (options.rs:75-123)
    o = _lib.OptionsT()
    C.memset(C.byref(o), 0xFF, C.sizeof(o))
    lib.ivp_options_default(C.byref(o))
    assert (o.method, o.rtol, o.atol, o.max_steps, o.n_eval, o.has_first_step, o.has_max_step, o.dense_output) == \
           (1, 1e-3, 1e-6, 0, 0, 0, 0, 0)
    assert not o.t_eval and not o.rtol_vec and o.fp_mode == 0 and o.profile == 0 and o.max_log == 0
    n, p = C.c_int32(), C.c_int32()
    dims = {0: (1, 1), 1: (2, 0), 2: (2, 1), 3: (6, 1), 4: (3, 3), 5: (3, 0), 6: (2, 0), 7: (2, 0)}
    for rid, d in dims.items():
        assert lib.ivp_rhs_dims(rid, C.byref(n), C.byref(p)) == 0 and (n.value, p.value) == d
    assert lib.ivp_rhs_dims(99, C.byref(n), C.byref(p)) == -100


def test_no_cpu_fallback_without_a_device(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    assert lib.ivp_device_count() == 0
    h = C.c_void_p()
    assert lib.ivp_ctx_create(C.byref(h), 0) == -102          # IVP_ERR_NO_DEVICE
    with pytest.raises(ivp_amd.IvpError, match="no CPU fallback"):
        ivp_amd.solve_ivp(ivp_amd.SHO(), 0.0, 1.0, [1.0, 0.0])


def test_degenerate_cases_are_host_bookkeeping_only():
    # zero interval / empty state never reach an integrator in the reference (solve_ivp.rs:110-176)
    s = ivp_amd.solve_ivp(ivp_amd.SHO(), 1.23, 1.23, [2.0, 3.0], ivp_amd.Options(dense_output=True))
    assert s.status == ivp_amd.Status.Success and s.nfev == 0 and np.array_equal(s.y[-1], [2.0, 3.0])
    assert np.array_equal(s.continuous_sol.evaluate_extrapolate(7.0), [2.0, 3.0])
    s = ivp_amd.solve_ivp(ivp_amd.SHO(), 1.0, 1.0, [2.0, 3.0], ivp_amd.Options(t_eval=[0.5, 1.0, 1.0 + 1e-13, 2.0]))
    assert np.array_equal(s.t, [1.0, 1.0 + 1e-13])
    s = ivp_amd.solve_ivp(ivp_amd.SHO(), 0.0, 10.0, [])
    assert np.array_equal(s.t, [0.0, 10.0]) and s.y.shape == (2, 0)


def test_enums_and_defaults_mirror_the_reference():
    M, S = ivp_amd.Method, ivp_amd.Status
    assert [m.name for m in M] == ["RK23", "DOPRI5", "DOP853", "RK4", "RADAU", "BDF"]          # options.rs:14-27
    assert [s.name for s in S] == ["Success", "UserInterrupt", "NeedLargerNMax", "StepSizeTooSmall",
                                   "ProbablyStiff", "SingularMatrix", "PoorConvergence"]       # status.rs:4-19
    assert S.Success.is_success() and S.UserInterrupt.is_success() and not S.ProbablyStiff.is_success()
    assert M.from_str("RK45") == M.DOPRI5 and M.from_str("radau5") == M.RADAU and M.from_str("?") == M.DOPRI5
    assert [M.RK4.coeffs_per_state(), M.RK23.coeffs_per_state(), M.DOPRI5.coeffs_per_state(),
            M.DOP853.coeffs_per_state(), M.RADAU.coeffs_per_state(), M.BDF.coeffs_per_state()] == [4, 4, 5, 8, 4, 7]
    o = ivp_amd.Options()
    assert (o.method, o.rtol, o.atol, o.max_steps, o.t_eval, o.first_step, o.max_step, o.dense_output) == \
           (M.DOPRI5, 1e-3, 1e-6, None, None, None, None, False)


def test_continuous_output_semantics():
    """ContinuousOutput (cont.rs): first matching segment within 1e-12, strict vs extrapolating evaluation."""
    from oracle import oracle as O
    s = O.solve_ivp("sho", 0.0, 2.0, [1.0, 0.0], method="DOP853", rtol=1e-8, atol=1e-10, dense_output=True)
    co = ivp_amd.ContinuousOutput(ivp_amd.Method.DOP853, 2, s.seg_cont, s.seg_xold, s.seg_h)
    assert co.t_span() == s.sol_span()
    for t in np.linspace(0.0, 2.0, 23):
        assert np.array_equal(co.evaluate(t), s.sol(t))
    assert co.evaluate(2.5) is None and co.evaluate(-0.5) is None
    assert np.array_equal(co.evaluate_extrapolate(2.5), s.sol_extrapolate(2.5))
    assert np.array_equal(co.evaluate_extrapolate(-0.5), s.sol_extrapolate(-0.5))


def test_shard_bounds_cover_the_batch():
    from ivp_amd.distributed import shard_bounds
    for B in (1, 7, 8, 100_000, 100_003):
        for world in (1, 2, 3, 8):
            cuts = [shard_bounds(B, world, r) for r in range(world)]
            assert cuts[0][0] == 0 and cuts[-1][1] == B
            assert all(cuts[i][1] == cuts[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in cuts]
            assert max(sizes) - min(sizes) <= 1


_WORKER = r"""
import os, sys
sys.path.insert(0, {root!r})
import numpy as np, torch, torch.distributed as dist
from ivp_amd import workloads as W, distributed as D
import ivp_amd
from oracle import oracle as O

def oracle_solve(f, t0, t1, y0, p, opt):      # tests may use the oracle as the per-shard integrator
    r = O.solve_batch("cr3bp", y0, p, t0, t1, method="DOPRI5", rtol=opt.rtol, atol=opt.atol, detpow=True)
    return {{k: r[k] for k in ("y_end", "t_end", "h_next", "status", "nfev", "nstep", "naccpt", "nrejct")}}

dist.init_process_group("gloo", init_method="tcp://127.0.0.1:{port}", rank=int(sys.argv[1]), world_size=2)
B = 101                                        # odd: unequal shards
y0, p, t0, _ = W.cr3bp_batch(B)
perm = W.shard_permutation(B)
opt = ivp_amd.Options(method="DOPRI5", rtol=1e-6, atol=1e-9)
got = D.solve_ivp_sharded(ivp_amd.CR3BP(), t0, 2.0, y0, p, opt, permutation=perm, solve_fn=oracle_solve)
ref = O.solve_batch("cr3bp", y0, p, t0, 2.0, method="DOPRI5", rtol=1e-6, atol=1e-9, detpow=True)
for k in ("y_end", "t_end", "h_next", "status", "nfev", "nstep", "naccpt", "nrejct"):
    assert np.array_equal(np.asarray(got[k]).astype(ref[k].dtype), ref[k]), k
# a batch smaller than the world: rank 1's shard is empty, it still takes part in the collective (no deadlock)
got1 = D.solve_ivp_sharded(ivp_amd.CR3BP(), t0, 2.0, y0[:, :1], p[:, :1], opt, solve_fn=oracle_solve)
assert np.array_equal(got1["y_end"], ref["y_end"][:, :1]) and got1["naccpt"][0] == ref["naccpt"][0]
dist.barrier()
dist.destroy_process_group()
print("rank", sys.argv[1], "ok")
"""


def test_sharded_solve_and_gather_under_gloo_world_size_2(tmp_path):
    """N > 1 path: contiguous shards after the fixed permutation, one packed all-gather, original order restored.
    The per-shard integrator is injected (CPU oracle) because there is no GPU here; on the GPU box the same
    function runs with the HIP path and RCCL."""
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    script = tmp_path / "worker.py"
    script.write_text(_WORKER.format(root=ROOT, port=port))
    procs = [subprocess.Popen([sys.executable, str(script), str(r)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
             for r in range(2)]
    outs = [p.communicate(timeout=300)[0].decode() for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o


_WORKER_SOL = r"""
import os, sys
sys.path.insert(0, {root!r})
import numpy as np, torch, torch.distributed as dist
from ivp_amd import workloads as W, distributed as D
import ivp_amd
from oracle import oracle as O

KW = dict(method="DOPRI5", rtol=1e-6, atol=1e-9, detpow=True)

def oracle_solve(f, t0, t1, y0, p, opt, log=False):      # the per-shard integrator of this CPU test
    kw = dict(KW)
    if opt.t_eval is not None:
        kw["t_eval"] = np.asarray(opt.t_eval)
    r = O.solve_batch("cr3bp", y0, p, t0, t1, **kw)
    out = {{k: r[k] for k in ("y_end", "t_end", "h_next", "status", "nfev", "nstep", "naccpt", "nrejct")}}
    if opt.t_eval is not None:
        out["y_eval"], out["n_filled"] = r["y_eval"], r["n_filled"]
    if log:
        ts, ys = [], []
        for b in range(y0.shape[1]):
            s = O.solve_ivp("cr3bp", float(np.atleast_1d(t0)[0]), float(np.atleast_1d(t1)[0]), y0[:, b], params=p[:, b], **KW)
            ts.append(s.t); ys.append(s.y)
        out["n_log"] = np.array([len(t) for t in ts], dtype=np.int32)
        out["t_log"] = np.concatenate(ts) if ts else np.zeros(0)
        out["y_log"] = np.concatenate(ys) if ys else np.zeros((0, 6))
    return out

dist.init_process_group("gloo", init_method="tcp://127.0.0.1:{port}", rank=int(sys.argv[1]), world_size=2)
B = 53                                         # odd: unequal shards
y0, p, t0, _ = W.cr3bp_batch(B)
perm = W.shard_permutation(B)
te = np.linspace(0.0, 2.0, 6)
# ---- t_eval samples ride in the same arena as the end states: ONE collective ----
opt = ivp_amd.Options(method="DOPRI5", rtol=1e-6, atol=1e-9, t_eval=te)
got = D.solve_ivp_sharded(ivp_amd.CR3BP(), t0, 2.0, y0, p, opt, permutation=perm, solve_fn=oracle_solve)
ref = O.solve_batch("cr3bp", y0, p, t0, 2.0, t_eval=te, **KW)
for k in ("y_end", "t_end", "status", "naccpt", "y_eval", "n_filled"):
    assert np.array_equal(np.asarray(got[k]).astype(ref[k].dtype), ref[k]), k
# ---- Solution.t / Solution.y of every trajectory: CSR log, counts in the arena + ONE collective of the records ----
opt = ivp_amd.Options(method="DOPRI5", rtol=1e-6, atol=1e-9)
got = D.solve_ivp_sharded(ivp_amd.CR3BP(), t0, 2.0, y0, p, opt, permutation=perm, solve_fn=oracle_solve, log=True)
off = got["log_offsets"]
assert off[0] == 0 and off[-1] == got["t_log"].shape[0] == int(got["n_log"].sum())
for b in range(B):                             # ORIGINAL trajectory order, record for record
    s = O.solve_ivp("cr3bp", t0, 2.0, y0[:, b], params=p[:, b], **KW)
    assert np.array_equal(got["t_log"][off[b]:off[b + 1]], s.t) and np.array_equal(got["y_log"][off[b]:off[b + 1]], s.y), b
# a batch smaller than the world: rank 1 has no records, still takes part in both collectives
got1 = D.solve_ivp_sharded(ivp_amd.CR3BP(), t0, 2.0, y0[:, :1], p[:, :1], opt, solve_fn=oracle_solve, log=True)
s = O.solve_ivp("cr3bp", t0, 2.0, y0[:, 0], params=p[:, 0], **KW)
assert np.array_equal(got1["t_log"], s.t) and np.array_equal(got1["y_log"], s.y)
dist.barrier()
dist.destroy_process_group()
print("rank", sys.argv[1], "ok")
"""


def test_sharded_solution_gather_carries_t_eval_samples_and_the_csr_log_under_gloo(tmp_path):
    """BASELINE C4 says "gather of sol.y": the reference's Solution.y is the whole sampled trajectory
    (src/solve/solution.rs:7-20, src/solve/solout.rs:344-428).  World size 2 under gloo, unequal shards, fixed
    permutation: t_eval samples and the accepted-step log of every trajectory arrive on every rank in the original
    order and equal the oracle's records."""
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    script = tmp_path / "worker_sol.py"
    script.write_text(_WORKER_SOL.format(root=ROOT, port=port))
    procs = [subprocess.Popen([sys.executable, str(script), str(r)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
             for r in range(2)]
    outs = [p.communicate(timeout=300)[0].decode() for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o


def test_method_defaults_are_the_reference_struct_defaults():
    """ivp_options_method_defaults(): dopri5.rs:34-72, dop853.rs:34-63, rk23.rs:17-37 (no GPU needed)."""
    import ctypes as C
    from ivp_amd import _lib
    lib = _lib.load()
    o = _lib.OptionsT()
    lib.ivp_options_default(C.byref(o))
    assert o.has_settings == 0
    want = {1: (2.3e-16, 0.9, 0.2, 10.0, 0.04, 1000), 2: (2.3e-16, 0.9, 0.333, 6.0, 0.0, 1000), 0: (2.3e-16, 0.9, 0.2, 10.0, 0.0, 1000)}
    for m, w in want.items():
        assert lib.ivp_options_method_defaults(C.byref(o), m) == 0
        assert (o.method, o.uround, o.safety_factor, o.scale_min, o.scale_max, o.beta, o.stiff_test) == (m, *w)
    for m in (3, 4, 5, 17):
        assert lib.ivp_options_method_defaults(C.byref(o), m) == -100


def test_graft_entry_build_passes():
    """The driver's "does it build" check must keep working when the ABI version moves."""
    import __graft_entry__ as g
    g.build()


_WORKER_OG = r"""
import os, sys
sys.path.insert(0, {root!r})
import torch, torch.distributed as dist
from ivp_amd.distributed import OverlappedGather

rank = int(sys.argv[1])
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:{port}", rank=rank, world_size=2)
og = OverlappedGather((3, 5), torch.float64, torch.device("cpu"))
bufs = [torch.empty((3, 5), dtype=torch.float64) for _ in range(2)]
seen = []
for step in range(7):                         # the loop bench.py runs: slot -> produce -> launch
    k = og.slot()                             # waits for the gather that last read bufs[k]
    if step >= 2:                             # that gather carried step - 2: every rank's result of that step
        want = torch.stack([torch.full((3, 5), 100.0 * r + (step - 2), dtype=torch.float64) for r in range(2)])
        assert torch.equal(og.gathered[k], want), (step, og.gathered[k])
        seen.append(step - 2)
    bufs[k].fill_(100.0 * rank + step)        # "integrate" step into the free buffer
    og.launch(k, bufs[k])
og.drain()
for step in (5, 6):
    want = torch.stack([torch.full((3, 5), 100.0 * r + step, dtype=torch.float64) for r in range(2)])
    assert torch.equal(og.gathered[step & 1], want)
assert seen == [0, 1, 2, 3, 4]
dist.barrier()
dist.destroy_process_group()
print("rank", rank, "ok")
"""


def test_overlapped_gather_double_buffering_under_gloo_world_size_2(tmp_path):
    """bench.py --gpus N: step i's all-gather overlaps step i+1; a buffer is only rewritten after its gather is done
    and every gather delivers every rank's result of that step."""
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    script = tmp_path / "worker_og.py"
    script.write_text(_WORKER_OG.format(root=ROOT, port=port))
    procs = [subprocess.Popen([sys.executable, str(script), str(r)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
             for r in range(2)]
    outs = [p.communicate(timeout=300)[0].decode() for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o


_WORKER_BENCH = r"""
import os, sys
sys.path.insert(0, {root!r})
import numpy as np, torch, torch.distributed as dist
from ivp_amd import workloads as W, distributed as D
from oracle import oracle as O

rank = int(sys.argv[1]); world = 2
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:{port}", rank=rank, world_size=world)
B = 64                                           # bench.py requires B % world == 0 (equal arenas)
y0, p, t0, t1 = W.cr3bp_batch(B)
t1 = 2.0
perm = W.shard_permutation(B)
lo, hi = D.shard_bounds(B, world, rank)
idx = perm[lo:hi]
ys, ps = np.ascontiguousarray(y0[:, idx]), np.ascontiguousarray(p[:, idx])
calls = [0]
def solve_into(sol):                             # bench.py passes ivp_amd.solve_ivp_batch(..., out=sol) here
    r = O.solve_batch("cr3bp", ys, ps, t0, t1, method="DOPRI5", rtol=1e-6, atol=1e-9, detpow=True)
    for k in ("y_end", "t_end", "h_next", "status", "nfev", "nstep", "naccpt", "nrejct"):
        getattr(sol, k).copy_(torch.as_tensor(r[k].astype(np.int64) if r[k].dtype == np.uint64 else r[k]))
    calls[0] += 1
    return sol
el, out, og, arenas = D.run_steps(solve_into, 6, hi - lo, torch.device("cpu"), steps=3, warmup=1, gather=True, d2h=False)
assert calls[0] == 4 and el > 0
# the gather of the LAST step: every rank holds every shard; undo the permutation and compare with one whole-batch solve
g = arenas[0].split(og.gathered[(og.steps - 1) & 1], [hi - lo] * world)
full = D._unpermute(g, perm, B)
ref = O.solve_batch("cr3bp", y0, p, t0, t1, method="DOPRI5", rtol=1e-6, atol=1e-9, detpow=True)
for k in ("y_end", "t_end", "h_next", "status", "nfev", "nstep", "naccpt", "nrejct"):
    assert np.array_equal(full[k].numpy().astype(ref[k].dtype), ref[k]), k
el2, _, og2, _ = D.run_steps(solve_into, 6, hi - lo, torch.device("cpu"), steps=2, warmup=0, gather=False, d2h=True)
assert og2 is None
dist.barrier()
dist.destroy_process_group()
print("rank", rank, "ok")
"""


def test_bench_step_loop_strong_scaling_under_gloo_world_size_2(tmp_path):
    """The exact loop bench.py --gpus N times (ivp_amd.distributed.run_steps): shard after the fixed permutation,
    integrate into a byte arena, overlapped all-gather of the arena per step, every rank ends up with the whole batch.
    The per-shard integrator is the CPU oracle here; on the GPU box it is the HIP path and the backend is RCCL."""
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    script = tmp_path / "worker_bench.py"
    script.write_text(_WORKER_BENCH.format(root=ROOT, port=port))
    procs = [subprocess.Popen([sys.executable, str(script), str(r)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
             for r in range(2)]
    outs = [p.communicate(timeout=300)[0].decode() for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o
