#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X batched explicit-RK integrator.

Workload (BASELINE.json configs[1], "C2"): 100 000 independent 6-state CR3BP trajectories (perturbed
Arenstorf orbits, one period), DOPRI5, rtol=1e-6, atol=1e-9, inputs resident in HBM.
One "step" = one complete solve of that batch (init kernel + chunked stepping kernels until every
trajectory has reached t_end).  Metric: accepted RK steps per second, aggregate over the batch and
over all ranks.

  python bench.py [--gpus N] [--steps K] [--warmup W]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

N > 1 (BASELINE config C4 as written): one process per GPU; the SAME batch of 100 000 trajectories is cut into N
contiguous shards after a fixed permutation (trajectories are independent, so there is no data-path collective) and the
end-state arenas are gathered with ONE RCCL all-gather per step inside the timed region: "scaling": "strong".  A weak-
scaling figure (100 000 trajectories PER GPU) is measured in the same run and reported as the "weak" object.

Timing: the headline steps run with options.profile = 0; the per-launch HIP-event durations that feed the roofline
objects come from a separate instrumented pass of the same K steps; "wall_ms_with_d2h" repeats the steps with a D2H
copy of the results into pinned host memory after every solve.

Rank 0 prints ONE JSON line.  Extra objects: "roofline" (HBM view, as the contract asks),
"roofline_fp64" (counted FP64 flops), "roofline_issue" (the resource that actually binds the stepping kernel:
vector-instruction issue, from the committed SQ-counter profile) and "cpu_baseline"
(the CPU oracle, i.e. the C restatement of the reference algorithm, timed on this host's cores).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
# the CPU baseline's OpenMP runtime reads these when the oracle library is first loaded: one thread per core, spread over the sockets
os.environ.setdefault("OMP_PROC_BIND", "spread")
os.environ.setdefault("OMP_PLACES", "cores")

import numpy as np  # noqa: E402
import torch  # noqa: E402
from kernel_sha import kernel_sources_sha256  # noqa: E402

PROFILE_ROUND = "r04"        # the counter profiles under profiles/ this file reads (each carries the hash of the kernels it was taken on)
HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FP64_PEAK_TFLOPS = 78.6      # MI355X FP64 vector peak (FMA = 2 flop), SURVEY.md section 8d
# Algorithmic work of ONE DOPRI5 step attempt on the 6-state CR3BP system:
#   stage AXPYs 46 + error combo 13 + norm 8 + dense 19 = 86 flop/component, 6 RHS evaluations of ~45 flop,
#   plus 13 sqrt, ~38 div and 2 pow counted as one flop each.
FLOP_PER_ATTEMPT = 86 * 6 + 6 * 45 + 13 + 38 + 2
# Algorithmic HBM bytes per trajectory per stepping-kernel launch: state in + state out
#   in : y[6] k1[6] x h facold hlamb (16 f64) + mu + flags,status (2x4) + nstep,naccpt (2x8) + perm id 4
#   out: y[6] k1[6] x h facold hlamb (16 f64) + flags,status + 4 counters read-modify-write (4x16) + perm id 4
BYTES_PER_LANE_LAUNCH = (16 * 8 + 8 + 8 + 16 + 4) + (16 * 8 + 8 + 64 + 4)


WORKLOADS = {
    "c2": dict(gen="cr3bp_batch", seed=20260102, B=100_000, problem="CR3BP", method="DOPRI5", rtol=1e-6, atol=1e-9,
               flop_per_attempt=FLOP_PER_ATTEMPT, kernel="chunk_kernel_t<DOPRI5, RhsCr3bp> (bulk launches) + coop_chunk_kernel<RhsCr3bp> (tail launch)",
               metric="accepted RK steps/sec (aggregate), batched CR3BP DOPRI5 @ rtol=1e-6",
               desc="C2: 100k independent 6-state CR3BP trajectories (perturbed Arenstorf orbits, one period), "
                    "DOPRI5 rtol=1e-6 atol=1e-9; one step = whole batch integrated to t_end"),
    # DOP853 on the 2-state Van der Pol system: ~0.4 kFLOP per attempt (dense stages elided)
    "c3": dict(gen="vdp_batch", seed=20260103, B=1_000_000, problem="VanDerPol", method="DOP853", rtol=1e-8, atol=1e-10,
               flop_per_attempt=400, kernel="chunk_kernel_t<DOP853, RhsVdp>",
               metric="accepted RK steps/sec (aggregate), batched Van der Pol DOP853 @ rtol=1e-8",
               desc="C3: 1M independent 2-state Van der Pol (mu=1) trajectories, per-trajectory t_end in [50,100], "
                    "DOP853 rtol=1e-8 atol=1e-10"),
    "c5": dict(gen="vdp_stiff_batch", seed=20260105, B=10_000, problem="VanDerPol", method="BDF", rtol=1e-4, atol=1e-6,
               flop_per_attempt=None, kernel="chunk_kernel_t<BDF, RhsVdp>",
               metric="accepted BDF steps/sec (aggregate), batched stiff Van der Pol (mu~1000) @ rtol=1e-4",
               desc="C5: 10k stiff Van der Pol (mu ~ 1000) trajectories, t in [0,3000], BDF order 1-5, rtol=1e-4 atol=1e-6"),
}


def bytes_per_lane_launch(n, n_params, workload):
    """Algorithmic HBM bytes per trajectory per stepping-kernel launch: state in + state out."""
    if workload == "c5":   # BDF: y, D[8][n], J, LU, scalars, 6 counters
        st = (n + 8 * n + 2 * n * n + 4) * 8 + 4 + 4 + 4
        return (st + n_params * 8 + 16 + 4) + (st + 6 * 16 + 4)
    st = (2 * n + 4) * 8
    return (st + n_params * 8 + 8 + 16 + 4) + (st + 8 + 64 + 4)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=100_000,
                    help="trajectories in the whole batch (BASELINE C2 / C4: 100000).  With --gpus 8 ONE 100k batch cannot reach the >= 6x strong-"
                         "scaling target (its slowest trajectory's sequential attempts do not shrink with the shard); the single-GPU prediction "
                         "(profiles/r04_strong_scaling_prediction.json, DESIGN.md section 7) puts the threshold at 6.4M trajectories and 8M safely "
                         "above it (6.4x): --gpus 8 --batch 8000000 --max-steps 1000")
    ap.add_argument("--max-steps", type=int, default=0,
                    help="Options.max_steps of every solve (0 = None = unlimited, the reference's default and the headline's).  Batches of more than "
                         "~400k perturbed Arenstorf orbits contain collision orbits (one needs > 200 000 steps) that end with NeedLargerNMax under a budget")
    ap.add_argument("--fp", choices=["strict", "fma", "fast"], default="strict",
                    help="arithmetic mode of the kernels: strict (headline: the reference's IEEE operation sequence) or fma "
                         "(the defined FMA mode, bit-comparable with oracle/liboracle_fma.so; 'fast' is its older name)")
    ap.add_argument("--chunk", type=int, default=0, help="step attempts per launch (0 = library default)")
    ap.add_argument("--workload", choices=["c2", "c3", "c5"], default="c2",
                    help="c2 = BASELINE headline (default; with --gpus N > 1 it is C4: the same batch sharded); "
                         "c3 = 1M Van der Pol DOP853 rtol 1e-8; c5 = 10k stiff Van der Pol BDF")
    ap.add_argument("--no-weak", action="store_true", help="N > 1: skip the secondary weak-scaling figure")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-fast", action="store_true", help="skip the secondary fast-FP-mode / pipelined measurements")
    ap.add_argument("--output", choices=["end", "t_eval", "csr_log", "dense", "events", "all"], default="end",
                    help="N = 1: measure the device DefaultSolOut output modes of the workload instead of the end-state "
                         "headline (t_eval: 128 samples per trajectory; csr_log: every accepted step, count + fill passes; "
                         "dense: step log + dense-output segments; events: a hiprtc CR3BP with the y = 0 crossing event); "
                         "the default run carries t_eval and csr_log as the secondary object `outputs`")
    args = ap.parse_args()

    # RCCL prints a version banner on stdout while the process group comes up; keep stdout clean for the ONE JSON
    # line by pointing fd 1 at stderr until the result is ready.
    sys.stdout.flush()
    saved_stdout = os.dup(1)
    os.dup2(2, 1)

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE = {world}: launch with `python -m torch.distributed.run --nnodes=1 "
                         f"--nproc-per-node {args.gpus} --master-addr 127.0.0.1 bench.py --gpus {args.gpus} ...` (one rank per GPU)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    launched = "RANK" in os.environ and "MASTER_PORT" in os.environ   # under torch.distributed.run
    if world > 1 or launched:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)  # "nccl" is RCCL on ROCm
        assert dist.get_world_size() == args.gpus, (dist.get_world_size(), args.gpus)

    import ivp_amd
    from ivp_amd import workloads as W
    from ivp_amd.distributed import ResultArena, run_steps, shard_bounds

    if args.fp == "fast":
        args.fp = "fma"
    fp = ivp_amd.FpMode.FMA if args.fp == "fma" else ivp_amd.FpMode.STRICT
    wl = WORKLOADS[args.workload]
    B = args.batch if args.workload == "c2" or args.batch != 100_000 else wl["B"]
    prob = getattr(ivp_amd, wl["problem"])()
    n_state = prob.n
    ctx = ivp_amd.Context(local_rank)
    mk_opts = lambda profile: ivp_amd.Options(method=wl["method"], rtol=wl["rtol"], atol=wl["atol"], fp_mode=fp,
                                              chunk_attempts=args.chunk, profile=profile, max_steps=args.max_steps or None)

    def barrier_sync():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def run_case(y0, p, t1, profile, with_gather, with_d2h=False):
        """W warm-up + K timed steps of one shard on this rank (ivp_amd.distributed.run_steps: one step = one complete
        solve of the shard, + the all-gather of its end-state arena when `with_gather`, overlapped with the next
        step's integration and waited for inside the timed region).  Returns elapsed seconds (this rank), the last
        result, with profile the accumulated launch statistics, and the gather object."""
        m = y0.shape[1]
        y0d, pd = torch.as_tensor(y0, device=dev), torch.as_tensor(p, device=dev)
        t1d = torch.as_tensor(t1, device=dev) if np.ndim(t1) else t1
        opts = mk_opts(profile)
        acc = dict(kern_ms=0.0, launches=0.0, slots=0.0, lane_launches=0.0, coop_ms=0.0, coop_launches=0.0,
                   decl=0.0, decl_coop=0.0, decl_ms=0.0, decl_coop_ms=0.0)

        def on_step(out):
            if profile:
                st = out.stats
                acc["kern_ms"] += st["step_kernel_ms"]; acc["launches"] += st["launches"]; acc["slots"] += st["lane_attempt_slots"]
                acc["lane_launches"] += st["lane_launches"]; acc["coop_ms"] += st["coop_kernel_ms"]; acc["coop_launches"] += st["coop_launches"]
                acc["decl"] += st["declined_launches"]; acc["decl_coop"] += st["declined_coop_launches"]
                acc["decl_ms"] += st["declined_ms"]; acc["decl_coop_ms"] += st["declined_coop_ms"]

        el, out, og, _ = run_steps(lambda sol: ivp_amd.solve_ivp_batch(prob, 0.0, t1d, y0d, pd, opts, ctx, sol), n_state, m, dev,
                                   args.steps, args.warmup, gather=with_gather, d2h=with_d2h, on_step=on_step)
        return el, out, acc, og

    def reduce_max_sum(elapsed, accepted):
        t_el = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        t_acc = torch.tensor([float(accepted)], dtype=torch.float64, device=dev)
        if dist is not None:
            dist.all_reduce(t_el, op=dist.ReduceOp.MAX)
            dist.all_reduce(t_acc, op=dist.ReduceOp.SUM)
        return float(t_el.item()), float(t_acc.item())

    # ---- STRONG scaling (headline; BASELINE C4 as written): ONE batch of B trajectories, fixed permutation for
    # N > 1, contiguous shards of B/N, one RCCL all-gather of the end-state arena per step inside the timed region ----
    if B % world:
        raise SystemExit(f"--batch {B} must be a multiple of --gpus {world} (equal shards: one all-gather of equal-sized arenas)")
    y0, p, t0, t1 = getattr(W, wl["gen"])(B, seed=wl["seed"])
    assert t0 == 0.0
    lo, hi = shard_bounds(B, world, rank)
    if world > 1:
        idx = W.shard_permutation(B)[lo:hi]
        sh = (np.ascontiguousarray(y0[:, idx]), np.ascontiguousarray(p[:, idx]), t1[idx] if np.ndim(t1) else t1)
    else:
        sh = (y0, p, t1)
    el, out, _, og = run_case(*sh, profile=0, with_gather=dist is not None)
    acc_rank = int(out.naccpt.sum().item())
    ok = bool((out.status == 0).all().item())
    # per-trajectory Status codes of this rank's shard (src/status.rs order).  C5: 9 of the 10 000 stiff Van der Pol
    # trajectories end AT t_end with StepSizeTooSmall -- the reference's stagnation guard `(x + 0.1 |h|) == x` firing on a clamped
    # last step of a few ulp (bdf.rs:308-328), reproduced bit for bit by the oracle (tests/test_gpu_parity.py compares the status words).
    codes, counts = torch.unique(out.status, return_counts=True)
    names = {int(v): v.name for v in ivp_amd.Status}
    status_counts = {names.get(int(c), str(int(c))): int(k) for c, k in zip(codes.tolist(), counts.tolist())}
    nstep_rank = float(out.nstep.sum().item())
    nrej_rank = float(out.nrejct.sum().item())
    elapsed, total_acc = reduce_max_sum(el, acc_rank)
    gathered_ok = None
    if og is not None:   # every rank now holds every shard's end state: check the gathered status words and counters
        g = ResultArena(n_state, hi - lo, dev).split(og.gathered[(og.steps - 1) & 1], [hi - lo] * world)
        gathered_ok = bool((g["status"] == 0).all().item()) and int(g["naccpt"].sum().item()) == int(total_acc)

    # ---- instrumented pass (NOT the headline): per-launch HIP-event durations for the roofline objects ----
    el_p, out_p, acc, _ = run_case(*sh, profile=1, with_gather=False)
    # ---- results on the host: the same steps followed by a D2H copy of the end-state arena into pinned memory ----
    el_h, _, _, _ = run_case(*sh, profile=0, with_gather=False, with_d2h=True)
    el_h, _ = reduce_max_sum(el_h, 0)

    # ---- WEAK scaling (secondary figure, N > 1): B trajectories PER RANK (a parameter sweep: rank r takes seed + r) ----
    weak = None
    if world > 1 and not args.no_weak:
        yw, pw, _, t1w = getattr(W, wl["gen"])(B, seed=wl["seed"] + rank)
        el_w, out_w, _, _ = run_case(yw, pw, t1w, profile=0, with_gather=True)
        el_w, acc_w = reduce_max_sum(el_w, int(out_w.naccpt.sum().item()))
        weak = {"scaling": "weak", "trajectories_per_gpu": B, "value": acc_w * args.steps / el_w, "unit": "steps/s",
                "ms_per_step": el_w / args.steps * 1e3,
                "note": "every rank integrates its own batch of the full size (seed + rank) and all-gathers its end-state arena"}

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = total_acc * args.steps / elapsed
        launches, kern_ms, coop_ms, coop_launches = acc["launches"], acc["kern_ms"], acc["coop_ms"], acc["coop_launches"]
        attempts = nstep_rank * args.steps   # DOPRI5/DOP853/BDF: nstep counts every attempt (dopri5.rs:285)
        avg_launch_ms = kern_ms / max(launches, 1)
        attempts_per_launch = attempts / max(launches, 1)
        lanes_per_launch = acc["lane_launches"] / max(launches, 1)  # trajectories that load + store their state
        flop_per_attempt = wl["flop_per_attempt"]
        bytes_per_launch = lanes_per_launch * bytes_per_lane_launch(n_state, prob.n_params, args.workload)
        gbs = bytes_per_launch / (avg_launch_ms * 1e-3) / 1e9 if avg_launch_ms > 0 else 0.0
        tflops = (attempts_per_launch * flop_per_attempt / (avg_launch_ms * 1e-3) / 1e12) if (avg_launch_ms > 0 and flop_per_attempt) else None
        # PMC-derived figures come from committed profiles of this command (separate rocprofv3 --pmc passes: counters cannot be
        # collected inside a timed run).  Each profile is stamped with the hash of the kernel sources it was taken on; a
        # profile of OTHER kernels than the ones just timed is reported as stale and nothing is derived from it.
        sha_now = kernel_sources_sha256()
        traffic = traffic_src = None
        traffic_stale = None
        pmc_name = f"{PROFILE_ROUND}_pmc_hbm_bytes_per_launch_{args.fp}.json"
        pmc = os.path.join(ROOT, "profiles", pmc_name)
        if os.path.exists(pmc) and args.workload == "c2" and world == 1 and B == 100_000:
            try:
                j = json.load(open(pmc))
                traffic_stale = j.get("kernel_sources_sha256") != sha_now
                if not traffic_stale:
                    traffic = j.get("hbm_bytes_per_launch")
                traffic_src = f"profiles/{pmc_name} (" + str(j.get("run", "rocprofv3 --pmc passes of tools/profile_c2.sh")) + \
                              "): PMC counters of a separate profiled run of this command, not of this process" + \
                              ("; STALE: taken on other kernel sources than this run's (hash mismatch), not used" if traffic_stale else "")
            except Exception:
                traffic = None
        # VALU issue view of the dominant kernel: the thread-per-trajectory stepping kernel is bound by vector-instruction
        # issue (one wave64 FP64 instruction per 4 cycles per SIMD at best), not by bytes or by counted flops (strict mode
        # spends separate multiply and add instructions, 11 on a division, 18 on a square root).  Instructions per launch
        # come from the committed SQ-counter profile of this command (profiles/<PROFILE_ROUND>_sq_counters_<workload>_<fp>.json,
        # SQ_INSTS_VALU, tools/profile_sq.sh), the launch duration is this run's.
        issue = None
        sq_name = f"{PROFILE_ROUND}_sq_counters_{args.workload}_{args.fp}.json"
        sqf = os.path.join(ROOT, "profiles", sq_name)
        chunk_launches = launches - coop_launches
        chunk_ms = (kern_ms - coop_ms) / max(chunk_launches, 1)
        if os.path.exists(sqf) and world == 1 and B == wl["B"] and chunk_ms > 0:
            try:
                sq = json.load(open(sqf))
                if sq.get("kernel_sources_sha256") != sha_now:
                    raise LookupError("stale")
                ent = next(v for k, v in sq.items() if "chunk_kernel_t" in k and "coop" not in k)
                valu = float(ent["mean_per_dispatch"]["SQ_INSTS_VALU"])
                if "per_solve" in ent:   # instructions of a whole solve / this run's launches per solve: robust against a
                    valu = float(ent["per_solve"]["SQ_INSTS_VALU"]) / max(chunk_launches / args.steps, 1e-9)   # different cut into launches
                peak = 256 * 4 * 2.4e9 / 4.0   # SIMDs x (clock / 4 cycles per wave64 instruction), at the 2.4 GHz peak clock
                issue = {"bound": "valu_issue", "kernel": "chunk_kernel_t", "achieved": valu / (chunk_ms * 1e-3), "peak": peak,
                         "unit": "wave-instructions/s", "frac": valu / (chunk_ms * 1e-3) / peak, "valu_instructions_per_launch": valu,
                         "avg_launch_ms": chunk_ms,
                         "source": f"profiles/{sq_name} (SQ_INSTS_VALU per launch, separate rocprofv3 --pmc "
                                   "run of this command) / this run's launch duration"}
            except LookupError:
                issue = {"bound": "valu_issue", "kernel": "chunk_kernel_t", "stale": True,
                         "source": f"profiles/{sq_name}: taken on other kernel sources than this run's (hash mismatch), nothing derived"}
            except Exception:
                issue = None
        fp64_frac = (tflops / FP64_PEAK_TFLOPS) if tflops is not None else None
        res = {
            "metric": wl["metric"],
            "value": value,
            "unit": "steps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": wl["desc"] + (f"; C4: sharded over {world} GPUs after the fixed permutation (seed 20260104), "
                                           "RCCL all-gather of the end-state arena per step" if world > 1 else ""),
                "trajectories_total": B,
                "trajectories_per_gpu": hi - lo,
                "fp_mode": args.fp,
                "chunk_attempts": args.chunk or 64, "max_steps": args.max_steps or None,
                "parallelism": f"dp{world} (independent shards" + (", one RCCL all-gather of the end states per step)" if world > 1 else ")"),
            },
            "wall_ms_to_t_end": ms_per_step,
            "wall_ms_with_d2h": el_h / args.steps * 1e3,
            "instrumented_ms_per_step": el_p / args.steps * 1e3,
            "timing_note": "value / ms_per_step: options.profile = 0 (no per-launch events); roofline launch durations come "
                           "from a separate instrumented pass of the same K steps (instrumented_ms_per_step, this rank)",
            "accepted_steps_per_batch": total_acc,
            "all_success": ok,
            "status_counts": status_counts,
            "gathered_ok": gathered_ok,
            "wave_lane_utilisation": attempts / acc["slots"] if acc["slots"] else None,
            "attempts_per_s": attempts / el * 1.0, "rejection_ratio": nrej_rank / max(nstep_rank, 1.0),
            "roofline": {
                "bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
                "traffic": traffic, "traffic_source": traffic_src, "traffic_stale": traffic_stale,
                # the resource that actually bounds this path: FP64 vector arithmetic, counted flops of the
                # whole solve over its wall time -- the HBM figures above are what north_star asks to be reported, and small
                "binding": {"bound": "valu_fp64", "achieved": (attempts * flop_per_attempt / elapsed / 1e12) if flop_per_attempt else None,
                            "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                            "frac": (attempts * flop_per_attempt / elapsed / 1e12 / FP64_PEAK_TFLOPS) if flop_per_attempt else None,
                            "kernel_time_frac": fp64_frac,
                            "note": "algorithmic flops of every step attempt of one solve / wall time of the solve; kernel_time_frac: the same "
                                    "flops / time inside the stepping kernels"},
                "kernel": wl["kernel"], "avg_launch_ms": avg_launch_ms,
                "launches_per_step": launches / args.steps,
                # per kernel name, for a one-to-one check against the rocprofv3 --kernel-trace --stats rows
                # (bulk, cooperative) launch PAIRS: exactly one launch of a pair works, its partner returns at once (~3 us).
                # launches_per_step / avg_launch_ms are the working launches; *_incl_declined is what a kernel trace averages
                # (rocprofv3 --stats counts the declined partner launches as calls of the same kernel)
                "per_kernel": {
                    "chunk_kernel_t": {"launches_per_step": (launches - coop_launches) / args.steps,
                                       "avg_launch_ms": (kern_ms - coop_ms - (acc["decl_ms"] - acc["decl_coop_ms"])) / max(launches - coop_launches, 1),
                                       "calls_per_step_incl_declined": (launches - coop_launches + acc["decl"] - acc["decl_coop"]) / args.steps,
                                       "avg_call_ms_incl_declined": (kern_ms - coop_ms) / max(launches - coop_launches + acc["decl"] - acc["decl_coop"], 1)},
                    "coop_chunk_kernel": {"launches_per_step": coop_launches / args.steps,
                                          "avg_launch_ms": ((coop_ms - acc["decl_coop_ms"]) / coop_launches) if coop_launches else None,
                                          "calls_per_step_incl_declined": (coop_launches + acc["decl_coop"]) / args.steps,
                                          "avg_call_ms_incl_declined": (coop_ms / (coop_launches + acc["decl_coop"])) if (coop_launches + acc["decl_coop"]) else None},
                },
                "note": "state is device-resident: HBM traffic is per trajectory per launch, not per step; "
                        "this fraction is informational, the binding resource is FP64 VALU issue (roofline_fp64)",
            },
            "roofline_fp64": {
                "bound": "valu_fp64", "achieved": tflops, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": (tflops / FP64_PEAK_TFLOPS) if tflops is not None else None, "flop_per_attempt": flop_per_attempt,
                "attempts_per_launch": attempts_per_launch,
            },
            "roofline_issue": issue,
            "kernel_sources_sha256": sha_now,
        }
        if world > 1:
            res["rccl_ranks"] = dist.get_world_size()
        if weak is not None:
            res["weak"] = weak
        if world > 1:
            # the builder's own single-GPU prediction of this curve, on the record next to the measured number
            # (every rank's shard timed alone on one MI355X; max over ranks = what the job takes, gather excluded)
            try:
                pred = json.load(open(os.path.join(ROOT, "profiles", f"{PROFILE_ROUND}_strong_scaling_prediction.json")))
                res["strong_scaling_prediction"] = {"source": f"profiles/{PROFILE_ROUND}_strong_scaling_prediction.json (tools/strong_scaling_prediction.py, strict mode, "
                                                              "one MI355X: every rank's shard timed alone, gather excluded)", "prediction": pred,
                                                    "stale": pred.get("kernel_sources_sha256") != sha_now,
                                                    "baseline_target": ">= 6x at 8 GPUs (BASELINE.json): out of reach for ONE 100k batch (the slowest "
                                                                       "trajectory's 702 sequential attempts do not shrink with the shard); the prediction "
                                                                       "file names the batch size that reaches it"}
            except Exception:   # noqa: BLE001
                pass
            res["latency_floor_note"] = ("strong scaling of one C2 batch is bounded by its slowest trajectory: 702 sequential "
                                         "step attempts (192 in the bulk kernel + 510 in the lane-cooperative tail kernel) "
                                         "do not shrink with the shard -- see DESIGN.md section 7")
        single = dist is None
        y0d, pd = torch.as_tensor(y0, device=dev), torch.as_tensor(p, device=dev)
        if args.fp == "strict" and not args.no_fast and single and args.workload == "c2":   # single process only (it synchronises)
            res["fma_fp_mode"] = fma_mode_numbers(ivp_amd, prob, t0, t1, y0d, pd, ctx, args, barrier_sync)
        if not args.no_fast and single and args.workload == "c2":
            res["pipelined"] = pipelined_numbers(ivp_amd, prob, t0, t1, y0d, pd, fp, args)
        if single and args.workload in ("c2", "c3") and (args.output != "end" or not args.no_fast):
            modes = ["t_eval", "csr_log"] if args.output == "end" else (["t_eval", "csr_log", "dense", "events"] if args.output == "all" else [args.output])
            if args.workload != "c2":
                modes = [m for m in modes if m != "events"]
            res["outputs"] = output_mode_numbers(ivp_amd, prob, wl, t0, t1, y0d, pd, fp, ctx, args, modes, ms_per_step)
        if "outputs" in res and isinstance(res["outputs"].get("csr_log"), dict) and "ms_per_solve" in res["outputs"]["csr_log"]:
            # the reference's own default contract next to the end-state headline: every accepted step returned (Solution.t / .y)
            c = res["outputs"]["csr_log"]
            res["full_contract"] = {"what": "Solution.t / Solution.y of every trajectory (every accepted step, CSR, device-resident) from ONE integration",
                                    "ms_per_step": c["ms_per_solve"], "value": total_acc / (c["ms_per_solve"] * 1e-3), "unit": "steps/s",
                                    "records": c["records"], "bytes": c["bytes_written"], "vs_end_state": c["ms_per_solve"] / ms_per_step,
                                    "two_pass_ms_per_step": c["two_pass_ms_per_solve"], "passes": c["log_info"].get("passes")}
        if not args.no_cpu_baseline and world == 1 and args.workload == "c2":
            res["cpu_baseline"] = cpu_baseline(y0, p, t0, t1)
            res["accuracy"] = accuracy_vs_truth(ivp_amd, prob, mk_opts(0), ctx, dev)
        sys.stdout.flush()
        os.dup2(saved_stdout, 1)
        print(json.dumps(res), flush=True)
        os.dup2(2, 1)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def fma_mode_numbers(ivp_amd, prob, t0, t1, y0d, pd, ctx, args, sync_all):
    """Secondary, informational: the same workload in the FMA arithmetic mode (explicit fused multiply-adds at the marked
    sites, one reciprocal per primary in the right-hand side; bit-comparable with oracle/liboracle_fma.so, identical in
    every kernel variant -- tests/test_fma_mode.py); differs from strict at the 1e-16-per-operation level."""
    opts = ivp_amd.Options(method="DOPRI5", rtol=1e-6, atol=1e-9, fp_mode=ivp_amd.FpMode.FMA, chunk_attempts=args.chunk)
    out = None
    for _ in range(max(2, args.warmup)):
        out = ivp_amd.solve_ivp_batch(prob, t0, t1, y0d, pd, opts, ctx, out)
    sync_all()
    k = max(5, args.steps // 2)
    t = time.perf_counter()
    for _ in range(k):
        out = ivp_amd.solve_ivp_batch(prob, t0, t1, y0d, pd, opts, ctx, out)
    sync_all()
    dt = time.perf_counter() - t
    acc = float(out.naccpt.sum().item())
    return {"value": acc * k / dt, "unit": "steps/s", "ms_per_step": dt / k * 1e3, "steps": k, "this_rank_only": True,
            "fp_mode": "fma", "oracle": "oracle/liboracle_fma.so (bit-exact, tests/test_fma_mode.py)"}


CR3BP_EVENT_SRC = r"""
__device__ void ode(double t, const double* s, double* d, const double* p)
{
    const double mu = p[0];
    const double x = s[0], y = s[1], z = s[2], vx = s[3], vy = s[4], vz = s[5];
    const double a = x + mu, b = x - 1.0 + mu;
    const double r1 = sqrt(a * a + y * y + z * z), r2 = sqrt(b * b + y * y + z * z);
    const double r13 = r1 * r1 * r1, r23 = r2 * r2 * r2;
    d[0] = vx; d[1] = vy; d[2] = vz;
    d[3] = x + 2.0 * vy - (1.0 - mu) * (x + mu) / r13 - mu * (x - 1.0 + mu) / r23;
    d[4] = y - 2.0 * vx - (1.0 - mu) * y / r13 - mu * y / r23;
    d[5] = -(1.0 - mu) * z / r13 - mu * z / r23;
}
__device__ void events(double t, const double* s, double* g, const double* p) { g[0] = s[1]; }   // crossings of the x axis
"""


def output_mode_numbers(ivp_amd, prob, wl, t0, t1, y0d, pd, fp, ctx, args, modes, end_state_ms):
    """The device DefaultSolOut (src/solve/solout.rs:127-431) at the workload's full size: wall time per complete solve,
    bytes the mode writes (records x record size: algorithmic, what the reference pushes into its Vecs) and the HBM rate
    that corresponds to -- against 8 TB/s.  Buffers are allocated once and reused (`out=`); every solve is complete."""
    import torch
    from ivp_amd import workloads as W
    n = prob.n
    B = int(y0d.shape[1])
    base = dict(method=wl["method"], rtol=wl["rtol"], atol=wl["atol"], fp_mode=fp, chunk_attempts=args.chunk)
    k = max(3, min(10, args.steps // 2))
    t_hi = float(t1.max()) if np.ndim(t1) else float(t1)
    t1d = torch.as_tensor(t1, device=y0d.device) if np.ndim(t1) else t1

    def timed(fn):
        out = fn(None)
        out = fn(out)
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(k):
            out = fn(out)
        torch.cuda.synchronize()
        return (time.perf_counter() - t) / k * 1e3, out

    res = {"solves_timed": k, "end_state_ms_per_solve": end_state_ms, "hbm_peak_GBs": HBM_PEAK_GBS,
           "note": "bytes = records x record size (algorithmic); GB/s = bytes / wall time of the whole solve, integration included"}
    for mode in modes:
        try:
            if mode == "t_eval":
                ne = 128
                te = np.linspace(0.0, t_hi, ne)
                o = ivp_amd.Options(t_eval=te, **base)
                ms, out = timed(lambda prev: ivp_amd.solve_ivp_batch(prob, t0, t1d, y0d, pd, o, ctx, prev))
                rec = int(out.n_filled.sum().item())
                byts = rec * (n * 8 + 4)
                res[mode] = {"samples_per_trajectory": ne, "records": rec}
            elif mode == "csr_log":
                # The reference's DEFAULT contract: no t_eval => every accepted step in Solution.t / Solution.y
                # (solout.rs:387-428).  Headline form: ONE integration into the page pool + the gather kernel
                # (ivp_batch_solve_logged_device), result buffers reused like every other mode here (`out=`).  Beside it: the
                # same call allocating its result (records fetched from the pool into fresh buffers of exactly `total`), and
                # the older counted two-pass form (counting solve + scan + filling solve).
                o = ivp_amd.Options(**base)
                ms, out = timed(lambda prev: ivp_amd.solve_ivp_batch_logged(prob, t0, t1d, y0d, pd, o, ctx, out=prev))
                info = dict(out.log_info)
                rec = int(out.t_log.shape[0])
                del out
                ms_fresh, out = timed(lambda prev: ivp_amd.solve_ivp_batch_logged(prob, t0, t1d, y0d, pd, o, ctx))
                del out
                ms_two, out = timed(lambda prev: ivp_amd.solve_ivp_batch_logged(prob, t0, t1d, y0d, pd, o, ctx, two_pass=True))
                assert int(out.t_log.shape[0]) == rec
                del out
                torch.cuda.empty_cache()
                byts = rec * (n + 1) * 8
                res[mode] = {"records": rec, "form": "one integration: page pool + gather kernel (ivp_batch_solve_logged_device), result buffers reused",
                             "log_info": info, "ms_per_solve_allocating_the_result": ms_fresh,
                             "two_pass_ms_per_solve": ms_two, "two_pass_form": "counting solve + exclusive scan + filling solve (round 3)",
                             "gather_traffic_bytes": 2 * byts, "gather_hbm_floor_ms": 2 * byts / (HBM_PEAK_GBS * 1e9) * 1e3}
            elif mode == "dense":
                ml = int(out_max_log(ivp_amd, prob, t0, t1d, y0d, pd, base, ctx))
                o = ivp_amd.Options(dense_output=True, max_log=ml, **base)
                ms, out = timed(lambda prev: ivp_amd.solve_ivp_batch(prob, t0, t1d, y0d, pd, o, ctx, prev))
                nseg, nlog = int(out.n_seg.sum().item()), int(out.n_log.sum().item())
                nc = int(out.seg_cont.shape[1])
                byts = nseg * (nc + 2) * 8 + nlog * (n + 1) * 8
                res[mode] = {"max_log": ml, "segments": nseg, "log_records": nlog, "coefficients_per_segment": nc,
                             "buffer_GB": (out.seg_cont.numel() + out.y_log.numel()) * 8 / 1e9}
                del out
                torch.cuda.empty_cache()
            elif mode == "events":
                f = ivp_amd.DeviceIVP(CR3BP_EVENT_SRC, n=6, params=(W.ARENSTORF_MU,), ctx=ctx, events=[ivp_amd.EventConfig()])
                o = ivp_amd.Options(max_events=16, **base)
                ms, out = timed(lambda prev: ivp_amd.solve_ivp_batch(f, t0, t1d, y0d, pd, o, ctx, prev))
                o0 = ivp_amd.Options(**base)
                f0 = ivp_amd.DeviceIVP(CR3BP_EVENT_SRC.split("__device__ void events")[0], n=6, params=(W.ARENSTORF_MU,), ctx=ctx)
                ms0, _ = timed(lambda prev: ivp_amd.solve_ivp_batch(f0, t0, t1d, y0d, pd, o0, ctx, prev))
                rec = int(out.n_event_hits.sum().item())
                byts = rec * (n + 1) * 8
                res[mode] = {"event": "g = y (x-axis crossings), direction All, not terminal; hiprtc right-hand side", "records": rec,
                             "same_rhs_without_events_ms_per_solve": ms0, "max_events": 16}
            else:
                continue
            res[mode].update({"ms_per_solve": ms, "bytes_written": byts, "achieved_GBs": byts / (ms * 1e-3) / 1e9,
                              "frac_of_hbm_peak": byts / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                              "hbm_floor_ms": byts / (HBM_PEAK_GBS * 1e9) * 1e3})
        except Exception as e:   # noqa: BLE001
            res[mode] = {"error": repr(e)}
    return res


def out_max_log(ivp_amd, prob, t0, t1d, y0d, pd, base, ctx):
    """Longest accepted-step log of the batch (+ the initial record): the dense [max_log, ..., B] buffers are sized for it."""
    cnt = ivp_amd.solve_ivp_batch(prob, t0, t1d, y0d, pd, ivp_amd.Options(count_log=True, **base), ctx)
    return int(cnt.n_log.max().item()) + 1


def pipelined_numbers(ivp_amd, prob, t0, t1, y0d, pd, fp_mode, args, streams=4):
    """Secondary, informational: the same K complete solves with `streams` batches in flight (ivp_amd/pipeline.py).
    The headline `value` above keeps ONE batch in flight, i.e. it pays the full latency of every batch's tail."""
    from ivp_amd.pipeline import BatchPipeline
    opts = ivp_amd.Options(method="DOPRI5", rtol=1e-6, atol=1e-9, fp_mode=fp_mode, chunk_attempts=args.chunk)
    pipe = BatchPipeline(streams, y0d.device.index or 0)
    k = max(streams * 4, (args.steps // streams) * streams)
    outs = [None] * streams
    batches = [dict(t0=t0, t1=t1, y0=y0d, params=pd) for _ in range(streams)]
    res = pipe.map(prob, batches, opts)                       # warm-up, allocates the result buffers
    outs = list(res)                                          # one reusable result set per stream
    batches = [dict(t0=t0, t1=t1, y0=y0d, params=pd) for i in range(k)]
    t = time.perf_counter()
    res = pipe.map(prob, batches, opts, out_per_context=outs)
    dt = time.perf_counter() - t
    acc = float(res[-1].naccpt.sum().item())
    return {"streams": streams, "steps": k, "value": acc * k / dt, "unit": "steps/s", "ms_per_step": dt / k * 1e3,
            "note": "independent batches overlapped on separate HIP streams, driven by one host thread through "
                    "ivp_batch_submit_device / ivp_batch_poll; every solve is complete and unshared"}


def reference_crate_timing():
    """if the GPU node has a Rust toolchain, time the genuine `ivp` 0.5.1 crate.  The crate is
    not vendored here (and /root/reference does not exist on the GPU box), so this needs cargo AND a local checkout
    named by IVP_REFERENCE_CRATE; otherwise the branch reports why it was skipped."""
    import shutil
    cargo = shutil.which("cargo")
    crate = os.environ.get("IVP_REFERENCE_CRATE")
    if not cargo:
        return {"available": False, "why": "cargo not found on this host (no Rust toolchain in the image)"}
    if not crate or not os.path.isfile(os.path.join(crate, "Cargo.toml")):
        return {"available": False, "why": "cargo present but IVP_REFERENCE_CRATE does not name a checkout of Ryan-D-Gast/ivp"}
    import subprocess
    try:
        t = time.perf_counter()
        subprocess.check_call([cargo, "run", "--release", "--example", "cr3bp"], cwd=crate, stdout=subprocess.DEVNULL,
                              stderr=subprocess.DEVNULL, timeout=600)
        return {"available": True, "what": "cargo run --release --example cr3bp (one Arenstorf orbit incl. process start)",
                "wall_s": time.perf_counter() - t}
    except Exception as e:   # noqa: BLE001
        return {"available": False, "why": f"cargo run failed: {e}"}


def _cgroup_cpu_quota():
    """CPU quota of this process's cgroup in cores (cgroup v2 cpu.max / v1 cfs quota), or None when unlimited / unknown."""
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        return None if q == "max" else float(q) / float(per)
    except Exception:
        pass
    try:
        q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return None if q <= 0 else q / per
    except Exception:
        return None


def cpu_baseline(y0, p, t0, t1):
    """The CPU oracle (C restatement of the reference algorithm, libm pow, -ffp-contract=off) on this host's cores: B
    back-to-back solve_ivp calls, OpenMP over trajectories (dynamic schedule), threads bound one per core
    (OMP_PROC_BIND=spread, OMP_PLACES=cores, set before the library loads).  SURVEY.md section 8d asks for "all host cores";
    what a container may USE is its cgroup quota, not its affinity mask, so the leg sweeps 16 / 32 / 64 / 128 / all threads
    (median of three runs each after a warm-up, the whole sweep ~5 s), reports the best with its thread count and states
    the quota beside it.  The single-thread figure is measured on the first 4096 trajectories."""
    from oracle import oracle as O
    O.build()
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        cores = os.cpu_count() or 1
    cores = max(1, cores)
    n1 = 4096
    t = time.perf_counter()
    r1 = O.solve_batch("cr3bp", y0[:, :n1], p[:, :n1], t0, t1, method="DOPRI5", rtol=1e-6, atol=1e-9, threads=1)
    dt1 = time.perf_counter() - t
    single = r1["total_accepted"] / dt1
    nall = y0.shape[1]
    sweep, r = {}, None
    t_leg = time.perf_counter()
    for th in sorted({min(c, cores) for c in (16, 32, 64, 128, cores)}):
        walls = []
        for rep in range(4):   # one warm-up + three timed runs
            t = time.perf_counter()
            r = O.solve_batch("cr3bp", y0, p, t0, t1, method="DOPRI5", rtol=1e-6, atol=1e-9, threads=th)
            if rep:
                walls.append(time.perf_counter() - t)
        sweep[th] = float(np.median(walls))
        if time.perf_counter() - t_leg > 20.0:   # a slow host: the leg stays bounded
            break
    best_threads = min(sweep, key=sweep.get)
    best_dt = sweep[best_threads]
    model = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    rej = float(r["nrejct"].sum()) / max(float(r["nstep"].sum()), 1.0)
    acc = r["total_accepted"]
    quota = _cgroup_cpu_quota()
    return {
        "value": acc / best_dt, "unit": "steps/s", "cores": best_threads, "kind": "port",
        "sample": f"the same {nall} CR3BP trajectories, one solve_ivp call each, OpenMP over trajectories, OMP_PROC_BIND="
                  f"{os.environ.get('OMP_PROC_BIND')} OMP_PLACES={os.environ.get('OMP_PLACES')}; median of 3 runs after a warm-up per thread count: "
                  + "; ".join(f"{th} threads {acc / dt:.3e} steps/s" for th, dt in sorted(sweep.items()))
                  + f"; single thread on the first {n1}: {single:.3e} steps/s",
        "thread_sweep_steps_per_s": {str(th): acc / dt for th, dt in sorted(sweep.items())},
        "affinity_cpus": cores, "cgroup_cpu_quota_cores": quota,
        "all_host_cores_note": ("the affinity mask lists %d logical CPUs" % cores)
                               + (", the cgroup lets this process run %.1f of them at a time" % quota if quota else ", no cgroup CPU quota is set")
                               + ": the best thread count of the sweep is the headline",
        "single_core_value": single, "cpu_model": model, "wall_s": best_dt,
        "attempts_per_s": float(r["nstep"].sum()) / best_dt, "rejection_ratio": rej,
        "what": "oracle/ivp_oracle.c: C restatement of the reference (Rust) algorithm; the crate cannot be built here",
        "reference_crate": reference_crate_timing(),
    }


def accuracy_vs_truth(ivp_amd, prob, opts, ctx, dev):
    """End-state error of the GPU path and of the CPU oracle against the committed SciPy DOP853 @ 1e-13 truth for the
    first trajectories of the C2 batch (tests/golden/scipy_truth.json; BASELINE target: within 10x of the CPU's)."""
    import torch
    from ivp_amd import workloads as W
    from oracle import oracle as O
    truth = json.load(open(os.path.join(ROOT, "tests", "golden", "scipy_truth.json")))["truth"]["cr3bp"]
    n = int(truth["subset"])
    y0, p, t0, t1 = W.cr3bp_batch(256)
    y0, p = np.ascontiguousarray(y0[:, :n]), np.ascontiguousarray(p[:, :n])
    ref = np.asarray(truth["y_end"]).T
    r = ivp_amd.solve_ivp_batch(prob, t0, t1, torch.as_tensor(y0, device=dev), torch.as_tensor(p, device=dev),
                                ivp_amd.Options(method="DOPRI5", rtol=1e-6, atol=1e-9, fp_mode=opts.fp_mode), ctx)
    o = O.solve_batch("cr3bp", y0, p, t0, t1, method="DOPRI5", rtol=1e-6, atol=1e-9)
    eg = float(np.abs(r.y_end.cpu().numpy() - ref).max())
    ec = float(np.abs(o["y_end"] - ref).max())
    return {"subset": n, "truth": "SciPy DOP853 rtol=atol=1e-13 (tests/golden/scipy_truth.json)", "gpu_max_abs_err": eg,
            "cpu_max_abs_err": ec, "ratio": eg / ec if ec > 0 else None}


if __name__ == "__main__":
    main()
