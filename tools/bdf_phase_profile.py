"""Where a BDF wave-attempt spends its shader-clock ticks, phase by phase (C5 workload).

tools/bdf_phase_profile.sh builds a copy of the library whose BDF translation unit is compiled with -DIVP_PHASE_PROF
(the IVP_PHASE(k) markers of bdf_core.h become LDS tick counters, ~120 ticks each) into exp_libs/ and runs this script
on the GPU box.  B / IVP_TUNE_BDF_LPW choose the batch and the trajectories per wave (B=256, LPW=1: a lone lane).
"""
import sys, os, ctypes as C
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ivp_amd._lib as L
L.LIB_PATH = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "exp_libs", "libivp_hip_prof.so")
import ivp_amd
from ivp_amd import workloads as W
dev = torch.device('cuda:0')
B = int(os.environ.get("B", "10000"))
y0, p, t0, t1 = W.vdp_stiff_batch(B)
opts = ivp_amd.Options(method="BDF", rtol=1e-4, atol=1e-6, profile=1, chunk_attempts=4096)
y0d = torch.as_tensor(y0, device=dev); pd = torch.as_tensor(p, device=dev)
ctx = ivp_amd.Context(0)
lib = L.load()
ticks = (C.c_ulonglong * 16)()
out = ivp_amd.solve_ivp_batch(ivp_amd.VanDerPol(), t0, t1, y0d, pd, opts, ctx)
lib.ivp_debug_phase_ticks(ticks, 1)
out = ivp_amd.solve_ivp_batch(ivp_amd.VanDerPol(), t0, t1, y0d, pd, opts, ctx, out)
lib.ivp_debug_phase_ticks(ticks, 1)
t = np.array(list(ticks), dtype=np.float64)
names = ["0 tail of previous attempt + loop", "1 step-size clamps", "2 change_d", "3 predictor/scale/psi", "4 LU refresh", "5 Newton",
         "6 error estimate", "7 accept: D update/output", "8 neighbour error norms", "9 three powers"]
att = t[15]
print("lpw", os.environ.get("IVP_TUNE_BDF_LPW"), "kernel ms", out.stats["step_kernel_ms"], "wave-attempts", att, "total ticks/attempt", t[:10].sum() / att)
for k, n in enumerate(names):
    print(f"{t[k] / att:9.1f} ticks/attempt  {100 * t[k] / t[:10].sum():5.1f} %  {n}")
