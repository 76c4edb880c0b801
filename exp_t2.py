import time,sys,os
sys.path.insert(0,os.getcwd())
import torch, numpy as np, ivp_amd, warnings; from ivp_amd import workloads as W
warnings.simplefilter("ignore")
src=open("tests/helpers/events_dump.py").read(); SRC=src.split('SRC = r"""')[1].split('"""')[0]
ONE=SRC.replace("g[0] = s[1]; g[1] = s[0];","g[0] = s[1];")
y0,p,t0,t1=W.cr3bp_batch(20000); dev=torch.device("cuda:0")
yd,pd=torch.as_tensor(y0, device=dev), torch.as_tensor(p, device=dev)
def run(tag, f, **kw):
    o = ivp_amd.Options(method="DOPRI5", rtol=1e-6, atol=1e-9, profile=True, **kw)
    for rep in range(2):
        t=time.time(); r = ivp_amd.solve_ivp_batch(f, t0, t1, yd, pd, o); torch.cuda.synchronize(); dt=time.time()-t
    st=r.stats
    print(tag, round(dt,3), "s; launches", st["launches"], "coop", st["coop_launches"], "kernel ms", round(st["step_kernel_ms"],2), "coop ms", round(st["coop_kernel_ms"],2), "hits", int(r.n_event_hits.sum()), "max nstep", int(r.nstep.max()), flush=True)
f1=ivp_amd.DeviceIVP(ONE, n=6, params=(W.ARENSTORF_MU,), events=[ivp_amd.EventConfig()])
run("1 event, max_events 16", f1, max_events=16)
run("1 event, max_events 6", f1, max_events=6)
f2=ivp_amd.DeviceIVP(SRC, n=6, params=(W.ARENSTORF_MU,), events=[ivp_amd.EventConfig(), ivp_amd.EventConfig()])
run("2 events all dirs, max_events 16", f2, max_events=16)
f3=ivp_amd.DeviceIVP(SRC, n=6, params=(W.ARENSTORF_MU,), events=[ivp_amd.EventConfig(), ivp_amd.EventConfig().negative()])
run("2 events, second negative, 16", f3, max_events=16)
run("2 events, second negative, 6", f3, max_events=6)
