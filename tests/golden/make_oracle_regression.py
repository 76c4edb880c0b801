"""Regression fixture of the CPU oracle's OWN outputs (fixture iii): step counts and end
states of the libm-pow build for a handful of problems, so that an accidental change of the restatement is
noticed.  These are not independent truth (see scipy_*.json for that).  Run: python tests/golden/make_oracle_regression.py"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from ivp_amd import workloads as W  # noqa: E402
from oracle import oracle as O  # noqa: E402

CASES = [
    ("arenstorf", "cr3bp", 0.0, W.ARENSTORF_PERIOD, [W.ARENSTORF_X0, 0, 0, 0, W.ARENSTORF_VY0, 0], [W.ARENSTORF_MU]),
    ("vdp", "vdp", 0.0, 100.0, [2.0, 0.0], [1.0]),
    ("lorenz", "lorenz", 0.0, 5.0, [1.0, 1.0, 1.0], [10.0, 28.0, 8.0 / 3.0]),
    ("decay", "decay", 0.0, 10.0, [1.0], [0.5]),
]
METHODS = [("RK23", 1e-4, 1e-7), ("DOPRI5", 1e-6, 1e-9), ("DOP853", 1e-8, 1e-10), ("BDF", 1e-5, 1e-8), ("RK4", None, None)]

if __name__ == "__main__":
    out = []
    for name, rhs, t0, t1, y0, p in CASES:
        for m, rt, at in METHODS:
            kw = {} if m == "RK4" else dict(rtol=rt, atol=at)
            s = O.solve_ivp(rhs, t0, t1, y0, params=p, method=m, **kw)
            out.append(dict(case=name, rhs=rhs, t0=t0, t1=t1, y0=list(map(float, y0)), params=p, method=m, rtol=rt, atol=at,
                            nfev=s.nfev, njev=s.njev, nlu=s.nlu, nstep=s.nstep, naccpt=s.naccpt, nrejct=s.nrejct,
                            status=s.status, y_end=[float(v) for v in s.y[-1]], t_end=float(s.t[-1])))
    json.dump(out, open(os.path.join(HERE, "oracle_regression.json"), "w"), indent=1)
    print("wrote", len(out), "cases")
