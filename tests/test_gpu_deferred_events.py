"""Deferred event refinement (no terminal event: the stepping kernels note crossings, event_kernel_t finds the roots afterwards;
DESIGN.md section 4) against the inline search, on the GPU, through the hiprtc path and the default launch policy (bulk kernel +
lane-cooperative tail).  The mode is a per-process library setting (IVP_TUNE_DEFER_EVENTS), so each run is a child process."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("method,B", [("DOPRI5", 20000), ("DOP853", 3000)])
def test_deferred_roots_equal_inline_roots_bit_for_bit(tmp_path, method, B):
    outs = []
    for mode in ("0", "1"):
        out = str(tmp_path / f"ev_{mode}.npz")
        env = dict(os.environ, IVP_TUNE_DEFER_EVENTS=mode)
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "helpers", "events_dump.py"), out, method, str(B)], env=env,
                           capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(np.load(out))
    a, d = outs
    assert sorted(a.files) == sorted(d.files)
    hits = a["end.n_event_hits"]
    assert hits.shape[0] == 2 and hits[0].max() > 6 and hits.sum() > 4 * B   # both events fire, the first one past its 6 slots
    for k in a.files:
        assert np.array_equal(a[k], d[k], equal_nan=True), k
