"""Parse hipcc -Rpass-analysis=kernel-resource-usage remarks (stdin) into one line per kernel."""
import re
import subprocess
import sys

name = sys.argv[1]
rows, cur = [], None
PATS = (("vgpr", r" VGPRs: (\d+)"), ("sspill", r"SGPRs Spill: (\d+)"), ("vspill", r"VGPRs Spill: (\d+)"),
        ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"), ("occ", r"Occupancy \[waves/SIMD\]: (\d+)"),
        ("lds", r"LDS Size \[bytes/block\]: (\d+)"))
for line in sys.stdin:
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = {"fn": m.group(1)}
        rows.append(cur)
        continue
    if cur is None:
        continue
    for key, pat in PATS:
        m = re.search(pat, line)
        if m:
            cur[key] = int(m.group(1))
names = subprocess.run(["c++filt"] + [r["fn"] for r in rows], capture_output=True, text=True).stdout.split("\n")
for r, dem in zip(rows, names):
    dem = re.sub(r"\(IvpKArgs\)$", "", dem).replace("void ", "")
    g = lambda k: r.get(k, 0)
    print("%-13s vgpr=%3d sgpr_spill=%3d vgpr_spill=%3d scratch=%5d occ=%d lds=%6d  %s"
          % (name, g("vgpr"), g("sspill"), g("vspill"), g("scratch"), g("occ"), g("lds"), dem))
