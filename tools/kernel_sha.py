#!/usr/bin/env python3
"""sha256 of the sources that define the stepping kernels and their launch schedule.

Every counter profile under profiles/ (SQ counters, HBM bytes) is stamped with this hash when it is collected
(tools/pmc_sq.py, tools/pmc_hbm.py); bench.py recomputes it and refuses to derive a roofline fraction from a profile of
other kernels than the ones it has just timed ("stale": true).

  python tools/kernel_sha.py        prints the hash of the working tree
"""
import hashlib
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNEL_SOURCES = ("ivp_amd/csrc/rk_core.h", "ivp_amd/csrc/rk_global.h", "ivp_amd/csrc/rk_coop.h", "ivp_amd/csrc/bdf_core.h",
                  "ivp_amd/csrc/ivp_kargs.h", "ivp_amd/csrc/ivp_capi.cpp", "ivp_amd/csrc/Makefile")


def kernel_sources_sha256(root: str = ROOT) -> str:
    h = hashlib.sha256()
    for rel in KERNEL_SOURCES:
        with open(os.path.join(root, rel), "rb") as f:
            h.update(rel.encode() + b"\0" + f.read() + b"\0")
    return h.hexdigest()


if __name__ == "__main__":
    print(kernel_sources_sha256())
