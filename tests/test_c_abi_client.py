"""A plain-C client (tests/c_abi/abi_smoke.c) compiled with gcc against include/ivp_hip.h and linked to
libivp_hip.so: the header is valid C11, the boundary takes nothing but pointers and sizes, and without a HIP device
the library refuses to compute (no CPU fallback)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build(tmp_path):
    from ivp_amd import _lib
    _lib.build()
    exe = str(tmp_path / "abi_smoke")
    subprocess.check_call(["gcc", "-std=c11", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "c_abi", "abi_smoke.c"), "-o", exe,
                           "-L", os.path.join(ROOT, "ivp_amd"), "-livp_hip", "-lm",
                           "-Wl,-rpath," + os.path.join(ROOT, "ivp_amd")])
    return exe


def test_c_client_compiles_links_and_runs_without_a_gpu(tmp_path):
    exe = _build(tmp_path)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, (r.returncode, r.stdout, r.stderr)
    assert "abi v5 ok" in r.stdout or "device solve ok" in r.stdout


@pytest.mark.gpu
def test_c_client_solves_on_the_gpu(tmp_path):
    exe = _build(tmp_path)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.returncode, r.stdout, r.stderr)
    assert "device solve ok" in r.stdout
    assert "multi-context solve ok" in r.stdout   # ivp_batch_solve_multi_host: two contexts, bit-identical to one
    assert "logged solve ok" in r.stdout          # ivp_batch_solve_logged: Solution.t / Solution.y in one call, library-owned Vecs
