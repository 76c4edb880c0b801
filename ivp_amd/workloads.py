"""Synthetic batched workloads named by BASELINE.json.

Pure numpy: the same generators feed the GPU path, the CPU oracle, the golden-fixture script
and bench.py, so every consumer integrates bit-identical inputs.
"""
from __future__ import annotations

import numpy as np

# examples/cr3bp.rs:40-53 (Arenstorf orbit, Hairer/Norsett/Wanner)
ARENSTORF_X0 = 0.994
ARENSTORF_VY0 = -2.00158510637908252240537862224
ARENSTORF_MU = 0.012277471
ARENSTORF_PERIOD = 17.0652165601579625588917206249


def cr3bp_batch(B: int, seed: int = 20260102):
    """C2/C4: B perturbed Arenstorf orbits. Returns (y0[6,B], params[1,B], t0, t1).

    Trajectory 0 is the unperturbed orbit of examples/cr3bp.rs:40-53.
    """
    rng = np.random.default_rng(seed)
    u = rng.uniform(-1.0, 1.0, size=(3, B))
    u[:, 0] = 0.0
    y0 = np.zeros((6, B))
    y0[0] = ARENSTORF_X0 * (1.0 + 1e-4 * u[0])
    y0[4] = ARENSTORF_VY0 * (1.0 + 1e-4 * u[1])
    mu = ARENSTORF_MU * (1.0 + 1e-3 * u[2])
    return y0, mu.reshape(1, B).copy(), 0.0, ARENSTORF_PERIOD


def vdp_batch(B: int, seed: int = 20260103):
    """C3: B Van der Pol (mu=1) oscillators with per-trajectory end times in [50,100].

    Returns (y0[2,B], params[1,B], t0, t1[B]); RHS is benches/benchmark.py:22-27.
    """
    rng = np.random.default_rng(seed)
    u = rng.uniform(-1.0, 1.0, size=(3, B))
    y0 = np.zeros((2, B))
    y0[0] = 2.0 * (1.0 + 0.5 * u[0])
    y0[1] = 2.0 * u[1]
    t1 = 50.0 + 50.0 * np.abs(u[2])
    return y0, np.ones((1, B)), 0.0, t1


def vdp_stiff_batch(B: int, seed: int = 20260105):
    """C5: B stiff Van der Pol oscillators (mu ~ 1000), t in [0, 3000] (benches/benchmark.py:118-126), for BDF.

    Returns (y0[2,B], params[1,B], t0, t1).  Trajectory 0 is the benchmark's own problem (y0 = [2, 0], mu = 1000).
    """
    rng = np.random.default_rng(seed)
    y0 = np.zeros((2, B))
    y0[0] = 2.0 * (1.0 + 0.05 * rng.standard_normal(B))
    y0[1] = 0.05 * rng.standard_normal(B)
    mu = 1000.0 * (1.0 + 0.1 * rng.uniform(-1.0, 1.0, B))
    y0[:, 0] = [2.0, 0.0]
    mu[0] = 1000.0
    return y0, mu.reshape(1, B).copy(), 0.0, 3000.0


def shard_permutation(B: int, seed: int = 20260104) -> np.ndarray:
    """C4: fixed permutation applied before contiguous sharding (equalises step-count skew)."""
    return np.random.default_rng(seed).permutation(B)
