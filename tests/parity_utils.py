"""Helpers shared by the parity tests."""
import numpy as np


def first_ambiguous_iteration(fitness: np.ndarray, tau: float) -> int:
    """Index (0-based) of the first iteration whose pbest / gbest decisions are closer
    than ``tau``; ``len(fitness)`` if none.

    PSO decisions are strict '<' compares on fp32 fitness values, so two correct
    implementations whose fitness differs by rounding noise follow the same trajectory
    only while every compare has a margin above that noise.  A recorded trajectory is
    therefore compared up to this iteration, and per-step beyond it.
    """
    t_n, n = fitness.shape
    pbest = np.full(n, np.inf, dtype=np.float32)
    gbest = np.float32(np.inf)
    for t in range(t_n):
        f = fitness[t]
        fin = np.isfinite(pbest)
        if fin.any() and np.min(np.abs(f[fin] - pbest[fin])) < tau:
            return t
        pbest = np.where(f < pbest, f, pbest)
        srt = np.sort(pbest)
        best = srt[0]
        if best < gbest:
            # a second candidate that also beats the old gbest and is within tau of the winner
            if n > 1 and srt[1] < gbest and srt[1] - best < tau:
                return t
            gbest = best
    return t_n


def vq_margins(distances: np.ndarray) -> np.ndarray:
    """Gap between the smallest and the second smallest DISTINCT-code distance of every row
    of ``distances`` [n, K]; exact duplicates of the winner are not counted (they tie by
    construction and are decided by the lowest-index rule)."""
    d = np.sort(distances, axis=1)
    gap = d[:, 1:] - d[:, :1]
    gap = np.where(gap == 0, np.inf, gap)
    return gap.min(axis=1)
