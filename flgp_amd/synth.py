"""Seeded synthetic point clouds for the configurations in BASELINE.json / SURVEY.md §8(d).

The generator is a SplitMix64 counter hash -> uniform(0,1) -> Box-Muller, written in
vectorised numpy so the same arrays can be fed to the HIP path, the CPU oracle and the
bench on any box (no dependence on numpy's global RNG or its version).

All matrices are returned column-major (Fortran order) float64, the layout R and the
C-ABI use (include/flgp_hip.h).
"""
from __future__ import annotations

import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def splitmix64(counter: np.ndarray) -> np.ndarray:
    """SplitMix64 output function applied to a uint64 counter array."""
    with np.errstate(over="ignore"):
        z = counter.astype(np.uint64) + np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def _stream_base(seed: int, stream: int) -> np.uint64:
    with np.errstate(over="ignore"):
        return splitmix64(np.array([np.uint64(seed) * np.uint64(0x100000001B3) + np.uint64(stream)], dtype=np.uint64))[0]


def uniform(seed: int, stream: int, count: int, offset: int = 0) -> np.ndarray:
    """count uniforms in (0,1): element i depends only on (seed, stream, offset+i)."""
    base = _stream_base(seed, stream)
    with np.errstate(over="ignore"):
        ctr = np.arange(offset, offset + count, dtype=np.uint64) + base
    bits = splitmix64(ctr) >> np.uint64(11)
    return (bits.astype(np.float64) + 0.5) * (1.0 / 9007199254740992.0)


def normal(seed: int, stream: int, count: int, offset: int = 0) -> np.ndarray:
    """count standard normals (Box-Muller on uniform pairs 2i, 2i+1)."""
    u = uniform(seed, stream, 2 * count, 2 * offset)
    rad = np.sqrt(-2.0 * np.log(u[0::2]))
    return rad * np.cos(2.0 * np.pi * u[1::2])


def gaussian_mixture(n: int, d: int, components: int = 16, center_scale: float = 2.0,
                     seed: int = 20241022, row_offset: int = 0) -> np.ndarray:
    """C3/C4 workload: equal-weight Gaussian mixture, centres ~ N(0, center_scale^2 I),
    unit component covariance.  Rows [row_offset, row_offset+n) of the global cloud, so a
    rank can generate only its own row block."""
    centers = (center_scale * normal(seed, 1, components * d)).reshape(components, d)
    comp = np.minimum((uniform(seed, 2, n, row_offset) * components).astype(np.int64), components - 1)
    noise = normal(seed, 3, n * d, row_offset * d).reshape(n, d)
    return np.asfortranarray(centers[comp] + noise)


def swiss_roll(n: int, seed: int = 20241022):
    """C2 workload: (u cos u, h, u sin u), standardised per column (README.md:54-55 style);
    y = sin(u) + 0.1 N(0,1)."""
    u = 1.5 * np.pi + 3.0 * np.pi * uniform(seed, 11, n)
    h = 21.0 * uniform(seed, 12, n)
    X = np.stack([u * np.cos(u), h, u * np.sin(u)], axis=1)
    X = (X - X.mean(0)) / X.std(0, ddof=1)
    X = X / np.sqrt(X.shape[1])
    y = np.sin(u) + 0.1 * normal(seed, 13, n)
    return np.asfortranarray(X), y


def torus(n: int = 4800, seed: int = 1234):
    """C1 workload: the README's six concentric circles (README.md:44-55) with this RNG."""
    n_each = n // 6
    th = 2.0 * np.pi * uniform(seed, 21, n)
    X = np.stack([np.cos(th), np.sin(th)], axis=1)
    Y = np.zeros(n)
    for i in range(6):
        sl = slice(i * n_each, (i + 1) * n_each)
        X[sl] *= 0.5 + 0.1 * i
        Y[sl] = 1.0 if i % 2 == 0 else 0.0
    X = (X - X.mean(0)) / X.std(0, ddof=1)
    X = X / np.sqrt(X.shape[1])
    return np.asfortranarray(X), Y


def random_anchor_rows(n: int, s: int, seed: int = 20241022) -> np.ndarray:
    """s distinct row indices by a seeded partial Fisher-Yates (the semantics of
    subsample="random", src/Utils.cpp:46-48)."""
    perm = np.arange(n, dtype=np.int64)
    u = uniform(seed, 31, s)
    for i in range(s):
        j = i + int(u[i] * (n - i))
        perm[i], perm[j] = perm[j], perm[i]
    return perm[:s].copy()


def anchors_from_rows(X: np.ndarray, rows: np.ndarray) -> np.ndarray:
    return np.asfortranarray(X[rows, :])
