"""Synthetic coefficient fields for benchmarks: one fp64 value per fine
element from a splitmix64 stream (row-major, ex fastest), broadcast to its 4 Gauss points.
D100 = uniform [1,100] (reference-like, include/Diffusion.h:62); D1e4 = log-uniform 10^(4u)."""
import numpy as np

_GAMMA = np.uint64(0x9E3779B97F4A7C15)
_C1 = np.uint64(0xBF58476D1CE4E5B9)
_C2 = np.uint64(0x94D049BB133111EB)


def splitmix64_uniform(seed, count):
    with np.errstate(over="ignore"):
        z = np.uint64(seed) + _GAMMA * np.arange(1, count + 1, dtype=np.uint64)
        z = (z ^ (z >> np.uint64(30))) * _C1
        z = (z ^ (z >> np.uint64(27))) * _C2
        z = z ^ (z >> np.uint64(31))
    return (z >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def fill_coefficient(seed, dist, n_elems_per_side):
    """-> per-qp field [NE*NE*4] (layout 1 of slod_set_coefficient)."""
    u = splitmix64_uniform(seed, n_elems_per_side * n_elems_per_side)
    if dist == "D100":
        v = 1.0 + 99.0 * u
    elif dist == "D1e4":
        v = np.power(1.0e4, u)
    elif dist == "const":
        v = np.ones_like(u)
    else:
        raise ValueError(dist)
    return np.repeat(v, 4)
