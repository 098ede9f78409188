"""Seedable, libm-free synthetic 3-D fields.

f = smooth (amplitude ~10) + medium scale (0.1) + white noise (0.01), built only from
IEEE +,-,*,/ and integer hashing so that numpy here, the C++ host generator
(csrc/wr_synth.cpp) and the HIP generator (csrc/wr_kernels.hip: k_synth) produce
bit-identical doubles on any machine.  The shape follows the reference's sample generator
examples/generic/create_in_field.f90:60-80 (10 sin x sin^2 y cos z) with polynomial / tent
stand-ins for the trigonometric factors.
"""
import numpy as np

_GOLD = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)


def splitmix64(seed, idx):
    """SplitMix64 output for state = seed + (idx+1)*GOLD (counter mode), uint64 arrays."""
    with np.errstate(over="ignore"):
        z = np.uint64(seed) + (idx.astype(np.uint64) + np.uint64(1)) * _GOLD
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        return z ^ (z >> np.uint64(31))


def _tent(t):
    return 1.0 - np.abs(2.0 * t - 1.0)


def field(nx, ny, nz, seed=12345, z0=0, z1=None):
    """float64 array shaped (z1-z0, ny, nx), x fastest; planes z0..z1-1 of the nx*ny*nz field."""
    z1 = nz if z1 is None else z1
    i = np.arange(nx, dtype=np.float64)[None, None, :]
    j = np.arange(ny, dtype=np.float64)[None, :, None]
    k = np.arange(z0, z1, dtype=np.float64)[:, None, None]
    u, v, w = i / float(nx), j / float(ny), k / float(nz)
    tv = _tent(v)
    a = 10.0 * (4.0 * u * (1.0 - u)) * (tv * tv) * (1.0 - 2.0 * w)
    u8, v8, w8 = 8.0 * u, 8.0 * v, 8.0 * w
    b = 0.1 * _tent(u8 - np.floor(u8)) * _tent(v8 - np.floor(v8)) * _tent(w8 - np.floor(w8))
    lin = (np.arange(z0, z1, dtype=np.uint64)[:, None, None] * np.uint64(ny)
           + np.arange(ny, dtype=np.uint64)[None, :, None]) * np.uint64(nx) \
        + np.arange(nx, dtype=np.uint64)[None, None, :]
    r = (splitmix64(seed, lin) >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)
    c = 0.01 * (r - 0.5)
    return (a + b) + c
