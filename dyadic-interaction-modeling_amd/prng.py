"""Deterministic, machine-independent tensor generator.

No checkpoints or datasets of the reference are available, so every weight and
every synthetic dyad clip is *regenerated* from integers: a counter-based
splitmix64 stream keyed by ``seed`` and the FNV-1a hash of the tensor name.  The
integer part is exact on any machine; the affine map to floats is done in float64
and rounded once to float32, so uniform tensors are bit-identical everywhere
(golden fixtures under tests/golden rely on this: the 93 MB of VQ-VAE weights are
not committed, they are regenerated on both sides).
"""
import numpy as np

_GOLD = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)
_MASK = (1 << 64) - 1


def fnv1a64(name: str) -> int:
    h = 0xCBF29CE484222325
    for b in name.encode("utf-8"):
        h ^= b
        h = (h * 0x100000001B3) & _MASK
    return h


def _mix(z: np.ndarray) -> np.ndarray:
    z = (z ^ (z >> np.uint64(30))) * _M1
    z = (z ^ (z >> np.uint64(27))) * _M2
    return z ^ (z >> np.uint64(31))


def raw_u64(seed: int, name: str, n: int, offset: int = 0) -> np.ndarray:
    """n 64-bit words of stream (seed, name), starting at counter ``offset``."""
    key = np.uint64((int(seed) ^ fnv1a64(name)) & _MASK)
    with np.errstate(over="ignore"):
        ctr = np.arange(offset + 1, offset + n + 1, dtype=np.uint64)
        return _mix(key + ctr * _GOLD)


def uniform01(seed: int, name: str, n: int, offset: int = 0) -> np.ndarray:
    """float64 in [0,1) with 24 significant bits (exactly representable in f32)."""
    return (raw_u64(seed, name, n, offset) >> np.uint64(40)).astype(np.float64) * (1.0 / (1 << 24))


def uniform(seed: int, name: str, shape, lo: float, hi: float) -> np.ndarray:
    n = int(np.prod(shape)) if len(shape) else 1
    u = uniform01(seed, name, n)
    return (lo + (hi - lo) * u).astype(np.float32).reshape(shape)


def normal(seed: int, name: str, shape) -> np.ndarray:
    """Box-Muller N(0,1) in float64, rounded once to float32."""
    n = int(np.prod(shape)) if len(shape) else 1
    m = (n + 1) // 2
    u = uniform01(seed, name, 2 * m)
    u1 = 1.0 - u[:m]            # (0,1]
    u2 = u[m:]
    r = np.sqrt(-2.0 * np.log(u1))
    z = np.concatenate([r * np.cos(2.0 * np.pi * u2), r * np.sin(2.0 * np.pi * u2)])[:n]
    return z.astype(np.float32).reshape(shape)


def exponential(seed: int, name: str, shape) -> np.ndarray:
    """Exp(1) noise for the injected-noise sampler, strictly positive."""
    n = int(np.prod(shape)) if len(shape) else 1
    u = uniform01(seed, name, n)
    e = -np.log1p(-u)           # u in [0,1) -> e in [0, inf)
    e = np.maximum(e, 2.0 ** -30)
    return e.astype(np.float32).reshape(shape)


def integers(seed: int, name: str, shape, lo: int, hi: int) -> np.ndarray:
    """ints uniform in [lo, hi)."""
    n = int(np.prod(shape)) if len(shape) else 1
    r = raw_u64(seed, name, n) >> np.uint64(11)
    return (lo + (r % np.uint64(hi - lo)).astype(np.int64)).reshape(shape)
