"""Synthetic frame generator fixed by BASELINE.md §2 / SURVEY.md §8(d).

Integer-only, so every language restates it identically.  Seed S = 0xFE11C5,
frame index f:

    h   = splitmix64(S ^ (f << 40) ^ (y << 20) ^ x)
    tri(t, P): a = (t mod P) * 510 // P;  a if a <= 255 else 510 - a
    S1 "natural-like": clamp(((tri(3x+17f,1531) + tri(5y,1187)) >> 1) + (h & 7) - 3, 0, 255)
    S2 "noise":        h & 0xFF
    S3 "flat":         128
    RGB (from S1): G = v, R = clamp(G + ((h >> 8) & 3) - 1), B = clamp(G - ((h >> 16) & 3) + 1)
"""
import numpy as np

SEED = 0xFE11C5
_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(z):
    z = (z + np.uint64(0x9E3779B97F4A7C15)) & _M64
    z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
    z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
    return z ^ (z >> np.uint64(31))


def _tri(t, period):
    a = (t % period) * 510 // period
    return np.where(a <= 255, a, 510 - a)


def _hash(width, height, frame, seed):
    x = np.arange(width, dtype=np.uint64)[None, :]
    y = np.arange(height, dtype=np.uint64)[:, None]
    with np.errstate(over="ignore"):
        key = np.uint64(seed) ^ (np.uint64(frame) << np.uint64(40)) ^ (y << np.uint64(20)) ^ x
        return _splitmix64(key)


def gray8(width, height, frame=0, kind="S1", seed=SEED):
    """One 8-bit grayscale frame, shape (height, width), dtype uint8."""
    if kind == "S3":
        return np.full((height, width), 128, dtype=np.uint8)
    h = _hash(width, height, frame, seed)
    if kind == "S2":
        return (h & np.uint64(0xFF)).astype(np.uint8)
    if kind != "S1":
        raise ValueError(kind)
    x = np.arange(width, dtype=np.int64)[None, :]
    y = np.arange(height, dtype=np.int64)[:, None]
    base = (_tri(3 * x + 17 * frame, 1531) + _tri(5 * y, 1187)) >> 1
    v = base + (h & np.uint64(7)).astype(np.int64) - 3
    return np.clip(v, 0, 255).astype(np.uint8)


def rgb8(width, height, frame=0, seed=SEED):
    """One 8-bit RGB frame, shape (height, width, 3), interleaved, dtype uint8."""
    h = _hash(width, height, frame, seed)
    g = gray8(width, height, frame, "S1", seed).astype(np.int64)
    r = np.clip(g + ((h >> np.uint64(8)) & np.uint64(3)).astype(np.int64) - 1, 0, 255)
    b = np.clip(g - ((h >> np.uint64(16)) & np.uint64(3)).astype(np.int64) + 1, 0, 255)
    return np.stack([r, g, b], axis=-1).astype(np.uint8)


def gray16(width, height, frame=0, seed=SEED):
    """16-bit variant used only by parity tests: S1 shape scaled to 16 bits plus hash noise."""
    h = _hash(width, height, frame, seed)
    g = gray8(width, height, frame, "S1", seed).astype(np.int64)
    v = g * 257 + ((h >> np.uint64(24)) & np.uint64(0x3FF)).astype(np.int64) - 512
    return np.clip(v, 0, 65535).astype(np.uint16)
