"""The synthetic frames of synth.py generated directly in HBM (torch is only the allocator and
the integer ALU here).  Bit-identical to synth.gray8 / synth.rgb8: int64 arithmetic wraps like
uint64, and logical right shifts are emulated by masking."""
import torch

from .synth import SEED

_C1 = 0x9E3779B97F4A7C15 - (1 << 64)
_C2 = 0xBF58476D1CE4E5B9 - (1 << 64)
_C3 = 0x94D049BB133111EB - (1 << 64)


def _lsr(z, s):
    return (z >> s) & ((1 << (64 - s)) - 1)


def _splitmix64(z):
    z = z + _C1
    z = (z ^ _lsr(z, 30)) * _C2
    z = (z ^ _lsr(z, 27)) * _C3
    return z ^ _lsr(z, 31)


def _tri(t, period):
    a = (t % period) * 510 // period
    return torch.where(a <= 255, a, 510 - a)


def _hash(width, height, frame, seed, device):
    x = torch.arange(width, dtype=torch.int64, device=device)[None, :]
    y = torch.arange(height, dtype=torch.int64, device=device)[:, None]
    return _splitmix64((seed ^ (frame << 40)) ^ (y << 20) ^ x)


def gray8(width, height, frame=0, kind="S1", seed=SEED, device="cuda"):
    if kind == "S3":
        return torch.full((height, width), 128, dtype=torch.uint8, device=device)
    h = _hash(width, height, frame, seed, device)
    if kind == "S2":
        return (h & 0xFF).to(torch.uint8)
    x = torch.arange(width, dtype=torch.int64, device=device)[None, :]
    y = torch.arange(height, dtype=torch.int64, device=device)[:, None]
    base = (_tri(3 * x + 17 * frame, 1531) + _tri(5 * y, 1187)) >> 1
    return (base + (h & 7) - 3).clamp_(0, 255).to(torch.uint8)


def rgb8(width, height, frame=0, seed=SEED, device="cuda"):
    h = _hash(width, height, frame, seed, device)
    g = gray8(width, height, frame, "S1", seed, device).to(torch.int64)
    r = (g + (_lsr(h, 8) & 3) - 1).clamp_(0, 255)
    b = (g - (_lsr(h, 16) & 3) + 1).clamp_(0, 255)
    return torch.stack([r, g, b], dim=-1).to(torch.uint8)
