"""TEST INFRASTRUCTURE ONLY -- counter-based synthetic tensors (numpy twin of
ominix-mlx_amd/csrc/fill.hip so CPU oracle and GPU path see bit-identical inputs).

There is no checkpoint and no network; SURVEY.md section 8d prescribes seeded
synthetic weights.  value(i) = (2*u24(i) - 1) * amp, u24 from a 32-bit avalanche
hash of (seed, i); rounded RNE to the tensor dtype.  amp = std * sqrt(3) gives the
requested standard deviation.
"""
from __future__ import annotations

import zlib

import numpy as np

from . import ref_core as rc


def name_seed(name: str, base_seed: int = 0x0C0FFEE5) -> int:
    """base seed XOR crc32(tensor name)."""
    return (base_seed ^ zlib.crc32(name.encode("utf-8"))) & 0xFFFFFFFF


def hash_u32(idx: np.ndarray, seed: int) -> np.ndarray:
    x = (idx.astype(np.uint64) * np.uint64(0x9E3779B1) + np.uint64(seed)) & np.uint64(0xFFFFFFFF)
    x ^= x >> np.uint64(16)
    x = (x * np.uint64(0x85EBCA6B)) & np.uint64(0xFFFFFFFF)
    x ^= x >> np.uint64(13)
    x = (x * np.uint64(0xC2B2AE35)) & np.uint64(0xFFFFFFFF)
    x ^= x >> np.uint64(16)
    return x.astype(np.uint32)


def uniform_pm(shape, seed: int, amp: float, offset: float = 0.0, dt: str = "bf16") -> np.ndarray:
    """offset + amp * (2*u - 1), u = (hash >> 8) * 2^-24, computed in float32 exactly as
    the device kernel does, then rounded to dt.  Returned as float32 on the dt grid."""
    n = int(np.prod(shape))
    h = hash_u32(np.arange(n, dtype=np.uint64), seed)
    u = (h >> np.uint32(8)).astype(np.float32) * np.float32(1.0 / 16777216.0)
    v = np.float32(offset) + np.float32(amp) * (np.float32(2.0) * u - np.float32(1.0))
    return rc.rnd(v.astype(np.float32), dt).reshape(shape)


def tensor(name: str, shape, std: float = 0.02, offset: float = 0.0, dt: str = "bf16",
           base_seed: int = 0x0C0FFEE5) -> np.ndarray:
    return uniform_pm(shape, name_seed(name, base_seed), std * float(np.sqrt(3.0)), offset, dt)


def prompt_ids(n: int, vocab: int) -> np.ndarray:
    """prompt = (i*7919 + 13) mod V."""
    return ((np.arange(n, dtype=np.int64) * 7919 + 13) % vocab).astype(np.uint32)
