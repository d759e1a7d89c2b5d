"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the MLX random generator.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
anything under oracle/.  The shipped product path never does.

The reference's seeded unit tests (mlx-rs/src/fast.rs:231-298,
mlx-rs/src/nn/{positional_encoding.rs:432-462, normalization.rs:666-733,
activation.rs:1156-1180,1291-1320, linear.rs:224-252}) draw their inputs from
MLX's global RNG.  MLX core (ml-explore/mlx v0.30.1, fetched at build time by
mlx-rs/mlx-sys/src/mlx-c/CMakeLists.txt:35-39) is NOT vendored in the reference
tree, so this file restates the published algorithm (Threefry-2x32, 20 rounds,
JAX-compatible key splitting) and is pinned by reproducing the *input*
statistics those reference tests assert (tests/test_oracle_kats.py).

Entry points mirror mlx-rs/src/random.rs:
    seed(s)                  random.rs:88-91
    key(s)                   random.rs:98-100
    uniform(lo, hi, shape)   random.rs (uniform::<_, f32>)
"""
from __future__ import annotations

import numpy as np

_M32 = np.uint64(0xFFFFFFFF)
_ROT = ((13, 15, 26, 6), (17, 29, 16, 24))


def _rotl(x: np.ndarray, r: int) -> np.ndarray:
    return ((x << np.uint64(r)) | (x >> np.uint64(32 - r))) & _M32


def threefry2x32(key, c0: np.ndarray, c1: np.ndarray):
    """Threefry-2x32, 20 rounds; all arithmetic mod 2^32 (held in uint64 lanes)."""
    k0 = np.uint64(key[0])
    k1 = np.uint64(key[1])
    ks = (k0, k1, np.uint64(0x1BD11BDA) ^ k0 ^ k1)
    x0 = (c0.astype(np.uint64) + ks[0]) & _M32
    x1 = (c1.astype(np.uint64) + ks[1]) & _M32
    for g in range(5):
        for r in _ROT[g % 2]:
            x0 = (x0 + x1) & _M32
            x1 = _rotl(x1, r)
            x1 = x1 ^ x0
        x0 = (x0 + ks[(g + 1) % 3]) & _M32
        x1 = (x1 + ks[(g + 2) % 3] + np.uint64(g + 1)) & _M32
    return x0.astype(np.uint32), x1.astype(np.uint32)


def bits(key, n: int) -> np.ndarray:
    """n uint32 words. Layout per SURVEY Appendix C (odd n puts the unpaired word in the middle)."""
    out = np.empty(n, dtype=np.uint32)
    h = n // 2
    if n % 2 == 0:
        i = np.arange(h, dtype=np.uint64)
        a, b = threefry2x32(key, i, i + np.uint64(h))
        out[:h] = a
        out[h:] = b
    else:
        i = np.arange(h, dtype=np.uint64)
        a, b = threefry2x32(key, i, i + np.uint64(h + 1))
        out[:h] = a
        out[h + 1:] = b
        m, _ = threefry2x32(key, np.array([h], dtype=np.uint64), np.array([0], dtype=np.uint64))
        out[h] = m[0]
    return out


def key(seed_value: int):
    s = int(seed_value) & 0xFFFFFFFFFFFFFFFF
    return (np.uint32(s >> 32), np.uint32(s & 0xFFFFFFFF))


def split2(k):
    b = bits(k, 4)
    return (b[0], b[1]), (b[2], b[3])


class _State:
    state = key(0)


def seed(s: int) -> None:
    _State.state = key(s)


def _next_key():
    k0, k1 = split2(_State.state)
    _State.state = k0
    return k1


def uniform(lo: float, hi: float, shape, k=None) -> np.ndarray:
    """float32 uniform in [lo, hi), row-major fill."""
    if k is None:
        k = _next_key()
    n = int(np.prod(shape))
    b = bits(k, n)
    u = b.astype(np.float32) / np.float32(4294967295.0)
    u = np.minimum(u, np.nextafter(np.float32(1.0), np.float32(0.0)))
    lo32, hi32 = np.float32(lo), np.float32(hi)
    return (lo32 + (hi32 - lo32) * u).astype(np.float32).reshape(shape)


def split(k, num: int = 2):
    """mlx-rs/src/random.rs:103-115 (`split_device` -> mlx_random_split_num): `num` sub-keys, row i = words
    (2i, 2i+1) of a 2*num-word draw from `k`."""
    b = bits(k, 2 * num)
    return [(b[2 * i], b[2 * i + 1]) for i in range(num)]


def _log32(x: np.ndarray) -> np.ndarray:
    """float32 log, correctly rounded (MLX evaluates log as a float32 primitive; its last-bit behaviour is
    the platform's libm / simd routine, so the oracle and the HIP kernel both take the correctly rounded value)."""
    with np.errstate(divide="ignore"):
        return np.log(x.astype(np.float64)).astype(np.float32)


def gumbel(shape, k=None) -> np.ndarray:
    """mlx-rs/src/random.rs:397-414 (`gumbel_device` -> mlx_random_gumbel): -log(-log(uniform(0,1))), every
    primitive in float32.  KAT: random.rs:690-694 (key 0 -> 0.13)."""
    u = uniform(0.0, 1.0, shape, k)
    return (-_log32(-_log32(u))).astype(np.float32)


def categorical(logits: np.ndarray, k=None, num_samples=None) -> np.ndarray:
    """mlx-rs/src/random.rs:456-497 (`categorical_device`, axis -1): argmax(logits + gumbel) with the noise
    drawn in the shape [..., V] (or [..., V, num_samples]); logits are promoted to float32 by the add.
    KATs: random.rs:697-718 (`test_logits`, `test_logits_count`)."""
    lg = np.asarray(logits).astype(np.float32)
    if num_samples is None:
        g = gumbel(lg.shape, k)
        return np.argmax((lg + g).astype(np.float32), axis=-1).astype(np.uint32)
    g = gumbel(lg.shape + (int(num_samples),), k)
    return np.argmax((lg[..., None] + g).astype(np.float32), axis=-2).astype(np.uint32)


class RandomState:
    """mlx-rs/src/random.rs:21-41: the key sequence behind `key = None` (`seed`, then one `split` per draw)."""

    def __init__(self, seed_value: int):
        self.state = key(seed_value)

    def next(self):
        k0, k1 = split(self.state, 2)
        self.state = k0
        return k1


_ERFINV_TAIL = (3.03697567e-10, 2.93243101e-8, 1.22150334e-6, 2.84108955e-5, 3.93552968e-4, 3.02698812e-3, 4.83185798e-3,
                -2.64646143e-1, 8.40016484e-1)
_ERFINV_CORE = (5.43877832e-9, 1.43285448e-7, 1.22774793e-6, 1.12963626e-7, -5.61530760e-5, -1.47697632e-4, 2.31468678e-3,
                1.15392581e-2, -2.32015476e-1, 8.86226892e-1)


def erfinv32(a: np.ndarray) -> np.ndarray:
    """float32 inverse error function: the single-precision polynomial of M. Giles as published by N. Juffa (two
    branches on t = log(1 - a^2), Horner with fused multiply-adds, <= 2.4 ulp) -- the form MLX v0.30.1 evaluates
    for `erfinv`.  MLX core is not vendored in the reference, so this is pinned by the reference's `test_normal` KAT
    (random.rs:582-586) and by scipy.special.erfinv to 3 ulp (tests/test_oracle_kats.py)."""
    a = np.asarray(a, np.float32)
    a64 = a.astype(np.float64)
    t = (a64 * (0.0 - a64) + 1.0).astype(np.float32)      # fmaf(a, -a, 1): one rounding
    t = _log32(t)

    def horner(coef):
        p = np.full(a.shape, np.float32(coef[0]), np.float32)
        for c in coef[1:]:
            p = (p.astype(np.float64) * t.astype(np.float64) + np.float64(np.float32(c))).astype(np.float32)   # fmaf
        return p
    with np.errstate(invalid="ignore", over="ignore"):
        p = np.where(np.abs(t) > np.float32(6.125), horner(_ERFINV_TAIL), horner(_ERFINV_CORE))
    return (a * p).astype(np.float32)


def normal(shape, k=None, loc: float = 0.0, scale: float = 1.0) -> np.ndarray:
    """mlx-rs/src/random.rs:186-212 (`normal_device` -> mlx_random_normal): sqrt(2) * erfinv(uniform(nextafter(-1, 0), 1)),
    float32; then * scale + loc when given.  KAT: random.rs:582-586 (key 0 -> -0.20)."""
    u = uniform(np.nextafter(np.float32(-1.0), np.float32(0.0)), 1.0, shape, k)
    z = (np.float32(np.sqrt(2.0)) * erfinv32(u)).astype(np.float32)
    if scale != 1.0:
        z = (z * np.float32(scale)).astype(np.float32)
    if loc != 0.0:
        z = (z + np.float32(loc)).astype(np.float32)
    return z
