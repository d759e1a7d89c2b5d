"""TEST INFRASTRUCTURE ONLY -- CPU (numpy) restatement of the mlx-rs-core hot path.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
anything under oracle/.  The product path (ominix-mlx_amd/) never does and
fails loudly when the HIP library is missing.

What this restates (reference file:line in each function):
  a1 scaled_dot_product_attention   mlx-rs-core/src/utils.rs:191-209, mlx-rs/src/fast.rs:121-151
  a2 KVCache / ConcatKeyValueCache  mlx-rs-core/src/cache.rs:44-194
  a3 RoPE                           mlx-rs-core/src/utils.rs:52-97, mlx-rs/src/fast.rs:15-46
  a4 RMSNorm / LayerNorm            mlx-rs/src/fast.rs:165-219, mlx-rs/src/nn/normalization.rs
  a5 Linear                         mlx-rs/src/nn/linear.rs:87-92
  a8 fused_swiglu                   mlx-rs-core/src/metal_kernels.rs:11-18
  a9 fused_modulate                 mlx-rs-core/src/metal_kernels.rs:28-94
  a10 sampler (greedy)              mlx-rs-core/src/sampler.rs:9-18

The arithmetic of a1/a3/a4/a5 lives in MLX core v0.30.1 (ml-explore/mlx, fetched
by mlx-rs/mlx-sys/src/mlx-c/CMakeLists.txt:35-39) which is absent from the
reference tree; the formulas here are its published semantics and are PINNED for
rope / rms_norm / layer_norm / silu / softmax / linear / matmul by the
reference's own seeded known-answer tests (tests/test_oracle_kats.py).
PARITY UNPINNED for: SDPA, KV cache, fused_swiglu, fused_modulate, sampler
tie-break (the reference holds no value test for them).

Number model: every op takes/returns float32 numpy arrays whose values lie on
the grid of the logical dtype ("bf16" | "f16" | "f32").  Accumulations are done
in float64 (order-free best estimate of MLX's fp32 accumulation) and the result
is rounded ONCE to the logical dtype, which is where MLX rounds (op output).
"""
from __future__ import annotations

import math
from typing import Optional, Tuple

import numpy as np

# --------------------------------------------------------------------------
# dtype grid helpers
# --------------------------------------------------------------------------


def bf16_round(x) -> np.ndarray:
    """Round-to-nearest-even float32 -> bfloat16 grid, returned as float32."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    u = x.view(np.uint32).astype(np.uint64)
    r = ((u >> np.uint64(16)) & np.uint64(1)) + np.uint64(0x7FFF)
    out = ((u + r) & np.uint64(0xFFFF0000)).astype(np.uint32)
    nan = np.isnan(x)
    if nan.any():
        out = np.where(nan, np.uint32(0x7FC00000), out)
    return out.view(np.float32).reshape(x.shape)


def to_bf16_bits(x) -> np.ndarray:
    """float32 (already on the bf16 grid or not) -> uint16 bf16 bit patterns (RNE)."""
    return (bf16_round(x).view(np.uint32) >> np.uint32(16)).astype(np.uint16)


def from_bf16_bits(b) -> np.ndarray:
    b = np.ascontiguousarray(b, dtype=np.uint16)
    return (b.astype(np.uint32) << np.uint32(16)).view(np.float32).reshape(b.shape)


def rnd(x, dt: str) -> np.ndarray:
    if dt == "bf16":
        return bf16_round(x)
    if dt == "f16":
        return np.asarray(x, dtype=np.float32).astype(np.float16).astype(np.float32)
    if dt == "f32":
        return np.asarray(x, dtype=np.float64).astype(np.float32) if np.asarray(x).dtype == np.float64 \
            else np.asarray(x, dtype=np.float32)
    raise ValueError(f"unknown dtype {dt}")


# --------------------------------------------------------------------------
# a4  RMSNorm / LayerNorm      mlx-rs/src/fast.rs:165-219
# --------------------------------------------------------------------------


def rms_norm(x, weight: Optional[np.ndarray], eps: float, dt: str = "f32") -> np.ndarray:
    """y = w * x * rsqrt(mean(x^2) + eps) over the last axis (fast.rs:171-179,
    nn/normalization.rs:262-270).  KAT: fast.rs:254-273 (seed 103)."""
    x64 = np.asarray(x, dtype=np.float64)
    ms = np.mean(x64 * x64, axis=-1, keepdims=True)
    y = x64 / np.sqrt(ms + np.float64(np.float32(eps)))
    if weight is not None:
        y = y * np.asarray(weight, dtype=np.float64)
    return rnd(y, dt)


def layer_norm(x, weight, bias, eps: float, dt: str = "f32") -> np.ndarray:
    """(x-mu) * rsqrt(var+eps) * w + b, biased variance (fast.rs:204-218).
    KAT: fast.rs:277-298 (seed 635)."""
    x64 = np.asarray(x, dtype=np.float64)
    mu = np.mean(x64, axis=-1, keepdims=True)
    var = np.mean((x64 - mu) ** 2, axis=-1, keepdims=True)
    y = (x64 - mu) / np.sqrt(var + np.float64(np.float32(eps)))
    if weight is not None:
        y = y * np.asarray(weight, dtype=np.float64)
    if bias is not None:
        y = y + np.asarray(bias, dtype=np.float64)
    return rnd(y, dt)


# --------------------------------------------------------------------------
# a3  RoPE      mlx-rs/src/fast.rs:15-46 ; nn/positional_encoding.rs:112-137
# --------------------------------------------------------------------------


def rope(x, dims: int, traditional: bool, base, scale: float, offset: int, dt: str = "f32", freqs=None) -> np.ndarray:
    """Rotary embedding on [..., T, D]; position axis = -2, rotates the first
    `dims` features.  Non-traditional pairs (i, i+dims/2); traditional pairs
    (2i, 2i+1).  theta_i = base^(-2i/dims), or 1 / freqs[i] when custom `freqs`
    (float32 [dims/2]) replace the base (fast.rs:15-46).  KAT: fast.rs:232-250 (seed 71)."""
    x = np.asarray(x, dtype=np.float32)
    T, D = x.shape[-2], x.shape[-1]
    half = dims // 2
    i = np.arange(half, dtype=np.float64)
    if freqs is not None:
        assert base is None, "rope: exactly one of base and freqs"
        inv_freq = 1.0 / np.asarray(freqs, dtype=np.float32).astype(np.float64)
    else:
        inv_freq = np.exp(-i * (math.log(float(np.float32(base))) / half))
    pos = (np.arange(T, dtype=np.float64) + float(offset)) * float(np.float32(scale))
    ang = pos[:, None] * inv_freq[None, :]          # [T, half]
    c, s = np.cos(ang), np.sin(ang)
    out = np.asarray(x, dtype=np.float64).copy()
    if traditional:
        x1 = out[..., 0:dims:2].copy()
        x2 = out[..., 1:dims:2].copy()
        out[..., 0:dims:2] = x1 * c - x2 * s
        out[..., 1:dims:2] = x1 * s + x2 * c
    else:
        x1 = out[..., :half].copy()
        x2 = out[..., half:dims].copy()
        out[..., :half] = x1 * c - x2 * s
        out[..., half:dims] = x1 * s + x2 * c
    return rnd(out, dt)


def initialize_rope(dims: int, base: float, traditional: bool, scaling_config: Optional[dict]):
    """mlx-rs-core/src/utils.rs:52-97 -> (dims, traditional, base, scale).  Only
    `default` / `linear` types are accepted; anything else raises like utils.rs:96."""
    rope_type = "default"
    if scaling_config is not None:
        rope_type = scaling_config.get("type", scaling_config.get("rope_type", "default"))
    if rope_type in ("default", "linear"):
        scale = 1.0
        if rope_type == "linear":
            if "factor" not in scaling_config:
                raise ValueError('key "factor" is not found in scaling config')
            try:
                scale = 1.0 / float(scaling_config["factor"])
            except (TypeError, ValueError):
                raise ValueError('key "factor" is not a valid float')
        return dict(dims=dims, traditional=traditional, base=base, scale=scale)
    raise ValueError(f"Unsupported RoPE type {rope_type!r}")


# --------------------------------------------------------------------------
# a5  Linear      mlx-rs/src/nn/linear.rs:87-92
# --------------------------------------------------------------------------


def linear(x, w, b=None, dt: str = "f32") -> np.ndarray:
    """y = x @ W^T (+ b), W:[out,in].  KATs: nn/linear.rs:224-252 (seed 744),
    ops/arithmetic.rs:1921-1937 (matmul)."""
    y = np.asarray(x, dtype=np.float64) @ np.asarray(w, dtype=np.float64).T
    if b is not None:
        # addmm (linear.rs:88-90): one fused op, rounded once
        y = y + np.asarray(b, dtype=np.float64)
    return rnd(y, dt)


def matmul(a, b, dt: str = "f32") -> np.ndarray:
    return rnd(np.asarray(a, dtype=np.float64) @ np.asarray(b, dtype=np.float64), dt)


# --------------------------------------------------------------------------
# elementwise: silu (nn/activation.rs:876-880), softmax, fused kernels
# --------------------------------------------------------------------------


def sigmoid(x, dt: str = "f32") -> np.ndarray:
    x64 = np.asarray(x, dtype=np.float64)
    return rnd(1.0 / (1.0 + np.exp(-x64)), dt)


def silu(x, dt: str = "f32") -> np.ndarray:
    """nn::silu = x * sigmoid(x) (nn/activation.rs:876-880; compiled closure: each
    primitive's result is held in the tensor dtype).  KAT: activation.rs:1291-1320 (seed 22)."""
    x32 = np.asarray(x, dtype=np.float32)
    return rnd(np.asarray(x32, dtype=np.float64) * np.asarray(sigmoid(x32, dt), dtype=np.float64), dt)


def multiply(a, b, dt: str = "f32") -> np.ndarray:
    return rnd(np.asarray(a, dtype=np.float64) * np.asarray(b, dtype=np.float64), dt)


def add(a, b, dt: str = "f32") -> np.ndarray:
    return rnd(np.asarray(a, dtype=np.float64) + np.asarray(b, dtype=np.float64), dt)


def softmax(x, axis: int = -1, dt: str = "f32") -> np.ndarray:
    """softmax with fp32-or-better internal precision (precise=true), output rounded
    to dt.  KAT: nn/activation.rs:1156-1180 (seed 853, rows sum to 1)."""
    x64 = np.asarray(x, dtype=np.float64)
    m = np.max(x64, axis=axis, keepdims=True)
    e = np.exp(x64 - m)
    return rnd(e / np.sum(e, axis=axis, keepdims=True), dt)


def fused_swiglu(x, gate, dt: str = "f32") -> np.ndarray:
    """mlx-rs-core/src/metal_kernels.rs:11-18: out = gate/(1+exp(-gate)) * x, argument
    order (up, gate) at every call site.  The Metal kernel evaluates in T; the
    single-rounding result below differs from a per-op-rounded evaluation by <= 1 ulp
    of T and is the value the tests allow 2 ulp against."""
    g = np.asarray(gate, dtype=np.float64)
    return rnd(g / (1.0 + np.exp(-g)) * np.asarray(x, dtype=np.float64), dt)


def fused_modulate(x, shift, scale, eps: float = 1e-6, dt: str = "f32") -> np.ndarray:
    """The MATHEMATICAL function of metal_kernels.rs:28-94 -- per row mean, var = E[x^2]-mean^2 clamped >= 0,
    (1+scale) * (x-mean)*rsqrt(var+eps) + shift, shift/scale [B,H] broadcast over the sequence -- evaluated in float64
    with ONE rounding to `dt`.  This is what the product kernel (omx_fused_modulate: fp32 statistics) is held to.

    It is NOT a rounding-faithful restatement of the Metal kernel, which (a) accumulates the statistics in T (a bf16 sum
    of 3072 values is good to a few per cent) and (b) indexes scale[i] / shift[i] with the column alone, i.e. reads
    batch 0's modulation for every batch row (:90-91).  `fused_modulate_metal_literal` below restates exactly that;
    tests/test_oracle_kats.py shows the two agree for f32 / B == 1 and how far the T-precision statistics move a bf16
    result.  The kernel has no caller in the reference workspace (the DiT uses LayerNorm + modulate
    in separate ops), so no reference-visible output depends on (a) or (b)."""
    x64 = np.asarray(x, dtype=np.float64)
    mean = np.mean(x64, axis=-1, keepdims=True)
    var = np.maximum(np.mean(x64 * x64, axis=-1, keepdims=True) - mean * mean, 0.0)
    norm = (x64 - mean) / np.sqrt(var + np.float64(np.float32(eps)))
    sh = np.asarray(shift, dtype=np.float64)
    sc = np.asarray(scale, dtype=np.float64)
    if x64.ndim == 3 and sh.ndim == 2:
        sh, sc = sh[:, None, :], sc[:, None, :]
    return rnd((1.0 + sc) * norm + sh, dt)


def fused_modulate_metal_literal(x, shift, scale, dt: str = "f32", threads: int = 256) -> np.ndarray:
    """metal_kernels.rs:28-94 line by line, every intermediate held in T (`dt`): 256 threads stride the row accumulating
    local_sum / local_sum_sq in T (:44-52), the tree reduction 128..1 adds in T (:59-74), mean = sum / T(dim),
    var = max(sum_sq / T(dim) - mean*mean, 0), inv_std = rsqrt(var + T(1e-6)) (:78-84), out = (T(1) + scale[i]) *
    ((x - mean) * inv_std) + shift[i] with scale / shift indexed by the COLUMN ONLY (:87-92) -- rows of every batch
    element read the first `dim` entries of the flattened shift / scale.  x [..., dim]; shift, scale any shape with at
    least `dim` elements."""
    x = np.asarray(x, dtype=np.float32)
    dim = x.shape[-1]
    rows = x.reshape(-1, dim)
    sh = np.asarray(shift, dtype=np.float32).reshape(-1)[:dim].astype(np.float64)
    sc = np.asarray(scale, dtype=np.float32).reshape(-1)[:dim].astype(np.float64)
    r = lambda v: rnd(np.asarray(v, dtype=np.float64), dt).astype(np.float64)   # one operation, rounded to T
    out = np.empty_like(rows)
    for ri in range(rows.shape[0]):
        row = r(rows[ri])
        ls, lq = np.zeros(threads), np.zeros(threads)
        for i0 in range(0, dim, threads):                      # every thread's next element, in lock step
            v = np.zeros(threads)
            n = min(threads, dim - i0)
            v[:n] = row[i0:i0 + n]
            ls[:n] = r(ls[:n] + v[:n])
            lq[:n] = r(lq[:n] + r(v[:n] * v[:n]))
        half = threads // 2
        while half >= 1:
            ls[:half] = r(ls[:half] + ls[half:2 * half])
            lq[:half] = r(lq[:half] + lq[half:2 * half])
            half //= 2
        mean = r(ls[0] / r(float(dim)))
        var = np.maximum(r(r(lq[0] / r(float(dim))) - r(mean * mean)), 0.0)
        inv_std = r(1.0 / np.sqrt(r(var + r(1e-6))))
        normalized = r(r(row - mean) * inv_std)
        out[ri] = r(r(r(1.0 + sc) * normalized) + sh)
    return out.reshape(x.shape).astype(np.float32)


# --------------------------------------------------------------------------
# masks      mlx-rs-core/src/utils.rs:134-188
# --------------------------------------------------------------------------


def create_causal_mask(N: int, offset: int = 0, window_size: Optional[int] = None) -> np.ndarray:
    """utils.rs:134-153: bool [N, offset+N]; mask = l >= r (& l <= r + window)."""
    rinds = np.arange(offset + N)[None, :]
    linds = np.arange(offset, offset + N)[:, None]
    mask = linds >= rinds
    if window_size is not None:
        mask = mask & (linds <= rinds + window_size)
    return mask


def create_attention_mask(T: int, cache_offset: Optional[int], cache_max_size: Optional[int] = None,
                          return_array: Optional[bool] = None):
    """utils.rs:156-188.  Returns None (T==1), the string "causal", or a bool array."""
    ra = bool(return_array) if return_array is not None else False
    if T > 1:
        offset, window = 0, None
        if cache_offset is not None:
            offset = cache_offset
            if cache_max_size is not None:
                window = cache_max_size
                offset = min(offset, window)
                ra = ra or (offset + T) > window
        if ra:
            return create_causal_mask(T, offset, window)
        return "causal"
    return None


# --------------------------------------------------------------------------
# a1  SDPA      mlx-rs/src/fast.rs:121-151
# --------------------------------------------------------------------------


def scaled_dot_product_attention(q, k, v, scale: float, mask=None, dt: str = "f32") -> np.ndarray:
    """O = softmax(scale * Q K^T (+mask)) V.  q [B,H,Tq,D]; k,v [B,Hkv,Tk,D]; GQA by
    head group (H/Hkv consecutive q heads share a kv head, fast.rs:118).  mask: None |
    "causal" (bottom-right aligned when Tq<Tk) | bool array (keep where true) | float
    array (additive), broadcastable to [B,H,Tq,Tk].  Scores/softmax in fp32-or-better
    (fast.rs:116); output rounded to dt."""
    q64 = np.asarray(q, dtype=np.float64)
    k64 = np.asarray(k, dtype=np.float64)
    v64 = np.asarray(v, dtype=np.float64)
    B, H, Tq, D = q64.shape
    Hkv, Tk = k64.shape[1], k64.shape[2]
    assert H % Hkv == 0
    g = H // Hkv
    kk = np.repeat(k64, g, axis=1)
    vv = np.repeat(v64, g, axis=1)
    s = np.einsum("bhqd,bhkd->bhqk", q64 * float(np.float32(scale)), kk)
    if isinstance(mask, str):
        if mask == "causal":
            qi = np.arange(Tq)[:, None] + (Tk - Tq)
            ki = np.arange(Tk)[None, :]
            s = np.where(qi >= ki, s, -np.inf)
        elif mask != "":
            raise ValueError(f"Invalid mask mode {mask!r}")
    elif mask is not None:
        m = np.asarray(mask)
        if m.dtype == np.bool_:
            s = np.where(m, s, -np.inf)
        else:
            s = s + m.astype(np.float64)
    smax = np.max(s, axis=-1, keepdims=True)
    e = np.exp(s - smax)
    p = e / np.sum(e, axis=-1, keepdims=True)
    return rnd(np.einsum("bhqk,bhkd->bhqd", p, vv), dt)


# --------------------------------------------------------------------------
# a2  KV caches      mlx-rs-core/src/cache.rs
# --------------------------------------------------------------------------


class ConcatKeyValueCache:
    """cache.rs:44-85."""

    def __init__(self):
        self.keys = None
        self.values = None
        self._offset = 0

    def offset(self) -> int:
        return self._offset

    def max_size(self):
        return None

    def reset(self):
        pass  # trait default (cache.rs:19): does nothing

    def trim(self, n: int) -> int:
        """Not in the reference (speculative.rs:165-169): drop the last n positions of the concatenated arrays."""
        n = max(0, min(int(n), self._offset))
        if n and self.keys is not None:
            self.keys, self.values = self.keys[..., : self._offset - n, :], self.values[..., : self._offset - n, :]
            self._offset -= n
        return n

    def update_and_fetch(self, keys, values):
        if self.keys is not None and self.values is not None:
            self.keys = np.concatenate([self.keys, keys], axis=-2)
            self.values = np.concatenate([self.values, values], axis=-2)
        else:
            self.keys, self.values = np.array(keys), np.array(values)
        self._offset = self.keys.shape[-2]
        return self.keys, self.values


class KVCache:
    """cache.rs:91-194: step-256 preallocated cache with in-place slice writes."""

    def __init__(self, step: int = 256):
        self.keys = None
        self.values = None
        self._offset = 0
        self.step = step

    def offset(self) -> int:
        return self._offset

    def max_size(self):
        return None

    def reset(self):
        self._offset = 0

    def capacity(self) -> int:
        return 0 if self.keys is None else self.keys.shape[2]

    def trim(self, n: int) -> int:
        """Not in the reference (speculative.rs:165-169 notes the trait lacks it): the offset moves back, the buffers stay."""
        n = max(0, min(int(n), self._offset))
        self._offset -= n
        return n

    def update_and_fetch(self, keys, values):
        prev = self._offset
        num_new = keys.shape[2]
        needs_grow = self.keys is None or (prev + num_new) > self.keys.shape[2]
        if needs_grow:
            b, hkv, _, kd = keys.shape
            vd = values.shape[3]
            n_steps = (self.step + num_new - 1) // self.step
            new_size = n_steps * self.step
            new_k = np.zeros((b, hkv, new_size, kd), dtype=keys.dtype)
            new_v = np.zeros((b, hkv, new_size, vd), dtype=values.dtype)
            if self.keys is not None and self.values is not None:
                old_k, old_v = self.keys, self.values
                if prev % self.step != 0:
                    old_k, old_v = old_k[:, :, :prev, :], old_v[:, :, :prev, :]
                self.keys = np.concatenate([old_k, new_k], axis=2)
                self.values = np.concatenate([old_v, new_v], axis=2)
            else:
                self.keys, self.values = new_k, new_v
        self._offset += num_new
        self.keys[:, :, prev:self._offset, :] = keys
        self.values[:, :, prev:self._offset, :] = values
        return self.keys[:, :, :self._offset, :], self.values[:, :, :self._offset, :]


# --------------------------------------------------------------------------
# a10  sampler      mlx-rs-core/src/sampler.rs:9-18
# --------------------------------------------------------------------------


def sample_greedy(logits) -> np.ndarray:
    """temp == 0 -> argmax(logits, -1) as u32, first max index on ties."""
    return np.argmax(np.asarray(logits), axis=-1).astype(np.uint32)


def sample(logits, temp: float, key) -> np.ndarray:
    """`DefaultSampler::sample` (mlx-rs-core/src/sampler.rs:9-18; same body as qwen3-mlx/src/model.rs:733-741):
    temp == 0 -> argmax; otherwise `categorical(logits * array!(1.0 / temp))`.  `array!(f32)` makes the
    product float32 whatever the logits' dtype (bf16 x f32 promotes to f32).  `key` is the PRNG key the
    categorical draw uses (the reference passes None = next key of the global `RandomState`)."""
    from . import mlx_rng
    if temp == 0.0:
        return sample_greedy(logits)
    inv = np.float32(np.float32(1.0) / np.float32(temp))
    scaled = (np.asarray(logits).astype(np.float32) * inv).astype(np.float32)
    return mlx_rng.categorical(scaled, key)


def argmax_margin(logits) -> np.ndarray:
    """top1 - top2 gap per row; used by tests as the guard under which token-id
    equality between two fp32-summation orders is meaningful."""
    l = np.asarray(logits, dtype=np.float64)
    part = np.partition(l, -2, axis=-1)
    return part[..., -1] - part[..., -2]


# --------------------------------------------------------------------------
# affine quantisation      mlx-rs/src/ops/quantization.rs:41-153
# --------------------------------------------------------------------------


def quantize(w, group_size: int = 64, bits: int = 4) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """MLX affine quantisation (published algorithm): per group of `group_size` along
    the last axis, scale = (max-min)/(2^bits-1) with the sign/edge adjustment MLX
    applies, bias = edge; q = round((w-bias)/scale) packed LSB-first into uint32.
    Bound KAT: ops/quantization.rs:289-305."""
    w = np.asarray(w, dtype=np.float32)
    assert w.shape[-1] % group_size == 0 and (32 % bits) == 0
    # MLX's affine_quantize works in float32; with symmetric weights edge/scale sits at k + 0.5, so the
    # precision of this arithmetic decides q0 -- keep every step in float32 like the published kernel
    f32 = np.float32
    n_bins = f32((1 << bits) - 1)
    g = w.reshape(*w.shape[:-1], w.shape[-1] // group_size, group_size)
    w_max = g.max(axis=-1)
    w_min = g.min(axis=-1)
    eps = f32(1e-7)
    mask = np.abs(w_min) > np.abs(w_max)
    scale = np.maximum(((w_max - w_min).astype(f32) / n_bins).astype(f32), eps)
    scale = np.where(mask, scale, -scale).astype(f32)
    edge = np.where(mask, w_min, w_max).astype(f32)
    q0 = np.rint((edge / scale).astype(f32))
    scale = np.where(q0 != 0, (edge / np.where(q0 != 0, q0, f32(1))).astype(f32), scale).astype(f32)
    bias = np.where(q0 == 0, f32(0), edge).astype(f32)
    q = np.clip(np.rint(((g - bias[..., None]).astype(f32) / scale[..., None]).astype(f32)), 0, n_bins).astype(np.uint32)
    q = q.reshape(*w.shape)
    per_word = 32 // bits
    qw = q.reshape(*w.shape[:-1], w.shape[-1] // per_word, per_word)
    shifts = (np.arange(per_word, dtype=np.uint32) * np.uint32(bits))
    packed = np.bitwise_or.reduce(qw << shifts, axis=-1).astype(np.uint32)
    return packed, scale.astype(np.float32), bias.astype(np.float32)


def dequantize(packed, scales, biases, group_size: int = 64, bits: int = 4, dt: str = "f32") -> np.ndarray:
    """w = scale * q + bias; element j of a row is the `bits`-wide field at bit
    (j*bits) mod 32 of word floor(j*bits/32) (LSB first)."""
    packed = np.asarray(packed, dtype=np.uint32)
    per_word = 32 // bits
    shifts = (np.arange(per_word, dtype=np.uint32) * np.uint32(bits))
    q = ((packed[..., None] >> shifts) & np.uint32((1 << bits) - 1)).reshape(*packed.shape[:-1], -1)
    sc = np.repeat(np.asarray(scales, dtype=np.float64), group_size, axis=-1)
    bi = np.repeat(np.asarray(biases, dtype=np.float64), group_size, axis=-1)
    return rnd(q.astype(np.float64) * sc + bi, dt)


def quantized_matmul(x, packed, scales, biases, group_size: int = 64, bits: int = 4, dt: str = "f32") -> np.ndarray:
    """x @ dequant(W)^T (transpose=true), mlx-rs/src/nn/quantized.rs:366-375."""
    w = dequantize(packed, scales, biases, group_size, bits, "f32")
    return rnd(np.asarray(x, dtype=np.float64) @ w.astype(np.float64).T, dt)
