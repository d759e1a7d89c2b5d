"""Host mirror of the mlx-rs / mlx-rs-core operator surface over the omx_* C ABI.

Names, argument meaning and error behaviour follow the reference so the parity tests read
like its own: mlx_rs::fast::{rms_norm, layer_norm, rope, scaled_dot_product_attention}
(mlx-rs/src/fast.rs), mlx_rs_core::{fused_swiglu, fused_modulate} (metal_kernels.rs),
nn::Linear::forward (nn/linear.rs:87-92), DefaultSampler (sampler.rs).  `Tensor` is the
stand-in for mlx_rs::Array: device memory + shape + dtype, owned by the library allocator.
Everything executes eagerly on the null stream (eval == stream sync).
"""
from __future__ import annotations

import ctypes
import sys
from typing import Optional, Sequence

import numpy as np

from . import (BFLOAT16, BOOL, FLOAT16, FLOAT32, MASK_ADDITIVE, MASK_BOOL, MASK_CAUSAL, MASK_NONE, UINT32, UINT8,
               OmxError, check, lib, require_device)

_ITEMSIZE = {BFLOAT16: 2, FLOAT16: 2, FLOAT32: 4, UINT32: 4, UINT8: 1, BOOL: 1}
_NAMES = {"bf16": BFLOAT16, "f16": FLOAT16, "f32": FLOAT32, "u32": UINT32, "u8": UINT8, "bool": BOOL}


def dtype_code(dt) -> int:
    return _NAMES[dt] if isinstance(dt, str) else int(dt)


def _f32_to_bf16_bits(x: np.ndarray) -> np.ndarray:
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32).astype(np.uint64)
    r = ((u >> np.uint64(16)) & np.uint64(1)) + np.uint64(0x7FFF)
    return ((u + r) >> np.uint64(16)).astype(np.uint16)


class Tensor:
    """Device array (stand-in for mlx_rs::Array)."""

    def __init__(self, shape: Sequence[int], dtype, ptr: Optional[int] = None, owner=None):
        self.shape = tuple(int(s) for s in shape)
        self.dtype = dtype_code(dtype)
        self.nbytes = int(np.prod(self.shape, dtype=np.int64)) * _ITEMSIZE[self.dtype]
        self._owner = owner
        if ptr is None:
            require_device()
            p = ctypes.c_void_p()
            check(lib.omx_malloc(ctypes.byref(p), self.nbytes))
            self.ptr = p.value
            self._owned = True
        else:
            self.ptr = int(ptr)
            self._owned = False

    def __del__(self):
        if sys is None or sys.is_finalizing():   # (module globals are already None late in shutdown)
            return
        if getattr(self, "_owned", False) and self.ptr:
            lib.omx_free(self.ptr)
            self.ptr = None

    @property
    def size(self) -> int:
        return int(np.prod(self.shape, dtype=np.int64))

    @staticmethod
    def from_numpy(a: np.ndarray, dtype="bf16") -> "Tensor":
        code = dtype_code(dtype)
        raw_bf16 = type(a).__name__ == "Bf16Bits"        # loader.read_safetensors: uint16 bit patterns, no conversion
        a = np.asarray(a)
        if code == BFLOAT16 and raw_bf16:
            host = np.ascontiguousarray(a, dtype=np.uint16)
        elif code == BFLOAT16 and a.dtype == np.float16:
            host = _f32_to_bf16_bits(a.astype(np.float32))
        elif code == BFLOAT16:
            if a.dtype.kind in "ui" and a.dtype.itemsize == 2:
                # a sharder / stacker dropped the Bf16Bits marking: converting bit patterns as VALUES would upload garbage
                raise OmxError("from_numpy(bf16): a 16-bit integer array is ambiguous (raw bf16 bits lose their loader.Bf16Bits "
                               "marking through numpy copies); pass float data or re-view it with loader.keep_kind")
            host = _f32_to_bf16_bits(a.astype(np.float32))
        elif code == FLOAT16:
            host = np.ascontiguousarray(a, dtype=np.float16)
        elif code == FLOAT32:
            host = np.ascontiguousarray(a, dtype=np.float32)
        elif code == UINT32:
            host = np.ascontiguousarray(a, dtype=np.uint32)
        elif code in (UINT8, BOOL):
            host = np.ascontiguousarray(a, dtype=np.uint8)
        else:
            raise OmxError(f"unsupported dtype {dtype}")
        t = Tensor(a.shape, code)
        if t.nbytes:
            check(lib.omx_memcpy_h2d(t.ptr, host.ctypes.data, t.nbytes, None))
            check(lib.omx_synchronize(None))
        return t

    def numpy(self) -> np.ndarray:
        """Copy back; bf16/f16 are widened to float32."""
        code = self.dtype
        raw = {BFLOAT16: np.uint16, FLOAT16: np.float16, FLOAT32: np.float32, UINT32: np.uint32,
               UINT8: np.uint8, BOOL: np.uint8}[code]
        host = np.empty(self.shape, dtype=raw)
        if self.nbytes:
            check(lib.omx_memcpy_d2h(host.ctypes.data, self.ptr, self.nbytes, None))
        if code == BFLOAT16:
            return (host.astype(np.uint32) << np.uint32(16)).view(np.float32).reshape(self.shape)
        if code == FLOAT16:
            return host.astype(np.float32)
        if code == BOOL:
            return host.astype(bool)
        return host

    def view(self, shape) -> "Tensor":
        t = Tensor(shape, self.dtype, ptr=self.ptr, owner=self)
        assert t.nbytes == self.nbytes
        return t

    def slice_rows(self, start_elem: int, shape) -> "Tensor":
        """Contiguous sub-view starting `start_elem` elements in (no copy)."""
        return Tensor(shape, self.dtype, ptr=self.ptr + start_elem * _ITEMSIZE[self.dtype], owner=self)


def empty_like(x: Tensor, shape=None) -> Tensor:
    return Tensor(x.shape if shape is None else shape, x.dtype)


def synchronize() -> None:
    check(lib.omx_synchronize(None))


def _p(t: Optional[Tensor]):
    return None if t is None else t.ptr


def fill_uniform(shape, seed: int, amp: float, offset: float = 0.0, dtype="bf16") -> Tensor:
    t = Tensor(shape, dtype)
    check(lib.omx_fill_uniform(t.ptr, t.size, seed & 0xFFFFFFFF, amp, offset, t.dtype, None))
    return t


# ---- mlx_rs::fast -----------------------------------------------------------------------

def rms_norm(x: Tensor, weight: Optional[Tensor], eps: float) -> Tensor:
    """fast::rms_norm(x, weight, eps) -- mlx-rs/src/fast.rs:165-180."""
    dim = x.shape[-1]
    if weight is not None and weight.shape != (dim,):
        raise OmxError(f"rms_norm: weight shape {weight.shape} does not match last axis {dim}")
    out = empty_like(x)
    check(lib.omx_rms_norm(out.ptr, x.ptr, _p(weight), x.size // max(dim, 1), dim, eps, x.dtype, None))
    return out


def layer_norm(x: Tensor, weight: Optional[Tensor], bias: Optional[Tensor], eps: float) -> Tensor:
    """fast::layer_norm -- mlx-rs/src/fast.rs:204-218."""
    dim = x.shape[-1]
    out = empty_like(x)
    check(lib.omx_layer_norm(out.ptr, x.ptr, _p(weight), _p(bias), x.size // max(dim, 1), dim, eps, x.dtype, None))
    return out


def rope(x: Tensor, dims: int, traditional: bool, base, scale: float, offset: int, freqs: Optional[Tensor] = None) -> Tensor:
    """fast::rope(a, dims, traditional, base, scale, offset, freqs) -- fast.rs:15-46: exactly one of `base` and `freqs`
    (float32 [dims / 2]).  Position axis is -2."""
    if len(x.shape) < 2:
        raise OmxError("rope: input must have at least 2 dimensions")
    if (base is None) == (freqs is None):
        raise OmxError("rope: exactly one of base and freqs must be given")
    T, D = x.shape[-2], x.shape[-1]
    out = empty_like(x)
    if freqs is not None:
        if freqs.dtype != FLOAT32 or tuple(freqs.shape) != (dims // 2,):
            raise OmxError(f"rope: freqs must be a float32 vector of dims / 2 = {dims // 2} entries")
        check(lib.omx_rope_freqs(out.ptr, x.ptr, x.size // max(T * D, 1), T, D, dims, int(traditional), freqs.ptr, scale, offset,
                                 x.dtype, None))
        return out
    check(lib.omx_rope(out.ptr, x.ptr, x.size // max(T * D, 1), T, D, dims, int(traditional), base, scale, offset,
                       x.dtype, None))
    return out


def scaled_dot_product_attention(q: Tensor, k: Tensor, v: Tensor, scale: float, mask=None,
                                 kv_strides=None) -> Tensor:
    """mlx_rs_core::scaled_dot_product_attention (utils.rs:191-209) ->
    fast::scaled_dot_product_attention (fast.rs:121-151).  q [B,H,Tq,D]; k,v [B,Hkv,Tk,D].
    mask: None | "causal" | Tensor(bool/u8 [Tq,Tk]) | Tensor(float [Tq,Tk])."""
    B, H, Tq, D = q.shape
    Hkv, Tk = k.shape[1], k.shape[2]
    if isinstance(mask, str):
        if mask not in ("", "causal"):
            raise OmxError(f"Invalid mask mode {mask!r}")
        mode, mptr = (MASK_CAUSAL if mask == "causal" else MASK_NONE), None
    elif mask is None:
        mode, mptr = MASK_NONE, None
    else:
        mode = MASK_BOOL if mask.dtype in (BOOL, UINT8) else MASK_ADDITIVE
        mptr = mask.ptr
    bs, hs = kv_strides if kv_strides is not None else (Hkv * Tk * D, Tk * D)
    out = Tensor((B, H, Tq, D), q.dtype)
    check(lib.omx_sdpa(out.ptr, q.ptr, k.ptr, v.ptr, B, H, Hkv, Tq, Tk, D, bs, hs, scale, mode, mptr, q.dtype, None))
    return out


# ---- mlx_rs_core::metal_kernels -----------------------------------------------------------

def fused_swiglu(x: Tensor, gate: Tensor) -> Tensor:
    """fused_swiglu(x, gate) = silu(gate) * x -- mlx-rs-core/src/metal_kernels.rs:188-236."""
    if x.shape != gate.shape:
        raise OmxError(f"fused_swiglu: shape mismatch {x.shape} vs {gate.shape}")
    out = empty_like(x)
    check(lib.omx_fused_swiglu(out.ptr, x.ptr, gate.ptr, x.size, x.dtype, None))
    return out


def fused_modulate(x: Tensor, shift: Tensor, scale: Tensor, eps: float = 1e-6) -> Tensor:
    """fused_modulate(x[B,S,H], shift[B,H], scale[B,H]) -- metal_kernels.rs:260-339."""
    B, S, H = x.shape
    out = empty_like(x)
    check(lib.omx_fused_modulate(out.ptr, x.ptr, shift.ptr, scale.ptr, B, S, H, eps, x.dtype, None))
    return out


# ---- nn::Linear / Embedding / sampler -------------------------------------------------------

def linear(x: Tensor, w: Tensor, bias: Optional[Tensor] = None) -> Tensor:
    """nn::Linear::forward: x @ W^T (+ b), W [out,in] -- mlx-rs/src/nn/linear.rs:87-92."""
    N, K = w.shape
    if x.shape[-1] != K:
        raise OmxError(f"linear: input features {x.shape[-1]} != weight in-features {K}")
    M = x.size // K
    out = Tensor(tuple(x.shape[:-1]) + (N,), x.dtype)
    check(lib.omx_linear(out.ptr, x.ptr, w.ptr, _p(bias), M, N, K, x.dtype, None))
    return out


def linear_swiglu(x: Tensor, w: Tensor, n_plain: int):
    """Linear + fused_swiglu over the trailing [gate | up] features in one launch (include/omx.h: omx_linear_swiglu).
    -> (plain [M, n_plain] or None, act [M, half])."""
    N, K = w.shape
    if x.shape[-1] != K:
        raise OmxError(f"linear_swiglu: input features {x.shape[-1]} != weight in-features {K}")
    if (N - n_plain) % 2 or N <= n_plain:
        raise OmxError(f"linear_swiglu: {N} weight rows do not split into {n_plain} plain + an even gate/up pair")
    half = (N - n_plain) // 2
    M = x.size // K
    plain = Tensor(tuple(x.shape[:-1]) + (n_plain,), x.dtype) if n_plain else None
    act = Tensor(tuple(x.shape[:-1]) + (half,), x.dtype)
    check(lib.omx_linear_swiglu(_p(plain), act.ptr, x.ptr, w.ptr, M, n_plain, half, K, x.dtype, None))
    return plain, act


def take_rows(table: Tensor, ids: Tensor) -> Tensor:
    """nn::Embedding::forward (gather rows)."""
    dim = table.shape[-1]
    out = Tensor(tuple(ids.shape) + (dim,), table.dtype)
    check(lib.omx_take_rows(out.ptr, table.ptr, ids.ptr, ids.size, dim, table.dtype, None))
    return out


def add(a: Tensor, b: Tensor) -> Tensor:
    out = empty_like(a)
    check(lib.omx_add(out.ptr, a.ptr, b.ptr, a.size, a.dtype, None))
    return out


def argmax(logits: Tensor) -> Tensor:
    """DefaultSampler temp == 0: argmax over the last axis, u32 (sampler.rs:9-12)."""
    n = logits.shape[-1]
    out = Tensor(logits.shape[:-1], UINT32)
    check(lib.omx_argmax(out.ptr, logits.ptr, logits.size // n, n, logits.dtype, None))
    return out


def cast(x: Tensor, dtype) -> Tensor:
    """mlx_rs as_dtype between bf16 / f16 / f32 (one rounding)."""
    out = Tensor(x.shape, dtype)
    check(lib.omx_cast(out.ptr, out.dtype, x.ptr, x.dtype, x.size, None))
    return out


def random_key(seed: int) -> Tensor:
    """mlx_rs::random::key (random.rs:98-100): [2] u32 = (seed >> 32, seed & 0xffffffff)."""
    k = Tensor((2,), UINT32)
    check(lib.omx_random_key(k.ptr, int(seed) & 0xFFFFFFFFFFFFFFFF, None))
    return k


def random_split(key: Tensor, num: int = 2) -> Tensor:
    """mlx_rs::random::split (random.rs:103-115): [num, 2] u32 sub-keys."""
    out = Tensor((num, 2), UINT32)
    check(lib.omx_random_split(out.ptr, key.ptr, num, None))
    return out


def random_bits(key: Tensor, shape) -> Tensor:
    out = Tensor(tuple(shape), UINT32)
    check(lib.omx_random_bits(out.ptr, key.ptr, out.size, None))
    return out


def random_uniform(key: Tensor, shape, lo: float = 0.0, hi: float = 1.0) -> Tensor:
    """mlx_rs::random::uniform::<_, f32> with scalar bounds."""
    out = Tensor(tuple(shape), FLOAT32)
    check(lib.omx_random_uniform(out.ptr, key.ptr, out.size, float(lo), float(hi), None))
    return out


def random_gumbel(key: Tensor, shape) -> Tensor:
    """mlx_rs::random::gumbel::<f32> (random.rs:397-414)."""
    out = Tensor(tuple(shape), FLOAT32)
    check(lib.omx_random_gumbel(out.ptr, key.ptr, out.size, None))
    return out


def random_normal(key: Tensor, shape, loc: float = 0.0, scale: float = 1.0) -> Tensor:
    """mlx_rs::random::normal::<f32> (random.rs:186-212)."""
    out = Tensor(tuple(shape), FLOAT32)
    check(lib.omx_random_normal(out.ptr, key.ptr, out.size, float(loc), float(scale), None))
    return out


def random_categorical(logits: Tensor, key: Tensor, num_samples=None, inv_temp: float = 1.0) -> Tensor:
    """mlx_rs::random::categorical over the last axis (random.rs:456-497); `inv_temp` folds the sampler's
    `logits * array!(1/temp)` (sampler.rs:14) into the same pass."""
    n = logits.shape[-1]
    s = 1 if num_samples is None else int(num_samples)
    out = Tensor(tuple(logits.shape[:-1]) + (() if num_samples is None else (s,)), UINT32)
    check(lib.omx_random_categorical(out.ptr, logits.ptr, logits.size // n, n, s, float(inv_temp), key.ptr, logits.dtype, None))
    return out


def quantize(w: Tensor, group_size: int = 64, bits: int = 4):
    """mlx_rs::ops::quantize (ops/quantization.rs:41-84): w [..., K] -> (packed u32 [..., K*bits/32], scales, biases)."""
    K = w.shape[-1]
    rows = w.size // K
    packed = Tensor(tuple(w.shape[:-1]) + (K * bits // 32,), UINT32)
    scales = Tensor(tuple(w.shape[:-1]) + (K // group_size,), w.dtype)
    biases = Tensor(tuple(w.shape[:-1]) + (K // group_size,), w.dtype)
    check(lib.omx_quantize(packed.ptr, scales.ptr, biases.ptr, w.ptr, rows, K, group_size, bits, w.dtype, None))
    return packed, scales, biases


def dequantize(packed: Tensor, scales: Tensor, biases: Optional[Tensor], group_size: int = 64, bits: int = 4) -> Tensor:
    """mlx_rs::ops::dequantize (ops/quantization.rs:118-153)."""
    K = packed.shape[-1] * 32 // bits
    out = Tensor(tuple(packed.shape[:-1]) + (K,), scales.dtype)     # the result has the scales' dtype (bf16, or f16 for a float16 checkpoint)
    check(lib.omx_dequantize(out.ptr, packed.ptr, scales.ptr, _p(biases), out.size // K, K, group_size, bits, scales.dtype, None))
    return out


def quantized_matmul(x: Tensor, packed: Tensor, scales: Tensor, biases: Optional[Tensor], group_size: int = 64,
                     bits: int = 4) -> Tensor:
    """x @ dequant(W)^T, transpose=true (nn::QuantizedLinear::forward, nn/quantized.rs:366-375)."""
    N, K = packed.shape[0], packed.shape[1] * 32 // bits
    if x.shape[-1] != K:
        raise OmxError(f"quantized_matmul: input features {x.shape[-1]} != weight in-features {K}")
    if x.dtype != scales.dtype:
        # MLX promotes with result_type(x, scales, biases): bf16 x f16 (and anything x f32) is float32.  The packed kernels run in ONE
        # 16-bit type, so a mixed call takes the long way with MLX's result dtype and range (casting a bf16 activation to a
        # float16 checkpoint's dtype returned float16 and overflowed above 65 504): the weight dequantised in the scales' dtype (what MLX's
        # kernel multiplies with), both operands widened to float32, a float32 product
        w32 = cast(dequantize(packed, scales, biases, group_size, bits), FLOAT32)
        return linear(cast(x, FLOAT32), w32)
    out = Tensor(tuple(x.shape[:-1]) + (N,), x.dtype)
    # one dtype for x, out, scales and biases: bf16, or f16 -- a float16 checkpoint runs in float16 end to end, like in MLX
    check(lib.omx_quantized_matmul(out.ptr, x.ptr, packed.ptr, scales.ptr, _p(biases), x.size // K, N, K, group_size, bits,
                                   scales.dtype, None))
    return out


def gather_qmm(x: Tensor, packed: Tensor, scales: Tensor, biases: Optional[Tensor], rhs_indices: Tensor, x_div: int = 1,
               group_size: int = 64, bits: int = 4) -> Tensor:
    """QuantizedSwitchLinear::apply (mixtral-mlx/src/model.rs:195-201): row i uses expert rhs_indices[i] and
    activation row i // x_div; packed [E, N, K*bits/32]."""
    E, N, K = packed.shape[0], packed.shape[1], packed.shape[2] * 32 // bits
    n = rhs_indices.size
    out = Tensor((n, N), x.dtype)
    check(lib.omx_gather_qmm(out.ptr, x.ptr, packed.ptr, scales.ptr, _p(biases), rhs_indices.ptr, n, x_div, N, K, E, group_size,
                             bits, scales.dtype, None))
    return out
