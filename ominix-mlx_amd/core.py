"""Host mirror of the `mlx-rs-core` crate (the L3 layer of SURVEY.md section 1) written against the
mlx-c compatible surface of libomx_hip.so, one C call per mlx-rs call, so that it exercises the
boundary exactly the way the reference's Rust does:

    KeyValueCache / KVCache / ConcatKeyValueCache   mlx-rs-core/src/cache.rs:7-194
    initialize_rope, create_causal_mask,
    create_attention_mask, scaled_dot_product_attention   mlx-rs-core/src/utils.rs:52-209
    DefaultSampler                                   mlx-rs-core/src/sampler.rs:9-18
    fused_swiglu, fused_modulate                     mlx-rs-core/src/metal_kernels.rs:188-339

Same names, argument meaning and error behaviour.  (In a real deployment this layer stays Rust;
only mlx-sys is re-pointed -- INTEGRATION.md.)
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import numpy as np

from . import OmxError
from . import mlx_c as mx
from .mlx_c import Array


class ConcatKeyValueCache:
    """cache.rs:44-85."""

    def __init__(self):
        self.keys: Optional[Array] = None
        self.values: Optional[Array] = None
        self._offset = 0

    def offset(self) -> int:
        return self._offset

    def max_size(self) -> Optional[int]:
        return None

    def reset(self) -> None:
        pass   # trait default (cache.rs:17-19)

    def trim(self, n: int) -> int:
        """The `trim(n)` that speculative.rs:165-169 says the `KeyValueCache` trait lacks: forget the last n cached positions
        (a slice of the concatenated arrays).  Returns the number actually trimmed."""
        n = max(0, min(int(n), self._offset))
        if n and self.keys is not None and self.values is not None:
            keep = self._offset - n
            ks, vs = list(self.keys.shape), list(self.values.shape)
            ks[-2], vs[-2] = keep, keep
            self.keys = mx.slice(self.keys, [0] * len(ks), ks)
            self.values = mx.slice(self.values, [0] * len(vs), vs)
            self._offset = keep
        return n

    def update_and_fetch(self, keys: Array, values: Array) -> Tuple[Array, Array]:
        if self.keys is not None and self.values is not None:
            self.keys = mx.concatenate_axis([self.keys, keys], -2)
            self.values = mx.concatenate_axis([self.values, values], -2)
        else:
            self.keys, self.values = keys, values
        self._offset = self.keys.shape[-2]
        return self.keys, self.values


class KVCache:
    """cache.rs:91-194: step-256 pre-allocated buffers, slice_update writes, [..,:offset,:] views."""

    def __init__(self, step: int = 256):
        self.keys: Optional[Array] = None
        self.values: Optional[Array] = None
        self._offset = 0
        self.step = step

    @staticmethod
    def with_step(step: int) -> "KVCache":
        return KVCache(step)

    def offset(self) -> int:
        return self._offset

    def max_size(self) -> Optional[int]:
        return None

    def reset(self) -> None:
        self._offset = 0

    def trim(self, n: int) -> int:
        """`trim(n)` (missing from the reference's trait, speculative.rs:165-169): the buffers stay, the offset moves back -- later
        writes land on the forgotten positions, reads stop before them.  Returns the number actually trimmed."""
        n = max(0, min(int(n), self._offset))
        self._offset -= n
        return n

    def update_and_fetch(self, keys: Array, values: Array) -> Tuple[Array, Array]:
        prev = self._offset
        b, n_kv_heads, num_new, k_head_dim = keys.shape
        v_head_dim = values.shape[3]
        needs_grow = self.keys is None or (prev + num_new) > self.keys.shape[2]
        if needs_grow:
            n_steps = (self.step + num_new - 1) // self.step
            new_size = n_steps * self.step
            new_k = mx.zeros([b, n_kv_heads, new_size, k_head_dim], keys.dtype)
            new_v = mx.zeros([b, n_kv_heads, new_size, v_head_dim], values.dtype)
            if self.keys is not None and self.values is not None:
                old_k, old_v = self.keys, self.values
                if prev % self.step != 0:
                    old_k = mx.slice(old_k, [0, 0, 0, 0], [b, n_kv_heads, prev, k_head_dim])
                    old_v = mx.slice(old_v, [0, 0, 0, 0], [b, n_kv_heads, prev, v_head_dim])
                self.keys = mx.concatenate_axis([old_k, new_k], 2)
                self.values = mx.concatenate_axis([old_v, new_v], 2)
            else:
                self.keys, self.values = new_k, new_v
        self._offset += num_new
        cap = self.keys.shape[2]
        # k.index_mut((Ellipsis, prev..offset, ..), &keys)  ->  mlx_slice_update (indexmut_impl.rs:29)
        self.keys = mx.slice_update(self.keys, keys, [0, 0, prev, 0], [b, n_kv_heads, self._offset, k_head_dim])
        self.values = mx.slice_update(self.values, values, [0, 0, prev, 0], [b, n_kv_heads, self._offset, v_head_dim])
        assert self.keys.shape[2] == cap
        return (mx.slice(self.keys, [0, 0, 0, 0], [b, n_kv_heads, self._offset, k_head_dim]),
                mx.slice(self.values, [0, 0, 0, 0], [b, n_kv_heads, self._offset, v_head_dim]))


def initialize_rope(dims: int, base: float, traditional: bool, scaling_config: Optional[dict],
                    _max_position_embeddings: int = 0) -> dict:
    """utils.rs:52-97: returns the nn::Rope parameters (dims, traditional, base, scale)."""
    rope_type = "default"
    if scaling_config is not None:
        rope_type = scaling_config.get("type", scaling_config.get("rope_type", "default"))
    if rope_type in ("default", "linear"):
        scale = 1.0
        if rope_type == "linear":
            if "factor" not in scaling_config:
                raise OmxError('key "factor" is not found in scaling config')
            try:
                scale = 1.0 / float(scaling_config["factor"])
            except (TypeError, ValueError):
                raise OmxError('key "factor" is not a valid float')
        return {"dims": dims, "traditional": traditional, "base": base, "scale": scale}
    raise OmxError(f"Unsupported RoPE type {rope_type!r}")


def apply_rope(rope: dict, x: Array, offset: int = 0) -> Array:
    """nn::Rope::forward (nn/positional_encoding.rs:112-137) -> mlx_fast_rope."""
    return mx.rope(x, rope["dims"], rope["traditional"], rope["base"], rope["scale"], offset)


def create_causal_mask(N: int, offset: Optional[int] = None, window_size: Optional[int] = None) -> Array:
    """utils.rs:134-153: bool [N, offset+N], l >= r (and l <= r + window).  The reference builds it with
    arange/greater_equal/logical_and graph ops; the index arithmetic is integer-exact, so it is built on
    the host here and uploaded as the same bool array."""
    offset = offset or 0
    rinds = np.arange(offset + N)[None, :]
    linds = np.arange(offset, offset + N)[:, None]
    mask = linds >= rinds
    if window_size is not None:
        mask &= linds <= rinds + window_size
    return Array.from_numpy(mask, mx.BOOL)


def create_attention_mask(h: Array, cache: List, return_array: Optional[bool] = None):
    """utils.rs:156-188 -> None | "causal" | Array."""
    ra = bool(return_array) if return_array is not None else False
    T = h.shape[1]
    if T > 1:
        offset, window = 0, None
        c = cache[0] if cache else None
        if c is not None:
            offset = c.offset()
            if c.max_size() is not None:
                window = c.max_size()
                offset = min(offset, window)
                ra = ra or (offset + T) > window
        return create_causal_mask(T, offset, window) if ra else "causal"
    return None


def scaled_dot_product_attention(queries: Array, keys: Array, values: Array, _cache, scale: float, mask=None) -> Array:
    """utils.rs:191-209 (the cache argument is ignored there too)."""
    return mx.scaled_dot_product_attention(queries, keys, values, scale, mask)


class RandomState:
    """mlx-rs/src/random.rs:21-41: the key sequence behind `key = None` on the Rust side (`seed`, one split per draw)."""

    def __init__(self, seed: int = 0):
        self.state = mx.random_key(seed)

    def seed(self, seed: int) -> None:
        self.state = mx.random_key(seed)

    def next(self) -> Array:
        self.state, sub = mx.random_split(self.state, 2)
        return sub


GLOBAL_RANDOM_STATE: Optional[RandomState] = None


def seed(value: int) -> None:
    """mlx_rs::random::seed (random.rs:88-91)."""
    global GLOBAL_RANDOM_STATE
    GLOBAL_RANDOM_STATE = RandomState(value)


def _resolve_key(key: Optional[Array]) -> Array:
    """random.rs:57-68 (`resolve`): the given key, or the next key of the global state."""
    global GLOBAL_RANDOM_STATE
    if key is not None:
        return key
    if GLOBAL_RANDOM_STATE is None:
        GLOBAL_RANDOM_STATE = RandomState(0)   # the reference seeds from the clock; a fixed seed keeps runs repeatable
    return GLOBAL_RANDOM_STATE.next()


class DefaultSampler:
    """sampler.rs:9-18: temp == 0 -> argmax(-1); otherwise categorical(logits * array!(1/temp))."""

    def sample(self, logits: Array, temp: float, key: Optional[Array] = None) -> Array:
        if temp == 0.0:
            return mx.argmax_axis(logits, -1)
        scaled = mx.multiply(logits, Array.from_numpy(np.float32(1.0) / np.float32(temp), mx.FLOAT32))
        return mx.random_categorical(scaled, -1, None, _resolve_key(key))


fused_swiglu = mx.fused_swiglu
fused_modulate = mx.fused_modulate
