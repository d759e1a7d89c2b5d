"""Host mirror of qwen3-mlx's `Model` + `Generate` (qwen3-mlx/src/model.rs:473-498, 743-844)
over the fused decode engine of libomx_hip.so (include/omx.h, omx_qwen3_*)."""
from __future__ import annotations

import ctypes
import sys
from typing import Dict, Iterator, Optional

import numpy as np

from . import OmxError, check, lib, require_device
from .ops import Tensor

c_int, c_float, c_void_p, c_uint32 = ctypes.c_int, ctypes.c_float, ctypes.c_void_p, ctypes.c_uint32


class Qwen3Config(ctypes.Structure):
    """omx_qwen3_config == the ModelArgs fields the forward uses (model.rs:47-64)."""
    _fields_ = [("hidden_size", c_int), ("num_hidden_layers", c_int), ("intermediate_size", c_int),
                ("num_attention_heads", c_int), ("num_key_value_heads", c_int), ("head_dim", c_int),
                ("vocab_size", c_int), ("rms_norm_eps", c_float), ("rope_theta", c_float), ("rope_scale", c_float),
                ("tie_word_embeddings", c_int), ("max_context", c_int), ("tp_rank", c_int), ("tp_size", c_int),
                ("quant_bits", c_int), ("quant_group", c_int), ("num_experts", c_int), ("num_experts_per_tok", c_int),
                ("moe_intermediate_size", c_int), ("moe_mode", c_int), ("norm_topk_prob", c_int), ("no_qk_norm", c_int),
                ("ep_rank", c_int), ("ep_size", c_int), ("attention_bias", c_int), ("quant_scales_f16", c_int)]


ENGINE_SIGNATURES = {
    "omx_fill_uniform_2d": (c_int, [c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64,
                                    ctypes.c_int64, c_uint32, c_float, c_float, c_int, c_void_p]),
    "omx_qwen3_create": (c_int, [ctypes.POINTER(c_void_p), ctypes.POINTER(Qwen3Config)]),
    "omx_qwen3_destroy": (c_int, [c_void_p]),
    "omx_qwen3_set_weight": (c_int, [c_void_p, ctypes.c_char_p, c_void_p, ctypes.c_size_t]),
    "omx_qwen3_synth_weights": (c_int, [c_void_p, c_uint32]),
    "omx_qwen3_synth_weights_peaked": (c_int, [c_void_p, c_uint32]),
    "omx_qwen3_set_comm": (c_int, [c_void_p, c_void_p, c_void_p]),
    "omx_qwen3_set_sampler": (c_int, [c_void_p, ctypes.c_float, ctypes.c_uint64]),
    "omx_qwen3_sampler_state": (c_int, [c_void_p, c_void_p, c_int]),
    "omx_qwen3_encode": (c_int, [c_void_p, ctypes.POINTER(c_uint32), c_int, c_void_p, ctypes.POINTER(c_int), c_int, c_void_p]),
    "omx_qwen3_reset": (c_int, [c_void_p]),
    "omx_qwen3_offset": (c_int, [c_void_p, ctypes.POINTER(c_int)]),
    "omx_qwen3_prefill": (c_int, [c_void_p, ctypes.POINTER(c_uint32), c_int, ctypes.POINTER(c_uint32)]),
    "omx_qwen3_decode": (c_int, [c_void_p, c_int, ctypes.POINTER(c_uint32)]),
    "omx_qwen3_last_logits": (c_int, [c_void_p, c_void_p, c_int]),
    "omx_qwen3_last_decode_ms": (c_int, [c_void_p, ctypes.POINTER(c_float)]),
    "omx_qwen3_last_prefill_ms": (c_int, [c_void_p, ctypes.POINTER(c_float)]),
    "omx_qwen3_debug_read": (c_int, [c_void_p, ctypes.c_char_p, c_void_p, ctypes.c_size_t]),
    "omx_qwen3_stream": (c_int, [c_void_p, ctypes.POINTER(c_void_p)]),
    "omx_qwen3_step_bytes": (c_int, [c_void_p, c_int, ctypes.POINTER(ctypes.c_double)]),
    "omx_qwen3_decode_path": (c_int, [c_void_p, ctypes.POINTER(c_int)]),
    "omx_qwen3_debug_trace_step": (c_int, [c_void_p, c_void_p, ctypes.c_size_t, ctypes.POINTER(c_int)]),
    "omx_qwen3_time_step_kernels": (c_int, [c_void_p, c_int, ctypes.POINTER(ctypes.c_float)]),
    "omx_qwen3_debug_trace_engine": (c_int, [c_void_p, c_void_p, ctypes.c_size_t, ctypes.POINTER(c_int)]),
    "omx_qwen3_verify": (c_int, [c_void_p, ctypes.POINTER(c_uint32), c_int, ctypes.POINTER(c_uint32)]),
    "omx_qwen3_verify_logits": (c_int, [c_void_p, c_int, c_void_p, c_int]),
    "omx_qwen3_trim": (c_int, [c_void_p, c_int, c_uint32]),
    "omx_qwen3_get_weight": (c_int, [c_void_p, ctypes.c_char_p, ctypes.POINTER(c_void_p), ctypes.POINTER(ctypes.c_size_t)]),
    "omx_bench_qwen3_per_op": (c_int, [c_void_p, ctypes.POINTER(Qwen3Config), ctypes.POINTER(c_uint32), c_int, c_int, ctypes.POINTER(c_uint32),
                                       ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double)]),
}
for _n, (_r, _a) in ENGINE_SIGNATURES.items():
    _f = getattr(lib, _n)
    _f.restype, _f.argtypes = _r, _a


def rope_scale_from_config(rope_scaling: Optional[dict]) -> float:
    """mlx_rs_core::initialize_rope (utils.rs:52-97): only default / linear are accepted."""
    rope_type = "default"
    if rope_scaling is not None:
        rope_type = rope_scaling.get("type", rope_scaling.get("rope_type", "default"))
    if rope_type == "default":
        return 1.0
    if rope_type == "linear":
        if "factor" not in rope_scaling:
            raise OmxError('key "factor" is not found in scaling config')
        try:
            return 1.0 / float(rope_scaling["factor"])
        except (TypeError, ValueError):
            raise OmxError('key "factor" is not a valid float')
    raise OmxError(f"Unsupported RoPE type {rope_type!r}")


def expected_shape(c: Qwen3Config, name: str):
    """Shape the engine will read for checkpoint tensor `name` on THIS rank (after the TP / EP slicing), or None for a
    name the forward does not use.  Mirrors resolve_weights in csrc/engine.hip."""
    tp, ep = max(c.tp_size, 1), max(c.ep_size, 1)
    hd, D = c.hidden_size, c.head_dim
    H, Hkv, I, V = c.num_attention_heads // tp, max(1, c.num_key_value_heads // tp), c.intermediate_size // tp, c.vocab_size
    Im, E = c.moe_intermediate_size, c.num_experts
    if E > 0 and tp > 1:      # expert tensor parallel: this rank's intermediate columns of every expert
        Im //= tp
    leaf_kind = None
    for suffix in (".weight", ".scales", ".biases", ".bias"):
        if name.endswith(suffix):
            leaf_kind, stem = suffix[1:], name[:-len(suffix)]
            break
    if leaf_kind is None:
        return None
    quant = bool(c.quant_bits)

    def lin(n, k, stack=()):
        """[n, k] Linear: dense, or the packed triplet of a quantized checkpoint."""
        if leaf_kind == "bias":
            return (n,)
        if not quant:
            return tuple(stack) + (n, k) if leaf_kind == "weight" else None
        if leaf_kind == "weight":
            return tuple(stack) + (n, k * c.quant_bits // 32)
        return tuple(stack) + (n, k // c.quant_group)

    if stem == "model.embed_tokens":
        return lin(V, hd)
    if stem == "lm_head":
        return lin(V // tp, hd)
    if stem == "model.norm":
        return (hd,) if leaf_kind == "weight" else None
    parts = stem.split(".")
    if len(parts) < 4 or parts[0] != "model" or parts[1] != "layers":
        return None
    sub = ".".join(parts[3:])
    table = {"self_attn.q_proj": (H * D, hd), "self_attn.k_proj": (Hkv * D, hd), "self_attn.v_proj": (Hkv * D, hd),
             "self_attn.o_proj": (hd, H * D), "mlp.gate_proj": (I, hd), "mlp.up_proj": (I, hd), "mlp.down_proj": (hd, I)}
    if sub in table:
        return lin(*table[sub])
    if sub in ("input_layernorm", "post_attention_layernorm"):
        return (hd,) if leaf_kind == "weight" else None
    if sub in ("self_attn.q_norm", "self_attn.k_norm"):
        return (D,) if leaf_kind == "weight" else None
    if E > 0:
        El = E // ep
        for mp in ("block_sparse_moe.", "mlp."):
            if sub == mp + "gate":
                return lin(E, hd)
            if sub in (mp + "switch_mlp.gate_proj", mp + "switch_mlp.up_proj"):
                return lin(Im, hd, (El,))
            if sub == mp + "switch_mlp.down_proj":
                return lin(hd, Im, (El,))
    return None


class Model:
    """qwen3_mlx::Model (dense Qwen3) resident on one MI355X (or one TP shard of it)."""

    def __init__(self, *, hidden_size, num_hidden_layers, intermediate_size, num_attention_heads,
                 num_key_value_heads, head_dim, vocab_size, rms_norm_eps=1e-6, rope_theta=1e6,
                 tie_word_embeddings=False, rope_scaling=None, max_context=4096, tp_rank=0, tp_size=1, quantization=None,
                 num_experts=0, num_experts_per_tok=0, moe_intermediate_size=0, moe_mode="qwen3_moe", norm_topk_prob=False,
                 qk_norm=True, ep_rank=0, ep_size=1, attention_bias=False, **_ignored):
        """quantization: config.json's {"bits": 4|8, "group_size": 64} (model.rs:63) or None for a bf16 checkpoint; + "scales_dtype":
        "float16" when the checkpoint's scales / biases are float16 (loader.load_model reads it off the tensors' dtype).
        num_experts > 0: sparse-MoE feed-forward in every layer -- moe_mode "qwen3_moe" (qwen3_moe.rs ModelArgs :60-87) or
        "mixtral" (mixtral-mlx ModelArgs :54-80, with qk_norm=False and moe_intermediate_size = intermediate_size)."""
        require_device()
        q = quantization or {}
        self.cfg = Qwen3Config(hidden_size, num_hidden_layers, intermediate_size, num_attention_heads,
                               num_key_value_heads, head_dim, vocab_size, rms_norm_eps, rope_theta,
                               rope_scale_from_config(rope_scaling), int(bool(tie_word_embeddings)), max_context,
                               tp_rank, tp_size, int(q.get("bits", 0)), int(q.get("group_size", 64 if q else 0)),
                               int(num_experts), int(num_experts_per_tok), int(moe_intermediate_size),
                               {"mixtral": 0, "qwen3_moe": 1}[moe_mode], int(bool(norm_topk_prob)), int(not qk_norm),
                               int(ep_rank), int(ep_size), int(bool(attention_bias)),
                               int(str(q.get("scales_dtype", "bfloat16")).lower() in ("float16", "f16", "half")))
        self._h = c_void_p()
        check(lib.omx_qwen3_create(ctypes.byref(self._h), ctypes.byref(self.cfg)))
        self._keep = []

    def close(self) -> None:
        h = getattr(self, "_h", None)
        if h is not None and h.value:
            lib.omx_qwen3_destroy(h)
            self._h = c_void_p()

    def __del__(self):
        # never call into HIP while the interpreter (and possibly the HIP runtime / a profiler
        # layered on it) is being torn down
        if sys is not None and not sys.is_finalizing():   # (module globals are already None late in shutdown)
            self.close()

    @property
    def vocab_local(self) -> int:
        return self.cfg.vocab_size // self.cfg.tp_size

    def load_weights(self, weights: Dict[str, np.ndarray]) -> None:
        """ModuleParametersExt::load_safetensors equivalent for in-memory arrays keyed by HF name."""
        if self.cfg.tp_size > 1:   # slice the logical checkpoint with the shared shard plan (tp.py)
            from . import tp
            weights = tp.shard_state_dict(weights, self.cfg.tp_rank, self.cfg.tp_size, bool(self.cfg.tie_word_embeddings),
                                          self.cfg.num_key_value_heads, self.cfg.head_dim)
        if self.cfg.ep_size > 1:   # expert parallel: this rank keeps its slice of every stacked expert tensor
            from . import ep
            weights = {k: (ep.shard_experts(v, self.cfg.ep_rank, self.cfg.ep_size) if ".switch_mlp." in k else v)
                       for k, v in weights.items()}
        for name, arr in weights.items():
            # the engine takes raw device pointers: a tensor whose shape disagrees with the config would be read past its
            # end, so every known name is checked here (the reference raises a shape error on load)
            want = self.expected_shape(name)
            if want is not None and tuple(arr.shape) != want:
                raise OmxError(f"ShapeMismatch: {name} has shape {tuple(arr.shape)}, the config expects {want}")
            dt = np.asarray(arr).dtype
            if self.cfg.quant_bits and name.endswith((".scales", ".biases")):
                # float16 triplets (an MLX float16 checkpoint) stay float16 on the device: the packed-weight kernels widen each group's
                # scale / bias to its exact float32 value (a host-side rounding to bf16 shifts every weight of a group by the same amount
                # -- measured outside the decoder's logit bound).  The model must have been created for them.
                f16 = dt == np.float16
                if f16 != bool(self.cfg.quant_scales_f16):
                    raise OmxError(f"{name}: {'float16' if f16 else str(dt)} scales / biases, but the model was created with "
                                   f"quantization scales_dtype={'float16' if self.cfg.quant_scales_f16 else 'bfloat16'}")
                if f16:
                    t = Tensor.from_numpy(arr, "f16")
                    self._keep.append(t)
                    check(lib.omx_qwen3_set_weight(self._h, name.encode(), t.ptr, t.nbytes))
                    continue
            if self.cfg.quant_bits and name.endswith(".weight") and name[:-7] + ".scales" in weights and dt != np.uint32:
                raise OmxError(f"{name}: a quantized weight must be packed uint32, found {dt}")
            # quantized checkpoints: "<prefix>.weight" is packed uint32 (ops/quantization.rs:41-84), scales / biases bf16; in a float16
            # checkpoint every other tensor (the norm weights) is float16 too -- the model then runs in float16 end to end, like in MLX
            if self.cfg.quant_scales_f16 and dt != np.uint32 and type(arr).__name__ == "Bf16Bits":
                # a BF16 tensor of a checkpoint whose triplets are float16 (loader.read_safetensors hands raw bits): its VALUES go up as float16
                arr = (np.asarray(arr).astype(np.uint32) << np.uint32(16)).view(np.float32)
            t = Tensor.from_numpy(arr, "u32" if dt == np.uint32 else "f16" if self.cfg.quant_scales_f16 else "bf16")
            self._keep.append(t)
            check(lib.omx_qwen3_set_weight(self._h, name.encode(), t.ptr, t.nbytes))

    def expected_shape(self, name: str):
        return expected_shape(self.cfg, name)

    def synth_weights(self, base_seed: int = 0x0C0FFEE5, peaked: bool = False) -> None:
        """peaked: embedding std 64 and lm_head[v] = table[(v + 1) mod V] -- greedy tokens count down with top-1 margins far above the
        bf16 bound (full-size parity tests assert token equality; oracle/ref_qwen3.py synth_weights(peaked=True))."""
        fn = lib.omx_qwen3_synth_weights_peaked if peaked else lib.omx_qwen3_synth_weights
        check(fn(self._h, base_seed & 0xFFFFFFFF))

    def set_comm(self, comm_ptr: int, allreduce_fn_ptr: int) -> None:
        check(lib.omx_qwen3_set_comm(self._h, comm_ptr, allreduce_fn_ptr))

    def encode(self, input_ids, attention_mask=None, extract_layers=(8, 17, 26)):
        """Qwen3TextEncoder::encode (flux-klein-mlx/src/qwen3_encoder.rs:403-455): hidden states after the tapped
        layers (0-indexed, raw, no final norm) concatenated on the last axis -> device Tensor [n, len(taps)*hidden]."""
        from .ops import Tensor
        ids = np.ascontiguousarray(np.asarray(input_ids, dtype=np.uint32).ravel())
        taps = (c_int * len(extract_layers))(*[int(t) for t in extract_layers])
        out = Tensor((ids.size, len(extract_layers) * self.cfg.hidden_size), "f16" if self.cfg.quant_scales_f16 else "bf16")
        am = None
        if attention_mask is not None:
            am = np.ascontiguousarray(np.asarray(attention_mask).ravel() != 0, dtype=np.uint8)
            if am.size != ids.size:
                raise OmxError("encode: attention_mask and input_ids differ in length")
        check(lib.omx_qwen3_encode(self._h, ids.ctypes.data_as(ctypes.POINTER(c_uint32)), ids.size,
                                   am.ctypes.data if am is not None else None, taps, len(extract_layers), out.ptr))
        return out

    def set_sampler(self, temperature: float, seed: int = 0) -> None:
        """DefaultSampler (mlx-rs-core/src/sampler.rs:9-18): 0 = greedy, otherwise categorical(logits / temperature)
        drawn on the device with the key sequence of `mlx_rs::random::seed(seed)`."""
        check(lib.omx_qwen3_set_sampler(self._h, float(temperature), int(seed) & 0xFFFFFFFFFFFFFFFF))

    def sampler_state(self) -> tuple:
        """The two words of the sampler's key sequence (after set_sampler): the reference's speculative loop draws both models' tokens from
        ONE global sequence, which two models reproduce by handing this state over (speculative.py)."""
        st = (c_uint32 * 2)()
        check(lib.omx_qwen3_sampler_state(self._h, st, 0))
        return int(st[0]), int(st[1])

    def set_sampler_state(self, state) -> None:
        st = (c_uint32 * 2)(int(state[0]), int(state[1]))
        check(lib.omx_qwen3_sampler_state(self._h, st, 1))

    def reset(self) -> None:
        check(lib.omx_qwen3_reset(self._h))

    def offset(self) -> int:
        v = c_int()
        check(lib.omx_qwen3_offset(self._h, ctypes.byref(v)))
        return v.value

    def prefill(self, prompt) -> int:
        p = np.ascontiguousarray(prompt, dtype=np.uint32)
        first = c_uint32()
        check(lib.omx_qwen3_prefill(self._h, p.ctypes.data_as(ctypes.POINTER(c_uint32)), p.size, ctypes.byref(first)))
        return first.value

    def decode(self, n: int) -> np.ndarray:
        out = np.empty(n, dtype=np.uint32)
        check(lib.omx_qwen3_decode(self._h, n, out.ctypes.data_as(ctypes.POINTER(c_uint32))))
        return out

    def verify(self, tokens) -> np.ndarray:
        """speculative.rs:132-161 `verify_draft_tokens`: all tokens in one batched pass on top of the cache; greedy token per position."""
        ids = np.ascontiguousarray(np.asarray(tokens, dtype=np.uint32).ravel())
        out = np.empty(ids.size, dtype=np.uint32)
        check(lib.omx_qwen3_verify(self._h, ids.ctypes.data_as(ctypes.POINTER(c_uint32)), ids.size, out.ctypes.data_as(ctypes.POINTER(c_uint32))))
        return out

    def verify_logits(self, row: int) -> np.ndarray:
        raw = np.empty(self.vocab_local, dtype=np.uint16)
        check(lib.omx_qwen3_verify_logits(self._h, row, raw.ctypes.data, raw.size))
        return (raw.astype(np.uint32) << np.uint32(16)).view(np.float32)

    def trim(self, n: int, next_token: int) -> None:
        """KeyValueCache::trim(n) (missing in the reference, speculative.rs:165-169) + the next step's input token."""
        check(lib.omx_qwen3_trim(self._h, n, int(next_token)))

    def last_decode_ms(self) -> float:
        v = c_float()
        check(lib.omx_qwen3_last_decode_ms(self._h, ctypes.byref(v)))
        return v.value

    def last_prefill_ms(self) -> float:
        v = c_float()
        check(lib.omx_qwen3_last_prefill_ms(self._h, ctypes.byref(v)))
        return v.value

    def per_op_route(self, prompt, n_new: int) -> dict:
        """The DROP-IN route on this model's weights (csrc/per_op_route.hip): qwen3-mlx's Model::forward + Generate::next replayed call for
        call through the mlx-c handle ABI, as an unmodified crate would drive it -- greedy tokens (the prompt's, then n_new), host
        wall-clock per decoded token, mlx_* calls per token.  Independent of the engine's own KV cache and step graph."""
        ids = np.ascontiguousarray(np.asarray(prompt, dtype=np.uint32).ravel())
        toks = np.zeros(n_new + 1, np.uint32)
        pre, per, calls = ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
        check(lib.omx_bench_qwen3_per_op(self._h, ctypes.byref(self.cfg), ids.ctypes.data_as(ctypes.POINTER(c_uint32)), ids.size, n_new,
                                         toks.ctypes.data_as(ctypes.POINTER(c_uint32)), ctypes.byref(pre), ctypes.byref(per), ctypes.byref(calls)))
        return {"tokens": toks, "prefill_ms": pre.value, "ms_per_token": per.value, "calls_per_token": calls.value}

    def per_op_route_forced(self, prompt, forced, logit_steps) -> dict:
        """The same route TEACHER-FORCED (the oracle pins): after the prompt, position i is fed forced[i]; returns the route's own greedy
        token at every step and the float32-widened bf16 logits rows of `logit_steps` (0 = the prompt's last position)."""
        ids = np.ascontiguousarray(np.asarray(prompt, dtype=np.uint32).ravel())
        forced = np.ascontiguousarray(np.asarray(forced, dtype=np.uint32).ravel())
        steps = np.ascontiguousarray(np.asarray(logit_steps, dtype=np.int32).ravel())
        toks = np.zeros(forced.size + 1, np.uint32)
        raw = np.zeros((steps.size, self.cfg.vocab_size), np.uint16)
        fn = lib.omx_bench_qwen3_per_op_ex
        fn.restype = ctypes.c_int
        fn.argtypes = [c_void_p, c_void_p, ctypes.POINTER(c_uint32), c_int, c_int, ctypes.POINTER(c_uint32), c_void_p, c_void_p, c_void_p,
                       ctypes.POINTER(c_uint32), ctypes.POINTER(ctypes.c_int32), c_int, c_void_p]
        pre, per, calls = ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
        check(fn(self._h, ctypes.addressof(self.cfg), ids.ctypes.data_as(ctypes.POINTER(c_uint32)), ids.size, forced.size,
                 toks.ctypes.data_as(ctypes.POINTER(c_uint32)), ctypes.addressof(pre), ctypes.addressof(per), ctypes.addressof(calls),
                 forced.ctypes.data_as(ctypes.POINTER(c_uint32)), steps.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)), steps.size, raw.ctypes.data))
        return {"tokens": toks, "logits": (raw.astype(np.uint32) << np.uint32(16)).view(np.float32)}

    def last_logits(self) -> np.ndarray:
        raw = np.empty(self.vocab_local, dtype=np.uint16)
        check(lib.omx_qwen3_last_logits(self._h, raw.ctypes.data, raw.size))
        if self.cfg.quant_scales_f16:      # a float16 model's logits are float16
            return raw.view(np.float16).astype(np.float32)
        return (raw.astype(np.uint32) << np.uint32(16)).view(np.float32)

    def stream(self) -> int:
        s = c_void_p()
        check(lib.omx_qwen3_stream(self._h, ctypes.byref(s)))
        return s.value or 0

    def decode_path(self) -> str:
        """'graph' | 'eager' | 'aql' (packets on the engine's own HSA queue, OMX_STEP_AQL) once the first step ran ('unbuilt' before)."""
        v = c_int()
        check(lib.omx_qwen3_decode_path(self._h, ctypes.byref(v)))
        return ("unbuilt", "graph", "eager", "aql")[v.value]

    KERNEL_CLASSES = ("qkv", "attention", "o", "gate_up", "down", "lm_head", "step_engine")

    def time_step_kernels(self, steps: int = 4) -> dict:
        """Average in-step duration (microseconds) of each per-layer kernel, HIP events on the step's stream around every launch of
        `steps` real (eager) decode steps: each launch's own HIP start / stop events -- omx_qwen3_time_step_kernels."""
        us = (ctypes.c_float * 7)()
        check(lib.omx_qwen3_time_step_kernels(self._h, steps, us))
        return dict(zip(self.KERNEL_CLASSES, (float(v) for v in us)))

    def step_bytes(self, ctx: int) -> float:
        v = ctypes.c_double()
        check(lib.omx_qwen3_step_bytes(self._h, ctx, ctypes.byref(v)))
        return v.value


class Generate:
    """qwen3_mlx::Generate (model.rs:743-844): iterator yielding sampled tokens; the first `next`
    prefills the prompt.  temp == 0 is the greedy path of sample() (model.rs:733-735); temp != 0 draws
    categorical(logits / temp) (model.rs:736-739) from the key sequence seeded with `seed`."""

    def __init__(self, model: Model, temp: float, prompt_token, chunk: int = 16, seed: int = 0):
        model.set_sampler(temp, seed)
        self.model, self.prompt, self.chunk = model, np.asarray(prompt_token, dtype=np.uint32).ravel(), chunk
        self._prefilled = False
        self._buf = []

    def __iter__(self) -> Iterator[int]:
        return self

    def __next__(self) -> int:
        if not self._prefilled:
            self._prefilled = True
            return self.model.prefill(self.prompt)
        if not self._buf:
            self._buf = list(self.model.decode(self.chunk))
        return int(self._buf.pop(0))
