"""Checkpoint formats next to the path: safetensors shards listed by
`model.safetensors.index.json`, `config.json` with an optional `"quantization": {"bits", "group_size"}`
block, MLX-quantized (weight, scales, biases) triplets, and Mixtral's per-expert -> stacked renaming.

    load_model        qwen3_mlx::load_model / load_model_quantized   (qwen3-mlx/src/model.rs:509-727)
    load_all_weights  (model.rs:543-560)
    sanitize_weights  mixtral_mlx::sanitize_weights: experts.{e}.{w1,w2,w3} -> switch_mlp.{gate,down,up}_proj
                      (mixtral-mlx/src/model.rs:480-510)

The safetensors container is parsed here (8-byte little-endian header length, JSON header, raw tensor bytes);
tensors are numpy views of a memory map, so a shard is paged in once, straight into the upload."""
from __future__ import annotations

import json
import os
from typing import Dict

import numpy as np

_DTYPES = {"F32": np.float32, "F16": np.float16, "BF16": np.uint16, "U32": np.uint32, "I32": np.int32, "U8": np.uint8,
           "I64": np.int64, "F64": np.float64, "BOOL": np.uint8, "U16": np.uint16, "I16": np.int16, "I8": np.int8}


class Bf16Bits(np.ndarray):
    """uint16 array holding raw bfloat16 bit patterns (what a BF16 safetensors tensor is mapped as)."""


def keep_kind(src, out: np.ndarray) -> np.ndarray:
    """`out` was cut / stacked from `src` by a numpy call that returns a base ndarray (ascontiguousarray, concatenate,
    stack, asarray): give it back src's bf16-bit-pattern marking, so that the upload does not value-convert the bits."""
    return out.view(Bf16Bits) if isinstance(src, Bf16Bits) and not isinstance(out, Bf16Bits) else out


def read_safetensors(path: str) -> Dict[str, np.ndarray]:
    """name -> array (a view of a read-only memory map).  BF16 tensors come back as `Bf16Bits` (uint16 bits)."""
    with open(path, "rb") as f:
        head = f.read(8)
        if len(head) != 8:
            raise ValueError(f"{path}: not a safetensors file (short header)")
        n = int.from_bytes(head, "little")
        if n <= 0 or n > 100 * 1024 * 1024:
            raise ValueError(f"{path}: implausible safetensors header length {n}")
        meta = json.loads(f.read(n).decode("utf-8"))
    mm = np.memmap(path, dtype=np.uint8, mode="r", offset=8 + n)
    out = {}
    for name, info in meta.items():
        if name == "__metadata__":
            continue
        dt = info["dtype"]
        if dt not in _DTYPES:
            raise ValueError(f"{path}: tensor {name} has unsupported dtype {dt}")
        b, e = info["data_offsets"]
        arr = np.frombuffer(mm, dtype=_DTYPES[dt], count=(e - b) // np.dtype(_DTYPES[dt]).itemsize, offset=b).reshape(info["shape"])
        out[name] = arr.view(Bf16Bits) if dt == "BF16" else arr
    return out


def write_safetensors(path: str, tensors: Dict[str, np.ndarray], bf16_names=()) -> None:
    """Minimal writer (tests, synthetic checkpoints).  Arrays named in `bf16_names` must be uint16 bf16 bit patterns."""
    header, blobs, off = {}, [], 0
    rev = {np.dtype(v): k for k, v in _DTYPES.items() if k not in ("BF16", "BOOL", "U16")}
    for name, a in tensors.items():
        a = np.ascontiguousarray(a)
        dt = "BF16" if name in bf16_names else rev[a.dtype]
        raw = a.tobytes()
        header[name] = {"dtype": dt, "shape": list(a.shape), "data_offsets": [off, off + len(raw)]}
        blobs.append(raw)
        off += len(raw)
    hj = json.dumps(header, separators=(",", ":")).encode("utf-8")
    hj += b" " * ((8 - len(hj) % 8) % 8)
    with open(path, "wb") as f:
        f.write(len(hj).to_bytes(8, "little"))
        f.write(hj)
        for raw in blobs:
            f.write(raw)


def load_all_weights(model_dir: str) -> Dict[str, np.ndarray]:
    """Every tensor of every shard named by model.safetensors.index.json (or the single model.safetensors)."""
    index = os.path.join(model_dir, "model.safetensors.index.json")
    if os.path.exists(index):
        with open(index) as f:
            files = sorted(set(json.load(f)["weight_map"].values()))
    elif os.path.exists(os.path.join(model_dir, "model.safetensors")):
        files = ["model.safetensors"]
    else:
        raise FileNotFoundError(f"{model_dir}: neither model.safetensors.index.json nor model.safetensors")
    out = {}
    for fn in files:
        out.update(read_safetensors(os.path.join(model_dir, fn)))
    return out


def sanitize_weights(weights: Dict[str, np.ndarray], num_hidden_layers: int, num_local_experts: int) -> Dict[str, np.ndarray]:
    """Mixtral: stack `block_sparse_moe.experts.{e}.{w1,w2,w3}.{weight,scales,biases}` over e into
    `block_sparse_moe.switch_mlp.{gate_proj,down_proj,up_proj}.*`; already-stacked checkpoints pass through."""
    if "model.layers.0.block_sparse_moe.experts.0.w1.weight" not in weights:
        return weights
    out = dict(weights)
    for layer in range(num_hidden_layers):
        prefix = f"model.layers.{layer}"
        for old, new in (("w1", "gate_proj"), ("w2", "down_proj"), ("w3", "up_proj")):
            for comp in ("weight", "scales", "biases"):
                if f"{prefix}.block_sparse_moe.experts.0.{old}.{comp}" not in out:
                    continue
                parts = []
                for e in range(num_local_experts):
                    key = f"{prefix}.block_sparse_moe.experts.{e}.{old}.{comp}"
                    if key not in out:
                        raise KeyError(f"WeightNotFound: {key}")
                    parts.append(out.pop(key))
                stacked = np.stack(parts, 0)
                out[f"{prefix}.block_sparse_moe.switch_mlp.{new}.{comp}"] = keep_kind(parts[0], stacked)
    return out


def model_args(model_dir: str) -> dict:
    """config.json -> keyword arguments of engine.Model (ModelArgs, model.rs:47-64)."""
    with open(os.path.join(model_dir, "config.json")) as f:
        c = json.load(f)
    head_dim = c.get("head_dim") or c["hidden_size"] // c["num_attention_heads"]
    args = dict(hidden_size=c["hidden_size"], num_hidden_layers=c["num_hidden_layers"], intermediate_size=c["intermediate_size"],
                num_attention_heads=c["num_attention_heads"], num_key_value_heads=c.get("num_key_value_heads", c["num_attention_heads"]),
                head_dim=head_dim, vocab_size=c["vocab_size"], rms_norm_eps=c.get("rms_norm_eps", 1e-6),
                rope_theta=c.get("rope_theta", 1e6), tie_word_embeddings=c.get("tie_word_embeddings", False),
                rope_scaling=c.get("rope_scaling"), quantization=c.get("quantization"))
    model_type = c.get("model_type", "")
    if model_type == "qwen2":     # qwen3-mlx/src/qwen2.rs: q/k/v bias, no q/k norm
        args.update(attention_bias=True, qk_norm=False)
    if model_type == "mixtral" or "num_local_experts" in c:
        # mixtral-mlx ModelArgs (model.rs:54-80): experts as wide as intermediate_size, top-2 of 8 by default, no q/k norm
        args.update(num_experts=c.get("num_local_experts", 8), num_experts_per_tok=c.get("num_experts_per_tok", 2),
                    moe_intermediate_size=c["intermediate_size"], moe_mode="mixtral", qk_norm=False,
                    rms_norm_eps=c.get("rms_norm_eps", 1e-5))
    elif c.get("num_experts", 0) > 0:
        # qwen3-mlx qwen3_moe ModelArgs (qwen3_moe.rs:60-87); every layer sparse (decoder_sparse_step 1, no mlp_only_layers)
        if c.get("decoder_sparse_step", 1) != 1 or c.get("mlp_only_layers"):
            raise ValueError("load_model: mixed dense / sparse layers (decoder_sparse_step, mlp_only_layers) are not supported")
        args.update(num_experts=c["num_experts"], num_experts_per_tok=c["num_experts_per_tok"],
                    moe_intermediate_size=c["moe_intermediate_size"], moe_mode="qwen3_moe", norm_topk_prob=c.get("norm_topk_prob", False))
    return args


def load_model(model_dir: str, max_context: int = 4096, **overrides):
    """qwen3_mlx::load_model: config.json + shards -> a ready engine.Model (bf16 or MLX-quantized)."""
    from . import engine
    args = dict(model_args(model_dir), max_context=max_context, **overrides)
    weights = load_all_weights(model_dir)
    if args.get("moe_mode") == "mixtral":
        weights = sanitize_weights(weights, args["num_hidden_layers"], args["num_experts"])
    if args.get("quantization") is not None and any(k.endswith(".scales") and np.asarray(v).dtype == np.float16 for k, v in weights.items()):
        args["quantization"] = dict(args["quantization"], scales_dtype="float16")     # a float16 checkpoint's triplets (engine.Model)
    m = engine.Model(**args)
    tied = bool(args["tie_word_embeddings"])
    quant = args.get("quantization") is not None
    keep = {k: v for k, v in weights.items() if not (tied and k.startswith("lm_head."))}
    if quant:
        for k, v in keep.items():
            if k.endswith(".weight") and k[:-7] + ".scales" in keep and v.dtype != np.uint32:
                raise ValueError(f"{k}: a quantized weight must be packed uint32, found {v.dtype}")
    m.load_weights(keep)
    return m
