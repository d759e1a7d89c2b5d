"""Tensor-parallel shard plan of the dense decoder.

Host logic shared by every consumer of the TP path: `engine.Model.load_weights` slices host
checkpoints with it, the C++ engine's synthetic generator follows the same plan
(`omx_qwen3_synth_weights`: row0/col0 of `omx_fill_uniform_2d`), and the world-size-2 gloo test
(tests/test_tp_plan.py) checks it against the single-device oracle.

    row split  (output features): q_proj, k_proj, v_proj, gate_proj, up_proj, lm_head
    col split  (input features):  o_proj, down_proj   -> partial sums, one all-reduce(sum) each
    replicated:                   norms, embedding table (the tied head uses a row shard of it)
Heads are contiguous per rank: rank r owns q heads [r*H/n, (r+1)*H/n) and KV heads [r*Hkv/n, ...).
"""
from __future__ import annotations

import numpy as np

from .loader import keep_kind

ROW_SPLIT = ("self_attn.q_proj.weight", "self_attn.k_proj.weight", "self_attn.v_proj.weight",
             "mlp.gate_proj.weight", "mlp.up_proj.weight", "lm_head.weight",
             # Qwen2's projection biases (qwen3-mlx/src/qwen2.rs:112-124) follow the rows of their Linear
             "self_attn.q_proj.bias", "self_attn.k_proj.bias", "self_attn.v_proj.bias")
COL_SPLIT = ("self_attn.o_proj.weight", "mlp.down_proj.weight")
# sparse-MoE models under tensor parallelism ("expert tensor parallel"): every expert's intermediate columns are split, the router is
# replicated -- stacks [E, I, hidden] (gate / up: rows of each expert) and [E, hidden, I] (down: columns of each expert)
EXPERT_ROW_SPLIT = ("switch_mlp.gate_proj.weight", "switch_mlp.up_proj.weight")
EXPERT_COL_SPLIT = ("switch_mlp.down_proj.weight",)


# quantized checkpoints (round 4): "<stem>.weight" is the packed uint32 matrix [out, in * bits / 32], ".scales" / ".biases" are
# [out, in / group] -- rows split like the dense rows, K slices (whole groups: in / world is a multiple of the group size) as equal
# parts of the last axis of all three
QUANT_LEAVES = (".scales", ".biases")


def _split_kind(name: str, table) -> bool:
    if name.endswith(table):
        return True
    return name.endswith(QUANT_LEAVES) and (name.rsplit(".", 1)[0] + ".weight").endswith(table)


def kv_replication(num_key_value_heads: int, world: int) -> int:
    """How many ranks share one KV head: 1 while every rank owns whole KV heads; world / Hkv once there are fewer KV heads than
    ranks ("Qwen2.5-7B has Hkv=4 -> replicate KV heads x2 at TP=8") -- rank r then holds KV head r // rep."""
    return 1 if num_key_value_heads >= world else world // num_key_value_heads


def check_divisible(*, num_attention_heads, num_key_value_heads, intermediate_size, vocab_size, world, **_):
    for what, n in (("num_attention_heads", num_attention_heads), ("intermediate_size", intermediate_size), ("vocab_size", vocab_size)):
        if n % world:
            raise ValueError(f"InvalidConfig: {what}={n} is not divisible by tp_size={world}")
    if num_key_value_heads >= world:
        if num_key_value_heads % world:
            raise ValueError(f"InvalidConfig: num_key_value_heads={num_key_value_heads} is not divisible by tp_size={world}")
    else:
        rep = world // num_key_value_heads
        group = num_attention_heads // num_key_value_heads
        if world % num_key_value_heads or group % rep:
            raise ValueError(f"InvalidConfig: num_key_value_heads={num_key_value_heads} cannot be replicated over tp_size={world} "
                             f"(query group of {group} heads, {rep} ranks per KV head)")


def shard(name: str, arr: np.ndarray, rank: int, world: int, num_key_value_heads: int = 0, head_dim: int = 0) -> np.ndarray:
    """This rank's slice of a logical weight (contiguous copy).  With fewer KV heads than ranks (pass num_key_value_heads and
    head_dim) the k / v projections are not split further: rank r takes the rows (and bias entries) of KV head r // rep."""
    if world == 1:
        return arr
    if 0 < num_key_value_heads < world and (".self_attn.k_proj." in name or ".self_attn.v_proj." in name):
        head = rank // kv_replication(num_key_value_heads, world)
        return keep_kind(arr, np.ascontiguousarray(arr[head * head_dim:(head + 1) * head_dim]))
    # (packed expert stacks, round 5: the triplet of a slice is the slice of the triplet -- rows of every expert's gate / up on all three
    #  leaves; equal parts of the last axis, whole groups, of the down projection's packed words, scales and biases)
    if _split_kind(name, EXPERT_ROW_SPLIT):
        n = arr.shape[1] // world
        return keep_kind(arr, np.ascontiguousarray(arr[:, rank * n:(rank + 1) * n, :]))
    if _split_kind(name, EXPERT_COL_SPLIT):
        if arr.shape[2] % world:
            raise ValueError(f"InvalidConfig: {name} has a last axis of {arr.shape[2]}, not divisible by tp_size={world}")
        n = arr.shape[2] // world
        return keep_kind(arr, np.ascontiguousarray(arr[:, :, rank * n:(rank + 1) * n]))
    if _split_kind(name, ROW_SPLIT):
        n = arr.shape[0] // world
        return keep_kind(arr, np.ascontiguousarray(arr[rank * n:(rank + 1) * n]))
    if _split_kind(name, COL_SPLIT):
        if arr.shape[1] % world:
            raise ValueError(f"InvalidConfig: {name} has {arr.shape[1]} columns, not divisible by tp_size={world}")
        n = arr.shape[1] // world
        return keep_kind(arr, np.ascontiguousarray(arr[:, rank * n:(rank + 1) * n]))
    return arr


def shard_state_dict(weights: dict, rank: int, world: int, tie_word_embeddings: bool = False, num_key_value_heads: int = 0,
                     head_dim: int = 0) -> dict:
    out = {k: shard(k, v, rank, world, num_key_value_heads, head_dim) for k, v in weights.items()}
    if tie_word_embeddings and world > 1:
        for leaf in (".weight",) + QUANT_LEAVES:      # (a quantized table: the shard of its whole triplet)
            if "model.embed_tokens" + leaf in weights:
                out["lm_head" + leaf] = shard("lm_head" + leaf, weights["model.embed_tokens" + leaf], rank, world)
    return out


def argmax_key(value: float, index: int) -> int:
    """Packed (orderable float32 bits << 32) | ~index: a plain unsigned max over ranks picks the largest
    logit and, on ties, the smallest vocabulary index -- the single-device argmax (sampler.rs:9-12)."""
    u = int(np.float32(value).view(np.uint32))
    u = (~u & 0xFFFFFFFF) if (u & 0x80000000) else (u | 0x80000000)
    if value != value:
        u = 0
    return (u << 32) | (~index & 0xFFFFFFFF)


def key_to_index(key: int) -> int:
    return ~key & 0xFFFFFFFF
