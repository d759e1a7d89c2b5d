"""TEST INFRASTRUCTURE ONLY -- CPU (numpy) restatement of the qwen3-mlx dense decoder
forward and greedy generation loop, i.e. the *caller* that drives the mlx-rs-core hot
path.  Never imported by the product path.

Follows, op by op (each op's output rounded to the activation dtype, as MLX does):
  Attention::forward        qwen3-mlx/src/model.rs:161-215
  Mlp::forward              qwen3-mlx/src/model.rs:263-267
  TransformerBlock::forward qwen3-mlx/src/model.rs:321-332
  Qwen3Model::forward       qwen3-mlx/src/model.rs:394-424
  Model::forward            qwen3-mlx/src/model.rs:480-490   (lm_head or tied embedding)
  sample / Generate         qwen3-mlx/src/model.rs:733-741, 804-843
PARITY UNPINNED at model level: the reference holds no golden token ids / logits
(qwen3-mlx has 0 tests); the primitives used here are pinned
individually in tests/test_oracle_kats.py.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, List, Optional

import numpy as np

from . import ref_core as rc
from . import synth


@dataclass
class Qwen3Config:
    """Fields of ModelArgs (qwen3-mlx/src/model.rs:47-64)."""
    hidden_size: int = 256
    num_hidden_layers: int = 2
    intermediate_size: int = 768
    num_attention_heads: int = 4
    num_key_value_heads: int = 2
    head_dim: int = 64
    vocab_size: int = 1024
    rms_norm_eps: float = 1e-6
    rope_theta: float = 1e6
    tie_word_embeddings: bool = False
    rope_scaling: Optional[dict] = None
    max_position_embeddings: int = 40960
    # sparse-MoE feed-forward in every layer (0 = dense): qwen3-mlx/src/qwen3_moe.rs ModelArgs :60-87 ("qwen3_moe") or
    # mixtral-mlx/src/model.rs ModelArgs :54-80 ("mixtral", which also has no q/k norm: qk_norm False)
    num_experts: int = 0
    num_experts_per_tok: int = 0
    moe_intermediate_size: int = 0
    moe_mode: str = "qwen3_moe"
    norm_topk_prob: bool = False
    qk_norm: bool = True
    attention_bias: bool = False     # Qwen2 (qwen3-mlx/src/qwen2.rs:112-124): q/k/v Linear with bias, and qk_norm False

    @staticmethod
    def qwen3_8b():
        return Qwen3Config(4096, 36, 12288, 32, 8, 128, 151936, 1e-6, 1e6, False)

    @staticmethod
    def qwen3_0_6b():
        return Qwen3Config(1024, 28, 3072, 16, 8, 128, 151936, 1e-6, 1e6, True)


def weight_shapes(cfg: Qwen3Config) -> Dict[str, tuple]:
    """HF key names the loader expects (qwen3-mlx/src/model.rs:631-716)."""
    h, I, H, Hkv, D, V = (cfg.hidden_size, cfg.intermediate_size, cfg.num_attention_heads,
                          cfg.num_key_value_heads, cfg.head_dim, cfg.vocab_size)
    s = {"model.embed_tokens.weight": (V, h), "model.norm.weight": (h,)}
    for i in range(cfg.num_hidden_layers):
        p = f"model.layers.{i}."
        s[p + "self_attn.q_proj.weight"] = (H * D, h)
        s[p + "self_attn.k_proj.weight"] = (Hkv * D, h)
        s[p + "self_attn.v_proj.weight"] = (Hkv * D, h)
        s[p + "self_attn.o_proj.weight"] = (h, H * D)
        if cfg.attention_bias:
            s[p + "self_attn.q_proj.bias"] = (H * D,)
            s[p + "self_attn.k_proj.bias"] = (Hkv * D,)
            s[p + "self_attn.v_proj.bias"] = (Hkv * D,)
        if cfg.qk_norm:
            s[p + "self_attn.q_norm.weight"] = (D,)
            s[p + "self_attn.k_norm.weight"] = (D,)
        if cfg.num_experts:
            mp = p + ("block_sparse_moe." if cfg.moe_mode == "mixtral" else "mlp.")
            E, Im = cfg.num_experts, cfg.moe_intermediate_size
            s[mp + "gate.weight"] = (E, h)
            s[mp + "switch_mlp.gate_proj.weight"] = (E, Im, h)
            s[mp + "switch_mlp.up_proj.weight"] = (E, Im, h)
            s[mp + "switch_mlp.down_proj.weight"] = (E, h, Im)
        else:
            s[p + "mlp.gate_proj.weight"] = (I, h)
            s[p + "mlp.up_proj.weight"] = (I, h)
            s[p + "mlp.down_proj.weight"] = (h, I)
        s[p + "input_layernorm.weight"] = (h,)
        s[p + "post_attention_layernorm.weight"] = (h,)
    if not cfg.tie_word_embeddings:
        s["lm_head.weight"] = (V, h)
    return s


def weight_spec(name: str):
    """(std, offset) of the synthetic generator per tensor class:
    matrices N(0,0.02^2)-like, norm weights 1 + small noise."""
    if name.endswith("norm.weight") or name.endswith("layernorm.weight"):
        return 0.01, 1.0
    return 0.02, 0.0


def synth_weights(cfg: Qwen3Config, dt: str = "bf16", peaked: bool = False) -> Dict[str, np.ndarray]:
    """peaked: the twin of omx_qwen3_synth_weights_peaked -- embedding std 64 (it dominates the layers' ~10-rms contribution to the
    residual stream) and lm_head[v] = row (v + 1) mod V of the same table at std 0.02, so the greedy successor of token t is t - 1 with a
    top-1 margin far above the bf16 bound while the other logits still carry the layers' arithmetic."""
    out = {}
    for name, shape in weight_shapes(cfg).items():
        std, off = weight_spec(name)
        out[name] = synth.tensor(name, shape, std, off, dt)
    if peaked:
        assert "lm_head.weight" in out, "peaked weights need an untied lm_head"
        shape = out["model.embed_tokens.weight"].shape
        out["model.embed_tokens.weight"] = synth.tensor("model.embed_tokens.weight", shape, 64.0, 0.0, dt)
        out["lm_head.weight"] = np.roll(synth.tensor("model.embed_tokens.weight", shape, 0.02, 0.0, dt), -1, axis=0)
    return out


QUANTIZED = ("self_attn.q_proj", "self_attn.k_proj", "self_attn.v_proj", "self_attn.o_proj", "mlp.gate_proj", "mlp.up_proj",
             "mlp.down_proj", "mlp.gate", "block_sparse_moe.gate", "switch_mlp.gate_proj", "switch_mlp.up_proj", "switch_mlp.down_proj")


def quantize_weights(cfg: Qwen3Config, weights: Dict[str, np.ndarray], bits: int = 4, group_size: int = 64) -> Dict[str, np.ndarray]:
    """What an MLX-quantized checkpoint of this model holds (qwen3-mlx/src/model.rs:621-727): every Linear and the
    embedding as (weight uint32, scales, biases) triplets -- mlx quantize() of the bf16 tensors, scales/biases in bf16;
    norm weights unchanged."""
    out = {}
    for name, w in weights.items():
        prefix = name[:-len(".weight")]
        if prefix.endswith(QUANTIZED) or prefix in ("model.embed_tokens", "lm_head"):
            w2 = w.reshape(-1, w.shape[-1])                     # expert stacks [E, out, in]: quantised row by row like any matrix
            q, s, b = rc.quantize(w2, group_size, bits)
            lead = w.shape[:-1]
            out[prefix + ".weight"] = q.reshape(*lead, -1)
            out[prefix + ".scales"], out[prefix + ".biases"] = rc.bf16_round(s).reshape(*lead, -1), rc.bf16_round(b).reshape(*lead, -1)
        else:
            out[name] = w
    return out


class Qwen3Oracle:
    """quant = (bits, group_size): `weights` is a quantized checkpoint (quantize_weights); every Linear becomes
    quantized_matmul (nn/quantized.rs:366-375), the embedding dequantises the gathered rows (:192-203) and a tied head is
    QuantizedEmbedding::as_linear (:166-180)."""

    def __init__(self, cfg: Qwen3Config, weights: Dict[str, np.ndarray], dt: str = "bf16", quant=None):
        self.cfg, self.w, self.dt, self.quant = cfg, weights, dt, quant
        self.rope = rc.initialize_rope(cfg.head_dim, cfg.rope_theta, False, cfg.rope_scaling)
        self.scale = float(np.float32(1.0) / np.sqrt(np.float32(cfg.head_dim)))

    def lin(self, x, prefix: str):
        if self.quant is None:
            return rc.linear(x, self.w[prefix + ".weight"], None, self.dt)
        bits, group = self.quant
        return rc.quantized_matmul(x, self.w[prefix + ".weight"], self.w[prefix + ".scales"], self.w[prefix + ".biases"], group,
                                   bits, self.dt)

    # model.rs:161-215
    def attention(self, i: int, x, mask, cache):
        cfg, dt = self.cfg, self.dt
        p = f"model.layers.{i}.self_attn."
        B, L, _ = x.shape
        if cfg.attention_bias:      # qwen2.rs:172-174
            q = rc.linear(x, self.w[p + "q_proj.weight"], self.w[p + "q_proj.bias"], dt)
            k = rc.linear(x, self.w[p + "k_proj.weight"], self.w[p + "k_proj.bias"], dt)
            v = rc.linear(x, self.w[p + "v_proj.weight"], self.w[p + "v_proj.bias"], dt)
        else:
            q = self.lin(x, p + "q_proj")
            k = self.lin(x, p + "k_proj")
            v = self.lin(x, p + "v_proj")
        q = q.reshape(B, L, cfg.num_attention_heads, -1).transpose(0, 2, 1, 3)
        k = k.reshape(B, L, cfg.num_key_value_heads, -1).transpose(0, 2, 1, 3)
        v = v.reshape(B, L, cfg.num_key_value_heads, -1).transpose(0, 2, 1, 3)
        if cfg.qk_norm:      # Qwen3 (model.rs:181-184); Mixtral attention has none (mixtral model.rs:120-160)
            q = rc.rms_norm(q, self.w[p + "q_norm.weight"], cfg.rms_norm_eps, dt)
            k = rc.rms_norm(k, self.w[p + "k_norm.weight"], cfg.rms_norm_eps, dt)
        off = cache.offset()
        r = self.rope
        q = rc.rope(q, r["dims"], r["traditional"], r["base"], r["scale"], off, dt)
        k = rc.rope(k, r["dims"], r["traditional"], r["base"], r["scale"], off, dt)
        k, v = cache.update_and_fetch(k, v)
        if mask is not None:
            m = mask
        elif L > 1:
            m = "causal"
        else:
            m = None
        o = rc.scaled_dot_product_attention(q, k, v, self.scale, m, dt)
        o = o.transpose(0, 2, 1, 3).reshape(B, L, -1)
        return self.lin(o, p + "o_proj")

    # model.rs:263-267
    def mlp(self, i: int, x):
        p = f"model.layers.{i}.mlp."
        dt = self.dt
        cfg = self.cfg
        if cfg.num_experts:   # MoeBlock::forward (qwen3_moe.rs:475-503) / MixtralSparseMoeBlock::forward (mixtral model.rs:296-308)
            from . import ref_moe
            mp = f"model.layers.{i}." + ("block_sparse_moe." if cfg.moe_mode == "mixtral" else "mlp.")
            B, L, h = x.shape
            if self.quant is not None:   # QuantizedLinear gate + QuantizedSwitchLinear experts (mixtral model.rs:560-600)
                bits, group = self.quant
                x2 = x.reshape(B * L, h)
                logits = rc.quantized_matmul(x2, self.w[mp + "gate.weight"], self.w[mp + "gate.scales"], self.w[mp + "gate.biases"], group, bits, dt)
                inds, scores = (ref_moe.route_logits_mixtral(logits, cfg.num_experts_per_tok, dt) if cfg.moe_mode == "mixtral"
                                else ref_moe.route_logits_qwen3_moe(logits, cfg.num_experts_per_tok, cfg.norm_topk_prob, dt))
                trip = lambda nm: tuple(self.w[mp + f"switch_mlp.{nm}.{c}"] for c in ("weight", "scales", "biases"))
                y = ref_moe.switch_glu_q(x2, inds, trip("gate_proj"), trip("up_proj"), trip("down_proj"), group, bits, dt)
                weighted = rc.rnd(y.astype(np.float64) * scores[..., None].astype(np.float64), dt)
                return rc.rnd(np.sum(weighted.astype(np.float64), axis=1), dt).reshape(B, L, h)
            y, _, _ = ref_moe.moe_block(x.reshape(B * L, h), self.w[mp + "gate.weight"], self.w[mp + "switch_mlp.gate_proj.weight"],
                                        self.w[mp + "switch_mlp.up_proj.weight"], self.w[mp + "switch_mlp.down_proj.weight"],
                                        cfg.num_experts_per_tok, cfg.moe_mode, cfg.norm_topk_prob, dt)
            return y.reshape(B, L, h)
        g = self.lin(x, p + "gate_proj")
        u = self.lin(x, p + "up_proj")
        act = rc.multiply(rc.silu(g, dt), u, dt)
        return self.lin(act, p + "down_proj")

    # model.rs:321-332
    def block(self, i: int, x, mask, cache):
        cfg, dt = self.cfg, self.dt
        p = f"model.layers.{i}."
        xn = rc.rms_norm(x, self.w[p + "input_layernorm.weight"], cfg.rms_norm_eps, dt)
        h = rc.add(x, self.attention(i, xn, mask, cache), dt)
        hn = rc.rms_norm(h, self.w[p + "post_attention_layernorm.weight"], cfg.rms_norm_eps, dt)
        return rc.add(h, self.mlp(i, hn), dt)

    # model.rs:394-424 + 480-490
    def forward(self, tokens: np.ndarray, caches: List):
        """tokens [B, L] -> logits [B, L, V] (lm_head applied to ALL positions, as the reference does)."""
        cfg = self.cfg
        tokens = np.asarray(tokens)
        if self.quant is None:
            h = self.w["model.embed_tokens.weight"][tokens]      # Embedding gather
        else:
            bits, group = self.quant
            h = rc.dequantize(self.w["model.embed_tokens.weight"][tokens], self.w["model.embed_tokens.scales"][tokens],
                              self.w["model.embed_tokens.biases"][tokens], group, bits, self.dt)
        T = h.shape[1]
        off = caches[0].offset() if caches else None
        m = rc.create_attention_mask(T, off, None, True)        # every caller passes Some(true) (model.rs:401)
        mask = m if isinstance(m, np.ndarray) else None
        if not caches:
            caches.extend(rc.KVCache() for _ in range(cfg.num_hidden_layers))
        for i in range(cfg.num_hidden_layers):
            h = self.block(i, h, mask, caches[i])
        h = rc.rms_norm(h, self.w["model.norm.weight"], cfg.rms_norm_eps, self.dt)
        return self.lin(h, "model.embed_tokens" if cfg.tie_word_embeddings else "lm_head")

    # flux-klein-mlx/src/qwen3_encoder.rs:141-224 (attention under causal AND padding mask, additive -1e9 in the
    # activation dtype), :403-455 (`forward_with_hidden_states`, `encode`): raw hidden states after the tapped layers,
    # concatenated on the last axis.  The blocks are the decoder's (q/k norm, RoPE offset 0, SwiGLU MLP); no KV cache.
    def encode(self, input_ids, attention_mask=None, extract_layers=(8, 17, 26)) -> np.ndarray:
        ids = np.asarray(input_ids).reshape(1, -1)
        T = ids.shape[1]
        if self.quant is None:
            h = self.w["model.embed_tokens.weight"][ids]
        else:   # a packed checkpoint's QuantizedEmbedding (nn/quantized.rs:252-283), as in forward()
            bits, group = self.quant
            h = rc.dequantize(self.w["model.embed_tokens.weight"][ids], self.w["model.embed_tokens.scales"][ids],
                              self.w["model.embed_tokens.biases"][ids], group, bits, self.dt)
        mask = None
        if attention_mask is not None:
            am = np.asarray(attention_mask).reshape(-1) != 0
            keep = (np.arange(T)[None, :] <= np.arange(T)[:, None]) & am[None, :]
            neg = rc.rnd(np.float32(-1e9), self.dt)
            mask = ((np.float32(1.0) - keep.astype(np.float32)) * neg).astype(np.float32)[None, None]
        taps = []
        for i in range(max(extract_layers) + 1):
            h = self.block(i, h, mask, rc.KVCache())
            if i in extract_layers:
                taps.append(h)
        return np.concatenate(taps, axis=-1)[0]

    # model.rs:804-843 (yield order == plain sequential decoding); temp / seed: model.rs:733-741, 785, 815 with the
    # global RandomState of mlx-rs/src/random.rs:21-41 seeded by `random::seed(seed)` (one split per sampled token)
    def generate(self, prompt: np.ndarray, n_new: int, caches: Optional[List] = None, return_logits: bool = False,
                 temp: float = 0.0, seed: int = 0):
        from . import mlx_rng
        state = mlx_rng.RandomState(seed)
        pick = (lambda l: rc.sample_greedy(l)) if temp == 0.0 else (lambda l: rc.sample(l, temp, state.next()))
        caches = [] if caches is None else caches
        logits = self.forward(np.asarray(prompt)[None, :], caches)
        last = logits[:, -1, :]
        y = pick(last)
        toks, all_logits = [int(y[0])], [last[0]]
        for _ in range(n_new - 1):
            logits = self.forward(y[:, None].astype(np.int64), caches)
            last = logits[:, -1, :]
            y = pick(last)
            toks.append(int(y[0]))
            all_logits.append(last[0])
        if return_logits:
            return np.array(toks, dtype=np.uint32), np.stack(all_logits)
        return np.array(toks, dtype=np.uint32)
