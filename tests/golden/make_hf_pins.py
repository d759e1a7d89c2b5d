"""Pins the MODEL-LEVEL oracle on an independent implementation (VERDICT r1 "Next" #4).

Runs in the BUILD container only (it imports `transformers`, which never travels to the GPU box): tiny random
Qwen3ForCausalLM / Qwen2ForCausalLM / MixtralForCausalLM / Qwen3MoeForCausalLM models in fp32 on the CPU, weights rounded to
bf16-representable values first (so that the bf16 engine later loads exactly the tensors the HF model ran on).  For each model:
    1. the oracle (oracle/ref_qwen3.py, dt="f32": float64 accumulation, one f32 rounding per op output) must reproduce the HF
       logits of every prompt position and of greedy decode steps to <= 1e-4 of the largest logit -- checked HERE, the script
       fails otherwise -- and the same greedy tokens;
    2. weights (bf16 bits), prompt, HF logits and HF greedy tokens are written to tests/golden/hf_<name>.npz.
tests/test_oracle_pins.py re-checks (1) from the fixture on any machine; tests/test_gpu_hf_pins.py runs the bf16 ENGINE on the
fixture weights against the HF logits.  Also checked here: oracle SDPA vs torch.nn.functional.scaled_dot_product_attention
(that part needs no fixture -- torch is on every box -- and is repeated as a plain CPU test).

    python tests/golden/make_hf_pins.py
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import ref_core as rc, ref_qwen3 as rq  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def bf16_bits(a: np.ndarray) -> np.ndarray:
    return (rc.bf16_round(a.astype(np.float32)).view(np.uint32) >> np.uint32(16)).astype(np.uint16)


def build(kind: str):
    """(hf_model, oracle config) of one tiny architecture; dims chosen so that every engine kernel constraint holds."""
    from transformers import (MixtralConfig, MixtralForCausalLM, Qwen2Config, Qwen2ForCausalLM, Qwen3Config, Qwen3ForCausalLM,
                              Qwen3MoeConfig, Qwen3MoeForCausalLM)
    # small on purpose (the fixtures are committed): hidden 128, 2 layers, vocabulary 256 -- still GQA, both head widths, both MoEs
    common = dict(hidden_size=128, num_hidden_layers=2, num_attention_heads=4, num_key_value_heads=2, vocab_size=256,
                  max_position_embeddings=512, rope_theta=1e6, attention_dropout=0.0, use_cache=False)
    if kind == "qwen3":
        cfg = Qwen3Config(intermediate_size=384, head_dim=64, rms_norm_eps=1e-6, tie_word_embeddings=False, **common)
        return Qwen3ForCausalLM(cfg), rq.Qwen3Config(128, 2, 384, 4, 2, 64, 256, 1e-6, 1e6, False)
    if kind == "qwen3_tied_d128":
        c2 = dict(common, num_attention_heads=4, num_key_value_heads=1)
        cfg = Qwen3Config(intermediate_size=256, head_dim=128, rms_norm_eps=1e-6, tie_word_embeddings=True, **c2)
        return Qwen3ForCausalLM(cfg), rq.Qwen3Config(128, 2, 256, 4, 1, 128, 256, 1e-6, 1e6, True)
    if kind == "qwen2":
        c2 = dict(common, num_attention_heads=2, num_key_value_heads=1)             # head_dim = hidden / heads = 64
        cfg = Qwen2Config(intermediate_size=384, rms_norm_eps=1e-6, tie_word_embeddings=False, **c2)
        return Qwen2ForCausalLM(cfg), rq.Qwen3Config(128, 2, 384, 2, 1, 64, 256, 1e-6, 1e6, False, qk_norm=False, attention_bias=True)
    if kind == "mixtral":
        cfg = MixtralConfig(intermediate_size=128, head_dim=64, rms_norm_eps=1e-5, num_local_experts=4, num_experts_per_tok=2,
                            tie_word_embeddings=False, router_jitter_noise=0.0, sliding_window=None, **common)
        return MixtralForCausalLM(cfg), rq.Qwen3Config(128, 2, 128, 4, 2, 64, 256, 1e-5, 1e6, False, num_experts=4, num_experts_per_tok=2,
                                                        moe_intermediate_size=128, moe_mode="mixtral", qk_norm=False)
    if kind == "qwen3_moe":
        cfg = Qwen3MoeConfig(intermediate_size=384, moe_intermediate_size=64, head_dim=64, rms_norm_eps=1e-6, num_experts=8,
                             num_experts_per_tok=2, norm_topk_prob=True, decoder_sparse_step=1, mlp_only_layers=[],
                             tie_word_embeddings=False, router_aux_loss_coef=0.0, **common)
        return Qwen3MoeForCausalLM(cfg), rq.Qwen3Config(128, 2, 384, 4, 2, 64, 256, 1e-6, 1e6, False, num_experts=8, num_experts_per_tok=2,
                                                        moe_intermediate_size=64, moe_mode="qwen3_moe", norm_topk_prob=True)
    raise ValueError(kind)


def to_oracle_names(kind: str, sd: dict, cfg: rq.Qwen3Config) -> dict:
    """HF state_dict -> the checkpoint keys the reference loads (expert stacks: sanitize_weights of mixtral-mlx/src/model.rs:480-510;
    transformers 5 keeps experts fused as gate_up_proj [E, 2I, h] = [gate; up] and down_proj [E, h, I])."""
    out = {}
    for k, v in sd.items():
        a = v.detach().float().numpy()
        if ".mlp.experts.gate_up_proj" in k:
            base = k.replace(".mlp.experts.gate_up_proj", "")
            mp = base + (".block_sparse_moe." if cfg.moe_mode == "mixtral" else ".mlp.")
            half = a.shape[1] // 2
            out[mp + "switch_mlp.gate_proj.weight"], out[mp + "switch_mlp.up_proj.weight"] = a[:, :half], a[:, half:]
        elif ".mlp.experts.down_proj" in k:
            base = k.replace(".mlp.experts.down_proj", "")
            out[base + (".block_sparse_moe." if cfg.moe_mode == "mixtral" else ".mlp.") + "switch_mlp.down_proj.weight"] = a
        elif k.endswith(".mlp.gate.weight") and cfg.moe_mode == "mixtral" and cfg.num_experts:
            out[k.replace(".mlp.gate.weight", ".block_sparse_moe.gate.weight")] = a
        elif k == "lm_head.weight" and cfg.tie_word_embeddings:
            continue
        else:
            out[k] = a
    return out


def pin(kind: str, seed: int, n_prompt: int = 24, n_new: int = 4):
    torch.manual_seed(seed)
    model, cfg = build(kind)
    model = model.float().eval()
    with torch.no_grad():
        for p in model.parameters():
            p.mul_(3.0 if p.ndim > 1 else 1.0)      # livelier logits than the 0.02 init gives at 2 layers
            p.copy_(p.bfloat16().float())            # bf16-representable: the engine loads exactly these values
    weights = to_oracle_names(kind, model.state_dict(), cfg)
    prompt = ((np.arange(n_prompt, dtype=np.int64) * 7919 + 13) % cfg.vocab_size).astype(np.uint32)
    ids = [int(t) for t in prompt]
    hf_logits, hf_tokens = [], []
    with torch.no_grad():
        full = model(torch.tensor([ids])).logits[0].numpy()           # every prompt position
        for _ in range(n_new):
            lg = model(torch.tensor([ids])).logits[0, -1].numpy()
            hf_logits.append(lg)
            ids.append(int(lg.argmax()))
            hf_tokens.append(ids[-1])
    hf_logits = np.stack(hf_logits)
    # ---- the oracle in f32 against it ----
    oracle = rq.Qwen3Oracle(cfg, weights, dt="f32")
    tokens, logits = oracle.generate(prompt, n_new, return_logits=True)
    scale = np.abs(hf_logits).max()
    err = np.abs(logits - hf_logits).max() / scale
    all_pos = oracle.forward(prompt[None, :].astype(np.int64), [])[0]            # lm_head on every prompt position (model.rs:480-489)
    err_all = np.abs(all_pos - full).max() / np.abs(full).max()
    print(f"{kind:18s} oracle(f32) vs transformers fp32: decode-step logits rel err {err:.2e}, all prompt positions {err_all:.2e}, "
          f"tokens {list(map(int, tokens))} vs {hf_tokens}")
    assert err <= 1e-4 and err_all <= 1e-4, f"{kind}: oracle disagrees with transformers"
    assert [int(t) for t in tokens] == hf_tokens, f"{kind}: greedy tokens differ"
    import dataclasses
    import json
    np.savez_compressed(os.path.join(OUT, f"hf_{kind}.npz"), cfg=np.array(json.dumps(dataclasses.asdict(cfg))), prompt=prompt,
                        hf_logits=hf_logits.astype(np.float32),
                        hf_prompt_logits=full.astype(np.float32), hf_tokens=np.array(hf_tokens, np.uint32),
                        **{"w:" + k: bf16_bits(v) for k, v in weights.items()})


def sdpa_against_torch():
    """oracle.scaled_dot_product_attention (f32) vs torch's, all mask kinds, GQA, Tq != Tk."""
    import torch.nn.functional as F
    g = np.random.default_rng(0)
    for (B, H, Hkv, Tq, Tk, D) in [(1, 8, 2, 1, 37, 64), (2, 4, 4, 9, 9, 32), (1, 6, 3, 5, 21, 128)]:
        q, k, v = (g.standard_normal(s).astype(np.float32) for s in ((B, H, Tq, D), (B, Hkv, Tk, D), (B, Hkv, Tk, D)))
        scale = D ** -0.5
        rep = H // Hkv
        kt, vt = torch.tensor(k).repeat_interleave(rep, 1), torch.tensor(v).repeat_interleave(rep, 1)
        bool_mask = g.random((Tq, Tk)) > 0.3
        bool_mask[:, 0] = True
        add_mask = (g.standard_normal((Tq, Tk)) * 2).astype(np.float32)
        causal = np.tril(np.ones((Tq, Tk), bool), Tk - Tq)
        for name, om, tm in (("none", None, None), ("causal", "causal", torch.tensor(causal)), ("bool", bool_mask, torch.tensor(bool_mask)),
                             ("additive", add_mask, torch.tensor(add_mask))):
            want = F.scaled_dot_product_attention(torch.tensor(q), kt, vt, attn_mask=tm, scale=scale).numpy()
            got = rc.scaled_dot_product_attention(q, k, v, scale, om, "f32")
            e = np.abs(got - want).max()
            assert e <= 2e-6 * max(1.0, np.abs(want).max()), (name, e)
    print("sdpa: oracle == torch.nn.functional.scaled_dot_product_attention (none / causal / bool / additive, GQA)")


if __name__ == "__main__":
    sdpa_against_torch()
    for i, kind in enumerate(["qwen3", "qwen3_tied_d128", "qwen2", "mixtral", "qwen3_moe"]):
        pin(kind, seed=100 + i)
