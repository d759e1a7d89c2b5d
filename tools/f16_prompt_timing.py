#!/usr/bin/env python3
"""Time a float16 checkpoint's prompt at Qwen3-8B width: the float16 matrix-core pass (round 4) against the token-serial decode form.

A float16-scale model takes uploaded triplets only, so this builds random 4-bit triplets (uint32 nibbles, float16 scales / biases
that centre the weights) for OMX_F16_LAYERS layers (default 6) of the 8B shapes and reports ms per layer and the 36-layer
extrapolation (the embedding gather and the head are outside the per-layer figure: two layer counts are timed and differenced).

    python tools/f16_prompt_timing.py            # on a GPU box
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import omx_import  # noqa: E402

omx_import.load_package()
from ominix_mlx_amd import engine  # noqa: E402

HD, D, H, HKV, I, V = 4096, 128, 32, 8, 12288, 151936
BITS, GROUP = 4, 64
MOE = os.environ.get("OMX_F16_MOE", "0") == "1"      # Mixtral-8x7B shapes: 8 experts of 14336, top-2, no q / k norm, vocab 32000
if MOE:
    I, V, E, TOPK = 14336, 32000, 8, 2


def triplet(rng, n, k):
    w = rng.integers(0, 2 ** 32, size=(n, k * BITS // 32), dtype=np.uint32)
    s = (rng.random((n, k // GROUP), dtype=np.float32) * 0.002 + 0.002).astype(np.float16)
    b = (-7.5 * s.astype(np.float32)).astype(np.float16)
    return w, s, b


def weights(layers):
    rng = np.random.default_rng(7)
    out = {}

    def put(stem, n, k):
        out[stem + ".weight"], out[stem + ".scales"], out[stem + ".biases"] = triplet(rng, n, k)

    put("model.embed_tokens", V, HD)
    put("lm_head", V, HD)
    out["model.norm.weight"] = np.ones(HD, np.float32)
    for l in range(layers):
        p = f"model.layers.{l}."
        put(p + "self_attn.q_proj", H * D, HD)
        put(p + "self_attn.k_proj", HKV * D, HD)
        put(p + "self_attn.v_proj", HKV * D, HD)
        put(p + "self_attn.o_proj", HD, H * D)
        if MOE:
            put(p + "block_sparse_moe.gate", E, HD)
            for nm, (n_, k_) in (("gate_proj", (I, HD)), ("up_proj", (I, HD)), ("down_proj", (HD, I))):
                w, sc, b = triplet(rng, E * n_, k_)
                stem = p + "block_sparse_moe.switch_mlp." + nm
                out[stem + ".weight"], out[stem + ".scales"], out[stem + ".biases"] = (w.reshape(E, n_, -1), sc.reshape(E, n_, -1), b.reshape(E, n_, -1))
        else:
            put(p + "mlp.gate_proj", I, HD)
            put(p + "mlp.up_proj", I, HD)
            put(p + "mlp.down_proj", HD, I)
        norms = (("input_layernorm", HD), ("post_attention_layernorm", HD)) + (() if MOE else (("self_attn.q_norm", D), ("self_attn.k_norm", D)))
        for n_, w_ in norms:
            out[p + n_ + ".weight"] = np.ones(w_, np.float32)
    return out


def run(layers, n_prompt, serial, reps=3):
    os.environ["OMX_PREFILL_SERIAL"] = "1" if serial else "0"
    m = engine.Model(hidden_size=HD, num_hidden_layers=layers, intermediate_size=I, num_attention_heads=H, num_key_value_heads=HKV,
                     head_dim=D, vocab_size=V, rms_norm_eps=1e-6, rope_theta=1e6, tie_word_embeddings=False, rope_scaling=None,
                     max_context=n_prompt + 64, quantization={"bits": BITS, "group_size": GROUP, "scales_dtype": "float16"},
                     **(dict(num_experts=E, num_experts_per_tok=TOPK, moe_intermediate_size=I, moe_mode="mixtral", qk_norm=False) if MOE else {}))
    m.load_weights(weights(layers))
    prompt = (np.arange(n_prompt, dtype=np.uint32) * 7919 + 13) % V
    best = 1e30
    for _ in range(reps):
        m.reset()
        m.prefill(prompt)
        best = min(best, m.last_prefill_ms())
    m.close()
    return best


NL = 32 if MOE else 36


def main():
    layers = int(os.environ.get("OMX_F16_LAYERS", "2" if MOE else "6"))
    n_prompt = int(os.environ.get("OMX_F16_PROMPT", "2048"))
    rows = []
    for serial in (False, True):
        n = n_prompt if not serial else min(n_prompt, 256)
        a, b = run(layers, n, serial, 3 if not serial else 1), run(layers // 2, n, serial, 3 if not serial else 1)
        per_layer = (a - b) / (layers - layers // 2)
        rows.append((serial, n, a, per_layer))
        print(f"{'token-serial' if serial else 'batched f16 '} prompt {n:5d}: {a:9.2f} ms at {layers} layers, {per_layer:8.3f} ms / layer -> "
              f"{per_layer * NL * (n_prompt / n):9.1f} ms for {NL} layers x {n_prompt} tokens", flush=True)
    print(f"speed-up per layer at {n_prompt} tokens: {rows[1][3] * (n_prompt / rows[1][1]) / rows[0][3]:.1f}x")


if __name__ == "__main__":
    main()
