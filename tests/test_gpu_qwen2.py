"""Qwen2 wiring of the decode engine (qwen3-mlx/src/qwen2.rs:100-218: q/k/v Linear WITH bias, no q/k norm, same MLP / norms /
RoPE) at shapes that have no tuned GEMV width -- Qwen2-0.5B (hidden 896, 14 heads of 64, 7 query heads per KV head) and
Qwen2.5-7B proportions (hidden 3584, intermediate 18944) -- so the generic streaming GEMV and the G = 7 attention carry it.
Oracle and tolerances as tests/test_gpu_qwen3.py."""
import numpy as np
import pytest

from oracle import ref_core as rc
from oracle import ref_qwen3 as rq
from oracle import synth

pytestmark = pytest.mark.gpu

CONFIGS = {
    "qwen2_0p5b_proportions": rq.Qwen3Config(896, 2, 4864, 14, 2, 64, 2048, 1e-6, 1e6, True, None, 40960, 0, 0, 0, "qwen3_moe", False, False, True),
    "qwen2p5_7b_proportions_1_layer": rq.Qwen3Config(3584, 1, 18944, 28, 4, 128, 1024, 1e-6, 1e6, False, None, 40960, 0, 0, 0, "qwen3_moe", False, False, True),
}


def _engine(omx, cfg, weights=None):
    from ominix_mlx_amd import engine
    m = engine.Model(hidden_size=cfg.hidden_size, num_hidden_layers=cfg.num_hidden_layers, intermediate_size=cfg.intermediate_size,
                     num_attention_heads=cfg.num_attention_heads, num_key_value_heads=cfg.num_key_value_heads, head_dim=cfg.head_dim,
                     vocab_size=cfg.vocab_size, rms_norm_eps=cfg.rms_norm_eps, rope_theta=cfg.rope_theta,
                     tie_word_embeddings=cfg.tie_word_embeddings, max_context=256, qk_norm=False, attention_bias=True)
    m.synth_weights() if weights is None else m.load_weights(weights)
    return m


@pytest.mark.parametrize("name", list(CONFIGS))
def test_qwen2_decode_matches_oracle(omx, name):
    cfg = CONFIGS[name]
    weights = rq.synth_weights(cfg)
    assert any(k.endswith("q_proj.bias") for k in weights) and not any("q_norm" in k for k in weights)
    oracle = rq.Qwen3Oracle(cfg, weights)
    n_prompt, n_new = 24, 6
    prompt = synth.prompt_ids(n_prompt, cfg.vocab_size)
    ref_tokens, ref_logits = oracle.generate(prompt, n_new, return_logits=True)
    m = _engine(omx, cfg)
    first = m.prefill(prompt)
    logits0 = m.last_logits()
    got = np.concatenate([[first], m.decode(n_new - 1)]).astype(np.uint32)
    assert m.decode_path() == "graph"
    bound = 2.0 ** -7 * np.abs(ref_logits).max() * np.sqrt(cfg.num_hidden_layers)
    assert np.abs(logits0 - ref_logits[0]).max() <= bound
    margins = rc.argmax_margin(ref_logits)
    for i in range(n_new):
        if got[i] != ref_tokens[i]:
            assert margins[i] <= 2 * bound, f"token {i}: margin {margins[i]:.4f}"
            break
    m2 = _engine(omx, cfg, weights)
    got2 = np.concatenate([[m2.prefill(prompt)], m2.decode(n_new - 1)]).astype(np.uint32)
    np.testing.assert_array_equal(got2, got)


def test_qwen2_serial_and_batched_prefill_agree(omx, monkeypatch):
    cfg = CONFIGS["qwen2_0p5b_proportions"]
    prompt = synth.prompt_ids(50, cfg.vocab_size)
    outs = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("OMX_PREFILL_SERIAL", mode)
        m = _engine(omx, cfg)
        outs[mode] = (m.prefill(prompt), m.last_logits())
    bound = 2.0 ** -7 * np.abs(outs["1"][1]).max() * np.sqrt(cfg.num_hidden_layers)
    assert np.abs(outs["0"][1] - outs["1"][1]).max() <= bound
