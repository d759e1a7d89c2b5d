"""GPU: the four-wave 256 x 256 / 128 x 256 GEMM tile (csrc/gemm.hip gemm_nt_w4_kernel, round 5) forced ON against forced OFF on every
epilogue form it serves by default -- ADVICE r5: the bit-for-bit A/B of tests/test_gpu_mfma.py only drove ops.linear with an optional bias.
Every output element is the same chain of 16x16x32 MFMAs over k in both kernels, so whole passes must agree bit for bit:

  * segmented launches: q | k | v with per-segment bias (Qwen2), gate | up with the SwiGLU epilogue in both activation roundings, ragged halves;
  * the grouped (MoE) form: row gather, uneven expert row counts, tiles that belong to no expert;
  * residual / gated-residual parked epilogues and per-segment bias of the DiT blocks;
  * widths with N % 8 != 0 and interior tiles next to ragged ones.
"""
import numpy as np
import pytest

from oracle import ref_core as rc, ref_klein as rk, ref_qwen3 as rq, synth
from test_gpu_primitives import rand

pytestmark = pytest.mark.gpu


def _ab(monkeypatch, fn):
    outs = []
    for v in ("1", "0"):
        monkeypatch.setenv("OMX_GEMM_W4", v)
        outs.append(fn())
    return outs


@pytest.mark.parametrize("M,n_plain,half,K", [(300, 0, 520, 256), (1000, 64, 1100, 384), (257, 128, 68, 128), (2048, 0, 1536, 1024)])
def test_segmented_swiglu_launch(omx, monkeypatch, M, n_plain, half, K):
    T = omx.ops.Tensor
    x = T.from_numpy(rc.bf16_round(rand((M, K), 5)))
    w = T.from_numpy(rc.bf16_round(rand((n_plain + 2 * half, K), 6) * 0.05))

    def run():
        plain, act = omx.ops.linear_swiglu(x, w, n_plain)
        return (plain.numpy() if plain is not None else np.zeros(0)), act.numpy()
    a, b = _ab(monkeypatch, run)
    np.testing.assert_array_equal(a[0], b[0])
    np.testing.assert_array_equal(a[1], b[1])


@pytest.mark.parametrize("M,N,K,bias", [(513, 1236, 256, True), (700, 204, 640, False), (1025, 2052, 128, True)])
def test_widths_that_are_not_multiples_of_eight(omx, monkeypatch, M, N, K, bias):
    T = omx.ops.Tensor
    x = T.from_numpy(rc.bf16_round(rand((M, K), 7)))
    w = T.from_numpy(rc.bf16_round(rand((N, K), 8) * 0.05))
    b = T.from_numpy(rc.bf16_round(rand((N,), 9))) if bias else None
    for tile in ("256", None):
        if tile:
            monkeypatch.setenv("OMX_GEMM_TILE", tile)
        else:
            monkeypatch.delenv("OMX_GEMM_TILE", raising=False)
            monkeypatch.setenv("OMX_GEMM_ROWS128", "1")
        a, c = _ab(monkeypatch, lambda: omx.ops.linear(x, w, b).numpy())
        np.testing.assert_array_equal(a, c)


@pytest.mark.parametrize("name", ["qwen3", "qwen2_bias"])
def test_dense_prompt_pass(omx, monkeypatch, name):
    """q | k | v segmented (Qwen2: with its per-segment biases), gate | up + SwiGLU, o / down with the residual epilogue, 300 ragged rows."""
    from ominix_mlx_amd import engine
    if name == "qwen3":
        cfg = dict(hidden_size=512, num_hidden_layers=2, intermediate_size=1536, num_attention_heads=8, num_key_value_heads=4, head_dim=64,
                   vocab_size=2048, rms_norm_eps=1e-6, rope_theta=1e6, tie_word_embeddings=False)
    else:
        cfg = dict(hidden_size=896, num_hidden_layers=2, intermediate_size=4864, num_attention_heads=14, num_key_value_heads=2, head_dim=64,
                   vocab_size=2048, rms_norm_eps=1e-6, rope_theta=1e6, tie_word_embeddings=True, qk_norm=False, attention_bias=True)
    prompt = synth.prompt_ids(300, cfg["vocab_size"])

    def run():
        m = engine.Model(max_context=512, **cfg)
        m.synth_weights()
        tok = int(m.prefill(prompt))
        lg = m.last_logits()
        m.close()
        return tok, lg
    (t1, l1), (t0, l0) = _ab(monkeypatch, run)
    assert t1 == t0
    np.testing.assert_array_equal(l1, l0)


@pytest.mark.parametrize("moe_mode,E,k", [("mixtral", 4, 2), ("qwen3_moe", 8, 2)])
def test_grouped_expert_prompt_pass(omx, monkeypatch, moe_mode, E, k):
    """The grouped form: rows gathered per expert (uneven counts: 193 tokens x top-k over E experts), tiles past an expert's rows return early."""
    from ominix_mlx_amd import engine
    cfg = dict(hidden_size=1024, num_hidden_layers=2, intermediate_size=1024, num_attention_heads=8, num_key_value_heads=2, head_dim=128,
               vocab_size=2048, rms_norm_eps=1e-5, rope_theta=1e6, tie_word_embeddings=False, num_experts=E, num_experts_per_tok=k,
               moe_intermediate_size=512 if moe_mode == "qwen3_moe" else 1024, moe_mode=moe_mode, norm_topk_prob=moe_mode == "qwen3_moe",
               qk_norm=moe_mode == "qwen3_moe")
    prompt = synth.prompt_ids(193, cfg["vocab_size"])

    def run():
        m = engine.Model(max_context=256, **cfg)
        m.synth_weights()
        tok = int(m.prefill(prompt))
        lg = m.last_logits()
        m.close()
        return tok, lg
    (t1, l1), (t0, l0) = _ab(monkeypatch, run)
    assert t1 == t0
    np.testing.assert_array_equal(l1, l0)


def test_dit_blocks(omx, monkeypatch):
    """Double + single blocks of the DiT: fused qkv + mlp projections with per-segment bias, gated-residual and residual parked epilogues."""
    from ominix_mlx_amd import klein
    T = omx.ops.Tensor
    p = rk.KleinParams.tiny()
    g = np.random.default_rng(17)
    grid, s_txt = (18, 17), 70
    latent = T.from_numpy(rc.bf16_round(g.standard_normal((grid[0] * grid[1], p.in_channels)).astype(np.float32)))
    txt = T.from_numpy(rc.bf16_round(g.standard_normal((s_txt, p.txt_embed_dim)).astype(np.float32)))

    def run():
        m = klein.FluxKlein(p.in_channels, p.hidden_size, p.txt_embed_dim, p.num_heads, p.depth, p.depth_single, p.head_dim, p.mlp_hidden)
        m.synth_weights()
        rcos, rsin = klein.compute_rope(klein.create_txt_ids(s_txt), klein.create_img_ids(*grid))
        return m.forward_with_rope(latent, txt, 750.0, rcos, rsin).numpy()
    a, b = _ab(monkeypatch, run)
    np.testing.assert_array_equal(a, b)
