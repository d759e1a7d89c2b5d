"""GPU: the HIP path against the committed golden vectors (no oracle call in these tests -- the
expected values are data under tests/golden/)."""
import os

import numpy as np
import pytest

from test_gpu_primitives import assert_bf16_close

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_sdpa_golden(omx):
    T = omx.ops.Tensor
    d = np.load(os.path.join(G, "sdpa.npz"))
    got = omx.ops.scaled_dot_product_attention(T.from_numpy(d["q"]), T.from_numpy(d["k"]), T.from_numpy(d["v"]), float(d["scale"])).numpy()
    assert_bf16_close(got, d["out_decode"], 1, atol=2e-3 * np.abs(d["out_decode"]).max())
    got = omx.ops.scaled_dot_product_attention(T.from_numpy(d["qp"]), T.from_numpy(d["kp"]), T.from_numpy(d["vp"]), float(d["scale"]),
                                               T.from_numpy(d["mask"], "bool")).numpy()
    assert_bf16_close(got, d["out_prefill"], 2, atol=4e-3 * np.abs(d["out_prefill"]).max())


def test_qwen3_tiny_golden_tokens_and_logits(omx):
    from ominix_mlx_amd import engine
    d = np.load(os.path.join(G, "qwen3_tiny.npz"))
    h, L, I, H, Hkv, D, V = [int(v) for v in d["config"]]
    m = engine.Model(hidden_size=h, num_hidden_layers=L, intermediate_size=I, num_attention_heads=H, num_key_value_heads=Hkv,
                     head_dim=D, vocab_size=V, max_context=512)
    m.synth_weights()
    first = m.prefill(d["prompt"])
    l0 = m.last_logits()
    rest = m.decode(11)
    got = np.concatenate([[first], rest])
    bound = 2.0 ** -7 * max(np.abs(d["logits_first"]).max(), np.abs(d["logits_last"]).max()) * np.sqrt(L)
    assert np.abs(l0 - d["logits_first"]).max() <= bound
    for i in range(12):
        if got[i] != d["tokens"][i]:
            assert d["margins"][i] <= 2 * bound, f"token {i} differs outside the near-tie guard"
            break
    else:
        assert np.abs(m.last_logits() - d["logits_last"]).max() <= bound


def test_mel_and_klein_golden(omx):
    from ominix_mlx_amd import audio, klein
    T = omx.ops.Tensor
    d = np.load(os.path.join(G, "mel.npz"))
    feats = audio.MelFrontend().forward(d["audio"]).numpy()[0]
    strong = d["feats"] >= d["feats"].max() - np.log(1e5)
    assert np.abs(feats - d["feats"])[strong].max() < 1e-3
    d = np.load(os.path.join(G, "klein_tiny.npz"))
    m = klein.FluxKlein(128, 256, 512, 2, 2, 2, 128, 768)
    m.synth_weights()
    rcos, rsin = klein.compute_rope(klein.create_txt_ids(16), klein.create_img_ids(4, 6))
    out = m.forward_with_rope(T.from_numpy(d["latent"]), T.from_numpy(d["txt"]), float(d["timestep"]), rcos, rsin).numpy()
    assert np.abs(out - d["out"]).max() <= 2.0 ** -6 * np.abs(d["out"]).max() * 2


def test_moe_golden(omx):
    from ominix_mlx_amd import moe
    T = omx.ops.Tensor
    d = np.load(os.path.join(G, "moe.npz"))
    rand = lambda shape, seed, scale: (np.random.default_rng(seed).uniform(-1, 1, size=shape) * scale).astype(np.float32)
    E, h, I = 8, 512, 1024
    # same seeded generator as make_golden.py (seeds 60..63); Tensor.from_numpy rounds to bf16 (RNE) on upload
    gw, wg, wu, wd = rand((E, h), 60, 0.5), rand((E, I, h), 61, 0.05), rand((E, I, h), 62, 0.05), rand((E, h, I), 63, 0.05)
    blk = moe.SparseMoeBlock(T.from_numpy(gw), T.from_numpy(wg), T.from_numpy(wu), T.from_numpy(wd), 2, "mixtral")
    out, inds, scores = blk.forward(T.from_numpy(d["x"]), return_routing=True)
    np.testing.assert_array_equal(inds.numpy(), d["inds"])
    assert_bf16_close(out.numpy(), d["out"], 2, atol=2.0 ** -7 * np.abs(d["out"]).max())
