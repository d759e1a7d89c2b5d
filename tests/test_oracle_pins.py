"""CPU: the model-level oracle pinned on an INDEPENDENT implementation (VERDICT r1 "Next" #4).

tests/golden/hf_*.npz were written by tests/golden/make_hf_pins.py in the build container: tiny random Qwen3 / Qwen2 / Mixtral /
Qwen3-MoE models of the `transformers` library run in fp32 on the CPU -- weights (bf16-representable), prompt, logits of every
prompt position and of four greedy steps.  Here (no transformers needed) the oracle, in f32, must reproduce them: until round 2
everything above the primitives (SDPA, KV cache, MoE routing, whole forwards) was "oracle-relative"; with this it is anchored to
code that shares nothing with this repository.  Bound: 1e-4 of the largest logit (measured: ~1e-6)."""
import glob
import json
import os

import numpy as np
import pytest

from oracle import ref_core as rc, ref_qwen3 as rq

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
FIXTURES = sorted(glob.glob(os.path.join(GOLDEN, "hf_*.npz")))


def load_pin(path):
    z = np.load(path)
    cfg = rq.Qwen3Config(**json.loads(str(z["cfg"])))
    weights = {k[2:]: (z[k].astype(np.uint32) << np.uint32(16)).view(np.float32) for k in z.files if k.startswith("w:")}
    return cfg, weights, z


def test_fixtures_present():
    assert len(FIXTURES) == 5, "tests/golden/hf_*.npz missing: run tests/golden/make_hf_pins.py in the build container"


@pytest.mark.parametrize("path", FIXTURES, ids=[os.path.basename(p)[3:-4] for p in FIXTURES])
def test_oracle_reproduces_transformers_logits(path):
    cfg, weights, z = load_pin(path)
    oracle = rq.Qwen3Oracle(cfg, weights, dt="f32")
    tokens, logits = oracle.generate(z["prompt"], len(z["hf_tokens"]), return_logits=True)
    np.testing.assert_array_equal(tokens, z["hf_tokens"])
    assert np.abs(logits - z["hf_logits"]).max() <= 1e-4 * np.abs(z["hf_logits"]).max()
    every = oracle.forward(z["prompt"][None, :].astype(np.int64), [])[0]
    assert np.abs(every - z["hf_prompt_logits"]).max() <= 1e-4 * np.abs(z["hf_prompt_logits"]).max()


@pytest.mark.parametrize("path", FIXTURES, ids=[os.path.basename(p)[3:-4] for p in FIXTURES])
def test_bf16_oracle_stays_within_the_bf16_bound_of_the_fp32_reference(path):
    """The oracle in the path's own dtype (bf16 op outputs, as MLX rounds them) against the unrounded fp32 transformers logits:
    within 2^-6 * max|logit| * sqrt(layers) -- the figure tests/test_gpu_hf_pins.py then holds the ENGINE to."""
    cfg, weights, z = load_pin(path)
    oracle = rq.Qwen3Oracle(cfg, weights, dt="bf16")
    _, logits = oracle.generate(z["prompt"], 1, return_logits=True)
    assert np.abs(logits[0] - z["hf_logits"][0]).max() <= 2.0 ** -6 * np.abs(z["hf_logits"]).max() * np.sqrt(cfg.num_hidden_layers)


def test_oracle_sdpa_equals_torch_sdpa():
    """rc.scaled_dot_product_attention (f32) vs torch.nn.functional.scaled_dot_product_attention on the CPU: no mask, causal
    (bottom-right aligned when Tq < Tk), boolean keep-mask, additive mask, GQA by head group."""
    import torch
    import torch.nn.functional as F
    g = np.random.default_rng(0)
    for (B, H, Hkv, Tq, Tk, D) in [(1, 8, 2, 1, 37, 64), (2, 4, 4, 9, 9, 32), (1, 6, 3, 5, 21, 128)]:
        q, k, v = (g.standard_normal(s).astype(np.float32) for s in ((B, H, Tq, D), (B, Hkv, Tk, D), (B, Hkv, Tk, D)))
        scale = D ** -0.5
        kt, vt = torch.tensor(k).repeat_interleave(H // Hkv, 1), torch.tensor(v).repeat_interleave(H // Hkv, 1)
        bool_mask = g.random((Tq, Tk)) > 0.3
        bool_mask[:, 0] = True
        add_mask = (g.standard_normal((Tq, Tk)) * 2).astype(np.float32)
        causal = np.tril(np.ones((Tq, Tk), bool), Tk - Tq)
        for om, tm in ((None, None), ("causal", torch.tensor(causal)), (bool_mask, torch.tensor(bool_mask)), (add_mask, torch.tensor(add_mask))):
            want = F.scaled_dot_product_attention(torch.tensor(q), kt, vt, attn_mask=tm, scale=scale).numpy()
            got = rc.scaled_dot_product_attention(q, k, v, scale, om, "f32")
            assert np.abs(got - want).max() <= 2e-6 * max(1.0, np.abs(want).max())


# ---- round 3: the remaining oracle-relative MODELS pinned on torch (tests/golden/make_torch_pins.py) ----

def _torch_pin(name):
    path = os.path.join(GOLDEN, f"torch_{name}.npz")
    assert os.path.exists(path), f"{path} missing: run tests/golden/make_torch_pins.py in the build container"
    return np.load(path)


def _close(got, ref, tol=1e-6):
    assert np.abs(np.asarray(got, np.float64) - ref).max() <= tol * np.abs(ref).max()


def test_klein_blocks_reproduce_torch():
    """oracle/ref_klein.py double and single block == the torch restatement (F.layer_norm / F.rms_norm / F.scaled_dot_product_attention /
    complex-number RoPE; klein_model.rs:399-522, 603-674) on the committed inputs; fixture outputs are float32."""
    from oracle import ref_klein as rk
    z = _torch_pin("klein")
    p = rk.KleinParams.tiny()
    oracle = rk.KleinOracle(p, rk.synth_weights(p))
    St, ph, pw = int(z["St"]), int(z["ph"]), int(z["pw"])
    cos, sin = rk.compute_rope(np.concatenate([rk.create_txt_ids(St), rk.create_img_ids(ph, pw)], 0))
    f64 = lambda a: np.asarray(a, np.float64)
    mods = [f64(m)[None] for m in z["mods"]]
    img, txt = oracle.double_block(0, f64(z["img"]), f64(z["txt"]), mods[:6], mods[6:12], cos, sin)
    _close(img, z["double_img"]); _close(txt, z["double_txt"])
    x = np.concatenate([f64(z["txt"]), f64(z["img"])], 0)
    _close(oracle.single_block(0, x, mods[12:15], cos, sin), z["single"])


def test_paraformer_layers_reproduce_torch():
    from oracle import ref_paraformer as rp
    from test_gpu_paraformer import TINY
    z = _torch_pin("paraformer")
    w = rp.synth_checkpoint(TINY, int(z["seed"]))
    heads = int(z["heads"])
    _close(rp.sanm_encoder_layer(z["x"], rp._enc_params(w, "encoder.layers.0"), heads), z["enc_out"])
    _close(rp.sanm_encoder_layer(z["x0"], rp._enc_params(w, "encoder.encoders0.0"), heads), z["enc0_out"])
    _close(rp.decoder_layer(z["xd"], z["enc_out"], rp._dec_params(w, "decoder.layers.1"), heads), z["dec_out"])


def test_vae_blocks_reproduce_torch():
    from oracle import ref_vae as rv
    z = _torch_pin("vae")
    cfg = dict(ch=32, ch_mult=(1, 2), num_res_blocks=1, z_channels=8)
    weights = rv.synth_decoder_weights(int(z["seed"]), **cfg)
    oracle = rv.VaeDecoderOracle(weights, **cfg)
    f64 = lambda a: np.asarray(a, np.float64)
    _close(oracle.resnet(f64(z["x_mid"]), "mid_block_resnets_0."), z["resnet_mid"])
    _close(oracle.attn(f64(z["x_mid"]), "mid_block_attentions_0."), z["attn_mid"])
    name = next(k[:-len("conv_shortcut.weight")] for k in weights if k.endswith("conv_shortcut.weight"))
    _close(oracle.resnet(f64(z["x_sc"]), name), z["resnet_sc"])
    _close(rv.silu(oracle.gn(f64(z["x_out"]), "conv_norm_out")), z["gn_silu"])
    _close(oracle.forward(z["z"]), z["decoded"])
