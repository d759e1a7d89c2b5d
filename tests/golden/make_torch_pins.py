"""Pins the remaining oracle-relative MODELS on an independent implementation (VERDICT r2 "Next" #6): torch (float64, CPU) restatements,
built from torch's own primitives -- F.layer_norm / F.rms_norm / F.group_norm / F.scaled_dot_product_attention / F.conv1d / F.conv2d /
F.interpolate / complex-number RoPE -- of
    * one Klein double block + one single block   (flux-klein-mlx/src/klein_model.rs:399-522, 603-674, rope :53-162)
    * one SAN-M encoder layer + one Paraformer decoder layer (funasr-mlx/src/paraformer.rs:496-634, 981-1053)
    * the VAE ResnetBlock (with and without conv_shortcut), AttnBlock and the decoder's final GroupNorm
      (flux-klein-mlx/src/autoencoder.rs:86-237)
and checks HERE that oracle/ref_klein.py, ref_paraformer.py, ref_vae.py agree with them to <= 1e-9 of the largest output (the
script fails otherwise).  Inputs, weights (float32, bf16-exact where the GPU tests load them as bf16) and the torch outputs are written to
tests/golden/torch_{klein,paraformer,vae}.npz; tests/test_oracle_pins.py re-checks the oracle from the fixtures on any machine and
the GPU tests compare the device kernels with the torch outputs.

    python tests/golden/make_torch_pins.py
"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import ref_core as rc, ref_klein as rk, ref_paraformer as rp, ref_vae as rv  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
T64 = lambda a: torch.from_numpy(np.asarray(a, np.float64))


def check(name, got, ref, tol=1e-9):
    err = float(np.abs(np.asarray(got, np.float64) - np.asarray(ref, np.float64)).max() / max(np.abs(ref).max(), 1e-30))
    print(f"  {name:34s} oracle vs torch: {err:.2e}")
    assert err <= tol, f"{name}: the oracle disagrees with torch ({err:.3e})"


# ------------------------------------------------------------------ Klein ----
def rope_complex(x, cos, sin):
    """x [S, H, D] with interleaved pairs: (x0 + i x1) * (cos + i sin)."""
    xc = torch.view_as_complex(x.reshape(*x.shape[:-1], -1, 2).contiguous())
    rot = torch.complex(cos[:, None, 0::2], sin[:, None, 0::2])
    return torch.view_as_real(xc * rot).reshape(x.shape)


def t_modulate(x, shift, scale, h):
    return (1.0 + scale) * F.layer_norm(x, (h,), eps=1e-6) + shift


def t_attn(q, k, v):
    """q [Sq, H, D], k / v [Sk, H, D] -> [Sq, H * D] through torch's SDPA (default scale 1 / sqrt(D))."""
    o = F.scaled_dot_product_attention(q.transpose(0, 1)[None], k.transpose(0, 1)[None], v.transpose(0, 1)[None])[0]
    return o.transpose(0, 1).reshape(q.shape[0], -1)


def klein_double(w, b, img, txt, img_mod, txt_mod, cos, sin, H, D):
    St, h = txt.shape[0], img.shape[1]
    xs, mods = {"img": img, "txt": txt}, {"img": img_mod, "txt": txt_mod}
    q, k, v = {}, {}, {}
    for st in ("txt", "img"):
        xm = t_modulate(xs[st], mods[st][0], mods[st][1], h)
        c, s = (cos[:St], sin[:St]) if st == "txt" else (cos[St:], sin[St:])
        qq = F.rms_norm(F.linear(xm, w[b + f"{st}_to_q.weight"]).reshape(-1, H, D), (D,), w[b + f"{st}_norm_q.weight"], 1e-5)
        kk = F.rms_norm(F.linear(xm, w[b + f"{st}_to_k.weight"]).reshape(-1, H, D), (D,), w[b + f"{st}_norm_k.weight"], 1e-5)
        q[st], k[st] = rope_complex(qq, c, s), rope_complex(kk, c, s)
        v[st] = F.linear(xm, w[b + f"{st}_to_v.weight"]).reshape(-1, H, D)
    kk, vv = torch.cat([k["txt"], k["img"]]), torch.cat([v["txt"], v["img"]])       # joint attention: keys [txt, img]
    out = {}
    for st in ("img", "txt"):
        g1, sh2, sc2, g2 = mods[st][2:6]
        x = xs[st] + F.linear(t_attn(q[st], kk, vv), w[b + f"{st}_to_out.weight"]) * g1
        gate, up = F.linear(t_modulate(x, sh2, sc2, h), w[b + f"{st}_mlp_in.weight"]).chunk(2, dim=-1)
        out[st] = x + F.linear(F.silu(gate) * up, w[b + f"{st}_mlp_out.weight"]) * g2
    return out["img"], out["txt"]


def klein_single(w, b, x, mod, cos, sin, H, D, mh):
    h = x.shape[1]
    shift, scale, g = mod
    q, k, v, mg, mu = F.linear(t_modulate(x, shift, scale, h), w[b + "to_qkv_mlp.weight"]).split([H * D, H * D, H * D, mh, mh], dim=-1)
    q = rope_complex(F.rms_norm(q.reshape(-1, H, D), (D,), w[b + "norm_q.weight"], 1e-5), cos, sin)
    k = rope_complex(F.rms_norm(k.reshape(-1, H, D), (D,), w[b + "norm_k.weight"], 1e-5), cos, sin)
    a = t_attn(q, k, v.reshape(-1, H, D))
    return x + F.linear(torch.cat([a, F.silu(mg) * mu], -1), w[b + "to_out.weight"]) * g


def pin_klein():
    print("Klein blocks")
    p = rk.KleinParams.tiny()
    wnp = {k: np.asarray(v, np.float32) for k, v in rk.synth_weights(p).items()}
    w = {k: T64(v) for k, v in wnp.items()}
    g = np.random.default_rng(11)
    St, ph, pw = 16, 4, 6
    Si, h = ph * pw, p.hidden_size
    img = rc.bf16_round(g.standard_normal((Si, h)).astype(np.float32)); txt = rc.bf16_round(g.standard_normal((St, h)).astype(np.float32))
    mods = rc.bf16_round((0.3 * g.standard_normal((15, h))).astype(np.float32))          # 6 img + 6 txt + 3 single
    cos, sin = rk.compute_rope(np.concatenate([rk.create_txt_ids(St), rk.create_img_ids(ph, pw)], 0))
    oracle = rk.KleinOracle(p, wnp)
    f64 = lambda a: np.asarray(a, np.float64)
    o_img, o_txt = oracle.double_block(0, f64(img), f64(txt), [f64(m)[None] for m in mods[:6]], [f64(m)[None] for m in mods[6:12]], cos, sin)
    t_img, t_txt = klein_double(w, "double_blocks.0.", T64(img), T64(txt), [T64(m)[None] for m in mods[:6]], [T64(m)[None] for m in mods[6:12]],
                                T64(cos), T64(sin), p.num_heads, p.head_dim)
    check("double block img", o_img, t_img.numpy()); check("double block txt", o_txt, t_txt.numpy())
    x = np.concatenate([txt, img], 0)
    o_s = oracle.single_block(0, f64(x), [f64(m)[None] for m in mods[12:15]], cos, sin)
    t_s = klein_single(w, "single_blocks.0.", T64(x), [T64(m)[None] for m in mods[12:15]], T64(cos), T64(sin), p.num_heads, p.head_dim, p.mlp_hidden)
    check("single block", o_s, t_s.numpy())
    np.savez_compressed(os.path.join(OUT, "torch_klein.npz"), img=img, txt=txt, mods=mods, St=St, ph=ph, pw=pw,
                        double_img=t_img.numpy().astype(np.float32), double_txt=t_txt.numpy().astype(np.float32),
                        single=t_s.numpy().astype(np.float32))


# -------------------------------------------------------------- Paraformer ----
def t_fsmn(v, w):
    """depthwise Conv1d over time (groups = channels, zero padding k // 2, no bias), NLC in and out."""
    k = w.shape[1]
    return F.conv1d(v.t()[None], w[:, None, :], padding=k // 2, groups=v.shape[1])[0].t()


def t_mha(q, k, v, heads):
    T, dim = q.shape
    D = dim // heads
    sp = lambda t: t.reshape(t.shape[0], heads, D).transpose(0, 1)[None]
    scale = float(np.float32(D) ** np.float32(-0.5))          # the reference's `scale` is an f32 value (paraformer.rs:466)
    return F.scaled_dot_product_attention(sp(q), sp(k), sp(v), scale=scale)[0].transpose(0, 1).reshape(T, dim)


def sanm_encoder_layer(x, p, heads):
    dim = p["out_w"].shape[0]
    h = F.layer_norm(x, (x.shape[1],), p["norm1_w"], p["norm1_b"], 1e-5)
    q, k, v = F.linear(h, p["qkv_w"], p["qkv_b"]).chunk(3, dim=-1)
    a = F.linear(t_mha(q, k, v, heads), p["out_w"], p["out_b"]) + (t_fsmn(v, p["fsmn_w"]) + v)
    x = x + a if x.shape[1] == dim else a
    h = F.layer_norm(x, (dim,), p["norm2_w"], p["norm2_b"], 1e-5)
    return x + F.linear(F.relu(F.linear(h, p["ffn_up_w"], p["ffn_up_b"])), p["ffn_down_w"], p["ffn_down_b"])


def para_decoder_layer(x, enc, p, heads):
    dim = x.shape[1]
    h = F.relu(F.linear(F.layer_norm(x, (dim,), p["norm1_w"], p["norm1_b"], 1e-5), p["ffn_up_w"], p["ffn_up_b"]))
    tgt = F.linear(F.layer_norm(h, (h.shape[1],), p["ffn_norm_w"], p["ffn_norm_b"], 1e-5), p["ffn_down_w"])
    h = F.layer_norm(tgt, (dim,), p["norm2_w"], p["norm2_b"], 1e-5)
    x1 = x + (t_fsmn(h, p["fsmn_w"]) + h)
    h = F.layer_norm(x1, (dim,), p["norm3_w"], p["norm3_b"], 1e-5)
    k, v = F.linear(enc, p["kv_w"], p["kv_b"]).chunk(2, dim=-1)
    return x1 + F.linear(t_mha(F.linear(h, p["q_w"], p["q_b"]), k, v, heads), p["out_w"], p["out_b"])


PARA_TINY = dict(n_mels=80, lfr_m=7, encoder_dim=512, encoder_layers=3, encoder_heads=4, encoder_ffn_dim=1024, decoder_dim=512,
                 decoder_layers=2, decoder_heads=4, decoder_ffn_dim=1024, vocab_size=640, sanm_kernel_size=11, cif_l_order=1,
                 cif_r_order=1, cif_threshold=1.0, cif_tail_threshold=0.45)


def pin_paraformer():
    """Layers at the model's real widths (512, 4 heads of 128: what the device kernels are built for); the weights are NOT stored --
    oracle/ref_paraformer.py synth_checkpoint(PARA_TINY, 7) regenerates them (numpy PCG64, bf16-exact) -- only inputs and torch outputs."""
    print("Paraformer layers")
    g = np.random.default_rng(12)
    w = rp.synth_checkpoint(PARA_TINY, 7)
    heads = 4
    enc_p, dec_p = rp._enc_params(w, "encoder.layers.0"), rp._dec_params(w, "decoder.layers.1")
    T, Tq, dim = 45, 11, 512
    x, xd = g.standard_normal((T, dim)).astype(np.float32), g.standard_normal((Tq, dim)).astype(np.float32)
    t_enc = sanm_encoder_layer(T64(x), {k: T64(v) for k, v in enc_p.items()}, heads).numpy()
    check("SAN-M encoder layer", rp.sanm_encoder_layer(x, enc_p, heads), t_enc)
    enc0_p = rp._enc_params(w, "encoder.encoders0.0")                  # first layer: in_dim 560 != 512, no attention residual
    x0 = g.standard_normal((T, PARA_TINY["n_mels"] * PARA_TINY["lfr_m"])).astype(np.float32)
    t_enc0 = sanm_encoder_layer(T64(x0), {k: T64(v) for k, v in enc0_p.items()}, heads).numpy()
    check("SAN-M first layer (560 -> 512)", rp.sanm_encoder_layer(x0, enc0_p, heads), t_enc0)
    t_dec = para_decoder_layer(T64(xd), T64(t_enc), {k: T64(v) for k, v in dec_p.items()}, heads).numpy()
    check("decoder layer", rp.decoder_layer(xd, t_enc, dec_p, heads), t_dec)
    np.savez_compressed(os.path.join(OUT, "torch_paraformer.npz"), x=x, x0=x0, xd=xd, heads=heads, seed=7,
                        enc_out=t_enc.astype(np.float32), enc0_out=t_enc0.astype(np.float32), dec_out=t_dec.astype(np.float32))


# --------------------------------------------------------------------- VAE ----
def nchw(x):   # [H, W, C] -> [1, C, H, W]
    return x.permute(2, 0, 1)[None]


def nhwc(x):
    return x[0].permute(1, 2, 0)


def t_conv(x, w, b, pad):      # weights in the reference's layout [out, kH, kW, in]
    return F.conv2d(x, w.permute(0, 3, 1, 2), b, padding=pad)


def vae_resnet(x, w, p):
    h = t_conv(F.silu(F.group_norm(x, 32, w[p + "norm1.weight"], w[p + "norm1.bias"], 1e-5)), w[p + "conv1.weight"], w[p + "conv1.bias"], 1)
    h = t_conv(F.silu(F.group_norm(h, 32, w[p + "norm2.weight"], w[p + "norm2.bias"], 1e-5)), w[p + "conv2.weight"], w[p + "conv2.bias"], 1)
    sc = t_conv(x, w[p + "conv_shortcut.weight"], w[p + "conv_shortcut.bias"], 0) if p + "conv_shortcut.weight" in w else x
    return h + sc


def vae_attn(x, w, p):
    B, C, H, W = x.shape
    h = F.group_norm(x, 32, w[p + "group_norm.weight"], w[p + "group_norm.bias"], 1e-5).reshape(C, H * W).t()
    q, k, v = (F.linear(h, w[p + n + ".weight"], w[p + n + ".bias"]) for n in ("to_q", "to_k", "to_v"))
    o = F.scaled_dot_product_attention(q[None, None], k[None, None], v[None, None])[0, 0]          # one head of width C
    return x + F.linear(o, w[p + "to_out.weight"], w[p + "to_out.bias"]).t().reshape(1, C, H, W)


def pin_vae():
    print("VAE blocks")
    cfg = dict(ch=32, ch_mult=(1, 2), num_res_blocks=1, z_channels=8)
    wnp = rv.synth_decoder_weights(3, **cfg)
    w = {k: T64(v) for k, v in wnp.items()}
    oracle = rv.VaeDecoderOracle(wnp, **cfg)
    g = np.random.default_rng(13)
    Hh, Ww = 6, 5
    mid_c = wnp["mid_block_resnets_0.conv1.weight"].shape[0]
    x = rc.bf16_round(g.standard_normal((Hh, Ww, mid_c)).astype(np.float32))
    out = {"x_mid": x}
    t = nhwc(vae_resnet(nchw(T64(x)), w, "mid_block_resnets_0.")).numpy()
    check("ResnetBlock (identity shortcut)", oracle.resnet(np.asarray(x, np.float64), "mid_block_resnets_0."), t); out["resnet_mid"] = t
    t = nhwc(vae_attn(nchw(T64(x)), w, "mid_block_attentions_0.")).numpy()
    check("AttnBlock", oracle.attn(np.asarray(x, np.float64), "mid_block_attentions_0."), t); out["attn_mid"] = t
    # a block whose channel count changes: conv_shortcut (autoencoder.rs:113-125)
    name = next(k[:-len("conv_shortcut.weight")] for k in wnp if k.endswith("conv_shortcut.weight"))
    cin = wnp[name + "conv1.weight"].shape[3]
    xs = rc.bf16_round(g.standard_normal((Hh, Ww, cin)).astype(np.float32))
    t = nhwc(vae_resnet(nchw(T64(xs)), w, name)).numpy()
    check("ResnetBlock (conv_shortcut)", oracle.resnet(np.asarray(xs, np.float64), name), t); out["x_sc"] = xs; out["resnet_sc"] = t
    cout = wnp["conv_norm_out.weight"].shape[0]
    xo = rc.bf16_round(g.standard_normal((Hh, Ww, cout)).astype(np.float32))
    t = nhwc(F.silu(F.group_norm(nchw(T64(xo)), 32, w["conv_norm_out.weight"], w["conv_norm_out.bias"], 1e-5))).numpy()
    check("GroupNorm + silu", rv.silu(oracle.gn(np.asarray(xo, np.float64), "conv_norm_out")), t); out["x_out"] = xo; out["gn_silu"] = t
    # whole decoder through torch as well (upsample = F.interpolate nearest)
    z = rc.bf16_round(g.standard_normal((3, 4, cfg["z_channels"])).astype(np.float32))
    hh = nchw(T64(z)) / np.float32(oracle.scale) + np.float32(oracle.shift)
    hh = t_conv(hh, w["post_quant_conv.weight"], w["post_quant_conv.bias"], 0)
    hh = t_conv(hh, w["conv_in.weight"], w["conv_in.bias"], 1)
    hh = vae_resnet(hh, w, "mid_block_resnets_0."); hh = vae_attn(hh, w, "mid_block_attentions_0."); hh = vae_resnet(hh, w, "mid_block_resnets_1.")
    for b, i in enumerate(reversed(range(len(cfg["ch_mult"])))):
        for j in range(cfg["num_res_blocks"] + 1):
            hh = vae_resnet(hh, w, f"up_blocks.{b}.resnets.{j}.")
        if i > 0:
            hh = t_conv(F.interpolate(hh, scale_factor=2, mode="nearest"), w[f"up_blocks.{b}.upsamplers_0_conv.weight"], w[f"up_blocks.{b}.upsamplers_0_conv.bias"], 1)
    hh = t_conv(F.silu(F.group_norm(hh, 32, w["conv_norm_out.weight"], w["conv_norm_out.bias"], 1e-5)), w["conv_out.weight"], w["conv_out.bias"], 1)
    t = nhwc(hh).numpy()
    check("whole decoder", oracle.forward(z), t); out["z"] = z; out["decoded"] = t
    np.savez_compressed(os.path.join(OUT, "torch_vae.npz"), seed=3, **{k: np.asarray(v, np.float32) for k, v in out.items()})


if __name__ == "__main__":
    torch.manual_seed(0)
    pin_klein()
    pin_paraformer()
    pin_vae()
    print("fixtures written to", OUT)
