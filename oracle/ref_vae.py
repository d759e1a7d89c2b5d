"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the FLUX VAE decoder.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import anything under oracle/.

Follows flux-klein-mlx/src/autoencoder.rs: AutoEncoderConfig :22-81, ResnetBlock::forward :139-157,
AttnBlock::forward :195-232, Decoder::new :279-372 (block/channel bookkeeping), Decoder::forward :375-412;
GroupNorm = mlx-rs/src/nn/normalization.rs:363-392 (pytorch-compatible grouping, layer_norm over (H*W, C/G), eps 1e-5);
Conv2d NHWC with weight [out, kH, kW, in]; Upsample(2, nearest).  float64 throughout.  Parity unpinned: the reference
holds no value test for the VAE (only shape handling in its example), so GPU results are oracle-relative."""
from __future__ import annotations

from typing import Dict

import numpy as np


def conv2d(x, w, b, pad: int):
    """x [H, W, Cin], w [Cout, kH, kW, Cin], b [Cout]."""
    x = np.asarray(x, np.float64)
    w = np.asarray(w, np.float64)
    H, W, _ = x.shape
    kH, kW = w.shape[1], w.shape[2]
    xp = np.pad(x, ((pad, pad), (pad, pad), (0, 0)))
    out = np.zeros((H + 2 * pad - kH + 1, W + 2 * pad - kW + 1, w.shape[0]))
    for i in range(kH):
        for j in range(kW):
            out += xp[i:i + out.shape[0], j:j + out.shape[1], :] @ w[:, i, j, :].T
    return out + np.asarray(b, np.float64)


def group_norm(x, weight, bias, groups: int = 32, eps: float = 1e-5):
    H, W, C = x.shape
    g = x.reshape(H * W, groups, C // groups)
    mean = g.mean(axis=(0, 2), keepdims=True)
    var = g.var(axis=(0, 2), keepdims=True)
    y = ((g - mean) / np.sqrt(var + eps)).reshape(H, W, C)
    return y * np.asarray(weight, np.float64) + np.asarray(bias, np.float64)


def silu(x):
    return x / (1.0 + np.exp(-x))


def upsample_nearest2(x):
    return np.repeat(np.repeat(x, 2, axis=0), 2, axis=1)


def decoder_weight_shapes(ch=128, ch_mult=(1, 2, 4, 4), num_res_blocks=2, z_channels=32, out_ch=3) -> Dict[str, tuple]:
    n = len(ch_mult)
    block_in = ch * ch_mult[-1]
    s: Dict[str, tuple] = {"post_quant_conv.weight": (z_channels, 1, 1, z_channels), "post_quant_conv.bias": (z_channels,),
                           "conv_in.weight": (block_in, 3, 3, z_channels), "conv_in.bias": (block_in,)}

    def resnet(p, cin, cout):
        s[p + "norm1.weight"] = s[p + "norm1.bias"] = (cin,)
        s[p + "conv1.weight"], s[p + "conv1.bias"] = (cout, 3, 3, cin), (cout,)
        s[p + "norm2.weight"] = s[p + "norm2.bias"] = (cout,)
        s[p + "conv2.weight"], s[p + "conv2.bias"] = (cout, 3, 3, cout), (cout,)
        if cin != cout:
            s[p + "conv_shortcut.weight"], s[p + "conv_shortcut.bias"] = (cout, 1, 1, cin), (cout,)

    resnet("mid_block_resnets_0.", block_in, block_in)
    resnet("mid_block_resnets_1.", block_in, block_in)
    a = "mid_block_attentions_0."
    s[a + "group_norm.weight"] = s[a + "group_norm.bias"] = (block_in,)
    for nm in ("to_q", "to_k", "to_v", "to_out"):
        s[a + nm + ".weight"], s[a + nm + ".bias"] = (block_in, block_in), (block_in,)
    cur = block_in
    for b, i in enumerate(reversed(range(n))):
        cout = ch * ch_mult[i]
        for j in range(num_res_blocks + 1):
            resnet(f"up_blocks.{b}.resnets.{j}.", cur if j == 0 else cout, cout)
        if i > 0:
            s[f"up_blocks.{b}.upsamplers_0_conv.weight"], s[f"up_blocks.{b}.upsamplers_0_conv.bias"] = (cout, 3, 3, cout), (cout,)
        cur = cout
    s["conv_norm_out.weight"] = s["conv_norm_out.bias"] = (ch,)
    s["conv_out.weight"], s["conv_out.bias"] = (out_ch, 3, 3, ch), (out_ch,)
    return s


def synth_decoder_weights(seed: int = 0, **cfg) -> Dict[str, np.ndarray]:
    """Random weights with the magnitudes of a trained VAE (variance-preserving convs, norm scales near 1), bf16-exact."""
    from . import ref_core as rc
    g = np.random.default_rng(seed)
    out = {}
    for name, shape in decoder_weight_shapes(**cfg).items():
        if name.endswith("norm1.weight") or name.endswith("norm2.weight") or name.endswith("norm.weight") or name.endswith("norm_out.weight"):
            v = 1.0 + 0.1 * g.standard_normal(shape)
        elif name.endswith(".bias"):
            v = 0.05 * g.standard_normal(shape)
        else:
            fan_in = int(np.prod(shape[1:]))
            v = g.standard_normal(shape) / np.sqrt(fan_in)
        out[name] = rc.bf16_round(v.astype(np.float32))
    return out


class VaeDecoderOracle:
    def __init__(self, weights, ch=128, ch_mult=(1, 2, 4, 4), num_res_blocks=2, z_channels=32, out_ch=3, scale_factor=0.3611,
                 shift_factor=0.1159):
        self.w = {k: np.asarray(v, np.float64) for k, v in weights.items()}
        self.ch, self.ch_mult, self.nrb, self.z, self.out_ch = ch, tuple(ch_mult), num_res_blocks, z_channels, out_ch
        self.scale, self.shift = scale_factor, shift_factor

    def conv(self, x, name, pad):
        return conv2d(x, self.w[name + ".weight"], self.w[name + ".bias"], pad)

    def gn(self, x, name):
        return group_norm(x, self.w[name + ".weight"], self.w[name + ".bias"])

    def resnet(self, x, p):
        h = self.conv(silu(self.gn(x, p + "norm1")), p + "conv1", 1)
        h = self.conv(silu(self.gn(h, p + "norm2")), p + "conv2", 1)
        sc = self.conv(x, p + "conv_shortcut", 0) if (p + "conv_shortcut.weight") in self.w else x
        return h + sc

    def attn(self, x, p):
        H, W, C = x.shape
        h = self.gn(x, p + "group_norm").reshape(H * W, C)
        lin = lambda t, n: t @ self.w[p + n + ".weight"].T + self.w[p + n + ".bias"]
        q, k, v = lin(h, "to_q"), lin(h, "to_k"), lin(h, "to_v")
        s = (q @ k.T) / np.sqrt(np.float32(C))
        s = np.exp(s - s.max(axis=-1, keepdims=True))
        o = (s / s.sum(axis=-1, keepdims=True)) @ v
        return x + lin(o, "to_out").reshape(H, W, C)

    def forward(self, z):
        z = np.asarray(z, np.float64) / np.float32(self.scale) + np.float32(self.shift)
        z = self.conv(z, "post_quant_conv", 0)
        h = self.conv(z, "conv_in", 1)
        h = self.resnet(h, "mid_block_resnets_0.")
        h = self.attn(h, "mid_block_attentions_0.")
        h = self.resnet(h, "mid_block_resnets_1.")
        for b, i in enumerate(reversed(range(len(self.ch_mult)))):
            for j in range(self.nrb + 1):
                h = self.resnet(h, f"up_blocks.{b}.resnets.{j}.")
            if i > 0:
                h = self.conv(upsample_nearest2(h), f"up_blocks.{b}.upsamplers_0_conv", 1)
        return self.conv(silu(self.gn(h, "conv_norm_out")), "conv_out", 1)
