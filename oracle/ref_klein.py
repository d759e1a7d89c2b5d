"""TEST INFRASTRUCTURE ONLY -- CPU (numpy) restatement of the FLUX.2-klein DiT forward
(BASELINE config 5).  Never imported by the product path.

Follows flux-klein-mlx/src/klein_model.rs:
    compute_rope_freqs / apply_rope        :53-162   (4 axes x 32 dims, theta 2000, interleaved pairs)
    SharedModulation::forward              :248-254  (Linear(silu(vec)), split)
    KleinDoubleBlock::forward              :399-522
    KleinSingleBlock::forward              :603-674
    FluxKlein::forward_with_rope           :799-854  (final layer chunk order [scale, shift], final_norm = RmsNorm)
    modulate / gate                        :909-925
  flux-klein-mlx/src/layers.rs:256-283 timestep_embedding ([cos | sin], freq_i = exp(-ln(1e4) i/half))
  flux-klein-mlx/examples/generate_klein.rs:519-556 create_img_ids / create_txt_ids, :441-443 Euler step

dtype: the reference example runs this model in float32 (generate_klein.rs:391-392), and with half-precision
weights MLX's promotion rules would still make the activations float32 from the first `modulate` on (its
`array!(1.0f32)` / `array!(scale)` operands are f32, klein_model.rs:916, 474-476).  So the restatement is
evaluated in float64 on the given (bf16-rounded) inputs and weights and is the f32-path value; the MI355X
build computes in bf16 with fp32 accumulation and is compared with a bf16-level tolerance.
PARITY UNPINNED: the reference's klein tests check shapes and the sampler only;
the Euler step KAT (sampler.rs:390-407) is reproduced in tests/test_klein_oracle.py.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict

import numpy as np

from . import synth


@dataclass
class KleinParams:
    """FluxKleinParams (klein_model.rs:166-196)."""
    in_channels: int = 128
    hidden_size: int = 3072
    txt_embed_dim: int = 7680
    num_heads: int = 24
    depth: int = 5
    depth_single: int = 20
    head_dim: int = 128
    mlp_hidden: int = 9216

    @staticmethod
    def tiny():
        return KleinParams(128, 256, 512, 2, 2, 2, 128, 768)


AXES_DIM = (32, 32, 32, 32)
THETA = 2000.0
RMS_EPS = 1e-5        # RmsNorm::DEFAULT_EPS (mlx-rs/src/nn/normalization.rs)
LN_EPS = 1e-6


def weight_shapes(p: KleinParams) -> Dict[str, tuple]:
    h, m, D = p.hidden_size, p.mlp_hidden, p.head_dim
    s = {"x_embedder.weight": (h, p.in_channels), "context_embedder.weight": (h, p.txt_embed_dim),
         "time_embed_1.weight": (h, 256), "time_embed_2.weight": (h, h),
         "double_mod_img.linear.weight": (6 * h, h), "double_mod_txt.linear.weight": (6 * h, h),
         "single_mod.linear.weight": (3 * h, h), "norm_out.weight": (2 * h, h), "proj_out.weight": (p.in_channels, h)}
    for i in range(p.depth):
        b = f"double_blocks.{i}."
        for st in ("img", "txt"):
            for n in ("to_q", "to_k", "to_v", "to_out"):
                s[b + f"{st}_{n}.weight"] = (h, h)
            s[b + f"{st}_norm_q.weight"] = (D,)
            s[b + f"{st}_norm_k.weight"] = (D,)
            s[b + f"{st}_mlp_in.weight"] = (2 * m, h)
            s[b + f"{st}_mlp_out.weight"] = (h, m)
    for i in range(p.depth_single):
        b = f"single_blocks.{i}."
        s[b + "to_qkv_mlp.weight"] = (3 * h + 2 * m, h)
        s[b + "to_out.weight"] = (h, h + m)
        s[b + "norm_q.weight"] = (D,)
        s[b + "norm_k.weight"] = (D,)
    return s


def synth_weights(p: KleinParams, dt: str = "bf16") -> Dict[str, np.ndarray]:
    out = {}
    for name, shape in weight_shapes(p).items():
        norm = name.endswith(("norm_q.weight", "norm_k.weight"))
        out[name] = synth.tensor("klein." + name, shape, 0.01 if norm else 0.02, 1.0 if norm else 0.0, dt)
    return out


def create_txt_ids(seq_len: int) -> np.ndarray:
    ids = np.zeros((seq_len, 4), np.float32)
    ids[:, 3] = np.arange(seq_len)
    return ids


def create_img_ids(h: int, w: int) -> np.ndarray:
    ids = np.zeros((h * w, 4), np.float32)
    ids[:, 1] = np.repeat(np.arange(h), w)
    ids[:, 2] = np.tile(np.arange(w), h)
    return ids


def compute_rope(ids: np.ndarray):
    """ids [S, 4] (txt rows first, then img) -> cos, sin [S, 128], each frequency duplicated [c0,c0,c1,c1,..]."""
    cs, sn = [], []
    for axis, dim in enumerate(AXES_DIM):
        half = dim // 2
        inv = np.array([1.0 / np.float32(THETA) ** (np.float32(2.0 * i) / np.float32(dim)) for i in range(half)], np.float32)
        ang = (ids[:, axis:axis + 1].astype(np.float32) * inv[None, :]).astype(np.float32)      # f32 multiply, as MLX
        cs.append(np.repeat(np.cos(ang.astype(np.float64)), 2, axis=1))
        sn.append(np.repeat(np.sin(ang.astype(np.float64)), 2, axis=1))
    return np.concatenate(cs, 1).astype(np.float32), np.concatenate(sn, 1).astype(np.float32)


def apply_rope(x, cos, sin):
    """x [S, heads, D]; interleaved pairs (2i, 2i+1)."""
    x0, x1 = x[..., 0::2], x[..., 1::2]
    c, s = cos[:, None, 0::2], sin[:, None, 0::2]
    out = np.empty_like(x)
    out[..., 0::2] = x0 * c - x1 * s
    out[..., 1::2] = x1 * c + x0 * s
    return out


def timestep_embedding(t: float, dim: int = 256, max_period: float = 10000.0) -> np.ndarray:
    half = dim // 2
    freqs = np.exp(-np.log(np.float32(max_period)) * np.arange(half, dtype=np.float32) / np.float32(half)).astype(np.float32)
    args = (np.float32(t) * freqs).astype(np.float64)
    return np.concatenate([np.cos(args), np.sin(args)])[None, :]


def _silu(x):
    return x / (1.0 + np.exp(-x))


def _ln(x):
    mu = x.mean(-1, keepdims=True)
    var = ((x - mu) ** 2).mean(-1, keepdims=True)
    return (x - mu) / np.sqrt(var + LN_EPS)


def _rms(x, w):
    return x / np.sqrt((x * x).mean(-1, keepdims=True) + RMS_EPS) * w


def _attn(q, k, v, D):
    """q [Sq, H, D], k/v [Sk, H, D] -> [Sq, H*D]; explicit softmax(q k^T / sqrt(D)) v (klein_model.rs:470-483)."""
    s = np.einsum("qhd,khd->hqk", q, k) / np.sqrt(np.float32(D))
    s = s - s.max(-1, keepdims=True)
    p = np.exp(s)
    p /= p.sum(-1, keepdims=True)
    return np.einsum("hqk,khd->qhd", p, v).reshape(q.shape[0], -1)


class KleinOracle:
    """`tp = (world, allreduce)`: this instance is ONE tensor-parallel rank -- `weights` are its shards
    (klein.shard_state_dict), heads and MLP width are the per-rank ones, and `allreduce` sums the partial
    outputs of the row-split projections where the device engine calls RCCL (tests/test_klein_tp.py)."""

    def __init__(self, p: KleinParams, weights: Dict[str, np.ndarray], tp=None):
        self.p = p
        self.w = {k: np.asarray(v, np.float64) for k, v in weights.items()}
        world, self.reduce = tp if tp is not None else (1, lambda x: x)
        self.H = p.num_heads // world              # local heads
        self.hq = self.H * p.head_dim              # local attention width
        self.m = p.mlp_hidden // world             # local MLP width

    def lin(self, x, name):
        return x @ self.w[name].T

    def modulation(self, vec, name, n):
        return np.split(self.lin(_silu(vec), name), n, axis=-1)

    def double_block(self, i, img, txt, img_mod, txt_mod, cos, sin):
        p, b = self.p, f"double_blocks.{i}."
        H, D, St = self.H, p.head_dim, txt.shape[0]
        xs = {"img": img, "txt": txt}
        mods = {"img": img_mod, "txt": txt_mod}
        q, k, v = {}, {}, {}
        for st in ("img", "txt"):
            sh1, sc1 = mods[st][0], mods[st][1]
            xm = (1.0 + sc1) * _ln(xs[st]) + sh1
            rc_, rs_ = (cos[:St], sin[:St]) if st == "txt" else (cos[St:], sin[St:])
            q[st] = apply_rope(_rms(self.lin(xm, b + f"{st}_to_q.weight").reshape(-1, H, D), self.w[b + f"{st}_norm_q.weight"]), rc_, rs_)
            k[st] = apply_rope(_rms(self.lin(xm, b + f"{st}_to_k.weight").reshape(-1, H, D), self.w[b + f"{st}_norm_k.weight"]), rc_, rs_)
            v[st] = self.lin(xm, b + f"{st}_to_v.weight").reshape(-1, H, D)
        kk = np.concatenate([k["txt"], k["img"]], 0)
        vv = np.concatenate([v["txt"], v["img"]], 0)
        out = {}
        for st in ("img", "txt"):
            g1, sh2, sc2, g2 = mods[st][2], mods[st][3], mods[st][4], mods[st][5]
            a = self.reduce(self.lin(_attn(q[st], kk, vv, D), b + f"{st}_to_out.weight"))
            x = xs[st] + a * g1
            xm = (1.0 + sc2) * _ln(x) + sh2
            proj = self.lin(xm, b + f"{st}_mlp_in.weight")
            gate_, up_ = proj[:, :self.m], proj[:, self.m:]
            out[st] = x + self.reduce(self.lin(_silu(gate_) * up_, b + f"{st}_mlp_out.weight")) * g2
        return out["img"], out["txt"]

    def single_block(self, i, x, mod, cos, sin):
        p, b = self.p, f"single_blocks.{i}."
        H, D, h, m = self.H, p.head_dim, self.hq, self.m
        shift, scale, g = mod
        proj = self.lin((1.0 + scale) * _ln(x) + shift, b + "to_qkv_mlp.weight")
        q, k, v, mg, mu = np.split(proj, [h, 2 * h, 3 * h, 3 * h + m], axis=-1)
        q = apply_rope(_rms(q.reshape(-1, H, D), self.w[b + "norm_q.weight"]), cos, sin)
        k = apply_rope(_rms(k.reshape(-1, H, D), self.w[b + "norm_k.weight"]), cos, sin)
        a = _attn(q, k, v.reshape(-1, H, D), D)
        out = self.reduce(self.lin(np.concatenate([a, _silu(mg) * mu], -1), b + "to_out.weight"))
        return x + out * g

    def forward_with_rope(self, img, txt, timestep, cos, sin, return_intermediates=False):
        """img [S_img, in_channels], txt [S_txt, txt_embed_dim], timestep = t*1000 (generate_klein.rs:434)."""
        p = self.p
        St = txt.shape[0]
        img = self.lin(np.asarray(img, np.float64), "x_embedder.weight")
        txt = self.lin(np.asarray(txt, np.float64), "context_embedder.weight")
        vec = self.lin(_silu(self.lin(timestep_embedding(timestep), "time_embed_1.weight")), "time_embed_2.weight")
        img_mod = self.modulation(vec, "double_mod_img.linear.weight", 6)
        txt_mod = self.modulation(vec, "double_mod_txt.linear.weight", 6)
        single_mod = self.modulation(vec, "single_mod.linear.weight", 3)
        inter = {}
        for i in range(p.depth):
            img, txt = self.double_block(i, img, txt, img_mod, txt_mod, cos, sin)
        inter["after_double_img"], inter["after_double_txt"] = img, txt
        x = np.concatenate([txt, img], 0)
        for i in range(p.depth_single):
            x = self.single_block(i, x, single_mod, cos, sin)
        inter["after_single"] = x
        img_out = x[St:]
        scale, shift = np.split(self.lin(_silu(vec), "norm_out.weight"), 2, axis=-1)      # chunk order [scale, shift] (:847-849)
        out = self.lin((1.0 + scale) * _rms(img_out, 1.0) + shift, "proj_out.weight")
        return (out, inter) if return_intermediates else out


def euler_step(latent, v, t_curr: float, t_next: float):
    """generate_klein.rs:441-443 / sampler.rs step: latent += (t_next - t_curr) * v."""
    return latent + (t_next - t_curr) * v


# --------------------------------------------------------------------------
# 8f rank 3: the sampling loop around the DiT (test infrastructure, like everything in oracle/)
#   flux-klein-mlx/src/sampler.rs:104-186, 291-301; examples/generate_klein.rs:456-468, 558-604
# --------------------------------------------------------------------------
_f = np.float32


def compute_empirical_mu(image_seq_len: int, num_steps: int):
    """generate_klein.rs:560-577 (f32 arithmetic, one rounding per operation)."""
    a1, b1, a2, b2 = _f(8.73809524e-05), _f(1.89833333), _f(0.00016927), _f(0.45666666)
    n = _f(image_seq_len)
    if image_seq_len > 4300:
        return a2 * n + b2
    m_200 = a2 * n + b2
    m_10 = a1 * n + b1
    a = (m_200 - m_10) / _f(190.0)
    b = m_200 - _f(200.0) * a
    return a * _f(num_steps) + b


def generalized_time_snr_shift(t, mu, sigma=1.0):
    """generate_klein.rs:579-588."""
    t, mu, sigma = _f(t), _f(mu), _f(sigma)
    if t <= 0.0:
        return _f(0.0)
    if t >= 1.0:
        return _f(1.0)
    return np.exp(mu) / (np.exp(mu) + np.power(_f(1.0) / t - _f(1.0), sigma))


def official_schedule(num_steps: int, image_seq_len: int):
    """sampler.rs:291-301 / generate_klein.rs:591-604."""
    mu = compute_empirical_mu(image_seq_len, num_steps)
    return [generalized_time_snr_shift(_f(1.0) - _f(i) / _f(num_steps), mu, 1.0) for i in range(num_steps + 1)]


def sampler_timesteps(steps: int, is_schnell: bool, shift: float = 1.15):
    """sampler.rs:104-132."""
    out = []
    for i in range(steps + 1):
        t = _f(1.0) - _f(i) / _f(steps)
        if not is_schnell:
            e = np.exp(_f(shift))
            t = e * t / (_f(1.0) + (e - _f(1.0)) * t)
        out.append(_f(t))
    return out


def add_noise(data, noise, t):
    """sampler.rs:151-164: x_t = t * noise + (1 - t) * data, t broadcast over [batch, 1, 1]."""
    t = np.asarray(t, np.float32).reshape(-1, 1, 1)
    return t * np.asarray(noise, np.float32) + (_f(1.0) - t) * np.asarray(data, np.float32)


def sampler_step(x_t, v_pred, t: float, t_prev: float):
    """sampler.rs:174-186."""
    return np.asarray(x_t, np.float32) + (_f(t_prev) - _f(t)) * np.asarray(v_pred, np.float32)


def unpack_latents(latent, patch_h: int, patch_w: int, z: int = 32, p: int = 2):
    """generate_klein.rs:462-468: reshape [ph, pw, z, p, p] -> transpose (0, 1, 4, 2, 5, 3) of the batched form."""
    x = np.asarray(latent).reshape(1, patch_h, patch_w, z, p, p).transpose(0, 1, 4, 2, 5, 3)
    return x.reshape(patch_h * p, patch_w * p, z)


def denoise(oracle: "KleinOracle", txt_embed, patch_h: int, patch_w: int, num_steps: int, noise):
    """generate_klein.rs:412-446 from a given prior sample `noise` [seq, in_channels] float32."""
    cos, sin = compute_rope(np.concatenate([create_txt_ids(txt_embed.shape[0]), create_img_ids(patch_h, patch_w)], 0))
    ts = official_schedule(num_steps, patch_h * patch_w)
    latent = np.asarray(noise, np.float32)
    for i in range(num_steps):
        v = oracle.forward_with_rope(latent, txt_embed, float(ts[i] * _f(1000.0)), cos, sin)
        latent = euler_step(latent, np.asarray(v, np.float32), ts[i], ts[i + 1]).astype(np.float32)
    return latent
