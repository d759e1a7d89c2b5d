"""Host side of the FLUX.2-klein generation loop around the DiT, mirroring
flux-klein-mlx/examples/generate_klein.rs steps 6-8 and flux-klein-mlx/src/sampler.rs:

    FluxSamplerConfig / FluxSampler.timesteps, time_shift, add_noise, step     sampler.rs:22-186
    official_schedule (= flux_official_schedule), compute_empirical_mu,
    generalized_time_snr_shift                                                 sampler.rs:291-301, generate_klein.rs:558-604
    denoise: prior noise, RoPE tables once, Euler steps                        generate_klein.rs:412-446
    unpack_latents [seq, 128] -> [H, W, 32] for the VAE                        generate_klein.rs:456-468

Scalar schedule arithmetic is float32 like the Rust (`f32`); tensors stay on the device (the master latent in
float32, the DiT reads a bf16 copy).  The VAE decoder itself (autoencoder.rs) is not part of this build."""
from __future__ import annotations

import numpy as np

from . import check, lib
from .ops import Tensor

f32 = np.float32


class FluxSamplerConfig:
    """sampler.rs:22-78."""

    def __init__(self, num_steps: int, guidance_scale: float, is_schnell: bool, shift: float):
        self.num_steps, self.guidance_scale, self.is_schnell, self.shift = num_steps, guidance_scale, is_schnell, shift

    @staticmethod
    def schnell():
        return FluxSamplerConfig(4, 0.0, True, 1.0)

    @staticmethod
    def dev():
        return FluxSamplerConfig(50, 3.5, False, 1.15)


class FluxSampler:
    """sampler.rs:80-186 (schedule and the two tensor formulas, on host scalars / numpy arrays)."""

    def __init__(self, config: FluxSamplerConfig):
        self.config = config

    def time_shift(self, t) -> np.float32:
        e = f32(np.exp(f32(self.config.shift)))
        t = f32(t)
        return f32(f32(e * t) / f32(f32(1.0) + f32(f32(e - f32(1.0)) * t)))

    def timesteps(self, num_steps=None):
        steps = self.config.num_steps if num_steps is None else num_steps
        ts = [f32(f32(1.0) - f32(f32(i) / f32(steps))) for i in range(steps + 1)]
        return ts if self.config.is_schnell else [self.time_shift(t) for t in ts]

    @staticmethod
    def add_noise(data, noise, t):
        t = np.asarray(t, f32).reshape(-1, 1, 1)
        return (t * np.asarray(noise, f32) + (f32(1.0) - t) * np.asarray(data, f32)).astype(f32)

    @staticmethod
    def step(x_t, v_pred, t: float, t_prev: float):
        dt = f32(f32(t_prev) - f32(t))
        return (np.asarray(x_t, f32) + dt * np.asarray(v_pred, f32)).astype(f32)


def compute_empirical_mu(image_seq_len: int, num_steps: int) -> np.float32:
    a1, b1, a2, b2 = f32(8.73809524e-05), f32(1.89833333), f32(0.00016927), f32(0.45666666)
    n = f32(image_seq_len)
    if image_seq_len > 4300:
        return f32(f32(a2 * n) + b2)
    m_200 = f32(f32(a2 * n) + b2)
    m_10 = f32(f32(a1 * n) + b1)
    a = f32(f32(m_200 - m_10) / f32(190.0))
    b = f32(m_200 - f32(f32(200.0) * a))
    return f32(f32(a * f32(num_steps)) + b)


def generalized_time_snr_shift(t, mu, sigma=1.0) -> np.float32:
    t, mu, sigma = f32(t), f32(mu), f32(sigma)
    if t <= 0.0:
        return f32(0.0)
    if t >= 1.0:
        return f32(1.0)
    e = f32(np.exp(mu))
    return f32(e / f32(e + f32(np.power(f32(f32(1.0) / t - f32(1.0)), sigma))))


def official_schedule(num_steps: int, image_seq_len: int):
    """sampler.rs:291-301 == generate_klein.rs:591-604: num_steps + 1 values from 1.0 down to 0.0."""
    mu = compute_empirical_mu(image_seq_len, num_steps)
    return [generalized_time_snr_shift(f32(f32(1.0) - f32(f32(i) / f32(num_steps))), mu, 1.0) for i in range(num_steps + 1)]


def unpack_latents(latent: np.ndarray, patch_h: int, patch_w: int, z_channels: int = 32, patch_size: int = 2) -> np.ndarray:
    """[seq = patch_h*patch_w, z*p*p] -> [patch_h*p, patch_w*p, z] (generate_klein.rs:462-468)."""
    x = np.asarray(latent).reshape(patch_h, patch_w, z_channels, patch_size, patch_size)
    return x.transpose(0, 3, 1, 4, 2).reshape(patch_h * patch_size, patch_w * patch_size, z_channels)


def denoise(model, txt_embed: Tensor, height: int, width: int, num_steps: int, seed: int = 0, on_step=None) -> np.ndarray:
    """generate_klein.rs:392-446 for one image: returns the final latent [seq, in_channels] float32 (host).
    model: klein.FluxKlein; txt_embed: device bf16 [s_txt, txt_embed_dim] (engine.Model.encode output)."""
    from . import klein, ops
    patch_h, patch_w = height // 16, width // 16            # VAE /8, then 2x2 patches
    seq, ch = patch_h * patch_w, model.cfg.in_channels
    ts = official_schedule(num_steps, seq)
    # mlx_rs::random::normal with key = None: the next key of the global state seeded with `seed`
    state = ops.random_key(seed)
    sub = ops.random_split(state, 2)                        # row 1 = the draw's key (RandomState::next)
    key = Tensor((2,), "u32")
    check(lib.omx_memcpy_d2d(key.ptr, sub.ptr + 8, 8, None))
    latent = ops.random_normal(key, (seq, ch))
    latent16 = ops.cast(latent, "bf16")
    s_txt = txt_embed.shape[0]
    rope_cos, rope_sin = klein.compute_rope(klein.create_txt_ids(s_txt), klein.create_img_ids(patch_h, patch_w))
    for i in range(num_steps):
        t_curr, t_next = ts[i], ts[i + 1]
        v = model.forward_with_rope(latent16, txt_embed, float(f32(t_curr * f32(1000.0))), rope_cos, rope_sin)
        check(lib.omx_klein_euler_step(latent.ptr, v.ptr, float(f32(t_next - t_curr)), latent16.ptr, seq * ch, None))
        if on_step is not None:
            on_step(i, float(t_curr), float(t_next), model.last_ms())
    return latent.numpy()
