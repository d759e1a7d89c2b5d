"""Host mirror of the FLUX VAE decoder (flux-klein-mlx/src/autoencoder.rs `Decoder`, config :22-81) over omx_vae_*;
`sanitize_vae_weights` is weights.rs:164-217 (diffusers keys -> the decoder's names, OIHW -> OHWI)."""
from __future__ import annotations

import ctypes
import sys

import numpy as np

from . import check, lib, require_device
from .ops import Tensor

c_int, c_float, c_void_p = ctypes.c_int, ctypes.c_float, ctypes.c_void_p


class VaeConfig(ctypes.Structure):
    _fields_ = [("ch", c_int), ("ch_mult", c_int * 8), ("n_mult", c_int), ("num_res_blocks", c_int), ("z_channels", c_int),
                ("out_ch", c_int), ("scale_factor", c_float), ("shift_factor", c_float)]


VAE_SIGNATURES = {
    "omx_vae_decoder_create": (c_int, [ctypes.POINTER(c_void_p), ctypes.POINTER(VaeConfig)]),
    "omx_vae_decoder_destroy": (c_int, [c_void_p]),
    "omx_vae_decoder_set_weight": (c_int, [c_void_p, ctypes.c_char_p, c_void_p]),
    "omx_vae_decoder_out_shape": (c_int, [c_void_p, c_int, c_int, ctypes.POINTER(c_int), ctypes.POINTER(c_int), ctypes.POINTER(c_int)]),
    "omx_vae_decode": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int]),
    "omx_vae_decoder_last_ms": (c_int, [c_void_p, ctypes.POINTER(c_float)]),
}
for _n, (_r, _a) in VAE_SIGNATURES.items():
    _f = getattr(lib, _n)
    _f.restype, _f.argtypes = _r, _a


def decoder_weight_shapes(ch=128, ch_mult=(1, 2, 4, 4), num_res_blocks=2, z_channels=32, out_ch=3) -> dict:
    """Parameter names and shapes of `Decoder::new` (autoencoder.rs:279-372), Conv2d weights as [out, kH, kW, in]."""
    n = len(ch_mult)
    block_in = ch * ch_mult[-1]
    s = {"post_quant_conv.weight": (z_channels, 1, 1, z_channels), "post_quant_conv.bias": (z_channels,),
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


def random_decoder_weights(seed: int = 0, **cfg) -> dict:
    """Synthetic weights of the right shapes and magnitudes (variance-preserving convolutions, norm scales near 1) for
    benchmarks without a checkpoint."""
    g = np.random.default_rng(seed)
    out = {}
    for name, shape in decoder_weight_shapes(**cfg).items():
        if name.endswith((".bias",)):
            v = 0.05 * g.standard_normal(shape)
        elif len(shape) == 1:
            v = 1.0 + 0.1 * g.standard_normal(shape)
        else:
            v = g.standard_normal(shape) / np.sqrt(int(np.prod(shape[1:])))
        out[name] = v.astype(np.float32)
    return out


def sanitize_vae_weights(weights: dict) -> dict:
    """weights.rs:164-217: keep `post_quant_conv.*` and `decoder.*`, rename to the decoder's fields, Conv2d OIHW -> OHWI."""
    out = {}
    for key, value in weights.items():
        v = np.asarray(value)
        if key.startswith("post_quant_conv."):
            out[key] = v.transpose(0, 2, 3, 1) if key.endswith(".weight") and v.ndim == 4 else v
            continue
        if not key.startswith("decoder."):
            continue
        k = key[8:]
        for a, b in (("mid_block.attentions.0.", "mid_block_attentions_0."), ("mid_block.resnets.0.", "mid_block_resnets_0."),
                     ("mid_block.resnets.1.", "mid_block_resnets_1."), (".to_out.0.", ".to_out."),
                     (".upsamplers.0.conv.", ".upsamplers_0_conv.")):
            k = k.replace(a, b)
        out[k] = v.transpose(0, 2, 3, 1) if k.endswith(".weight") and v.ndim == 4 else v
    return out


class VaeDecoder:
    def __init__(self, ch=128, ch_mult=(1, 2, 4, 4), num_res_blocks=2, z_channels=32, out_ch=3, scale_factor=0.3611,
                 shift_factor=0.1159):
        require_device()
        mult = (c_int * 8)(*list(ch_mult) + [0] * (8 - len(ch_mult)))
        self.cfg = VaeConfig(ch, mult, len(ch_mult), num_res_blocks, z_channels, out_ch, scale_factor, shift_factor)
        self._h = c_void_p()
        check(lib.omx_vae_decoder_create(ctypes.byref(self._h), ctypes.byref(self.cfg)))
        self._keep = []

    def __del__(self):
        if sys is not None and not sys.is_finalizing() and getattr(self, "_h", None) is not None and self._h.value:
            lib.omx_vae_decoder_destroy(self._h)
            self._h = c_void_p()

    def load_weights(self, weights: dict) -> None:
        """Decoder-named tensors (see `sanitize_vae_weights`); a missing `post_quant_conv` is the identity (autoencoder.rs:286-301)."""
        weights = dict(weights)
        z = self.cfg.z_channels
        if "post_quant_conv.weight" not in weights:
            weights["post_quant_conv.weight"] = np.eye(z, dtype=np.float32).reshape(z, 1, 1, z)
            weights["post_quant_conv.bias"] = np.zeros(z, np.float32)
        for name, arr in weights.items():
            t = arr if isinstance(arr, Tensor) else Tensor.from_numpy(np.ascontiguousarray(arr), "bf16")
            self._keep.append(t)
            check(lib.omx_vae_decoder_set_weight(self._h, name.encode(), t.ptr))

    def decode(self, latent: Tensor) -> Tensor:
        """latent [h, w, z_channels] bf16 (device) -> image [8h, 8w, 3] bf16 in [-1, 1]."""
        h, w = latent.shape[0], latent.shape[1]
        oh, ow, oc = c_int(), c_int(), c_int()
        check(lib.omx_vae_decoder_out_shape(self._h, h, w, ctypes.byref(oh), ctypes.byref(ow), ctypes.byref(oc)))
        out = Tensor((oh.value, ow.value, oc.value), "bf16")
        check(lib.omx_vae_decode(self._h, out.ptr, latent.ptr, h, w))
        return out

    def last_ms(self) -> float:
        v = c_float()
        check(lib.omx_vae_decoder_last_ms(self._h, ctypes.byref(v)))
        return v.value


def to_rgb8(image: np.ndarray) -> np.ndarray:
    """generate_klein.rs:477-491: ((x + 1) * 127.5) clamped to [0, 255], rounded to u8."""
    x = (np.asarray(image, np.float32) + np.float32(1.0)) * np.float32(127.5)
    return np.round(np.clip(x, 0.0, 255.0)).astype(np.uint8)


def write_ppm(path, rgb: np.ndarray) -> None:
    """generate_klein.rs:493-504: binary PPM (P6)."""
    h, w, _ = rgb.shape
    with open(path, "wb") as fh:
        fh.write(f"P6\n{w} {h}\n255\n".encode() + np.ascontiguousarray(rgb, np.uint8).tobytes())
