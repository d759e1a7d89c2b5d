"""Host mirror of flux-klein-mlx's `FluxKlein` (klein_model.rs:686-870) over omx_klein_*: `compute_rope`
(ids -> cos/sin), `forward_with_rope`, and the Euler denoise loop of examples/generate_klein.rs:431-446."""
from __future__ import annotations

import ctypes
import sys

import numpy as np

from . import check, lib, require_device
from .ops import Tensor

c_int, c_float, c_void_p, c_uint32, c_size_t = ctypes.c_int, ctypes.c_float, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_size_t


class KleinConfig(ctypes.Structure):
    _fields_ = [(n, c_int) for n in ("in_channels", "hidden_size", "txt_embed_dim", "num_heads", "depth", "depth_single",
                                     "head_dim", "mlp_hidden", "tp_rank", "tp_size")]


KLEIN_SIGNATURES = {
    "omx_klein_create": (c_int, [ctypes.POINTER(c_void_p), ctypes.POINTER(KleinConfig)]),
    "omx_klein_destroy": (c_int, [c_void_p]),
    "omx_klein_set_weight": (c_int, [c_void_p, ctypes.c_char_p, c_void_p, ctypes.c_size_t]),
    "omx_klein_synth_weights": (c_int, [c_void_p, c_uint32]),
    "omx_klein_set_comm": (c_int, [c_void_p, c_void_p, c_void_p]),
    "omx_klein_forward_with_rope": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_float, c_void_p, c_void_p]),
    "omx_klein_euler_step": (c_int, [c_void_p, c_void_p, c_float, c_void_p, ctypes.c_int64, c_void_p]),
    "omx_klein_last_ms": (c_int, [c_void_p, ctypes.POINTER(c_float)]),
    "omx_klein_debug_read": (c_int, [c_void_p, ctypes.c_char_p, c_void_p, c_size_t]),
}
for _n, (_r, _a) in KLEIN_SIGNATURES.items():
    _f = getattr(lib, _n)
    _f.restype, _f.argtypes = _r, _a

AXES_DIM, THETA = (32, 32, 32, 32), 2000.0


def create_txt_ids(seq_len: int) -> np.ndarray:
    """generate_klein.rs:543-556: (0, 0, 0, s)."""
    ids = np.zeros((seq_len, 4), np.float32)
    ids[:, 3] = np.arange(seq_len)
    return ids


def create_img_ids(h: int, w: int) -> np.ndarray:
    """generate_klein.rs:519-535: (0, y, x, 0)."""
    ids = np.zeros((h * w, 4), np.float32)
    ids[:, 1] = np.repeat(np.arange(h), w)
    ids[:, 2] = np.tile(np.arange(w), h)
    return ids


def compute_rope(txt_ids: np.ndarray, img_ids: np.ndarray):
    """FluxKlein::compute_rope / compute_rope_freqs (klein_model.rs:53-110, 786-797): [txt, img] order, every
    frequency duplicated for its interleaved pair.  The tables are tiny ([S, 128]) and position-only, so they are
    built once on the host (f32 angle, as the reference) and uploaded."""
    ids = np.concatenate([txt_ids, img_ids], 0).astype(np.float32)
    cs, sn = [], []
    for axis, dim in enumerate(AXES_DIM):
        half = dim // 2
        inv = (np.float32(1.0) / np.power(np.float32(THETA), np.float32(2.0) * np.arange(half, dtype=np.float32) / np.float32(dim))).astype(np.float32)
        ang = (ids[:, axis:axis + 1] * inv[None, :]).astype(np.float32)
        cs.append(np.repeat(np.cos(ang.astype(np.float64)), 2, axis=1))
        sn.append(np.repeat(np.sin(ang.astype(np.float64)), 2, axis=1))
    return (Tensor.from_numpy(np.concatenate(cs, 1).astype(np.float32), "f32"),
            Tensor.from_numpy(np.concatenate(sn, 1).astype(np.float32), "f32"))


def shard_state_dict(weights: dict, hidden_size: int, mlp_hidden: int, rank: int, world: int) -> dict:
    """Tensor-parallel shard plan of the DiT, shared by `FluxKlein.load_weights`,
    the device generator (`omx_klein_synth_weights`) and the world-size-2 gloo test (tests/test_klein_tp.py).
    Heads are contiguous per rank (width hl = hidden/world), MLP columns too (ml = mlp_hidden/world):
        row split   to_q / to_k / to_v                      rows  [r*hl, +hl)
                    mlp_in  = [gate; up]                    rows  [r*ml, +ml) of each half
                    to_qkv_mlp = [q; k; v; gate; up]        the rank's rows of each of the five parts
        col split   to_out (double), mlp_out                cols  [r*hl, +hl) / [r*ml, +ml)   -> partial sums
                    to_out (single) over [attn | mlp]       cols  [r*hl, +hl) and h + [r*ml, +ml)
        replicated  embedders, time/modulation MLPs, q/k norms, norm_out, proj_out"""
    if world == 1:
        return dict(weights)
    h, mh, r = hidden_size, mlp_hidden, rank
    if h % world or mh % world:
        raise ValueError(f"InvalidConfig: hidden_size={h} / mlp_hidden={mh} not divisible by tp_size={world}")
    hl, ml = h // world, mh // world
    from .loader import keep_kind
    rows = lambda a, segs: keep_kind(a, np.ascontiguousarray(np.concatenate([a[s:s + n] for s, n in segs], 0)))
    cols = lambda a, segs: keep_kind(a, np.ascontiguousarray(np.concatenate([a[:, s:s + n] for s, n in segs], 1)))
    out = {}
    for name, a in weights.items():
        leaf = name.rsplit(".", 2)[-2] if name.endswith(".weight") else name
        if leaf.endswith(("to_q", "to_k", "to_v")):
            a = rows(a, [(r * hl, hl)])
        elif leaf.endswith("mlp_in"):
            a = rows(a, [(r * ml, ml), (mh + r * ml, ml)])
        elif leaf.endswith("mlp_out"):
            a = cols(a, [(r * ml, ml)])
        elif leaf == "to_qkv_mlp":
            a = rows(a, [(r * hl, hl), (h + r * hl, hl), (2 * h + r * hl, hl), (3 * h + r * ml, ml), (3 * h + mh + r * ml, ml)])
        elif leaf.endswith("to_out"):
            a = cols(a, [(r * hl, hl), (h + r * ml, ml)] if name.startswith("single_blocks.") else [(r * hl, hl)])
        out[name] = a
    return out


class FluxKlein:
    def __init__(self, in_channels=128, hidden_size=3072, txt_embed_dim=7680, num_heads=24, depth=5, depth_single=20,
                 head_dim=128, mlp_hidden=9216, tp_rank=0, tp_size=1):
        require_device()
        self.cfg = KleinConfig(in_channels, hidden_size, txt_embed_dim, num_heads, depth, depth_single, head_dim, mlp_hidden,
                               tp_rank, tp_size)
        self._h = c_void_p()
        check(lib.omx_klein_create(ctypes.byref(self._h), ctypes.byref(self.cfg)))
        self._keep = []

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            lib.omx_klein_destroy(self._h)
            self._h = c_void_p()

    def __del__(self):
        if sys is not None and not sys.is_finalizing():   # (module globals are already None late in shutdown)
            self.close()

    def set_comm(self, comm_ptr: int, allreduce_fn_ptr: int) -> None:
        """RCCL communicator + address of ncclAllReduce (comm.rccl_comm), before the first forward."""
        check(lib.omx_klein_set_comm(self._h, comm_ptr, allreduce_fn_ptr))

    def load_weights(self, weights: dict) -> None:
        """Logical (unsharded) tensors by the reference's internal names; sliced here under tensor parallelism."""
        weights = shard_state_dict(weights, self.cfg.hidden_size, self.cfg.mlp_hidden, self.cfg.tp_rank, self.cfg.tp_size)
        for name, arr in weights.items():
            t = Tensor.from_numpy(arr, "bf16")
            self._keep.append(t)
            check(lib.omx_klein_set_weight(self._h, name.encode(), t.ptr, t.nbytes))

    def synth_weights(self, base_seed: int = 0x0C0FFEE5) -> None:
        check(lib.omx_klein_synth_weights(self._h, base_seed & 0xFFFFFFFF))

    def forward_with_rope(self, img: Tensor, txt: Tensor, timestep: float, rope_cos: Tensor, rope_sin: Tensor) -> Tensor:
        s_img, s_txt = img.shape[-2], txt.shape[-2]
        out = Tensor(img.shape, img.dtype)
        check(lib.omx_klein_forward_with_rope(self._h, out.ptr, img.ptr, txt.ptr, s_img, s_txt, timestep, rope_cos.ptr, rope_sin.ptr))
        return out

    def last_ms(self) -> float:
        v = c_float()
        check(lib.omx_klein_last_ms(self._h, ctypes.byref(v)))
        return v.value
