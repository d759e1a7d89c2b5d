"""Host mirror of the sparse-MoE block (mixtral-mlx/src/model.rs:280-313 `MixtralSparseMoeBlock`,
qwen3-mlx/src/qwen3_moe.rs:440-508 `MoeBlock`) over omx_moe_forward."""
from __future__ import annotations

import ctypes

from . import UINT32, check, lib
from .ops import Tensor

c_int, c_void_p, c_size_t = ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t
MOE_SIGNATURES = {
    "omx_moe_block_partials": (c_int, [c_void_p, c_void_p, c_void_p, ctypes.c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                       c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "omx_moe_workspace_bytes": (c_int, [c_int, c_int, c_int, c_int, c_int, ctypes.POINTER(c_size_t)]),
    "omx_moe_forward": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                                c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "omx_moe_block_forward": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, ctypes.c_float, c_void_p, c_void_p, c_void_p, c_void_p,
                                      c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "omx_moe_block_forward_q": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, ctypes.c_float, c_void_p] + [c_void_p] * 12 + [c_int] * 9 + [c_void_p]),
    # the batched expert-parallel block up to the expert outputs (out: struct of four pointers) -> comm.omx_peer_moe_combine
    "omx_moe_block_slots_ep": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                                       c_int, c_int, c_void_p]),
    "omx_moe_block_partial_ep": (c_int, [c_void_p, c_void_p, c_void_p, ctypes.c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                         c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "omx_moe_forward_q": (c_int, [c_void_p] * 12 + [c_int] * 9 + [c_void_p, c_void_p, c_void_p]),
    "omx_moe_block_forward_q_ex": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, ctypes.c_float, c_void_p] + [c_void_p] * 12 + [c_int] * 10 + [c_void_p]),
    # expert tensor parallel (tp_size > 1 on a sparse-MoE engine): this rank's columns of every expert, f32 slot partials, combine
    "omx_moe_block_partial_tp": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, ctypes.c_float, c_void_p, c_void_p, c_void_p,
                                         c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "omx_moe_combine_slots": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    # the two sharded forms on MLX-packed expert stacks (round 5): partial, x, norm_w, eps, xn, 12 triplet pointers, shapes, shard, format
    "omx_moe_block_partial_ep_q": (c_int, [c_void_p, c_void_p, c_void_p, ctypes.c_float, c_void_p] + [c_void_p] * 12 + [c_int] * 12 + [c_void_p]),
    "omx_moe_block_partial_tp_q": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, ctypes.c_float] + [c_void_p] * 12 + [c_int] * 10 + [c_void_p]),
    "omx_moe_combine_slots_ex": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
}
for _n, (_r, _a) in MOE_SIGNATURES.items():
    _f = getattr(lib, _n)
    _f.restype, _f.argtypes = _r, _a


class SparseMoeBlock:
    """gate: Linear [E, hidden]; switch_mlp.{gate,up,down}_proj stacked [E, ...] (model.rs:478-506)."""

    def __init__(self, gate_w: Tensor, w_gate: Tensor, w_up: Tensor, w_down: Tensor, num_experts_per_tok: int,
                 mode: str = "mixtral", norm_topk_prob: bool = True):
        self.gate_w, self.w_gate, self.w_up, self.w_down = gate_w, w_gate, w_up, w_down
        self.E, self.hidden = gate_w.shape
        self.inter = w_gate.shape[1]
        self.k = num_experts_per_tok
        self.mode = {"mixtral": 0, "qwen3_moe": 1}[mode]
        self.norm = int(norm_topk_prob)

    def forward(self, x: Tensor, return_routing: bool = False):
        n = x.size // self.hidden
        out = Tensor(x.shape, x.dtype)
        inds = Tensor((n, self.k), UINT32) if return_routing else None
        scores = Tensor((n, self.k), x.dtype) if return_routing else None
        check(lib.omx_moe_forward(out.ptr, x.ptr, self.gate_w.ptr, self.w_gate.ptr, self.w_up.ptr, self.w_down.ptr, n,
                                  self.hidden, self.inter, self.E, self.k, self.mode, self.norm,
                                  inds.ptr if inds else None, scores.ptr if scores else None, None))
        return (out, inds, scores) if return_routing else out


class QuantizedSparseMoeBlock:
    """MixtralSparseMoeBlock with `QuantizedSwitchLinear` experts (mixtral-mlx/src/model.rs:182-313): each of gate / up / down
    is a (packed u32 [E, out, in*bits/32], scales, biases) triplet of device Tensors; the router gate is bf16 [E, hidden]."""

    def __init__(self, gate_w: Tensor, q_gate, q_up, q_down, num_experts_per_tok: int, group_size: int = 64, bits: int = 4,
                 mode: str = "mixtral", norm_topk_prob: bool = True):
        self.gate_w, self.q_gate, self.q_up, self.q_down = gate_w, tuple(q_gate), tuple(q_up), tuple(q_down)
        self.E, self.hidden = gate_w.shape
        self.inter = self.q_gate[0].shape[1]
        self.k, self.group, self.bits = num_experts_per_tok, group_size, bits
        self.mode = {"mixtral": 0, "qwen3_moe": 1}[mode]
        self.norm = int(norm_topk_prob)

    def forward(self, x: Tensor, return_routing: bool = False):
        n = x.size // self.hidden
        out = Tensor(x.shape, x.dtype)
        inds = Tensor((n, self.k), UINT32) if return_routing else None
        scores = Tensor((n, self.k), x.dtype) if return_routing else None
        ptrs = [t.ptr for trip in (self.q_gate, self.q_up, self.q_down) for t in trip]
        check(lib.omx_moe_forward_q(out.ptr, x.ptr, self.gate_w.ptr, *ptrs, n, self.hidden, self.inter, self.E, self.k, self.mode,
                                    self.norm, self.group, self.bits, inds.ptr if inds else None, scores.ptr if scores else None, None))
        return (out, inds, scores) if return_routing else out
