"""TEST INFRASTRUCTURE ONLY -- CPU (numpy) restatement of the sparse-MoE block of the hot path
(SURVEY.md 8a rows a6, a7).  Never imported by the product path.

Follows:
  MixtralSparseMoeBlock::forward   mixtral-mlx/src/model.rs:296-308   (top-k on logits, softmax over the
                                    SELECTED logits, precise)
  MoeBlock::forward (Qwen3-MoE)    qwen3-mlx/src/qwen3_moe.rs:475-503 (softmax over ALL experts, top-k,
                                    optional renormalisation)
  SwitchGLU::forward_experts       mixtral-mlx/src/model.rs:243-274   (gate/up gather-matmul, fused_swiglu,
                                    down gather-matmul; the sort path only permutes rows)
The reference stores expert weights 4-bit (gather_qmm, model.rs:194-201, loader rejects anything else at
:554-556); the bf16 build of BASELINE.json config 3 uses dense stacked weights [E, out, in], which is the
`gather_mm` form of the same op (mlx-rs/src/ops/quantization.rs:169-203).  PARITY UNPINNED: no reference
test covers the MoE block.

Top-k order: argpartition leaves the order of the k selected experts unspecified; the result is a
commutative sum over them.  This restatement selects in descending score order, ties to the lower index.
"""
from __future__ import annotations

import numpy as np

from . import ref_core as rc


def topk_indices(scores: np.ndarray, k: int) -> np.ndarray:
    """Indices of the k largest entries of the last axis, descending, ties -> lower index."""
    order = np.argsort(-scores.astype(np.float64), axis=-1, kind="stable")
    return order[..., :k]


def route_logits_mixtral(gates, k: int, dt: str = "bf16"):
    """model.rs:298-302 from the gate Linear's output (dense or quantised)."""
    inds = topk_indices(gates, k)
    sel = np.take_along_axis(gates, inds, axis=-1)
    return inds, rc.softmax(sel, -1, dt)


def route_mixtral(x, gate_w, k: int, dt: str = "bf16"):
    """model.rs:296-302 -> (inds [.., k], scores [.., k] in dt)."""
    return route_logits_mixtral(rc.linear(x, gate_w, None, dt), k, dt)


def route_logits_qwen3_moe(logits, k: int, norm_topk_prob: bool, dt: str = "bf16"):
    """qwen3_moe.rs:479-494 from the gate Linear's output."""
    gates = rc.softmax(logits, -1, dt)
    inds = topk_indices(gates, k)
    sel = np.take_along_axis(gates, inds, axis=-1)
    if norm_topk_prob and k > 1:
        s = rc.rnd(np.sum(sel.astype(np.float64), axis=-1, keepdims=True), dt)
        sel = rc.rnd(sel.astype(np.float64) / s.astype(np.float64), dt)
    return inds, sel


def route_qwen3_moe(x, gate_w, k: int, norm_topk_prob: bool, dt: str = "bf16"):
    """qwen3_moe.rs:478-494."""
    gates = rc.softmax(rc.linear(x, gate_w, None, dt), -1, dt)
    inds = topk_indices(gates, k)
    sel = np.take_along_axis(gates, inds, axis=-1)
    if norm_topk_prob and k > 1:
        s = rc.rnd(np.sum(sel.astype(np.float64), axis=-1, keepdims=True), dt)
        sel = rc.rnd(sel.astype(np.float64) / s.astype(np.float64), dt)
    return inds, sel


def switch_glu(x, inds, w_gate, w_up, w_down, dt: str = "bf16"):
    """SwitchGLU::forward_experts: x [N, h], inds [N, k] -> [N, k, h]."""
    N, k = inds.shape
    out = np.zeros((N, k, w_down.shape[1]), np.float32)
    for n in range(N):
        for j in range(k):
            e = int(inds[n, j])
            g = rc.linear(x[n:n + 1], w_gate[e], None, dt)
            u = rc.linear(x[n:n + 1], w_up[e], None, dt)
            act = rc.fused_swiglu(u, g, dt)                         # argument order (up, gate), model.rs:257
            out[n, j] = rc.linear(act, w_down[e], None, dt)[0]
    return out


def quantize_experts(w: np.ndarray, group: int = 64, bits: int = 4):
    """[E, out, in] -> MLX triplet (packed u32 [E, out, in*bits/32], scales, biases [E, out, in/group]); the on-disk form of
    `switch_mlp.{gate,up,down}_proj` in the reference's Mixtral checkpoints (mixtral-mlx/src/model.rs:466-548)."""
    q, s, b = zip(*(rc.quantize(w[e], group, bits) for e in range(w.shape[0])))
    return np.stack(q), rc.bf16_round(np.stack(s)), rc.bf16_round(np.stack(b))


def switch_glu_q(x, inds, qg, qu, qd, group: int, bits: int, dt: str = "bf16"):
    """SwitchGLU::forward_experts on quantised stacks: gather_qmm x3 + fused_swiglu (model.rs:195-201, 243-274)."""
    N, k = inds.shape
    out = np.zeros((N, k, qd[0].shape[1]), np.float32)
    for n in range(N):
        for j in range(k):
            e = int(inds[n, j])
            g = rc.quantized_matmul(x[n:n + 1], qg[0][e], qg[1][e], qg[2][e], group, bits, dt)
            u = rc.quantized_matmul(x[n:n + 1], qu[0][e], qu[1][e], qu[2][e], group, bits, dt)
            act = rc.fused_swiglu(u, g, dt)
            out[n, j] = rc.quantized_matmul(act, qd[0][e], qd[1][e], qd[2][e], group, bits, dt)[0]
    return out


def moe_block_q(x, gate_w, qg, qu, qd, k: int, group: int = 64, bits: int = 4, dt: str = "bf16"):
    """MixtralSparseMoeBlock::forward (model.rs:296-308) with quantised experts and a bf16 router."""
    inds, scores = route_mixtral(x, gate_w, k, dt)
    y = switch_glu_q(x, inds, qg, qu, qd, group, bits, dt)
    weighted = rc.rnd(y.astype(np.float64) * scores[..., None].astype(np.float64), dt)
    return rc.rnd(np.sum(weighted.astype(np.float64), axis=1), dt), inds, scores


def moe_block(x, gate_w, w_gate, w_up, w_down, k: int, mode: str = "mixtral", norm_topk_prob: bool = True,
              dt: str = "bf16"):
    """x [N, h] -> [N, h]:  sum_j scores_j * expert_{inds_j}(x)   (model.rs:304-307)."""
    if mode == "mixtral":
        inds, scores = route_mixtral(x, gate_w, k, dt)
    else:
        inds, scores = route_qwen3_moe(x, gate_w, k, norm_topk_prob, dt)
    y = switch_glu(x, inds, w_gate, w_up, w_down, dt)
    weighted = rc.rnd(y.astype(np.float64) * scores[..., None].astype(np.float64), dt)     # y.multiply(scores)
    return rc.rnd(np.sum(weighted.astype(np.float64), axis=1), dt), inds, scores           # .sum_axis(2)
