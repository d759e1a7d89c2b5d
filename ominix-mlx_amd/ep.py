"""Expert-parallel sparse-MoE block (BASELINE config 3): the experts of
`MixtralSparseMoeBlock` (mixtral-mlx/src/model.rs:280-313) / `MoeBlock` (qwen3_moe.rs:440-508) are
sharded over the ranks of one node, tokens are sharded too, and the block becomes

    route (local tokens)  ->  all-to-all(v) dispatch of token rows to the ranks that own their experts
    -> SwitchGLU on the received rows (local experts)  ->  all-to-all(v) back  ->  weighted sum (local)

Rank r owns the contiguous experts [r*E/n, (r+1)*E/n); the router weight is replicated.  Every
(token, slot) pair is computed by exactly one expert exactly as on one device, and the weighted sum
runs in slot order, so the result equals the single-device block row for row.

`plan_dispatch` is plain numpy (shared by the device path below and by the world-size-2 gloo test,
tests/test_ep_plan.py).  The exchange object hides the fabric: `comm.RcclExchange` (ncclSend/ncclRecv
groups over xGMI), `LoopbackExchange` (several ranks driven from one process: tests on one GPU)."""
from __future__ import annotations

import ctypes
from typing import List

import numpy as np

from . import UINT32, OmxError, check, lib
from .ops import Tensor, take_rows

c_int, c_void_p = ctypes.c_int, ctypes.c_void_p
EP_SIGNATURES = {
    "omx_moe_route": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "omx_moe_experts": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "omx_moe_combine": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
}
for _n, (_r, _a) in EP_SIGNATURES.items():
    _f = getattr(lib, _n)
    _f.restype, _f.argtypes = _r, _a


def experts_of_rank(n_experts: int, rank: int, world: int) -> range:
    if n_experts % world:
        raise ValueError(f"InvalidConfig: num_local_experts={n_experts} is not divisible by ep_size={world}")
    per = n_experts // world
    return range(rank * per, (rank + 1) * per)


def shard_experts(stacked: np.ndarray, rank: int, world: int) -> np.ndarray:
    """This rank's slice of a stacked expert tensor [E, ...] (switch_mlp.{gate,up,down}_proj.weight)."""
    r = experts_of_rank(stacked.shape[0], rank, world)
    from .loader import keep_kind
    return keep_kind(stacked, np.ascontiguousarray(stacked[r.start:r.stop]))


def plan_dispatch(inds: np.ndarray, n_experts: int, world: int):
    """inds [N, k] (global expert ids of the local tokens) ->
         order        [N*k] slot indices (slot = token*k + j) grouped by destination rank, stable inside a rank
         send_counts  [world] rows sent to each rank
         local_expert [N*k] expert id relative to the destination rank, in `order`
       The rows come back in the same order, so  y_slots[order[i]] = y_returned[i]."""
    flat = np.asarray(inds, np.int64).reshape(-1)
    per = n_experts // world
    dest = flat // per
    order = np.argsort(dest, kind="stable")
    send_counts = np.bincount(dest, minlength=world).astype(np.int64)
    local_expert = (flat - dest * per)[order]
    return order, send_counts, local_expert


class LoopbackExchange:
    """All ranks live in this process (tests): rank r deposits, then every rank collects."""

    def __init__(self, world: int):
        self.world = world
        self._counts: List = [None] * world
        self._rows: List = [None] * world

    def put_counts(self, rank, send_counts):
        self._counts[rank] = list(send_counts)

    def get_counts(self, rank):
        return [self._counts[src][rank] for src in range(self.world)]

    def put_rows(self, rank, host_rows: np.ndarray, send_counts):
        offs = np.concatenate([[0], np.cumsum(send_counts)])
        self._rows[rank] = [host_rows[offs[p]:offs[p + 1]] for p in range(self.world)]

    def get_rows(self, rank):
        return np.concatenate([self._rows[src][rank] for src in range(self.world)], axis=0)


class ExpertParallelMoe:
    """One rank's shard of the block on its GPU.  gate_w [E, hidden] replicated; w_* [E/world, ...] local."""

    def __init__(self, gate_w: Tensor, w_gate: Tensor, w_up: Tensor, w_down: Tensor, n_experts: int,
                 num_experts_per_tok: int, rank: int, world: int, exchange, mode: str = "mixtral",
                 norm_topk_prob: bool = True):
        self.gate_w, self.w_gate, self.w_up, self.w_down = gate_w, w_gate, w_up, w_down
        self.E, self.hidden = n_experts, gate_w.shape[1]
        self.E_local = len(experts_of_rank(n_experts, rank, world))
        if w_gate.shape[0] != self.E_local:
            raise OmxError(f"expert shard holds {w_gate.shape[0]} experts, rank {rank}/{world} of {n_experts} needs {self.E_local}")
        self.inter = w_gate.shape[1]
        self.k, self.rank, self.world, self.exchange = num_experts_per_tok, rank, world, exchange
        self.mode = {"mixtral": 0, "qwen3_moe": 1}[mode]
        self.norm = int(norm_topk_prob)

    # the block is split at the two exchanges so that a loopback test can interleave several ranks
    def dispatch(self, x: Tensor):
        n = x.size // self.hidden
        self._n, self._dtype = n, x.dtype
        self._inds = Tensor((n, self.k), UINT32)
        self._scores = Tensor((n, self.k), x.dtype)
        check(lib.omx_moe_route(self._inds.ptr, self._scores.ptr, x.ptr, self.gate_w.ptr, n, self.hidden, self.E, self.k,
                                self.mode, self.norm, None))
        order, send_counts, local_expert = plan_dispatch(self._inds.numpy(), self.E, self.world)   # host sync
        self._order, self._send_counts = order, [int(c) for c in send_counts]
        rows = take_rows(x.view((n, self.hidden)), Tensor.from_numpy((order // self.k).astype(np.uint32), "u32"))
        return rows, Tensor.from_numpy(local_expert.astype(np.uint32), "u32")

    def experts(self, rows: Tensor, expert_ids: Tensor, m: int) -> Tensor:
        y = Tensor((max(m, 1), self.hidden), self._dtype)
        check(lib.omx_moe_experts(y.ptr, rows.ptr, expert_ids.ptr, m, self.w_gate.ptr, self.w_up.ptr, self.w_down.ptr,
                                  self.hidden, self.inter, self.E_local, None))
        return y

    def combine(self, y_returned: Tensor) -> Tensor:
        inv = np.empty_like(self._order)
        inv[self._order] = np.arange(self._order.size)
        y_slots = take_rows(y_returned.view((max(self._order.size, 1), self.hidden)), Tensor.from_numpy(inv.astype(np.uint32), "u32"))
        out = Tensor((self._n, self.hidden), self._dtype)
        check(lib.omx_moe_combine(out.ptr, y_slots.ptr, self._scores.ptr, self._n, self.hidden, self.k, None))
        return out

    def forward(self, x: Tensor) -> Tensor:
        """The whole block with a fabric exchange (comm.RcclExchange)."""
        ex = self.exchange
        rows, eids = self.dispatch(x)
        recv_counts = ex.counts(self._send_counts)
        row_bytes = self.hidden * 2
        got_rows = ex.rows(rows, self._send_counts, recv_counts, row_bytes)
        got_eids = ex.rows(eids, self._send_counts, recv_counts, 4)
        m = sum(recv_counts)
        y = self.experts(Tensor((max(m, 1), self.hidden), self._dtype, ptr=got_rows.ptr, owner=got_rows),
                         Tensor((max(m, 1),), UINT32, ptr=got_eids.ptr, owner=got_eids), m)
        back = ex.rows(y, recv_counts, self._send_counts, row_bytes)
        return self.combine(Tensor((max(sum(self._send_counts), 1), self.hidden), self._dtype, ptr=back.ptr, owner=back))
