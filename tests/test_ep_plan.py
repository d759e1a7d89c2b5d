"""World-size-2 gloo test (CPU) of the expert-parallel MoE block (SURVEY.md section 8e row 2;
ominix-mlx_amd/ep.py): tokens and experts are sharded over the ranks, token rows travel through
all-to-all(v) exactly where the device path sends them over RCCL, every stage in between is the CPU
restatement of the reference (oracle/ref_moe.py), and the result must EQUAL the single-device
restatement of MixtralSparseMoeBlock::forward / MoeBlock::forward row for row."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import ref_core as rc, ref_moe as rm, synth

E, K, H, I = 8, 2, 64, 128
TOKENS = (5, 3)          # ragged: rank 0 holds 5 tokens, rank 1 holds 3


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _weights():
    return (synth.tensor("gate.weight", (E, H)), synth.tensor("switch_mlp.gate_proj.weight", (E, I, H)),
            synth.tensor("switch_mlp.up_proj.weight", (E, I, H)), synth.tensor("switch_mlp.down_proj.weight", (E, H, I)))


def _tokens():
    return synth.tensor("ep.x", (sum(TOKENS), H), std=1.0)


def _a2a(rows: np.ndarray, send_counts, recv_counts):
    out = torch.zeros((int(sum(recv_counts)),) + rows.shape[1:], dtype=torch.float32)
    dist.all_to_all_single(out, torch.from_numpy(np.ascontiguousarray(rows, np.float32)),
                           output_split_sizes=[int(c) for c in recv_counts], input_split_sizes=[int(c) for c in send_counts])
    return out.numpy()


def _rank_main(rank, world, port, mode, ret):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import omx_import
    omx_import.load_package()
    from ominix_mlx_amd import ep
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    gate_w, w_gate, w_up, w_down = _weights()
    lo = sum(TOKENS[:rank])
    x = _tokens()[lo:lo + TOKENS[rank]]
    wg, wu, wd = (ep.shard_experts(w, rank, world) for w in (w_gate, w_up, w_down))
    # route (replicated router), plan, dispatch
    inds, scores = rm.route_mixtral(x, gate_w, K) if mode == "mixtral" else rm.route_qwen3_moe(x, gate_w, K, True)
    order, send_counts, local_expert = ep.plan_dispatch(inds, E, world)
    cnt = torch.zeros(world, dtype=torch.int64)
    dist.all_to_all_single(cnt, torch.from_numpy(send_counts.copy()))
    recv_counts = cnt.numpy()
    rows = _a2a(x[order // K], send_counts, recv_counts)
    eids = _a2a(local_expert[:, None].astype(np.float32), send_counts, recv_counts)[:, 0].astype(np.int64)
    assert eids.size == 0 or (eids.min() >= 0 and eids.max() < E // world)
    # local experts on the received rows, then the way back
    y = rm.switch_glu(rows, eids[:, None], wg, wu, wd)[:, 0] if rows.shape[0] else np.zeros((0, H), np.float32)
    back = _a2a(y, recv_counts, send_counts)
    y_slots = np.empty((x.shape[0] * K, H), np.float32)
    y_slots[order] = back
    y_slots = y_slots.reshape(x.shape[0], K, H)
    weighted = rc.rnd(y_slots.astype(np.float64) * scores[..., None].astype(np.float64), "bf16")
    out = rc.rnd(np.sum(weighted.astype(np.float64), axis=1), "bf16")
    ret[rank] = (out, int(recv_counts.sum()))
    dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["mixtral", "qwen3_moe"])
def test_expert_parallel_block_equals_single_device(mode):
    world = 2
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_rank_main, args=(world, _free_port(), mode, ret), nprocs=world, join=True)
        got = np.concatenate([ret[r][0] for r in range(world)], axis=0)
        assert sum(ret[r][1] for r in range(world)) == sum(TOKENS) * K      # every (token, slot) row was served once
    gate_w, w_gate, w_up, w_down = _weights()
    want, _, _ = rm.moe_block(_tokens(), gate_w, w_gate, w_up, w_down, K, mode)
    np.testing.assert_array_equal(got, want)


def test_plan_dispatch_is_a_permutation_grouped_by_owner():
    import omx_import
    omx_import.load_package()
    from ominix_mlx_amd import ep
    rng = np.random.default_rng(3)
    inds = np.stack([rng.permutation(E)[:K] for _ in range(37)])
    order, counts, local = ep.plan_dispatch(inds, E, 4)
    assert sorted(order.tolist()) == list(range(37 * K)) and counts.sum() == 37 * K
    dest = inds.reshape(-1)[order] // (E // 4)
    assert np.all(np.diff(dest) >= 0)                                   # grouped by destination rank
    np.testing.assert_array_equal(dest * (E // 4) + local, inds.reshape(-1)[order])
    with pytest.raises(ValueError):
        ep.experts_of_rank(E, 0, 3)
