"""One-shot peer-store all-reduce (csrc/peer_allreduce.hip) across PROCESSES: run under torch.distributed.run, one process per rank
(gloo bootstrap, so that two ranks may share the one GPU of a test box; on a multi-GPU node every rank takes its LOCAL_RANK device
and the stores cross xGMI).  Each rank: self-test against the rank-ordered sum, latency of a 4096-float reduction, then a
tensor-parallel greedy decode of a small Qwen3 through the engine with every all-reduce (hidden partials, argmax key) on the peer
path -- serial prefill first, then the batched one, whose [T, hidden] reductions (like the seeded large messages checked before it)
take the two-shot path; no call needs RCCL.  Rank r writes <out>/rank<r>.json; rank 0 also prints per-call latency percentiles and a
log2 histogram for the step's reduction sizes.
usage: python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 --master-port P tools/peer_allreduce_check.py <out dir> [8b]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["OMX_PREFILL_SERIAL"] = "1"
rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ.get("LOCAL_RANK", "0"))
out_dir = sys.argv[1]

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

torch.cuda.set_device(local % torch.cuda.device_count())
dist.init_process_group("gloo", rank=rank, world_size=world)
import omx_import  # noqa: E402

omx = omx_import.load_package()
from ominix_mlx_amd import comm, engine  # noqa: E402
from oracle import synth  # noqa: E402  (prompt ids only: the checker's seeded inputs)

res = {"rank": rank, "world": world, "device": torch.cuda.current_device()}
pc = comm.PeerComm(comm.torch_all_gather_bytes(dist), rank, world)
res["self_test"] = bool(pc.self_test())      # one-shot rounds, then (two-shot path on) seeded 1 .. 32 MB messages with a late rank + the MoE combine round
res["scope"], res["same_device"] = pc.scope, bool(pc.same_device)
T = omx.ops.Tensor
import numpy as np  # noqa: E402

t = T.from_numpy(np.ones(4096, np.float32), "f32")
dist.barrier()
for _ in range(20):
    pc.allreduce_f32(t)
    t = T.from_numpy(np.ones(4096, np.float32), "f32")
omx.ops.synchronize()
dist.barrier()
reps = 200
t0 = time.perf_counter()
for _ in range(reps):
    omx.lib.omx_peer_allreduce(t.ptr, t.ptr, 4096, comm.NCCL_FLOAT32, 0, pc.comm, None)
omx.ops.synchronize()
res["allreduce_16k_us"] = (time.perf_counter() - t0) * 1e6 / reps
res["aborted_after_loop"] = pc.aborted()


def latency_histogram(n_floats, calls=400):
    """PER-CALL latency of the reduction (each call issued and synchronised on its own, all ranks entering together only at the start:
    rank skew shows up as a tail): percentiles and a log2 histogram in microseconds -- enough to read a first multi-GPU run from its log."""
    buf = T.from_numpy(np.ones(n_floats, np.float32), "f32")
    omx.ops.synchronize()
    dist.barrier()
    us = np.empty(calls)
    for i in range(calls):
        a = time.perf_counter()
        omx.lib.omx_peer_allreduce(buf.ptr, buf.ptr, n_floats, comm.NCCL_FLOAT32, 0, pc.comm, None)
        omx.ops.synchronize()
        us[i] = (time.perf_counter() - a) * 1e6
    edges = [0] + [2 ** k for k in range(1, 15)]
    hist = np.histogram(us, bins=edges + [1e12])[0]
    return {"n_floats": n_floats, "calls": calls, "min": round(float(us.min()), 2), "p50": round(float(np.percentile(us, 50)), 2),
            "p90": round(float(np.percentile(us, 90)), 2), "p99": round(float(np.percentile(us, 99)), 2), "max": round(float(us.max()), 2),
            "log2_buckets_us": {f"<{edges[i + 1]}" if i + 1 < len(edges) else f">={edges[-1]}": int(c) for i, c in enumerate(hist) if c}}


# the two sizes of a TP decode step (hidden-sized f32 partials; the 8-byte argmax key goes through the same path) and a prefill-sized one
res["latency_us"] = [latency_histogram(n) for n in (4096, 2, 4096 * 64)]
if os.environ.get("OMX_PEER_CHECK_BIG") == "1":      # prompt-sized messages through the two-shot path (profiles/r04_peer_two_shot.txt)
    res["latency_us"] += [latency_histogram(n, 100) for n in (2048 * 4096 // 2, 2048 * 4096)]
if rank == 0:
    for h in res["latency_us"]:
        print(f"[peer all-reduce] {h['n_floats']:>7d} f32 x {h['calls']} calls, us per call (host clock incl. launch + sync): "
              f"min {h['min']} p50 {h['p50']} p90 {h['p90']} p99 {h['p99']} max {h['max']}  {h['log2_buckets_us']}", flush=True)



def bf16_round(x):
    u = np.ascontiguousarray(x, np.float32).view(np.uint32).astype(np.uint64)
    return ((u + ((u >> np.uint64(16)) & np.uint64(1)) + np.uint64(0x7FFF)) & np.uint64(0xFFFF0000)).astype(np.uint32).view(np.float32)


def large_check(n, dtype, seed):
    """The two-shot path (messages above 8192 words; no RCCL communicator behind this PeerComm): every rank derives ALL ranks' inputs
    from seeds, so the rank-ordered f32 sum (rounded once for bf16) is known locally and must come back bit for bit."""
    parts = [np.random.default_rng(seed * 16 + r).standard_normal(n).astype(np.float32) for r in range(world)]
    if dtype == "bf16":
        parts = [bf16_round(p) for p in parts]
    want = parts[0].copy()
    for p in parts[1:]:
        want = want + p
    if dtype == "bf16":
        want = bf16_round(want)
    buf = T.from_numpy(parts[rank], dtype)
    rc = omx.lib.omx_peer_allreduce(buf.ptr, buf.ptr, n, comm.NCCL_BFLOAT16 if dtype == "bf16" else comm.NCCL_FLOAT32, 0, pc.comm, None)
    omx.ops.synchronize()
    got = buf.numpy().astype(np.float32).ravel()
    return {"n": n, "dtype": dtype, "rc": int(rc), "equal": bool(rc == 0 and np.array_equal(got, want))}


dist.barrier()
res["large"] = [large_check(n, dt, i) for i, (n, dt) in enumerate([(262144, "f32"), (1000004, "f32"), (2048 * 4096, "bf16"), (40 * 1024 + 8, "bf16"),
                                                                   (8200, "f32"), (2048 * 4096, "bf16")])]
res["aborted_after_large"] = pc.aborted()

cfg = dict(hidden_size=1024, num_hidden_layers=2, intermediate_size=3072, num_attention_heads=8, num_key_value_heads=2, head_dim=128,
           vocab_size=4096, rms_norm_eps=1e-6, rope_theta=1e6, tie_word_embeddings=False)
if len(sys.argv) > 2 and sys.argv[2] == "8b":      # two layers of the real Qwen3-8B shapes (the shards bench.py --gpus N runs)
    import bench
    cfg = dict(bench.QWEN3_8B)
    cfg["num_hidden_layers"] = 2
m = engine.Model(max_context=256, tp_rank=rank, tp_size=world, **cfg)
m.synth_weights()
m.set_comm(pc.comm, pc.fn)
prompt = synth.prompt_ids(40, cfg["vocab_size"])
first = m.prefill(prompt)
rest = m.decode(15)
res["tokens"] = [int(first)] + [int(x) for x in rest]
dist.barrier()
t0 = time.perf_counter()
m.decode(64)
res["step_ms"] = (time.perf_counter() - t0) * 1e3 / 64
res["decode_path"] = m.decode_path()
# the same prompt as ONE batched pass: its [T, hidden] bf16 reductions take the two-shot path
os.environ["OMX_PREFILL_SERIAL"] = "0"
m.reset()
dist.barrier()
res["tokens_batched"] = [int(m.prefill(prompt))] + [int(x) for x in m.decode(15)]
res["aborted"] = pc.aborted()
dist.barrier()
m.close()

# expert parallel: a small Mixtral-shaped model, this rank's experts, a 200-token prompt as ONE batched pass.  On
# this communicator the MoE block's weighted sum is the all-to-all combine + all-gather kernel (peer_moe_combine_kernel);
# OMX_EP_COMBINE=allreduce keeps the [T, hidden] f32 all-reduce (here: the two-shot path).  Same roundings: the two must agree bit for bit.
moe_cfg = dict(hidden_size=512, num_hidden_layers=2, intermediate_size=1024, num_attention_heads=8, num_key_value_heads=2, head_dim=64,
               vocab_size=2048, rms_norm_eps=1e-5, rope_theta=1e6, tie_word_embeddings=False, num_experts=8, num_experts_per_tok=2,
               moe_intermediate_size=1024, moe_mode="mixtral", norm_topk_prob=0, qk_norm=False)
moe_prompt = synth.prompt_ids(200, moe_cfg["vocab_size"])
res["stage_bytes"] = int(omx.lib.omx_peer_comm_stage_bytes(pc.comm))
for label, env in (("exchange", None), ("allreduce", "allreduce")):
    if env is None:
        os.environ.pop("OMX_EP_COMBINE", None)
    else:
        os.environ["OMX_EP_COMBINE"] = env
    before = pc.counts()
    em = engine.Model(max_context=512, ep_rank=rank, ep_size=world, **moe_cfg)
    em.synth_weights()
    em.set_comm(pc.comm, pc.fn)
    dist.barrier()
    toks = [int(em.prefill(moe_prompt))] + [int(x) for x in em.decode(6)]
    em.reset()
    dist.barrier()
    toks += [int(em.prefill(moe_prompt[:9]))] + [int(x) for x in em.decode(3)]      # a short prompt: the block's GEMV form + all-reduce in both modes
    res["ep_" + label] = {"tokens": toks, "logits_crc": int(np.frombuffer(em.last_logits().tobytes(), np.uint32).sum() & 0xFFFFFFFF),
                          "prefill_ms": em.last_prefill_ms(), "launches": {k: v - before[k] for k, v in pc.counts().items()}}
    dist.barrier()
    em.close()
os.environ.pop("OMX_EP_COMBINE", None)
res["aborted_after_ep"] = pc.aborted()
pc.close()
os.makedirs(out_dir, exist_ok=True)
with open(os.path.join(out_dir, f"rank{rank}.json"), "w") as f:
    json.dump(res, f)
dist.barrier()
dist.destroy_process_group()
