"""What ONE rank of a tensor-parallel Qwen3-8B decode step costs at its real shard shapes (VERDICT r3 "Next" 4a), on one GPU: the
engine of rank 0 of TP = 1 / 2 / 4 / 8 with all 36 layers, its all-reduces replaced by the identity of a one-rank in-process
communicator -- ONE small kernel launch per reduction on the step's stream, captured in the step graph like a one-hop peer-store
reduction, without its link latency.  Prints per TP degree: the measured step, the rank's algorithmic bytes, and the split the fit
t = bytes / 6.67 TB/s + launches x 3.6 us + reductions x r implies (r solved from the measurement).  The xGMI hop itself
(tools/peer_allreduce_check.py: ~4.4 us per 16 KB one-shot reduction between processes on one GPU) comes on top per reduction.
usage: python tools/tp_shard_step.py [steps] [prompt] [dense|mixtral] [loopback|peer|fused] > profiles/rNN_tp_shard_step.md"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import omx_import  # noqa: E402
omx = omx_import.load_package()
from ominix_mlx_amd import comm, engine  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 64
prompt_n = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
MIXTRAL = len(sys.argv) > 3 and sys.argv[3] == "mixtral"      # expert tensor parallel shards of Mixtral-8x7B
# (round 6) how a reduction is issued: "loopback" = one small launch of an in-process one-rank communicator (the default, round 4's table);
# "peer" = the one-hop peer-store all-reduce itself on a ONE-rank communicator (stores into and polls its own inbox: the kernel a real rank
# runs, without the link); "fused" = the same with OMX_PEER_FUSED=1: the O / down GEMVs reduce their rows in their own epilogue, no launch
COMM = sys.argv[4] if len(sys.argv) > 4 else "loopback"
if COMM == "fused":
    os.environ["OMX_PEER_FUSED"] = "1"
cfg = dict(bench.MIXTRAL_8X7B) if MIXTRAL else dict(bench.QWEN3_8B)
ids = bench.prompt_ids(prompt_n, cfg["vocab_size"])
L = cfg["num_hidden_layers"]
BW, C_LAUNCH = 6.67e12, 3.6e-6
print("| TP | step (ms) | tok/s per replica | rank bytes / token (GB) | bytes / 6.67 TB/s (ms) | 145 launches x 3.6 us (ms) | "
      "left for the 73 reductions (ms) | per reduction launch (us) | speed-up vs TP 1 | byte-only bound | prompt pass, steady (ms) |")
print("|---|---|---|---|---|---|---|---|---|---|---|")
base = None
for tp in (1, 2, 4, 8):
    m = engine.Model(max_context=prompt_n + 3 * steps + 32, tp_rank=0, tp_size=tp, **cfg)
    group = None
    peer = None
    if tp > 1 and COMM == "loopback":
        group = comm.LoopbackGroup(1, max(1 << 24, prompt_n * cfg['hidden_size'] * 4))
        m.set_comm(group.rank_comm(0), group.allreduce_fn)
    elif tp > 1:
        peer = comm.PeerComm(lambda b: [b], 0, 1)
        m.set_comm(peer.comm, peer.fn)
    m.synth_weights()
    m.prefill(ids)
    m.reset()
    m.prefill(ids)                                        # (the first call pays the one-time scratch allocation)
    prompt_ms = m.last_prefill_ms()
    m.decode(8)
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        m.decode(steps)
        best = min(best, (time.perf_counter() - t0) / steps)
    nbytes = m.step_bytes(prompt_n + 8 + steps)
    path = m.decode_path()
    m.close()
    if group is not None:
        group.close()
    if peer is not None:
        peer.close()
    n_launch = (7 * L + 3) if MIXTRAL else 145            # per layer q/k/v, attention + o, router, experts gate/up, down, (fold, combine)
    t_bytes, t_launch = nbytes / BW, n_launch * C_LAUNCH
    n_red = 2 * L + 1 if tp > 1 else 0
    left = best - t_bytes - t_launch
    base = base or best
    print(f"| {tp} ({path}{'' if COMM == 'loopback' or tp == 1 else ', ' + COMM}) | {best * 1e3:.3f} | {1 / best:.1f} | {nbytes / 1e9:.3f} | {t_bytes * 1e3:.3f} | {t_launch * 1e3:.3f} | "
          f"{left * 1e3:.3f} | {(left / n_red * 1e6 if n_red else 0.0):.2f} | {base / best:.2f}x | {base / t_bytes:.2f}x | {prompt_ms:.1f} |", flush=True)
