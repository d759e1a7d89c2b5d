#!/usr/bin/env python3
"""bench.py -- BASELINE.json headline metric on MI355X: greedy decode tokens/s of the
Qwen3-8B-shaped bf16 model ("qwen3-mlx 7B bf16 TP=1 on one MI355X, 2k prefill + 256 decode",
BASELINE.json configs[1]; Qwen3-8B is the in-family size the reference pins).

A "step" is ONE decode token through the whole hot path (36 layers + lm_head + greedy sample)
at a context that starts at --prompt (2048) tokens.  Synthetic weights / prompt (no network).
One process per GPU; for N > 1 the model is tensor-parallel (column-split q/k/v/gate/up/lm_head,
row-split o/down, KV heads sharded) with RCCL all-reduce over xGMI -- strong scaling (one token
stream, fixed total work).

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` for the dominant
kernel (gate/up GEMV + SwiGLU) and `cpu_baseline` (plain-C port of the same decode step timed on
the host cores; oracle/c/omx_oracle.c).
"""
from __future__ import annotations

import argparse
import ctypes
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

N_WINDOWS = 3        # timed windows of --steps decode tokens each (the first one is `value`)
SECONDARY_TIMEOUT_S = int(os.environ.get("OMX_BENCH_SECONDARY_TIMEOUT", "420"))   # watchdog of the collective secondaries at N > 1
HBM_PEAK_GBPS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy ceiling)

QWEN3_8B = dict(hidden_size=4096, num_hidden_layers=36, intermediate_size=12288, num_attention_heads=32,
                num_key_value_heads=8, head_dim=128, vocab_size=151936, rms_norm_eps=1e-6, rope_theta=1e6,
                tie_word_embeddings=False)
QWEN3_0_6B = dict(hidden_size=1024, num_hidden_layers=28, intermediate_size=3072, num_attention_heads=16,
                  num_key_value_heads=8, head_dim=128, vocab_size=151936, rms_norm_eps=1e-6, rope_theta=1e6,
                  tie_word_embeddings=True)
MIXTRAL_8X7B = dict(hidden_size=4096, num_hidden_layers=32, intermediate_size=14336, num_attention_heads=32, num_key_value_heads=8,
                    head_dim=128, vocab_size=32000, rms_norm_eps=1e-5, rope_theta=1e6, num_experts=8, num_experts_per_tok=2,
                    moe_intermediate_size=14336, moe_mode="mixtral", qk_norm=False)     # mixtral-mlx/src/model.rs:44-52
QWEN2P5_7B = dict(hidden_size=3584, num_hidden_layers=28, intermediate_size=18944, num_attention_heads=28, num_key_value_heads=4,
                  head_dim=128, vocab_size=152064, rms_norm_eps=1e-6, rope_theta=1e6, tie_word_embeddings=False,
                  qk_norm=False, attention_bias=True)      # the qwen2.rs wiring ("also report Qwen2.5-7B shapes")
MODELS = {"qwen3-8b": QWEN3_8B, "qwen3-0.6b": QWEN3_0_6B, "mixtral-8x7b": MIXTRAL_8X7B, "qwen2.5-7b": QWEN2P5_7B}


def prompt_ids(n, vocab):
    """SURVEY.md section 8d synthetic prompt: token i = (i * 7919 + 13) mod V."""
    import numpy as np
    return ((np.arange(n, dtype=np.int64) * 7919 + 13) % vocab).astype(np.uint32)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=128)
    ap.add_argument("--warmup", type=int, default=16)
    ap.add_argument("--prompt", type=int, default=2048)
    ap.add_argument("--model", default="qwen3-8b", choices=list(MODELS))
    ap.add_argument("--layers", type=int, default=0, help="debug: override layer count (invalidates the metric)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-only", action="store_true", help="internal: the CPU leg's child process (no GPU use)")
    ap.add_argument("--cpu-ctx", type=int, default=0)
    ap.add_argument("--cpu-cfg", default="")
    ap.add_argument("--no-flux", action="store_true", help="skip the secondary FLUX.2-klein sec/step measurement")
    ap.add_argument("--flux-tp", action="store_true",
                    help="with --gpus N > 1: also time FLUX.2-klein tensor-parallel over the N GPUs (all ranks take part)")
    ap.add_argument("--dry-run", action="store_true",
                    help="launcher / rendezvous / barrier / max-over-ranks plumbing only (gloo, no GPU, no engine): the CPU test of --gpus N")
    return ap.parse_args()


def self_launch(args):
    """`python bench.py --gpus N` outside torch.distributed.run: start the N ranks as a CHILD process (before this process
    has touched the GPU -- an exec from a GPU-initialised process takes the box down), relay its output, return its code."""
    import subprocess
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // args.gpus)))
    # --standalone: the launcher picks (and keeps) a free rendezvous port itself -- a bind-then-close probe can lose the port to
    # another job between the probe and the launch
    cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nnodes=1",
           f"--nproc-per-node={args.gpus}", os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def dry_run(args):
    """The distributed skeleton of main() without a GPU: gloo rendezvous, barrier, a stand-in step, MAX over ranks, one JSON line."""
    import torch
    import torch.distributed as dist
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo")
        dist.barrier()
    t0 = time.perf_counter()
    x = torch.ones(1024)
    for _ in range(args.steps):
        x = x * 1.0001
        if world > 1:
            dist.all_reduce(x)
            x /= world
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        dist.destroy_process_group()
    if rank == 0:
        out = {"metric": "decode_tokens_per_sec", "value": None, "unit": "tokens/s", "n_gpus": world, "steps": args.steps,
               "warmup": args.warmup, "ms_per_step": round(elapsed * 1e3 / max(args.steps, 1), 4), "dry_run": True,
               "config": {"workload": "launcher dry run (gloo, no engine)", "parallelism": f"tp{world}"}}
        # the same objects the real run appends (collective_secondaries): at N > 1 the flag-less command measures all three curves
        for key, par in collective_plan(args, world).items():
            out[key] = {"value": None, "n_gpus": world, "parallelism": par, "dry_run": True}
        print(json.dumps(out), flush=True)


def collective_plan(args, world):
    """Which secondary workloads a run appends as COLLECTIVE measurements (every rank takes part) and with what parallelism:
    BASELINE config 5 (FLUX.2-klein, tensor parallel) and config 3 (Mixtral-8x7B, expert parallel) ride on the default
    `bench.py --gpus N` command whenever a communicator exists (N > 1, or OMX_BENCH_FORCE_COMM=1 on one rank), so the driver's
    one command per N yields all three scaling curves.  --no-flux skips them; a MoE primary model is its own EP run."""
    collective = world > 1 or os.environ.get("OMX_BENCH_FORCE_COMM") == "1"
    if not collective or args.no_flux or MODELS[args.model].get("num_experts", 0) > 0:
        return {}
    plan = {"secondary": f"tp{world}"}
    if 8 % world == 0:            # Mixtral-8x7B: 8 experts over the ranks
        plan["mixtral"] = f"ep{world}"
    return plan


# Pre-flight of the N > 1 command on a box with ONE GPU (OMX_BENCH_ONE_GPU=1): every rank is its own process on device 0, gloo does
# the rendezvous / barriers / max-over-ranks, and EVERY device all-reduce -- the step's 16 KB ones and the [T, hidden] / 28 MB ones that
# otherwise go to RCCL -- runs on the peer communicator (csrc/peer_allreduce.hip: one-shot below 32 KB, two-shot above), whose inboxes
# and stages are HIP IPC mappings whether the peer sits on another GPU or on this one.  Not a measurement: a line from this mode says so.
ONE_GPU = os.environ.get("OMX_BENCH_ONE_GPU") == "1"
if ONE_GPU:
    # N ranks share the CUs: the large-path all-reduce kernels of the ranks that wait must not sit on every CU the last rank's GEMMs need
    os.environ.setdefault("OMX_PEER_LARGE_BLOCKS", str(max(8, 64 // max(1, int(os.environ.get("WORLD_SIZE", "1"))))))
# where the scalars of the host-side reductions (rank status, MAX of the windows' times) live: on the device under the nccl backend;
# under gloo (the one-GPU pre-flight) on the host -- a CUDA tensor reduced through gloo's staging path leaves the process in a state in
# which every later kernel of BOTH ranks runs 3.3x slower (tools/two_rank_windows.py TRW_MODE=cuda_tensor: 12.1 against 3.67 ms per
# step, toggling with every further such call: the "slow first window" of the first pre-flight runs)
REDUCE_DEVICE = "cpu" if ONE_GPU else "cuda"
if ONE_GPU:
    # launches whose workgroups ALL wait on each other (attention + O in one launch, the GEMV that reduces over the peers in its
    # epilogue) need their whole grid resident: with one GPU per rank it is, with two ranks' twin launches interleaved on the same CUs
    # neither ever is -- both give up after their bounded waits.  The pre-flight takes the forms that only wait on producers.
    os.environ.setdefault("OMX_ATTN_OPROJ", "0")
    os.environ.setdefault("OMX_PEER_FUSED", "0")


def init_dist(n_gpus):
    """One process per GPU (torchrun env).  Returns (rank, world, local_rank, dist or None)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if n_gpus != world:
        raise SystemExit(f"--gpus {n_gpus} but WORLD_SIZE={world}: launch with --nproc-per-node {n_gpus} (or without torch.distributed.run)")
    import torch
    if ONE_GPU:
        local = 0
    torch.cuda.set_device(local)
    dist = None
    if world > 1:
        import torch.distributed as dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        with c_stdout_to_stderr():
            if ONE_GPU:                 # RCCL refuses two ranks on one device: gloo carries the host-side plumbing
                dist_mod.init_process_group("gloo")
            else:
                dist_mod.init_process_group("nccl", device_id=torch.device(f"cuda:{local}"))
            dist_mod.barrier()          # torch creates its communicator lazily: do it here, banner and all
        dist = dist_mod
    return rank, world, local, dist


class c_stdout_to_stderr:
    """RCCL prints a version / host banner on the C stdout when a communicator is created; the contract is ONE JSON line on stdout.
    While active, file descriptor 1 points at stderr; the C buffer is flushed before the descriptor is restored."""

    def __enter__(self):
        sys.stdout.flush()
        self._saved = os.dup(1)
        os.dup2(2, 1)
        return self

    def __exit__(self, *exc):
        try:
            ctypes.CDLL(None).fflush(None)
        finally:
            os.dup2(self._saved, 1)
            os.close(self._saved)
        return False


def peer_comm(dist, rank, world, rccl):
    """The step's small all-reduces (16 KB of f32 partials, the argmax key) as ONE kernel of tagged xGMI peer stores
    (csrc/peer_allreduce.hip) instead of an RCCL ring launch each; everything larger stays on RCCL.  Used only when every rank
    mapped every inbox AND reproduced the rank-ordered sums of the self-test exactly; otherwise the run stays on RCCL and says so.
    OMX_PEER_ALLREDUCE=0: RCCL only (A/B).  -> (PeerComm or None, note for the JSON line)"""
    if os.environ.get("OMX_PEER_ALLREDUCE", "1") == "0":
        return None, "rccl (OMX_PEER_ALLREDUCE=0)"
    import torch
    from ominix_mlx_amd import comm
    peer, err = None, ""
    try:
        gather = comm.torch_all_gather_bytes(dist) if dist is not None else (lambda b: [b])
        peer = comm.PeerComm(gather, rank, world, rccl=rccl)
        peer.self_test()
    except Exception as e:   # noqa: BLE001  (any failure: this run uses RCCL)
        err = str(e) or type(e).__name__
    ok = torch.tensor([0 if err else 1], dtype=torch.int32, device=REDUCE_DEVICE)
    if dist is not None:
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
    if int(ok.item()) == 1:
        scope = f"; two-shot / exchange hand-offs at {peer.scope} scope, self-tested on 1-32 MB messages with a late rank" if peer.stage_bytes() else ""
        if rccl is None:
            return peer, ("peer communicator only (csrc/peer_allreduce.hip: one-shot below 32 KB, two-shot above; self-test exact on every rank" + scope + ")" +
                          (" -- ALL RANKS ON ONE GPU (OMX_BENCH_ONE_GPU=1): a pre-flight of the multi-process path, not a measurement" if ONE_GPU else ""))
        return peer, "peer-store one-shot (csrc/peer_allreduce.hip; self-test exact on every rank" + scope + "), RCCL above 32 KB"
    if peer is not None and not err:
        err = "a peer rank failed its self-test"
    print(f"[bench rank {rank}] peer all-reduce not used: {err}", file=sys.stderr, flush=True)
    return None, f"rccl (peer-store path declined: {err[:200]})"


def rccl_comm(dist, rank, world):
    """RCCL communicator for the C++ engine, bootstrapped through torch.distributed (ominix-mlx_amd/comm.py).
    Returns None -- on EVERY rank, agreed through the torch process group -- when any rank cannot load librccl or create / warm up its
    communicator: the run then goes on with the peer communicator alone (its two-shot path takes the large reductions; bench.py's
    one-GPU pre-flight exercises exactly that configuration)."""
    import torch
    from ominix_mlx_amd import comm

    def all_ok(ok):
        if dist is None:
            return bool(ok)
        t = torch.tensor([1 if ok else 0], dtype=torch.int32, device=REDUCE_DEVICE)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return int(t.item()) == 1

    err = ""
    try:
        comm.load_rccl()                       # (not a collective: a rank that cannot load the library must not leave the others in one)
    except Exception as e:   # noqa: BLE001
        err = str(e) or type(e).__name__
    if not all_ok(not err):
        print(f"[bench rank {rank}] RCCL not used: librccl not loadable on every rank ({err or 'another rank'})", file=sys.stderr, flush=True)
        return None
    keep = None
    try:
        with c_stdout_to_stderr():
            keep = comm.rccl_comm(dist, rank, world)
    except Exception as e:   # noqa: BLE001
        err = str(e) or type(e).__name__
    if not all_ok(keep is not None):
        print(f"[bench rank {rank}] RCCL not used: communicator creation failed ({err or 'on another rank'})", file=sys.stderr, flush=True)
        return None
    return keep


def time_dominant_kernel(omx, cfg, world, iters=3):
    """roofline.achieved: average duration of the dominant kernel (RMSNorm + gate/up GEMV + SwiGLU,
    52 % of a step's bytes) over distinct per-layer weight buffers (no Infinity-Cache reuse), HIP
    events on the launch stream (omx_bench_gemv)."""
    lib = omx.lib
    lib.omx_bench_gemv.restype = ctypes.c_int
    lib.omx_bench_gemv.argtypes = [ctypes.c_int] * 7 + [ctypes.POINTER(ctypes.c_float)]
    N, K = cfg["intermediate_size"] // world, cfg["hidden_size"]
    nbytes = 2 * N * K * 2
    copies = max(2, int(1.2e9 // nbytes))
    ms = ctypes.c_float()
    omx.check(lib.omx_bench_gemv(N, K, 1, 2, 0, copies, copies * iters, ctypes.byref(ms)))
    return nbytes, ms.value * 1e-3


DOMINANT_KERNEL = "gemv_kernel<8, 1, 1, 1, 2>"      # RMSNorm + gate/up GEMV + SwiGLU: 52 % of a step's bytes
PMC_JSON = os.path.join(ROOT, "profiles", "r06_pmc_fetch_size.json")
# greedy token after the 2048-token synthetic prompt, per (model, prompt length) on ONE GPU, synthetic weights: the engine is
# deterministic, so a different token means a different computation.  (Committed from the round-2 run; checked at world 1 only --
# tensor-parallel partial sums round differently.)
FIRST_TOKEN = {("qwen3-8b", 2048): 140044}


def pmc_traffic(kernel):
    """HBM bytes per launch of `kernel` from the committed PMC pass (profiles/r06_pmc_fetch_size.json, regenerated from the trimmed
    raw counter_collection.csv by tools/pmc_report.py): rocprofv3 --pmc FETCH_SIZE in its own pass, x 1024 x 2 (gfx950 correction)."""
    try:
        table = json.load(open(PMC_JSON))
        return int(table[kernel.replace(" ", "")]["hbm_bytes"])
    except (OSError, KeyError, ValueError):
        return None


def flux_secondary(omx, steps=8, rank=0, world=1, comm=None):
    """Second half of BASELINE.json's metric string: FLUX.2-klein 1024x1024 sec/step (bf16, synthetic
    weights/latents; full 5 double + 20 single block model, S = 512 txt + 4096 img tokens).  world > 1:
    tensor-parallel over the node (heads and MLP columns sharded, bf16 all-reduce over RCCL)."""
    import numpy as np
    from ominix_mlx_amd import klein
    g = 1024 // 16
    s_img, s_txt = g * g, 512
    m = klein.FluxKlein(tp_rank=rank, tp_size=world)
    if comm is not None:
        m.set_comm(comm[1], comm[2])
    m.synth_weights()
    lat = omx.ops.fill_uniform((s_img, 128), 1, 1.7)
    txt = omx.ops.fill_uniform((s_txt, 7680), 2, 1.7)
    rcos, rsin = klein.compute_rope(klein.create_txt_ids(s_txt), klein.create_img_ids(g, g))
    m.forward_with_rope(lat, txt, 1000.0, rcos, rsin)
    ts = []
    for i in range(steps):
        m.forward_with_rope(lat, txt, (1.0 - i / 28.0) * 1000.0, rcos, rsin)
        ts.append(m.last_ms())
    m.close()
    ms = float(np.median(ts))
    flop = 34.79e12   # 28.27 TFLOP linear + 6.52 TFLOP attention at S = 4608
    return {"metric": "flux_klein_1024_sec_per_step", "value": round(ms / 1e3, 5), "unit": "s/step", "higher_is_better": False,
            "n_gpus": world, "steps": steps, "dtype": "bf16", "data": "synthetic", "parallelism": f"tp{world}",
            "spread": {"mean": round(float(np.mean(ts)) / 1e3, 5), "std": round(float(np.std(ts)) / 1e3, 5),
                       "note": f"{steps} consecutive steps of the 28-step schedule (timesteps 1000 (1 - i / 28)), device time of each; value = median"},
            "roofline": {"bound": "mfma", "achieved": round(flop / ms / 1e9, 1), "peak": 2500.0 * world, "unit": "TFLOP/s",
                         "frac": round(flop / ms / 1e9 / (2500.0 * world), 4)}}


def quantized_secondary(omx, cfg, args, bits=4):
    """The reference's flagship mode: the same model as an MLX 4-bit checkpoint (group 64), same
    protocol (2048-token prompt, warm-up, timed greedy decode steps); the decode step streams the packed weights."""
    from ominix_mlx_amd import engine
    max_ctx = args.prompt + args.warmup + args.steps + 16
    m = engine.Model(max_context=max_ctx, quantization={"bits": bits, "group_size": 64}, **cfg)
    m.synth_weights()
    prompt = prompt_ids(args.prompt, cfg["vocab_size"])
    first = m.prefill(prompt)
    prefill_first_ms = m.last_prefill_ms()
    # steady state: the same prompt again on the emptied cache (the scratch is allocated, and the layers' dequantised matrices are kept
    # between prompts -- engine.hip dq_cache, OMX_DEQUANT_CACHE)
    m.reset()
    again = m.prefill(prompt)
    prefill_steady_ms = m.last_prefill_ms()
    if args.warmup:
        m.decode(args.warmup)
    t0 = time.perf_counter()
    toks = m.decode(args.steps)
    omx.check(omx.lib.omx_synchronize(m.stream()))
    elapsed = time.perf_counter() - t0
    ctx_mid = args.prompt + args.warmup + args.steps // 2
    step_bytes = m.step_bytes(ctx_mid)
    out = {"metric": f"decode_tokens_per_sec_{bits}bit", "value": round(args.steps / elapsed, 2), "unit": "tokens/s", "n_gpus": 1,
           "steps": args.steps, "ms_per_step": round(elapsed * 1e3 / args.steps, 4), "higher_is_better": True, "dtype": f"u{bits}/bf16",
           "data": "synthetic", "config": {"workload": f"{args.model} as an MLX {bits}-bit group-64 checkpoint, same protocol"},
           "step_roofline": {"algorithmic_bytes_per_token": int(step_bytes),
                             "achieved_GBps": round(step_bytes / (elapsed / args.steps) / 1e9, 1),
                             "frac_of_hbm_peak": round(step_bytes / (elapsed / args.steps) / 1e9 / HBM_PEAK_GBPS, 4)},
           "prefill_device_ms": round(prefill_first_ms, 2), "prefill_device_ms_steady": round(prefill_steady_ms, 2),
           "first_tokens": [int(first)] + [int(t) for t in toks[:4]], "first_token_again": int(again)}
    # the same checkpoint through the DROP-IN route: qwen3-mlx's forward with every Linear a quantized_matmul on the (weight, scales, biases)
    # triplet, the embedding dequantising its rows -- what the unmodified crate calls on the reference's flagship format
    try:
        m.per_op_route(prompt, 1)                 # (first call: the route's buffers and the packed matrices' derived forms are built here)
        r = m.per_op_route(prompt, 64)
        out["per_op_route"] = {"metric": f"decode_tokens_per_sec_per_op_route_{bits}bit", "value": round(1e3 / r["ms_per_token"], 2), "unit": "tokens/s",
                               "ms_per_token_host_wall": round(r["ms_per_token"], 3), "mlx_calls_per_token": round(r["calls_per_token"], 1),
                               "tokens_timed": 64, "vs_engine": round((1e3 / r["ms_per_token"]) / (args.steps / elapsed), 3),
                               "first_tokens": [int(t) for t in r["tokens"][:5]],
                               "note": "csrc/per_op_route.hip on this model's packed weights through the mlx-c ABI (mlx_quantized_matmul, mlx_dequantize, "
                                       "mlx_take_axis ...), the deferred list rewriting the calls onto the packed-GEMV family (csrc/mlxc_lazy.hpp)"}
    except Exception as e:   # a report, never a reason to lose the measured line
        out["per_op_route"] = {"value": None, "error": str(e)[:300]}
    m.close()
    return out


def mixtral_secondary(omx, steps=64, warm=8, n_prompt=2048, rank=0, world=1, comm=None, dist=None, mode="ep"):
    """BASELINE config 3: Mixtral-8x7B shapes (mixtral-mlx/src/model.rs:44-52) in bf16, sparse-MoE decode engine: router + top-2
    expert GEMVs per layer inside the step graph.  One GPU: all 93 GB of weights on it.  comm != None: expert parallel over `world`
    ranks -- 8 / world experts per rank, attention and router replicated, one f32 all-reduce per MoE block (the prompt as ONE batched
    pass with a [T, hidden] all-reduce per layer); every rank calls this, the time is the MAX over ranks.
    mode "etp": expert TENSOR parallel instead -- attention heads and every expert's intermediate columns sharded over the ranks
    (batch-1 decode under expert parallelism streams a token's two experts from at most two ranks; here all ranks stream 1 / world
    of them); the prompt is one batched pass there too (all experts at this rank's columns, the same f32 [T, hidden] all-reduce)."""
    import numpy as np
    from ominix_mlx_amd import engine
    cfg = dict(hidden_size=4096, num_hidden_layers=32, intermediate_size=14336, num_attention_heads=32, num_key_value_heads=8,
               head_dim=128, vocab_size=32000, rms_norm_eps=1e-5, rope_theta=1e6, num_experts=8, num_experts_per_tok=2,
               moe_intermediate_size=14336, moe_mode="mixtral", qk_norm=False)
    if comm is not None and mode == "etp":
        m = engine.Model(max_context=n_prompt + warm + steps + 8, tp_rank=rank, tp_size=world, **cfg)
        m.set_comm(comm[1], comm[2])
    elif comm is not None:
        m = engine.Model(max_context=n_prompt + warm + steps + 8, ep_rank=rank, ep_size=world, **cfg)
        m.set_comm(comm[1], comm[2])
    else:
        m = engine.Model(max_context=n_prompt + warm + steps + 8, **cfg)
    m.synth_weights()

    def barrier():
        if dist is not None:
            import torch
            dist.barrier()
            torch.cuda.synchronize()
        omx.check(omx.lib.omx_synchronize(m.stream()))

    barrier()
    first = m.prefill(prompt_ids(n_prompt, cfg["vocab_size"]))
    m.decode(warm)
    barrier()
    t0 = time.perf_counter()
    toks = m.decode(steps)
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        import torch
        t = torch.tensor([dt], dtype=torch.float64, device=REDUCE_DEVICE)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    step_bytes = m.step_bytes(n_prompt + warm + steps // 2)
    out = {"metric": "decode_tokens_per_sec_mixtral_8x7b_bf16", "value": round(steps / dt, 2), "unit": "tokens/s", "n_gpus": world,
           "steps": steps, "ms_per_step": round(dt / steps * 1e3, 4), "dtype": "bf16", "data": "synthetic", "parallelism": f"{'etp' if mode == 'etp' else 'ep'}{world}", "prompt": n_prompt,
           "decode_path": m.decode_path(),
           "step_roofline": {"algorithmic_bytes_per_token": int(step_bytes), "achieved_GBps": round(step_bytes / (dt / steps) / 1e9, 1),
                             "frac_of_hbm_peak": round(step_bytes / (dt / steps) / 1e9 / HBM_PEAK_GBPS, 4)},
           "prefill_device_ms": round(m.last_prefill_ms(), 2), "first_tokens": [int(first)] + [int(t) for t in toks[:4]]}
    m.close()
    return out


def paraformer_secondary(omx, reps=5):
    """BASELINE.json configs[3]: Paraformer-large on 30 s of 16 kHz audio (mel/STFT + LFR + CMVN -> 50-layer SAN-M encoder
    -> CIF -> 16-layer decoder -> token ids), one MI355X, synthetic checkpoint with the reference's keys; audio resident in
    HBM when the timed region starts.  The reference publishes 400 ms on an M3 Max."""
    import numpy as np
    from ominix_mlx_amd import audio, paraformer
    m = paraformer.Paraformer(paraformer.random_checkpoint())
    sr, secs = 16000, 30
    t = np.arange(sr * secs) / sr
    wave = (0.3 * np.sin(2 * np.pi * 220 * t) * (1 + 0.5 * np.sin(2 * np.pi * 3 * t)) + 0.03 * np.random.default_rng(0).standard_normal(t.size)).astype(np.float32)
    fe = audio.MelFrontend()
    wave_d = omx.ops.Tensor.from_numpy(wave, "f32")

    def run():
        t0 = time.perf_counter()
        mel = fe.forward(wave_d)
        tok, n = m.transcribe_from_mel(mel.view(mel.shape[1:]))
        return time.perf_counter() - t0, n

    run()
    best, n = min(run() for _ in range(reps))
    flop = 0.17e12   # the whole pass: 50 SAN-M encoder layers over 501 frames + 16 decoder layers over ~216 tokens
    return {"metric": "paraformer_30s_audio_seconds", "value": round(best, 5), "unit": "s", "higher_is_better": False, "n_gpus": 1,
            "rtf": round(best / secs, 6), "tokens": int(n), "dtype": "f32", "data": "synthetic",
            "vs_reference_m3max_400ms": round(0.4 / best, 1),
            # ~650 launches of <= 16 us over 200-500-row GEMMs: latency-bound, the float32 matrix-core peak is not what limits it
            "roofline": {"bound": "mfma", "achieved": round(flop / best / 1e12, 2), "peak": 157.0, "unit": "TFLOP/s (f32 matrix cores)",
                         "frac": round(flop / best / 1e12 / 157.0, 4), "traffic": None}}


def physical_cores():
    """Physical cores this process may run on (unique (package, core) pairs of /proc/cpuinfo within the affinity mask)."""
    allowed = os.sched_getaffinity(0)
    cores, cpu, pkg = set(), None, 0
    try:
        for line in open("/proc/cpuinfo"):
            k, _, v = line.partition(":")
            k = k.strip()
            if k == "processor":
                cpu = int(v)
            elif k == "physical id":
                pkg = int(v)
            elif k == "core id" and cpu in allowed:
                cores.add((pkg, int(v)))
    except OSError:
        pass
    n = max(1, len(cores) or len(allowed))
    # a container's CPU bandwidth quota (cgroup v2 cpu.max / v1 cfs_quota_us): more runnable threads than that are throttled, not run
    # (the GPU boxes of this pool: 256 visible CPUs, quota 16 -- 128 pinned threads gave a bimodal 0.6 .. 12 tok/s)
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = min(n, max(1, int(q) // int(per)))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // per))
        except (OSError, ValueError):
            pass
    return n


def cpu_baseline(cfg, ctx):
    """The CPU leg runs in a CHILD process (`bench.py --cpu-baseline-only`) that never touches the GPU: OpenMP reads its thread
    count and binding when the runtime is loaded, so a clean process with OMP_NUM_THREADS = physical cores, OMP_PLACES=cores,
    OMP_PROC_BIND=close is the only way to pin them; the child first-touches its buffers with the same static schedule the GEMVs use."""
    threads = physical_cores()
    env = dict(os.environ, OMP_NUM_THREADS=str(threads), OMP_PLACES="cores", OMP_PROC_BIND="close", OMP_DYNAMIC="false")
    cmd = [sys.executable, os.path.abspath(__file__), "--cpu-baseline-only", "--cpu-ctx", str(ctx), "--cpu-cfg", json.dumps(cfg)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    if r.returncode != 0:
        raise RuntimeError(f"cpu baseline child rc={r.returncode}: {r.stderr[-300:]}")
    return json.loads(r.stdout.strip().splitlines()[-1])


def cpu_baseline_child(cfg, ctx, n_layers_sample=4, warm=2, reps=7):
    """Plain-C port (oracle/c/omx_oracle.c, OpenMP) of the same decode step, timed on a bounded sample: `n_layers_sample` of the
    model's layers at context `ctx` plus the lm_head, `warm` untimed passes (page first-touch, caches), then the MEDIAN of `reps`
    timed passes, scaled to the full layer count."""
    import numpy as np
    from oracle import c_oracle
    lib = c_oracle.load()
    hd, I, H, Hkv, D, V = (cfg["hidden_size"], cfg["intermediate_size"], cfg["num_attention_heads"],
                           cfg["num_key_value_heads"], cfg["head_dim"], cfg["vocab_size"])
    cap = ctx + 8
    rngs = iter(range(1, 10000))

    def filled(n, amp=0.0346, off=0.0):
        a = np.empty(n, np.uint16)
        lib.oracle_fill_uniform_bf16(c_oracle.ptr(a), n, next(rngs), np.float32(amp), np.float32(off))
        return a

    layers = []
    for _ in range(n_layers_sample):
        arrs = [filled(H * D * hd), filled(Hkv * D * hd), filled(Hkv * D * hd), filled(hd * H * D), filled(I * hd),
                filled(I * hd), filled(hd * I), filled(D, 0.017, 1.0), filled(D, 0.017, 1.0), filled(hd, 0.017, 1.0),
                filled(hd, 0.017, 1.0), filled(Hkv * cap * D, 1.0), filled(Hkv * cap * D, 1.0)]
        layers.append((arrs, c_oracle.Layer(*[c_oracle.ptr(a) for a in arrs])))
    lc = c_oracle.LayerCfg(hd, I, H, Hkv, D, cap, cfg["rms_norm_eps"], cfg["rope_theta"], 1.0)
    scratch = np.zeros(lib.oracle_qwen3_scratch_elems(ctypes.byref(lc)), np.uint16)
    h = filled(hd, 1.0)
    head, norm_w, logits = filled(V * hd), filled(hd, 0.017, 1.0), np.empty(V, np.uint16)
    samples = []
    for it in range(warm + reps):
        t0 = time.perf_counter()
        for _, L in layers:
            lib.oracle_qwen3_layer_decode(ctypes.byref(lc), ctypes.byref(L), c_oracle.ptr(h), ctx, c_oracle.ptr(scratch))
        t1 = time.perf_counter()
        lib.oracle_qwen3_head(c_oracle.ptr(h), c_oracle.ptr(norm_w), c_oracle.ptr(head), hd, V, cfg["rms_norm_eps"],
                              c_oracle.ptr(logits), c_oracle.ptr(scratch))
        t2 = time.perf_counter()
        if it >= warm:
            samples.append((t1 - t0) / n_layers_sample * cfg["num_hidden_layers"] + (t2 - t1))
    samples.sort()
    step_s = samples[len(samples) // 2]
    threads = int(os.environ.get("OMP_NUM_THREADS", "0")) or os.cpu_count()
    print(json.dumps({"value": round(1.0 / step_s, 4), "unit": "tokens/s", "cores": threads, "kind": "port",
                      "spread": [round(1.0 / samples[-1], 4), round(1.0 / samples[0], 4)],
                      "sample": f"{n_layers_sample} of {cfg['num_hidden_layers']} decoder layers at ctx {ctx} + lm_head, {warm} warm passes then the "
                                f"median of {reps}, scaled to the full model; C port of the oracle, OpenMP, {threads} threads pinned one per "
                                f"physical core (OMP_PLACES=cores OMP_PROC_BIND=close; thread count = physical cores capped by the container's CPU quota), in a child process"}), flush=True)


def main():
    args = parse()
    if args.cpu_baseline_only:
        return cpu_baseline_child(json.loads(args.cpu_cfg), args.cpu_ctx)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))
    if args.dry_run:
        return dry_run(args)
    rank, world, local, dist = init_dist(args.gpus)
    import torch
    import omx_import
    omx = omx_import.load_package()
    from ominix_mlx_amd import engine

    cfg = dict(MODELS[args.model])
    if args.layers:
        cfg["num_hidden_layers"] = args.layers
    max_ctx = args.prompt + args.warmup + N_WINDOWS * args.steps + 16
    moe = cfg.get("num_experts", 0) > 0
    if moe:   # BASELINE config 3: experts sharded over the ranks (expert parallel), attention replicated, one all-reduce per layer
        model = engine.Model(max_context=max_ctx, ep_rank=rank, ep_size=world, **cfg)
    else:
        model = engine.Model(max_context=max_ctx, tp_rank=rank, tp_size=world, **cfg)
    keep = peer = None
    peer_note = None
    if world > 1 or os.environ.get("OMX_BENCH_FORCE_COMM") == "1":   # (the flag: run the N > 1 code path -- RCCL all-reduces in the
        keep = None if ONE_GPU else rccl_comm(dist, rank, world)    #  step graph, batched TP prefill -- on a one-rank communicator)
        peer, peer_note = peer_comm(dist, rank, world, keep)
        if peer is not None:
            model.set_comm(peer.comm, peer.fn)
        elif keep is None:
            raise SystemExit(f"rank {rank}: no RCCL communicator and the peer communicator was not usable either ({peer_note})")
        else:
            model.set_comm(keep[1], keep[2])
    model.synth_weights()

    prompt = prompt_ids(args.prompt, cfg["vocab_size"])
    if dist is not None:
        dist.barrier()      # ranks enter their first collective together (the peer all-reduce's waits are bounded)
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    first = model.prefill(prompt)
    torch.cuda.synchronize()
    prefill_s = time.perf_counter() - t0
    prefill_first_ms = model.last_prefill_ms()
    # steady state: the same prompt once more on the emptied cache (the first call pays the one-time scratch allocation)
    model.reset()
    if dist is not None:
        dist.barrier()
    first2 = model.prefill(prompt)
    prefill_steady_ms = model.last_prefill_ms()
    if dist is not None and peer is not None and peer.aborted():
        raise SystemExit(f"rank {rank}: the peer communicator gave up waiting for a rank during the prompt passes (bounded wait expired: its "
                         f"result was voided) -- first {int(first)}, second {int(first2)}")
    if int(first2) != int(first):
        raise SystemExit(f"rank {rank}: the second prefill of the same prompt sampled {int(first2)}, the first {int(first)}")
    warm_toks = model.decode(args.warmup) if args.warmup else []

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        omx.check(omx.lib.omx_synchronize(model.stream()))

    barrier()
    t0 = time.perf_counter()
    toks = model.decode(args.steps)
    t_dec = time.perf_counter() - t0
    barrier()
    elapsed = time.perf_counter() - t0
    if os.environ.get("OMX_BENCH_DEBUG") == "1":
        print(f"[bench rank {rank}] window 1: decode call {t_dec * 1e3:.1f} ms, with the closing barrier {elapsed * 1e3:.1f} ms, device {model.last_decode_ms():.1f} ms, path {model.decode_path()}, offset {model.offset()}", file=sys.stderr, flush=True)
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=REDUCE_DEVICE)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    dev_ms = model.last_decode_ms()
    # the reference's protocol is 3 runs, mean +- sigma (mistral-mlx/examples/benchmark_mistral.rs:62-105): `value` stays the first
    # window of exactly K steps; two more windows of K steps each (the context keeps growing) give the spread
    window_tok_s = [args.steps / elapsed]
    for _ in range(N_WINDOWS - 1):
        barrier()
        t0 = time.perf_counter()
        model.decode(args.steps)
        barrier()
        w = time.perf_counter() - t0
        if os.environ.get("OMX_BENCH_DEBUG") == "1":
            print(f"[bench rank {rank}] next window: {w * 1e3:.1f} ms, device {model.last_decode_ms():.1f} ms, path {model.decode_path()}, offset {model.offset()}", file=sys.stderr, flush=True)
        if dist is not None:
            t = torch.tensor([w], dtype=torch.float64, device=REDUCE_DEVICE)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            w = float(t.item())
        window_tok_s.append(args.steps / w)
    if peer is not None and peer.aborted():   # a wait inside the peer all-reduce gave up: the tokens of this run are void
        raise SystemExit(f"rank {rank}: the peer-store all-reduce gave up waiting for a peer during the run; no valid measurement")

    # collective secondaries (collective_plan): every rank takes part, rank 0 reports.  The primary model is closed first: Mixtral's
    # shard (93 GB / world + replicated attention) must fit next to nothing else of this size
    plan = collective_plan(args, world)
    flux_tp = mixtral_ep = None
    primary_closed = False
    if plan:
        model_stats = {"step_bytes": model.step_bytes(args.prompt + args.warmup + args.steps // 2), "prefill_ms": model.last_prefill_ms()}
        model.close()
        primary_closed = True
    # rank 0 assembles the primary line BEFORE the collective secondaries: they have never run on real xGMI links in the build loop, and
    # a rank stuck inside a collective cannot be recovered in-process -- the watchdog below then still prints the measured line
    ctx_mid = args.prompt + args.warmup + args.steps // 2

    def assemble():
        step_bytes = (model_stats["step_bytes"] if primary_closed else model.step_bytes(ctx_mid)) * (1 if moe else world)   # whole-job algorithmic bytes per token (EP ranks share one token's experts)
        ms_per_step = elapsed * 1e3 / args.steps
        tok_s = args.steps / elapsed
        k_bytes, iso_s = time_dominant_kernel(omx, cfg, 1 if moe else world)
        in_step = None
        if world == 1 and not moe and keep is None:
            # the figure the roofline object reports: HIP events on the step's stream around every launch of 4 further (eager) decode steps
            try:
                in_step = model.time_step_kernels(4)
            except Exception as e:      # a measurement hook must never cost the measured line: fall back to the isolated launches
                print(f"in-step kernel timing failed ({e}); roofline.achieved from isolated launches", file=sys.stderr)
                in_step = None
        # the DROP-IN route (what an unmodified qwen3-mlx gets: no fused engine, ~1 400 eager mlx_* calls per token through the handle ABI,
        # csrc/per_op_route.hip replays Model::forward + Generate::next natively on this model's weights) next to the engine's number
        per_op = None
        if world == 1 and not moe and keep is None and not cfg.get("quantization"):
            try:
                n_po = 64
                r0 = model.per_op_route(prompt, 1)        # (first call: the handle route's buffers are allocated here -- 50 ms .. 0.5 s by box)
                from ominix_mlx_amd import mlx_c
                ls0 = mlx_c.lazy_stats()
                r = model.per_op_route(prompt, n_po)
                ls1 = mlx_c.lazy_stats()
                n_pass = n_po + 1                         # decode passes of the call (Generate::next keeps one in flight) beside the one prompt pass
                eng = [int(first)] + [int(t) for t in warm_toks[:n_po]]
                got = [int(t) for t in r["tokens"]]
                n_cmp = min(len(eng), len(got))
                agree = next((i for i in range(n_cmp) if eng[i] != got[i]), n_cmp)
                per_op = {"metric": "decode_tokens_per_sec_per_op_route", "value": round(1e3 / r["ms_per_token"], 2), "unit": "tokens/s",
                          "ms_per_token_host_wall": round(r["ms_per_token"], 3), "mlx_calls_per_token": round(r["calls_per_token"], 1),
                          "host_us_per_mlx_call": round(r["ms_per_token"] * 1e3 / r["calls_per_token"], 2),
                          "prefill_ms_host_wall": round(r["prefill_ms"], 1), "prefill_ms_host_wall_first_call": round(r0["prefill_ms"], 1),
                          "tokens_timed": n_po, "context": args.prompt,
                          "first_tokens": got[:5], "leading_tokens_equal_to_engine": f"{agree} of {n_cmp}",
                          "vs_engine": round((1e3 / r["ms_per_token"]) / tok_s, 3),
                          # the deferred list behind the ABI (csrc/mlxc_lazy.hpp): what the ~1 380 calls of a token became
                          "deferred": {"ops_recorded_per_token": round((ls1["recorded"] - ls0["recorded"]) / (n_pass + 1), 1),
                                       "launches_per_token": round((ls1["launched_as_recorded"] - ls0["launched_as_recorded"] +
                                                                    ls1["fused_launches"] - ls0["fused_launches"]) / (n_pass + 1), 1),
                                       "of_which_fused_gemv_family": round((ls1["fused_launches"] - ls0["fused_launches"]) / (n_pass + 1), 1),
                                       "host_ms_per_token_in_flushes": round((ls1["flush_host_ns"] - ls0["flush_host_ns"]) / 1e6 / (n_pass + 1), 3),
                                       "note": "averages over the call's passes INCLUDING its one 2 048-token prompt pass"},
                          "note": "qwen3-mlx Model::forward + Generate::next replayed call for call through the mlx-c handle ABI by native code on the "
                                  "engine's own weights (csrc/per_op_route.hip): the route an UNMODIFIED crate takes.  Round 6: the ABI records the calls and "
                                  "executes them at mlx_async_eval / item, rewriting the decode idioms onto the engine's GEMV family (csrc/mlxc_lazy.hpp; "
                                  "OMX_MLX_LAZY=0 is round 5's eager execution); `value` above is the omx_qwen3_* engine"}
            except Exception as e:   # a report, never a reason to lose the measured line
                per_op = {"metric": "decode_tokens_per_sec_per_op_route", "value": None, "error": str(e)[:300]}
        # what this box streams in ONE dependency-free launch of the step's bytes (the plain GEMV kernel over a single [N, 4096] matrix as
        # large as a decode step's traffic): the ceiling the step's fraction of the 8 TB/s spec is to be read against (VERDICT r5 "Next" 2)
        ceiling = None
        if world == 1 and not moe and keep is None and not args.layers:
            try:
                import ctypes
                fn = omx.lib.omx_bench_gemv
                fn.restype = ctypes.c_int
                fn.argtypes = [ctypes.c_int] * 7 + [ctypes.POINTER(ctypes.c_float)]
                n_rows = int(model.step_bytes(ctx_mid) // (2 * 4096)) // 512 * 512
                ms = ctypes.c_float(0)
                omx.check(fn(n_rows, 4096, 0, 0, 0, 1, 5, ctypes.byref(ms)))
                ceiling = n_rows * 4096 * 2 / (ms.value * 1e-3) / 1e9
            except Exception as e:      # a report, never a reason to lose the measured line
                print(f"streaming ceiling not measured ({e})", file=sys.stderr)
        k_s = in_step["gate_up"] * 1e-6 if in_step else iso_s
        achieved = k_bytes / k_s / 1e9
        H, Hkv, D, hd, I, V = (cfg["num_attention_heads"], cfg["num_key_value_heads"], cfg["head_dim"], cfg["hidden_size"],
                               cfg["intermediate_size"], cfg["vocab_size"])
        class_bytes = {"qkv": 2 * (H + 2 * Hkv) * D * hd, "attention": 2 * 2 * Hkv * D * (ctx_mid + args.steps // 2), "o": 2 * H * D * hd,
                       "gate_up": 4 * I * hd, "down": 2 * I * hd, "lm_head": 2 * V * hd}
        first_ok = None
        want_first = FIRST_TOKEN.get((args.model, args.prompt)) if world == 1 and not args.layers else None
        if want_first is not None:
            first_ok = int(first) == want_first
        out = {
            "metric": "decode_tokens_per_sec", "value": round(tok_s, 2), "unit": "tokens/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": f"{args.model}" + (" (Qwen3-8B shapes for BASELINE 'Qwen3-7B')" if args.model == "qwen3-8b" else "") + " bf16 greedy decode, batch 1, "
                                   f"{args.prompt}-token prompt then {args.warmup}+{args.steps} decode tokens",
                       "parallelism": (f"ep{world}" if moe else f"tp{world}"), "context_at_timing": ctx_mid,
                       "layers": cfg["num_hidden_layers"], **({"allreduce": peer_note} if peer_note else {}),
                       # reductions issued through the peer communicator so far, by path (one-shot <= 32 KB, two-shot chunks, MoE exchange,
                       # handed to RCCL; calls inside a captured graph count once) -- what the first multi-GPU run's log is read from
                       **({"allreduce_launches": peer.counts()} if peer is not None else {})},
            "roofline": {"bound": "hbm", "kernel": "gemv_kernel<rmsnorm, gate/up, swiglu>", "achieved": round(achieved, 1),
                         "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 4),
                         "traffic": pmc_traffic(DOMINANT_KERNEL) if world == 1 and args.model == "qwen3-8b" else None,
                         "traffic_source": "profiles/r06_pmc_fetch_size.json <- profiles/r06_pmc_fetch_size_step.csv (rocprofv3 --pmc FETCH_SIZE over real decode steps, "
                                           "own pass, x2 gfx950 correction; tools/pmc_profile.sh, tools/pmc_report.py)",
                         "algorithmic_bytes_per_launch": k_bytes, "avg_launch_us": round(k_s * 1e6, 2),
                         "timing": ("in-step: every one of the 36 launches per step of 4 eager decode steps, run right after the timed region, carries its own "
                                    "HIP start/stop event pair (hipExtLaunchKernelGGL: the dispatch's begin/end timestamps on the step's stream)"
                                    if in_step else "isolated launches over rotating weight buffers (omx_bench_gemv), HIP events"),
                         "isolated_launch_us": round(iso_s * 1e6, 2)},
            "step_roofline": {"algorithmic_bytes_per_token": int(step_bytes),
                              "achieved_GBps": round(step_bytes / (ms_per_step * 1e-3) / 1e9, 1),
                              "frac_of_hbm_peak": round(step_bytes / (ms_per_step * 1e-3) / 1e9 / (HBM_PEAK_GBPS * world), 4),
                              "roofline_tokens_per_sec": round(HBM_PEAK_GBPS * 1e9 * world / step_bytes, 1),
                              "device_ms_per_step_hip_events": round(dev_ms / args.steps, 4),
                              **({"ceiling_GBps": round(ceiling, 1), "frac_of_ceiling": round(step_bytes / (ms_per_step * 1e-3) / 1e9 / ceiling, 4),
                                  "ceiling_tokens_per_sec": round(ceiling * 1e9 / step_bytes, 1),
                                  "ceiling_note": "one launch of the plain bf16 GEMV kernel streaming the step's algorithmic bytes as a single matrix, "
                                                  "HIP events, this box, this run: no dependency, no launch boundary inside"} if ceiling else {})},
            "prefill": {"tokens": args.prompt, "seconds": round(prefill_s, 4), "device_ms": round(prefill_first_ms, 3),
                        "device_ms_steady": round(prefill_steady_ms, 3),
                        "tokens_per_sec": round(args.prompt / max(prefill_steady_ms, 1e-6) * 1e3, 1),
                        "mode": "batched on the tensor-parallel shards: MFMA GEMMs + flash attention, two bf16 all-reduces of [T, hidden] per layer, the last token through the decode step" if world > 1 else
                                "batched: MFMA GEMMs + flash attention over all n tokens (first call: includes the one-time scratch allocation), norm + lm_head + sampler on the last row"},
            "windows": {"n": len(window_tok_s), "steps_each": args.steps, "tokens_per_sec": [round(v, 2) for v in window_tok_s],
                        "mean": round(sum(window_tok_s) / len(window_tok_s), 2),
                        "std": round((sum((v - sum(window_tok_s) / len(window_tok_s)) ** 2 for v in window_tok_s) / len(window_tok_s)) ** 0.5, 2),
                        "protocol": "mistral-mlx/examples/benchmark_mistral.rs:62-105 (3 runs, mean +- sigma); `value` = window 1"},
            "first_tokens": [int(first)] + [int(t) for t in toks[:4]],
            "first_token_check": {"expected": want_first, "ok": first_ok},
        }
        if per_op is not None:
            out["per_op_route"] = per_op
        if in_step:
            if in_step.get("step_engine", 0.0) == 0.0:   # (the persistent step's launch: only with OMX_STEP_ENGINE set)
                in_step.pop("step_engine", None)
            else:
                class_bytes["step_engine"] = step_bytes - class_bytes["lm_head"]
                in_step = {k: v for k, v in in_step.items() if v > 0.0}
            if in_step.get("o", 1.0) == 0.0:      # the O projection rides in the attention launch (csrc/attn_step.hip): one figure, both byte counts
                in_step.pop("o")
                class_bytes["attention"] += class_bytes["o"]
                in_step = {("attention+o" if k == "attention" else k): v for k, v in in_step.items()}
                class_bytes["attention+o"] = class_bytes["attention"]
            out["step_kernels"] = {k: {"avg_us": round(v, 2), "algorithmic_bytes": int(class_bytes[k]),
                                       "frac_of_hbm_peak": round(class_bytes[k] / (v * 1e-6) / 1e9 / HBM_PEAK_GBPS, 4)} for k, v in in_step.items()}
        return out, first_ok, want_first

    out, first_ok, want_first = assemble() if rank == 0 else (None, None, None)
    if plan:
        import threading

        out_lock = threading.Lock()

        def give_up():
            if rank == 0:
                with out_lock:                      # (the main thread adds the finished secondaries under the same lock)
                    line = dict(out)
                note = {"value": None, "error": f"collective secondary did not finish within {SECONDARY_TIMEOUT_S} s (watchdog)"}
                line.setdefault("secondary", dict(note, metric="flux_klein_1024_sec_per_step"))
                if "mixtral" in plan:
                    line.setdefault("mixtral", dict(note, metric="decode_tokens_per_sec_mixtral_8x7b_bf16"))
                print(json.dumps(line), flush=True)
            # status 3 = "primary line valid, a collective secondary hung": ranks stuck inside GPU collectives must not look like a
            # clean run (every rank's own watchdog fires: the ranks leave by themselves, nobody waits for a stuck peer)
            os._exit(3)

        # the secondaries' communicator: the peer one when it is up (one hop for the Mixtral step's 16 KB reductions; anything larger it
        # hands to the RCCL communicator behind it -- or, with no RCCL at all, reduces itself in two shots), else RCCL
        sec_comm = (None, peer.comm, peer.fn) if peer is not None else keep
        dog = threading.Timer(SECONDARY_TIMEOUT_S, give_up)
        dog.daemon = True
        dog.start()
        try:
            flux_tp = flux_secondary(omx, rank=rank, world=world, comm=sec_comm)
        except Exception as e:
            flux_tp = {"metric": "flux_klein_1024_sec_per_step", "value": None, "error": str(e)}
        if rank == 0 and flux_tp is not None:
            with out_lock:
                out["secondary"] = flux_tp      # (kept if the watchdog fires during the next workload)
        if "mixtral" in plan:
            try:
                mixtral_ep = mixtral_secondary(omx, rank=rank, world=world, comm=sec_comm, dist=dist)
            except Exception as e:
                mixtral_ep = {"metric": "decode_tokens_per_sec_mixtral_8x7b_bf16", "value": None, "error": str(e)}
            if world > 1 and 14336 % (64 * world) == 0:    # the same model with expert TENSOR parallelism (what scales batch-1 decode)
                try:
                    mixtral_ep["expert_tensor_parallel"] = mixtral_secondary(omx, rank=rank, world=world, comm=sec_comm, dist=dist, mode="etp")
                except Exception as e:
                    mixtral_ep["expert_tensor_parallel"] = {"value": None, "error": str(e)}
        dog.cancel()
    if rank != 0:
        if not primary_closed:
            model.close()
        return
    if not primary_closed:
        model.close()
    if flux_tp is not None:
        out["secondary"] = flux_tp
    if mixtral_ep is not None:
        out["mixtral"] = mixtral_ep
    if world == 1 and not args.no_flux and not moe and not plan:
        try:
            out["secondary"] = flux_secondary(omx)
        except Exception as e:
            out["secondary"] = {"metric": "flux_klein_1024_sec_per_step", "value": None, "error": str(e)}
    if world == 1 and not args.no_flux and not moe:
        try:
            out["quantized"] = quantized_secondary(omx, cfg, args)
        except Exception as e:
            out["quantized"] = {"metric": "decode_tokens_per_sec_4bit", "value": None, "error": str(e)}
        try:
            if "mixtral" not in out:
                out["mixtral"] = mixtral_secondary(omx)
        except Exception as e:
            out["mixtral"] = {"metric": "decode_tokens_per_sec_mixtral_8x7b_bf16", "value": None, "error": str(e)}
        try:
            out["paraformer"] = paraformer_secondary(omx)
        except Exception as e:
            out["paraformer"] = {"metric": "paraformer_30s_audio_seconds", "value": None, "error": str(e)}
    if not args.no_cpu_baseline and world == 1:      # the CPU leg is timed at N = 1 only (one baseline per box, not per rank count)
        try:
            out["cpu_baseline"] = cpu_baseline(cfg, ctx_mid)
        except Exception as e:   # the baseline is a report, never a reason to lose the measured line
            out["cpu_baseline"] = {"value": None, "unit": "tokens/s", "cores": os.cpu_count(), "kind": "port",
                                   "sample": f"failed: {e}"}
    print(json.dumps(out), flush=True)
    if first_ok is False:
        raise SystemExit(f"first token {int(first)} != committed {want_first}: the engine computed something else; the line above is not a valid measurement")


if __name__ == "__main__":
    main()
