"""The driver's N > 1 command -- `python -m torch.distributed.run --nproc-per-node 2 bench.py --gpus 2` with no other workload flag --
run for real on the test box's ONE GPU (bench.py OMX_BENCH_ONE_GPU=1): two PROCESSES, gloo for the host-side plumbing, every device
all-reduce on the peer communicator (csrc/peer_allreduce.hip: HIP IPC inboxes and stages; one-shot below 32 KB, two-shot above).
What the CPU launcher test (gloo dry run, no engine) and the loopback tests (threads, no processes) leave open is closed here: the
bench's rank control flow with real engines -- barriers, MAX over ranks, watchdog threads, the peer self-test, graph-captured steps
with the reduction kernels inside, batched TP prefill, then all three collective secondaries (FLUX TP 2, Mixtral EP 2 and expert-TP 2).
RCCL itself cannot run with two ranks on one device and stays untested here."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_flagless_two_rank_bench_on_one_gpu(omx, tmp_path):
    port = 29700 + (os.getpid() % 200)
    env = dict(os.environ, OMX_BENCH_ONE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "16", "--warmup", "4"],
                       env=env, capture_output=True, text=True, timeout=1500, stdin=subprocess.DEVNULL, cwd=str(tmp_path))
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines                      # ONE JSON line on stdout, from rank 0
    out = json.loads(lines[0])
    assert out["metric"] == "decode_tokens_per_sec" and out["n_gpus"] == 2 and out["steps"] == 16 and out["value"] > 0
    assert out["config"]["parallelism"] == "tp2" and "ONE GPU" in out["config"]["allreduce"]
    assert out["prefill"]["device_ms_steady"] > 0 and "batched" in out["prefill"]["mode"]
    assert len(out["windows"]["tokens_per_sec"]) == 3
    flux, mix = out["secondary"], out["mixtral"]
    assert flux["parallelism"] == "tp2" and flux["value"] and flux["value"] > 0, flux
    assert mix["parallelism"] == "ep2" and mix["value"] and mix["value"] > 0 and mix["decode_path"] == "graph", mix
    etp = mix["expert_tensor_parallel"]
    assert etp["parallelism"] == "etp2" and etp["value"] and etp["value"] > 0 and etp["decode_path"] == "graph", etp
    print(f"two ranks on one GPU: Qwen3-8B TP 2 {out['value']:.1f} tok/s (windows {out['windows']['tokens_per_sec']}), FLUX TP 2 "
          f"{flux['value'] * 1e3:.1f} ms / step, Mixtral EP 2 {mix['value']:.1f} tok/s, expert-TP 2 {etp['value']:.1f} tok/s -- shared GPU, not a measurement")
