"""CPU: `python bench.py --gpus N` launches its own ranks (VERDICT r1 "Next" #2).  The dry run keeps the distributed skeleton of
the real bench -- child torch.distributed.run, 127.0.0.1 rendezvous, barriers around the timed region, MAX over ranks, one JSON
line from rank 0 -- and swaps the GPU engine for a stand-in step, so the launcher is testable without a GPU."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*flags):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *flags], capture_output=True, text=True, timeout=240, env=env)
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    return p.returncode, lines, p.stderr


def test_bench_self_launches_two_ranks():
    rc, lines, err = _run("--gpus", "2", "--steps", "4", "--warmup", "1", "--dry-run")
    assert rc == 0, err[-2000:]
    assert len(lines) == 1, lines                     # exactly ONE JSON line, from rank 0
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 4 and out["warmup"] == 1 and out["dry_run"] is True
    assert out["config"]["parallelism"] == "tp2" and out["ms_per_step"] > 0


def test_bench_dry_run_single_rank_and_world_mismatch():
    rc, lines, _ = _run("--steps", "2", "--dry-run")
    assert rc == 0 and json.loads(lines[0])["n_gpus"] == 1
    # under a launcher whose world size disagrees with --gpus the bench refuses (no silent single-rank number)
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "1"], capture_output=True, text=True,
                       timeout=240, env=env)
    assert p.returncode != 0 and "WORLD_SIZE=1" in (p.stderr + p.stdout)


def test_flagless_multi_gpu_command_covers_all_three_workloads():
    """VERDICT r2 "Next" #4: the driver runs `bench.py --gpus N` with no other flags; at N > 1 that one command must also measure
    BASELINE config 5 (FLUX tensor parallel, object "secondary") and config 3 (Mixtral expert parallel, object "mixtral") -- the
    dry run reports the plan the real run executes (bench.collective_plan)."""
    rc, lines, err = _run("--gpus", "2", "--steps", "2", "--warmup", "1", "--dry-run")
    assert rc == 0, err[-2000:]
    out = json.loads(lines[0])
    assert out["secondary"]["parallelism"] == "tp2" and out["mixtral"]["parallelism"] == "ep2"
    # one rank with a communicator forced: the N > 1 code path of all three, on one GPU
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["OMX_BENCH_FORCE_COMM"] = "1"
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--dry-run"], capture_output=True, text=True,
                       timeout=240, env=env)
    out = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][0])
    assert out["secondary"]["parallelism"] == "tp1" and out["mixtral"]["parallelism"] == "ep1"
    # plain single-GPU run: no collective objects in the plan (the single-device secondaries are measured instead)
    rc, lines, _ = _run("--steps", "2", "--dry-run")
    out = json.loads(lines[0])
    assert "secondary" not in out and "mixtral" not in out
    # 3 ranks do not divide 8 experts: FLUX only
    import bench
    class A: no_flux = False; model = "qwen3-8b"
    assert bench.collective_plan(A, 3) == {"secondary": "tp3"} and bench.collective_plan(A, 8) == {"secondary": "tp8", "mixtral": "ep8"}
