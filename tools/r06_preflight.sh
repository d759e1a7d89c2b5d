#!/bin/bash
# Round-6 pre-flight of the driver's N > 1 command: N processes on the ONE GPU of a test box (OMX_BENCH_ONE_GPU=1): NOT a measurement -- evidence that
# the TP / EP / expert-TP control flow of N = 2, 4, 8 runs to the end with the round's code (peer communicator scopes, self-tests, new kernels).
set -u
root="${GRAFT_REPO_ROOT:-$(pwd)}"; out="$root/gpurun_out/r06pf"; mkdir -p "$out"
cd "$root"
for n in 2 4 8; do
  OMX_BENCH_ONE_GPU=1 HSA_ENABLE_IPC_MODE_LEGACY=0 timeout -k 10 900 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29600 + n)) \
      bench.py --gpus $n --steps 16 --warmup 4 > "$out/n$n.out" 2> "$out/n$n.err"
  echo "N=$n rc=$?"
  tail -1 "$out/n$n.out" > "$out/r06_one_gpu_preflight_n$n.json"
  python3 - <<PY
import json
try:
    d = json.load(open("$out/r06_one_gpu_preflight_n$n.json"))
    print("  value", d["value"], "first", d.get("first_tokens", [None])[0], "scope", d["config"].get("allreduce", "")[:160])
    print("  secondary", (d.get("secondary") or {}).get("value"), "mixtral", (d.get("mixtral") or {}).get("value"), "expert-TP", ((d.get("mixtral") or {}).get("expert_tensor_parallel") or {}).get("value"))
except Exception as e:
    print("  no JSON line:", e)
PY
  tail -c 600 "$out/n$n.err"
done
