#!/bin/bash
# Round-2 kernel stats of the secondary workloads (GPU box, repo root): prefill, FLUX step, MoE block.
set -u
root="${GRAFT_REPO_ROOT:-$(pwd)}"; out="$root/gpurun_out/r02"; mkdir -p "$out"
cd /tmp; export TMPDIR=/tmp; cd "$root"
for w in "prefill tools/prefill_bench.py 2048" "flux tools/flux_bench.py" "moe tools/moe_bench.py"; do
  set -- $w; name=$1; shift
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/p_$name" -o p -- python3 "$@" > "$out/$name.log" 2>&1 < /dev/null
  f=$(find "$out/p_$name" -name '*kernel_stats.csv' | head -1)
  [ -n "$f" ] && cp "$f" "$out/r02_${name}_kernel_stats.csv" && echo "$name:" && head -n 5 "$f" | cut -c1-150
  rm -rf "$out/p_$name"
done
