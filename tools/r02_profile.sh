#!/bin/bash
# Round-2 measurement set (GPU box, repo root): the bench line, the rocprofv3 kernel stats of the SAME command, the PMC passes.
set -u
root="${GRAFT_REPO_ROOT:-$(pwd)}"; out="$root/gpurun_out/r02"; mkdir -p "$out"
cd /tmp; export TMPDIR=/tmp; cd "$root"
python3 bench.py > "$out/bench_n1.json" 2> "$out/bench_n1.err"; echo "bench rc=$?"
tail -c 600 "$out/bench_n1.err"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/prof" -o b -- python3 bench.py --no-cpu-baseline > "$out/bench_prof.json" 2> "$out/bench_prof.err"
f=$(find "$out/prof" -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" "$out/r02_bench_kernel_stats.csv" && head -12 "$f"
rm -rf "$out/prof"
bash tools/pmc_profile.sh
