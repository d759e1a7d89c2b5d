#!/bin/bash
# Round-5 measurement set (GPU box, repo root): the bench line, the rocprofv3 kernel stats of the SAME command, the PMC passes
# (counters only, program directly after `--`).
set -u
root="${GRAFT_REPO_ROOT:-$(pwd)}"; out="$root/gpurun_out/r05"; mkdir -p "$out/keep"
cd /tmp; export TMPDIR=/tmp; cd "$root"
python3 bench.py > "$out/r05_bench_n1.json" 2> "$out/bench_n1.err"; echo "bench rc=$?"
tail -c 400 "$out/bench_n1.err"
timeout -k 5 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/prof" -o b -- python3 bench.py --no-cpu-baseline > "$out/bench_prof.json" 2> "$out/bench_prof.err"
f=$(find "$out/prof" -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" "$out/r05_bench_kernel_stats.csv" && head -8 "$f"
rm -rf "$out/prof"
# PMC: HBM bytes of the decode step's kernels, matrix-core busy of the prefill / FLUX kernels
timeout -k 5 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$out/fetch" -o f -- python3 tools/decode_steps.py 8 > "$out/fetch.log" 2>&1
f=$(find "$out/fetch" -name '*counter_collection.csv' | head -1)
[ -n "$f" ] && python3 tools/pmc_report.py trim "$f" "$out/keep/r05_pmc_fetch_size_step.csv" gemv_kernel attn_step_kernel embed_kernel sample_finalize
timeout -k 5 900 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d "$out/mfma" -o m -- python3 tools/prefill_flux_steps.py > "$out/mfma.log" 2>&1
f=$(find "$out/mfma" -name '*counter_collection.csv' | head -1)
[ -n "$f" ] && python3 tools/pmc_report.py trim "$f" "$out/keep/r05_pmc_mfma_busy.csv" gemm_bf16 attn_prefill flash dit_ gemm_
rm -rf "$out/fetch" "$out/mfma"
# FLUX step and the 2048-token prefill with the round's kernels (four-wave flash attention, four-wave GEMM tile on full-chip grids)
for w in flux prefill; do
  if [ $w = flux ]; then cmd="tools/flux_bench.py"; else cmd="tools/prefill_bench.py 2048"; fi
  timeout -k 5 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/$w" -o s -- python3 $cmd > "$out/$w.log" 2>&1
  f=$(find "$out/$w" -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" "$out/keep/r05_${w}_kernel_stats.csv" && head -5 "$f"
  rm -rf "$out/$w"
done
ls -la "$out" "$out/keep"
