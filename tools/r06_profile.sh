#!/bin/bash
# Round-6 measurement set (GPU box, repo root): the bench line, the rocprofv3 kernel stats of the SAME command, the PMC passes (counters
# only, program directly after `--`), the 4-bit step, the drop-in route, the FLUX step and the 2 048-token prompt.
set -u
root="${GRAFT_REPO_ROOT:-$(pwd)}"; out="$root/gpurun_out/r06"; mkdir -p "$out/keep"
cd /tmp; export TMPDIR=/tmp; cd "$root"
python3 bench.py > "$out/keep/r06_bench_n1.json" 2> "$out/bench_n1.err"; echo "bench rc=$?"
tail -c 300 "$out/bench_n1.err"
timeout -k 5 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/prof" -o b -- python3 bench.py --no-cpu-baseline > "$out/bench_prof.json" 2> "$out/bench_prof.err"
f=$(find "$out/prof" -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" "$out/keep/r06_bench_kernel_stats.csv" && head -6 "$f" | cut -c1-160
rm -rf "$out/prof"
# PMC: HBM bytes of the decode step's kernels (bf16 and 4-bit), matrix-core busy of the prefill / FLUX kernels
timeout -k 5 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$out/fetch" -o f -- python3 tools/decode_steps.py 8 > "$out/fetch.log" 2>&1
f=$(find "$out/fetch" -name '*counter_collection.csv' | head -1)
[ -n "$f" ] && python3 tools/pmc_report.py trim "$f" "$out/keep/r06_pmc_fetch_size_step.csv" gemv_kernel attn_step_kernel embed_kernel sample_finalize
timeout -k 5 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$out/fetchq" -o f -- python3 tools/decode_steps.py 8 2048 4 > "$out/fetchq.log" 2>&1
f=$(find "$out/fetchq" -name '*counter_collection.csv' | head -1)
[ -n "$f" ] && python3 tools/pmc_report.py trim "$f" "$out/keep/r06_pmc_fetch_size_q4_step.csv" qgemv4m_kernel attn_step_kernel qembed_kernel
timeout -k 5 900 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d "$out/mfma" -o m -- python3 tools/prefill_flux_steps.py > "$out/mfma.log" 2>&1
f=$(find "$out/mfma" -name '*counter_collection.csv' | head -1)
[ -n "$f" ] && python3 tools/pmc_report.py trim "$f" "$out/keep/r06_pmc_mfma_busy.csv" gemm_bf16 attn_prefill flash dit_ gemm_
rm -rf "$out/fetch" "$out/fetchq" "$out/mfma"
for w in q4 route route_q4 flux prefill paraformer; do
  case $w in
    q4) cmd="tools/quant_decode.py 4 2048";;
    route) cmd="tools/per_op_route_time.py 2048 64 0";;
    route_q4) cmd="tools/per_op_route_time.py 2048 64 4";;
    flux) cmd="tools/flux_bench.py";;
    prefill) cmd="tools/prefill_bench.py 2048";;
    paraformer) cmd="tools/paraformer_bench.py";;
  esac
  timeout -k 5 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/$w" -o s -- python3 $cmd > "$out/$w.log" 2>&1
  f=$(find "$out/$w" -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" "$out/keep/r06_${w}_kernel_stats.csv" && head -4 "$f" | cut -c1-160
  rm -rf "$out/$w"
done
python3 tools/per_op_route_time.py 2048 64 0 > "$out/keep/r06_route_bf16.json" 2>/dev/null
python3 tools/per_op_route_time.py 2048 64 4 > "$out/keep/r06_route_q4.json" 2>/dev/null
ls -la "$out/keep"
