#!/bin/bash
# Round-2 PMC passes (run on the GPU box from the repo root; each rocprofv3 call has counters ONLY -- no sys/hip/hsa trace domains --
# and the program directly after `--`).  Raw files land under gpurun_out/r02_pmc/, trimmed copies under gpurun_out/r02_pmc/keep/ for
# profiles/.
#   usage: bash tools/pmc_profile.sh
set -u
root="${GRAFT_REPO_ROOT:-$(pwd)}"
out="$root/gpurun_out/r02_pmc"; mkdir -p "$out/keep"
cd /tmp; export TMPDIR=/tmp; cd "$root"
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$out/fetch" -o f -- python3 tools/decode_steps.py 8 > "$out/fetch.log" 2>&1
f=$(find "$out/fetch" -name '*counter_collection.csv' | head -1)
[ -n "$f" ] && python3 tools/pmc_report.py trim "$f" "$out/keep/r02_pmc_fetch_size_step.csv" gemv_kernel attn_step_kernel embed_kernel sample_finalize
timeout 900 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d "$out/mfma" -o m -- python3 tools/prefill_flux_steps.py > "$out/mfma.log" 2>&1
f=$(find "$out/mfma" -name '*counter_collection.csv' | head -1)
[ -n "$f" ] && python3 tools/pmc_report.py trim "$f" "$out/keep/r02_pmc_mfma_busy.csv" gemm_bf16 attn_prefill flash dit_ gemm_
tail -n 3 "$out/fetch.log"; tail -n 3 "$out/mfma.log"
ls -la "$out/keep"
