#!/bin/bash
# usage: tools/pmc_run.sh <kernel-substring> <out-dir> <script.py> -- collects PMC groups one pass each (rocprofv3 --pmc only)
cd /tmp; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
pat="$1"; out="$2"; script="$3"
groups=("SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD" "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM")
i=0
for c in "${groups[@]}"; do
  i=$((i+1))
  timeout 150 rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$out/g$i" -o x -- python3 "$script" > /dev/null 2>&1
  f=$(find "$out/g$i" -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && python3 tools/pmc_sum.py "$f" "$pat"
done
