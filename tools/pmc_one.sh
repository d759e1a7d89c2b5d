#!/bin/bash
# usage: tools/pmc_one.sh <kernel-substring> <script.py> "<COUNTERS...>"
cd /tmp; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
out=gpurun_out/pmc_one_$$
timeout 150 rocprofv3 --pmc $3 --kernel-trace --output-format csv -d "$out" -o x -- python3 "$2" > /dev/null 2>&1
f=$(find "$out" -name '*counter_collection.csv' | head -1)
[ -n "$f" ] && python3 tools/pmc_sum.py "$f" "$1"
rm -rf "$out"
