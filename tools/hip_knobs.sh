#!/bin/bash
# A/B of HIP runtime knobs on the decode bench (one box, back to back).  usage: bash tools/hip_knobs.sh KEY=VAL ...
run() { timeout 300 python3 bench.py --no-flux --no-cpu-baseline 2>/dev/null < /dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; }
echo "base $(run)"
for kv in "$@"; do echo "$kv $(env $kv bash -c "$(declare -f run); run")"; done
echo "base $(run)"
