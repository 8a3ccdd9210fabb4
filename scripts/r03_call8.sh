#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r03i
mkdir -p $O
B="python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-roofline"
run() { local label=$1; shift; env "$@" 2>> $O/bench.err | tail -1 > "$O/run_${label// /_}.json"; python -c "
import json
try:
  d=json.load(open('$O/run_${label// /_}.json')); print('$label |', d['value'], 'pairs/s', d['ms_per_step'], 'ms')
except Exception as e: print('$label failed', e)" | tee -a $O/runs.txt; }
for i in 1 2 3; do
  run "window 16384-100000 (default) $i" timeout 120 $B
  run "window 8192-inf $i" PCMI_WGRAD_X3T=8192 PCMI_WGRAD_X3T_MAX=100000000 timeout 120 $B
  run "window 16384-inf $i" PCMI_WGRAD_X3T=16384 PCMI_WGRAD_X3T_MAX=100000000 timeout 120 $B
done
echo done
