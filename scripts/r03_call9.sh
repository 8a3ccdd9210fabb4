#!/bin/bash
# Round 3, GPU call 9: weight-gradient stream confined to a subset of the compute units (CU mask) -- does the chain gain?
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r03j
mkdir -p $O
B="python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-roofline"
run() { local label=$1; shift; env "$@" 2>> $O/bench.err | tail -1 > "$O/run_${label// /_}.json"; python -c "
import json
try:
  d=json.load(open('$O/run_${label// /_}.json')); print('$label |', d['value'], 'pairs/s', d['ms_per_step'], 'ms')
except Exception as e: print('$label failed', e)" | tee -a $O/runs.txt; }
run "default a" timeout 120 $B
run "mask ffffffff (all CUs, default priority)" PCMI_WGRAD_CU_MASK=ffffffff timeout 120 $B
run "mask 00ffffff (24 of 32)" PCMI_WGRAD_CU_MASK=00ffffff timeout 120 $B
run "mask 0fffffff (28 of 32)" PCMI_WGRAD_CU_MASK=0fffffff timeout 120 $B
run "mask 0000ffff (16 of 32)" PCMI_WGRAD_CU_MASK=0000ffff timeout 120 $B
run "mask 3fffffff (30 of 32)" PCMI_WGRAD_CU_MASK=3fffffff timeout 120 $B
run "mask 55555555 (every other)" PCMI_WGRAD_CU_MASK=55555555 timeout 120 $B
run "default b" timeout 120 $B
tail -3 $O/bench.err
echo done
