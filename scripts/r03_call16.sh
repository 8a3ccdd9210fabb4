#!/bin/bash
# Round 3, GPU call 16: step rate with the deterministic (atomic-free) loss-gradient scatters
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r03w
mkdir -p $O
B="python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-roofline"
run() { local label=$1; shift; env "$@" 2>> $O/bench.err | tail -1 > "$O/run_${label// /_}.json"; python -c "
import json
try:
  d=json.load(open('$O/run_${label// /_}.json')); print('$label |', d['value'], 'pairs/s', d['ms_per_step'], 'ms')
except Exception as e: print('$label failed', e)" | tee -a $O/runs.txt; }
run "nce a" timeout 120 $B
run "nce b" timeout 120 $B
run "nce c" timeout 120 $B
run "hardest a" timeout 120 $B --loss hardest
run "hardest b" timeout 120 $B --loss hardest
echo done
