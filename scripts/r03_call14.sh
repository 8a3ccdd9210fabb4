#!/bin/bash
# Round 3, GPU call 14: regression test of the map hand-out race; x3t weight-gradient kernel at less than 2 workgroups per CU
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r03u
mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "handed_out or conv32r" 2>&1 | tail -3 | tee $O/tests.txt
B="python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-roofline"
run() { local label=$1; shift; env "$@" 2>> $O/bench.err | tail -1 > "$O/run_${label// /_}.json"; python -c "
import json
try:
  d=json.load(open('$O/run_${label// /_}.json')); print('$label |', d['value'], 'pairs/s', d['ms_per_step'], 'ms')
except Exception as e: print('$label failed', e)" | tee -a $O/runs.txt; }
run "occ 2.0 a" timeout 120 $B
run "occ 1.75" PCMI_X3T_OCC10=17 timeout 120 $B
run "occ 1.5" PCMI_X3T_OCC10=15 timeout 120 $B
run "occ 1.25" PCMI_X3T_OCC10=12 timeout 120 $B
run "occ 1.0" PCMI_X3T_OCC10=10 timeout 120 $B
run "occ 2.0 b" timeout 120 $B
tail -3 $O/bench.err
echo done
