#!/bin/bash
# Round 3, GPU call 10: L2 locality of the level-1 gathers.  PMC says 80 % of the gathered rows of the 96->96 conv come
# over the fabric; an LRU model of one XCD (scripts/l2_locality_model.py) says that dispatching the unit-balanced launch in
# R rounds over R-times-finer mask-sort chunks would bring that to 10-40 % (R = 4).  Costs: more partial tiles, +5 % MFMA.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r03k
mkdir -p $O
PCMI_SK_ROUNDS=4 PCMI_SORT_SUB=4 timeout 300 python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "streamk_matches_plain or conv16_x3_split or wgrad_x3t_split" 2>&1 | tail -3 | tee $O/tests.txt
B="python bench.py --steps 30 --warmup 8 --no-cpu-baseline"
run() { local label=$1; shift; env "$@" 2>> $O/bench.err | tail -1 > "$O/run_${label// /_}.json"; python -c "
import json
try:
  d=json.load(open('$O/run_${label// /_}.json')); r=d['roofline']
  print('$label |', d['value'], 'pairs/s', d['ms_per_step'], 'ms | dominant', r['ms'], 'ms |', ' '.join('%s=%.4f' % (k['kernel'][:28].replace(' ','_'), k['ms']) for k in d.get('kernels', [])[1:4]))
except Exception as e: print('$label failed', e)" | tee -a $O/runs.txt; }
run "R1 S1 a" timeout 150 $B
run "R2 S2" PCMI_SK_ROUNDS=2 PCMI_SORT_SUB=2 timeout 150 $B
run "R4 S4" PCMI_SK_ROUNDS=4 PCMI_SORT_SUB=4 timeout 150 $B
run "R1 S4" PCMI_SK_ROUNDS=1 PCMI_SORT_SUB=4 timeout 150 $B
run "R4 S1" PCMI_SK_ROUNDS=4 PCMI_SORT_SUB=1 timeout 150 $B
run "R2 S4" PCMI_SK_ROUNDS=2 PCMI_SORT_SUB=4 timeout 150 $B
run "R1 S1 b" timeout 150 $B
tail -3 $O/bench.err
echo done
