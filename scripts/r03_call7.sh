#!/bin/bash
# Round 3, GPU call 7: the tests that now share the device's ReLU patterns with the oracle; A/B of the offset-split target
# and the unit-balanced launch threshold under the new default.
set -u
ulimit -c 0
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
TAG=${TAG:-r03h}
O=gpurun_out/$TAG
mkdir -p $O
B="python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-roofline"
timeout 600 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -rP -k "trainer_iteration or segmentation_trainer or semseg" > $O/pytest_masks.log 2>&1
echo "mask tests exit $?"; grep -E "passed|failed|worst state|worst parameters|sat on the other side" $O/pytest_masks.log | cut -c1-300 | tail -12
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke exit $?"; tail -1 $O/smoke.log
run() { local label=$1; shift; env "$@" 2>> $O/bench.err | tail -1 > "$O/run_${label// /_}.json"; python -c "
import json
try:
  d=json.load(open('$O/run_${label// /_}.json')); print('$label |', d['value'], 'pairs/s', d['ms_per_step'], 'ms')
except Exception as e: print('$label failed', e)" | tee -a $O/runs.txt; }
run "default a" timeout 120 $B
run "ksplit 15" PCMI_KSPLIT_TARGET=15 timeout 120 $B
run "ksplit 40" PCMI_KSPLIT_TARGET=40 timeout 120 $B
run "ksplit 10" PCMI_KSPLIT_TARGET=10 timeout 120 $B
run "streamk 512" PCMI_SPCONV_STREAMK=512 timeout 120 $B
run "streamk 128" PCMI_SPCONV_STREAMK=128 timeout 120 $B
run "default b" timeout 120 $B
run "conv16 128" PCMI_CONV16=128 timeout 120 $B
run "x3t off" PCMI_WGRAD_X3T=0 timeout 120 $B
run "x3t all" PCMI_WGRAD_X3T=8192 PCMI_WGRAD_X3T_MAX=100000000 timeout 120 $B
echo done
