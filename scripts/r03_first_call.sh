#!/bin/bash
# First GPU call of round 3 (~4 min): the paths written after round 2's GPU budget was spent, then the open questions of
# DESIGN.md "What comes next".  Everything is logged under gpurun_out/$TAG.
set -u
ulimit -c 0
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
TAG=${TAG:-r03a}
O=gpurun_out/$TAG
mkdir -p $O
B="timeout 100 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-roofline"
run() {
  local label=$1; shift
  local line
  line=$(env "$@" 2>> $O/ab.err | tail -1)
  echo "$label | $(echo "$line" | python -c "
import sys,json
d=json.loads(sys.stdin.read()); h=d['config'].get('host_phase_ms_per_step',{})
print(d['value'], 'pairs/s', d['ms_per_step'], 'ms |', {k: v for k, v in h.items() if not k.endswith('_cpu')})" 2>/dev/null)" | tee -a $O/ab.txt
}
PCMI_TEST_EXPERIMENTAL=1 timeout 400 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > $O/pytest_gpu.log 2>&1
echo "pytest exit $?" | tee -a $O/ab.txt; tail -3 $O/pytest_gpu.log
for rep in 1 2 3; do
  run "default"                       $B
  run "plan defer"                    PCMI_PLAN_DEFER=1 $B
  run "helper thread, switch 0.1 ms"  $B --set misc.prefetch_thread=True --set misc.switch_interval=0.0001
  run "helper thread + plan defer"    PCMI_PLAN_DEFER=1 $B --set misc.prefetch_thread=True --set misc.switch_interval=0.0001
done
echo done
