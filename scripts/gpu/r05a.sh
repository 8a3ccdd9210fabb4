#!/bin/bash
# Round 5, call 1: the whole GPU suite on the new build (blocking run-ahead wait, 3 gradient buckets, lean backward
# statistics, bucket hand-over tests, hardest trace, fp64-anchored BN / NCE tolerances), the driver's bench line with the new
# keys (roofline.in_step_*, extra.rotating_batches, extra.fp32_mfma), and A/B series of the two switches that exist as
# environment variables: PCMI_BN_LEAN_ROWS (48-register backward statistics) and PCMI_WGRAD_SIDE2 (small weight
# gradients on a stream of their own).
set -u
ulimit -c 0
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$ROOT"
export TMPDIR=/tmp
TAG=${TAG:-r05a}
O=$ROOT/gpurun_out/$TAG
mkdir -p $O
T0=$(date +%s)
stamp() { echo "[$(( $(date +%s) - T0 )) s] $*" | tee -a $O/stages.log; }
python -c "import __graft_entry__ as g; g.build()" > $O/build.txt 2>&1
stamp "1 new / changed GPU tests first"
timeout 900 python -m pytest tests/test_gpu_bucket_sync.py tests/test_gpu_trace.py -m gpu -q --tb=short -p no:cacheprovider -s > $O/pytest_new.log 2>&1
echo "pytest(new) exit $?" | tee -a $O/stages.log; grep -E "passed|failed|skipped" $O/pytest_new.log | tail -3
stamp "2 bench as the driver runs it"
timeout 700 python bench.py > $O/bench_line.json 2> $O/bench.err
echo "bench exit $?" >> $O/stages.log; cut -c1-400 $O/bench_line.json; echo
B="python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-roofline --no-extra"
run() {  # label, env...
  local label=$1; shift
  for i in 1 2 3; do
    env "$@" timeout 150 $B 2>> $O/bench.err | tail -1 > $O/ab_${label}_$i.json
    python -c "
import json
try:
  d = json.load(open('$O/ab_${label}_$i.json')); print('$label run $i |', d['value'], 'pairs/s', d['ms_per_step'], 'ms | enqueue', d['config']['host_enqueue_ms_per_step'], '| fwd host', d['config'].get('host_phase_ms_per_step', {}).get('forward'), 'cpu', d['config'].get('host_phase_ms_per_step', {}).get('forward_cpu'))
except Exception as e:
  print('$label run $i failed:', e)" | tee -a $O/ab.txt
  done
}
stamp "3 A/B"
run base PCMI_NOP=1
run lean_off PCMI_BN_LEAN_ROWS=0
run lean_l2 PCMI_BN_LEAN_ROWS=16384
run side2 PCMI_WGRAD_SIDE2=1
run side2_lean_off PCMI_WGRAD_SIDE2=1 PCMI_BN_LEAN_ROWS=0
stamp "4 forced reducer (3 buckets, 16 channels)"
for i in 1 2; do
  timeout 150 $B --set misc.force_reducer=True 2>> $O/bench.err | tail -1 > $O/forced_$i.json
  python -c "
import json
d = json.load(open('$O/forced_$i.json')); print('forced reducer run $i |', d['value'], 'pairs/s', d['ms_per_step'], 'ms |', json.dumps(d['config']['collective']))" | tee -a $O/ab.txt
done
PCMI_RCCL_MAX_CHANNELS=0 timeout 150 $B --set misc.force_reducer=True --set misc.bucket_mb=32 2>> $O/bench.err | tail -1 > $O/forced_r04form.json
python -c "
import json
d = json.load(open('$O/forced_r04form.json')); print('forced reducer, round-4 form (5 buckets, RCCL default channels) |', d['value'], 'pairs/s', d['ms_per_step'], 'ms |', json.dumps(d['config']['collective']))" | tee -a $O/ab.txt
stamp "5 GPU test suite + smoke"
timeout 1200 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider --durations=8 --deselect tests/test_gpu_bucket_sync.py --deselect tests/test_gpu_trace.py > $O/pytest_gpu.log 2>&1
echo "pytest exit $?" | tee -a $O/stages.log; grep -E "passed|failed" $O/pytest_gpu.log | tail -2; grep -E "^FAILED|^ERROR" $O/pytest_gpu.log | head
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke exit $?" | tee -a $O/stages.log; tail -1 $O/smoke.log
stamp "done"
