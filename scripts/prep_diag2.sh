#!/bin/bash
# After the pair selection moved to threaded int32 passes: the default step, and the helper-thread preparation with a short
# GIL switch interval.  gpurun_out/$TAG/runs.txt
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
TAG=${TAG:-r02i}
O=gpurun_out/$TAG
mkdir -p $O
run() {
  local label=$1; shift
  local line
  line=$(env "$@" 2>> $O/diag.err | tail -1)
  echo "$line" >> $O/lines.jsonl
  echo "$label | $(echo "$line" | python -c "
import sys,json
d=json.loads(sys.stdin.read()); c=d['config']; h=c.get('host_phase_ms_per_step',{})
print(d['value'], 'pairs/s', d['ms_per_step'], 'ms |', {k: v for k, v in h.items() if not k.endswith('_cpu')})" 2>/dev/null)" | tee -a $O/runs.txt
}
B="timeout 100 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline"
run "default"                               $B
run "default"                               $B
run "default"                               $B
run "prefetch thread, switch 0.1 ms"        $B --set misc.prefetch_thread=True --set misc.switch_interval=0.0001
run "prefetch thread, switch 0.1 ms"        $B --set misc.prefetch_thread=True --set misc.switch_interval=0.0001
run "default, hardest"                      $B --loss hardest
echo done
