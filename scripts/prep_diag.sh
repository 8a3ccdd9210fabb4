#!/bin/bash
# Where the batch preparation (host-synchronous planning on the plan stream) spends its time, and the step under the
# alternatives: helper-thread / inline prefetch of the next batch, more hardware queues.  gpurun_out/$TAG/runs.txt
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
TAG=${TAG:-r02h}
O=gpurun_out/$TAG
mkdir -p $O
run() {
  local label=$1; shift
  local line
  line=$(env "$@" 2>> $O/diag.err | tail -1)
  echo "$label | $(echo "$line" | python -c "
import sys,json
d=json.loads(sys.stdin.read()); c=d['config']; h=c.get('host_phase_ms_per_step',{})
print(d['value'], 'pairs/s', d['ms_per_step'], 'ms |', {k: v for k, v in h.items() if not k.endswith('_cpu')})" 2>/dev/null)" | tee -a $O/runs.txt
}
B="timeout 100 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline"
run "default"                 $B
run "default"                 $B
run "prefetch thread"         $B --set misc.prefetch_thread=True
run "prefetch inline"         $B --set misc.prefetch_thread=False
run "8 hw queues"             GPU_MAX_HW_QUEUES=8 $B
run "8 hw queues + thread"    GPU_MAX_HW_QUEUES=8 $B --set misc.prefetch_thread=True
run "one-tile conv launches"  PCMI_SPCONV_STREAMK=0 $B
echo done
