#!/bin/bash
# Second look at the slow runs: bench.py WITH the isolated-kernel timings (are the kernels themselves slow in a slow
# process, or only the step?) and the CPU the host thread ran on; N runs back to back.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
TAG=${TAG:-r02g}
O=gpurun_out/$TAG
mkdir -p $O
timeout 120 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=short -p no:cacheprovider -k "prepacked" > $O/pytest_prepack.log 2>&1
echo "prepack test exit $?" | tee -a $O/runs.txt; tail -2 $O/pytest_prepack.log
lscpu | grep -E "NUMA|Socket|Model name|^CPU\(s\)" >> $O/runs.txt
cat /sys/class/drm/card0/device/numa_node >> $O/runs.txt 2>/dev/null
for i in $(seq 1 ${N:-6}); do
  line=$(timeout 100 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>> $O/diag.err | tail -1)
  echo "$line" > $O/line_$i.json
  echo "run $i | $(echo "$line" | python -c "
import sys,json
d=json.loads(sys.stdin.read()); c=d['config']; k={e['kernel'].split(' @')[0][:34]: e['ms'] for e in d.get('kernels',[])}
print(d['value'], 'pairs/s', d['ms_per_step'], 'ms | cpu', c.get('host_cpu'), '| fwd phase', c.get('host_phase_ms_per_step',{}).get('forward'), '| kernels ms', list(k.values())[:5], list(k.values())[-2:])" 2>/dev/null)" | tee -a $O/runs.txt
done
echo done
