#!/bin/bash
# Round 3, GPU call 6: validation of the new default (split-precision conv from 512 rows) -- whole suite, bench, kernel stats.
set -u
ulimit -c 0
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
TAG=${TAG:-r03g}
O=gpurun_out/$TAG
mkdir -p $O
T0=$(date +%s)
stamp() { echo "[$(( $(date +%s) - T0 )) s] $*" | tee -a $O/stages.log; }
python - <<'PY' | tee $O/host.txt
import os, time, torch
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)), "torch threads", torch.get_num_threads(), "loadavg", os.getloadavg())
a = torch.randn(4096, 4096); t = time.perf_counter(); (a @ a).sum().item(); print("4096^3 fp32 matmul s:", round(time.perf_counter() - t, 3))
PY
B="python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-roofline"
stamp "bench x3"
for i in 1 2 3; do timeout 120 $B 2>> $O/bench.err | tail -1 > $O/run_$i.json; python -c "
import json; d=json.load(open('$O/run_$i.json')); print('run $i |', d['value'], 'pairs/s', d['ms_per_step'], 'ms')" | tee -a $O/runs.txt; done
stamp "full GPU suite"
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider --durations=10 -rP > $O/pytest_gpu.log 2>&1
echo "pytest exit $?" | tee -a $O/stages.log; grep -E "passed|failed|s call" $O/pytest_gpu.log | tail -13; grep -E "^FAILED|^ERROR" $O/pytest_gpu.log | head -20
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke exit $?" | tee -a $O/stages.log; tail -1 $O/smoke.log
stamp "rocprofv3 kernel stats"
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/$O/prof" -o bench -- \
    python "$GRAFT_REPO_ROOT/bench.py" --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > "$GRAFT_REPO_ROOT/$O/prof.log" 2>&1 )
echo "prof exit $?" >> $O/stages.log
find $O/prof -name "*kernel_trace*" -size +8M -delete 2>/dev/null
stamp "done"
