#!/bin/bash
set -u
ulimit -c 0
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { echo "build failed"; tail -5 gpurun_out/build.log; }
timeout 600 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider \
  -k "${TEST_K:-spconv_parity or spconv_golden or streamk or adjointness or engine_matches or refsrc or (trainer_iteration and nce)}" > gpurun_out/pytest_e.log 2>&1
echo "pytest exit: $?" >> gpurun_out/pytest_e.log
grep -E "passed|failed|error|exit|FAILED|Error" gpurun_out/pytest_e.log | tail -12
timeout 300 python scripts/kbench.py > gpurun_out/kbench_e.txt 2>&1; tail -32 gpurun_out/kbench_e.txt | cut -c1-150
i=0
for e in "X=1" "PCMI_CONV16=0" "GPU_MAX_HW_QUEUES=16 CB=True" "GPU_MAX_HW_QUEUES=4 CB=True" "GPU_MAX_HW_QUEUES=2 CB=False"; do
  i=$((i+1))
  extra=""
  case "$e" in *CB=True*) extra="--set misc.concurrent_backward=True";; esac
  env $e timeout 300 python bench.py --steps 20 --warmup 5 --no-roofline --no-cpu-baseline $extra > "gpurun_out/bench_e_$i.log" 2>&1
  echo "$e: $(tail -1 "gpurun_out/bench_e_$i.log" | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config'].get('host_phase_ms_per_step'))")"
done
echo done
