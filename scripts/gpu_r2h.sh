#!/bin/bash
set -u
ulimit -c 0
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { echo "build failed"; tail -5 gpurun_out/build.log; }
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider --durations=5 \
  -k "${TEST_K:-spconv_parity or spconv_golden or adjointness or conv16 or engine_matches or refsrc or (trainer_iteration and nce)}" > gpurun_out/pytest_h.log 2>&1
echo "pytest exit: $?" >> gpurun_out/pytest_h.log
grep -E "passed|failed|error|exit|FAILED|Error|assert" gpurun_out/pytest_h.log | tail -25
KBENCH_SUSTAINED=0 timeout 300 python scripts/kbench.py > gpurun_out/kbench_h.txt 2>&1; sed -n 3,14p gpurun_out/kbench_h.txt | cut -c1-150
i=0
for e in "X=1" "PCMI_WAVE_WG=0"; do
  i=$((i+1))
  env $e timeout 300 python bench.py --steps 20 --warmup 5 --no-roofline --no-cpu-baseline > "gpurun_out/bench_h_$i.log" 2>&1
  echo "$e: $(tail -1 "gpurun_out/bench_h_$i.log" | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")"
done
echo done
