#!/bin/bash
set -u
ulimit -c 0
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { echo "build failed"; tail -5 gpurun_out/build.log; }
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider \
  -k "${TEST_K:-streamk or adjointness or conv16 or (spconv_parity and (big or mid)) or refsrc or segmentation_trainer}" > gpurun_out/pytest_n.log 2>&1
echo "pytest exit: $?" >> gpurun_out/pytest_n.log
grep -E "passed|failed|error|exit|FAILED|Error|assert" gpurun_out/pytest_n.log | tail -12
echo "== KC auto"; KBENCH_LEVELS=0,1 KBENCH_SUSTAINED=0 timeout 200 python scripts/kbench.py 2>&1 | grep "^L" | cut -c1-150
echo "== KC 32"; PCMI_CONV16_KC=32 KBENCH_LEVELS=0 KBENCH_SUSTAINED=0 timeout 200 python scripts/kbench.py 2>&1 | grep "^L0" | cut -c1-150
i=0
for e in "X=1" "PCMI_CONV16_KC=32"; do
  i=$((i+1))
  env $e timeout 300 python bench.py --steps 25 --warmup 5 --no-roofline --no-cpu-baseline > "gpurun_out/bench_n_$i.log" 2>&1
  echo "$e: $(tail -1 "gpurun_out/bench_n_$i.log" | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")"
done
echo done
