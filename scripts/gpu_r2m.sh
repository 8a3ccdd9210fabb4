#!/bin/bash
set -u
ulimit -c 0
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { echo "build failed"; tail -5 gpurun_out/build.log; }
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider \
  -k "${TEST_K:-engine_matches or refsrc or trainer_iteration or full_config or segmentation or rccl or batchnorm or checkpoint}" > gpurun_out/pytest_m.log 2>&1
echo "pytest exit: $?" >> gpurun_out/pytest_m.log
grep -E "passed|failed|error|exit|FAILED|Error|assert" gpurun_out/pytest_m.log | tail -12
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
i=0
for e in "X=1" "PCMI_CONV_BN_STATS=0" "X=2"; do
  i=$((i+1))
  env $e timeout 300 python bench.py --steps 25 --warmup 5 --no-roofline --no-cpu-baseline > "gpurun_out/bench_m_$i.log" 2>&1
  echo "$e: $(tail -1 "gpurun_out/bench_m_$i.log" | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")"
done
echo done
