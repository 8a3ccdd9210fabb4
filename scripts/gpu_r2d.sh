#!/bin/bash
set -u
ulimit -c 0
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { echo "build failed"; tail -5 gpurun_out/build.log; }
timeout 600 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x \
  -k "${TEST_K:-batchnorm or bn_eval or spconv_parity or engine_matches or (trainer_iteration and nce)}" > gpurun_out/pytest_d.log 2>&1
echo "pytest exit: $?" >> gpurun_out/pytest_d.log
grep -E "passed|failed|error|exit|FAILED|Error" gpurun_out/pytest_d.log | tail -8
i=0
for e in "X=1" "CB=True" "PCMI_KSPLIT_TARGET=50" "PCMI_KSPLIT_TARGET=12"; do
  i=$((i+1))
  extra=""
  if [ "$e" = "CB=True" ]; then extra="--set misc.concurrent_backward=True"; fi
  env $e timeout 300 python bench.py --steps 20 --warmup 5 --no-roofline --no-cpu-baseline $extra > "gpurun_out/bench_d_$i.log" 2>&1
  echo "$e: $(tail -1 "gpurun_out/bench_d_$i.log" | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config'].get('host_phase_ms_per_step'))")"
done
timeout 300 python scripts/kbench.py > gpurun_out/kbench_d.txt 2>&1; tail -32 gpurun_out/kbench_d.txt | cut -c1-150
echo done
