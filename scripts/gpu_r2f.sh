#!/bin/bash
set -u
ulimit -c 0
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { echo "build failed"; tail -5 gpurun_out/build.log; }
timeout 900 python -m pytest tests/test_gpu_loader.py tests/test_gpu_semseg.py tests/test_gpu_checkpoint.py tests/test_gpu_parity.py -m gpu -q --tb=short -p no:cacheprovider --durations=5 \
  -k "${TEST_K:-loader or semseg or any_width or cross_entropy or segmentation or checkpoint or kernel_order or conv16 or streamk or (spconv_parity and small)}" > gpurun_out/pytest_f.log 2>&1
echo "pytest exit: $?" >> gpurun_out/pytest_f.log
grep -E "passed|failed|error|exit|FAILED|Error|assert" gpurun_out/pytest_f.log | tail -25
i=0
for e in "PCMI_WGRAD_PRIORITY=0 CB=True" "PCMI_WGRAD_PRIORITY=0 CB=False"; do
  i=$((i+1))
  extra=""
  case "$e" in *CB=True*) extra="--set misc.concurrent_backward=True";; esac
  env $e timeout 300 python bench.py --steps 20 --warmup 5 --no-roofline --no-cpu-baseline $extra > "gpurun_out/bench_f_$i.log" 2>&1
  echo "$e: $(tail -1 "gpurun_out/bench_f_$i.log" | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config'].get('host_phase_ms_per_step'))")"
done
echo done
