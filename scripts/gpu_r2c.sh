#!/bin/bash
set -u
ulimit -c 0
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { echo "build failed"; tail -5 gpurun_out/build.log; }
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider --durations=6 \
  -k "${TEST_K:-batchnorm or bn_eval or spconv_parity or stem_conv or spconv_golden or engine_matches or adjointness or refsrc or (trainer_iteration and nce) or rccl}" \
  > gpurun_out/pytest_c.log 2>&1
echo "pytest exit: $?" >> gpurun_out/pytest_c.log
grep -E "passed|failed|error|exit|FAILED|Error" gpurun_out/pytest_c.log | tail -15
for e in "X=1" "PCMI_BN_FUSE_FINAL=0" "PCMI_WGRAD_DIRECT=0 PCMI_WGRAD_ARRIVE_MAX=0" ${EXTRA_ENVS:-}; do
  env $e timeout 300 python bench.py --steps 20 --warmup 5 --no-roofline --no-cpu-baseline > "gpurun_out/bench_c_${e%%=*}.log" 2>&1
  echo "$e: $(tail -1 "gpurun_out/bench_c_${e%%=*}.log" | cut -c1-200)"
done
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/prof_c" -o bench -- python "$GRAFT_REPO_ROOT/bench.py" --steps 6 --warmup 2 --no-cpu-baseline --no-roofline > "$GRAFT_REPO_ROOT/gpurun_out/prof_c.log" 2>&1
cd "$GRAFT_REPO_ROOT"
find gpurun_out/prof_c -name "*kernel_trace*" -size +8M -delete
echo done
