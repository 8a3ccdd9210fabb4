#!/bin/bash
set -u
ulimit -c 0
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { echo "build failed"; tail -5 gpurun_out/build.log; }
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider --durations=5 \
  -k "${TEST_K:-nce_parity or stem_conv or refsrc or trainer_iteration or engine_matches or rccl}" > gpurun_out/pytest_i.log 2>&1
echo "pytest exit: $?" >> gpurun_out/pytest_i.log
grep -E "passed|failed|error|exit|FAILED|Error|assert" gpurun_out/pytest_i.log | tail -25
i=0
for e in "X=1" "PCMI_NCE_SPLITS=1"; do
  i=$((i+1))
  env $e timeout 300 python bench.py --steps 20 --warmup 5 --no-roofline --no-cpu-baseline > "gpurun_out/bench_i_$i.log" 2>&1
  echo "$e: $(tail -1 "gpurun_out/bench_i_$i.log" | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")"
done
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/prof_i" -o bench -- python "$GRAFT_REPO_ROOT/bench.py" --steps 6 --warmup 2 --no-cpu-baseline --no-roofline > "$GRAFT_REPO_ROOT/gpurun_out/prof_i.log" 2>&1
cd "$GRAFT_REPO_ROOT"
find gpurun_out/prof_i -name "*kernel_trace*" -size +8M -delete
grep -E "nce_|stem" gpurun_out/prof_i/bench_kernel_stats.csv | cut -d, -f1-4 | cut -c1-120
echo done
