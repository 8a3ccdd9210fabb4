#!/bin/bash
# round-2 GPU session A: new parity tests, A/B of the concurrent backward, kernel stats
set -u
ulimit -c 0
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { echo "build failed"; tail -5 gpurun_out/build.log; }
if [ "${SKIP_TESTS:-0}" != "1" ]; then
  timeout ${TEST_TIMEOUT:-900} python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider --durations=12 -s \
    -k "${TEST_K:-refsrc or network_features or hardest or full_config or bench_and_1cm or trainer_iteration or engine_matches}" \
    > gpurun_out/pytest_a.log 2>&1
  echo "pytest exit: $?" >> gpurun_out/pytest_a.log
  grep -E "passed|failed|error|exit|FAILED|flips|kink|worst" gpurun_out/pytest_a.log | tail -40
fi
for cb in False True; do
  timeout 300 python bench.py --steps ${STEPS:-15} --warmup 5 --no-roofline --no-cpu-baseline --set misc.concurrent_backward=$cb > gpurun_out/bench_cb$cb.log 2>&1
  tail -1 gpurun_out/bench_cb$cb.log | cut -c1-400
done
if [ "${SKIP_PROF:-0}" != "1" ]; then
  cd /tmp
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/prof_a" -o bench -- python "$GRAFT_REPO_ROOT/bench.py" --steps 6 --warmup 2 --no-cpu-baseline --no-roofline --set misc.concurrent_backward=${PROF_CB:-True} > "$GRAFT_REPO_ROOT/gpurun_out/prof_a.log" 2>&1
  cd "$GRAFT_REPO_ROOT"
  find gpurun_out/prof_a -name "*kernel_trace*" -size +8M -delete
  find gpurun_out/prof_a -name "*kernel_stats*" | head -2
fi
echo done
