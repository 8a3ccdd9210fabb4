#!/bin/bash
# Runs on the GPU box (via gpurun): GPU parity tests, smoke, a short bench, and a rocprofv3 kernel-trace
# of the bench.  Everything is logged under gpurun_out/ (merged back by gpurun).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
mkdir -p gpurun_out
STEPS=${STEPS:-10}
WARM=${WARM:-3}
echo "== rocminfo" > gpurun_out/env.log
(rocminfo | grep -E "Marketing Name|gfx|Compute Unit" | head -8; nproc; free -g | head -2) >> gpurun_out/env.log 2>&1
if [ "${SKIP_TESTS:-0}" != "1" ]; then
  timeout ${TEST_TIMEOUT:-1500} python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider ${PYTEST_ARGS:-} > gpurun_out/pytest_gpu.log 2>&1
  echo "pytest exit: $?" >> gpurun_out/pytest_gpu.log
  tail -5 gpurun_out/pytest_gpu.log
fi
if [ "${SKIP_SMOKE:-0}" != "1" ]; then
  timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1
  echo "smoke exit: $?" >> gpurun_out/smoke.log
  tail -3 gpurun_out/smoke.log
fi
if [ "${SKIP_BENCH:-0}" != "1" ]; then
  timeout 900 python bench.py --steps $STEPS --warmup $WARM --layer-table gpurun_out/layer_table.tsv ${BENCH_ARGS:-} > gpurun_out/bench.log 2>&1
  echo "bench exit: $?" >> gpurun_out/bench.log
  tail -3 gpurun_out/bench.log
fi
if [ "${SKIP_PROF:-0}" != "1" ]; then
  cd /tmp
  timeout 900 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/gpurun_out/prof" -o bench -- python "$GRAFT_REPO_ROOT/bench.py" --steps 5 --warmup 2 --no-cpu-baseline --no-roofline > "$GRAFT_REPO_ROOT/gpurun_out/prof.log" 2>&1
  echo "prof exit: $?" >> "$GRAFT_REPO_ROOT/gpurun_out/prof.log"
  cd "$GRAFT_REPO_ROOT"
  find gpurun_out/prof -name "*kernel_stats*" | head
  # keep the merged payload small: drop the raw trace, keep the stats
  find gpurun_out/prof -name "*kernel_trace*" -size +20M -delete
fi
echo done
