#!/bin/bash
# Runs on the GPU box (via gpurun): GPU parity tests (one process per group, so a device fault in
# one group cannot hide the others), smoke, a short bench, and a rocprofv3 kernel-trace of the
# bench.  Everything is logged under gpurun_out/ (merged back by gpurun).
set -u
ulimit -c 0
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
mkdir -p gpurun_out
STEPS=${STEPS:-10}
WARM=${WARM:-3}
(rocminfo | grep -E "Marketing Name|gfx|Compute Unit" | head -8; nproc; free -g | head -2) > gpurun_out/env.log 2>&1
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1 || { echo "build failed"; tail -5 gpurun_out/build.log; }
if [ "${SKIP_TESTS:-0}" != "1" ]; then
  : > gpurun_out/pytest_gpu.log
  GROUPS_DEFAULT="maps_ single_voxel spconv_parity stem_conv spconv_golden batchnorm bn_eval nce_parity gather_scatter pdist hardest_loss sgd_step network_features engine_matches trainer_iteration rccl_reducer full_size"
  for grp in ${TEST_GROUPS:-$GROUPS_DEFAULT}; do
    echo "=== group $grp" >> gpurun_out/pytest_gpu.log
    timeout ${TEST_TIMEOUT:-900} python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -k "$grp" >> gpurun_out/pytest_gpu.log 2>&1
    echo "=== group $grp exit: $?" >> gpurun_out/pytest_gpu.log
  done
  grep -E "^=== group .* exit|passed|failed|error" gpurun_out/pytest_gpu.log | tail -40
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
  if [ "${BENCH_AUTOGRAD:-1}" = "1" ]; then
    timeout 600 python bench.py --steps $STEPS --warmup $WARM --engine autograd --no-roofline --no-cpu-baseline > gpurun_out/bench_autograd.log 2>&1
    tail -1 gpurun_out/bench_autograd.log
  fi
fi
if [ "${SKIP_PROF:-0}" != "1" ]; then
  cd /tmp
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/prof" -o bench -- python "$GRAFT_REPO_ROOT/bench.py" --steps 5 --warmup 2 --no-cpu-baseline --no-roofline > "$GRAFT_REPO_ROOT/gpurun_out/prof.log" 2>&1
  echo "prof exit: $?" >> "$GRAFT_REPO_ROOT/gpurun_out/prof.log"
  cd "$GRAFT_REPO_ROOT"
  find gpurun_out/prof -name "*kernel_stats*" | head
  find gpurun_out/prof -name "*kernel_trace*" -size +8M -delete
fi
if [ "${SKIP_PMC:-0}" != "1" ]; then
  # HBM traffic / MFMA counters of the dominant kernels: counters in their own passes, kernel-trace only
  cd /tmp
  for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE"; do
    tag=$(echo $pass | cut -d" " -f1)
    timeout 600 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/pmc_$tag" -o pmc -- python "$GRAFT_REPO_ROOT/scripts/pmc_probe.py" > "$GRAFT_REPO_ROOT/gpurun_out/pmc_$tag.log" 2>&1
    echo "pmc $tag exit: $?" >> "$GRAFT_REPO_ROOT/gpurun_out/pmc_$tag.log"
  done
  cd "$GRAFT_REPO_ROOT"
  find gpurun_out -name "*kernel_trace*" -size +8M -delete
  ls gpurun_out/pmc_* | head -20
fi
echo done
