#!/bin/bash
# Full GPU validation of a round: every -m gpu test, smoke, the driver's bench line (roofline + cpu baseline),
# the HardestContrastive and 1 cm bench lines, rocprofv3 kernel stats and the PMC passes of the dominant kernels.
# Env: SKIP_TESTS / SKIP_1CM / SKIP_PMC = 1 to skip parts; TAG (default r02) names the files.
set -u
ulimit -c 0
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
TAG=${TAG:-r02}
O=gpurun_out/$TAG
mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { echo "build failed"; tail -5 $O/build.log; }
if [ "${SKIP_TESTS:-0}" != "1" ]; then
  timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider --durations=10 > $O/pytest_gpu.log 2>&1
  echo "pytest exit: $?" >> $O/pytest_gpu.log
  grep -E "passed|failed|exit|FAILED|Error" $O/pytest_gpu.log | tail -15
  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke exit: $?" >> $O/smoke.log; tail -2 $O/smoke.log
fi
timeout 600 python bench.py --layer-table $O/layer_table.tsv > $O/bench_line.json 2> $O/bench.err; tail -c 600 $O/bench_line.json; echo
timeout 300 python bench.py --loss hardest --steps 20 --warmup 5 --no-roofline --no-cpu-baseline > $O/bench_hardest_line.json 2>> $O/bench.err; cut -c1-220 $O/bench_hardest_line.json
if [ "${SKIP_1CM:-0}" != "1" ]; then
  timeout 900 python bench.py --voxel 0.01 --steps 10 --warmup 3 --no-roofline --no-cpu-baseline > $O/bench_1cm_line.json 2>> $O/bench.err; cut -c1-260 $O/bench_1cm_line.json
fi
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/$O/prof" -o bench -- python "$GRAFT_REPO_ROOT/bench.py" --steps 8 --warmup 2 --no-cpu-baseline --no-roofline > "$GRAFT_REPO_ROOT/$O/prof.log" 2>&1
cd "$GRAFT_REPO_ROOT"
find $O/prof -name "*kernel_trace*" -size +8M -delete
if [ "${SKIP_PMC:-0}" != "1" ]; then
  cd /tmp
  for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE"; do
    tag=$(echo $pass | cut -d" " -f1)
    timeout 400 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d "$GRAFT_REPO_ROOT/$O/pmc_$tag" -o pmc -- python "$GRAFT_REPO_ROOT/scripts/pmc_probe.py" > "$GRAFT_REPO_ROOT/$O/pmc_$tag.log" 2>&1
    echo "pmc $tag exit: $?" >> "$GRAFT_REPO_ROOT/$O/pmc_$tag.log"
  done
  cd "$GRAFT_REPO_ROOT"
  find $O -name "*kernel_trace*" -size +8M -delete
fi
echo done
