#!/bin/bash
# End-of-round validation of the default configuration in ONE short GPU-box call (~5.5 min), most valuable first:
#   1 bench.py as the driver runs it (roofline included)               -> bench_line.json
#   2 rocprofv3 --kernel-trace --stats of the same training loop       -> prof/
#   3 the whole GPU test suite + smoke
#   4 rocprofv3 --pmc passes of scripts/pmc_probe.py (HBM traffic, MFMA activity of the dominant kernels)
#   5 A/B lines (per-launch weight packing, fp32-MFMA convolutions), HardestContrastive and 1 cm bench lines
# Env: TAG (default r02e), SKIP_TESTS=1, SKIP_PMC=1.
set -u
ulimit -c 0
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
TAG=${TAG:-r02e}
O=gpurun_out/$TAG
mkdir -p $O
T0=$(date +%s)
stamp() { echo "[$(( $(date +%s) - T0 )) s] $*" | tee -a $O/stages.log; }
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline"

stamp "1 bench (default)"
timeout 300 $B --layer-table $O/layer_table.tsv > $O/bench_line.json 2> $O/bench.err
echo "bench exit $?" >> $O/stages.log; cut -c1-330 $O/bench_line.json; echo

stamp "2 rocprofv3 kernel stats"
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/$O/prof" -o bench -- \
    python "$GRAFT_REPO_ROOT/bench.py" --steps 8 --warmup 2 --no-cpu-baseline --no-roofline > "$GRAFT_REPO_ROOT/$O/prof.log" 2>&1 )
echo "prof exit $?" >> $O/stages.log
find $O/prof -name "*kernel_trace*" -size +8M -delete 2>/dev/null

if [ "${SKIP_TESTS:-0}" != "1" ]; then
  stamp "3 GPU test suite"
  PCMI_X3_REPORT_DIR=$O/x3_err timeout 420 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider --durations=6 > $O/pytest_gpu.log 2>&1
  echo "pytest exit $?" | tee -a $O/stages.log; tail -3 $O/pytest_gpu.log
  timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke exit $?" | tee -a $O/stages.log
fi

if [ "${SKIP_PMC:-0}" != "1" ]; then
  stamp "4 PMC passes"
  for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE"; do
    tag=$(echo $pass | cut -d" " -f1)
    ( cd /tmp && PMC_PROBE_ONLY=96 timeout 120 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d "$GRAFT_REPO_ROOT/$O/pmc_$tag" -o pmc -- \
        python "$GRAFT_REPO_ROOT/scripts/pmc_probe.py" > "$GRAFT_REPO_ROOT/$O/pmc_$tag.log" 2>&1 )
    echo "pmc $tag exit $?" >> $O/stages.log
  done
  find $O -name "*kernel_trace*" -size +8M -delete 2>/dev/null
fi

stamp "5 A/B and the other configurations"
PCMI_X3_PREPACK=0 timeout 100 $B --no-roofline > $O/bench_noprepack_line.json 2>> $O/bench.err; cut -c1-200 $O/bench_noprepack_line.json; echo
PCMI_CONV16_X3=0 timeout 100 $B --no-roofline > $O/bench_fp32_line.json 2>> $O/bench.err; cut -c1-200 $O/bench_fp32_line.json; echo
timeout 100 $B --no-roofline > $O/bench_repeat_line.json 2>> $O/bench.err; cut -c1-200 $O/bench_repeat_line.json; echo
timeout 100 $B --no-roofline --loss hardest > $O/bench_hardest_line.json 2>> $O/bench.err; cut -c1-200 $O/bench_hardest_line.json; echo
timeout 150 python bench.py --voxel 0.01 --steps 10 --warmup 3 --no-roofline --no-cpu-baseline > $O/bench_1cm_line.json 2>> $O/bench.err; cut -c1-220 $O/bench_1cm_line.json; echo
stamp "done"
