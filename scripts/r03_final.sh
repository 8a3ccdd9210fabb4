#!/bin/bash
# Round 3, final GPU call: the driver's bench line (roofline + cpu_baseline), rocprofv3 kernel stats of the same loop,
# the PMC passes of scripts/pmc_probe.py (HBM traffic, MFMA activity, LDS conflicts), 10 consecutive bench processes,
# the other configurations, the whole GPU suite + smoke.  Env: TAG, SKIP_TESTS=1, SKIP_PMC=1.
set -u
ulimit -c 0
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
TAG=${TAG:-r03f}
O=gpurun_out/$TAG
mkdir -p $O
T0=$(date +%s)
stamp() { echo "[$(( $(date +%s) - T0 )) s] $*" | tee -a $O/stages.log; }
B="python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-roofline"
stamp "1 bench as the driver runs it"
timeout 400 python bench.py --layer-table $O/layer_table.tsv > $O/bench_line.json 2> $O/bench.err
echo "bench exit $?" >> $O/stages.log; cut -c1-400 $O/bench_line.json; echo
stamp "2 rocprofv3 kernel stats"
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/$O/prof" -o bench -- \
    python "$GRAFT_REPO_ROOT/bench.py" --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > "$GRAFT_REPO_ROOT/$O/prof.log" 2>&1 )
echo "prof exit $?" >> $O/stages.log
find $O/prof -name "*kernel_trace*" -size +8M -delete 2>/dev/null
stamp "3 PMC passes"
[ "${SKIP_PMC:-0}" = "1" ] || for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE" \
            "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
  tag=$(echo $pass | cut -d" " -f1)
  ( cd /tmp && PMC_PROBE_ONLY=96 timeout 150 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d "$GRAFT_REPO_ROOT/$O/pmc_$tag" -o pmc -- \
      python "$GRAFT_REPO_ROOT/scripts/pmc_probe.py" > "$GRAFT_REPO_ROOT/$O/pmc_$tag.log" 2>&1 )
  echo "pmc $tag exit $?" >> $O/stages.log
done
find $O -name "*kernel_trace*" -size +8M -delete 2>/dev/null
stamp "4 ten consecutive bench processes"
for i in 1 2 3 4 5 6 7 8 9 10; do
  timeout 120 $B 2>> $O/bench.err | tail -1 > $O/run_$i.json
  python -c "
import sys, json
try:
  d = json.load(open('$O/run_$i.json')); h = d['config'].get('host_phase_ms_per_step', {})
  print('run $i |', d['value'], 'pairs/s', d['ms_per_step'], 'ms | enqueue', d['config']['host_enqueue_ms_per_step'], '|', {k: v for k, v in h.items() if not k.endswith('_cpu')})
except Exception as e:
  print('run $i failed:', e)" | tee -a $O/runs.txt
done
stamp "5 the other configurations"
timeout 120 $B --loss hardest > $O/bench_hardest_line.json 2>> $O/bench.err; cut -c1-200 $O/bench_hardest_line.json; echo
timeout 200 python bench.py --voxel 0.01 --steps 10 --warmup 3 --no-roofline --no-cpu-baseline > $O/bench_1cm_line.json 2>> $O/bench.err; cut -c1-220 $O/bench_1cm_line.json; echo
PCMI_CONV16_X3=0 PCMI_WGRAD_X3T=0 timeout 120 $B > $O/bench_fp32_line.json 2>> $O/bench.err; cut -c1-200 $O/bench_fp32_line.json; echo
if [ "${SKIP_TESTS:-0}" != "1" ]; then
  stamp "6 GPU test suite + smoke"
  timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider --durations=8 -rP > $O/pytest_gpu.log 2>&1
  echo "pytest exit $?" | tee -a $O/stages.log; grep -E "passed|failed" $O/pytest_gpu.log | tail -2; grep -E "^FAILED|^ERROR" $O/pytest_gpu.log | head
  timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke exit $?" | tee -a $O/stages.log; tail -1 $O/smoke.log
fi
stamp "done"
