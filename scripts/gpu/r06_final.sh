#!/bin/bash
# Round 6, the closing GPU call of a build: the driver's bench line (rooflines with in-step times, extras incl. rotating batches and
# the fp32-instruction leg, cpu_baseline), rocprofv3 kernel stats + trace of the same loop, the PMC passes of
# scripts/pmc_probe.py (per probe segment), ten consecutive bench processes, forced-reducer lines, the whole GPU suite + smoke.
set -u
ulimit -c 0
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$ROOT"
export TMPDIR=/tmp
TAG=${TAG:-r06z}
O=$ROOT/gpurun_out/$TAG
mkdir -p $O
T0=$(date +%s)
stamp() { echo "[$(( $(date +%s) - T0 )) s] $*" | tee -a $O/stages.log; }
python -c "import __graft_entry__ as g; g.build()" > $O/build.txt 2>&1
B="python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-roofline --no-extra"
stamp "1 bench as the driver runs it"
timeout 700 python bench.py --layer-table $O/layer_table.tsv > $O/bench_line.json 2> $O/bench.err
echo "bench exit $?" >> $O/stages.log; cut -c1-300 $O/bench_line.json; echo
cp $O/layer_table.tsv.times.tsv $O/layer_table_with_in_step_times.tsv 2>/dev/null
stamp "2 rocprofv3 kernel stats"
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/prof" -o bench -- \
    python "$ROOT/bench.py" --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-extra > "$O/prof.log" 2>&1 )
echo "prof exit $?" >> $O/stages.log
find $O/prof -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
find $O/prof -name "*kernel_trace.csv" -exec cp {} $O/kernel_trace.csv \;
rm -rf $O/prof
stamp "3 PMC passes"
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE" \
            "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
  tag=$(echo $pass | cut -d" " -f1)
  ( cd /tmp && PMC_PROBE_ONLY=96 timeout 200 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d "$O/pmc_$tag" -o pmc -- \
      python "$ROOT/scripts/pmc_probe.py" > "$O/pmc_$tag.log" 2>&1 )
  echo "pmc $tag exit $?" >> $O/stages.log
  find $O/pmc_$tag -name "*kernel_trace*" -delete 2>/dev/null
  find $O/pmc_$tag -name "*counter_collection.csv" -exec cp {} $O/pmc_$tag/pmc_counter_collection.csv \; 2>/dev/null
done
stamp "4 ten consecutive bench processes"
for i in 1 2 3 4 5 6 7 8 9 10; do
  timeout 120 $B 2>> $O/bench.err | grep '^{' | tail -1 > $O/run_$i.json
  python -c "
import json
try:
  d = json.load(open('$O/run_$i.json')); h = d['config'].get('host_phase_ms_per_step', {}); print('run $i |', d['value'], 'pairs/s', d['ms_per_step'], 'ms | enqueue', d['config']['host_enqueue_ms_per_step'], '| forward host', h.get('forward'), 'cpu', h.get('forward_cpu'))
except Exception as e:
  print('run $i failed:', e)" | tee -a $O/runs.txt
done
stamp "5 other lines"
timeout 150 $B --set misc.force_reducer=True 2>> $O/bench.err | grep '^{' | tail -1 > $O/bench_forced_reducer_line.json; cut -c1-160 $O/bench_forced_reducer_line.json; echo
timeout 150 $B --set misc.force_reducer=True 2>> $O/bench.err | grep '^{' | tail -1 > $O/bench_forced_reducer_line_2.json; cut -c1-100 $O/bench_forced_reducer_line_2.json; echo
timeout 150 python bench.py --loss hardest --steps 30 --warmup 8 --no-cpu-baseline --no-roofline --no-extra 2>> $O/bench.err | grep '^{' | tail -1 > $O/bench_hardest_line.json; cut -c1-160 $O/bench_hardest_line.json; echo
stamp "6 GPU test suite + smoke"
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider --durations=8 > $O/pytest_gpu.log 2>&1
echo "pytest exit $?" | tee -a $O/stages.log; grep -E "passed|failed" $O/pytest_gpu.log | tail -2; grep -E "^FAILED|^ERROR" $O/pytest_gpu.log | head
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke exit $?" | tee -a $O/stages.log; tail -1 $O/smoke.log
stamp "done"
