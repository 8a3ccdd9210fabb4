#!/bin/bash
# One GPU-box call sized for a SHORT budget (~10 min of box time), most valuable output first; every stage under its own
# timeout and logged under gpurun_out/$TAG (merged back by gpurun even if the call is cut):
#   1 bench.py as the driver runs it (default configuration)          -> bench_line.json
#   2 rocprofv3 --kernel-trace --stats of the same training loop      -> prof/ (kernel_stats CSV)
#   3 split-precision conv kernel: parity tests, A/B timing, whole step with PCMI_CONV16_X3=1
#   4 the GPU tests of the newest code paths, then the rest of the suite in what time is left
# Env: TAG (default r02b), SKIP_X3=1, SKIP_TESTS=1, TEST_TIMEOUT (seconds for stage 4, default 420).
set -u
ulimit -c 0
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
TAG=${TAG:-r02b}
O=gpurun_out/$TAG
mkdir -p $O
T0=$(date +%s)
stamp() { echo "[$(( $(date +%s) - T0 )) s] $*" | tee -a $O/stages.log; }

stamp "build check"
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { echo "build failed"; tail -5 $O/build.log; }

stamp "1 bench (default)"
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --layer-table $O/layer_table.tsv > $O/bench_line.json 2> $O/bench.err
echo "bench exit $?" >> $O/stages.log; cut -c1-400 $O/bench_line.json; echo

stamp "2 rocprofv3 kernel stats"
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/$O/prof" -o bench -- \
    python "$GRAFT_REPO_ROOT/bench.py" --steps 8 --warmup 2 --no-cpu-baseline --no-roofline > "$GRAFT_REPO_ROOT/$O/prof.log" 2>&1 )
echo "prof exit $?" >> $O/stages.log
find $O/prof -name "*kernel_trace*" -size +8M -delete 2>/dev/null
find $O/prof -name "*kernel_stats*" | head -3

if [ "${SKIP_X3:-0}" != "1" ]; then
  stamp "3a x3 parity tests"
  PCMI_X3_REPORT_DIR=$O/x3_err timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=short -p no:cacheprovider -k "x3" > $O/pytest_x3.log 2>&1
  echo "x3 tests exit $?" >> $O/stages.log; tail -4 $O/pytest_x3.log
  stamp "3b x3 A/B timing"
  timeout 200 python scripts/x3_bench.py > $O/x3_bench.txt 2>&1
  echo "x3 bench exit $?" >> $O/stages.log; grep -E "fwd" $O/x3_bench.txt | head -20
  stamp "3c whole step with the split-precision kernel"
  PCMI_CONV16_X3=1 timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > $O/bench_x3_line.json 2>> $O/bench.err
  echo "bench x3 exit $?" >> $O/stages.log; cut -c1-260 $O/bench_x3_line.json; echo
fi

if [ "${SKIP_TESTS:-0}" != "1" ]; then
  stamp "4a newest GPU tests"
  timeout ${TEST_TIMEOUT:-420} python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider --durations=8 \
    -k "joint or conv16_pipelined or wgrad_buffer or trainer_iteration or full_config or engine_matches" > $O/pytest_new.log 2>&1
  echo "new tests exit $?" >> $O/stages.log; tail -4 $O/pytest_new.log
  stamp "4b smoke"
  timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke exit $?" | tee -a $O/stages.log
  stamp "4c the rest of the suite"
  timeout ${TEST_TIMEOUT:-420} python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x \
    -k "not (joint or conv16_pipelined or wgrad_buffer or trainer_iteration or full_config or engine_matches or x3)" > $O/pytest_rest.log 2>&1
  echo "rest exit $?" >> $O/stages.log; tail -4 $O/pytest_rest.log
fi
stamp "done"
