#!/bin/bash
# Round 5, call 8: BatchNorm backward with the two segments side by side (parameter gradients accumulated on the side
# stream): tests, A/B against the sequential form (PCMI_BN_SMALL_PAR=0), kernel statistics, forced reducer.
set -u
ulimit -c 0
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$ROOT"
export TMPDIR=/tmp
TAG=${TAG:-r05h}
O=$ROOT/gpurun_out/$TAG
mkdir -p $O
T0=$(date +%s)
stamp() { echo "[$(( $(date +%s) - T0 )) s] $*" | tee -a $O/stages.log; }
python -c "import __graft_entry__ as g; g.build()" > $O/build.txt 2>&1
B="python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-roofline --no-extra"
line() {  # file label
  python - "$1" "$2" <<'PY' | tee -a $O/ab.txt
import json, sys
try:
  txt = [l for l in open(sys.argv[1]) if l.startswith("{")]
  d = json.loads(txt[-1]); c = d["config"]; h = c.get("host_phase_ms_per_step", {})
  print(sys.argv[2], "|", d["value"], "pairs/s", d["ms_per_step"], "ms | loss", c["final_loss"], "| enqueue", c["host_enqueue_ms_per_step"], "| bwd_step host", h.get("backward_step"), "cpu", h.get("backward_step_cpu"), "|", json.dumps(c.get("collective"))[:200] if c.get("collective") else "")
except Exception as e:
  print(sys.argv[2], "failed:", e)
PY
}
run() {  # label n env... [-- bench args]
  local label=$1 n=$2; shift 2
  local envs=() args=()
  while [ $# -gt 0 ] && [ "$1" != "--" ]; do envs+=("$1"); shift; done
  [ $# -gt 0 ] && shift
  args=("$@")
  for i in $(seq 1 $n); do
    env "${envs[@]}" timeout 150 $B "${args[@]}" > $O/ab_${label}_$i.json 2>> $O/bench.err
    line $O/ab_${label}_$i.json "$label run $i"
  done
}
stamp "1 tests"
timeout 900 python -m pytest "tests/test_gpu_parity.py::test_network_features_loss_and_grads" "tests/test_gpu_parity.py::test_joint_pair_pass_matches_two_passes" \
  "tests/test_gpu_parity.py::test_trainer_iteration_matches_oracle" tests/test_gpu_bucket_sync.py tests/test_gpu_trace.py \
  "tests/test_gpu_fullsize.py::test_full_config_gradients_match_oracle" "tests/test_gpu_fullsize.py::test_full_config_step_is_bit_reproducible" \
  -m gpu -q --tb=short -p no:cacheprovider > $O/pytest_sel.log 2>&1
echo "pytest(sel) exit $?" | tee -a $O/stages.log; grep -E "passed|failed|skipped" $O/pytest_sel.log | tail -3; grep -E "^FAILED|^ERROR" $O/pytest_sel.log | head -20
stamp "2 A/B"
run par_off 3 PCMI_BN_SMALL_PAR=0
run par_on 3 PCMI_NOP=1
run par_on_768 2 PCMI_BN_SMALL_ROWS=768
run forced 2 PCMI_NOP=1 -- --set misc.force_reducer=True
stamp "3 rocprofv3 kernel stats"
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/prof" -o bench -- \
    python "$ROOT/bench.py" --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-extra > "$O/prof.log" 2>&1 )
echo "prof exit $?" >> $O/stages.log
find $O/prof -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
find $O/prof -name "*kernel_trace.csv" -exec cp {} $O/kernel_trace.csv \;
rm -rf $O/prof
grep -E "bn_small|bn_param" $O/kernel_stats.csv | cut -c1-150
stamp "done"
