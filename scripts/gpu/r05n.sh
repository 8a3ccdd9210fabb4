#!/bin/bash
# Round 5, call 17: the forward-orientation weight packs of a training pass on the side stream (PCMI_X3_PACK_FWD_SIDE),
# joined in front of the first layer that reads one: parity, A/B
set -u
ulimit -c 0
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$ROOT"
export TMPDIR=/tmp
TAG=${TAG:-r05n}
O=$ROOT/gpurun_out/$TAG
mkdir -p $O
T0=$(date +%s)
stamp() { echo "[$(( $(date +%s) - T0 )) s] $*" | tee -a $O/stages.log; }
python -c "import __graft_entry__ as g; g.build()" > $O/build.txt 2>&1
B="python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-roofline --no-extra"
line() {
  python - "$1" "$2" <<'PY' | tee -a $O/ab.txt
import json, sys
try:
  txt = [l for l in open(sys.argv[1]) if l.startswith("{")]
  d = json.loads(txt[-1]); c = d["config"]
  print(sys.argv[2], "|", d["value"], "pairs/s", d["ms_per_step"], "ms | loss", c["final_loss"])
except Exception as e:
  print(sys.argv[2], "failed:", e)
PY
}
run() {
  local label=$1 n=$2; shift 2
  for i in $(seq 1 $n); do
    env "$@" timeout 150 $B > $O/ab_${label}_$i.json 2>> $O/bench.err
    line $O/ab_${label}_$i.json "$label run $i"
  done
}
export OMP_NUM_THREADS=32
stamp "1 tests"
timeout 300 python -m pytest "tests/test_gpu_parity.py::test_engine_prepacked_weights_are_bit_identical" "tests/test_gpu_parity.py::test_engine_prepack_follows_each_pass_across_size_classes" \
  "tests/test_gpu_parity.py::test_engine_matches_autograd_path" "tests/test_gpu_parity.py::test_joint_pair_pass_matches_two_passes" \
  "tests/test_gpu_fullsize.py::test_full_config_step_is_bit_reproducible" "tests/test_gpu_bucket_sync.py::test_bucket_consumer_sees_the_final_gradients" "tests/test_gpu_trace.py" \
  -m gpu -q --tb=short -p no:cacheprovider --durations=5 > $O/pytest_sel.log 2>&1
echo "pytest(sel) exit $?" | tee -a $O/stages.log; grep -E "passed|failed|skipped" $O/pytest_sel.log | tail -2; grep -E "^FAILED|^ERROR" $O/pytest_sel.log | head -12
stamp "2 A/B"
run pack_on_chain 3 PCMI_X3_PACK_FWD_SIDE=0
run pack_on_side 3 PCMI_NOP=1
run pack_on_chain_b 2 PCMI_X3_PACK_FWD_SIDE=0
run pack_on_side_b 2 PCMI_NOP=1
stamp "3 rocprofv3 kernel stats"
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/prof" -o bench -- \
    python "$ROOT/bench.py" --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-extra > "$O/prof.log" 2>&1 )
find $O/prof -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
find $O/prof -name "*kernel_trace.csv" -exec cp {} $O/kernel_trace.csv \;
rm -rf $O/prof
grep -E "x3_pack" $O/kernel_stats.csv | cut -c1-150 | head -12
stamp "done"
