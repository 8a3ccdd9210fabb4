#!/bin/bash
# Round 5, call 10: the dense 1x1 weight gradients on the split-precision tile kernel (K = 1): parity, A/B, kernel stats
set -u
ulimit -c 0
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$ROOT"
export TMPDIR=/tmp
TAG=${TAG:-r05j}
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
stamp "1 tests"
timeout 600 python -m pytest "tests/test_gpu_parity.py::test_dense_1x1_weight_gradient_on_the_split_kernel" "tests/test_gpu_parity.py::test_wgrad_x3t_split_precision_matches_fp64" \
  "tests/test_gpu_parity.py::test_spconv_parity" "tests/test_gpu_fullsize.py::test_full_config_gradients_match_oracle" -m gpu -q --tb=short -p no:cacheprovider -s > $O/pytest_sel.log 2>&1
echo "pytest(sel) exit $?" | tee -a $O/stages.log; grep -E "passed|failed|skipped" $O/pytest_sel.log | tail -2; grep -E "^FAILED|^ERROR|dense wgrad" $O/pytest_sel.log | head -12
stamp "2 A/B"
run dense_off 3 PCMI_WGRAD_X3T_DENSE=0
run dense_on 3 PCMI_NOP=1
run dense_off_b 2 PCMI_WGRAD_X3T_DENSE=0
run dense_on_b 2 PCMI_NOP=1
stamp "3 rocprofv3 kernel stats"
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/prof" -o bench -- \
    python "$ROOT/bench.py" --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-extra > "$O/prof.log" 2>&1 )
find $O/prof -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
rm -rf $O/prof
grep -E "wgrad" $O/kernel_stats.csv | cut -c1-150 | head -12
stamp "done"
