#!/bin/bash
# Round 5, call 13: the stem's weight gradient on the fp32 matrix cores (stem_wgrad_mfma_kernel) and the hardest trainer's
# four gathers through one gradient buffer (GatherManyFunction): parity, A/B, kernel statistics
set -u
ulimit -c 0
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$ROOT"
export TMPDIR=/tmp
TAG=${TAG:-r05l}
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
timeout 420 python -m pytest "tests/test_gpu_parity.py::test_stem_conv_parity" "tests/test_gpu_parity.py::test_gather_many_shares_one_gradient_buffer" \
  "tests/test_gpu_parity.py::test_gather_scatter_rows" "tests/test_gpu_parity.py::test_network_features_loss_and_grads" \
  "tests/test_gpu_parity.py::test_trainer_iteration_matches_oracle" "tests/test_gpu_parity.py::test_hardest_loss_parity" \
  "tests/test_gpu_fullsize.py::test_full_config_step_is_bit_reproducible" "tests/test_gpu_fullsize.py::test_full_config_gradients_match_oracle" \
  "tests/test_gpu_fullsize.py::test_full_config_hardest_trainer_matches_oracle" "tests/test_gpu_trace.py" \
  -m gpu -q --tb=short -p no:cacheprovider --durations=5 > $O/pytest_sel.log 2>&1
echo "pytest(sel) exit $?" | tee -a $O/stages.log; grep -E "passed|failed|skipped" $O/pytest_sel.log | tail -2; grep -E "^FAILED|^ERROR" $O/pytest_sel.log | head -12
stamp "2 A/B"
run stem_pairs 3 PCMI_STEM_WGRAD_MFMA=0
run stem_mfma 3 PCMI_NOP=1
run stem_pairs_b 2 PCMI_STEM_WGRAD_MFMA=0
run stem_mfma_b 2 PCMI_NOP=1
H="python bench.py --loss hardest --steps 30 --warmup 8 --no-cpu-baseline --no-roofline --no-extra"
B="$H --set misc.fused_pair_gather=False"; run hardest_four_buffers 2 PCMI_NOP=1
B="$H"; run hardest_one_buffer 2 PCMI_NOP=1
B="$H --set misc.fused_pair_gather=False"; run hardest_four_buffers_b 2 PCMI_NOP=1
B="$H"; run hardest_one_buffer_b 2 PCMI_NOP=1
stamp "3 rocprofv3 kernel stats"
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/prof" -o bench -- \
    python "$ROOT/bench.py" --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-extra > "$O/prof.log" 2>&1 )
find $O/prof -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
find $O/prof -name "*kernel_trace.csv" -exec cp {} $O/kernel_trace.csv \;
rm -rf $O/prof
grep -E "stem" $O/kernel_stats.csv | cut -c1-150 | head -12
stamp "done"
