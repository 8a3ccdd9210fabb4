#!/bin/bash
# Round 5, call 14: the stem's weight gradient on the chain instead of the side stream (PCMI_STEM_WGRAD_ON_CHAIN), its slab count
# (PCMI_STEM_SLABS) and the 32-lane slab reduction: parity, A/B
set -u
ulimit -c 0
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$ROOT"
export TMPDIR=/tmp
TAG=${TAG:-r05m}
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
timeout 300 python -m pytest "tests/test_gpu_parity.py::test_stem_conv_parity" "tests/test_gpu_parity.py::test_engine_matches_autograd_path" \
  "tests/test_gpu_fullsize.py::test_full_config_step_is_bit_reproducible" "tests/test_gpu_bucket_sync.py::test_bucket_consumer_sees_the_final_gradients" \
  -m gpu -q --tb=short -p no:cacheprovider --durations=5 > $O/pytest_sel.log 2>&1
echo "pytest(sel) exit $?" | tee -a $O/stages.log; grep -E "passed|failed|skipped" $O/pytest_sel.log | tail -2; grep -E "^FAILED|^ERROR" $O/pytest_sel.log | head -12
stamp "2 A/B"
run side_stream 3 PCMI_STEM_WGRAD_ON_CHAIN=0
run on_chain 3 PCMI_NOP=1
run on_chain_256_slabs 2 PCMI_STEM_SLABS=256
run on_chain_384_slabs 2 PCMI_STEM_SLABS=384
run side_stream_b 2 PCMI_STEM_WGRAD_ON_CHAIN=0
run on_chain_b 2 PCMI_NOP=1
run on_chain_256_slabs_b 2 PCMI_STEM_SLABS=256
stamp "3 rocprofv3 kernel stats"
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/prof" -o bench -- \
    python "$ROOT/bench.py" --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-extra > "$O/prof.log" 2>&1 )
find $O/prof -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
find $O/prof -name "*kernel_trace.csv" -exec cp {} $O/kernel_trace.csv \;
rm -rf $O/prof
grep -E "stem" $O/kernel_stats.csv | cut -c1-150 | head -12
stamp "done"
