#!/bin/bash
# Round 5, call 3: (i) forced reducer on the build without the idle second side stream, (ii) BatchNorm in one launch on the
# coarse levels (bn_small_*_kernel): parity tests, then the step with PCMI_BN_SMALL_ROWS=0 / 768 / 1536, (iii) kernel stats.
set -u
ulimit -c 0
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$ROOT"
export TMPDIR=/tmp
TAG=${TAG:-r05c}
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
  print(sys.argv[2], "|", d["value"], "pairs/s", d["ms_per_step"], "ms | enqueue", c["host_enqueue_ms_per_step"], "| fwd host", h.get("forward"), "cpu", h.get("forward_cpu"),
        "| bwd_step host", h.get("backward_step"), "cpu", h.get("backward_step_cpu"), "|", json.dumps(c.get("collective")) if c.get("collective") else "")
except Exception as e:
  print(sys.argv[2], "failed:", e)
PY
}
run() {  # label n env...
  local label=$1 n=$2; shift 2
  for i in $(seq 1 $n); do
    env "$@" timeout 150 $B > $O/ab_${label}_$i.json 2>> $O/bench.err
    line $O/ab_${label}_$i.json "$label run $i"
  done
}
stamp "1 BatchNorm / network tests"
timeout 900 python -m pytest "tests/test_gpu_parity.py::test_batchnorm_parity" "tests/test_gpu_parity.py::test_batchnorm_one_launch_form_matches_the_three_launch_form" \
   "tests/test_gpu_parity.py::test_batchnorm_backward_lean_statistics_match_the_wide_kernel" tests/test_gpu_bucket_sync.py -m gpu -q --tb=short -p no:cacheprovider > $O/pytest_bn.log 2>&1
echo "pytest(bn) exit $?" | tee -a $O/stages.log; grep -E "passed|failed|skipped" $O/pytest_bn.log | tail -3; grep -E "^FAILED|^ERROR" $O/pytest_bn.log | head -20
stamp "2 forced reducer"
timeout 150 $B --set misc.force_reducer=True > $O/forced_1.json 2>> $O/bench.err; line $O/forced_1.json "forced run 1"
timeout 150 $B --set misc.force_reducer=True > $O/forced_2.json 2>> $O/bench.err; line $O/forced_2.json "forced run 2"
PCMI_RCCL_MAX_CHANNELS=0 timeout 150 $B --set misc.force_reducer=True --set misc.bucket_mb=32 > $O/forced_r04form.json 2>> $O/bench.err; line $O/forced_r04form.json "forced, 5 buckets / RCCL default channels"
stamp "3 small BN A/B"
run small_off 3 PCMI_BN_SMALL_ROWS=0
run small_768 3 PCMI_BN_SMALL_ROWS=768
run small_1536 3 PCMI_NOP=1
stamp "4 whole suite"
timeout 1200 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider --durations=5 --deselect tests/test_gpu_bucket_sync.py > $O/pytest_gpu.log 2>&1
echo "pytest exit $?" | tee -a $O/stages.log; grep -E "passed|failed" $O/pytest_gpu.log | tail -2; grep -E "^FAILED|^ERROR" $O/pytest_gpu.log | head -20
stamp "5 rocprofv3 kernel stats"
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/prof" -o bench -- \
    python "$ROOT/bench.py" --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-extra > "$O/prof.log" 2>&1 )
echo "prof exit $?" >> $O/stages.log
find $O/prof -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
find $O/prof -name "*kernel_trace.csv" -exec cp {} $O/kernel_trace.csv \;
rm -rf $O/prof
head -14 $O/kernel_stats.csv | cut -c1-150
stamp "done"
