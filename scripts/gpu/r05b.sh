#!/bin/bash
# Round 5, call 2: (i) why the forced 1-rank reducer run took 45 ms/step in call 1 (full JSON this time; with the
# host-wait changes switched off), (ii) the sleeping run-ahead wait against the spinning one, (iii) wgrad_x3p_kernel with
# the conversion of absent slots skipped against round 4's form (variant build libpcmi_x3p_convert_absent.so): kernel
# timings (scripts/kbench.py) and the step, (iv) the tests that cover both, (v) kernel statistics of the step.
set -u
ulimit -c 0
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$ROOT"
export TMPDIR=/tmp
TAG=${TAG:-r05b}
O=$ROOT/gpurun_out/$TAG
mkdir -p $O
T0=$(date +%s)
stamp() { echo "[$(( $(date +%s) - T0 )) s] $*" | tee -a $O/stages.log; }
python -c "import __graft_entry__ as g; g.build()" > $O/build.txt 2>&1
OLD=$ROOT/pointcontrast_amd/libpcmi_x3p_convert_absent.so
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
stamp "1 tests of the touched kernels"
timeout 600 python -m pytest tests/test_gpu_bucket_sync.py "tests/test_gpu_parity.py::test_wgrad_x3t_split_precision_matches_fp64" "tests/test_gpu_parity.py::test_spconv_parity" "tests/test_gpu_parity.py::test_batchnorm_backward_lean_statistics_match_the_wide_kernel" tests/test_gpu_fullsize.py -m gpu -q --tb=short -p no:cacheprovider > $O/pytest_sel.log 2>&1
echo "pytest(sel) exit $?" | tee -a $O/stages.log; grep -E "passed|failed|skipped" $O/pytest_sel.log | tail -3; grep -E "^FAILED|^ERROR" $O/pytest_sel.log | head
stamp "2 forced reducer"
for i in 1 2; do
  timeout 150 $B --set misc.force_reducer=True > $O/forced_$i.json 2>> $O/bench.err; line $O/forced_$i.json "forced run $i"
done
PCMI_BLOCKING_EVENTS=0 PCMI_THROTTLE_SLEEP_US=0 timeout 150 $B --set misc.force_reducer=True > $O/forced_spin.json 2>> $O/bench.err; line $O/forced_spin.json "forced, spinning waits / plain events"
PCMI_RCCL_MAX_CHANNELS=0 timeout 150 $B --set misc.force_reducer=True --set misc.bucket_mb=32 > $O/forced_r04form.json 2>> $O/bench.err; line $O/forced_r04form.json "forced, 5 buckets / RCCL default channels"
NCCL_DEBUG=WARN timeout 150 $B --set misc.force_reducer=True --set misc.reducer_profile=False > $O/forced_noprofile.json 2>> $O/bench.err; line $O/forced_noprofile.json "forced, no reducer profile events"
stamp "3 host wait A/B"
run sleep50 3 PCMI_NOP=1
run spin 3 PCMI_THROTTLE_SLEEP_US=0 PCMI_BLOCKING_EVENTS=0
stamp "4 x3p absent-slot conversion: kernels"
for v in new old; do
  L=""; [ $v = old ] && L=$OLD
  echo "== $v" >> $O/kbench.txt
  PCMI_LIB=$L KBENCH_LEVELS=0,1 KBENCH_SUSTAINED=0 PYTHONPATH=. timeout 200 python scripts/kbench.py 2>/dev/null | grep "3^3" >> $O/kbench.txt
done
cat $O/kbench.txt
stamp "5 x3p absent-slot conversion: step"
run x3p_old 3 PCMI_LIB=$OLD
run x3p_new 3 PCMI_NOP=1
stamp "6 rocprofv3 kernel stats"
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/prof" -o bench -- \
    python "$ROOT/bench.py" --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-extra > "$O/prof.log" 2>&1 )
echo "prof exit $?" >> $O/stages.log
find $O/prof -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
find $O/prof -name "*kernel_trace.csv" -exec cp {} $O/kernel_trace.csv \;
rm -rf $O/prof
head -12 $O/kernel_stats.csv | cut -c1-150
stamp "done"
