#!/bin/bash
# Round 6, call 1: (i) what the weight-gradient stream costs the backward chain -- the step with NO weight gradients, and with
# only the level-1 ones (>= 65536 rows) left out (timing diagnostics, wrong gradients): the bound of running those outside
# the backward pass; (ii) the unit-balanced launch on the levels that split their offsets over blockIdx.z (PCMI_SK_MID);
# (iii) a raw kernel trace of the product step for the per-layer timeline (scripts/trace_layers.py).
set -u
ulimit -c 0
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$ROOT"
export TMPDIR=/tmp
TAG=${TAG:-r06a}
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
  print(sys.argv[2], "|", d["value"], "pairs/s", d["ms_per_step"], "ms | loss", c["final_loss"], "| enqueue", c["host_enqueue_ms_per_step"])
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
stamp "1 A/B"
run base 2 PCMI_NOP=1
run no_wgrad_at_all 2 PCMI_DEBUG_SKIP_WGRAD_ROWS=1
run no_level1_wgrad 2 PCMI_DEBUG_SKIP_WGRAD_ROWS=65536
run no_level12_wgrad 1 PCMI_DEBUG_SKIP_WGRAD_ROWS=16384
run sk_mid_256 2 PCMI_SK_MID=256
run sk_mid_64 2 PCMI_SK_MID=64
run sk_mid_64_s8 1 PCMI_SK_MID=64 PCMI_SK_MID_STEPS=8
run sk_mid_64_s32 1 PCMI_SK_MID=64 PCMI_SK_MID_STEPS=32
stamp "2 parity of the SK-mid launch"
PCMI_SK_MID=64 timeout 600 python -m pytest "tests/test_gpu_parity.py::test_network_features_loss_and_grads" "tests/test_gpu_fullsize.py::test_full_config_gradients_match_oracle" \
  "tests/test_gpu_parity.py::test_full_size_streamk_matches_plain" -m gpu -q --tb=short -p no:cacheprovider > $O/pytest_sel.log 2>&1
echo "pytest(sel) exit $?" | tee -a $O/stages.log; grep -E "passed|failed|skipped" $O/pytest_sel.log | tail -3; grep -E "^FAILED|^ERROR" $O/pytest_sel.log | head
stamp "3 rocprofv3 kernel trace (product)"
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/prof" -o bench -- \
    python "$ROOT/bench.py" --steps 8 --warmup 3 --no-cpu-baseline --no-roofline --no-extra > "$O/prof.log" 2>&1 )
echo "prof exit $?" >> $O/stages.log
find $O/prof -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
find $O/prof -name "*kernel_trace.csv" -exec cp {} $O/kernel_trace.csv \;
rm -rf $O/prof
python scripts/trace_layers.py $O/kernel_trace.csv > $O/trace_layers.txt 2>&1
python scripts/trace_timeline.py $O/kernel_trace.csv --phases > $O/timeline_phases.txt 2>&1
# keep the merged output small: one step of the trace is enough
python - <<PY
import csv
rows = list(csv.DictReader(open("$O/kernel_trace.csv")))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
sgd = [i for i, r in enumerate(rows) if "sgd_kernel" in r["Kernel_Name"]]
if len(sgd) >= 6:
  rows = rows[sgd[3] : sgd[5] + 1]
with open("$O/kernel_trace_2steps.csv", "w", newline="") as f:
  w = csv.DictWriter(f, fieldnames=list(rows[0].keys())); w.writeheader(); w.writerows(rows)
PY
rm -f $O/kernel_trace.csv
stamp "done"
