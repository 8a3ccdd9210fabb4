#!/bin/bash
# Round 6, call 2: timing prototype of running the level-1 weight gradients beside the NEXT forward pass instead of beside
# the backward chain (PCMI_DEBUG_LATE_WGRAD + PCMI_DEBUG_SKIP_WGRAD_ROWS: wrong gradients, exact schedule), with the
# tile-stationary kernel (one 8-wave workgroup per CU on 224 CUs) and with the pair-list fp32 kernel (small workgroups).
set -u
ulimit -c 0
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$ROOT"
export TMPDIR=/tmp
TAG=${TAG:-r06b}
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
run base 1 PCMI_NOP=1
run no_level1_wgrad 1 PCMI_DEBUG_SKIP_WGRAD_ROWS=65536
run late_x3p 2 PCMI_DEBUG_SKIP_WGRAD_ROWS=65536 PCMI_DEBUG_LATE_WGRAD=65536
run late_fp32 2 PCMI_DEBUG_SKIP_WGRAD_ROWS=65536 PCMI_DEBUG_LATE_WGRAD=65536 PCMI_DEBUG_LATE_FP32=1
run late12_x3p 1 PCMI_DEBUG_SKIP_WGRAD_ROWS=16384 PCMI_DEBUG_LATE_WGRAD=16384
stamp "2 rocprofv3 kernel trace (late_x3p)"
( cd /tmp && PCMI_DEBUG_SKIP_WGRAD_ROWS=65536 PCMI_DEBUG_LATE_WGRAD=65536 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/prof" -o bench -- \
    python "$ROOT/bench.py" --steps 8 --warmup 3 --no-cpu-baseline --no-roofline --no-extra > "$O/prof.log" 2>&1 )
echo "prof exit $?" >> $O/stages.log
find $O/prof -name "*kernel_trace.csv" -exec cp {} $O/kernel_trace.csv \;
rm -rf $O/prof
python scripts/trace_layers.py $O/kernel_trace.csv > $O/trace_layers_late_x3p.txt 2>&1
python scripts/trace_timeline.py $O/kernel_trace.csv --phases > $O/timeline_phases_late_x3p.txt 2>&1
rm -f $O/kernel_trace.csv
stamp "done"
