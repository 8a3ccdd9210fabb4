#!/bin/bash
# Round 6, call 9: threshold sweep on the round-6 build (the kernels around these thresholds changed this round: relu_bits made the
# lean statistics kernel 45 registers, the offset split changed the coarse launches): two processes per setting, alternating order.
set -u
ulimit -c 0
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$ROOT"
export TMPDIR=/tmp
TAG=${TAG:-r06i}
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
  d = json.loads(txt[-1]); c = d["config"]
  print("%-34s | %s pairs/s %s ms | loss %s" % (sys.argv[2], d["value"], d["ms_per_step"], c["final_loss"]))
except Exception as e:
  print(sys.argv[2], "failed:", e)
PY
}
run1() {  # label idx (env via ENVV)
  local label=$1 i=$2; shift 2
  env $ENVV timeout 150 $B "$@" > $O/ab_${label}_$i.json 2>> $O/bench.err
  line $O/ab_${label}_$i.json "$label run $i"
}
stamp "sweep"
for i in 1 2; do
  ENVV="PCMI_NOP=1" run1 base $i
  ENVV="PCMI_BN_LEAN_ROWS=32768" run1 lean_rows_32768 $i
  ENVV="PCMI_BN_LEAN_ROWS=16384" run1 lean_rows_16384 $i
  ENVV="PCMI_BN_SMALL_BWD_ROWS=1536" run1 bn_small_bwd_1536 $i
  ENVV="PCMI_BN_SMALL_ROWS=768" run1 bn_small_fwd_768 $i
  ENVV="PCMI_WGRAD_X3T=16384" run1 wgrad_x3t_16384 $i
  ENVV="PCMI_WGRAD_X3T=4096" run1 wgrad_x3t_4096 $i
  ENVV="PCMI_SPCONV_STREAMK=128" run1 streamk_128 $i
  ENVV="PCMI_CONV32R=0" run1 conv32r_off $i
  ENVV="PCMI_CONV16=2048" run1 conv16_2048 $i
  ENVV="PCMI_THROTTLE_SLEEP_US=20" run1 sleep_20us $i
  ENVV="GPU_MAX_HW_QUEUES=4" run1 hwq_4 $i
  ENVV="PCMI_NOP=1" run1 base_again $i
done
stamp "done"
