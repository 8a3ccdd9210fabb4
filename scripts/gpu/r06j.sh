#!/bin/bash
# Round 6, call 10: what a pre-split operand image could buy the tile-stationary weight-gradient kernel -- wgrad_x3p_kernel with ONE
# producer component compiled out (wrong results; spconv_wgrad_x3.o built with -DPCMI_X3_DIAG_NO_SPLIT / _NO_GATHER / both / + no LDS
# writes, every other object the product's: libpcmi_wg_*.so via PCMI_LIB), stand-alone (scripts/kbench.py, levels 0-1) and in the step.
set -u
ulimit -c 0
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$ROOT"
export TMPDIR=/tmp
TAG=${TAG:-r06j}
O=$ROOT/gpurun_out/$TAG
mkdir -p $O
T0=$(date +%s)
stamp() { echo "[$(( $(date +%s) - T0 )) s] $*" | tee -a $O/stages.log; }
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
stamp "1 stand-alone"
for v in product nosplit nogather nosplit_nogather noprod; do
  if [ $v = product ]; then L=$ROOT/pointcontrast_amd/libpcmi.so; else L=$ROOT/pointcontrast_amd/libpcmi_wg_$v.so; fi
  echo "== $v" | tee -a $O/kbench.txt
  PCMI_LIB=$L KBENCH_LEVELS=0,1 timeout 200 python scripts/kbench.py 2>&1 | grep -E "3\^3 (128|96)" | tee -a $O/kbench.txt
done
stamp "2 in the step"
for i in 1 2; do
  for v in product nosplit nosplit_nogather noprod; do
    if [ $v = product ]; then L=$ROOT/pointcontrast_amd/libpcmi.so; else L=$ROOT/pointcontrast_amd/libpcmi_wg_$v.so; fi
    PCMI_LIB=$L timeout 150 $B > $O/ab_${v}_$i.json 2>> $O/bench.err
    line $O/ab_${v}_$i.json "$v run $i"
  done
done
stamp "done"
