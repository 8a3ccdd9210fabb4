#!/bin/bash
# Round 6, call 11: what the CONSUMER side of wgrad_x3p_kernel costs by component -- producers compiled out (no split, no gather
# traffic, no LDS writes: wrong results) and on top of that no fragment reads / no MFMAs / neither / no per-slot barrier / the six
# products of an accumulator back to back; stand-alone (scripts/kbench.py, level 0), then the b2b order in the step.
set -u
ulimit -c 0
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$ROOT"
export TMPDIR=/tmp
TAG=${TAG:-r06k}
O=$ROOT/gpurun_out/$TAG
mkdir -p $O
B="python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-roofline --no-extra"
for v in product b2b noprod noprod_b2b noprod_nofrag noprod_nomfma noprod_nofrag_nomfma noprod_nobarrier noprod_nofrag_nobarrier; do
  if [ $v = product ]; then L=$ROOT/pointcontrast_amd/libpcmi.so; else L=$ROOT/pointcontrast_amd/libpcmi_wg_$v.so; fi
  echo "== $v" | tee -a $O/kbench.txt
  PCMI_LIB=$L KBENCH_LEVELS=0,1 timeout 200 python scripts/kbench.py 2>&1 | grep -E "3\^3 (128|96)" | sed 's/.*| wgrad/wgrad/' | tee -a $O/kbench.txt
done
for i in 1 2; do
  for v in product b2b; do
    if [ $v = product ]; then L=$ROOT/pointcontrast_amd/libpcmi.so; else L=$ROOT/pointcontrast_amd/libpcmi_wg_$v.so; fi
    PCMI_LIB=$L timeout 150 $B > $O/ab_${v}_$i.json 2>> $O/bench.err
    python - $O/ab_${v}_$i.json "$v run $i" <<'PY' | tee -a $O/ab.txt
import json, sys
try:
  d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1]); print(sys.argv[2], "|", d["value"], "pairs/s", d["ms_per_step"], "ms | loss", d["config"]["final_loss"])
except Exception as e:
  print(sys.argv[2], "failed:", e)
PY
  done
done
