#!/bin/bash
# Round 6, call 18: whole-library compiler-flag variants against the product build (PCMI_LIB), stand-alone times of the level-1 / 2
# convolutions and the step: VARIANTS="ilp memclause nopostsched novec" (-mllvm -amdgpu-sched-strategy=max-ilp / max-memory-clause,
# -mllvm -enable-post-misched=false, -fno-vectorize).
set -u
ulimit -c 0
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$ROOT"
export TMPDIR=/tmp
O=$ROOT/gpurun_out/${TAG:-r06r}
mkdir -p $O
V="product ${VARIANTS:-ilp memclause nopostsched novec}"
for v in $V; do
  if [ $v = product ]; then L=$ROOT/pointcontrast_amd/libpcmi.so; else L=$ROOT/pointcontrast_amd/libpcmi_$v.so; fi
  echo "== $v" | tee -a $O/kbench.txt
  PCMI_LIB=$L KBENCH_LEVELS=0,1 timeout 200 python scripts/kbench.py 2>&1 | grep -E "3\^3 (128|96)" | tee -a $O/kbench.txt
done
B="python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-roofline --no-extra"
for i in 1 2; do
  for v in $V; do
    if [ $v = product ]; then L=$ROOT/pointcontrast_amd/libpcmi.so; else L=$ROOT/pointcontrast_amd/libpcmi_$v.so; fi
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
