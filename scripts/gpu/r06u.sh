#!/bin/bash
# Round 6, call 21: spconv16x_kernel multiplying BOTH 16-row groups of a wave whenever either has a neighbour in the step
# (-DPCMI_X3_BOTH_GROUPS: no per-group branches, 143 instead of 165 registers, MFMAs on zeros for half-absent steps) against the
# per-group skipping of the product: stand-alone (all levels), the step, parity of the convolutions.
set -u
ulimit -c 0
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$ROOT"
export TMPDIR=/tmp
O=$ROOT/gpurun_out/${TAG:-r06u}
mkdir -p $O
V="product ${VARIANTS:-both}"
for v in $V; do
  if [ $v = product ]; then L=$ROOT/pointcontrast_amd/libpcmi.so; else L=$ROOT/pointcontrast_amd/libpcmi_$v.so; fi
  echo "== $v" | tee -a $O/kbench.txt
  PCMI_LIB=$L timeout 300 python scripts/kbench.py 2>&1 | grep -E "3\^3" | sed 's/ | wgrad.*//' | tee -a $O/kbench.txt
done
B="python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-roofline --no-extra"
for i in 1 2 3; do
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
for v in $V; do
  if [ $v = product ]; then continue; fi
  PCMI_LIB=$ROOT/pointcontrast_amd/libpcmi_$v.so timeout 600 python -m pytest tests/test_gpu_parity.py -k "conv16 or spconv_parity or streamk" -m gpu -q --tb=short -p no:cacheprovider 2>&1 | tail -2
done
