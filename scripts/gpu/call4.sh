#!/bin/bash
# round 4, GPU call 4: compile-level A/B of the split-precision kernels -- no SLP packing of the operand split (v_pk_add_f32 beside
# MFMAs), weight block requested one column tile earlier -- stand-alone (kbench, levels 0-1) and in the step
set -u
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$ROOT"
export TMPDIR=/tmp
O=$ROOT/gpurun_out/r04d
mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.txt 2>&1
for v in base noslp earlyb earlyb_noslp; do
  L=$ROOT/pointcontrast_amd/libpcmi_$v.so; [ $v = base ] && L=$ROOT/pointcontrast_amd/libpcmi.so
  PCMI_LIB=$L KBENCH_SUSTAINED=0 KBENCH_LEVELS=0,1 timeout 120 python scripts/kbench.py > $O/kbench_$v.txt 2>&1
done
B="python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-roofline --no-extra"
for r in a b; do
  for v in base noslp earlyb earlyb_noslp; do
    L=$ROOT/pointcontrast_amd/libpcmi_$v.so; [ $v = base ] && L=$ROOT/pointcontrast_amd/libpcmi.so
    PCMI_LIB=$L timeout 150 $B > $O/step_${v}_$r.json 2>> $O/ab.err
  done
done
python - <<PY
import json,glob
for f in sorted(glob.glob("$O/*.json")):
  try:
    d=[json.loads(l) for l in open(f).read().splitlines() if l.startswith("{")][-1]; print(f.split('/')[-1], d['value'], d['ms_per_step'])
  except Exception as e: print(f, 'failed', e)
PY
grep -h "^L[01] 3^3 \(96->96\|128->96\)" $O/kbench_*.txt | cut -c1-150
echo done
