#!/bin/bash
# round 4, GPU call 5: what bounds the two split-precision kernels -- timing with one component removed at a time (wrong results,
# stand-alone kbench on level 0 only), and the back-to-back MFMA order as a real candidate (kbench + step)
set -u
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$ROOT"
export TMPDIR=/tmp
O=$ROOT/gpurun_out/r04e
mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.txt 2>&1
for v in base b2b dg_nogather dg_nodma dg_nogather_nodma dg_nosplit dg_nomfma dg_nobarrier; do
  L=$ROOT/pointcontrast_amd/libpcmi_$v.so; [ $v = base ] && L=$ROOT/pointcontrast_amd/libpcmi.so
  PCMI_LIB=$L KBENCH_SUSTAINED=0 KBENCH_LEVELS=0 timeout 100 python scripts/kbench.py > $O/kbench_$v.txt 2>&1
  echo "== $v"; grep -h "^L0 3^3 \(96->96\|128->96\)" $O/kbench_$v.txt | cut -c1-150
done
B="python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-roofline --no-extra"
for r in a b; do
  for v in base b2b; do
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
echo done
