#!/bin/bash
# round 4, GPU call 13: how many CUs the producer / consumer weight-gradient kernel should take (row blocks 32 / 24 / 16 of 8 x NG workgroups)
set -u
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$ROOT"
export TMPDIR=/tmp
O=$ROOT/gpurun_out/r04m
mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.txt 2>&1
B="python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-roofline --no-extra"
for r in a b; do
  for rb in 32 24 16; do
    PCMI_WGRAD_X3P_RB=$rb timeout 150 $B > $O/step_rb${rb}_$r.json 2>> $O/ab.err
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
