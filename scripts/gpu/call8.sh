#!/bin/bash
# round 4, GPU call 8: upper bound of a ReLU bit mask for BatchNorm backward (timing diagnostic: the y reads skipped, wrong gradients)
set -u
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$ROOT"
export TMPDIR=/tmp
O=$ROOT/gpurun_out/r04h
mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.txt 2>&1
B="python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-roofline --no-extra"
for r in a b c; do
  timeout 150 $B > $O/step_base_$r.json 2>> $O/ab.err
  PCMI_DIAG_BN_NO_MASK=1 timeout 150 $B > $O/step_nomask_$r.json 2>> $O/ab.err
done
python - <<PY
import json,glob
for f in sorted(glob.glob("$O/*.json")):
  try:
    d=[json.loads(l) for l in open(f).read().splitlines() if l.startswith("{")][-1]; print(f.split('/')[-1], d['value'], d['ms_per_step'])
  except Exception as e: print(f, 'failed', e)
PY
echo done
