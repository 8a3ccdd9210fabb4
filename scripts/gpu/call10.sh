#!/bin/bash
# round 4, GPU call 10: producer / consumer weight-gradient kernel -- parity (both forms), stand-alone A/B, step A/B
set -u
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$ROOT"
export TMPDIR=/tmp
O=$ROOT/gpurun_out/r04j
mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.txt 2>&1
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "wgrad_x3t" > $O/gpu_tests_wgrad.txt 2>&1; echo "pytest rc $?" >> $O/gpu_tests_wgrad.txt
tail -4 $O/gpu_tests_wgrad.txt
for m in 0 1; do
  PCMI_WGRAD_X3P=$m KBENCH_SUSTAINED=0 KBENCH_LEVELS=0,1 timeout 120 python scripts/kbench.py > $O/kbench_x3p$m.txt 2>&1
  echo "== x3p $m"; grep -h "^L[01] 3^3 \(96->96\|128->96\)" $O/kbench_x3p$m.txt | cut -c1-150
done
B="python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-roofline --no-extra"
for r in a b; do
  for m in 0 1; do
    PCMI_WGRAD_X3P=$m timeout 150 $B > $O/step_x3p${m}_$r.json 2>> $O/ab.err
  done
done
python - <<PY
import json,glob
for f in sorted(glob.glob("$O/*.json")):
  try:
    d=[json.loads(l) for l in open(f).read().splitlines() if l.startswith("{")][-1]; print(f.split('/')[-1], d['value'], d['ms_per_step'], d['config']['final_loss'])
  except Exception as e: print(f, 'failed', e)
PY
echo done
