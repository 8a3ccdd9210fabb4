#!/bin/bash
# round 4, GPU call 12: producer / consumer weight gradients with every CU used (row blocks not rounded to 8) -- parity, stand-alone, step
set -u
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$ROOT"
export TMPDIR=/tmp
O=$ROOT/gpurun_out/r04l
mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.txt 2>&1
PCMI_WGRAD_X3P_FILL=1 timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "wgrad_x3t" > $O/gpu_tests_wgrad.txt 2>&1; echo "pytest rc $?" >> $O/gpu_tests_wgrad.txt
tail -3 $O/gpu_tests_wgrad.txt
for m in 0 1; do
  if [ $m = 1 ]; then export PCMI_WGRAD_X3P_FILL=1; else unset PCMI_WGRAD_X3P_FILL; fi
  KBENCH_SUSTAINED=0 KBENCH_LEVELS=0,1 timeout 120 python scripts/kbench.py > $O/kbench_fill$m.txt 2>&1
  echo "== fill $m"; grep -h "^L[01] 3^3 \(96->96\|128->96\)" $O/kbench_fill$m.txt | cut -c100-150
done
B="python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-roofline --no-extra"
for r in a b c; do
  unset PCMI_WGRAD_X3P_FILL
  PCMI_WGRAD_X3P=0 timeout 150 $B > $O/step_onerole_$r.json 2>> $O/ab.err
  timeout 150 $B > $O/step_pc_$r.json 2>> $O/ab.err
  PCMI_WGRAD_X3P_FILL=1 timeout 150 $B > $O/step_pcfill_$r.json 2>> $O/ab.err
done
python - <<PY
import json,glob
for f in sorted(glob.glob("$O/*.json")):
  try:
    d=[json.loads(l) for l in open(f).read().splitlines() if l.startswith("{")][-1]; print(f.split('/')[-1], d['value'], d['ms_per_step'], d['config']['final_loss'])
  except Exception as e: print(f, 'failed', e)
PY
echo done
