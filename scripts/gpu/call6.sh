#!/bin/bash
# round 4, GPU call 6: deep operand ring (both operands two steps ahead, counted waits) -- stand-alone A/B first, parity, step A/B
set -u
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$ROOT"
export TMPDIR=/tmp
O=$ROOT/gpurun_out/r04f
mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.txt 2>&1
for m in 0 3; do
  PCMI_X3_RING=$m KBENCH_SUSTAINED=0 timeout 200 python scripts/kbench.py > $O/kbench_ring$m.txt 2>&1
  echo "== ring mask $m"; grep -h "^L[0-4] 3^3" $O/kbench_ring$m.txt | cut -c1-120
done
B="python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-roofline --no-extra"
for r in a b; do
  for m in 0 3 1 2; do
    PCMI_X3_RING=$m timeout 150 $B > $O/step_ring${m}_$r.json 2>> $O/ab.err
  done
done
python - <<PY
import json,glob
for f in sorted(glob.glob("$O/*.json")):
  try:
    d=[json.loads(l) for l in open(f).read().splitlines() if l.startswith("{")][-1]; print(f.split('/')[-1], d['value'], d['ms_per_step'], d['config']['final_loss'])
  except Exception as e: print(f, 'failed', e)
PY
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "spconv_parity or conv16_x3 or streamk or prepack or test_network_features_loss_and_grads" > $O/gpu_tests_conv.txt 2>&1; echo "pytest rc $?" >> $O/gpu_tests_conv.txt
tail -3 $O/gpu_tests_conv.txt
echo done
