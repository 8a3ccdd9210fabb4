#!/bin/bash
# round 4, GPU call 3: split-sum fused into the BatchNorm statistics pass -- bit-identity test, BN / network tests, step A/B
set -u
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$ROOT"
export TMPDIR=/tmp
O=$ROOT/gpurun_out/r04c
mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.txt 2>&1
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_refsrc.py tests/test_gpu_trace.py -m gpu -q -x -k "split_sum or batchnorm or network_features or engine or joint_pair or trainer_iteration or refsrc or trace or prepack" > $O/gpu_tests_bn.txt 2>&1; echo "pytest rc $?" >> $O/gpu_tests_bn.txt
tail -4 $O/gpu_tests_bn.txt
B="python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-roofline --no-extra"
for r in a b c; do
  for m in 1 0; do
    PCMI_FUSE_SPLIT_BN=$m timeout 150 $B > $O/step_fuse${m}_$r.json 2>> $O/ab.err
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
