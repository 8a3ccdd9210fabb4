#!/bin/bash
mkdir -p gpurun_out/r04p; O=gpurun_out/r04p; rm -f $O/bn2.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "bn or batchnorm or norm or engine or network or trainer_iteration" 2>&1 | tail -5 > $O/pytest_bn.log
for i in 1 2 3; do
  timeout 200 python bench.py --steps 30 --warmup 10 --no-extra --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('separate final', d['value'], d['ms_per_step'])" >> $O/bn2.txt
  PCMI_BN_FUSED_FINAL=1 timeout 200 python bench.py --steps 30 --warmup 10 --no-extra --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('fused final   ', d['value'], d['ms_per_step'])" >> $O/bn2.txt
done
cat $O/pytest_bn.log $O/bn2.txt
