#!/bin/bash
mkdir -p gpurun_out/r04p; O=gpurun_out/r04p; rm -f $O/bn3.txt
timeout 400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_trace.py -q -x -k "batchnorm_parity or joint_pair or twenty" --durations=5 2>&1 | tail -12 > $O/pytest_bn3.log
for i in 1 2 3; do
  timeout 200 python bench.py --steps 30 --warmup 10 --no-extra --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('merge in apply ', d['value'], d['ms_per_step'], d.get('final_loss'))" >> $O/bn3.txt
  PCMI_BN_MERGE_IN_APPLY=0 timeout 200 python bench.py --steps 30 --warmup 10 --no-extra --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('separate final ', d['value'], d['ms_per_step'], d.get('final_loss'))" >> $O/bn3.txt
done
PCMI_BN_MERGE_IN_APPLY=0 PCMI_BN_FUSED_FINAL=1 timeout 200 python bench.py --steps 30 --warmup 10 --no-extra --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('fused final    ', d['value'], d['ms_per_step'], d.get('final_loss'))" >> $O/bn3.txt
cat $O/pytest_bn3.log $O/bn3.txt
