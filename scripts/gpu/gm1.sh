#!/bin/bash
mkdir -p gpurun_out/r04q; O=gpurun_out/r04q; rm -f $O/gm1.txt
timeout 500 python -m pytest tests/test_gpu_parity.py -q -x -k "spconv_parity or conv16_x3 or joint_pair or spconv_golden or engine_prepack" 2>&1 | tail -4 > $O/pytest_gm1.log
for i in 1 2 3; do
  timeout 200 python bench.py --steps 30 --warmup 10 --no-extra --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('group-major', d['value'], d['ms_per_step'])" >> $O/gm1.txt
  PCMI_X3_GROUP_MAJOR=0 timeout 200 python bench.py --steps 30 --warmup 10 --no-extra --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('tile-major ', d['value'], d['ms_per_step'])" >> $O/gm1.txt
done
cat $O/pytest_gm1.log $O/gm1.txt
