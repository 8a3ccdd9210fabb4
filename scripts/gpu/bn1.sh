#!/bin/bash
mkdir -p gpurun_out/r04n; O=gpurun_out/r04n; rm -f $O/bn1.txt
for i in 1 2 3; do
  timeout 200 python bench.py --steps 30 --warmup 10 --no-extra --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('base      ', d['value'], d['ms_per_step'])" >> $O/bn1.txt
  PCMI_LIB=$PWD/pointcontrast_amd/libpcmi_bn_nofinish.so timeout 200 python bench.py --steps 30 --warmup 10 --no-extra --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bn_nofinish', d['value'], d['ms_per_step'])" >> $O/bn1.txt
done
cat $O/bn1.txt
