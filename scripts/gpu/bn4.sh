#!/bin/bash
mkdir -p gpurun_out/r04p; O=gpurun_out/r04p; rm -f $O/bn4.txt
for v in base bn_skipf bn_skipfb bn_skipall base; do
  L=""; [ $v != base ] && L=$PWD/pointcontrast_amd/libpcmi_$v.so
  PCMI_LIB=$L timeout 200 python bench.py --steps 30 --warmup 10 --no-extra --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', d['value'], d['ms_per_step'])" >> $O/bn4.txt
done
cat $O/bn4.txt
