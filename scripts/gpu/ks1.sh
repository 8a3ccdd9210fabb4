#!/bin/bash
mkdir -p gpurun_out/r04s; O=gpurun_out/r04s; rm -f $O/ks1.txt
for v in none 16 8 4 2 none; do
  E=""; [ $v != none ] && E="PCMI_KSPLIT_MAX_MB=$v"
  env $E timeout 200 python bench.py --steps 30 --warmup 10 --no-extra --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('max partial MB $v', d['value'], d['ms_per_step'])" >> $O/ks1.txt
done
cat $O/ks1.txt
