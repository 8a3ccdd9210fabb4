#!/bin/bash
mkdir -p gpurun_out/r04p; O=gpurun_out/r04p; rm -f $O/bn5.txt
timeout 400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_trace.py -q -x -k "batchnorm_parity or joint_pair or twenty or engine_matches" 2>&1 | tail -4 > $O/pytest_bn5.log
for i in 1 2 3; do
  timeout 200 python bench.py --steps 30 --warmup 10 --no-extra --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('wide final ', d['value'], d['ms_per_step'])" >> $O/bn5.txt
  PCMI_BN_FUSED_FINAL=1 timeout 200 python bench.py --steps 30 --warmup 10 --no-extra --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('fused final', d['value'], d['ms_per_step'])" >> $O/bn5.txt
done
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_b -o p -- python /root/repo/bench.py --steps 10 --warmup 3 --no-extra --no-cpu-baseline > /dev/null 2>&1
f=$(find /tmp/prof_b -name "*kernel_stats.csv" | head -1); cp $f /root/repo/$O/bn5_kernel_stats.csv
cd /root/repo
cat $O/pytest_bn5.log $O/bn5.txt
grep -E "colreduce|bn_" $O/bn5_kernel_stats.csv | awk -F'","' '{printf "%-50s calls %s avg %.1f us\n", substr($1,2,50), $2, $4/1000}'
