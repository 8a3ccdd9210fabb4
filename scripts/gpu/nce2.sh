#!/bin/bash
mkdir -p gpurun_out/r04n; O=gpurun_out/r04n
cd /tmp && export TMPDIR=/tmp
for v in 1 0; do
  PCMI_NCE_X3=$v PYTHONPATH=/root/repo timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$v -o p -- python /root/repo/scripts/loss_bench.py 4096 > /dev/null 2>&1
  f=$(find /tmp/prof_$v -name "*kernel_stats.csv" | head -1)
  cp $f /root/repo/$O/loss_kernel_stats_x3_$v.csv
done
cd /root/repo
head -14 $O/loss_kernel_stats_x3_1.csv | cut -c1-200; head -14 $O/loss_kernel_stats_x3_0.csv | cut -c1-200
