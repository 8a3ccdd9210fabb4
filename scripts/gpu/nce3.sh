#!/bin/bash
mkdir -p gpurun_out/r04n; O=gpurun_out/r04n
timeout 600 python -m pytest tests/test_gpu_parity.py -q -k "nce_parity or gather_scatter" 2>&1 | tail -5 > $O/pytest_nce.log
cd /tmp && export TMPDIR=/tmp
for v in 1; do
  PCMI_NCE_X3=$v PYTHONPATH=/root/repo timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$v -o p -- python /root/repo/scripts/loss_bench.py 4096 > /dev/null 2>&1
  f=$(find /tmp/prof_$v -name "*kernel_stats.csv" | head -1)
  cp $f /root/repo/$O/loss_kernel_stats_x3_$v.csv
done
cd /root/repo
for i in 1 2; do
  timeout 200 python bench.py --steps 30 --warmup 10 --no-extra --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('x3  ', d['value'], d['ms_per_step'])" >> $O/ab3.txt
  PCMI_NCE_X3=0 timeout 200 python bench.py --steps 30 --warmup 10 --no-extra --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('valu', d['value'], d['ms_per_step'])" >> $O/ab3.txt
done
cat $O/pytest_nce.log $O/ab3.txt
python - <<'PY'
import csv
for r in csv.DictReader(open('/root/repo/gpurun_out/r04n/loss_kernel_stats_x3_1.csv')):
    print('  %-40s calls %4s avg %8.1f us min %7.1f max %7.1f' % (r['Name'].split('(')[0][-40:], r['Calls'], float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3, float(r['MaxNs'])/1e3))
PY
