#!/bin/bash
mkdir -p gpurun_out/r04n; O=gpurun_out/r04n
cd /tmp && export TMPDIR=/tmp
for v in base nce_nofinish nce_noloop nce_neither; do
  L=""; [ $v != base ] && L=/root/repo/pointcontrast_amd/libpcmi_$v.so
  PCMI_LIB=$L PYTHONPATH=/root/repo timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$v -o p -- python /root/repo/scripts/loss_bench.py 4096 > /dev/null 2>&1
  f=$(find /tmp/prof_$v -name "*kernel_stats.csv" | head -1)
  cp $f /root/repo/$O/diag_$v.csv
done
cd /root/repo
python - <<'PY'
import csv
for v in ("base","nce_nofinish","nce_noloop","nce_neither"):
  print(v)
  for r in csv.DictReader(open('/root/repo/gpurun_out/r04n/diag_%s.csv'%v)):
    if 'nce_' in r['Name']: print('  %-30s calls %4s avg %8.1f us min %7.1f max %7.1f' % (r['Name'].split('(')[1][-30:] if 'anonymous' in r['Name'] else r['Name'][:30], r['Calls'], float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3, float(r['MaxNs'])/1e3))
PY
