#!/bin/bash
mkdir -p gpurun_out/r04n; O=gpurun_out/r04n
R=$PWD
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_h -o p -- python $R/bench.py --loss hardest --steps 10 --warmup 3 --no-extra --no-cpu-baseline > $R/$O/hardest_line.json 2>/dev/null
f=$(find /tmp/prof_h -name "*kernel_stats.csv" | head -1); cp $f $R/$O/hardest_kernel_stats.csv
cd $R
python - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/r04n/hardest_kernel_stats.csv')))
for r in rows:
  n=r['Name']
  if any(k in n for k in ('hardest','pdist','keyset','scatter','gather_rows','rocprim','at::native','l2norm','nce')):
    print('%-60s calls/step %6.1f ms/step %7.4f avg %7.1f us' % (n[:60], int(r['Calls'])/13.0, float(r['TotalDurationNs'])/13e6, float(r['AverageNs'])/1e3))
PY
tail -c 600 $O/hardest_line.json
