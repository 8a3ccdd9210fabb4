#!/bin/bash
mkdir -p gpurun_out/r04n; O=gpurun_out/r04n
R=$PWD
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -q -k "nce or hardest or scatter or pdist or trainer_iteration or keyset" 2>&1 | tail -30 > $O/pytest_hard.log
for i in 1 2; do
  timeout 200 python bench.py --loss hardest --steps 30 --warmup 10 --no-extra --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('hardest', d['value'], d['ms_per_step'], d.get('final_loss'))" >> $O/hard2.txt
done
timeout 200 python bench.py --steps 30 --warmup 10 --no-extra --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('nce', d['value'], d['ms_per_step'])" >> $O/hard2.txt
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_h -o p -- python $R/bench.py --loss hardest --steps 10 --warmup 3 --no-extra --no-cpu-baseline > /dev/null 2>&1
f=$(find /tmp/prof_h -name "*kernel_stats.csv" | head -1); cp $f $R/$O/hardest_kernel_stats2.csv
cd $R
cat $O/pytest_hard.log $O/hard2.txt
python - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/r04n/hardest_kernel_stats2.csv')))
for r in rows:
  n=r['Name']
  if any(k in n for k in ('hardest','pdist','keyset','scatter','gather_rows','l2norm')):
    print('%-60s calls/step %6.1f ms/step %7.4f avg %7.1f us' % (n[:60], int(r['Calls'])/13.0, float(r['TotalDurationNs'])/13e6, float(r['AverageNs'])/1e3))
PY
