#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of the HBM-priced kernels under 0.40 (scripts/pmc_probe.py, PMC_PROBE_SET=gathers)
R=$PWD; O=$R/gpurun_out/r04g; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for pass in FETCH_SIZE WRITE_SIZE; do
  PMC_PROBE_SET=gathers timeout 150 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $O/pmc_$pass -o pmc -- python $R/scripts/pmc_probe.py > $O/pmc_$pass.log 2>&1
  find $O/pmc_$pass -name "*counter_collection.csv" -exec cp {} $O/pmc_$pass.csv \;
  find $O/pmc_$pass -name "*kernel_trace.csv" -exec cp {} $O/trace_$pass.csv \;
  rm -rf $O/pmc_$pass
done
grep -h "ALGO\|CAL" $O/pmc_FETCH_SIZE.log
python - <<'PY'
import csv, collections
for c in ("FETCH_SIZE","WRITE_SIZE"):
  acc=collections.OrderedDict()
  for r in csv.DictReader(open('/root/repo/gpurun_out/r04g/pmc_%s.csv'%c)):
    k=r['Kernel_Name'][:70]
    acc.setdefault(k,[]).append(float(r['Counter_Value']))
  print(c)
  for k,v in acc.items(): print('  %-70s n=%d  %s' % (k, len(v), ' '.join('%.1f'%x for x in v[:9])))
PY
