#!/bin/bash
# round 4, GPU call 1: GPU suite on the new tree, the driver's bench line (extras, both rooflines), plain vs forced-reducer
# step, stream phases, kernel trace for the offline timeline
set -u
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$ROOT"
export TMPDIR=/tmp
O=$ROOT/gpurun_out/r04a
mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.txt 2>&1
timeout 900 python -m pytest tests -m gpu -q --durations=12 > $O/gpu_tests.txt 2>&1; echo "pytest rc $?" >> $O/gpu_tests.txt
tail -5 $O/gpu_tests.txt
timeout 400 python bench.py > $O/bench_line.json 2> $O/bench.err; echo "bench rc $?"
B="python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-roofline --no-extra"
for r in a b; do
  timeout 150 $B > $O/plain_$r.json 2>> $O/ab.err
  timeout 150 $B --set misc.force_reducer=True > $O/forced_$r.json 2>> $O/ab.err
done
timeout 150 $B --set misc.gpu_profile=True > $O/phases.json 2>> $O/ab.err
python - <<PY
import json,glob
for f in sorted(glob.glob("$O/*.json")):
  try:
    d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], d['value'], d['ms_per_step'], d['config'].get('gpu_phase_ms_per_step'), (d['config'].get('collective') or {}).get('overlap'))
  except Exception as e: print(f, 'failed', e)
PY
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof -o bench -- python $ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-extra > $O/prof_run.txt 2>&1
find $O/prof -name "*kernel_trace.csv" -exec cp {} $O/kernel_trace.csv \;
find $O/prof -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
rm -rf $O/prof
ls -la $O
echo done
