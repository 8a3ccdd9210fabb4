#!/bin/bash
# round 4, GPU call 2: DEEP conv schedule -- parity subset, stand-alone kernel A/B (kbench), step A/B, kernel trace CSV
set -u
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$ROOT"
export TMPDIR=/tmp
O=$ROOT/gpurun_out/r04b
mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.txt 2>&1
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "spconv_parity or conv16 or streamk or linearity or prepack or network_features or full_config_forward" > $O/gpu_tests_conv.txt 2>&1; echo "pytest rc $?" >> $O/gpu_tests_conv.txt
tail -3 $O/gpu_tests_conv.txt
for m in 0 7; do
  KBENCH_SUSTAINED=0 PCMI_X3_DEEP=$m timeout 200 python scripts/kbench.py > $O/kbench_deep$m.txt 2>&1
done
B="python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-roofline --no-extra"
for r in a b; do
  for m in 0 7 5 2; do
    PCMI_X3_DEEP=$m timeout 150 $B > $O/step_deep${m}_$r.json 2>> $O/ab.err
  done
done
timeout 150 $B --set misc.force_reducer=True --set misc.reducer_profile=False > $O/forced_noprofile.json 2>> $O/ab.err
python - <<PY
import json,glob
for f in sorted(glob.glob("$O/*.json")):
  try:
    d=[json.loads(l) for l in open(f).read().splitlines() if l.startswith("{")][-1]; print(f.split('/')[-1], d['value'], d['ms_per_step'])
  except Exception as e: print(f, 'failed', e)
PY
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o bench -- python $ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-extra > $O/prof_run.txt 2>&1
find $O/prof -name "*kernel_trace.csv" -exec cp {} $O/kernel_trace.csv \;
find $O/prof -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
rm -rf $O/prof
ls -la $O
echo done
