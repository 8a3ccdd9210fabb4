#!/bin/bash
# Round 6, call 19: bn_bwd_apply_kernel with the mask form and the residual-gradient mode as template parameters and every load of an
# element requested up front -- which fits 48 registers since -fno-slp-vectorize (46-48; 50-58 before) -- against the run-time-branch
# form (libpcmi_oldbn.so: the previous commit's library): parity / bit-identity tests, the step (alternating), per-layer times.
set -u
ulimit -c 0
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$ROOT"
export TMPDIR=/tmp
O=$ROOT/gpurun_out/${TAG:-r06s}
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_timing.py -k "batchnorm or bn_ or relu_bits or network_features or bit_identical" -m gpu -q --tb=short -p no:cacheprovider > $O/pytest_sel.log 2>&1
echo "pytest(sel) exit $?" | tee -a $O/stages.log; tail -4 $O/pytest_sel.log
B="python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-roofline --no-extra"
for i in 1 2 3 4; do
  for v in oldbn product; do
    if [ $v = product ]; then L=$ROOT/pointcontrast_amd/libpcmi.so; else L=$ROOT/pointcontrast_amd/libpcmi_$v.so; fi
    PCMI_LIB=$L timeout 150 $B > $O/ab_${v}_$i.json 2>> $O/bench.err
    python - $O/ab_${v}_$i.json "$v run $i" <<'PY' | tee -a $O/ab.txt
import json, sys
try:
  d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1]); print(sys.argv[2], "|", d["value"], "pairs/s", d["ms_per_step"], "ms | loss", d["config"]["final_loss"])
except Exception as e:
  print(sys.argv[2], "failed:", e)
PY
  done
done
for v in oldbn product; do
  if [ $v = product ]; then L=$ROOT/pointcontrast_amd/libpcmi.so; else L=$ROOT/pointcontrast_amd/libpcmi_$v.so; fi
  PCMI_LIB=$L timeout 200 python bench.py --steps 10 --warmup 4 --no-cpu-baseline --no-extra > $O/fam_$v.json 2>> $O/bench.err
  python - $O/fam_$v.json $v <<'PY' | tee -a $O/families.txt
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print(sys.argv[2], " | ".join("%s %.3f" % (f["family"][:22], f["ms_per_step"]) for f in d["families"][:5]))
PY
done
