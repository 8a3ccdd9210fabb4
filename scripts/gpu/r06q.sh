#!/bin/bash
# Round 6, call 17: order of the staging waves' gather requests and conversions inside a step of wgrad_x3p_kernel (the requests of
# four waves queue on the CU's one address unit: 780 of a step's 2830 cycles): 0 = requests first (the product), 1 = conversion
# first, 2 = a channel's conversion behind every few requests.  Stand-alone, stamps of order 2, the step.
set -u
ulimit -c 0
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$ROOT"
export TMPDIR=/tmp
O=$ROOT/gpurun_out/${TAG:-r06q}
mkdir -p $O
for v in o0 o1 o2; do
  echo "== $v" | tee -a $O/kbench.txt
  PCMI_LIB=$ROOT/pointcontrast_amd/libpcmi_wg_$v.so KBENCH_LEVELS=0,1 timeout 200 python scripts/kbench.py 2>&1 | grep -E "3\^3 (128|96)" | sed 's/ fwd .*| wgrad/ wgrad/' | tee -a $O/kbench.txt
done
PCMI_LIB=$ROOT/pointcontrast_amd/libpcmi_wg_o2s.so KBENCH_LEVELS=0,1 timeout 200 python scripts/kbench.py 2>&1 | grep -E "x3p (stamp|phases)" | sort | uniq -c | sort -nr | head -12 | tee $O/stamps_o2.txt
B="python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-roofline --no-extra"
for i in 1 2 3; do
  for v in o0 o1 o2; do
    PCMI_LIB=$ROOT/pointcontrast_amd/libpcmi_wg_$v.so timeout 150 $B > $O/ab_${v}_$i.json 2>> $O/bench.err
    python - $O/ab_${v}_$i.json "$v run $i" <<'PY' | tee -a $O/ab.txt
import json, sys
try:
  d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1]); print(sys.argv[2], "|", d["value"], "pairs/s", d["ms_per_step"], "ms | loss", d["config"]["final_loss"])
except Exception as e:
  print(sys.argv[2], "failed:", e)
PY
  done
done
timeout 600 python -m pytest tests/test_gpu_parity.py -k "wgrad_x3t" -m gpu -q --tb=short -p no:cacheprovider -x 2>&1 | tail -2
