#!/bin/bash
# Round 6, call 13: wgrad_x3q_kernel (twelve waves, eight of them staging, an offset pair per workgroup; PCMI_WGRAD_X3P=2) against
# wgrad_x3p_kernel (=1): parity, stand-alone times (scripts/kbench.py), the step (alternating processes), kernel stats.
set -u
ulimit -c 0
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$ROOT"
export TMPDIR=/tmp
TAG=${TAG:-r06m}
O=$ROOT/gpurun_out/$TAG
mkdir -p $O
T0=$(date +%s)
stamp() { echo "[$(( $(date +%s) - T0 )) s] $*" | tee -a $O/stages.log; }
B="python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-roofline --no-extra"
line() {
  python - "$1" "$2" <<'PY' | tee -a $O/ab.txt
import json, sys
try:
  d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1]); print(sys.argv[2], "|", d["value"], "pairs/s", d["ms_per_step"], "ms | loss", d["config"]["final_loss"])
except Exception as e:
  print(sys.argv[2], "failed:", e)
PY
}
stamp "1 parity"
timeout 600 python -m pytest tests/test_gpu_parity.py -k "wgrad or dense_1x1" -m gpu -q --tb=short -p no:cacheprovider -x > $O/pytest_sel.log 2>&1
echo "pytest(sel) exit $?" | tee -a $O/stages.log; tail -5 $O/pytest_sel.log
stamp "2 stand-alone"
for m in 2 1; do
  echo "== PCMI_WGRAD_X3P=$m" | tee -a $O/kbench.txt
  PCMI_WGRAD_X3P=$m timeout 300 python scripts/kbench.py 2>&1 | grep -E "3\^3|1x1" | sed 's/ fwd .*| wgrad/ wgrad/' | tee -a $O/kbench.txt
done
stamp "3 step"
for i in 1 2 3; do
  for m in 1 2; do
    PCMI_WGRAD_X3P=$m timeout 150 $B > $O/ab_x3p${m}_$i.json 2>> $O/bench.err
    line $O/ab_x3p${m}_$i.json "PCMI_WGRAD_X3P=$m run $i"
  done
done
stamp "4 more parity (network gradients, reproducibility)"
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -k "network_features or bit_reproducible or full_config_gradients" -m gpu -q --tb=short -p no:cacheprovider > $O/pytest_net.log 2>&1
echo "pytest(net) exit $?" | tee -a $O/stages.log; tail -5 $O/pytest_net.log
stamp "done"
