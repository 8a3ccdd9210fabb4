#!/bin/bash
# Round 6, call 15: the product build with -fno-slp-vectorize: the whole GPU suite, the twelve-wave weight-gradient form again
# (PCMI_WGRAD_X3P=2 against the default 1, now that a conversion costs fewer cycles), kernel stats of the bench loop.
set -u
ulimit -c 0
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$ROOT"
export TMPDIR=/tmp
TAG=${TAG:-r06o}
O=$ROOT/gpurun_out/$TAG
mkdir -p $O
T0=$(date +%s)
stamp() { echo "[$(( $(date +%s) - T0 )) s] $*" | tee -a $O/stages.log; }
B="python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-roofline --no-extra"
stamp "1 step A/B of the weight-gradient forms"
for i in 1 2 3; do
  for m in 1 2; do
    PCMI_WGRAD_X3P=$m timeout 150 $B > $O/ab_x3p${m}_$i.json 2>> $O/bench.err
    python - $O/ab_x3p${m}_$i.json "PCMI_WGRAD_X3P=$m run $i" <<'PY' | tee -a $O/ab.txt
import json, sys
try:
  d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1]); print(sys.argv[2], "|", d["value"], "pairs/s", d["ms_per_step"], "ms | loss", d["config"]["final_loss"])
except Exception as e:
  print(sys.argv[2], "failed:", e)
PY
  done
done
stamp "2 stand-alone"
for m in 1 2; do
  echo "== PCMI_WGRAD_X3P=$m" | tee -a $O/kbench.txt
  PCMI_WGRAD_X3P=$m KBENCH_LEVELS=0,1 timeout 300 python scripts/kbench.py 2>&1 | grep -E "3\^3 (96|128)" | sed 's/ fwd .*| wgrad/ wgrad/' | tee -a $O/kbench.txt
done
stamp "3 GPU suite"
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > $O/pytest_gpu.log 2>&1
echo "pytest exit $?" | tee -a $O/stages.log; tail -8 $O/pytest_gpu.log
stamp "done"
