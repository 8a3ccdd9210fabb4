#!/bin/bash
# Round 6, call 24: the residual branch of a BasicBlock (1x1 convolution + BatchNorm of the block's input) on the executor's side
# stream beside the block's main path in the forward pass (PCMI_FWD_BRANCH, engine.hip): bit-identity, timing-mode and network
# parity tests, then the step with the branches forked / in program order (alternating).  SWITCH=PCMI_X3_PACK_SIDE: the same for the
# forward weight pack on that stream.
set -u
ulimit -c 0
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$ROOT"
export TMPDIR=/tmp
O=$ROOT/gpurun_out/${TAG:-r06x}
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_timing.py tests/test_gpu_parity.py -k "residual_branches or timing or time_ops or network_features or bit_reproducible or prepacked" -m gpu -q --tb=short -p no:cacheprovider -x > $O/pytest_sel.log 2>&1
echo "pytest(sel) exit $?" | tee -a $O/stages.log; tail -4 $O/pytest_sel.log
B="python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-roofline --no-extra"
SW=${SWITCH:-PCMI_FWD_BRANCH}
for i in 1 2 3 4; do
  for m in 0 1; do
    env $SW=$m timeout 150 $B > $O/ab_${m}_$i.json 2>> $O/bench.err
    python - $O/ab_${m}_$i.json "$SW=$m run $i" <<'PY' | tee -a $O/ab.txt
import json, sys
try:
  d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1]); print(sys.argv[2], "|", d["value"], "pairs/s", d["ms_per_step"], "ms | loss", d["config"]["final_loss"])
except Exception as e:
  print(sys.argv[2], "failed:", e)
PY
  done
done
