#!/bin/bash
# Round 6, call 20: compute units the tile-stationary weight-gradient kernel is sized for (PCMI_WGRAD_X3P_CUS; 256 -> 32 row blocks =
# 224 workgroups at 27 offsets, 192 -> 24 = 168, 128 -> 16 = 112), again on the round-6 kernels: the chain's LDS-heavy kernels
# only run on the compute units it leaves.  The step, alternating processes.
set -u
ulimit -c 0
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$ROOT"
export TMPDIR=/tmp
O=$ROOT/gpurun_out/${TAG:-r06t}
mkdir -p $O
B="python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-roofline --no-extra"
for i in 1 2 3; do
  for c in 256 192 128; do
    PCMI_WGRAD_X3P_CUS=$c timeout 150 $B > $O/ab_${c}_$i.json 2>> $O/bench.err
    python - $O/ab_${c}_$i.json "cus=$c run $i" <<'PY' | tee -a $O/ab.txt
import json, sys
try:
  d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1]); print(sys.argv[2], "|", d["value"], "pairs/s", d["ms_per_step"], "ms | loss", d["config"]["final_loss"])
except Exception as e:
  print(sys.argv[2], "failed:", e)
PY
  done
done
