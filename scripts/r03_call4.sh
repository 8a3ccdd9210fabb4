#!/bin/bash
# Round 3, GPU call 4: A/B of the split-precision conv on the coarse levels (PCMI_CONV16 threshold, weights packed per
# level size), the weight-gradient kernel's residency, pair selection on the planning stream.
set -u
ulimit -c 0
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
TAG=${TAG:-r03e}
O=gpurun_out/$TAG
mkdir -p $O
T0=$(date +%s)
stamp() { echo "[$(( $(date +%s) - T0 )) s] $*" | tee -a $O/stages.log; }
line() { python - "$1" "$2" <<'PY' | tee -a $O/runs.txt
import sys, json
try:
  d = json.load(open(sys.argv[1])); h = d["config"].get("host_phase_ms_per_step", {})
  print(sys.argv[2], "|", d["value"], "pairs/s", d["ms_per_step"], "ms | enqueue", d["config"]["host_enqueue_ms_per_step"], "|",
        {k: v for k, v in h.items() if not k.endswith("_cpu")})
except Exception as e:
  print(sys.argv[2], "failed:", e)
PY
}
B="python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-roofline"
stamp "targeted tests"
timeout 600 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x \
  -k "pair_selection or 1cm or prepacked or trainer_iteration or full_config_forward or conv16 or refsrc" > $O/pytest_targeted.log 2>&1
echo "targeted exit $?" | tee -a $O/stages.log; grep -E "passed|failed" $O/pytest_targeted.log | tail -1; grep -E "^FAILED|^ERROR" $O/pytest_targeted.log | head
stamp "bench A/B"
run() { local label=$1; shift; env "$@" 2>> $O/bench.err | tail -1 > "$O/run_${label// /_}.json"; line "$O/run_${label// /_}.json" "$label"; }
run "default 1" timeout 120 $B
run "default 2" timeout 120 $B
run "conv16 2048 a" PCMI_CONV16=2048 timeout 120 $B
run "conv16 2048 b" PCMI_CONV16=2048 timeout 120 $B
run "conv16 512 a" PCMI_CONV16=512 timeout 120 $B
run "conv16 512 b" PCMI_CONV16=512 timeout 120 $B
run "x3t 1 wg per cu" PCMI_WGRAD_X3T_WGS=1 timeout 120 $B
run "x3t 1 wg per cu, all sizes" PCMI_WGRAD_X3T_WGS=1 PCMI_WGRAD_X3T=8192 PCMI_WGRAD_X3T_MAX=10000000 timeout 120 $B
run "host pair selection" timeout 120 $B --set misc.device_pair_selection=False
run "conv16 512 + x3t off" PCMI_CONV16=512 PCMI_WGRAD_X3T=0 timeout 120 $B
stamp "tests with PCMI_CONV16=512"
PCMI_CONV16=512 timeout 600 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x \
  --durations=10 -k "network_features or full_config_forward or refsrc or trainer_iteration or joint_pair or engine_matches" > $O/pytest_conv16_512.log 2>&1
echo "conv16=512 tests exit $?" | tee -a $O/stages.log; grep -E "passed|failed|s call" $O/pytest_conv16_512.log | tail -12; grep -E "^FAILED|^ERROR" $O/pytest_conv16_512.log | head
stamp "done"
