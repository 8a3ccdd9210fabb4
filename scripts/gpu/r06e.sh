#!/bin/bash
# Round 6, call 5: the optimiser per gradient bucket beside the backward pass (misc.bucket_sgd: GradReducer.after_bucket ->
# FlatSGD.step_range on the communication stream) and the committed form of the offset-split rule (levels under 16384 rows):
# bit-identity test, step A/B (alternating), forced 1-rank reducer lines.
set -u
ulimit -c 0
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$ROOT"
export TMPDIR=/tmp
TAG=${TAG:-r06e}
O=$ROOT/gpurun_out/$TAG
mkdir -p $O
T0=$(date +%s)
stamp() { echo "[$(( $(date +%s) - T0 )) s] $*" | tee -a $O/stages.log; }
python -c "import __graft_entry__ as g; g.build()" > $O/build.txt 2>&1
B="python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-roofline --no-extra"
line() {  # file label
  python - "$1" "$2" <<'PY' | tee -a $O/ab.txt
import json, sys
try:
  txt = [l for l in open(sys.argv[1]) if l.startswith("{")]
  d = json.loads(txt[-1]); c = d["config"]
  ov = (c.get("collective") or {}).get("overlap") or {}
  print(sys.argv[2], "|", d["value"], "pairs/s", d["ms_per_step"], "ms | loss", c["final_loss"], "| enqueue", c["host_enqueue_ms_per_step"], "| exposed", ov.get("exposed_after_backward_ms"))
except Exception as e:
  print(sys.argv[2], "failed:", e)
PY
}
run1() {  # label idx extra-args... (env via ENVV)
  local label=$1 i=$2; shift 2
  env $ENVV timeout 150 $B "$@" > $O/ab_${label}_$i.json 2>> $O/bench.err
  line $O/ab_${label}_$i.json "$label run $i"
}
stamp "1 test"
timeout 600 python -m pytest tests/test_gpu_timing.py tests/test_gpu_bucket_sync.py "tests/test_gpu_parity.py::test_rccl_reducer_path_single_rank" "tests/test_gpu_fullsize.py::test_full_config_step_is_bit_reproducible" -m gpu -q --tb=short -p no:cacheprovider > $O/pytest_sel.log 2>&1
echo "pytest(sel) exit $?" | tee -a $O/stages.log; grep -E "passed|failed|skipped" $O/pytest_sel.log | tail -3; grep -E "^FAILED|^ERROR" $O/pytest_sel.log | head
stamp "2 A/B"
for i in 1 2 3; do
  ENVV="PCMI_KSPLIT_RULE=0" run1 both_off $i --set misc.bucket_sgd=False
  ENVV="PCMI_NOP=1" run1 rule_only $i --set misc.bucket_sgd=False
  ENVV="PCMI_NOP=1" run1 rule_and_bucket_sgd $i
done
stamp "3 forced 1-rank reducer"
for i in 1 2; do
  ENVV="PCMI_NOP=1" run1 forced_sgd_behind $i --set misc.force_reducer=True --set misc.bucket_sgd=False
  ENVV="PCMI_NOP=1" run1 forced_sgd_per_bucket $i --set misc.force_reducer=True
done
stamp "done"
