#!/bin/bash
# Round 5, call 9: the 1024-thread backward form of the one-launch BatchNorm at 768-1536 rows per segment: parity tests, A/B
set -u
ulimit -c 0
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$ROOT"
export TMPDIR=/tmp
TAG=${TAG:-r05i}
O=$ROOT/gpurun_out/$TAG
mkdir -p $O
T0=$(date +%s)
stamp() { echo "[$(( $(date +%s) - T0 )) s] $*" | tee -a $O/stages.log; }
python -c "import __graft_entry__ as g; g.build()" > $O/build.txt 2>&1
B="python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-roofline --no-extra"
line() {
  python - "$1" "$2" <<'PY' | tee -a $O/ab.txt
import json, sys
try:
  txt = [l for l in open(sys.argv[1]) if l.startswith("{")]
  d = json.loads(txt[-1]); c = d["config"]
  print(sys.argv[2], "|", d["value"], "pairs/s", d["ms_per_step"], "ms | loss", c["final_loss"])
except Exception as e:
  print(sys.argv[2], "failed:", e)
PY
}
run() {
  local label=$1 n=$2; shift 2
  for i in $(seq 1 $n); do
    env "$@" timeout 150 $B > $O/ab_${label}_$i.json 2>> $O/bench.err
    line $O/ab_${label}_$i.json "$label run $i"
  done
}
stamp "1 tests"
timeout 600 python -m pytest "tests/test_gpu_parity.py::test_batchnorm_parity" "tests/test_gpu_parity.py::test_batchnorm_one_launch_form_matches_the_three_launch_form" -m gpu -q --tb=short -p no:cacheprovider > $O/pytest_bn.log 2>&1
echo "pytest(bn) exit $?" | tee -a $O/stages.log; grep -E "passed|failed|skipped" $O/pytest_bn.log | tail -2; grep -E "^FAILED|^ERROR" $O/pytest_bn.log | head
timeout 100 python scripts/debug/bn_small_probe.py 2>&1 | grep -v amdgpu.ids | cut -c1-200 | tee $O/probe.txt
stamp "2 A/B"
run bwd768 3 PCMI_NOP=1
run bwd1536 3 PCMI_BN_SMALL_BWD_ROWS=1536
run bwd768_b 2 PCMI_NOP=1
run bwd1536_b 2 PCMI_BN_SMALL_BWD_ROWS=1536
stamp "done"
