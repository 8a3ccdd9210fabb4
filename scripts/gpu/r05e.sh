#!/bin/bash
# Round 5, call 5: BatchNorm in one launch (fixed stores) -- tests, A/B over the row threshold; the pair gather with one
# gradient buffer (misc.fused_pair_gather) A/B; kernel statistics.
set -u
ulimit -c 0
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$ROOT"
export TMPDIR=/tmp
TAG=${TAG:-r05e}
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
  d = json.loads(txt[-1]); c = d["config"]; h = c.get("host_phase_ms_per_step", {})
  print(sys.argv[2], "|", d["value"], "pairs/s", d["ms_per_step"], "ms | loss", c["final_loss"], "| enqueue", c["host_enqueue_ms_per_step"], "| fwd host", h.get("forward"), "cpu", h.get("forward_cpu"),
        "| bwd_step host", h.get("backward_step"), "cpu", h.get("backward_step_cpu"))
except Exception as e:
  print(sys.argv[2], "failed:", e)
PY
}
run() {  # label n env... [-- bench args]
  local label=$1 n=$2; shift 2
  local envs=() args=()
  while [ $# -gt 0 ] && [ "$1" != "--" ]; do envs+=("$1"); shift; done
  [ $# -gt 0 ] && shift
  args=("$@")
  for i in $(seq 1 $n); do
    env "${envs[@]}" timeout 150 $B "${args[@]}" > $O/ab_${label}_$i.json 2>> $O/bench.err
    line $O/ab_${label}_$i.json "$label run $i"
  done
}
stamp "1 A/B"
run small_off 3 PCMI_BN_SMALL_ROWS=0
run small_512 2 PCMI_BN_SMALL_ROWS=512
run small_768 3 PCMI_BN_SMALL_ROWS=768
run small_1536 3 PCMI_NOP=1
run gather_unfused 3 PCMI_NOP=1 -- --set misc.fused_pair_gather=False
stamp "2 whole suite"
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider --durations=5 > $O/pytest_gpu.log 2>&1
echo "pytest exit $?" | tee -a $O/stages.log; grep -E "passed|failed" $O/pytest_gpu.log | tail -2; grep -E "^FAILED|^ERROR" $O/pytest_gpu.log | head -20
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke exit $?" | tee -a $O/stages.log; tail -1 $O/smoke.log
stamp "3 rocprofv3 kernel stats"
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/prof" -o bench -- \
    python "$ROOT/bench.py" --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-extra > "$O/prof.log" 2>&1 )
echo "prof exit $?" >> $O/stages.log
find $O/prof -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
find $O/prof -name "*kernel_trace.csv" -exec cp {} $O/kernel_trace.csv \;
rm -rf $O/prof
grep -E "bn_small|colreduce|bn_apply|bn_bwd" $O/kernel_stats.csv | cut -c1-170
stamp "done"
