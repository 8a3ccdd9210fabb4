#!/bin/bash
# Round 3, GPU call 1: the whole GPU suite with the new full-size tests and tightened tolerances, smoke, then the
# run-to-run distribution of the default bench (10 consecutive processes) as the baseline for the "reproducible number" work.
set -u
ulimit -c 0
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
TAG=${TAG:-r03a}
O=gpurun_out/$TAG
mkdir -p $O
T0=$(date +%s)
stamp() { echo "[$(( $(date +%s) - T0 )) s] $*" | tee -a $O/stages.log; }
stamp "pytest -m gpu"
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider --durations=12 -rP > $O/pytest_gpu.log 2>&1
echo "pytest exit $?" | tee -a $O/stages.log
grep -E "passed|failed" $O/pytest_gpu.log | tail -2
grep -E "^FAILED|^ERROR" $O/pytest_gpu.log | head -20
stamp "smoke"
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke exit $?" | tee -a $O/stages.log; tail -1 $O/smoke.log
stamp "10 bench runs"
for i in 1 2 3 4 5 6 7 8 9 10; do
  timeout 120 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-roofline 2>> $O/bench.err | tail -1 > $O/run_$i.json
  python - "$O/run_$i.json" "$i" <<'PY' | tee -a $O/runs.txt
import sys, json
try:
  d = json.load(open(sys.argv[1])); h = d["config"].get("host_phase_ms_per_step", {})
  print("run", sys.argv[2], "|", d["value"], "pairs/s", d["ms_per_step"], "ms | enqueue", d["config"]["host_enqueue_ms_per_step"], "|",
        {k: v for k, v in h.items() if not k.endswith("_cpu")})
except Exception as e:
  print("run", sys.argv[2], "failed:", e)
PY
done
stamp "done"
