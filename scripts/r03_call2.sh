#!/bin/bash
# Round 3, GPU call 2: the one-synchronisation plan and the tile-stationary split-precision weight gradients.
set -u
ulimit -c 0
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
TAG=${TAG:-r03b}
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
timeout 600 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -rP -x \
  -k "plan_unet or deferred_insert or wgrad_x3t or spconv_parity or maps_match or full_size_map" > $O/pytest_targeted.log 2>&1
echo "targeted exit $?" | tee -a $O/stages.log
grep -E "passed|failed" $O/pytest_targeted.log | tail -2; grep -E "^FAILED|^ERROR|wgrad errors" $O/pytest_targeted.log | cut -c1-300 | head -20
stamp "kbench wgrad A/B"
for v in 0 8192 4096; do
  echo "== PCMI_WGRAD_X3T=$v" >> $O/kbench.txt
  PCMI_WGRAD_X3T=$v KBENCH_LEVELS=0,1,2 KBENCH_SUSTAINED=0 timeout 200 python scripts/kbench.py >> $O/kbench.txt 2>> $O/kbench.err
done
grep -E "==|3\^3 (96->96|128->96|64->64|128->128|192->128)" $O/kbench.txt | cut -c1-200
stamp "bench A/B"
for i in 1 2 3; do timeout 120 $B 2>> $O/bench.err | tail -1 > $O/run_default_$i.json; line $O/run_default_$i.json "default $i"; done
for i in 1 2; do PCMI_WGRAD_X3T=0 timeout 120 $B 2>> $O/bench.err | tail -1 > $O/run_nox3t_$i.json; line $O/run_nox3t_$i.json "PCMI_WGRAD_X3T=0 $i"; done
PCMI_PLAN_CHAIN=0 timeout 120 $B 2>> $O/bench.err | tail -1 > $O/run_nochain.json; line $O/run_nochain.json "PCMI_PLAN_CHAIN=0"
PCMI_WGRAD_X3T=4096 timeout 120 $B 2>> $O/bench.err | tail -1 > $O/run_x3t4096.json; line $O/run_x3t4096.json "PCMI_WGRAD_X3T=4096"
stamp "full GPU suite"
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider --durations=8 -rP > $O/pytest_gpu.log 2>&1
echo "pytest exit $?" | tee -a $O/stages.log
grep -E "passed|failed" $O/pytest_gpu.log | tail -2; grep -E "^FAILED|^ERROR" $O/pytest_gpu.log | head -20
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke exit $?" | tee -a $O/stages.log; tail -1 $O/smoke.log
stamp "rocprofv3 kernel stats"
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/$O/prof" -o bench -- \
    python "$GRAFT_REPO_ROOT/bench.py" --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > "$GRAFT_REPO_ROOT/$O/prof.log" 2>&1 )
echo "prof exit $?" >> $O/stages.log
find $O/prof -name "*kernel_trace*" -size +8M -delete 2>/dev/null
stamp "done"
