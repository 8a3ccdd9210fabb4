#!/bin/bash
# Round 3, GPU call 3: validation of the pruned tree (one conv generation and 16 switches fewer, no backward_pair), the
# conflict-free weight-fragment layout of the split-precision conv, the weight pack off the chain, device-side pair
# selection, the device-resident loader stage; bench lines of the three configurations.
set -u
ulimit -c 0
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
TAG=${TAG:-r03c}
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
stamp "full GPU suite"
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider --durations=8 -rP > $O/pytest_gpu.log 2>&1
echo "pytest exit $?" | tee -a $O/stages.log
grep -E "passed|failed" $O/pytest_gpu.log | tail -2; grep -E "^FAILED|^ERROR" $O/pytest_gpu.log | head -20
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke exit $?" | tee -a $O/stages.log; tail -1 $O/smoke.log
stamp "kbench level 0/1"
KBENCH_LEVELS=0,1 KBENCH_SUSTAINED=0 timeout 200 python scripts/kbench.py > $O/kbench.txt 2>> $O/kbench.err; grep "3^3" $O/kbench.txt | cut -c1-200
stamp "bench A/B"
for i in 1 2 3; do timeout 120 $B 2>> $O/bench.err | tail -1 > $O/run_default_$i.json; line $O/run_default_$i.json "default $i"; done
for i in 1 2; do PCMI_X3_PACK_ASYNC=0 timeout 120 $B 2>> $O/bench.err | tail -1 > $O/run_packsync_$i.json; line $O/run_packsync_$i.json "PCMI_X3_PACK_ASYNC=0 $i"; done
timeout 120 $B --set misc.device_pair_selection=False 2>> $O/bench.err | tail -1 > $O/run_hostsel.json; line $O/run_hostsel.json "host pair selection"
PCMI_CONV16=2048 timeout 120 $B 2>> $O/bench.err | tail -1 > $O/run_conv16_2048.json; line $O/run_conv16_2048.json "PCMI_CONV16=2048"
PCMI_WGRAD_X3T=0 timeout 120 $B 2>> $O/bench.err | tail -1 > $O/run_nox3t.json; line $O/run_nox3t.json "PCMI_WGRAD_X3T=0"
timeout 120 $B --loss hardest 2>> $O/bench.err | tail -1 > $O/run_hardest.json; line $O/run_hardest.json "hardest"
timeout 200 python bench.py --voxel 0.01 --steps 10 --warmup 3 --no-roofline --no-cpu-baseline 2>> $O/bench.err | tail -1 > $O/run_1cm.json; line $O/run_1cm.json "1 cm"
timeout 200 python bench.py --voxel 0.01 --steps 10 --warmup 3 --no-roofline --no-cpu-baseline --set misc.device_pair_selection=False 2>> $O/bench.err | tail -1 > $O/run_1cm_hostsel.json; line $O/run_1cm_hostsel.json "1 cm, host pair selection"
stamp "loader bench"
timeout 300 python scripts/loader_bench.py 8 > $O/loader_bench.txt 2>&1; cat $O/loader_bench.txt | tail -6
stamp "done"
