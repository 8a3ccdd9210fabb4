#!/bin/bash
# Round 6, call 3: the executor's all-op timing mode (tests/test_gpu_timing.py), the new full-size parity tests (hardest-contrastive
# gradients at configs[2], backward at the 1 cm shape), and the driver-form bench line with families[] + per-layer times.
set -u
ulimit -c 0
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$ROOT"
export TMPDIR=/tmp
TAG=${TAG:-r06c}
O=$ROOT/gpurun_out/$TAG
mkdir -p $O
T0=$(date +%s)
stamp() { echo "[$(( $(date +%s) - T0 )) s] $*" | tee -a $O/stages.log; }
python -c "import __graft_entry__ as g; g.build()" > $O/build.txt 2>&1
stamp "1 tests"
timeout 900 python -m pytest tests/test_gpu_timing.py "tests/test_gpu_fullsize.py::test_full_config_hardest_gradients_match_oracle" \
  "tests/test_gpu_fullsize.py::test_1cm_config_gradients_match_oracle" "tests/test_gpu_parity.py::test_network_features_loss_and_grads" \
  -m gpu -q --tb=short -p no:cacheprovider -s --durations=6 > $O/pytest_sel.log 2>&1
echo "pytest(sel) exit $?" | tee -a $O/stages.log; grep -E "passed|failed|skipped" $O/pytest_sel.log | tail -3; grep -E "^FAILED|^ERROR|worst gradient|level-1 tensors|chain ops" $O/pytest_sel.log | head -20
stamp "2 bench as the driver runs it"
timeout 900 python bench.py --layer-table $O/layer_table.tsv > $O/bench_line.json 2> $O/bench.err
echo "bench exit $?" >> $O/stages.log; cut -c1-300 $O/bench_line.json; echo
python - <<PY
import json
d = json.loads([l for l in open("$O/bench_line.json") if l.startswith("{")][-1])
print("value", d["value"], "ms", d["ms_per_step"])
for f in d.get("families") or []:
  print("  %-100s %8.3f ms %6s launches  frac %s" % (f["family"][:100], f["ms_per_step"], f.get("launches_per_step"), f.get("frac")))
print(d.get("families_note"))
print("nce:", [k for k in d["kernels"] if k["kernel"].startswith("nce")])
print("cpu_baseline:", d.get("cpu_baseline"))
PY
stamp "done"
