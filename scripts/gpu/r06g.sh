#!/bin/bash
# Round 6, call 7: every load of a batch really in flight -- the streaming / reduction kernels whose batched loads hipcc had
# serialised behind run-time-uniform conditions (BatchNorm statistics / apply forward and backward, split_reduce, sk_fixup,
# wgrad_slab_sum, wgrad_reduce: mask form / residual / accumulation as template parameters, clamped indices instead of
# `ok ? load : 0`), and relu_bits on top of that.  Parity first, then the step with the bits on / off, then per-layer times.
set -u
ulimit -c 0
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$ROOT"
export TMPDIR=/tmp
TAG=${TAG:-r06g}
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
  print(sys.argv[2], "|", d["value"], "pairs/s", d["ms_per_step"], "ms | loss", c["final_loss"], "| enqueue", c["host_enqueue_ms_per_step"])
except Exception as e:
  print(sys.argv[2], "failed:", e)
PY
}
run1() {  # label idx (env via ENVV)
  local label=$1 i=$2; shift 2
  env $ENVV timeout 150 $B "$@" > $O/ab_${label}_$i.json 2>> $O/bench.err
  line $O/ab_${label}_$i.json "$label run $i"
}
stamp "1 tests"
timeout 1200 python -m pytest tests/test_gpu_timing.py tests/test_gpu_trace.py tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_semseg.py tests/test_gpu_refsrc.py \
  -k "not maps and not loader and not pair_selection and not gather and not pdist and not nce_parity and not sgd_step and not hardest_loss_parity" \
  -m gpu -q --tb=short -p no:cacheprovider --durations=5 > $O/pytest_sel.log 2>&1
echo "pytest(sel) exit $?" | tee -a $O/stages.log; grep -E "passed|failed|skipped" $O/pytest_sel.log | tail -3; grep -E "^FAILED|^ERROR" $O/pytest_sel.log | head -20
stamp "2 A/B"
for i in 1 2 3; do
  ENVV="PCMI_BN_RELU_BITS=0" run1 fp32_mask $i
  ENVV="PCMI_NOP=1" run1 relu_bits $i
done
stamp "3 per-layer in-step times"
PCMI_BN_RELU_BITS=0 timeout 300 python bench.py --steps 10 --warmup 5 --no-cpu-baseline --no-extra --layer-table $O/layers_off.tsv > $O/line_off.json 2>> $O/bench.err
timeout 300 python bench.py --steps 10 --warmup 5 --no-cpu-baseline --no-extra --layer-table $O/layers_on.tsv > $O/line_on.json 2>> $O/bench.err
python - <<PY | tee $O/layers_ab.txt
import json
for name in ("off", "on"):
  d = json.loads([l for l in open("$O/line_%s.json" % name) if l.startswith("{")][-1])
  print("relu_bits", name, "| step", d["ms_per_step"], "|", [(f["family"][:18], f["ms_per_step"], f.get("launches_per_step")) for f in d["families"]])
  print("   kernels:", [(k["kernel"][:28], k["ms"]) for k in d["kernels"]])
PY
stamp "4 kernel stats"
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/prof" -o bench -- \
    python "$ROOT/bench.py" --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-extra > "$O/prof.log" 2>&1 )
find $O/prof -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
rm -rf $O/prof
grep -E "split_reduce|sk_fixup|slab_sum|wgrad_reduce|colreduce|bn_" $O/kernel_stats.csv | cut -c1-60,200-400 | cut -d, -f1-4 | head -30
stamp "done"
