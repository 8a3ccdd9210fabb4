#!/bin/bash
# Round 6, call 6: the ReLU pattern of the fused BatchNorm outputs as one bit per element for the backward pass
# (csrc/norm.hip: relu_bits; PCMI_BN_RELU_BITS): bit-identity + BatchNorm / network parity, step A/B (alternating),
# per-layer in-step BatchNorm times of both arms.
set -u
ulimit -c 0
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$ROOT"
export TMPDIR=/tmp
TAG=${TAG:-r06f}
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
timeout 900 python -m pytest tests/test_gpu_timing.py tests/test_gpu_trace.py tests/test_gpu_parity.py -k "relu_bits or timing or bucket or batchnorm or network_features or trainer_iteration or engine_matches or joint_pair or twenty" \
  -m gpu -q --tb=short -p no:cacheprovider -s > $O/pytest_sel.log 2>&1
echo "pytest(sel) exit $?" | tee -a $O/stages.log; grep -E "passed|failed|skipped" $O/pytest_sel.log | tail -3; grep -E "^FAILED|^ERROR|rows per joint" $O/pytest_sel.log | head
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
def load(p):
  rows = {}
  for l in open(p):
    if l.startswith("#") or l.startswith("op"): continue
    f = l.rstrip("\n").split("\t")
    rows[int(f[0])] = f
  return rows
a, b = load("$O/layers_off.tsv.times.tsv"), load("$O/layers_on.tsv.times.tsv")
tf = tb = uf = ub = 0.0
print("BatchNorm ops with >= 40000 rows: fwd ms off -> on | bwd ms off -> on")
for q in sorted(a):
  x, y = a[q], b[q]
  if x[2] != "bn": continue
  f0, f1, b0, b1 = float(x[10]), float(y[10]), float(x[11]), float(y[11])
  tf += f0; uf += f1; tb += b0; ub += b1
  if int(x[7]) >= 40000:
    print("%3d c=%3s rows=%6s | %.4f -> %.4f | %.4f -> %.4f" % (q, x[5], x[7], f0, f1, b0, b1))
print("all BatchNorms: fwd %.3f -> %.3f ms, bwd %.3f -> %.3f ms" % (tf, uf, tb, ub))
for name in ("off", "on"):
  d = json.loads([l for l in open("$O/line_%s.json" % name) if l.startswith("{")][-1])
  print(name, [(f["family"][:22], f["ms_per_step"]) for f in d["families"] if f["family"].startswith("BatchNorm")], "kernels bn:", [k["ms"] for k in d["kernels"] if k["kernel"].startswith("bn_")])
PY
stamp "done"
