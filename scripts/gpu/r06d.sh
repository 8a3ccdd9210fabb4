#!/bin/bash
# Round 6, call 4: the offset split by residency rounds (PCMI_KSPLIT_RULE, csrc/spconv.hip: make_plan) against the round-2 rule
# ("2.5 workgroups per CU"): step A/B, per-layer in-step times of both arms, parity of the convolutions it touches.
set -u
ulimit -c 0
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$ROOT"
export TMPDIR=/tmp
TAG=${TAG:-r06d}
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
run() {  # label n env...
  local label=$1 n=$2; shift 2
  for i in $(seq 1 $n); do
    env "$@" timeout 150 $B > $O/ab_${label}_$i.json 2>> $O/bench.err
    line $O/ab_${label}_$i.json "$label run $i"
  done
}
stamp "1 A/B"
run rule_off 3 PCMI_KSPLIT_RULE=0
run rule_on 3 PCMI_NOP=1
run rule_off 1 PCMI_KSPLIT_RULE=0
run rule_on 1 PCMI_NOP=1
stamp "2 per-layer in-step times"
PCMI_KSPLIT_RULE=0 timeout 300 python bench.py --steps 10 --warmup 5 --no-cpu-baseline --no-extra --layer-table $O/layers_rule_off.tsv > $O/line_rule_off.json 2>> $O/bench.err
timeout 300 python bench.py --steps 10 --warmup 5 --no-cpu-baseline --no-extra --layer-table $O/layers_rule_on.tsv > $O/line_rule_on.json 2>> $O/bench.err
python - <<PY | tee $O/layers_ab.txt
def load(p):
  rows = {}
  for l in open(p):
    if l.startswith("#") or l.startswith("op"): continue
    f = l.rstrip("\n").split("\t")
    rows[int(f[0])] = f
  return rows
a, b = load("$O/layers_rule_off.tsv.times.tsv"), load("$O/layers_rule_on.tsv.times.tsv")
print("op layer K cin cout rows_out | fwd ms off -> on | bwd ms off -> on")
tf = tb = uf = ub = 0.0
for q in sorted(a):
  x, y = a[q], b[q]
  if x[2] != "conv": continue
  f0, f1, b0, b1 = float(x[10]), float(y[10]), float(x[11]), float(y[11])
  tf += f0; uf += f1; tb += b0; ub += b1
  if abs(f0 - f1) > 0.003 or abs(b0 - b1) > 0.004:
    print("%3d %-26s %2s %3s %3s %6s | %.4f -> %.4f | %.4f -> %.4f" % (q, x[1], x[3], x[4], x[5], x[7], f0, f1, b0, b1))
print("all convolutions: fwd %.3f -> %.3f ms, bwd-data %.3f -> %.3f ms" % (tf, uf, tb, ub))
PY
stamp "3 parity"
timeout 900 python -m pytest tests/test_gpu_parity.py -k "spconv_parity or spconv_golden or network_features or conv16 or full_size" -m gpu -q --tb=short -p no:cacheprovider > $O/pytest_sel.log 2>&1
echo "pytest(sel) exit $?" | tee -a $O/stages.log; grep -E "passed|failed|skipped" $O/pytest_sel.log | tail -3; grep -E "^FAILED|^ERROR" $O/pytest_sel.log | head
stamp "done"
