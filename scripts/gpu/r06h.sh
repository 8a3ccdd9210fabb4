#!/bin/bash
# Round 6, call 8: A/B of the reduction / table kernels with all loads of a batch in flight (split_reduce, sk_fixup,
# wgrad_slab_sum, wgrad_reduce, colsum_partial, the unit-balanced launch's table build) against the committed forms of the same
# four sources (libpcmi_oldreduce.so: the previous commit's spconv.hip / spconv_x3.hip / spconv_wgrad.hip / spconv_wgrad_x3.hip,
# everything else identical), alternating processes; then relu_bits off on the new build; kernel stats of both builds.
set -u
ulimit -c 0
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$ROOT"
export TMPDIR=/tmp
TAG=${TAG:-r06h}
O=$ROOT/gpurun_out/$TAG
mkdir -p $O
T0=$(date +%s)
stamp() { echo "[$(( $(date +%s) - T0 )) s] $*" | tee -a $O/stages.log; }
python -c "import __graft_entry__ as g; g.build()" > $O/build.txt 2>&1
OLD=$ROOT/pointcontrast_amd/libpcmi_oldreduce.so
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
stamp "1 tests (new build)"
timeout 900 python -m pytest tests/test_gpu_timing.py tests/test_gpu_parity.py tests/test_gpu_fullsize.py -k "relu_bits or spconv_parity or streamk or batchnorm or network_features or bit_reproducible or wgrad or conv16 or grouped or dense_1x1 or stem" \
  -m gpu -q --tb=short -p no:cacheprovider > $O/pytest_sel.log 2>&1
echo "pytest(sel) exit $?" | tee -a $O/stages.log; grep -E "passed|failed|skipped" $O/pytest_sel.log | tail -3; grep -E "^FAILED|^ERROR" $O/pytest_sel.log | head -20
stamp "2 A/B old / new reduction kernels"
for i in 1 2 3 4; do
  ENVV="PCMI_LIB=$OLD" run1 old_reduce $i
  ENVV="PCMI_NOP=1" run1 new_reduce $i
done
stamp "3 relu_bits off on the new build"
for i in 1 2; do
  ENVV="PCMI_BN_RELU_BITS=0" run1 new_fp32_mask $i
done
stamp "4 kernel stats, both builds"
for arm in old new; do
  ( cd /tmp && if [ $arm = old ]; then export PCMI_LIB=$OLD; fi; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/prof_$arm" -o bench -- \
      python "$ROOT/bench.py" --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-extra > "$O/prof_$arm.log" 2>&1 )
  find $O/prof_$arm -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats_$arm.csv \;
  rm -rf $O/prof_$arm
done
python - <<PY | tee $O/kernel_ab.txt
import csv
def load(p):
  d = {}
  for r in csv.DictReader(open(p)):
    d[r["Name"]] = (int(r["Calls"]), float(r["TotalDurationNs"]))
  return d
a, b = load("$O/kernel_stats_old.csv"), load("$O/kernel_stats_new.csv")
st = lambda d: [v[0] for k, v in d.items() if "sgd_kernel" in k][0]
def fam(d, keys):
  return sum(v[1] for k, v in d.items() if any(x in k for x in keys)) / st(d) / 1e6, sum(v[0] for k, v in d.items() if any(x in k for x in keys)) / st(d)
print("ms per step of kernel time (launches): old -> new")
for name, keys in [("split_reduce", ["split_reduce"]), ("sk_fixup", ["sk_fixup"]), ("wgrad_slab_sum", ["slab_sum"]), ("wgrad_reduce", ["wgrad_reduce"]), ("colsum_partial", ["colsum_partial"]),
                   ("spconv16x unit-balanced", ["spconv16x_kernel<3, true", "spconv16x_kernel<4, true"]), ("spconv16x offset split", ["spconv16x_kernel<2, false", "spconv16x_kernel<3, false", "spconv16x_kernel<4, false"]),
                   ("wgrad_x3p", ["wgrad_x3p"]), ("BatchNorm", ["bn_", "colreduce"]), ("everything", [""])]:
  x, y = fam(a, keys), fam(b, keys)
  print("  %-26s %.3f (%.1f) -> %.3f (%.1f)" % (name, x[0], x[1], y[0], y[1]))
PY
stamp "done"
