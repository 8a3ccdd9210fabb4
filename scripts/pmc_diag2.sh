#!/bin/bash
# stall / LDS attribution of the dominant kernels: one PMC pass (8 SQ counters) over scripts/pmc_probe.py
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp PMC_PROBE_ONLY=96
mkdir -p gpurun_out/diag2
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/diag2/build.log 2>&1
cd /tmp
for pass in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES SQ_WAVES"; do
  tag=$(echo $pass | cut -d" " -f1)
  timeout 300 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/diag2/$tag" -o pmc -- python "$GRAFT_REPO_ROOT/scripts/pmc_probe.py" > "$GRAFT_REPO_ROOT/gpurun_out/diag2/$tag.log" 2>&1
done
cd "$GRAFT_REPO_ROOT"
find gpurun_out/diag2 -name "*kernel_trace*" -delete
python - <<'PY'
import csv, collections, glob
for f in sorted(glob.glob("gpurun_out/diag2/*/pmc_counter_collection.csv")):
  d = collections.defaultdict(lambda: collections.defaultdict(list))
  for r in csv.DictReader(open(f)):
    d[r["Kernel_Name"].split("(")[0].replace("void pcmi::", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
  for k, v in d.items():
    if "spconv16" in k or "wgrad" in k or "fixup" in k:
      print(k, {c: "%.3g" % (sum(x) / len(x)) for c, x in v.items()})
PY
