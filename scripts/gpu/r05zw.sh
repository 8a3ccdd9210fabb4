#!/bin/bash
# Round 5, after the closing call: the driver's command once more on the committed tree, with profiles/pmc_traffic.json now
# carrying that tree's kernel-source digest (roofline.traffic filled from the PMC passes of r05z).
set -u
ulimit -c 0
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$ROOT"
export TMPDIR=/tmp
O=$ROOT/gpurun_out/r05zw
mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.txt 2>&1
timeout 600 python bench.py > $O/bench_line.json 2> $O/bench.err
echo "bench exit $?"; cut -c1-200 $O/bench_line.json
python - <<'PY'
import json
d = json.load(open("gpurun_out/r05zw/bench_line.json"))
print(d["value"], d["ms_per_step"], d["roofline"]["traffic"], d.get("roofline_other", {}).get("traffic"))
PY
