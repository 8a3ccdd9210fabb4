#!/bin/bash
set -u
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$ROOT"; export TMPDIR=/tmp
O=$ROOT/gpurun_out/r05d; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.txt 2>&1
timeout 200 python scripts/debug/bn_small_probe.py 2>&1 | grep -v amdgpu.ids | cut -c1-260 > $O/probe.txt
cat $O/probe.txt
