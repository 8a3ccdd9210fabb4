#!/bin/bash
set -u
ulimit -c 0
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
mkdir -p gpurun_out
export KBENCH_LEVELS=0,1 KBENCH_SUSTAINED=0
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1
echo "== default"; timeout 200 python scripts/kbench.py 2>&1 | grep "^L" | cut -c1-150
echo "== streamk off"; PCMI_SPCONV_STREAMK=0 timeout 200 python scripts/kbench.py 2>&1 | grep "^L0" | cut -c1-150
echo "== W4 build"
PCMI_EXTRA_HIPCC_FLAGS="-DPCMI_CONV16_W4=1" python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build_w4.log 2>&1 || tail -3 gpurun_out/build_w4.log
PCMI_EXTRA_HIPCC_FLAGS="-DPCMI_CONV16_W4=1" timeout 200 python scripts/kbench.py 2>&1 | grep "^L" | cut -c1-150
PCMI_EXTRA_HIPCC_FLAGS="-DPCMI_CONV16_W4=1" timeout 300 python -m pytest tests -m gpu -q -p no:cacheprovider -k "conv16 or streamk" 2>&1 | tail -2
PCMI_EXTRA_HIPCC_FLAGS="-DPCMI_CONV16_W4=1" timeout 300 python bench.py --steps 20 --warmup 5 --no-roofline --no-cpu-baseline 2>/dev/null | cut -c1-160
echo done
