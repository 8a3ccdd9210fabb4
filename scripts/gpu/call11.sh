#!/bin/bash
# round 4, GPU call 11: what bounds the producer / consumer weight-gradient kernel (components compiled out, level 0 stand-alone)
set -u
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$ROOT"
export TMPDIR=/tmp
O=$ROOT/gpurun_out/r04k
mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.txt 2>&1
for v in base dg_nogather dg_nosplit dg_nomfma dg_noldsw dg_nosplit_noldsw_nogather; do
  L=$ROOT/pointcontrast_amd/libpcmi_$v.so; [ $v = base ] && L=$ROOT/pointcontrast_amd/libpcmi.so
  PCMI_LIB=$L KBENCH_SUSTAINED=0 KBENCH_LEVELS=0 timeout 100 python scripts/kbench.py > $O/kbench_$v.txt 2>&1
  echo "== $v"; grep -h "^L0 3^3 \(96->96\|128->96\)" $O/kbench_$v.txt | cut -c100-150
done
echo done
