#!/bin/bash
# Round 6, call 12: shader-clock stamps around every per-slot barrier of one workgroup of wgrad_x3p_kernel and around the phases of
# one producer wave (-DPCMI_X3_DIAG_STAMP): who waits for whom, and where a producer's step goes.
set -u
ulimit -c 0
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$ROOT"
export TMPDIR=/tmp
O=$ROOT/gpurun_out/${TAG:-r06l}
mkdir -p $O
for v in ${VARIANTS:-stamp stamp_nosplit stamp_noldsw stamp_skeleton}; do
  PCMI_LIB=$ROOT/pointcontrast_amd/libpcmi_wg_$v.so KBENCH_LEVELS=0,1 timeout 200 python scripts/kbench.py > $O/raw_$v.txt 2>&1
done
python - $O <<'PY' | tee $O/stamps.txt
import re, sys, glob, os, collections
for f in sorted(glob.glob(sys.argv[1] + "/raw_*.txt")):
  st, ph = collections.defaultdict(list), collections.defaultdict(list)
  for l in open(f):
    m = re.search(r"x3p stamp: rows (\d+) C (\d+).*consumer work (\d+) wait (\d+) \| producer work (\d+) wait (\d+) \| slot (\d+)", l)
    if m: st[(int(m.group(2)), int(m.group(1)))].append([float(x) for x in m.groups()[2:]])
    m = re.search(r"x3p phases: C (\d+) rows (\d+) .*issue (\d+) \| X convert\+write (\d+) \| G/tables (\d+) \| barrier (\d+)", l)
    if m: ph[(int(m.group(1)), int(m.group(2)))].append([float(x) for x in m.groups()[2:]])
  print("==", os.path.basename(f))
  for k in sorted(st):
    a = [sum(c) / len(st[k]) for c in zip(*st[k])]
    b = [sum(c) / len(ph[k]) for c in zip(*ph[k])] if ph[k] else [0] * 4
    print("  C %3d rows %6d | consumer work %5.0f wait %5.0f | producer work %5.0f wait %4.0f | slot %5.0f | producer phases: issue %5.0f X %5.0f G/tables %5.0f barrier %5.0f" % (k + tuple(a) + tuple(b)))
PY
