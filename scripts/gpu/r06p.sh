#!/bin/bash
# Round 6, call 16: wave priorities inside wgrad_x3p_kernel (s_setprio: multiplying waves 3 / 1, staging waves 3) -- stamps per slot
# and stand-alone times; the two unstamped candidates also in the step.
set -u
ulimit -c 0
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$ROOT"
export TMPDIR=/tmp
O=$ROOT/gpurun_out/${TAG:-r06p}
mkdir -p $O
for v in base c3 c1 p3; do
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
for v in product c3_nostamp p3_nostamp; do
  if [ $v = product ]; then L=$ROOT/pointcontrast_amd/libpcmi.so; else L=$ROOT/pointcontrast_amd/libpcmi_wg_$v.so; fi
  echo "== $v" | tee -a $O/kbench.txt
  PCMI_LIB=$L KBENCH_LEVELS=0,1 timeout 200 python scripts/kbench.py 2>&1 | grep -E "3\^3 (128|96)" | sed 's/ fwd .*| wgrad/ wgrad/' | tee -a $O/kbench.txt
done
B="python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-roofline --no-extra"
for i in 1 2; do
  for v in product c3_nostamp p3_nostamp; do
    if [ $v = product ]; then L=$ROOT/pointcontrast_amd/libpcmi.so; else L=$ROOT/pointcontrast_amd/libpcmi_wg_$v.so; fi
    PCMI_LIB=$L timeout 150 $B > $O/ab_${v}_$i.json 2>> $O/bench.err
    python - $O/ab_${v}_$i.json "$v run $i" <<'PY' | tee -a $O/ab.txt
import json, sys
try:
  d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1]); print(sys.argv[2], "|", d["value"], "pairs/s", d["ms_per_step"], "ms | loss", d["config"]["final_loss"])
except Exception as e:
  print(sys.argv[2], "failed:", e)
PY
  done
done
