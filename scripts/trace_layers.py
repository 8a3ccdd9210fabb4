"""Per-launch timeline of ONE training step from a rocprofv3 --kernel-trace CSV of bench.py: every kernel of the compute
queue in launch order with its start (us from the step's first kernel), duration, the gap to the previous kernel's end on
the same queue, grid size and a short name -- forward part, loss, backward part -- and next to each backward kernel how
many weight-gradient kernels were running at its start.  Then the same step aggregated per family.
Usage: python scripts/trace_layers.py kernel_trace.csv [step]"""
import csv
import re
import sys
from collections import defaultdict


def short(n):
  n = re.sub(r"^void ", "", n)
  n = re.sub(r"pcmi::(\(anonymous namespace\)::)?", "", n)
  m = re.match(r"([A-Za-z0-9_:]+(<[^>(]*>)?)", n)
  return (m.group(1) if m else n)[:44]


def main(path, step=None):
  rows = []
  with open(path) as f:
    for r in csv.DictReader(f):
      grid = r.get("Grid_Size_X") or r.get("Grid_Size") or "0"
      wg = r.get("Workgroup_Size_X") or r.get("Workgroup_Size") or "1"
      try:
        wgs = int(grid) * int(r.get("Grid_Size_Y") or 1) * int(r.get("Grid_Size_Z") or 1) // max(
            1, int(wg) * int(r.get("Workgroup_Size_Y") or 1) * int(r.get("Workgroup_Size_Z") or 1))
      except ValueError:
        wgs = 0
      rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r["Queue_Id"], wgs))
  rows.sort()
  sgd = [(s, e, q) for s, e, n, q, w in rows if "sgd_kernel" in n]
  if len(sgd) < 3:
    print("fewer than three sgd steps in the trace")
    return
  i = step if step is not None else len(sgd) // 2
  mainq = sgd[0][2]
  t0, t1 = sgd[i - 1][1], sgd[i][1]
  ks = [r for r in rows if r[0] >= t0 and r[1] <= t1]
  main = [r for r in ks if r[3] == mainq]
  side = [r for r in ks if r[3] != mainq and ("wgrad" in r[2] or "slab" in r[2])]
  origin = main[0][0]
  print("step %d: %.3f ms, %d kernels on the compute queue, %d weight-gradient kernels elsewhere" %
        (i, (t1 - t0) / 1e6, len(main), len(side)))
  print("%9s %8s %7s %7s %3s  %s" % ("start_us", "dur_us", "gap_us", "wgs", "wg+", "kernel"))
  prev_end = None
  fam = defaultdict(lambda: [0, 0.0, 0.0])
  phase = "fwd"
  for s, e, n, q, w in main:
    if "nce_" in n or "hardest" in n or "pdist" in n:
      phase = "loss"
    elif phase == "loss" and not ("gather" in n or "scatter" in n or "elementwise" in n or "fill" in n.lower()):
      phase = "bwd"
    gap = (s - prev_end) / 1e3 if prev_end is not None else 0.0
    prev_end = e
    conc = sum(1 for ss, ee, nn, qq, ww in side if ss <= s < ee)
    sn = short(n)
    print("%9.1f %8.1f %7.1f %7d %3d  %s %s" % ((s - origin) / 1e3, (e - s) / 1e3, gap, w, conc, phase, sn))
    f = fam[(phase, sn)]
    f[0] += 1
    f[1] += (e - s) / 1e3
    f[2] += max(gap, 0.0)
  print("\nper family (phase, kernel): launches, kernel us, gap us in front")
  for k in sorted(fam, key=lambda k: -fam[k][1]):
    print("  %-4s %-44s %4d %9.1f %8.1f" % (k[0], k[1], fam[k][0], fam[k][1], fam[k][2]))
  for ph in ("fwd", "loss", "bwd"):
    print("  %s total: %d launches, %.1f us of kernels, %.1f us of gaps" %
          (ph, sum(v[0] for k, v in fam.items() if k[0] == ph), sum(v[1] for k, v in fam.items() if k[0] == ph),
           sum(v[2] for k, v in fam.items() if k[0] == ph)))
  print("\nweight-gradient queue(s): %d kernels, %.1f us" % (len(side), sum(e - s for s, e, n, q, w in side) / 1e3))
  for s, e, n, q, w in side:
    print("%9.1f %8.1f %7d  %s" % ((s - origin) / 1e3, (e - s) / 1e3, w, short(n)))


if __name__ == "__main__":
  main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else None)
