"""Per-iteration timeline from a rocprofv3 --kernel-trace CSV of bench.py: GPU-busy time (union of kernel
intervals), per-queue busy time, and the gaps, per training step (delimited by sgd_kernel launches).
Usage: python scripts/trace_timeline.py gpurun_out/prof/bench_kernel_trace.csv"""
import csv
import sys
from collections import defaultdict


def union(iv):
  iv = sorted(iv)
  tot, cs, ce = 0, None, None
  for s, e in iv:
    if cs is None:
      cs, ce = s, e
    elif s <= ce:
      ce = max(ce, e)
    else:
      tot += ce - cs
      cs, ce = s, e
  if cs is not None:
    tot += ce - cs
  return tot


def main(path):
  rows = []
  with open(path) as f:
    for r in csv.DictReader(f):
      rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r["Queue_Id"], r.get("Stream_Id", "")))
  rows.sort()
  sgd = [e for s, e, n, q, st in rows if "sgd_kernel" in n]
  print("%d kernels, %d sgd steps" % (len(rows), len(sgd)))
  for i in range(1, len(sgd)):
    t0, t1 = sgd[i - 1], sgd[i]
    ks = [(s, e, n, q) for s, e, n, q, st in rows if s >= t0 and e <= t1]
    busy = union([(s, e) for s, e, n, q in ks])
    perq = defaultdict(list)
    for s, e, n, q in ks:
      perq[q].append((s, e))
    ksum = sum(e - s for s, e, n, q in ks)
    cls = defaultdict(int)
    for s, e, n, q in ks:
      key = ("spconv" if ("spconv_mfma" in n or "spconv16" in n or "stem" in n) else "wgrad" if "wgrad" in n else "bn" if ("bn_" in n or "colreduce" in n or "colsum" in n) else
             "reduce" if ("reduce" in n or "fixup" in n) else "coords" if any(x in n for x in ("kmap", "scan", "sort", "insert", "stride", "mask", "permute", "tile", "rocprim")) else "other")
      cls[key] += e - s
    print("step %d: wall %.2f ms | GPU busy (any kernel) %.2f ms = %.0f%% | sum of kernel time %.2f ms | %d kernels | queues: %s | %s" %
          (i, (t1 - t0) / 1e6, busy / 1e6, 100.0 * busy / (t1 - t0), ksum / 1e6, len(ks),
           " ".join("q%s=%.1f" % (q, union(v) / 1e6) for q, v in sorted(perq.items())),
           " ".join("%s=%.1f" % (k, v / 1e6) for k, v in sorted(cls.items()))))


if __name__ == "__main__":
  main(sys.argv[1])
