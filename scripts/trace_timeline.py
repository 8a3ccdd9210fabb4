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


if __name__ == "__main__" and "--phases" not in sys.argv:
  main(sys.argv[1])


def phases(path):
  """Second view (python scripts/trace_timeline.py trace.csv --phases): per training step, the compute queue's forward
  part (up to the loss kernels) and backward part (from them to sgd_kernel), each as kernel time + gaps, and how far the
  weight-gradient queue trails the end of the backward chain (the part of it the optimiser has to wait for)."""
  rows = []
  with open(path) as f:
    for r in csv.DictReader(f):
      rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r["Queue_Id"]))
  rows.sort()
  sgd = [(s, e, q) for s, e, n, q in rows if "sgd_kernel" in n]
  if len(sgd) < 2:
    print("no sgd steps in the trace")
    return
  mainq = sgd[0][2]
  acc = defaultdict(float)
  cnt = 0
  for i in range(1, len(sgd)):
    t0, t1 = sgd[i - 1][1], sgd[i][0]
    ks = [(s, e, n, q) for s, e, n, q in rows if s >= t0 and e <= t1]
    main = [(s, e, n) for s, e, n, q in ks if q == mainq]
    loss = [(s, e) for s, e, n in main if "nce_" in n or "hardest" in n or "pdist" in n]
    if not main or not loss:
      continue
    lf, ll = min(s for s, e in loss), max(e for s, e in loss)
    fwd = [(s, e, n) for s, e, n in main if e <= lf]
    bwd = [(s, e, n) for s, e, n in main if s >= ll]
    wg = [(s, e) for s, e, n, q in ks if "wgrad" in n or "slab" in n]
    d = {"step_ms": (sgd[i][1] - sgd[i - 1][1]) / 1e6,
         "fwd_span_ms": (lf - fwd[0][0]) / 1e6 if fwd else 0.0, "fwd_kernel_ms": sum(e - s for s, e, n in fwd) / 1e6, "fwd_launches": len(fwd),
         "loss_span_ms": (ll - lf) / 1e6,
         "bwd_span_ms": (t1 - ll) / 1e6, "bwd_kernel_ms": sum(e - s for s, e, n in bwd) / 1e6, "bwd_launches": len(bwd),
         "wgrad_kernel_ms": sum(e - s for s, e in wg) / 1e6, "wgrad_busy_ms": union(wg) / 1e6,
         "wgrad_tail_after_chain_ms": (max(e for s, e in wg) - max(e for s, e, n in bwd)) / 1e6 if wg and bwd else 0.0,
         "idle_before_fwd_ms": (fwd[0][0] - t0) / 1e6 if fwd else 0.0}
    # kernel time of the chain by kernel family, forward / backward
    for tag, part in (("fwd", fwd), ("bwd", bwd)):
      for s, e, n in part:
        fam = ("conv16x_sk" if "spconv16x_kernel<3, true" in n or "spconv16x_kernel<4, true" in n else
               "conv16x" if "spconv16x" in n else "conv_other" if ("spconv" in n or "stem" in n) else
               "bn_stats" if "colreduce" in n else "bn_apply" if "bn_" in n else "split_reduce" if "split_reduce" in n else
               "sk_fixup" if "fixup" in n else "other")
        d["%s.%s_ms" % (tag, fam)] = d.get("%s.%s_ms" % (tag, fam), 0.0) + (e - s) / 1e6
    for k, v in d.items():
      acc[k] += v
    cnt += 1
  print("%d steps averaged (compute queue %s)" % (cnt, mainq))
  for k in sorted(acc, key=lambda x: (("." in x), x)):
    print("  %-28s %8.3f" % (k, acc[k] / cnt))


if __name__ == "__main__" and "--phases" in sys.argv:
  phases(sys.argv[1])
