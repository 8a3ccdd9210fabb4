"""CPU model of the row order of a 3^3 / stride-1 kernel map (sortrows.hip): how many (tile, offset) steps and
16-row-group products the convolution / weight-gradient kernels issue under different sort keys.

The kernels skip an offset for a 128-row tile (spconv16x_kernel: workgroup steps), a 16-row group (issued MFMAs) or a
64-row tile (wgrad_x3p_kernel: slots) only if NO row of the unit has that neighbour, so the cost of an order is the sum
over units of popcount(OR of the row masks).  The ideal is the mean popcount of a row (17.1 on the bench batch).
  python scripts/row_order_model.py [level]        (CPU only)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from scripts.tile_schedule_sim import occupancy_masks

POP = np.array([bin(i).count("1") for i in range(1 << 16)], np.int64)


def popcount(m):
  m = m.astype(np.uint64)
  return POP[m & 0xFFFF] + POP[(m >> 16) & 0xFFFF]


def unit_cost(m, unit):
  n = len(m)
  pad = (-n) % unit
  mm = np.concatenate([m, np.zeros(pad, m.dtype)]).reshape(-1, unit)
  return int(popcount(np.bitwise_or.reduce(mm, axis=1)).sum())


def report(name, mask, perm):
  m = mask[perm]
  n = len(m)
  ideal = popcount(mask).sum()
  print("%-44s  steps128 %7d (x%.3f)  slots64 %7d (x%.3f)  groups16 %8d (x%.3f)" % (
      name, unit_cost(m, 128), unit_cost(m, 128) * 128 / ideal, unit_cost(m, 64), unit_cost(m, 64) * 64 / ideal,
      unit_cost(m, 16), unit_cost(m, 16) * 16 / ideal))


def chunked(n, chunk, keyfn):
  out = []
  for c0 in range(0, n, chunk):
    idx = np.arange(c0, min(n, c0 + chunk))
    out.append(idx[keyfn(idx)])
  return np.concatenate(out)


def bit_permuted(mask, order):
  """key whose MOST significant bit is offset order[0], ..."""
  key = np.zeros(len(mask), np.uint64)
  for rank, k in enumerate(order):
    key |= ((mask.astype(np.uint64) >> np.uint64(k)) & np.uint64(1)) << np.uint64(26 - rank)
  return key


def greedy_bits(mask, idx):
  """recursive bisection: at every node split on the bit that is closest to half set in the node (a decision-tree
  order).  Returns the positions of idx in the new order."""
  order = np.empty(len(idx), np.int64)
  stack = [(np.arange(len(idx)), 0, (1 << 27) - 1)]
  m = mask[idx]
  pos = 0
  out = []
  # iterative DFS, left (bit clear) first
  work = [(np.arange(len(idx)), (1 << 27) - 1)]
  while work:
    rows, avail = work.pop()
    if len(rows) <= 16 or avail == 0:
      out.append(rows[np.argsort(m[rows], kind="stable")])
      continue
    mm = m[rows]
    best, bestd = -1, None
    for k in range(27):
      if not (avail >> k) & 1:
        continue
      f = ((mm >> np.uint32(k)) & 1).mean()
      d = abs(f - 0.5)
      if f > 0 and f < 1 and (bestd is None or d < bestd):
        best, bestd = k, d
    if best < 0:
      out.append(rows)
      continue
    bit = ((mm >> np.uint32(best)) & 1).astype(bool)
    work.append((rows[bit], avail & ~(1 << best)))
    work.append((rows[~bit], avail & ~(1 << best)))
  return np.concatenate(out)


def main():
  ns = bench.parse_args([]) if hasattr(bench, "parse_args") else None
  from pointcontrast_amd.lib.synthetic import make_batch
  b = make_batch(seed=0, batch_size=4, voxel_size=0.025)
  C0, C1 = b["sinput0_C"].astype(np.int64), b["sinput1_C"].astype(np.int64)
  C1 = C1.copy()
  C1[:, 0] += 4
  C = np.concatenate([C0, C1])
  level = int(sys.argv[1]) if len(sys.argv) > 1 else 1
  for _ in range(level - 1):
    q = C.copy()
    q[:, 1:] = np.floor_divide(q[:, 1:], 2)
    _, first = np.unique(q, axis=0, return_index=True)
    C = q[np.sort(first)]
  n = len(C)
  print("rows", n)
  mask = occupancy_masks(C)
  print("mean neighbours per row %.2f" % popcount(mask).mean())
  tiles = -(-n // 128)
  chunk = -(-tiles // 8) * 128
  report("loader order", mask, np.arange(n))
  report("(chunk, mask) -- sortrows.hip", mask, chunked(n, chunk, lambda idx: np.argsort(mask[idx], kind="stable")))
  freq = np.array([((mask >> np.uint32(k)) & 1).mean() for k in range(27)])
  print("offset frequencies", np.round(freq, 2))
  by_half = np.argsort(np.abs(freq - 0.5))  # most balanced bit first
  key = bit_permuted(mask, by_half)
  report("(chunk, bits by |f - 0.5|)", mask, chunked(n, chunk, lambda idx: np.argsort(key[idx], kind="stable")))
  key = bit_permuted(mask, by_half[::-1])
  report("(chunk, reversed)", mask, chunked(n, chunk, lambda idx: np.argsort(key[idx], kind="stable")))
  pc = popcount(mask)
  report("(chunk, popcount, mask)", mask, chunked(n, chunk, lambda idx: np.lexsort((mask[idx], pc[idx]))))
  report("(chunk, decision tree)", mask, chunked(n, chunk, lambda idx: greedy_bits(mask, idx)))
  report("(global, mask)", mask, np.argsort(mask, kind="stable"))
  report("(global, decision tree)", mask, greedy_bits(mask, np.arange(n)))
  for nch in (4, 2):
    ch = -(-tiles // nch) * 128
    report("(%d chunks, decision tree)" % nch, mask, chunked(n, ch, lambda idx: greedy_bits(mask, idx)))


if __name__ == "__main__":
  main()
