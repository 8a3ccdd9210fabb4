"""CPU model of the one-tile-per-workgroup launch of the level-1 3^3 conv ("what bounds it"):
builds the 27-bit occupancy masks of the bench batch, applies the (XCD chunk, mask) row sort of sortrows.hip,
costs every tile by the number of offsets that occur in it and list-schedules the tiles on 256 CUs x 3
resident workgroups (processor sharing inside a CU).  Prints makespan vs the perfectly balanced bound for
128- / 64- / 32-row tiles -- the numbers that motivated the unit-balanced (stream-K) launch.
  python scripts/tile_schedule_sim.py        (CPU only, ~1 min)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench


def occupancy_masks(C):
  n = len(C)
  key = lambda c: ((c[:, 0] * 4096 + (c[:, 1] + 1024)) * 4096 + (c[:, 2] + 1024)) * 4096 + (c[:, 3] + 1024)
  sk = np.sort(key(C))
  mask, bit = np.zeros(n, np.uint32), 0
  for dx in (-1, 0, 1):
    for dy in (-1, 0, 1):
      for dz in (-1, 0, 1):
        q = C.copy()
        q[:, 1] += dx; q[:, 2] += dy; q[:, 3] += dz
        kq = key(q)
        pos = np.minimum(np.searchsorted(sk, kq), n - 1)
        mask |= (sk[pos] == kq).astype(np.uint32) << bit
        bit += 1
  return mask


def tile_costs(mask, TM, chunk):
  n = len(mask)
  perm = np.concatenate([np.arange(c0, min(n, c0 + chunk))[np.argsort(mask[c0:c0 + chunk], kind="stable")]
                         for c0 in range(0, n, chunk)])
  m = mask[perm]
  return np.array([bin(int(np.bitwise_or.reduce(m[t * TM:(t + 1) * TM]))).count("1") for t in range(-(-n // TM))], float)


def makespan(costs, slots_per_cu=3, ncu=256, single_wave_rate=0.7):
  """Workgroups are dispatched in order to the CU with a free slot; the resident workgroups of a CU share its
  matrix pipes equally, one workgroup alone reaches `single_wave_rate` of the pipe."""
  queue = list(costs)[::-1]
  cus = [[] for _ in range(ncu)]
  for _ in range(slots_per_cu):
    for c in range(ncu):
      if queue:
        cus[c].append(queue.pop())
  now = 0.0
  while True:
    best = None
    for c in range(ncu):
      if cus[c]:
        rate = min(single_wave_rate, 1.0 / len(cus[c]))
        tf = min(cus[c]) / rate
        best = tf if best is None else min(best, tf)
    if best is None:
      return now
    now += best
    for c in range(ncu):
      if cus[c]:
        rate = min(single_wave_rate, 1.0 / len(cus[c]))
        left = [w - best * rate for w in cus[c]]
        done = sum(1 for w in left if w <= 1e-9)
        cus[c] = [w for w in left if w > 1e-9]
        for _ in range(done):
          if queue:
            cus[c].append(queue.pop())


if __name__ == "__main__":
  C = bench.get_batch(0, 4, 0.025)["sinput0_C"].numpy().astype(np.int64)
  mask = occupancy_masks(C)
  n = len(C)
  chunk = -(-(-(-n // 128)) // 8) * 128
  print("rows %d, occupied offsets per row %.2f" % (n, np.mean([bin(int(m)).count("1") for m in mask[:20000]])))
  for TM in (128, 64, 32):
    cost = tile_costs(mask, TM, chunk) * (TM / 128.0)
    T, xt = len(cost), -(-len(cost) // 8)
    order = [(b & 7) * xt + (b >> 3) for b in range(8 * xt)]  # the XCD-contiguous tile order of spconv.hip
    mk = makespan([cost[t] for t in order if t < T])
    print("%3d-row tiles: %5d tiles, %8.0f tile-units in total, balanced bound %6.1f per CU, makespan %6.1f (%.0f %%)"
          % (TM, T, cost.sum(), cost.sum() / 256, mk, 100 * cost.sum() / 256 / mk))
