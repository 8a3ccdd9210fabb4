"""CPU restatement of the loader-side geometry (TEST INFRASTRUCTURE, see oracle/__init__.py).

  sparse_quantize_index : ME.utils.sparse_quantize(xyz / voxel, return_index=True) as used at
                          pc/lib/ddp_data_loaders.py:228-229 (floor, first occurrence, ascending)
  match_radius          : get_matching_indices, pc/lib/ddp_data_loaders.py:36-49 -- every (i, j) with
                          |T p_i - q_j| <= r.  The reference asks an open3d KD-tree (absent here); the RESULT SET is
                          defined by the distances, which this file evaluates in float64 with a fixed operation order
                          (each product / sum separately rounded) so that the device kernels can match bit for bit.
                          tests/test_oracle_loader.py checks it against scipy's cKDTree.
"""
import numpy as np


def sparse_quantize_index(xyz, voxel_size):
  q = np.floor(np.asarray(xyz, dtype=np.float64) / np.float64(voxel_size)).astype(np.int64)
  if len(q) == 0:
    return np.zeros(0, np.int64)
  q0 = q - q.min(0)
  m = q0.max(0) + 1
  key = (q0[:, 0] * m[1] + q0[:, 1]) * m[2] + q0[:, 2]
  _, first = np.unique(key, return_index=True)
  return np.sort(first).astype(np.int64)


def apply_rigid(trans, xyz):
  """((R0 x + R1 y) + R2 z) + t per output coordinate, every operation rounded on its own (no fused multiply-add)."""
  T = np.asarray(trans, dtype=np.float64)
  p = np.asarray(xyz, dtype=np.float64)
  out = np.empty_like(p)
  for r in range(3):
    out[:, r] = ((T[r, 0] * p[:, 0] + T[r, 1] * p[:, 1]) + T[r, 2] * p[:, 2]) + T[r, 3]
  return out


def match_radius(xyz0, trans, xyz1, radius):
  """int64 [P, 2], sorted by (i, j)."""
  src = apply_rigid(trans, xyz0)
  dst = np.asarray(xyz1, dtype=np.float64)
  if len(src) == 0 or len(dst) == 0:
    return np.zeros((0, 2), np.int64)
  r = np.float64(radius)
  r2 = r * r
  cell = np.floor(dst / r).astype(np.int64)
  lo = cell.min(0) - 1
  dims = cell.max(0) - lo + 2

  def key_of(c):
    c = c - lo
    return (c[:, 0] * dims[1] + c[:, 1]) * dims[2] + c[:, 2]

  dkey = key_of(cell)
  order = np.argsort(dkey, kind="stable")
  skey = dkey[order]
  scell = np.floor(src / r).astype(np.int64)
  ii, jj = [], []
  for dz in (-1, 0, 1):
    for dy in (-1, 0, 1):
      for dx in (-1, 0, 1):
        c = scell + np.array([dx, dy, dz])
        ok = ((c - lo) >= 0).all(1) & ((c - lo) < dims).all(1)
        k = key_of(np.where(ok[:, None], c, lo))
        a, b = np.searchsorted(skey, k, "left"), np.searchsorted(skey, k, "right")
        n = np.where(ok, b - a, 0)
        i = np.repeat(np.arange(len(src)), n)
        j = order[np.repeat(a, n) + (np.arange(n.sum()) - np.repeat(np.cumsum(n) - n, n))]
        ex, ey, ez = src[i, 0] - dst[j, 0], src[i, 1] - dst[j, 1], src[i, 2] - dst[j, 2]
        d2 = (ex * ex + ey * ey) + ez * ez
        hit = d2 <= r2
        ii.append(i[hit])
        jj.append(j[hit])
  ii, jj = np.concatenate(ii), np.concatenate(jj)
  o = np.lexsort((jj, ii))
  return np.stack([ii[o], jj[o]], 1).astype(np.int64)
