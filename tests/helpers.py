"""Shared test helpers (random sparse inputs, dense scatter)."""
import numpy as np
import torch


def random_coords(n, extent=12, batch=2, seed=0, negative=True):
  """n unique (b,x,y,z) rows inside a cube, clustered enough to have neighbours."""
  rng = np.random.RandomState(seed)
  lo = -extent // 2 if negative else 0
  pts = set()
  while len(pts) < n:
    b = rng.randint(batch)
    c = rng.randint(lo, lo + extent, 3)
    pts.add((b, int(c[0]), int(c[1]), int(c[2])))
  arr = np.array(sorted(pts), dtype=np.int32)
  rng.shuffle(arr)
  return arr


def surface_coords(n_side=24, batch=2, seed=0):
  """Voxels of wavy surfaces (ScanNet-like: 2-D manifolds in 3-D), shuffled rows."""
  rng = np.random.RandomState(seed)
  out = []
  for b in range(batch):
    u, v = np.meshgrid(np.arange(n_side), np.arange(n_side))
    z = np.round(3 * np.sin(u / 4.0 + b) + 2 * np.cos(v / 3.0)).astype(int)
    pts = np.stack([u.ravel() - n_side // 2, v.ravel() - n_side // 2, z.ravel()], 1)
    pts2 = np.stack([z.ravel() + 5, u.ravel() - n_side // 2, v.ravel() - n_side // 2], 1)
    p = np.unique(np.concatenate([pts, pts2]), axis=0)
    out.append(np.concatenate([np.full((len(p), 1), b), p], 1))
  arr = np.concatenate(out).astype(np.int32)
  rng.shuffle(arr)
  return arr


def to_dense(coords, feats, origin, shape):
  """[B, C, X, Y, Z] grid with feats scattered at coords - origin."""
  B = int(coords[:, 0].max()) + 1
  g = torch.zeros((B, feats.shape[1]) + tuple(shape), dtype=feats.dtype)
  c = torch.from_numpy(coords.astype(np.int64))
  g[c[:, 0], :, c[:, 1] - origin[0], c[:, 2] - origin[1], c[:, 3] - origin[2]] = feats
  return g
