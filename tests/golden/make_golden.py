"""Generates tests/golden/golden_small.npz from the CPU oracle (run from the repo root:
`python tests/golden/make_golden.py`).  The reference itself cannot run here (MinkowskiEngine
absent), so these vectors pin the ORACLE (itself pinned against dense torch conv3d) and give the
GPU tests a committed, seed-independent target for the integer tables and a few float outputs."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle import sparse_ref as sr, loss_ref as lr  # noqa: E402
from helpers import surface_coords  # noqa: E402


def main():
  coords = surface_coords(n_side=14, batch=2, seed=7)
  cm = sr.CoordsManagerRef(coords)
  out = {"coords": coords}
  key = 0
  for lvl in range(3):
    for name, region in (("cube", sr.HYPERCUBE), ("hybrid", sr.HYBRID)):
      out["nbr_%s_l%d" % (name, lvl)] = cm.kernel_map(key, key, 3, region).nbr
    ck = cm.stride(key, 2)
    out["coords_l%d" % (lvl + 1)] = cm.coords[ck]
    km = cm.kernel_map(key, ck, 2)
    out["child_l%d" % lvl] = km.nbr
    out["s2_offs_l%d" % lvl] = km.offs
    key = ck
  g = torch.Generator().manual_seed(11)
  n = len(coords)
  x = torch.randn(n, 32, generator=g)
  W = torch.randn(27, 32, 64, generator=g) * 0.1
  out["x"], out["W"] = x.numpy(), W.numpy()
  out["y_hybrid"] = sr.sparse_conv(x, W, cm.kernel_map(0, 0, 3, sr.HYBRID)).numpy()
  k1 = cm.key_at_stride(2)
  W2 = torch.randn(8, 32, 32, generator=g) * 0.2
  out["W2"] = W2.numpy()
  y2 = sr.sparse_conv(x, W2, cm.kernel_map(0, k1, 2))
  out["y_down"] = y2.numpy()
  out["y_up"] = sr.sparse_conv(y2, W2, cm.kernel_map(0, k1, 2).swapped()).numpy()
  q = torch.nn.functional.normalize(torch.randn(300, 32, generator=g), dim=1)
  k = torch.nn.functional.normalize(torch.randn(300, 32, generator=g), dim=1)
  out["q"], out["k"] = q.numpy(), k.numpy()
  out["nce_T0.4"] = lr.nce_loss(q, k, torch.arange(300), torch.arange(300), 0.4).numpy()
  out["nce_T0.07"] = lr.nce_loss(q, k, torch.arange(300), torch.arange(300), 0.07).numpy()
  path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden_small.npz")
  np.savez_compressed(path, **out)
  print(path, {k: getattr(v, "shape", None) for k, v in out.items()})


if __name__ == "__main__":
  main()
