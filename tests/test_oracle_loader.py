"""oracle/loader_ref.py (the CPU restatement the device loader kernels are compared with) against independent
implementations: scipy's cKDTree for the radius search, the synthetic generator's own first-occurrence sub-sampling."""
import numpy as np
from scipy.spatial import cKDTree

from oracle import loader_ref as lf
from pointcontrast_amd.lib import synthetic


def _pair(seed, crop):
  rng = np.random.RandomState(seed)
  a, b = synthetic.make_frame_pair(rng)
  c = a[rng.randint(len(a))]
  a, b = a[np.linalg.norm(a - c, axis=1) < crop], b[np.linalg.norm(b - c, axis=1) < crop]
  T0, T1 = synthetic.sample_random_trans(a, rng), synthetic.sample_random_trans(b, rng)
  a, b = a @ T0[:3, :3].T + T0[:3, 3], b @ T1[:3, :3].T + T1[:3, 3]
  return a, b, T1 @ np.linalg.inv(T0)


def test_first_occurrence_index():
  a, _, _ = _pair(0, 0.8)
  sel = lf.sparse_quantize_index(a, 0.025)
  assert (sel == synthetic.sparse_quantize_index(a / 0.025)).all()
  q = np.floor(a / 0.025).astype(np.int64)
  assert len(np.unique(q[sel], axis=0)) == len(sel) == len(np.unique(q, axis=0)) and (np.diff(sel) > 0).all()
  for i in sel[:200]:  # really the FIRST point of its voxel
    assert not (q[:i] == q[i]).all(1).any()
  assert len(lf.sparse_quantize_index(np.zeros((0, 3)), 0.025)) == 0


def test_match_radius_against_kdtree():
  a, b, T = _pair(1, 0.7)
  a, b = a[lf.sparse_quantize_index(a, 0.025)], b[lf.sparse_quantize_index(b, 0.025)]
  r = 1.5 * 0.025
  got = lf.match_radius(a, T, b, r)
  src = a @ T[:3, :3].T + T[:3, 3]
  nb = cKDTree(b).query_ball_point(src, r)
  want = np.array([(i, j) for i, js in enumerate(nb) for j in sorted(js)], dtype=np.int64).reshape(-1, 2)
  assert len(got) > 1000
  # identical sets up to pairs whose distance is within round-off of the radius (the two sides round differently)
  gs, ws = set(map(tuple, got.tolist())), set(map(tuple, want.tolist()))
  for i, j in gs ^ ws:
    assert abs(np.linalg.norm(src[i] - b[j]) - r) < 1e-12
  assert len(gs ^ ws) <= 2
  assert (np.lexsort((got[:, 1], got[:, 0])) == np.arange(len(got))).all(), "sorted by (i, j)"
  assert lf.match_radius(a[:0], T, b, r).shape == (0, 2) and lf.match_radius(a, T, b[:0], r).shape == (0, 2)
