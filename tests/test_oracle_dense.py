"""Pins the oracle's sparse conv semantics against dense torch convolutions:
MinkowskiEngine itself is absent, so this is the independent anchor of Appendix A3-A5."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import sparse_ref as sr
from helpers import random_coords, to_dense


def _dense_kernel(W, offs, ksize):
  """[K,cin,cout] sparse kernel -> [cout,cin,k,k,k] dense kernel (index = offset - min offset)."""
  K, cin, cout = W.shape
  lo = offs.min()
  Wd = torch.zeros(cout, cin, ksize, ksize, ksize, dtype=W.dtype)
  for k in range(K):
    o = offs[k] - lo
    Wd[:, :, o[0], o[1], o[2]] = W[k].t()
  return Wd


@pytest.mark.parametrize("region", [sr.HYPERCUBE, sr.HYBRID])
def test_k3s1_equals_dense_conv3d(region):
  torch.manual_seed(0)
  coords = random_coords(300, extent=10, batch=2, seed=1)
  cin, cout = 5, 7
  feats = torch.randn(len(coords), cin, dtype=torch.float64)
  W = torch.randn(27, cin, cout, dtype=torch.float64)
  cm = sr.CoordsManagerRef(coords)
  km = cm.kernel_map(0, 0, 3, region)
  out = sr.sparse_conv(feats, W, km)
  origin = coords[:, 1:].min(0)
  shape = coords[:, 1:].max(0) - origin + 1
  dense = F.conv3d(to_dense(coords, feats, origin, shape), _dense_kernel(W, sr.region_offsets(3, region), 3), padding=1)
  c = torch.from_numpy(coords.astype(np.int64))
  ref = dense[c[:, 0], :, c[:, 1] - origin[0], c[:, 2] - origin[1], c[:, 3] - origin[2]]
  assert torch.allclose(out, ref, atol=1e-10)


def test_k2s2_equals_dense_strided_conv3d_with_negative_coords():
  torch.manual_seed(1)
  coords = random_coords(250, extent=12, batch=2, seed=2, negative=True)
  cin, cout = 4, 6
  feats = torch.randn(len(coords), cin, dtype=torch.float64)
  W = torch.randn(8, cin, cout, dtype=torch.float64)
  cm = sr.CoordsManagerRef(coords)
  ck = cm.stride(0, 2)
  out = sr.sparse_conv(feats, W, cm.kernel_map(0, ck, 2))
  lo = coords[:, 1:].min(0)
  origin = np.floor_divide(lo, 2) * 2  # even origin: floor-division parents line up with the grid
  shape = coords[:, 1:].max(0) - origin + 1
  shape = shape + shape % 2
  dense = F.conv3d(to_dense(coords, feats, origin, shape), _dense_kernel(W, sr.region_offsets(2), 2), stride=2)
  cc = torch.from_numpy(cm.coords[ck].astype(np.int64))
  ref = dense[cc[:, 0], :, (cc[:, 1] - origin[0]) // 2, (cc[:, 2] - origin[1]) // 2, (cc[:, 3] - origin[2]) // 2]
  assert torch.allclose(out, ref, atol=1e-10)
  # every coarse cell holds >= 1 child and every child has exactly one parent/offset
  assert sum(len(p[0]) for p in cm.kernel_map(0, ck, 2).pairs) == len(coords)
  # the number of coarse rows equals the number of non-empty dense output cells
  occupied = F.conv3d(to_dense(coords, torch.ones(len(coords), 1, dtype=torch.float64), origin, shape),
                      torch.ones(1, 1, 2, 2, 2, dtype=torch.float64), stride=2)
  assert int((occupied > 0).sum()) == len(cc)


def test_k2s2_transpose_equals_dense_conv_transpose3d():
  torch.manual_seed(2)
  coords = random_coords(200, extent=10, batch=1, seed=3)
  cin, cout = 3, 5
  cm = sr.CoordsManagerRef(coords)
  ck = cm.stride(0, 2)
  coarse = cm.coords[ck]
  feats = torch.randn(len(coarse), cin, dtype=torch.float64)
  W = torch.randn(8, cin, cout, dtype=torch.float64)
  out = sr.sparse_conv(feats, W, cm.kernel_map(0, ck, 2).swapped())
  origin = np.floor_divide(coords[:, 1:].min(0), 2) * 2
  shape_c = (coarse[:, 1:].max(0) - origin) // 2 + 1
  gc = torch.zeros((1, cin) + tuple(shape_c), dtype=torch.float64)
  c = torch.from_numpy(coarse.astype(np.int64))
  gc[c[:, 0], :, (c[:, 1] - origin[0]) // 2, (c[:, 2] - origin[1]) // 2, (c[:, 3] - origin[2]) // 2] = feats
  offs = sr.region_offsets(2)
  Wt = torch.zeros(cin, cout, 2, 2, 2, dtype=torch.float64)
  for k in range(8):
    Wt[:, :, offs[k][0], offs[k][1], offs[k][2]] = W[k]
  dense = F.conv_transpose3d(gc, Wt, stride=2)
  f = torch.from_numpy(coords.astype(np.int64))
  ref = dense[f[:, 0], :, f[:, 1] - origin[0], f[:, 2] - origin[1], f[:, 3] - origin[2]]
  assert torch.allclose(out, ref, atol=1e-10)


def test_three_voxel_known_answer():
  """Hand-computable: voxels at x=0,1,3 on a line, cin=cout=1, weight slice k = value 10*k+1."""
  coords = np.array([[0, 0, 0, 0], [0, 1, 0, 0], [0, 3, 0, 0]], dtype=np.int32)
  feats = torch.tensor([[1.0], [2.0], [4.0]])
  W = (torch.arange(27, dtype=torch.float32) * 10 + 1).view(27, 1, 1)
  cm = sr.CoordsManagerRef(coords)
  out = sr.sparse_conv(feats, W, cm.kernel_map(0, 0, 3))
  # HYPERCUBE, axis 0 fastest: offset (-1,0,0) is k=12, centre k=13, (+1,0,0) is k=14
  assert torch.equal(out, torch.tensor([[1 * 131.0 + 2 * 141.0], [1 * 121.0 + 2 * 131.0], [4 * 131.0]]))
  ck = cm.stride(0, 2)
  assert cm.coords[ck].tolist() == [[0, 0, 0, 0], [0, 2, 0, 0]]  # first-occurrence order
  W2 = (torch.arange(8, dtype=torch.float32) + 1).view(8, 1, 1)
  out2 = sr.sparse_conv(feats, W2, cm.kernel_map(0, ck, 2))
  assert torch.equal(out2, torch.tensor([[1 * 1.0 + 2 * 2.0], [4 * 2.0]]))  # x offset 1 is k=1


def test_hybrid_offsets_order_and_mirror():
  h = sr.region_offsets(3, sr.HYBRID)
  assert h[:9].tolist() == [[0, 0, 0], [-1, 0, 0], [1, 0, 0], [0, -1, 0], [0, 1, 0], [-1, -1, 0], [-1, 1, 0], [1, -1, 0],
                            [1, 1, 0]]
  assert len({tuple(o) for o in h.tolist()}) == 27
  c = sr.region_offsets(3, sr.HYPERCUBE)
  assert c[0].tolist() == [-1, -1, -1] and c[1].tolist() == [0, -1, -1] and c[13].tolist() == [0, 0, 0]
  assert (c[::-1] == -c).all()  # hypercube mirror is index reversal


def test_sparse_quantize_first_occurrence():
  pts = np.array([[0.2, 0.1, 0.0], [1.7, 0, 0], [0.9, 0.5, 0.3], [-0.1, 0, 0], [1.2, 0.9, 0.9], [-0.9, 0.2, 0.1]])
  assert sr.sparse_quantize(pts).tolist() == [0, 1, 3]


def test_oracle_gradcheck_fp64():
  torch.manual_seed(3)
  coords = random_coords(40, extent=5, batch=1, seed=4)
  cm = sr.CoordsManagerRef(coords)
  km = cm.kernel_map(0, 0, 3, sr.HYBRID)
  x = torch.randn(len(coords), 2, dtype=torch.float64, requires_grad=True)
  W = torch.randn(27, 2, 3, dtype=torch.float64, requires_grad=True)
  assert torch.autograd.gradcheck(lambda a, b: sr.sparse_conv(a, b, km), (x, W), atol=1e-6)
