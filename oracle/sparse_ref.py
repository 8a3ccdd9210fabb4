"""CPU restatement of the MinkowskiEngine-0.4.3 behaviour the hot path relies on.

TEST INFRASTRUCTURE (see oracle/__init__.py).  numpy for the integer work,
torch-CPU (autograd) for the float work.  Each function cites the reference
call site that depends on it and the SURVEY.md Appendix-A item it encodes.

Parity status: ME itself is absent -> ME-specific conventions are *unpinned*;
conv / strided conv / transposed conv semantics are pinned against dense torch
conv3d in tests/test_oracle_dense.py.
"""
import numpy as np
import torch

HYPERCUBE = 0  # ME.RegionType.HYPERCUBE (pc/model/modules/common.py:50-52)
HYPERCROSS = 1
HYBRID = 3     # ME.RegionType.HYBRID    (pc/model/modules/common.py:59)

_OFF = 1 << 17  # per-axis bias of the packed key (|coord| < 2^17)


def pack_keys(coords):
  """(b,x,y,z) int rows -> one int64 key per row (order-preserving per field)."""
  c = np.asarray(coords, dtype=np.int64)
  return (c[:, 0] << 54) | ((c[:, 1] + _OFF) << 36) | ((c[:, 2] + _OFF) << 18) | (c[:, 3] + _OFF)


def region_offsets(kernel_size, region=HYPERCUBE, D=3):
  """Kernel offsets in weight-slice order (Appendix A7), unit = one tensor stride.

  HYPERCUBE: axis 0 fastest; odd size s -> {-(s-1)/2..(s-1)/2}, even -> {0..s-1}.
  HYBRID with all-cube axes (block convs, pc/model/modules/common.py:108-114):
  centre first, then per axis every existing offset copied with that axis set to
  each non-centre value.
  """
  s = int(kernel_size)
  centre = (s - 1) // 2 if s % 2 == 1 else 0
  if region == HYPERCUBE:
    K = s ** D
    offs = np.zeros((K, D), dtype=np.int32)
    for k in range(K):
      r = k
      for d in range(D):
        offs[k, d] = r % s - centre
        r //= s
    return offs
  if region == HYBRID:
    lst = [[0] * D]
    for d in range(D):
      new = []
      for o in lst:
        for v in range(s):
          if v == centre:
            continue
          o2 = list(o)
          o2[d] = v - centre
          new.append(o2)
      lst.extend(new)
    return np.asarray(lst, dtype=np.int32)
  raise ValueError("region %r not used by the hot path" % (region,))


def sparse_quantize(coords, return_index=True):
  """ME.utils.sparse_quantize (pc/lib/ddp_data_loaders.py:228-229, Appendix A10):
  floor -> int32 -> ascending indices of the first occurrence of each voxel."""
  q = np.floor(np.asarray(coords)).astype(np.int32)
  keys = pack_keys(np.concatenate([np.zeros((len(q), 1), np.int32), q], 1))
  _, first = np.unique(keys, return_index=True)
  first = np.sort(first)
  return first if return_index else q[first]


class KernelMapRef:
  """Per-offset pair lists of one (in_key, out_key, kernel) triple.

  nbr[k, j]  = in-row that feeds out-row j through weight slice k, or -1.
  pairs[k]   = (in_rows, out_rows) compacted, ascending in out_row.
  """

  def __init__(self, nbr, n_in):
    self.nbr = nbr
    self.n_in = int(n_in)
    self.n_out = int(nbr.shape[1])
    self.K = int(nbr.shape[0])
    self.pairs = []
    for k in range(self.K):
      out_rows = np.nonzero(nbr[k] >= 0)[0].astype(np.int32)
      self.pairs.append((nbr[k, out_rows].astype(np.int32), out_rows))
    self.offs = np.concatenate([[0], np.cumsum([len(p[0]) for p in self.pairs])]).astype(np.int64)

  def swapped(self):
    """In/out exchanged with the same weight-slice index (transposed conv, A5)."""
    m = KernelMapRef.__new__(KernelMapRef)
    m.K, m.n_in, m.n_out = self.K, self.n_out, self.n_in
    m.pairs = [(o, i) for (i, o) in self.pairs]
    m.offs = self.offs
    m.nbr = None
    return m


class CoordsManagerRef:
  """ME CoordsManager restated (Appendix A1, A3-A5).  One per SparseTensor
  (pc/lib/ddp_trainer.py:392-398 builds a fresh one for every forward)."""

  def __init__(self, coords):
    coords = np.ascontiguousarray(np.asarray(coords, dtype=np.int32))
    assert coords.ndim == 2 and coords.shape[1] == 4, "coords must be [N, 1+3], batch index first"
    keys = pack_keys(coords)
    if len(np.unique(keys)) != len(keys):
      raise ValueError("duplicate coordinates")
    self.coords = {0: coords}
    self.tensor_stride = {0: 1}
    self._by_stride = {1: 0}
    self._sorted = {}
    self._kmaps = {}

  def size(self, key):
    return len(self.coords[key])

  def _lookup(self, key, query_keys):
    if key not in self._sorted:
      k = pack_keys(self.coords[key])
      order = np.argsort(k, kind="stable")
      self._sorted[key] = (k[order], order.astype(np.int32))
    sk, order = self._sorted[key]
    pos = np.searchsorted(sk, query_keys)
    pos[pos >= len(sk)] = len(sk) - 1
    hit = sk[pos] == query_keys
    return np.where(hit, order[pos], -1).astype(np.int32)

  def stride(self, in_key, stride=2):
    """Strided coordinates (A4): unique floor(c / (s*ts)) * (s*ts), rows in
    first-occurrence order of the input rows (ME's row order is hash-iteration
    dependent and unobservable by the reference; we fix it deterministically)."""
    ts = self.tensor_stride[in_key] * stride
    if ts in self._by_stride:
      return self._by_stride[ts]
    c = self.coords[in_key]
    q = c.copy()
    q[:, 1:] = np.floor_divide(c[:, 1:], ts) * ts
    keys = pack_keys(q)
    _, first = np.unique(keys, return_index=True)
    first = np.sort(first)
    out_key = len(self.coords)
    self.coords[out_key] = np.ascontiguousarray(q[first])
    self.tensor_stride[out_key] = ts
    self._by_stride[ts] = out_key
    return out_key

  def key_at_stride(self, ts):
    return self._by_stride[ts]

  def kernel_map(self, in_key, out_key, kernel_size, region=HYPERCUBE):
    """in/out pairs per weight slice (A3, A4): pair (i, j, k) iff
    c_out[j] + o_k * ts_in == c_in[i]."""
    ck = (in_key, out_key, kernel_size, region)
    if ck in self._kmaps:
      return self._kmaps[ck]
    offs = region_offsets(kernel_size, region)
    ts = self.tensor_stride[in_key]
    cout = self.coords[out_key]
    nbr = np.empty((len(offs), len(cout)), dtype=np.int32)
    for k, o in enumerate(offs):
      q = cout.copy()
      q[:, 1:] += o.astype(np.int32) * ts
      nbr[k] = self._lookup(in_key, pack_keys(q))
    km = KernelMapRef(nbr, self.size(in_key))
    self._kmaps[ck] = km
    return km


def sparse_conv(feats, weight, kmap, bias=None):
  """out[j] = sum_k in[i] @ W[k] over the map's pairs (A3/A4/A5; A11 for the
  backward, which autograd derives from these same pairs)."""
  out = feats.new_zeros((kmap.n_out, weight.shape[-1]))
  for k, (i_rows, o_rows) in enumerate(kmap.pairs):
    if len(i_rows) == 0:
      continue
    it = torch.from_numpy(i_rows.astype(np.int64))
    ot = torch.from_numpy(o_rows.astype(np.int64))
    out = out.index_add(0, ot, feats.index_select(0, it) @ weight[k])
  if bias is not None:
    out = out + bias
  return out


class SparseTensorRef:
  """(features, coords_key, coords_manager) triple, as ME.SparseTensor
  (pc/lib/ddp_trainer.py:392; pc/model/res16unet.py:262-266)."""

  def __init__(self, feats, coords=None, coords_key=None, coords_manager=None):
    if coords_manager is None:
      coords_manager = CoordsManagerRef(coords.numpy() if torch.is_tensor(coords) else coords)
      coords_key = 0
    self.F = feats
    self.coords_key = coords_key
    self.coords_man = coords_manager

  @property
  def C(self):
    return self.coords_man.coords[self.coords_key]

  def __add__(self, other):  # `out += residual`, pc/model/modules/resnet_block.py:57
    assert other.coords_key == self.coords_key and other.coords_man is self.coords_man
    return SparseTensorRef(self.F + other.F, coords_key=self.coords_key, coords_manager=self.coords_man)

  __iadd__ = __add__

  @property
  def tensor_stride(self):
    return self.coords_man.tensor_stride[self.coords_key]
