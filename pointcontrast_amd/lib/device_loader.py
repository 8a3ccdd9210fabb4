"""Device versions of the two loader steps that precede the hot path in every dataset item:
first-occurrence voxelisation (ME.utils.sparse_quantize, pc/lib/ddp_data_loaders.py:228-229) and the radius
correspondence search (get_matching_indices, :36-49 -- in the reference a Python loop of ~20k open3d KD-tree queries per
item).  libpcmi kernels (csrc/loader.hip), bit-exact against oracle/loader_ref.py.

Two forms: numpy in / numpy out (sparse_quantize_index, get_matching_indices -- what the dataset classes switch to with
data.device_geometry=True), and a device-resident form (pair_geometry_device) that uploads the two frames once, keeps
the voxelised points, voxel coordinates and correspondences ON the device and synchronises twice per item (the two
kernels report counts): a pipeline stage whose outputs feed ME.SparseTensor / the device-side pair selection without
returning to the host.  scripts/loader_bench.py times both against the host path (items/s)."""
import ctypes as C

import numpy as np
import torch

from .._lib import lib, check
from ..runtime import ptr, cur_stream, ws_args


def _dev(device):
  return torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())


def sparse_quantize_index(xyz, voxel_size, device=None, return_coords=False):
  """Ascending indices of the first point of every occupied voxel of floor(xyz / voxel_size)."""
  dev = _dev(device)
  x = torch.as_tensor(np.ascontiguousarray(xyz, dtype=np.float64)).to(dev)
  n = x.shape[0]
  idx = torch.empty(max(n, 1), dtype=torch.int32, device=dev)
  coords = torch.empty((max(n, 1), 3), dtype=torch.int32, device=dev) if return_coords else None
  nu = C.c_int64()
  with torch.cuda.device(dev):
    ws, wsb = ws_args(lib.pcmi_voxelize_workspace_bytes(n), dev)
    check(lib.pcmi_voxelize(ptr(x), n, float(voxel_size), ptr(idx), ptr(coords), C.byref(nu), ws, wsb, cur_stream(dev)))
  torch.cuda.current_stream(dev).synchronize()
  sel = idx[:nu.value].cpu().numpy().astype(np.int64)
  return (sel, coords[:nu.value].cpu().numpy()) if return_coords else sel


def get_matching_indices(xyz0, xyz1, trans, search_radius, device=None):
  """All (i, j) with |trans(xyz0[i]) - xyz1[j]| <= search_radius, int64 [P, 2] sorted by (i, j)."""
  dev = _dev(device)
  a = torch.as_tensor(np.ascontiguousarray(xyz0, dtype=np.float64)).to(dev)
  b = torch.as_tensor(np.ascontiguousarray(xyz1, dtype=np.float64)).to(dev)
  n0, n1 = a.shape[0], b.shape[0]
  T = (C.c_double * 12)(*np.asarray(trans, dtype=np.float64)[:3, :4].reshape(-1).tolist())
  cnt = C.c_int64()
  cap = max(16 * n0, 1024)
  with torch.cuda.device(dev):
    ws, wsb = ws_args(lib.pcmi_match_radius_workspace_bytes(n0, n1), dev)
    while True:
      pairs = torch.empty((cap, 2), dtype=torch.int32, device=dev)
      rc = lib.pcmi_match_radius(ptr(a), n0, T, ptr(b), n1, float(search_radius), ptr(pairs), cap, C.byref(cnt), ws, wsb,
                                 cur_stream(dev))
      if rc == -7 and cnt.value > cap:  # PCMI_ERR_WORKSPACE: more pairs than room -> retry with the reported count
        cap = cnt.value
        continue
      check(rc)
      break
  torch.cuda.current_stream(dev).synchronize()
  return pairs[:cnt.value].cpu().numpy().astype(np.int64)


def pair_geometry_device(xyz0, xyz1, trans, voxel_size, search_radius, device=None):
  """One dataset item's geometry, device-resident: first-occurrence voxelisation of both frames, the surviving points,
  their voxel coordinates floor(xyz / voxel_size) and all correspondences within search_radius of trans(xyz0[i]).
  Returns dict(xyz0, xyz1 [n, 3] fp64, coords0, coords1 [n, 3] int32, matches [P, 2] int32 sorted by (i, j)) of device
  tensors.  Same arithmetic as the numpy forms (bit-exact against oracle/loader_ref.py)."""
  dev = _dev(device)
  out = {}
  with torch.cuda.device(dev):
    st = cur_stream(dev)
    pts = []
    for tag, xyz in (("0", xyz0), ("1", xyz1)):
      x = torch.as_tensor(np.ascontiguousarray(xyz, dtype=np.float64)).to(dev, non_blocking=True)
      n = x.shape[0]
      idx = torch.empty(max(n, 1), dtype=torch.int32, device=dev)
      coords = torch.empty((max(n, 1), 3), dtype=torch.int32, device=dev)
      nu = C.c_int64()
      ws, wsb = ws_args(lib.pcmi_voxelize_workspace_bytes(n), dev)
      check(lib.pcmi_voxelize(ptr(x), n, float(voxel_size), ptr(idx), ptr(coords), C.byref(nu), ws, wsb, st))  # syncs (count)
      sel = idx[:nu.value].long()
      out["xyz" + tag] = x.index_select(0, sel)
      out["coords" + tag] = coords[:nu.value]
      pts.append(out["xyz" + tag])
    a, b = pts
    n0, n1 = a.shape[0], b.shape[0]
    T = (C.c_double * 12)(*np.asarray(trans, dtype=np.float64)[:3, :4].reshape(-1).tolist())
    cnt = C.c_int64()
    cap = max(16 * n0, 1024)
    ws, wsb = ws_args(lib.pcmi_match_radius_workspace_bytes(n0, n1), dev)
    while True:
      pairs = torch.empty((cap, 2), dtype=torch.int32, device=dev)
      rc = lib.pcmi_match_radius(ptr(a), n0, T, ptr(b), n1, float(search_radius), ptr(pairs), cap, C.byref(cnt), ws, wsb, st)
      if rc == -7 and cnt.value > cap:
        cap = cnt.value
        continue
      check(rc)
      break
    out["matches"] = pairs[:cnt.value]
  return out
