"""Throughput of the loader-side geometry of one dataset item (first-occurrence voxelisation of both
frames + radius correspondence search, pc/lib/ddp_data_loaders.py:36-49,228-241) in items/s:
  host      numpy sparse_quantize + scipy cKDTree ball queries (this package's host path; the reference loops ~20k open3d
            KD-tree queries per item in Python)
  device    csrc/loader.hip through the numpy-in / numpy-out wrappers (upload, 3 kernels' worth of launches, download)
  resident  the same kernels with the outputs left on the device (lib/device_loader.py::pair_geometry_device)
on synthetic ScanNet-shaped frame pairs (640x480 depth frames, ~300k points each).  Usage: python scripts/loader_bench.py [items]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from pointcontrast_amd.lib import synthetic, device_loader as dl
from pointcontrast_amd.lib.ddp_data_loaders import get_matching_indices
import pointcontrast_amd.minkowski as ME

n_items = int(sys.argv[1]) if len(sys.argv) > 1 else 8
voxel, radius = 0.025, 1.5 * 0.025
rng = np.random.RandomState(0)
items = []
for _ in range(n_items):
  a, b = synthetic.make_frame_pair(rng)
  items.append((a, b, np.eye(4)))
print("frames: %d items, %d / %d points in the first pair" % (n_items, len(items[0][0]), len(items[0][1])), flush=True)


def host(a, b, T):
  a = a[ME.utils.sparse_quantize(a / voxel, return_index=True)]
  b = b[ME.utils.sparse_quantize(b / voxel, return_index=True)]
  return len(get_matching_indices(a, b, T, radius))


def device(a, b, T):
  a = a[dl.sparse_quantize_index(a, voxel)]
  b = b[dl.sparse_quantize_index(b, voxel)]
  return len(dl.get_matching_indices(a, b, T, radius))


def resident(a, b, T):
  return int(dl.pair_geometry_device(a, b, T, voxel, radius)["matches"].shape[0])


res = {}
for name, fn in (("host", host), ("device", device), ("resident", resident)):
  counts = [fn(*items[0])]  # warm-up (page faults, allocator, first launches)
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  counts = [fn(*it) for it in items]
  torch.cuda.synchronize()
  dt = time.perf_counter() - t0
  res[name] = (n_items / dt, counts)
  print("%-9s %8.2f items/s  (%.1f ms per item, %d correspondences in item 0)" % (name, n_items / dt, dt / n_items * 1e3, counts[0]), flush=True)
assert res["host"][1] == res["device"][1] == res["resident"][1], "the three paths found different numbers of correspondences"
print("device / host = %.1fx, resident / host = %.1fx" % (res["device"][0] / res["host"][0], res["resident"][0] / res["host"][0]))
