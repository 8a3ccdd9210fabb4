"""Pair data loader with the reference's batch contract (pc/lib/ddp_data_loaders.py:52-112,
:272-309): a dict of CPU tensors
  sinput{0,1}_C int32 [N,4] (batch id FIRST), sinput{0,1}_F fp32 [N,3], correspondences int32
  [P,2] sorted by column 0 with per-item row offsets, pcd{0,1}, T_gt, len_batch.
Two datasets yield the same 8-tuple items (xyz0, xyz1, coords0, coords1, feats0, feats1, matches, trans):
  * ScanNetMatchPairDataset -- the reference's (:119-270): a pair list (txt, two npz paths per line) under
    data.dataset_root_dir / data.scannet_match_dir, each npz holding a 'pcd' array; random scale / rotation,
    first-occurrence voxelisation, radius-1.5-voxel correspondences, Jitter;
  * SyntheticScanNetPairDataset -- ScanNet itself is not available here (licence + no network): the seeded generator
    of lib/synthetic.py pushed through the same per-item chain (default, used by bench.py and the tests)."""
import logging
import os
import random

import numpy as np
import torch
import torch.utils.data
from scipy.spatial import cKDTree

from . import synthetic
from .data_sampler import DistributedInfSampler, InfSampler


class FeatureJitter:
  """The loader's only feature transform (pc/lib/transforms.py:21-30, applied through Compose at
  pc/lib/ddp_data_loaders.py:281-288): with probability p the features get N(mu, sigma) noise; coordinates pass."""

  def __init__(self, mu=0.0, sigma=0.01, p=0.95):
    self.mu, self.sigma, self.p = mu, sigma, p

  def __call__(self, coords, feats):
    if random.random() < self.p:
      feats = feats + np.random.normal(self.mu, self.sigma, feats.shape)
    return coords, feats


def get_matching_indices(xyz0, xyz1, trans, search_radius, K=None):
  """All (i, j) with |trans(xyz0[i]) - xyz1[j]| <= radius, ordered by i (pc/lib/ddp_data_loaders.py:36-49; the
  reference walks an open3d KD-tree point by point -- here one batched cKDTree query)."""
  src = _apply_rigid(trans, xyz0)
  nb = cKDTree(xyz1).query_ball_point(src, search_radius * (1 + 1e-9))  # candidates; the exact test follows
  if K is not None:
    nb = [x[:K] for x in nb]
  cnt = np.fromiter((len(x) for x in nb), np.int64, len(nb))
  ii = np.repeat(np.arange(len(nb)), cnt)
  jj = np.concatenate([np.sort(np.asarray(x, np.int64)) for x in nb]) if cnt.sum() else np.zeros(0, np.int64)
  # membership decided by the same separately-rounded float64 expression the device kernel evaluates
  ex, ey, ez = src[ii, 0] - xyz1[jj, 0], src[ii, 1] - xyz1[jj, 1], src[ii, 2] - xyz1[jj, 2]
  keep = ((ex * ex + ey * ey) + ez * ez) <= np.float64(search_radius) * np.float64(search_radius)
  return np.stack([ii[keep], jj[keep]], 1)


def _apply_rigid(trans, xyz):
  """((R0 x + R1 y) + R2 z) + t, each operation rounded on its own (what csrc/loader.hip computes)."""
  T, p = np.asarray(trans, dtype=np.float64), np.asarray(xyz, dtype=np.float64)
  out = np.empty_like(p)
  for r in range(3):
    out[:, r] = ((T[r, 0] * p[:, 0] + T[r, 1] * p[:, 1]) + T[r, 2] * p[:, 2]) + T[r, 3]
  return out


class ScanNetMatchPairDataset(torch.utils.data.Dataset):
  """pc/lib/ddp_data_loaders.py:119-270."""

  def __init__(self, phase, transform=None, random_rotation=True, random_scale=True, manual_seed=False, config=None):
    if phase != "train":
      raise NotImplementedError("only the train split is defined (as in the reference)")
    self.phase, self.transform = phase, transform
    self.voxel_size = config.data.voxel_size
    self.matching_search_voxel_size = config.data.voxel_size * config.trainer.positive_pair_search_voxel_size_multiplier
    self.random_scale, self.random_rotation = random_scale, random_rotation
    self.min_scale, self.max_scale = config.trainer.min_scale, config.trainer.max_scale
    self.rotation_range = config.trainer.rotation_range
    self.randg = np.random.RandomState()
    if manual_seed:
      self.reset_seed()
    # data.device_geometry: voxelisation + correspondence search as libpcmi kernels (lib/device_loader.py) instead of
    # numpy / cKDTree on the host; identical results (both bit-exact against oracle/loader_ref.py).  The item is then
    # produced by the process that owns the GPU: use misc.train_num_thread=0.
    self.device_geometry = bool(config.data.get("device_geometry", False))
    self.root = config.data.dataset_root_dir
    if not self.root or not config.data.scannet_match_dir:
      raise ValueError("ScanNetMatchPairDataset needs data.dataset_root_dir and data.scannet_match_dir (the pair list)")
    fname_txt = os.path.join(self.root, config.data.scannet_match_dir)
    logging.info("Loading the subset %s from %s", phase, fname_txt)
    with open(fname_txt) as f:
      self.files = [ln.split()[:2] for ln in f if len(ln.split()) >= 2]

  def reset_seed(self, seed=0):
    self.randg.seed(seed)

  def __len__(self):
    return len(self.files)

  def __getitem__(self, idx):
    from .. import minkowski as ME
    xyz0 = np.load(os.path.join(self.root, self.files[idx][0]))["pcd"]
    xyz1 = np.load(os.path.join(self.root, self.files[idx][1]))["pcd"]
    radius = self.matching_search_voxel_size
    if self.random_scale and random.random() < 0.95:
      scale = self.min_scale + (self.max_scale - self.min_scale) * random.random()
      radius *= scale
      xyz0, xyz1 = scale * xyz0, scale * xyz1
    if self.random_rotation:
      T0 = synthetic.sample_random_trans(xyz0, self.randg, self.rotation_range)
      T1 = synthetic.sample_random_trans(xyz1, self.randg, self.rotation_range)
      trans = T1 @ np.linalg.inv(T0)
      xyz0 = xyz0 @ T0[:3, :3].T + T0[:3, 3]
      xyz1 = xyz1 @ T1[:3, :3].T + T1[:3, 3]
    else:
      trans = np.identity(4)
    if self.device_geometry:
      from . import device_loader as dl
      xyz0 = xyz0[dl.sparse_quantize_index(xyz0, self.voxel_size)]
      xyz1 = xyz1[dl.sparse_quantize_index(xyz1, self.voxel_size)]
      matches = dl.get_matching_indices(xyz0, xyz1, trans, radius)
    else:
      xyz0 = xyz0[ME.utils.sparse_quantize(xyz0 / self.voxel_size, return_index=True)]
      xyz1 = xyz1[ME.utils.sparse_quantize(xyz1 / self.voxel_size, return_index=True)]
      matches = get_matching_indices(xyz0, xyz1, trans, radius)
    feats0, feats1 = np.ones((len(xyz0), 3)), np.ones((len(xyz1), 3))
    coords0, coords1 = np.floor(xyz0 / self.voxel_size), np.floor(xyz1 / self.voxel_size)
    if self.transform:
      coords0, feats0 = self.transform(coords0, feats0)
      coords1, feats1 = self.transform(coords1, feats1)
    return xyz0, xyz1, coords0, coords1, feats0, feats1, matches, trans


class SyntheticScanNetPairDataset(torch.utils.data.Dataset):

  def __init__(self, phase="train", config=None, num_pairs=None, crop=None, **_unused):
    self.voxel_size = config.data.voxel_size
    self.search_mult = config.trainer.positive_pair_search_voxel_size_multiplier
    self.num_pairs = num_pairs or config.data.get("num_pairs", 64)
    self.seed = config.misc.get("seed", 0)
    self.crop = crop if crop is not None else config.data.get("crop", None)

  def __len__(self):
    return self.num_pairs

  def __getitem__(self, idx):
    rng = np.random.RandomState(self.seed * 100003 + idx)
    return synthetic.make_pair_item(rng, self.voxel_size, self.search_mult, crop=self.crop)


def default_collate_pair_fn(list_data):
  d = synthetic.collate_pairs(list_data)
  out = {k: (torch.from_numpy(v) if isinstance(v, np.ndarray) else v) for k, v in d.items()}
  return out


ALL_DATASETS = [ScanNetMatchPairDataset, SyntheticScanNetPairDataset]
dataset_str_mapping = {d.__name__: d for d in ALL_DATASETS}


def make_data_loader(config, batch_size, num_threads=0):
  if config.data.dataset not in dataset_str_mapping:
    raise ValueError("Dataset %s does not exist in %s" % (config.data.dataset, ", ".join(dataset_str_mapping)))
  Dataset = dataset_str_mapping[config.data.dataset]
  dset = Dataset(phase="train", transform=FeatureJitter(), random_scale=config.trainer.use_random_scale,
                 random_rotation=config.trainer.use_random_rotation, config=config)
  batch_size = batch_size // config.misc.num_gpus  # per-GPU batch, pc/lib/ddp_data_loaders.py:292
  if getattr(dset, "device_geometry", False) and num_threads > 0:
    # __getitem__ then launches HIP kernels (lib/device_loader.py): a forked DataLoader worker cannot use the parent's
    # HIP context ("Cannot re-initialize CUDA in forked subprocess") -- the items are produced on the training process
    raise ValueError("data.device_geometry=True needs misc.train_num_thread=0 (got %d): the dataset runs libpcmi "
                     "kernels in __getitem__, which forked DataLoader workers cannot do" % num_threads)
  # The training loop calls iter() once and next() until opt.max_iter (pc/lib/ddp_trainer.py:128-140), so the loader
  # must never run dry: infinite samplers on ANY number of GPUs (the reference's single-GPU DataLoader is finite and
  # raises StopIteration after one epoch).
  sampler = DistributedInfSampler(dset) if config.misc.num_gpus > 1 else InfSampler(dset, shuffle=True)
  return torch.utils.data.DataLoader(dset, batch_size=batch_size, shuffle=False, num_workers=num_threads,
                                     collate_fn=default_collate_pair_fn, pin_memory=torch.cuda.is_available(), sampler=sampler,
                                     drop_last=True)


class FixedBatchLoader:
  """Replays pre-generated batches forever (benchmarks / tests: inputs staged before timing)."""

  def __init__(self, batches, batch_size, pin_memory=None):
    self.batches, self.batch_size = list(batches), batch_size
    # the staged batches in pinned host memory (what DataLoader(pin_memory=True) hands out): uploads are then truly
    # asynchronous copies without a pass through a staging buffer
    if pin_memory is None:
      pin_memory = torch.cuda.is_available()
    if pin_memory:
      self.batches = [{k: (v.pin_memory() if torch.is_tensor(v) and not v.is_cuda and not v.is_pinned() else v) for k, v in b.items()}
                      for b in self.batches]

  def __len__(self):
    return len(self.batches)

  def __iter__(self):
    i = 0
    while True:
      yield self.batches[i % len(self.batches)]
      i += 1
