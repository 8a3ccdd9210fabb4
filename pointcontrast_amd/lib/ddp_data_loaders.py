"""Pair data loader with the reference's batch contract (pc/lib/ddp_data_loaders.py:52-112,
:272-309): a dict of CPU tensors
  sinput{0,1}_C int32 [N,4] (batch id FIRST), sinput{0,1}_F fp32 [N,3], correspondences int32
  [P,2] sorted by column 0 with per-item row offsets, pcd{0,1}, T_gt, len_batch.
ScanNet itself is not available (licence + no network), so the dataset is the seeded
synthetic generator of lib/synthetic.py pushed through the same per-item chain; a real
ScanNetMatchPairDataset only has to yield the same item tuples."""
import numpy as np
import torch
import torch.utils.data

from . import synthetic
from .data_sampler import DistributedInfSampler


class SyntheticScanNetPairDataset(torch.utils.data.Dataset):

  def __init__(self, phase="train", config=None, num_pairs=None, crop=None):
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


ALL_DATASETS = [SyntheticScanNetPairDataset]
dataset_str_mapping = {d.__name__: d for d in ALL_DATASETS}


def make_data_loader(config, batch_size, num_threads=0):
  Dataset = dataset_str_mapping[config.data.dataset]
  dset = Dataset(phase="train", config=config)
  batch_size = batch_size // config.misc.num_gpus  # per-GPU batch, pc/lib/ddp_data_loaders.py:292
  sampler = DistributedInfSampler(dset) if config.misc.num_gpus > 1 else None
  return torch.utils.data.DataLoader(dset, batch_size=batch_size, shuffle=False if sampler else True,
                                     num_workers=num_threads, collate_fn=default_collate_pair_fn, pin_memory=False,
                                     sampler=sampler, drop_last=True)


class FixedBatchLoader:
  """Replays pre-generated batches forever (benchmarks / tests: inputs staged before timing)."""

  def __init__(self, batches, batch_size):
    self.batches, self.batch_size = list(batches), batch_size

  def __len__(self):
    return len(self.batches)

  def __iter__(self):
    i = 0
    while True:
      yield self.batches[i % len(self.batches)]
      i += 1
