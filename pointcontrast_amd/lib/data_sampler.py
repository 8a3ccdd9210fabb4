"""Infinite samplers (pc/lib/data_sampler.py:13-73): every rank walks the SAME permutation
(same seed on all ranks, pc/ddp_train.py:28) with a rank-strided index."""
import torch
from torch.utils.data.sampler import Sampler


class InfSampler(Sampler):

  def __init__(self, data_source, shuffle=False):
    self.data_source, self.shuffle = data_source, shuffle
    self.reset_permutation()

  def reset_permutation(self):
    n = len(self.data_source)
    self._perm = (torch.randperm(n) if self.shuffle else torch.arange(n)).tolist()

  def __iter__(self):
    return self

  def __next__(self):
    if not self._perm:
      self.reset_permutation()
    return self._perm.pop()

  def __len__(self):
    return len(self.data_source)


class DistributedInfSampler(InfSampler):

  def __init__(self, data_source, num_replicas=None, rank=None, shuffle=True):
    import torch.distributed as dist
    self.num_replicas = dist.get_world_size() if num_replicas is None else num_replicas
    self.rank = dist.get_rank() if rank is None else rank
    self.it = 0
    self.num_samples = -(-len(data_source) // self.num_replicas)
    super().__init__(data_source, shuffle)

  def __next__(self):
    idx = self.it * self.num_replicas + self.rank
    value = self._perm[idx % len(self._perm)]
    self.it += 1
    if self.it * self.num_replicas >= len(self._perm):
      self.reset_permutation()
      self.it = 0
    return value

  def __len__(self):
    return self.num_samples
