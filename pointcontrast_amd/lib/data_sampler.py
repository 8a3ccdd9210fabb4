"""Samplers that never run dry (pc/lib/data_sampler.py:13-73): the training loop calls ``next()`` until
``opt.max_iter`` on ONE iterator, so the index stream is infinite.

Observable behaviour kept from the reference (same seed on all ranks, pc/ddp_train.py:28, hence the same
permutations everywhere): ``InfSampler`` hands out a fresh permutation from its END towards its start;
``DistributedInfSampler`` rank r takes positions r, r + world, r + 2 world, ... of the shared permutation and draws a new
permutation once the walk has covered it.  Written as generators over a permutation source."""
import torch
from torch.utils.data.sampler import Sampler


class InfSampler(Sampler):

  def __init__(self, data_source, shuffle=False):
    self.data_source, self.shuffle = data_source, shuffle
    self._first = self._permutation()  # drawn at construction, as the reference does (position in the RNG stream)
    self._stream = self._walk()

  def _permutation(self):
    n = len(self.data_source)
    return (torch.randperm(n) if self.shuffle else torch.arange(n)).tolist()

  def _walk(self):
    perm = self._first
    while True:
      yield from reversed(perm)
      perm = self._permutation()  # lazily, by the call that finds the previous permutation used up

  def __iter__(self):
    return self

  def __next__(self):
    return next(self._stream)

  def __len__(self):
    return len(self.data_source)


class DistributedInfSampler(InfSampler):

  def __init__(self, data_source, num_replicas=None, rank=None, shuffle=True):
    import torch.distributed as dist
    if (num_replicas is None or rank is None) and not (dist.is_available() and dist.is_initialized()):
      raise RuntimeError("DistributedInfSampler needs an initialised process group (or explicit num_replicas / rank)")
    self.num_replicas = dist.get_world_size() if num_replicas is None else num_replicas
    self.rank = dist.get_rank() if rank is None else rank
    self.num_samples = -(-len(data_source) // self.num_replicas)
    super().__init__(data_source, shuffle)

  def _walk(self):
    perm = self._first
    while True:
      step = 0
      while True:  # position of this rank in round `step`; the walk ends with the round that reaches the end
        value = perm[(step * self.num_replicas + self.rank) % len(perm)]
        step += 1
        last = step * self.num_replicas >= len(perm)
        if last:  # the next permutation is drawn in the SAME call that hands out the last index (as the reference:
          perm = self._permutation()  # the position of this draw in the global torch RNG stream is observable)
        yield value
        if last:
          break

  def __len__(self):
    return self.num_samples
