"""Oracle restatement of the two contrastive losses and the optimiser step.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Host-RNG draws of the reference
(np.random.choice, Uniform.sample) are *inputs* here so the device path and the
oracle see the same index sets.
"""
import numpy as np
import torch
import torch.nn.functional as F


def nce_select_pairs(pos_pairs, uniform, sampled_inds=None):
  """pc/lib/ddp_trainer.py:403-417.  One key per unique query voxel:
  k_sel = pairs[:,1][floor(u * count) + cumsum_excl(count)]; optional npos
  sub-sample by ``sampled_inds``.  pos_pairs int [P,2] sorted by column 0."""
  pp = torch.as_tensor(pos_pairs).long()
  q_unique, count = pp[:, 0].unique(return_counts=True)
  off = torch.floor(torch.as_tensor(uniform, dtype=torch.float32) * count).long()
  cums = torch.cat([torch.zeros(1, dtype=torch.long), torch.cumsum(count, 0)[:-1]])
  k_sel = pp[:, 1][off + cums]
  if sampled_inds is not None:
    si = torch.as_tensor(sampled_inds).long()
    q_unique, k_sel = q_unique[si], k_sel[si]
  return q_unique, k_sel


def nce_loss(F0, F1, q_idx, k_idx, T, sampled_inds=None):
  """pc/lib/ddp_trainer.py:409-426 + pc/lib/criterion.py:15-19:
  CE(q @ k^T / T, arange).
  sampled_inds: the reference gathers in TWO stages (q = F0[q_unique]; q = q[sampled_inds], :409-415); pass the
  unsampled selection plus sampled_inds to reproduce that spelling (tests/test_reference_trainer_source.py runs it
  beside the reference's own file).  The loss is the same either way; where several queries picked the same key the
  gradient w.r.t. F1 is a scatter-add whose last bit depends on the order of the adds (torch's CPU kernel uses parallel
  atomics unless torch.use_deterministic_algorithms(True))."""
  q = F0[q_idx]
  k = F1[k_idx]
  if sampled_inds is not None:
    q = q[sampled_inds]
    k = k[sampled_inds]
  logits = torch.mm(q, k.t()) / T
  labels = torch.arange(q.shape[0])
  return F.cross_entropy(logits, labels)


def hash_pairs(a, b, M):
  """pc/lib/ddp_trainer.py:39-51 with D=2: a + b*M in int64."""
  return np.asarray(a, dtype=np.int64) + np.asarray(b, dtype=np.int64) * np.int64(M)


def hardest_contrastive_loss(F0, F1, positive_pairs, sel0, sel1, pos_sel, pos_thresh=0.1,
                             neg_thresh=1.4, forced=None):
  """pc/lib/ddp_trainer.py:186-238.  sel0/sel1: hard-negative candidate rows
  (np.random.choice at :199-200); pos_sel: sampled positive-pair indices (:203)
  or None when P <= num_pos.  Returns (pos_loss, neg_loss, aux) where aux holds
  the mined indices and masks for integer parity checks.
  forced = (D01ind, D10ind): use these hard negatives instead of torch's arg-min (tie-aware comparisons: at an fp32
  tie two correct implementations may mine different rows; the caller verifies that the forced rows ARE minima)."""
  pp = np.asarray(positive_pairs, dtype=np.int64)
  N0, N1 = len(F0), len(F1)
  hash_seed = max(N0, N1)
  sample = pp if pos_sel is None else pp[np.asarray(pos_sel)]
  pos_ind0 = torch.from_numpy(sample[:, 0].copy())
  pos_ind1 = torch.from_numpy(sample[:, 1].copy())
  subF0, subF1 = F0[torch.from_numpy(np.asarray(sel0, np.int64))], F1[torch.from_numpy(np.asarray(sel1, np.int64))]
  posF0, posF1 = F0[pos_ind0], F1[pos_ind1]

  def pdist(A, B):  # :182-184
    D2 = torch.sum((A.unsqueeze(1) - B.unsqueeze(0)).pow(2), 2)
    return torch.sqrt(D2 + 1e-7)

  if forced is None:
    D01min, D01ind = pdist(posF0, subF1).min(1)
    D10min, D10ind = pdist(posF1, subF0).min(1)
  else:
    D01ind, D10ind = (torch.from_numpy(np.asarray(f).astype(np.int64)) for f in forced)
    D01min = torch.sqrt((posF0 - subF1[D01ind]).pow(2).sum(1) + 1e-7)
    D10min = torch.sqrt((posF1 - subF0[D10ind]).pow(2).sum(1) + 1e-7)
  pos_keys = hash_pairs(pp[:, 0], pp[:, 1], hash_seed)
  n01 = np.asarray(sel1)[D01ind.numpy()]
  n10 = np.asarray(sel0)[D10ind.numpy()]
  neg_keys0 = hash_pairs(pos_ind0.numpy(), n01, hash_seed)
  neg_keys1 = hash_pairs(n10, pos_ind1.numpy(), hash_seed)
  mask0 = torch.from_numpy(np.logical_not(np.isin(neg_keys0, pos_keys)))
  mask1 = torch.from_numpy(np.logical_not(np.isin(neg_keys1, pos_keys)))
  pos_loss = F.relu((posF0 - posF1).pow(2).sum(1) - pos_thresh)
  neg_loss0 = F.relu(neg_thresh - D01min[mask0]).pow(2)
  neg_loss1 = F.relu(neg_thresh - D10min[mask1]).pow(2)
  aux = dict(D01ind=D01ind.numpy(), D10ind=D10ind.numpy(), mask0=mask0.numpy(), mask1=mask1.numpy())
  return pos_loss.mean(), (neg_loss0.mean() + neg_loss1.mean()) / 2, aux


def make_sgd(params, lr, momentum=0.8, weight_decay=1e-4):
  """pc/lib/ddp_trainer.py:107-111 (opt.momentum, NOT sgd_momentum; defaults.yaml:43-53)."""
  return torch.optim.SGD(params, lr=lr, momentum=momentum, weight_decay=weight_decay)
