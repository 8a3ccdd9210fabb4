"""Generates tests/golden/golden_trace.npz: an UNSYNCHRONISED 20-iteration loss trace of the PointInfoNCE training step
-- and, with `hardest`, tests/golden/golden_trace_hardest.npz: the same for HardestContrastiveLossTrainer
(pc/lib/ddp_trainer.py:186-238,278-326; the candidate / positive draws of :198-206 injected per step).

Run from the repo root (CPU only, ~10 minutes each):
    python tests/golden/make_golden_trace.py
    python tests/golden/make_golden_trace.py hardest

What runs: the oracle's restatement of the reference iteration (pc/lib/ddp_trainer.py:380-440 -- two forwards of
Res16UNet14, pair selection with injected draws, PointInfoNCE, backward, SGD(lr 0.1, momentum 0.8, wd 1e-4); pinned
bit-identically to the reference's own source by tests/test_reference_trainer_source.py), 20 consecutive iterations on
ONE fixed synthetic pair, nothing re-seeded from anywhere in between, twice: in float32 (the reference's arithmetic)
and in float64 (the truth both are compared with).

Why both.  The iteration at lr 0.1 is a sensitive map: the fp32 oracle's own trace leaves the fp64 trace by 1e-4 after
three steps and by more than 1e-3 after five (ReLU kinks, BatchNorm over a few hundred rows) -- so "the device's trace
equals the oracle's to 1e-3 at every step" is not a property ANY fp32 implementation has, the reference's included.
What a correct fp32 implementation does have: its distance to the fp64 trace grows like the fp32 oracle's does.
tests/test_gpu_trace.py holds the device to that envelope, step by step, and to the same overall descent.

The fixture carries the batch, the per-step draws (as seeds), both traces and a checksum of the initial weights
(torch.manual_seed(0) on the CPU generator; the GPU box runs the same torch build).
"""
import copy
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

STEPS, NPOS, T, LR, MODEL, BN_MOMENTUM = 20, 256, 0.4, 0.1, "Res16UNet14", 0.05


def make_batch():
  from pointcontrast_amd.lib import synthetic
  from pointcontrast_amd.lib.ddp_data_loaders import default_collate_pair_fn
  rng = np.random.RandomState(0)
  return default_collate_pair_fn([synthetic.make_pair_item(rng, 0.025, crop=0.6)])


def draws_of(step, nq):
  """The host-RNG draws of iteration `step` (what the reference takes from torch / numpy global generators)."""
  d = dict(uniform=torch.rand(nq, generator=torch.Generator().manual_seed(1000 + step)))
  if nq > NPOS:
    d["sampled_inds"] = np.random.RandomState(1000 + step).choice(nq, NPOS, replace=False)
  return d


HN_POS, HN_SAMPLES, POS_THRESH, NEG_THRESH = 1024, 256, 0.1, 1.4  # pc/config/defaults.yaml (per batch of 1 pair)


def hardest_draws_of(step, N0, N1, P):
  """The three np.random.choice calls of pc/lib/ddp_trainer.py:198-206 for iteration `step`, from a seeded generator."""
  rng = np.random.RandomState(2000 + step)
  sel0 = rng.choice(N0, min(N0, HN_SAMPLES), replace=False)
  sel1 = rng.choice(N1, min(N1, HN_SAMPLES), replace=False)
  pos_sel = rng.choice(P, HN_POS, replace=False) if P > HN_POS else None
  return dict(sel0=sel0, sel1=sel1, pos_sel=pos_sel)


def initial_model():
  from oracle import model_ref as mr
  torch.manual_seed(0)
  m = mr.MODELS[MODEL](3, 32, bn_momentum=BN_MOMENTUM)
  m.train()
  return m


def weight_checksum(model):
  return float(sum(p.detach().double().abs().sum() for p in model.parameters()))


def main_hardest():
  from oracle import loss_ref as lr, sparse_ref as sr
  torch.set_num_threads(min(16, os.cpu_count() or 1))
  batch = make_batch()
  pp = batch["correspondences"].numpy()
  N0, N1 = batch["sinput0_C"].shape[0], batch["sinput1_C"].shape[0]
  m32 = initial_model()
  m64 = copy.deepcopy(m32).double()
  chk = weight_checksum(m32)
  o32, o64 = lr.make_sgd(m32.parameters(), LR), lr.make_sgd(m64.parameters(), LR)
  trace = {torch.float32: [], torch.float64: []}
  for step in range(STEPS):
    d = hardest_draws_of(step, N0, N1, len(pp))
    for m, o, dt in ((m32, o32, torch.float32), (m64, o64, torch.float64)):
      o.zero_grad()
      F0 = m(sr.SparseTensorRef(batch["sinput0_F"].to(dt), coords=batch["sinput0_C"].numpy())).F
      F1 = m(sr.SparseTensorRef(batch["sinput1_F"].to(dt), coords=batch["sinput1_C"].numpy())).F
      pos, neg, _ = lr.hardest_contrastive_loss(F0, F1, pp, d["sel0"], d["sel1"], d["pos_sel"], POS_THRESH, NEG_THRESH)
      loss = pos + neg
      loss.backward()
      o.step()
      trace[dt].append(float(loss.detach()))
    print("step %2d  fp32 %.6f  fp64 %.6f  rel %.2e" % (step, trace[torch.float32][-1], trace[torch.float64][-1],
                                                       abs(trace[torch.float32][-1] - trace[torch.float64][-1]) / abs(trace[torch.float64][-1])),
          flush=True)
  out = os.path.join(ROOT, "tests", "golden", "golden_trace_hardest.npz")
  np.savez_compressed(out, loss32=np.array(trace[torch.float32]), loss64=np.array(trace[torch.float64]),
                      weight_checksum=np.array(chk), steps=np.array(STEPS), lr=np.array(LR),
                      **{k: batch[k].numpy() for k in ("sinput0_C", "sinput0_F", "sinput1_C", "sinput1_F", "correspondences")})
  print("wrote", out)


def main():
  from oracle import loss_ref as lr, sparse_ref as sr
  torch.set_num_threads(min(16, os.cpu_count() or 1))
  batch = make_batch()
  pp = batch["correspondences"].numpy()
  nq = len(np.unique(pp[:, 0]))
  m32 = initial_model()
  m64 = copy.deepcopy(m32).double()
  chk = weight_checksum(m32)
  o32, o64 = lr.make_sgd(m32.parameters(), LR), lr.make_sgd(m64.parameters(), LR)
  trace = {torch.float32: [], torch.float64: []}
  for step in range(STEPS):
    d = draws_of(step, nq)
    qi, ki = lr.nce_select_pairs(pp, d["uniform"], d.get("sampled_inds"))
    for m, o, dt in ((m32, o32, torch.float32), (m64, o64, torch.float64)):
      o.zero_grad()
      F0 = m(sr.SparseTensorRef(batch["sinput0_F"].to(dt), coords=batch["sinput0_C"].numpy())).F
      F1 = m(sr.SparseTensorRef(batch["sinput1_F"].to(dt), coords=batch["sinput1_C"].numpy())).F
      loss = lr.nce_loss(F0, F1, qi, ki, T)
      loss.backward()
      o.step()
      trace[dt].append(float(loss.detach()))
    print("step %2d  fp32 %.6f  fp64 %.6f  rel %.2e" % (step, trace[torch.float32][-1], trace[torch.float64][-1],
                                                       abs(trace[torch.float32][-1] - trace[torch.float64][-1]) / abs(trace[torch.float64][-1])),
          flush=True)
  out = os.path.join(ROOT, "tests", "golden", "golden_trace.npz")
  np.savez_compressed(out, loss32=np.array(trace[torch.float32]), loss64=np.array(trace[torch.float64]),
                      weight_checksum=np.array(chk), steps=np.array(STEPS), npos=np.array(NPOS), T=np.array(T), lr=np.array(LR),
                      **{k: batch[k].numpy() for k in ("sinput0_C", "sinput0_F", "sinput1_C", "sinput1_F", "correspondences")})
  print("wrote", out)


if __name__ == "__main__":
  main_hardest() if "hardest" in sys.argv[1:] else main()
