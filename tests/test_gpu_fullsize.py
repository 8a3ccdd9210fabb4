"""GPU parity at BASELINE.json's FULL sizes (pytest -m gpu): the configurations the bench lines are quoted on, through
the same code path the bench runs (trainer + native executor, the pair as one two-segment pass), against the oracle.

  configs[1]  2.5 cm, B = 4, Res16UNet34C, PointInfoNCE: parameter GRADIENTS of every tensor (mask-imposed, see
              test_gpu_parity._network_case) -- features / loss at this size are test_full_config_forward_and_loss_*;
  configs[2]  the same batch through HardestContrastiveLossTrainer (4096 positives, 1024 hard-negative candidates per
              side), tie-aware, loss at 1e-4 (pc/lib/ddp_trainer.py:278-326);
  configs[4]  1 cm voxels (one pair, ~250 k rows in the joint pass): features of both clouds (max-norm and per row)
              and the PointInfoNCE loss at 1e-4 (pc/lib/ddp_trainer.py:380-440).
The oracle's share of these tests is tens of seconds of host time each on the GPU box's cores.
"""
import numpy as np
import pytest
import torch

from test_gpu_parity import DEV, ME, _assert_mined_valid, _network_case, assert_rows_close  # noqa: F401

pytestmark = pytest.mark.gpu


def _trainer(cls_name, batch, batch_size, extra=()):
  from oracle import model_ref as mr
  from pointcontrast_amd.lib import ddp_trainer
  from pointcontrast_amd.lib.config import get_config
  from pointcontrast_amd.lib.ddp_data_loaders import FixedBatchLoader
  # host_threads: the trainer otherwise caps torch's intra-op pool at 8 threads, which is what the ORACLE below runs on
  cfg = get_config(["net.model=Res16UNet34C", "misc.nceT=0.4", "misc.npos=4096", "opt.lr=0.1", "misc.engine=native",
                    "misc.host_threads=4096"] + list(extra))
  loader = FixedBatchLoader([batch], batch_size=batch_size)
  torch.manual_seed(21)
  trainer = getattr(ddp_trainer, cls_name)(cfg, loader)
  ref = mr.MODELS["Res16UNet34C"](3, 32, bn_momentum=cfg.opt.bn_momentum)
  ref.load_state_dict({k: v.cpu() for k, v in trainer.model.state_dict().items()})
  ref.train()
  return cfg, trainer, ref, loader


def _torch_batch(b):
  return {k: torch.from_numpy(np.ascontiguousarray(v)) if isinstance(v, np.ndarray) else v for k, v in b.items()}


def test_full_config_gradients_match_oracle(ME):
  """configs[1] at full size: every parameter-gradient tensor of Res16UNet34C on the B = 4, ~87k-voxel batch within
  10x the fp32 oracle's own error against the fp64 oracle (floor 5e-5), ReLU masks of the fp64 oracle imposed on all
  three runs -- the rule of test_network_features_loss_and_grads, at the size the bench is quoted on."""
  report = _network_case(ME, "Res16UNet34C", None, 4, 0, npos=4096, voxel_size=0.025)
  msg = "; ".join("%s dev=%.2e ref32=%.2e" % (n_, d_, r_) for d_, r_, n_, g_ in report[:6])
  print("full-size worst gradient tensors:", msg)
  level1 = ("conv0p1s1", "bn0", "convtr7p2s2", "bntr7", "block8", "final")
  bad = [(n_, d_, r_) for d_, r_, n_, _ in report if d_ > max(10 * r_, 5e-5)]
  assert not [x for x in bad if x[0].startswith(level1)], "level-1 gradient tensors off: %s" % (bad[:6],)
  assert not bad, "gradient tensors off: %s | worst: %s" % (bad[:6], msg)


def test_full_config_hardest_gradients_match_oracle(ME):
  """configs[2] at full size, GRADIENTS (round 5 compared the losses only; gradients at <= 24 k rows): every parameter-
  gradient tensor of Res16UNet34C under the HardestContrastive loss -- 4096 positives, 1024 hard-negative candidates per
  cloud on the B = 4, ~87k-voxel batch -- by the fp64-mask rule of test_full_config_gradients_match_oracle; the negatives
  the device mines are verified as arg-mins and shared with the oracle (pc/lib/ddp_trainer.py:186-238,278-326)."""
  report = _network_case(ME, "Res16UNet34C", None, 4, 0, npos=4096, voxel_size=0.025, loss="hardest", n_hard=1024)
  msg = "; ".join("%s dev=%.2e ref32=%.2e" % (n_, d_, r_) for d_, r_, n_, g_ in report[:6])
  print("full-size hardest-contrastive worst gradient tensors:", msg)
  bad = [(n_, d_, r_) for d_, r_, n_, _ in report if d_ > max(10 * r_, 5e-5)]
  assert not bad, "gradient tensors off: %s | worst: %s" % (bad[:6], msg)


def test_1cm_config_gradients_match_oracle(ME):
  """configs[4] shape, BACKWARD (round 5: forward + loss only): one 1 cm pair -- > 250 k rows over the two clouds, ~15
  neighbours per voxel at level 1 -- through forward, PointInfoNCE and backward; every parameter-gradient tensor by the
  fp64-mask rule, the level-1 tensors (whose kernels see the 250 k rows) named first."""
  report = _network_case(ME, "Res16UNet34C", None, 1, 3, npos=4096, voxel_size=0.01)
  msg = "; ".join("%s dev=%.2e ref32=%.2e" % (n_, d_, r_) for d_, r_, n_, g_ in report[:6])
  print("1 cm worst gradient tensors:", msg)
  level1 = ("conv0p1s1", "bn0", "convtr7p2s2", "bntr7", "block8", "final")
  l1 = [(n_, d_, r_) for d_, r_, n_, _ in report if n_.startswith(level1)]
  print("1 cm level-1 tensors:", "; ".join("%s dev=%.2e ref32=%.2e" % x for x in l1[:8]))
  assert len(l1) >= 3
  bad = [(n_, d_, r_) for d_, r_, n_, _ in report if d_ > max(10 * r_, 5e-5)]
  assert not [x for x in bad if x[0].startswith(level1)], "level-1 gradient tensors off at 1 cm: %s" % (bad[:6],)
  assert not bad, "gradient tensors off at 1 cm: %s | worst: %s" % (bad[:6], msg)


def test_full_config_hardest_trainer_matches_oracle(ME):
  """configs[2]: the full B = 4 Res16UNet34C batch through HardestContrastiveLossTrainer on the native executor.  The
  device's mined negatives must be arg-mins of the ORACLE's features up to 1e-4 (the two feature sets differ by
  ~1e-5, so near-ties of that size are legitimate), are handed to the oracle, and pos / neg / total loss agree to 1e-4."""
  from oracle import loss_ref as lr, sparse_ref as sr
  from pointcontrast_amd.lib import synthetic
  from pointcontrast_amd.lib.timer import AverageMeter, Timer
  b = _torch_batch(synthetic.make_batch(seed=0, batch_size=4, voxel_size=0.025))
  cfg, trainer, ref, loader = _trainer("HardestContrastiveLossTrainer", b, 4)
  assert trainer.config.trainer.num_pos_per_batch * 4 == 4096 and trainer.config.trainer.num_hn_samples_per_batch * 4 == 1024
  pp = b["correspondences"].numpy()
  N0, N1 = b["sinput0_C"].shape[0], b["sinput1_C"].shape[0]
  assert N0 > 80000 and len(pp) > 4096
  r = np.random.RandomState(5)
  draws = dict(sel0=r.choice(N0, 1024, replace=False), sel1=r.choice(N1, 1024, replace=False),
               pos_sel=r.choice(len(pp), 4096, replace=False))
  res = trainer._train_iter(iter(loader), [AverageMeter(), Timer(), Timer()], draws=draws)
  with torch.no_grad():
    F0 = ref(sr.SparseTensorRef(b["sinput0_F"], coords=b["sinput0_C"].numpy())).F
    F1 = ref(sr.SparseTensorRef(b["sinput1_F"], coords=b["sinput1_C"].numpy())).F
    mined = {k: v.cpu().numpy() for k, v in trainer._last_mined.items()}
    _assert_mined_valid(F0, F1, pp, draws, mined, tol=1e-4)
    pos, neg, aux = lr.hardest_contrastive_loss(F0, F1, pp, draws["sel0"], draws["sel1"], draws["pos_sel"],
                                                forced=(mined["D01ind"], mined["D10ind"]))
  assert (mined["mask0"].astype(bool) == aux["mask0"]).all() and (mined["mask1"].astype(bool) == aux["mask1"]).all()
  got = {k: float(v) for k, v in res.items()}
  print("hardest, full size: device", got, "oracle pos %.6f neg %.6f" % (float(pos), float(neg)))
  assert abs(got["pos_loss"] - float(pos)) <= 1e-4 * abs(float(pos)) + 1e-7
  assert abs(got["neg_loss"] - float(neg)) <= 1e-4 * abs(float(neg))
  assert abs(got["loss"] - float(pos + neg)) <= 1e-4 * abs(float(pos + neg))


def test_1cm_config_features_and_loss_match_oracle(ME):
  """configs[4] shape: 1 cm voxels, one pair (~250 k rows in the joint two-segment pass, ~15 neighbours per voxel at
  level 1), Res16UNet34C, npos 4096, T 0.4, prepared and forwarded exactly as PointNCELossTrainer._train_iter does:
  features of both clouds -- max-norm and worst row -- and the PointInfoNCE loss at 1e-4."""
  from oracle import loss_ref as lr, sparse_ref as sr
  from pointcontrast_amd import functional as PF
  from pointcontrast_amd.lib import synthetic
  b = _torch_batch(synthetic.make_batch(seed=3, batch_size=1, voxel_size=0.01))
  N0, N1 = b["sinput0_C"].shape[0], b["sinput1_C"].shape[0]
  assert N0 + N1 > 150000, (N0, N1)
  cfg, trainer, ref, loader = _trainer("PointNCELossTrainer", b, 1)
  trainer.model.train()
  pp = b["correspondences"].numpy()
  nq = len(np.unique(pp[:, 0]))
  draws = dict(uniform=torch.rand(nq, generator=torch.Generator().manual_seed(7)),
               sampled_inds=np.random.RandomState(7).choice(nq, 4096, replace=False))
  prep = trainer._prepare(b, draws)
  F0, F1 = trainer._forward_pair(prep)
  torch.cuda.current_stream().wait_event(prep["sel_event"])  # the selection runs on the planning stream
  q = PF.GatherRowsFunction.apply(F0, prep["q_idx"])
  k = PF.GatherRowsFunction.apply(F1, prep["k_idx"])
  ld = float(PF.NCELossFunction.apply(q, k, 0.4))
  with torch.no_grad():
    R0 = ref(sr.SparseTensorRef(b["sinput0_F"], coords=b["sinput0_C"].numpy())).F
    R1 = ref(sr.SparseTensorRef(b["sinput1_F"], coords=b["sinput1_C"].numpy())).F
    qi, ki = lr.nce_select_pairs(pp, draws["uniform"], draws["sampled_inds"])
    lref = float(lr.nce_loss(R0, R1, qi, ki, 0.4))
  assert torch.equal(prep["q_idx"].cpu(), qi) and torch.equal(prep["k_idx"].cpu(), ki)
  assert_rows_close(F0, R0, 1e-4, "1 cm features cloud 0 (%d rows)" % N0)
  assert_rows_close(F1, R1, 1e-4, "1 cm features cloud 1 (%d rows)" % N1)
  print("1 cm: N0 %d N1 %d loss device %.6f oracle %.6f" % (N0, N1, ld, lref))
  assert abs(ld - lref) <= 1e-4 * abs(lref), (ld, lref)
  trainer.engine._held[0] = None


def test_full_config_step_is_bit_reproducible():
  """configs[1] at full size, twice: two trainers built from the same seed take two PointInfoNCE iterations on the same
  batch with the same draws -- weights, SGD momentum and losses must be IDENTICAL bit for bit.  Every accumulation of
  the step has a fixed order (no float atomics: partial tiles / slabs / row-block partials summed in index order, the
  weight gradients on their own stream included), so concurrency between the compute, plan and weight-gradient streams
  may change the timing of a step but not one bit of its result."""
  from pointcontrast_amd.lib import synthetic
  from pointcontrast_amd.lib.ddp_data_loaders import FixedBatchLoader
  from pointcontrast_amd.lib.timer import AverageMeter, Timer
  batch = _torch_batch(synthetic.make_batch(seed=0, batch_size=4, voxel_size=0.025))
  pp = batch["correspondences"].numpy()
  nq = len(np.unique(pp[:, 0]))
  runs = []
  for _ in range(2):
    _, trainer, _, loader = _trainer("PointNCELossTrainer", batch, 4)
    losses = []
    for step in range(2):
      draws = dict(uniform=torch.rand(nq, generator=torch.Generator().manual_seed(step)),
                   sampled_inds=np.random.RandomState(step).choice(nq, 4096, replace=False))
      losses.append(float(trainer._train_iter(iter(FixedBatchLoader([batch], 4)), [AverageMeter(), Timer(), Timer()], draws=draws)["loss"]))
    torch.cuda.synchronize()
    runs.append((losses, trainer.flat.w.clone(), trainer.flat.v.clone()))
  assert runs[0][0] == runs[1][0], (runs[0][0], runs[1][0])
  assert torch.equal(runs[0][1], runs[1][1]), "weights differ between two identical runs"
  assert torch.equal(runs[0][2], runs[1][2]), "SGD momentum differs between two identical runs"
