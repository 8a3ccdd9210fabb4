"""The reference's OWN training-step source, executed on the CPU over the oracle ops.

pc/lib/ddp_trainer.py is imported unmodified (tests/refsrc.py) with `MinkowskiEngine` = oracle/me_shim.py; its classes
are instantiated without their GPU-asserting constructors and fed CPU tensors:
  * `_hash` and `HardestContrastiveLossTrainer.contrastive_hardest_negative_loss` (:39-51, :182-238)
  * `PointNCELossTrainer._train_iter` (:380-440): the reference's whole iteration -- two forwards of ITS Res16UNet34C,
    pair selection, logits / T, NCESoftmaxLoss, backward, SGD step -- with `.cuda()` neutralised by the test
and compared with the restatements the GPU parity tests use (oracle/loss_ref.py, oracle/model_ref.py): same host-RNG
draws (same seeds, same call order), bit-identical losses, gradients and updated weights.
Skipped where /root/reference is absent."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import refsrc  # noqa: E402

pytestmark = pytest.mark.skipif(not refsrc.reference_available(), reason="/root/reference is not present on this host")


@pytest.fixture(scope="module")
def ref_trainer():
  from oracle import me_shim
  return refsrc.import_reference_trainer(me_shim.install)


def test_hash_and_hardest_loss_equal_the_reference_source(ref_trainer):
  from oracle import loss_ref as lr
  rng = np.random.RandomState(0)
  a, b = rng.randint(0, 5000, 300), rng.randint(0, 7000, 300)
  assert (ref_trainer._hash([a, b], 7000) == lr.hash_pairs(a, b, 7000)).all()
  assert (ref_trainer._hash(np.stack([a, b], 1), 7000) == lr.hash_pairs(a, b, 7000)).all()
  torch.manual_seed(1)
  N0, N1 = 900, 800
  F0 = torch.nn.functional.normalize(torch.randn(N0, 32), dim=1)
  F1 = torch.nn.functional.normalize(F0[:N1] + 0.2 * torch.randn(N1, 32), dim=1)
  i = np.sort(rng.randint(0, N1, 2000))
  pp = np.unique(np.stack([i, np.clip(i + rng.randint(-1, 2, 2000), 0, N1 - 1)], 1), axis=0)
  tr = object.__new__(ref_trainer.HardestContrastiveLossTrainer)
  tr.pos_thresh, tr.neg_thresh = 0.1, 1.4
  F0a, F1a = F0.clone().requires_grad_(True), F1.clone().requires_grad_(True)
  np.random.seed(7)
  pos_a, neg_a = tr.contrastive_hardest_negative_loss(F0a, F1a, torch.from_numpy(pp), num_pos=256, num_hn_samples=128)
  (pos_a + neg_a).backward()
  # the same three draws, in the reference's order (:199-203)
  np.random.seed(7)
  sel0 = np.random.choice(N0, 128, replace=False)
  sel1 = np.random.choice(N1, 128, replace=False)
  pos_sel = np.random.choice(len(pp), 256, replace=False)
  F0b, F1b = F0.clone().requires_grad_(True), F1.clone().requires_grad_(True)
  pos_b, neg_b, _ = lr.hardest_contrastive_loss(F0b, F1b, pp, sel0, sel1, pos_sel)
  (pos_b + neg_b).backward()
  assert float(pos_a) == float(pos_b) and float(neg_a) == float(neg_b)
  assert torch.equal(F0a.grad, F0b.grad) and torch.equal(F1a.grad, F1b.grad)


@pytest.fixture(autouse=True)
def _serial_scatter_adds():
  """torch's CPU index_put_(accumulate=True) -- the backward of F1[k_sel] where several queries picked one key -- adds
  with parallel atomics unless deterministic algorithms are requested; bit-for-bit comparisons need the serial order."""
  was = torch.are_deterministic_algorithms_enabled()
  torch.use_deterministic_algorithms(True)
  yield
  torch.use_deterministic_algorithms(was)


def _cpu_only(monkeypatch):
  """The reference's file calls .cuda() on a label tensor and on its criterion and empties the CUDA cache: no-ops here."""
  from oracle import sparse_ref as sr
  monkeypatch.setattr(torch.Tensor, "cuda", lambda self, *a, **k: self)
  monkeypatch.setattr(torch.nn.Module, "cuda", lambda self, *a, **k: self)
  monkeypatch.setattr(torch.cuda, "empty_cache", lambda: None)
  monkeypatch.setattr(sr.SparseTensorRef, "to", lambda self, device: self, raising=False)


def _reference_trainer(ref_trainer, cls, cfg):
  Model = ref_trainer.load_model("Res16UNet34C")
  assert Model.__module__ == "model.res16unet"
  model = Model(3, 32, cfg, D=3)
  refsrc.fill_deterministic(model)
  model.train()
  tr = object.__new__(cls)  # the constructors assert a GPU and build data loaders; the iteration needs only these
  tr.config, tr.model, tr.cur_device, tr.batch_size = cfg, model, "cpu", 1
  tr.optimizer = torch.optim.SGD(model.parameters(), lr=cfg.opt.lr, momentum=cfg.opt.momentum, weight_decay=cfg.opt.weight_decay)
  return tr, model


def _batch():
  G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "golden_refsrc.npz"))
  batch = {k: torch.from_numpy(G[k]) for k in ("sinput0_C", "sinput0_F", "sinput1_C", "sinput1_F", "correspondences")}
  batch["pcd0"], batch["pcd1"] = batch["sinput0_F"], batch["sinput1_F"]  # only their lengths are read

  class Loader:  # the file calls data_loader_iter.next() (torch 1.5 spelling)
    def next(self):
      return batch

  return G, batch, Loader()


def test_hardest_iteration_equals_the_reference_source(ref_trainer, monkeypatch):
  from oracle import loss_ref as lr, model_ref as mr, sparse_ref as sr
  from pointcontrast_amd.lib.config import get_config
  cfg = get_config(["opt.lr=0.1", "trainer.num_pos_per_batch=512", "trainer.num_hn_samples_per_batch=256"])
  _cpu_only(monkeypatch)
  tr, model = _reference_trainer(ref_trainer, ref_trainer.HardestContrastiveLossTrainer, cfg)
  tr.pos_thresh, tr.neg_thresh = cfg.trainer.pos_thresh, cfg.trainer.neg_thresh
  G, batch, loader = _batch()
  T = ref_trainer.Timer
  np.random.seed(5)
  loss_a, pos_a, neg_a = tr._train_iter(loader, [ref_trainer.AverageMeter(), T(), T()])
  own = mr.MODELS["Res16UNet34C"](3, 32, bn_momentum=cfg.opt.bn_momentum)
  refsrc.fill_deterministic(own)
  own.train()
  opt = lr.make_sgd(own.parameters(), cfg.opt.lr, momentum=cfg.opt.momentum, weight_decay=cfg.opt.weight_decay)
  opt.zero_grad()
  F0 = own(sr.SparseTensorRef(batch["sinput0_F"], coords=G["sinput0_C"])).F
  F1 = own(sr.SparseTensorRef(batch["sinput1_F"], coords=G["sinput1_C"])).F
  pp = G["correspondences"]
  np.random.seed(5)
  sel0 = np.random.choice(len(F0), min(len(F0), 256), replace=False)
  sel1 = np.random.choice(len(F1), min(len(F1), 256), replace=False)
  pos_sel = np.random.choice(len(pp), 512, replace=False) if len(pp) > 512 else None
  pos_b, neg_b, _ = lr.hardest_contrastive_loss(F0, F1, pp, sel0, sel1, pos_sel, cfg.trainer.pos_thresh, cfg.trainer.neg_thresh)
  (pos_b + neg_b).backward()
  opt.step()
  assert (pos_a, neg_a) == (pos_b.item(), neg_b.item())
  for (k, a), (_, b) in zip(model.state_dict().items(), own.state_dict().items()):
    assert torch.equal(a, b), "after the SGD step: " + k


def test_nce_iteration_equals_the_reference_source(ref_trainer, monkeypatch):
  from oracle import loss_ref as lr, model_ref as mr, sparse_ref as sr
  from pointcontrast_amd.lib.config import get_config
  cfg = get_config(["misc.nceT=0.4", "misc.npos=512", "opt.lr=0.1"])
  _cpu_only(monkeypatch)
  tr, model = _reference_trainer(ref_trainer, ref_trainer.PointNCELossTrainer, cfg)
  tr.T, tr.npos = cfg.misc.nceT, cfg.misc.npos
  G, batch, loader = _batch()
  T = ref_trainer.Timer
  torch.manual_seed(11)
  np.random.seed(12)
  loss_ref_src = tr._train_iter(loader, [ref_trainer.AverageMeter(), T(), T()])
  # ---- the restatement, same draws ----
  own = mr.MODELS["Res16UNet34C"](3, 32, bn_momentum=cfg.opt.bn_momentum)
  refsrc.fill_deterministic(own)
  own.train()
  opt = lr.make_sgd(own.parameters(), cfg.opt.lr, momentum=cfg.opt.momentum, weight_decay=cfg.opt.weight_decay)
  pp = G["correspondences"]
  nq = len(np.unique(pp[:, 0]))
  torch.manual_seed(11)
  uniform = torch.distributions.Uniform(0, 1).sample([nq])
  np.random.seed(12)
  sampled = np.random.choice(nq, cfg.misc.npos, replace=False)
  opt.zero_grad()
  F0 = own(sr.SparseTensorRef(batch["sinput0_F"], coords=G["sinput0_C"])).F
  F1 = own(sr.SparseTensorRef(batch["sinput1_F"], coords=G["sinput1_C"])).F
  qi, ki = lr.nce_select_pairs(pp, uniform, None)
  loss = lr.nce_loss(F0, F1, qi, ki, cfg.misc.nceT, sampled_inds=sampled)  # the reference's two-stage gather
  loss.backward()
  opt.step()
  assert float(loss) == loss_ref_src, (float(loss), loss_ref_src)
  for (k, a), (_, b) in zip(model.state_dict().items(), own.state_dict().items()):
    assert torch.equal(a, b), "after the SGD step: " + k
