"""The oracle reproduces the committed golden vectors (guards against oracle drift) and its
float building blocks agree with torch's own implementations."""
import os

import numpy as np
import torch

from oracle import loss_ref as lr, model_ref as mr, sparse_ref as sr

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "golden_small.npz"))


def test_integer_tables_match_golden():
  cm = sr.CoordsManagerRef(G["coords"])
  key = 0
  for lvl in range(3):
    assert (cm.kernel_map(key, key, 3, sr.HYPERCUBE).nbr == G["nbr_cube_l%d" % lvl]).all()
    assert (cm.kernel_map(key, key, 3, sr.HYBRID).nbr == G["nbr_hybrid_l%d" % lvl]).all()
    ck = cm.stride(key, 2)
    assert (cm.coords[ck] == G["coords_l%d" % (lvl + 1)]).all()
    assert (cm.kernel_map(key, ck, 2).nbr == G["child_l%d" % lvl]).all()
    key = ck


def test_float_outputs_match_golden():
  cm = sr.CoordsManagerRef(G["coords"])
  x, W, W2 = torch.from_numpy(G["x"]), torch.from_numpy(G["W"]), torch.from_numpy(G["W2"])
  y = sr.sparse_conv(x, W, cm.kernel_map(0, 0, 3, sr.HYBRID))
  assert np.allclose(y.numpy(), G["y_hybrid"], rtol=1e-5, atol=1e-5)
  k1 = cm.stride(0, 2)
  y2 = sr.sparse_conv(x, W2, cm.kernel_map(0, k1, 2))
  assert np.allclose(y2.numpy(), G["y_down"], rtol=1e-5, atol=1e-5)
  q, k = torch.from_numpy(G["q"]), torch.from_numpy(G["k"])
  idx = torch.arange(300)
  assert abs(float(lr.nce_loss(q, k, idx, idx, 0.4)) - float(G["nce_T0.4"])) < 1e-5


def test_nce_loss_closed_form():
  torch.manual_seed(0)
  q, k = torch.randn(50, 8), torch.randn(50, 8)
  idx = torch.arange(50)
  logits = q @ k.t() / 0.3
  ref = (torch.logsumexp(logits, 1) - logits.diag()).mean()
  assert torch.allclose(lr.nce_loss(q, k, idx, idx, 0.3), ref, atol=1e-6)


def test_model_param_count_and_names():
  m = mr.Res16UNet34CRef(3, 32)
  assert sum(p.numel() for p in m.parameters()) == 37847808
  names = list(m.state_dict().keys())
  assert "conv0p1s1.kernel" in names and "block2.0.downsample.1.bn.running_var" in names and "final.bias" in names
  assert m.final.kernel.shape == (96, 32) and m.conv0p1s1.kernel.shape == (27, 3, 32)
  # momentum quirk: block BNs 0.1, top-level and downsample BNs opt.bn_momentum (0.05)
  assert m.bn0.bn.momentum == 0.05 and m.block1[0].norm1.bn.momentum == 0.1 and m.block2[0].downsample[1].bn.momentum == 0.05


def test_hardest_loss_runs_and_masks_true_positives():
  torch.manual_seed(1)
  F0 = torch.nn.functional.normalize(torch.randn(60, 8), dim=1)
  F1 = F0.clone()  # identical features: the hardest negative of i is j = i, a true positive -> masked out
  pp = np.stack([np.arange(60), np.arange(60)], 1)
  pos, neg, aux = lr.hardest_contrastive_loss(F0, F1, pp, np.arange(60), np.arange(60), None)
  assert float(pos) == 0.0 and aux["mask0"].sum() == 0 and np.isnan(float(neg))
