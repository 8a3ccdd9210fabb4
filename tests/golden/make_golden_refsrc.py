"""Generates tests/golden/golden_refsrc.npz by executing the REFERENCE'S OWN model source.

Run from the repo root in the build container (needs /root/reference):
    python tests/golden/make_golden_refsrc.py

What runs: /root/reference/pretrain/pointcontrast/model/{res16unet,resnet}.py + modules/{common,resnet_block}.py,
imported unmodified (tests/refsrc.py), with `MinkowskiEngine` resolved to
  (a) oracle/me_shim.py  (CPU oracle ops)      -> features F0/F1 of Res16UNet34C on a seeded 2-pair batch, the
      PointInfoNCE loss (pc/lib/ddp_trainer.py:400-426 restated in oracle/loss_ref.py), BatchNorm running
      statistics after the two forwards, state-dict names / shapes;
  (b) pointcontrast_amd.minkowski (symbolic)   -> the network program the native executor runs
      (pointcontrast_amd/engine.py::canonical_program), as JSON.
MinkowskiEngine itself is not available, so these vectors pin the WIRING to the reference source;
the per-op arithmetic stays pinned by tests/test_oracle_dense.py.  The fixture travels to the GPU box, where
/root/reference does not exist.
"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import refsrc  # noqa: E402

MODEL = "Res16UNet34C"
STAT_BNS = ("bn0", "block1.0.norm1", "block5.0.downsample.1", "bntr7", "block8.1.norm2")
NPOS, T = 512, 0.4


def make_inputs():
  from pointcontrast_amd.lib import synthetic
  from pointcontrast_amd.lib.ddp_data_loaders import default_collate_pair_fn
  from pointcontrast_amd.lib.ddp_trainer import PointNCELossTrainer
  rng = np.random.RandomState(20260925)
  batch = default_collate_pair_fn([synthetic.make_pair_item(rng, 0.025, crop=0.4) for _ in range(2)])
  pp = batch["correspondences"]
  nq = len(np.unique(pp[:, 0].numpy()))
  draws = dict(uniform=torch.rand(nq, generator=torch.Generator().manual_seed(1)),
               sampled_inds=np.random.RandomState(1).choice(nq, NPOS, replace=False))
  qi, ki = PointNCELossTrainer.select_pairs(pp, NPOS, draws)
  out = {k: batch[k].numpy() for k in ("sinput0_C", "sinput0_F", "sinput1_C", "sinput1_F", "correspondences")}
  out["q_idx"], out["k_idx"] = qi.numpy(), ki.numpy()
  return out


def run_reference_source(inp):
  """(a): the reference's model classes over the oracle-backed MinkowskiEngine stand-in."""
  from oracle import loss_ref as lr, me_shim
  from pointcontrast_amd.lib.config import get_config
  pkg = refsrc.import_reference_models(me_shim.install)
  cfg = get_config([])
  model = pkg.load_model(MODEL)(3, cfg.net.model_n_out, cfg, D=3)
  refsrc.fill_deterministic(model)
  model.train()
  F = []
  for s in "01":
    st = me_shim.SparseTensor(torch.from_numpy(inp["sinput%s_F" % s]), coords=inp["sinput%s_C" % s])
    F.append(model(st).F)
  loss = lr.nce_loss(F[0], F[1], torch.from_numpy(inp["q_idx"]), torch.from_numpy(inp["k_idx"]), T)
  sd = model.state_dict()
  out = {"F0": F[0].detach().numpy(), "F1": F[1].detach().numpy(), "loss": np.float64(loss.item())}
  for b in STAT_BNS:
    out["rm:" + b] = sd[b + ".bn.running_mean"].numpy().copy()
    out["rv:" + b] = sd[b + ".bn.running_var"].numpy().copy()
  out["state_dict_layout"] = json.dumps([[k, list(v.shape)] for k, v in sd.items()])
  return out, model


def reference_program():
  """(b): the same source lowered by this package's tracer."""
  import pointcontrast_amd.minkowski as ME
  from pointcontrast_amd.engine import canonical_program, lower_model
  from pointcontrast_amd.lib.config import get_config
  from pointcontrast_amd.lib.distributed import FlatParameters
  pkg = refsrc.import_reference_models(ME.install)
  cfg = get_config([])
  model = pkg.load_model(MODEL)(3, cfg.net.model_n_out, cfg, D=3)
  return canonical_program(lower_model(model, FlatParameters(model.parameters())))


def generate():
  inp = make_inputs()
  out, _ = run_reference_source(inp)
  out.update(inp)
  out["program"] = json.dumps(reference_program())
  return out


def main():
  out = generate()
  path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden_refsrc.npz")
  np.savez_compressed(path, **out)
  print(path, os.path.getsize(path), {k: getattr(v, "shape", None) for k, v in out.items()})


if __name__ == "__main__":
  main()
