"""Wiring parity against the reference's OWN model source.

MinkowskiEngine cannot be installed here, but pc/model/*.py can be imported unmodified over a stand-in
`MinkowskiEngine` module (tests/refsrc.py).  Two stand-ins are used:
  * oracle/me_shim.py (CPU oracle ops): the reference source becomes the oracle's model; the travelling
    restatement oracle/model_ref.py and the committed fixture tests/golden/golden_refsrc.npz are checked
    against it bit for bit;
  * pointcontrast_amd.minkowski (symbolic lowering): the reference source must lower to exactly the network
    program this package's own model classes lower to -- the program the GPU parity
    tests and bench.py execute.
Tests that need /root/reference skip where it is absent (the GPU box); the fixture-based ones run everywhere.
"""
import json
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import refsrc  # noqa: E402

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "golden_refsrc.npz")
G = np.load(GOLD)
needs_reference = pytest.mark.skipif(not refsrc.reference_available(), reason="/root/reference is not present on this host")


def _cfg():
  from pointcontrast_amd.lib.config import get_config
  return get_config([])


def _oracle_forward(model, dtype=torch.float32):
  from oracle import sparse_ref as sr
  return [model(sr.SparseTensorRef(torch.from_numpy(G["sinput%s_F" % s]).to(dtype), coords=G["sinput%s_C" % s])).F for s in "01"]


@needs_reference
@pytest.mark.parametrize("name", ["Res16UNet34C", "Res16UNet34"])
def test_model_ref_equals_reference_source_over_the_oracle(name):
  """oracle/model_ref.py (the restatement that travels to the GPU box) against the reference's source run over
  the same oracle ops: same state-dict names and shapes, bit-identical features, loss and parameter gradients."""
  from oracle import loss_ref as lr, me_shim, model_ref as mr
  cfg = _cfg()
  pkg = refsrc.import_reference_models(me_shim.install)
  ref = pkg.load_model(name)(3, 32, cfg, D=3)
  assert type(ref).__module__ == "model.res16unet" and sys.modules.get("model") is None  # the reference's class
  assert type(ref).__mro__[2].forward.__code__.co_filename.startswith(refsrc.REF_PC)  # Res16UNetBase.forward
  own = mr.MODELS[name](3, 32, bn_momentum=cfg.opt.bn_momentum, normalize_feature=cfg.net.normalize_feature)
  assert [(k, tuple(v.shape)) for k, v in ref.state_dict().items()] == [(k, tuple(v.shape)) for k, v in own.state_dict().items()]
  refsrc.fill_deterministic(ref)
  refsrc.fill_deterministic(own)
  ref.train()
  own.train()
  qi, ki = torch.from_numpy(G["q_idx"]), torch.from_numpy(G["k_idx"])
  outs = []
  for m in (ref, own):
    F = _oracle_forward(m)
    loss = lr.nce_loss(F[0], F[1], qi, ki, 0.4)
    loss.backward()
    outs.append((F, loss))
  for a, b in zip(outs[0][0], outs[1][0]):
    assert torch.equal(a, b), "features differ: max |d| = %g" % float((a - b).abs().max())
  assert float(outs[0][1].detach()) == float(outs[1][1].detach())
  gp = dict(own.named_parameters())
  for k, p in ref.named_parameters():
    assert torch.equal(p.grad, gp[k].grad), k
  for (k, a), (_, b) in zip(ref.state_dict().items(), own.state_dict().items()):
    assert torch.equal(a, b), "state after two training forwards: " + k  # incl. the block-BN momentum quirk


@needs_reference
def test_golden_fixture_is_what_the_reference_source_produces():
  """tests/golden/golden_refsrc.npz is current: regenerating it from the reference source gives the same arrays."""
  sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
  import make_golden_refsrc as mk
  new = mk.generate()
  assert set(new) == set(G.files)
  for k in G.files:
    a, b = np.asarray(new[k]), G[k]
    if a.dtype.kind in "US":
      assert str(a) == str(b), k
    elif a.dtype.kind == "f":
      assert np.allclose(a, b, rtol=1e-6, atol=1e-7), k
    else:
      assert (a == b).all(), k


def test_model_ref_reproduces_the_reference_source_fixture():
  """Runs everywhere (also on the GPU box's CPU): the oracle restatement against the vectors the reference source
  produced -- features, loss, BatchNorm running statistics, state-dict layout."""
  from oracle import loss_ref as lr, model_ref as mr
  cfg = _cfg()
  own = mr.MODELS["Res16UNet34C"](3, 32, bn_momentum=cfg.opt.bn_momentum)
  assert [[k, list(v.shape)] for k, v in own.state_dict().items()] == json.loads(str(G["state_dict_layout"]))
  refsrc.fill_deterministic(own)
  own.train()
  F = _oracle_forward(own)
  for i in range(2):
    assert np.allclose(F[i].detach().numpy(), G["F%d" % i], rtol=1e-5, atol=1e-6)
  loss = lr.nce_loss(F[0], F[1], torch.from_numpy(G["q_idx"]), torch.from_numpy(G["k_idx"]), 0.4)
  assert abs(float(loss) - float(G["loss"])) <= 1e-6 * abs(float(G["loss"]))
  sd = own.state_dict()
  for k in G.files:
    if k[:3] in ("rm:", "rv:"):
      name = k[3:] + (".bn.running_mean" if k[:2] == "rm" else ".bn.running_var")
      assert np.allclose(sd[name].numpy(), G[k], rtol=1e-5, atol=1e-7), k


@needs_reference
@pytest.mark.parametrize("name", ["Res16UNet34C", "Res16UNet34"])
def test_reference_source_lowers_to_the_same_program(name, built_lib):
  """pc/model/*.py, unmodified, with `MinkowskiEngine` = pointcontrast_amd.minkowski.
  Its unfused spelling (bn -> relu, out += residual -> relu, F / torch.norm(F)) is folded by the tracer into the
  fused ops; the resulting program, parameter layout and seeded initial values equal those of this package's
  own model classes."""
  import pointcontrast_amd.minkowski as ME
  from pointcontrast_amd.engine import canonical_program, lower_model
  from pointcontrast_amd.lib.distributed import FlatParameters
  from pointcontrast_amd.model import load_model
  cfg = _cfg()
  pkg = refsrc.import_reference_models(ME.install)
  torch.manual_seed(0)
  ref = pkg.load_model(name)(3, 32, cfg, D=3)
  torch.manual_seed(0)
  own = load_model(name)(3, 32, cfg, D=3)
  assert list(ref.state_dict()) == list(own.state_dict())
  for (k, a), (_, b) in zip(ref.state_dict().items(), own.state_dict().items()):
    assert torch.equal(a, b), "seeded initial values differ: " + k
  p_ref = canonical_program(lower_model(ref, FlatParameters(ref.parameters())))
  p_own = canonical_program(lower_model(own, FlatParameters(own.parameters())))
  assert p_ref == p_own
  if name == "Res16UNet34C":
    assert json.loads(json.dumps(p_ref)) == json.loads(str(G["program"]))


def test_own_model_lowers_to_the_reference_source_program(built_lib):
  """Runs everywhere: this package's Res16UNet34C lowers to the program recorded from the reference source."""
  from pointcontrast_amd.engine import canonical_program, lower_model
  from pointcontrast_amd.lib.distributed import FlatParameters
  from pointcontrast_amd.model import load_model
  own = load_model("Res16UNet34C")(3, 32, _cfg(), D=3)
  assert [[k, list(v.shape)] for k, v in own.state_dict().items()] == json.loads(str(G["state_dict_layout"]))
  prog = canonical_program(lower_model(own, FlatParameters(own.parameters())))
  assert json.loads(json.dumps(prog)) == json.loads(str(G["program"]))


def test_tracer_refuses_what_it_cannot_fold(built_lib):
  """An add / ReLU that is not the tail of a BatchNorm, or arithmetic on .F other than the L2 normalisation, must
  fail loudly at lowering time (there is no eager fallback inside the native engine)."""
  import pointcontrast_amd.minkowski as ME
  from pointcontrast_amd.engine import _Tracer
  tr = _Tracer({})
  x = tr.new(32, 0)
  with pytest.raises(NotImplementedError):
    ME.MinkowskiReLU()(x)
  with pytest.raises(NotImplementedError):
    x + x
  with pytest.raises((NotImplementedError, TypeError)):
    x.F * 2
  with pytest.raises(NotImplementedError):
    ME.SparseTensor(x.F, coords_key=None, coords_manager=None)
