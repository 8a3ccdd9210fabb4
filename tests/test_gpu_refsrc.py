"""GPU side of the reference-source wiring parity (see tests/test_reference_source.py).

tests/golden/golden_refsrc.npz was produced by executing the reference's unmodified pc/model/*.py over the CPU
oracle ops (tests/golden/make_golden_refsrc.py); /root/reference does not exist on the GPU box, so these tests work
from that fixture: the device model -- per-layer path and native executor -- must reproduce the reference
source's features, loss and BatchNorm running statistics, and the program the executor runs must be the one the
reference source lowers to."""
import json
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import refsrc  # noqa: E402
from test_gpu_parity import DEV, assert_close  # noqa: E402

pytestmark = pytest.mark.gpu
G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "golden_refsrc.npz"))


def _model():
  from pointcontrast_amd.lib.config import get_config
  from pointcontrast_amd.model import load_model
  m = load_model("Res16UNet34C")(3, 32, get_config([]), D=3)
  refsrc.fill_deterministic(m)
  return m.to(DEV).train()


def _inputs(ME):
  return [ME.SparseTensor(torch.from_numpy(G["sinput%s_F" % s]), coords=torch.from_numpy(G["sinput%s_C" % s])).to(DEV) for s in "01"]


@pytest.mark.parametrize("engine", ["autograd", "native"])
def test_device_model_reproduces_reference_source_vectors(engine):
  import pointcontrast_amd.minkowski as ME
  from pointcontrast_amd import functional as PF
  from pointcontrast_amd.engine import NativeEngine, canonical_program, lower_model
  from pointcontrast_amd.lib.distributed import FlatParameters
  m = _model()
  sts = _inputs(ME)
  if engine == "native":
    flat = FlatParameters(m.parameters())
    prog = canonical_program(lower_model(m, flat))
    assert json.loads(json.dumps(prog)) == json.loads(str(G["program"])), "not the program the reference source lowers to"
    eng = NativeEngine(m, flat)
    F = [eng.forward(i, sts[i]) for i in range(2)]
  else:
    F = [m(st).F for st in sts]
  for i in range(2):
    assert_close(F[i], torch.from_numpy(G["F%d" % i]), 1e-4, "%s features cloud %d vs reference source" % (engine, i))
  q = PF.GatherRowsFunction.apply(F[0], torch.from_numpy(G["q_idx"]).to(DEV))
  k = PF.GatherRowsFunction.apply(F[1], torch.from_numpy(G["k_idx"]).to(DEV))
  loss = float(PF.NCELossFunction.apply(q, k, 0.4))
  assert abs(loss - float(G["loss"])) <= 1e-4 * abs(float(G["loss"])), (loss, float(G["loss"]))
  sd = m.state_dict()
  for key in G.files:
    if key[:3] in ("rm:", "rv:"):
      name = key[3:] + (".bn.running_mean" if key[:2] == "rm" else ".bn.running_var")
      assert_close(sd[name], torch.from_numpy(G[key]), 1e-4, key)


def test_unfused_reference_spelling_runs_on_the_device():
  """The reference writes bn -> relu, `out += residual` -> relu and F / torch.norm(F) as separate ops
  (pc/model/modules/resnet_block.py:44-60, pc/model/res16unet.py:262-266).  Through the per-layer path those are
  separate libpcmi kernels (pcmi_bn_fwd_train, pcmi_add, pcmi_relu_fwd/bwd); results and gradients must equal the
  fused spelling this package's own blocks use."""
  import pointcontrast_amd.minkowski as ME
  from pointcontrast_amd.model.modules.common import ConvType, conv, get_norm, NormType
  torch.manual_seed(0)
  ct = ConvType.SPATIAL_HYPERCUBE_TEMPORAL_HYPERCROSS
  c1, n1 = conv(32, 64, 3, conv_type=ct, D=3).to(DEV), get_norm(NormType.BATCH_NORM, 64, 3).to(DEV)
  c2, n2 = conv(64, 64, 3, conv_type=ct, D=3).to(DEV), get_norm(NormType.BATCH_NORM, 64, 3).to(DEV)
  cd, nd = conv(32, 64, 1, D=3).to(DEV), get_norm(NormType.BATCH_NORM, 64, 3).to(DEV)
  relu = ME.MinkowskiReLU(inplace=True)
  C = torch.from_numpy(G["sinput1_C"])
  x0 = torch.randn(len(C), 32)
  params = [p for mod in (c1, n1, c2, n2, cd, nd) for p in mod.parameters()]

  def run(fused):
    for mod in (n1, n2, nd):
      mod.bn.reset_running_stats()
    x = ME.SparseTensor(x0.clone().requires_grad_(True), coords=C).to(DEV)
    xf = x.F
    xf.retain_grad()
    if fused:
      out = n1(c1(x), relu=True)
      out = n2(c2(out), residual=nd(cd(x)), relu=True)
      out = ME.l2_normalize(out)
    else:  # the reference's spelling, line by line
      residual = x
      out = c1(x)
      out = n1(out)
      out = relu(out)
      out = c2(out)
      out = n2(out)
      residual = nd(cd(x))
      out += residual
      out = relu(out)
      out = ME.SparseTensor(out.F / torch.norm(out.F, p=2, dim=1, keepdim=True), coords_key=out.coords_key,
                            coords_manager=out.coords_man)
    for p in params:
      p.grad = None
    (out.F * torch.linspace(-1, 1, 64, device=DEV)).sum().backward()
    return out.F.detach().clone(), [p.grad.clone() for p in params], n2.bn.running_var.clone()

  fa, ga, ra = run(True)
  fb, gb, rb = run(False)
  assert_close(fb, fa, 1e-6, "unfused vs fused features")
  assert_close(rb, ra, 1e-6, "running var")
  for i, (a, b) in enumerate(zip(ga, gb)):
    assert_close(b, a, 1e-4, "parameter gradient %d" % i)
