"""Row N3 (first slice): the backbone reused downstream with out_channels = number of classes
(downstream/semseg/models/res16unet.py:202-260: 1x1 head of width num_labels, no feature normalisation) and the
semantic-segmentation loss (downstream/semseg/lib/train.py:64,124: CrossEntropyLoss(ignore_index)).  Needs
convolutions of ANY width (csrc/widths.hip stages zero-padded operands for the 32-channel matrix-core kernels)."""
import numpy as np
import pytest
import torch

from test_gpu_parity import DEV, assert_close, _conv_case

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ME():
  import pointcontrast_amd.minkowski as me
  return me


@pytest.mark.parametrize("size,kind,cin,cout", [("small", "1x1", 96, 20), ("small", "k3_hybrid", 48, 20), ("small", "down", 40, 72),
                                                ("small", "up", 72, 40), ("tiny", "k3_cube", 16, 16), ("mid", "1x1", 96, 13),
                                                ("mid", "k3_hybrid", 32, 20)])
def test_spconv_any_width(ME, size, kind, cin, cout):
  res = _conv_case(ME, size, kind, cin, cout, bias=(kind == "1x1"))
  for name, (got, ref) in res.items():
    assert_close(got, ref, 1e-4, "%s %s %d->%d %s" % (size, kind, cin, cout, name))


@pytest.mark.parametrize("n,c", [(1000, 20), (70000, 20), (333, 13), (5000, 41)])
def test_softmax_cross_entropy_with_ignore_label(n, c):
  from pointcontrast_amd import functional as PF
  torch.manual_seed(n)
  x = torch.randn(n, c) * 3
  lb = torch.randint(0, c, (n,))
  lb[torch.rand(n) < 0.2] = 255
  xr = x.clone().requires_grad_(True)
  lref = torch.nn.functional.cross_entropy(xr, lb, ignore_index=255)
  (lref * 1.3).backward()
  xd = x.to(DEV).requires_grad_(True)
  ld = PF.SoftmaxCrossEntropyFunction.apply(xd, lb.to(DEV), 255)
  (ld * 1.3).backward()
  assert abs(float(ld) - float(lref)) <= 1e-5 * abs(float(lref))
  assert_close(xd.grad, xr.grad, 1e-5, "dlogits")
  all_ignored = PF.SoftmaxCrossEntropyFunction.apply(x.to(DEV), torch.full((n,), 255), 255)
  assert float(all_ignored) == 0.0
  # a label that is neither a class nor the ignore label: torch raises, the kernel poisons the loss (never drops the row)
  bad = lb.clone()
  bad[n // 2] = c + 3
  with pytest.raises((IndexError, RuntimeError)):
    torch.nn.functional.cross_entropy(x, bad, ignore_index=255)
  assert np.isnan(float(PF.SoftmaxCrossEntropyFunction.apply(x.to(DEV), bad.to(DEV), 255)))


@pytest.mark.parametrize("engine", ["autograd", "native"])
def test_segmentation_head_forward_loss_and_head_gradients(ME, engine):
  """Res16UNet14 with a 20-class head and normalize_feature=False: logits, CE loss (ignore 255) and the gradients of
  the head against the oracle, through the per-layer path and through the native executor."""
  from oracle import model_ref as mr, sparse_ref as sr
  from pointcontrast_amd import functional as PF
  from pointcontrast_amd.engine import NativeEngine
  from pointcontrast_amd.lib import synthetic
  from pointcontrast_amd.lib.config import get_config
  from pointcontrast_amd.lib.distributed import FlatParameters
  from pointcontrast_amd.model import load_model
  cfg = get_config(["net.normalize_feature=False"])
  torch.manual_seed(0)
  ref = mr.MODELS["Res16UNet14"](3, 20, bn_momentum=cfg.opt.bn_momentum, normalize_feature=False)
  dev = load_model("Res16UNet14")(3, 20, cfg, D=3)
  dev.load_state_dict(ref.state_dict())
  dev = dev.to(DEV).train()
  ref.train()
  b = synthetic.make_batch(seed=3, batch_size=1, crop=0.6)
  C, F = b["sinput0_C"], torch.from_numpy(b["sinput0_F"])
  labels = torch.from_numpy(np.random.RandomState(0).randint(0, 20, len(C)))
  labels[::7] = 255
  yr = ref(sr.SparseTensorRef(F, coords=C)).F
  lref = torch.nn.functional.cross_entropy(yr, labels, ignore_index=255)
  lref.backward()
  st = ME.SparseTensor(F, coords=torch.from_numpy(C)).to(DEV)
  if engine == "native":
    flat = FlatParameters(dev.parameters())
    eng = NativeEngine(dev, flat)
    yd = eng.forward(0, st).requires_grad_(True)
  else:
    yd = dev(st).F
  assert yd.shape == (len(C), 20)
  assert_close(yd, yr, 1e-4, "segmentation logits (%s)" % engine)
  ld = PF.SoftmaxCrossEntropyFunction.apply(yd, labels.to(DEV), 255)
  assert abs(float(ld) - float(lref)) <= 1e-4 * abs(float(lref))
  ld.backward()
  if engine == "native":
    flat.zero_grad()
    eng.backward(0, yd.grad)
  assert_close(dev.final.kernel.grad, ref.final.kernel.grad, 1e-4, "head kernel gradient")
  assert_close(dev.final.bias.grad, ref.final.bias.grad, 1e-4, "head bias gradient")
  # upstream of the head, through ONE ReLU: an activation within fp32 round-off of zero may sit on the other side of the
  # kink on the device and then moves ITS channel's sum by one row's term (~1/rows); every other channel must agree
  gd, gr = dev.block8[-1].norm2.bn.bias.grad.cpu().double(), ref.block8[-1].norm2.bn.bias.grad.double()
  err = (gd - gr).abs() / gr.abs().max()
  assert int((err > 1e-4).sum()) <= 2 and float(err.max()) <= 1e-2, "upstream gradient: %d channels off, worst %.2e" % (
      int((err > 1e-4).sum()), float(err.max()))


def test_segmentation_trainer_steps_match_oracle(tmp_path):
  """pointcontrast_amd.downstream.semseg.SegmentationTrainer: pre-trained backbone loaded by name and shape (the
  32-wide contrastive head is skipped), then two fine-tuning iterations -- forward, CE(ignore 255), backward,
  SGD(momentum 0.9, dampening 0.1: the reference's defaults)
  + PolyLR -- against the oracle model + torch CrossEntropyLoss / SGD / the reference's PolyLR formula.  Every step
  starts from the device's state (see test_trainer_iteration_matches_oracle)."""
  from oracle import model_ref as mr, sparse_ref as sr
  from pointcontrast_amd.downstream import semseg as ss
  from pointcontrast_amd.lib import synthetic
  from pointcontrast_amd.lib.config import get_config
  from pointcontrast_amd.model import load_model
  torch.manual_seed(5)
  pre = load_model("Res16UNet14")(3, 32, get_config([]), D=3)  # "pre-trained" weights with the contrastive head
  torch.save({"state_dict": {"module." + k: v for k, v in pre.state_dict().items()}}, tmp_path / "pretrained.pth")
  tr = ss.SegmentationTrainer(20, model="Res16UNet14", lr=0.05, max_iter=50, pretrained=str(tmp_path / "pretrained.pth"))
  sd = tr.model.state_dict()
  assert torch.equal(sd["block3.0.conv1.kernel"].cpu(), pre.state_dict()["block3.0.conv1.kernel"])
  assert sd["final.kernel"].shape == (256, 20)
  ref = mr.MODELS["Res16UNet14"](3, 20, bn_momentum=0.02, normalize_feature=False)
  ref.train()
  # the reference's fine-tuning optimiser: downstream/semseg/lib/solvers.py:52-60 with config/default.yaml:16-19
  opt = torch.optim.SGD(ref.parameters(), lr=0.05, momentum=0.9, dampening=0.1, weight_decay=1e-4)
  b = synthetic.make_batch(seed=8, batch_size=2, crop=0.6)
  C, F = torch.from_numpy(b["sinput0_C"]), torch.from_numpy(b["sinput0_F"])
  target = torch.from_numpy(np.random.RandomState(1).randint(0, 20, len(C)))
  target[::5] = 255
  for step in range(2):
    ref.load_state_dict({k: v.detach().cpu().clone() for k, v in tr.model.state_dict().items()})
    if step > 0:
      dev_params = dict(tr.model.named_parameters())
      for name, p in ref.named_parameters():
        opt.state[p]["momentum_buffer"] = tr.optimizer.state[dev_params[name]]["momentum_buffer"].detach().cpu().clone()
    lr_now = tr.scheduler.get_last_lr()[0]
    assert abs(lr_now - 0.05 * (1 - step / 51) ** 0.9) < 1e-9
    for g in opt.param_groups:
      g["lr"] = lr_now
    res = tr.train_iter(C, F, target)
    opt.zero_grad()
    masks = [m.cpu() for m in tr.engine.relu_masks(0)]  # the device's ReLU patterns (see test_trainer_iteration_matches_oracle)
    with mr.relu_masks(apply=masks) as kinks:
      logits = ref(sr.SparseTensorRef(F, coords=C.numpy())).F
    assert kinks.flips <= 1e-4 * kinks.total
    loss = torch.nn.functional.cross_entropy(logits, target, ignore_index=255)
    loss.backward()
    opt.step()
    assert abs(float(res["loss"]) - float(loss)) <= 1e-4 * abs(float(loss)), (step, float(res["loss"]), float(loss))
    ref_score = ss.precision_at_one(logits.detach().max(1)[1], target)
    assert abs(res["score"] - ref_score) <= 0.5, (res["score"], ref_score)  # arg-max ties
  report = sorted(((float((dict(tr.model.named_parameters())[k].detach().cpu() - p.detach()).abs().max()) /
                    max(float(p.detach().abs().max()), 1e-6), k) for k, p in ref.named_parameters()), reverse=True)
  print("worst parameters after the fine-tuning step:", report[:4])
  assert report[0][0] <= 1e-4  # observed on MI355X with the shared ReLU patterns: 2.7e-6 (1.1e-3 without them)
  miou, ious, hist = tr.evaluate(C, F, target.numpy())
  assert hist.shape == (20, 20) and hist.sum() == int((target != 255).sum()) and 0.0 <= miou <= 100.0
