"""Semantic-segmentation fine-tuning step on the pre-training backbone.

What the reference's downstream/semseg does per iteration (downstream/semseg/lib/train.py:46-232), on libpcmi:
  model      Res16UNet34C with out_channels = number of classes and no feature normalisation
             (downstream/semseg/models/res16unet.py:202-260) -- the SAME backbone kernels; the head's odd width goes
             through csrc/widths.hip
  weights    pre-trained checkpoint loaded by name and shape (downstream/semseg/lib/utils.py:19-43), the head stays random
  loss       nn.CrossEntropyLoss(ignore_index=ignore_label) on model(x).F (train.py:64,124) -> pcmi_softmax_ce_fwd/bwd
  optimiser  SGD(lr, sgd_momentum, dampening, weight_decay) + PolyLR (lib/solvers.py:27-31,50-59,75-76) -> FlatSGD + PolyLR
  metrics    precision_at_one (lib/utils.py:117-128), fast_hist / per_class_iu -> mIoU (:131-138)
Datasets, augmentation, validation loop, tensorboard and checkpoint bookkeeping of the downstream trainer are outside
the hot path and not provided.
"""
import numpy as np
import torch
from torch.optim.lr_scheduler import LambdaLR

from .. import functional as PF
from .. import minkowski as ME
from ..engine import NativeEngine
from ..lib import checkpoint as ck
from ..lib.config import get_config
from ..lib.distributed import FlatParameters
from ..lib.solver import FlatSGD
from ..model import load_model


class PolyLR(LambdaLR):
  """DeepLab learning-rate policy lr * (1 - step / (max_iter + 1)) ** power (downstream/semseg/lib/solvers.py:12-31)."""

  def __init__(self, optimizer, max_iter, power=0.9, last_step=-1):
    super().__init__(optimizer, lambda s: (1 - s / (max_iter + 1)) ** power, last_step)

  @property
  def last_step(self):
    return self.last_epoch


def precision_at_one(pred, target, ignore_label=255):
  """Percentage of correctly labelled points among those whose label is not ignored (lib/utils.py:117-128)."""
  pred, target = pred.reshape(-1), target.reshape(-1)
  keep = target != ignore_label
  if int(keep.sum()) == 0:
    return float("nan")
  return float((pred[keep] == target[keep]).float().mean() * 100.0)


def fast_hist(pred, label, n):
  """n x n confusion matrix of the points with a valid label (lib/utils.py:131-133)."""
  pred, label = np.asarray(pred), np.asarray(label)
  k = (label >= 0) & (label < n)
  return np.bincount(n * label[k].astype(int) + pred[k], minlength=n ** 2).reshape(n, n)


def per_class_iu(hist):
  """Intersection over union per class; mIoU = nanmean (lib/utils.py:136-138)."""
  with np.errstate(divide="ignore", invalid="ignore"):
    return np.diag(hist) / (hist.sum(1) + hist.sum(0) - np.diag(hist))


class SegmentationTrainer:
  """One process per GPU; `train_iter(coords, feats, target)` = forward, cross-entropy, backward, SGD + PolyLR step."""

  def __init__(self, num_labels, in_channels=3, model="Res16UNet34C", lr=0.1, momentum=0.9, dampening=0.1,
               weight_decay=1e-4, max_iter=60000, poly_power=0.9, ignore_label=255, bn_momentum=0.02, pretrained=None,
               kernel_order="hybrid", device=None):
    assert torch.cuda.is_available(), "the fine-tuning step runs on a gfx950 GPU (no CPU path)"
    self.device = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
    cfg = get_config(["net.normalize_feature=False", "opt.bn_momentum=%g" % bn_momentum])
    self.model = load_model(model)(in_channels, num_labels, cfg, D=3).to(self.device)
    if pretrained is not None:  # a pre-training checkpoint: everything whose name and shape match (not the head)
      state = torch.load(pretrained, map_location="cpu", weights_only=False) if isinstance(pretrained, str) else pretrained
      weights = ck.convert_kernel_order(self.model, ck.strip_prefixes(state.get("state_dict", state)), kernel_order)
      own = self.model.state_dict()
      own.update(ck.load_state_with_same_shape(self.model, weights))
      self.model.load_state_dict(own)
    self.flat = FlatParameters(self.model.parameters())
    self.engine = NativeEngine(self.model, self.flat, in_channels=in_channels, n_passes=1)
    # downstream/semseg/lib/solvers.py:52-60: SGD(lr, momentum=sgd_momentum 0.9, dampening=sgd_dampening 0.1, weight_decay)
    self.optimizer = FlatSGD(self.flat, lr=lr, momentum=momentum, weight_decay=weight_decay, dampening=dampening)
    self.scheduler = PolyLR(self.optimizer, max_iter=max_iter, power=poly_power)
    self.ignore_label, self.num_labels, self.curr_iter = ignore_label, num_labels, 0

  def forward(self, coords, feats, training=True):
    st = ME.SparseTensor(feats, coords=coords).to(self.device)
    return self.engine.forward(0, st, training=training)

  def train_iter(self, coords, feats, target):
    self.model.train()
    self.optimizer.zero_grad()
    if torch.is_tensor(target) and not target.is_cuda and target.numel():  # free on the host; as torch's CrossEntropyLoss
      bad = (target != self.ignore_label) & ((target < 0) | (target >= self.num_labels))
      if bool(bad.any()):
        raise IndexError("Target %d is out of bounds (classes 0..%d, ignore label %d)" %
                         (int(target[bad][0]), self.num_labels - 1, self.ignore_label))
    logits = self.forward(coords, feats).requires_grad_(True)
    tgt = target.to(self.device)
    loss = PF.SoftmaxCrossEntropyFunction.apply(logits, tgt, self.ignore_label)
    loss.backward()
    self.engine.backward(0, logits.grad)
    self.optimizer.step()
    self.scheduler.step()
    self.curr_iter += 1
    pred = logits.detach().max(1)[1]
    return {"loss": loss.detach(), "score": precision_at_one(pred, tgt, self.ignore_label), "pred": pred}

  @torch.no_grad()
  def evaluate(self, coords, feats, target):
    """(mIoU in %, per-class IoU, confusion matrix) of one batch in eval mode (running BN estimates)."""
    self.model.eval()
    pred = self.forward(coords, feats, training=False).max(1)[1].cpu().numpy()
    hist = fast_hist(pred, np.asarray(target), self.num_labels)
    ious = per_class_iu(hist) * 100.0
    return float(np.nanmean(ious)), ious, hist
