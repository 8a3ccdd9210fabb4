"""Oracle restatement of the Res16UNet family (torch-CPU, autograd).

TEST INFRASTRUCTURE (see oracle/__init__.py).  Follows
  pc/model/res16unet.py:36-268 (wiring, forward), :270-275 (34 / 34C),
  pc/model/resnet.py:93-140 (BN init, _make_layer, 1x1 downsample),
  pc/model/modules/resnet_block.py:44-60 (BasicBlock.forward),
  downstream/semseg/models/res16unet.py:263-265 (Res16UNet14 = LAYERS (1,)*8).
State-dict names equal the reference's (conv ``.kernel``/``.bias``, BN
``.bn.*``) so oracle and device model exchange weights by name.
"""
import math

import torch
import torch.nn as nn

from . import sparse_ref as sr


class ConvRef(nn.Module):
  """ME.MinkowskiConvolution / ConvolutionTranspose restated (Appendix A3-A6, A8)."""

  def __init__(self, cin, cout, kernel_size, stride=1, region=sr.HYPERCUBE, bias=False,
               transpose=False):
    super().__init__()
    self.kernel_size, self.stride, self.region, self.transpose = kernel_size, stride, region, transpose
    K = kernel_size ** 3
    shape = (cin, cout) if K == 1 else (K, cin, cout)
    self.kernel = nn.Parameter(torch.empty(shape))
    self.bias = nn.Parameter(torch.empty(1, cout)) if bias else None
    n = (cout if transpose else cin) * K
    bound = 1.0 / math.sqrt(n)
    with torch.no_grad():
      self.kernel.uniform_(-bound, bound)
      if bias:
        self.bias.uniform_(-bound, bound)

  def forward(self, x):
    cm = x.coords_man
    if self.kernel.dim() == 2:
      out = x.F @ self.kernel
      if self.bias is not None:
        out = out + self.bias
      return sr.SparseTensorRef(out, coords_key=x.coords_key, coords_manager=cm)
    if self.transpose:
      out_key = cm.key_at_stride(x.tensor_stride // self.stride)
      km = cm.kernel_map(out_key, x.coords_key, self.kernel_size, self.region).swapped()
    elif self.stride > 1:
      out_key = cm.stride(x.coords_key, self.stride)
      km = cm.kernel_map(x.coords_key, out_key, self.kernel_size, self.region)
    else:
      out_key = x.coords_key
      km = cm.kernel_map(x.coords_key, out_key, self.kernel_size, self.region)
    out = sr.sparse_conv(x.F, self.kernel, km, self.bias)
    return sr.SparseTensorRef(out, coords_key=out_key, coords_manager=cm)


class BatchNormRef(nn.Module):
  """ME.MinkowskiBatchNorm == BatchNorm1d on .F, exposed as ``.bn`` (A9)."""

  def __init__(self, c, momentum):
    super().__init__()
    self.bn = nn.BatchNorm1d(c, eps=1e-5, momentum=momentum)

  def forward(self, x):
    return sr.SparseTensorRef(self.bn(x.F), coords_key=x.coords_key, coords_manager=x.coords_man)


_RELU_IMPL = [torch.relu]  # every ReLU of the oracle goes through this slot (see relu_masks)


def relu_features(f):
  return _RELU_IMPL[0](f)


class relu_masks:
  """Context manager for deterministic gradient comparisons.  An activation that lies within fp32 round-off of
  zero gets opposite ReLU masks in two implementations and then moves whole gradient tensors by percents -- a
  property of ReLU, not an error of either side.  ``relu_masks(record=lst)`` appends the mask (x > 0) of every
  ReLU call, in call order, to ``lst``; ``relu_masks(apply=lst)`` replaces ReLU by ``x * mask`` with the recorded
  masks, so that two runs (fp64 / fp32 oracle, device) differentiate exactly the same piecewise-linear function.
  ``segment`` ("head" / "tail"): the masks come from a JOINT two-cloud pass of the device (rows of cloud 0 first, then
  cloud 1, at every level -- NativeEngine.relu_masks) and this forward is cloud 0 / cloud 1: the first / last rows of
  every mask are used.  ``flips`` / ``total`` count the activations whose own sign disagrees with the imposed mask."""

  def __init__(self, record=None, apply=None, segment=None):
    assert (record is None) != (apply is None) and segment in (None, "head", "tail")
    self.record, self.apply, self.pos, self.segment, self.flips, self.total = record, apply, 0, segment, 0, 0

  def _fn(self, f):
    if self.record is not None:
      self.record.append((f > 0).clone())
      return torch.relu(f)
    m = self.apply[self.pos]
    self.pos += 1
    if self.segment is not None and m.shape[0] >= f.shape[0] and m.shape[1:] == f.shape[1:]:
      m = m[:f.shape[0]] if self.segment == "head" else m[m.shape[0] - f.shape[0]:]
    assert m.shape == f.shape, "ReLU call %d: mask %s vs activation %s" % (self.pos - 1, tuple(m.shape), tuple(f.shape))
    self.flips += int(((f > 0) != m).sum())
    self.total += m.numel()
    return f * m.to(f.dtype)

  def __enter__(self):
    self._old = _RELU_IMPL[0]
    _RELU_IMPL[0] = self._fn
    return self

  def __exit__(self, *exc):
    _RELU_IMPL[0] = self._old
    if self.apply is not None and exc[0] is None:
      assert self.pos == len(self.apply), "%d ReLU calls, %d masks" % (self.pos, len(self.apply))
    return False


def _relu(x):
  return sr.SparseTensorRef(relu_features(x.F), coords_key=x.coords_key, coords_manager=x.coords_man)


class BasicBlockRef(nn.Module):
  """pc/model/modules/resnet_block.py:13-60.  Block BNs keep momentum 0.1
  (the reference never forwards bn_momentum into the block, resnet.py:120-139)."""

  def __init__(self, inplanes, planes, downsample=None):
    super().__init__()
    self.conv1 = ConvRef(inplanes, planes, 3, region=sr.HYBRID)
    self.norm1 = BatchNormRef(planes, 0.1)
    self.conv2 = ConvRef(planes, planes, 3, region=sr.HYBRID)
    self.norm2 = BatchNormRef(planes, 0.1)
    self.downsample = downsample

  def forward(self, x):
    out = _relu(self.norm1(self.conv1(x)))
    out = self.norm2(self.conv2(out))
    res = x if self.downsample is None else self.downsample(x)
    return sr.SparseTensorRef(relu_features(out.F + res.F), coords_key=out.coords_key,
                              coords_manager=out.coords_man)


class Res16UNetRef(nn.Module):
  PLANES = (32, 64, 128, 256, 256, 256, 256, 256)
  LAYERS = (2, 2, 2, 2, 2, 2, 2, 2)
  INIT_DIM = 32

  def __init__(self, in_channels, out_channels, bn_momentum=0.05, normalize_feature=True,
               conv1_kernel_size=3):
    super().__init__()
    P, L = self.PLANES, self.LAYERS
    self.normalize_feature = normalize_feature
    self.inplanes = self.INIT_DIM
    self.conv0p1s1 = ConvRef(in_channels, self.inplanes, conv1_kernel_size)
    self.bn0 = BatchNormRef(self.inplanes, bn_momentum)
    enc = [("conv1p1s2", "bn1", "block1"), ("conv2p2s2", "bn2", "block2"),
           ("conv3p4s2", "bn3", "block3"), ("conv4p8s2", "bn4", "block4")]
    for i, (cn, bn, blk) in enumerate(enc):
      setattr(self, cn, ConvRef(self.inplanes, self.inplanes, 2, stride=2))
      setattr(self, bn, BatchNormRef(self.inplanes, bn_momentum))
      setattr(self, blk, self._make_layer(P[i], L[i], bn_momentum))
    dec = [("convtr4p16s2", "bntr4", "block5", P[2]), ("convtr5p8s2", "bntr5", "block6", P[1]),
           ("convtr6p4s2", "bntr6", "block7", P[0]), ("convtr7p2s2", "bntr7", "block8", self.INIT_DIM)]
    for i, (cn, bn, blk, skip) in enumerate(dec):
      setattr(self, cn, ConvRef(self.inplanes, P[4 + i], 2, stride=2, transpose=True))
      setattr(self, bn, BatchNormRef(P[4 + i], bn_momentum))
      self.inplanes = P[4 + i] + skip
      setattr(self, blk, self._make_layer(P[4 + i], L[4 + i], bn_momentum))
    self.final = ConvRef(P[7], out_channels, 1, bias=True)
    for m in self.modules():  # pc/model/resnet.py:93-97
      if isinstance(m, BatchNormRef):
        nn.init.constant_(m.bn.weight, 1)
        nn.init.constant_(m.bn.bias, 0)

  def _make_layer(self, planes, blocks, bn_momentum):
    down = None
    if self.inplanes != planes:
      down = nn.Sequential(ConvRef(self.inplanes, planes, 1), BatchNormRef(planes, bn_momentum))
    layers = [BasicBlockRef(self.inplanes, planes, down)]
    self.inplanes = planes
    for _ in range(1, blocks):
      layers.append(BasicBlockRef(planes, planes))
    return nn.Sequential(*layers)

  def forward(self, x):
    def cat(a, b):  # MinkowskiOps.cat, pc/model/res16unet.py:235
      assert a.coords_key == b.coords_key
      return sr.SparseTensorRef(torch.cat([a.F, b.F], 1), coords_key=a.coords_key,
                                coords_manager=a.coords_man)
    out_p1 = _relu(self.bn0(self.conv0p1s1(x)))
    out_b1p2 = self.block1(_relu(self.bn1(self.conv1p1s2(out_p1))))
    out_b2p4 = self.block2(_relu(self.bn2(self.conv2p2s2(out_b1p2))))
    out_b3p8 = self.block3(_relu(self.bn3(self.conv3p4s2(out_b2p4))))
    out = self.block4(_relu(self.bn4(self.conv4p8s2(out_b3p8))))
    out = self.block5(cat(_relu(self.bntr4(self.convtr4p16s2(out))), out_b3p8))
    out = self.block6(cat(_relu(self.bntr5(self.convtr5p8s2(out))), out_b2p4))
    out = self.block7(cat(_relu(self.bntr6(self.convtr6p4s2(out))), out_b1p2))
    out = self.block8(cat(_relu(self.bntr7(self.convtr7p2s2(out))), out_p1))
    out = self.final(out)
    if self.normalize_feature:  # pc/model/res16unet.py:262-266, no eps
      out = sr.SparseTensorRef(out.F / torch.norm(out.F, p=2, dim=1, keepdim=True),
                               coords_key=out.coords_key, coords_manager=out.coords_man)
    return out


class Res16UNet14Ref(Res16UNetRef):
  LAYERS = (1, 1, 1, 1, 1, 1, 1, 1)


class Res16UNet34Ref(Res16UNetRef):
  LAYERS = (2, 3, 4, 6, 2, 2, 2, 2)


class Res16UNet34CRef(Res16UNet34Ref):
  PLANES = (32, 64, 128, 256, 256, 128, 96, 96)


MODELS = {"Res16UNet14": Res16UNet14Ref, "Res16UNet34": Res16UNet34Ref, "Res16UNet34C": Res16UNet34CRef}
