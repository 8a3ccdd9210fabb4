"""Base classes of the sparse residual networks (pc/model/resnet.py:15-22, 93-140)."""
import torch.nn as nn

from .. import minkowski as ME
from .modules.common import ConvType, NormType, conv, get_norm


class Model(ME.MinkowskiNetwork):
  OUT_PIXEL_DIST = -1

  def __init__(self, in_channels, out_channels, config, D, **kwargs):
    super().__init__(D)
    self.in_channels, self.out_channels, self.config = in_channels, out_channels, config


class ResNetBase(Model):
  BLOCK = None
  LAYERS = ()
  INIT_DIM = 64
  PLANES = (64, 128, 256, 512)
  OUT_PIXEL_DIST = 32
  CONV_TYPE = ConvType.HYPERCUBE

  def __init__(self, in_channels, out_channels, config, D=3, **kwargs):
    assert self.BLOCK is not None and self.OUT_PIXEL_DIST > 0
    super().__init__(in_channels, out_channels, config, D, **kwargs)
    self.network_initialization(in_channels, out_channels, config, D)
    self.weight_initialization()

  def network_initialization(self, in_channels, out_channels, config, D):
    raise NotImplementedError("only the Res16UNet family is on the pre-training path")

  def weight_initialization(self):
    for m in self.modules():
      if isinstance(m, ME.MinkowskiBatchNorm):
        nn.init.constant_(m.bn.weight, 1)
        nn.init.constant_(m.bn.bias, 0)

  def _make_layer(self, block, planes, blocks, stride=1, dilation=1, norm_type=NormType.BATCH_NORM, bn_momentum=0.1):
    """A stage of `blocks` residual blocks.  As in the reference the 1x1 projection shortcut gets
    `bn_momentum` while the BNs inside the blocks keep the block default (0.1): bn_momentum is
    not forwarded to block(...) (pc/model/resnet.py:120-139)."""
    out_planes = planes * block.expansion
    shortcut = None
    if stride != 1 or self.inplanes != out_planes:
      shortcut = nn.Sequential(
          conv(self.inplanes, out_planes, kernel_size=1, stride=stride, bias=False, D=self.D),
          get_norm(norm_type, out_planes, D=self.D, bn_momentum=bn_momentum))
    stage = [block(self.inplanes, planes, stride=stride, dilation=dilation, downsample=shortcut,
                   conv_type=self.CONV_TYPE, D=self.D)]
    self.inplanes = out_planes
    stage += [block(self.inplanes, planes, stride=1, dilation=dilation, conv_type=self.CONV_TYPE, D=self.D)
              for _ in range(1, blocks)]
    return nn.Sequential(*stage)
