"""Residual block of the Res16UNet family (pc/model/modules/resnet_block.py:13-60), with the
BatchNorm -> (+residual) -> ReLU tail fused into one libpcmi kernel."""
import torch.nn as nn

from .common import ConvType, NormType, conv, get_norm


class BasicBlockBase(nn.Module):
  expansion = 1
  NORM_TYPE = NormType.BATCH_NORM

  def __init__(self, inplanes, planes, stride=1, dilation=1, downsample=None, conv_type=ConvType.HYPERCUBE,
               bn_momentum=0.1, D=3):
    super().__init__()
    mk = lambda cin, s: conv(cin, planes, kernel_size=3, stride=s, dilation=dilation, conv_type=conv_type, D=D)
    self.conv1, self.norm1 = mk(inplanes, stride), get_norm(self.NORM_TYPE, planes, D, bn_momentum=bn_momentum)
    self.conv2, self.norm2 = mk(planes, 1), get_norm(self.NORM_TYPE, planes, D, bn_momentum=bn_momentum)
    self.downsample = downsample

  def forward(self, x):
    out = self.norm1(self.conv1(x), relu=True)
    out = self.conv2(out)  # before the shortcut, the order of the reference's forward (:46-55)
    if self.downsample is None:
      shortcut = x
    else:  # nn.Sequential(1x1 conv, BN) built by ResNetBase._make_layer
      shortcut = self.downsample[1](self.downsample[0](x))
    # relu(norm2(conv2(out)) + shortcut) in one pass over the activation
    return self.norm2(out, residual=shortcut, relu=True)


class BasicBlock(BasicBlockBase):
  NORM_TYPE = NormType.BATCH_NORM
