"""Res16UNet family: 4 strided-conv encoder stages, 4 transposed-conv decoder stages with skip
concatenation, 1x1 head and L2-normalised output (pc/model/res16unet.py:17-275).

Module and parameter names match the reference so checkpoints interchange
(conv0p1s1, bn0, conv{i}p{s}s2, bn{i}, block{i}, convtr{i}p{s}s2, bntr{i}, final).
BatchNorm+ReLU pairs run as one fused libpcmi kernel.
"""
from .. import minkowski as ME
from .modules.common import ConvType, NormType, conv, conv_tr, get_norm
from .modules.resnet_block import BasicBlock
from .resnet import ResNetBase


class Res16UNetBase(ResNetBase):
  BLOCK = None
  PLANES = (32, 64, 128, 256, 256, 256, 256, 256)
  DILATIONS = (1, 1, 1, 1, 1, 1, 1, 1)
  LAYERS = (2, 2, 2, 2, 2, 2, 2, 2)
  INIT_DIM = 32
  OUT_PIXEL_DIST = 1
  NORM_TYPE = NormType.BATCH_NORM
  NON_BLOCK_CONV_TYPE = ConvType.SPATIAL_HYPERCUBE
  CONV_TYPE = ConvType.SPATIAL_HYPERCUBE_TEMPORAL_HYPERCROSS

  # (conv name, bn name, block name, tensor stride of the conv input)
  ENCODER = (("conv1p1s2", "bn1", "block1"), ("conv2p2s2", "bn2", "block2"),
             ("conv3p4s2", "bn3", "block3"), ("conv4p8s2", "bn4", "block4"))
  DECODER = (("convtr4p16s2", "bntr4", "block5"), ("convtr5p8s2", "bntr5", "block6"),
             ("convtr6p4s2", "bntr6", "block7"), ("convtr7p2s2", "bntr7", "block8"))

  def __init__(self, in_channels, out_channels, config, D=3):
    super().__init__(in_channels, out_channels, config, D)
    self.normalize_feature = config.net.normalize_feature

  def network_initialization(self, in_channels, out_channels, config, D):
    assert D == 3, "the pre-training path is 3-D"
    mom = config.opt.bn_momentum
    nb = dict(conv_type=self.NON_BLOCK_CONV_TYPE, D=D)
    stage = lambda i: self._make_layer(self.BLOCK, self.PLANES[i], self.LAYERS[i], dilation=self.DILATIONS[i],
                                       norm_type=self.NORM_TYPE, bn_momentum=mom)
    self.inplanes = self.INIT_DIM
    self.conv0p1s1 = conv(in_channels, self.inplanes, kernel_size=config.net.conv1_kernel_size, stride=1, dilation=1, **nb)
    self.bn0 = get_norm(self.NORM_TYPE, self.inplanes, D, bn_momentum=mom)
    for i, (cname, bname, blk) in enumerate(self.ENCODER):
      setattr(self, cname, conv(self.inplanes, self.inplanes, kernel_size=2, stride=2, dilation=1, **nb))
      setattr(self, bname, get_norm(self.NORM_TYPE, self.inplanes, D, bn_momentum=mom))
      setattr(self, blk, stage(i))
    # width of the skip tensor concatenated in decoder stage i (pc/model/res16unet.py:143,163,183,203): the outputs of
    # block3 / block2 / block1 (PLANES[k] * expansion) and of the stem (INIT_DIM, no block -> no expansion)
    skip = [self.PLANES[2] * self.BLOCK.expansion, self.PLANES[1] * self.BLOCK.expansion,
            self.PLANES[0] * self.BLOCK.expansion, self.INIT_DIM]
    for i, (cname, bname, blk) in enumerate(self.DECODER):
      up = self.PLANES[4 + i]
      setattr(self, cname, conv_tr(self.inplanes, up, kernel_size=2, upsample_stride=2, dilation=1, bias=False, **nb))
      setattr(self, bname, get_norm(self.NORM_TYPE, up, D, bn_momentum=mom))
      self.inplanes = up + skip[i]
      setattr(self, blk, stage(4 + i))
    self.final = conv(self.PLANES[7], out_channels, kernel_size=1, stride=1, bias=True, D=D)
    self.relu = ME.MinkowskiReLU(inplace=True)

  def forward(self, x):
    if x.coords_man is not None:
      x.coords_man.plan_unet(len(self.ENCODER))  # all levels / maps up front, on the plan stream
    out = self.bn0(self.conv0p1s1(x), relu=True)
    skips = [out]
    for cname, bname, blk in self.ENCODER:
      out = getattr(self, bname)(getattr(self, cname)(out), relu=True)
      out = getattr(self, blk)(out)
      skips.append(out)
    skips.pop()  # the deepest stage output is the decoder input, not a skip
    for cname, bname, blk in self.DECODER:
      out = getattr(self, bname)(getattr(self, cname)(out), relu=True)
      out = ME.MinkowskiOps.cat(out, skips.pop())
      out = getattr(self, blk)(out)
    out = self.final(out)
    return ME.l2_normalize(out) if self.normalize_feature else out


class Res16UNet14(Res16UNetBase):
  """downstream/semseg/models/res16unet.py:263-265 (BASELINE config #1 plumbing model)."""
  BLOCK = BasicBlock
  LAYERS = (1, 1, 1, 1, 1, 1, 1, 1)


class Res16UNet34(Res16UNetBase):
  BLOCK = BasicBlock
  LAYERS = (2, 3, 4, 6, 2, 2, 2, 2)


class Res16UNet34C(Res16UNet34):
  PLANES = (32, 64, 128, 256, 256, 128, 96, 96)
