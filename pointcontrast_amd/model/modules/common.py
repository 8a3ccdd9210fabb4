"""Layer factories of the pre-training models, same public names as the reference's
pc/model/modules/common.py (ConvType, NormType, conv, conv_tr, get_norm) so model code written
against it runs unchanged on the libpcmi-backed Minkowski surface."""
from enum import Enum

from ... import minkowski as ME


class NormType(Enum):
  BATCH_NORM = 0
  SPARSE_LAYER_NORM = 1
  SPARSE_INSTANCE_NORM = 2
  SPARSE_SWITCH_NORM = 3


class ConvType(Enum):
  """Kernel-region selector (pc/model/modules/common.py:28-45)."""
  HYPERCUBE = 0
  SPATIAL_HYPERCUBE = 1
  SPATIO_TEMPORAL_HYPERCUBE = 2
  HYPERCROSS = 3
  SPATIAL_HYPERCROSS = 4
  SPATIO_TEMPORAL_HYPERCROSS = 5
  SPATIAL_HYPERCUBE_TEMPORAL_HYPERCROSS = 6

  def __int__(self):
    return self.value


# ConvType -> (ME region, per-axis region list or None); D=3 only, so "temporal" axes vanish and
# the hybrid type degenerates to a cube with ME's HYBRID offset enumeration (Appendix A7).
_REGION = {
    ConvType.HYPERCUBE: (ME.RegionType.HYPERCUBE, None),
    ConvType.SPATIAL_HYPERCUBE: (ME.RegionType.HYPERCUBE, None),
    ConvType.SPATIO_TEMPORAL_HYPERCUBE: (ME.RegionType.HYPERCUBE, None),
    ConvType.HYPERCROSS: (ME.RegionType.HYPERCROSS, None),
    ConvType.SPATIAL_HYPERCROSS: (ME.RegionType.HYPERCROSS, None),
    ConvType.SPATIO_TEMPORAL_HYPERCROSS: (ME.RegionType.HYPERCROSS, None),
    ConvType.SPATIAL_HYPERCUBE_TEMPORAL_HYPERCROSS: (ME.RegionType.HYBRID, [ME.RegionType.HYPERCUBE] * 3),
}


def convert_conv_type(conv_type, kernel_size, D):
  assert isinstance(conv_type, ConvType), "conv_type must be of ConvType"
  assert D == 3, "the pre-training path is 3-D"
  region_type, axis_types = _REGION[conv_type]
  if isinstance(kernel_size, (list, tuple)):
    kernel_size = list(kernel_size[:3])
  return region_type, axis_types, kernel_size


def get_norm(norm_type, n_channels, D, bn_momentum=0.1):
  if norm_type == NormType.BATCH_NORM:
    return ME.MinkowskiBatchNorm(n_channels, momentum=bn_momentum)
  raise ValueError("Norm type: %s not supported on the pre-training path" % (norm_type,))


def _make(cls, in_planes, out_planes, kernel_size, stride, dilation, bias, conv_type, D):
  assert D > 0, "Dimension must be a positive integer"
  region_type, axis_types, kernel_size = convert_conv_type(conv_type, kernel_size, D)
  gen = ME.KernelGenerator(kernel_size, stride, dilation, region_type=region_type, axis_types=axis_types, dimension=D)
  return cls(in_channels=in_planes, out_channels=out_planes, kernel_size=kernel_size, stride=stride,
             dilation=dilation, has_bias=bias, kernel_generator=gen, dimension=D)


def conv(in_planes, out_planes, kernel_size, stride=1, dilation=1, bias=False, conv_type=ConvType.HYPERCUBE, D=-1):
  return _make(ME.MinkowskiConvolution, in_planes, out_planes, kernel_size, stride, dilation, bias, conv_type, D)


def conv_tr(in_planes, out_planes, kernel_size, upsample_stride=1, dilation=1, bias=False,
            conv_type=ConvType.HYPERCUBE, D=-1):
  return _make(ME.MinkowskiConvolutionTranspose, in_planes, out_planes, kernel_size, upsample_stride, dilation, bias,
               conv_type, D)
