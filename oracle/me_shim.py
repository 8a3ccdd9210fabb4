"""CPU stand-in for the `MinkowskiEngine` 0.4.3 package, backed by the oracle ops.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Purpose: execute the reference's OWN model
source (pc/model/res16unet.py, resnet.py, modules/common.py, modules/resnet_block.py --
imported unmodified from /root/reference when it is present) with
``sys.modules["MinkowskiEngine"]`` pointing here, so that the network WIRING the parity tests
compare against is the reference's source and not a restatement.  The arithmetic behind every
symbol is oracle/sparse_ref.py + oracle/model_ref.py (ConvRef / BatchNormRef), i.e. the same
single implementation that tests/test_oracle_dense.py pins against dense torch conv3d.

Symbols = exactly what pc/model/*.py touches:
  SparseTensor, MinkowskiNetwork, MinkowskiConvolution(.Transpose), MinkowskiBatchNorm,
  MinkowskiInstanceNorm (name only), MinkowskiReLU, MinkowskiGlobalPooling (name only),
  Minkowski{Avg,Sum}Pooling / AvgUnpooling (names only), KernelGenerator, RegionType,
  MinkowskiOps.cat, utils.sparse_quantize.
`install()` registers this module as MinkowskiEngine (+ MinkowskiEngine.MinkowskiOps).
"""
import sys
import types
from enum import Enum

import torch
import torch.nn as nn

from . import model_ref as mr
from . import sparse_ref as sr


class RegionType(Enum):
  """ME.RegionType (pc/model/modules/common.py:47-60 iterates it and reads .value)."""
  HYPERCUBE = 0
  HYPERCROSS = 1
  CUSTOM = 2
  HYBRID = 3


SparseTensor = sr.SparseTensorRef


def _scalar(v, what):
  if isinstance(v, (list, tuple)):
    assert len(set(v[:3])) == 1, "%s must be isotropic in the 3 spatial axes" % what
    return int(v[0])
  return int(v)


class KernelGenerator:
  """ME.KernelGenerator(kernel_size, stride, dilation, region_type=, axis_types=, dimension=)
  (pc/model/modules/common.py:127-128, 151-157)."""

  def __init__(self, kernel_size=-1, stride=1, dilation=1, is_transpose=False, region_type=RegionType.HYPERCUBE,
               region_offsets=None, axis_types=None, dimension=-1):
    assert dimension == 3
    self.kernel_size, self.stride, self.dilation = _scalar(kernel_size, "kernel_size"), _scalar(stride, "stride"), _scalar(dilation, "dilation")
    assert self.dilation == 1
    self.region_type, self.axis_types = region_type, axis_types
    if region_type == RegionType.HYBRID:
      assert axis_types is not None and all(a == RegionType.HYPERCUBE for a in axis_types[:3])
      self.region = sr.HYBRID
    else:
      assert region_type == RegionType.HYPERCUBE, region_type
      self.region = sr.HYPERCUBE


class MinkowskiNetwork(nn.Module):

  def __init__(self, D):
    super().__init__()
    self.D = D


class _Conv(mr.ConvRef):
  _transpose = False

  def __init__(self, in_channels, out_channels, kernel_size=-1, stride=1, dilation=1, has_bias=False,
               kernel_generator=None, dimension=None):
    assert dimension == 3
    if kernel_generator is None:
      kernel_generator = KernelGenerator(kernel_size, stride, dilation, dimension=dimension)
    kg = kernel_generator
    assert kg.kernel_size == _scalar(kernel_size, "kernel_size") and kg.stride == _scalar(stride, "stride")
    super().__init__(in_channels, out_channels, kg.kernel_size, stride=kg.stride, region=kg.region, bias=has_bias,
                     transpose=self._transpose)
    self.kernel_generator = kg


class MinkowskiConvolution(_Conv):
  _transpose = False


class MinkowskiConvolutionTranspose(_Conv):
  _transpose = True


class MinkowskiBatchNorm(mr.BatchNormRef):

  def __init__(self, num_features, eps=1e-5, momentum=0.1, affine=True, track_running_stats=True):
    assert eps == 1e-5 and affine and track_running_stats
    super().__init__(num_features, momentum)


class MinkowskiReLU(nn.Module):

  def __init__(self, inplace=False):
    super().__init__()

  def forward(self, x):
    return mr._relu(x)


def cat(*tensors):
  """MinkowskiOps.cat (pc/model/res16unet.py:235,242,249,256)."""
  key = tensors[0].coords_key
  assert all(t.coords_key == key for t in tensors), "cat: tensors must share coords_key"
  return sr.SparseTensorRef(torch.cat([t.F for t in tensors], dim=1), coords_key=key, coords_manager=tensors[0].coords_man)


def _name_only(name):

  class _Missing(nn.Module):

    def __init__(self, *a, **k):
      raise NotImplementedError("%s is imported but never instantiated by the pre-training path" % name)

  _Missing.__name__ = name
  return _Missing


MinkowskiInstanceNorm = _name_only("MinkowskiInstanceNorm")
MinkowskiGlobalPooling = _name_only("MinkowskiGlobalPooling")
MinkowskiAvgPooling = _name_only("MinkowskiAvgPooling")
MinkowskiAvgUnpooling = _name_only("MinkowskiAvgUnpooling")
MinkowskiSumPooling = _name_only("MinkowskiSumPooling")

MinkowskiOps = types.ModuleType("MinkowskiEngine.MinkowskiOps")
MinkowskiOps.cat = cat
utils = types.SimpleNamespace(sparse_quantize=sr.sparse_quantize)


def install():
  """sys.modules["MinkowskiEngine"] = this module (what `import MinkowskiEngine as ME`, `from MinkowskiEngine
  import ...` and `import MinkowskiEngine.MinkowskiOps as me` in the reference's files then resolve to)."""
  me = sys.modules[__name__]
  sys.modules["MinkowskiEngine"] = me
  sys.modules["MinkowskiEngine.MinkowskiOps"] = MinkowskiOps
  return me
