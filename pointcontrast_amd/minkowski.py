"""The slice of the MinkowskiEngine 0.4.3 Python surface that the PointContrast
pre-training path uses, backed by libpcmi (HIP kernels on gfx950).

Symbols (exhaustive list of what pc/ imports from ME -- SURVEY.md 8b):
  SparseTensor (.F .C .coords_key .coords_man .tensor_stride .to), CoordsKey,
  CoordsManager, MinkowskiNetwork, MinkowskiConvolution,
  MinkowskiConvolutionTranspose, MinkowskiBatchNorm (.bn), MinkowskiReLU,
  KernelGenerator, RegionType, MinkowskiOps.cat, utils.sparse_quantize.
Usage mirrors the reference: ``import pointcontrast_amd.minkowski as ME``.
Anything else ME offers (pooling, instance norm, pruning ...) is outside the hot
path and raises NotImplementedError.
"""
import contextlib
import ctypes as C
import math
import sys
import threading
import types
from enum import Enum

import numpy as np
import torch
import torch.nn as nn

from . import _lib
from ._lib import lib, check, KMap
from . import functional as PF
from .runtime import ptr, handle_pool, require_cuda


class RegionType(Enum):
  """ME.RegionType (values as in ME 0.4.x; pc/model/modules/common.py:47-60)."""
  HYPERCUBE = 0
  HYPERCROSS = 1
  CUSTOM = 2
  HYBRID = 3


def _region_code(region_type, axis_types):
  if region_type == RegionType.HYPERCUBE:
    return 0
  if region_type == RegionType.HYBRID:
    if axis_types is not None and any(a != RegionType.HYPERCUBE for a in axis_types):
      raise NotImplementedError("HYBRID regions with non-cube axes are not on the 3-D hot path")
    return 3
  raise NotImplementedError("region type %s is not used by the pre-training path" % (region_type,))


class KernelGenerator:
  """ME.KernelGenerator: (kernel_size, stride, dilation, region) of a conv
  (pc/model/modules/common.py:127-128,151-157)."""

  def __init__(self, kernel_size=-1, stride=1, dilation=1, is_transpose=False, region_type=RegionType.HYPERCUBE,
               region_offsets=None, axis_types=None, dimension=-1):
    assert dimension == 3, "only D=3 is on the hot path"

    def scalar(v, name):
      if isinstance(v, (list, tuple)):
        assert len(set(v[:3])) == 1, "%s must be isotropic" % name
        return int(v[0])
      return int(v)

    self.kernel_size = scalar(kernel_size, "kernel_size")
    self.stride = scalar(stride, "stride")
    self.dilation = scalar(dilation, "dilation")
    assert self.dilation == 1, "dilation != 1 is not used by Res16UNet (DILATIONS = 1, pc/model/res16unet.py:20)"
    self.region_type, self.axis_types, self.dimension = region_type, axis_types, dimension
    self.region_code = _region_code(region_type, axis_types)
    self.kernel_volume = self.kernel_size ** 3

  def get_kernel(self, tensor_stride=1):
    """(region_type, offsets [K,3] in units of the tensor stride, None) in weight-slice order."""
    buf = (C.c_int32 * (27 * 3))()
    K = C.c_int()
    check(lib.pcmi_kernel_offsets(self.kernel_size, self.region_code, buf, C.byref(K)))
    offs = np.ctypeslib.as_array(buf)[:K.value * 3].reshape(K.value, 3).copy() * int(tensor_stride)
    return self.region_type, torch.from_numpy(offs), None


class CoordsKey:
  """Identifies one coordinate set inside a CoordsManager."""

  def __init__(self, key, tensor_stride, D=3):
    self.key, self.tensor_stride, self.D = int(key), int(tensor_stride), D

  def getKey(self):
    return self.key

  def getTensorStride(self):
    return [self.tensor_stride] * self.D

  def __eq__(self, other):
    return isinstance(other, CoordsKey) and self.key == other.key and self.tensor_stride == other.tensor_stride

  def __hash__(self):
    return hash((self.key, self.tensor_stride))

  def __repr__(self):
    return "CoordsKey(key=%d, tensor_stride=%d)" % (self.key, self.tensor_stride)


class CoordsManager:
  """ME.CoordsManager on the device: hash, strided coordinate sets and kernel maps are
  HIP kernels (ME 0.4.x ran them on the CPU).  Coordinate work is enqueued on a side
  ("plan") stream so that building the next SparseTensor's maps overlaps the compute stream.
  The per-call entry points (stride, kernel_map) are host-synchronous on that stream; plan_unet
  synchronises once (level sizes) and leaves its maps enqueued -- the native engine orders its
  stream behind the plan itself (pcmi_net_forward), per-layer callers get their maps through
  kernel_map, which waits for what it hands out.

  defer_check: the insert is only enqueued; duplicate / out-of-range coordinates are then
  reported by the next synchronising call (plan_unet, stride, kernel_map) or by check()."""

  def __init__(self, coords, D=3, defer_check=False):
    assert D == 3
    require_cuda(coords, "CoordsManager")
    assert coords.dtype == torch.int32 and coords.dim() == 2 and coords.shape[1] == 4, \
        "coords must be an IntTensor [N, 1+3] with the batch index first"
    self.device = coords.device
    self.D = D
    self._plan = handle_pool.plan_stream(self.device)
    self._h, prev_use = handle_pool.acquire(self.device)
    self._maps = {}
    self._alive = True
    with torch.cuda.device(self.device):
      if prev_use is not None:
        self._plan.wait_event(prev_use)  # kernels of the previous owner may still read the arena
      check(lib.pcmi_coords_reset(self._h))
      # the caller's tensor was produced on the current stream
      self._plan.wait_stream(torch.cuda.current_stream(self.device))
      c = coords.contiguous()
      c.record_stream(self._plan)
      insert = lib.pcmi_coords_insert_deferred if defer_check else lib.pcmi_coords_insert
      check(insert(self._h, ptr(c), c.shape[0], self._st()))

  def check(self):
    """Reports a deferred insert's duplicate / out-of-range coordinates (synchronises the plan stream if needed)."""
    with torch.cuda.device(self.device):
      check(lib.pcmi_coords_check(self._h, self._st()))

  def _st(self):
    return C.c_void_p(self._plan.cuda_stream)

  def __del__(self):
    try:
      if getattr(self, "_alive", False):
        self._alive = False
        handle_pool.release(self.device, self._h)
    except Exception:
      pass

  # -- keys ---------------------------------------------------------------------------------
  def size(self, key):
    n, ts = C.c_int64(), C.c_int()
    check(lib.pcmi_coords_size(self._h, key.key if isinstance(key, CoordsKey) else int(key), C.byref(n), C.byref(ts)))
    return n.value

  def set_split(self, n_first):
    """Declares a two-segment batch: rows [0, n_first) are the first point cloud of a pair, the rest the second
    (disjoint batch indices).  The native engine then keeps BatchNorm statistics per segment, as the reference's two
    forward calls do (pcmi_coords_set_split).  Call before any strided level exists."""
    check(lib.pcmi_coords_set_split(self._h, int(n_first)))
    self.n_first = int(n_first)

  def split(self, key=0):
    """Rows of the first segment at `key` (None: single-segment batch)."""
    v = C.c_int64()
    check(lib.pcmi_coords_split(self._h, key.key if isinstance(key, CoordsKey) else int(key), C.byref(v)))
    return None if v.value < 0 else v.value

  def key(self, k):
    n, ts = C.c_int64(), C.c_int()
    check(lib.pcmi_coords_size(self._h, int(k), C.byref(n), C.byref(ts)))
    return CoordsKey(k, ts.value, self.D)

  def stride(self, in_key, stride=2):
    ok, n = C.c_int(), C.c_int64()
    with torch.cuda.device(self.device):
      check(lib.pcmi_coords_stride(self._h, in_key.key, int(stride), C.byref(ok), C.byref(n), self._st()))
    return CoordsKey(ok.value, in_key.tensor_stride * stride, self.D)

  def key_at_stride(self, tensor_stride):
    k = C.c_int()
    check(lib.pcmi_coords_key_at_stride(self._h, int(tensor_stride), C.byref(k)))
    return CoordsKey(k.value, tensor_stride, self.D)

  def get_coords(self, key):
    n = self.size(key)
    with torch.cuda.device(self.device), torch.cuda.stream(self._plan):
      out = torch.empty((n, 4), dtype=torch.int32, device=self.device)
      check(lib.pcmi_coords_get(self._h, key.key, ptr(out), self._st()))
    self._plan.synchronize()
    return out

  # -- kernel maps -----------------------------------------------------------------------------
  def kernel_map(self, in_key, out_key, kernel_size, stride, region_code):
    ck = (in_key.key, out_key.key, kernel_size, stride, region_code if kernel_size % 2 else 0)
    m = self._maps.get(ck)
    if m is None:
      m = KMap()
      with torch.cuda.device(self.device):
        check(lib.pcmi_kmap_get(self._h, in_key.key, out_key.key, kernel_size, stride, region_code, C.byref(m),
                                self._st()))
      self._maps[ck] = m  # the arena behind the pointers lives as long as this manager
    return m

  def export_map(self, m):
    """(nbr [K, n_out], pair_in [M], pair_out [M]) as torch int32 tensors (parity / debug)."""
    with torch.cuda.device(self.device), torch.cuda.stream(self._plan):
      nbr = torch.empty((m.K, m.n_out), dtype=torch.int32, device=self.device)
      pin = torch.empty(m.M, dtype=torch.int32, device=self.device)
      pout = torch.empty(m.M, dtype=torch.int32, device=self.device)
      check(lib.pcmi_kmap_export(C.byref(m), ptr(nbr), ptr(pin), ptr(pout), self._st()))
    self._plan.synchronize()
    return nbr, pin, pout

  def get_kernel_map(self, in_key, out_key, stride=1, kernel_size=3, region_type=RegionType.HYPERCUBE):
    """Per-offset (in_rows, out_rows) int tensors -- ME's get_kernel_map view of the pairs."""
    m = self.kernel_map(in_key, out_key, kernel_size, stride, _region_code(region_type, None))
    _, pin, pout = self.export_map(m)
    offs = list(m.offs_host[:m.K + 1])
    return [(pin[offs[k]:offs[k + 1]], pout[offs[k]:offs[k + 1]]) for k in range(m.K)]

  def plan_unet(self, n_down=4, first_region=0, block_region=3):
    """Performance hint: build all levels and maps of a U-Net forward now (on the plan stream)."""
    with torch.cuda.device(self.device):
      check(lib.pcmi_coords_plan_unet(self._h, n_down, first_region, block_region, self._st()))


_tls = threading.local()


@contextlib.contextmanager
def deferred_upload_sync():
  """Inside this context (per thread) ``SparseTensor.to(device)`` does NOT make the current compute stream wait for
  the upload / coordinate hash it enqueues on the plan stream; the event is left on the result as ``upload_event``
  and the consumer waits on it when it actually uses the tensor (``wait_upload``).  Used by the trainer's helper
  thread, which prepares the NEXT batch while the current step is still being enqueued: an immediate wait would
  stall the compute stream in the middle of that step."""
  old = getattr(_tls, "defer", False)
  _tls.defer = True
  try:
    yield
  finally:
    _tls.defer = old


class SymTensor:
  """Stand-in for a SparseTensor while a model is being lowered to a libpcmi network program
  (pointcontrast_amd/engine.py): the modules record ops on `tracer` instead of launching kernels.
  Model code written in the reference's unfused style -- ``relu(bn(x))``, ``out += residual; relu(out)``,
  ``SparseTensor(out.F / torch.norm(out.F, p=2, dim=1, keepdim=True), ...)`` (pc/model/modules/resnet_block.py:44-60,
  pc/model/res16unet.py:262-266) -- lowers to the same fused ops: the tracer folds the add / ReLU into the
  BatchNorm that produced the tensor."""

  def __init__(self, tracer, tid, channels, level):
    self.tracer, self.id, self.channels, self.level = tracer, tid, channels, level
    self.coords_man = self.coords_key = None

  @property
  def F(self):
    return SymFeatures(self, "raw")

  def __add__(self, other):
    return self.tracer.add(self, other)

  __iadd__ = __add__


class SymFeatures:
  """`.F` of a SymTensor.  The only arithmetic a lowered model may do on it is the reference's L2 row
  normalisation ``F / torch.norm(F, p=2, dim=1, keepdim=True)``; anything else cannot be expressed as a libpcmi op."""

  def __init__(self, sym, kind):
    self.sym, self.kind = sym, kind

  @classmethod
  def __torch_function__(cls, func, types, args=(), kwargs=None):
    kwargs = kwargs or {}
    if func is torch.norm and len(args) == 1 and isinstance(args[0], SymFeatures) and args[0].kind == "raw" and \
        kwargs.get("p", 2) == 2 and kwargs.get("dim") == 1 and kwargs.get("keepdim") is True:
      return SymFeatures(args[0].sym, "rownorm")
    raise NotImplementedError("%s on the features of a symbolic tensor cannot be lowered to the native engine" %
                              getattr(func, "__name__", func))

  def __truediv__(self, other):
    if self.kind == "raw" and isinstance(other, SymFeatures) and other.kind == "rownorm" and other.sym is self.sym:
      return SymFeatures(self.sym, "l2normalized")
    raise NotImplementedError("only F / torch.norm(F, p=2, dim=1, keepdim=True) can be lowered")


class SparseTensor:
  """ME.SparseTensor: a feature matrix plus (coords_key, coords_manager)
  (pc/lib/ddp_trainer.py:290-297,392-398; pc/model/res16unet.py:262-266).

  As in the reference a tensor may be built from CPU tensors and moved with ``.to(device)``;
  the coordinate hash is created on the device at that point."""

  def __new__(cls, feats=None, *args, **kwargs):
    if isinstance(feats, SymFeatures):  # model being lowered: SparseTensor(F / ||F||, coords_key=..., coords_manager=...)
      if feats.kind != "l2normalized":
        raise NotImplementedError("a SparseTensor built from raw symbolic features cannot be lowered")
      return feats.sym.tracer.l2norm(feats.sym)  # a SymTensor: __init__ is skipped
    return super().__new__(cls)

  def __init__(self, feats, coords=None, coords_key=None, coords_manager=None, force_creation=False,
               allow_duplicate_coords=False, tensor_stride=1):
    assert torch.is_tensor(feats) and feats.dim() == 2, "features must be a 2-D tensor"
    self._F = feats
    self._cpu_coords = None
    if coords_manager is not None:
      assert coords_key is not None
      self.coords_man, self.coords_key = coords_manager, coords_key
    else:
      assert coords is not None, "either coords or (coords_key, coords_manager) is required"
      if isinstance(coords, np.ndarray):
        coords = torch.from_numpy(coords)
      assert coords.shape[0] == feats.shape[0], "one coordinate row per feature row"
      coords = coords.int()
      if feats.is_cuda:
        self.coords_man = CoordsManager(coords.to(feats.device))
        self.coords_key = self.coords_man.key(0)
      else:
        self._cpu_coords = coords
        self.coords_man, self.coords_key = None, None

  def to(self, device, defer_check=False):
    """defer_check: see CoordsManager (the caller plans the network right behind this call)."""
    device = torch.device(device if not isinstance(device, int) else "cuda:%d" % device)
    if self.coords_man is None:
      # Uploads and the coordinate hash run on the side ("plan") stream: they then never queue behind the
      # compute stream's backlog, so the next batch can be prepared while the previous step still runs.
      plan, cur = handle_pool.plan_stream(device), torch.cuda.current_stream(device)
      with torch.cuda.device(device), torch.cuda.stream(plan):
        c = self._cpu_coords.to(device, non_blocking=True)
        f = self._F.to(device, non_blocking=True)
        cm = CoordsManager(c, defer_check=defer_check)
      ev = torch.cuda.Event()
      ev.record(plan)
      f.record_stream(cur)
      out = SparseTensor(f, coords_key=cm.key(0), coords_manager=cm)
      if getattr(_tls, "defer", False):
        out.upload_event = ev  # see deferred_upload_sync
      else:
        cur.wait_event(ev)  # stream-ordered: the features are on the device before the compute stream reads them
      return out
    return SparseTensor(self._F.to(device), coords_key=self.coords_key, coords_manager=self.coords_man)

  def wait_upload(self):
    """Makes the current stream wait for an upload deferred by deferred_upload_sync (no-op otherwise)."""
    ev = getattr(self, "upload_event", None)
    if ev is not None:
      torch.cuda.current_stream(self._F.device).wait_event(ev)
      self.upload_event = None

  @property
  def F(self):
    return self._F

  @property
  def feats(self):
    return self._F

  @property
  def C(self):
    if self.coords_man is None:
      return self._cpu_coords
    return self.coords_man.get_coords(self.coords_key)

  @property
  def coords(self):
    return self.C

  @property
  def tensor_stride(self):
    ts = self.coords_key.tensor_stride if self.coords_key is not None else 1
    return [ts] * 3

  @property
  def D(self):
    return 3

  @property
  def device(self):
    return self._F.device

  def __len__(self):
    return self._F.shape[0]

  def __add__(self, other):
    assert isinstance(other, SparseTensor) and other.coords_key == self.coords_key
    return SparseTensor(PF.AddFunction.apply(self._F, other._F), coords_key=self.coords_key,
                        coords_manager=self.coords_man)

  def __iadd__(self, other):  # `out += residual`, pc/model/modules/resnet_block.py:57
    return self.__add__(other)

  def __repr__(self):
    return "SparseTensor(F=%s, key=%s)" % (tuple(self._F.shape), self.coords_key)


class MinkowskiNetwork(nn.Module):
  """ME.MinkowskiNetwork: nn.Module that records the spatial dimension (pc/model/resnet.py:15-22)."""

  def __init__(self, D):
    super().__init__()
    self.D = D


class _ConvBase(nn.Module):
  transpose = False

  def __init__(self, in_channels, out_channels, kernel_size=-1, stride=1, dilation=1, has_bias=False,
               kernel_generator=None, dimension=None):
    super().__init__()
    assert dimension == 3, "only D=3 is on the hot path"
    if kernel_generator is None:
      kernel_generator = KernelGenerator(kernel_size, stride, dilation, dimension=dimension)
    self.kernel_generator = kernel_generator
    self.in_channels, self.out_channels = in_channels, out_channels
    self.kernel_size, self.stride = kernel_generator.kernel_size, kernel_generator.stride
    self.dimension, self.has_bias = dimension, has_bias
    K = kernel_generator.kernel_volume
    self.kernel_volume = K
    # 1x1 kernels are stored 2-D ("use_mm")
    self.kernel = nn.Parameter(torch.empty((in_channels, out_channels) if K == 1 else (K, in_channels, out_channels)))
    self.bias = nn.Parameter(torch.empty(1, out_channels)) if has_bias else None
    self.reset_parameters()

  def reset_parameters(self):
    n = (self.out_channels if self.transpose else self.in_channels) * self.kernel_volume
    stdv = 1.0 / math.sqrt(n)
    with torch.no_grad():
      self.kernel.uniform_(-stdv, stdv)
      if self.bias is not None:
        self.bias.uniform_(-stdv, stdv)

  def forward(self, x):
    if isinstance(x, SymTensor):
      return x.tracer.conv(self, x)
    assert isinstance(x, SparseTensor)
    cm, in_key = x.coords_man, x.coords_key
    assert cm is not None, "move the SparseTensor to the GPU first (.to(device))"
    if self.kernel_volume == 1 and self.stride == 1:
      out = PF.SparseConvFunction.apply(x.F, self.kernel, self.bias, None, False, x.F.shape[0], cm)
      self.last_work = (x.F.shape[0], x.F.shape[0], x.F.shape[0], 1)  # (pairs, n_in, n_out, K)
      return SparseTensor(out, coords_key=in_key, coords_manager=cm)
    region = self.kernel_generator.region_code
    if self.transpose:
      out_key = cm.key_at_stride(in_key.tensor_stride // self.stride)
      kmap = cm.kernel_map(out_key, in_key, self.kernel_size, self.stride, region)
    elif self.stride > 1:
      out_key = cm.stride(in_key, self.stride)
      kmap = cm.kernel_map(in_key, out_key, self.kernel_size, self.stride, region)
    else:
      out_key = in_key
      kmap = cm.kernel_map(in_key, out_key, self.kernel_size, 1, region)
    n_out = kmap.n_in if self.transpose else kmap.n_out
    self.last_work = (kmap.M, x.F.shape[0], n_out, kmap.K)
    out = PF.SparseConvFunction.apply(x.F, self.kernel, self.bias, kmap, self.transpose, n_out, cm)
    return SparseTensor(out, coords_key=out_key, coords_manager=cm)

  def extra_repr(self):
    return "in=%d, out=%d, kernel_size=%d, stride=%d, region=%s" % (
        self.in_channels, self.out_channels, self.kernel_size, self.stride, self.kernel_generator.region_type.name)


class MinkowskiConvolution(_ConvBase):
  """ME.MinkowskiConvolution (pc/model/modules/common.py:130-139)."""
  transpose = False


class MinkowskiConvolutionTranspose(_ConvBase):
  """ME.MinkowskiConvolutionTranspose with generate_new_coords=False: the output lands on the
  already existing finer coordinate set (pc/model/modules/common.py:159-168; Appendix A5)."""
  transpose = True


class MinkowskiBatchNorm(nn.Module):
  """ME.MinkowskiBatchNorm: BatchNorm1d over the rows; parameters live in ``.bn`` so that
  state-dict names (``<mod>.bn.weight`` ...) and ``m.bn.weight`` (pc/model/resnet.py:95-97)
  keep working.  The arithmetic is libpcmi's; ``.bn`` is only the parameter container."""

  def __init__(self, num_features, eps=1e-5, momentum=0.1, affine=True, track_running_stats=True):
    super().__init__()
    assert affine and track_running_stats
    if num_features % 4 != 0 or num_features > 1024:  # fail at model build, not at the first launch
      raise ValueError("MinkowskiBatchNorm: libpcmi's normalisation kernels need a channel count that is a multiple "
                       "of 4 and <= 1024 (got %d)" % num_features)
    self.bn = nn.BatchNorm1d(num_features, eps=eps, momentum=momentum)
    # num_batches_tracked only matters for momentum=None; count on the host and fold it into the
    # buffer when a state_dict is taken instead of launching a 1-element kernel per forward
    self._untracked = 0
    self.register_state_dict_pre_hook(MinkowskiBatchNorm._flush_tracked)

  @staticmethod
  def _flush_tracked(module, prefix, keep_vars):
    if module._untracked:
      with torch.no_grad():
        module.bn.num_batches_tracked += module._untracked
      module._untracked = 0

  def forward(self, x, residual=None, relu=False):
    """``residual`` / ``relu`` select the fused epilogue (the unfused call is forward(x))."""
    if isinstance(x, SymTensor):
      return x.tracer.bn(self, x, residual, relu)
    bn = self.bn
    res = residual.F if residual is not None else None
    if self.training:
      y = PF.BatchNormFunction.apply(x.F, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.momentum, bn.eps,
                                     res, relu)
      self._untracked += 1
    else:
      y = PF.batch_norm_eval(x.F, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps, res, relu)
    return SparseTensor(y, coords_key=x.coords_key, coords_manager=x.coords_man)


class MinkowskiReLU(nn.Module):

  def __init__(self, inplace=False):
    super().__init__()

  def forward(self, x):
    if isinstance(x, SymTensor):
      return x.tracer.relu(x)
    return SparseTensor(PF.ReLUFunction.apply(x.F), coords_key=x.coords_key, coords_manager=x.coords_man)


def cat(*tensors):
  """MinkowskiOps.cat: channel concat of tensors on one coordinate set
  (pc/model/res16unet.py:235,242,249,256).  The copy is a torch.cat (memory plumbing)."""
  if isinstance(tensors[0], SymTensor):
    return tensors[0].tracer.cat(tensors)
  key = tensors[0].coords_key
  for t in tensors:
    assert t.coords_key == key, "cat: tensors must share coords_key"
  return SparseTensor(torch.cat([t.F for t in tensors], dim=1), coords_key=key, coords_manager=tensors[0].coords_man)


def l2_normalize(x):
  """F / ||F||_2 per row on the same coordinates (pc/model/res16unet.py:262-266)."""
  if isinstance(x, SymTensor):
    return x.tracer.l2norm(x)
  return SparseTensor(PF.L2NormalizeFunction.apply(x.F), coords_key=x.coords_key, coords_manager=x.coords_man)


def _sparse_quantize(coords, feats=None, labels=None, return_index=False, quantization_size=None):
  """ME.utils.sparse_quantize (pc/lib/ddp_data_loaders.py:228-229): floor -> int32 ->
  ascending first-occurrence indices of the distinct voxels.  Host-side (loader)."""
  c = coords.numpy() if torch.is_tensor(coords) else np.asarray(coords)
  if quantization_size is not None:
    c = c / quantization_size
  q = np.floor(c).astype(np.int64)
  q0 = q - q.min(0)
  m = q0.max(0) + 1
  key = q0[:, 0]
  for d in range(1, q.shape[1]):
    key = key * m[d] + q0[:, d]
  _, first = np.unique(key, return_index=True)
  first = np.sort(first)
  if return_index:
    return first
  out = [q[first].astype(np.int32)]
  if feats is not None:
    out.append(feats[first])
  if labels is not None:
    out.append(labels[first])
  return out[0] if len(out) == 1 else tuple(out)


def _not_on_hot_path(name):

  class _Missing(nn.Module):

    def __init__(self, *a, **k):
      raise NotImplementedError("%s is not used by the PointContrast pre-training path and is not provided" % name)

  _Missing.__name__ = name
  return _Missing


MinkowskiInstanceNorm = _not_on_hot_path("MinkowskiInstanceNorm")
MinkowskiGlobalPooling = _not_on_hot_path("MinkowskiGlobalPooling")
MinkowskiAvgPooling = _not_on_hot_path("MinkowskiAvgPooling")
MinkowskiAvgUnpooling = _not_on_hot_path("MinkowskiAvgUnpooling")
MinkowskiSumPooling = _not_on_hot_path("MinkowskiSumPooling")

MinkowskiOps = types.ModuleType("MinkowskiEngine.MinkowskiOps")
MinkowskiOps.cat = cat
utils = types.SimpleNamespace(sparse_quantize=_sparse_quantize)


def install():
  """Registers this module as ``MinkowskiEngine`` (and its ``MinkowskiOps`` submodule), so that code written
  against ME 0.4.3 -- ``import MinkowskiEngine as ME``, ``from MinkowskiEngine import SparseTensor``,
  ``import MinkowskiEngine.MinkowskiOps as me`` (pc/model/res16unet.py:10-12, pc/lib/ddp_trainer.py:26) -- runs on
  libpcmi without touching its import lines.  Call before importing such code.  Refuses to shadow a real ME."""
  me = sys.modules[__name__]
  cur = sys.modules.get("MinkowskiEngine")
  if cur is not None and cur is not me and hasattr(cur, "MinkowskiEngineBackend"):
    raise RuntimeError("a real MinkowskiEngine is already imported")
  sys.modules["MinkowskiEngine"] = me
  sys.modules["MinkowskiEngine.MinkowskiOps"] = MinkowskiOps
  return me
