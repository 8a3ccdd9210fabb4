"""Checkpoint / weight-file interoperability with the reference.

Layout (pc/lib/ddp_trainer.py:151-169): ``{curr_iter, state_dict, optimizer, scheduler, config}`` saved as
``weights/<name>.pth`` with a ``weights/weights.pth`` symlink; state-dict names are the reference's
(``<conv>.kernel`` [K, Cin, Cout] -- 1x1 kernels 2-D [Cin, Cout] --, ``<conv>.bias`` [1, Cout],
``<bn>.bn.{weight,bias,running_mean,running_var,num_batches_tracked}``), which is what the released
``nce.pth`` / ``hardest_contrastive.pth`` files and the downstream loaders (downstream/semseg/lib/utils.py:19-43) use.

The one thing a weight file does not record is WHICH OFFSET each of the 27 slices of a 3^3 kernel belongs to.  The
block convolutions are built with ConvType.SPATIAL_HYPERCUBE_TEMPORAL_HYPERCROSS -> ME.RegionType.HYBRID with three
cube axes (pc/model/modules/common.py:59,108-114).  libpcmi enumerates such a region centre-first (SURVEY.md
Appendix A7, recalled from ME 0.4.3's Python kernel generator); MinkowskiEngine itself is not available here, so that
recollection cannot be pinned.  ``kernel_order`` is therefore a documented switch: a file whose 27-slice block kernels
are stored in HYPERCUBE order (x fastest, -1..+1 per axis) is mapped with ``kernel_order="hypercube"``; the default
``"hybrid"`` takes the slices as they are.  Everything else (names, shapes, 1x1 and 2^3 kernels, BatchNorm buffers)
needs no conversion.
"""
import logging

import numpy as np
import torch

from .. import minkowski as ME

_PREFIXES = ("module.", "encoder.")


def strip_prefixes(weights):
  """DataParallel / encoder wrappers prefix every key (downstream/semseg/lib/utils.py:23-29)."""
  out = dict(weights)
  for pre in _PREFIXES:
    if out and next(iter(out)).startswith(pre):
      logging.info("Loading weights with %s prefix...", pre)
      out = {k.partition(pre)[2]: v for k, v in out.items()}
  return out


def load_state_with_same_shape(model, weights):
  """The tensors of `weights` whose (prefix-stripped) name and shape match the model's
  (downstream/semseg/lib/utils.py:19-43)."""
  own = model.state_dict()
  weights = strip_prefixes(weights)
  keep = {k: v for k, v in weights.items() if k in own and v.size() == own[k].size()}
  logging.info("Loading weights:" + ", ".join(keep.keys()))
  return keep


def _offsets(region_type):
  gen = ME.KernelGenerator(3, 1, 1, region_type=region_type,
                           axis_types=[ME.RegionType.HYPERCUBE] * 3 if region_type == ME.RegionType.HYBRID else None, dimension=3)
  return gen.get_kernel()[1].numpy()


def slice_permutation(src_region, dst_region):
  """perm with  kernel_dst[k] = kernel_src[perm[k]]:  slice k of the destination enumeration is the slice of the
  source enumeration that has the same offset."""
  src, dst = _offsets(src_region), _offsets(dst_region)
  lut = {tuple(o): k for k, o in enumerate(src.tolist())}
  return np.array([lut[tuple(o)] for o in dst.tolist()], dtype=np.int64)


def hybrid_kernel_names(model):
  """State-dict names of the 27-slice kernels whose region is HYBRID (the block convolutions)."""
  return [name + ".kernel" for name, m in model.named_modules()
          if isinstance(m, ME._ConvBase) and m.kernel_volume == 27 and m.kernel_generator.region_type == ME.RegionType.HYBRID]


def convert_kernel_order(model, weights, kernel_order="hybrid", inverse=False):
  """Re-orders the 27 slices of the block-conv kernels of `weights` from the file's enumeration to libpcmi's
  (``inverse=True``: the other way, for writing a file in that enumeration).  kernel_order: "hybrid" (identity) or
  "hypercube" (module docstring)."""
  if kernel_order in (None, "hybrid", "same"):
    return weights
  if kernel_order != "hypercube":
    raise ValueError("kernel_order must be 'hybrid' or 'hypercube', not %r" % (kernel_order,))
  if inverse:
    perm = slice_permutation(ME.RegionType.HYBRID, ME.RegionType.HYPERCUBE)
  else:
    perm = slice_permutation(ME.RegionType.HYPERCUBE, ME.RegionType.HYBRID)
  out = dict(weights)
  idx = torch.from_numpy(perm)
  for name in hybrid_kernel_names(model):
    if name in out and out[name].dim() == 3 and out[name].shape[0] == 27:
      out[name] = out[name].index_select(0, idx.to(out[name].device))
  return out


def load_state(model, weights, lenient_weight_loading=False, kernel_order="hybrid"):
  """pc/lib/ddp_trainer.py:54-69 (+ prefix stripping and the slice-order switch)."""
  weights = convert_kernel_order(model, strip_prefixes(weights), kernel_order)
  if lenient_weight_loading:
    own = model.state_dict()
    keep = {k: v for k, v in weights.items() if k in own and v.size() == own[k].size()}
    logging.info("Load weights:" + ", ".join(keep.keys()))
    own.update(keep)
    weights = own
  model.load_state_dict(weights, strict=True)
