"""ctypes binding of libpcmi.so (the C ABI declared in include/pcmi.h).

The library is the product: there is no Python/PyTorch fallback for any op.  If
the shared object is missing or a symbol is absent this module raises at import
time.  `import torch` must come first so that libpcmi's DT_NEEDED
libamdhip64.so.7 resolves to the HIP runtime PyTorch already loaded (one runtime
per process -> torch streams / device pointers are valid inside libpcmi).
"""
import ctypes as C
import os

import torch  # noqa: F401  (must precede the CDLL, see docstring)

_HERE = os.path.dirname(os.path.abspath(__file__))
# PCMI_LIB: another build of the SAME library (scripts/ablate_conv16.sh's A/B kernels); never a different implementation
LIB_PATH = os.path.abspath(os.environ["PCMI_LIB"]) if os.environ.get("PCMI_LIB") else os.path.join(_HERE, "libpcmi.so")

if not os.path.exists(LIB_PATH):
  raise ImportError(
      "pointcontrast_amd: %s is missing -- build it with `python -m pointcontrast_amd.build` "
      "(hipcc, gfx950).  There is no fallback path." % LIB_PATH)

lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)

MAX_K = 27
c_i64, c_i32, c_f32, c_vp, c_sz = C.c_int64, C.c_int32, C.c_float, C.c_void_p, C.c_size_t


class KMap(C.Structure):
  """struct pcmi_kmap (include/pcmi.h)."""
  _fields_ = [
      ("K", c_i32), ("kernel_size", c_i32), ("stride", c_i32), ("region", c_i32),
      ("n_in", c_i64), ("n_out", c_i64), ("M", c_i64),
      ("nbr", c_vp), ("pair_in", c_vp), ("pair_out", c_vp), ("offs", c_vp),
      ("offs_host", c_i64 * (MAX_K + 1)), ("mirror", c_i32 * MAX_K),
      ("perm", c_vp), ("nbr_perm", c_vp),
      ("tile_mask", c_vp), ("tile_pref", c_vp), ("n_tiles", c_i64),
  ]


_KP = C.POINTER(KMap)


class NetTensor(C.Structure):
  """struct pcmi_net_tensor."""
  _fields_ = [("level", c_i32), ("channels", c_i32), ("parent", c_i32), ("col_off", c_i32)]


class NetOp(C.Structure):
  """struct pcmi_net_op."""
  _fields_ = [("type", c_i32), ("in_", c_i32), ("in2", c_i32), ("out", c_i32),
              ("cin", c_i32), ("cout", c_i32), ("kernel_size", c_i32), ("stride", c_i32), ("region", c_i32),
              ("transpose", c_i32), ("relu", c_i32), ("has_bias", c_i32),
              ("w_off", c_i64), ("b_off", c_i64), ("running_mean", c_vp), ("running_var", c_vp),
              ("momentum", c_f32), ("eps", c_f32)]


READY_FN = C.CFUNCTYPE(None, c_vp, C.c_int)

# name -> (restype, argtypes); every symbol of include/pcmi.h
PROTOTYPES = {
    "pcmi_version": (C.c_int, []),
    "pcmi_last_error": (C.c_char_p, []),
    "pcmi_device_info": (C.c_int, [C.POINTER(C.c_int), C.c_char_p, C.c_int]),
    "pcmi_coords_create": (C.c_int, [C.c_int, C.POINTER(c_vp)]),
    "pcmi_coords_destroy": (C.c_int, [c_vp]),
    "pcmi_coords_reset": (C.c_int, [c_vp]),
    "pcmi_coords_insert": (C.c_int, [c_vp, c_vp, c_i64, c_vp]),
    "pcmi_coords_insert_deferred": (C.c_int, [c_vp, c_vp, c_i64, c_vp]),
    "pcmi_coords_check": (C.c_int, [c_vp, c_vp]),
    "pcmi_coords_stride": (C.c_int, [c_vp, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(c_i64), c_vp]),
    "pcmi_coords_key_at_stride": (C.c_int, [c_vp, C.c_int, C.POINTER(C.c_int)]),
    "pcmi_coords_size": (C.c_int, [c_vp, C.c_int, C.POINTER(c_i64), C.POINTER(C.c_int)]),
    "pcmi_coords_set_split": (C.c_int, [c_vp, c_i64]),
    "pcmi_coords_split": (C.c_int, [c_vp, C.c_int, C.POINTER(c_i64)]),
    "pcmi_coords_get": (C.c_int, [c_vp, C.c_int, c_vp, c_vp]),
    "pcmi_coords_plan_unet": (C.c_int, [c_vp, C.c_int, C.c_int, C.c_int, c_vp]),
    "pcmi_coords_arena_bytes": (C.c_int, [c_vp, C.POINTER(c_sz)]),
    "pcmi_kernel_offsets": (C.c_int, [C.c_int, C.c_int, C.POINTER(c_i32), C.POINTER(C.c_int)]),
    "pcmi_kmap_get": (C.c_int, [c_vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _KP, c_vp]),
    "pcmi_kmap_export": (C.c_int, [_KP, c_vp, c_vp, c_vp, c_vp]),
    "pcmi_spconv_workspace_bytes": (c_sz, [c_i64, c_i64, C.c_int, C.c_int, C.c_int, c_i64]),
    "pcmi_spconv_fwd": (C.c_int, [c_vp, c_i64, c_i64, C.c_int, c_vp, C.c_int, _KP, C.c_int, c_vp, c_vp, c_i64,
                                  c_i64, c_vp, c_sz, c_vp]),
    "pcmi_spconv_bwd_data": (C.c_int, [c_vp, c_i64, c_i64, C.c_int, c_vp, C.c_int, _KP, C.c_int, c_vp, c_i64,
                                       c_i64, c_vp, c_sz, c_vp]),
    "pcmi_spconv_bwd_weight": (C.c_int, [c_vp, c_i64, c_i64, C.c_int, c_vp, c_i64, c_i64, C.c_int, _KP, C.c_int,
                                         c_vp, c_vp, c_vp, c_sz, c_vp]),
    "pcmi_spconv_split_precision": (C.c_int, []),
    "pcmi_bn_workspace_bytes": (c_sz, [c_i64, C.c_int]),
    "pcmi_bn_fwd_train": (C.c_int, [c_vp, c_i64, c_i64, C.c_int, c_vp, c_vp, c_vp, c_vp, c_f32, c_f32, c_vp,
                                    c_i64, C.c_int, c_vp, c_i64, c_vp, c_vp, c_vp, c_sz, c_vp]),
    "pcmi_bn_fwd_eval": (C.c_int, [c_vp, c_i64, c_i64, C.c_int, c_vp, c_vp, c_vp, c_vp, c_f32, c_vp, c_i64,
                                   C.c_int, c_vp, c_i64, c_vp]),
    "pcmi_bn_bwd": (C.c_int, [c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_i64, C.c_int, c_vp, c_vp, c_vp, c_vp,
                              c_i64, c_vp, c_i64, c_vp, c_vp, c_vp, c_sz, c_vp]),
    "pcmi_relu_fwd": (C.c_int, [c_vp, c_i64, c_i64, C.c_int, c_vp, c_i64, c_vp]),
    "pcmi_relu_bwd": (C.c_int, [c_vp, c_i64, c_vp, c_i64, c_i64, C.c_int, c_vp, c_i64, c_vp]),
    "pcmi_add": (C.c_int, [c_vp, c_i64, c_vp, c_i64, c_i64, C.c_int, c_vp, c_i64, c_vp]),
    "pcmi_l2norm_fwd": (C.c_int, [c_vp, c_i64, c_i64, C.c_int, c_vp, c_i64, c_vp, c_vp]),
    "pcmi_l2norm_bwd": (C.c_int, [c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, C.c_int, c_vp, c_i64, c_vp]),
    "pcmi_gather_rows": (C.c_int, [c_vp, c_i64, c_vp, c_i64, C.c_int, c_vp, c_i64, c_vp]),
    "pcmi_scatter_add_rows": (C.c_int, [c_vp, c_i64, c_vp, c_i64, C.c_int, c_vp, c_i64, c_vp]),
    "pcmi_nce_workspace_bytes": (c_sz, [c_i64, C.c_int]),
    "pcmi_nce_fwd": (C.c_int, [c_vp, c_vp, c_i64, C.c_int, c_f32, c_vp, c_vp, c_vp, c_sz, c_vp]),
    "pcmi_nce_bwd": (C.c_int, [c_vp, c_vp, c_vp, c_i64, C.c_int, c_f32, c_vp, c_vp, c_vp, c_vp, c_sz, c_vp]),
    "pcmi_pdist_argmin": (C.c_int, [c_vp, c_i64, c_vp, c_i64, C.c_int, c_vp, c_vp, c_vp]),
    "pcmi_keyset_bytes": (c_sz, [c_i64]),
    "pcmi_keyset_build": (C.c_int, [c_vp, c_i64, c_i64, c_vp, c_sz, c_vp]),
    "pcmi_keyset_mask_absent": (C.c_int, [c_vp, c_sz, c_vp, c_vp, c_i64, c_i64, c_vp, c_vp]),
    "pcmi_hardest_workspace_bytes": (c_sz, [c_i64]),
    "pcmi_hardest_loss_fwd": (C.c_int, [c_vp, c_vp, c_i64, C.c_int, c_vp, c_vp, c_vp, c_vp, c_f32, c_f32, c_vp, c_vp,
                                        c_vp, c_sz, c_vp]),
    "pcmi_hardest_loss_bwd": (C.c_int, [c_vp, c_vp, c_i64, c_vp, c_vp, C.c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp,
                                        c_f32, c_f32, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "pcmi_voxelize_workspace_bytes": (c_sz, [c_i64]),
    "pcmi_voxelize": (C.c_int, [c_vp, c_i64, C.c_double, c_vp, c_vp, C.POINTER(c_i64), c_vp, c_sz, c_vp]),
    "pcmi_match_radius_workspace_bytes": (c_sz, [c_i64, c_i64]),
    "pcmi_match_radius": (C.c_int, [c_vp, c_i64, C.POINTER(C.c_double), c_vp, c_i64, C.c_double, c_vp, c_i64,
                                    C.POINTER(c_i64), c_vp, c_sz, c_vp]),
    "pcmi_softmax_ce_workspace_bytes": (c_sz, [c_i64]),
    "pcmi_softmax_ce_fwd": (C.c_int, [c_vp, c_i64, c_i64, C.c_int, c_vp, C.c_int, c_vp, c_vp, c_sz, c_vp]),
    "pcmi_softmax_ce_bwd": (C.c_int, [c_vp, c_i64, c_i64, C.c_int, c_vp, C.c_int, c_vp, c_vp, c_vp, c_i64, c_vp]),
    "pcmi_pair_select_workspace_bytes": (C.c_size_t, [c_i64]),
    "pcmi_pair_select": (C.c_int, [c_vp, c_i64, c_i64, c_vp, c_vp, c_i64, c_vp, c_vp, c_vp, C.c_size_t, c_vp]),
    "pcmi_pairs_scan_host": (C.c_int, [c_vp, c_i64, C.POINTER(c_i64), C.POINTER(C.c_int)]),
    "pcmi_sgd_step": (C.c_int, [c_vp, c_vp, c_vp, c_i64, c_f32, c_f32, c_f32, c_f32, c_vp]),
    "pcmi_sgd_step_dampened": (C.c_int, [c_vp, c_vp, c_vp, c_i64, c_f32, c_f32, c_f32, c_f32, c_f32, C.c_int, c_vp]),
    "pcmi_net_create": (C.c_int, [C.POINTER(NetTensor), C.c_int, C.POINTER(NetOp), C.c_int, C.c_int, C.c_int, C.c_int,
                                  C.POINTER(c_vp)]),
    "pcmi_net_destroy": (C.c_int, [c_vp]),
    "pcmi_net_forward": (C.c_int, [c_vp, C.c_int, c_vp, c_vp, c_i64, c_i64, c_vp, C.c_int, c_vp, c_i64, c_vp]),
    "pcmi_net_backward": (C.c_int, [c_vp, C.c_int, c_vp, c_i64, c_vp, c_vp, C.POINTER(c_i64), C.c_int, READY_FN, c_vp,
                                    c_vp]),
    "pcmi_net_apply_running_stats": (C.c_int, [c_vp, C.c_int, c_vp]),
    "pcmi_net_stream_wait_bucket": (C.c_int, [c_vp, c_vp]),
    "pcmi_net_export_tensor": (C.c_int, [c_vp, C.c_int, C.c_int, C.POINTER(c_i64), C.POINTER(C.c_int), c_vp, c_i64, c_vp]),
    "pcmi_net_memory_bytes": (C.c_int, [c_vp, C.POINTER(c_sz)]),
    "pcmi_net_time_ops": (C.c_int, [c_vp, C.POINTER(C.c_int), C.c_int, C.c_int]),
    "pcmi_net_timed_ms": (C.c_int, [c_vp, C.c_int, C.POINTER(c_f32), C.POINTER(c_f32), C.POINTER(c_f32), C.c_int]),
    "pcmi_net_time_all": (C.c_int, [c_vp, C.c_int]),
    "pcmi_net_timed_launches": (C.c_int, [c_vp, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_int,
                                          C.POINTER(C.c_int), C.c_int]),
    "pcmi_net_timed_groups_ms": (C.c_int, [c_vp, C.c_int, C.POINTER(c_f32), C.c_int, C.POINTER(C.c_int)]),
}

for _name, (_res, _args) in PROTOTYPES.items():
  _fn = getattr(lib, _name)  # AttributeError here == the library does not export what pcmi.h declares
  _fn.restype = _res
  _fn.argtypes = _args


class PcmiError(RuntimeError):
  pass


def check(rc):
  if rc != 0:
    raise PcmiError("libpcmi error %d: %s" % (rc, lib.pcmi_last_error().decode()))


def version():
  return lib.pcmi_version()
