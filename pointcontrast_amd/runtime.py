"""Device plumbing shared by the op wrappers: raw pointers, streams, the scratch
workspace and the pooled coordinate-manager handles.  PyTorch is used for device
memory (caching allocator) and streams only."""
import ctypes as C
import threading

import torch

from . import _lib
from ._lib import lib, check


def ptr(t):
  """Device pointer of a tensor (None -> NULL)."""
  return None if t is None else C.c_void_p(t.data_ptr())


def cur_stream(device=None):
  return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def require_cuda(t, who):
  if not t.is_cuda:
    raise _lib.PcmiError(
        "%s: tensor is on %s; the pointcontrast_amd ops run only as HIP kernels on a gfx950 device "
        "(there is no CPU path)" % (who, t.device))


class _Workspace:
  """One growable scratch buffer per device, used in stream order by the ops of
  the compute stream (ops never allocate device memory themselves)."""

  def __init__(self):
    self._buf = {}

  def get(self, nbytes, device):
    key = torch.device(device).index
    b = self._buf.get(key)
    if b is None or b.numel() < nbytes:
      new_size = max(int(nbytes * 1.25) + 4096, 1 << 24)
      # the old buffer may still be in use by enqueued kernels: the caching allocator keeps it
      # alive until the stream passes the point of release
      b = torch.empty(new_size, dtype=torch.uint8, device=device)
      self._buf[key] = b
    return b


workspace = _Workspace()


def ws_args(nbytes, device):
  b = workspace.get(nbytes, device)
  return C.c_void_p(b.data_ptr()), C.c_size_t(b.numel())


# ---------------------------------------------------------------------------------------------
# coordinate-manager handle pool
# ---------------------------------------------------------------------------------------------
class _HandlePool:
  """pcmi_coords handles are reset and reused across iterations so that their device arenas are
  allocated once.  A handle returned to the pool carries an event recorded on the compute
  stream; the next user makes its plan stream wait on it before the arena is overwritten."""

  def __init__(self):
    self._free = {}
    self._lock = threading.Lock()
    self._plan_streams = {}

  def plan_stream(self, device):
    key = torch.device(device).index
    s = self._plan_streams.get(key)
    if s is None:
      # highest priority: the planning calls are host-synchronous (they read back a few counts), so the helper thread
      # -- and through it the enqueueing thread -- waits for these small kernels; they must not queue behind the
      # compute / weight-gradient streams' workgroups
      s = torch.cuda.Stream(device=device, priority=-1)
      self._plan_streams[key] = s
    return s

  MIN_FREE = 4  # reuse the OLDEST released handle, and only once a few are idle, so that the
  # release event of the handle we pick is (almost always) already complete and the
  # host never stalls on the previous iteration

  def acquire(self, device):
    key = torch.device(device).index
    with self._lock:
      lst = self._free.setdefault(key, [])
      if len(lst) >= self.MIN_FREE:
        h, ev = lst.pop(0)
      else:
        h, ev = None, None
    if h is None:
      with torch.cuda.device(device):
        hp = C.c_void_p()
        check(lib.pcmi_coords_create(3, C.byref(hp)))
      return hp, None
    return h, ev

  def release(self, device, handle):
    key = torch.device(device).index
    ev = torch.cuda.Event()
    ev.record(torch.cuda.current_stream(device))
    with self._lock:
      self._free.setdefault(key, []).append((handle, ev))


handle_pool = _HandlePool()
