"""Native network executor: lowers a model built from the pointcontrast_amd.minkowski modules into
a libpcmi network program (include/pcmi.h, "Network executor") and runs a whole forward or
backward as ONE C call.

The autograd path (functional.py) costs one Python autograd node + several ctypes calls per layer
-- ~2000 launches and ~90 ms of host time per 4-pair iteration of Res16UNet34C, twice the GPU time.
Here the module tree is traced once with symbolic tensors; per iteration the host does two
pcmi_net_forward and two pcmi_net_backward calls plus the loss.  Same kernels, same arithmetic.
"""
import ctypes as C

import torch

from . import minkowski as ME
from ._lib import lib, check, NetOp, NetTensor, READY_FN
from .runtime import ptr, cur_stream, require_cuda

OP_CONV, OP_BN, OP_L2NORM = 0, 1, 2


class _Tracer:

  def __init__(self, offset_of):
    self.offset_of = offset_of  # id(parameter) -> offset in the flat buffer
    self.tensors, self.ops, self.bn_modules = [], [], []
    self.consumed = set()  # tensor ids some op already reads: such a tensor can no longer absorb an add / ReLU
    self.bn_op = {}        # tensor id -> the (still open) BN op dict that produced it

  def new(self, channels, level):
    self.tensors.append(dict(level=level, channels=channels, parent=-1, col_off=0))
    return ME.SymTensor(self, len(self.tensors) - 1, channels, level)

  def _off(self, p):
    return self.offset_of[id(p)]

  def _use(self, *ts):
    for t in ts:
      if t is not None:
        self.consumed.add(t.id)

  def conv(self, mod, x):
    assert x.channels == mod.in_channels, "conv input width %d != %d" % (x.channels, mod.in_channels)
    self._use(x)
    if mod.kernel_volume == 1:
      level = x.level
    elif mod.transpose:
      level = x.level - 1
    else:
      level = x.level + (1 if mod.stride == 2 else 0)
    out = self.new(mod.out_channels, level)
    self.ops.append(dict(type=OP_CONV, in_=x.id, in2=-1, out=out.id, cin=mod.in_channels, cout=mod.out_channels,
                         kernel_size=mod.kernel_size, stride=mod.stride, region=mod.kernel_generator.region_code,
                         transpose=int(mod.transpose), relu=0, has_bias=int(mod.bias is not None),
                         w_off=self._off(mod.kernel), b_off=self._off(mod.bias) if mod.bias is not None else 0))
    return out

  def bn(self, mod, x, residual, relu):
    bn = mod.bn
    self._use(x, residual)
    out = self.new(x.channels, x.level)
    self.ops.append(dict(type=OP_BN, in_=x.id, in2=residual.id if residual is not None else -1, out=out.id,
                         cin=x.channels, cout=x.channels, relu=int(bool(relu)), w_off=self._off(bn.weight),
                         b_off=self._off(bn.bias), running_mean=bn.running_mean.data_ptr(),
                         running_var=bn.running_var.data_ptr(), momentum=float(bn.momentum), eps=float(bn.eps)))
    self.bn_modules.append(mod)
    self.bn_op[out.id] = self.ops[-1]
    return out

  # -- the reference's unfused spelling (pc/model/modules/resnet_block.py:44-60): folded into the producing BN ----
  def _open_bn(self, x, what):
    op = self.bn_op.get(x.id)
    if op is None or x.id in self.consumed or op["relu"]:
      raise NotImplementedError("%s of a tensor that is not the fresh, unconsumed output of a BatchNorm cannot be "
                                "lowered to the native engine (its ops are conv, BN(+residual)(+ReLU), L2-norm)" % what)
    return op

  def add(self, a, b):
    """``out += residual``: the residual becomes the second input of the BatchNorm that produced ``out``.  The BN op
    moves to the end of the program (nothing consumed its output yet), i.e. behind the residual's producer."""
    try:
      op, res = self._open_bn(a, "an add"), b
    except NotImplementedError:
      op, res = self._open_bn(b, "an add"), a
    if op["in2"] >= 0:
      raise NotImplementedError("a second add into the same BatchNorm output cannot be lowered")
    assert res.channels == op["cout"] and res.level == self.tensors[op["out"]]["level"], "add: shapes differ"
    self._use(res)
    op["in2"] = res.id
    self.ops.remove(op)
    self.ops.append(op)
    return ME.SymTensor(self, op["out"], op["cout"], res.level)

  def relu(self, x):
    op = self._open_bn(x, "a ReLU")
    op["relu"] = 1
    return ME.SymTensor(self, x.id, x.channels, x.level)

  def cat(self, ts):
    self._use(*ts)
    level = ts[0].level
    parent = self.new(sum(t.channels for t in ts), level)
    col = 0
    for t in ts:
      rec = self.tensors[t.id]
      assert t.level == level and rec["parent"] == -1, "a tensor can be concatenated once, on its own level"
      rec["parent"], rec["col_off"] = parent.id, col
      col += t.channels
    return parent

  def l2norm(self, x):
    self._use(x)
    out = self.new(x.channels, x.level)
    self.ops.append(dict(type=OP_L2NORM, in_=x.id, in2=-1, out=out.id, cin=x.channels, cout=x.channels))
    return out


def lower_model(model, flat, in_channels=3):
  """Traces `model` with a symbolic tensor; returns the program (ctypes arrays + metadata)."""
  tr = _Tracer({id(p): off for p, off in zip(flat.params, flat.offsets)})
  x = tr.new(in_channels, 0)
  y = model(x)
  assert isinstance(y, ME.SymTensor), "the model's forward did not stay symbolic"
  T = (NetTensor * len(tr.tensors))()
  for i, t in enumerate(tr.tensors):
    T[i] = NetTensor(t["level"], t["channels"], t["parent"], t["col_off"])
  O = (NetOp * len(tr.ops))()
  for i, o in enumerate(tr.ops):
    O[i] = NetOp(o["type"], o["in_"], o["in2"], o["out"], o.get("cin", 0), o.get("cout", 0), o.get("kernel_size", 0),
                 o.get("stride", 1), o.get("region", 0), o.get("transpose", 0), o.get("relu", 0), o.get("has_bias", 0),
                 o.get("w_off", 0), o.get("b_off", 0), o.get("running_mean", None), o.get("running_var", None),
                 o.get("momentum", 0.0), o.get("eps", 0.0))
  return dict(T=T, O=O, tensors=tr.tensors, ops=tr.ops, input=x.id, output=y.id, out_channels=y.channels,
              n_down=max(t["level"] for t in tr.tensors), bn_modules=tr.bn_modules)


def canonical_program(prog):
  """The program with tensors renumbered in order of first use (input first), as plain tuples: two models lower to
  the same network iff these are equal (tests: the reference's unmodified model source vs. this package's)."""
  tensors, ops = prog["tensors"], prog["ops"]
  order = {}

  def num(t):
    if t < 0:
      return -1
    if t not in order:
      order[t] = len(order)
      if tensors[t]["parent"] >= 0:
        num(tensors[t]["parent"])
    return order[t]

  num(prog["input"])
  keys = ("type", "cin", "cout", "kernel_size", "stride", "region", "transpose", "relu", "has_bias", "w_off", "b_off",
          "momentum", "eps")
  cops = []
  for o in ops:
    cops.append((num(o["in_"]), num(o["in2"]), num(o["out"])) + tuple(o.get(k, 0) for k in keys))
  inv = sorted(order, key=order.get)
  ctens = [(tensors[t]["level"], tensors[t]["channels"], num(tensors[t]["parent"]), tensors[t]["col_off"]) for t in inv]
  return dict(tensors=ctens, ops=cops, input=order[prog["input"]], output=order[prog["output"]])


def create_net(prog, n_passes=2):
  h = C.c_void_p()
  check(lib.pcmi_net_create(prog["T"], len(prog["tensors"]), prog["O"], len(prog["ops"]), prog["input"], prog["output"],
                            n_passes, C.byref(h)))
  return h


class NativeEngine:
  """model: a network of ME modules whose forward maps one SparseTensor to one SparseTensor
  (Res16UNet family).  flat: lib.distributed.FlatParameters of the same model."""

  def __init__(self, model, flat, in_channels=3, n_passes=2):
    require_cuda(flat.w, "NativeEngine")
    self.model, self.flat, self.n_passes = model, flat, n_passes
    prog = lower_model(model, flat, in_channels)
    self.in_channels, self.out_channels, self.n_down = in_channels, prog["out_channels"], prog["n_down"]
    self._bn_modules = prog["bn_modules"]
    self.n_ops, self.n_tensors = len(prog["ops"]), len(prog["tensors"])
    self._tensors = prog["tensors"]  # {level, channels, parent, col_off} per tensor id
    self._ops = prog["ops"]  # the lowered program (dicts): which op writes which tensor (activation / relu_masks)
    self._h = create_net(prog, n_passes)
    self._held = [None] * n_passes
    self._pair_stream = None
    self.pair_marks = None

  def __del__(self):
    try:
      if getattr(self, "_h", None):
        torch.cuda.synchronize()
        lib.pcmi_net_destroy(self._h)
        self._h = None
    except Exception:
      pass

  def forward(self, pass_id, st, training=True, defer_running_stats=False):
    """st: ME.SparseTensor on the device.  Returns the output features [N, out_channels].
    Enqueued on torch's current stream.  defer_running_stats: see apply_running_stats."""
    x = st.F
    require_cuda(x, "NativeEngine.forward")
    x = x if (x.stride(1) == 1 and x.dtype == torch.float32) else x.float().contiguous()
    cm = st.coords_man
    cm.plan_unet(self.n_down)
    n = x.shape[0]
    out = torch.empty((n, self.out_channels), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
      mode = (1 if training else 0) | (2 if (training and defer_running_stats) else 0)  # PCMI_NET_DEFER_RUNNING_STATS
      check(lib.pcmi_net_forward(self._h, pass_id, cm._h, ptr(x), x.stride(0), n, ptr(self.flat.w), mode, ptr(out),
                                 self.out_channels, cur_stream(x.device)))
    if training:
      calls = 2 if getattr(cm, "n_first", None) not in (None, 0, n) else 1  # a two-segment batch is two forward calls
      for m in self._bn_modules:
        m._untracked += calls
      self._held[pass_id] = (st, x, out)  # coordinates / input / output stay alive until backward
    return out

  def apply_running_stats(self, pass_id):
    """Applies the BatchNorm running-estimate updates a forward(..., defer_running_stats=True) of this pass left
    pending (on the current stream, which must be ordered after that forward)."""
    with torch.cuda.device(self.flat.w.device):
      check(lib.pcmi_net_apply_running_stats(self._h, pass_id, cur_stream(self.flat.w.device)))

  def forward_pair(self, st0, st1):
    """The two forwards of a training iteration (ddp_trainer.py:404-407 of the reference runs them back to back):
    pass 1 is enqueued on a side stream, pass 0 on the current one, so that the small levels of one pass fill the
    CUs the other leaves idle.  BatchNorm running estimates are still updated in the order pass 0, pass 1."""
    dev = self.flat.w.device
    cur = torch.cuda.current_stream(dev)
    if self._pair_stream is None:
      self._pair_stream = torch.cuda.Stream(device=dev)
    side = self._pair_stream
    side.wait_stream(cur)  # inputs / parameters were produced on the current stream
    marks = self.pair_marks  # None, or a list: timing events of the two forwards (bench: misc.gpu_profile)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)] if marks is not None else None
    with torch.cuda.stream(side):
      if ev:
        ev[0].record(side)
      f1 = self.forward(1, st1, defer_running_stats=True)
      if ev:
        ev[1].record(side)
    if ev:
      ev[2].record(cur)
    f0 = self.forward(0, st0)
    if ev:
      ev[3].record(cur)
      marks.append(ev)
    cur.wait_stream(side)
    f1.record_stream(cur)
    self.apply_running_stats(1)
    return f0, f1

  def pair_marks_ms(self, skip=0):
    """Mean stream times of the recorded forward pairs (after a synchronize): pass 1 on its stream, pass 0 on the
    current one, and how far pass 0's first kernel trails pass 1's."""
    m = (self.pair_marks or [])[skip:]
    if not m:
      return {}
    avg = lambda f: round(sum(f(e) for e in m) / len(m), 3)
    return {"pass1": avg(lambda e: e[0].elapsed_time(e[1])), "pass0": avg(lambda e: e[2].elapsed_time(e[3])),
            "pass0_starts_after_pass1_start": avg(lambda e: e[0].elapsed_time(e[2])),
            "pass0_ends_after_pass1_end": avg(lambda e: e[1].elapsed_time(e[3]))}

  def backward(self, pass_id, d_out, reducer=None):
    """Accumulates parameter gradients of pass `pass_id` into flat.g.  With a GradReducer the
    buckets are all-reduced (on its side stream) as soon as this pass completes them -- pass it
    only on the LAST backward of the iteration."""
    assert self._held[pass_id] is not None, "forward(training=True) first"
    d = d_out if (d_out.stride(1) == 1 and d_out.stride(0) % 4 == 0) else d_out.contiguous()
    errors = []
    cb, lo_arr, nb = self._ready_args(reducer, self._wait_bucket, errors)
    with torch.cuda.device(d.device):
      rc = lib.pcmi_net_backward(self._h, pass_id, ptr(d), d.stride(0), ptr(self.flat.w), ptr(self.flat.g), lo_arr, nb, cb,
                                 None, cur_stream(d.device))
    self._held[pass_id] = None
    if (errors or rc) and reducer is not None and (reducer.active or getattr(reducer, "after_bucket", None) is not None):
      # the step is lost either way; leave the reducer in a state the NEXT step can start from
      reducer.abort()
    if errors:  # raised inside a bucket callback: ctypes would have printed and dropped it
      raise errors[0]
    check(rc)

  def _wait_bucket(self, stream):
    """Inside a bucket-ready callback: `stream` (the reducer's communication stream) waits for the chain AND the
    executor's weight-gradient stream up to the bucket -- the backward chain itself does not join them there."""
    check(lib.pcmi_net_stream_wait_bucket(self._h, C.c_void_p(stream.cuda_stream)))

  @staticmethod
  def _ready_args(reducer, wait_bucket=None, errors=None):
    """errors: a list that receives what a bucket callback raised.  The callback runs inside a ctypes trampoline, which
    prints and DROPS a Python exception: the bucket would stay un-reduced on this rank only -- silent divergence between
    the ranks.  The caller re-raises the first entry once pcmi_net_backward has returned."""
    cb, lo_arr, nb = READY_FN(), None, 0
    if reducer is not None and (reducer.active or (getattr(reducer, "after_bucket", None) is not None and reducer.cuda)):
      order = sorted(range(len(reducer.buckets)), key=lambda b: reducer.buckets[b][0])
      lo_arr = (C.c_int64 * len(order))(*[reducer.buckets[b][0] for b in order])
      nb = len(order)

      def ready(_ctx, q):
        try:
          if wait_bucket is None:  # (tests drive the callback with CPU stand-ins for the executor)
            reducer._launch(order[q])
          else:
            reducer._launch(order[q], order_behind=wait_bucket)
        except BaseException as e:  # noqa: B902  (must not escape into the C caller)
          if errors is None:
            raise
          errors.append(e)

      cb = READY_FN(ready)
    return cb, lo_arr, nb

  def time_ops(self, ops, n_sets=8):
    """Timing events around the convolution launches of program ops `ops` inside the next `n_sets` passes (forward and
    backward-data); [] stops.  See include/pcmi.h (pcmi_net_time_ops)."""
    arr = (C.c_int * max(len(ops), 1))(*ops)
    self._timed = (list(ops), n_sets)
    check(lib.pcmi_net_time_ops(self._h, arr, len(ops), n_sets))

  def time_all(self, n_sets=8):
    """Timing events around EVERY op of the program inside the next `n_sets` passes (grouped weight gradients stay grouped
    and are timed as launches: timed_groups_ms); 0 stops.  See include/pcmi.h (pcmi_net_time_all)."""
    self._timed = (list(range(self.n_ops)) if n_sets else [], n_sets)
    check(lib.pcmi_net_time_all(self._h, n_sets))

  def timed_groups_ms(self, n_sets=None):
    """[[ms of every grouped weight-gradient launch] per recorded set] after time_all."""
    _, sets = getattr(self, "_timed", ([], 0))
    out = []
    for s_ in range(min(n_sets or sets, sets)):
      ms, n = (C.c_float * 16)(), C.c_int()
      check(lib.pcmi_net_timed_groups_ms(self._h, s_, ms, 16, C.byref(n)))
      out.append([ms[i] for i in range(n.value)])
    return out

  def timed_launches(self, n_sets=None):
    """[(fwd, bwd, wgrad launches per op, [launches per grouped weight-gradient flush])] of the recorded sets."""
    ops, sets = getattr(self, "_timed", ([], 0))
    out = []
    for s_ in range(min(n_sets or sets, sets)):
      f, b, w, g = (C.c_int * len(ops))(), (C.c_int * len(ops))(), (C.c_int * len(ops))(), (C.c_int * 16)()
      check(lib.pcmi_net_timed_launches(self._h, s_, f, b, w, len(ops), g, 16))
      out.append((list(f), list(b), list(w), [x for x in g if x]))
    return out

  def timed_ms(self, n_sets=None):
    """[(fwd_ms, bwd_data_ms, wgrad_ms) per op] of the first `n_sets` recorded sets (waits for them; -1 = not recorded)."""
    ops, sets = getattr(self, "_timed", ([], 0))
    out = []
    for s_ in range(min(n_sets or sets, sets)):
      f, b, w = (C.c_float * len(ops))(), (C.c_float * len(ops))(), (C.c_float * len(ops))()
      check(lib.pcmi_net_timed_ms(self._h, s_, f, b, w, len(ops)))
      out.append((list(f), list(b), list(w)))
    return out

  def activation(self, pass_id, tensor_id):
    """Copy of activation tensor `tensor_id` of the last forward of pass `pass_id` ([rows, channels], current stream)."""
    n, c = C.c_int64(), C.c_int()
    dev = self.flat.w.device
    with torch.cuda.device(dev):
      check(lib.pcmi_net_export_tensor(self._h, pass_id, tensor_id, C.byref(n), C.byref(c), None, 0, cur_stream(dev)))
      out = torch.empty((n.value, c.value), dtype=torch.float32, device=dev)
      check(lib.pcmi_net_export_tensor(self._h, pass_id, tensor_id, None, None, ptr(out), c.value, cur_stream(dev)))
    return out

  def relu_masks(self, pass_id):
    """(y > 0) of every fused BatchNorm(+residual)+ReLU output of the last forward, in program order -- the order of
    the ReLU calls of the model's forward (a block's norm1, then its `out += residual; relu`).  Tests hand these to the
    oracle (oracle.model_ref.relu_masks(apply=...)) so that both differentiate the same piecewise-linear function."""
    return [self.activation(pass_id, o["out"]) > 0 for o in self._ops if o["type"] == OP_BN and o.get("relu")]

  def memory_bytes(self):
    b = C.c_size_t()
    check(lib.pcmi_net_memory_bytes(self._h, C.byref(b)))
    return b.value
