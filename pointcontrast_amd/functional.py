"""torch.autograd.Function wrappers over the libpcmi C ABI.

Every forward/backward below is one or a few calls into libpcmi.so on the
current torch stream; tensors are plain fp32 row-major [rows, channels].  There is
no torch-op fallback: a CPU tensor raises.
"""
import ctypes as C

import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from ._lib import lib, check, KMap
from .runtime import ptr, cur_stream, ws_args, require_cuda


def _rows(t):
  assert t.dim() == 2 and t.stride(1) == 1, "expected a row-major [rows, channels] tensor"
  return t.shape[0], t.shape[1], t.stride(0)


def _c(t):
  """Contiguous-rows view (ld % 4 == 0, 16-byte aligned base) or a packed copy."""
  if t.stride(1) == 1 and t.stride(0) % 4 == 0 and t.data_ptr() % 16 == 0 and t.stride(0) >= t.shape[1]:
    return t
  return t.contiguous()


def _kmap_ref(kmap):
  return C.byref(kmap) if kmap is not None else None


class SparseConvFunction(Function):
  """MinkowskiConvolution(.Transpose)Function: fwd / bwd-data / bwd-weight
  (replaces MEB.Convolution{Forward,Backward}GPU; pc/model/modules/common.py:130-168)."""

  @staticmethod
  def forward(ctx, feats, kernel, bias, kmap, transpose, n_out, owner=None):
    require_cuda(feats, "sparse conv")
    feats = _c(feats)
    n_in, cin, in_ld = _rows(feats)
    cout = kernel.shape[-1]
    K = kmap.K if kmap is not None else 1
    assert kernel.is_contiguous() and kernel.numel() == K * cin * cout, "kernel shape does not match the map"
    out = torch.empty((n_out, cout), dtype=torch.float32, device=feats.device)
    M = kmap.M if kmap is not None else n_in
    ws, wsb = ws_args(lib.pcmi_spconv_workspace_bytes(n_in, n_out, cin, cout, K, M), feats.device)
    check(lib.pcmi_spconv_fwd(ptr(feats), in_ld, n_in, cin, ptr(kernel), cout, _kmap_ref(kmap), int(transpose),
                              ptr(bias), ptr(out), cout, n_out, ws, wsb, cur_stream(feats.device)))
    ctx.save_for_backward(feats, kernel)
    ctx.kmap, ctx.transpose, ctx.has_bias = kmap, int(transpose), bias is not None
    ctx.owner = owner  # keeps the coordinate manager (and the arena behind kmap) alive until backward
    return out

  @staticmethod
  @once_differentiable
  def backward(ctx, gout):
    feats, kernel = ctx.saved_tensors
    kmap = ctx.kmap
    gout = _c(gout)
    n_out, cout, g_ld = _rows(gout)
    n_in, cin, in_ld = _rows(feats)
    K = kmap.K if kmap is not None else 1
    M = kmap.M if kmap is not None else n_in
    dev = feats.device
    st = cur_stream(dev)
    ws, wsb = ws_args(lib.pcmi_spconv_workspace_bytes(n_in, n_out, cin, cout, K, M), dev)
    gin = gw = gb = None
    if ctx.needs_input_grad[0]:
      gin = torch.empty((n_in, cin), dtype=torch.float32, device=dev)
      check(lib.pcmi_spconv_bwd_data(ptr(gout), g_ld, n_out, cout, ptr(kernel), cin, _kmap_ref(kmap), ctx.transpose,
                                     ptr(gin), cin, n_in, ws, wsb, st))
    if ctx.needs_input_grad[1]:
      gw = torch.empty_like(kernel)
      if ctx.has_bias and ctx.needs_input_grad[2]:
        gb = torch.empty((1, cout), dtype=torch.float32, device=dev)
      check(lib.pcmi_spconv_bwd_weight(ptr(feats), in_ld, n_in, cin, ptr(gout), g_ld, n_out, cout, _kmap_ref(kmap),
                                       ctx.transpose, ptr(gw), ptr(gb), ws, wsb, st))
    return gin, gw, gb, None, None, None, None


class BatchNormFunction(Function):
  """Training-mode BatchNorm1d over the rows, optionally fused with the residual add and
  ReLU that follow it in the reference blocks (pc/model/modules/resnet_block.py:44-60)."""

  @staticmethod
  def forward(ctx, x, gamma, beta, running_mean, running_var, momentum, eps, residual, relu):
    require_cuda(x, "batch norm")
    x = _c(x)
    n, c, x_ld = _rows(x)
    res = _c(residual) if residual is not None else None
    y = torch.empty((n, c), dtype=torch.float32, device=x.device)
    mean = torch.empty(c, dtype=torch.float32, device=x.device)
    invstd = torch.empty(c, dtype=torch.float32, device=x.device)
    ws, wsb = ws_args(lib.pcmi_bn_workspace_bytes(n, c), x.device)
    check(lib.pcmi_bn_fwd_train(ptr(x), x_ld, n, c, ptr(gamma), ptr(beta), ptr(running_mean), ptr(running_var),
                                float(momentum), float(eps), ptr(res), res.stride(0) if res is not None else 0,
                                int(relu), ptr(y), c, ptr(mean), ptr(invstd), ws, wsb, cur_stream(x.device)))
    ctx.save_for_backward(x, gamma, mean, invstd, y if relu else None)
    ctx.has_res, ctx.relu = residual is not None, bool(relu)
    return y

  @staticmethod
  @once_differentiable
  def backward(ctx, dy):
    x, gamma, mean, invstd, y = ctx.saved_tensors
    dy = _c(dy)
    n, c, x_ld = _rows(x)
    dev = x.device
    dx = torch.empty((n, c), dtype=torch.float32, device=dev)
    dres = torch.empty((n, c), dtype=torch.float32, device=dev) if ctx.has_res else None
    dgamma = torch.empty(c, dtype=torch.float32, device=dev)
    dbeta = torch.empty(c, dtype=torch.float32, device=dev)
    ws, wsb = ws_args(lib.pcmi_bn_workspace_bytes(n, c), dev)
    check(lib.pcmi_bn_bwd(ptr(dy), dy.stride(0), ptr(x), x_ld, ptr(y), c if y is not None else 0, n, c, ptr(gamma),
                          ptr(mean), ptr(invstd), ptr(dx), c, ptr(dres), c, ptr(dgamma), ptr(dbeta), ws, wsb,
                          cur_stream(dev)))
    return dx, dgamma, dbeta, None, None, None, None, dres, None


def batch_norm_eval(x, gamma, beta, running_mean, running_var, eps, residual=None, relu=False):
  require_cuda(x, "batch norm (eval)")
  x = _c(x)
  n, c, x_ld = _rows(x)
  res = _c(residual) if residual is not None else None
  y = torch.empty((n, c), dtype=torch.float32, device=x.device)
  check(lib.pcmi_bn_fwd_eval(ptr(x), x_ld, n, c, ptr(gamma), ptr(beta), ptr(running_mean), ptr(running_var),
                             float(eps), ptr(res), res.stride(0) if res is not None else 0, int(relu), ptr(y), c,
                             cur_stream(x.device)))
  return y


class ReLUFunction(Function):

  @staticmethod
  def forward(ctx, x):
    require_cuda(x, "relu")
    x = _c(x)
    n, c, ld = _rows(x)
    y = torch.empty((n, c), dtype=torch.float32, device=x.device)
    check(lib.pcmi_relu_fwd(ptr(x), ld, n, c, ptr(y), c, cur_stream(x.device)))
    ctx.save_for_backward(y)
    return y

  @staticmethod
  @once_differentiable
  def backward(ctx, dy):
    (y,) = ctx.saved_tensors
    dy = _c(dy)
    n, c, _ = _rows(y)
    dx = torch.empty_like(y)
    check(lib.pcmi_relu_bwd(ptr(dy), dy.stride(0), ptr(y), c, n, c, ptr(dx), c, cur_stream(y.device)))
    return dx


class AddFunction(Function):

  @staticmethod
  def forward(ctx, a, b):
    require_cuda(a, "add")
    a, b = _c(a), _c(b)
    n, c, _ = _rows(a)
    y = torch.empty((n, c), dtype=torch.float32, device=a.device)
    check(lib.pcmi_add(ptr(a), a.stride(0), ptr(b), b.stride(0), n, c, ptr(y), c, cur_stream(a.device)))
    return y

  @staticmethod
  def backward(ctx, dy):
    return dy, dy


class L2NormalizeFunction(Function):
  """F / ||F||_2 per row, no eps (pc/model/res16unet.py:262-266)."""

  @staticmethod
  def forward(ctx, x):
    require_cuda(x, "l2 normalise")
    x = _c(x)
    n, c, ld = _rows(x)
    y = torch.empty((n, c), dtype=torch.float32, device=x.device)
    norm = torch.empty(n, dtype=torch.float32, device=x.device)
    check(lib.pcmi_l2norm_fwd(ptr(x), ld, n, c, ptr(y), c, ptr(norm), cur_stream(x.device)))
    ctx.save_for_backward(y, norm)
    return y

  @staticmethod
  @once_differentiable
  def backward(ctx, dy):
    y, norm = ctx.saved_tensors
    dy = _c(dy)
    n, c, _ = _rows(y)
    dx = torch.empty_like(y)
    check(lib.pcmi_l2norm_bwd(ptr(dy), dy.stride(0), ptr(y), c, ptr(norm), n, c, ptr(dx), c, cur_stream(y.device)))
    return dx


class GatherRowsFunction(Function):
  """F[idx] with a scatter-add backward (pc/lib/ddp_trainer.py:209-213,409-410)."""

  @staticmethod
  def forward(ctx, src, idx):
    require_cuda(src, "gather rows")
    src = _c(src)
    idx = idx.to(device=src.device, dtype=torch.int64).contiguous()
    n, c = idx.shape[0], src.shape[1]
    out = torch.empty((n, c), dtype=torch.float32, device=src.device)
    check(lib.pcmi_gather_rows(ptr(src), src.stride(0), ptr(idx), n, c, ptr(out), c, cur_stream(src.device)))
    ctx.save_for_backward(idx)
    ctx.n_src = src.shape[0]
    return out

  @staticmethod
  @once_differentiable
  def backward(ctx, dout):
    (idx,) = ctx.saved_tensors
    dout = _c(dout)
    n, c = dout.shape
    dsrc = torch.zeros((ctx.n_src, c), dtype=torch.float32, device=dout.device)
    check(lib.pcmi_scatter_add_rows(ptr(dout), dout.stride(0), ptr(idx), n, c, ptr(dsrc), c, cur_stream(dout.device)))
    return dsrc, None


class GatherManyFunction(Function):
  """(F[idx_0], F[idx_1], ...) of ONE feature matrix -- the two clouds of a pair run as one two-segment tensor
  (lib/ddp_trainer.py: misc.joint_pair), the second cloud's indices already shifted past the first cloud's rows -- with all
  gradients scattered into ONE zero-filled buffer.  Through one GatherRowsFunction per index set on the slices F[:n0] /
  F[n0:] autograd enqueued nine small kernels between the PointInfoNCE loss and the backward pass (two fills + two scatters
  of the halves, two more full-size fills + two slice copies + an add: 92 us on the chain in profiles/r04zy_*; more for
  the four index sets of the hardest-contrastive loss); here: one fill and one scatter per index set (a scatter adds to
  what is there: index sets may share rows)."""

  @staticmethod
  def forward(ctx, src, *idxs):
    require_cuda(src, "gather many")
    src = _c(src)
    idxs = tuple(i.to(device=src.device, dtype=torch.int64).contiguous() for i in idxs)
    c = src.shape[1]
    outs = []
    for idx in idxs:
      out = torch.empty((idx.shape[0], c), dtype=torch.float32, device=src.device)
      check(lib.pcmi_gather_rows(ptr(src), src.stride(0), ptr(idx), idx.shape[0], c, ptr(out), c, cur_stream(src.device)))
      outs.append(out)
    ctx.save_for_backward(*idxs)
    ctx.n_src = src.shape[0]
    return tuple(outs)

  @staticmethod
  @once_differentiable
  def backward(ctx, *douts):
    idxs = ctx.saved_tensors
    c = douts[0].shape[1]
    dsrc = torch.zeros((ctx.n_src, c), dtype=torch.float32, device=douts[0].device)
    for idx, d in zip(idxs, douts):  # in argument order (deterministic; rows shared between index sets accumulate)
      d = _c(d)
      check(lib.pcmi_scatter_add_rows(ptr(d), d.stride(0), ptr(idx), d.shape[0], c, ptr(dsrc), c, cur_stream(d.device)))
    return (dsrc,) + (None,) * len(idxs)


GatherPairFunction = GatherManyFunction  # (the two-index-set case: q / k of the PointInfoNCE loss)


class NCELossFunction(Function):
  """mean_i(logsumexp_j(q_i.k_j/T) - q_i.k_i/T) without materialising the logits
  (torch.mm + CrossEntropyLoss at pc/lib/ddp_trainer.py:419-426)."""

  @staticmethod
  def forward(ctx, q, k, T):
    require_cuda(q, "nce loss")
    q, k = q.contiguous(), k.contiguous()
    n, c = q.shape
    lse = torch.empty(n, dtype=torch.float32, device=q.device)
    loss = torch.empty((), dtype=torch.float32, device=q.device)
    ws, wsb = ws_args(lib.pcmi_nce_workspace_bytes(n, c), q.device)
    check(lib.pcmi_nce_fwd(ptr(q), ptr(k), n, c, 1.0 / float(T), ptr(lse), ptr(loss), ws, wsb, cur_stream(q.device)))
    ctx.save_for_backward(q, k, lse)
    ctx.inv_T = 1.0 / float(T)
    return loss

  @staticmethod
  @once_differentiable
  def backward(ctx, gloss):
    q, k, lse = ctx.saved_tensors
    n, c = q.shape
    dq, dk = torch.empty_like(q), torch.empty_like(k)
    g = gloss.to(torch.float32).contiguous()
    ws, wsb = ws_args(lib.pcmi_nce_workspace_bytes(n, c), q.device)
    check(lib.pcmi_nce_bwd(ptr(q), ptr(k), ptr(lse), n, c, ctx.inv_T, ptr(g), ptr(dq), ptr(dk), ws, wsb,
                           cur_stream(q.device)))
    return dq, dk, None


def pdist_argmin(a, b):
  """(min_s sqrt(|a_p - b_s|^2 + 1e-7), argmin) -- pc/lib/ddp_trainer.py:182-184,218-219."""
  require_cuda(a, "pdist_argmin")
  a, b = a.contiguous(), b.contiguous()
  p, c = a.shape
  dmin = torch.empty(p, dtype=torch.float32, device=a.device)
  amin = torch.empty(p, dtype=torch.int32, device=a.device)
  check(lib.pcmi_pdist_argmin(ptr(a), p, ptr(b), b.shape[0], c, ptr(dmin), ptr(amin), cur_stream(a.device)))
  return dmin, amin


class PairKeySet:
  """Device hash set of the int64 keys i + j*M of the positive pairs
  (_hash + np.isin at pc/lib/ddp_trainer.py:39-51,224-234)."""

  def __init__(self, pairs_i32, M):
    require_cuda(pairs_i32, "pair key set")
    pairs = pairs_i32.to(torch.int32).contiguous()
    self.M = int(M)
    nbytes = lib.pcmi_keyset_bytes(pairs.shape[0])
    self.buf = torch.empty(nbytes, dtype=torch.uint8, device=pairs.device)
    check(lib.pcmi_keyset_build(ptr(pairs), pairs.shape[0], self.M, ptr(self.buf), nbytes, cur_stream(pairs.device)))

  def absent(self, a_i64, b_i64):
    a, b = a_i64.to(torch.int64).contiguous(), b_i64.to(torch.int64).contiguous()
    mask = torch.empty(a.shape[0], dtype=torch.uint8, device=a.device)
    check(lib.pcmi_keyset_mask_absent(ptr(self.buf), self.buf.numel(), ptr(a), ptr(b), a.shape[0], self.M, ptr(mask),
                                      cur_stream(a.device)))
    return mask


class HardestLossFunction(Function):
  """losses = [pos_loss, neg_loss] of pc/lib/ddp_trainer.py:235-238 given the mined minima."""

  @staticmethod
  def forward(ctx, posF0, posF1, subF0, subF1, d01min, d01ind, mask0, d10min, d10ind, mask1, pos_thresh, neg_thresh):
    require_cuda(posF0, "hardest loss")
    posF0, posF1, subF0, subF1 = posF0.contiguous(), posF1.contiguous(), subF0.contiguous(), subF1.contiguous()
    p, c = posF0.shape
    dev = posF0.device
    losses = torch.empty(2, dtype=torch.float32, device=dev)
    stats = torch.empty(8, dtype=torch.float32, device=dev)
    ws, wsb = ws_args(lib.pcmi_hardest_workspace_bytes(p), dev)
    check(lib.pcmi_hardest_loss_fwd(ptr(posF0), ptr(posF1), p, c, ptr(d01min), ptr(mask0), ptr(d10min), ptr(mask1),
                                    float(pos_thresh), float(neg_thresh), ptr(losses), ptr(stats), ws, wsb,
                                    cur_stream(dev)))
    ctx.save_for_backward(posF0, posF1, subF0, subF1, d01min, d01ind, mask0, d10min, d10ind, mask1, stats)
    ctx.thresh = (float(pos_thresh), float(neg_thresh))
    return losses

  @staticmethod
  @once_differentiable
  def backward(ctx, gl):
    posF0, posF1, subF0, subF1, d01min, d01ind, mask0, d10min, d10ind, mask1, stats = ctx.saved_tensors
    p, c = posF0.shape
    gl = gl.to(torch.float32).contiguous()
    g0, g1 = torch.empty_like(posF0), torch.empty_like(posF1)
    gs0, gs1 = torch.zeros_like(subF0), torch.zeros_like(subF1)
    check(lib.pcmi_hardest_loss_bwd(ptr(posF0), ptr(posF1), p, ptr(subF0), ptr(subF1), c, ptr(d01min), ptr(d01ind),
                                    ptr(mask0), ptr(d10min), ptr(d10ind), ptr(mask1), ctx.thresh[0], ctx.thresh[1],
                                    ptr(stats), ptr(gl), ptr(g0), ptr(g1), ptr(gs0), ptr(gs1), cur_stream(posF0.device)))
    return (g0, g1, gs0, gs1) + (None,) * 8


class SoftmaxCrossEntropyFunction(Function):
  """torch.nn.CrossEntropyLoss(ignore_index=ignore_label) on logits [n, c] / int labels [n]
  (downstream/semseg/lib/train.py:64,124), as libpcmi kernels."""

  @staticmethod
  def forward(ctx, logits, labels, ignore_label):
    require_cuda(logits, "softmax cross-entropy")
    x = logits if (logits.stride(1) == 1 and logits.dtype == torch.float32) else logits.float().contiguous()
    lb = labels.to(device=x.device, dtype=torch.int32).contiguous()
    n, c = x.shape
    out2 = torch.empty(2, dtype=torch.float32, device=x.device)
    ws, wsb = ws_args(lib.pcmi_softmax_ce_workspace_bytes(n), x.device)
    check(lib.pcmi_softmax_ce_fwd(ptr(x), x.stride(0), n, c, ptr(lb), int(ignore_label), ptr(out2), ws, wsb, cur_stream(x.device)))
    ctx.save_for_backward(x, lb, out2)
    ctx.ignore = int(ignore_label)
    return out2[0]

  @staticmethod
  @once_differentiable
  def backward(ctx, gloss):
    x, lb, out2 = ctx.saved_tensors
    n, c = x.shape
    dx = torch.empty((n, c), dtype=torch.float32, device=x.device)
    g = gloss.reshape(1).to(dtype=torch.float32, device=x.device).contiguous()
    check(lib.pcmi_softmax_ce_bwd(ptr(x), x.stride(0), n, c, ptr(lb), ctx.ignore, ptr(out2), ptr(g), ptr(dx), c,
                                  cur_stream(x.device)))
    return dx, None, None


def pairs_scan_host(pos_pairs):
  """(number of runs of column 0, sorted?) of HOST correspondences [P, 2] int32 -- one native pass (pcmi_pairs_scan_host)."""
  assert (not pos_pairs.is_cuda) and pos_pairs.dtype == torch.int32 and pos_pairs.dim() == 2 and pos_pairs.shape[1] == 2 \
      and pos_pairs.is_contiguous(), "correspondences must be a contiguous CPU int32 [P, 2] tensor"
  n, srt = C.c_int64(), C.c_int()
  check(lib.pcmi_pairs_scan_host(C.c_void_p(pos_pairs.data_ptr()), pos_pairs.shape[0], C.byref(n), C.byref(srt)))
  return n.value, bool(srt.value)


def pair_select(pairs_dev, n_unique, uniform_dev, sampled_dev=None, workspace=None):
  """Device side of the PointInfoNCE pair selection (pc/lib/ddp_trainer.py:400-417; pcmi_pair_select): row indices
  (q_idx into F0, k_idx into F1), int64 [npos] (or [n_unique] without a sub-sample).  Enqueued on torch's current stream;
  workspace: a caller-owned uint8 tensor (a call on a stream other than the compute stream must not use the shared
  scratch buffer of the compute-stream ops)."""
  require_cuda(pairs_dev, "pair selection")
  assert pairs_dev.dtype == torch.int32 and pairs_dev.is_contiguous() and uniform_dev.dtype == torch.float32
  assert uniform_dev.numel() == n_unique and (sampled_dev is None or sampled_dev.dtype == torch.int64)
  n_sel = sampled_dev.numel() if sampled_dev is not None else n_unique
  q = torch.empty(n_sel, dtype=torch.int64, device=pairs_dev.device)
  k = torch.empty(n_sel, dtype=torch.int64, device=pairs_dev.device)
  P = pairs_dev.shape[0]
  need = lib.pcmi_pair_select_workspace_bytes(P)
  if workspace is not None:
    assert workspace.is_cuda and workspace.dtype == torch.uint8 and workspace.numel() >= need
    ws, wsb = C.c_void_p(workspace.data_ptr()), C.c_size_t(workspace.numel())
  else:
    ws, wsb = ws_args(need, pairs_dev.device)
  check(lib.pcmi_pair_select(ptr(pairs_dev), P, n_unique, ptr(uniform_dev), ptr(sampled_dev), n_sel, ptr(q), ptr(k), ws, wsb,
                             cur_stream(pairs_dev.device)))
  return q, k


def sgd_step(w, g, v, lr, momentum, weight_decay, grad_scale=1.0, dampening=0.0, first_step=True):
  """torch.optim.SGD.step on flat buffers (pc/lib/ddp_trainer.py:107-111,319,435; with dampening:
  downstream/semseg/lib/solvers.py:52-60).  first_step: torch fills a fresh momentum buffer with the gradient itself."""
  require_cuda(w, "sgd step")
  check(lib.pcmi_sgd_step_dampened(ptr(w), ptr(g), ptr(v), w.numel(), float(lr), float(momentum), float(dampening),
                                   float(weight_decay), float(grad_scale), int(bool(first_step)), cur_stream(w.device)))
