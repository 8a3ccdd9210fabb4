"""Training step of the contrastive pre-training (pc/lib/ddp_trainer.py), on libpcmi.

Same classes and control flow as the reference -- ContrastiveLossTrainer,
HardestContrastiveLossTrainer (:171-326), PointNCELossTrainer (:328-440), `_train_iter`,
checkpoint layout (:151-169) -- with these deliberate differences:
  * every sparse op, both losses and the optimiser are libpcmi HIP kernels (no ME, no torch
    conv / mm / CrossEntropy / SGD kernels);
  * DDP is FlatParameters + GradReducer (RCCL buckets on a side stream) instead of
    torch DistributedDataParallel; BN buffers stay per-rank (broadcast_buffers=False, :101);
  * the host-RNG draws of the reference (np.random.choice, Uniform.sample) can be injected
    (`draws=`) so that parity tests feed the oracle the same index sets;
  * `_train_iter` returns the loss as a 0-dim device tensor; `.item()` happens only when the
    loop logs (stat_freq), so the host never waits on the GPU inside an iteration
    (the reference syncs every iteration at :316-318,433);
  * no torch.cuda.empty_cache() per iteration (:321,437) and no autograd anomaly mode (:36).
"""
import logging
import os
import threading
import time

import numpy as np
import torch

from .. import functional as PF
from .. import minkowski as ME
from ..model import load_model
from . import distributed as du
from .solver import FlatSGD
from .timer import AverageMeter, Timer


def _hash(arr, M):
  """int64 key of index tuples (pc/lib/ddp_trainer.py:39-51); kept for host-side checks."""
  if isinstance(arr, np.ndarray):
    N, D = arr.shape
    cols = [arr[:, d] for d in range(D)]
  else:
    N, D = len(arr[0]), len(arr)
    cols = arr
  hv = np.zeros(N, dtype=np.int64)
  for d in range(D):
    hv += np.asarray(cols[d], dtype=np.int64) * np.int64(M) ** d
  return hv


from .checkpoint import load_state  # noqa: E402,F401  (pc/lib/ddp_trainer.py:54-69)


class ContrastiveLossTrainer:

  def __init__(self, config, data_loader):
    assert config.misc.use_gpu and torch.cuda.is_available(), "the pre-training path runs on a gfx950 GPU"
    num_feats = 3  # ones (+ jitter), pc/lib/ddp_data_loaders.py:248-252
    self.config = config
    # The host side of an iteration is a few small torch-CPU / numpy ops (pair selection, index staging).
    # On a many-core host torch's default intra-op pool (one thread per core) makes each of them cost tens of
    # milliseconds of thread wake-ups; a handful of threads is faster and leaves the cores to the loader workers.
    if torch.get_num_threads() > config.misc.get("host_threads", 8):
      torch.set_num_threads(config.misc.get("host_threads", 8))
    self.world_size = du.get_world_size()
    self.is_master = du.is_master_proc()
    self.cur_device = torch.device("cuda", torch.cuda.current_device())

    Model = load_model(config.net.model)
    model = Model(num_feats, config.net.model_n_out, config, D=3).to(self.cur_device)
    self.model = model
    self.flat = du.FlatParameters(model.parameters())
    if self.world_size > 1:
      # what DDP's constructor does (pc/lib/ddp_trainer.py:96-102: rank 0's parameters everywhere) -- as ONE collective
      # over the flat buffer every parameter is a view of, not one per tensor (round 4: 250 small broadcasts)
      torch.distributed.broadcast(self.flat.w, src=0)
    self.reducer = du.GradReducer(self.flat, bucket_mb=config.misc.get("bucket_mb", du.DEFAULT_BUCKET_MB),
                                   force=config.misc.get("force_reducer", False),
                                   profile=config.misc.get("reducer_profile", False))
    # misc.engine: "native" = whole forward / backward as one libpcmi call each (engine.py);
    #              "autograd" = per-layer torch.autograd.Function path (same kernels)
    self.engine = None
    self.host_ms, self._host_t, self._host_c = {}, 0.0, 0.0
    self._prefetch_thread, self._prefetch_err = None, None
    if config.misc.get("switch_interval", None):
      # helper-thread experiments: CPython hands the GIL over every 5 ms by default, which is a third of an iteration --
      # the enqueueing thread and the preparing thread then take turns at that granularity
      import sys
      sys.setswitchinterval(float(config.misc.switch_interval))
    self._gpu_marks = []
    if config.misc.get("engine", "native") == "native":
      from ..engine import NativeEngine
      self.engine = NativeEngine(model, self.flat, in_channels=num_feats)
    self.optimizer = FlatSGD(self.flat, lr=config.opt.lr, momentum=config.opt.momentum,
                             weight_decay=config.opt.weight_decay, grad_scale=self.reducer.grad_scale)
    if self.engine is not None and config.misc.get("bucket_sgd", False):
      # misc.bucket_sgd=True: the optimiser steps a gradient bucket as soon as it is final (and all-reduced), on the
      # reducer's communication stream beside the rest of the backward pass; optimizer.step() in _backward_and_step covers
      # what is left (nothing, when every bucket fired).  Same arithmetic bit for bit (tests/test_gpu_timing.py).  OFF by
      # default: on one GPU it measured 14.33 against 14.14 ms per step (profiles/r06e_*) -- three more launches on a
      # fourth active stream and three Python callbacks inside the backward enqueue cost more than the 0.13 ms SGD launch
      # they take off the end of the step; with the 1-rank reducer forced on it is neutral (14.79 / 14.84 ms).
      self.reducer.after_bucket = lambda b, lo, hi: self.optimizer.step_range(lo, hi)
    self.scheduler = torch.optim.lr_scheduler.ExponentialLR(self.optimizer, config.opt.exp_gamma)
    self.curr_iter = 0
    self.batch_size = data_loader.batch_size if data_loader is not None else config.trainer.batch_size
    self.data_loader = data_loader
    self.neg_thresh, self.pos_thresh = config.trainer.neg_thresh, config.trainer.pos_thresh
    self.stat_freq, self.lr_update_freq = config.trainer.stat_freq, config.trainer.lr_update_freq

    if config.misc.weight:
      state = torch.load(config.misc.weight, map_location="cpu", weights_only=False)
      load_state(model, state["state_dict"], config.misc.lenient_weight_loading,
                 kernel_order=config.misc.get("weight_kernel_order", "hybrid"))
    ckpt = "weights/weights.pth"
    if os.path.isfile(ckpt):
      state = torch.load(ckpt, map_location="cpu", weights_only=False)
      self.curr_iter = state["curr_iter"]
      load_state(model, state["state_dict"])
      self.optimizer.load_state_dict(state["optimizer"])
      self.scheduler.load_state_dict(state["scheduler"])
      if self.is_master:
        logging.info("=> loaded checkpoint '%s' (curr_iter %d)", ckpt, self.curr_iter)

  # -- checkpoint: {curr_iter, state_dict, optimizer, scheduler, config} + weights.pth symlink ----
  def _save_checkpoint(self, curr_iter, filename="checkpoint"):
    if not self.is_master:
      return
    os.makedirs("weights", mode=0o755, exist_ok=True)
    cfg = self.config.to_dict() if hasattr(self.config, "to_dict") else self.config
    state = {"curr_iter": curr_iter, "state_dict": self.model.state_dict(), "optimizer": self.optimizer.state_dict(),
             "scheduler": self.scheduler.state_dict(), "config": cfg}
    path = os.path.join("weights", filename + ".pth")
    logging.info("Saving checkpoint: %s ...", path)
    torch.save(state, path)
    link = "weights/weights.pth"
    if os.path.lexists(link):
      os.remove(link)
    os.symlink(filename + ".pth", link)

  def _upload(self, idx_cpu, slot):
    """Index vector -> device through a persistent pinned staging buffer: the copy is truly
    asynchronous (a pageable H2D copy would block the host until the stream has drained)."""
    n = idx_cpu.numel()
    st = getattr(self, "_staging", None)
    if st is None:
      st = self._staging = {}
    buf, ev = st.get(slot, (None, None))
    if buf is None or buf.numel() < n:
      buf = torch.empty(int(n * 1.25) + 8192, dtype=torch.int64).pin_memory()
      ev = None
    if ev is not None:
      ev.synchronize()  # the previous iteration's copy out of this buffer (long finished)
    buf[:n].copy_(idx_cpu.reshape(-1))
    out = buf[:n].to(self.cur_device, non_blocking=True)
    ev = torch.cuda.Event()
    ev.record()
    st[slot] = (buf, ev)
    return out.view(idx_cpu.shape)

  def _upload_any(self, t_cpu, slot):
    """A small host vector of any dtype -> device through a persistent pinned staging buffer (see _upload)."""
    st = getattr(self, "_staging_any", None)
    if st is None:
      st = self._staging_any = {}
    n = t_cpu.numel()
    buf, ev = st.get(slot, (None, None))
    if buf is None or buf.numel() < n or buf.dtype != t_cpu.dtype:
      # 25 % head-room (as the device workspace below): the buffer converges after a few batches instead of being
      # re-pinned -- milliseconds per hipHostMalloc, on the preparation path -- at every new maximum
      buf, ev = torch.empty(int(n * 1.25) + 8192, dtype=t_cpu.dtype).pin_memory(), None
    if ev is not None:
      ev.synchronize()
    buf[:n].copy_(t_cpu.reshape(-1))
    out = buf[:n].to(self.cur_device, non_blocking=True)
    ev = torch.cuda.Event()
    ev.record()
    st[slot] = (buf, ev)
    return out

  # -- shared pieces of one iteration ----------------------------------------------------------
  # A batch is *prepared* (uploads, coordinate hash / levels / kernel maps on the plan stream, host-side index
  # selection) independently of the network state, so with misc.prefetch the next batch is prepared right after
  # the current step has been enqueued and overlaps with the GPU still working on it.
  def _prepare(self, input_dict, draws=None):
    if self.engine is not None and self.config.misc.get("joint_pair", True):
      # The two clouds of the pair as ONE sparse tensor (cloud 1's batch indices shifted past cloud 0's): one
      # forward / backward pass with half the launches and twice the rows per launch.  The reference calls its
      # network once per cloud (ddp_trainer.py:404-407); what differs between one call and two is only the BatchNorm
      # batch statistics, and the engine keeps those per segment (set_split).
      sub = self._prep_mark  # misc.host_profile: where the preparation spends its host time
      sub(None)
      C0, C1 = input_dict["sinput0_C"], input_dict["sinput1_C"]
      n_batch0 = int(C0[:, 0].max()) + 1 if C0.shape[0] else 0
      C1 = C1.clone()
      C1[:, 0] += n_batch0
      Cj, Fj = torch.cat([C0, C1]), torch.cat([input_dict["sinput0_F"], input_dict["sinput1_F"]])
      sub("prep.concat")
      # (defer_check: the insert's duplicate / range status comes back with plan_unet's one synchronisation)
      sj = ME.SparseTensor(Fj, coords=Cj).to(self.cur_device, defer_check=True)
      sj.coords_man.set_split(C0.shape[0])
      sub("prep.upload_and_hash")
      sj.coords_man.plan_unet(self.engine.n_down)
      sub("prep.levels_and_maps")
      prep = {"input": input_dict, "sj": sj, "n0": int(C0.shape[0])}
      self._prepare_loss(prep, draws)
      sub("prep.pair_selection")
      return prep
    else:
      planned = self.engine is not None
      s0 = ME.SparseTensor(input_dict["sinput0_F"], coords=input_dict["sinput0_C"]).to(self.cur_device, defer_check=planned)
      s1 = ME.SparseTensor(input_dict["sinput1_F"], coords=input_dict["sinput1_C"]).to(self.cur_device, defer_check=planned)
      if self.engine is not None:
        s0.coords_man.plan_unet(self.engine.n_down)
        s1.coords_man.plan_unet(self.engine.n_down)
      prep = {"input": input_dict, "s0": s0, "s1": s1}
    self._prepare_loss(prep, draws)
    return prep

  def _prepare_loss(self, prep, draws):
    pass

  def _join_draws(self):
    """Waits for a helper thread that is consuming the GLOBAL numpy generator (HardestContrastiveLossTrainer's draws).
    Called before anything else may touch that generator -- another batch's draws, or next(data_loader_iter) whose
    dataset transforms use np.random / random in-process (num_workers = 0) -- so that the stream is consumed in the
    reference's order whatever the prefetch mode.  A failure is re-raised where the draws are used."""
    th = getattr(self, "_draw_thread", None)
    if th is not None:
      th.join()
      self._draw_thread = None

  def _prep_mark(self, phase):
    if not self.config.misc.get("host_profile", False):
      return
    now = time.perf_counter()
    if phase is not None:
      self.host_ms[phase] = self.host_ms.get(phase, 0.0) + (now - self._prep_t) * 1e3
    self._prep_t = now

  def _next_prepared(self, data_loader_iter, data_timer, draws):
    th = getattr(self, "_prefetch_thread", None)
    if th is not None:  # batch prepared by the helper thread while the previous step was being enqueued
      th.join()
      self._prefetch_thread = None
      err, self._prefetch_err = self._prefetch_err, None
      if err is not None:
        raise err
    nxt = getattr(self, "_prefetched", None)
    self._prefetched = None
    if nxt is not None:
      if draws is None:
        for key in ("s0", "s1", "sj"):
          if hasattr(nxt.get(key), "wait_upload"):
            nxt[key].wait_upload()
        return nxt, 0.0
      logging.warning("injected draws: the prefetched batch is discarded and a fresh one is prepared")
    data_timer.tic()
    self._join_draws()
    input_dict = next(data_loader_iter)
    data_time = data_timer.toc(average=False)
    return self._prepare(input_dict, draws), data_time

  def _prefetch_worker(self, input_dict):
    try:
      if self.cur_device.type == "cuda":
        torch.cuda.set_device(self.cur_device)  # the current device is per thread
      with ME.deferred_upload_sync():  # the compute stream waits when the batch is CONSUMED (_next_prepared)
        self._prefetched = self._prepare(input_dict)
    except BaseException as e:  # re-raised on the training thread by _next_prepared
      self._prefetch_err = e

  def _prefetch_start(self, data_loader_iter, draws):
    """misc.prefetch_thread: the next batch's uploads / coordinate planning (host-synchronous on the plan stream:
    ~10 ms of waiting per batch) run on a helper thread WHILE this step is enqueued, instead of after it.  The C
    calls release the GIL; the helper only touches its own coordinate handles and the plan stream."""
    if getattr(self, "_last_iter", False):
      return
    if draws is None and self._prefetch_mode() == "thread":
      self._prefetch_err = None
      self._join_draws()
      th = threading.Thread(target=self._prefetch_worker, args=(next(data_loader_iter),), daemon=True)
      th.start()
      self._prefetch_thread = th

  def _host_mark(self, phase):
    """Host-clock phase accounting (misc.host_profile=True): where the enqueueing thread spends an iteration.
    With misc.gpu_profile=True the same marks also drop timing events into the compute stream (gpu_phase_ms())."""
    if self.config.misc.get("gpu_profile", False):
      ev = torch.cuda.Event(enable_timing=True)
      ev.record()
      self._gpu_marks.append((phase, ev))
    if not self.config.misc.get("host_profile", False):
      return
    now, cpu = time.perf_counter(), time.thread_time()
    if phase is not None:  # wall clock, and CPU time of this thread (wall - cpu = blocked, e.g. on a full queue)
      self.host_ms[phase] = self.host_ms.get(phase, 0.0) + (now - self._host_t) * 1e3
      self.host_ms[phase + "_cpu"] = self.host_ms.get(phase + "_cpu", 0.0) + (cpu - self._host_c) * 1e3
    self._host_t, self._host_c = now, cpu

  def gpu_phase_ms(self, skip=0):
    """Stream time between consecutive marks, averaged over the recorded iterations (after a synchronize)."""
    acc, cnt, prev, it = {}, {}, None, 0
    for phase, ev in self._gpu_marks:
      if phase is None:
        it += 1
        if prev is not None and it > skip + 1:
          acc["(between iterations)"] = acc.get("(between iterations)", 0.0) + prev.elapsed_time(ev)
          cnt["(between iterations)"] = cnt.get("(between iterations)", 0) + 1
      elif it > skip:
        acc[phase] = acc.get(phase, 0.0) + prev.elapsed_time(ev)
        cnt[phase] = cnt.get(phase, 0) + 1
      prev = ev
    return {k: round(acc[k] / cnt[k], 3) for k in acc}

  def _prefetch_mode(self):
    """"thread" | "inline" | None.  With the pair run as one two-segment pass (misc.joint_pair) the default is None:
    the batch is prepared at the top of its own iteration, which already overlaps with the GPU (the host is one
    backward pass ahead of it); measured on the bench configuration 20.7 ms / iteration against 29.6 with the helper
    thread and 29.0 with the inline prefetch, whose plan kernels then interleave with the doubled-size backward
    kernels (BatchNorm backward 12 -> 19 ms).  misc.prefetch / misc.prefetch_thread set explicitly still win."""
    m = self.config.misc
    if "prefetch" not in m and "prefetch_thread" not in m and self.engine is not None and m.get("joint_pair", True):
      return None
    if not m.get("prefetch", True):
      return None
    return "thread" if m.get("prefetch_thread", True) else "inline"

  def _prefetch(self, data_loader_iter, draws):
    if getattr(self, "_last_iter", False):
      return
    if draws is None and self._prefetch_mode() == "inline":
      self._join_draws()
      self._prefetched = self._prepare(next(data_loader_iter))

  def _forward_pair(self, prep):
    if "sj" in prep:  # the pair as one two-segment pass (see _prepare)
      F = self.engine.forward(0, prep["sj"], self.model.training).requires_grad_(True)
      self._feats = (F,)
      return F[:prep["n0"]], F[prep["n0"]:]
    s0, s1 = prep["s0"], prep["s1"]
    if self.engine is None:
      return self.model(s0).F, self.model(s1).F
    if self.model.training and self.config.misc.get("concurrent_forward", True):
      F0, F1 = self.engine.forward_pair(s0, s1)  # two streams, same results
      F0, F1 = F0.requires_grad_(True), F1.requires_grad_(True)
    else:
      F0 = self.engine.forward(0, s0, self.model.training).requires_grad_(True)
      F1 = self.engine.forward(1, s1, self.model.training).requires_grad_(True)
    self._feats = (F0, F1)
    return F0, F1

  def _backward_and_step(self, loss, result):
    loss.backward()  # autograd path: through both networks; engine path: only down to F0 / F1
    if self.engine is not None and len(self._feats) == 1:
      self.engine.backward(0, self._feats[0].grad, reducer=self.reducer)
      self._feats = None
    elif self.engine is not None:
      F0, F1 = self._feats
      self.engine.backward(1, F1.grad)
      self.engine.backward(0, F0.grad, reducer=self.reducer)  # last pass: buckets become final -> RCCL
      self._feats = None
    self.reducer.finish()
    if self.world_size > 1 or self.reducer.active:
      result = du.scaled_all_reduce_dict({k: v.detach().clone() for k, v in result.items()}, self.world_size)
    self.optimizer.step()
    return result

  def _prefetch_drain(self):
    """Joins the batch-preparation helper and drops what it prepared (end of train(), or an exception in a step):
    the helper must not be inside HIP / plan-stream calls while the process group and the interpreter tear down."""
    th, self._prefetch_thread = getattr(self, "_prefetch_thread", None), None
    if th is not None:
      th.join()
    self._prefetched, self._prefetch_err = None, None

  def train(self):
    try:
      self._train_loop()
    finally:
      self._prefetch_drain()

  def _train_loop(self):
    curr_iter = self.curr_iter
    it = iter(self.data_loader)
    data_meter, data_timer, total_timer = AverageMeter(), Timer(), Timer()
    while curr_iter < self.config.opt.max_iter:
      curr_iter += 1
      self._last_iter = curr_iter >= self.config.opt.max_iter  # nothing is prefetched past the last iteration
      epoch = curr_iter / max(len(self.data_loader), 1)
      result = self._train_iter(it, [data_meter, data_timer, total_timer])
      if curr_iter % self.lr_update_freq == 0 or curr_iter == 1:
        lr = self.scheduler.get_last_lr()
        self.scheduler.step()
        if self.is_master:
          logging.info(" Epoch: %s, LR: %s", epoch, lr)
          self._save_checkpoint(curr_iter, "checkpoint_" + str(curr_iter))
      if curr_iter % self.stat_freq == 0 and self.is_master:
        vals = {k: float(v) for k, v in (result.items() if isinstance(result, dict) else [("loss", result)])}
        logging.info("Train Epoch: %.3f [%d/%d], Current Loss: %.3e %s\tData time: %.4f, Train time: %.4f, "
                     "Iter time: %.4f, LR: %s", epoch, curr_iter, len(self.data_loader), vals["loss"],
                     {k: round(v, 5) for k, v in vals.items() if k != "loss"}, data_meter.avg,
                     total_timer.avg - data_meter.avg, total_timer.avg, self.scheduler.get_last_lr())
        data_meter.reset()
        total_timer.reset()
    self.curr_iter = curr_iter


class HardestContrastiveLossTrainer(ContrastiveLossTrainer):

  @staticmethod
  def _draw_hardest(N0, N1, P, num_pos, num_hn_samples, draws):
    """np.random.choice in the reference's order (pc/lib/ddp_trainer.py:198-206): candidates of cloud 0, of cloud 1,
    positives.  Each call permutes the whole range (350k rows, 840k pairs per 4-pair batch: milliseconds of host time)."""
    draws = draws or {}
    sel0 = draws["sel0"] if "sel0" in draws else np.random.choice(N0, min(N0, num_hn_samples), replace=False)
    sel1 = draws["sel1"] if "sel1" in draws else np.random.choice(N1, min(N1, num_hn_samples), replace=False)
    if "pos_sel" in draws:
      pos_sel = draws["pos_sel"]
    else:
      pos_sel = np.random.choice(P, num_pos, replace=False) if P > num_pos else None
    return sel0, sel1, pos_sel

  def _upload_hardest(self, N0, N1, pp, drawn, stream, slot=0, keys=None):
    """The drawn rows and (unless `keys` has it) the hash set of ALL positive pairs on the device, through pinned staging
    buffers, on `stream` (None: the current one).  Round 3 had five pageable copies here -- each drains the compute
    stream before it returns -- and a 50 us insert kernel between the forward pass and the loss."""
    sel0, sel1, pos_sel = drawn
    sample = pp if pos_sel is None else pp[np.asarray(pos_sel)]
    host = lambda a, dt: torch.as_tensor(np.ascontiguousarray(a)).to(dt)
    cur = torch.cuda.current_stream(self.cur_device)
    with torch.cuda.stream(stream if stream is not None else cur):
      up = lambda a, dt, name: self._upload_any(host(a, dt), ("hardest", name, slot))
      out = dict(sel0=up(sel0, torch.int64, "sel0"), sel1=up(sel1, torch.int64, "sel1"),
                 pos0=up(sample[:, 0], torch.int64, "pos0"), pos1=up(sample[:, 1], torch.int64, "pos1"))
      # cloud 1's rows in the pair's ONE feature matrix (PF.GatherManyFunction; misc.joint_pair)
      out["sel1_joint"], out["pos1_joint"] = out["sel1"] + N0, out["pos1"] + N0
      held = [out["sel0"], out["sel1"], out["pos0"], out["pos1"], out["sel1_joint"], out["pos1_joint"]]
      if keys is None:
        pairs_d = up(pp.reshape(-1), torch.int32, "pairs").view(-1, 2)
        keys = PF.PairKeySet(pairs_d, max(N0, N1))
        held += [pairs_d, keys.buf]
      out.update(keys=keys, n=(N0, N1), event=None)
      if stream is not None:
        out["event"] = torch.cuda.Event()
        out["event"].record(stream)
        for t_ in held:
          t_.record_stream(cur)
    return out

  def _prepare_loss(self, prep, draws):
    """What the loss needs besides the features depends on the batch only (pc/lib/ddp_trainer.py:198-213,224-226), so it is
    started with the rest of the batch preparation: the key set of the positive pairs is built on the planning stream
    now; the three np.random.choice calls -- the longest host-side item of this trainer's iteration -- run in a helper
    thread (numpy's shuffle releases the GIL) while this thread enqueues the forward pass, and are joined in front of
    the loss -- and in front of anything else that may touch np.random (_join_draws: the next batch's draws, the loader's
    next item), so the global generator is consumed in the reference's order in every prefetch mode."""
    from ..runtime import handle_pool
    slot = getattr(self, "_slot", 0)
    self._slot = slot ^ 1  # two sets of staging buffers: a prefetched batch must not overwrite the live one
    inp = prep["input"]
    pp = inp["correspondences"]
    pp = pp.numpy() if torch.is_tensor(pp) else np.asarray(pp)
    N0, N1 = int(inp["sinput0_C"].shape[0]), int(inp["sinput1_C"].shape[0])
    plan, cur = handle_pool.plan_stream(self.cur_device), torch.cuda.current_stream(self.cur_device)
    with torch.cuda.stream(plan):
      pairs_d = self._upload_any(torch.as_tensor(np.ascontiguousarray(pp.reshape(-1))).to(torch.int32),
                                 ("hardest", "pairs", slot)).view(-1, 2)
      keys = PF.PairKeySet(pairs_d, max(N0, N1))
      pairs_d.record_stream(cur)
      keys.buf.record_stream(cur)
    box = {}
    args = (N0, N1, len(pp), self.config.trainer.num_pos_per_batch * self.batch_size,
            self.config.trainer.num_hn_samples_per_batch * self.batch_size, draws)

    def work():
      try:
        box["drawn"] = self._draw_hardest(*args)
      except BaseException as e:  # re-raised by the joining thread
        box["error"] = e

    self._join_draws()  # (a prefetched batch: the previous batch's draws come first, as in the reference's loop)
    th = threading.Thread(target=work, name="pcmi-hardest-draws", daemon=True)
    th.start()
    self._draw_thread = th
    prep["hardest"] = dict(thread=th, box=box, keys=keys, n=(N0, N1), pp=pp, slot=slot, plan=plan)

  def _finish_prepared_loss(self, h):
    h["thread"].join()
    if "error" in h["box"]:
      raise h["box"]["error"]
    return self._upload_hardest(h["n"][0], h["n"][1], h["pp"], h["box"]["drawn"], h["plan"], h["slot"], keys=h["keys"])

  def contrastive_hardest_negative_loss(self, F0, F1, positive_pairs, num_pos=5192, num_hn_samples=2048,
                                        draws=None, prepared=None, joint=None):
    """pc/lib/ddp_trainer.py:186-238.  positive_pairs: CPU int tensor / array [P,2].
    draws: optional dict(sel0, sel1, pos_sel) replacing the np.random.choice calls.
    prepared: the samples / key set of _prepare_loss (then positive_pairs, num_*, draws are not looked at)."""
    N0, N1 = F0.shape[0], F1.shape[0]
    if prepared is None:
      if not hasattr(self, "cur_device"):  # (a bare instance, as the parity tests make one)
        self.cur_device = F0.device
      pp = positive_pairs.numpy() if torch.is_tensor(positive_pairs) else np.asarray(positive_pairs)
      prepared = self._upload_hardest(N0, N1, pp, self._draw_hardest(N0, N1, len(pp), num_pos, num_hn_samples, draws), None)
    elif "thread" in prepared:
      prepared = self._finish_prepared_loss(prepared)
    assert prepared["n"] == (N0, N1), "hardest samples were drawn for %s rows, the features have %s" % (prepared["n"], (N0, N1))
    if prepared.get("event") is not None:  # uploads + key set ran on the planning stream (_prepare_loss)
      torch.cuda.current_stream(F0.device).wait_event(prepared["event"])
    sel0_d, sel1_d, pos0_d, pos1_d, keys = (prepared[k] for k in ("sel0", "sel1", "pos0", "pos1", "keys"))

    if joint is not None and "sel1_joint" in prepared:
      # joint: the pair's one feature matrix [N0 + N1, c] (F0 / F1 are its two row ranges): four gathers, ONE gradient buffer
      subF0, subF1, posF0, posF1 = PF.GatherManyFunction.apply(joint, sel0_d, prepared["sel1_joint"], pos0_d, prepared["pos1_joint"])
    else:
      subF0, subF1 = PF.GatherRowsFunction.apply(F0, sel0_d), PF.GatherRowsFunction.apply(F1, sel1_d)
      posF0, posF1 = PF.GatherRowsFunction.apply(F0, pos0_d), PF.GatherRowsFunction.apply(F1, pos1_d)
    with torch.no_grad():
      D01min, D01ind = PF.pdist_argmin(posF0, subF1)
      D10min, D10ind = PF.pdist_argmin(posF1, subF0)
      neg1 = sel1_d[D01ind.long()]  # row of F1 mined for each posF0
      neg0 = sel0_d[D10ind.long()]
      mask0 = keys.absent(pos0_d, neg1)
      mask1 = keys.absent(neg0, pos1_d)
    losses = PF.HardestLossFunction.apply(posF0, posF1, subF0, subF1, D01min, D01ind, mask0, D10min, D10ind, mask1,
                                          self.pos_thresh, self.neg_thresh)
    self._last_mined = dict(D01ind=D01ind, D10ind=D10ind, mask0=mask0, mask1=mask1)
    return losses[0], losses[1]

  def _train_iter(self, data_loader_iter, timers, draws=None):
    self.model.train()
    data_meter, data_timer, total_timer = timers
    mark = self._host_mark
    mark(None)
    self.optimizer.zero_grad()
    total_timer.tic()
    prep, data_time = self._next_prepared(data_loader_iter, data_timer, draws)
    self._prefetch_start(data_loader_iter, draws)
    mark("next_prepared")
    F0, F1 = self._forward_pair(prep)
    mark("forward")
    pos_loss, neg_loss = self.contrastive_hardest_negative_loss(
        F0, F1, prep["input"]["correspondences"],
        num_pos=self.config.trainer.num_pos_per_batch * self.batch_size,
        num_hn_samples=self.config.trainer.num_hn_samples_per_batch * self.batch_size, draws=draws,
        prepared=prep.get("hardest"),
        joint=self._feats[0] if ("sj" in prep and len(self._feats) == 1 and self.config.misc.get("fused_pair_gather", True)) else None)
    loss = pos_loss + neg_loss
    mark("loss")
    result = self._backward_and_step(loss, {"loss": loss.detach(), "pos_loss": pos_loss.detach(),
                                            "neg_loss": neg_loss.detach()})
    mark("backward_step")
    self._prefetch(data_loader_iter, draws)
    mark("prefetch")
    total_timer.toc()
    data_meter.update(data_time)
    return result


class PointNCELossTrainer(ContrastiveLossTrainer):

  def __init__(self, config, data_loader):
    super().__init__(config, data_loader)
    self.T, self.npos = config.misc.nceT, config.misc.npos

  @staticmethod
  def select_pairs(pos_pairs, npos, draws=None):
    """One key per unique query voxel, optional npos sub-sample
    (pc/lib/ddp_trainer.py:403-417).  The reference runs unique / cumsum on the GPU and the
    RNG on the host; the correspondences arrive on the host anyway, so the whole selection
    is host-side here (identical arithmetic: fp32 floor(u * count)) and only the two index
    vectors are uploaded."""
    # ~840k correspondences per 4-pair batch, on the thread that enqueues the iteration (it was 3.8 ms of a 16.6 ms step
    # as numpy passes over int64 copies): the two full-length passes are torch's threaded kernels on the int32 columns
    # as they are; everything after the run starts works on the ~70k unique queries.
    pp = pos_pairs if torch.is_tensor(pos_pairs) else torch.from_numpy(np.asarray(pos_pairs))
    col0, col1 = pp[:, 0], pp[:, 1]
    n = col0.shape[0]
    if n > 1 and bool((col0[1:] < col0[:-1]).any()):  # the loader's contract is "sorted by query row"
      order = torch.from_numpy(np.argsort(col0.numpy(), kind="stable"))
      col0, col1 = col0[order], col1[order]
    # unique(return_counts=True) of a sorted column = its run starts / run lengths (no sort)
    starts = torch.zeros(min(n, 1), dtype=torch.int64)
    if n > 1:
      starts = torch.cat([starts, torch.nonzero(col0[1:] != col0[:-1]).squeeze(1) + 1])
    count = torch.diff(starts, append=torch.tensor([n], dtype=torch.int64)) if n else starts
    q_unique = col0[starts].long()
    draws = draws or {}
    # (torch.distributions.Uniform(0, 1).sample([n]) of the reference IS torch.rand(n) * 1 + 0: same values, same
    #  generator consumption, without building a distribution object per iteration)
    uniform = draws["uniform"] if "uniform" in draws else torch.rand(len(count))
    off = torch.floor(torch.as_tensor(uniform, dtype=torch.float32) * count).long()
    k_sel = col1[off + starts].long()  # starts = exclusive cumsum of the counts
    if npos < q_unique.shape[0]:
      si = draws["sampled_inds"] if "sampled_inds" in draws else np.random.choice(q_unique.shape[0], npos, replace=False)
      si = torch.as_tensor(np.asarray(si)).long()
      q_unique, k_sel = q_unique[si], k_sel[si]
    return q_unique, k_sel

  def select_pairs_device(self, pos_pairs, npos, draws=None, slot=0, defer_wait=False):
    """select_pairs with the run detection and the gathers on the device (csrc/pairs.hip): the host keeps what consumes
    the random-number streams -- torch.rand(n_unique), np.random.choice(n_unique, npos) -- in the same order as
    select_pairs, plus one native pass over column 0 for n_unique.  Returns device int64 (q_idx, k_idx); bit-identical
    to select_pairs (tests/test_gpu_parity.py::test_device_pair_selection_is_bit_identical).  None: the correspondences
    are not a sorted contiguous int32 tensor (the caller falls back to the host path).  defer_wait: also returns the event
    the consumer's stream has to wait for (the selection runs on the planning stream); otherwise the current stream
    waits for it here."""
    pp = pos_pairs if torch.is_tensor(pos_pairs) else torch.from_numpy(np.asarray(pos_pairs))
    if pp.is_cuda or pp.dtype != torch.int32 or pp.dim() != 2 or not pp.is_contiguous() or pp.shape[0] == 0:
      return None
    nq, is_sorted = PF.pairs_scan_host(pp)
    if not is_sorted:
      return None
    draws = draws or {}
    uniform = draws["uniform"] if "uniform" in draws else torch.rand(nq)
    uniform = torch.as_tensor(uniform, dtype=torch.float32)
    assert uniform.numel() == nq, "uniform draws: %d for %d unique queries" % (uniform.numel(), nq)
    si = None
    if npos < nq:
      si = draws["sampled_inds"] if "sampled_inds" in draws else np.random.choice(nq, npos, replace=False)
      si = torch.as_tensor(np.asarray(si)).long()
    # Upload + selection go to the planning stream: nothing of it is needed before the loss, and on the compute stream
    # the 6.7 MB copy of the correspondences sat in front of the forward pass (237.7 against 240.1 pairs/s with the
    # selection on the host, profiles/r03c_bench_ab.txt).  The compute stream waits for `sel_event` in front of the gathers.
    from ..runtime import handle_pool
    from .._lib import lib
    plan, cur = handle_pool.plan_stream(self.cur_device), torch.cuda.current_stream(self.cur_device)
    with torch.cuda.stream(plan):
      pairs_d = pp.to(self.cur_device, non_blocking=True) if pp.is_pinned() else self._upload_any(pp, ("pairs", slot)).view(pp.shape)
      u_d = self._upload_any(uniform, ("uniform", slot))
      si_d = self._upload_any(si, ("sampled", slot)) if si is not None else None
      need = lib.pcmi_pair_select_workspace_bytes(pp.shape[0])
      ws_cache = getattr(self, "_pair_ws", None)
      if ws_cache is None:
        ws_cache = self._pair_ws = {}
      ws = ws_cache.get(slot)
      if ws is None or ws.numel() < need:
        ws = ws_cache[slot] = torch.empty(int(need * 1.25) + 4096, dtype=torch.uint8, device=self.cur_device)
      q_d, k_d = PF.pair_select(pairs_d, nq, u_d, si_d, workspace=ws)
      ev = torch.cuda.Event()
      ev.record(plan)
    for t_ in (q_d, k_d, pairs_d, u_d) + ((si_d,) if si_d is not None else ()):
      t_.record_stream(cur)
    if defer_wait:
      return q_d, k_d, ev
    cur.wait_event(ev)
    return q_d, k_d

  def _prepare_loss(self, prep, draws):
    slot = getattr(self, "_slot", 0)
    self._slot = slot ^ 2  # two staging buffer pairs: the prefetched batch must not overwrite the live one
    if self.config.misc.get("device_pair_selection", True):
      sel = self.select_pairs_device(prep["input"]["correspondences"], self.npos, draws, slot, defer_wait=True)
      if sel is not None:
        prep["q_idx"], prep["k_idx"], prep["sel_event"] = sel
        if "n0" in prep:  # the pair as one tensor: the keys' rows in it (PF.GatherPairFunction), still on the planning stream
          from ..runtime import handle_pool
          plan, cur = handle_pool.plan_stream(self.cur_device), torch.cuda.current_stream(self.cur_device)
          with torch.cuda.stream(plan):
            prep["k_idx_joint"] = prep["k_idx"] + prep["n0"]
            prep["sel_event"] = torch.cuda.Event()
            prep["sel_event"].record(plan)
          prep["k_idx_joint"].record_stream(cur)
        return
    q_idx, k_idx = self.select_pairs(prep["input"]["correspondences"], self.npos, draws)
    prep["q_idx"], prep["k_idx"] = self._upload(q_idx, slot), self._upload(k_idx, slot + 1)
    if "n0" in prep:
      prep["k_idx_joint"] = self._upload(k_idx + prep["n0"], slot + 4)

  def _train_iter(self, data_loader_iter, timers, draws=None):
    self.model.train()
    data_meter, data_timer, total_timer = timers
    mark = self._host_mark
    mark(None)
    self.optimizer.zero_grad()
    total_timer.tic()
    prep, data_time = self._next_prepared(data_loader_iter, data_timer, draws)
    self._prefetch_start(data_loader_iter, draws)
    mark("next_prepared")
    F0, F1 = self._forward_pair(prep)
    mark("forward")
    if prep.get("sel_event") is not None:  # the pair selection ran on the planning stream (select_pairs_device)
      torch.cuda.current_stream(self.cur_device).wait_event(prep["sel_event"])
    if "k_idx_joint" in prep and len(self._feats) == 1 and self.config.misc.get("fused_pair_gather", True):
      # both gathers from the pair's ONE feature matrix, both gradients into one buffer (no slice / accumulate kernels)
      q, k = PF.GatherPairFunction.apply(self._feats[0], prep["q_idx"], prep["k_idx_joint"])
    else:
      q = PF.GatherRowsFunction.apply(F0, prep["q_idx"])
      k = PF.GatherRowsFunction.apply(F1, prep["k_idx"])
    loss = PF.NCELossFunction.apply(q, k, self.T)
    mark("loss")
    result = self._backward_and_step(loss, {"loss": loss.detach()})
    mark("backward_step")
    self._prefetch(data_loader_iter, draws)
    mark("prefetch")
    total_timer.toc()
    data_meter.update(data_time)
    return result
