"""world_size-2 gloo test of the gradient reducer: bucketed all-reduce over the flat gradient
buffer == gradients of the concatenated batch / world (what DDP gives the reference,
pc/lib/ddp_trainer.py:96-102), and per-rank buffers stay local."""
import os
import socket

import torch
import torch.multiprocessing as mp


def _free_port():
  s = socket.socket()
  s.bind(("127.0.0.1", 0))
  p = s.getsockname()[1]
  s.close()
  return p


def _worker(rank, world, port, q):
  os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
  from pointcontrast_amd.lib import distributed as du
  du.init_process_group(rank, world, backend="gloo")
  torch.manual_seed(0)
  model = torch.nn.Sequential(torch.nn.Linear(6, 16), torch.nn.ReLU(), torch.nn.Linear(16, 16), torch.nn.ReLU(),
                              torch.nn.Linear(16, 3))
  flat = du.FlatParameters(model.parameters())
  red = du.GradReducer(flat, bucket_mb=0.00015)  # ~40 floats per bucket -> three buckets
  assert len(red.buckets) >= 3 and red.buckets[0][1] == flat.numel
  torch.manual_seed(100)
  full = torch.randn(8, 6)
  x = full[rank * 4:(rank + 1) * 4]
  for _ in range(2):  # two iterations: hook bookkeeping must reset
    flat.zero_grad()
    (model(x).pow(2).sum() + model(2 * x).sum()).backward()  # two uses of the shared weights
    red.finish()
  # a step that fails after SOME buckets were launched (a bucket callback raised on every rank, NativeEngine.backward):
  # abort() waits for what is in flight and forgets the step; the next step must reduce EVERY bucket again -- without it
  # the buckets marked launched would be skipped and the ranks would step on un-reduced gradients
  flat.zero_grad()
  red._launch(0)
  assert red._launched[0] and red._works
  red.abort()
  assert not any(red._launched) and not red._works and not any(red._pending)
  flat.zero_grad()
  n_before = red.n_launched_total
  (model(x).pow(2).sum() + model(2 * x).sum()).backward()
  red.finish()
  assert red.n_launched_total == n_before + len(red.buckets)
  avg = flat.g.clone() * red.grad_scale
  # single-process reference on the concatenated batch
  ref = torch.nn.Sequential(torch.nn.Linear(6, 16), torch.nn.ReLU(), torch.nn.Linear(16, 16), torch.nn.ReLU(),
                            torch.nn.Linear(16, 3))
  ref.load_state_dict(model.state_dict())
  (ref(full).pow(2).sum() + ref(2 * full).sum()).backward()
  refg = torch.cat([p.grad.reshape(-1) for p in ref.parameters()])
  got = torch.cat([flat.view(avg, i).reshape(-1) for i in range(len(flat.params))])
  ok = torch.allclose(got * world, refg, atol=1e-5)
  res = du.scaled_all_reduce_dict({"loss": torch.tensor(float(rank + 1))}, world)
  q.put((rank, bool(ok), float(res["loss"])))
  du.destroy_process_group()


def test_grad_reducer_world2_gloo():
  ctx = mp.get_context("spawn")
  q = ctx.Queue()
  port = _free_port()
  procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
  for p in procs:
    p.start()
  out = [q.get(timeout=120) for _ in procs]
  for p in procs:
    p.join(timeout=60)
    assert p.exitcode == 0
  assert all(ok for _, ok, _ in out), out
  assert all(abs(l - 1.5) < 1e-6 for _, _, l in out), out


def _engine_worker(rank, world, port, q):
  """The REAL GradReducer driven through NativeEngine._ready_args by a CPU stand-in for pcmi_net_backward's bucket
  logic (csrc/engine.hip::run_backward: a bucket is final after the lowest-index op that owns parameters of it; the
  executor reports it by its position in the ASCENDING list of bucket offsets)."""
  os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
  from pointcontrast_amd.engine import NativeEngine, lower_model
  from pointcontrast_amd.lib import distributed as du
  from pointcontrast_amd.lib.config import get_config
  from pointcontrast_amd.model import load_model
  du.init_process_group(rank, world, backend="gloo")
  torch.manual_seed(0)
  model = load_model("Res16UNet14")(3, 32, get_config([]), D=3)
  flat = du.FlatParameters(model.parameters())
  red = du.GradReducer(flat, bucket_mb=4.0)
  assert red.active and len(red.buckets) >= 3
  prog = lower_model(model, flat)
  cb, lo_arr, nb = NativeEngine._ready_args(red)
  lo = [lo_arr[i] for i in range(nb)]
  assert lo == sorted(lo) and lo[0] == 0 and nb == len(red.buckets)

  def bucket_of(off):
    b = 0
    for qq in range(nb):
      if off >= lo[qq]:
        b = qq
    return b

  ops = prog["ops"]
  last = [-1] * nb
  for i in range(len(ops) - 1, -1, -1):
    o = ops[i]
    if o["type"] == 2:
      continue
    last[bucket_of(o["w_off"])] = i
    if o["type"] == 1 or o.get("has_bias"):
      last[bucket_of(o["b_off"])] = i
  fired = []
  for it in range(2):  # two iterations: the reducer's bookkeeping must reset
    flat.zero_grad()
    for i in range(len(ops) - 1, -1, -1):  # "backward": this op's parameter gradients become final
      o = ops[i]
      if o["type"] == 0:
        K = o["kernel_size"] ** 3
        flat.g[o["w_off"]:o["w_off"] + K * o["cin"] * o["cout"]] += float(rank + 1) * (1 + i % 7)
        if o.get("has_bias"):
          flat.g[o["b_off"]:o["b_off"] + o["cout"]] += float(rank + 1)
      elif o["type"] == 1:
        flat.g[o["w_off"]:o["w_off"] + o["cout"]] += float(rank + 1) * 2
        flat.g[o["b_off"]:o["b_off"] + o["cout"]] += float(rank + 1) * 3
      for b in range(nb):
        if last[b] == i:
          n_before = red.n_launched_total
          cb(None, b)
          assert red.n_launched_total == n_before + 1, "bucket %d launched twice or not at all" % b
          if it == 0:
            fired.append(b)
    red.finish()
  assert red.n_launched_total == 2 * nb
  # expected: the sum over ranks of what each rank wrote (rank + 1 -> 1 + 2 = 3 with world 2)
  ok = True
  tot = sum(r + 1 for r in range(world))
  for i, o in enumerate(ops):
    if o["type"] == 0:
      K = o["kernel_size"] ** 3
      ok &= bool((flat.g[o["w_off"]:o["w_off"] + K * o["cin"] * o["cout"]] == tot * (1 + i % 7)).all())
    elif o["type"] == 1:
      ok &= bool((flat.g[o["w_off"]:o["w_off"] + o["cout"]] == tot * 2).all() and (flat.g[o["b_off"]:o["b_off"] + o["cout"]] == tot * 3).all())
  q.put((rank, ok, fired == sorted(fired, reverse=True), nb))
  du.destroy_process_group()


def test_reducer_through_engine_bucket_mapping_world2_gloo():
  ctx = mp.get_context("spawn")
  q = ctx.Queue()
  port = _free_port()
  procs = [ctx.Process(target=_engine_worker, args=(r, 2, port, q)) for r in range(2)]
  for p in procs:
    p.start()
  out = [q.get(timeout=180) for _ in procs]
  for p in procs:
    p.join(timeout=60)
    assert p.exitcode == 0
  assert all(ok for _, ok, _, _ in out), out
  assert all(desc for _, _, desc, _ in out), "buckets must become final from the END of the flat buffer: %r" % (out,)


def _trainer_worker(rank, world, port, q):
  """The trainer's OWN control flow for the N > 1 step (lib/ddp_trainer.py::_backward_and_step: loss.backward ->
  engine.backward(reducer) -> reducer.finish -> scaled_all_reduce_dict -> optimizer.step, and zero_grad on the next
  iteration) on gloo, with CPU stand-ins for exactly the two things that need the GPU: the network executor (a torch
  module whose backward writes the flat gradient buffer in the executor's order and reports buckets through the real
  NativeEngine._ready_args callback) and the fused SGD kernel (the same formula in torch).  pc/lib/ddp_trainer.py:96-102
  (DDP wrap), :428-435 (backward, step)."""
  os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
  import types
  from pointcontrast_amd.engine import NativeEngine
  from pointcontrast_amd.lib import distributed as du
  from pointcontrast_amd.lib.ddp_trainer import PointNCELossTrainer
  from pointcontrast_amd.lib.solver import FlatSGD
  du.init_process_group(rank, world, backend="gloo")

  def make_model():
    torch.manual_seed(7)
    return torch.nn.Sequential(torch.nn.Linear(5, 24), torch.nn.Tanh(), torch.nn.Linear(24, 24), torch.nn.Tanh(),
                               torch.nn.Linear(24, 8))

  class CpuSGD(FlatSGD):  # csrc/loss.hip::sgd_kernel restated
    def step(self, closure=None):
      g = self.param_groups[0]
      with torch.no_grad():
        grad = self.grad_scale * self.flat.g + g["weight_decay"] * self.flat.w
        self.flat.v.mul_(g["momentum"]).add_(grad)
        self.flat.w.sub_(g["lr"] * self.flat.v)

  class CpuEngine:  # stands in for NativeEngine: parameter gradients appear from the END of the flat buffer
    def __init__(self, model, flat):
      self.model, self.flat, self.fired = model, flat, []

    def forward(self, x):
      self._out = self.model(x)
      return self._out.detach().requires_grad_(True)

    def backward(self, pass_id, d_out, reducer=None):
      cb, lo_arr, nb = NativeEngine._ready_args(reducer)
      lo = [lo_arr[i] for i in range(nb)]
      grads = torch.autograd.grad(self._out, self.flat.params, d_out)
      first_param_of = {}  # bucket -> lowest parameter index inside it: the bucket is final after that parameter
      for i, off in enumerate(self.flat.offsets):
        b = max(qq for qq in range(nb) if off >= lo[qq])
        first_param_of.setdefault(b, i)
      for i in range(len(self.flat.params) - 1, -1, -1):
        self.flat.view(self.flat.g, i).add_(grads[i])
        for b, first in first_param_of.items():
          if first == i:
            self.fired.append(b)
            cb(None, b)

  def loss_fn(F):  # decomposes over the samples: the mean over a 2x batch = the mean of the two ranks' means
    return (F.pow(2).sum(1) - F[:, 0]).mean()

  model = make_model()
  flat = du.FlatParameters(model.parameters())
  red = du.GradReducer(flat, bucket_mb=0.0005)
  assert red.active and len(red.buckets) >= 3
  tr = PointNCELossTrainer.__new__(PointNCELossTrainer)
  tr.config = types.SimpleNamespace(misc={})
  tr.engine, tr.reducer, tr.world_size = CpuEngine(model, flat), red, world
  tr.optimizer = CpuSGD(flat, lr=0.1, momentum=0.8, weight_decay=1e-4, grad_scale=red.grad_scale)
  torch.manual_seed(11)
  data = [torch.randn(2 * 6, 5) for _ in range(2)]
  losses = []
  for it in range(2):  # two iterations: bucket bookkeeping, zero_grad and momentum carry over
    tr.optimizer.zero_grad()
    F = tr.engine.forward(data[it][rank * 6:(rank + 1) * 6])
    tr._feats = (F,)
    loss = loss_fn(F)
    res = tr._backward_and_step(loss, {"loss": loss.detach()})
    losses.append(float(res["loss"]))
  assert tr.engine.fired[:len(red.buckets)] == sorted(tr.engine.fired[:len(red.buckets)], reverse=True)
  assert red.n_launched_total == 2 * len(red.buckets)
  # single process, 2x batch, torch's own SGD
  ref = make_model()
  opt = torch.optim.SGD(ref.parameters(), lr=0.1, momentum=0.8, weight_decay=1e-4)
  ref_losses = []
  for it in range(2):
    opt.zero_grad()
    parts = [loss_fn(ref(data[it][r * 6:(r + 1) * 6])) for r in range(world)]
    total = sum(parts) / world
    total.backward()
    opt.step()
    ref_losses.append(float(total))
  err = max(float((p - r).abs().max()) for p, r in zip(model.parameters(), ref.parameters()))
  q.put((rank, err, losses, ref_losses, flat.w.clone().numpy()))
  du.destroy_process_group()


def test_trainer_step_control_flow_world2_gloo():
  """Post-step weights are identical on both ranks and equal the single-process step on the 2x batch; the logged loss is
  the mean over ranks."""
  ctx = mp.get_context("spawn")
  q = ctx.Queue()
  port = _free_port()
  procs = [ctx.Process(target=_trainer_worker, args=(r, 2, port, q)) for r in range(2)]
  for p in procs:
    p.start()
  out = sorted([q.get(timeout=180) for _ in procs], key=lambda t: t[0])
  for p in procs:
    p.join(timeout=60)
    assert p.exitcode == 0
  assert (out[0][4] == out[1][4]).all(), "ranks ended the step with different weights"
  for rank, err, losses, ref_losses, _ in out:
    assert err <= 1e-6, (rank, err)
    assert all(abs(a - b) <= 1e-6 * max(1.0, abs(b)) for a, b in zip(losses, ref_losses)), (losses, ref_losses)
