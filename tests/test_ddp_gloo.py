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
