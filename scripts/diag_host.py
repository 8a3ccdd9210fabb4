"""Host-side diagnostics on the GPU box: launch throughput and where an iteration's host time goes."""
import os, sys, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

NT = int(os.environ.get("DIAG_THREADS", "0"))
if NT:
  torch.set_num_threads(NT)
print("torch threads", torch.get_num_threads(), "interop", torch.get_num_interop_threads(), flush=True)

from pointcontrast_amd._lib import lib, check
from pointcontrast_amd.runtime import ptr, cur_stream
import bench

dev = torch.device("cuda:0")
a = torch.zeros(8, 32, device=dev); b = torch.ones(8, 32, device=dev); y = torch.empty(8, 32, device=dev)
torch.cuda.synchronize()


def probe(label, stream_obj, n=3000):
  with torch.cuda.stream(stream_obj) if stream_obj is not None else torch.cuda.stream(torch.cuda.current_stream()):
    s = cur_stream(dev)
    t0 = time.perf_counter()
    for _ in range(n):
      lib.pcmi_add(ptr(a), 32, ptr(b), 32, 8, 32, ptr(y), 32, s)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
  print("%-28s enqueue %.2f us/launch, drained after %.2f ms more" % (label, (t1 - t0) / n * 1e6, (t2 - t1) * 1e3), flush=True)


probe("default stream", None)
side = torch.cuda.Stream()
probe("side stream", side)
probe("default stream again", None)
t0 = time.perf_counter()
for _ in range(3000):
  torch.add(a, b, out=y)
t1 = time.perf_counter(); torch.cuda.synchronize()
print("torch.add default stream      enqueue %.2f us/launch" % ((t1 - t0) / 3000 * 1e6), flush=True)

# ---- iteration phases, host clock only (no syncs inside) ---------------------------------------------
from pointcontrast_amd.lib.config import get_config
from pointcontrast_amd.lib.ddp_data_loaders import FixedBatchLoader
from pointcontrast_amd.lib import ddp_trainer
from pointcontrast_amd import functional as PF
import pointcontrast_amd.minkowski as ME
cfg = get_config(["net.model=Res16UNet34C", "misc.nceT=0.4", "misc.npos=4096", "opt.lr=0.1"])
batch = bench.get_batch(0, 4, 0.025)
tr = ddp_trainer.PointNCELossTrainer(cfg, FixedBatchLoader([batch], 4))
tr.model.train()


def one_iter(detail):
  T = [time.perf_counter()]
  def mark(): T.append(time.perf_counter())
  tr.optimizer.zero_grad(); mark()
  s0 = ME.SparseTensor(batch["sinput0_F"], coords=batch["sinput0_C"]).to(dev); mark()
  s1 = ME.SparseTensor(batch["sinput1_F"], coords=batch["sinput1_C"]).to(dev); mark()
  s0.coords_man.plan_unet(4); s1.coords_man.plan_unet(4); mark()
  F0 = tr.engine.forward(0, s0).requires_grad_(True); mark()
  F1 = tr.engine.forward(1, s1).requires_grad_(True); mark()
  qi, ki = tr.select_pairs(batch["correspondences"], 4096); mark()
  q = PF.GatherRowsFunction.apply(F0, qi.to(dev)); k = PF.GatherRowsFunction.apply(F1, ki.to(dev))
  loss = PF.NCELossFunction.apply(q, k, 0.4); mark()
  loss.backward(); mark()
  tr.engine.backward(1, F1.grad); mark()
  tr.engine.backward(0, F0.grad); mark()
  tr.optimizer.step(); mark()
  torch.cuda.synchronize(); mark()
  names = ["zero_grad", "sparse0", "sparse1", "plan", "fwd0", "fwd1", "select", "loss", "loss.bwd", "bwd1", "bwd0", "sgd", "drain"]
  d = np.diff(T) * 1e3
  if detail:
    print(" ".join("%s=%.2f" % (n, v) for n, v in zip(names, d)), "| total %.2f ms" % (T[-1] - T[0]) * 1 if False else "| total %.2f ms" % ((T[-1] - T[0]) * 1e3), flush=True)
  return d


for i in range(3):
  one_iter(False)
for i in range(6):
  one_iter(True)
import gc
gc.disable()
print("gc disabled", flush=True)
for i in range(4):
  one_iter(True)
