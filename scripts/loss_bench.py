"""Times the loss block of one iteration on the GPU (HIP events on the current stream): row gather, PointInfoNCE forward,
backward, scatter-add of the row gradients -- python scripts/loss_bench.py [n] [c].  PCMI_NCE_X3=0 selects the fp32
VALU kernels of loss.hip for an A/B."""
import sys
import time

import torch

from pointcontrast_amd import functional as PF


def timed(fn, reps=50):
  for _ in range(5):
    fn()
  torch.cuda.synchronize()
  a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  a.record()
  for _ in range(reps):
    fn()
  b.record()
  torch.cuda.synchronize()
  return a.elapsed_time(b) / reps * 1e3  # us


def main():
  n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
  c = int(sys.argv[2]) if len(sys.argv) > 2 else 32
  dev = torch.device("cuda:0")
  torch.manual_seed(0)
  F0 = torch.nn.functional.normalize(torch.randn(87380, c), dim=1).to(dev).requires_grad_(True)
  F1 = torch.nn.functional.normalize(torch.randn(87371, c), dim=1).to(dev).requires_grad_(True)
  qi = torch.randperm(87380)[:n].sort().values.to(dev)
  ki = torch.randint(0, 87371, (n,)).to(dev)
  out = {}

  def fwd():
    out["q"] = PF.GatherRowsFunction.apply(F0, qi)
    out["k"] = PF.GatherRowsFunction.apply(F1, ki)
    out["loss"] = PF.NCELossFunction.apply(out["q"], out["k"], 0.4)

  def fwd_bwd():
    fwd()
    out["loss"].backward()
    F0.grad = F1.grad = None

  t_f, t_fb = timed(fwd), timed(fwd_bwd)
  q, k = out["q"].detach(), out["k"].detach()
  t_nce_f = timed(lambda: PF.NCELossFunction.apply(q, k, 0.4))
  qg, kg = q.clone().requires_grad_(True), k.clone().requires_grad_(True)

  def nce_fb():
    PF.NCELossFunction.apply(qg, kg, 0.4).backward()
    qg.grad = kg.grad = None

  t_nce_fb = timed(nce_fb)
  g = torch.randn(n, c, device=dev)

  def scatter():
    out["k"].backward(g, retain_graph=True)
    F1.grad = None

  fwd()
  t_sc = timed(scatter)
  print("n=%d c=%d | gather+nce fwd %.1f us | + backward + scatter %.1f us | nce fwd alone %.1f | nce fwd+bwd alone %.1f | "
        "one scatter (zero fill + kernel) %.1f" % (n, c, t_f, t_fb, t_nce_f, t_nce_fb, t_sc))


if __name__ == "__main__":
  main()
