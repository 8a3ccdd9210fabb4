"""Debug probe of bn_small_fwd / bwd_kernel: errors against torch (float64), per column / row when wrong, determinism."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from pointcontrast_amd import functional as PF
DEV = "cuda:0"
for n, c in ((65, 16), (77, 32), (300, 64), (768, 256), (769, 128), (1350, 128), (1536, 256)):
  torch.manual_seed(0)
  x = (torch.randn(n, c) * 2.0 + 0.7)
  res, gy = torch.randn(n, c), torch.randn(n, c)
  bn = torch.nn.BatchNorm1d(c, eps=1e-5, momentum=0.05).double()
  with torch.no_grad():
    bn.weight.uniform_(0.5, 1.5); bn.bias.uniform_(-0.5, 0.5)
  x64, r64 = x.double().requires_grad_(True), res.double().requires_grad_(True)
  yr = torch.relu(bn(x64) + r64)
  yr.backward(gy.double())
  outs = []
  for rep in range(2):
    xd, rd = x.to(DEV).requires_grad_(True), res.to(DEV).requires_grad_(True)
    g, b = bn.weight.detach().float().to(DEV).requires_grad_(True), bn.bias.detach().float().to(DEV).requires_grad_(True)
    rm, rv = torch.zeros(c, device=DEV), torch.ones(c, device=DEV)
    y = PF.BatchNormFunction.apply(xd, g, b, rm, rv, 0.05, 1e-5, rd, True)
    y.backward(gy.to(DEV))
    torch.cuda.synchronize()
    outs.append([t.detach().cpu() for t in (y, xd.grad, rd.grad, g.grad, b.grad)])
  refs = [yr.detach(), x64.grad, r64.grad, bn.weight.grad, bn.bias.grad]
  same = all(torch.equal(a, b_) for a, b_ in zip(outs[0], outs[1]))
  line = []
  for name, got, ref in zip(("y", "dx", "dres", "dgamma", "dbeta"), outs[0], refs):
    e = float((got.double() - ref).abs().max() / ref.abs().max())
    line.append("%s %.1e" % (name, e))
    if e > 1e-4 and got.dim() == 2:
      err = (got.double() - ref).abs()
      bad_rows = (err.max(1).values > 1e-3 * float(ref.abs().max())).nonzero().flatten().tolist()
      bad_cols = (err.max(0).values > 1e-3 * float(ref.abs().max())).nonzero().flatten().tolist()
      print("   %s wrong: %d rows %s..., %d cols %s..." % (name, len(bad_rows), bad_rows[:12], len(bad_cols), bad_cols[:12]))
  print("n %4d c %3d: %s | identical over two calls: %s" % (n, c, "  ".join(line), same))
