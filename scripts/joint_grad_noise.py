"""Diagnostic: the joint two-segment pass against single passes, with the gradient of one cloud zeroed -- which cloud's
contribution to which parameter differs?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import pointcontrast_amd.minkowski as ME
from pointcontrast_amd.engine import NativeEngine
from pointcontrast_amd.lib import synthetic
from pointcontrast_amd.lib.config import get_config
from pointcontrast_amd.lib.distributed import FlatParameters
from pointcontrast_amd.model import load_model

DEV = torch.device("cuda:0")
cfg = get_config([])
torch.manual_seed(3)
dev = load_model("Res16UNet14")(3, 32, cfg, D=3).to(DEV)
dev.train()
flat = FlatParameters(dev.parameters())
eng = NativeEngine(dev, flat)
b = synthetic.make_batch(seed=6, batch_size=1, crop=0.6)
F = [torch.from_numpy(b["sinput%s_F" % s]) for s in "01"]
Cs = [torch.from_numpy(b["sinput%s_C" % s]) for s in "01"]
C1 = Cs[1].clone(); C1[:, 0] += int(Cs[0][:, 0].max()) + 1
n0 = Cs[0].shape[0]
names = {id(p): n for n, p in dev.named_parameters()}
sts = [ME.SparseTensor(F[i], coords=Cs[i]).to(DEV) for i in range(2)]
fe = [eng.forward(i, sts[i]) for i in range(2)]
g = [torch.randn_like(f) for f in fe]
single = []
for i in range(2):
  eng.forward(i, sts[i])
  flat.zero_grad(); eng.backward(i, g[i]); torch.cuda.synchronize()
  single.append(flat.g.clone())

def joint(g0, g1, Fa=None, Ca=None, Cb=None, na=None):
  Fa = F if Fa is None else Fa
  sj = ME.SparseTensor(torch.cat(Fa), coords=torch.cat([Cs[0] if Ca is None else Ca, C1 if Cb is None else Cb])).to(DEV)
  sj.coords_man.set_split(n0 if na is None else na)
  eng.forward(0, sj)
  flat.zero_grad(); eng.backward(0, torch.cat([g0, g1])); torch.cuda.synchronize()
  return flat.g.clone()

def rel(a, e):
  scale = float(a.abs().max())
  out = []
  for i, p in enumerate(flat.params):
    x, y = flat.view(a, i), flat.view(e, i)
    out.append((float((x - y).abs().max()) / max(float(x.abs().max()), 1e-4 * scale), names[id(p)]))
  out.sort(reverse=True)
  return out

z = [torch.zeros_like(t) for t in g]
for label, a, e in (("joint(g0, 0) vs pass 0 alone", single[0], joint(g[0], z[1])), ("joint(0, g1) vs pass 1 alone", single[1], joint(z[0], g[1])),
                    ("joint(g0, g1) vs sum", single[0] + single[1], joint(g[0], g[1]))):
  r = rel(a, e)
  print(label + ": " + "; ".join("%s %.2e" % (n, v) for v, n in r[:5]))

# the same cloud twice (second copy under the next batch index): its contribution as the SECOND segment
C0b = Cs[0].clone(); C0b[:, 0] += int(Cs[0][:, 0].max()) + 1
r = rel(single[0], joint(z[0], g[0], Fa=[F[0], F[0]], Ca=Cs[0], Cb=C0b, na=n0))
print("cloud 0 as second segment of (cloud 0, cloud 0): " + "; ".join("%s %.2e" % (n, v) for v, n in r[:4]))
# swapped order: cloud 1 first
C0s = Cs[0].clone(); C0s[:, 0] += int(Cs[1][:, 0].max()) + 1
r = rel(single[1], joint(g[1], z[0], Fa=[F[1], F[0]], Ca=Cs[1], Cb=C0s, na=Cs[1].shape[0]))
print("cloud 1 as FIRST segment of (cloud 1, cloud 0): " + "; ".join("%s %.2e" % (n, v) for v, n in r[:4]))
r = rel(single[0], joint(z[1], g[0], Fa=[F[1], F[0]], Ca=Cs[1], Cb=C0s, na=Cs[1].shape[0]))
print("cloud 0 as SECOND segment of (cloud 1, cloud 0): " + "; ".join("%s %.2e" % (n, v) for v, n in r[:4]))
print("rows", n0, Cs[1].shape[0])
