"""backward_pair vs two sequential backwards, per parameter tensor (debug aid; GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import pointcontrast_amd.minkowski as ME
from pointcontrast_amd.engine import NativeEngine
from pointcontrast_amd.lib import synthetic
from pointcontrast_amd.lib.config import get_config
from pointcontrast_amd.lib.distributed import FlatParameters
from pointcontrast_amd.model import load_model
name = sys.argv[1] if len(sys.argv) > 1 else "Res16UNet14"
crop, batch = (0.6, 1) if name == "Res16UNet14" else (0.9, 2)
torch.manual_seed(3)
dev = load_model(name)(3, 32, get_config([]), D=3).to("cuda:0").train()
flat = FlatParameters(dev.parameters())
eng = NativeEngine(dev, flat)
b = synthetic.make_batch(seed=6, batch_size=batch, crop=crop)
sts = [ME.SparseTensor(torch.from_numpy(b["sinput%s_F" % s]), coords=torch.from_numpy(b["sinput%s_C" % s])).to("cuda:0") for s in "01"]
names = {id(p): n for n, p in dev.named_parameters()}
for rep in range(3):
  f = eng.forward_pair(sts[0], sts[1])
  g = [torch.randn_like(x) for x in f]
  flat.zero_grad()
  eng.backward(1, g[1]); eng.backward(0, g[0])
  torch.cuda.synchronize()
  g_seq = flat.g.clone()
  eng.forward_pair(sts[0], sts[1])
  flat.zero_grad()
  eng.backward_pair(g[0], g[1])
  torch.cuda.synchronize()
  bad = []
  for i, p in enumerate(flat.params):
    a, e = flat.view(g_seq, i), flat.view(flat.g, i)
    d = float((a - e).abs().max())
    if d != 0.0:
      bad.append((d, float(a.abs().max()), names[id(p)], tuple(p.shape)))
  print("rep %d: %d of %d tensors differ" % (rep, len(bad), len(flat.params)), sorted(bad, reverse=True)[:8], flush=True)
