"""Stall-attribution probe: launches the level-1 3^3 96->96 forward / backward-data / weight-gradient kernels a few
times so that `rocprofv3 --kernel-trace --pmc <SQ / TCP counters>` passes can say where their cycles go.
  python scripts/pmc_diag.py            # under rocprofv3: the workload
  python scripts/pmc_diag.py --report gpurun_out/diag_*    # tabulates the counter_collection CSVs"""
import csv, glob, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def report(dirs):
  tab = {}
  for d in dirs:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
      for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void pcmi::", "")
        if "mfma" not in k:
          continue
        e = tab.setdefault(k, {}).setdefault(r["Counter_Name"], [0.0, 0])
        e[0] += float(r["Counter_Value"])
        e[1] += 1
  for k, cs in sorted(tab.items()):
    print(k)
    for c, (s, n) in sorted(cs.items()):
      print("    %-36s %16.0f" % (c, s / n))


if len(sys.argv) > 1 and sys.argv[1] == "--report":
  report(sys.argv[2:])
  sys.exit(0)

import ctypes as C
import torch
import bench
import pointcontrast_amd.minkowski as ME
from pointcontrast_amd._lib import lib, check
from pointcontrast_amd.runtime import ptr, cur_stream, ws_args

dev = torch.device("cuda:0")
batch = bench.get_batch(0, 4, 0.025)
st = ME.SparseTensor(batch["sinput0_F"], coords=batch["sinput0_C"]).to(dev)
cm, key = st.coords_man, st.coords_key
cm.plan_unet(4)
m = cm.kernel_map(key, key, 3, 1, 3)
N = st.F.shape[0]
for cin, cout in ((96, 96), (32, 32)):
  W = torch.randn((27, cin, cout), device=dev) * 0.05
  x, g = torch.randn(N, cin, device=dev), torch.randn(N, cout, device=dev)
  yy, gin, gw = torch.empty(N, cout, device=dev), torch.empty(N, cin, device=dev), torch.empty_like(W)
  ws, wsb = ws_args(lib.pcmi_spconv_workspace_bytes(N, N, cin, cout, 27, m.M), dev)
  s = cur_stream(dev)
  for _ in range(3):
    check(lib.pcmi_spconv_fwd(ptr(x), cin, N, cin, ptr(W), cout, C.byref(m), 0, None, ptr(yy), cout, N, ws, wsb, s))
    check(lib.pcmi_spconv_bwd_weight(ptr(x), cin, N, cin, ptr(g), cout, N, cout, C.byref(m), 0, ptr(gw), None, ws, wsb, s))
torch.cuda.synchronize()
print("done")
