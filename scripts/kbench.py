"""Kernel micro-benchmark on the real maps of the bench batch: conv fwd / bwd-data / bwd-weight for the
layer shapes of Res16UNet34C at every level (HIP events on the launch stream).  Usage on the GPU box:
  python scripts/kbench.py ; PCMI_SPCONV_STREAMK=0 python scripts/kbench.py"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import pointcontrast_amd.minkowski as ME
from pointcontrast_amd._lib import lib, check
from pointcontrast_amd.runtime import ptr, cur_stream, ws_args

dev = torch.device("cuda:0")
batch = bench.get_batch(0, 4, 0.025)
st = bench.level1_tensor(batch, dev, joint=os.environ.get("KBENCH_JOINT", "1") == "1")  # default: as the training step
cm = st.coords_man
cm.plan_unet(4)
keys = [st.coords_key]
for _ in range(4):
  keys.append(cm.stride(keys[-1], 2))
print("streamk env", os.environ.get("PCMI_SPCONV_STREAMK"), "rows", [cm.size(k) for k in keys], flush=True)
tot = {}


def run(label, kmap, K, cin, cout, n_in, n_out, transpose=0):
  W = torch.randn((K, cin, cout) if K > 1 else (cin, cout), device=dev) * 0.05
  x, g = torch.randn(n_in, cin, device=dev), torch.randn(n_out, cout, device=dev)
  y, gin, gw = torch.empty(n_out, cout, device=dev), torch.empty(n_in, cin, device=dev), torch.empty_like(W)
  M = kmap.M if kmap is not None else n_in
  ws, wsb = ws_args(lib.pcmi_spconv_workspace_bytes(n_in, n_out, cin, cout, K, M), dev)
  kr = C.byref(kmap) if kmap is not None else None
  s = cur_stream(dev)
  f = lambda: check(lib.pcmi_spconv_fwd(ptr(x), cin, n_in, cin, ptr(W), cout, kr, transpose, None, ptr(y), cout, n_out, ws, wsb, s))
  b = lambda: check(lib.pcmi_spconv_bwd_data(ptr(g), cout, n_out, cout, ptr(W), cin, kr, transpose, ptr(gin), cin, n_in, ws, wsb, s))
  w = lambda: check(lib.pcmi_spconv_bwd_weight(ptr(x), cin, n_in, cin, ptr(g), cout, n_out, cout, kr, transpose, ptr(gw), None, ws, wsb, s))
  tf, tb, tw = (bench.time_kernel(k, iters=10, warm=2) * 1e3 for k in (f, b, w))
  gf = 2 * M * cin * cout * 1e-9
  print("%-34s M=%8d  fwd %7.3f ms %6.1f TF | bwd %7.3f ms %6.1f TF | wgrad %7.3f ms %6.1f TF" %
        (label, M, tf, gf / tf, tb, gf / tb, tw, gf / tw), flush=True)
  for k, v in (("fwd", tf), ("bwd", tb), ("wgrad", tw)):
    tot[k] = tot.get(k, 0) + v


shapes = {0: [(128, 96), (96, 96), (32, 32)], 1: [(32, 32), (128, 96), (96, 96)], 2: [(32, 64), (64, 64), (192, 128), (128, 128)],
          3: [(64, 128), (128, 128), (384, 256), (256, 256)], 4: [(128, 256), (256, 256)]}
only = os.environ.get("KBENCH_LEVELS")  # e.g. "0" or "0,1": only the 3^3 convs of these levels, nothing else
if only is not None:
  shapes = {l: v for l, v in shapes.items() if str(l) in only.split(",")}
for lvl, lst in shapes.items():
  m = cm.kernel_map(keys[lvl], keys[lvl], 3, 1, 3)
  n = cm.size(keys[lvl])
  for cin, cout in lst:
    run("L%d 3^3 %d->%d" % (lvl, cin, cout), m, 27, cin, cout, n, n)
for lvl, (c, cu_in, cu_out) in enumerate([] if only is not None else [(32, 96, 96), (32, 128, 96), (64, 256, 128), (128, 256, 256)]):
  m2 = cm.kernel_map(keys[lvl], keys[lvl + 1], 2, 2, 0)
  run("L%d->%d 2^3/s2 %d->%d" % (lvl, lvl + 1, c, c), m2, 8, c, c, m2.n_in, m2.n_out)
  run("L%d->%d 2^3/s2^T %d->%d" % (lvl + 1, lvl, cu_in, cu_out), m2, 8, cu_in, cu_out, m2.n_out, m2.n_in, 1)
n0 = cm.size(keys[0])
if only is None:
  run("L0 1x1 128->96", None, 1, 128, 96, n0, n0)
  run("L0 1x1 96->32", None, 1, 96, 32, n0, n0)
print("sum of the listed shapes: fwd %.2f ms, bwd %.2f ms, wgrad %.2f ms" % (tot["fwd"], tot["bwd"], tot["wgrad"]))

# sustained clocks: the same level-1 96->96 forward in consecutive blocks of 100 launches (a drop from the first
# block to the later ones is the power/thermal governor, not the kernel)
if os.environ.get("KBENCH_SUSTAINED", "1") == "1":
  m = cm.kernel_map(keys[0], keys[0], 3, 1, 3)
  W = torch.randn(27, 96, 96, device=dev) * 0.05
  x, y = torch.randn(n0, 96, device=dev), torch.empty(n0, 96, device=dev)
  ws, wsb = ws_args(lib.pcmi_spconv_workspace_bytes(n0, n0, 96, 96, 27, m.M), dev)
  s = cur_stream(dev)
  f = lambda: check(lib.pcmi_spconv_fwd(ptr(x), 96, n0, 96, ptr(W), 96, C.byref(m), 0, None, ptr(y), 96, n0, ws, wsb, s))
  print("sustained 96->96 fwd, ms per launch in blocks of 100:",
        " ".join("%.3f" % (bench.time_kernel(f, iters=100, warm=0) * 1e3) for _ in range(10)), flush=True)
