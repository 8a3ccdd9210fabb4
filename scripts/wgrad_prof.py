"""Per-wave cycle accounting of wgrad_mfma_kernel<3,3> on the level-1 96->96 conv of the bench batch.  Needs the profile
build:  ABLATIONS=9 bash scripts/ablate_conv16.sh ;  PCMI_LIB=pointcontrast_amd/libpcmi_abl9.so python scripts/wgrad_prof.py"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
import pointcontrast_amd.minkowski as ME
from pointcontrast_amd._lib import lib, check
from pointcontrast_amd.runtime import ptr, cur_stream, ws_args

dev = torch.device("cuda:0")
batch = bench.get_batch(0, 4, 0.025)
st = bench.level1_tensor(batch, dev, joint=os.environ.get("PROBE_JOINT", "1") == "1")  # as the training step launches it
cm = st.coords_man
cm.plan_unet(4)
key = st.coords_key
n = cm.size(key)
m = cm.kernel_map(key, key, 3, 1, 3)
cin = cout = int(os.environ.get("PROF_C", "96"))
x, g = torch.randn(n, cin, device=dev), torch.randn(n, cout, device=dev)
gw = torch.empty(27, cin, cout, device=dev)
ws, wsb = ws_args(lib.pcmi_spconv_workspace_bytes(n, n, cin, cout, 27, m.M), dev)
s = cur_stream(dev)
f = lambda: check(lib.pcmi_spconv_bwd_weight(ptr(x), cin, n, cin, ptr(g), cout, n, cout, C.byref(m), 0, ptr(gw), None, ws, wsb, s))
t = bench.time_kernel(f, iters=50, warm=20)
torch.cuda.synchronize()
f()
torch.cuda.synchronize()
buf = np.zeros(4096 * 8, dtype=np.uint64)
lib.pcmi_debug_wgrad_prof.argtypes = [C.c_void_p]
check(lib.pcmi_debug_wgrad_prof(buf.ctypes.data))
raw = buf.reshape(4096, 8)
p = raw[raw[:, 0] > 0].astype(np.float64)
print("wgrad %d->%d, M=%d: %.3f ms per call (kernel + slab reduction; instrumented); %d waves reported" % (cin, cout, m.M, t * 1e3, len(p)))
tot = p[:, 0]
print("ticks per wave: mean %.0f min %.0f max %.0f (%.0f ticks/us if the longest wave spans the call)" % (tot.mean(), tot.min(), tot.max(), tot.max() / (t * 1e6)))
for i, nm in ((1, "prologue (chunk look-up)"), (2, "main loop"), (3, "wave reduction + write-out")):
  print("  %-28s mean %9.0f (%5.1f %%) min %9.0f max %9.0f" % (nm, p[:, i].mean(), 100 * p[:, i].mean() / tot.mean(), p[:, i].min(), p[:, i].max()))
gr = p[:, 4]
print("  64-pair groups per wave: mean %.1f min %.0f max %.0f; main-loop ticks per group %.0f (= %.1f per MFMA; 32 steps x 9 MFMAs x 64 cycles = 18432 alone on the pipe, x2 waves/SIMD)" %
      (gr.mean(), gr.min(), gr.max(), p[:, 2].sum() / gr.sum(), p[:, 2].sum() / gr.sum() / 288))
print("  pairs per chunk: mean %.0f min %.0f max %.0f" % (p[:, 7].mean(), p[:, 7].min(), p[:, 7].max()))
print("  wave total percentiles 1/10/50/90/99: " + " ".join("%.0f" % v for v in np.percentile(tot, [1, 10, 50, 90, 99])))
hwid, xcc = raw[raw[:, 0] > 0][:, 5].astype(np.int64), raw[raw[:, 0] > 0][:, 6].astype(np.int64) & 15
cu, se = (hwid >> 8) & 15, (hwid >> 13) & 7
cuid = (xcc * 8 + se) * 16 + cu
per_cu = np.bincount(cuid)
per_cu = per_cu[per_cu > 0]
print("  waves per CU histogram %s" % dict(zip(*np.unique(per_cu, return_counts=True))))
