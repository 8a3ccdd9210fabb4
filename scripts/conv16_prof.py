"""Per-wave cycle accounting of spconv16_kernel<3> on the level-1 96->96 conv of the bench batch.  Needs the profile
build:  ABLATIONS=9 bash scripts/ablate_conv16.sh ;  PCMI_LIB=pointcontrast_amd/libpcmi_abl9.so python scripts/conv16_prof.py"""
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
W = torch.randn(27, cin, cout, device=dev) * 0.05
x, y = torch.randn(n, cin, device=dev), torch.empty(n, cout, device=dev)
ws, wsb = ws_args(lib.pcmi_spconv_workspace_bytes(n, n, cin, cout, 27, m.M), dev)
s = cur_stream(dev)
f = lambda: check(lib.pcmi_spconv_fwd(ptr(x), cin, n, cin, ptr(W), cout, C.byref(m), 0, None, ptr(y), cout, n, ws, wsb, s))
t = bench.time_kernel(f, iters=50, warm=20)
torch.cuda.synchronize()
f()
torch.cuda.synchronize()
buf = np.zeros(4096 * 12, dtype=np.uint64)
lib.pcmi_debug_conv_prof.argtypes = [C.c_void_p]
check(lib.pcmi_debug_conv_prof(buf.ctypes.data))
raw = buf.reshape(4096, 12)
keep = raw[:, 6] > 0
hw = raw[keep][:, 8:12]
p = raw[:, :8].astype(np.float64)
p = p[keep]
names = ["prologue (s_idx, klist)", "pipeline fill", "issue next loads", "B preload + MFMA", "steps (count)", "store_b + barrier", "piece total", "MFMA count"]
print("conv %d->%d, M=%d rows=%d: %.3f ms per call (instrumented); %d waves reported" % (cin, cout, m.M, n, t * 1e3, len(p)))
tot = p[:, 6]
print("counter ticks per wave: mean %.0f  min %.0f  max %.0f   (kernel ~%.0f us => %.1f ticks/us)" % (tot.mean(), tot.min(), tot.max(), t * 1e6, tot.max() / (t * 1e6)))
for i, nm in enumerate(names[:6]):
  print("  %-26s mean %9.0f  (%5.1f %% of mean total)  min %9.0f  max %9.0f" % (nm, p[:, i].mean(), 100 * p[:, i].mean() / tot.mean(), p[:, i].min(), p[:, i].max()))
mf = p[:, 7]
print("  MFMAs per wave: mean %.0f min %.0f max %.0f ; MFMA-phase ticks per MFMA: %.2f" % (mf.mean(), mf.min(), mf.max(), p[:, 3].sum() / mf.sum()))
q = np.percentile(tot, [1, 10, 50, 90, 99])
print("  wave total percentiles 1/10/50/90/99: " + " ".join("%.0f" % v for v in q))
blk = tot.reshape(-1, 4).max(axis=1) if len(tot) % 4 == 0 else tot
print("  block total (max over its waves) percentiles 1/50/99: " + " ".join("%.0f" % v for v in np.percentile(blk, [1, 50, 99])))

# what explains a workgroup's duration?  least squares  total ~ c0 + c1 steps + c2 max-wave MFMAs
if len(tot) % 4 == 0:
  B = p.reshape(-1, 4, 8)
  bt = B[:, :, 6].max(axis=1)
  steps = B[:, :, 4].max(axis=1)
  mf_max = B[:, :, 7].max(axis=1)
  mf_sum = B[:, :, 7].sum(axis=1)
  A = np.stack([np.ones_like(bt), steps, mf_max], 1)
  coef, res, *_ = np.linalg.lstsq(A, bt, rcond=None)
  pred = A @ coef
  print("  block total ~ %.0f + %.0f * steps + %.1f * max-wave MFMAs ; R^2 = %.3f ; residual std %.0f (total std %.0f)" %
        (coef[0], coef[1], coef[2], 1 - ((bt - pred) ** 2).sum() / ((bt - bt.mean()) ** 2).sum(), (bt - pred).std(), bt.std()))
  print("  steps per block: mean %.1f min %.0f max %.0f ; block MFMA sum: mean %.0f min %.0f max %.0f" % (steps.mean(), steps.min(), steps.max(), mf_sum.mean(), mf_sum.min(), mf_sum.max()))
  nb = len(bt)
  xcd = np.arange(nb) & 7
  print("  mean block total by XCD (blockIdx & 7): " + " ".join("%.0f" % bt[xcd == j].mean() for j in range(8)))
  print("  mean block MFMA sum by XCD:             " + " ".join("%.0f" % mf_sum[xcd == j].mean() for j in range(8)))
  order = np.argsort(bt)
  for nm, sel in (("fastest 5 %", order[: nb // 20]), ("slowest 5 %", order[-nb // 20:])):
    print("  %s blocks: total %.0f steps %.1f max-wave MFMAs %.0f  load %.0f mfma %.0f barrier %.0f" %
          (nm, bt[sel].mean(), steps[sel].mean(), mf_max[sel].mean(), B[sel, :, 2].mean(), B[sel, :, 3].mean(), B[sel, :, 5].mean()))

# placement: HW_ID bits (gfx9): wave[3:0] simd[5:4] pipe[7:6] cu[11:8] sh[12] se[15:13]; XCC_ID[3:0]
hwid, xcc = hw[:, 1].astype(np.int64), hw[:, 2].astype(np.int64) & 15
simd, cu, sh, se = (hwid >> 4) & 3, (hwid >> 8) & 15, (hwid >> 12) & 1, (hwid >> 13) & 7
t_start = hw[:, 0].astype(np.float64)
t_end = hw[:, 3].astype(np.float64)
cuid = ((xcc * 8 + se) * 2 + sh) * 16 + cu
print("  distinct XCC %d, SE %d, SH %d, CU ids %d -> distinct CUs %d" % (len(set(xcc)), len(set(se)), len(set(sh)), len(set(cu)), len(set(cuid))))
per_cu = np.bincount(cuid)
per_cu = per_cu[per_cu > 0]
print("  waves per CU: min %d max %d ; histogram %s" % (per_cu.min(), per_cu.max(), dict(zip(*np.unique(per_cu, return_counts=True)))))
key = cuid * 4 + simd
per_simd = np.bincount(key)
per_simd = per_simd[per_simd > 0]
print("  waves per (CU, SIMD): histogram %s" % dict(zip(*np.unique(per_simd, return_counts=True))))
if len(tot) % 4 == 0:
  same = (simd.reshape(-1, 4)[:, :, None] == simd.reshape(-1, 4)[:, None, :]).sum(axis=(1, 2)) - 4
  print("  blocks with two of their waves on one SIMD: %d of %d" % ((same > 0).sum(), len(same)))
  cu_of_block = cuid.reshape(-1, 4)[:, 0]
  bt_by_cu = {}
  for c, v in zip(cu_of_block, bt):
    bt_by_cu.setdefault(c, []).append(v)
  nblk = np.array([len(v) for v in bt_by_cu.values()])
  mean_t = np.array([np.mean(v) for v in bt_by_cu.values()])
  print("  blocks per CU histogram %s" % dict(zip(*np.unique(nblk, return_counts=True))))
  for k in np.unique(nblk):
    print("    CUs with %d blocks: mean block total %.0f" % (k, mean_t[nblk == k].mean()))
# xcc clocks are not synchronised: spans per XCC
for j in sorted(set(xcc)):
  sel = xcc == j
  print("  XCC %d: %d waves, first start .. last end = %.0f ticks, start spread %.0f, mean wave total %.0f" %
        (j, sel.sum(), t_end[sel].max() - t_start[sel].min(), t_start[sel].max() - t_start[sel].min(), (t_end[sel] - t_start[sel]).mean()))
