"""A/B timing of the split-precision conv kernel (csrc/spconv_x3.hip, PCMI_CONV16_X3) against the fp32-MFMA kernel on the
real maps of the bench batch: 3^3 conv forward / backward-data of the matrix-bound layer shapes of levels 1 and 2, as the
training step launches them (the pair as one two-segment tensor).  HIP events on the launch stream.  Usage on the GPU box:
  python scripts/x3_bench.py > gpurun_out/x3_bench.txt"""
import ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from pointcontrast_amd._lib import lib, check
from pointcontrast_amd.runtime import ptr, cur_stream, ws_args

dev = torch.device("cuda:0")
batch = bench.get_batch(0, 4, 0.025)
st = bench.level1_tensor(batch, dev, joint=os.environ.get("KBENCH_JOINT", "1") == "1")
cm = st.coords_man
cm.plan_unet(4)
keys = [st.coords_key, cm.stride(st.coords_key, 2)]
print("rows", [cm.size(k) for k in keys], flush=True)
MODES = [("fp32", {"PCMI_CONV16_X3": "0"}), ("x3 dma", {"PCMI_CONV16_X3": "1", "PCMI_X3_DMA": "1"}),
         ("x3 reg", {"PCMI_CONV16_X3": "1", "PCMI_X3_DMA": "0"}),
         ("x3 nt2", {"PCMI_CONV16_X3": "1", "PCMI_X3_DMA": "1", "PCMI_X3_MAXNT": "2"})]  # 128-wide outputs as 2 x 64
results = []


def run(label, kmap, cin, cout, n):
  W = torch.randn(27, cin, cout, device=dev) * 0.05
  x, g = torch.randn(n, cin, device=dev), torch.randn(n, cout, device=dev)
  y, gin = torch.empty(n, cout, device=dev), torch.empty(n, cin, device=dev)
  ws, wsb = ws_args(lib.pcmi_spconv_workspace_bytes(n, n, cin, cout, 27, kmap.M), dev)
  kr = C.byref(kmap)
  s = cur_stream(dev)
  f = lambda: check(lib.pcmi_spconv_fwd(ptr(x), cin, n, cin, ptr(W), cout, kr, 0, None, ptr(y), cout, n, ws, wsb, s))
  b = lambda: check(lib.pcmi_spconv_bwd_data(ptr(g), cout, n, cout, ptr(W), cin, kr, 0, ptr(gin), cin, n, ws, wsb, s))
  gf = 2 * kmap.M * cin * cout * 1e-9
  row = {"shape": label, "pairs": int(kmap.M), "gflop": round(gf, 3)}
  ref = {}
  for name, env in MODES:
    if name == "x3 nt2" and cout % 128 != 0:
      continue
    os.environ.update(env)
    tf, tb = (bench.time_kernel(k, iters=20, warm=3) * 1e3 for k in (f, b))
    if name == "fp32":
      ref = {"y": y.clone(), "gin": gin.clone()}
      err = ""
    else:
      ef = float((y - ref["y"]).abs().max() / ref["y"].abs().max())
      eb = float((gin - ref["gin"]).abs().max() / ref["gin"].abs().max())
      err = "  max|diff|/max vs fp32: fwd %.1e bwd %.1e" % (ef, eb)
      row[name + " err"] = [ef, eb]
    row[name] = {"fwd_ms": round(tf, 4), "bwd_ms": round(tb, 4), "fwd_tflops": round(gf / tf, 1), "bwd_tflops": round(gf / tb, 1)}
    print("%-22s %-7s fwd %7.3f ms %6.1f TF | bwd %7.3f ms %6.1f TF%s" % (label, name, tf, gf / tf, tb, gf / tb, err), flush=True)
  os.environ["PCMI_CONV16_X3"] = "0"
  os.environ.pop("PCMI_X3_MAXNT", None)
  results.append(row)


shapes = {0: [(96, 96), (128, 96)], 1: [(64, 64), (128, 128), (192, 128)]}
for lvl, lst in shapes.items():
  m = cm.kernel_map(keys[lvl], keys[lvl], 3, 1, 3)
  n = cm.size(keys[lvl])
  for cin, cout in lst:
    run("L%d 3^3 %d->%d" % (lvl + 1, cin, cout), m, cin, cout, n)
print(json.dumps(results))
