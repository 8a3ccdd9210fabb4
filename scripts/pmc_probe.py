"""Runs the dominant kernels a few times each so that rocprofv3 --pmc passes can attribute HBM traffic
(FETCH_SIZE / WRITE_SIZE) and MFMA activity to them.  The first kernels are a calibration pair with a
known byte count in the same access width (float4 per lane): pcmi_add on 3 x 256 MiB buffers."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import pointcontrast_amd.minkowski as ME
from pointcontrast_amd._lib import lib, check
from pointcontrast_amd.runtime import ptr, cur_stream, ws_args

dev = torch.device("cuda:0")
REP = 3
# calibration: y = a + b, 2 x 256 MiB read + 256 MiB written per launch (well past the 256 MiB L3)
n, c = 1 << 19, 128
a, b, y = (torch.randn(n, c, device=dev) for _ in range(3))
for _ in range(REP):
  check(lib.pcmi_add(ptr(a), c, ptr(b), c, n, c, ptr(y), c, cur_stream(dev)))
print("CAL eltwise_kernel<2> read_bytes=%d write_bytes=%d" % (2 * n * c * 4, n * c * 4))
del a, b, y

batch = bench.get_batch(0, 4, 0.025)
st = bench.level1_tensor(batch, dev, joint=os.environ.get("PROBE_JOINT", "1") == "1")  # as the training step launches it
cm, key = st.coords_man, st.coords_key
m = cm.kernel_map(key, key, 3, 1, 3)
N = st.F.shape[0]


SEG = [0]
_mk = torch.zeros(64, 4, device=dev)


def segment(label):
  """Marks the start of a probe segment with a tiny eltwise launch (one workgroup): scripts/pmc_summary.py cuts the
  dispatch sequence at these markers, so that launches of the SAME kernel on different shapes (wgrad_x3p_kernel<3,3,4>
  at level 1 and at level 2 -- same name, same grid) are separate rows instead of one averaged row."""
  check(lib.pcmi_add(ptr(_mk), 4, ptr(_mk), 4, 64, 4, ptr(_mk), 4, cur_stream(dev)))
  print("SEG %d %s" % (SEG[0], label))
  SEG[0] += 1


def conv(cin, cout, kmap, K, n_in, n_out, modes="fbw", tag=""):
  segment("%s%d->%d K=%d rows=%d modes=%s" % (tag, cin, cout, K, n_out, modes))
  W = torch.randn((K, cin, cout), device=dev) * 0.05
  x, g = torch.randn(n_in, cin, device=dev), torch.randn(n_out, cout, device=dev)
  yy, gin, gw = torch.empty(n_out, cout, device=dev), torch.empty(n_in, cin, device=dev), torch.empty_like(W)
  ws, wsb = ws_args(lib.pcmi_spconv_workspace_bytes(n_in, n_out, cin, cout, K, kmap.M), dev)
  s = cur_stream(dev)
  for _ in range(REP):
    if "f" in modes:
      check(lib.pcmi_spconv_fwd(ptr(x), cin, n_in, cin, ptr(W), cout, C.byref(kmap), 0, None, ptr(yy), cout, n_out, ws, wsb, s))
    if "b" in modes:
      check(lib.pcmi_spconv_bwd_data(ptr(g), cout, n_out, cout, ptr(W), cin, C.byref(kmap), 0, ptr(gin), cin, n_in, ws, wsb, s))
    if "w" in modes:
      check(lib.pcmi_spconv_bwd_weight(ptr(x), cin, n_in, cin, ptr(g), cout, n_out, cout, C.byref(kmap), 0, ptr(gw), None, ws, wsb, s))
  M = kmap.M
  print("ALGO conv %s%d->%d K=%d pairs=%d fwd_bytes=%d bwd_bytes=%d wgrad_bytes=%d flops=%d" %
        (tag, cin, cout, K, M, M * (4 * cin + 8) + n_out * 4 * cout + 4 * K * cin * cout,
         M * (4 * cout + 8) + n_in * 4 * cin + 4 * K * cin * cout, M * 4 * (cin + cout) + 8 * M + 4 * K * cin * cout,
         2 * M * cin * cout))


if os.environ.get("PMC_PROBE_SET") == "gathers":
  # the three HBM-priced entries of bench.py's kernels[] that sit under 0.40 of the HBM peak:
  # how many bytes do they actually move?  (forward only; REP launches each, distinct kernel names)
  ck = cm.stride(key, 2)
  m2 = cm.kernel_map(key, ck, 2, 2, 0)
  conv(32, 32, m2, 8, m2.n_in, m2.n_out, modes="f", tag="gather 2^3/s2 ")
  m1 = cm.kernel_map(ck, ck, 3, 1, 3)
  n2 = cm.size(ck)
  x = torch.randn(n2, 32, device=dev)  # (the level-2 3^3 launch shares its kernel with the one above: told apart by launch order)
  conv(32, 32, m1, 27, n2, n2, modes="f", tag="level2 3^3 ")
  conv(3, 32, cm.kernel_map(key, key, 3, 1, 0), 27, N, N, modes="f", tag="stem ")
  torch.cuda.synchronize()
  print("done")
  sys.exit(0)
conv(96, 96, m, 27, N, N)
# the stride-2 level (~40k rows): its weight gradients take the tile-stationary split-precision kernel (wgrad_x3t_kernel)
ck = cm.stride(key, 2)
m1 = cm.kernel_map(ck, ck, 3, 1, 3)
print("LEVEL2 rows=%d pairs=%d" % (cm.size(ck), m1.M))
conv(96, 96, m1, 27, cm.size(ck), cm.size(ck), tag="level2:")  # (its own segment: the kernels share their names with level 1)
conv(128, 96, m, 27, N, N, tag="in128:")  # the other level-1 shape of the decoder (block8.0.conv1)
if os.environ.get("PMC_PROBE_ONLY") != "96":
  conv(32, 32, m, 27, N, N)
torch.cuda.synchronize()
print("done")
