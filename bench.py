"""Benchmark of the PointContrast pre-training hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W
  N > 1 either way: under a launcher (python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr
  127.0.0.1 --master-port P bench.py --gpus N ...: RANK / WORLD_SIZE come from the environment), or bare -- without
  WORLD_SIZE in the environment bench.py starts its N ranks itself (pointcontrast_amd/lib/multiprocessing.py).

Metric (BASELINE.json): scene-pairs/sec of the full training iteration -- 2 forwards of
Res16UNet34C, PointInfoNCE (or HardestContrastive with --loss hardest), backward, gradient
all-reduce (N > 1), SGD step -- on seeded synthetic ScanNet-shaped pairs (2.5 cm voxels,
B = 4 pairs per GPU, ~85k voxels per forward, 4096 correspondences), inputs staged on the host
in the reference's batch format before the timed region starts (coordinates / features are
uploaded inside the step exactly as the reference's ME.SparseTensor(...).to(device) does).
One process per GPU, weak scaling: every rank has its own 4 pairs (seed = rank).

Rank 0 prints ONE JSON line; besides the contract keys it carries
  roofline     : the dominant kernel (fp32-MFMA gather-GEMM of the level-1 96->96 block conv)
                 timed live with HIP events, algorithmic FLOPs / launch time vs the 157.3 TFLOP/s
                 fp32 matrix peak (MI355X_MICROARCH.md); `kernels` lists the other hot kernels
                 incl. the HBM-bound ones against the 8 TB/s peak
  cpu_baseline : the CPU oracle's identical iteration on a bounded sample (1 pair), N=1 only.
"""
import argparse
import json
import os

# compute, plan, pair-forward, weight-gradient and RCCL streams should each get a hardware queue (ROCm default: 4)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
  sys.path.insert(0, ROOT)

PEAK_FP32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md "Peak FP32 (matrix)"
PEAK_HBM_GBS = 8000.0          # MI355X_MICROARCH.md "HBM3E peak BW" (spec)
PEAK_BF16_MFMA_TFLOPS = 2500.0  # MI355X_MICROARCH.md "Peak BF16/FP16 MFMA", dense


BATCH_KEYS = ["sinput0_C", "sinput0_F", "sinput1_C", "sinput1_F", "correspondences"]


def _batch_path(seed, batch_size, voxel_size):
  cache = os.path.join(os.environ.get("TMPDIR", "/tmp"), "pcmi_bench_cache")
  os.makedirs(cache, exist_ok=True)
  return os.path.join(cache, "s%d_b%d_v%g.npz" % (seed, batch_size, voxel_size))


def generate_batch(seed, batch_size, voxel_size):
  """Generates the seeded synthetic batch into the local cache (numpy only: safe in a spawned helper process)."""
  path = _batch_path(seed, batch_size, voxel_size)
  if not os.path.exists(path):
    from pointcontrast_amd.lib import synthetic
    d = synthetic.make_batch(seed=seed, batch_size=batch_size, voxel_size=voxel_size)
    tmp = path + ".tmp%d.npz" % os.getpid()
    np.savez(tmp, **{k: d[k] for k in BATCH_KEYS})
    os.replace(tmp, path)
  return path


def get_batch(seed, batch_size, voxel_size):
  """Seeded synthetic batch in the reference's collate format, cached on local disk."""
  import torch
  z = np.load(generate_batch(seed, batch_size, voxel_size))
  return {k: torch.from_numpy(np.ascontiguousarray(z[k])) for k in BATCH_KEYS}


def get_batches(seeds, batch_size, voxel_size):
  """Several seeds: the missing ones are generated side by side in spawned helper processes (~10 s of numpy each)."""
  missing = [sd for sd in seeds if not os.path.exists(_batch_path(sd, batch_size, voxel_size))]
  if len(missing) > 1:
    import multiprocessing as mp
    with mp.get_context("spawn").Pool(min(len(missing), 4)) as pool:
      pool.starmap(generate_batch, [(sd, batch_size, voxel_size) for sd in missing])
  return [get_batch(sd, batch_size, voxel_size) for sd in seeds]


def level1_tensor(batch, device, joint=True):
  """The full-resolution sparse tensor as the training step launches on it: with misc.joint_pair (default) BOTH clouds
  of the pair as one two-segment tensor, else cloud 0."""
  import torch
  import pointcontrast_amd.minkowski as ME
  if not joint:
    return ME.SparseTensor(batch["sinput0_F"], coords=batch["sinput0_C"]).to(device)
  C0, C1 = batch["sinput0_C"], batch["sinput1_C"].clone()
  C1[:, 0] += int(C0[:, 0].max()) + 1
  st = ME.SparseTensor(torch.cat([batch["sinput0_F"], batch["sinput1_F"]]), coords=torch.cat([C0, C1])).to(device)
  st.coords_man.set_split(C0.shape[0])
  return st


def _split_precision():
  from pointcontrast_amd._lib import lib
  return bool(lib.pcmi_spconv_split_precision())


def conv_work(model):
  """Algorithmic FLOPs / bytes of one forward from the real per-layer pair counts
  (flops = 2*M*Cin*Cout; bytes = M*(4*Cin + 8) + N_out*4*Cout + 4*K*Cin*Cout)."""
  from pointcontrast_amd.minkowski import _ConvBase
  flops = bytes_ = 0
  rows = []
  for name, m in model.named_modules():
    if isinstance(m, _ConvBase) and hasattr(m, "last_work"):
      M, n_in, n_out, K = m.last_work
      f = 2 * M * m.in_channels * m.out_channels
      b = M * (4 * m.in_channels + 8) + n_out * 4 * m.out_channels + 4 * K * m.in_channels * m.out_channels
      flops += f
      bytes_ += b
      rows.append((name, K, m.in_channels, m.out_channels, int(M), int(n_out), f, b))
  return flops, bytes_, rows


def time_kernel(fn, iters=20, warm=3, warm_ms=40.0):
  """Average duration of fn() over `iters` back-to-back launches (HIP events on the launch stream), after `warm`
  launches and at least `warm_ms` of continuous work: the GPU has been idle while the host prepared the operands, and
  the first launches after an idle period ran up to 10 % slower than the same launch in a busy stream (kbench.py: the
  shape measured first was always the slow one), which is not the state the training step runs the kernel in."""
  import time
  import torch
  for _ in range(warm):
    fn()
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  while (time.perf_counter() - t0) * 1e3 < warm_ms:
    for _ in range(8):
      fn()
    torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()  # libpcmi launches on torch's current stream, the stream these events are recorded on
  for _ in range(iters):
    fn()
  e1.record()
  e1.synchronize()
  return e0.elapsed_time(e1) * 1e-3 / iters


def kernel_rooflines(batch, device, joint=True):
  """Times the hot kernels in isolation on the level-1 map the training step uses (HIP events): the pair as one
  two-segment tensor with misc.joint_pair, else cloud 0."""
  import torch
  import pointcontrast_amd.minkowski as ME
  from pointcontrast_amd import functional as PF
  from pointcontrast_amd._lib import lib, check
  from pointcontrast_amd.runtime import ptr, cur_stream, ws_args
  import ctypes as C
  st = level1_tensor(batch, device, joint)
  cm, key = st.coords_man, st.coords_key
  n = st.F.shape[0]
  m = cm.kernel_map(key, key, 3, 1, 3)
  out = []
  x3_on = bool(lib.pcmi_spconv_split_precision())

  def conv_entry(label, cin, cout, kmap, K, n_in, n_out, transpose=False, mode="fwd"):
    W = torch.randn((K, cin, cout) if K > 1 else (cin, cout), device=device) * 0.05
    x = torch.randn(n_in, cin, device=device)
    g = torch.randn(n_out, cout, device=device)
    M = kmap.M if kmap is not None else n_in
    y = torch.empty(n_out, cout, device=device)
    gin = torch.empty(n_in, cin, device=device)
    gw = torch.empty_like(W)
    ws, wsb = ws_args(lib.pcmi_spconv_workspace_bytes(n_in, n_out, cin, cout, K, M), device)
    kref = C.byref(kmap) if kmap is not None else None
    s = cur_stream(device)
    if mode == "fwd":
      fn = lambda: check(lib.pcmi_spconv_fwd(ptr(x), cin, n_in, cin, ptr(W), cout, kref, int(transpose), None, ptr(y),
                                             cout, n_out, ws, wsb, s))
      byts = M * (4 * cin + 8) + n_out * 4 * cout + 4 * K * cin * cout
    elif mode == "bwd_data":
      fn = lambda: check(lib.pcmi_spconv_bwd_data(ptr(g), cout, n_out, cout, ptr(W), cin, kref, int(transpose), ptr(gin),
                                                  cin, n_in, ws, wsb, s))
      byts = M * (4 * cout + 8) + n_in * 4 * cin + 4 * K * cin * cout
    else:
      fn = lambda: check(lib.pcmi_spconv_bwd_weight(ptr(x), cin, n_in, cin, ptr(g), cout, n_out, cout, kref,
                                                    int(transpose), ptr(gw), None, ws, wsb, s))
      byts = M * 4 * (cin + cout) + 8 * M + 4 * K * cin * cout
    print("[bench] timing %s" % label, file=sys.stderr, flush=True)
    t = time_kernel(fn)
    flops = 2 * M * cin * cout
    intensity = flops / byts
    bound = "mfma" if intensity > PEAK_FP32_MFMA_TFLOPS * 1e12 / (PEAK_HBM_GBS * 1e9) else "hbm"
    ent = {"kernel": label, "ms": round(t * 1e3, 4), "pairs": int(M), "gflop": round(flops * 1e-9, 3),
           "algo_mb": round(byts * 1e-6, 2), "bound": bound}
    # which launches the split-precision kernel takes (csrc/spconv.hip: run_gathered): 3^3 / 2^3 table launches of the
    # 16-row kernel with >= 64 channels on both sides -- forward and backward-data, not the weight gradients
    split = (x3_on and mode in ("fwd", "bwd_data") and kmap is not None and min(cin, cout) >= 64 and not transpose
             and min(n_in, n_out) >= int(os.environ.get("PCMI_CONV16", "512")) > 0)
    # ... and the weight gradients of the 3^3 / stride-1 convolutions the tile-stationary kernel takes
    # (csrc/spconv_wgrad_x3.hip: wgrad_x3t_eligible)
    x3t_rows = int(os.environ.get("PCMI_WGRAD_X3T", "8192"))
    tw = lambda c: c % 96 == 0 or c % 64 == 0
    split = split or (mode == "bwd_weight" and kmap is not None and K == 27 and not transpose and min(cin, cout) >= 64
                      and tw(cin) and tw(cout) and 0 < x3t_rows <= n_out)
    if bound == "mfma" and split:
      # fp32 products from six bf16 MFMAs: the matrix-pipe bound of THIS arithmetic is the dense bf16 peak / 6; the
      # fraction of the fp32 instruction's own peak is given beside it (it can exceed 1: that is the point of the split)
      ent.update(achieved=round(flops / t * 1e-12, 3), peak=round(PEAK_BF16_MFMA_TFLOPS / 6, 1), unit="TFLOP/s",
                 arithmetic="fp32 operands as 3 bf16 terms, 6 v_mfma_f32_16x16x32_bf16 per fp32-equivalent tile, fp32 accumulate",
                 peak_fp32_mfma=PEAK_FP32_MFMA_TFLOPS, frac_of_fp32_mfma_peak=round(flops / t * 1e-12 / PEAK_FP32_MFMA_TFLOPS, 4))
    elif bound == "mfma":
      ent.update(achieved=round(flops / t * 1e-12, 3), peak=PEAK_FP32_MFMA_TFLOPS, unit="TFLOP/s")
    else:
      ent.update(achieved=round(byts / t * 1e-9, 1), peak=PEAK_HBM_GBS, unit="GB/s")
    ent["frac"] = round(ent["achieved"] / ent["peak"], 4)
    out.append(ent)
    return ent

  kname = "spconv16x (bf16x3 split)" if x3_on else "spconv16p (fp32 MFMA)"
  dominant = conv_entry("%s fwd 3^3 96->96 @level1 (%d rows)" % (kname, n), 96, 96, m, 27, n, n)
  conv_entry("%s bwd_data 3^3 96->96 @level1" % kname, 96, 96, m, 27, n, n, mode="bwd_data")
  conv_entry("wgrad_x3p (bf16x3 split, tile-stationary, staging / multiplying waves) 3^3 96->96 @level1", 96, 96, m, 27, n, n, mode="bwd_weight")
  conv_entry("%s fwd 3^3 128->96 @level1" % kname, 128, 96, m, 27, n, n)
  ck = cm.stride(key, 2)
  m2 = cm.kernel_map(key, ck, 2, 2, 0)
  r32 = int(os.environ.get("PCMI_CONV32R", "8192"))  # csrc/spconv32r.hip: conv32r_min_rows
  k32 = "spconv32r (weights resident in LDS)" if 0 < r32 <= m2.n_out else "spconv16p (128-row tiles)"
  conv_entry("%s fwd 2^3/s2 32->32 (gather)" % k32, 32, 32, m2, 8, m2.n_in, m2.n_out)
  conv_entry("spconv_mfma pair fwd 2^3/s2^T 96->96 (scatter)", 96, 96, m2, 8, m2.n_out, m2.n_in, transpose=True)
  m1 = cm.kernel_map(ck, ck, 3, 1, 3)
  conv_entry("%s fwd 3^3 32->32 @level2 (%d rows)" % (k32, m2.n_out), 32, 32, m1, 27, m2.n_out, m2.n_out)
  conv_entry("wgrad_x3p (bf16x3 split, tile-stationary, staging / multiplying waves) 3^3 96->96 @level2 (%d rows)" % m2.n_out, 96, 96, m1, 27, m2.n_out, m2.n_out,
             mode="bwd_weight")
  conv_entry("stem32_fwd 3^3 3->32 @level1 (lane per row)", 3, 32, cm.kernel_map(key, key, 3, 1, 0), 27, n, n)
  # PointInfoNCE block, n = 4096 positives, 32 channels: logits GEMM forward, (recompute + contraction) x 2 backward
  qn = torch.nn.functional.normalize(torch.randn(4096, 32, device=device), dim=1).requires_grad_(True)
  kn = torch.nn.functional.normalize(torch.randn(4096, 32, device=device), dim=1).requires_grad_(True)

  # The two C-ABI calls directly, on preallocated buffers.  (Through autograd -- Function.apply + .backward(), two allocator
  # round trips and ~10 small torch ops per iteration on the host -- the 4 launches of ~20 us each were HOST-bound: the same
  # build read 0.088 ms in one process and 0.2215 ms in another)
  qd, kd = qn.detach().contiguous(), kn.detach().contiguous()
  lse = torch.empty(4096, dtype=torch.float32, device=device)
  lossv = torch.empty((), dtype=torch.float32, device=device)
  gone = torch.ones((), dtype=torch.float32, device=device)
  dq, dk = torch.empty_like(qd), torch.empty_like(kd)
  nws, nwsb = ws_args(lib.pcmi_nce_workspace_bytes(4096, 32), device)
  sn = cur_stream(device)

  def nce_step():
    check(lib.pcmi_nce_fwd(ptr(qd), ptr(kd), 4096, 32, 1.0 / 0.4, ptr(lse), ptr(lossv), nws, nwsb, sn))
    check(lib.pcmi_nce_bwd(ptr(qd), ptr(kd), ptr(lse), 4096, 32, 1.0 / 0.4, ptr(gone), ptr(dq), ptr(dk), nws, nwsb, sn))

  print("[bench] timing nce", file=sys.stderr, flush=True)
  t = time_kernel(nce_step, iters=100)
  fl = 5 * 2 * 4096 * 4096 * 32
  # (pack + forward, pack + backward: 4 launches; the time is launch / cross-workgroup hand-off latency, not matrix work:
  #  profiles/r04n_nce_component_removal.txt)
  x3 = os.environ.get("PCMI_NCE_X3", "1") != "0"
  peak = PEAK_BF16_MFMA_TFLOPS / 6 if x3 else PEAK_FP32_MFMA_TFLOPS
  out.append({"kernel": "nce fwd + bwd, n=4096 c=32 (5 tile GEMMs, %s)" % ("bf16x3 split on the matrix cores" if x3 else "fp32 VALU"),
              "ms": round(t * 1e3, 4), "gflop": round(fl * 1e-9, 3), "bound": "mfma", "achieved": round(fl / t * 1e-12, 3),
              "peak": round(peak, 1), "unit": "TFLOP/s", "frac": round(fl / t * 1e-12 / peak, 4)})
  # BatchNorm (train) fused with ReLU on [n, 96]: 2 reads + 1 write of the activation
  x = torch.randn(n, 96, device=device)
  gam, bet = torch.ones(96, device=device), torch.zeros(96, device=device)
  rm, rv = torch.zeros(96, device=device), torch.ones(96, device=device)
  print("[bench] timing bn", file=sys.stderr, flush=True)
  t = time_kernel(lambda: PF.BatchNormFunction.apply(x, gam, bet, rm, rv, 0.1, 1e-5, None, True))
  byts = 12 * n * 96
  out.append({"kernel": "bn_fwd_train+relu [n,96] (3 kernels)", "ms": round(t * 1e3, 4), "algo_mb": round(byts * 1e-6, 2),
              "bound": "hbm", "achieved": round(byts / t * 1e-9, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
              "frac": round(byts / t * 1e-9 / PEAK_HBM_GBS, 4)})
  return dominant, out


def cpu_baseline(batch_size_sample=1):
  """The CPU oracle's identical iteration (Res16UNet34C, PointInfoNCE, SGD) on `batch_size_sample`
  synthetic pairs: one untimed + one timed iteration, all host cores."""
  import torch
  from oracle import loss_ref as lr, model_ref as mr, sparse_ref as sr
  # torch-CPU index_select / index_add_ / mm stop scaling long before a 256-thread host is full
  # (and oversubscription makes them slower), so the baseline uses at most 32 threads
  cores = min(os.cpu_count() or 1, int(os.environ.get("PCMI_CPU_BASELINE_THREADS", "32")))
  torch.set_num_threads(cores)
  b = get_batch(seed=0, batch_size=batch_size_sample, voxel_size=0.025)
  torch.manual_seed(0)
  model = mr.Res16UNet34CRef(3, 32)
  model.train()
  opt = lr.make_sgd(model.parameters(), 0.1)
  pp = b["correspondences"].numpy()
  nq = len(np.unique(pp[:, 0]))
  rng = np.random.RandomState(0)
  times = []
  n_timed = int(os.environ.get("PCMI_CPU_BASELINE_ITERS", "3"))  # >= 3 timed iterations, same inputs
  for it in range(1 + n_timed):
    t0 = time.perf_counter()
    opt.zero_grad()
    F0 = model(sr.SparseTensorRef(b["sinput0_F"], coords=b["sinput0_C"].numpy())).F
    F1 = model(sr.SparseTensorRef(b["sinput1_F"], coords=b["sinput1_C"].numpy())).F
    si = rng.choice(nq, 4096, replace=False) if nq > 4096 else None
    qi, ki = lr.nce_select_pairs(pp, torch.rand(nq), si)
    loss = lr.nce_loss(F0, F1, qi, ki, 0.4)
    loss.backward()
    opt.step()
    times.append(time.perf_counter() - t0)
  timed = times[1:]
  return {"value": round(batch_size_sample * len(timed) / sum(timed), 4), "unit": "scene-pairs/sec",
          "cores": torch.get_num_threads(), "host_cores": os.cpu_count(), "kind": "port",
          "sample": "%d pair(s) (N0=%d, N1=%d voxels), full oracle iteration (2 fwd + NCE + bwd + SGD), 1 untimed + %d "
                    "timed (%s s each), time.perf_counter"
                    % (batch_size_sample, b["sinput0_C"].shape[0], b["sinput1_C"].shape[0], len(timed),
                       "/".join("%.1f" % t for t in timed))}


def run_cpu_baseline_bounded(limit_s=300):
  """Runs cpu_baseline() in a child process so that a slow host cannot stall the bench line."""
  import subprocess
  code = "import json, bench; print('CPUBASE' + json.dumps(bench.cpu_baseline(1)))"
  try:
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True, timeout=limit_s)
    for line in r.stdout.splitlines():
      if line.startswith("CPUBASE"):
        return json.loads(line[len("CPUBASE"):])
    return {"value": None, "unit": "scene-pairs/sec", "cores": 0, "host_cores": os.cpu_count(), "kind": "port",
            "sample": "cpu baseline failed: " + (r.stderr or "")[-200:]}
  except subprocess.TimeoutExpired:
    return {"value": None, "unit": "scene-pairs/sec", "cores": 0, "host_cores": os.cpu_count(), "kind": "port",
            "sample": "1 pair did not finish within %d s on this host" % limit_s}


def workload_label(args):
  """Which BASELINE.json config the run has the shape of (by voxel size / loss / batch), else 'custom'."""
  if args.model != "Res16UNet34C" or args.batch != 4:
    return "custom (not a BASELINE config)"
  if abs(args.voxel - 0.025) < 1e-9:
    return "BASELINE configs[1]" if args.loss == "nce" else "BASELINE configs[2]"
  if abs(args.voxel - 0.01) < 1e-9 and args.loss == "nce":
    return "BASELINE configs[4] shape (1 cm voxels) on %d GPU(s)" % args.gpus
  return "custom (not a BASELINE config)"


def log(msg):
  print("[bench] " + msg, file=sys.stderr, flush=True)


def timed_leg(loss, voxel, batch_size, steps, warmup, model="Res16UNet34C", overrides=(), seeds=(0,)):
  """One more workload on THIS GPU in the same process (N = 1 only): the full training iteration of `loss` at `voxel`,
  `warmup` untimed + `steps` timed iterations bracketed by synchronisations.  Used for the extra.* legs of the JSON line
  (BASELINE configs[2] = HardestContrastive, configs[4] shape = 1 cm voxels), so that the driver's own run witnesses
  them next to the headline number."""
  import torch
  from pointcontrast_amd.lib import ddp_trainer
  from pointcontrast_amd.lib.config import get_config
  from pointcontrast_amd.lib.ddp_data_loaders import FixedBatchLoader
  from pointcontrast_amd.lib.timer import AverageMeter, Timer
  cfg = get_config(["net.model=%s" % model, "misc.nceT=0.4", "misc.npos=4096", "opt.lr=0.1", "misc.num_gpus=1",
                    "trainer.batch_size=%d" % batch_size] + list(overrides))
  batches = get_batches(list(seeds), batch_size, voxel)
  batch = batches[0]
  loader = FixedBatchLoader(batches, batch_size=batch_size)  # replayed in turn: step i takes batch i mod len(seeds)
  torch.manual_seed(0)
  np.random.seed(0)
  cls = ddp_trainer.PointNCELossTrainer if loss == "nce" else ddp_trainer.HardestContrastiveLossTrainer
  trainer = cls(cfg, loader)
  it, timers = iter(loader), [AverageMeter(), Timer(), Timer()]
  for _ in range(warmup):
    res = trainer._train_iter(it, timers)
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  for _ in range(steps):
    res = trainer._train_iter(it, timers)
  torch.cuda.synchronize()
  dt = time.perf_counter() - t0
  out = {"value": round(batch_size * steps / dt, 3), "unit": "scene-pairs/sec", "ms_per_step": round(dt / steps * 1e3, 3),
         "steps": steps, "warmup": warmup, "final_loss": round(float(res["loss"]), 5),
         "voxels_per_forward_pair": [int(batch["sinput0_C"].shape[0]), int(batch["sinput1_C"].shape[0])]}
  if len(batches) > 1:
    out["batches"] = [{"seed": int(sd), "voxels": [int(b["sinput0_C"].shape[0]), int(b["sinput1_C"].shape[0])],
                       "correspondences": int(b["correspondences"].shape[0])} for sd, b in zip(seeds, batches)]
  del trainer
  torch.cuda.empty_cache()
  return out


def in_step_times(trainer, it, timers, ops, sets=6):
  """What the level-1 96 -> 96 convolution launches cost INSIDE the training step (HIP events the executor records
  around them on the streams they run on: pcmi_net_time_ops), over `sets` further iterations with nothing synchronised
  in between.  In the step the weights are packed once per pass (x3_pack_many_kernel), so a forward / backward-data
  'launch' here is the main kernel + its fix-up pass; the weight-gradient launch is the kernel + its slab sum."""
  import torch
  eng = trainer.engine
  eng.time_ops(ops, n_sets=sets)
  for _ in range(sets):
    trainer._train_iter(it, timers)
  torch.cuda.synchronize()
  recs = eng.timed_ms(sets)
  eng.time_ops([])
  mean = lambda xs: round(sum(xs) / len(xs), 4) if xs else None
  pick = lambda kind: [t for r in recs for t in r[kind] if t >= 0]
  return {"fwd_ms": mean(pick(0)), "bwd_data_ms": mean(pick(1)), "wgrad_ms": mean(pick(2)), "ops_timed": len(ops), "iterations": sets}


def family_table(trainer, it, timers, batch, device, sets=6, layer_times_path=None):
  """families[] of the bench line: what every family of launches costs INSIDE the training step,
  next to its algorithmic work and the peak that bounds it.

  How: pcmi_net_time_all -- HIP events around every op of the executor's program on the stream it runs on (forward and
  backward launches on the compute stream, weight gradients on the side stream; the grouped coarse-level launches as
  launches) over `sets` further iterations with nothing synchronised in between, the launch count of every call from the
  library's own counter -- plus torch events around the three phases outside the executor: batch preparation (planning
  stream: uploads, coordinate hash, strided levels, kernel maps, mask sort, pair selection), the loss block (pair gathers,
  PointInfoNCE forward + backward, gradient scatter) and the SGD step.  `ms` is stream time between the two events of a
  call, i.e. kernels + the gaps between them + waiting for compute units beside the other streams -- what the call costs
  where it runs, not its stand-alone kernel time.  Work: exact pair counts of the step's own kernel maps (the pair as one
  two-segment tensor); conv flops = 2 M cin cout per launch kind, bytes = SURVEY 8d's formula; BatchNorm 12 / 20 B per
  element forward / backward."""
  import torch
  import pointcontrast_amd.minkowski as ME
  eng = trainer.engine
  ops, tens = eng._ops, eng._tensors
  # --- exact pair counts / rows of the step's tensors (the same coordinates the step plans)
  st = level1_tensor(batch, device, joint=True)
  cm = st.coords_man
  keys, rows = [st.coords_key], [st.F.shape[0]]
  for _ in range(eng.n_down):
    keys.append(cm.stride(keys[-1], 2))
    rows.append(int(cm.size(keys[-1])))
  work = []
  for o in ops:
    li, lo = tens[o["in_"]]["level"], tens[o["out"]]["level"]
    if o["type"] != 0:
      work.append(dict(M=0, n_in=rows[li], n_out=rows[lo], K=0))
      continue
    ks = o["kernel_size"]
    if ks == 1:
      M, K = rows[li], 1
    else:
      m = cm.kernel_map(keys[lo], keys[li], ks, o["stride"], o["region"]) if o["transpose"] else cm.kernel_map(keys[li], keys[lo], ks, o["stride"], o["region"])
      M, K = int(m.M), ks ** 3
    work.append(dict(M=M, n_in=rows[li], n_out=rows[lo], K=K))
  # --- events outside the executor
  marks = {"plan": [], "loss": [], "sgd": []}
  plan_stream = ME.handle_pool.plan_stream(device)
  orig_prepare, orig_fwd, orig_bwd, orig_step = trainer._prepare, eng.forward, eng.backward, trainer.optimizer.step
  ev = lambda: torch.cuda.Event(enable_timing=True)

  def prepare(*a, **k):
    e0, e1 = ev(), ev()
    e0.record(plan_stream)
    r = orig_prepare(*a, **k)
    e1.record(plan_stream)
    marks["plan"].append((e0, e1))
    return r

  pending = {}

  def forward(*a, **k):
    r = orig_fwd(*a, **k)
    pending["e0"] = ev()
    pending["e0"].record()
    return r

  def backward(*a, **k):
    e1 = ev()
    e1.record()
    marks["loss"].append((pending.pop("e0"), e1))
    return orig_bwd(*a, **k)

  def step(*a, **k):
    e0, e1 = ev(), ev()
    e0.record()
    r = orig_step(*a, **k)
    e1.record()
    marks["sgd"].append((e0, e1))
    return r

  trainer._prepare, eng.forward, eng.backward, trainer.optimizer.step = prepare, forward, backward, step
  try:
    eng.time_all(sets)
    for _ in range(sets):
      trainer._train_iter(it, timers)
    torch.cuda.synchronize()
    recs, groups, counts = eng.timed_ms(sets), eng.timed_groups_ms(sets), eng.timed_launches(sets)
  finally:
    trainer._prepare, eng.forward, eng.backward, trainer.optimizer.step = orig_prepare, orig_fwd, orig_bwd, orig_step
    eng.time_all(0)
  n = len(ops)
  mean = lambda xs: sum(xs) / len(xs) if xs else 0.0
  per_op = []
  for q in range(n):
    per_op.append(tuple(mean([r[kind][q] for r in recs if r[kind][q] >= 0]) for kind in range(3)) +
                  tuple(mean([c[kind][q] for c in counts]) for kind in range(3)))
  grp_ms, grp_n = mean([sum(g) for g in groups]), mean([sum(c[3]) for c in counts])
  names = {off: nme[:-len(".kernel")] for (nme, prm), off in zip(trainer.model.named_parameters(), trainer.flat.offsets) if nme.endswith(".kernel")}
  if layer_times_path:
    with open(layer_times_path, "w") as f:
      f.write("op\tlayer\ttype\tK\tcin\tcout\trows_in\trows_out\tpairs\tgflop_per_launch\tfwd_ms\tbwd_ms\twgrad_ms\tfwd_launches\tbwd_launches\twgrad_launches\n")
      for q, (o, w) in enumerate(zip(ops, work)):
        t = per_op[q]
        f.write("%d\t%s\t%s\t%d\t%d\t%d\t%d\t%d\t%d\t%.4f\t%.4f\t%.4f\t%.4f\t%.1f\t%.1f\t%.1f\n" % (
            q, names.get(o.get("w_off"), "-") if o["type"] == 0 else "-", ("conv", "bn", "l2norm")[o["type"]], w["K"], o.get("cin", 0),
            o.get("cout", 0), w["n_in"], w["n_out"], w["M"], 2e-9 * w["M"] * o.get("cin", 0) * o.get("cout", 0), t[0], t[1], t[2], t[3], t[4], t[5]))
      f.write("# grouped coarse-level weight gradients: %.4f ms, %.1f launches per step (timed as launches, not per layer: wgrad_ms 0 above)\n" % (grp_ms, grp_n))
  fam = {}

  def add(key, ms, launches, gflop=0.0, gb=0.0):
    d = fam.setdefault(key, dict(ms=0.0, launches=0.0, gflop=0.0, gb=0.0))
    d["ms"] += ms
    d["launches"] += launches
    d["gflop"] += gflop
    d["gb"] += gb

  for q, (o, w) in enumerate(zip(ops, work)):
    f_ms, b_ms, w_ms, f_n, b_n, w_n = per_op[q]
    if o["type"] == 0:
      cin, cout = o["cin"], o["cout"]
      fl = 2e-9 * w["M"] * cin * cout
      by_f = 1e-9 * (w["M"] * (4 * cin + 8) + w["n_out"] * 4 * cout + 4 * w["K"] * cin * cout)
      by_b = 1e-9 * (w["M"] * (4 * cout + 8) + w["n_in"] * 4 * cin + 4 * w["K"] * cin * cout)
      by_w = 1e-9 * (w["M"] * 4 * (cin + cout) + 8 * w["M"] + 4 * w["K"] * cin * cout)
      lvl1 = min(tens[o["in_"]]["level"], tens[o["out"]]["level"]) == 0
      key = "conv level 1 (fwd + bwd-data; stem, 2^3 down / up, block8, head)" if lvl1 else "conv coarse (strides 2-16, fwd + bwd-data, incl. split-reduce / fix-up launches)"
      has_b = b_n > 0
      add(key, f_ms + b_ms, f_n + b_n, fl * (2 if has_b else 1), by_f + (by_b if has_b else 0.0))
      add("weight gradients (side stream; incl. slab sums / reductions)", w_ms, w_n, fl, by_w)
    elif o["type"] == 1:
      e = 1e-9 * w["n_in"] * o["cout"]
      add("BatchNorm forward (statistics + merge + apply, +ReLU / +residual)", f_ms, f_n, 0.0, 12 * e)
      add("BatchNorm backward (statistics + merge + apply)", b_ms, b_n, 0.0, 20 * e)
    else:
      e = 1e-9 * w["n_in"] * o["cout"]
      add("L2 normalisation (fwd + bwd)", f_ms + b_ms, f_n + b_n, 0.0, (8 + 16) * e)
  add("weight gradients (side stream; incl. slab sums / reductions)", grp_ms, grp_n)
  el = lambda key: mean([a.elapsed_time(b) for a, b in marks[key][1:]])  # (the first iteration's events predate time_all's sync)
  npar = trainer.flat.numel
  out = []
  peak6 = PEAK_BF16_MFMA_TFLOPS / 6
  for key, d in fam.items():
    ent = {"family": key, "ms_per_step": round(d["ms"], 4), "launches_per_step": round(d["launches"], 1)}
    if d["gflop"] > 0:
      ent.update(gflop=round(d["gflop"], 2), achieved_tflops=round(d["gflop"] * 1e-3 / (d["ms"] * 1e-3), 2) if d["ms"] > 0 else None,
                 peak_tflops=round(peak6, 1), bound="mfma (six-product split arithmetic; the 32-channel layers of the family are HBM-bound and counted by time)",
                 algo_gb=round(d["gb"], 3))
      ent["frac"] = round(ent["achieved_tflops"] / peak6, 4) if ent["achieved_tflops"] else None
    else:
      ent.update(algo_gb=round(d["gb"], 3), achieved_gbs=round(d["gb"] / (d["ms"] * 1e-3), 1) if d["ms"] > 0 else None, peak_gbs=PEAK_HBM_GBS, bound="hbm")
      ent["frac"] = round(ent["achieved_gbs"] / PEAK_HBM_GBS, 4) if ent["achieved_gbs"] else None
    out.append(ent)
  out.append({"family": "batch preparation (planning stream: upload, hash, levels, kernel maps, mask sort, pair selection)",
              "ms_per_step": round(el("plan"), 4), "bound": "latency (integer / index work, off the chain)"})
  out.append({"family": "loss block (pair gathers, PointInfoNCE fwd + bwd, gradient scatter)", "ms_per_step": round(el("loss"), 4),
              "gflop": round(5 * 2 * 4096 * 4096 * 32 * 1e-9, 3), "bound": "latency (4096 x 4096 x 32: 13 us at the six-product bound)"})
  sgd_ms = el("sgd")
  out.append({"family": "SGD step (one launch over the flat buffers)", "ms_per_step": round(sgd_ms, 4), "launches_per_step": 1,
              "algo_gb": round(npar * 20e-9, 3), "achieved_gbs": round(npar * 20e-9 / (sgd_ms * 1e-3), 1) if sgd_ms > 0 else None,
              "peak_gbs": PEAK_HBM_GBS, "bound": "hbm", "frac": round(npar * 20e-9 / (sgd_ms * 1e-3) / PEAK_HBM_GBS, 4) if sgd_ms > 0 else None})
  return {"note": "in-step stream time per family, mean over %d iterations (HIP events on the streams the launches run on; the families "
                  "of the compute stream add up to the chain, the weight gradients run beside it on the side stream)" % sets,
          "families": out}


def fp32_instruction_leg(args, steps=10, warmup=5, limit_s=240):
  """The same iteration with every split-precision kernel switched off (PCMI_CONV16_X3=0 PCMI_WGRAD_X3T=0 PCMI_NCE_X3=0:
  the fp32 MFMA instruction throughout -- literally the reference's arithmetic), in a child process of this run."""
  import subprocess
  env = dict(os.environ, PCMI_CONV16_X3="0", PCMI_WGRAD_X3T="0", PCMI_NCE_X3="0")
  for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
    env.pop(k, None)
  cmd = [sys.executable, os.path.abspath(__file__), "--steps", str(steps), "--warmup", str(warmup), "--batch", str(args.batch),
         "--voxel", str(args.voxel), "--loss", args.loss, "--model", args.model, "--no-cpu-baseline", "--no-roofline", "--no-extra"]
  for a in args.set:
    cmd += ["--set", a]
  try:
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=limit_s)
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    d = json.loads(line[-1])
    return {"value": d["value"], "unit": d["unit"], "ms_per_step": d["ms_per_step"], "steps": d["steps"], "warmup": d["warmup"],
            "final_loss": d["config"]["final_loss"], "conv_arithmetic": d["config"]["conv_arithmetic"],
            "switches": "PCMI_CONV16_X3=0 PCMI_WGRAD_X3T=0 PCMI_NCE_X3=0"}
  except Exception as e:
    return {"value": None, "error": "%s: %s" % (type(e).__name__, str(e)[:300])}


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument("--gpus", type=int, default=1)
  ap.add_argument("--steps", type=int, default=30)
  ap.add_argument("--warmup", type=int, default=8)  # (arena / pinned-buffer growth and the clocks of a fresh box settle within ~6 iterations)
  ap.add_argument("--batch", type=int, default=4, help="scene pairs per GPU (BASELINE config: 4)")
  ap.add_argument("--voxel", type=float, default=0.025)
  ap.add_argument("--loss", choices=["nce", "hardest"], default="nce")
  ap.add_argument("--model", default="Res16UNet34C")
  ap.add_argument("--engine", choices=["native", "autograd"], default="native")
  ap.add_argument("--no-cpu-baseline", action="store_true")
  ap.add_argument("--no-roofline", action="store_true")
  ap.add_argument("--no-extra", action="store_true", help="skip the extra.hardest / extra.voxel_1cm legs (N = 1 default run)")
  ap.add_argument("--layer-table", default=None, help="write the per-layer work table to this path")
  ap.add_argument("--set", action="append", default=[], metavar="a.b=c", help="extra config override (A/B experiments)")
  args = ap.parse_args()

  import __graft_entry__
  if "WORLD_SIZE" not in os.environ and args.gpus > 1:
    # No external launcher: start the N ranks ourselves, as the reference's entry point does (pc/ddp_train.py:57-59 ->
    # lib/multiprocessing.py:36-56).  The library is built ONCE, here, before any rank exists; every rank is this same
    # script with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in its environment; a failing rank stops the others and its
    # exit code becomes ours.  Rank 0 inherits this stdout, so the JSON line arrives where the caller reads it.
    __graft_entry__.build()
    from pointcontrast_amd.lib.multiprocessing import launch_script_ranks
    sys.exit(launch_script_ranks(args.gpus, [os.path.abspath(__file__)] + sys.argv[1:]))
  import torch
  import torch.distributed as dist
  assert torch.cuda.is_available(), "bench.py measures the HIP path and needs an MI355X (no CPU fallback)"
  world = int(os.environ.get("WORLD_SIZE", "1"))
  rank = int(os.environ.get("RANK", "0"))
  __graft_entry__.build()  # every rank: an exclusive file lock inside serialises them, all but the first find it built
  from pointcontrast_amd.lib import distributed as du
  forced = any(a.replace(" ", "") in ("misc.force_reducer=True", "misc.force_reducer=1") for a in args.set)
  if world > 1:
    du.init_process_group()
    dist.barrier()
  elif forced:
    # --set misc.force_reducer=True on one GPU: the N > 1 code path (bucket callbacks -> RCCL all-reduce on the side
    # stream -> one wait in front of SGD) in a 1-rank group -- what it costs the per-GPU step when nothing is exchanged
    from pointcontrast_amd.lib.multiprocessing import free_port
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(free_port()), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    du.init_process_group(0, 1)
  assert world == args.gpus, "--gpus %d but WORLD_SIZE=%d" % (args.gpus, world)
  device = torch.device("cuda", torch.cuda.current_device())

  from pointcontrast_amd.lib.config import get_config
  from pointcontrast_amd.lib.ddp_data_loaders import FixedBatchLoader
  from pointcontrast_amd.lib import ddp_trainer
  from pointcontrast_amd.lib.timer import AverageMeter, Timer
  cfg = get_config(["net.model=%s" % args.model, "misc.nceT=0.4", "misc.npos=4096", "opt.lr=0.1",
                    "misc.num_gpus=%d" % world, "trainer.batch_size=%d" % (args.batch * world),
                    "misc.engine=%s" % args.engine, "misc.host_profile=True",
                    "misc.reducer_profile=%s" % (world > 1 or forced)] + list(args.set))
  batch = get_batch(seed=rank, batch_size=args.batch, voxel_size=args.voxel)
  loader = FixedBatchLoader([batch], batch_size=args.batch)
  torch.manual_seed(0)
  np.random.seed(rank)
  cls = ddp_trainer.PointNCELossTrainer if args.loss == "nce" else ddp_trainer.HardestContrastiveLossTrainer
  trainer = cls(cfg, loader)
  if cfg.misc.get("gpu_profile", False) and trainer.engine is not None:
    trainer.engine.pair_marks = []
  it = iter(loader)
  timers = [AverageMeter(), Timer(), Timer()]

  log("warmup")
  for _ in range(args.warmup):
    res = trainer._train_iter(it, timers)
  torch.cuda.synchronize()
  if world > 1:
    dist.barrier()
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  for _ in range(args.steps):
    res = trainer._train_iter(it, timers)
  host_enqueue = time.perf_counter() - t0  # host side of the K steps (nothing inside them synchronises)
  torch.cuda.synchronize()
  if world > 1:
    dist.barrier()
  torch.cuda.synchronize()
  elapsed = time.perf_counter() - t0
  loss_val = float(res["loss"])
  per_rank = None
  if world > 1:
    # what every rank saw: its own wall time of the K steps, its host-side enqueue time and how long its all-reduces stayed
    # exposed after its backward pass -- so that the first real N > 1 run says WHICH rank everyone waited for (the ranks'
    # batches differ by +-10 % in voxel count; GradReducer.finish() makes every step as long as the slowest rank's)
    try:
      rep = trainer.reducer.overlap_report(skip_steps=args.warmup) or {}
    except Exception:  # noqa: BLE001  (measurement garnish: every rank must still reach the collectives below)
      rep = {}
    mine = torch.tensor([elapsed, host_enqueue, float(rep.get("exposed_after_backward_ms") or 0.0),
                         float(batch["sinput0_C"].shape[0] + batch["sinput1_C"].shape[0])], device=device, dtype=torch.float64)
    allr = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(allr, mine)
    per_rank = [{"rank": r, "ms_per_step": round(float(t[0]) / args.steps * 1e3, 3), "host_enqueue_ms_per_step": round(float(t[1]) / args.steps * 1e3, 3),
                 "exposed_after_backward_ms": round(float(t[2]), 3), "voxels_per_pass": int(t[3])} for r, t in enumerate(allr)]
    tt = torch.tensor([elapsed], device=device, dtype=torch.float64)
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    elapsed = float(tt.item())

  if rank == 0:
    n0, n1 = batch["sinput0_C"].shape[0], batch["sinput1_C"].shape[0]
    if args.engine == "native":  # per-layer pair counts are recorded by the module path: one untimed forward
      import pointcontrast_amd.minkowski as ME
      with torch.no_grad():
        trainer.model(ME.SparseTensor(batch["sinput1_F"], coords=batch["sinput1_C"]).to(device))
      torch.cuda.synchronize()
    flops, byts, rows = conv_work(trainer.model)  # cloud 1
    out = {
        "metric": "scene-pairs/sec, ScanNet %s Res16UNet34C %s" % (
            "2.5cm" if abs(args.voxel - 0.025) < 1e-9 else "%gcm" % (args.voxel * 100),
            "PointInfoNCE" if args.loss == "nce" else "HardestContrastive"),
        "value": round(args.batch * world * args.steps / elapsed, 3), "unit": "scene-pairs/sec",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "%s: %s, %s loss, voxel %.3g m, %d pairs/GPU, %d+%d active voxels per "
                               "forward pair on rank 0, npos 4096, T 0.4, SGD(lr 0.1, mom 0.8, wd 1e-4)"
                               % (workload_label(args), args.model, args.loss, args.voxel, args.batch, n0, n1),
                   "global_batch": args.batch * world, "parallelism": "dp%d" % world, "engine": args.engine,
                   "pair_execution": ("one two-segment pass (BatchNorm statistics per cloud)"
                                      if args.engine == "native" and cfg.misc.get("joint_pair", True) else "one pass per cloud"),
                   "conv_arithmetic": ("fp32 in / fp32 out; matrix-bound forward / backward-data convolutions contract fp32 operands "
                                       "split into 3 bf16 terms on the bf16 matrix cores with fp32 accumulation (within fp32 "
                                       "round-off of the fp32 instruction); everything else fp32 MFMA / VALU"
                                       if _split_precision() else "fp32 MFMA / VALU throughout"),
                   "final_loss": round(loss_val, 5),
                   "host_cpu": getattr(os, "sched_getcpu", lambda: -1)(),
                   "host_enqueue_ms_per_step": round(host_enqueue / args.steps * 1e3, 3),
                   **({"gpu_phase_ms_per_step": trainer.gpu_phase_ms(skip=args.warmup)} if trainer._gpu_marks else {}),
                   **({"forward_pair_ms": trainer.engine.pair_marks_ms(skip=args.warmup)}
                      if getattr(trainer.engine, "pair_marks", None) else {}),
                   **({"host_phase_ms_per_step": {k: round(v / (args.steps + args.warmup), 3) for k, v in trainer.host_ms.items()}}
                      if trainer.host_ms else {}),
                   "conv_gflop_per_forward": round(flops * 1e-9, 2), "conv_algo_gb_per_forward": round(byts * 1e-9, 3),
                   # N > 1: what RCCL saw -- ranks, backend, gradient buckets, and how much of the all-reduce was hidden
                   "collective": ({"ranks": dist.get_world_size(), "backend": dist.get_backend(),
                                   "allreduce_mb_per_step": round(trainer.flat.numel * 4 / 2 ** 20, 1),
                                   "buckets": len(trainer.reducer.buckets),
                                   "bucket_mb": [round((hi - lo) * 4 / 2 ** 20, 1) for lo, hi, _ in trainer.reducer.buckets],
                                   "rccl_max_channels": du.rccl_channel_cap(),
                                   "overlap": trainer.reducer.overlap_report(skip_steps=args.warmup),
                                   "per_rank": per_rank,
                                   "step_skew_ms": (round(max(p_["ms_per_step"] for p_ in per_rank) - min(p_["ms_per_step"] for p_ in per_rank), 3)
                                                    if per_rank else None)}
                                  if (world > 1 or forced) else None)},
    }
    if args.layer_table:
      with open(args.layer_table, "w") as f:
        f.write("layer\tK\tcin\tcout\tpairs\tn_out\tflops\talgo_bytes\n")
        for r in rows:
          f.write("\t".join(str(v) for v in r) + "\n")
    log("timed region done: %.2f ms/step" % (elapsed / args.steps * 1e3))
    if not args.no_roofline:
      instep = None
      if trainer.engine is not None:
        ops96 = [i for i, o in enumerate(trainer.engine._ops)
                 if o["type"] == 0 and o.get("cin") == 96 and o.get("cout") == 96 and o.get("kernel_size") == 3
                 and o.get("stride") == 1 and trainer.engine._tensors[o["out"]]["level"] == 0]
        if ops96:
          try:
            instep = in_step_times(trainer, it, timers, ops96)
          except Exception as e:  # measurement garnish: never lose the headline to it
            instep = {"error": "%s: %s" % (type(e).__name__, str(e)[:200])}
          log("in-step kernel times: %s" % instep)
        if cfg.misc.get("joint_pair", True) and args.engine == "native":
          try:
            ft = family_table(trainer, it, timers, batch, device,
                              layer_times_path=(args.layer_table + ".times.tsv") if args.layer_table else None)
            out["families"], out["families_note"] = ft["families"], ft["note"]
          except Exception as e:  # measurement garnish: never lose the headline to it
            out["families"], out["families_note"] = None, "%s: %s" % (type(e).__name__, str(e)[:300])
          log("families done")
      dom, kernels = kernel_rooflines(batch, device, joint=bool(cfg.misc.get("joint_pair", True)) and args.engine == "native")
      log("rooflines done")
      # Which kernel is "the dominant one": launches per step of a shape (from the lowered program) x its stand-alone
      # launch time = its estimated share of the step; the forward / backward-data convolution and the weight-gradient
      # kernel of the level-1 96 -> 96 layers are both reported (`roofline` = the larger share, `roofline_other` = the
      # other), each with its own fraction -- the weight-gradient kernel is the one further from its bound.
      n96 = sum(1 for o in (trainer.engine._ops if trainer.engine is not None else [])
                if o["type"] == 0 and o.get("cin") == 96 and o.get("cout") == 96 and o.get("kernel_size") == 3
                and o.get("stride") == 1 and trainer.engine._tensors[o["out"]]["level"] == 0) or 3
      wg = next(k for k in kernels if k["kernel"].startswith("wgrad") and "@level1" in k["kernel"])
      share = {id(dom): 2 * n96 * dom["ms"], id(wg): n96 * wg["ms"]}  # fwd + bwd-data launches; one gradient launch per layer
      first, second = (dom, wg) if share[id(dom)] >= share[id(wg)] else (wg, dom)
      # HBM-side bytes per launch of the same kernel / shape from the separate rocprofv3 --pmc passes
      # (scripts/pmc_probe.py -> scripts/pmc_summary.py -> profiles/pmc_traffic.json); reported only when the file was
      # collected on THIS build (kernel_sources_sha16 = sha256 over the kernel sources, headers and flags) and on this
      # run's level-1 tensor (pair count); else null
      from pointcontrast_amd.build import sources_digest
      pmc, traffic_note = None, "profiles/pmc_traffic.json absent"
      try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
          pmc = json.load(f)
        if pmc.get("kernel_sources_sha16") != sources_digest()[:16]:
          traffic_note = "PMC passes were collected on another build (%s), not on this one (%s)" % (
              pmc.get("kernel_sources_sha16", "unstamped"), sources_digest()[:16])
          pmc = None
        elif pmc.get("algorithmic", {}).get("96->96", {}).get("pairs") != dom["pairs"]:
          traffic_note = "PMC passes were collected on another level-1 tensor"
          pmc = None
        else:
          traffic_note = "PMC passes of this build: profiles/%s" % pmc.get("source", "?")
      except (OSError, ValueError):
        pass

      def traffic_of(ent):
        if pmc is None:
          return None
        per = pmc["bytes_per_launch"]
        if ent["kernel"].startswith("wgrad"):
          names, extra = ("wgrad_x3p_kernel<3, 3, 4>", "wgrad_x3t_kernel<3, 3, 4>", "wgrad_mfma_kernel<3, 3, true, true>"), "wgrad_slab_sum_kernel"
        elif ent["kernel"].startswith("spconv16x"):
          names, extra = ("spconv16x_kernel<3, true, true>",), "sk_fixup_kernel"
        else:
          names, extra = ("spconv16p_kernel<3, false, true>",), "sk_fixup_kernel"
        for name in names:
          if name in per:  # (+ the launch's second pass: the fix-up / slab-sum kernel, whatever its template arguments)
            return per[name] + sum(v for k, v in per.items() if k.startswith(extra))
        return None

      def roofline_of(ent):
        r = {"bound": ent["bound"], "achieved": ent["achieved"], "peak": ent["peak"], "unit": ent["unit"],
             "frac": ent["frac"], "traffic": traffic_of(ent), "traffic_unit": "bytes/launch (PMC, calibrated)",
             "traffic_note": traffic_note, "kernel": ent["kernel"], "ms": ent["ms"],
             "launches_per_step": (2 if ent is dom else 1) * n96, "est_step_share_ms": round(share[id(ent)], 3),
             **{k: ent[k] for k in ("arithmetic", "peak_fp32_mfma", "frac_of_fp32_mfma_peak") if k in ent}}
        # the same launches INSIDE the step (events around them on their own streams, next to the other streams' work)
        if instep and "error" not in instep:
          if ent is dom:
            both = [t for t in (instep["fwd_ms"], instep["bwd_data_ms"]) if t]
            ms_in = sum(both) / len(both) if both else None
            r["in_step"] = {"fwd_ms": instep["fwd_ms"], "bwd_data_ms": instep["bwd_data_ms"]}
          else:
            ms_in = instep["wgrad_ms"]
          if ms_in:
            r["in_step_ms"] = round(ms_in, 4)
            r["in_step_frac"] = round(ent["gflop"] * 1e-3 / (ms_in * 1e-3) / ent["peak"], 4)
            r["in_step_note"] = ("mean over %d iterations x %d layers, HIP events on the launch stream inside pcmi_net_forward / "
                                 "_backward; weights pre-packed once per pass" % (instep["iterations"], instep["ops_timed"]))
        elif instep:
          r["in_step_error"] = instep["error"]
        return r

      out["roofline"] = roofline_of(first)
      out["roofline_other"] = roofline_of(second)
      out["kernels"] = kernels
    if world == 1 and not args.no_extra and workload_label(args) == "BASELINE configs[1]":
      # the other two single-GPU-measurable BASELINE configurations, short legs in the same process (same build, same box)
      del trainer, it, loader
      torch.cuda.empty_cache()
      out["extra"] = {}
      labels = {"hardest": "BASELINE configs[2]: HardestContrastive, 2.5 cm",
                "voxel_1cm": "BASELINE configs[4] shape: PointInfoNCE, 1 cm voxels, on 1 GPU",
                "rotating_batches": "BASELINE configs[1] on FOUR different batches (seeds 0-3) replayed in turn: every step sees "
                                    "other level sizes than the one before (arena / pinned-buffer growth, weight-pack size classes "
                                    "inside the timed region), as a real loader's batches do (pc/lib/ddp_trainer.py:389)"}
      for key, kw in (("rotating_batches", dict(loss="nce", voxel=0.025, steps=args.steps, warmup=8, seeds=(0, 1, 2, 3))),
                      ("hardest", dict(loss="hardest", voxel=0.025, steps=20, warmup=6)),
                      ("voxel_1cm", dict(loss="nce", voxel=0.01, steps=6, warmup=3))):
        log("extra leg: %s" % key)
        try:
          leg = timed_leg(batch_size=args.batch, model=args.model, overrides=list(args.set), **kw)
          leg["workload"] = labels[key]
          if key == "rotating_batches":
            leg["vs_headline"] = round(leg["value"] / out["value"], 4)
          out["extra"][key] = leg
        except Exception as e:  # the headline number must not be lost to a failing extra leg
          out["extra"][key] = {"value": None, "error": "%s: %s" % (type(e).__name__, str(e)[:300])}
      log("extra leg: fp32_mfma")
      out["extra"]["fp32_mfma"] = fp32_instruction_leg(args)
      out["extra"]["fp32_mfma"]["workload"] = "BASELINE configs[1] on the fp32 MFMA instruction throughout (no split-precision kernel)"
    if world == 1 and not args.no_cpu_baseline:
      out["cpu_baseline"] = run_cpu_baseline_bounded()
    print(json.dumps(out), flush=True)
  if world > 1:
    dist.barrier()
  if world > 1 or forced:
    du.destroy_process_group()


if __name__ == "__main__":
  main()
