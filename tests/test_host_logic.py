"""CPU-side tests: C-ABI library loads and exports every symbol of include/pcmi.h, host-only
entry points, configuration / sampler / batch-contract logic, pair selection vs the oracle."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol(built_lib):
  hdr = open(os.path.join(ROOT, "include", "pcmi.h")).read()
  hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
  declared = sorted(set(re.findall(r"\b(pcmi_[a-z0-9_]+)\s*\(", hdr)))
  assert len(declared) >= 40
  lib = ctypes.CDLL(built_lib)
  missing = [s for s in declared if not hasattr(lib, s)]
  assert not missing, missing
  from pointcontrast_amd import _lib
  assert sorted(_lib.PROTOTYPES) == declared, set(declared) ^ set(_lib.PROTOTYPES)
  assert _lib.version() == 100


def test_kernel_offsets_host_entry_matches_oracle(built_lib):
  from oracle import sparse_ref as sr
  from pointcontrast_amd._lib import lib, check
  for ks, region in ((3, 0), (3, 3), (2, 0), (1, 0)):
    buf = (ctypes.c_int32 * 81)()
    K = ctypes.c_int()
    check(lib.pcmi_kernel_offsets(ks, region, buf, ctypes.byref(K)))
    got = np.ctypeslib.as_array(buf)[:K.value * 3].reshape(-1, 3)
    assert (got == sr.region_offsets(ks, region)).all(), (ks, region)


def test_errors_are_reported_not_swallowed(built_lib):
  from pointcontrast_amd._lib import lib, check, PcmiError
  buf = (ctypes.c_int32 * 81)()
  K = ctypes.c_int()
  with pytest.raises(PcmiError, match="not on the hot path"):
    check(lib.pcmi_kernel_offsets(5, 0, buf, ctypes.byref(K)))


def test_ops_refuse_cpu_tensors(built_lib):
  """No CPU fallback: the product path must fail loudly without a device."""
  from pointcontrast_amd import functional as PF
  from pointcontrast_amd._lib import PcmiError
  with pytest.raises(PcmiError, match="no CPU path"):
    PF.ReLUFunction.apply(torch.zeros(4, 4))
  with pytest.raises(PcmiError, match="no CPU path"):
    PF.NCELossFunction.apply(torch.zeros(4, 32), torch.zeros(4, 32), 0.4)


def test_config_defaults_and_overrides():
  from pointcontrast_amd.lib.config import get_config
  c = get_config(["misc.nceT=0.4", "trainer.trainer=PointNCELossTrainer", "opt.lr=0.05", "data.voxel_size=0.01"])
  assert c.misc.nceT == 0.4 and c.opt.lr == 0.05 and c.trainer.trainer == "PointNCELossTrainer"
  assert c.opt.momentum == 0.8 and c.opt.weight_decay == 1e-4 and c.opt.bn_momentum == 0.05  # defaults.yaml:43-53
  assert c.trainer.num_pos_per_batch == 1024 and c.trainer.num_hn_samples_per_batch == 256 and c.misc.npos == 4096


def test_samplers():
  from pointcontrast_amd.lib.data_sampler import DistributedInfSampler, InfSampler
  torch.manual_seed(0)
  s = InfSampler(list(range(5)), shuffle=False)
  assert [next(s) for _ in range(7)] == [4, 3, 2, 1, 0, 4, 3]
  torch.manual_seed(0)
  a = DistributedInfSampler(list(range(10)), num_replicas=2, rank=0)
  torch.manual_seed(0)
  b = DistributedInfSampler(list(range(10)), num_replicas=2, rank=1)
  xa, xb = [next(a) for _ in range(5)], [next(b) for _ in range(5)]
  assert sorted(xa + xb) == list(range(10))  # ranks stride one shared permutation


def test_synthetic_batch_contract():
  from pointcontrast_amd.lib.ddp_data_loaders import default_collate_pair_fn
  from pointcontrast_amd.lib import synthetic
  rng = np.random.RandomState(3)
  items = [synthetic.make_pair_item(rng, 0.025, crop=0.5) for _ in range(2)]
  b = default_collate_pair_fn(items)
  C0, F0, corr = b["sinput0_C"], b["sinput0_F"], b["correspondences"]
  assert C0.dtype == torch.int32 and F0.dtype == torch.float32 and corr.dtype == torch.int32
  assert C0.shape[1] == 4 and F0.shape == (C0.shape[0], 3)
  assert set(C0[:, 0].tolist()) == {0, 1}  # batch index FIRST
  assert len(np.unique(C0.numpy(), axis=0)) == len(C0)  # unique voxels
  assert (np.diff(corr[:, 0].numpy()) >= 0).all()  # sorted by query row (needed by the NCE selection)
  assert corr[:, 0].max() < C0.shape[0] and corr[:, 1].max() < b["sinput1_C"].shape[0]
  n0 = b["len_batch"][0][0]
  second = corr[corr[:, 0] >= n0]
  assert (second[:, 1] >= b["len_batch"][0][1]).all()  # offsets applied to both columns


def test_pair_selection_matches_oracle():
  from oracle import loss_ref as lr
  from pointcontrast_amd.lib.ddp_trainer import PointNCELossTrainer, _hash
  rng = np.random.RandomState(0)
  q = np.sort(rng.randint(0, 300, 2000))
  pp = np.stack([q, rng.randint(0, 500, 2000)], 1).astype(np.int32)
  nq = len(np.unique(q))
  u = torch.rand(nq)
  si = rng.choice(nq, 100, replace=False)
  a = PointNCELossTrainer.select_pairs(torch.from_numpy(pp), 100, dict(uniform=u, sampled_inds=si))
  b = lr.nce_select_pairs(pp, u, si)
  assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
  assert (_hash(pp.astype(np.int64), 1000) == lr.hash_pairs(pp[:, 0], pp[:, 1], 1000)).all()


def test_pair_selection_consumes_the_reference_rng_streams_and_accepts_unsorted_pairs():
  """The un-injected path of select_pairs: torch.rand(n) is value- and generator-identical to the reference's
  torch.distributions.Uniform(0, 1).sample([n]) (pc/lib/ddp_trainer.py:408), the sub-sample is the reference's
  np.random.choice(n, npos, replace=False) (:412) -- so under the same seeds the selection equals the oracle fed with the
  reference's draws; an unsorted pair list is sorted first (stable), an empty one selects nothing."""
  from oracle import loss_ref as lr
  from pointcontrast_amd.lib.ddp_trainer import PointNCELossTrainer
  rng = np.random.RandomState(1)
  q = np.sort(rng.randint(0, 5000, 60000))
  pp = np.stack([q, rng.randint(0, 9000, 60000)], 1).astype(np.int32)
  nq = len(np.unique(q))
  torch.manual_seed(7)
  np.random.seed(11)
  a = PointNCELossTrainer.select_pairs(torch.from_numpy(pp), 4096)
  torch.manual_seed(7)
  np.random.seed(11)
  u = torch.distributions.Uniform(0, 1).sample([nq])
  si = np.random.choice(nq, 4096, replace=False)
  b = lr.nce_select_pairs(pp, u, si)
  assert torch.equal(a[0], torch.as_tensor(b[0]).long()) and torch.equal(a[1], torch.as_tensor(b[1]).long())
  assert a[0].dtype == torch.int64 and a[0].shape == (4096,)
  # unsorted input == the same pairs stably sorted by query row
  shuffled = pp[rng.permutation(len(pp))]
  u2 = torch.rand(nq)
  x = PointNCELossTrainer.select_pairs(torch.from_numpy(shuffled), 10 ** 9, dict(uniform=u2))
  y = PointNCELossTrainer.select_pairs(shuffled[np.argsort(shuffled[:, 0], kind="stable")], 10 ** 9, dict(uniform=u2))
  assert torch.equal(x[0], y[0]) and torch.equal(x[1], y[1]) and x[0].shape == (nq,)
  e = PointNCELossTrainer.select_pairs(torch.zeros((0, 2), dtype=torch.int32), 4096)
  assert e[0].numel() == 0 and e[1].numel() == 0
  one = PointNCELossTrainer.select_pairs(torch.tensor([[5, 7]], dtype=torch.int32), 4096, dict(uniform=torch.tensor([0.9])))
  assert one[0].tolist() == [5] and one[1].tolist() == [7]


def test_model_lowers_to_a_valid_network_program(built_lib):
  """Tracing + pcmi_net_create need no GPU: the static program of Res16UNet34C is well formed
  (every gradient contribution is either a first write or a full accumulate, slices fit)."""
  from pointcontrast_amd._lib import lib, check
  from pointcontrast_amd.engine import create_net, lower_model
  from pointcontrast_amd.lib.config import get_config
  from pointcontrast_amd.lib.distributed import FlatParameters
  from pointcontrast_amd.model import load_model
  cfg = get_config([])
  for name, n_conv in (("Res16UNet14", 33), ("Res16UNet34C", 63)):
    model = load_model(name)(3, 32, cfg, D=3)
    flat = FlatParameters(model.parameters())
    prog = lower_model(model, flat)
    kinds = [o["type"] for o in prog["ops"]]
    assert kinds.count(0) == n_conv and kinds.count(2) == 1 and prog["out_channels"] == 32 and prog["n_down"] == 4
    assert kinds.count(1) == n_conv - 1  # every conv but the head is followed by a BatchNorm
    cats = [t for t in prog["tensors"] if t["parent"] >= 0]
    assert len(cats) == 8  # 4 concatenations, 2 slices each, zero copy
    h = create_net(prog)
    check(lib.pcmi_net_destroy(h))
    offs = sorted(o["w_off"] for o in prog["ops"] if o["type"] == 0)
    assert len(set(offs)) == n_conv and offs[0] == 0


def test_checkpoint_layout_interoperates_with_reference_format(built_lib, tmp_path):
  """a checkpoint in the reference's layout ({state_dict, optimizer, scheduler, curr_iter, config},
  pc/lib/ddp_trainer.py:113-133) whose state_dict uses the reference's parameter names (restated by the oracle
  model: '<conv>.kernel' [K, Cin, Cout], '<bn>.bn.weight', 'final.bias') loads into the device model key for
  key, strict -- no CPU compute involved, only module construction."""
  from oracle import model_ref as mr
  from pointcontrast_amd.lib.config import get_config
  from pointcontrast_amd.lib.ddp_trainer import load_state
  from pointcontrast_amd.model import load_model
  cfg = get_config([])
  for name in ("Res16UNet14", "Res16UNet34C"):
    ref = mr.MODELS[name](3, 32)
    dev = load_model(name)(3, 32, cfg, D=3)
    sd_ref, sd_dev = ref.state_dict(), dev.state_dict()
    assert list(sd_ref.keys()) == list(sd_dev.keys()), "parameter / buffer names or order differ"
    for k in sd_ref:
      assert tuple(sd_ref[k].shape) == tuple(sd_dev[k].shape), k
    path = tmp_path / ("%s.pth" % name)
    torch.save({"curr_iter": 7, "state_dict": sd_ref, "optimizer": {}, "scheduler": {}, "config": {}}, path)
    state = torch.load(path, map_location="cpu", weights_only=False)
    load_state(dev, state["state_dict"])  # strict
    assert torch.equal(dev.state_dict()["conv0p1s1.kernel"], sd_ref["conv0p1s1.kernel"])
    # lenient loading keeps matching tensors and skips a head of another width (ddp_trainer.py:45-55 of the reference)
    other = load_model(name)(3, 16, cfg, D=3)
    load_state(other, state["state_dict"], lenient_weight_loading=True)
    assert torch.equal(other.state_dict()["bn0.bn.weight"], sd_ref["bn0.bn.weight"])


def test_engine_bucket_callback_mapping(built_lib):
  """The native executor takes the reducer's buckets as ASCENDING flat offsets and reports a finished bucket by its
  position in that list; NativeEngine._ready_args must translate back to the reducer's own bucket index (the
  reducer numbers its buckets from the END of the flat buffer, the order backward finishes them)."""
  from pointcontrast_amd.engine import NativeEngine

  class StubReducer:
    active = True
    buckets = [(700, 1000, None), (300, 700, None), (0, 300, None)]  # (lo, hi, _), last parameters first

    def __init__(self):
      self.launched = []

    def _launch(self, b):
      self.launched.append(b)

  r = StubReducer()
  cb, lo_arr, nb = NativeEngine._ready_args(r)
  assert nb == 3 and list(lo_arr) == [0, 300, 700]
  for q in (2, 1, 0):  # the executor finishes the highest offsets first
    cb(None, q)
  assert r.launched == [0, 1, 2]
  r.active = False
  cb, lo_arr, nb = NativeEngine._ready_args(r)
  assert nb == 0 and lo_arr is None


def test_prefetch_thread_keeps_batch_order_and_reraises(built_lib):
  """misc.prefetch_thread: batch N+1 is prepared on a helper thread while step N is enqueued; the training thread
  must get the batches in loader order and see the helper's exceptions (host logic only: _prepare is stubbed)."""
  import threading
  from pointcontrast_amd.lib import ddp_trainer
  from pointcontrast_amd.lib.config import get_config
  from pointcontrast_amd.lib.timer import Timer
  tr = object.__new__(ddp_trainer.PointNCELossTrainer)
  tr.config = get_config([])
  tr.cur_device = torch.device("cpu")
  tr._prefetch_thread, tr._prefetch_err, tr._prefetched = None, None, None
  tr.engine = None  # per-layer path: the helper thread is the default
  seen_threads = []

  def fake_prepare(input_dict, draws=None):
    seen_threads.append(threading.current_thread().name)
    if input_dict == "bad":
      raise ValueError("boom")
    return {"input": input_dict}

  tr._prepare = fake_prepare
  it = iter(["a", "b", "c", "bad", "never"])
  got = []
  for _ in range(3):
    prep, _ = tr._next_prepared(it, Timer(), None)
    tr._prefetch_start(it, None)
    got.append(prep["input"])
  assert got == ["a", "b", "c"]
  assert seen_threads[0] == threading.current_thread().name and all(n != seen_threads[0] for n in seen_threads[1:])
  with pytest.raises(ValueError, match="boom"):
    tr._next_prepared(it, Timer(), None)  # the helper failed on "bad"
  # fixed draws (parity tests) bypass the helper thread entirely
  tr._prefetch_start(it, {"uniform": None})
  assert tr._prefetch_thread is None
  # native engine with the pair as one two-segment pass (the default): no prefetch unless asked for explicitly
  tr.engine = object()
  assert tr._prefetch_mode() is None
  tr.config = get_config(["misc.prefetch_thread=True"])
  assert tr._prefetch_mode() == "thread"
  tr.config = get_config(["misc.prefetch_thread=False"])
  assert tr._prefetch_mode() == "inline"
  tr.config = get_config(["misc.joint_pair=False"])
  assert tr._prefetch_mode() == "thread"


def test_scannet_match_pair_dataset_and_infinite_loader(tmp_path):
  """The reference's dataset class (pc/lib/ddp_data_loaders.py:119-270) on a tiny on-disk pair list: item contract,
  correspondences within the search radius and sorted by query row; the single-GPU loader never runs dry
  (train() calls next() far more often than len(loader))."""
  from pointcontrast_amd.lib import synthetic
  from pointcontrast_amd.lib.config import get_config
  from pointcontrast_amd.lib.ddp_data_loaders import dataset_str_mapping, make_data_loader
  rng = np.random.RandomState(0)
  names = []
  for i in range(3):
    a, b = synthetic.make_frame_pair(rng)
    c = a[rng.randint(len(a))]
    for tag, x in (("a", a), ("b", b)):
      np.savez(tmp_path / ("f%d%s.npz" % (i, tag)), pcd=x[np.linalg.norm(x - c, axis=1) < 0.4])
    names.append("f%da.npz f%db.npz 0.5\n" % (i, i))
  (tmp_path / "pairs.txt").write_text("".join(names))
  cfg = get_config(["data.dataset=ScanNetMatchPairDataset", "data.dataset_root_dir=%s" % tmp_path,
                    "data.scannet_match_dir=pairs.txt", "trainer.batch_size=2", "trainer.use_random_scale=True"])
  dset = dataset_str_mapping["ScanNetMatchPairDataset"](phase="train", config=cfg, random_scale=False)
  xyz0, xyz1, c0, c1, f0, f1, m, trans = dset[1]
  assert len(dset) == 3 and c0.shape == xyz0.shape and f0.shape == (len(xyz0), 3) and trans.shape == (4, 4)
  assert len(np.unique(c0, axis=0)) == len(c0), "one point per voxel"
  assert len(m) > 0 and (np.diff(m[:, 0]) >= 0).all()
  d = np.linalg.norm(xyz0[m[:, 0]] @ trans[:3, :3].T + trans[:3, 3] - xyz1[m[:, 1]], axis=1)
  assert d.max() <= 1.5 * cfg.data.voxel_size + 1e-9
  loader = make_data_loader(cfg, cfg.trainer.batch_size)
  it = iter(loader)
  for _ in range(2 * len(dset)):  # > one epoch
    batch = next(it)
    assert batch["sinput0_C"].shape[1] == 4 and batch["sinput0_C"].dtype == torch.int32
    assert (np.diff(batch["correspondences"][:, 0].numpy()) >= 0).all()
  with pytest.raises(ValueError, match="does not exist"):
    make_data_loader(get_config(["data.dataset=Nope"]), 4)
  # device-side geometry launches HIP kernels in __getitem__: forked loader workers are refused, not left to crash
  cfg_dev = get_config(["data.dataset=ScanNetMatchPairDataset", "data.dataset_root_dir=%s" % tmp_path,
                        "data.scannet_match_dir=pairs.txt", "trainer.batch_size=2", "data.device_geometry=True"])
  with pytest.raises(ValueError, match="train_num_thread=0"):
    make_data_loader(cfg_dev, 2, num_threads=2)
  make_data_loader(cfg_dev, 2, num_threads=0)


def test_checkpoint_prefixes_and_kernel_order_switch(built_lib):
  """lib/checkpoint.py: 'module.' / 'encoder.' prefixes are stripped (downstream/semseg/lib/utils.py:23-29); the
  slice-order switch permutes exactly the 27-slice HYBRID kernels, by offset, and is its own inverse."""
  from oracle import sparse_ref as sr
  from pointcontrast_amd.lib import checkpoint as ck
  from pointcontrast_amd.lib.config import get_config
  from pointcontrast_amd.model import load_model
  import pointcontrast_amd.minkowski as ME
  torch.manual_seed(0)
  m = load_model("Res16UNet14")(3, 32, get_config([]), D=3)
  sd = {k: v.clone() for k, v in m.state_dict().items()}
  wrapped = {"module." + k: v for k, v in sd.items()}
  assert set(ck.load_state_with_same_shape(m, wrapped)) == set(sd)
  perm = ck.slice_permutation(ME.RegionType.HYPERCUBE, ME.RegionType.HYBRID)
  cube, hyb = sr.region_offsets(3, sr.HYPERCUBE), sr.region_offsets(3, sr.HYBRID)
  assert (cube[perm] == hyb).all() and sorted(perm.tolist()) == list(range(27))
  names = ck.hybrid_kernel_names(m)
  assert "block1.0.conv1.kernel" in names and "conv0p1s1.kernel" not in names and "final.kernel" not in names
  as_file = ck.convert_kernel_order(m, sd, "hypercube", inverse=True)   # what a hypercube-ordered file would hold
  assert not torch.equal(as_file["block1.0.conv1.kernel"], sd["block1.0.conv1.kernel"])
  assert torch.equal(as_file["conv0p1s1.kernel"], sd["conv0p1s1.kernel"])
  back = ck.convert_kernel_order(m, as_file, "hypercube")
  for k in sd:
    assert torch.equal(back[k], sd[k]), k
  m2 = load_model("Res16UNet14")(3, 32, get_config([]), D=3)
  ck.load_state(m2, {"module." + k: v for k, v in as_file.items()}, kernel_order="hypercube")
  assert torch.equal(m2.state_dict()["block4.0.conv2.kernel"], sd["block4.0.conv2.kernel"])


def test_semseg_scheduler_and_metrics_against_reference_source(built_lib):
  """downstream/semseg pieces restated in pointcontrast_amd/downstream/semseg.py: PolyLR against the reference's own
  class (lib/solvers.py imports only torch: executed from /root/reference where present), metrics against hand counts."""
  import importlib.util
  from pointcontrast_amd.downstream import semseg as ss
  p = torch.nn.Parameter(torch.zeros(3))
  opt = torch.optim.SGD([p], lr=0.1, momentum=0.9)
  sch = ss.PolyLR(opt, max_iter=100, power=0.9)
  lrs = []
  for _ in range(5):
    opt.step()
    sch.step()
    lrs.append(sch.get_last_lr()[0])
  assert abs(lrs[0] - 0.1 * (1 - 1 / 101) ** 0.9) < 1e-12 and lrs == sorted(lrs, reverse=True)
  ref_path = "/root/reference/downstream/semseg/lib/solvers.py"
  if os.path.isfile(ref_path):
    spec = importlib.util.spec_from_file_location("ref_semseg_solvers", ref_path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    opt2 = torch.optim.SGD([torch.nn.Parameter(torch.zeros(3))], lr=0.1, momentum=0.9)
    sch2 = mod.PolyLR(opt2, max_iter=100, power=0.9)
    ref_lrs = []
    for _ in range(5):
      opt2.step()
      sch2.step()
      ref_lrs.append(sch2.get_last_lr()[0])
    assert ref_lrs == lrs
  pred = np.array([0, 1, 1, 2, 2, 2, 0])
  label = np.array([0, 1, 2, 2, 2, 255, 1])
  h = ss.fast_hist(pred, label, 3)
  assert h.tolist() == [[1, 0, 0], [1, 1, 0], [0, 1, 2]]
  iu = ss.per_class_iu(h)
  assert np.allclose(iu, [1 / 2, 1 / 3, 2 / 3])
  assert abs(ss.precision_at_one(torch.from_numpy(pred), torch.from_numpy(label)) - 100 * 4 / 6) < 1e-4


def test_samplers_reproduce_the_reference_classes():
  """lib/data_sampler.py against the reference's own InfSampler / DistributedInfSampler (pc/lib/data_sampler.py imports
  only torch: executed from /root/reference where present): identical index streams AND identical positions of the
  permutation draws in the global torch RNG stream (other consumers of that stream -- the NCE trainer's Uniform
  draws -- sit between them)."""
  import importlib.util
  path = "/root/reference/pretrain/pointcontrast/lib/data_sampler.py"
  if not os.path.isfile(path):
    pytest.skip("/root/reference is not present on this host")
  spec = importlib.util.spec_from_file_location("ref_sampler", path)
  ref = importlib.util.module_from_spec(spec)
  spec.loader.exec_module(ref)
  from pointcontrast_amd.lib import data_sampler as ds
  data = list(range(11))

  def stream(mod, make, n=45):
    torch.manual_seed(3)
    s = make(mod)
    out = []
    for i in range(n):
      out.append(next(s))
      out.append(float(torch.rand(1)))  # another consumer of the RNG between two indices
    return out

  for shuffle in (False, True):
    assert stream(ref, lambda m: m.InfSampler(data, shuffle)) == stream(ds, lambda m: m.InfSampler(data, shuffle))
  for world in (2, 3, 4):
    for rank in range(world):
      mk = lambda m: m.DistributedInfSampler(data, world, rank, True)
      assert stream(ref, mk) == stream(ds, mk), (world, rank)
      assert len(mk(ref)) == len(mk(ds))


def test_pairs_scan_host_counts_runs_and_detects_unsorted_input(built_lib):
  """pcmi_pairs_scan_host (csrc/pairs.hip, host side of the device pair selection): number of unique queries of a sorted
  column 0 in one native pass, and the sortedness flag that sends unsorted correspondences to the host path."""
  import torch
  from pointcontrast_amd import functional as PF
  rng = np.random.RandomState(0)
  for n in (0, 1, 7, 70000, 300001):
    q = np.sort(rng.randint(0, max(n // 12, 1), n)).astype(np.int32)
    pp = torch.from_numpy(np.stack([q, rng.randint(0, 1000, n).astype(np.int32)], 1).copy())
    runs, ok = PF.pairs_scan_host(pp)
    assert ok and runs == len(np.unique(q)), (n, runs)
    if n > 10:
      pp[n // 2, 0] = -1
      assert PF.pairs_scan_host(pp)[1] is False


def test_flat_sgd_first_step_flag_survives_a_checkpoint(built_lib):
  """torch's SGD copies the gradient into the momentum buffer on the first step (no dampening); FlatSGD's buffers exist
  from the start, so "has stepped" travels in the state dict -- a checkpoint written before the first step must not
  resume with dampening applied (downstream fine-tuning: dampening 0.1, downstream/semseg/lib/solvers.py:52-60)."""
  import torch
  from pointcontrast_amd.lib import distributed as du
  from pointcontrast_amd.lib.solver import FlatSGD

  def make():
    torch.manual_seed(0)
    m = torch.nn.Linear(4, 3)
    flat = du.FlatParameters(m.parameters())
    return m, flat, FlatSGD(flat, lr=0.1, momentum=0.9, dampening=0.1)

  _, _, fresh = make()
  sd = fresh.state_dict()
  assert sd["param_groups"][0]["pcmi_steps_taken"] == 0
  _, _, o = make()
  o._fresh = False
  o.load_state_dict(sd)
  assert o._fresh and "pcmi_steps_taken" not in o.param_groups[0]
  # after a step
  _, flat, stepped = make()
  stepped._fresh = False
  flat.v.fill_(0.5)
  sd = stepped.state_dict()
  assert sd["param_groups"][0]["pcmi_steps_taken"] == 1
  _, flat2, o = make()
  o.load_state_dict(sd)
  assert not o._fresh and all(torch.equal(flat2.view(flat2.v, i), flat.view(flat.v, i)) for i in range(2))
  # a checkpoint without the flag (torch.optim.SGD's, or an older build's): derived from the buffers
  sd_old = stepped.state_dict()
  del sd_old["param_groups"][0]["pcmi_steps_taken"]
  _, _, o = make()
  o.load_state_dict(sd_old)
  assert not o._fresh
  flat.v.zero_()
  sd_zero = stepped.state_dict()
  del sd_zero["param_groups"][0]["pcmi_steps_taken"]
  _, _, o = make()
  o._fresh = False
  o.load_state_dict(sd_zero)
  assert o._fresh


def test_hardest_draws_consume_the_global_generator_in_the_reference_order():
  """HardestContrastiveLossTrainer._draw_hardest (run in a helper thread during the forward pass) = the three
  np.random.choice calls of pc/lib/ddp_trainer.py:198-206 in their order: candidates of cloud 0, of cloud 1, positives;
  no draw for the positives when there are no more than num_pos of them; injected draws consume nothing."""
  import numpy as np
  from pointcontrast_amd.lib.ddp_trainer import HardestContrastiveLossTrainer as T
  N0, N1, P, num_pos, num_hn = 5000, 4800, 9000, 1024, 256
  np.random.seed(7)
  sel0, sel1, pos = T._draw_hardest(N0, N1, P, num_pos, num_hn, None)
  after = np.random.rand()
  np.random.seed(7)
  r0 = np.random.choice(N0, min(N0, num_hn), replace=False)
  r1 = np.random.choice(N1, min(N1, num_hn), replace=False)
  rp = np.random.choice(P, num_pos, replace=False)
  assert np.array_equal(sel0, r0) and np.array_equal(sel1, r1) and np.array_equal(pos, rp) and after == np.random.rand()
  np.random.seed(7)
  s0, s1, none = T._draw_hardest(N0, N1, 1000, num_pos, num_hn, None)  # P <= num_pos: every pair is used, nothing drawn
  assert none is None and np.array_equal(s0, r0) and np.array_equal(s1, r1)
  np.random.seed(7)
  inj = dict(sel0=np.arange(3), sel1=np.arange(4), pos_sel=np.arange(5))
  got = T._draw_hardest(N0, N1, P, num_pos, num_hn, inj)
  assert all(np.array_equal(a, b) for a, b in zip(got, (inj["sel0"], inj["sel1"], inj["pos_sel"])))
  np.random.seed(7)
  first = np.random.rand()
  np.random.seed(7)
  T._draw_hardest(N0, N1, P, num_pos, num_hn, inj)
  assert np.random.rand() == first, "injected draws must not touch the global generator"


def test_build_flags_are_part_of_the_measurement_stamp(monkeypatch):
  """profiles/pmc_traffic.json is stamped with sources_digest(); a PMC pass belongs to a BUILD, so the digest must move with
  the compiler flags too (round 6: -fno-slp-vectorize changed every matrix-bound kernel's duration without touching a
  source line), and the product flags must carry that switch."""
  from pointcontrast_amd import build as b
  assert "-fno-slp-vectorize" in b.FLAGS and "--offload-arch=gfx950" in b.FLAGS
  d0 = b.sources_digest()
  monkeypatch.setattr(b, "FLAGS", [f for f in b.FLAGS if f != "-fno-slp-vectorize"])
  assert b.sources_digest() != d0
