"""Row N2 on the device: checkpoints in the reference's layout (pc/lib/ddp_trainer.py:151-169) round-trip through the
trainer on the GPU, 'module.'-prefixed files load, and the 27-slice order switch of lib/checkpoint.py is semantically
right (a HYPERCUBE-enumerated kernel converted to the HYBRID enumeration computes the same convolution)."""
import os

import numpy as np
import pytest
import torch

from test_gpu_parity import DEV, assert_close

pytestmark = pytest.mark.gpu


def _trainer(cfg_over, batch):
  from pointcontrast_amd.lib.config import get_config
  from pointcontrast_amd.lib.ddp_data_loaders import FixedBatchLoader
  from pointcontrast_amd.lib.ddp_trainer import PointNCELossTrainer
  cfg = get_config(["net.model=Res16UNet14", "misc.nceT=0.4", "misc.npos=256", "opt.lr=0.1", "misc.prefetch=False"] + cfg_over)
  return PointNCELossTrainer(cfg, FixedBatchLoader([batch], batch_size=2))


def test_checkpoint_round_trip_on_the_device(tmp_path, monkeypatch):
  from pointcontrast_amd.lib import synthetic
  from pointcontrast_amd.lib.ddp_data_loaders import default_collate_pair_fn
  from pointcontrast_amd.lib.timer import AverageMeter, Timer
  import pointcontrast_amd.minkowski as ME
  monkeypatch.chdir(tmp_path)
  rng = np.random.RandomState(4)
  batch = default_collate_pair_fn([synthetic.make_pair_item(rng, 0.025, crop=0.6) for _ in range(2)])
  torch.manual_seed(1)
  tr = _trainer([], batch)
  it, timers = iter(tr.data_loader), [AverageMeter(), Timer(), Timer()]
  for _ in range(2):
    tr._train_iter(it, timers)
  tr.scheduler.step()
  tr._save_checkpoint(2, "checkpoint_2")
  assert os.path.islink("weights/weights.pth") and os.readlink("weights/weights.pth") == "checkpoint_2.pth"
  state = torch.load("weights/weights.pth", map_location="cpu", weights_only=False)
  assert set(state) == {"curr_iter", "state_dict", "optimizer", "scheduler", "config"} and state["curr_iter"] == 2
  assert state["state_dict"]["conv0p1s1.kernel"].shape == (27, 3, 32) and state["state_dict"]["final.kernel"].dim() == 2
  tr.model.eval()
  st = ME.SparseTensor(batch["sinput0_F"], coords=batch["sinput0_C"]).to(DEV)
  with torch.no_grad():
    want = tr.model(st).F.clone()
  # a fresh trainer in the same directory resumes from weights/weights.pth (pc/lib/ddp_trainer.py:118-131)
  torch.manual_seed(99)
  tr2 = _trainer([], batch)
  assert tr2.curr_iter == 2
  assert tr2.scheduler.get_last_lr() == tr.scheduler.get_last_lr()
  assert torch.equal(tr2.flat.w, tr.flat.w) and torch.equal(tr2.flat.v, tr.flat.v), "weights / SGD momentum not restored"
  tr2.model.eval()
  with torch.no_grad():
    got = tr2.model(ME.SparseTensor(batch["sinput0_F"], coords=batch["sinput0_C"]).to(DEV)).F
  assert torch.equal(got, want), "features differ after the round trip"
  # both continue identically (same batch, same draws)
  pp = batch["correspondences"].numpy()
  nq = len(np.unique(pp[:, 0]))
  draws = dict(uniform=torch.rand(nq, generator=torch.Generator().manual_seed(5)),
               sampled_inds=np.random.RandomState(5).choice(nq, 256, replace=False))
  l1 = float(tr._train_iter(iter(tr.data_loader), timers, draws=draws)["loss"])
  l2 = float(tr2._train_iter(iter(tr2.data_loader), timers, draws=draws)["loss"])
  assert l1 == l2
  # misc.weight: a DataParallel-style file ('module.' prefix) with a head of another width, lenient loading
  wrapped = {"module." + k: v for k, v in state["state_dict"].items()}
  torch.save({"state_dict": wrapped}, "wrapped.pth")
  os.remove("weights/weights.pth")
  os.remove("weights/checkpoint_2.pth")
  tr3 = _trainer(["misc.weight=wrapped.pth", "misc.lenient_weight_loading=True", "net.model_n_out=64"], batch)
  assert torch.equal(tr3.model.state_dict()["block4.0.conv2.kernel"].cpu(), state["state_dict"]["block4.0.conv2.kernel"])
  assert tr3.model.final.kernel.shape[1] == 64  # the 32-wide head of the file was skipped


def test_kernel_order_switch_is_semantically_right():
  """A 3^3 conv whose 27 slices are enumerated in HYPERCUBE order (region 0) and the same weights converted by
  lib/checkpoint.convert_kernel_order into the HYBRID enumeration (region 3) are the same convolution."""
  import pointcontrast_amd.minkowski as ME
  from pointcontrast_amd.lib import checkpoint as ck
  from pointcontrast_amd.model.modules.common import ConvType, conv
  from helpers import surface_coords
  torch.manual_seed(2)
  cube = conv(32, 64, 3, conv_type=ConvType.SPATIAL_HYPERCUBE, D=3).to(DEV)
  hyb = conv(32, 64, 3, conv_type=ConvType.SPATIAL_HYPERCUBE_TEMPORAL_HYPERCROSS, D=3).to(DEV)

  class Holder(torch.nn.Module):
    def __init__(self):
      super().__init__()
      self.c = hyb

  h = Holder()
  converted = ck.convert_kernel_order(h, {"c.kernel": cube.kernel.detach().clone()}, "hypercube")
  h.load_state_dict(converted)
  C = surface_coords(20, 2, seed=3)
  x = torch.randn(len(C), 32)
  y0 = cube(ME.SparseTensor(x, coords=torch.from_numpy(C)).to(DEV)).F
  y1 = hyb(ME.SparseTensor(x, coords=torch.from_numpy(C)).to(DEV)).F
  assert_close(y1, y0, 1e-6, "hypercube-ordered kernel converted to the hybrid enumeration")
