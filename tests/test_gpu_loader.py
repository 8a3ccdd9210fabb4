"""Row N1 on the device: first-occurrence voxelisation and radius correspondence search (csrc/loader.hip) bit-exact
against the CPU restatement oracle/loader_ref.py (pc/lib/ddp_data_loaders.py:36-49, :228-241), incl. empty / single /
duplicate inputs, and through the reference's dataset class."""
import numpy as np
import pytest
import torch

from test_oracle_loader import _pair

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed,crop,voxel", [(0, 0.5, 0.025), (1, 9.0, 0.025), (2, 9.0, 0.05), (3, 1.2, 0.01)])
def test_voxelize_and_match_are_bit_exact(seed, crop, voxel):
  from oracle import loader_ref as lf
  from pointcontrast_amd.lib import device_loader as dl
  a, b, T = _pair(seed, crop)
  sel, coords = dl.sparse_quantize_index(a, voxel, return_coords=True)
  ref = lf.sparse_quantize_index(a, voxel)
  assert sel.dtype == np.int64 and (sel == ref).all()
  assert (coords == np.floor(a[ref] / voxel).astype(np.int32)).all()
  a, b = a[ref], b[lf.sparse_quantize_index(b, voxel)]
  r = 1.5 * voxel
  got = dl.get_matching_indices(a, b, T, r)
  want = lf.match_radius(a, T, b, r)
  assert got.shape == want.shape and (got == want).all()
  assert len(got) > 100 and (np.diff(got[:, 0]) >= 0).all()


def test_device_resident_item_geometry_matches_the_numpy_form():
  """pair_geometry_device (the loader stage with its outputs left on the device) = the numpy-in / numpy-out wrappers =
  the oracle: surviving points, voxel coordinates, correspondences."""
  from oracle import loader_ref as lf
  from pointcontrast_amd.lib import device_loader as dl
  a, b, T = _pair(5, 9.0)
  voxel, r = 0.025, 1.5 * 0.025
  out = dl.pair_geometry_device(a, b, T, voxel, r)
  sa, sb = lf.sparse_quantize_index(a, voxel), lf.sparse_quantize_index(b, voxel)
  assert out["xyz0"].is_cuda and out["matches"].is_cuda and out["matches"].dtype == torch.int32
  assert (out["xyz0"].cpu().numpy() == a[sa]).all() and (out["xyz1"].cpu().numpy() == b[sb]).all()
  assert (out["coords0"].cpu().numpy() == np.floor(a[sa] / voxel).astype(np.int32)).all()
  assert (out["coords1"].cpu().numpy() == np.floor(b[sb] / voxel).astype(np.int32)).all()
  want = lf.match_radius(a[sa], T, b[sb], r)
  assert (out["matches"].cpu().numpy().astype(np.int64) == want).all() and len(want) > 100


def test_loader_kernels_edge_cases():
  from oracle import loader_ref as lf
  from pointcontrast_amd._lib import PcmiError
  from pointcontrast_amd.lib import device_loader as dl
  assert len(dl.sparse_quantize_index(np.zeros((0, 3)), 0.025)) == 0
  one = np.array([[0.3, -0.2, 1.7]])
  assert dl.sparse_quantize_index(one, 0.025).tolist() == [0]
  dup = np.array([[0.01, 0.01, 0.01], [0.02, 0.02, 0.02], [-0.01, 0.0, 0.0], [0.011, 0.012, 0.013], [-0.02, 0.0, 0.0]])
  assert dl.sparse_quantize_index(dup, 0.025).tolist() == lf.sparse_quantize_index(dup, 0.025).tolist() == [0, 2]
  I = np.eye(4)
  assert dl.get_matching_indices(np.zeros((0, 3)), one, I, 0.1).shape == (0, 2)
  assert dl.get_matching_indices(one, np.zeros((0, 3)), I, 0.1).shape == (0, 2)
  assert dl.get_matching_indices(one, one, I, 0.1).tolist() == [[0, 0]]
  far = np.array([[5.0, 5.0, 5.0]])
  assert dl.get_matching_indices(one, far, I, 0.1).shape == (0, 2)
  # negative coordinates / cells straddling zero
  rng = np.random.RandomState(0)
  p = rng.uniform(-0.2, 0.2, (3000, 3))
  q = p + rng.normal(0, 0.01, p.shape)
  assert (dl.get_matching_indices(p, q, I, 0.03) == lf.match_radius(p, I, q, 0.03)).all()
  with pytest.raises(PcmiError, match="matches"):  # a radius far above the point spacing is refused, not truncated
    dl.get_matching_indices(p, q, I, 0.5)


def test_reference_dataset_with_device_geometry(tmp_path):
  """ScanNetMatchPairDataset (pc/lib/ddp_data_loaders.py:119-270) with data.device_geometry=True yields exactly the item
  the host path yields (same random transforms)."""
  import random
  from pointcontrast_amd.lib import synthetic
  from pointcontrast_amd.lib.config import get_config
  from pointcontrast_amd.lib.ddp_data_loaders import dataset_str_mapping
  rng = np.random.RandomState(0)
  a, b = synthetic.make_frame_pair(rng)
  np.savez(tmp_path / "a.npz", pcd=a)
  np.savez(tmp_path / "b.npz", pcd=b)
  (tmp_path / "pairs.txt").write_text("a.npz b.npz 0.6\n")
  items = []
  for devgeo in (False, True):
    cfg = get_config(["data.dataset=ScanNetMatchPairDataset", "data.dataset_root_dir=%s" % tmp_path,
                      "data.scannet_match_dir=pairs.txt", "data.device_geometry=%s" % devgeo])
    d = dataset_str_mapping["ScanNetMatchPairDataset"](phase="train", config=cfg, random_scale=False, manual_seed=True)
    random.seed(0)
    np.random.seed(0)
    items.append(d[0])
  for x, y in zip(*items):
    assert np.asarray(x).shape == np.asarray(y).shape and (np.asarray(x) == np.asarray(y)).all()
  assert len(items[0][6]) > 10000
