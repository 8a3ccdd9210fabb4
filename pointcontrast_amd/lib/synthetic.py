"""Seeded synthetic ScanNet-frame-shaped scene pairs.

There is no dataset (and no network) here, so the benchmark / tests feed the
path with pairs that have the *shape* of the reference's input: two depth
frames (640x480, fx=fy=577) of the same furnished room seen from nearby poses,
back-projected to world space, then pushed through exactly the reference's
per-item chain and collate:
  random rotation about a random axis + centring  (pc/lib/ddp_data_loaders.py:137-142, :217-223)
  voxel first-occurrence sub-sampling             (:228-239, ME.utils.sparse_quantize)
  correspondences within 1.5 voxel                (:36-49, :241)
  feats = ones (+ N(0, 0.01) w.p. 0.95)           (:248-252, pc/lib/transforms.py:21-30)
  coords = floor(xyz / voxel)                     (:258-259)
  collate: batch id first, offsets on matches     (:52-112)
Everything is numpy on the host: this is the loader side of the boundary
(SURVEY.md 8f row N1), not the device hot path.
"""
import numpy as np
from scipy.spatial import cKDTree


def _rotation(axis, theta):
  """Rodrigues form of expm(cross(I, axis/|axis| * theta)) (ddp_data_loaders.py:115-116)."""
  a = axis / np.linalg.norm(axis)
  K = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
  return np.eye(3) + np.sin(theta) * K + (1 - np.cos(theta)) * (K @ K)


def sample_random_trans(pcd, rng, rotation_range=360):
  T = np.eye(4)
  R = _rotation(rng.rand(3) - 0.5, rotation_range * np.pi / 180.0 * (rng.rand(1)[0] - 0.5))
  T[:3, :3] = R
  T[:3, 3] = R.dot(-np.mean(pcd, axis=0))
  return T


def _make_room(rng):
  """Axis-aligned boxes: the room shell is handled separately; returns furniture [n,2,3]."""
  room = np.array([5.0, 4.0, 2.7])
  boxes = []
  for _ in range(rng.randint(6, 10)):
    size = np.array([rng.uniform(0.4, 1.6), rng.uniform(0.4, 1.2), rng.uniform(0.4, 1.4)])
    lo = np.array([rng.uniform(0.1, room[0] - size[0] - 0.1), rng.uniform(0.1, room[1] - size[1] - 0.1), 0.0])
    boxes.append(np.stack([lo, lo + size]))
  return room, np.asarray(boxes)


def _raycast(origin, dirs, room, boxes):
  """Nearest hit of rays (origin inside the room) with the shell and the boxes."""
  inv = 1.0 / np.where(np.abs(dirs) < 1e-12, 1e-12, dirs)
  # shell: exit distance of the room AABB
  t_hi = np.maximum((0.0 - origin) * inv, (room - origin) * inv)
  t = np.min(t_hi, axis=1)
  for b in boxes:
    t0 = (b[0] - origin) * inv
    t1 = (b[1] - origin) * inv
    tn = np.max(np.minimum(t0, t1), axis=1)
    tf = np.min(np.maximum(t0, t1), axis=1)
    hit = (tn < tf) & (tn > 1e-3)
    t = np.where(hit & (tn < t), tn, t)
  return t


def _camera_rays(R, width=640, height=480, f=577.0):
  u, v = np.meshgrid(np.arange(width) + 0.5 - width / 2, np.arange(height) + 0.5 - height / 2)
  d = np.stack([u / f, v / f, np.ones_like(u)], -1).reshape(-1, 3)
  d /= np.linalg.norm(d, axis=1, keepdims=True)
  return d @ R.T


def _look_at(yaw, pitch):
  cy, sy, cp, sp = np.cos(yaw), np.sin(yaw), np.cos(pitch), np.sin(pitch)
  fwd = np.array([cy * cp, sy * cp, -sp])
  right = np.array([sy, -cy, 0.0])
  down = np.cross(fwd, right)
  return np.stack([right, down, fwd], 1)  # camera (x right, y down, z fwd) -> world


def make_frame_pair(rng, noise=0.003, min_cells=16000, max_cells=24500):
  """Two world-space point clouds ('pcd' arrays of the reference's npz files).

  Poses are rejection-sampled so that each frame covers 16k-24.5k cells of a
  2.5 cm grid (ScanNet frames: ~20k), independent of the voxel size used later."""
  while True:
    room, boxes = _make_room(rng)
    cam = np.array([rng.uniform(0.6, 4.4), rng.uniform(0.6, 3.4), rng.uniform(1.2, 1.7)])
    to_centre = np.array([2.5, 2.0]) - cam[:2]
    yaw = np.arctan2(to_centre[1], to_centre[0]) + rng.uniform(-0.5, 0.5)
    pitch = rng.uniform(0.15, 0.45)
    out = []
    for i in range(2):
      o = cam + (rng.uniform(-0.4, 0.4, 3) * np.array([1, 1, 0.2]) if i else 0)
      o = np.clip(o, [0.3, 0.3, 1.0], [4.7, 3.7, 1.9])
      y = yaw + (rng.uniform(-0.4, 0.4) if i else 0)
      p = pitch + (rng.uniform(-0.1, 0.1) if i else 0)
      dirs = _camera_rays(_look_at(y, p))
      t = _raycast(o, dirs, room, boxes)
      t = t + rng.normal(0, 1, t.shape) * noise * t
      out.append(o + dirs * t[:, None])
    cells = [len(sparse_quantize_index(x / 0.025)) for x in out]
    if min(cells) >= min_cells and max(cells) <= max_cells:
      return out[0], out[1]


def sparse_quantize_index(xyz_over_voxel):
  """Ascending first-occurrence indices of distinct voxels (ME.utils.sparse_quantize(..., return_index=True))."""
  q = np.floor(xyz_over_voxel).astype(np.int64)
  q -= q.min(0)
  m = q.max(0) + 1
  key = (q[:, 0] * m[1] + q[:, 1]) * m[2] + q[:, 2]
  _, first = np.unique(key, return_index=True)
  return np.sort(first)


def make_pair_item(rng, voxel_size=0.025, search_mult=1.5, crop=None, jitter=True):
  """One dataset item: (xyz0, xyz1, coords0, coords1, feats0, feats1, matches, trans)."""
  xyz0, xyz1 = make_frame_pair(rng)
  if crop is not None:  # small plumbing-size pairs (BASELINE config #1)
    c = xyz0[rng.randint(len(xyz0))]
    xyz0 = xyz0[np.linalg.norm(xyz0 - c, axis=1) < crop]
    xyz1 = xyz1[np.linalg.norm(xyz1 - c, axis=1) < crop]
  T0 = sample_random_trans(xyz0, rng)
  T1 = sample_random_trans(xyz1, rng)
  trans = T1 @ np.linalg.inv(T0)
  xyz0 = xyz0 @ T0[:3, :3].T + T0[:3, 3]
  xyz1 = xyz1 @ T1[:3, :3].T + T1[:3, 3]
  xyz0 = xyz0[sparse_quantize_index(xyz0 / voxel_size)]
  xyz1 = xyz1[sparse_quantize_index(xyz1 / voxel_size)]
  src = xyz0 @ trans[:3, :3].T + trans[:3, 3]
  tree = cKDTree(xyz1)
  nb = tree.query_ball_point(src, search_mult * voxel_size)
  cnt = np.fromiter((len(x) for x in nb), np.int64, len(nb))
  ii = np.repeat(np.arange(len(nb)), cnt)
  jj = np.concatenate([np.sort(np.asarray(x, np.int64)) for x in nb]) if cnt.sum() else np.zeros(0, np.int64)
  matches = np.stack([ii, jj], 1)
  feats = []
  for n in (len(xyz0), len(xyz1)):
    f = np.ones((n, 3))
    if jitter and rng.rand() < 0.95:
      f = f + rng.normal(0, 0.01, f.shape)
    feats.append(f)
  coords0 = np.floor(xyz0 / voxel_size)
  coords1 = np.floor(xyz1 / voxel_size)
  return xyz0, xyz1, coords0, coords1, feats[0], feats[1], matches, trans


def collate_pairs(items):
  """default_collate_pair_fn restated (ddp_data_loaders.py:52-112): numpy in, dict of numpy out."""
  C0, C1, F0, F1, M, X0, X1, T, lens = [], [], [], [], [], [], [], [], []
  s0 = s1 = 0
  for b, (xyz0, xyz1, c0, c1, f0, f1, m, tr) in enumerate(items):
    n0, n1 = len(c0), len(c1)
    C0.append(np.concatenate([np.full((n0, 1), b), c0], 1))
    C1.append(np.concatenate([np.full((n1, 1), b), c1], 1))
    F0.append(f0); F1.append(f1); X0.append(xyz0); X1.append(xyz1); T.append(tr)
    if len(m) == 0:
      m = np.zeros((1, 2), np.int64)
    M.append(m + np.array([[s0, s1]]))
    lens.append([n0, n1])
    s0 += n0; s1 += n1
  return {
      "pcd0": np.concatenate(X0).astype(np.float32), "pcd1": np.concatenate(X1).astype(np.float32),
      "sinput0_C": np.concatenate(C0).astype(np.int32), "sinput0_F": np.concatenate(F0).astype(np.float32),
      "sinput1_C": np.concatenate(C1).astype(np.int32), "sinput1_F": np.concatenate(F1).astype(np.float32),
      "correspondences": np.concatenate(M).astype(np.int32),
      "T_gt": np.concatenate(T).astype(np.float32), "len_batch": lens,
  }


def make_batch(seed=0, batch_size=4, voxel_size=0.025, crop=None):
  rng = np.random.RandomState(seed)
  return collate_pairs([make_pair_item(rng, voxel_size, crop=crop) for _ in range(batch_size)])
