"""CPU emulation of the index arithmetic of stem_wgrad_mfma_kernel<3, 27> (csrc/spconv_wgrad.hip): the
3-channel stem's weight gradient  gW[k][c][n] = sum over rows r of x[nbr[k][r]][c] * g[r][n]  (reference: the weight
gradient of `conv0p1s1`, pc/model/res16unet.py:47-54,207, as MinkowskiConvolution's backward defines it) as ONE
[81 x rows] @ [rows x 32] product on v_mfma_f32_32x32x2_f32.

Followed as the kernel does it: the grid-stride loop over 64-row blocks, the thread -> (tile row, offset) assignment of
the tile build (lane = row of the block, wave w takes offsets w, w + 4, ...; offset-major neighbour table), the zero
columns 81..95, the operand roles of the instruction (lane (i, h) supplies A[i][h] = X81[row + h][32 mt + i] and
B[h][i] = g[row + h][n0 + i]; accumulator register j of lane (i, h) is D[(j & 3) + 8 (j >> 2) + 4 h][i]), the rows a wave
multiplies (16 w + 2 s + h), the wave-ordered reduction into one [81][cout] slab per workgroup and
stem_slab_reduce_kernel's 8-element x 32-lane sum.  Values are float64 here: the test is about WHERE every number goes."""
import numpy as np
import pytest

K, CIN, E, MT, RB = 27, 3, 81, 3, 64


def mfma_32x32x2(a, b, acc):
  """a, b: [64 lanes]; acc: [64 lanes][16].  D[m][n] += sum_h A[m][h] B[h][n]."""
  A, B = np.zeros((32, 2)), np.zeros((2, 32))
  for lane in range(64):
    i, h = lane & 31, lane >> 5
    A[i, h] = a[lane]
    B[h, i] = b[lane]
  D = A @ B
  out = acc.copy()
  for lane in range(64):
    i, h = lane & 31, lane >> 5
    for j in range(16):
      out[lane, j] += D[(j & 3) + 8 * (j >> 2) + 4 * h, i]
  return out


def workgroup(bx, by, n_wg, x, g, nbr, n_rows, cout):
  """One workgroup of stem_wgrad_mfma_kernel: its [81][32] contribution to the slab (columns n0 .. n0 + 31)."""
  n0 = 32 * by
  n_blocks = -(-n_rows // RB)
  acc = np.zeros((4, MT, 64, 16))  # [wave][mt][lane][j]
  tile = np.full((RB, 32 * MT + 1), np.nan)
  tile[:, E:] = 0.0  # the kernel's first loop: columns 81 .. 96 of every tile row
  for b in range(bx, n_blocks, n_wg):
    r0 = b * RB
    for wave in range(4):  # tile build: thread (wave, lane) -> tile row `lane`, offsets wave, wave + 4, ...
      for lane in range(64):
        for j in range((K + 3) // 4):
          k = wave + 4 * j
          if k >= K:
            continue
          ix = nbr[k * n_rows + r0 + lane] if r0 + lane < n_rows else -1
          tile[lane, CIN * k:CIN * k + CIN] = x[ix] if ix >= 0 else 0.0
    assert not np.isnan(tile).any(), "a tile cell no thread writes"
    for wave in range(4):
      for s in range(8):
        a = np.zeros((MT, 64))
        bv = np.zeros(64)
        for lane in range(64):
          i, h = lane & 31, lane >> 5
          row = 16 * wave + 2 * s + h
          bv[lane] = g[r0 + row, n0 + i] if r0 + row < n_rows else 0.0
          for mt in range(MT):
            a[mt, lane] = tile[row, 32 * mt + i]
        for mt in range(MT):
          acc[wave, mt] = mfma_32x32x2(a[mt], bv, acc[wave, mt])
  red = np.zeros((32 * MT, 32))
  for wave in range(4):  # wave order
    for mt in range(MT):
      for lane in range(64):
        i, h = lane & 31, lane >> 5
        for j in range(16):
          red[32 * mt + (j & 3) + 8 * (j >> 2) + 4 * h, i] += acc[wave, mt, lane, j]
  assert np.all(red[E:] == 0.0), "the padding rows 81..95 of the product must stay zero"
  return red[:E]


def slab_reduce(slabs, per):
  """stem_slab_reduce_kernel: 8 elements x 32 slab lanes per workgroup, lanes folded in lane order."""
  n_slabs = slabs.shape[0]
  flat = slabs.reshape(n_slabs, per)
  out = np.zeros(per)
  for block in range(-(-per // 8)):
    for el in range(8):
      e = block * 8 + el
      if e >= per:
        continue
      lanes = [sum(flat[b, e] for b in range(cl, n_slabs, 32)) for cl in range(32)]
      s = lanes[0]
      for q in range(1, 32):
        s += lanes[q]
      out[e] = s
  return out


@pytest.mark.parametrize("n_rows,cout,n_wg", [(200, 32, 3), (64, 32, 1), (65, 64, 2), (391, 32, 7)])
def test_stem_weight_gradient_index_arithmetic(n_rows, cout, n_wg):
  rng = np.random.RandomState(n_rows)
  x = rng.randn(n_rows, CIN)
  g = rng.randn(n_rows, cout)
  nbr = rng.randint(0, n_rows, size=(K, n_rows))
  nbr[rng.rand(K, n_rows) < 0.4] = -1  # absent neighbours
  nbr[K // 2] = np.arange(n_rows)      # the centre offset: every row is its own neighbour
  nbr_flat = nbr.reshape(-1)           # offset-major, as pcmi_kmap_t::nbr
  n_wg = min(n_wg, -(-n_rows // RB))
  slabs = np.zeros((n_wg, E, cout))
  for bx in range(n_wg):
    for by in range(cout // 32):
      slabs[bx, :, 32 * by:32 * by + 32] = workgroup(bx, by, n_wg, x, g, nbr_flat, n_rows, cout)
  gw = slab_reduce(slabs, E * cout).reshape(K, CIN, cout)
  ref = np.zeros((K, CIN, cout))
  for k in range(K):
    present = nbr[k] >= 0
    ref[k] = x[nbr[k][present]].T @ g[present]
  assert np.abs(gw - ref).max() <= 1e-12 * max(1.0, np.abs(ref).max())
