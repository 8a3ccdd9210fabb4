"""CPU emulation of the index arithmetic of csrc/nce_x3.hip (PointInfoNCE on the matrix cores): the two operand images of
nce_pack_kernel (row fragments; slot-ordered column fragments), the fragment layouts of v_mfma_f32_16x16x32_bf16 (lane
(i, kk) supplies A[i][8kk..8kk+7] and B[8kk..8kk+7][i]; D[4kk + r][i]), and the claim the kernel is built on: the D
layout of the first product S^T = other . own^T IS the A layout of the second d_own = W . other once the second's B
operand lists the 32 rows of a chunk in the order slot(kk, e) = 16 (e >> 2) + 4 kk + (e & 3) -- so the softmax weights
go from accumulator to operand inside the lane.  Workgroup / wave / chunk decomposition, the diagonal chunk, zero
padding, the split of the other operand over gridDim.y and the forward's log2-domain online log-sum-exp with its
"nothing seen yet" floor are followed as the kernels do them; values are fp64 here (the three-term bf16 split is
tests/test_x3_numerics.py).  Against torch's cross entropy and its autograd."""
import numpy as np
import pytest
import torch

OWN, CHUNK = 128, 32
LOG2E, LN2, FLOOR = 1.4426950408889634, 0.6931471805599453, -1e30


def pack_images(x, n_pad):
  """nce_pack_kernel: (row fragments [16-row tile][lane][8], column fragments [32-row chunk][16-column tile][lane][8])."""
  n, c = x.shape
  rows = np.zeros((n_pad // 16, 64, 8))
  cols = np.zeros((n_pad // CHUNK, 2, 64, 8))
  for tile in range(n_pad // 16):
    for lane in range(64):
      i, kk = lane & 15, lane >> 4
      r = tile * 16 + i
      if r < n and 8 * kk < c:
        rows[tile, lane] = x[r, 8 * kk:8 * kk + 8]
  for chunk in range(n_pad // CHUNK):
    for ct in range(2):
      for lane in range(64):
        i, kk = lane & 15, lane >> 4
        d = 16 * ct + i
        for e in range(8):
          b = chunk * CHUNK + 16 * (e >> 2) + 4 * kk + (e & 3)
          if b < n and d < c:
            cols[chunk, ct, lane, e] = x[b, d]
  return rows, cols


def mfma(a_frag, b_frag, acc):
  """One v_mfma_f32_16x16x32: a_frag / b_frag [64 lanes][8], acc [64 lanes][4] (D[4 kk + r][i] in lane (i, kk))."""
  A, B = np.zeros((16, 32)), np.zeros((32, 16))
  for lane in range(64):
    i, kk = lane & 15, lane >> 4
    A[i, 8 * kk:8 * kk + 8] = a_frag[lane]
    B[8 * kk:8 * kk + 8, i] = b_frag[lane]
  D = A @ B
  out = acc.copy()
  for lane in range(64):
    i, kk = lane & 15, lane >> 4
    for r in range(4):
      out[lane, r] += D[4 * kk + r, i]
  return out


def plan(n, n_cu=256):
  """nce_x3_plan."""
  n_pad = -(-n // OWN) * OWN
  tiles, chunks = n_pad // OWN, -(-n // CHUNK)
  want = -(-n_cu // tiles)
  splits = max(1, min(want, -(-chunks // 4), 32))
  span = -(-chunks // splits) * CHUNK
  return n_pad, tiles, -(-chunks * CHUNK // span), span


def emulate_backward(q, k, lse, inv_T, gs):
  n, c = q.shape
  n_pad, tiles, splits, span = plan(n)
  img = {"q": pack_images(q, n_pad), "k": pack_images(k, n_pad)}
  n32 = -(-n // CHUNK) * CHUNK
  grads = []
  for side, (own, oth) in enumerate((("q", "k"), ("k", "q"))):
    d_own = np.zeros((n_pad, c))
    own_rows, (oth_rows, oth_cols) = img[own][0], img[oth]
    for tile in range(tiles):
      for split in range(splits):  # the partials of the splits are added in split order by the last workgroup to arrive
        part = np.zeros((OWN, 32))
        for wave in range(4):
          a_base = tile * OWN + wave * 32
          dacc = np.zeros((2, 2, 64, 4))
          for b0 in range(split * span, min(n32, split * span + span), CHUNK):
            for g in range(2):
              bo = own_rows[(a_base >> 4) + g]
              w = np.zeros((64, 8))
              for tt in range(2):
                sacc = mfma(oth_rows[b0 // 16 + tt], bo, np.zeros((64, 4)))
                for lane in range(64):
                  i, kk = lane & 15, lane >> 4
                  a = a_base + 16 * g + i
                  for r in range(4):
                    b = b0 + 16 * tt + 4 * kk + r
                    ls = lse[min(b, n - 1)] if side else lse[min(a, n - 1)]
                    p = 2.0 ** (sacc[lane, r] * inv_T * LOG2E - ls * LOG2E)
                    diag = b0 == a_base and 16 * tt + 4 * kk + r == 16 * g + i
                    w[lane, 4 * tt + r] = ((p - 1.0) if diag else p) * gs  # the accumulator of tile tt IS slots 4 tt .. 4 tt + 3
              for ct in range(c // 16):
                dacc[g, ct] = mfma(w, oth_cols[b0 // CHUNK, ct], dacc[g, ct])
          for g in range(2):
            for ct in range(c // 16):
              for lane in range(64):
                i, kk = lane & 15, lane >> 4
                for r in range(4):
                  part[wave * 32 + 16 * g + 4 * kk + r, 16 * ct + i] = dacc[g, ct, lane, r]
        d_own[tile * OWN:(tile + 1) * OWN] += part[:, :c]
    grads.append(d_own[:n])
  return grads


def emulate_forward(q, k, inv_T):
  n, c = q.shape
  n_pad, tiles, splits, span = plan(n)
  q_rows, k_rows = pack_images(q, n_pad)[0], pack_images(k, n_pad)[0]
  n32 = -(-n // CHUNK) * CHUNK
  pm, pl = np.full((splits, n_pad), FLOOR), np.zeros((splits, n_pad))
  for tile in range(tiles):
    for split in range(splits):
      for wave in range(4):
        a_base = tile * OWN + wave * 32
        m, l = np.full((2, 64), FLOOR), np.zeros((2, 64))
        for b0 in range(split * span, min(n32, split * span + span), CHUNK):
          for g in range(2):
            v = np.zeros((64, 8))
            for tt in range(2):
              sacc = mfma(k_rows[b0 // 16 + tt], q_rows[(a_base >> 4) + g], np.zeros((64, 4)))
              for lane in range(64):
                kk = lane >> 4
                for r in range(4):
                  x = sacc[lane, r] * inv_T * LOG2E
                  v[lane, 4 * tt + r] = FLOOR if b0 + 16 * tt + 4 * kk + r >= n else x
            mn = np.maximum(m[g], v.max(1))
            l[g] = l[g] * 2.0 ** (m[g] - mn) + (2.0 ** (v - mn[:, None])).sum(1)
            m[g] = mn
        for g in range(2):
          for d in (16, 32):  # the four lane quads of a column hold disjoint key rows
            om, ol = m[g][np.arange(64) ^ d], l[g][np.arange(64) ^ d]
            mn = np.maximum(m[g], om)
            l[g] = l[g] * 2.0 ** (m[g] - mn) + ol * 2.0 ** (om - mn)
            m[g] = mn
          for i in range(16):
            pm[split, a_base + 16 * g + i], pl[split, a_base + 16 * g + i] = m[g][i], l[g][i]
  lse = np.zeros(n)
  for a in range(n):
    mm, ll = FLOOR, 0.0
    for sp in range(splits):
      mn = max(mm, pm[sp, a])
      ll = ll * 2.0 ** (mm - mn) + pl[sp, a] * 2.0 ** (pm[sp, a] - mn)
      mm = mn
    lse[a] = (mm + np.log2(ll)) * LN2
  loss = float(np.mean(lse - (q * k).sum(1) * inv_T))
  return lse, loss


@pytest.mark.parametrize("n,c,T", [(1, 32, 0.4), (33, 32, 0.4), (130, 16, 0.07), (200, 32, 0.07)])
def test_nce_x3_index_arithmetic_matches_cross_entropy(n, c, T):
  rng = np.random.RandomState(n)
  q = rng.randn(n, c)
  q /= np.linalg.norm(q, axis=1, keepdims=True)
  k = q + 0.3 * rng.randn(n, c)
  k /= np.linalg.norm(k, axis=1, keepdims=True)
  qt, kt = torch.tensor(q, requires_grad=True), torch.tensor(k, requires_grad=True)
  logits = qt @ kt.t() / T
  ref = torch.nn.functional.cross_entropy(logits, torch.arange(n))
  (ref * 1.7).backward()
  lse, loss = emulate_forward(q, k, 1.0 / T)
  assert abs(loss - float(ref.detach())) <= 1e-12 * max(1.0, abs(float(ref.detach())))
  assert np.allclose(lse, torch.logsumexp(logits, 1).detach().numpy(), rtol=0, atol=1e-12)
  dq, dk = emulate_backward(q, k, lse, 1.0 / T, 1.7 / T / n)
  assert np.allclose(dq, qt.grad.numpy(), rtol=0, atol=1e-13), np.abs(dq - qt.grad.numpy()).max()
  assert np.allclose(dk, kt.grad.numpy(), rtol=0, atol=1e-13), np.abs(dk - kt.grad.numpy()).max()


def test_slot_order_is_a_permutation_of_the_chunk():
  seen = sorted(16 * (e >> 2) + 4 * kk + (e & 3) for kk in range(4) for e in range(8))
  assert seen == list(range(32))


@pytest.mark.parametrize("n", [1, 64, 300, 1000, 4096, 4097, 8192, 100000])
def test_plan_covers_every_key_chunk_once(n):
  n_pad, tiles, splits, span = plan(n)
  assert n_pad % OWN == 0 and n_pad >= n and tiles * OWN == n_pad and span % CHUNK == 0 and 1 <= splits <= 32
  n32 = -(-n // CHUNK) * CHUNK
  covered = [b0 for s in range(splits) for b0 in range(s * span, min(n32, s * span + span), CHUNK)]
  assert covered == list(range(0, n32, CHUNK))
  assert all(s * span < n32 for s in range(splits)), "a split without a chunk"
