"""CPU emulation of the index arithmetic of spconv32r_kernel (csrc/spconv32r.hip): group -> wave assignment, the table
look-ups of lane (i, kk) (offsets kk, kk + 4, ...), the occupied-offset mask from the four 16-lane ballot fields, the
compaction into the wave-private list, the weight staging (memory [c][n] and, for backward-data, [n][c]) with its
36-float rows, the fragment mapping of v_mfma_f32_16x16x4_f32 (lane (i, kk) supplies A[i][kk] and B[kk][i];
D[4 kk + r][i]) over the float4 a lane gathers, the refill slots of the kDepth-deep ring, and the output rows through
perm -- against a plain numpy contraction.  Also the bank claim of the header: the B fragment reads of a 32-lane half
wave hit 32 distinct banks."""
import numpy as np

LDB, W, D = 36, 12, 8


def _emulate(x, wmem, transposed, nbr, perm, wsel, K, KMAX, n_wg):
  """x [n_in, 32]; wmem [K, 32, 32] as it lies in memory; nbr [K, n_rows]; returns out [n_rows, 32]."""
  n_rows = nbr.shape[1]
  QN = (KMAX + 3) // 4
  # ---- weights -> "LDS": B_ks[c][n] at s_w[(ks * 32 + c) * LDB + n]
  s_w = np.full(KMAX * 32 * LDB, np.nan)
  WL = -(-KMAX * 256 // (W * 64))
  for t in range(W * 64):
    for u in range(WL):
      e = t + u * W * 64
      ks, r = e >> 8, e & 255
      if ks >= K:
        continue
      v = wmem[ks].reshape(-1)[(r >> 3) * 32 + (r & 7) * 4:(r >> 3) * 32 + (r & 7) * 4 + 4]
      if not transposed:  # memory [c][n]: a float4 of n
        base = (ks * 32 + (r >> 3)) * LDB + (r & 7) * 4
        s_w[base:base + 4] = v
      else:               # memory [n][c]: a float4 of c, scattered over four staged rows
        for q in range(4):
          s_w[(ks * 32 + (r & 7) * 4 + q) * LDB + (r >> 3)] = v[q]
  out = np.full((n_rows, 32), np.nan)
  n_groups = -(-n_rows // 16)
  seen = np.zeros(n_groups, int)
  for b in range(n_wg):
    for wave in range(W):
      g = wave * n_wg + b
      while g < n_groups:
        seen[g] += 1
        lanes = [(l & 15, l >> 4) for l in range(64)]
        nb = np.full((64, QN), -1, np.int64)
        for l, (i, kk) in enumerate(lanes):
          pos = g * 16 + i
          for q in range(QN):
            k = 4 * q + kk
            if pos < n_rows and k < K:
              nb[l, q] = nbr[k, pos]
        occ = 0
        for q in range(QN):
          ballot = [nb[l, q] >= 0 for l in range(64)]
          for c in range(4):
            if 4 * q + c < KMAX and any(ballot[16 * c:16 * c + 16]):
              occ |= 1 << (4 * q + c)
        cnt = bin(occ).count("1")
        s_off = np.full((KMAX, 16), -7, np.int64)
        s_ks = np.full(32, -1)
        for l, (i, kk) in enumerate(lanes):
          for q in range(QN):
            k = 4 * q + kk
            if k < KMAX and (occ >> k) & 1:
              j = bin(occ & ((1 << k) - 1)).count("1")
              s_off[j, i] = nb[l, q]  # (row index instead of byte offset; -1 = absent)
              if i == 0:
                s_ks[j] = wsel[k]
        assert (s_off[:cnt] != -7).all(), "a list entry was never written"
        # ring of D slots, refilled on every path; slot d serves offsets d, d + D, ...
        acc = np.zeros((2, 64, 4))
        ring = [None] * D
        fill = lambda j: [(s_off[j, i] if j < cnt else -1, kk) for (i, kk) in lanes]
        for d in range(D):
          ring[d] = fill(d)
        for j0 in range(0, cnt, D):
          for d in range(D):
            j = j0 + d
            if j < cnt:
              ks = s_ks[j]
              A = np.zeros((16, 32))
              for l, (i, kk) in enumerate(lanes):
                row, kk_ = ring[d][l]
                assert kk_ == kk
                if row >= 0:
                  A[i, 4 * kk:4 * kk + 4] = x[row, 4 * kk:4 * kk + 4]                      # float4 at byte 16 kk
                  A[i, 16 + 4 * kk:16 + 4 * kk + 4] = x[row, 16 + 4 * kk:16 + 4 * kk + 4]  # ... and at 64 + 16 kk
              for q in range(8):
                rowq = 16 * (q >> 2) + (q & 3)
                for ct in range(2):
                  # one MFMA step: contraction index kk; A[i][kk] = this lane's float value, B[kk][n = i]
                  Am = np.zeros((16, 4))
                  Bm = np.zeros((4, 16))
                  for l, (i, kk) in enumerate(lanes):
                    Am[i, kk] = A[i, 16 * (q >> 2) + 4 * kk + (q & 3)]
                    Bm[kk, i] = s_w[(ks * 32 + 4 * kk + rowq) * LDB + ct * 16 + i]
                  Dm = Am @ Bm
                  for l, (i, kk) in enumerate(lanes):
                    for r in range(4):
                      acc[ct, l, r] += Dm[4 * kk + r, i]
            ring[d] = fill(j + D)
        for l, (i, kk) in enumerate(lanes):
          for r in range(4):
            pos = g * 16 + 4 * kk + r
            if pos < n_rows:
              orow = perm[pos] if perm is not None else pos
              out[orow, i] = acc[0, l, r]
              out[orow, 16 + i] = acc[1, l, r]
        g += n_wg * W
  assert (seen == 1).all(), "every 16-row group is processed exactly once"
  return out


def _case(K, KMAX, n_rows, n_in, transposed, seed):
  rng = np.random.RandomState(seed)
  nbr = rng.randint(0, n_in, size=(K, n_rows))
  nbr[rng.rand(K, n_rows) < 0.45] = -1
  nbr[K // 2, : n_rows // 3] = -1          # an offset absent from whole groups
  if K > 3:
    nbr[1, :] = -1                         # ... and one absent everywhere
  perm = rng.permutation(n_rows)
  wsel = rng.permutation(K) if transposed else np.arange(K)
  x = rng.randn(n_in, 32)
  wlog = rng.randn(K, 32, 32)              # B_ks[c][n]
  wmem = wlog.transpose(0, 2, 1).copy() if transposed else wlog
  got = _emulate(x, wmem, transposed, nbr, perm, wsel, K, KMAX, n_wg=3)
  ref = np.zeros((n_rows, 32))
  for k in range(K):
    ok = nbr[k] >= 0
    ref[perm[ok]] += x[nbr[k][ok]] @ wlog[wsel[k]]
  assert not np.isnan(got).any()
  np.testing.assert_allclose(got, ref, rtol=1e-12, atol=1e-12)


def test_conv32r_index_arithmetic_k27_forward():
  _case(K=27, KMAX=27, n_rows=150, n_in=170, transposed=False, seed=0)


def test_conv32r_index_arithmetic_k27_backward_data():
  _case(K=27, KMAX=27, n_rows=99, n_in=99, transposed=True, seed=1)


def test_conv32r_index_arithmetic_k8():
  _case(K=8, KMAX=8, n_rows=70, n_in=300, transposed=False, seed=2)


def test_conv32r_b_fragment_reads_are_conflict_free():
  """ds_read_b32 of lane (i, kk) at float index (4 kk + row) * LDB + ct * 16 + i: the 32 lanes of a half wave (kk in
  {0, 1} or {2, 3}) must fall into 32 distinct 4-byte banks for every (row, ct)."""
  for row in range(32 - 12):
    for ct in range(2):
      for half in range(2):
        banks = {((4 * kk + row) * LDB + ct * 16 + i) % 32 for kk in (2 * half, 2 * half + 1) for i in range(16)}
        assert len(banks) == 32
