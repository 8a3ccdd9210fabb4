"""The arithmetic behind csrc/spconv_x3.hip, restated in numpy (CPU): an fp32 number is the exact sum of three bf16 numbers
(round-to-nearest-even splits), and a contraction computed from the six largest bf16 x bf16 cross products with fp32
accumulation (one rounding per 32-channel MFMA) is as close to the float64 result as an fp32 fused-multiply-add chain -- the
arithmetic of v_mfma_f32_16x16x4_f32 and of the reference's fp32 GEMMs.  The device kernel itself is compared with the fp32
kernel and with float64 in tests/test_gpu_parity.py::test_conv16_x3_split_precision_matches_fp32 (-m gpu)."""
import numpy as np


def bf16_rne(x):
  """float32 -> nearest bf16 (ties to even), returned as float32 (what v_cvt_pk_bf16_f32 computes)."""
  u = np.asarray(x, np.float32).view(np.uint32).astype(np.uint64)
  u = ((u + ((u >> 16) & 1) + 0x7FFF) >> 16) << 16
  return u.astype(np.uint32).view(np.float32)


def split3(x):
  h = bf16_rne(x)
  r1 = (x - h).astype(np.float32)
  m = bf16_rne(r1)
  r2 = (r1 - m).astype(np.float32)
  return h, m, bf16_rne(r2)


def contract(terms, K, chunk=32):
  """sum over 32-channel chunks; per chunk and term one exact product sum (bf16 x bf16 fits fp32, the MFMA adds them in
  wider precision), rounded to fp32 when it joins the accumulator."""
  acc = np.zeros((terms[0][0].shape[0], terms[0][1].shape[1]), np.float32)
  for c0 in range(0, K, chunk):
    for a, b in terms:
      acc = (acc.astype(np.float64) + a[:, c0:c0 + chunk].astype(np.float64) @ b[c0:c0 + chunk].astype(np.float64)).astype(np.float32)
  return acc


def test_three_bf16_terms_are_exact_and_six_products_match_fp32_class():
  rng = np.random.RandomState(0)
  K = 27 * 96  # the contraction of the level-1 96 -> 96 convolution
  A = (rng.randn(48, K) * np.exp(rng.randn(48, 1))).astype(np.float32)  # rows of very different scale
  B = (rng.randn(K, 96) / np.sqrt(K)).astype(np.float32)
  ah, am, al = split3(A)
  bh, bm, bl = split3(B)
  assert np.array_equal(ah.astype(np.float64) + am + al, A.astype(np.float64)), "x = h + m + l must be exact"
  assert np.array_equal(bh.astype(np.float64) + bm + bl, B.astype(np.float64))
  for part in (ah, am, al, bh, bm, bl):  # every term is a bf16 value
    assert np.array_equal(part.view(np.uint32) & 0xFFFF, np.zeros(part.shape, np.uint32))
  ref = A.astype(np.float64) @ B.astype(np.float64)
  scale = np.abs(ref).max()
  # the device kernel's order: the small cross terms first
  y6 = contract([(al, bh), (ah, bl), (am, bm), (am, bh), (ah, bm), (ah, bh)], K)
  chain = np.zeros_like(y6)
  for k in range(K):  # fp32 fused-multiply-add chain: one rounding per product
    chain = (chain.astype(np.float64) + A[:, k:k + 1].astype(np.float64) * B[k:k + 1].astype(np.float64)).astype(np.float32)
  e6, ec = np.abs(y6 - ref).max() / scale, np.abs(chain - ref).max() / scale
  e1 = np.abs(contract([(ah, bh)], K) - ref).max() / scale
  assert ec < 5e-6 and e6 <= max(2 * ec, 2e-6), (e6, ec)
  assert e1 > 1e-4, "plain bf16 operands would NOT meet the 1e-4 bar (%.1e)" % e1
