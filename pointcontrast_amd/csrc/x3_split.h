// Three-term bf16 split of fp32 operands for the bf16 matrix cores (shared by spconv_x3.hip and spconv_wgrad_x3.hip).
#pragma once
#include <hip/hip_runtime.h>

namespace pcmi {

typedef float v4f __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int kKC = 32;  // contraction channels per staged chunk = per MFMA

// two floats -> two bf16 (round to nearest even) packed in one dword, first value in the low half
__device__ __forceinline__ unsigned cvt_pk_bf16(float a, float b) {
  const f32x2 v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}
__device__ __forceinline__ float bf16_lo(unsigned p) { return __builtin_bit_cast(float, p << 16); }
__device__ __forceinline__ float bf16_hi(unsigned p) { return __builtin_bit_cast(float, p & 0xffff0000u); }

// 8 floats (element e = x0[e] for e < 4, x1[e - 4] otherwise) -> three vectors of 8 bf16 with x = h + m + l.
// Special values: an element that rounds to +-inf in bf16 (|x| >= 3.39e38, or inf itself) has h = inf and residuals
// x - inf = -inf / NaN, so its products come out NaN where the fp32 kernel yields +-inf.  Zeroing the residuals of such
// an element does not restore inf either: inf * b_h + inf * b_m + inf * b_l mixes signs (the residual terms of a weight
// have arbitrary sign) and is NaN again.  An overflowed activation therefore shows up as NaN in its whole output tile
// (documented in INTEGRATION.md; PCMI_CONV16_X3=0 selects the fp32-MFMA kernel to localise a divergence).
__device__ __forceinline__ void split3(const v4f& x0, const v4f& x1, u32x4& h, u32x4& m, u32x4& l) {
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const float xa = p < 2 ? x0[2 * p] : x1[2 * p - 4], xb = p < 2 ? x0[2 * p + 1] : x1[2 * p - 3];
    const unsigned hp = cvt_pk_bf16(xa, xb);
    const float ra = xa - bf16_lo(hp), rb = xb - bf16_hi(hp);  // exact
    const unsigned mp = cvt_pk_bf16(ra, rb);
    const float sa = ra - bf16_lo(mp), sb = rb - bf16_hi(mp);  // exact
    h[p] = hp;
    m[p] = mp;
    l[p] = cvt_pk_bf16(sa, sb);
  }
}

}  // namespace pcmi
