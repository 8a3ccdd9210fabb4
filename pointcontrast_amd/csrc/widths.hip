// Sparse convolution with ANY channel counts.
//
// The matrix-core kernels (spconv.hip, spconv_wgrad.hip) contract and produce channels in units of 32.  Every width
// of the pre-training network is such a multiple (and the 3-channel stem has its own kernel), but the same backbone
// is reused downstream with out_channels = number of classes (downstream/semseg/models/res16unet.py:202-260: a 1x1
// head of width 20 / 13 / ...), and net.model_n_out may be 16.  For those the operands are staged zero-padded to the
// next multiple of 32 in the caller's workspace -- weights [K][Cp][Np], and the activation / gradient matrix whose
// width is odd -- the MFMA path runs on the padded problem (the zero channels contribute exact zeros), and the result
// is copied (or accumulated) back at its true width.  Cost: one extra pass over the padded operand; only layers with
// such widths pay it.
#include <algorithm>

#include "common.h"
#include "internal.h"

namespace pcmi {

static inline int up32(int c) { return (c + 31) / 32 * 32; }

// dst[s][r][c] (+)= (r < rows_src && c < c_src) ? src[s][r][c] : 0   for r < rows_dst, c < c_dst
__global__ __launch_bounds__(256) void pad2d_kernel(const float* __restrict__ src, int64_t src_ld, int64_t rows_src, int c_src,
                                                    int64_t src_slice, float* __restrict__ dst, int64_t dst_ld,
                                                    int64_t rows_dst, int c_dst, int64_t dst_slice, int64_t n_slices,
                                                    int accumulate) {
  const int64_t per = rows_dst * c_dst, total = per * n_slices;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int64_t sl = e / per, rem = e - sl * per;
    const int64_t r = rem / c_dst;
    const int c = (int)(rem - r * c_dst);
    const float v = (r < rows_src && c < c_src) ? src[sl * src_slice + r * src_ld + c] : 0.f;
    float* d = dst + sl * dst_slice + r * dst_ld + c;
    *d = accumulate ? *d + v : v;
  }
}

static int pad2d(const float* src, int64_t src_ld, int64_t rows_src, int c_src, int64_t src_slice, float* dst,
                 int64_t dst_ld, int64_t rows_dst, int c_dst, int64_t dst_slice, int64_t n_slices, int accumulate,
                 hipStream_t st) {
  const int64_t total = rows_dst * c_dst * n_slices;
  if (total == 0) return PCMI_OK;
  pad2d_kernel<<<(unsigned)std::min<int64_t>(ceil_div(total, 256), 4096), 256, 0, st>>>(
      src, src_ld, rows_src, c_src, src_slice, dst, dst_ld, rows_dst, c_dst, dst_slice, n_slices, accumulate);
  PCMI_LAUNCH_CHECK();
  return PCMI_OK;
}

struct Carve {  // bump allocator over the caller's workspace (256-byte aligned pieces)
  char* p;
  size_t left;
  float* take(size_t n_floats) {
    const size_t b = align_up(n_floats * sizeof(float), 256);
    if (b > left) return nullptr;
    float* r = (float*)p;
    p += b;
    left -= b;
    return r;
  }
};

static bool needs_padding(int cin, int cout) { return cin >= 8 && (cin % 32 != 0 || cout % 32 != 0); }

static size_t pad_bytes(int64_t n_in, int64_t n_out, int cin, int cout, int K) {
  const int Cp = up32(cin), Np = up32(cout);
  size_t f = align_up((size_t)K * Cp * Np * 4, 256) + align_up((size_t)Np * 4, 256);
  f += align_up((size_t)n_in * Cp * 4, 256) + align_up((size_t)n_out * Np * 4, 256);
  return f + 1024;
}

int spconv_forward(const float* in, int64_t in_ld, int64_t n_in, int cin, const float* weight, int cout,
                   const pcmi_kmap_t* map, int transpose, const float* bias, float* out, int64_t out_ld, int64_t n_out,
                   int accumulate, void* ws, size_t ws_bytes, hipStream_t st) {
  if (!needs_padding(cin, cout))
    return spconv_forward_m32(in, in_ld, n_in, cin, weight, cout, map, transpose, bias, out, out_ld, n_out, accumulate, ws,
                              ws_bytes, st);
  PCMI_REQUIRE(in && weight && out, PCMI_ERR_INVALID, "spconv_fwd: null argument");
  const int K = map ? map->K : 1, Cp = up32(cin), Np = up32(cout);
  Carve cv{(char*)ws, ws ? ws_bytes : 0};
  float* wp = cv.take((size_t)K * Cp * Np);
  float* bp = bias ? cv.take(Np) : nullptr;
  float* xp = Cp != cin ? cv.take((size_t)n_in * Cp) : nullptr;
  float* yp = Np != cout ? cv.take((size_t)n_out * Np) : nullptr;
  PCMI_REQUIRE(wp && (!bias || bp) && (Cp == cin || xp) && (Np == cout || yp), PCMI_ERR_WORKSPACE,
               "spconv_fwd: workspace too small for the padded widths (%d -> %d)", cin, cout);
  int rc = pad2d(weight, cout, cin, cout, (int64_t)cin * cout, wp, Np, Cp, Np, (int64_t)Cp * Np, K, 0, st);
  if (rc) return rc;
  if (bp && (rc = pad2d(bias, cout, 1, cout, 0, bp, Np, 1, Np, 0, 1, 0, st))) return rc;
  if (xp && (rc = pad2d(in, in_ld, n_in, cin, 0, xp, Cp, n_in, Cp, 0, 1, 0, st))) return rc;
  rc = spconv_forward_m32(xp ? xp : in, xp ? Cp : in_ld, n_in, Cp, wp, Np, map, transpose, bp, yp ? yp : out,
                          yp ? Np : out_ld, n_out, yp ? 0 : accumulate, cv.p, cv.left, st);
  if (rc) return rc;
  if (yp) rc = pad2d(yp, Np, n_out, Np, 0, out, out_ld, n_out, cout, 0, 1, accumulate, st);
  return rc;
}

int spconv_backward_data(const float* gout, int64_t gout_ld, int64_t n_out, int cout, const float* weight, int cin,
                         const pcmi_kmap_t* map, int transpose, float* gin, int64_t gin_ld, int64_t n_in, int accumulate,
                         void* ws, size_t ws_bytes, hipStream_t st) {
  if (!needs_padding(cin, cout))
    return spconv_backward_data_m32(gout, gout_ld, n_out, cout, weight, cin, map, transpose, gin, gin_ld, n_in, accumulate,
                                    ws, ws_bytes, st);
  PCMI_REQUIRE(gout && weight && gin, PCMI_ERR_INVALID, "spconv_bwd_data: null argument");
  const int K = map ? map->K : 1, Cp = up32(cin), Np = up32(cout);
  Carve cv{(char*)ws, ws ? ws_bytes : 0};
  float* wp = cv.take((size_t)K * Cp * Np);
  float* gp = Np != cout ? cv.take((size_t)n_out * Np) : nullptr;
  float* dxp = Cp != cin ? cv.take((size_t)n_in * Cp) : nullptr;
  PCMI_REQUIRE(wp && (Np == cout || gp) && (Cp == cin || dxp), PCMI_ERR_WORKSPACE,
               "spconv_bwd_data: workspace too small for the padded widths (%d -> %d)", cin, cout);
  int rc = pad2d(weight, cout, cin, cout, (int64_t)cin * cout, wp, Np, Cp, Np, (int64_t)Cp * Np, K, 0, st);
  if (rc) return rc;
  if (gp && (rc = pad2d(gout, gout_ld, n_out, cout, 0, gp, Np, n_out, Np, 0, 1, 0, st))) return rc;
  rc = spconv_backward_data_m32(gp ? gp : gout, gp ? Np : gout_ld, n_out, Np, wp, Cp, map, transpose, dxp ? dxp : gin,
                                dxp ? Cp : gin_ld, n_in, dxp ? 0 : accumulate, cv.p, cv.left, st);
  if (rc) return rc;
  if (dxp) rc = pad2d(dxp, Cp, n_in, Cp, 0, gin, gin_ld, n_in, cin, 0, 1, accumulate, st);
  return rc;
}

int spconv_backward_weight(const float* in, int64_t in_ld, int64_t n_in, int cin, const float* gout, int64_t gout_ld,
                           int64_t n_out, int cout, const pcmi_kmap_t* map, int transpose, float* gweight, float* gbias,
                           int accumulate, void* ws, size_t ws_bytes, hipStream_t st) {
  if (!needs_padding(cin, cout))
    return spconv_backward_weight_m32(in, in_ld, n_in, cin, gout, gout_ld, n_out, cout, map, transpose, gweight, gbias,
                                      accumulate, ws, ws_bytes, st);
  PCMI_REQUIRE(in && gout && gweight, PCMI_ERR_INVALID, "spconv_bwd_weight: null argument");
  const int K = map ? map->K : 1, Cp = up32(cin), Np = up32(cout);
  Carve cv{(char*)ws, ws ? ws_bytes : 0};
  float* gwp = cv.take((size_t)K * Cp * Np);
  float* gbp = gbias ? cv.take(Np) : nullptr;
  float* xp = Cp != cin ? cv.take((size_t)n_in * Cp) : nullptr;
  float* gp = Np != cout ? cv.take((size_t)n_out * Np) : nullptr;
  PCMI_REQUIRE(gwp && (!gbias || gbp) && (Cp == cin || xp) && (Np == cout || gp), PCMI_ERR_WORKSPACE,
               "spconv_bwd_weight: workspace too small for the padded widths (%d -> %d)", cin, cout);
  int rc = PCMI_OK;
  if (xp && (rc = pad2d(in, in_ld, n_in, cin, 0, xp, Cp, n_in, Cp, 0, 1, 0, st))) return rc;
  if (gp && (rc = pad2d(gout, gout_ld, n_out, cout, 0, gp, Np, n_out, Np, 0, 1, 0, st))) return rc;
  rc = spconv_backward_weight_m32(xp ? xp : in, xp ? Cp : in_ld, n_in, Cp, gp ? gp : gout, gp ? Np : gout_ld, n_out, Np, map,
                                  transpose, gwp, gbp, 0, cv.p, cv.left, st);
  if (rc) return rc;
  rc = pad2d(gwp, Np, Cp, Np, (int64_t)Cp * Np, gweight, cout, cin, cout, (int64_t)cin * cout, K, accumulate, st);
  if (rc) return rc;
  if (gbias) rc = pad2d(gbp, Np, 1, Np, 0, gbias, cout, 1, cout, 0, 1, accumulate, st);
  return rc;
}

}  // namespace pcmi

using namespace pcmi;

extern "C" {

size_t pcmi_spconv_workspace_bytes(int64_t n_in, int64_t n_out, int cin, int cout, int K, int64_t M) {
  if (!needs_padding(cin, cout)) return spconv_workspace_m32(n_in, n_out, cin, cout, K, M);
  return spconv_workspace_m32(n_in, n_out, up32(cin), up32(cout), K, M) + pad_bytes(n_in, n_out, cin, cout, K);
}

}  // extern "C"
