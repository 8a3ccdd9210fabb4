// BatchNorm (training statistics over all rows of a [n, c] activation), fused ReLU / residual
// epilogues, standalone ReLU / add, and the L2 row normalisation of the output features.
// All of these are HBM-bound streaming kernels: float4 per lane, every lane of a wave on one
// contiguous run of a row, per-channel reductions done in two deterministic levels
// (per-block partials in registers + LDS, then one thread per channel combining the blocks
// with Chan's parallel-variance update).
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <unordered_map>

#include "common.h"
#include "internal.h"

namespace pcmi {

// ---- per-stream arrival-counter pools (internal.h) ----------------------------------------------------------------
unsigned* stream_counters(hipStream_t st, size_t n) {
  struct Pool {
    unsigned* p = nullptr;
    size_t cap = 0;
  };
  static std::mutex mu;
  static std::unordered_map<hipStream_t, Pool> pools;
  std::lock_guard<std::mutex> lock(mu);
  Pool& pl = pools[st];
  if (n > pl.cap) {
    const size_t want = std::max<size_t>(align_up(n, 4096), 16384);
    unsigned* q = nullptr;
    // the old pool may still be in use by a launch in flight on `st`: drain the stream before replacing it
    // (zero-fill ON `st`: hipMemset on the null stream is not ordered against a non-blocking stream's launches)
    if (hipStreamSynchronize(st) != hipSuccess || hipMalloc((void**)&q, want * sizeof(unsigned)) != hipSuccess ||
        hipMemsetAsync(q, 0, want * sizeof(unsigned), st) != hipSuccess) {
      set_error("stream_counters: allocation of %zu counters failed", want);
      return nullptr;
    }
    if (pl.p) (void)hipFree(pl.p);
    pl.p = q;
    pl.cap = want;
  }
  return pl.p;
}

constexpr int kMaxRedBlocks = 1024;
// up to this many row blocks the statistics kernel finishes the reduction itself (last-arriving workgroup), beyond
// it a separate one-wave-per-channel kernel does (the serial merge of the partials would be a visible tail)
constexpr int kFuseFinalBlocks = 256;

struct RedGeom {
  int c4;              // float4 columns
  int rp;              // row lanes per block (256 / c4)
  int rows_per_block;  // multiple of rp
  int nblocks;
};

static RedGeom red_geom(int64_t n, int c) {
  RedGeom g;
  g.c4 = c / 4;
  g.rp = 256 / g.c4;
  int64_t rpb = std::max<int64_t>((int64_t)g.rp * 8, ceil_div(n, kMaxRedBlocks));
  rpb = ceil_div(rpb, g.rp) * g.rp;
  g.rows_per_block = (int)rpb;
  g.nblocks = (int)std::max<int64_t>(1, ceil_div(n, rpb));
  return g;
}

// ---- the ReLU pattern of a fused BatchNorm(+residual)+ReLU output as ONE BIT per element ("relu_bits") -----------------
// The backward pass needs of y only whether it is positive.  Reading the fp32 tensor for that is a third of the backward
// statistics' traffic and a quarter of the apply pass's (67 MB per level-1 BatchNorm of the bench batch, twice); the apply
// pass of the forward writes the pattern beside y -- bit (c mod 32) of word [row][c / 32], channels a multiple of 32 --
// and the backward kernels read 4 bytes where they read 128.  Same predicate (y > 0 of the value that was stored), so the
// gradients are the same bit for bit (tests/test_gpu_parity.py::test_relu_bits_*).  Round 6.
// nibble of the four channels of float4 column `col` -> a float4 that is positive where the bit is set
__device__ __forceinline__ float4 relu_bits_as_mask(uint32_t word, int col) {
  const uint32_t nib = word >> (4 * (col & 7));
  return make_float4((nib & 1u) ? 1.f : 0.f, (nib & 2u) ? 1.f : 0.f, (nib & 4u) ? 1.f : 0.f, (nib & 8u) ? 1.f : 0.f);
}
__device__ __forceinline__ uint32_t relu_nibble(const float4& o) {
  return (o.x > 0.f ? 1u : 0u) | (o.y > 0.f ? 2u : 0u) | (o.z > 0.f ? 4u : 0u) | (o.w > 0.f ? 8u : 0u);
}

// (n, mean, M2) of a set of rows merged with another set's (Chan et al.), four channels at a time
__device__ inline void chan_merge(float& n, float4& mean, float4& m2, float on, const float4& omean, const float4& om2) {
  const float tot = n + on;
  if (tot <= 0.f) return;
  const float wa = n / tot, wb = on / tot, cross = n * on / tot;
  const float4 d = make_float4(omean.x - mean.x, omean.y - mean.y, omean.z - mean.z, omean.w - mean.w);
  mean = make_float4(wa * mean.x + wb * omean.x, wa * mean.y + wb * omean.y, wa * mean.z + wb * omean.z,
                     wa * mean.w + wb * omean.w);
  m2 = make_float4(m2.x + om2.x + d.x * d.x * cross, m2.y + om2.y + d.y * d.y * cross, m2.z + om2.z + d.z * d.z * cross,
                   m2.w + om2.w + d.w * d.w * cross);
  n = tot;
}

// What the last-arriving workgroup of a fused statistics launch needs to finish the reduction (counter == nullptr:
// a separate kernel does it).
struct RedFinal {
  unsigned* counter;
  // MODE 0 (BatchNorm forward statistics)
  float eps, momentum;
  float* running_mean;
  float* running_var;
  float* save_mean;
  float* save_invstd;
  float* save_unbiased;
  // MODE 1 (BatchNorm backward sums)
  float* out_a;
  float* out_b;
  float* acc_a;
  float* acc_b;
  // two-segment launches (gridDim.y == 2, see colreduce_partial_kernel): floats between the segments' outputs
  int out_seg_stride;
};

// The merge of the per-block partials of one statistics launch (one segment), by ONE workgroup, in a fixed order ->
// deterministic.  Two levels, in the geometry of the main pass: thread (rl, col) folds the blocks rl, rl + rp, ... of its
// four channels (coalesced float4 loads, independent of each other), then the c4 column threads fold the rp lanes
// through LDS.  (A thread-per-channel loop over all blocks was one L2 latency per block: 20 us for 80 blocks.)
// All 256 threads call; threads t < c4 return the result: MODE 0 (a, b, cnt) = (mean, M2, rows), MODE 1 (sum a, sum b).
template <int MODE>
__device__ __forceinline__ void colreduce_merge(const float* __restrict__ part, int nblocks, int64_t n, int rows_per_block,
                                                int c4, int rp, int t, float4* s_a, float4* s_b, float* s_n, float4& a,
                                                float4& b, float& cnt) {
  const int col = t % c4, rl = t / c4;
  const int c = c4 * 4;
  cnt = 0.f;
  a = make_float4(0.f, 0.f, 0.f, 0.f);  // MODE 0: running mean, MODE 1: sum a
  b = a;                                 // MODE 0: running M2,   MODE 1: sum b
  if (rl < rp) {
#pragma unroll 4
    for (int q = rl; q < nblocks; q += rp) {
      const float4 pa = *reinterpret_cast<const float4*>(part + (int64_t)q * 2 * c + col * 4);
      const float4 pb = *reinterpret_cast<const float4*>(part + (int64_t)q * 2 * c + c + col * 4);
      if (MODE == 0) {
        const int64_t b0 = (int64_t)q * rows_per_block;
        chan_merge(cnt, a, b, (float)(min(b0 + (int64_t)rows_per_block, n) - b0), pa, pb);
      } else {
        a.x += pa.x; a.y += pa.y; a.z += pa.z; a.w += pa.w;
        b.x += pb.x; b.y += pb.y; b.z += pb.z; b.w += pb.w;
      }
    }
  }
  __syncthreads();  // (the caller's earlier use of the LDS arrays)
  s_a[t] = a;
  s_b[t] = b;
  s_n[t] = cnt;
  __syncthreads();
  if (t >= c4) return;
  a = s_a[t];
  b = s_b[t];
  cnt = s_n[t];
  for (int q = 1; q < rp; ++q) {
    const float4 va = s_a[q * c4 + t], vb = s_b[q * c4 + t];
    if (MODE == 0) {
      chan_merge(cnt, a, b, s_n[q * c4 + t], va, vb);
    } else {
      a.x += va.x; a.y += va.y; a.z += va.z; a.w += va.w;
      b.x += vb.x; b.y += vb.y; b.z += vb.z; b.w += vb.w;
    }
  }
}

// MODE 0: (mean, M2, rows) -> invstd / unbiased variance; both: what column thread t of the merging workgroup stores
__device__ __forceinline__ float4 bn_invstd(const float4& m2, float cnt, float eps) {
  const float4 var = make_float4(m2.x / cnt, m2.y / cnt, m2.z / cnt, m2.w / cnt);
  return make_float4(1.0f / sqrtf(var.x + eps), 1.0f / sqrtf(var.y + eps), 1.0f / sqrtf(var.z + eps), 1.0f / sqrtf(var.w + eps));
}
template <int MODE>
__device__ __forceinline__ void colreduce_write(const RedFinal& fin, int t, const float4& a, const float4& b, float cnt) {
  if (MODE == 0) {
    const float4 var = make_float4(b.x / cnt, b.y / cnt, b.z / cnt, b.w / cnt);
    reinterpret_cast<float4*>(fin.save_mean)[t] = a;
    reinterpret_cast<float4*>(fin.save_invstd)[t] = bn_invstd(b, cnt, fin.eps);
    const float ub = cnt > 1.f ? cnt / (cnt - 1.f) : 1.f;
    const float4 unb = make_float4(var.x * ub, var.y * ub, var.z * ub, var.w * ub);
    if (fin.save_unbiased) reinterpret_cast<float4*>(fin.save_unbiased)[t] = unb;
    if (fin.running_mean) {
      float4 rm = reinterpret_cast<float4*>(fin.running_mean)[t], rv = reinterpret_cast<float4*>(fin.running_var)[t];
      const float mo = fin.momentum, om = 1.f - fin.momentum;
      rm = make_float4(om * rm.x + mo * a.x, om * rm.y + mo * a.y, om * rm.z + mo * a.z, om * rm.w + mo * a.w);
      rv = make_float4(om * rv.x + mo * unb.x, om * rv.y + mo * unb.y, om * rv.z + mo * unb.z, om * rv.w + mo * unb.w);
      reinterpret_cast<float4*>(fin.running_mean)[t] = rm;
      reinterpret_cast<float4*>(fin.running_var)[t] = rv;
    }
  } else {
    reinterpret_cast<float4*>(fin.out_a)[t] = a;
    reinterpret_cast<float4*>(fin.out_b)[t] = b;
    if (fin.acc_a) {
      float4 ga = reinterpret_cast<float4*>(fin.acc_a)[t], gb = reinterpret_cast<float4*>(fin.acc_b)[t];
      reinterpret_cast<float4*>(fin.acc_a)[t] = make_float4(ga.x + a.x, ga.y + a.y, ga.z + a.z, ga.w + a.w);
      reinterpret_cast<float4*>(fin.acc_b)[t] = make_float4(gb.x + b.x, gb.y + b.y, gb.z + b.z, gb.w + b.w);
    }
  }
}

// MODE 0: (sum x, sum x^2) per block -> (mean_b, M2_b)
// MODE 1: (sum dy_eff, sum dy_eff * xhat)
// MASK (MODE 1): 0 no ReLU, 1 the pattern from y (fp32, `ymask`), 2 from `bits` (relu_bits).  A TEMPLATE parameter since
// round 6: as a run-time `if (ymask)` around the load inside the 4-row batch, hipcc put every row's mask load in a branch
// of its own with s_waitcnt vmcnt(0) in front of and behind it (hipcc -S) -- eight serialised memory round trips per batch
// where the batch exists to make it one.  For the same reason a row past the block is no longer `ok ? load : 0` but a load
// of the block's last row whose contribution is zeroed afterwards (v_cndmask, no branch).
template <int MODE, int MASK = 0>
__global__ __launch_bounds__(256) void colreduce_partial_kernel(
    const float* __restrict__ x, int64_t x_ld, const float* __restrict__ dy, int64_t dy_ld,
    const float* __restrict__ ymask, int64_t y_ld, const float* __restrict__ mean,
    const float* __restrict__ invstd, int64_t n, int c4, int rp, int rows_per_block,
    float* __restrict__ part /* [nblocks][2][c] */, RedFinal fin, int64_t seg_split = 0, int in_seg_stride = 0,
    int64_t part_seg_stride = 0, const uint32_t* __restrict__ bits = nullptr /* MODE 1: the ReLU pattern, see relu_bits */) {
  __shared__ float4 s_a[256];
  __shared__ float4 s_b[256];
  __shared__ unsigned s_last;
  const int t = threadIdx.x;
  const int col = t % c4, rl = t / c4;
  const int c = c4 * 4;
  // gridDim.y == 2: the rows [0, seg_split) and [seg_split, n) are two independent reductions (the two point clouds
  // of a pair in one sparse tensor: BatchNorm statistics per cloud) in ONE launch, each with its own partials,
  // arrival counter and outputs
  int64_t base = 0;
  if (gridDim.y > 1) {
    const int sg = blockIdx.y;
    base = sg ? seg_split : 0;
    n = sg ? n - seg_split : seg_split;
    part += sg * part_seg_stride;
    if (MODE == 1) {
      mean += sg * in_seg_stride;
      invstd += sg * in_seg_stride;
    }
    if (fin.counter) {
      fin.counter += sg;
      const int os = sg * fin.out_seg_stride;
      if (MODE == 0) {
        fin.save_mean += os;
        fin.save_invstd += os;
        if (fin.save_unbiased) fin.save_unbiased += os;
      } else {
        fin.out_a += os;
        fin.out_b += os;
      }
    }
    if ((int64_t)blockIdx.x * rows_per_block >= n) return;  // the shorter segment has fewer row blocks
  }
  x += base * x_ld;
  if (MODE == 1) {
    dy += base * dy_ld;
    if (ymask) ymask += base * y_ld;
    if (bits) bits += base * (c4 >> 3);
  }
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
  const int64_t r1 = min(r0 + (int64_t)rows_per_block, n);
  float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
  if (rl < rp) {
    float4 mu = a, is = a;
    if (MODE == 1) {
      mu = reinterpret_cast<const float4*>(mean)[col];
      is = reinterpret_cast<const float4*>(invstd)[col];
    }
    // rows in batches of kRowBatch: every load of a batch is issued before the first use (a row at a time was one
    // L2 / HBM latency per row and thread -- the small-activation BatchNorms were latency-, not bandwidth-bound)
    constexpr int kRowBatch = 4;
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int64_t rb = r0 + rl; rb < r1; rb += (int64_t)rp * kRowBatch) {
      float4 xv[kRowBatch], gv[kRowBatch], yv[kRowBatch];
      uint32_t wv[kRowBatch];
      bool ok[kRowBatch];
#pragma unroll
      for (int u = 0; u < kRowBatch; ++u) {  // all loads of the batch, none under a condition
        const int64_t r = rb + (int64_t)u * rp;
        ok[u] = r < r1;
        const int64_t rc = ok[u] ? r : r1 - 1;  // (r1 > rb >= r0: a valid row)
        xv[u] = *reinterpret_cast<const float4*>(x + rc * x_ld + col * 4);
        if constexpr (MODE == 1) {
          gv[u] = *reinterpret_cast<const float4*>(dy + rc * dy_ld + col * 4);
          if constexpr (MASK == 1) yv[u] = *reinterpret_cast<const float4*>(ymask + rc * y_ld + col * 4);
          if constexpr (MASK == 2) wv[u] = bits[rc * (c4 >> 3) + (col >> 3)];
        }
      }
#pragma unroll
      for (int u = 0; u < kRowBatch; ++u) {  // (a row past r1 adds nothing: its x / its gradient is zeroed here)
        const float4 xq = (MODE == 0 && !ok[u]) ? zero4 : xv[u];
        if (MODE == 0) {
          a.x += xq.x; a.y += xq.y; a.z += xq.z; a.w += xq.w;
          b.x = fmaf(xq.x, xq.x, b.x); b.y = fmaf(xq.y, xq.y, b.y);
          b.z = fmaf(xq.z, xq.z, b.z); b.w = fmaf(xq.w, xq.w, b.w);
        } else {
          float4 g = ok[u] ? gv[u] : zero4;
          if constexpr (MASK == 2) yv[u] = relu_bits_as_mask(wv[u], col);
          if constexpr (MASK != 0) {
            g.x = yv[u].x > 0.f ? g.x : 0.f; g.y = yv[u].y > 0.f ? g.y : 0.f;
            g.z = yv[u].z > 0.f ? g.z : 0.f; g.w = yv[u].w > 0.f ? g.w : 0.f;
          }
          a.x += g.x; a.y += g.y; a.z += g.z; a.w += g.w;
          b.x = fmaf(g.x, (xq.x - mu.x) * is.x, b.x); b.y = fmaf(g.y, (xq.y - mu.y) * is.y, b.y);
          b.z = fmaf(g.z, (xq.z - mu.z) * is.z, b.z); b.w = fmaf(g.w, (xq.w - mu.w) * is.w, b.w);
        }
      }
    }
  }
  s_a[t] = a;
  s_b[t] = b;
  __syncthreads();
  if (t < c4) {
    float4 sa = s_a[t], sb = s_b[t];
    for (int q = 1; q < rp; ++q) {
      const float4 va = s_a[q * c4 + t], vb = s_b[q * c4 + t];
      sa.x += va.x; sa.y += va.y; sa.z += va.z; sa.w += va.w;
      sb.x += vb.x; sb.y += vb.y; sb.z += vb.z; sb.w += vb.w;
    }
    float* p0 = part + (int64_t)blockIdx.x * 2 * c;
    if (MODE == 0) {
      const float nb = (float)(r1 - r0);
      float4 m, m2;
      m.x = sa.x / nb; m.y = sa.y / nb; m.z = sa.z / nb; m.w = sa.w / nb;
      m2.x = fmaxf(sb.x - sa.x * m.x, 0.f); m2.y = fmaxf(sb.y - sa.y * m.y, 0.f);
      m2.z = fmaxf(sb.z - sa.z * m.z, 0.f); m2.w = fmaxf(sb.w - sa.w * m.w, 0.f);
      reinterpret_cast<float4*>(p0)[t] = m;
      reinterpret_cast<float4*>(p0 + c)[t] = m2;
    } else {
      reinterpret_cast<float4*>(p0)[t] = sa;
      reinterpret_cast<float4*>(p0 + c)[t] = sb;
    }
  }
  // ---- fused final: the last workgroup to arrive merges the per-block partials (colreduce_merge) ----
  if (fin.counter == nullptr) return;
  const int nblocks = (int)((n + rows_per_block - 1) / rows_per_block);  // (= gridDim.x for a one-segment launch)
  if (!arrive_last(fin.counter, (unsigned)nblocks, &s_last)) return;
  __shared__ float s_n[256];
  float cnt;
  colreduce_merge<MODE>(part, nblocks, n, rows_per_block, c4, rp, t, s_a, s_b, s_n, a, b, cnt);
  if (t < c4) colreduce_write<MODE>(fin, t, a, b, cnt);
}

// The backward sums of colreduce_partial_kernel<1> within 48 registers and 4 KiB of LDS, for large activations.
// Why: one wgrad_x3p_kernel<3,3,4> workgroup per compute unit leaves 52 VGPRs per lane and 12 KiB of LDS
// (scripts/kernel_resources.py); the backward statistics of the level-1 BatchNorms run entirely beside such a launch and
// the 102-register kernel above was confined to the 32 compute units it leaves free -- 114-121 us for 200 MB where the
// same pass takes 69 us next to the smaller weight-gradient instantiation (profiles/r04zy_*).
// How: a thread owns TWO channels instead of four (half the accumulators / statistics in registers) and keeps four
// rows in flight through raw buffer loads with 32-bit byte offsets (one offset register per operand; a row past the
// block's end is an out-of-range offset that reads zeros and adds nothing).  Same partial layout [block][2][c] and the
// same blocks as the kernel above, so colreduce_final_kernel<1> merges either; the order in which a block's rows are
// added differs (256 / (c / 2) row lanes), fixed and deterministic all the same.
// MASKED: 0 no ReLU, 1 the pattern from y (fp32), 2 from relu_bits (`ymask` is then the bit tensor, y_ld its words per row)
template <int MASKED>
__global__ __launch_bounds__(256, 10) void bn_bwd_stats_lean_kernel(
    const float* __restrict__ x, int64_t x_ld, const float* __restrict__ dy, int64_t dy_ld,
    const float* __restrict__ ymask, int64_t y_ld, const float* __restrict__ mean, const float* __restrict__ invstd,
    int64_t n, int c, int rows_per_block, float* __restrict__ part, int64_t seg_split, int in_seg_stride,
    int64_t part_seg_stride) {
  __shared__ float2 s_a[256];
  __shared__ float2 s_b[256];
  typedef float lf2 __attribute__((ext_vector_type(2)));
  constexpr uint32_t kOut = 0x80000000u;
  constexpr int kFlags = 0x00020000;
  const int t = threadIdx.x;
  const int c2 = c >> 1, rp = 256 / c2;
  const int col = t % c2, rl = t / c2;
  int64_t base = 0;
  if (gridDim.y > 1) {  // two segments, as colreduce_partial_kernel
    const int sg = blockIdx.y;
    base = sg ? seg_split : 0;
    n = sg ? n - seg_split : seg_split;
    part += sg * part_seg_stride;
    mean += sg * in_seg_stride;
    invstd += sg * in_seg_stride;
    if ((int64_t)blockIdx.x * rows_per_block >= n) return;
  }
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
  const uint32_t first = (uint32_t)(base + r0) + (uint32_t)rl, last = (uint32_t)(base + min(r0 + (int64_t)rows_per_block, n));
  lf2 a = {0.f, 0.f}, b = {0.f, 0.f};
  if (rl < rp) {
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x), 0, 0x7FFFFFFF, kFlags);
    const __amdgpu_buffer_rsrc_t gr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(dy), 0, 0x7FFFFFFF, kFlags);
    const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(MASKED ? ymask : x), 0, 0x7FFFFFFF, kFlags);
    const uint32_t xs = (uint32_t)x_ld * 4u, gs = (uint32_t)dy_ld * 4u, ys = (uint32_t)y_ld * 4u, cb = (uint32_t)col * 8u;
    const lf2 mu = *reinterpret_cast<const lf2*>(mean + 2 * col), is = *reinterpret_cast<const lf2*>(invstd + 2 * col);
    constexpr int kRows = 4;
#pragma unroll 1
    for (uint32_t r = first; r < last; r += (uint32_t)(kRows * rp)) {
      lf2 xv[kRows], gv[kRows], yv[kRows];
      uint32_t wv[kRows];
#pragma unroll
      for (int u = 0; u < kRows; ++u) {
        const uint32_t ru = r + (uint32_t)(u * rp);
        const bool ok = ru < last;
        const uint32_t xo = ok ? ru * xs + cb : kOut;
        xv[u] = __builtin_bit_cast(lf2, __builtin_amdgcn_raw_buffer_load_b64(xr, xo, 0, 0));
        gv[u] = __builtin_bit_cast(lf2, __builtin_amdgcn_raw_buffer_load_b64(gr, ok ? ru * gs + cb : kOut, 0, 0));
        if constexpr (MASKED == 1) yv[u] = __builtin_bit_cast(lf2, __builtin_amdgcn_raw_buffer_load_b64(yr, ok ? ru * ys + cb : kOut, 0, 0));
        // MASKED == 2 (the caller guarantees x_ld == c): the row's words are c / 32 and the thread's two channels sit in word
        // col >> 4 at bits 2 (col & 15), + 1 -- i.e. the word's byte offset is the x element's byte offset / 32, rounded down to
        // 4: derived from the offset register x already has.  (An offset / shift pair of its own took the kernel from 48 to 49
        // registers = 56 allocated: it no longer fitted beside a weight-gradient workgroup, the reason this kernel exists.)
        if constexpr (MASKED == 2) wv[u] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_raw_buffer_load_b32(yr, ok ? ((xo >> 5) & ~3u) : kOut, 0, 0));
      }
#pragma unroll
      for (int u = 0; u < kRows; ++u) {
        lf2 g = gv[u];
        if constexpr (MASKED == 1) {
          g[0] = yv[u][0] > 0.f ? g[0] : 0.f;
          g[1] = yv[u][1] > 0.f ? g[1] : 0.f;
        }
        if constexpr (MASKED == 2) {  // (a row past the block: g is zero already)
          const uint32_t nib = wv[u] >> ((cb >> 2) & 30u);  // cb = 8 col
          g[0] = (nib & 1u) ? g[0] : 0.f;
          g[1] = (nib & 2u) ? g[1] : 0.f;
        }
        a[0] += g[0];
        a[1] += g[1];
        b[0] = fmaf(g[0], (xv[u][0] - mu[0]) * is[0], b[0]);
        b[1] = fmaf(g[1], (xv[u][1] - mu[1]) * is[1], b[1]);
      }
    }
  }
  s_a[t] = make_float2(a[0], a[1]);
  s_b[t] = make_float2(b[0], b[1]);
  __syncthreads();
  if (t < c2) {
    float2 sa = s_a[t], sb = s_b[t];
    for (int q = 1; q < rp; ++q) {
      const float2 va = s_a[q * c2 + t], vb = s_b[q * c2 + t];
      sa.x += va.x; sa.y += va.y;
      sb.x += vb.x; sb.y += vb.y;
    }
    float* p0 = part + (int64_t)blockIdx.x * 2 * c;
    reinterpret_cast<float2*>(p0)[t] = sa;
    reinterpret_cast<float2*>(p0 + c)[t] = sb;
  }
}

// The merge of a statistics launch's partials as a launch of its own.  Default since round 4: handing the partials to the
// last-arriving workgroup INSIDE the statistics launch (release, device-scope atomic, acquire, reloads that miss) cost
// the step 1.48 ms over its 129 BatchNorm statistics launches -- 11.5 us each, more than a kernel boundary and this
// kernel (profiles/r04p_bn_statistics_hand_over.txt).
// A workgroup owns kFinalCols float4 columns of one segment; its 256 / kFinalCols row lanes fold the blocks
// rl, rl + lanes, ... with every load of a batch in flight (the partials come from other XCDs: each batch is a round
// trip to memory -- the one-workgroup form of this kernel, 26 blocks per thread in batches of 4, was 7 us of the same
// latency over and over), then the column threads fold the lanes through LDS.  Fixed order -> deterministic.
constexpr int kFinalCols = 16, kFinalLanes = 256 / kFinalCols, kFinalBatch = 8;
template <int MODE>
__global__ __launch_bounds__(256) void colreduce_final_kernel(const float* __restrict__ part, int64_t n, int c4,
                                                              int rows_per_block, RedFinal fin, int64_t seg_split,
                                                              int64_t part_seg_stride) {
  __shared__ float4 s_a[256];
  __shared__ float4 s_b[256];
  __shared__ float s_n[256];
  const int t = threadIdx.x;
  if (gridDim.y > 1) {
    const int sg = blockIdx.y;
    n = sg ? n - seg_split : seg_split;
    part += sg * part_seg_stride;
    const int os = sg * fin.out_seg_stride;
    if (MODE == 0) {
      fin.save_mean += os;
      fin.save_invstd += os;
      if (fin.save_unbiased) fin.save_unbiased += os;
    } else {
      fin.out_a += os;
      fin.out_b += os;
    }
  }
  const int nblocks = (int)((n + rows_per_block - 1) / rows_per_block);
  const int c = c4 * 4;
  const int cl = t % kFinalCols, rl = t / kFinalCols;
  const int col = blockIdx.x * kFinalCols + cl;
  const bool live = col < c4;
  float cnt = 0.f;
  float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;  // MODE 0: running (mean, M2); MODE 1: (sum a, sum b)
  for (int q0 = rl; q0 < nblocks; q0 += kFinalLanes * kFinalBatch) {
    float4 pa[kFinalBatch], pb[kFinalBatch];
#pragma unroll
    for (int u = 0; u < kFinalBatch; ++u) {
      const int q = q0 + u * kFinalLanes;
      const bool ok = live && q < nblocks;
      pa[u] = ok ? *reinterpret_cast<const float4*>(part + (int64_t)q * 2 * c + col * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
      pb[u] = ok ? *reinterpret_cast<const float4*>(part + (int64_t)q * 2 * c + c + col * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int u = 0; u < kFinalBatch; ++u) {
      const int q = q0 + u * kFinalLanes;
      if (q >= nblocks) break;
      if (MODE == 0) {
        const int64_t b0 = (int64_t)q * rows_per_block;
        chan_merge(cnt, a, b, (float)(min(b0 + (int64_t)rows_per_block, n) - b0), pa[u], pb[u]);
      } else {
        a.x += pa[u].x; a.y += pa[u].y; a.z += pa[u].z; a.w += pa[u].w;
        b.x += pb[u].x; b.y += pb[u].y; b.z += pb[u].z; b.w += pb[u].w;
      }
    }
  }
  s_a[t] = a;
  s_b[t] = b;
  s_n[t] = cnt;
  __syncthreads();
  if (rl != 0 || !live) return;
  for (int q = 1; q < kFinalLanes; ++q) {
    const float4 va = s_a[q * kFinalCols + cl], vb = s_b[q * kFinalCols + cl];
    if (MODE == 0) {
      chan_merge(cnt, a, b, s_n[q * kFinalCols + cl], va, vb);
    } else {
      a.x += va.x; a.y += va.y; a.z += va.z; a.w += va.w;
      b.x += vb.x; b.y += vb.y; b.z += vb.z; b.w += vb.w;
    }
  }
  colreduce_write<MODE>(fin, col, a, b, cnt);
}

// One 64-lane wave per channel: lanes stride over the per-block partials, then the (n, mean, M2)
// triples are merged across lanes with Chan's update (associative, fixed order -> deterministic).
__global__ __launch_bounds__(256) void bn_stats_final_kernel(const float* __restrict__ part, int nblocks, int64_t n, int c,
                                                             int rows_per_block, float eps, float momentum,
                                                             float* __restrict__ running_mean,
                                                             float* __restrict__ running_var,
                                                             float* __restrict__ save_mean,
                                                             float* __restrict__ save_invstd,
                                                             float* __restrict__ save_unbiased) {
  const int lane = threadIdx.x & 63;
  const int ch = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (ch >= c) return;
  float cnt = 0.f, mean = 0.f, m2 = 0.f;
  for (int b = lane; b < nblocks; b += 64) {
    const int64_t r0 = (int64_t)b * rows_per_block;
    const float nb = (float)(min(r0 + (int64_t)rows_per_block, n) - r0);
    const float mb = part[(int64_t)b * 2 * c + ch], m2b = part[(int64_t)b * 2 * c + c + ch];
    const float tot = cnt + nb;
    const float delta = mb - mean;
    mean += delta * (nb / tot);
    m2 += m2b + delta * delta * (cnt * nb / tot);
    cnt = tot;
  }
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const float ocnt = __shfl_xor(cnt, d, 64), omean = __shfl_xor(mean, d, 64), om2 = __shfl_xor(m2, d, 64);
    const float tot = cnt + ocnt;
    if (tot > 0.f) {
      // symmetric form so that both partners compute bit-identical results
      const float delta = omean - mean;
      const float nmean = (cnt * mean + ocnt * omean) / tot;
      m2 = m2 + om2 + delta * delta * (cnt * ocnt / tot);
      mean = nmean;
      cnt = tot;
    }
  }
  if (lane != 0) return;
  const float var = m2 / cnt;
  save_mean[ch] = mean;
  save_invstd[ch] = 1.0f / sqrtf(var + eps);
  const float unbiased = cnt > 1.f ? m2 / (cnt - 1.f) : var;
  if (save_unbiased) save_unbiased[ch] = unbiased;
  if (running_mean) {
    running_mean[ch] = (1.f - momentum) * running_mean[ch] + momentum * mean;
    running_var[ch] = (1.f - momentum) * running_var[ch] + momentum * unbiased;
  }
}

__global__ __launch_bounds__(256) void colsum2_final_kernel(const float* __restrict__ part, int nblocks, int c,
                                                            float* __restrict__ out_a, float* __restrict__ out_b,
                                                            float* __restrict__ acc_a, float* __restrict__ acc_b) {
  const int lane = threadIdx.x & 63;
  const int ch = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (ch >= c) return;
  float a = 0.f, b = 0.f;
  for (int q = lane; q < nblocks; q += 64) {
    a += part[(int64_t)q * 2 * c + ch];
    b += part[(int64_t)q * 2 * c + c + ch];
  }
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    a += __shfl_xor(a, d, 64);
    b += __shfl_xor(b, d, 64);
  }
  if (lane == 0) {
    out_a[ch] = a;
    out_b[ch] = b;
    if (acc_a) {  // parameter gradients accumulated into a second (flat) buffer
      acc_a[ch] += a;
      acc_b[ch] += b;
    }
  }
}

// ---- small activations: the whole BatchNorm of a segment in ONE launch ---------------------------------------------
// The coarse levels of the U-Net (strides 8 and 16: 400-1400 rows per cloud and level on the bench batch, 128-256
// channels) run 30 of the network's 62 BatchNorms, each as three dependent launches per direction -- statistics over
// per-block partials, their merge, the apply pass: 19 us forward / 23 us backward of kernels that move 0.4-1.4 MB, plus
// two inter-kernel gaps on a chain that is latency-bound there.  With so few rows no cross-workgroup
// reduction is needed at all: a workgroup owns 16 channels (four float4 columns = 64 contiguous bytes of every row) of
// ALL rows of a segment, keeps its rows in registers (<= 12 per thread), reduces over its row lanes through shuffles
// and LDS in a fixed order, and applies -- one launch, no partials, no hand-over.  The statistics are two-pass (mean,
// then the centred second moment of the register-resident rows): at least as accurate as the merged partials.
// Geometry: thread t -> column cg = t & 3, row lane rl = t >> 2 (RL = THREADS / 4 lanes); rows rl, rl + RL, ...;
// grid = (c / 16, segments).  Eligible: c % 16 == 0 and <= kSmallMaxRows rows per segment (PCMI_BN_SMALL_ROWS).
constexpr int kSmallCG = 4;
constexpr int kSmallMaxRPT = 12;

// sum over all row lanes of the workgroup, per column group: every thread returns the total of ITS column group.
// Fixed order: xor-shuffles over the 16 row lanes of a wave, then the waves in index order.  s_w: [16][kSmallCG].
__device__ __forceinline__ float4 small_block_sum(float4 v, float4 (*s_w)[kSmallCG], int t) {
#pragma unroll
  for (int d = 4; d < 64; d <<= 1) {
    v.x += __shfl_xor(v.x, d, 64);
    v.y += __shfl_xor(v.y, d, 64);
    v.z += __shfl_xor(v.z, d, 64);
    v.w += __shfl_xor(v.w, d, 64);
  }
  const int lane = t & 63, wave = t >> 6, nw = (int)blockDim.x >> 6;
  __syncthreads();  // (the previous use of s_w)
  if (lane < kSmallCG) s_w[wave][lane] = v;
  __syncthreads();
  float4 s = s_w[0][t & 3];
  for (int w = 1; w < nw; ++w) {
    const float4 o = s_w[w][t & 3];
    s.x += o.x; s.y += o.y; s.z += o.z; s.w += o.w;
  }
  return s;
}

struct BnSmallFwd {
  const float* x; int64_t x_ld;
  const float* res; int64_t res_ld;
  float* y; int64_t y_ld;
  int64_t n, split;  // split == n: one segment
  const float* gamma; const float* beta;
  int relu;
  RedFinal fin;      // MODE 0 outputs (save_*, running_*, eps, momentum, out_seg_stride)
  uint16_t* bits;    // nullable: relu_bits as half words -- a workgroup's 16 channels are half of a 32-channel word; [row][c / 16]
};

// Rows go through raw buffer loads / stores: ONE offset register per operand (row lane x leading dimension + the
// thread's 16 bytes of the row), the row index j as a scalar offset -- so that the 12 rows a thread holds do not cost 12
// 64-bit addresses per operand -- and a row past the segment is an out-of-range offset: it loads zeros and its store
// is dropped (the scalar offset does not take part in the range check, so out of range stays out of range).
typedef float sf4 __attribute__((ext_vector_type(4)));
constexpr uint32_t kSmallOut = 0x80000000u;
__device__ __forceinline__ __amdgpu_buffer_rsrc_t small_rsrc(const float* p) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p), 0, 0x7FFFFFFF, 0x00020000);
}
__device__ __forceinline__ float4 small_load(__amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff) {
  const sf4 v = __builtin_bit_cast(sf4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
  return make_float4(v[0], v[1], v[2], v[3]);
}
// Stores take the WHOLE offset in the vector register (scalar offset 0).  With the row index as a scalar offset, as the
// loads have it, buffer_store_dwordx4 wrote a stale second dword in lanes 12-15 of every 16 (measured on MI355X, round 5:
// scripts/debug/bn_small_probe.py -- every row >= 64 with row % 4 == 3 had a wrong .y, plain pointer stores of the same
// registers were right): the store's data registers are written by VALU instructions right in front of it, and the
// compiler's hazard recogniser inserts the wait state for a > 64-bit MUBUF store only when soffset is NOT a register.
__device__ __forceinline__ void small_store(__amdgpu_buffer_rsrc_t r, uint32_t voff, const float4& v) {
  const sf4 t = {v.x, v.y, v.z, v.w};
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned, t), r, voff, 0, 0);
}

template <int RPT, int THREADS>
__global__ __launch_bounds__(THREADS) void bn_small_fwd_kernel(BnSmallFwd a) {
  __shared__ float4 s_w[16][kSmallCG];
  constexpr int RL = THREADS / kSmallCG;
  const int t = threadIdx.x, cg = t & 3, rl = t >> 2;
  const int sg = blockIdx.y;
  const int64_t base = sg ? a.split : 0;
  const int ns = (int)(gridDim.y > 1 ? (sg ? a.n - a.split : a.split) : a.n);
  const int col4 = blockIdx.x * kSmallCG + cg;
  RedFinal fin = a.fin;
  if (sg) {
    fin.save_mean += fin.out_seg_stride;
    fin.save_invstd += fin.out_seg_stride;
    if (fin.save_unbiased) fin.save_unbiased += fin.out_seg_stride;
  }
  const int cb = blockIdx.x * (4 * kSmallCG);  // first channel of the workgroup (uniform: part of the buffer base)
  const __amdgpu_buffer_rsrc_t xr = small_rsrc(a.x + base * a.x_ld + cb), yr = small_rsrc(a.y + base * a.y_ld + cb);
  const __amdgpu_buffer_rsrc_t rr = small_rsrc(a.res ? a.res + base * a.res_ld + cb : a.x);
  const uint32_t xs = (uint32_t)a.x_ld * 4u, ys = (uint32_t)a.y_ld * 4u, rs = (uint32_t)a.res_ld * 4u;
  const uint32_t xo = (uint32_t)rl * xs + (uint32_t)cg * 16u, yo = (uint32_t)rl * ys + (uint32_t)cg * 16u,
                 ro = (uint32_t)rl * rs + (uint32_t)cg * 16u;
  float4 xv[RPT];
#pragma unroll
  for (int j = 0; j < RPT; ++j) xv[j] = small_load(xr, rl + j * RL < ns ? xo : kSmallOut, (uint32_t)(j * RL) * xs);
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int j = 0; j < RPT; ++j) { s.x += xv[j].x; s.y += xv[j].y; s.z += xv[j].z; s.w += xv[j].w; }
  s = small_block_sum(s, s_w, t);
  const float cnt = (float)ns;
  const float4 mean = make_float4(s.x / cnt, s.y / cnt, s.z / cnt, s.w / cnt);
  float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int j = 0; j < RPT; ++j) {
    if (rl + j * RL < ns) {  // (rows past the segment were loaded as zeros: they must not count as -mean)
      const float dx = xv[j].x - mean.x, dy = xv[j].y - mean.y, dz = xv[j].z - mean.z, dw = xv[j].w - mean.w;
      q.x = fmaf(dx, dx, q.x); q.y = fmaf(dy, dy, q.y); q.z = fmaf(dz, dz, q.z); q.w = fmaf(dw, dw, q.w);
    }
  }
  q = small_block_sum(q, s_w, t);
  if (rl == 0) colreduce_write<0>(fin, col4, mean, q, cnt);  // mean, invstd, unbiased variance, running estimates
  const float4 is = bn_invstd(q, cnt, fin.eps);
  const float4 g = reinterpret_cast<const float4*>(a.gamma)[col4], b = reinterpret_cast<const float4*>(a.beta)[col4];
  const bool has_res = a.res != nullptr;
#pragma unroll
  for (int j = 0; j < RPT; ++j) {
    const bool ok = rl + j * RL < ns;
    float4 o;  // (the expression of bn_apply_kernel)
    o.x = (xv[j].x - mean.x) * is.x * g.x + b.x;
    o.y = (xv[j].y - mean.y) * is.y * g.y + b.y;
    o.z = (xv[j].z - mean.z) * is.z * g.z + b.z;
    o.w = (xv[j].w - mean.w) * is.w * g.w + b.w;
    if (has_res) {  // (uniform)
      const float4 rv = small_load(rr, ok ? ro : kSmallOut, (uint32_t)(j * RL) * rs);
      o.x += rv.x; o.y += rv.y; o.z += rv.z; o.w += rv.w;
    }
    if (a.relu) {
      o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f);
    }
    small_store(yr, ok ? yo + (uint32_t)(j * RL) * ys : kSmallOut, o);
    if (a.bits) {  // (uniform) the four lanes of a row lane hold the workgroup's 16 channels of row rl + j RL
      uint32_t h = relu_nibble(o) << (4 * cg);
      h |= __shfl_xor(h, 1, 64);
      h |= __shfl_xor(h, 2, 64);
      if (cg == 0 && ok) a.bits[(base + rl + j * RL) * (int64_t)gridDim.x + blockIdx.x] = (uint16_t)h;
    }
  }
}

struct BnSmallBwd {
  const float* dy; int64_t dy_ld;
  const float* x; int64_t x_ld;
  const float* ymask; int64_t y_ld;  // nullable
  int64_t n, split;                  // split == n: one segment
  const float* gamma;
  const float* mean; const float* invstd; int stat_stride;  // floats between the segments' statistics
  float* dx; int64_t dx_ld;
  float* dres; int64_t dres_ld; int dres_accumulate;        // nullable
  float* sum_g; float* sum_gx; int sum_stride;              // this call's sums per segment (dbeta, dgamma)
  float* acc_g; float* acc_gx;                               // nullable: parameter gradients, += segment 0, then += segment 1
  const uint16_t* bits;                                      // nullable: relu_bits (half words, as BnSmallFwd) instead of ymask
};

// The segments of a two-segment tensor are handled ONE AFTER THE OTHER by the same workgroup: the parameter gradients
// are (acc + sums of segment 0) + sums of segment 1, in that order -- what two consecutive one-segment calls accumulate
// (and what bn_bwd_apply_kernel does) -- without a hand-over between workgroups.
// gridDim.y == 2 (a two-segment tensor whose parameter gradients somebody else accumulates from `sum_g / sum_gx` --
// a.acc_g == nullptr, bn_param_acc_kernel): one workgroup per segment, side by side.
template <int RPT, int THREADS>
__global__ __launch_bounds__(THREADS) void bn_small_bwd_kernel(BnSmallBwd a) {
  __shared__ float4 s_w[16][kSmallCG];
  constexpr int RL = THREADS / kSmallCG;
  const int t = threadIdx.x, cg = t & 3, rl = t >> 2;
  const int col4 = blockIdx.x * kSmallCG + cg;
  const int cb = blockIdx.x * (4 * kSmallCG);
  const int n_seg = a.split < a.n ? 2 : 1;
  const int sg_first = gridDim.y > 1 ? (int)blockIdx.y : 0, sg_end = gridDim.y > 1 ? (int)blockIdx.y + 1 : n_seg;
  const float4 ga = reinterpret_cast<const float4*>(a.gamma)[col4];
  float4 pa = make_float4(0.f, 0.f, 0.f, 0.f), pb = pa;
  if (a.acc_g && rl == 0) {
    pa = reinterpret_cast<const float4*>(a.acc_g)[col4];
    pb = reinterpret_cast<const float4*>(a.acc_gx)[col4];
  }
  const uint32_t gs = (uint32_t)a.dy_ld * 4u, xs = (uint32_t)a.x_ld * 4u, ys = (uint32_t)a.y_ld * 4u, ds = (uint32_t)a.dx_ld * 4u,
                 rs = (uint32_t)a.dres_ld * 4u;
  const uint32_t lane_b = (uint32_t)cg * 16u;
  const bool masked = a.ymask != nullptr, has_res = a.dres != nullptr, bitmask = a.bits != nullptr;
#pragma unroll 1
  for (int sg = sg_first; sg < sg_end; ++sg) {
    const int64_t base = sg ? a.split : 0;
    const int ns = (int)(n_seg > 1 ? (sg ? a.n - a.split : a.split) : a.n);
    const float4 mu = reinterpret_cast<const float4*>(a.mean + sg * a.stat_stride)[col4];
    const float4 is = reinterpret_cast<const float4*>(a.invstd + sg * a.stat_stride)[col4];
    const __amdgpu_buffer_rsrc_t gr = small_rsrc(a.dy + base * a.dy_ld + cb), xr = small_rsrc(a.x + base * a.x_ld + cb);
    const __amdgpu_buffer_rsrc_t yr = small_rsrc(masked ? a.ymask + base * a.y_ld + cb : a.x);
    const __amdgpu_buffer_rsrc_t dr = small_rsrc(a.dx + base * a.dx_ld + cb);
    const __amdgpu_buffer_rsrc_t rr = small_rsrc(has_res ? a.dres + base * a.dres_ld + cb : a.dx);
    float4 gm[RPT], xh[RPT];  // masked gradient, normalised input
#pragma unroll
    for (int j = 0; j < RPT; ++j) {
      const bool ok = rl + j * RL < ns;
      gm[j] = small_load(gr, ok ? (uint32_t)rl * gs + lane_b : kSmallOut, (uint32_t)(j * RL) * gs);
      xh[j] = small_load(xr, ok ? (uint32_t)rl * xs + lane_b : kSmallOut, (uint32_t)(j * RL) * xs);
    }
    if (bitmask) {  // (uniform)
      uint32_t hw[RPT];
#pragma unroll
      for (int j = 0; j < RPT; ++j)
        hw[j] = rl + j * RL < ns ? (uint32_t)a.bits[(base + rl + j * RL) * (int64_t)gridDim.x + blockIdx.x] : 0xFFFFu;
#pragma unroll
      for (int j = 0; j < RPT; ++j) {
        const uint32_t nib = hw[j] >> (4 * cg);
        gm[j].x = (nib & 1u) ? gm[j].x : 0.f; gm[j].y = (nib & 2u) ? gm[j].y : 0.f;
        gm[j].z = (nib & 4u) ? gm[j].z : 0.f; gm[j].w = (nib & 8u) ? gm[j].w : 0.f;
      }
    } else if (masked) {  // (uniform)
#pragma unroll
      for (int j = 0; j < RPT; ++j) {  // (a row past the segment: gm is zero already, whatever its mask reads)
        const float4 yv = small_load(yr, rl + j * RL < ns ? (uint32_t)rl * ys + lane_b : kSmallOut, (uint32_t)(j * RL) * ys);
        gm[j].x = yv.x > 0.f ? gm[j].x : 0.f; gm[j].y = yv.y > 0.f ? gm[j].y : 0.f;
        gm[j].z = yv.z > 0.f ? gm[j].z : 0.f; gm[j].w = yv.w > 0.f ? gm[j].w : 0.f;
      }
    }
    float4 sa = make_float4(0.f, 0.f, 0.f, 0.f), sb = sa;
#pragma unroll
    for (int j = 0; j < RPT; ++j) {  // (a row past the segment: x read as zero, but gm = 0: it adds nothing to either sum)
      xh[j] = make_float4((xh[j].x - mu.x) * is.x, (xh[j].y - mu.y) * is.y, (xh[j].z - mu.z) * is.z, (xh[j].w - mu.w) * is.w);
      sa.x += gm[j].x; sa.y += gm[j].y; sa.z += gm[j].z; sa.w += gm[j].w;
      sb.x = fmaf(gm[j].x, xh[j].x, sb.x); sb.y = fmaf(gm[j].y, xh[j].y, sb.y);
      sb.z = fmaf(gm[j].z, xh[j].z, sb.z); sb.w = fmaf(gm[j].w, xh[j].w, sb.w);
    }
    sa = small_block_sum(sa, s_w, t);
    sb = small_block_sum(sb, s_w, t);
    if (rl == 0) {
      reinterpret_cast<float4*>(a.sum_g + sg * a.sum_stride)[col4] = sa;
      reinterpret_cast<float4*>(a.sum_gx + sg * a.sum_stride)[col4] = sb;
      pa = make_float4(pa.x + sa.x, pa.y + sa.y, pa.z + sa.z, pa.w + sa.w);
      pb = make_float4(pb.x + sb.x, pb.y + sb.y, pb.z + sb.z, pb.w + sb.w);
    }
    const float inv_n = 1.0f / (float)ns;
#pragma unroll
    for (int j = 0; j < RPT; ++j) {
      const bool ok = rl + j * RL < ns;
      float4 o;  // (the expression of bn_bwd_apply_kernel; xh = (x - mean) * invstd)
      o.x = ga.x * is.x * (gm[j].x - sa.x * inv_n - xh[j].x * sb.x * inv_n);
      o.y = ga.y * is.y * (gm[j].y - sa.y * inv_n - xh[j].y * sb.y * inv_n);
      o.z = ga.z * is.z * (gm[j].z - sa.z * inv_n - xh[j].z * sb.z * inv_n);
      o.w = ga.w * is.w * (gm[j].w - sa.w * inv_n - xh[j].w * sb.w * inv_n);
      small_store(dr, ok ? (uint32_t)(rl + j * RL) * ds + lane_b : kSmallOut, o);
      if (has_res) {  // (uniform)
        float4 g = gm[j];
        const uint32_t vo = ok ? (uint32_t)(rl + j * RL) * rs + lane_b : kSmallOut;
        if (a.dres_accumulate) {
          const float4 old = small_load(rr, vo, 0u);
          g.x += old.x; g.y += old.y; g.z += old.z; g.w += old.w;
        }
        small_store(rr, vo, g);
      }
    }
  }
  if (a.acc_g && rl == 0) {
    reinterpret_cast<float4*>(a.acc_g)[col4] = pa;
    reinterpret_cast<float4*>(a.acc_gx)[col4] = pb;
  }
}

// acc += sums of segment 0, then += sums of segment 1 (the order of two consecutive one-segment calls): the parameter
// gradients of a BatchNorm whose backward ran its two segments side by side.  One thread per four channels.
__global__ void bn_param_acc_kernel(float* __restrict__ acc_g, float* __restrict__ acc_gx, const float* __restrict__ sum_g,
                                    const float* __restrict__ sum_gx, int sum_stride, int c4, int n_seg) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= c4) return;
  float4 a = reinterpret_cast<float4*>(acc_g)[t], b = reinterpret_cast<float4*>(acc_gx)[t];
  for (int sg = 0; sg < n_seg; ++sg) {
    const float4 u = reinterpret_cast<const float4*>(sum_g + sg * sum_stride)[t];
    const float4 v = reinterpret_cast<const float4*>(sum_gx + sg * sum_stride)[t];
    a = make_float4(a.x + u.x, a.y + u.y, a.z + u.z, a.w + u.w);
    b = make_float4(b.x + v.x, b.y + v.y, b.z + v.z, b.w + v.w);
  }
  reinterpret_cast<float4*>(acc_g)[t] = a;
  reinterpret_cast<float4*>(acc_gx)[t] = b;
}

// y = relu?( (x - mean) * (invstd * gamma) + beta (+ residual) )
// RES / BITS: residual present / relu_bits written -- template parameters so that every load of an element is issued at the
// top of its iteration (round 6: the residual load sat in a uniform branch BEHIND the arithmetic with s_waitcnt vmcnt(0)
// after it, a third memory round trip per element; hipcc -S)
template <bool RES, bool BITS>
__global__ __launch_bounds__(256) void bn_apply_kernel(const float* __restrict__ x, int64_t x_ld,
                                                       int64_t n, int c4, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta,
                                                       const float* __restrict__ mean,
                                                       const float* __restrict__ invstd_or_var, float eps,
                                                       int use_var, const float* __restrict__ res,
                                                       int64_t res_ld, int relu, float* __restrict__ y,
                                                       int64_t y_ld, int64_t seg_split = INT64_MAX, int seg_stride4 = 0,
                                                       uint32_t* __restrict__ bits = nullptr /* relu_bits, c4 % 8 == 0 */) {
  const int64_t total = n * c4;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int64_t r = idx / c4;
    const int col = (int)(idx - r * c4);
    const int sc = col + (r >= seg_split ? seg_stride4 : 0);  // rows of the second segment: its own statistics
    const float4 xv = *reinterpret_cast<const float4*>(x + r * x_ld + col * 4);
    float4 rv = make_float4(0.f, 0.f, 0.f, 0.f);
    if constexpr (RES) rv = *reinterpret_cast<const float4*>(res + r * res_ld + col * 4);
    const float4 mu = reinterpret_cast<const float4*>(mean)[sc];
    float4 is = reinterpret_cast<const float4*>(invstd_or_var)[sc];
    const float4 g = reinterpret_cast<const float4*>(gamma)[col];
    const float4 b = reinterpret_cast<const float4*>(beta)[col];
    if (use_var) {
      is.x = 1.0f / sqrtf(is.x + eps); is.y = 1.0f / sqrtf(is.y + eps);
      is.z = 1.0f / sqrtf(is.z + eps); is.w = 1.0f / sqrtf(is.w + eps);
    }
    float4 o;
    o.x = (xv.x - mu.x) * is.x * g.x + b.x;
    o.y = (xv.y - mu.y) * is.y * g.y + b.y;
    o.z = (xv.z - mu.z) * is.z * g.z + b.z;
    o.w = (xv.w - mu.w) * is.w * g.w + b.w;
    if constexpr (RES) {
      o.x += rv.x; o.y += rv.y; o.z += rv.z; o.w += rv.w;
    }
    if (relu) {
      o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f);
    }
    *reinterpret_cast<float4*>(y + r * y_ld + col * 4) = o;
    if constexpr (BITS) {  // the 32-channel word of 8 consecutive float4 columns = 8 consecutive lanes: idx = col = lane mod 8,
      //            and total is a multiple of 8, so the 8 lanes of a word are active together
      uint32_t w = relu_nibble(o) << (4 * (col & 7));
      w |= __shfl_xor(w, 1, 64);
      w |= __shfl_xor(w, 2, 64);
      w |= __shfl_xor(w, 4, 64);
      if ((col & 7) == 0) bits[r * (c4 >> 3) + (col >> 3)] = w;
    }
  }
}

// dx = gamma * invstd * (g - sum_g/n - xhat * sum_gx/n),  dres = g   (g = relu-masked dy)
// MASK / DRES are template parameters and every load of an element is requested at the top of its iteration -- except the old
// residual gradient of DRES == 2, which a compiler barrier keeps below the arithmetic: four more live registers at the top cost
// the co-residency.  The register count is what decides this kernel in the step: a workgroup must fit beside a wgrad_x3p_kernel
// workgroup (52 registers per lane are left), or it runs on the 32 free compute units only and the level-1 launches get 14 %
// slower (profiles/r06g_*: the same form needed 50-58 registers then and lost).  Since the library is built with
// -fno-slp-vectorize the forms the step uses need 46-48 (the fp32-mask forms of PCMI_BN_RELU_BITS=0: 49-50): BatchNorm backward
// 2.54 -> 2.46 ms in the step, the step itself unchanged (profiles/r06s_*).
template <int MASK, int DRES>  // MASK 0: none, 1: fp32 y, 2: relu_bits; DRES 0: none, 1: store, 2: accumulate
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(
    const float* __restrict__ dy, int64_t dy_ld, const float* __restrict__ x, int64_t x_ld,
    const float* __restrict__ ymask, int64_t y_ld, int64_t n, int c4, const float* __restrict__ gamma,
    const float* __restrict__ mean, const float* __restrict__ invstd, const float* __restrict__ sum_g,
    const float* __restrict__ sum_gx, float* __restrict__ dx, int64_t dx_ld, float* __restrict__ dres,
    int64_t dres_ld, int dres_accumulate, int64_t seg_split = INT64_MAX, int stat_stride4 = 0, int sum_stride4 = 0,
    float* __restrict__ acc_g = nullptr, float* __restrict__ acc_gx = nullptr, const uint32_t* __restrict__ bits = nullptr) {
  const int64_t total = n * c4;
  const bool two = seg_split < n;
  const float inv_n0 = 1.0f / (float)(two ? seg_split : n), inv_n1 = two ? 1.0f / (float)(n - seg_split) : 0.f;
  if (acc_g && blockIdx.x == 0 && threadIdx.x < c4) {
    // two-segment launch: the parameter gradients (acc + sums of segment 0) + sums of segment 1, in that order --
    // what two consecutive one-segment calls accumulate
    float4 a = reinterpret_cast<float4*>(acc_g)[threadIdx.x], b = reinterpret_cast<float4*>(acc_gx)[threadIdx.x];
    for (int sg = 0; sg < (two ? 2 : 1); ++sg) {
      const float4 u = reinterpret_cast<const float4*>(sum_g)[threadIdx.x + sg * sum_stride4];
      const float4 v = reinterpret_cast<const float4*>(sum_gx)[threadIdx.x + sg * sum_stride4];
      a = make_float4(a.x + u.x, a.y + u.y, a.z + u.z, a.w + u.w);
      b = make_float4(b.x + v.x, b.y + v.y, b.z + v.z, b.w + v.w);
    }
    reinterpret_cast<float4*>(acc_g)[threadIdx.x] = a;
    reinterpret_cast<float4*>(acc_gx)[threadIdx.x] = b;
  }
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int64_t r = idx / c4;
    const int col = (int)(idx - r * c4);
    const bool second = r >= seg_split;
    const float inv_n = second ? inv_n1 : inv_n0;
    const int sc = col + (second ? stat_stride4 : 0), uc = col + (second ? sum_stride4 : 0);
    // every load of the element is requested here, before any of them is used
    float4 g = *reinterpret_cast<const float4*>(dy + r * dy_ld + col * 4);
    const float4 xv = *reinterpret_cast<const float4*>(x + r * x_ld + col * 4);
    uint32_t bw = 0;
    float4 yv = make_float4(1.f, 1.f, 1.f, 1.f);
    if constexpr (MASK == 2) bw = bits[r * (c4 >> 3) + (col >> 3)];
    if constexpr (MASK == 1) yv = *reinterpret_cast<const float4*>(ymask + r * y_ld + col * 4);
    if constexpr (MASK == 2) yv = relu_bits_as_mask(bw, col);
    if constexpr (MASK != 0) {
      g.x = yv.x > 0.f ? g.x : 0.f; g.y = yv.y > 0.f ? g.y : 0.f;
      g.z = yv.z > 0.f ? g.z : 0.f; g.w = yv.w > 0.f ? g.w : 0.f;
    }
    const float4 mu = reinterpret_cast<const float4*>(mean)[sc];
    const float4 is = reinterpret_cast<const float4*>(invstd)[sc];
    const float4 ga = reinterpret_cast<const float4*>(gamma)[col];
    const float4 sg = reinterpret_cast<const float4*>(sum_g)[uc];
    const float4 sx = reinterpret_cast<const float4*>(sum_gx)[uc];
    float4 o;
    o.x = ga.x * is.x * (g.x - sg.x * inv_n - (xv.x - mu.x) * is.x * sx.x * inv_n);
    o.y = ga.y * is.y * (g.y - sg.y * inv_n - (xv.y - mu.y) * is.y * sx.y * inv_n);
    o.z = ga.z * is.z * (g.z - sg.z * inv_n - (xv.z - mu.z) * is.z * sx.z * inv_n);
    o.w = ga.w * is.w * (g.w - sg.w * inv_n - (xv.w - mu.w) * is.w * sx.w * inv_n);
    *reinterpret_cast<float4*>(dx + r * dx_ld + col * 4) = o;
    if constexpr (DRES != 0) {
      if constexpr (DRES == 2) {  // (requested here, not at the top: four more live registers there cost the co-residency)
        asm volatile("" ::: "memory");  // (keeps the request below the arithmetic: the compiler would hoist it)
        const float4 dold = *reinterpret_cast<const float4*>(dres + r * dres_ld + col * 4);
        g.x += dold.x; g.y += dold.y; g.z += dold.z; g.w += dold.w;
      }
      *reinterpret_cast<float4*>(dres + r * dres_ld + col * 4) = g;
    }
  }
}

// OP 0: y = max(a, 0); OP 1: y = b > 0 ? a : 0 (a = dy, b = y); OP 2: y = a + b
template <int OP>
__global__ __launch_bounds__(256) void eltwise_kernel(const float* __restrict__ a, int64_t a_ld,
                                                      const float* __restrict__ b, int64_t b_ld, int64_t n,
                                                      int c4, float* __restrict__ y, int64_t y_ld) {
  const int64_t total = n * c4;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int64_t r = idx / c4;
    const int col = (int)(idx - r * c4);
    const float4 av = *reinterpret_cast<const float4*>(a + r * a_ld + col * 4);
    float4 o;
    if (OP == 0) {
      o = make_float4(fmaxf(av.x, 0.f), fmaxf(av.y, 0.f), fmaxf(av.z, 0.f), fmaxf(av.w, 0.f));
    } else {
      const float4 bv = *reinterpret_cast<const float4*>(b + r * b_ld + col * 4);
      if (OP == 1)
        o = make_float4(bv.x > 0.f ? av.x : 0.f, bv.y > 0.f ? av.y : 0.f, bv.z > 0.f ? av.z : 0.f,
                        bv.w > 0.f ? av.w : 0.f);
      else
        o = make_float4(av.x + bv.x, av.y + bv.y, av.z + bv.z, av.w + bv.w);
    }
    *reinterpret_cast<float4*>(y + r * y_ld + col * 4) = o;
  }
}

// L2 row normalisation: LPR lanes per row (power of two >= c/4), float4 per lane
template <bool BWD>
__global__ __launch_bounds__(256) void l2norm_kernel(const float* __restrict__ a, int64_t a_ld,
                                                     const float* __restrict__ yin, int64_t y_ld,
                                                     float* __restrict__ norm, int64_t n, int c4, int lpr,
                                                     float* __restrict__ out, int64_t out_ld) {
  const int rows_per_block = 256 / lpr;
  const int sub = threadIdx.x % lpr;
  const int64_t r = (int64_t)blockIdx.x * rows_per_block + threadIdx.x / lpr;
  const bool active = r < n && sub < c4;
  float4 av = make_float4(0.f, 0.f, 0.f, 0.f), yv = av;
  if (active) {
    av = *reinterpret_cast<const float4*>(a + r * a_ld + sub * 4);
    if (BWD) yv = *reinterpret_cast<const float4*>(yin + r * y_ld + sub * 4);
  }
  float s = BWD ? (av.x * yv.x + av.y * yv.y + av.z * yv.z + av.w * yv.w)
                : (av.x * av.x + av.y * av.y + av.z * av.z + av.w * av.w);
  for (int d = 1; d < lpr; d <<= 1) s += __shfl_xor(s, d, 64);
  if (!active) return;
  float4 o;
  if (!BWD) {
    const float nr = sqrtf(s);
    o = make_float4(av.x / nr, av.y / nr, av.z / nr, av.w / nr);
    if (sub == 0) norm[r] = nr;
  } else {
    const float nr = norm[r];
    o = make_float4((av.x - yv.x * s) / nr, (av.y - yv.y * s) / nr, (av.z - yv.z * s) / nr,
                    (av.w - yv.w * s) / nr);
  }
  *reinterpret_cast<float4*>(out + r * out_ld + sub * 4) = o;
}

// PCMI_BN_FUSED_FINAL=1: the statistics launch merges its own partials (last-arriving workgroup) instead of leaving
// them to colreduce_final_kernel -- the round-1..3 form, kept for the A/B (see colreduce_final_kernel)
// PCMI_BN_LEAN_ROWS: from this many rows the backward statistics take the 48-register form of colreduce_partial_kernel
// (0 = never).  Read per call (A/B in one process).  Same row blocks and the same partial layout, but NOT the same
// summation order inside a block: the lean kernel has 256 / (c / 2) row lanes where the wide one has 256 / (c / 4), so the
// last bits of dgamma / dbeta / dx change where n crosses this threshold (the one-launch kernels at PCMI_BN_SMALL_ROWS /
// _BWD_ROWS likewise).  Every form is deterministic for a given input; the result is not invariant under the row count
// (the golden traces were recorded with the default thresholds).
static int64_t bn_lean_rows() {
  const char* e = getenv("PCMI_BN_LEAN_ROWS");
  return e ? (int64_t)atoll(e) : (int64_t)65536;
}

static bool bn_apply_buf_ok(int64_t n, int64_t a, int64_t b, int64_t c_, int64_t d, int64_t e) {  // every tensor of n rows < 2 GiB
  const int64_t ld = std::max(std::max(std::max(a, b), std::max(c_, d)), e);
  return n * ld * 4 <= 0x7FFFFF00ll;
}
static bool bn_lean_eligible(int64_t n, int c, int64_t x_ld, int64_t dy_ld, int64_t y_ld) {
  const int64_t lean = bn_lean_rows();
  const int64_t ld = std::max(std::max(x_ld, dy_ld), y_ld);
  return lean > 0 && n >= lean && c % 2 == 0 && c / 2 <= 256 && n * ld * 4 <= 0x7FFFFF00ll;  // 32-bit byte offsets
}

// PCMI_BN_SMALL_ROWS / PCMI_BN_SMALL_BWD_ROWS: up to this many rows per segment a BatchNorm runs as ONE launch
// (bn_small_fwd_kernel / bn_small_bwd_kernel); 0 = never.  Read per call.  The bound of the kernels is 128 row lanes x 12
// rows.  Defaults 1536 forward, 768 backward -- measured on the bench batch (profiles/r05e_*): the forward kernel takes
// 6.1 us at 403 rows per segment and 9.3 us at 1350 (three launches: ~19 us + two gaps); the backward kernel, which
// walks the two segments of a pair one after the other (the parameter gradients are (acc + segment 0) + segment 1),
// 12.6 us at 403 rows but 32.6 us at 1350 -- slower than the three launches it replaces (~24 us).  Two more forms of
// the backward kernel at 1350 rows lost as well: the segments side by side with the parameter gradients added on the
// side stream (-1.2 % of the step, r05h_*), and 256 row lanes x 6 rows in 128 registers (-2.2 %, r05i_*).
static int64_t bn_small_rows(bool backward) {
  const char* e = getenv(backward ? "PCMI_BN_SMALL_BWD_ROWS" : "PCMI_BN_SMALL_ROWS");
  const int64_t v = e ? (int64_t)atoll(e) : (int64_t)(backward ? 768 : 1536);
  return std::min<int64_t>(v, 128 * kSmallMaxRPT);
}
static bool bn_small_eligible(int64_t longest_segment, int c, bool backward = false) {
  return longest_segment > 0 && longest_segment <= bn_small_rows(backward) && c % (4 * kSmallCG) == 0;
}

#define PCMI_BN_SMALL_DISPATCH(KERNEL, ARGS, GRID, LONGEST, ST)                                   \
  do {                                                                                            \
    const bool wide = (LONGEST) > 64 * kSmallMaxRPT;                                              \
    const int rl = wide ? 128 : 64;                                                               \
    const int need = (int)ceil_div((LONGEST), rl);                                                \
    if (!wide) {                                                                                  \
      if (need <= 4) KERNEL<4, 256><<<GRID, 256, 0, ST>>>(ARGS);                                   \
      else if (need <= 8) KERNEL<8, 256><<<GRID, 256, 0, ST>>>(ARGS);                              \
      else KERNEL<12, 256><<<GRID, 256, 0, ST>>>(ARGS);                                            \
    } else {                                                                                      \
      if (need <= 8) KERNEL<8, 512><<<GRID, 512, 0, ST>>>(ARGS);                                   \
      else KERNEL<12, 512><<<GRID, 512, 0, ST>>>(ARGS);                                            \
    }                                                                                             \
  } while (0)

static int bn_small_forward(const float* x, int64_t x_ld, int64_t n, int64_t split, int c, const float* gamma, const float* beta,
                            const float* residual, int64_t res_ld, int relu, float* y, int64_t y_ld, const RedFinal& fin,
                            hipStream_t st, uint32_t* relu_bits = nullptr) {
  BnSmallFwd a;
  a.x = x; a.x_ld = x_ld; a.res = residual; a.res_ld = res_ld; a.y = y; a.y_ld = y_ld;
  a.n = n; a.split = split; a.gamma = gamma; a.beta = beta; a.relu = relu; a.fin = fin;
  a.bits = reinterpret_cast<uint16_t*>(relu_bits);
  const bool two = split < n;
  const int64_t longest = two ? std::max(split, n - split) : n;
  const dim3 grid((unsigned)(c / (4 * kSmallCG)), two ? 2u : 1u);
  PCMI_BN_SMALL_DISPATCH(bn_small_fwd_kernel, a, grid, longest, st);
  PCMI_LAUNCH_CHECK();
  return PCMI_OK;
}

static int bn_small_backward(const BnSmallBwd& a, int c, hipStream_t st, bool parallel_segments = false) {
  const bool two = a.split < a.n;
  const int64_t longest = two ? std::max(a.split, a.n - a.split) : a.n;
  const dim3 grid((unsigned)(c / (4 * kSmallCG)), (two && parallel_segments) ? 2u : 1u);
  PCMI_BN_SMALL_DISPATCH(bn_small_bwd_kernel, a, grid, longest, st);
  PCMI_LAUNCH_CHECK();
  return PCMI_OK;
}

static bool fuse_final_enabled() {
  static const bool on = [] {
    const char* e = getenv("PCMI_BN_FUSED_FINAL");
    return e && e[0] == '1';
  }();
  return on;
}

static int check_rows(const char* who, const void* p, int64_t ld, int c) {
  PCMI_REQUIRE(p && c > 0 && c % 4 == 0 && c <= 1024 && ld % 4 == 0 && ld >= c && (uintptr_t)p % 16 == 0, PCMI_ERR_INVALID,
               "%s: needs 16-byte aligned rows, c %% 4 == 0, c <= 1024 (c=%d ld=%lld)", who, c, (long long)ld);
  return PCMI_OK;
}

// The streaming kernels above take what is uniform per launch (residual / mask form / accumulation) as template parameters:
// these pick the instantiation.
#define PCMI_BN_APPLY_LAUNCH(GRID, ST, RES_PTR, BITS_PTR, ...)                                          \
  do {                                                                                                  \
    if ((RES_PTR) && (BITS_PTR)) bn_apply_kernel<true, true><<<(GRID), 256, 0, (ST)>>>(__VA_ARGS__);    \
    else if (RES_PTR) bn_apply_kernel<true, false><<<(GRID), 256, 0, (ST)>>>(__VA_ARGS__);              \
    else if (BITS_PTR) bn_apply_kernel<false, true><<<(GRID), 256, 0, (ST)>>>(__VA_ARGS__);             \
    else bn_apply_kernel<false, false><<<(GRID), 256, 0, (ST)>>>(__VA_ARGS__);                          \
  } while (0)
#define PCMI_BN_BWD_APPLY_CASE(M, D, GRID, ST, ...) \
  case (M) * 3 + (D): bn_bwd_apply_kernel<M, D><<<(GRID), 256, 0, (ST)>>>(__VA_ARGS__); break
#define PCMI_BN_BWD_APPLY_LAUNCH(GRID, ST, MASK, DRESMODE, BUF, ...)                     \
  switch ((MASK) * 3 + (DRESMODE)) {                                                      \
    PCMI_BN_BWD_APPLY_CASE(0, 0, GRID, ST, __VA_ARGS__);                                  \
    PCMI_BN_BWD_APPLY_CASE(0, 1, GRID, ST, __VA_ARGS__);                                  \
    PCMI_BN_BWD_APPLY_CASE(0, 2, GRID, ST, __VA_ARGS__);                                  \
    PCMI_BN_BWD_APPLY_CASE(1, 0, GRID, ST, __VA_ARGS__);                                  \
    PCMI_BN_BWD_APPLY_CASE(1, 1, GRID, ST, __VA_ARGS__);                                  \
    PCMI_BN_BWD_APPLY_CASE(1, 2, GRID, ST, __VA_ARGS__);                                  \
    PCMI_BN_BWD_APPLY_CASE(2, 0, GRID, ST, __VA_ARGS__);                                  \
    PCMI_BN_BWD_APPLY_CASE(2, 1, GRID, ST, __VA_ARGS__);                                  \
    default: bn_bwd_apply_kernel<2, 2><<<(GRID), 256, 0, (ST)>>>(__VA_ARGS__); break;     \
  }
#define PCMI_BN_BWD_PARTIAL_LAUNCH(GRID, ST, MASK, ...)                                                 \
  do {                                                                                                  \
    const int m_ = (MASK);                                                                              \
    if (m_ == 2) colreduce_partial_kernel<1, 2><<<(GRID), 256, 0, (ST)>>>(__VA_ARGS__);                 \
    else if (m_ == 1) colreduce_partial_kernel<1, 1><<<(GRID), 256, 0, (ST)>>>(__VA_ARGS__);            \
    else colreduce_partial_kernel<1, 0><<<(GRID), 256, 0, (ST)>>>(__VA_ARGS__);                         \
  } while (0)

static unsigned stream_grid(int64_t total) {
  return (unsigned)std::max<int64_t>(1, std::min<int64_t>(ceil_div(total, 256), 256 * 8));
}

}  // namespace pcmi

using namespace pcmi;

extern "C" {

size_t pcmi_bn_workspace_bytes(int64_t n, int c) {
  (void)n;
  return (size_t)kMaxRedBlocks * 2 * c * sizeof(float) + 2 * (size_t)c * sizeof(float) + 512;
}

int pcmi_bn_fwd_train(const float* x, int64_t x_ld, int64_t n, int c, const float* gamma, const float* beta,
                      float* running_mean, float* running_var, float momentum, float eps, const float* residual,
                      int64_t res_ld, int relu, float* y, int64_t y_ld, float* save_mean, float* save_invstd,
                      void* ws, size_t ws_bytes, pcmi_stream_t stream) {
  return pcmi::bn_forward_train(x, x_ld, n, c, gamma, beta, running_mean, running_var, momentum, eps, residual, res_ld, relu, y,
                                y_ld, save_mean, save_invstd, nullptr, ws, ws_bytes, as_stream(stream));
}

}  // extern "C"

namespace pcmi {

// running_mean / running_var may be null (no update) and save_unbiased non-null: the executor then applies the
// running-estimate update later, in program order, with bn_running_update (two passes forwarded concurrently).
int bn_forward_train(const float* x, int64_t x_ld, int64_t n, int c, const float* gamma, const float* beta,
                     float* running_mean, float* running_var, float momentum, float eps, const float* residual,
                     int64_t res_ld, int relu, float* y, int64_t y_ld, float* save_mean, float* save_invstd,
                     float* save_unbiased, void* ws, size_t ws_bytes, hipStream_t st, uint32_t* relu_bits) {
  if (!relu || c % 32 != 0) relu_bits = nullptr;
  int rc = check_rows("bn_fwd_train(x)", x, x_ld, c);
  if (rc) return rc;
  rc = check_rows("bn_fwd_train(y)", y, y_ld, c);
  if (rc) return rc;
  if (residual && (rc = check_rows("bn_fwd_train(residual)", residual, res_ld, c))) return rc;
  PCMI_REQUIRE(gamma && beta && save_mean && save_invstd && n > 0 && c <= 1024, PCMI_ERR_INVALID, "bn_fwd_train: bad argument");
  PCMI_REQUIRE(ws && ws_bytes >= pcmi_bn_workspace_bytes(n, c), PCMI_ERR_WORKSPACE, "bn_fwd_train: workspace too small");
  const RedGeom g = red_geom(n, c);
  float* part = (float*)ws;
  RedFinal fin;
  memset(&fin, 0, sizeof(fin));
  const bool small = g.nblocks <= kFuseFinalBlocks, fuse = small && fuse_final_enabled();
  fin.eps = eps;
  fin.momentum = momentum;
  fin.running_mean = running_mean;
  fin.running_var = running_var;
  fin.save_mean = save_mean;
  fin.save_invstd = save_invstd;
  fin.save_unbiased = save_unbiased;
  if (bn_small_eligible(n, c)) return bn_small_forward(x, x_ld, n, n, c, gamma, beta, residual, res_ld, relu, y, y_ld, fin, st, relu_bits);
  if (fuse) {  // statistics + their final merge in ONE launch (last-arriving workgroup), then the apply pass
    fin.counter = stream_counters(st, 1);
    if (!fin.counter) return PCMI_ERR_HIP;
  }
  colreduce_partial_kernel<0><<<g.nblocks, 256, 0, st>>>(x, x_ld, nullptr, 0, nullptr, 0, nullptr, nullptr, n, g.c4, g.rp,
                                                        g.rows_per_block, part, fin);
  PCMI_LAUNCH_CHECK();
  if (small && !fuse) {
    colreduce_final_kernel<0><<<(unsigned)ceil_div(g.c4, kFinalCols), 256, 0, st>>>(part, n, g.c4, g.rows_per_block, fin, 0, 0);
    PCMI_LAUNCH_CHECK();
  }
  if (!small) {
    bn_stats_final_kernel<<<dim3((unsigned)ceil_div(c, 4)), 256, 0, st>>>(part, g.nblocks, n, c, g.rows_per_block, eps, momentum,
                                                                         running_mean, running_var, save_mean, save_invstd,
                                                                         save_unbiased);
    PCMI_LAUNCH_CHECK();
  }
  PCMI_BN_APPLY_LAUNCH(stream_grid(n * g.c4), st, residual, relu_bits, x, x_ld, n, g.c4, gamma, beta, save_mean, save_invstd, eps, 0,
                       residual, res_ld, relu, y, y_ld, INT64_MAX, 0, relu_bits);
  PCMI_LAUNCH_CHECK();
  return PCMI_OK;
}

// Two-segment BatchNorm forward (rows [0, split) and [split, n): own statistics each) in two launches: statistics of
// both segments (gridDim.y = 2, fused final), then one apply pass.  save_*: [2][3c] blocks (mean, invstd, unbiased)
// `stat_stride` floats apart; the running estimates are the caller's business (BnRunningUpdate with mean2).
int bn_forward_train2(const float* x, int64_t x_ld, int64_t n, int64_t split, int c, const float* gamma, const float* beta,
                      float eps, const float* residual, int64_t res_ld, int relu, float* y, int64_t y_ld, float* save_mean,
                      float* save_invstd, float* save_unbiased, int stat_stride, void* ws, size_t ws_bytes, hipStream_t st,
                      uint32_t* relu_bits) {
  if (!relu || c % 32 != 0) relu_bits = nullptr;
  int rc = check_rows("bn_fwd_train2(x)", x, x_ld, c);
  if (rc) return rc;
  rc = check_rows("bn_fwd_train2(y)", y, y_ld, c);
  if (rc) return rc;
  if (residual && (rc = check_rows("bn_fwd_train2(residual)", residual, res_ld, c))) return rc;
  PCMI_REQUIRE(gamma && beta && save_mean && save_invstd && split > 0 && split < n && stat_stride % 4 == 0, PCMI_ERR_INVALID,
               "bn_fwd_train2: bad argument");
  PCMI_REQUIRE(ws && ws_bytes >= pcmi_bn_workspace_bytes(n, c), PCMI_ERR_WORKSPACE, "bn_fwd_train2: workspace too small");
  const int64_t longest = std::max(split, n - split);
  RedGeom g = red_geom(longest, c);
  if (g.nblocks > kFuseFinalBlocks) {  // at most kFuseFinalBlocks row blocks per segment: the final merge stays fused
    g.rows_per_block = (int)(ceil_div(ceil_div(longest, kFuseFinalBlocks), g.rp) * g.rp);
    g.nblocks = (int)ceil_div(longest, g.rows_per_block);
  }
  float* part = (float*)ws;
  RedFinal fin;
  memset(&fin, 0, sizeof(fin));
  const bool fuse = fuse_final_enabled();
  if (fuse) {
    fin.counter = stream_counters(st, 2);
    if (!fin.counter) return PCMI_ERR_HIP;
  }
  fin.eps = eps;
  fin.save_mean = save_mean;
  fin.save_invstd = save_invstd;
  fin.save_unbiased = save_unbiased;
  fin.out_seg_stride = stat_stride;
  if (bn_small_eligible(longest, c)) {
    fin.counter = nullptr;
    return bn_small_forward(x, x_ld, n, split, c, gamma, beta, residual, res_ld, relu, y, y_ld, fin, st, relu_bits);
  }
  const int64_t part_seg = (int64_t)g.nblocks * 2 * c;
#if defined(PCMI_BN_DIAG_SKIP_SMALL_STATS)  // timing diagnostic (wrong results): as if the producer had left the partials behind
  if (longest >= PCMI_BN_DIAG_SKIP_SMALL_STATS)
#endif
  colreduce_partial_kernel<0><<<dim3((unsigned)g.nblocks, 2), 256, 0, st>>>(x, x_ld, nullptr, 0, nullptr, 0, nullptr, nullptr, n,
                                                                           g.c4, g.rp, g.rows_per_block, part, fin, split, 0,
                                                                           part_seg);
  PCMI_LAUNCH_CHECK();
  if (!fuse) {
    colreduce_final_kernel<0><<<dim3((unsigned)ceil_div(g.c4, kFinalCols), 2), 256, 0, st>>>(part, n, g.c4, g.rows_per_block, fin, split,
                                                                                              part_seg);
    PCMI_LAUNCH_CHECK();
  }
  PCMI_BN_APPLY_LAUNCH(stream_grid(n * g.c4), st, residual, relu_bits, x, x_ld, n, g.c4, gamma, beta, save_mean, save_invstd, eps, 0,
                       residual, res_ld, relu, y, y_ld, split, stat_stride / 4, relu_bits);
  PCMI_LAUNCH_CHECK();
  return PCMI_OK;
}

// Two-segment BatchNorm backward.  sums: [2][2c] scratch (dbeta, dgamma per segment); acc_*: parameter gradients
// (+= segment 0, then += segment 1).
int bn_backward2(const float* dy, int64_t dy_ld, const float* x, int64_t x_ld, const float* relu_mask_y, int64_t y_ld, int64_t n,
                 int64_t split, int c, const float* gamma, const float* save_mean, const float* save_invstd, int stat_stride,
                 float* dx, int64_t dx_ld, float* dres, int64_t dres_ld, int dres_accumulate, float* sums, float* acc_dgamma,
                 float* acc_dbeta, void* ws, size_t ws_bytes, hipStream_t st, int* deferred_acc, const uint32_t* relu_bits) {
  if (c % 32 != 0) relu_bits = nullptr;
  if (relu_bits) relu_mask_y = nullptr;  // the pattern comes from the bits: y is not read
  const int bits_ld = c / 32;
  int rc = check_rows("bn_bwd2(dy)", dy, dy_ld, c);
  if (rc) return rc;
  rc = check_rows("bn_bwd2(x)", x, x_ld, c);
  if (rc) return rc;
  rc = check_rows("bn_bwd2(dx)", dx, dx_ld, c);
  if (rc) return rc;
  if (relu_mask_y && (rc = check_rows("bn_bwd2(y)", relu_mask_y, y_ld, c))) return rc;
  if (dres && (rc = check_rows("bn_bwd2(dres)", dres, dres_ld, c))) return rc;
  PCMI_REQUIRE(gamma && save_mean && save_invstd && sums && split > 0 && split < n && stat_stride % 4 == 0, PCMI_ERR_INVALID,
               "bn_bwd2: bad argument");
  PCMI_REQUIRE(ws && ws_bytes >= pcmi_bn_workspace_bytes(n, c), PCMI_ERR_WORKSPACE, "bn_bwd2: workspace too small");
  const int64_t longest = std::max(split, n - split);
  RedGeom g = red_geom(longest, c);
  if (g.nblocks > kFuseFinalBlocks) {
    g.rows_per_block = (int)(ceil_div(ceil_div(longest, kFuseFinalBlocks), g.rp) * g.rp);
    g.nblocks = (int)ceil_div(longest, g.rows_per_block);
  }
  float* part = (float*)ws;
  RedFinal fin;
  memset(&fin, 0, sizeof(fin));
  // deferred_acc (nullable): the caller can add the two segments' sums to the parameter gradients itself
  // (bn_param_accumulate, on any stream ordered behind this call) -- then the segments run side by side, and the one-launch
  // form pays up to the forward threshold; else one workgroup walks them in order and accumulates.
  if (deferred_acc) *deferred_acc = 0;
  if (bn_small_eligible(longest, c, deferred_acc == nullptr)) {
    BnSmallBwd a;
    a.dy = dy; a.dy_ld = dy_ld; a.x = x; a.x_ld = x_ld; a.ymask = relu_mask_y; a.y_ld = y_ld; a.n = n; a.split = split;
    a.gamma = gamma; a.mean = save_mean; a.invstd = save_invstd; a.stat_stride = stat_stride;
    a.dx = dx; a.dx_ld = dx_ld; a.dres = dres; a.dres_ld = dres_ld; a.dres_accumulate = dres_accumulate;
    a.sum_g = sums; a.sum_gx = sums + c; a.sum_stride = 2 * c;
    a.acc_g = deferred_acc ? nullptr : acc_dbeta;
    a.acc_gx = deferred_acc ? nullptr : acc_dgamma;
    a.bits = reinterpret_cast<const uint16_t*>(relu_bits);
    if (deferred_acc) *deferred_acc = 1;
    return bn_small_backward(a, c, st, deferred_acc != nullptr);
  }
  // (the 48-register kernel finds a row's bit words through x's own offsets: with the bits it needs x_ld == c)
  const bool lean = bn_lean_eligible(n, c, x_ld, dy_ld, relu_mask_y ? y_ld : 0) && (!relu_bits || x_ld == c);
  const bool fuse = fuse_final_enabled() && !lean;  // (the lean statistics kernel never merges: it has no registers for it)
  if (fuse) {
    fin.counter = stream_counters(st, 2);
    if (!fin.counter) return PCMI_ERR_HIP;
  }
  fin.out_a = sums;      // dbeta of a segment
  fin.out_b = sums + c;  // dgamma
  fin.out_seg_stride = 2 * c;
  const int64_t part_seg = (int64_t)g.nblocks * 2 * c;
#if defined(PCMI_BN_DIAG_SKIP_SMALL_STATS_BWD)
  if (longest >= PCMI_BN_DIAG_SKIP_SMALL_STATS_BWD)
#endif
  if (lean && relu_bits)
    bn_bwd_stats_lean_kernel<2><<<dim3((unsigned)g.nblocks, 2), 256, 0, st>>>(x, x_ld, dy, dy_ld, reinterpret_cast<const float*>(relu_bits),
                                                                             bits_ld, save_mean, save_invstd, n, c, g.rows_per_block, part,
                                                                             split, stat_stride, part_seg);
  else if (lean && relu_mask_y)
    bn_bwd_stats_lean_kernel<1><<<dim3((unsigned)g.nblocks, 2), 256, 0, st>>>(x, x_ld, dy, dy_ld, relu_mask_y, y_ld, save_mean,
                                                                             save_invstd, n, c, g.rows_per_block, part, split,
                                                                             stat_stride, part_seg);
  else if (lean)
    bn_bwd_stats_lean_kernel<0><<<dim3((unsigned)g.nblocks, 2), 256, 0, st>>>(x, x_ld, dy, dy_ld, nullptr, 0, save_mean, save_invstd,
                                                                             n, c, g.rows_per_block, part, split, stat_stride,
                                                                             part_seg);
  else
    PCMI_BN_BWD_PARTIAL_LAUNCH(dim3((unsigned)g.nblocks, 2), st, relu_bits ? 2 : (relu_mask_y ? 1 : 0), x, x_ld, dy, dy_ld, relu_mask_y,
                               y_ld, save_mean, save_invstd, n, g.c4, g.rp, g.rows_per_block, part, fin, split, stat_stride, part_seg,
                               relu_bits);
  PCMI_LAUNCH_CHECK();
  if (!fuse) {
    colreduce_final_kernel<1><<<dim3((unsigned)ceil_div(g.c4, kFinalCols), 2), 256, 0, st>>>(part, n, g.c4, g.rows_per_block, fin, split,
                                                                                              part_seg);
    PCMI_LAUNCH_CHECK();
  }
  PCMI_BN_BWD_APPLY_LAUNCH(stream_grid(n * g.c4), st, relu_bits ? 2 : (relu_mask_y ? 1 : 0), dres ? (dres_accumulate ? 2 : 1) : 0,
                           bn_apply_buf_ok(n, dy_ld, x_ld, relu_mask_y ? y_ld : 0, dx_ld, dres ? dres_ld : 0), dy, dy_ld, x, x_ld,
                           relu_mask_y, y_ld, n, g.c4, gamma, save_mean, save_invstd, sums, sums + c, dx, dx_ld, dres, dres_ld,
                           dres_accumulate, split, stat_stride / 4, 2 * c / 4, acc_dbeta, acc_dgamma, relu_bits);
  PCMI_LAUNCH_CHECK();
  return PCMI_OK;
}

// (acc_dbeta, acc_dgamma) += the sums bn_backward2 left in `sums` ([2][2c]: dbeta, dgamma per segment) with *deferred_acc == 1
int bn_param_accumulate(const float* sums, int c, float* acc_dgamma, float* acc_dbeta, hipStream_t st) {
  PCMI_REQUIRE(sums && acc_dgamma && acc_dbeta && c % 4 == 0, PCMI_ERR_INVALID, "bn_param_accumulate: bad argument");
  bn_param_acc_kernel<<<(unsigned)ceil_div(c / 4, 64), 64, 0, st>>>(acc_dbeta, acc_dgamma, sums, sums + c, 2 * c, c / 4, 2);
  PCMI_LAUNCH_CHECK();
  return PCMI_OK;
}

__global__ __launch_bounds__(256) void bn_running_update_kernel(const BnRunningUpdate* __restrict__ tab) {
  const BnRunningUpdate u = tab[blockIdx.x];
  for (int ch = threadIdx.x; ch < u.c; ch += 256) {
    float rm = (1.f - u.momentum) * u.running_mean[ch] + u.momentum * u.mean[ch];
    float rv = (1.f - u.momentum) * u.running_var[ch] + u.momentum * u.unbiased[ch];
    if (u.mean2) {  // second segment of the same layer: the second of two forward calls
      rm = (1.f - u.momentum) * rm + u.momentum * u.mean2[ch];
      rv = (1.f - u.momentum) * rv + u.momentum * u.unbiased2[ch];
    }
    u.running_mean[ch] = rm;
    u.running_var[ch] = rv;
  }
}

int bn_running_update(const BnRunningUpdate* table_dev, int n_entries, hipStream_t st) {
  if (n_entries <= 0) return PCMI_OK;
  bn_running_update_kernel<<<n_entries, 256, 0, st>>>(table_dev);
  PCMI_LAUNCH_CHECK();
  return PCMI_OK;
}

}  // namespace pcmi

extern "C" {


int pcmi_bn_fwd_eval(const float* x, int64_t x_ld, int64_t n, int c, const float* gamma, const float* beta,
                     const float* running_mean, const float* running_var, float eps, const float* residual,
                     int64_t res_ld, int relu, float* y, int64_t y_ld, pcmi_stream_t stream) {
  int rc = check_rows("bn_fwd_eval(x)", x, x_ld, c);
  if (rc) return rc;
  rc = check_rows("bn_fwd_eval(y)", y, y_ld, c);
  if (rc) return rc;
  PCMI_REQUIRE(gamma && beta && running_mean && running_var, PCMI_ERR_INVALID, "bn_fwd_eval: bad argument");
  if (n == 0) return PCMI_OK;
  PCMI_BN_APPLY_LAUNCH(stream_grid(n * (c / 4)), as_stream(stream), residual, (uint32_t*)nullptr, x, x_ld, n, c / 4, gamma, beta,
                       running_mean, running_var, eps, 1, residual, res_ld, relu, y, y_ld, INT64_MAX, 0, (uint32_t*)nullptr);
  PCMI_LAUNCH_CHECK();
  return PCMI_OK;
}

}  // extern "C"

namespace pcmi {
// dgamma / dbeta: this call's sums (scratch, overwritten); acc_dgamma / acc_dbeta (nullable): += the same sums.
int bn_backward(const float* dy, int64_t dy_ld, const float* x, int64_t x_ld, const float* relu_mask_y, int64_t y_ld,
                int64_t n, int c, const float* gamma, const float* save_mean, const float* save_invstd, float* dx,
                int64_t dx_ld, float* dres, int64_t dres_ld, int dres_accumulate, float* dgamma, float* dbeta,
                float* acc_dgamma, float* acc_dbeta, void* ws, size_t ws_bytes, hipStream_t st, const uint32_t* relu_bits) {
  if (c % 32 != 0) relu_bits = nullptr;
  if (relu_bits) relu_mask_y = nullptr;
  const int bits_ld = c / 32;
  int rc = check_rows("bn_bwd(dy)", dy, dy_ld, c);
  if (rc) return rc;
  rc = check_rows("bn_bwd(x)", x, x_ld, c);
  if (rc) return rc;
  rc = check_rows("bn_bwd(dx)", dx, dx_ld, c);
  if (rc) return rc;
  if (relu_mask_y && (rc = check_rows("bn_bwd(y)", relu_mask_y, y_ld, c))) return rc;
  if (dres && (rc = check_rows("bn_bwd(dres)", dres, dres_ld, c))) return rc;
  PCMI_REQUIRE(gamma && save_mean && save_invstd && dgamma && dbeta && n > 0, PCMI_ERR_INVALID, "bn_bwd: bad argument");
  PCMI_REQUIRE(ws && ws_bytes >= pcmi_bn_workspace_bytes(n, c), PCMI_ERR_WORKSPACE, "bn_bwd: workspace too small");
  const RedGeom g = red_geom(n, c);
  float* part = (float*)ws;
  RedFinal fin;
  memset(&fin, 0, sizeof(fin));
  if (bn_small_eligible(n, c, true)) {
    BnSmallBwd a;
    a.dy = dy; a.dy_ld = dy_ld; a.x = x; a.x_ld = x_ld; a.ymask = relu_mask_y; a.y_ld = y_ld; a.n = n; a.split = n;
    a.gamma = gamma; a.mean = save_mean; a.invstd = save_invstd; a.stat_stride = 0;
    a.dx = dx; a.dx_ld = dx_ld; a.dres = dres; a.dres_ld = dres_ld; a.dres_accumulate = dres_accumulate;
    a.sum_g = dbeta; a.sum_gx = dgamma; a.sum_stride = 0; a.acc_g = acc_dbeta; a.acc_gx = acc_dgamma;
    a.bits = reinterpret_cast<const uint16_t*>(relu_bits);
    return bn_small_backward(a, c, st);
  }
  // (the 48-register kernel finds a row's bit words through x's own offsets: with the bits it needs x_ld == c)
  const bool lean = bn_lean_eligible(n, c, x_ld, dy_ld, relu_mask_y ? y_ld : 0) && (!relu_bits || x_ld == c);
  const bool small = g.nblocks <= kFuseFinalBlocks, fuse = small && fuse_final_enabled() && !lean;
  fin.out_a = dbeta;
  fin.out_b = dgamma;
  fin.acc_a = acc_dbeta;
  fin.acc_b = acc_dgamma;
  if (fuse) {
    fin.counter = stream_counters(st, 1);
    if (!fin.counter) return PCMI_ERR_HIP;
  }
  if (lean && relu_bits)
    bn_bwd_stats_lean_kernel<2><<<g.nblocks, 256, 0, st>>>(x, x_ld, dy, dy_ld, reinterpret_cast<const float*>(relu_bits), bits_ld,
                                                          save_mean, save_invstd, n, c, g.rows_per_block, part, 0, 0, 0);
  else if (lean && relu_mask_y)
    bn_bwd_stats_lean_kernel<1><<<g.nblocks, 256, 0, st>>>(x, x_ld, dy, dy_ld, relu_mask_y, y_ld, save_mean, save_invstd, n, c,
                                                          g.rows_per_block, part, 0, 0, 0);
  else if (lean)
    bn_bwd_stats_lean_kernel<0><<<g.nblocks, 256, 0, st>>>(x, x_ld, dy, dy_ld, nullptr, 0, save_mean, save_invstd, n, c,
                                                          g.rows_per_block, part, 0, 0, 0);
  else
    PCMI_BN_BWD_PARTIAL_LAUNCH(g.nblocks, st, relu_bits ? 2 : (relu_mask_y ? 1 : 0), x, x_ld, dy, dy_ld, relu_mask_y, y_ld, save_mean,
                               save_invstd, n, g.c4, g.rp, g.rows_per_block, part, fin, (int64_t)0, 0, (int64_t)0, relu_bits);
  PCMI_LAUNCH_CHECK();
  if (small && !fuse) {
    colreduce_final_kernel<1><<<(unsigned)ceil_div(g.c4, kFinalCols), 256, 0, st>>>(part, n, g.c4, g.rows_per_block, fin, 0, 0);
    PCMI_LAUNCH_CHECK();
  }
  if (!small) {
    colsum2_final_kernel<<<dim3((unsigned)ceil_div(c, 4)), 256, 0, st>>>(part, g.nblocks, c, dbeta, dgamma, acc_dbeta, acc_dgamma);
    PCMI_LAUNCH_CHECK();
  }
  PCMI_BN_BWD_APPLY_LAUNCH(stream_grid(n * g.c4), st, relu_bits ? 2 : (relu_mask_y ? 1 : 0), dres ? (dres_accumulate ? 2 : 1) : 0,
                           bn_apply_buf_ok(n, dy_ld, x_ld, relu_mask_y ? y_ld : 0, dx_ld, dres ? dres_ld : 0), dy, dy_ld, x, x_ld,
                           relu_mask_y, y_ld, n, g.c4, gamma, save_mean, save_invstd, dbeta, dgamma, dx, dx_ld, dres, dres_ld,
                           dres_accumulate, INT64_MAX, 0, 0, (float*)nullptr, (float*)nullptr, relu_bits);
  PCMI_LAUNCH_CHECK();
  return PCMI_OK;
}

}  // namespace pcmi

extern "C" {

int pcmi_bn_bwd(const float* dy, int64_t dy_ld, const float* x, int64_t x_ld, const float* relu_mask_y, int64_t y_ld,
                int64_t n, int c, const float* gamma, const float* save_mean, const float* save_invstd, float* dx,
                int64_t dx_ld, float* dres, int64_t dres_ld, float* dgamma, float* dbeta, void* ws, size_t ws_bytes,
                pcmi_stream_t stream) {
  return bn_backward(dy, dy_ld, x, x_ld, relu_mask_y, y_ld, n, c, gamma, save_mean, save_invstd, dx, dx_ld, dres, dres_ld, 0,
                     dgamma, dbeta, nullptr, nullptr, ws, ws_bytes, as_stream(stream));
}

int pcmi_relu_fwd(const float* x, int64_t x_ld, int64_t n, int c, float* y, int64_t y_ld, pcmi_stream_t stream) {
  int rc = check_rows("relu_fwd(x)", x, x_ld, c);
  if (rc) return rc;
  rc = check_rows("relu_fwd(y)", y, y_ld, c);
  if (rc) return rc;
  if (n == 0) return PCMI_OK;
  eltwise_kernel<0><<<stream_grid(n * (c / 4)), 256, 0, as_stream(stream)>>>(x, x_ld, nullptr, 0, n, c / 4, y, y_ld);
  PCMI_LAUNCH_CHECK();
  return PCMI_OK;
}

int pcmi_relu_bwd(const float* dy, int64_t dy_ld, const float* y, int64_t y_ld, int64_t n, int c, float* dx,
                  int64_t dx_ld, pcmi_stream_t stream) {
  int rc = check_rows("relu_bwd(dy)", dy, dy_ld, c);
  if (rc) return rc;
  rc = check_rows("relu_bwd(y)", y, y_ld, c);
  if (rc) return rc;
  rc = check_rows("relu_bwd(dx)", dx, dx_ld, c);
  if (rc) return rc;
  if (n == 0) return PCMI_OK;
  eltwise_kernel<1><<<stream_grid(n * (c / 4)), 256, 0, as_stream(stream)>>>(dy, dy_ld, y, y_ld, n, c / 4, dx, dx_ld);
  PCMI_LAUNCH_CHECK();
  return PCMI_OK;
}

int pcmi_add(const float* a, int64_t a_ld, const float* b, int64_t b_ld, int64_t n, int c, float* y, int64_t y_ld,
             pcmi_stream_t stream) {
  int rc = check_rows("add(a)", a, a_ld, c);
  if (rc) return rc;
  rc = check_rows("add(b)", b, b_ld, c);
  if (rc) return rc;
  rc = check_rows("add(y)", y, y_ld, c);
  if (rc) return rc;
  if (n == 0) return PCMI_OK;
  eltwise_kernel<2><<<stream_grid(n * (c / 4)), 256, 0, as_stream(stream)>>>(a, a_ld, b, b_ld, n, c / 4, y, y_ld);
  PCMI_LAUNCH_CHECK();
  return PCMI_OK;
}

static int l2_lanes(int c4) {
  int l = 1;
  while (l < c4) l <<= 1;
  return l;
}

int pcmi_l2norm_fwd(const float* x, int64_t x_ld, int64_t n, int c, float* y, int64_t y_ld, float* norm,
                    pcmi_stream_t stream) {
  int rc = check_rows("l2norm_fwd(x)", x, x_ld, c);
  if (rc) return rc;
  rc = check_rows("l2norm_fwd(y)", y, y_ld, c);
  if (rc) return rc;
  PCMI_REQUIRE(norm && c <= 256, PCMI_ERR_INVALID, "l2norm_fwd: bad argument (c=%d)", c);
  if (n == 0) return PCMI_OK;
  const int lpr = l2_lanes(c / 4);
  l2norm_kernel<false><<<dim3((unsigned)ceil_div(n, 256 / lpr)), 256, 0, as_stream(stream)>>>(x, x_ld, nullptr, 0, norm, n, c / 4,
                                                                                           lpr, y, y_ld);
  PCMI_LAUNCH_CHECK();
  return PCMI_OK;
}

int pcmi_l2norm_bwd(const float* dy, int64_t dy_ld, const float* y, int64_t y_ld, const float* norm, int64_t n, int c,
                    float* dx, int64_t dx_ld, pcmi_stream_t stream) {
  int rc = check_rows("l2norm_bwd(dy)", dy, dy_ld, c);
  if (rc) return rc;
  rc = check_rows("l2norm_bwd(y)", y, y_ld, c);
  if (rc) return rc;
  rc = check_rows("l2norm_bwd(dx)", dx, dx_ld, c);
  if (rc) return rc;
  PCMI_REQUIRE(norm && c <= 256, PCMI_ERR_INVALID, "l2norm_bwd: bad argument (c=%d)", c);
  if (n == 0) return PCMI_OK;
  const int lpr = l2_lanes(c / 4);
  l2norm_kernel<true><<<dim3((unsigned)ceil_div(n, 256 / lpr)), 256, 0, as_stream(stream)>>>(
      dy, dy_ld, y, y_ld, const_cast<float*>(norm), n, c / 4, lpr, dx, dx_ld);
  PCMI_LAUNCH_CHECK();
  return PCMI_OK;
}

}  // extern "C"
