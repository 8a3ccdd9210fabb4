// Launch arguments shared by the gathered-GEMM kernels of spconv.hip and spconv_x3.hip.
#pragma once
#include "common.h"

namespace pcmi {

struct ConvArgs {
  const float* x;        // gathered operand [*, x_ld]
  int64_t x_ld;
  int C;                 // contraction size (multiple of 32)
  const float* w;        // weights [K][cin][cout] in memory
  int64_t w_kstride;     // floats per weight slice (cin*cout)
  int64_t w_sc, w_sn;    // B_k[c][n] = w[wk*w_kstride + c*w_sc + n*w_sn]
  int N;                 // output channels (multiple of 32)
  const int32_t* nbr;    // [K][n_rows] or nullptr (identity); the permuted table when perm is set
  const int32_t* perm;   // nullable: tile position -> output row (mask-sorted processing order)
  const int32_t* pair_src;  // pair mode: gather row per pair
  const int32_t* pair_dst;  // pair mode: output row per pair
  const int64_t* offs;   // pair mode: [K+1] device group offsets
  int K;                 // number of offsets
  int32_t wsel[PCMI_MAX_KERNEL_VOLUME];  // weight slice used by offset k
  int64_t n_rows;        // output rows
  float* out;            // [n_rows, out_ld]  (or partial buffer when ksplit > 1)
  int64_t out_ld;
  int64_t split_stride;  // floats between partial buffers
  int ksplit;            // offsets are divided into ksplit contiguous ranges over blockIdx.z
  const float* bias;     // nullable, only when ksplit == 1
  int xcd_tiles;         // > 0: tiles per XCD of the XCD-contiguous tile order (grid.x = 8 * xcd_tiles)
  int accumulate;        // out += result instead of out = result (only when ksplit == 1)
  // unit-balanced mode (SK kernels): see spconv_mfma_kernel
  const uint32_t* sk_mask;  // [n_tiles] occupied offsets of a tile
  const int32_t* sk_pref;   // [n_tiles + 1] units before a tile
  int sk_tiles;
  float* sk_part;           // [gridDim.x][2][128][N] partial tiles
  // split-precision form (spconv_x3.hip): the weights as three bf16 terms in the kernel's LDS image order
  const void* wpack;
};

// ---- spconv_x3.hip: fp32 convolution on the bf16 matrix cores (three-term operand split) ---------------------------
// bytes of the packed weights of one launch (all K slices, every chunk / output slice), 256-byte aligned
size_t x3_pack_bytes(int K, int C, int N);
// packs B_k[c][n] = w[k * w_kstride + c * w_sc + n * w_sn] for k < K into `out` (x3_pack_bytes) for NT-wide slices
int x3_pack_weights(const ConvArgs& a, int NT, void* out, hipStream_t st);
// the 128-row-tile kernel itself (a.wpack set; grid as for spconv16p_kernel); NT in {2, 3, 4}
int x3_launch(int NT, bool sk, const ConvArgs& a, dim3 grid, hipStream_t st);
// output slice width (in units of 32 channels) of a table launch over `rows` output rows contracting C channels
// (0: that launch does not take the split kernel)
int x3_plan_nt(int64_t rows, int C, int N, int K);

// ---- spconv32r.hip: 32 -> 32 channel table convolutions with the weights resident in LDS ---------------------------
bool conv32r_eligible(const ConvArgs& a, int64_t x_bytes);  // a.nbr / a.perm / a.K / a.n_rows set
int conv32r_launch(bool w_transposed, const ConvArgs& a, hipStream_t st);

// ---- weights packed ahead of the launches (the network executor packs every eligible layer in ONE launch per forward
// pass instead of one pack launch in front of every convolution) ----------------------------------------------------
struct X3PackJob {   // one (layer, orientation): B_k[c][n] = w[k * w_kstride + c * w_sc + n * w_sn]
  const float* w;
  int64_t w_kstride, w_sc, w_sn;
  int K, C, N, NS;
  void* out;           // x3_pack_bytes(K, C, N)
  int64_t first_item;  // exclusive prefix of K * (C / 32) * N * 4 over the jobs
};
// packs the items [first_item, total_items) of the job list (a prefix / suffix of the jobs: the forward orientations first)
int x3_pack_many(const X3PackJob* jobs_dev, int n_jobs, int64_t total_items, hipStream_t st, int64_t first_item = 0);
struct X3Prepacked {
  const float* w;
  int transposed;  // 0: B = W[k] (forward), 1: B = W[k]^T (backward-data)
  int NT;
  const void* pack;
};
// The calling thread's table of packed weights: run_gathered uses an entry instead of packing when (weights pointer,
// orientation, slice width) match.  Set only while the owner guarantees the packs are current (the executor: from the
// pack launch at the top of a forward pass to the end of the matching backward); nullptr / 0 clears.
void x3_set_prepacked(const X3Prepacked* table, int n);
const void* x3_find_prepacked(const float* w, bool transposed, int NT);

}  // namespace pcmi
