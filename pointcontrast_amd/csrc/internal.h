// Cross-file internal entry points (same semantics as the C ABI, plus accumulate flags).
#pragma once
#include "common.h"

namespace pcmi {

int spconv_forward(const float* in, int64_t in_ld, int64_t n_in, int cin, const float* weight, int cout,
                   const pcmi_kmap_t* map, int transpose, const float* bias, float* out, int64_t out_ld,
                   int64_t n_out, int accumulate, void* ws, size_t ws_bytes, hipStream_t st);
int spconv_backward_data(const float* gout, int64_t gout_ld, int64_t n_out, int cout, const float* weight, int cin,
                         const pcmi_kmap_t* map, int transpose, float* gin, int64_t gin_ld, int64_t n_in,
                         int accumulate, void* ws, size_t ws_bytes, hipStream_t st);
int spconv_backward_weight(const float* in, int64_t in_ld, int64_t n_in, int cin, const float* gout,
                           int64_t gout_ld, int64_t n_out, int cout, const pcmi_kmap_t* map, int transpose,
                           float* gweight, float* gbias, int accumulate, void* ws, size_t ws_bytes, hipStream_t st);
// the same three on the MFMA path proper (channel counts multiples of 32, or the 3-channel stem); the entry points
// above add zero-padded staging for any other width (widths.hip)
int spconv_forward_m32(const float* in, int64_t in_ld, int64_t n_in, int cin, const float* weight, int cout,
                       const pcmi_kmap_t* map, int transpose, const float* bias, float* out, int64_t out_ld,
                       int64_t n_out, int accumulate, void* ws, size_t ws_bytes, hipStream_t st);
int spconv_backward_data_m32(const float* gout, int64_t gout_ld, int64_t n_out, int cout, const float* weight, int cin,
                             const pcmi_kmap_t* map, int transpose, float* gin, int64_t gin_ld, int64_t n_in,
                             int accumulate, void* ws, size_t ws_bytes, hipStream_t st);
int spconv_backward_weight_m32(const float* in, int64_t in_ld, int64_t n_in, int cin, const float* gout,
                               int64_t gout_ld, int64_t n_out, int cout, const pcmi_kmap_t* map, int transpose,
                               float* gweight, float* gbias, int accumulate, void* ws, size_t ws_bytes, hipStream_t st);
size_t spconv_workspace_m32(int64_t n_in, int64_t n_out, int cin, int cout, int K, int64_t M);
// Weight gradients of several layers in ONE launch (spconv_wgrad.hip: wgrad_mfma_group_kernel).  add() == true: the layer's
// gradient is enqueued by the next flush() on the stream given there (its operands must exist on that stream by then and
// stay untouched until it); false: not a candidate, launch it with spconv_backward_weight.
struct WgradGroupBuilder;
WgradGroupBuilder* wgrad_group_create();
void wgrad_group_destroy(WgradGroupBuilder* b);
int wgrad_group_size(const WgradGroupBuilder* b);
void wgrad_group_drop(WgradGroupBuilder* b);  // forget what was collected (a backward pass that failed half-way)
bool wgrad_group_add(WgradGroupBuilder* b, const float* in, int64_t in_ld, int64_t n_in, int cin, const float* gout,
                     int64_t gout_ld, int64_t n_out, int cout, const pcmi_kmap_t* map, float* gweight, int accumulate,
                     hipStream_t st);
int wgrad_group_flush(WgradGroupBuilder* b, hipStream_t st);

int bn_backward(const float* dy, int64_t dy_ld, const float* x, int64_t x_ld, const float* relu_mask_y, int64_t y_ld,
                int64_t n, int c, const float* gamma, const float* save_mean, const float* save_invstd, float* dx,
                int64_t dx_ld, float* dres, int64_t dres_ld, int dres_accumulate, float* dgamma, float* dbeta,
                float* acc_dgamma, float* acc_dbeta, void* ws, size_t ws_bytes, hipStream_t st,
                const uint32_t* relu_bits = nullptr);
// relu_bits (nullable, norm.hip "relu_bits"): the ReLU pattern of a fused BatchNorm(+residual)+ReLU output as one bit per
// element, [rows][c / 32] words, c a multiple of 32 -- written by the forward functions when relu != 0, read by the backward
// ones INSTEAD of relu_mask_y (which may then be null)

int bn_forward_train(const float* x, int64_t x_ld, int64_t n, int c, const float* gamma, const float* beta,
                     float* running_mean, float* running_var, float momentum, float eps, const float* residual,
                     int64_t res_ld, int relu, float* y, int64_t y_ld, float* save_mean, float* save_invstd,
                     float* save_unbiased, void* ws, size_t ws_bytes, hipStream_t st, uint32_t* relu_bits = nullptr);
// deferred running-estimate update of one BatchNorm layer: running = (1 - momentum) * running + momentum * batch
struct BnRunningUpdate {
  float* running_mean;
  float* running_var;
  const float* mean;
  const float* unbiased;
  int c;
  float momentum;
  const float* mean2;      // nullable: statistics of the second segment of a two-segment pass, applied after the first
  const float* unbiased2;
};
int bn_forward_train2(const float* x, int64_t x_ld, int64_t n, int64_t split, int c, const float* gamma, const float* beta,
                      float eps, const float* residual, int64_t res_ld, int relu, float* y, int64_t y_ld, float* save_mean,
                      float* save_invstd, float* save_unbiased, int stat_stride, void* ws, size_t ws_bytes, hipStream_t st,
                      uint32_t* relu_bits = nullptr);
int bn_backward2(const float* dy, int64_t dy_ld, const float* x, int64_t x_ld, const float* relu_mask_y, int64_t y_ld, int64_t n,
                 int64_t split, int c, const float* gamma, const float* save_mean, const float* save_invstd, int stat_stride,
                 float* dx, int64_t dx_ld, float* dres, int64_t dres_ld, int dres_accumulate, float* sums, float* acc_dgamma,
                 float* acc_dbeta, void* ws, size_t ws_bytes, hipStream_t st, int* deferred_acc = nullptr,
                 const uint32_t* relu_bits = nullptr);
// deferred_acc != nullptr: the call MAY leave the parameter-gradient accumulation to the caller (*deferred_acc = 1):
// bn_param_accumulate(sums, ...) on a stream ordered behind the call, with `sums` untouched until then
int bn_param_accumulate(const float* sums, int c, float* acc_dgamma, float* acc_dbeta, hipStream_t st);
int bn_running_update(const BnRunningUpdate* table_dev, int n_entries, hipStream_t st);

// coords.hip: a cached map as it is -- M == -1 / offs_host unset when pcmi_coords_plan_unet built it and nobody has asked
// for the counts yet (the kernels size their launches by bounds and read the device-side offsets); and the ordering of a
// consumer stream behind the handle's last plan
int kmap_get_nosync(pcmi_coords_t* h, int in_key, int out_key, int kernel_size, int stride, int region, pcmi_kmap_t* out,
                    pcmi_stream_t stream);
int coords_wait_plan(pcmi_coords_t* h, hipStream_t st);
// pairs of a map as far as the host knows: exact once the counts have arrived, else the bound (every output row of a
// stride-1 map has at most K neighbours; every fine row of a stride-2 map exactly one parent)
static inline int64_t kmap_pairs_bound(const pcmi_kmap_t& m) {
  return m.M >= 0 ? m.M : (m.stride == 1 ? (int64_t)m.K * m.n_out : m.n_in);
}

// spconv_wgrad_x3.hip: weight gradients of the 3^3 / stride-1 convolutions, output-tile stationary on the bf16 matrix cores
bool wgrad_x3t_eligible(const pcmi_kmap_t* map, int64_t n_in, int64_t n_out, int cin, int cout, int64_t in_ld, int64_t gout_ld);
size_t wgrad_x3t_workspace(int64_t n_rows, int cin, int cout);
int wgrad_x3t_run(const float* in, int64_t in_ld, const float* gout, int64_t gout_ld, int64_t n_rows, int cin, int cout,
                  const pcmi_kmap_t* map, float* gweight, int accumulate, void* ws, size_t ws_bytes, hipStream_t st);

// nce_x3.hip: PointInfoNCE forward / backward on the bf16 matrix cores (three-term split); c in {16, 32}, q / k contiguous
// and 16-byte aligned, ws of pcmi_nce_workspace_bytes and 16-byte aligned.  PCMI_NCE_X3=0 keeps the fp32 VALU kernels of loss.hip.
bool nce_x3_on();
size_t nce_x3_workspace_bytes(int64_t n);
int nce_x3_fwd(const float* q, const float* k, int64_t n, int c, float inv_T, float* lse, float* loss, void* ws, hipStream_t st);
int nce_x3_bwd(const float* q, const float* k, const float* lse, int64_t n, int c, float inv_T, const float* gscale, float* dq,
               float* dk, void* ws, hipStream_t st);

size_t sort_rows_temp_bytes(int64_t n);
int sort_rows_by_mask(const int32_t* nbr, int K, int64_t n, int64_t chunk_rows, uint32_t* mask_in, uint32_t* mask_out,
                      int32_t* iota, void* temp, size_t temp_bytes, int32_t* perm, int32_t* nbr_perm, hipStream_t st);
int tile_units(const uint32_t* sorted_key, int K, int64_t n, uint32_t* tile_mask, int32_t* cnt, int32_t* tile_pref,
               hipStream_t st);
// rows of the contiguous tile range one XCD processes (tile swizzle of the conv kernel, 128-row tiles)
static inline int64_t xcd_chunk_rows(int64_t n) { return ceil_div(ceil_div(n, 128), 8) * 128; }

// n zero-initialised arrival counters (common.h: arrive_last) for launches on `st`: one pool per stream -- launches
// on a stream are serialised and every kernel leaves its counters at zero.  nullptr on allocation failure.
unsigned* stream_counters(hipStream_t st, size_t n);

// compute units of the current device (cached; 256 on MI355X)
static inline int num_cu() {
  static int n = 0;
  if (n == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n = prop.multiProcessorCount;
    if (n <= 0) n = 256;
  }
  return n;
}

}  // namespace pcmi
