/*
 * pcmi.h -- C ABI of libpcmi.so: the MI355X-native (gfx950 / CDNA4) sparse-voxel
 * contrastive pre-training hot path.
 *
 * This library replaces, for the path named by BASELINE.json:north_star, what the
 * reference reaches through MinkowskiEngine 0.4.3's pybind11 backend
 * ("MinkowskiEngineBackend", third-party, not vendored under /root/reference) and
 * a handful of torch ops.  Each entry point cites the reference call site it
 * replaces ("pc/" = /root/reference/pretrain/pointcontrast/).
 *
 * Conventions
 *   - every function returns PCMI_OK (0) or a negative PCMI_ERR_* code; the message
 *     of the last failure on the calling thread is pcmi_last_error();
 *   - every pointer is a DEVICE pointer owned by the caller (e.g. a torch tensor's
 *     data_ptr) unless its name ends in _host;
 *   - every call takes the HIP stream to enqueue on (pcmi_stream_t == hipStream_t);
 *     nothing synchronises except where the comment says "syncs";
 *   - float tensors are row-major fp32 [rows, channels] with an explicit leading
 *     dimension (*_ld, in floats) so producers can write into column slices of a
 *     wider buffer (zero-copy concat, pc/model/res16unet.py:235,242,249,256);
 *   - ops never allocate device memory: scratch comes from the caller through
 *     (ws, ws_bytes), sized by the matching *_workspace_bytes() query.  The one
 *     exception is the coordinate manager, which owns an arena that is reused
 *     across pcmi_coords_reset() calls (one hipMalloc burst in the first
 *     iterations, none in steady state);
 *   - a handle is not thread-safe; distinct handles are independent.
 */
#ifndef PCMI_H_
#define PCMI_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PCMI_OK 0
#define PCMI_ERR_INVALID (-1)     /* bad argument / shape */
#define PCMI_ERR_HIP (-2)         /* HIP runtime failure */
#define PCMI_ERR_DUPLICATE (-3)   /* duplicate coordinates in pcmi_coords_insert */
#define PCMI_ERR_NOKEY (-4)       /* unknown coords key */
#define PCMI_ERR_RANGE (-5)       /* coordinate outside the packable range */
#define PCMI_ERR_UNSUPPORTED (-6) /* configuration outside the hot path */
#define PCMI_ERR_WORKSPACE (-7)   /* workspace too small */

/* ME.RegionType values used by the path (pc/model/modules/common.py:47-60). */
#define PCMI_REGION_HYPERCUBE 0
#define PCMI_REGION_HYBRID 3

#define PCMI_MAX_KERNEL_VOLUME 27

typedef void* pcmi_stream_t; /* hipStream_t */
typedef struct pcmi_coords pcmi_coords_t;

int pcmi_version(void);
const char* pcmi_last_error(void);
/* Number of compute units / arch name of the current HIP device (gfx950 expected). */
int pcmi_device_info(int* n_cu, char* arch_host, int arch_len);

/* ------------------------------------------------------------------------------------------
 * Coordinate manager -- replaces ME CoordsManager / CoordsKey
 *   ME.SparseTensor(feats, coords=...)             pc/lib/ddp_trainer.py:290-297,392-398
 *   strided coordinates of every stride-2 conv     pc/model/res16unet.py:58-64,75-81,92-98,109-115
 * Coordinates are int32 rows (batch, x, y, z), batch index FIRST
 * (pc/lib/ddp_data_loaders.py:68-76); |x|,|y|,|z| < 2^17, 0 <= batch < 1023.
 * Key 0 is the inserted set (tensor stride 1); row i of a feature matrix belongs to
 * row i of its key's coordinates.
 * ------------------------------------------------------------------------------------------ */
int pcmi_coords_create(int dimension /* must be 3 */, pcmi_coords_t** out);
int pcmi_coords_destroy(pcmi_coords_t* h);
/* Forget all keys and maps but keep the device arena (one manager per SparseTensor per
 * iteration in the reference; here a pooled handle is reset instead). */
int pcmi_coords_reset(pcmi_coords_t* h);
/* Build the hash of n rows -> key 0.  Syncs (returns PCMI_ERR_DUPLICATE / PCMI_ERR_RANGE). */
int pcmi_coords_insert(pcmi_coords_t* h, const int32_t* bxyz, int64_t n, pcmi_stream_t stream);
/* The same without the synchronisation: the insert is only enqueued, and a duplicate / out-of-range row is reported by
 * the next call on this handle that synchronises `stream` anyway (pcmi_coords_plan_unet, pcmi_coords_stride, a
 * pcmi_kmap_get miss) or by pcmi_coords_check.  For callers that plan the whole network right behind the insert
 * (the training step: one host wait per batch instead of one per call). */
int pcmi_coords_insert_deferred(pcmi_coords_t* h, const int32_t* bxyz, int64_t n, pcmi_stream_t stream);
int pcmi_coords_check(pcmi_coords_t* h, pcmi_stream_t stream); /* syncs if a deferred insert is still unchecked */
/* Two-segment batches (the two point clouds of a contrastive pair processed as ONE sparse tensor: the reference runs
 * its network once per cloud, ddp_trainer.py:404-407, so BatchNorm statistics are per cloud).  The caller inserts the
 * rows of the first cloud before those of the second, with disjoint batch indices, and declares the boundary;
 * pcmi_coords_split returns the boundary of any level (strided levels keep first-occurrence order, so the segments
 * stay contiguous) or -1 when none was declared.  set_split: after insert, before the first stride. */
int pcmi_coords_set_split(pcmi_coords_t* h, int64_t n_first);
int pcmi_coords_split(pcmi_coords_t* h, int key, int64_t* n_first);
/* Strided coordinates: unique floor(c / (stride*ts)) * (stride*ts); rows are in
 * first-occurrence order of the input rows.  Cached per tensor stride.  Syncs on a miss. */
int pcmi_coords_stride(pcmi_coords_t* h, int in_key, int stride, int* out_key, int64_t* n_out,
                       pcmi_stream_t stream);
int pcmi_coords_key_at_stride(pcmi_coords_t* h, int tensor_stride, int* key);
int pcmi_coords_size(pcmi_coords_t* h, int key, int64_t* n, int* tensor_stride);
/* Copy the key's coordinates into out_bxyz [n,4] (device). */
int pcmi_coords_get(pcmi_coords_t* h, int key, int32_t* out_bxyz, pcmi_stream_t stream);
/* Build every level (strides 2,4,..,2^n_down) and the kernel maps a Res16UNet forward
 * uses up front (typically on a side stream while the compute stream is still busy), so
 * the per-layer calls below all hit the cache and never sync.  Pure performance hint.  first_region: region of the 3^3 stem conv (HYPERCUBE),
 * block_region: region of the 3^3 block convs (HYBRID).
 * ONE host synchronisation per call (round 2: 16): the levels of a fresh handle are built as a chain whose kernels take
 * their row counts from the device and whose counts come back together; the maps are then enqueued without waiting for
 * their per-offset pair counts (pcmi_kmap_get hands those out later; no kernel of this library needs them on the
 * host).  The tables are complete when `stream` reaches the end of the call: a consumer on another stream must be
 * ordered behind it (pcmi_net_forward does that itself). */
int pcmi_coords_plan_unet(pcmi_coords_t* h, int n_down, int first_region, int block_region,
                          pcmi_stream_t stream);
/* Bytes currently reserved by the handle's arena (for memory accounting). */
int pcmi_coords_arena_bytes(pcmi_coords_t* h, size_t* bytes);

/* ------------------------------------------------------------------------------------------
 * Kernel maps -- replaces CoordsManager::getInOutMaps (reached from every
 * MinkowskiConvolution / MinkowskiConvolutionTranspose forward,
 * pc/model/modules/common.py:130-139,159-168).
 * Pair (i, j, k): in-row i feeds out-row j through weight slice k iff
 *   c_out[j] + offset_k * tensor_stride(in) == c_in[i].
 * Offsets in weight-slice order come from pcmi_kernel_offsets().
 * All pointers live in the handle's arena and stay valid until reset/destroy.
 * ------------------------------------------------------------------------------------------ */
typedef struct pcmi_kmap {
  int32_t K;               /* kernel volume (27, 8) */
  int32_t kernel_size;     /* 3 or 2 */
  int32_t stride;          /* 1 or 2 */
  int32_t region;
  int64_t n_in, n_out;
  int64_t M;               /* total number of pairs; -1 in a map taken while pcmi_coords_plan_unet's counts were still
                            * on their way (internal use: pcmi_kmap_get always returns it filled in) */
  const int32_t* nbr;      /* [K, n_out]: in-row for (k, out-row) or -1 */
  const int32_t* pair_in;  /* [M] grouped by k, ascending out-row inside a group */
  const int32_t* pair_out; /* [M] */
  const int64_t* offs;     /* [K+1] device prefix of the group sizes */
  int64_t offs_host[PCMI_MAX_KERNEL_VOLUME + 1];
  int32_t mirror[PCMI_MAX_KERNEL_VOLUME]; /* offset_{mirror[k]} == -offset_k (stride 1) */
  /* Processing order for the conv kernels (stride-1 maps of large levels, else NULL): rows sorted by their
   * 27-bit neighbour-occupancy mask so that a 32-row wave group shares its set of occupied offsets.
   * perm[i] = row handled at position i; nbr_perm[k][i] = nbr[k][perm[i]].  Results are unaffected. */
  const int32_t* perm;
  const int32_t* nbr_perm;
  /* Work units of the 128-row tiles of that order (NULL when perm is NULL): tile_mask[t] = OR of the occupancy
   * masks of the tile's rows, tile_pref[t] = number of (tile, occupied offset) units before tile t
   * (tile_pref[n_tiles] = total).  Lets the conv kernel give every workgroup the same number of units. */
  const uint32_t* tile_mask;
  const int32_t* tile_pref;
  int64_t n_tiles;
} pcmi_kmap_t;

/* Host-side enumeration of the kernel offsets in weight-slice order
 * (ME.KernelGenerator, pc/model/modules/common.py:127-128,151-157).  out_host: [K,3]. */
int pcmi_kernel_offsets(int kernel_size, int region, int32_t* out_host, int* K);
/* kernel_size 3 / stride 1 (in_key == out_key) or kernel_size 2 / stride 2
 * (out_key = strided key of in_key).  Cached; one stream sync on a miss (reads back the
 * K+1 group offsets). */
int pcmi_kmap_get(pcmi_coords_t* h, int in_key, int out_key, int kernel_size, int stride,
                  int region, pcmi_kmap_t* out, pcmi_stream_t stream);
/* Copy a map's tables into caller buffers (any may be NULL): nbr [K*n_out], pair_in [M],
 * pair_out [M].  For parity checks and ME-style get_kernel_map(); not on the hot path. */
int pcmi_kmap_export(const pcmi_kmap_t* map, int32_t* nbr, int32_t* pair_in, int32_t* pair_out,
                     pcmi_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Sparse convolution -- replaces MEB.Convolution{Forward,Backward}GPU and
 * ConvolutionTranspose{Forward,Backward}GPU (pc/model/modules/common.py:117-168; 63 modules
 * in Res16UNet34C).  weight is [K, cin, cout] fp32 (K == 1: [cin, cout], map == NULL: the
 * dense 1x1 "use_mm" path).  transpose == 0: conv along the map (in = map.n_in rows,
 * out = map.n_out rows).  transpose == 1: transposed conv (in = map.n_out rows,
 * out = map.n_in rows, same weight-slice index).
 *   fwd        out[j]  = sum_k in[i] @ W[k]      (+ bias)
 *   bwd_data   gin[i]  = sum_k gout[j] @ W[k]^T
 *   bwd_weight gW[k]   = sum_pairs in[i]^T gout[j]
 * All three overwrite their outputs (no accumulate) and are deterministic (no float atomics).
 * ------------------------------------------------------------------------------------------ */
size_t pcmi_spconv_workspace_bytes(int64_t n_in, int64_t n_out, int cin, int cout, int K,
                                   int64_t M);
int pcmi_spconv_fwd(const float* in, int64_t in_ld, int64_t n_in, int cin, const float* weight,
                    int cout, const pcmi_kmap_t* map, int transpose, const float* bias,
                    float* out, int64_t out_ld, int64_t n_out, void* ws, size_t ws_bytes,
                    pcmi_stream_t stream);
int pcmi_spconv_bwd_data(const float* gout, int64_t gout_ld, int64_t n_out, int cout,
                         const float* weight, int cin, const pcmi_kmap_t* map, int transpose,
                         float* gin, int64_t gin_ld, int64_t n_in, void* ws, size_t ws_bytes,
                         pcmi_stream_t stream);
int pcmi_spconv_bwd_weight(const float* in, int64_t in_ld, int64_t n_in, int cin,
                           const float* gout, int64_t gout_ld, int64_t n_out, int cout,
                           const pcmi_kmap_t* map, int transpose, float* gweight,
                           float* gbias /* nullable [cout] */, void* ws, size_t ws_bytes,
                           pcmi_stream_t stream);
/* 1 if the matrix-bound forward / backward-data launches (>= 64 channels on both sides, >= 512 rows) run the
 * split-precision kernel -- fp32 operands as three bf16 terms each on the bf16 matrix cores, fp32 accumulation,
 * results within fp32 round-off of the fp32-MFMA kernel (csrc/spconv_x3.hip) -- else 0 (environment
 * PCMI_CONV16_X3=0).  The reference has no counterpart: MinkowskiEngine's GEMMs are cuBLAS fp32
 * (pc/model/modules/common.py:117-168 reach them through ME.MinkowskiConvolution). */
int pcmi_spconv_split_precision(void);

/* ------------------------------------------------------------------------------------------
 * Normalisation / elementwise
 *   BatchNorm1d inside ME.MinkowskiBatchNorm      pc/model/modules/common.py:19-21
 *   MinkowskiReLU, `out += residual`              pc/model/modules/resnet_block.py:41,57
 *   L2 row normalisation of the output features   pc/model/res16unet.py:262-266
 * bn_fwd_train: batch statistics over the n rows (biased var for normalisation, unbiased
 * for the running estimate, as torch), y = relu?((x-mean)*invstd*gamma+beta (+ residual)).
 * bn_bwd: gradients of that fused expression; relu_mask_y (nullable) is the forward output
 * whose sign gives the ReLU mask; dres (nullable) receives the residual-branch gradient.
 * ------------------------------------------------------------------------------------------ */
size_t pcmi_bn_workspace_bytes(int64_t n, int c);
int pcmi_bn_fwd_train(const float* x, int64_t x_ld, int64_t n, int c, const float* gamma,
                      const float* beta, float* running_mean, float* running_var,
                      float momentum, float eps, const float* residual, int64_t res_ld, int relu,
                      float* y, int64_t y_ld, float* save_mean, float* save_invstd, void* ws,
                      size_t ws_bytes, pcmi_stream_t stream);
int pcmi_bn_fwd_eval(const float* x, int64_t x_ld, int64_t n, int c, const float* gamma,
                     const float* beta, const float* running_mean, const float* running_var,
                     float eps, const float* residual, int64_t res_ld, int relu, float* y,
                     int64_t y_ld, pcmi_stream_t stream);
int pcmi_bn_bwd(const float* dy, int64_t dy_ld, const float* x, int64_t x_ld,
                const float* relu_mask_y, int64_t y_ld, int64_t n, int c, const float* gamma,
                const float* save_mean, const float* save_invstd, float* dx, int64_t dx_ld,
                float* dres, int64_t dres_ld, float* dgamma, float* dbeta, void* ws,
                size_t ws_bytes, pcmi_stream_t stream);
int pcmi_relu_fwd(const float* x, int64_t x_ld, int64_t n, int c, float* y, int64_t y_ld,
                  pcmi_stream_t stream);
int pcmi_relu_bwd(const float* dy, int64_t dy_ld, const float* y, int64_t y_ld, int64_t n, int c,
                  float* dx, int64_t dx_ld, pcmi_stream_t stream);
int pcmi_add(const float* a, int64_t a_ld, const float* b, int64_t b_ld, int64_t n, int c,
             float* y, int64_t y_ld, pcmi_stream_t stream);
int pcmi_l2norm_fwd(const float* x, int64_t x_ld, int64_t n, int c, float* y, int64_t y_ld,
                    float* norm /* [n] */, pcmi_stream_t stream);
int pcmi_l2norm_bwd(const float* dy, int64_t dy_ld, const float* y, int64_t y_ld,
                    const float* norm, int64_t n, int c, float* dx, int64_t dx_ld,
                    pcmi_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Row gather / scatter-add used by the losses (F0[q_idx], pc/lib/ddp_trainer.py:209-213,409-410)
 * scatter_add accumulates into dst (caller zero-fills); duplicate indices are summed in increasing source-row
 * order, without float atomics (bit-reproducible; n^2 / 64 wave steps: meant for the losses' n of a few thousand).
 * ------------------------------------------------------------------------------------------ */
int pcmi_gather_rows(const float* src, int64_t src_ld, const int64_t* idx, int64_t n, int c,
                     float* dst, int64_t dst_ld, pcmi_stream_t stream);
int pcmi_scatter_add_rows(const float* src, int64_t src_ld, const int64_t* idx, int64_t n, int c,
                          float* dst, int64_t dst_ld, pcmi_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Positive-pair selection of the PointInfoNCE step (pc/lib/ddp_trainer.py:400-417; csrc/pairs.hip):
 *   q_unique, count = pos_pairs[:, 0].unique(return_counts=True); off = floor(uniform * count);
 *   k_sel = pos_pairs[:, 1][off + exclusive_cumsum(count)]; optional sub-sample [sampled] of both.
 * pairs: device int32 [n_pairs, 2] sorted by column 0 (the loader's contract, pc/lib/ddp_data_loaders.py:43-48,85-91);
 * uniform: device fp32 [n_unique] -- the host's torch.rand(n_unique) draws; sampled: nullable device int64 [n_sel] --
 * the host's np.random.choice(n_unique, npos) (NULL: n_sel == n_unique, every query in order).  Writes q_idx / k_idx
 * (device int64 [n_sel]: rows of F0 / F1).  The random draws stay on the host so that the reference's generator streams
 * are consumed identically; for them the host needs n_unique: pcmi_pairs_scan_host, one pass over column 0 of the
 * HOST copy of the correspondences (*sorted_host == 0: the column is not sorted -- sort before using either call).
 * ------------------------------------------------------------------------------------------ */
size_t pcmi_pair_select_workspace_bytes(int64_t n_pairs);
int pcmi_pair_select(const int32_t* pairs, int64_t n_pairs, int64_t n_unique, const float* uniform,
                     const int64_t* sampled, int64_t n_sel, int64_t* q_idx, int64_t* k_idx, void* ws,
                     size_t ws_bytes, pcmi_stream_t stream);
int pcmi_pairs_scan_host(const int32_t* pairs_host, int64_t n_pairs, int64_t* n_runs_host, int* sorted_host);

/* ------------------------------------------------------------------------------------------
 * PointInfoNCE block -- replaces torch.mm + nn.CrossEntropyLoss
 * (pc/lib/ddp_trainer.py:419-426, pc/lib/criterion.py:13-18):
 *   loss = mean_i( logsumexp_j(q_i.k_j / T) - q_i.k_i / T ),  q, k: [n, c] fp32.
 * The n x n logits are never materialised.  fwd writes lse[n] and *loss (device scalar);
 * bwd writes dq, dk for upstream gradient gscale (device scalar, nullable -> 1).
 * ------------------------------------------------------------------------------------------ */
size_t pcmi_nce_workspace_bytes(int64_t n, int c);
int pcmi_nce_fwd(const float* q, const float* k, int64_t n, int c, float inv_T, float* lse,
                 float* loss, void* ws, size_t ws_bytes, pcmi_stream_t stream);
int pcmi_nce_bwd(const float* q, const float* k, const float* lse, int64_t n, int c, float inv_T,
                 const float* gscale, float* dq, float* dk, void* ws, size_t ws_bytes,
                 pcmi_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Hardest-contrastive block (pc/lib/ddp_trainer.py:182-238)
 *   pdist_argmin: dmin[p] = min_s sqrt(|a_p - b_s|^2 + 1e-7), amin[p] = first arg min
 *                 (replaces pdist :182-184 + .min(1) :218-219; no [P,S,C] temporary)
 *   keyset: device hash set of int64 keys a + b*M (replaces _hash :39-51 + np.isin :231-234)
 *   hardest_loss: pos = mean(relu(|a-b|^2 - pos_thresh)); neg = (mean_masked(relu(nt-D01)^2)
 *                 + mean_masked(relu(nt-D10)^2)) / 2 (fwd), and the gradients w.r.t. the four
 *                 gathered matrices (bwd).
 * ------------------------------------------------------------------------------------------ */
int pcmi_pdist_argmin(const float* a, int64_t p, const float* b, int64_t s, int c, float* dmin,
                      int32_t* amin, pcmi_stream_t stream);
size_t pcmi_keyset_bytes(int64_t n_keys);
int pcmi_keyset_build(const int32_t* pairs /* [n,2] */, int64_t n, int64_t M, void* set,
                      size_t set_bytes, pcmi_stream_t stream);
/* mask[p] = 1 iff (a[p] + b[p]*M) is NOT in the set (i.e. the mined negative is kept). */
int pcmi_keyset_mask_absent(const void* set, size_t set_bytes, const int64_t* a, const int64_t* b,
                            int64_t n, int64_t M, uint8_t* mask, pcmi_stream_t stream);
int pcmi_hardest_loss_fwd(const float* posF0, const float* posF1, int64_t p, int c,
                          const float* d01min, const uint8_t* mask0, const float* d10min,
                          const uint8_t* mask1, float pos_thresh, float neg_thresh,
                          float* losses /* [2]: pos, neg */, float* stats /* [5], for bwd */,
                          void* ws, size_t ws_bytes, pcmi_stream_t stream);
/* gl: device [2] upstream gradients of (pos, neg).  dsub* are accumulated (caller zero-fills). */
int pcmi_hardest_loss_bwd(const float* posF0, const float* posF1, int64_t p, const float* subF0,
                          const float* subF1, int c, const float* d01min, const int32_t* d01ind,
                          const uint8_t* mask0, const float* d10min, const int32_t* d10ind,
                          const uint8_t* mask1, float pos_thresh, float neg_thresh,
                          const float* stats, const float* gl, float* dposF0, float* dposF1,
                          float* dsubF0, float* dsubF1, pcmi_stream_t stream);
size_t pcmi_hardest_workspace_bytes(int64_t p);

/* ------------------------------------------------------------------------------------------
 * Optimiser -- replaces torch.optim.SGD.step (pc/lib/ddp_trainer.py:107-111,319,435) on a
 * flat fp32 buffer:  g = grad_scale*g + wd*w;  v = mu*v + g;  w -= lr*v.
 * (dampening 0, no Nesterov; a zero-filled v reproduces torch's first-step buffer init.)
 * ------------------------------------------------------------------------------------------ */
/* ---- loader-side geometry on the device (csrc/loader.hip) -------------------------------------
 * pcmi_voxelize  = ME.utils.sparse_quantize(xyz / voxel_size, return_index=True) of the dataset item
 *   (pc/lib/ddp_data_loaders.py:228-229): first_index[0 .. *n_unique) = ascending indices of the first point of every
 *   occupied voxel, coords (nullable, [n_unique, 3]) = floor(xyz[first_index] / voxel_size).  xyz: device fp64 [n, 3].
 * pcmi_match_radius = get_matching_indices (pc/lib/ddp_data_loaders.py:36-49): all (i, j) with
 *   |R src_i + t - dst_j| <= radius, rigid3x4_host = [R | t] row-major (12 host doubles); pairs [*n_pairs, 2] sorted by
 *   (i, j).  pairs == NULL: only counts.  PCMI_ERR_WORKSPACE if pairs_capacity is too small (*n_pairs_host = needed).
 * Both synchronise the stream (they return counts) and are bit-exact against oracle/loader_ref.py. */
size_t pcmi_voxelize_workspace_bytes(int64_t n);
int pcmi_voxelize(const double* xyz, int64_t n, double voxel_size, int32_t* first_index, int32_t* coords,
                  int64_t* n_unique_host, void* ws, size_t ws_bytes, pcmi_stream_t stream);
size_t pcmi_match_radius_workspace_bytes(int64_t n0, int64_t n1);
int pcmi_match_radius(const double* src, int64_t n0, const double* rigid3x4_host, const double* dst, int64_t n1,
                      double radius, int32_t* pairs, int64_t pairs_capacity, int64_t* n_pairs_host, void* ws,
                      size_t ws_bytes, pcmi_stream_t stream);

/* Softmax cross-entropy over the rows of logits [n, c] with an ignore label -- the loss of the downstream semantic
 * segmentation fine-tuning that reuses this backbone with out_channels = number of classes
 * (downstream/semseg/lib/train.py:64,124: nn.CrossEntropyLoss(ignore_index=config.ignore_label)).
 * out2[0] = mean loss over the counted rows, out2[1] = their number (device).  _bwd: dlogits = gloss[0] * dloss/dlogits.
 * A label that is neither in [0, c) nor the ignore label (torch raises for it) makes the loss and its row of dlogits
 * NaN -- a mis-mapped dataset label fails loudly instead of dropping points from the loss. */
size_t pcmi_softmax_ce_workspace_bytes(int64_t n);
int pcmi_softmax_ce_fwd(const float* logits, int64_t ld, int64_t n, int c, const int32_t* labels,
                        int ignore_label, float* out2, void* ws, size_t ws_bytes, pcmi_stream_t stream);
int pcmi_softmax_ce_bwd(const float* logits, int64_t ld, int64_t n, int c, const int32_t* labels,
                        int ignore_label, const float* out2, const float* gloss, float* dlogits,
                        int64_t d_ld, pcmi_stream_t stream);

int pcmi_sgd_step(float* w, const float* g, float* v, int64_t n, float lr, float momentum,
                  float weight_decay, float grad_scale, pcmi_stream_t stream);
/* The same with torch's dampening (the downstream fine-tuning's optimiser: SGD(lr, sgd_momentum, dampening =
 * sgd_dampening 0.1, weight_decay), downstream/semseg/lib/solvers.py:52-60, config/default.yaml:16-19):
 *   v = mu*v + (1 - dampening)*g, except on the optimiser's first step (first_step != 0, v zero-filled), where torch
 *   initialises the buffer with g itself. */
int pcmi_sgd_step_dampened(float* w, const float* g, float* v, int64_t n, float lr, float momentum,
                           float dampening, float weight_decay, float grad_scale, int first_step,
                           pcmi_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Network executor -- the reference drives the 63 convs / 62 BNs of Res16UNet34C from Python
 * (pc/model/res16unet.py:206-268, autograd for the backward).  Here a model is lowered ONCE into
 * a static program (tensors + ops, built by tracing the Python module tree) and a whole forward
 * or backward is ONE call that enqueues every kernel from C++: no per-layer host overhead,
 * activations in a per-pass arena, concatenations as column slices of a shared buffer
 * (zero copy), gradients accumulated straight into the flat gradient buffer.
 *   tensor: (level, channels) rows = voxels of that level (tensor stride 2^level); parent >= 0
 *           makes it the column slice [col_off, col_off + channels) of tensor `parent`.
 *   op    : CONV (MinkowskiConvolution / Transpose), BN (+ residual, + ReLU), L2NORM.
 * The training step of the reference needs two passes per iteration (one per point cloud,
 * pc/lib/ddp_trainer.py:392-398): activations of pass p stay valid until pcmi_net_backward(p).
 * ------------------------------------------------------------------------------------------ */
#define PCMI_OP_CONV 0
#define PCMI_OP_BN 1
#define PCMI_OP_L2NORM 2

typedef struct pcmi_net pcmi_net_t;
typedef struct pcmi_net_tensor {
  int32_t level, channels, parent, col_off;
} pcmi_net_tensor_t;
typedef struct pcmi_net_op {
  int32_t type, in, in2, out;   /* tensor ids; in2 = residual input of a BN or -1 */
  int32_t cin, cout, kernel_size, stride, region, transpose, relu, has_bias;
  int64_t w_off, b_off;         /* flat-parameter offsets: conv kernel / bias, BN gamma / beta */
  float* running_mean;          /* BN buffers (device pointers; not part of the flat parameters) */
  float* running_var;
  float momentum, eps;
} pcmi_net_op_t;
/* Called from pcmi_net_backward as soon as every gradient of parameter range `bucket` has been ENQUEUED -- on the
 * backward stream and, for the weight gradients, on the executor's own low-priority stream.  The backward chain does not
 * wait for that stream at a bucket boundary; a consumer (the gradient all-reduce, pc/lib/ddp_trainer.py:96-102 = what
 * DistributedDataParallel's bucket hooks do) orders ITS stream behind both from inside the callback with
 * pcmi_net_stream_wait_bucket. */
typedef void (*pcmi_ready_fn)(void* ctx, int bucket);

int pcmi_net_create(const pcmi_net_tensor_t* tensors, int n_tensors, const pcmi_net_op_t* ops,
                    int n_ops, int input_tensor, int output_tensor, int n_passes, pcmi_net_t** out);
int pcmi_net_destroy(pcmi_net_t* net);
/* in_feats [n_rows, in_ld] and out_feats [n_rows, out_ld] are caller memory; coords must hold
 * key 0 with n_rows rows.  May sync the first times (arena growth, coordinate planning).
 * training: 0 = eval (running BN estimates), 1 = train (batch statistics, running estimates
 * updated in place as torch's BatchNorm1d), 1 | PCMI_NET_DEFER_RUNNING_STATS = train, but the
 * running-estimate updates of this pass are only applied by pcmi_net_apply_running_stats: two
 * passes (the two clouds of a pair, ddp_trainer.py:404-407) can then be forwarded concurrently
 * on two streams and still update the estimates in the reference's order (pass 0, then 1). */
#define PCMI_NET_DEFER_RUNNING_STATS 2
int pcmi_net_forward(pcmi_net_t* net, int pass, pcmi_coords_t* coords, const float* in_feats,
                     int64_t in_ld, int64_t n_rows, const float* params, int training,
                     float* out_feats, int64_t out_ld, pcmi_stream_t stream);
/* d_out: gradient w.r.t. out_feats.  Parameter gradients are ACCUMULATED into grads (flat, same
 * layout as params; the caller zero-fills it once per iteration).  bucket_lo_host: ascending
 * flat offsets splitting the parameters into n_buckets ranges (may be NULL / 0 with ready). */
int pcmi_net_backward(pcmi_net_t* net, int pass, const float* d_out, int64_t d_ld,
                      const float* params, float* grads, const int64_t* bucket_lo_host,
                      int n_buckets, pcmi_ready_fn ready, void* ready_ctx, pcmi_stream_t stream);
/* Inside a pcmi_ready_fn callback: `stream` waits (device side, no host wait) for everything that produced the bucket
 * the callback announces.  PCMI_ERR_INVALID outside a backward pass that has announced a bucket. */
int pcmi_net_stream_wait_bucket(pcmi_net_t* net, pcmi_stream_t stream);
int pcmi_net_apply_running_stats(pcmi_net_t* net, int pass, pcmi_stream_t stream);
/* Copy of one activation tensor of the last forward of `pass` (they stay in the pass's arena until its next forward)
 * into caller memory out [rows, out_ld]; rows / channels (nullable) report its shape, out == NULL only queries.  For
 * inspection and for tests that hand the device's ReLU patterns to the oracle (the reference has no counterpart: its
 * activations are autograd-owned torch tensors). */
int pcmi_net_export_tensor(pcmi_net_t* net, int pass, int tensor, int64_t* rows, int* channels, float* out,
                           int64_t out_ld, pcmi_stream_t stream);
int pcmi_net_memory_bytes(pcmi_net_t* net, size_t* bytes);
/* Measurement (bench.py roofline.in_step_ms; the reference has no counterpart: torch.autograd.profiler would be its
 * tool): timing events around the convolution launches of the ops `ops[0..n_ops)` (indices into the program) INSIDE the
 * passes -- forward, the backward(-data) launch and (on the executor's weight-gradient stream) the weight-gradient launch
 * of the same op; any op type -- in a ring of n_sets event sets, one per forward pass enqueued after this call (networks run as one
 * pass per iteration; the backward pass records into the set of the forward before it).  n_ops == 0 stops timing.
 * Synchronises the device.  pcmi_net_timed_ms waits for set `set` and returns the elapsed stream time per op in ms
 * (-1: not recorded). */
int pcmi_net_time_ops(pcmi_net_t* net, const int* ops, int n_ops, int n_sets);
int pcmi_net_timed_ms(pcmi_net_t* net, int set, float* fwd_ms, float* bwd_ms, float* wgrad_ms, int n_ops);
/* The same for EVERY op of the program (convolutions, BatchNorms, the L2 normalisation: forward and backward launches on
 * the pass's stream, weight gradients on the executor's side stream) -- bench.py's per-layer times and `families[]`.  The
 * weight gradients the executor collects into grouped launches (coarse levels) stay grouped; those launches are timed as
 * such: pcmi_net_timed_groups_ms returns up to `cap` of them for set `set` (n_out = how many).  n_sets == 0 stops timing.
 * pcmi_net_timed_ms then takes n_ops = the number of ops of the program. */
int pcmi_net_time_all(pcmi_net_t* net, int n_sets);
int pcmi_net_timed_groups_ms(pcmi_net_t* net, int set, float* ms, int cap, int* n_out);
/* Kernel launches each timed call of set `set` enqueued (forward, backward(-data), weight gradient; 0 = not recorded), and
 * -- groups != NULL after pcmi_net_time_all -- of its first groups_cap grouped weight-gradient launches. */
int pcmi_net_timed_launches(pcmi_net_t* net, int set, int* fwd, int* bwd, int* wgrad, int n_ops, int* groups, int groups_cap);

#ifdef __cplusplus
}
#endif
#endif /* PCMI_H_ */
