// Coordinate manager: device hash of packed (b,x,y,z) keys, strided coordinate sets,
// kernel maps.  Replaces ME 0.4.3's CPU CoordsManager (E1-E3 in SURVEY.md 2.2) with HIP
// kernels so coordinates never return to the host: only a few integers (row counts,
// per-offset pair counts) are read back.
//
// Data layout (all int32/uint64, in the handle's persistent arena):
//   level.coords  [n,4]            rows of (b,x,y,z)
//   level.hkeys   [cap] uint64     open-addressing table of packed keys (linear probing)
//   level.hvals   [cap] int32      row index of the key
//   level.parent  [n]              row of the enclosing voxel one level coarser
//   map.nbr       [K,n_out]        output-stationary neighbour table (-1 = absent)
//   map.pair_in / pair_out [M]     the same pairs compacted per offset, ascending out-row
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "internal.h"

namespace pcmi {

// ---------------------------------------------------------------------------------------------
// error string (thread local)
// ---------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";
thread_local long g_launches = 0;

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

// ---------------------------------------------------------------------------------------------
// bump arena over hipMalloc'd chunks; reset() keeps the chunks
// ---------------------------------------------------------------------------------------------
struct Arena {
  struct Chunk {
    char* base;
    size_t size, used;
  };
  std::vector<Chunk> chunks;
  size_t min_chunk;
  explicit Arena(size_t min_chunk_bytes) : min_chunk(min_chunk_bytes) {}
  void* alloc(size_t bytes) {
    bytes = align_up(bytes ? bytes : 1, 256);
    for (auto& c : chunks) {
      if (c.size - c.used >= bytes) {
        void* p = c.base + c.used;
        c.used += bytes;
        return p;
      }
    }
    size_t sz = std::max(bytes, min_chunk);
    if (!chunks.empty()) sz = std::max(sz, std::min<size_t>(chunks.back().size * 2, (size_t)1 << 30));
    void* p = nullptr;
    if (hipMalloc(&p, sz) != hipSuccess) {
      set_error("arena: hipMalloc(%zu) failed", sz);
      return nullptr;
    }
    chunks.push_back({(char*)p, sz, bytes});
    return p;
  }
  template <typename T>
  T* alloc_n(int64_t n) {
    return (T*)alloc(sizeof(T) * (size_t)std::max<int64_t>(n, 1));
  }
  void reset() {
    for (auto& c : chunks) c.used = 0;
  }
  size_t reserved() const {
    size_t s = 0;
    for (auto& c : chunks) s += c.size;
    return s;
  }
  ~Arena() {
    for (auto& c : chunks) (void)hipFree(c.base);
  }
};

struct Level {
  int64_t n = 0;
  int ts = 1;
  int32_t* coords = nullptr;
  uint64_t* hkeys = nullptr;
  int32_t* hvals = nullptr;
  uint32_t cap = 0;
  int32_t* parent = nullptr;  // valid once the next coarser level exists
  int child_key = -1;         // key of that coarser level
  int64_t split = -1;         // >= 0: rows [0, split) belong to the first segment (pcmi_coords_set_split)
};

struct MapEntry {
  int in_key, out_key, ksize, stride, region;
  pcmi_kmap_t map;
};

}  // namespace pcmi

struct pcmi_coords {
  pcmi::Arena persistent{(size_t)64 << 20};
  pcmi::Arena scratch{(size_t)32 << 20};
  std::vector<pcmi::Level> levels;
  std::vector<pcmi::MapEntry> maps;
  int32_t* d_flags = nullptr;  // [4] device status words (dup count, range errors, ...)
  int64_t* d_total = nullptr;  // device scalar for scan totals
  int64_t* d_counts = nullptr; // [kMaxLevels] rows of every level, [kMaxLevels] first-segment rows (plan_unet's level chain)
  int64_t* h_pinned = nullptr; // pinned host staging for small read-backs (4096 bytes)
  // pcmi_coords_plan_unet builds its maps WITHOUT waiting for their per-offset pair counts: the counts are copied to the
  // map's own slot of the pinned buffer behind the build, `ev_counts` is recorded behind the last copy, and the host-side
  // fields (offs_host, M) of those maps are filled in when somebody asks for them (pcmi_kmap_get / pcmi_kmap_export) --
  // nothing on the training path does: the kernels read the device-side offsets and size their launches by bounds.
  bool defer_maps = false;
  std::vector<std::pair<int, int>> pending;  // (index in `maps`, slot)
  hipEvent_t ev_counts = nullptr;
  // recorded on the planning stream at the end of pcmi_coords_plan_unet: a consumer on another stream
  // (pcmi_net_forward) waits for it before it reads the tables
  hipEvent_t ev_plan = nullptr;
  bool plan_recorded = false;
  bool insert_unchecked = false;  // pcmi_coords_insert_deferred: the status words have not been read yet
};
constexpr int kMaxLevels = 16;
constexpr int kStatusSlot = 2;  // h_pinned[2]: the two int32 status words of the insert (duplicates, out-of-range rows)
constexpr int kLevelSlot0 = 8;  // h_pinned[8 .. 8 + 2 * kMaxLevels): level sizes and segment boundaries of the level chain
constexpr int kMapSlot0 = 64, kMapSlotLen = PCMI_MAX_KERNEL_VOLUME + 1, kMapSlots = (512 - kMapSlot0) / kMapSlotLen;

static int resolve_pending_maps(pcmi_coords* h) {
  if (h->pending.empty()) return PCMI_OK;
  PCMI_HIP_CHECK(hipEventSynchronize(h->ev_counts));
  for (const auto& pr : h->pending) {
    pcmi_kmap_t& m = h->maps[pr.first].map;
    memcpy(m.offs_host, h->h_pinned + kMapSlot0 + (size_t)pr.second * kMapSlotLen, sizeof(int64_t) * (m.K + 1));
    m.M = m.offs_host[m.K];
  }
  h->pending.clear();
  return PCMI_OK;
}

// the duplicate / range status of the insert, once a synchronisation of its stream has made the pinned copy current
// (every synchronising call of this file ends with it: a deferred insert reports its error there)
static int check_insert_status(pcmi_coords* h) {
  if (!h->insert_unchecked) return PCMI_OK;
  h->insert_unchecked = false;
  const int32_t* f = (const int32_t*)(h->h_pinned + kStatusSlot);
  PCMI_REQUIRE(f[1] == 0, PCMI_ERR_RANGE, "coords_insert: %d rows outside the packable range", f[1]);
  PCMI_REQUIRE(f[0] == 0, PCMI_ERR_DUPLICATE, "coords_insert: %d duplicate coordinates", f[0]);
  return PCMI_OK;
}

// a duplicate / out-of-range row: the handle holds nothing afterwards (levels, maps and both arenas), whichever call
// found out -- the synchronous insert, the level chain or the end of pcmi_coords_plan_unet
static void discard_after_bad_insert(pcmi_coords_t* h) {
  h->levels.clear();
  h->maps.clear();
  h->pending.clear();
  h->plan_recorded = false;
  h->persistent.reset();
  h->scratch.reset();
}

namespace pcmi {

// ---------------------------------------------------------------------------------------------
// exclusive scan of flags (in[i] >= 0) or of values, int32 positions, multi-level
// ---------------------------------------------------------------------------------------------
constexpr int kScanThreads = 256;
constexpr int kScanItems = 16;
constexpr int kScanTile = kScanThreads * kScanItems;

template <bool FLAGS>
__global__ __launch_bounds__(kScanThreads) void scan_tile_kernel(const int32_t* __restrict__ in,
                                                                 int64_t n,
                                                                 int32_t* __restrict__ out,
                                                                 int32_t* __restrict__ tile_sums) {
  __shared__ int32_t wave_sums[kScanThreads / 64];
  const int64_t base = (int64_t)blockIdx.x * kScanTile + (int64_t)threadIdx.x * kScanItems;
  int32_t v[kScanItems];
  int32_t local = 0;
#pragma unroll
  for (int i = 0; i < kScanItems; ++i) {
    int64_t idx = base + i;
    int32_t x = 0;
    if (idx < n) {
      int32_t r = in[idx];
      x = FLAGS ? (r >= 0 ? 1 : 0) : r;
    }
    v[i] = local;
    local += x;
  }
  // inclusive scan of `local` across the wave
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int32_t incl = local;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    int32_t t = __shfl_up(incl, d, 64);
    if (lane >= d) incl += t;
  }
  if (lane == 63) wave_sums[wave] = incl;
  __syncthreads();
  int32_t wave_base = 0, total = 0;
#pragma unroll
  for (int w = 0; w < kScanThreads / 64; ++w) {
    int32_t s = wave_sums[w];
    if (w < wave) wave_base += s;
    total += s;
  }
  const int32_t thread_base = wave_base + incl - local;
#pragma unroll
  for (int i = 0; i < kScanItems; ++i) {
    int64_t idx = base + i;
    if (idx < n) out[idx] = thread_base + v[i];
  }
  if (threadIdx.x == 0) tile_sums[blockIdx.x] = total;
}

__global__ void scan_add_kernel(int32_t* __restrict__ out, int64_t n,
                                const int32_t* __restrict__ tile_offsets) {
  int64_t idx = (int64_t)blockIdx.x * kScanTile + threadIdx.x;
  const int32_t off = tile_offsets[blockIdx.x];
#pragma unroll
  for (int i = 0; i < kScanItems; ++i, idx += kScanThreads)
    if (idx < n) out[idx] += off;
}

template <bool FLAGS>
__global__ void scan_total_kernel(const int32_t* in, const int32_t* out, int64_t n,
                                  int64_t* total) {
  if (n == 0) {
    *total = 0;
    return;
  }
  int32_t r = in[n - 1];
  *total = (int64_t)out[n - 1] + (FLAGS ? (r >= 0 ? 1 : 0) : r);
}

// out[i] = exclusive prefix; if total != nullptr, *total = sum.  Scratch from `tmp`.
template <bool FLAGS>
static int exclusive_scan(const int32_t* in, int64_t n, int32_t* out, int64_t* total, Arena& tmp,
                          hipStream_t st) {
  if (n > 0) {
    const int64_t tiles = ceil_div(n, kScanTile);
    int32_t* sums = tmp.alloc_n<int32_t>(tiles);
    if (!sums) return PCMI_ERR_HIP;
    scan_tile_kernel<FLAGS><<<dim3((unsigned)tiles), kScanThreads, 0, st>>>(in, n, out, sums);
    PCMI_LAUNCH_CHECK();
    if (tiles > 1) {
      int32_t* sums_scanned = tmp.alloc_n<int32_t>(tiles);
      if (!sums_scanned) return PCMI_ERR_HIP;
      int rc = exclusive_scan<false>(sums, tiles, sums_scanned, nullptr, tmp, st);
      if (rc) return rc;
      scan_add_kernel<<<dim3((unsigned)tiles), kScanThreads, 0, st>>>(out, n, sums_scanned);
      PCMI_LAUNCH_CHECK();
    }
  }
  if (total) {
    scan_total_kernel<FLAGS><<<1, 1, 0, st>>>(in, out, n, total);
    PCMI_LAUNCH_CHECK();
  }
  return PCMI_OK;
}

// ---------------------------------------------------------------------------------------------
// hash table
// ---------------------------------------------------------------------------------------------
__device__ inline int32_t table_lookup(const uint64_t* __restrict__ keys,
                                       const int32_t* __restrict__ vals, uint32_t mask,
                                       uint64_t key) {
  uint32_t slot = hash_key(key) & mask;
  while (true) {
    const uint64_t k = keys[slot];
    if (k == key) return vals[slot];
    if (k == kEmptyKey) return -1;
    slot = (slot + 1) & mask;
  }
}

// insert key with value min(value, row); returns the slot
__device__ inline uint32_t table_insert_min(uint64_t* keys, int32_t* vals, uint32_t mask,
                                            uint64_t key, int32_t row, bool* existed) {
  uint32_t slot = hash_key(key) & mask;
  while (true) {
    const unsigned long long prev =
        atomicCAS((unsigned long long*)&keys[slot], (unsigned long long)kEmptyKey,
                  (unsigned long long)key);
    if (prev == kEmptyKey || prev == key) {
      *existed = (prev == key);
      atomicMin(&vals[slot], row);
      return slot;
    }
    slot = (slot + 1) & mask;
  }
}

__global__ void insert_kernel(const int32_t* __restrict__ coords, int64_t n, uint64_t* keys,
                              int32_t* vals, uint32_t mask, int32_t* status) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int4 c = reinterpret_cast<const int4*>(coords)[i];
  const bool ok = c.x >= 0 && c.x < 1023 && c.y > -kCoordBias && c.y < kCoordBias &&
                  c.z > -kCoordBias && c.z < kCoordBias && c.w > -kCoordBias && c.w < kCoordBias;
  if (!ok) {
    atomicAdd(&status[1], 1);
    return;
  }
  bool existed;
  table_insert_min(keys, vals, mask, pack_key(c.x, c.y, c.z, c.w), (int32_t)i, &existed);
  if (existed) atomicAdd(&status[0], 1);
}

// strided level, pass 1: insert the quantised key, remember the slot
// (the strided-level kernels take their row count from the device when n_dev != nullptr: pcmi_coords_plan_unet builds
//  the whole level chain without reading a count back; `n` is then the launch bound)
__global__ void stride_insert_kernel(const int32_t* __restrict__ coords, int64_t n, const int64_t* __restrict__ n_dev, int ts2,
                                     uint64_t* keys, int32_t* vals, uint32_t mask,
                                     uint32_t* __restrict__ slot_of) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (n_dev) n = *n_dev;
  if (i >= n) return;
  const int4 c = reinterpret_cast<const int4*>(coords)[i];
  const uint64_t key = pack_key(c.x, floor_div(c.y, ts2) * ts2, floor_div(c.z, ts2) * ts2,
                                floor_div(c.w, ts2) * ts2);
  bool existed;
  slot_of[i] = table_insert_min(keys, vals, mask, key, (int32_t)i, &existed);
}

// pass 2: first-occurrence flags (0 = this row is the lowest child of its parent, -1 otherwise)
__global__ void stride_flag_kernel(int64_t n, const int64_t* __restrict__ n_dev, const int32_t* __restrict__ vals,
                                   const uint32_t* __restrict__ slot_of,
                                   int32_t* __restrict__ flags) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (n_dev && i >= *n_dev) {  // between the actual count and the launch bound: never a first occurrence
    flags[i] = -1;
    return;
  }
  flags[i] = (vals[slot_of[i]] == (int32_t)i) ? 0 : -1;
}

// pass 3: parent row of every fine row; coordinates of the coarse rows
__global__ void stride_parent_kernel(const int32_t* __restrict__ coords, int64_t n, const int64_t* __restrict__ n_dev, int ts2,
                                     const int32_t* __restrict__ vals,
                                     const uint32_t* __restrict__ slot_of,
                                     const int32_t* __restrict__ pos,
                                     int32_t* __restrict__ parent,
                                     int32_t* __restrict__ coarse_coords) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (n_dev) n = *n_dev;
  if (i >= n) return;
  const int32_t first_child = vals[slot_of[i]];
  const int32_t row = pos[first_child];
  parent[i] = row;
  if (first_child == (int32_t)i) {
    const int4 c = reinterpret_cast<const int4*>(coords)[i];
    reinterpret_cast<int4*>(coarse_coords)[row] =
        make_int4(c.x, floor_div(c.y, ts2) * ts2, floor_div(c.z, ts2) * ts2,
                  floor_div(c.w, ts2) * ts2);
  }
}

// pass 4: table values become coarse row ids
__global__ void stride_relabel_kernel(int64_t n, const int64_t* __restrict__ n_dev, const int32_t* __restrict__ flags,
                                      const uint32_t* __restrict__ slot_of,
                                      const int32_t* __restrict__ pos, int32_t* vals) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (n_dev) n = *n_dev;
  if (i >= n) return;
  if (flags[i] == 0) vals[slot_of[i]] = pos[i];
}

// level chain on the device: counts[l] = rows of level l, counts[kMaxLevels + l] = rows of its first segment (-1: none)
__global__ void chain_init_kernel(int64_t* counts, int64_t n0, int64_t split0) {
  counts[0] = n0;
  counts[kMaxLevels] = split0;
}
// segment boundary of the coarse level = number of first occurrences among the fine rows [0, split) = pos[split]
__global__ void chain_split_kernel(const int32_t* __restrict__ pos, int64_t* counts, int l) {
  const int64_t n = counts[l], fs = counts[kMaxLevels + l], nc = counts[l + 1];
  counts[kMaxLevels + l + 1] = fs < 0 ? -1 : (fs > 0 && fs < n ? (int64_t)pos[fs] : (fs >= n ? nc : 0));
}

// ---------------------------------------------------------------------------------------------
// kernel maps
// ---------------------------------------------------------------------------------------------
struct OffsetTable {
  int8_t o[PCMI_MAX_KERNEL_VOLUME][3];
};

constexpr int kMapTile = 128;
constexpr int64_t kSortRowsMin = 4096;  // levels below this are latency-, not MFMA-bound: no sorted order  // rows per tile == rows per spconv workgroup tile

// 3^3 / stride-1 map.  One workgroup = one 128-row tile: the tile's coordinates are staged in
// LDS once, the 27*128 probes are spread over the 256 threads (consecutive threads take the
// consecutive offsets of one voxel, so probe-chain length variance is shared), results are
// staged in LDS and leave as 27 coalesced 512-byte row segments.
__global__ __launch_bounds__(256) void kmap_k3_kernel(const int32_t* __restrict__ coords,
                                                      int64_t n, int ts, OffsetTable offs,
                                                      const uint64_t* __restrict__ keys,
                                                      const int32_t* __restrict__ vals,
                                                      uint32_t mask, int32_t* __restrict__ nbr) {
  __shared__ int4 s_c[kMapTile];
  __shared__ int32_t s_nbr[27][kMapTile];
  const int64_t j0 = (int64_t)blockIdx.x * kMapTile;
  const int t = threadIdx.x;
  if (t < kMapTile) {
    const int64_t j = j0 + t;
    s_c[t] = j < n ? reinterpret_cast<const int4*>(coords)[j] : make_int4(-1, 0, 0, 0);
  }
  __syncthreads();
  for (int p = t; p < 27 * kMapTile; p += 256) {
    const int r = p / 27, k = p - r * 27;
    const int4 c = s_c[r];
    int32_t v = -1;
    if (c.x >= 0) {
      const int x = c.y + offs.o[k][0] * ts, y = c.z + offs.o[k][1] * ts,
                z = c.w + offs.o[k][2] * ts;
      if (x > -kCoordBias && x < kCoordBias && y > -kCoordBias && y < kCoordBias &&
          z > -kCoordBias && z < kCoordBias)
        v = table_lookup(keys, vals, mask, pack_key(c.x, x, y, z));
    }
    s_nbr[k][r] = v;
  }
  __syncthreads();
  for (int p = t; p < 27 * kMapTile; p += 256) {
    const int k = p / kMapTile, r = p - k * kMapTile;
    if (j0 + r < n) nbr[(int64_t)k * n + j0 + r] = s_nbr[k][r];
  }
}

// 2^3 / stride-2 map: child table [8, n_coarse]; every fine row has exactly one (parent, k).
__global__ void kmap_s2_kernel(const int32_t* __restrict__ coords, int64_t n_fine, int ts,
                               const int32_t* __restrict__ parent, int64_t n_coarse,
                               int32_t* __restrict__ child) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_fine) return;
  const int4 c = reinterpret_cast<const int4*>(coords)[i];
  const int ts2 = ts * 2;
  const int dx = (c.y - floor_div(c.y, ts2) * ts2) / ts, dy = (c.z - floor_div(c.z, ts2) * ts2) / ts,
            dz = (c.w - floor_div(c.w, ts2) * ts2) / ts;
  const int k = dx + 2 * dy + 4 * dz;  // axis 0 fastest (HYPERCUBE enumeration, even kernel)
  child[(int64_t)k * n_coarse + parent[i]] = (int32_t)i;
}

// compaction of the neighbour table into per-offset pair lists (ascending out-row)
__global__ void kmap_compact_kernel(const int32_t* __restrict__ nbr, int K, int64_t n_out,
                                    const int32_t* __restrict__ pos, const int64_t* total,
                                    int32_t* __restrict__ pair_in, int32_t* __restrict__ pair_out,
                                    int64_t* __restrict__ offs) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t tot = K * n_out;
  if (idx < tot) {
    const int32_t v = nbr[idx];
    if (v >= 0) {
      const int32_t p = pos[idx];
      pair_in[p] = v;
      pair_out[p] = (int32_t)(idx % n_out);
    }
    if (idx % n_out == 0) offs[idx / n_out] = pos[idx];
  }
  if (idx == 0) offs[K] = *total;
}

static void fill_offsets(int ksize, int region, int32_t (*o)[3], int* K_out) {
  const int s = ksize, D = 3;
  const int centre = (s % 2 == 1) ? (s - 1) / 2 : 0;
  if (region == PCMI_REGION_HYBRID && s % 2 == 1) {
    // centre first, then per axis every existing offset copied with that axis set to each
    // non-centre value (ME HYBRID region with all-HYPERCUBE axes)
    int cnt = 1;
    o[0][0] = o[0][1] = o[0][2] = 0;
    for (int d = 0; d < D; ++d) {
      const int existing = cnt;
      for (int e = 0; e < existing; ++e)
        for (int v = 0; v < s; ++v) {
          if (v == centre) continue;
          o[cnt][0] = o[e][0];
          o[cnt][1] = o[e][1];
          o[cnt][2] = o[e][2];
          o[cnt][d] = v - centre;
          ++cnt;
        }
    }
    *K_out = cnt;
    return;
  }
  int K = s * s * s;
  for (int k = 0; k < K; ++k) {
    int r = k;
    for (int d = 0; d < D; ++d) {
      o[k][d] = r % s - centre;
      r /= s;
    }
  }
  *K_out = K;
}

// small synchronous read-back: 8-byte scalars land in h_pinned[0], a map's K + 1 offsets in its slot area
static int read_back(pcmi_coords* h, const void* dev, size_t bytes, hipStream_t st) {
  PCMI_HIP_CHECK(hipMemcpyAsync(bytes <= 16 ? h->h_pinned : h->h_pinned + kMapSlot0, dev, bytes, hipMemcpyDeviceToHost, st));
  PCMI_HIP_CHECK(hipStreamSynchronize(st));
  return PCMI_OK;
}

static int new_table(pcmi_coords* h, Level& L, int64_t max_keys, hipStream_t st) {
  uint32_t cap = 1024;
  while ((int64_t)cap < max_keys + max_keys / 2 + 1) cap <<= 1;
  L.cap = cap;
  L.hkeys = h->persistent.alloc_n<uint64_t>(cap);
  L.hvals = h->persistent.alloc_n<int32_t>(cap);
  if (!L.hkeys || !L.hvals) return PCMI_ERR_HIP;
  PCMI_HIP_CHECK(hipMemsetAsync(L.hkeys, 0xFF, sizeof(uint64_t) * cap, st));
  PCMI_HIP_CHECK(hipMemsetAsync(L.hvals, 0x7F, sizeof(int32_t) * cap, st));
  return PCMI_OK;
}

}  // namespace pcmi

using namespace pcmi;

extern "C" {

int pcmi_version(void) { return 100; }
const char* pcmi_last_error(void) { return pcmi::g_err; }

int pcmi_device_info(int* n_cu, char* arch_host, int arch_len) {
  int dev = 0;
  PCMI_HIP_CHECK(hipGetDevice(&dev));
  hipDeviceProp_t prop;
  PCMI_HIP_CHECK(hipGetDeviceProperties(&prop, dev));
  if (n_cu) *n_cu = prop.multiProcessorCount;
  if (arch_host && arch_len > 0) {
    strncpy(arch_host, prop.gcnArchName, arch_len - 1);
    arch_host[arch_len - 1] = 0;
  }
  return PCMI_OK;
}

int pcmi_coords_create(int dimension, pcmi_coords_t** out) {
  PCMI_REQUIRE(dimension == 3 && out, PCMI_ERR_UNSUPPORTED, "coords: only D=3 is on the hot path");
  pcmi_coords* h = new pcmi_coords();
  if (hipMalloc((void**)&h->d_flags, 64 + sizeof(int64_t) * 2 * kMaxLevels) != hipSuccess ||
      hipHostMalloc((void**)&h->h_pinned, 4096) != hipSuccess ||
      hipEventCreateWithFlags(&h->ev_counts, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&h->ev_plan, hipEventDisableTiming) != hipSuccess) {
    set_error("coords_create: allocation failed");
    pcmi_coords_destroy(h);
    return PCMI_ERR_HIP;
  }
  h->d_total = (int64_t*)(h->d_flags + 8);
  h->d_counts = (int64_t*)(h->d_flags + 16);
  *out = h;
  return PCMI_OK;
}

int pcmi_coords_destroy(pcmi_coords_t* h) {
  if (!h) return PCMI_OK;
  if (h->d_flags) (void)hipFree(h->d_flags);
  if (h->h_pinned) (void)hipHostFree(h->h_pinned);
  if (h->ev_counts) (void)hipEventDestroy(h->ev_counts);
  if (h->ev_plan) (void)hipEventDestroy(h->ev_plan);
  delete h;
  return PCMI_OK;
}

int pcmi_coords_reset(pcmi_coords_t* h) {
  PCMI_REQUIRE(h, PCMI_ERR_INVALID, "null handle");
  h->levels.clear();
  h->maps.clear();
  h->pending.clear();
  h->defer_maps = false;
  h->plan_recorded = false;
  h->insert_unchecked = false;
  h->persistent.reset();
  h->scratch.reset();
  return PCMI_OK;
}

int pcmi_coords_arena_bytes(pcmi_coords_t* h, size_t* bytes) {
  PCMI_REQUIRE(h && bytes, PCMI_ERR_INVALID, "null argument");
  *bytes = h->persistent.reserved() + h->scratch.reserved();
  return PCMI_OK;
}

static int coords_insert_impl(pcmi_coords_t* h, const int32_t* bxyz, int64_t n, pcmi_stream_t stream, bool deferred) {
  PCMI_REQUIRE(h && (bxyz || n == 0) && n >= 0, PCMI_ERR_INVALID, "coords_insert: bad argument");
  PCMI_REQUIRE(h->levels.empty(), PCMI_ERR_INVALID, "coords_insert: handle already holds key 0 (reset first)");
  PCMI_REQUIRE(n < (1ll << 30), PCMI_ERR_RANGE, "coords_insert: too many rows");
  hipStream_t st = as_stream(stream);
  h->scratch.reset();
  Level L;
  L.n = n;
  L.ts = 1;
  L.coords = h->persistent.alloc_n<int32_t>(n * 4);
  if (!L.coords) return PCMI_ERR_HIP;
  int rc = new_table(h, L, n, st);
  if (rc) return rc;
  PCMI_HIP_CHECK(hipMemsetAsync(h->d_flags, 0, 64, st));
  if (n > 0) {
    PCMI_HIP_CHECK(hipMemcpyAsync(L.coords, bxyz, sizeof(int32_t) * 4 * n, hipMemcpyDeviceToDevice, st));
    insert_kernel<<<dim3((unsigned)ceil_div(n, 256)), 256, 0, st>>>(L.coords, n, L.hkeys, L.hvals,
                                                                   L.cap - 1, h->d_flags);
    PCMI_LAUNCH_CHECK();
  }
  // the status words travel to their pinned slot; the host looks at them behind the next synchronisation of `st`
  PCMI_HIP_CHECK(hipMemcpyAsync(h->h_pinned + kStatusSlot, h->d_flags, 8, hipMemcpyDeviceToHost, st));
  h->insert_unchecked = true;
  h->levels.push_back(L);
  if (deferred) return PCMI_OK;
  PCMI_HIP_CHECK(hipStreamSynchronize(st));
  rc = check_insert_status(h);
  if (rc) discard_after_bad_insert(h);
  return rc;
}

int pcmi_coords_insert(pcmi_coords_t* h, const int32_t* bxyz, int64_t n, pcmi_stream_t stream) {
  return coords_insert_impl(h, bxyz, n, stream, false);
}

int pcmi_coords_insert_deferred(pcmi_coords_t* h, const int32_t* bxyz, int64_t n, pcmi_stream_t stream) {
  return coords_insert_impl(h, bxyz, n, stream, true);
}

int pcmi_coords_check(pcmi_coords_t* h, pcmi_stream_t stream) {
  PCMI_REQUIRE(h, PCMI_ERR_INVALID, "null handle");
  if (!h->insert_unchecked) return PCMI_OK;
  PCMI_HIP_CHECK(hipStreamSynchronize(as_stream(stream)));
  const int rc = check_insert_status(h);
  if (rc) discard_after_bad_insert(h);
  return rc;
}

int pcmi_coords_size(pcmi_coords_t* h, int key, int64_t* n, int* tensor_stride) {
  PCMI_REQUIRE(h && key >= 0 && key < (int)h->levels.size(), PCMI_ERR_NOKEY, "coords: unknown key %d", key);
  if (n) *n = h->levels[key].n;
  if (tensor_stride) *tensor_stride = h->levels[key].ts;
  return PCMI_OK;
}

int pcmi_coords_set_split(pcmi_coords_t* h, int64_t n_first) {
  PCMI_REQUIRE(h && !h->levels.empty(), PCMI_ERR_INVALID, "coords_set_split: insert the coordinates first");
  PCMI_REQUIRE(h->levels.size() == 1, PCMI_ERR_INVALID, "coords_set_split: strided levels exist already");
  PCMI_REQUIRE(n_first >= 0 && n_first <= h->levels[0].n, PCMI_ERR_INVALID, "coords_set_split: %lld of %lld rows",
               (long long)n_first, (long long)h->levels[0].n);
  h->levels[0].split = n_first;
  return PCMI_OK;
}

int pcmi_coords_split(pcmi_coords_t* h, int key, int64_t* n_first) {
  PCMI_REQUIRE(h && n_first && key >= 0 && key < (int)h->levels.size(), PCMI_ERR_NOKEY, "coords_split: unknown key %d", key);
  *n_first = h->levels[key].split;
  return PCMI_OK;
}

int pcmi_coords_key_at_stride(pcmi_coords_t* h, int tensor_stride, int* key) {
  PCMI_REQUIRE(h && key, PCMI_ERR_INVALID, "null argument");
  for (size_t i = 0; i < h->levels.size(); ++i)
    if (h->levels[i].ts == tensor_stride) {
      *key = (int)i;
      return PCMI_OK;
    }
  set_error("coords: no key at tensor stride %d", tensor_stride);
  return PCMI_ERR_NOKEY;
}

int pcmi_coords_get(pcmi_coords_t* h, int key, int32_t* out_bxyz, pcmi_stream_t stream) {
  PCMI_REQUIRE(h && key >= 0 && key < (int)h->levels.size(), PCMI_ERR_NOKEY, "coords: unknown key %d", key);
  const Level& L = h->levels[key];
  if (L.n > 0)
    PCMI_HIP_CHECK(hipMemcpyAsync(out_bxyz, L.coords, sizeof(int32_t) * 4 * L.n,
                                  hipMemcpyDeviceToDevice, as_stream(stream)));
  return PCMI_OK;
}

int pcmi_coords_stride(pcmi_coords_t* h, int in_key, int stride, int* out_key, int64_t* n_out,
                       pcmi_stream_t stream) {
  PCMI_REQUIRE(h && in_key >= 0 && in_key < (int)h->levels.size(), PCMI_ERR_NOKEY, "coords_stride: unknown key %d", in_key);
  PCMI_REQUIRE(stride == 2, PCMI_ERR_UNSUPPORTED, "coords_stride: only stride 2 is on the hot path");
  if (h->levels[in_key].child_key >= 0) {
    const int ck = h->levels[in_key].child_key;
    if (out_key) *out_key = ck;
    if (n_out) *n_out = h->levels[ck].n;
    return PCMI_OK;
  }
  hipStream_t st = as_stream(stream);
  h->scratch.reset();
  const int64_t n = h->levels[in_key].n;
  const int ts2 = h->levels[in_key].ts * 2;
  Level C;
  C.ts = ts2;
  int rc = new_table(h, C, n, st);
  if (rc) return rc;
  int32_t* parent = h->persistent.alloc_n<int32_t>(n);
  uint32_t* slot_of = h->scratch.alloc_n<uint32_t>(n);
  int32_t* flags = h->scratch.alloc_n<int32_t>(n);
  int32_t* pos = h->scratch.alloc_n<int32_t>(n);
  if (!parent || !slot_of || !flags || !pos) return PCMI_ERR_HIP;
  const int32_t* fine = h->levels[in_key].coords;
  const dim3 grid((unsigned)std::max<int64_t>(ceil_div(n, 256), 1));
  stride_insert_kernel<<<grid, 256, 0, st>>>(fine, n, nullptr, ts2, C.hkeys, C.hvals, C.cap - 1, slot_of);
  PCMI_LAUNCH_CHECK();
  stride_flag_kernel<<<grid, 256, 0, st>>>(n, nullptr, C.hvals, slot_of, flags);
  PCMI_LAUNCH_CHECK();
  rc = exclusive_scan<true>(flags, n, pos, h->d_total, h->scratch, st);
  if (rc) return rc;
  // segment boundary of the coarse level: its rows are in first-occurrence order of the fine rows and a cell never
  // has children in both segments (they differ in the batch index), so it is the number of first occurrences among
  // the fine rows [0, split) = pos[split]
  const int64_t fsplit = h->levels[in_key].split;
  const bool carry = fsplit > 0 && fsplit < n;
  h->h_pinned[1] = 0;
  if (carry) PCMI_HIP_CHECK(hipMemcpyAsync(h->h_pinned + 1, pos + fsplit, sizeof(int32_t), hipMemcpyDeviceToHost, st));
  rc = read_back(h, h->d_total, 8, st);
  if (rc) return rc;
  rc = check_insert_status(h);
  if (rc) return rc;
  C.n = h->h_pinned[0];
  C.split = fsplit < 0 ? -1 : (carry ? (int64_t)(h->h_pinned[1] & 0xFFFFFFFFll) : (fsplit >= n ? C.n : 0));
  C.coords = h->persistent.alloc_n<int32_t>(C.n * 4);
  if (!C.coords) return PCMI_ERR_HIP;
  stride_parent_kernel<<<grid, 256, 0, st>>>(fine, n, nullptr, ts2, C.hvals, slot_of, pos, parent, C.coords);
  PCMI_LAUNCH_CHECK();
  stride_relabel_kernel<<<grid, 256, 0, st>>>(n, nullptr, flags, slot_of, pos, C.hvals);
  PCMI_LAUNCH_CHECK();
  h->levels.push_back(C);
  const int ck = (int)h->levels.size() - 1;
  h->levels[in_key].parent = parent;
  h->levels[in_key].child_key = ck;
  if (out_key) *out_key = ck;
  if (n_out) *n_out = C.n;
  return PCMI_OK;
}

int pcmi_kernel_offsets(int kernel_size, int region, int32_t* out_host, int* K) {
  PCMI_REQUIRE(out_host && K && (kernel_size == 3 || kernel_size == 2 || kernel_size == 1), PCMI_ERR_UNSUPPORTED,
               "kernel_offsets: kernel size %d not on the hot path", kernel_size);
  PCMI_REQUIRE(region == PCMI_REGION_HYPERCUBE || region == PCMI_REGION_HYBRID, PCMI_ERR_UNSUPPORTED,
               "kernel_offsets: region %d not on the hot path", region);
  int32_t o[PCMI_MAX_KERNEL_VOLUME][3];
  fill_offsets(kernel_size, region, o, K);
  memcpy(out_host, o, sizeof(int32_t) * 3 * (*K));
  return PCMI_OK;
}

}  // extern "C"

// nosync: a map the plan built is returned as it is (M == -1 while its counts are on their way); the kernels of this
// library do not need them (pcmi_net_forward uses this form)
static int kmap_get_impl(pcmi_coords_t* h, int in_key, int out_key, int kernel_size, int stride, int region,
                         pcmi_kmap_t* out, pcmi_stream_t stream, bool nosync) {
  PCMI_REQUIRE(h && out, PCMI_ERR_INVALID, "kmap_get: null argument");
  PCMI_REQUIRE(in_key >= 0 && in_key < (int)h->levels.size() && out_key >= 0 && out_key < (int)h->levels.size(),
               PCMI_ERR_NOKEY, "kmap_get: unknown key (%d -> %d)", in_key, out_key);
  PCMI_REQUIRE((kernel_size == 3 && stride == 1) || (kernel_size == 2 && stride == 2), PCMI_ERR_UNSUPPORTED,
               "kmap_get: kernel %d / stride %d is not on the hot path", kernel_size, stride);
  PCMI_REQUIRE(region == PCMI_REGION_HYPERCUBE || region == PCMI_REGION_HYBRID, PCMI_ERR_UNSUPPORTED,
               "kmap_get: region %d not on the hot path", region);
  if (kernel_size == 2) region = PCMI_REGION_HYPERCUBE;  // even kernels enumerate identically
  for (auto& e : h->maps)
    if (e.in_key == in_key && e.out_key == out_key && e.ksize == kernel_size && e.stride == stride &&
        e.region == region) {
      if (e.map.M < 0 && !h->defer_maps && !nosync) {  // built by the plan: the caller wants the host-side counts
        const int rc0 = resolve_pending_maps(h);
        if (rc0) return rc0;
      }
      *out = e.map;
      return PCMI_OK;
    }
  hipStream_t st = as_stream(stream);
  h->scratch.reset();
  const Level& Lin = h->levels[in_key];
  const Level& Lout = h->levels[out_key];
  pcmi_kmap_t m;
  memset(&m, 0, sizeof(m));
  int32_t o[PCMI_MAX_KERNEL_VOLUME][3];
  int K = 0;
  fill_offsets(kernel_size, region, o, &K);
  m.K = K;
  m.kernel_size = kernel_size;
  m.stride = stride;
  m.region = region;
  m.n_in = Lin.n;
  m.n_out = Lout.n;
  for (int k = 0; k < K; ++k) {
    m.mirror[k] = -1;
    for (int k2 = 0; k2 < K; ++k2)
      if (o[k2][0] == -o[k][0] && o[k2][1] == -o[k][1] && o[k2][2] == -o[k][2]) m.mirror[k] = k2;
  }
  const int64_t n_out = Lout.n;
  const int64_t tiles = std::max<int64_t>(ceil_div(n_out, kMapTile), 1);
  const int64_t tot = (int64_t)K * n_out;
  int defer_slot = -1;  // >= 0: the pair counts are on their way to that pinned slot (resolve_pending_maps)
  // pair lists are sized by their a-priori bound so that the only read-back is the K+1 offsets
  const int64_t pair_bound = stride == 1 ? tot : Lin.n;
  int32_t* nbr = h->persistent.alloc_n<int32_t>(tot);
  int64_t* offs = h->persistent.alloc_n<int64_t>(K + 1);
  int32_t* pair_in = h->persistent.alloc_n<int32_t>(pair_bound);
  int32_t* pair_out = h->persistent.alloc_n<int32_t>(pair_bound);
  int32_t* pos = h->scratch.alloc_n<int32_t>(tot);
  if (!nbr || !offs || !pair_in || !pair_out || !pos) return PCMI_ERR_HIP;
  if (stride == 1) {
    PCMI_REQUIRE(in_key == out_key, PCMI_ERR_INVALID, "kmap_get: stride-1 map needs in_key == out_key");
    OffsetTable ot;
    for (int k = 0; k < K; ++k)
      for (int d = 0; d < 3; ++d) ot.o[k][d] = (int8_t)o[k][d];
    if (n_out > 0) {
      kmap_k3_kernel<<<dim3((unsigned)tiles), 256, 0, st>>>(Lout.coords, n_out, Lin.ts, ot, Lin.hkeys,
                                                           Lin.hvals, Lin.cap - 1, nbr);
      PCMI_LAUNCH_CHECK();
    }
  } else {
    PCMI_REQUIRE(Lin.child_key == out_key, PCMI_ERR_INVALID,
                 "kmap_get: key %d is not the stride-2 child of key %d", out_key, in_key);
    PCMI_HIP_CHECK(hipMemsetAsync(nbr, 0xFF, sizeof(int32_t) * std::max<int64_t>(tot, 1), st));
    if (Lin.n > 0) {
      kmap_s2_kernel<<<dim3((unsigned)ceil_div(Lin.n, 256)), 256, 0, st>>>(Lin.coords, Lin.n, Lin.ts,
                                                                          Lin.parent, n_out, nbr);
      PCMI_LAUNCH_CHECK();
    }
  }
  // per-offset pair lists
  int rc = exclusive_scan<true>(nbr, tot, pos, h->d_total, h->scratch, st);
  if (rc) return rc;
  if (tot > 0) {
    kmap_compact_kernel<<<dim3((unsigned)ceil_div(tot, 256)), 256, 0, st>>>(nbr, K, n_out, pos, h->d_total,
                                                                           pair_in, pair_out, offs);
    PCMI_LAUNCH_CHECK();
    // the K + 1 offsets travel to this map's slot of the pinned buffer; a deferred build (pcmi_coords_plan_unet) does
    // not wait for them, any other build synchronises here and fills in every map that is still waiting
    if ((int)h->pending.size() >= kMapSlots) {
      rc = resolve_pending_maps(h);
      if (rc) return rc;
    }
    defer_slot = (int)h->pending.size();
    PCMI_HIP_CHECK(hipMemcpyAsync(h->h_pinned + kMapSlot0 + (size_t)defer_slot * kMapSlotLen, offs, sizeof(int64_t) * (K + 1),
                                  hipMemcpyDeviceToHost, st));
    // (ev_counts is recorded at the END of this build, behind the mask sort: whoever waits for the counts of a map --
    //  pcmi_kmap_get does, before it hands the map out -- has then also waited for every table of it)
  } else {
    PCMI_HIP_CHECK(hipMemsetAsync(offs, 0, sizeof(int64_t) * (K + 1), st));
  }
  m.M = defer_slot >= 0 ? -1 : 0;  // -1: the host-side counts are still on their way (see pcmi_kmap_t)
  m.perm = nullptr;
  m.nbr_perm = nullptr;
  m.tile_mask = nullptr;
  m.tile_pref = nullptr;
  m.n_tiles = 0;
  if (stride == 1 && n_out >= kSortRowsMin) {
    int32_t* perm = h->persistent.alloc_n<int32_t>(n_out);
    int32_t* nbr_perm = h->persistent.alloc_n<int32_t>(tot);
    uint32_t* mk_in = h->scratch.alloc_n<uint32_t>(n_out);
    uint32_t* mk_out = h->scratch.alloc_n<uint32_t>(n_out);
    int32_t* iota = h->scratch.alloc_n<int32_t>(n_out);
    const size_t tb = sort_rows_temp_bytes(n_out);
    void* temp = h->scratch.alloc(tb);
    if (!perm || !nbr_perm || !mk_in || !mk_out || !iota || !temp) return PCMI_ERR_HIP;
    rc = sort_rows_by_mask(nbr, K, n_out, xcd_chunk_rows(n_out), mk_in, mk_out, iota, temp, tb, perm, nbr_perm, st);
    if (rc) return rc;
    m.perm = perm;
    m.nbr_perm = nbr_perm;
    const int64_t n_tiles = ceil_div(n_out, 128);
    uint32_t* tile_mask = h->persistent.alloc_n<uint32_t>(n_tiles);
    int32_t* tile_pref = h->persistent.alloc_n<int32_t>(n_tiles + 1);
    int32_t* tile_cnt = h->scratch.alloc_n<int32_t>(n_tiles);
    if (!tile_mask || !tile_pref || !tile_cnt) return PCMI_ERR_HIP;
    rc = tile_units(mk_out, K, n_out, tile_mask, tile_cnt, tile_pref, st);
    if (rc) return rc;
    m.tile_mask = tile_mask;
    m.tile_pref = tile_pref;
    m.n_tiles = n_tiles;
  }
  m.nbr = nbr;
  m.pair_in = pair_in;
  m.pair_out = pair_out;
  m.offs = offs;
  h->maps.push_back({in_key, out_key, kernel_size, stride, region, m});
  if (defer_slot >= 0) {
    h->pending.push_back({(int)h->maps.size() - 1, defer_slot});
    // Until round 3 the event sat right behind the copy of the counts, IN FRONT of the mask sort: a per-call map was
    // handed out while row_mask / radix sort / permute_table / tile_units were still running on the plan stream, and a
    // convolution launched on the compute stream within ~100 us read perm / nbr_perm / tile_pref half-written (found
    // as a GPU memory fault in 1 of 12 runs of bench.py's stand-alone kernel timings, profiles/r03m_*).
    PCMI_HIP_CHECK(hipEventRecord(h->ev_counts, st));
  }
  if (defer_slot >= 0 && !h->defer_maps) {
    rc = resolve_pending_maps(h);
    if (rc) return rc;
    rc = check_insert_status(h);
    if (rc) return rc;
  }
  *out = h->maps.back().map;  // (deferred build: offs_host / M are not filled in, M == -1)
  return PCMI_OK;
}

namespace pcmi {
int kmap_get_nosync(pcmi_coords_t* h, int in_key, int out_key, int kernel_size, int stride, int region, pcmi_kmap_t* out,
                    pcmi_stream_t stream) {
  return kmap_get_impl(h, in_key, out_key, kernel_size, stride, region, out, stream, true);
}
// makes `stream` wait for the end of the handle's last pcmi_coords_plan_unet (enqueued on another stream, unsynchronised)
int coords_wait_plan(pcmi_coords_t* h, hipStream_t st) {
  if (h && h->plan_recorded) PCMI_HIP_CHECK(hipStreamWaitEvent(st, h->ev_plan, 0));
  return PCMI_OK;
}
}  // namespace pcmi

extern "C" {

int pcmi_kmap_get(pcmi_coords_t* h, int in_key, int out_key, int kernel_size, int stride, int region,
                  pcmi_kmap_t* out, pcmi_stream_t stream) {
  return kmap_get_impl(h, in_key, out_key, kernel_size, stride, region, out, stream, false);
}

int pcmi_kmap_export(const pcmi_kmap_t* map, int32_t* nbr, int32_t* pair_in, int32_t* pair_out,
                     pcmi_stream_t stream) {
  PCMI_REQUIRE(map, PCMI_ERR_INVALID, "kmap_export: null map");
  PCMI_REQUIRE(map->M >= 0, PCMI_ERR_INVALID, "kmap_export: the map's pair counts are not on the host yet (take it from pcmi_kmap_get)");
  hipStream_t st = as_stream(stream);
  const size_t tot = (size_t)map->K * map->n_out;
  if (nbr && tot) PCMI_HIP_CHECK(hipMemcpyAsync(nbr, map->nbr, sizeof(int32_t) * tot, hipMemcpyDeviceToDevice, st));
  if (pair_in && map->M)
    PCMI_HIP_CHECK(hipMemcpyAsync(pair_in, map->pair_in, sizeof(int32_t) * map->M, hipMemcpyDeviceToDevice, st));
  if (pair_out && map->M)
    PCMI_HIP_CHECK(hipMemcpyAsync(pair_out, map->pair_out, sizeof(int32_t) * map->M, hipMemcpyDeviceToDevice, st));
  return PCMI_OK;
}

}  // extern "C"

// Strided levels 1 .. n_down of a fresh handle (key 0 only) WITHOUT a read-back per level: every level is sized by its
// bound (a coarse level has at most as many rows as the level above it, hence at most n0), its kernels take the actual
// count from the device (counts[l], written by the scan of the level above), and the counts of all levels plus the
// insert's status words come back in ONE synchronisation at the end.  Same tables, same row order as pcmi_coords_stride.
static int build_level_chain(pcmi_coords_t* h, int n_down, hipStream_t st) {
  const int64_t n0 = h->levels[0].n;
  chain_init_kernel<<<1, 1, 0, st>>>(h->d_counts, n0, h->levels[0].split);
  PCMI_LAUNCH_CHECK();
  const dim3 grid((unsigned)std::max<int64_t>(ceil_div(n0, 256), 1));
  std::vector<int32_t*> parents;
  for (int l = 0; l < n_down; ++l) {
    h->scratch.reset();
    const Level& F = h->levels[l];
    Level C;
    C.ts = F.ts * 2;
    int rc = new_table(h, C, n0, st);
    if (rc) return rc;
    int32_t* parent = h->persistent.alloc_n<int32_t>(n0);
    C.coords = h->persistent.alloc_n<int32_t>(n0 * 4);
    uint32_t* slot_of = h->scratch.alloc_n<uint32_t>(n0);
    int32_t* flags = h->scratch.alloc_n<int32_t>(n0);
    int32_t* pos = h->scratch.alloc_n<int32_t>(n0);
    if (!parent || !C.coords || !slot_of || !flags || !pos) return PCMI_ERR_HIP;
    const int64_t* n_dev = h->d_counts + l;
    stride_insert_kernel<<<grid, 256, 0, st>>>(F.coords, n0, n_dev, C.ts, C.hkeys, C.hvals, C.cap - 1, slot_of);
    PCMI_LAUNCH_CHECK();
    stride_flag_kernel<<<grid, 256, 0, st>>>(n0, n_dev, C.hvals, slot_of, flags);
    PCMI_LAUNCH_CHECK();
    rc = exclusive_scan<true>(flags, n0, pos, h->d_counts + l + 1, h->scratch, st);
    if (rc) return rc;
    chain_split_kernel<<<1, 1, 0, st>>>(pos, h->d_counts, l);
    PCMI_LAUNCH_CHECK();
    stride_parent_kernel<<<grid, 256, 0, st>>>(F.coords, n0, n_dev, C.ts, C.hvals, slot_of, pos, parent, C.coords);
    PCMI_LAUNCH_CHECK();
    stride_relabel_kernel<<<grid, 256, 0, st>>>(n0, n_dev, flags, slot_of, pos, C.hvals);
    PCMI_LAUNCH_CHECK();
    C.n = -1;  // known after the read-back below
    h->levels.push_back(C);
    parents.push_back(parent);
  }
  PCMI_HIP_CHECK(hipMemcpyAsync(h->h_pinned + kLevelSlot0, h->d_counts, sizeof(int64_t) * 2 * kMaxLevels, hipMemcpyDeviceToHost, st));
  PCMI_HIP_CHECK(hipStreamSynchronize(st));
  int rc = check_insert_status(h);
  if (rc) {  // as the synchronous insert: the handle holds nothing (a table with a bad hash must not stay usable)
    discard_after_bad_insert(h);
    return rc;
  }
  for (int l = 0; l < n_down; ++l) {
    h->levels[l + 1].n = h->h_pinned[kLevelSlot0 + l + 1];
    h->levels[l + 1].split = h->h_pinned[kLevelSlot0 + kMaxLevels + l + 1];
    h->levels[l].parent = parents[l];
    h->levels[l].child_key = l + 1;
  }
  return PCMI_OK;
}

extern "C" {

int pcmi_coords_plan_unet(pcmi_coords_t* h, int n_down, int first_region, int block_region,
                          pcmi_stream_t stream) {
  PCMI_REQUIRE(h && !h->levels.empty(), PCMI_ERR_INVALID, "plan_unet: insert coordinates first");
  PCMI_REQUIRE(n_down >= 0 && n_down < kMaxLevels - 1 && n_down <= 8, PCMI_ERR_INVALID, "plan_unet: bad depth");
  hipStream_t st = as_stream(stream);
  pcmi_kmap_t tmp;
  int rc;
  // ---- levels: one synchronisation for the whole chain (a fresh handle; otherwise level by level, as pcmi_coords_stride) ----
  if (h->levels.size() == 1 && n_down > 0 && h->maps.empty()) {
    rc = build_level_chain(h, n_down, st);
    if (rc) return rc;
  }
  // ---- maps: built on `st` without waiting for their pair counts (see pcmi_coords::defer_maps) -------------------------
  struct Defer {  // also on the error paths: the flag never outlives the call
    pcmi_coords_t* h;
    ~Defer() { h->defer_maps = false; }
  } defer{h};
  h->defer_maps = true;
  int key = 0;
  rc = kmap_get_impl(h, 0, 0, 3, 1, first_region, &tmp, stream, true);
  if (rc) return rc;
  for (int l = 0; l <= n_down; ++l) {
    rc = kmap_get_impl(h, key, key, 3, 1, block_region, &tmp, stream, true);
    if (rc) return rc;
    if (l == n_down) break;
    int ck;
    int64_t n;
    rc = pcmi_coords_stride(h, key, 2, &ck, &n, stream);
    if (rc) return rc;
    rc = kmap_get_impl(h, key, ck, 2, 2, PCMI_REGION_HYPERCUBE, &tmp, stream, true);
    if (rc) return rc;
    key = ck;
  }
  h->defer_maps = false;
  // a deferred insert reports its duplicate / range status through this call (pcmi.h): when no level chain was built
  // (n_down == 0, or levels that already existed) nothing above has synchronised and looked at it yet
  if (h->insert_unchecked) {
    PCMI_HIP_CHECK(hipStreamSynchronize(st));
    rc = check_insert_status(h);
    if (rc) {
      discard_after_bad_insert(h);
      return rc;
    }
  }
  PCMI_HIP_CHECK(hipEventRecord(h->ev_plan, st));
  h->plan_recorded = true;
  return PCMI_OK;
}

}  // extern "C"
