// Attainable fp32 matrix-core rate on this GPU, measured the way the conv kernels use the pipe (diagnostic, not product):
//   mode 0: v_mfma_f32_16x16x4_f32 only, 12 independent accumulators per wave, operands in registers
//   mode 1: + one ds_read_b32 B fragment per 2 MFMAs from a padded LDS tile (the ratio of spconv16_kernel<3>)
//   mode 2: mode 1 + a workgroup barrier every 96 MFMAs (one K-chunk step of the conv)
//   mode 3: v_mfma_f32_32x32x2_f32 only, 6 independent accumulators
// at 1, 2 and 4 waves per SIMD, for a short (~0.3 ms) and a long (~5 ms) launch.  The guide's 157.3 TFLOP/s assumes
// 256 CUs x 4 SIMDs x 64 FLOP/cycle x 2.4 GHz; mode 0 tells which clock the part actually sustains under this load.
//   hipcc --offload-arch=gfx950 -O3 scripts/mfma_peak.hip -o scripts/_build/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int MODE, int WPS>
__global__ __launch_bounds__(256, WPS) void peak_kernel(float* out, int iters, float seed) {
  constexpr int LDB = 100;
  __shared__ float s_b[32 * LDB];
  const int t = threadIdx.x, lane = t & 63;
  for (int p = t; p < 32 * LDB; p += 256) s_b[p] = seed * (float)(p & 15);
  __syncthreads();
  float res = 0.f;
  if constexpr (MODE == 3) {
    f32x16 acc[6];
    for (int u = 0; u < 6; ++u)
      for (int j = 0; j < 16; ++j) acc[u][j] = 0.f;
    float av = seed * lane, bv = seed + lane;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int u = 0; u < 6; ++u) acc[u] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[u], 0, 0, 0);
    }
    for (int u = 0; u < 6; ++u)
      for (int j = 0; j < 16; ++j) res += acc[u][j];
  } else {
    f32x4 acc[12];
    for (int u = 0; u < 12; ++u) acc[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float av0 = seed * lane, av1 = seed - lane;
    const int i = lane & 15, kk = lane >> 4;
    for (int it = 0; it < iters; ++it) {
      const float* sb = s_b + i + (4 * kk) * LDB + ((it & 3) * LDB);
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        float bf[6];
        if constexpr (MODE == 0) {
#pragma unroll
          for (int ct = 0; ct < 6; ++ct) bf[ct] = av1 + (float)ct;
        } else {
#pragma unroll
          for (int ct = 0; ct < 6; ++ct) bf[ct] = sb[(16 * (q >> 2) + (q & 3)) * LDB + ct * 16];
        }
#pragma unroll
        for (int ct = 0; ct < 6; ++ct) acc[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(av0, bf[ct], acc[ct], 0, 0, 0);
#pragma unroll
        for (int ct = 0; ct < 6; ++ct) acc[6 + ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(av1, bf[ct], acc[6 + ct], 0, 0, 0);
      }
      if constexpr (MODE == 2) __syncthreads();
    }
    for (int u = 0; u < 12; ++u)
      for (int j = 0; j < 4; ++j) res += acc[u][j];
  }
  if (res == 12345.678f) out[blockIdx.x * 256 + t] = res;
}

template <int MODE, int WPS>
static void run(const char* label, float* out, int cus) {
  // MFMA cycles per iteration per wave: 96 x 32 (16x16x4) or 24 x 64 (32x32x2) -- 3072 / 1536
  const double flop_it = MODE == 3 ? 24.0 * 32 * 32 * 2 * 2 : 96.0 * 16 * 16 * 4 * 2;
  const int blocks = cus * WPS;
  for (int len = 0; len < 2; ++len) {
    const int per_wave = (len == 0 ? 240 : 4000) / WPS * (MODE == 3 ? 2 : 1);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    float best = 1e30f, sum = 0.f;
    const int reps = len == 0 ? 20 : 5;
    for (int r = -2; r < reps; ++r) {
      CK(hipEventRecord(e0, 0));
      peak_kernel<MODE, WPS><<<blocks, 256>>>(out, per_wave, 1e-3f);
      CK(hipEventRecord(e1, 0));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      if (r >= 0) { best = ms < best ? ms : best; sum += ms; }
    }
    const double flops = flop_it * per_wave * 4.0 * blocks;
    printf("%-44s waves/SIMD %d  %s launch: avg %.3f ms  %.1f TFLOP/s (best %.1f) = %.3f of 157.3\n", label, WPS, len == 0 ? "short" : "long ",
           sum / reps, flops / (sum / reps) * 1e-9, flops / best * 1e-9, flops / (sum / reps) * 1e-9 / 157.3);
  }
}

int main() {
  hipDeviceProp_t p;
  CK(hipGetDeviceProperties(&p, 0));
  printf("%s  CUs %d  clockRate %d kHz\n", p.name, p.multiProcessorCount, p.clockRate);
  float* out;
  CK(hipMalloc(&out, 256 * 1024 * 4 * sizeof(float)));
  const int cus = p.multiProcessorCount;
  run<0, 1>("16x16x4 regs only", out, cus);
  run<0, 2>("16x16x4 regs only", out, cus);
  run<0, 4>("16x16x4 regs only", out, cus);
  run<1, 1>("16x16x4 + ds_read_b32 per 2 MFMA", out, cus);
  run<1, 2>("16x16x4 + ds_read_b32 per 2 MFMA", out, cus);
  run<1, 4>("16x16x4 + ds_read_b32 per 2 MFMA", out, cus);
  run<2, 1>("16x16x4 + ds_read + barrier / 96 MFMA", out, cus);
  run<2, 2>("16x16x4 + ds_read + barrier / 96 MFMA", out, cus);
  run<2, 4>("16x16x4 + ds_read + barrier / 96 MFMA", out, cus);
  run<3, 1>("32x32x2 regs only", out, cus);
  run<3, 2>("32x32x2 regs only", out, cus);
  run<3, 4>("32x32x2 regs only", out, cus);
  return 0;
}
