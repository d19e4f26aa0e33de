// Micro-benchmark: sustained v_mfma_f32_32x32x64_f8f6f4 (fp4 x fp4, unit scales) rate on the GPU it runs on, operands in
// registers, random +-1.0 nibbles (the clock the chip holds depends on the data): the ceiling of the K4x matcher, whose
// 256-bit Hamming distance of 32 x 32 pairs is a chain of 4 such MFMAs.
//   hipcc -O3 --offload-arch=gfx950 -o tools/mfma_fp4_peak tools/mfma_fp4_peak.hip
#include <hip/hip_runtime.h>
#include <cstdio>

typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((ext_vector_type(16))) float f32x16;
constexpr int kIters = 8192;

__device__ __forceinline__ unsigned mix(unsigned x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

template <int CHAINS>
__global__ __launch_bounds__(256) void mfma_kernel(float* out, int seed) {
  i32x8 a[4], b[4];
#pragma unroll
  for (int s = 0; s < 4; ++s)
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      a[s][i] = i < 4 ? (int)((mix(threadIdx.x * 131u + s * 17u + i + seed) & 0x88888888u) | 0x22222222u) : 0;
      b[s][i] = i < 4 ? (int)((mix(threadIdx.x * 977u + s * 29u + i * 7u + blockIdx.x) & 0x88888888u) | 0x22222222u) : 0;
    }
  f32x16 acc[CHAINS];
  for (int c = 0; c < CHAINS; ++c) for (int i = 0; i < 16; ++i) acc[c][i] = 0.f;
  for (int it = 0; it < kIters; ++it) {
#pragma unroll
    for (int c = 0; c < CHAINS; ++c)
#pragma unroll
      for (int s = 0; s < 4; ++s) acc[c] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[s], b[s], acc[c], 4, 4, 0, 0, 0, 0);
  }
  float r = 0.f;
  for (int c = 0; c < CHAINS; ++c) for (int i = 0; i < 16; ++i) r += acc[c][i];
  out[blockIdx.x * 256 + threadIdx.x] = r;
}

template <int CHAINS>
void run(int blocks_per_cu, int n_cu, float* d_out) {
  const int grid = n_cu * blocks_per_cu;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(mfma_kernel<CHAINS>, dim3(grid), dim3(256), 0, 0, d_out, 1);
  hipDeviceSynchronize();
  float best = 1e30f, sum = 0.f;
  for (int r = 0; r < 8; ++r) {
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(mfma_kernel<CHAINS>, dim3(grid), dim3(256), 0, 0, d_out, r + 2);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms; sum += ms;
  }
  const double mfmas = (double)grid * 4.0 * kIters * 4.0 * CHAINS;
  const double flops = mfmas * 32.0 * 32.0 * 64.0 * 2.0;
  printf("chains=%d waves/SIMD=%d: best %.3f ms (mean %.3f), %.2f PFLOP/s, %.2f T 256-bit pairs/s, %.1f cycles per MFMA per SIMD @2.4GHz\n", CHAINS,
         blocks_per_cu, best, sum / 8, flops / (best * 1e-3) / 1e15, mfmas / 4.0 * 1024.0 / (best * 1e-3) / 1e12,
         (best * 1e-3 * 2.4e9) / (mfmas / (n_cu * 4.0)));
}

int main() {
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  const int n_cu = p.multiProcessorCount;
  float* d_out; hipMalloc(&d_out, (size_t)n_cu * 8 * 256 * sizeof(float));
  for (int bpc : {1, 2}) { run<1>(bpc, n_cu, d_out); run<2>(bpc, n_cu, d_out); }
  return 0;
}
