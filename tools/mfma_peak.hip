// Micro-benchmark: sustained v_mfma_f32_32x32x16_bf16 rate on the GPU it runs on, for the accumulation patterns the
// float-descriptor matcher can choose from (one dependent chain, 2 or 4 interleaved chains; 1, 2 or 4 waves per SIMD).
//   hipcc -O3 --offload-arch=gfx950 -o tools/mfma_peak tools/mfma_peak.hip
#include <hip/hip_runtime.h>
#include <cstdio>

typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
constexpr int kIters = 2048;

template <int CHAINS>
__global__ __launch_bounds__(256) void mfma_kernel(float* out, int seed) {
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (short)(0x3f80 + ((threadIdx.x + i + seed) & 7)); b[i] = (short)(0x3f80 + ((threadIdx.x * 3 + i) & 7)); }
  f32x16 acc[CHAINS];
  for (int c = 0; c < CHAINS; ++c) for (int i = 0; i < 16; ++i) acc[c][i] = 0.f;
  for (int it = 0; it < kIters; ++it) {
#pragma unroll
    for (int s = 0; s < 8 / CHAINS * CHAINS; ++s) acc[s % CHAINS] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[s % CHAINS], 0, 0, 0);
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
  float best = 1e30f;
  for (int r = 0; r < 5; ++r) {
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(mfma_kernel<CHAINS>, dim3(grid), dim3(256), 0, 0, d_out, r + 2);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
  }
  const double mfmas = (double)grid * 4.0 * kIters * (8 / CHAINS * CHAINS);
  const double flops = mfmas * 32.0 * 32.0 * 16.0 * 2.0;
  printf("chains=%d waves/SIMD=%d: %.3f ms, %.2f PFLOP/s, %.1f cycles per MFMA per SIMD @2.4GHz\n", CHAINS, blocks_per_cu, best,
         flops / (best * 1e-3) / 1e15, (best * 1e-3 * 2.4e9) / (mfmas / (n_cu * 4.0)));
}

int main() {
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  const int n_cu = p.multiProcessorCount;
  float* d_out; hipMalloc(&d_out, (size_t)n_cu * 8 * 256 * sizeof(float));
  for (int bpc : {1, 2, 4}) { run<1>(bpc, n_cu, d_out); run<2>(bpc, n_cu, d_out); run<4>(bpc, n_cu, d_out); }
  return 0;
}
