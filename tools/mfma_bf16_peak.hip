// Micro-benchmark: sustained v_mfma_f32_32x32x16_bf16 rate on the GPU it runs on, operands in registers, random bf16 data of the
// magnitude the float matcher sees (the clock the chip holds depends on the data): the ceiling of the L2 GEMM (csrc/l2.hip), whose
// 32 rows x 32 queries x 128 dimensions are a chain of 8 such MFMAs.
//   hipcc -O3 --offload-arch=gfx950 -o tools/mfma_bf16_peak tools/mfma_bf16_peak.hip
#include <hip/hip_runtime.h>
#include <cstdio>

typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

__device__ __forceinline__ unsigned mix(unsigned x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
// a bf16 in [-1, 1): sign and mantissa random, exponent 2^-1 .. 2^-8
__device__ __forceinline__ short rnd_bf16(unsigned h) { return (short)(((h & 1u) << 15) | ((119u + ((h >> 1) & 7u)) << 7) | ((h >> 4) & 0x7fu)); }

template <int CHAINS>
__global__ __launch_bounds__(256) void mfma_kernel(float* out, int seed, int zeros, int iters) {
  bf16x8 a[8], b[8];
#pragma unroll
  for (int s = 0; s < 8; ++s)
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      a[s][i] = zeros ? (short)0 : rnd_bf16(mix(threadIdx.x * 131u + s * 17u + i + seed));
      b[s][i] = zeros ? (short)0 : rnd_bf16(mix(threadIdx.x * 977u + s * 29u + i * 7u + blockIdx.x));
    }
  f32x16 acc[CHAINS];
  for (int c = 0; c < CHAINS; ++c) for (int i = 0; i < 16; ++i) acc[c][i] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int c = 0; c < CHAINS; ++c)
#pragma unroll
      for (int s = 0; s < 8; ++s) acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[s], b[s], acc[c], 0, 0, 0);
  }
  float r = 0.f;
  for (int c = 0; c < CHAINS; ++c) for (int i = 0; i < 16; ++i) r += acc[c][i];
  out[blockIdx.x * 256 + threadIdx.x] = r;
}

template <int CHAINS>
void run(int blocks_per_cu, int n_cu, float* d_out, int zeros, int iters) {
  const int grid = n_cu * blocks_per_cu;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(mfma_kernel<CHAINS>, dim3(grid), dim3(256), 0, 0, d_out, 1, zeros, iters);
  hipDeviceSynchronize();
  // launches back to back inside one timed region (an idle gap lets the clock fall, and the next launch pays for the ramp)
  const int reps = iters >= 4096 ? 8 : 128;
  float best = 1e30f, sum = 0.f;
  for (int r = 0; r < 8; ++r) {
    hipEventRecord(e0, 0);
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(mfma_kernel<CHAINS>, dim3(grid), dim3(256), 0, 0, d_out, r + 2, zeros, iters);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps; if (ms < best) best = ms; sum += ms;
  }
  const double mfmas = (double)grid * 4.0 * iters * 8.0 * CHAINS;
  const double flops = mfmas * 32.0 * 32.0 * 16.0 * 2.0;
  printf("%s %5d iterations chains=%d waves/SIMD=%d: best %.3f ms (mean %.3f), %.3f PFLOP/s best (%.3f mean), %.1f cycles per MFMA per SIMD @2.4GHz\n", zeros ? "zeros " : "random", iters, CHAINS, blocks_per_cu,
         best, sum / 8, flops / (best * 1e-3) / 1e15, flops / (sum / 8 * 1e-3) / 1e15, (best * 1e-3 * 2.4e9) / (mfmas / (n_cu * 4.0)));
}

int main() {
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  const int n_cu = p.multiProcessorCount;
  float* d_out; hipMalloc(&d_out, (size_t)n_cu * 8 * 256 * sizeof(float));
  // all-zero operands toggle nothing: the rate the matrix cores reach when power is not what limits the clock
  // and short launches (the matcher's GEMM pass is ~0.1 ms) next to long ones: the clock a burst gets is not the sustained one
  for (int zeros : {0, 1})
    for (int iters : {256, 8192})
      for (int bpc : {1, 2}) { run<1>(bpc, n_cu, d_out, zeros, iters); run<2>(bpc, n_cu, d_out, zeros, iters); }
  return 0;
}
