// Micro-benchmark: how much of the K4x matcher's per-block vector work (the 16-way maximum of the previous block's accumulators, the
// compare + branch, the fp4 expansion of the next rows) hides behind its v_mfma_f32_32x32x64_f8f6f4 chain, two waves per SIMD as in
// the product kernel. A "block" is NM dependent MFMAs into one of two alternating accumulators; the test of a block runs after the
// next block's first MFMA was issued (the product's software pipeline). Prints ns per block per SIMD for every variant; the sum
// model (MFMA-only time + VALU-only time) and the overlap model (the larger of the two) bracket what the hardware does.
//   hipcc -O3 --offload-arch=gfx950 -o tools/mfma_valu_overlap tools/mfma_valu_overlap.hip
#include <hip/hip_runtime.h>
#include <cstdio>

typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((ext_vector_type(16))) float f32x16;
constexpr int kIters = 8192;   // trips of two blocks

__device__ __forceinline__ unsigned mix(unsigned x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

template <int NM>
__device__ __forceinline__ f32x16 block(const i32x8 (&a)[4], const i32x8 (&b)[4]) {
  f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int s = 0; s < NM; ++s) acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[s], b[s], acc, 4, 4, 0, 0, 0, 0);
  return acc;
}

// TEST 1: the product's chain (max, 7 max3); 2: a tree (5 independent max3, then 2 max3 + max); 3: chain on 8 of the 16 registers
template <int TEST>
__device__ __forceinline__ bool test(const f32x16& acc, float thr) {
  if (TEST == 1) {
    int m = max(max(__float_as_int(acc[0]), __float_as_int(acc[1])), __float_as_int(acc[2]));
#pragma unroll
    for (int i = 3; i < 15; i += 2) m = max(max(m, __float_as_int(acc[i])), __float_as_int(acc[i + 1]));
    m = max(m, __float_as_int(acc[15]));
    return m > __float_as_int(thr);
  } else if (TEST == 2) {
    int g[5];
#pragma unroll
    for (int j = 0; j < 5; ++j) g[j] = max(max(__float_as_int(acc[3 * j]), __float_as_int(acc[3 * j + 1])), __float_as_int(acc[3 * j + 2]));
    const int u = max(max(g[0], g[1]), g[2]), v = max(max(g[3], g[4]), __float_as_int(acc[15]));
    return max(u, v) > __float_as_int(thr);
  } else {
    int m = max(max(__float_as_int(acc[0]), __float_as_int(acc[1])), __float_as_int(acc[2]));
#pragma unroll
    for (int i = 3; i < 7; i += 2) m = max(max(m, __float_as_int(acc[i])), __float_as_int(acc[i + 1]));
    m = max(m, __float_as_int(acc[7]));
    return m > __float_as_int(thr);
  }
}

template <int NM, int TEST, int XV>
__global__ __launch_bounds__(256, 2) void k(float* out, int seed, float thr, unsigned* sink) {
  i32x8 a[4], b[4];
#pragma unroll
  for (int s = 0; s < 4; ++s)
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      a[s][i] = i < 4 ? (int)((mix(threadIdx.x * 131u + s * 17u + i + seed) & 0x88888888u) | 0x22222222u) : 0;
      b[s][i] = i < 4 ? (int)((mix(threadIdx.x * 977u + s * 29u + i * 7u + blockIdx.x) & 0x88888888u) | 0x22222222u) : 0;
    }
  f32x16 acc_e, acc_o;
#pragma unroll
  for (int i = 0; i < 16; ++i) { acc_e[i] = -4096.f; acc_o[i] = -4096.f; }
  unsigned x = mix(threadIdx.x + seed), hits = 0;
  const unsigned sign = 0x88888888u, one = 0x22222222u;
  for (int it = 0; it < kIters; ++it) {
    if (NM) acc_e = block<NM ? NM : 1>(a, b);
    if (XV) {                                             // 4 independent shift + and_or pairs: one word's fp4 expansion
      a[0][0] = (int)(((x << 3) & sign) | one); a[0][1] = (int)(((x << 2) & sign) | one);
      a[0][2] = (int)(((x << 1) & sign) | one); a[0][3] = (int)((x & sign) | one);
      x += 0x9e3779b9u;
    }
    if (TEST) { if (__builtin_amdgcn_ballot_w64(test<TEST ? TEST : 1>(acc_o, thr)) != 0ull) { ++hits; thr += 1.f; asm volatile("global_store_dword %0, %1, off" :: "v"(sink + (threadIdx.x & 63u)), "v"(hits) : "memory"); } }
    if (NM) acc_o = block<NM ? NM : 1>(a, b);
    if (XV) {
      b[0][0] = (int)(((x << 3) & sign) | one); b[0][1] = (int)(((x << 2) & sign) | one);
      b[0][2] = (int)(((x << 1) & sign) | one); b[0][3] = (int)((x & sign) | one);
      x += 0x9e3779b9u;
    }
    if (TEST) { if (__builtin_amdgcn_ballot_w64(test<TEST ? TEST : 1>(acc_e, thr)) != 0ull) { ++hits; thr += 1.f; asm volatile("global_store_dword %0, %1, off" :: "v"(sink + (threadIdx.x & 63u)), "v"(hits) : "memory"); } }
    if (!NM) {                                            // keep the tested registers moving without the matrix pipe
#pragma unroll
      for (int i = 0; i < 16; i += 8) { acc_e[i] = __int_as_float(__float_as_int(acc_e[i]) ^ (int)(x & 1u)); acc_o[i] = __int_as_float(__float_as_int(acc_o[i]) ^ (int)(x & 1u)); }
    }
  }
  float r = 0.f;
  for (int i = 0; i < 16; ++i) r += acc_e[i] + acc_o[i];
  out[blockIdx.x * 256 + threadIdx.x] = r + (float)a[0][0] + (float)b[1][2];
  if (hits) atomicAdd(sink, hits);
}

// the same 1024 pairs x 128 bit positions as four 16 x 16 x 128 tiles: one MFMA each, no accumulator input, 4 result registers
typedef __attribute__((ext_vector_type(4))) float f32x4;
template <int XV>
__global__ __launch_bounds__(256, 2) void k16(float* out, int seed, float thr, unsigned* sink) {
  i32x8 a[4], b[4];
#pragma unroll
  for (int s = 0; s < 4; ++s)
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      a[s][i] = i < 4 ? (int)((mix(threadIdx.x * 131u + s * 17u + i + seed) & 0x88888888u) | 0x22222222u) : 0;
      b[s][i] = i < 4 ? (int)((mix(threadIdx.x * 977u + s * 29u + i * 7u + blockIdx.x) & 0x88888888u) | 0x22222222u) : 0;
    }
  f32x4 e[4], o[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) for (int i = 0; i < 4; ++i) { e[j][i] = -4096.f; o[j][i] = -4096.f; }
  unsigned x = mix(threadIdx.x + seed), hits = 0;
  const unsigned sign = 0x88888888u, one = 0x22222222u;
  const f32x4 z = {0.f, 0.f, 0.f, 0.f};
  auto test16 = [&](const f32x4 (&v)[4]) {
    int m = max(max(__float_as_int(v[0][0]), __float_as_int(v[0][1])), __float_as_int(v[0][2]));
    m = max(max(m, __float_as_int(v[0][3])), __float_as_int(v[1][0]));
    m = max(max(m, __float_as_int(v[1][1])), __float_as_int(v[1][2]));
    m = max(max(m, __float_as_int(v[1][3])), __float_as_int(v[2][0]));
    m = max(max(m, __float_as_int(v[2][1])), __float_as_int(v[2][2]));
    m = max(max(m, __float_as_int(v[2][3])), __float_as_int(v[3][0]));
    m = max(max(m, __float_as_int(v[3][1])), __float_as_int(v[3][2]));
    m = max(m, __float_as_int(v[3][3]));
    return m > __float_as_int(thr);
  };
  for (int it = 0; it < kIters; ++it) {
#pragma unroll
    for (int j = 0; j < 4; ++j) e[j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a[j], b[j], z, 4, 4, 0, 0, 0, 0);
    if (XV) {
      a[0][0] = (int)(((x << 3) & sign) | one); a[0][1] = (int)(((x << 2) & sign) | one);
      a[0][2] = (int)(((x << 1) & sign) | one); a[0][3] = (int)((x & sign) | one);
      x += 0x9e3779b9u;
    }
    if (__builtin_amdgcn_ballot_w64(test16(o)) != 0ull) { ++hits; thr += 1.f; asm volatile("global_store_dword %0, %1, off" :: "v"(sink + (threadIdx.x & 63u)), "v"(hits) : "memory"); }
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a[j], b[j], z, 4, 4, 0, 0, 0, 0);
    if (XV) {
      b[0][0] = (int)(((x << 3) & sign) | one); b[0][1] = (int)(((x << 2) & sign) | one);
      b[0][2] = (int)(((x << 1) & sign) | one); b[0][3] = (int)((x & sign) | one);
      x += 0x9e3779b9u;
    }
    if (__builtin_amdgcn_ballot_w64(test16(e)) != 0ull) { ++hits; thr += 1.f; asm volatile("global_store_dword %0, %1, off" :: "v"(sink + (threadIdx.x & 63u)), "v"(hits) : "memory"); }
  }
  float r = 0.f;
  for (int j = 0; j < 4; ++j) for (int i = 0; i < 4; ++i) r += e[j][i] + o[j][i];
  out[blockIdx.x * 256 + threadIdx.x] = r + (float)a[0][0] + (float)b[1][2];
  if (hits) atomicAdd(sink, hits);
}

template <int XV>
void run16(const char* what, int n_cu, float* d_out, unsigned* d_sink) {
  const int grid = n_cu * 2;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL((k16<XV>), dim3(grid), dim3(256), 0, 0, d_out, 1, 1e30f, d_sink);
  hipDeviceSynchronize();
  float best = 1e30f;
  for (int r = 0; r < 6; ++r) {
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((k16<XV>), dim3(grid), dim3(256), 0, 0, d_out, r + 2, 1e30f, d_sink);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
  }
  const double ns = best * 1e6 / (2.0 * kIters * 2.0);
  printf("%-58s %.3f ms  %6.1f ns per block per SIMD (%5.1f cycles @2.1 GHz)\n", what, best, ns, ns * 2.1);
}

template <int NM, int TEST, int XV>
double run(const char* what, int n_cu, float* d_out, unsigned* d_sink) {
  const int grid = n_cu * 2;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL((k<NM, TEST, XV>), dim3(grid), dim3(256), 0, 0, d_out, 1, 1e30f, d_sink);
  hipDeviceSynchronize();
  float best = 1e30f;
  for (int r = 0; r < 6; ++r) {
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((k<NM, TEST, XV>), dim3(grid), dim3(256), 0, 0, d_out, r + 2, 1e30f, d_sink);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
  }
  const double blocks_per_simd = 2.0 * kIters * 2.0;       // two waves per SIMD, two blocks per trip
  const double ns = best * 1e6 / blocks_per_simd;
  printf("%-58s %.3f ms  %6.1f ns per block per SIMD (%5.1f cycles @2.1 GHz)\n", what, best, ns, ns * 2.1);
  return ns;
}

int main() {
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  const int n_cu = p.multiProcessorCount;
  float* d_out; hipMalloc(&d_out, (size_t)n_cu * 2 * 256 * sizeof(float));
  unsigned* d_sink; (void)hipMalloc(&d_sink, 1024); (void)hipMemset(d_sink, 0, 1024);
  run<2, 0, 0>("2 MFMAs per block, nothing else", n_cu, d_out, d_sink);
  run<0, 1, 0>("chain test only", n_cu, d_out, d_sink);
  run<0, 1, 4>("chain test + expansion only", n_cu, d_out, d_sink);
  run<2, 1, 0>("2 MFMAs + chain test", n_cu, d_out, d_sink);
  run<2, 2, 0>("2 MFMAs + tree test", n_cu, d_out, d_sink);
  run<2, 3, 0>("2 MFMAs + chain over 8 registers", n_cu, d_out, d_sink);
  run<2, 1, 4>("2 MFMAs + chain test + expansion (the product's half)", n_cu, d_out, d_sink);
  run<2, 2, 4>("2 MFMAs + tree test + expansion", n_cu, d_out, d_sink);
  run16<0>("4 x (16x16x128, C = 0) + chain test", n_cu, d_out, d_sink);
  run16<4>("4 x (16x16x128, C = 0) + chain test + expansion", n_cu, d_out, d_sink);
  run<3, 0, 0>("3 MFMAs per block, nothing else", n_cu, d_out, d_sink);
  run<3, 1, 4>("3 MFMAs + chain test + expansion", n_cu, d_out, d_sink);
  run<4, 0, 0>("4 MFMAs per block, nothing else", n_cu, d_out, d_sink);
  run<4, 1, 4>("4 MFMAs + chain test + expansion (the product's whole)", n_cu, d_out, d_sink);
  run<4, 2, 4>("4 MFMAs + tree test + expansion", n_cu, d_out, d_sink);
  return 0;
}
