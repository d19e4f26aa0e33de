// Micro-benchmark: sustained issue rate of the integer VALU ops the Hamming matcher is made of
// (v_xor_b32 with an SGPR operand, accumulating v_bcnt_u32_b32), measured on the GPU it runs on.
// Gives the "known-good reference on the same hardware" that the VALU roofline in DESIGN.md uses.
//   hipcc -O3 --offload-arch=gfx950 -o valu_peak tools/valu_peak.hip && ./valu_peak
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %d at %d\n", (int)e_, __LINE__); return 1; } } while (0)

constexpr int kIters = 4096;

template <int MODE>
__global__ __launch_bounds__(256) void valu_kernel(uint32_t* out, uint32_t seed) {
  uint32_t a0 = threadIdx.x + seed, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 9, a5 = a0 * 11, a6 = a0 * 13,
           a7 = a0 * 15;
  uint32_t s = __builtin_amdgcn_readfirstlane(seed * 2654435761u);
  for (int i = 0; i < kIters; ++i) {
    if (MODE == 0) {          // 8 independent v_xor_b32 (SGPR operand)
      asm volatile("v_xor_b32 %0, %8, %0\n v_xor_b32 %1, %8, %1\n v_xor_b32 %2, %8, %2\n v_xor_b32 %3, %8, %3\n"
                   "v_xor_b32 %4, %8, %4\n v_xor_b32 %5, %8, %5\n v_xor_b32 %6, %8, %6\n v_xor_b32 %7, %8, %7\n"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "s"(s));
    } else if (MODE == 1) {   // 8 independent accumulating v_bcnt_u32_b32
      asm volatile("v_bcnt_u32_b32 %0, %0, %1\n v_bcnt_u32_b32 %1, %1, %2\n v_bcnt_u32_b32 %2, %2, %3\n"
                   "v_bcnt_u32_b32 %3, %3, %4\n v_bcnt_u32_b32 %4, %4, %5\n v_bcnt_u32_b32 %5, %5, %6\n"
                   "v_bcnt_u32_b32 %6, %6, %7\n v_bcnt_u32_b32 %7, %7, %0\n"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    } else if (MODE == 2) {   // the matcher's pattern: xor -> dependent accumulating bcnt, 4 chains interleaved
      asm volatile("v_xor_b32 %4, %8, %4\n v_xor_b32 %5, %8, %5\n v_xor_b32 %6, %8, %6\n v_xor_b32 %7, %8, %7\n"
                   "v_bcnt_u32_b32 %0, %4, %0\n v_bcnt_u32_b32 %1, %5, %1\n v_bcnt_u32_b32 %2, %6, %2\n"
                   "v_bcnt_u32_b32 %3, %7, %3\n"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "s"(s));
    } else if (MODE == 3) {   // 8 independent v_add_u32 (reference integer op)
      asm volatile("v_add_u32 %0, %8, %0\n v_add_u32 %1, %8, %1\n v_add_u32 %2, %8, %2\n v_add_u32 %3, %8, %3\n"
                   "v_add_u32 %4, %8, %4\n v_add_u32 %5, %8, %5\n v_add_u32 %6, %8, %6\n v_add_u32 %7, %8, %7\n"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "s"(s));
    } else if (MODE == 4) {   // 8 independent v_fma_f32 (the 157 TF fp32 vector peak is quoted on this op family)
      asm volatile("v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %1, %1, %2, %3\n v_fma_f32 %2, %2, %3, %4\n"
                   "v_fma_f32 %3, %3, %4, %5\n v_fma_f32 %4, %4, %5, %6\n v_fma_f32 %5, %5, %6, %7\n"
                   "v_fma_f32 %6, %6, %7, %0\n v_fma_f32 %7, %7, %0, %1\n"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    } else if (MODE == 5) {   // 4 v_pk_fma_f32 (packed: 2 fp32 lanes-ops per lane)
      asm volatile("v_pk_fma_f32 %0, %0, %1, %2\n v_pk_fma_f32 %1, %1, %2, %3\n v_pk_fma_f32 %2, %2, %3, %0\n"
                   "v_pk_fma_f32 %3, %3, %0, %1\n"
                   : "+v"(*(double*)&a0), "+v"(*(double*)&a2), "+v"(*(double*)&a4), "+v"(*(double*)&a6));
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}

template <int MODE>
int run(const char* name, int ops_per_iter, int blocks_per_cu, int n_cu, uint32_t* d_out) {
  const int grid = n_cu * blocks_per_cu;
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL(valu_kernel<MODE>, dim3(grid), dim3(256), 0, 0, d_out, 1u);
  CHECK(hipDeviceSynchronize());
  float best = 1e30f;
  for (int r = 0; r < 5; ++r) {
    CHECK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(valu_kernel<MODE>, dim3(grid), dim3(256), 0, 0, d_out, (uint32_t)r + 2u);
    CHECK(hipEventRecord(e1, 0));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  const double wave_instr = (double)grid * 4.0 * kIters * ops_per_iter;
  const double lane_ops = wave_instr * 64.0;
  const double per_simd_cycle = wave_instr / (n_cu * 4.0) / (best * 1e-3 * 2.4e9);
  printf("%-34s blocks/CU=%d  %8.3f ms  %7.2f T lane-op/s  %.3f wave-instr/cycle/SIMD @2.4GHz (=> %.2f cycles per wave64 instr)\n",
         name, blocks_per_cu, best, lane_ops / (best * 1e-3) / 1e12, per_simd_cycle, 1.0 / per_simd_cycle);
  return 0;
}

int main() {
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int n_cu = prop.multiProcessorCount;
  printf("device %s, %d CUs, clock %d kHz\n", prop.name, n_cu, prop.clockRate);
  uint32_t* d_out;
  CHECK(hipMalloc(&d_out, (size_t)n_cu * 8 * 256 * 4));
  for (int b : {1, 2, 4, 8}) {
    if (run<0>("v_xor_b32 (sgpr operand) x8 indep", 8, b, n_cu, d_out)) return 1;
    if (run<1>("v_bcnt_u32_b32 acc x8 indep", 8, b, n_cu, d_out)) return 1;
    if (run<2>("xor->bcnt, 4 chains (matcher mix)", 8, b, n_cu, d_out)) return 1;
    if (run<3>("v_add_u32 x8 indep", 8, b, n_cu, d_out)) return 1;
    if (run<4>("v_fma_f32 x8", 8, b, n_cu, d_out)) return 1;
    if (run<5>("v_pk_fma_f32 x4 (2 fp32/lane)", 4, b, n_cu, d_out)) return 1;
  }
  CHECK(hipFree(d_out));
  return 0;
}
