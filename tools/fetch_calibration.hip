// FETCH_SIZE / WRITE_SIZE calibration on known byte counts, in this repo's access patterns (MI355X_MICROARCH.md, HBM: "other access
// widths are uncalibrated: calibrate on a known byte count in your own access pattern before trusting an absolute"):
//   copy16    16 B per lane coalesced global_load_dwordx4 -> global_store_dwordx4   (K4x's DB rows: one uint4 per lane)
//   read16    16 B per lane loads, reduced to one word per wave                      (reads only)
//   read4     4 B per lane loads
//   lds16     global_load_lds_dwordx4 into an LDS slot (the float matcher's tile DMA, l2.hip), reads only
//   sread     s_load_dwordx16 by every wave (K4's DB rows)
// Every kernel streams SIZE bytes (default 2 GiB: eight times the Infinity Cache, so nothing is served on-die) exactly once.
// build: hipcc -O3 --offload-arch=gfx950 -o tools/fetch_calibration tools/fetch_calibration.hip;  run under
// rocprofv3 --pmc FETCH_SIZE (and, in a run of its own, WRITE_SIZE): tools/fetch_calibration.sh
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <string>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ __launch_bounds__(256) void copy16(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) dst[i] = src[i];
}
__global__ __launch_bounds__(256) void read16(const uint4* __restrict__ src, uint32_t* __restrict__ out, size_t n) {
  uint32_t acc = 0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) { const uint4 v = src[i]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
  if (acc == 0x12345678u) out[0] = acc;                       // (never true for the fill pattern: no stores)
}
__global__ __launch_bounds__(256) void read4(const uint32_t* __restrict__ src, uint32_t* __restrict__ out, size_t n) {
  uint32_t acc = 0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) acc ^= src[i];
  if (acc == 0x12345678u) out[0] = acc;
}
__global__ __launch_bounds__(256) void lds16(const uint4* __restrict__ src, uint32_t* __restrict__ out, size_t n) {
  __shared__ __align__(16) uint4 slot[4][64];                  // one 1 KB slot per wave, overwritten by every load (as l2.hip's ring slots are)
  const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
  typedef __attribute__((address_space(3))) void* lptr_t;
  const uint32_t lds_addr = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(size_t)(lptr_t)(&slot[wave][0]));
  const uint32_t voff = lane * 16u;
  for (size_t i = (size_t)blockIdx.x * 256 + wave * 64; i < n; i += (size_t)gridDim.x * 256) {
    const uint64_t a = (uint64_t)(src + i);                   // wave-uniform base, lane offset in the VGPR: l2.hip's form
    const uint64_t g = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(a >> 32)) << 32) |
                       (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)a);
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(g), "s"(lds_addr) : "memory", "m0");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (slot[wave][lane].x == 0x12345678u) out[0] = 1;
}
__global__ __launch_bounds__(256) void sread(const uint32_t* __restrict__ src, uint32_t* __restrict__ out, size_t n_words) {
  // every wave reads its own 64-byte rows with scalar loads (wave-uniform addresses), 16 dwords per load
  const size_t wave = (size_t)blockIdx.x * 4 + (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), n_waves = (size_t)gridDim.x * 4;
  uint32_t acc = 0;
  for (size_t r = wave; r * 16 < n_words; r += n_waves) {
    const uint32_t* p = src + r * 16;
    uint32_t v[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) v[j] = p[j];                                          // uniform address: the compiler emits s_load_dwordx16
#pragma unroll
    for (int j = 0; j < 16; ++j) acc ^= v[j];
  }
  if (acc == 0x12345678u) out[0] = acc;
}

int main(int argc, char** argv) {
  const size_t bytes = argc > 1 ? strtoull(argv[1], nullptr, 10) : (size_t)2 << 30;
  const char* which = argc > 2 ? argv[2] : "all";
  void *src = nullptr, *dst = nullptr; uint32_t* out = nullptr;
  CHECK(hipMalloc(&src, bytes)); CHECK(hipMalloc(&dst, bytes)); CHECK(hipMalloc(&out, 256));
  CHECK(hipMemset(src, 0x5a, bytes)); CHECK(hipMemset(dst, 0, bytes)); CHECK(hipMemset(out, 0, 256));
  CHECK(hipDeviceSynchronize());
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  const int grid = 256 * 16;
  auto run = [&](const char* name, auto launch, double moved) -> int {
    if (which[0] != 'a' && std::string(which) != name) return 0;
    launch(); CHECK(hipDeviceSynchronize());                    // warm-up (its dispatch is in the counters too: per-dispatch rows)
    CHECK(hipEventRecord(e0)); launch(); CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-7s %zu bytes streamed: %.3f ms = %.2f TB/s of memory traffic\n", name, bytes, ms, moved / (ms * 1e-3) / 1e12);
    return 0;
  };
  if (run("copy16", [&] { hipLaunchKernelGGL(copy16, dim3(grid), dim3(256), 0, 0, (const uint4*)src, (uint4*)dst, bytes / 16); }, 2.0 * bytes)) return 1;
  if (run("read16", [&] { hipLaunchKernelGGL(read16, dim3(grid), dim3(256), 0, 0, (const uint4*)src, out, bytes / 16); }, 1.0 * bytes)) return 1;
  if (run("read4", [&] { hipLaunchKernelGGL(read4, dim3(grid), dim3(256), 0, 0, (const uint32_t*)src, out, bytes / 4); }, 1.0 * bytes)) return 1;
  if (run("lds16", [&] { hipLaunchKernelGGL(lds16, dim3(grid), dim3(256), 0, 0, (const uint4*)src, out, bytes / 16); }, 1.0 * bytes)) return 1;
  if (run("sread", [&] { hipLaunchKernelGGL(sread, dim3(grid), dim3(256), 0, 0, (const uint32_t*)src, out, bytes / 4); }, 1.0 * bytes)) return 1;
  return 0;
}
