// Stage C device code (gfx950): the graph-constrained RANSAC verifier of wg-perception/tod.
// Reference: src/common/adjacency_ransac.cpp, sac_model_registration_graph.h, ransac.h, maximum_clique.cpp.
// All adjacency is kept as bit matrices (n rows of W 64-bit words); one wave owns one hypothesis.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

namespace tod {

typedef unsigned long long u64;

constexpr int kMaxWords = 512;     // n <= 32768 matches per object (conf/detection.ork:26,  n_features 5000 x the cell's k = 5 -> 25000)
constexpr int kWPL = 8;            // bitset words per lane (kMaxWords / 64)
constexpr uint32_t kMaxSampleChecks = 1000;   // sac_model_registration_graph.h:366
constexpr uint32_t kGateMinimal = 7;          // min(best_inlier_number_, 7) is always 7 (:85,:203,:268)
constexpr int kStepCap = 100000;              // maximum_clique.cpp:318

// One object's matches == one tod::AdjacencyRansac (adjacency_ransac.h:48-133), device resident.
struct ObjJob {
  uint32_t n, W;
  const float* train;     // n x 3  training_points_
  const float* query;     // n x 3  query_points_
  const uint32_t* qidx;   // n      query_indices_ (non-decreasing)
  const float* kpxy;      // n x 2  pixel of the keypoint of each match
  u64* phys;              // n x W  physical_adjacency_
  u64* samp;              // n x W  sample_adjacency_
  u64* finite;            // W      matches whose six coordinates are finite
  u64* valid;             // W      valid_indices_
  u64* deg7;              // W      valid vertices with sample degree >= 7
  uint32_t* sampdeg;      // n      sample degree inside the valid set
};

// one entry of the speculative draw table: the outcome of ONE drawIndexSampleHelper attempt that starts
// at a given position of the rand() stream
struct DrawEntry { uint32_t status, consumed, s0, s1, s2, pad; };   // status: 0 fail, 1 ok, 2 window overflow
enum { DRAW_FAIL = 0, DRAW_OK = 1, DRAW_OVERFLOW = 2 };

struct ChainOut {       // result of walking the draw table (getSamples x iterations)
  uint32_t n_done;      // iterations drawn in this walk
  uint32_t pos_end;     // table position after the last completed attempt
  uint32_t attempts;    // failed attempts of the unfinished getSamples call (carried to the next window)
  uint32_t flag;        // 0 = all requested iterations drawn, 1 = window exhausted, 2 = selection empty (1000 failures)
};

// ---------------------------------------------------------------------------------------------- wave helpers
__device__ __forceinline__ uint32_t lane_id() { return threadIdx.x & 63u; }

// wave-uniform value -> SGPR: branches on it are scalar, so the compiler cannot split the wave's lanes
__device__ __forceinline__ uint32_t uni(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }

__device__ __forceinline__ uint32_t wave_sum(uint32_t v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  return uni(v);
}
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v) {
  const uint32_t l = lane_id();
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    uint32_t t = __shfl_up(v, d);
    if (l >= (uint32_t)d) v += t;
  }
  return v;
}
__device__ __forceinline__ u64 shfl64(u64 v, uint32_t src) {
  uint32_t lo = __shfl((uint32_t)v, src), hi = __shfl((uint32_t)(v >> 32), src);
  return ((u64)hi << 32) | lo;
}
__device__ __forceinline__ uint32_t nth_set_bit(u64 w, uint32_t n) {
  for (uint32_t i = 0; i < n; ++i) w &= w - 1;
  return (uint32_t)__ffsll((long long)w) - 1u;
}

// A bitset over the n matches, distributed over the wave: lane l holds words l, l + 64, ...
struct WaveBits {
  u64 w[kWPL];
};
__device__ __forceinline__ void wb_load(WaveBits& b, const u64* p, uint32_t W) {
  const uint32_t l = lane_id();
#pragma unroll
  for (int j = 0; j < kWPL; ++j) b.w[j] = (j * 64u + l) < W ? p[j * 64u + l] : 0ull;
}
__device__ __forceinline__ void wb_and(WaveBits& b, const u64* p, uint32_t W) {
  const uint32_t l = lane_id();
#pragma unroll
  for (int j = 0; j < kWPL; ++j) b.w[j] &= (j * 64u + l) < W ? p[j * 64u + l] : 0ull;
}
__device__ __forceinline__ uint32_t wb_count(const WaveBits& b) {
  uint32_t c = 0;
#pragma unroll
  for (int j = 0; j < kWPL; ++j) c += (uint32_t)__popcll(b.w[j]);
  return wave_sum(c);
}
__device__ __forceinline__ void wb_clear(WaveBits& b, uint32_t v) {
  const uint32_t word = v >> 6, l = lane_id();
#pragma unroll
  for (int j = 0; j < kWPL; ++j)
    if (word == j * 64u + l) b.w[j] &= ~(1ull << (v & 63u));
}
__device__ __forceinline__ void wb_set(WaveBits& b, uint32_t v) {
  const uint32_t word = v >> 6, l = lane_id();
#pragma unroll
  for (int j = 0; j < kWPL; ++j)
    if (word == j * 64u + l) b.w[j] |= (1ull << (v & 63u));
}
// index of the idx-th set bit (ascending); idx < count and idx wave-uniform
__device__ __forceinline__ uint32_t wb_select(const WaveBits& b, uint32_t idx) {
  idx = uni(idx);
  uint32_t result = 0xFFFFFFFFu;
#pragma unroll
  for (int j = 0; j < kWPL; ++j) {
    const uint32_t c = (uint32_t)__popcll(b.w[j]);
    const uint32_t incl = wave_incl_scan(c);
    const uint32_t total = uni(__shfl(incl, 63));
    if (result == 0xFFFFFFFFu) {
      if (idx < total) {
        const bool mine = (incl - c) <= idx && idx < incl;
        const u64 bal = __ballot(mine);
        const uint32_t owner = (uint32_t)__ffsll((long long)bal) - 1u;
        const uint32_t bit = mine ? nth_set_bit(b.w[j], idx - (incl - c)) : 0u;
        result = (j * 64u + owner) * 64u + uni(__shfl(bit, owner));
      } else {
        idx -= total;
      }
    }
  }
  return result;
}

// ---------------------------------------------------------------------------------------------- float helpers
// Arithmetic order and precision follow the reference; the library is compiled with
// -ffp-contract=off so no product is fused into a following sum.
__device__ __forceinline__ float dist_sq3(const float* a, const float* b) {   // sac_model_registration_graph.h:52-58
  float t0 = a[0] - b[0], t1 = a[1] - b[1], t2 = a[2] - b[2];
  return t0 * t0 + t1 * t1 + t2 * t2;
}
__device__ __forceinline__ double norm3d(float x, float y, float z) {         // cv::norm(Vec3f): double accumulation
  return sqrt((double)x * (double)x + (double)y * (double)y + (double)z * (double)z);
}

}  // namespace tod
