// Stage C on gfx950: GuessGenerator::process (reference src/detection/GuessGenerator.cpp:127-250) and the
// verifier library under src/common (adjacency_ransac.cpp, sac_model_registration_graph.h, ransac.h,
// maximum_clique.cpp). Decisions D2-D4 of SURVEY.md App. A apply.
//
// Kernels
//   K6  adjacency_kernel     FillAdjacency (adjacency_ransac.cpp:127-172): one wave = 64 pair tests -> one
//                            64-bit word of the physical and of the sample bit matrix via __ballot
//   K7a draw_table_kernel    drawIndexSampleHelper (sac_model_registration_graph.h:102-132) evaluated
//                            speculatively from EVERY position of the rand() stream window, one wave each
//   K7b chain_kernel         getSamples (:141-168) x iterations: pointer chase through the table
//   K8  eval_kernel          selectWithinDistance (:171-269): 3-row AND + popcount, degree filter and the
//                            maximum-clique gate (maximum_clique.cpp:286-369) on an LDS-resident induced graph
//   K9  growth_kernel        Ransac's refinement loop (adjacency_ransac.cpp:255-308) incl. Kabsch/SVD (:304-347)
//   K11 invalidate_kernel    InvalidateQueryIndices / InvalidateIndices (:63-123)
// The RANSAC bookkeeping (ransac.h:95-135: strictly-better test, adaptive k with pow/log) is replayed on the
// host from the per-iteration consensus counts, so that libm results are those of the CPU reference.
#include <algorithm>
#include <chrono>
#include <atomic>
#include <climits>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <map>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

#include "ctx.h"
#include "verify_kernels.h"

using namespace tod;

namespace {

// Every kernel takes up to kMaxSlots argument sets and picks its own with the last grid dimension: the frames of a
// batch (each at its own point of its own RANSAC state machine) share launches, so a batch costs the launches and
// host round trips of one frame.
constexpr uint32_t kMaxSlots = 16;                         // 16 x 200 B of EvalArgs stays under the 4 KB kernarg limit
constexpr uint32_t kManySlots = 36;                        // kernels with ~110 B of arguments per slot (4 KB of kernarg in all)
template <class A, uint32_t N = kMaxSlots> struct Slots { A a[N]; };
// the same argument sets in device memory (lists of thousands of objects: launch_many)
template <class A> struct SlotsPtr { const A* a; };

struct AdjArgs { ObjJob job; float span, err; };
struct JobArgs { ObjJob job; };
// gate (optional): the kernel does nothing unless *gate >= gate_min -- the next round's preparation rides in the tick of the growth
// that decides whether there is a next round (GrowthOut::n_kp_inliers against min_inliers, GuessGenerator.cpp:205-206)
struct PrepArgs { ObjJob job; uint32_t* stats; const uint32_t* gate; uint32_t gate_min; };   // stats[0] = |valid|, [1] = sum of sample degrees inside valid, [2] = triangle found
struct DrawArgs { ObjJob job; const uint32_t* rnd; uint32_t window_len, S; DrawEntry* table; };
struct ChainArgs {
  const DrawEntry* table; uint32_t S, n_req, attempts0, out_base;
  uint32_t* iter_samples; uint32_t* iter_pos_after; ChainOut* out;
};
struct CopyArgs { const uint32_t* src; uint32_t* dst; uint32_t n; };   // src == nullptr: zero fill

// words from one address space to another (device <-> device-visible pinned host memory) or zero fill: the
// host's mailbox traffic rides in kernels, so a tick of the batch engine is launches + ONE stream synchronize
constexpr uint32_t kWideSlots = 32;                        // argument sets of <= 124 B: a 32-frame batch's tick in one launch
constexpr uint32_t kCopySlots = 160;                       // 24 B per copy: a tick's copies of 16 frames in one launch
__global__ __launch_bounds__(256) void copy_words_kernel(Slots<CopyArgs, kCopySlots> S) {
  const CopyArgs& a = S.a[blockIdx.y];
  for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < a.n; i += gridDim.x * 256u) a.dst[i] = a.src ? a.src[i] : 0u;
}

// TODHIP_DEBUG=1: ticks, flights and sprints (a timeline that costs a few lines per tick); 2: also every round, draw window, evaluation
inline int tod_debug_level() { static const int lv = [] { const char* e = getenv("TODHIP_DEBUG"); return e ? std::max(1, atoi(e)) : 0; }(); return lv; }   // read once
inline bool tod_debug() { return tod_debug_level() > 0; }
inline double dbg_us() {
  static const std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
  return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
}
inline int dbg_thread() { static std::atomic<int> next{0}; thread_local int id = next.fetch_add(1); return id; }   // which host thread (= which context's batch)
#define TOD_DBG2(...) do { if (tod_debug_level() > 1) { fprintf(stderr, "[todhip %.0f] ", dbg_us()); fprintf(stderr, __VA_ARGS__); fputc('\n', stderr); fflush(stderr); } } while (0)
#define TOD_DBG(...) do { if (tod_debug()) { char b_[512]; int n_ = snprintf(b_, sizeof(b_), "[todhip %.0f] ", dbg_us()); n_ += snprintf(b_ + n_, sizeof(b_) - n_, __VA_ARGS__); \
    snprintf(b_ + std::min<int>(n_, (int)sizeof(b_) - 16), 16, " {t%d}\n", dbg_thread()); fputs(b_, stderr); } } while (0)   /* one write per line: threads do not interleave */

// ------------------------------------------------------------------------------------------------ K6
// one pair of FillAdjacency (adjacency_ransac.cpp:136-166): q / t = query / training point, k = keypoint pixel of the two matches
__device__ __forceinline__ void pair_test(const float* q1, const float* q2, const float* t1, const float* t2, const float* k1, const float* k2,
                                          float span, float err, bool& phys, bool& samp) {
  phys = false; samp = false;
  float dq = dist_sq3(q1, q2);
  const float lim = (span + 2 * err) * (span + 2 * err);
  if (!(dq > lim)) {                                      // adjacency_ransac.cpp:144
    dq = sqrtf(dq);
    const float dt = (float)norm3d(t1[0] - t2[0], t1[1] - t2[1], t1[2] - t2[2]);
    const float a = fabsf(dt - dq);
    if (!(a > 4 * err)) {                                 // :151
      phys = true;
      const float px = (k1[0] - k2[0]) * (k1[0] - k2[0]) + (k1[1] - k2[1]) * (k1[1] - k2[1]);
      samp = (px > 20 * 20) && (a < 2 * err);             // :158-161
    }
  }
}

template <class H>
__global__ __launch_bounds__(256) void adjacency_kernel(H S) {
  TOD_LATENCY_PRIO();   // latency-bound: win issue arbitration against the VALU-saturating matcher
  const ObjJob& job = S.a[blockIdx.z].job;
  const float span = S.a[blockIdx.z].span, err = S.a[blockIdx.z].err;
  const uint32_t i = blockIdx.x;
  if (i >= job.n) return;
  const uint32_t word = blockIdx.y * 4u + (threadIdx.x >> 6);
  if (word >= job.W) return;
  const uint32_t j = word * 64u + lane_id();
  bool phys = false, samp = false;
  if (j < job.n && j != i) {
    const uint32_t lo = min(i, j), hi = max(i, j);       // the reference visits each pair once with i < j
    pair_test(job.query + 3 * lo, job.query + 3 * hi, job.train + 3 * lo, job.train + 3 * hi, job.kpxy + 2 * lo, job.kpxy + 2 * hi, span, err,
              phys, samp);
  }
  const u64 pb = __ballot(phys), sb = __ballot(samp);
  if (lane_id() == 0) {
    job.phys[(size_t)i * job.W + word] = pb;
    job.samp[(size_t)i * job.W + word] = sb;
  }
}

template <class H>
__global__ __launch_bounds__(256) void finite_kernel(H S) {
  TOD_LATENCY_PRIO();   // latency-bound: win issue arbitration against the VALU-saturating matcher
  const ObjJob& job = S.a[blockIdx.y].job;
  const uint32_t v = blockIdx.x * 256u + threadIdx.x;
  bool f = false;
  if (v < job.n) {
    f = true;
    for (int c = 0; c < 3; ++c) f = f && isfinite(job.train[3 * v + c]) && isfinite(job.query[3 * v + c]);
  }
  const u64 b = __ballot(f);
  const uint32_t word = v >> 6;
  if (lane_id() == 0 && word < job.W) {
    job.finite[word] = b;
    const uint32_t base = word * 64u;
    job.valid[word] = base + 64u <= job.n ? ~0ull : (base < job.n ? ((1ull << (job.n - base)) - 1ull) : 0ull);
  }
}

// per round: sample degree inside the valid set, the ">= 7" filter mask (:211-213), |valid| -- and whether the sample
// graph on the valid matches holds a triangle at all. Without one, no drawIndexSampleHelper attempt can succeed
// (sac_model_registration_graph.h:102-132 needs three mutually sample-adjacent indices), and a failing attempt consumes a
// number of rand() calls that does not depend on the values drawn: the top level draws and erases every valid index once
// (|valid| draws); under pick v the second level draws and erases every not-yet-erased neighbour of v (one draw each --
// its own third level is empty, so it returns before drawing), i.e. every edge is paid for exactly once, at whichever
// endpoint is picked first. A triangle-free object therefore advances the stream by exactly 1000 (|valid| + |E|) draws
// (getSamples' 1000 attempts, :141-168) and yields nothing: the host skips its draw table and chain walk altogether.
template <class H>
__global__ __launch_bounds__(256) void round_prep_kernel(H S) {
  TOD_LATENCY_PRIO();   // latency-bound: win issue arbitration against the VALU-saturating matcher
  const ObjJob& job = S.a[blockIdx.y].job;
  uint32_t* const stats = S.a[blockIdx.y].stats;
  if (S.a[blockIdx.y].gate && *S.a[blockIdx.y].gate < S.a[blockIdx.y].gate_min) return;   // block-uniform
  const uint32_t v = blockIdx.x * 256u + threadIdx.x;
  bool isv = false;
  uint32_t d = 0;
  if (v < job.n) {
    isv = (job.valid[v >> 6] >> (v & 63u)) & 1ull;
    if (isv)
      for (uint32_t w = 0; w < job.W; ++w) d += (uint32_t)__popcll(job.samp[(size_t)v * job.W + w] & job.valid[w]);
    job.sampdeg[v] = d;
  }
  const u64 b7 = __ballot(isv && d >= kGateMinimal), bv = __ballot(isv);
  const uint32_t dsum = wave_sum(d);
  if (lane_id() == 0 && (v >> 6) < job.W) {
    job.deg7[v >> 6] = b7;
    if (bv) atomicAdd(stats, (uint32_t)__popcll(bv));
    if (dsum) atomicAdd(stats + 1, dsum);
  }
  // triangle through v: a neighbour j > v that shares a neighbour with v (all inside valid). One finder is enough.
  if (isv && d >= 2u) {
    const u64* rv = job.samp + (size_t)v * job.W;
    for (uint32_t wj = v >> 6; wj < job.W; ++wj) {
      u64 nb = rv[wj] & job.valid[wj];
      if (wj == (v >> 6)) nb &= (v & 63u) == 63u ? 0ull : ~0ull << ((v & 63u) + 1u);
      while (nb) {
        if (__hip_atomic_load(stats + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return;
        const uint32_t j = wj * 64u + (uint32_t)__ffsll((long long)nb) - 1u;
        nb &= nb - 1ull;
        const u64* rj = job.samp + (size_t)j * job.W;
        u64 common = 0;
        for (uint32_t w = 0; w < job.W; ++w) common |= rv[w] & rj[w] & job.valid[w];
        if (common) { __hip_atomic_store(stats + 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return; }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------ K7a
__global__ __launch_bounds__(256) void draw_table_kernel(Slots<DrawArgs, kWideSlots> SL) {
  TOD_LATENCY_PRIO();   // latency-bound: win issue arbitration against the VALU-saturating matcher
  const ObjJob& job = SL.a[blockIdx.y].job;
  const uint32_t* __restrict__ rnd = SL.a[blockIdx.y].rnd;
  const uint32_t window_len = SL.a[blockIdx.y].window_len, S = SL.a[blockIdx.y].S;
  DrawEntry* const table = SL.a[blockIdx.y].table;
  const uint32_t s = blockIdx.x * 4u + (threadIdx.x >> 6);
  if (s >= S) return;
  const uint32_t W = job.W;
  WaveBits VA;
  wb_load(VA, job.valid, W);
  uint32_t nA = wb_count(VA);
  uint32_t pos = s, status = DRAW_FAIL, s0 = 0, s1 = 0, s2 = 0;
  while (nA > 0) {                                        // level "3 samples left"
    if (pos >= window_len) { status = DRAW_OVERFLOW; break; }
    const uint32_t a = wb_select(VA, rnd[pos++] % nA);    // valid_samples[rand() % size], :111
    if (a >= job.n) break;                                // cannot happen; keeps every address in bounds
    WaveBits VB = VA;
    wb_and(VB, job.samp + (size_t)a * W, W);              // set_intersection with the sample neighbours, :113-117
    uint32_t nB = wb_count(VB);
    bool ok = false;
    uint32_t b = 0, c = 0;
    while (nB > 0) {                                      // level "2 samples left"
      if (pos >= window_len) { status = DRAW_OVERFLOW; break; }
      b = wb_select(VB, rnd[pos++] % nB);
      if (b >= job.n) { nB = 0; break; }
      WaveBits VC = VB;
      wb_and(VC, job.samp + (size_t)b * W, W);
      const uint32_t nC = wb_count(VC);
      if (nC > 0) {                                       // level "1 sample left": any pick succeeds
        if (pos >= window_len) { status = DRAW_OVERFLOW; break; }
        c = wb_select(VC, rnd[pos++] % nC);
        ok = true;
        break;
      }
      wb_clear(VB, b);                                    // std::remove of the failed pick, :125-128
      --nB;
    }
    if (status == DRAW_OVERFLOW) break;
    if (ok) { status = DRAW_OK; s0 = c; s1 = b; s2 = a; break; }   // samples_ is deepest-first, :118-121
    wb_clear(VA, a);
    --nA;
  }
  if (lane_id() == 0) {
    DrawEntry e;
    e.status = status; e.consumed = pos - s; e.s0 = s0; e.s1 = s1; e.s2 = s2; e.pad = 0;
    table[s] = e;
  }
}

// K7a for objects of at most 128 matches (W <= 2): ONE LANE per stream position instead of one wave -- the bitsets fit
// two 64-bit registers, so a lane runs the helper's three levels on its own. Distractor objects (a few dozen random
// matches, hopeless, burning their whole iteration budget and hundreds of failed attempts) are what makes the table
// large, and they are small: 64 x fewer waves for the same table.
__device__ __forceinline__ uint32_t nth_set_bit128(u64 w0, u64 w1, uint32_t n) {   // n-th set bit (ascending), n < popc
  const uint32_t c0 = (uint32_t)__popcll(w0);
  u64 w = w0; uint32_t base = 0;
  if (n >= c0) { n -= c0; w = w1; base = 64u; }
  uint32_t pos = 0;
#pragma unroll
  for (uint32_t shift = 32; shift > 0; shift >>= 1) {
    const uint32_t cnt = (uint32_t)__popcll((w >> pos) & ((1ull << shift) - 1ull));
    if (n >= cnt) { n -= cnt; pos += shift; }
  }
  return base + pos;
}
__global__ __launch_bounds__(256) void draw_table_small_kernel(Slots<DrawArgs, kWideSlots> SL) {
  TOD_LATENCY_PRIO();   // latency-bound: win issue arbitration against the VALU-saturating matcher
  const ObjJob& job = SL.a[blockIdx.y].job;
  const uint32_t* __restrict__ rnd = SL.a[blockIdx.y].rnd;
  const uint32_t window_len = SL.a[blockIdx.y].window_len, S = SL.a[blockIdx.y].S;
  DrawEntry* const table = SL.a[blockIdx.y].table;
  const uint32_t s = blockIdx.x * 256u + threadIdx.x;
  if (s >= S) return;
  const uint32_t W = job.W;                                // 1 or 2
  u64 a0 = job.valid[0], a1 = W > 1u ? job.valid[1] : 0ull;
  uint32_t nA = (uint32_t)(__popcll(a0) + __popcll(a1));
  uint32_t pos = s, status = DRAW_FAIL, s0 = 0, s1 = 0, s2 = 0;
  while (nA > 0) {                                          // level "3 samples left"
    if (pos >= window_len) { status = DRAW_OVERFLOW; break; }
    const uint32_t a = nth_set_bit128(a0, a1, rnd[pos++] % nA);   // valid_samples[rand() % size], :111
    if (a >= job.n) break;                                  // cannot happen; keeps every address in bounds
    u64 b0 = a0 & job.samp[(size_t)a * W], b1 = W > 1u ? (a1 & job.samp[(size_t)a * W + 1]) : 0ull;   // :113-117
    uint32_t nB = (uint32_t)(__popcll(b0) + __popcll(b1));
    bool ok = false;
    uint32_t b = 0, c = 0;
    while (nB > 0) {                                        // level "2 samples left"
      if (pos >= window_len) { status = DRAW_OVERFLOW; break; }
      b = nth_set_bit128(b0, b1, rnd[pos++] % nB);
      if (b >= job.n) { nB = 0; break; }
      const u64 c0 = b0 & job.samp[(size_t)b * W], c1 = W > 1u ? (b1 & job.samp[(size_t)b * W + 1]) : 0ull;
      const uint32_t nC = (uint32_t)(__popcll(c0) + __popcll(c1));
      if (nC > 0) {                                         // level "1 sample left": any pick succeeds
        if (pos >= window_len) { status = DRAW_OVERFLOW; break; }
        c = nth_set_bit128(c0, c1, rnd[pos++] % nC);
        ok = true;
        break;
      }
      if (b < 64u) b0 &= ~(1ull << b); else b1 &= ~(1ull << (b - 64u));   // std::remove of the failed pick, :125-128
      --nB;
    }
    if (status == DRAW_OVERFLOW) break;
    if (ok) { status = DRAW_OK; s0 = c; s1 = b; s2 = a; break; }   // samples_ is deepest-first, :118-121
    if (a < 64u) a0 &= ~(1ull << a); else a1 &= ~(1ull << (a - 64u));
    --nA;
  }
  DrawEntry e;
  e.status = status; e.consumed = pos - s; e.s0 = s0; e.s1 = s1; e.s2 = s2; e.pad = 0;
  table[s] = e;
}

// ------------------------------------------------------------------------------------------------ K7b
// The walk is a pointer chase (position -> position + consumed): through global memory every hop costs a DRAM/L2
// round trip (~1.5 us; 2500 iterations of a hopeless object = 4 ms), so the hop data (status, consumed) of the whole
// window is first packed into LDS by all threads, one lane walks it there, recording where each successful attempt
// started, and all threads then fetch those attempts' triples.
constexpr uint32_t kChainLdsEntries = 36u * 1024u;       // 144 KB of packed (consumed << 2 | status) words
constexpr uint32_t kChainMaxReq = 4096u;                  // == kMaxEvalWaves
__global__ __launch_bounds__(256) void chain_kernel(Slots<ChainArgs, kWideSlots> SL) {
  TOD_LATENCY_PRIO();   // latency-bound: win issue arbitration against the VALU-saturating matcher
  extern __shared__ __align__(16) unsigned char lds_raw[];
  uint32_t* s_hop = reinterpret_cast<uint32_t*>(lds_raw);                       // min(S, kChainLdsEntries)
  const ChainArgs& a = SL.a[blockIdx.x];
  const DrawEntry* __restrict__ table = a.table;
  const uint32_t S = a.S, n_req = a.n_req, attempts0 = a.attempts0, out_base = a.out_base;
  uint32_t* const iter_samples = a.iter_samples; uint32_t* const iter_pos_after = a.iter_pos_after; ChainOut* const out = a.out;
  const uint32_t n_lds = min(S, kChainLdsEntries);
  uint32_t* s_start = s_hop + n_lds;                                            // n_req: table position of each drawn iteration
  __shared__ uint32_t s_done;
  for (uint32_t i = threadIdx.x; i < n_lds; i += 256u) s_hop[i] = (table[i].consumed << 2) | table[i].status;
  __syncthreads();
  if (threadIdx.x < 64u) {
    // Wave 0 walks, with wave-uniform control flow: the hop words of 64 consecutive positions sit in one register (lane i = position
    // base + i) and a hop inside the window is a v_readlane -- an attempt consumes a handful of draws, so a window serves ~10 hops
    // and the dependent LDS round trip (~100 cycles, what a hop cost before) is paid once per window.
    const uint32_t lane = threadIdx.x;
    uint32_t p = 0, done = 0, attempts = attempts0, flag = 0, base = 0xFFFFFF00u, win = 0;
    while (done < n_req) {
      bool got = false;
      while (true) {
        if (p >= S) { flag = 1; break; }
        if (p - base >= 64u) {                              // wave-uniform (also true on the first pass: base is far above)
          base = p;
          const uint32_t idx = base + lane;
          win = idx < n_lds ? s_hop[idx] : (idx < S ? ((table[idx].consumed << 2) | table[idx].status) : 0u);
        }
        const uint32_t hop = (uint32_t)__builtin_amdgcn_readlane((int)win, (int)(p - base));
        const uint32_t status = hop & 3u;
        if (status == DRAW_OVERFLOW) { flag = 1; break; }
        const uint32_t at = p;
        p = uni(p + (hop >> 2));
        if (status == DRAW_OK) {
          if (lane == 0) { s_start[done] = at; iter_pos_after[out_base + done] = p; }
          got = true;
          break;
        }
        if (++attempts >= kMaxSampleChecks) { flag = 2; break; }   // getSamples gives up: samples.clear(), :167
      }
      if (!got) break;
      attempts = 0;
      ++done;
    }
    if (lane == 0) {
      out->n_done = done; out->pos_end = p; out->attempts = attempts; out->flag = flag;
      s_done = done;
    }
  }
  __syncthreads();
  for (uint32_t i = threadIdx.x; i < s_done; i += 256u) {
    const DrawEntry e = table[s_start[i]];
    iter_samples[3 * (out_base + i) + 0] = e.s0;
    iter_samples[3 * (out_base + i) + 1] = e.s1;
    iter_samples[3 * (out_base + i) + 2] = e.s2;
  }
}

// ------------------------------------------------------------------------------------------------ K8
struct GateLds {
  u64* adjc;                                   // m x MW induced sample adjacency, graph index = rank in F
  u64* mask;                                   // MW
  uint16_t *flist, *cur, *nxt, *tmp;           // m each
  uint32_t *C, *deg, *keys;                    // m each
  uint32_t *S, *SOld, *lbase, *lsize, *lcap;   // m + 2 each
  uint32_t* trash;                             // 64 words: where the lanes that have nothing to record write (colour_first_fit64)
  uint16_t* lstack;                            // LDS part of the per-level vertex lists (the rest is in global memory)
  uint32_t lstack_cap;
};
__host__ __device__ inline uint32_t gate_lds_bytes(uint32_t m) {
  const uint32_t MW = (m + 63u) / 64u, ma = (m + 7u) & ~3u;      // ma >= m + 2
  return 8u * m * MW + 8u * MW + 8u * 4u * ma + 4u * 2u * ma + 256u + 64u;
}
__host__ __device__ inline uint32_t gate_small_bytes(uint32_t m) {           // everything except the adjacency matrix
  const uint32_t MW = (m + 63u) / 64u, ma = (m + 7u) & ~3u;
  return 8u * MW + 8u * 4u * ma + 4u * 2u * ma + 256u + 64u;
}
// ext_adjc != nullptr: the m x MW matrix lives in global scratch (graphs beyond one CU's LDS); same code path,
// the pointers are generic
// kExt is a template parameter so that, in the LDS instantiation, every pointer provably comes from the LDS allocation:
// the compiler then emits ds_read/ds_write for the adjacency rows instead of flat loads (the rows are on the critical
// path of Intersection and ColorSort)
// rows: how many adjacency rows to make room for (m, or the object's n when the graph keeps the object's vertex numbers)
template <bool kExt>
__device__ __forceinline__ GateLds gate_carve(unsigned char* base, uint32_t m, uint32_t lds_bytes, u64* ext_adjc = nullptr, uint32_t rows = 0) {
  const uint32_t MW = (m + 63u) / 64u, ma = (m + 7u) & ~3u;
  if (rows == 0u) rows = m;
  unsigned char* const base0 = base;
  GateLds L;
  if constexpr (kExt) { L.adjc = ext_adjc; } else { L.adjc = reinterpret_cast<u64*>(base); base += 8u * rows * MW; }
  L.mask = reinterpret_cast<u64*>(base); base += 8u * MW;
  L.C = reinterpret_cast<uint32_t*>(base); base += 4u * ma;
  L.deg = reinterpret_cast<uint32_t*>(base); base += 4u * ma;
  L.keys = reinterpret_cast<uint32_t*>(base); base += 4u * ma;
  L.S = reinterpret_cast<uint32_t*>(base); base += 4u * ma;
  L.SOld = reinterpret_cast<uint32_t*>(base); base += 4u * ma;
  L.lbase = reinterpret_cast<uint32_t*>(base); base += 4u * ma;
  L.lsize = reinterpret_cast<uint32_t*>(base); base += 4u * ma;
  L.lcap = reinterpret_cast<uint32_t*>(base); base += 4u * ma;
  L.trash = reinterpret_cast<uint32_t*>(base); base += 256u;
  L.flist = reinterpret_cast<uint16_t*>(base); base += 2u * ma;
  L.cur = reinterpret_cast<uint16_t*>(base); base += 2u * ma;
  L.nxt = reinterpret_cast<uint16_t*>(base); base += 2u * ma;
  L.tmp = reinterpret_cast<uint16_t*>(base); base += 2u * ma;
  // whatever the launch's LDS allocation has left is the first part of the level stack
  uint32_t used = ((uint32_t)(base - base0) + 15u) & ~15u;
  L.lstack = reinterpret_cast<uint16_t*>(base0 + used);
  L.lstack_cap = lds_bytes > used ? (lds_bytes - used) / 2u : 0u;
  return L;
}

// per-level vertex lists: entry i lives in LDS while it fits, in the wave's global stack beyond
struct LevelStack { uint16_t* lds; uint32_t lds_cap; uint16_t* glob; };
__device__ __forceinline__ uint16_t stk_get(const LevelStack& s, uint32_t i) {
  return i < s.lds_cap ? s.lds[i] : s.glob[i - s.lds_cap];
}
__device__ __forceinline__ void stk_put(const LevelStack& s, uint32_t i, uint16_t v) {
  if (i < s.lds_cap) s.lds[i] = v; else s.glob[i - s.lds_cap] = v;
}

__device__ __forceinline__ bool row_test(u64 roww, uint32_t h) {   // roww: lane l holds word l of the row
  const u64 wv = shfl64(roww, h >> 6);
  return (wv >> (h & 63u)) & 1ull;
}

__device__ __forceinline__ uint32_t rdlane(uint32_t v, uint32_t lane) {    // lane is wave-uniform: v_readlane_b32
  return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)lane);
}
__device__ __forceinline__ u64 rdlane64(u64 v, uint32_t lane) {
  return ((u64)rdlane((uint32_t)(v >> 32), lane) << 32) | rdlane((uint32_t)v, lane);
}

constexpr uint32_t kRegChunks = 8;                         // the column compaction's register path covers graphs of up to 512 vertices
constexpr uint32_t kSortChunks = 16;                       // DegreeSort's register path: lists of up to 1024 vertices

// DegreeSort (maximum_clique.cpp:263-284): (degree inside the list, vertex) ascending, then reversed.
// deg[] must hold the degree of list[i] at position i. Rank by counting; keys are unique.
// keys stay in registers (lane l: positions l, l + 64, ...); every key is broadcast once with v_readlane. NCH = chunks of 64
// positions, a template parameter so that the per-key work is straight-line code over exactly NCH registers.
template <uint32_t NCH>
__device__ __forceinline__ void rank_sort_regs(uint16_t* list, const uint32_t* deg, uint32_t r) {
  const uint32_t l = lane_id();
  uint32_t kreg[NCH], rank[NCH];
#pragma unroll
  for (uint32_t c = 0; c < NCH; ++c) {
    const uint32_t i = c * 64u + l;
    kreg[c] = i < r ? ((deg[i] << 16) | list[i]) : 0u;
    rank[c] = 0u;
  }
#pragma unroll
  for (uint32_t cj = 0; cj < NCH; ++cj) {
    const uint32_t cnt = cj * 64u < r ? min(64u, r - cj * 64u) : 0u;   // (the last chunk of a merged case may be empty)
    for (uint32_t lj = 0; lj < cnt; ++lj) {
      const uint32_t kj = rdlane(kreg[cj], lj);
#pragma unroll
      for (uint32_t c = 0; c < NCH; ++c) rank[c] += (kj > kreg[c]) ? 1u : 0u;
    }
  }
  __syncthreads();                                         // every lane holds its keys: the list can be overwritten
#pragma unroll
  for (uint32_t c = 0; c < NCH; ++c)
    if (c * 64u + l < r) list[rank[c]] = (uint16_t)(kreg[c] & 0xFFFFu);
  __syncthreads();
}

// kWide: the instantiation for objects of 513..1024 matches (eval_kernel<true>); the narrow one carries none of its code, so that
// the registers of the common case are allocated as if the wide case did not exist (it costs 6 % otherwise)
template <bool kWide>
__device__ __forceinline__ void rank_sort_desc(uint16_t* list, uint16_t* tmp, const uint32_t* deg, uint32_t r, uint32_t* keys) {
  const uint32_t l = lane_id();
  if (r <= kRegChunks * 64u) {
    switch ((r + 63u) / 64u) {                             // wave-uniform
      case 0: case 1: rank_sort_regs<1>(list, deg, r); break;
      case 2: rank_sort_regs<2>(list, deg, r); break;
      case 3: rank_sort_regs<3>(list, deg, r); break;
      case 4: rank_sort_regs<4>(list, deg, r); break;
      case 5: rank_sort_regs<5>(list, deg, r); break;
      case 6: rank_sort_regs<6>(list, deg, r); break;
      case 7: rank_sort_regs<7>(list, deg, r); break;
      default: rank_sort_regs<8>(list, deg, r); break;
    }
    return;
  }
  if constexpr (kWide) {
    if (r <= kSortChunks * 64u) {
      switch ((r + 63u) / 64u) {                           // wave-uniform
        case 9: rank_sort_regs<9>(list, deg, r); break;
        case 10: rank_sort_regs<10>(list, deg, r); break;
        case 11: rank_sort_regs<11>(list, deg, r); break;
        case 12: rank_sort_regs<12>(list, deg, r); break;
        case 13: case 14: rank_sort_regs<14>(list, deg, r); break;
        default: rank_sort_regs<16>(list, deg, r); break;
      }
      return;
    }
  }
  for (uint32_t i = l; i < r; i += 64u) keys[i] = (deg[i] << 16) | list[i];
  __syncthreads();
  for (uint32_t i0 = 0; i0 < r; i0 += 64u) {
    const uint32_t i = i0 + l;
    const uint32_t mine = i < r ? keys[i] : 0u;
    uint32_t rank = 0;
    for (uint32_t j = 0; j < r; ++j) rank += keys[j] > mine;
    if (i < r) tmp[rank] = (uint16_t)(mine & 0xFFFFu);
  }
  __syncthreads();
  for (uint32_t i = l; i < r; i += 64u) list[i] = tmp[i];
  __syncthreads();
}

// degrees of the members of list[0..r) inside the list, into L.deg[0..r)
__device__ __forceinline__ void degrees_in_list(const GateLds& L, const uint16_t* list, uint32_t r, uint32_t MW) {
  const uint32_t l = lane_id();
  if (l < MW) L.mask[l] = 0ull;
  __syncthreads();
  for (uint32_t i = l; i < r; i += 64u) atomicOr(&L.mask[list[i] >> 6], 1ull << (list[i] & 63u));
  __syncthreads();
  for (uint32_t i = l; i < r; i += 64u) {
    const u64* row = L.adjc + (size_t)list[i] * MW;
    uint32_t d = 0;
    for (uint32_t w = 0; w < MW; ++w) d += (uint32_t)__popcll(row[w] & L.mask[w]);
    L.deg[i] = d;
  }
  __syncthreads();
}

// The same first-fit colouring for at most 64 classes, with nothing but vector instructions between one vertex and the next. A lone
// wave pays for every hand-over between the vector and the scalar unit (ballot -> find-first-set -> lane compare -> exec mask, the
// shape of colour_first_fit below, costs ~530 cycles per vertex for ~45 instructions). Here the first free class is found
// lane-locally: the free-class mask stays in VCC, v_mbcnt counts the free classes below each lane, the one lane that is free
// with none below it joins -- a select and an or on its own registers. What ColorSort needs for its output order, (class, rank
// inside the class) per list position, is written by that lane itself: every lane stores one word, the others into a trash slot.
// Returns false (nothing written to the list or to C) when some vertex found all 64 classes taken: the caller then runs the
// two-set form below.
template <uint32_t MWT>
__device__ __forceinline__ bool colour_first_fit64(const GateLds& L, uint16_t* list, uint32_t r) {
  typedef uint32_t u32x16 __attribute__((ext_vector_type(MWT <= 8u ? 16 : 32)));   // (2 MWT halves; the tuple sizes the hardware indexes)
  const uint32_t l = lane_id();
  // class l's members as a bitset over graph vertices, 32-bit halves in ONE register tuple: the half that receives a vertex is
  // picked with the hardware's register indexing (s_set_gpr_idx, the index v >> 5 is wave-uniform) -- three instructions to read,
  // three to write, no branch tree and no per-word selects
  u32x16 cls = {};
  uint32_t rec = l << 16;                                  // (class l, members of class l so far): what a joining vertex records
  uint32_t* const my_trash = L.trash + l;
  for (uint32_t c0 = 0; c0 < r; c0 += 64u) {
    const uint32_t cnt = min(64u, r - c0);
    // lanes past the list hold its last vertex, so that the row prefetch one vertex ahead needs no clamp (lane 64 wraps to lane 0:
    // any vertex will do, the row is never used); and every lane keeps the BYTE offset of its vertex's row next to the vertex
    const uint32_t vmine = (uint32_t)list[min(c0 + l, r - 1u)];
    const uint32_t voff = vmine * (MWT * 8u);
    auto place = [&](const u64 (&row)[MWT], u64 (&next)[MWT], uint32_t li) {
      const uint32_t v = rdlane(vmine, li);
      {
        const u64* g = reinterpret_cast<const u64*>(reinterpret_cast<const unsigned char*>(L.adjc) + rdlane(voff, (li + 1u) & 63u));
#pragma unroll
        for (uint32_t w = 0; w < MWT; ++w) next[w] = g[w];
      }
      // (row & class) over all halves: one v_and_or per half. Left to itself the compiler builds and + and + or3 trees, three
      // instructions per two halves, which is shallower but longer -- and a lone wave is bound by what it must issue, not by depth
      uint32_t hit32 = (uint32_t)row[0] & cls[0];
      {
        const uint32_t rh = (uint32_t)(row[0] >> 32), ch = cls[1];
        asm("v_and_or_b32 %0, %1, %2, %0" : "+v"(hit32) : "v"(rh), "v"(ch));
      }
#pragma unroll
      for (uint32_t w = 1; w < MWT; ++w) {
        const uint32_t rl = (uint32_t)row[w], rh = (uint32_t)(row[w] >> 32), cl = cls[2u * w], ch = cls[2u * w + 1u];
        asm("v_and_or_b32 %0, %1, %2, %0" : "+v"(hit32) : "v"(rl), "v"(cl));
        asm("v_and_or_b32 %0, %1, %2, %0" : "+v"(hit32) : "v"(rh), "v"(ch));
      }
      const u64 fm = __ballot(hit32 == 0u);                // classes without a neighbour of v
      const uint32_t below = __builtin_amdgcn_mbcnt_hi((uint32_t)(fm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)fm, 0u));
      const uint32_t lw = hit32 == 0u ? below : 1u;
      const bool join = lw == 0u;                          // the first free class: exactly one lane, or none (overflow)
      uint32_t* const dst = join ? (L.keys + (c0 + li)) : my_trash;
      *dst = rec;                                          // (class, rank inside the class) of position c0 + li
      rec += join ? 1u : 0u;
      cls[v >> 5] |= join ? (1u << (v & 31u)) : 0u;
    };
    u64 rowA[MWT], rowB[MWT];
    {
      const u64* g = L.adjc + (size_t)rdlane(vmine, 0u) * MWT;
#pragma unroll
      for (uint32_t w = 0; w < MWT; ++w) rowA[w] = g[w];
    }
    uint32_t li = 0;
    for (; li + 2u <= cnt; li += 2u) {                     // two vertices per trip: the row buffers swap roles, nothing is copied
      place(rowA, rowB, li);
      place(rowB, rowA, li + 1u);
    }
    if (li < cnt) place(rowA, rowB, li);
  }
  const uint32_t cnt0 = rec & 0xFFFFu;                     // members of class l
  const uint32_t incl0 = wave_incl_scan(cnt0), total0 = uni(__shfl(incl0, 63));
  if (total0 != r) return false;                           // a vertex found no free class among 64
  const uint32_t base0 = incl0 - cnt0;
  __syncthreads();
  for (uint32_t i = l; i < r + 63u - ((r + 63u) & 63u); i += 64u) {   // whole waves: the shuffles need every lane
    const uint32_t rec = i < r ? L.keys[i] : 0u;
    const uint32_t k = rec >> 16;
    const uint32_t b = __shfl(base0, k & 63u);
    if (i < r) {
      const uint32_t pos = b + (rec & 0xFFFFu);
      L.tmp[pos] = list[i];
      L.C[pos] = k + 1u;
    }
  }
  __syncthreads();
  for (uint32_t i = l; i < r; i += 64u) list[i] = L.tmp[i];
  __syncthreads();
  return true;
}

// ColorSort (maximum_clique.cpp:219-261) on list[0..r), writing colours into the shared array C by absolute
// position (decision D3). With min_k == 1 every vertex joins a class and first-fit colouring in list order
// equals colouring class by class (each class = greedy independent set in list order), which is what the
// bit-parallel loop below does. With min_k >= 2 class 1 is never filled (:242-245), so every vertex gets
// k = 1 < min_k, the order is unchanged and only C[r-1] = 0 is written (:247-248).
// First-fit colouring in list order with one LANE per colour class (graphs of up to 512 vertices).
// ColorSort (maximum_clique.cpp:219-261) gives vertex i the smallest class none of whose members it is adjacent to; lane c
// keeps class c's members as a bitset over GRAPH vertices in MWT registers, so "is v adjacent to a member of class c" is
// (row_v & class_c) != 0 for all classes at once: MWT broadcast LDS reads of row_v (prefetched: the list order is known),
// MWT and-or pairs, one ballot, one find-first-set. Nothing on the critical path waits for LDS, and no position-space
// adjacency has to be built (the class-by-class forms that this replaces paid ~600 cycles per coloured vertex for both).
// A second register set serves classes 64..127 once the first 64 are in use; with more than 128 classes the caller falls
// back to the generic class-by-class loop. Output as ColorSort's: the list regrouped by class (each class in list order),
// C[position] = class.
// WIDE = false: classes 0..63 only (one register set); returns false as soon as a vertex finds all 64 taken, and the caller
// starts over with WIDE = true (two sets, 128 classes). Lists that need more than 64 classes are rare (dense graphs of
// several hundred vertices), so the common loop carries no second set and no range checks.
template <uint32_t MWT, bool WIDE>
__device__ __forceinline__ bool colour_first_fit(const GateLds& L, uint16_t* list, uint32_t r) {
  const uint32_t l = lane_id();
  u64 cls0[MWT], cls1[WIDE ? MWT : 1];
#pragma unroll
  for (uint32_t w = 0; w < MWT; ++w) cls0[w] = 0ull;
#pragma unroll
  for (uint32_t w = 0; w < (WIDE ? MWT : 1); ++w) cls1[w] = 0ull;
  uint32_t cnt0 = 0u, cnt1 = 0u;                           // members of class l / class 64 + l so far
  bool wide = false, overflow = false;                     // wave-uniform
  for (uint32_t c0 = 0; c0 < r; c0 += 64u) {
    const uint32_t cnt = min(64u, r - c0);
    const uint32_t vmine = (c0 + l) < r ? (uint32_t)list[c0 + l] : 0u;
    uint32_t rec = 0u;                                     // lane li: (class << 16 | index inside the class) of position c0 + li
    // one vertex: `row` holds its adjacency row (loaded one vertex ahead), `next` receives the following vertex's
    auto place = [&](const u64 (&row)[MWT], u64 (&next)[MWT], uint32_t li) {
      const uint32_t v = rdlane(vmine, li);
      {
        const u64* g = L.adjc + (size_t)rdlane(vmine, min(li + 1u, cnt - 1u)) * MWT;   // the chunk's last re-reads itself
#pragma unroll
        for (uint32_t w = 0; w < MWT; ++w) next[w] = g[w];
      }
      u64 hit = 0ull;
#pragma unroll
      for (uint32_t w = 0; w < MWT; ++w) hit |= row[w] & cls0[w];
      const u64 free0 = __ballot(hit == 0ull);
      uint32_t k = 0u;
      bool second = false;
      if (free0 != 0ull) {
        k = (uint32_t)__ffsll((long long)free0) - 1u;
      } else if (WIDE) {
        wide = true; second = true;
        u64 hit1 = 0ull;
#pragma unroll
        for (uint32_t w = 0; w < (WIDE ? MWT : 1); ++w) hit1 |= row[w] & cls1[w];
        const u64 free1 = __ballot(hit1 == 0ull);
        if (free1 == 0ull) overflow = true; else k = (uint32_t)__ffsll((long long)free1) - 1u;
      } else {
        overflow = true;
      }
      // vertex v joins class k (of the first or second set): only lane k executes the update; which register pair receives
      // the bit is a scalar branch on the (wave-uniform) word index, so the arrays are only ever indexed statically and stay
      // in registers, and one word is touched instead of all of them
      const uint32_t vw = v >> 6;
      const u64 bit = 1ull << (v & 63u);
      const uint32_t idx = second ? rdlane(cnt1, k) : rdlane(cnt0, k);
      if (l == k) {
        if (!second) {
          cnt0 += 1u;
#pragma unroll
          for (uint32_t w = 0; w < MWT; ++w)
            if (vw == w) { cls0[w] |= bit; asm volatile("" ::: "memory"); }   // (the empty asm keeps this a branch, not MWT selects)
        } else if (WIDE) {
          cnt1 += 1u;
#pragma unroll
          for (uint32_t w = 0; w < (WIDE ? MWT : 1); ++w)
            if (vw == w) { cls1[w] |= bit; asm volatile("" ::: "memory"); }
        }
      }
      if (l == li) rec = ((second ? k + 64u : k) << 16) | idx;
    };
    u64 rowA[MWT], rowB[MWT];
    {
      const u64* g = L.adjc + (size_t)rdlane(vmine, 0u) * MWT;
#pragma unroll
      for (uint32_t w = 0; w < MWT; ++w) rowA[w] = g[w];
    }
    uint32_t li = 0;
    for (; li + 2u <= cnt && !overflow; li += 2u) {        // two vertices per trip: the row buffers swap roles, nothing is copied
      place(rowA, rowB, li);
      place(rowB, rowA, li + 1u);
    }
    if (li < cnt && !overflow) place(rowA, rowB, li);
    if (overflow) return false;                            // nothing has been written to the list or to C
    if (c0 + l < r) L.keys[c0 + l] = rec;
  }
  // class c's block starts after all smaller classes: exclusive prefix of the class sizes over the lanes
  const uint32_t incl0 = wave_incl_scan(cnt0), total0 = uni(__shfl(incl0, 63));
  const uint32_t base0 = incl0 - cnt0;
  uint32_t base1 = 0u;
  if (wide) { const uint32_t incl1 = wave_incl_scan(cnt1); base1 = total0 + incl1 - cnt1; }
  __syncthreads();
  for (uint32_t i = l; i < r + 63u - ((r + 63u) & 63u); i += 64u) {   // whole waves: the shuffles need every lane
    const uint32_t rec = i < r ? L.keys[i] : 0u;
    const uint32_t k = rec >> 16;
    uint32_t b = __shfl(base0, k & 63u);
    if (wide) { const uint32_t b1 = __shfl(base1, k & 63u); b = k >= 64u ? b1 : b; }
    if (i < r) {
      const uint32_t pos = b + (rec & 0xFFFFu);
      L.tmp[pos] = list[i];
      L.C[pos] = k + 1u;
    }
  }
  __syncthreads();
  for (uint32_t i = l; i < r; i += 64u) list[i] = L.tmp[i];
  __syncthreads();
  return true;
}

template <bool kWide>
__device__ __forceinline__ void colour_sort(const GateLds& L, uint16_t* list, uint32_t r, uint32_t MW, uint32_t qmax, uint32_t qsz) {
  const uint32_t l = lane_id();
  const int min_k = max(1, (int)qmax - (int)qsz + 1);
  if (min_k >= 2) {
    if (l == 0) L.C[r - 1] = 0u;
    __syncthreads();
    return;
  }
  if (r <= 2u) {
    // one or two vertices (about half of all calls deep in the tree): the order cannot change; the second vertex opens
    // class 2 iff it is adjacent to the first
    if (l == 0) {
      L.C[0] = 1u;
      if (r == 2u) {
        const uint32_t a = list[0], b = list[1];
        L.C[1] = ((L.adjc[(size_t)a * MW + (b >> 6)] >> (b & 63u)) & 1ull) ? 2u : 1u;
      }
    }
    __syncthreads();
    return;
  }
  if (MW <= 8u) {                                          // graphs of up to 512 vertices: one lane per colour class
    bool done = false;
    switch (MW) {                                          // wave-uniform
      case 1: done = colour_first_fit64<1>(L, list, r) || colour_first_fit<1, true>(L, list, r); break;
      case 2: done = colour_first_fit64<2>(L, list, r) || colour_first_fit<2, true>(L, list, r); break;
      case 3: done = colour_first_fit64<3>(L, list, r) || colour_first_fit<3, true>(L, list, r); break;
      case 4: done = colour_first_fit64<4>(L, list, r) || colour_first_fit<4, true>(L, list, r); break;
      case 5: done = colour_first_fit64<5>(L, list, r) || colour_first_fit<5, true>(L, list, r); break;
      case 6: done = colour_first_fit64<6>(L, list, r) || colour_first_fit<6, true>(L, list, r); break;
      case 7: done = colour_first_fit64<7>(L, list, r) || colour_first_fit<7, true>(L, list, r); break;
      default: done = colour_first_fit64<8>(L, list, r) || colour_first_fit<8, true>(L, list, r); break;
    }
    if (done) return;                                      // else: more than 128 classes -> the generic loop below
  }
  if constexpr (kWide) {
    if (MW > 8u && MW <= 16u) {                            // up to 1024 vertices: the same, with a 32-register class tuple
      bool done = false;
      switch (MW) {                                        // wave-uniform
        case 9: done = colour_first_fit64<9>(L, list, r) || colour_first_fit<9, true>(L, list, r); break;
        case 10: done = colour_first_fit64<10>(L, list, r) || colour_first_fit<10, true>(L, list, r); break;
        case 11: done = colour_first_fit64<11>(L, list, r) || colour_first_fit<11, true>(L, list, r); break;
        case 12: done = colour_first_fit64<12>(L, list, r) || colour_first_fit<12, true>(L, list, r); break;
        case 13: done = colour_first_fit64<13>(L, list, r) || colour_first_fit<13, true>(L, list, r); break;
        case 14: done = colour_first_fit64<14>(L, list, r) || colour_first_fit<14, true>(L, list, r); break;
        case 15: done = colour_first_fit64<15>(L, list, r) || colour_first_fit<15, true>(L, list, r); break;
        default: done = colour_first_fit64<16>(L, list, r) || colour_first_fit<16, true>(L, list, r); break;
      }
      if (done) return;
    }
  }
  // generic class-by-class colouring (graphs beyond 1024 vertices, or more than 128 classes)
  const uint32_t nchunks = (r + 63u) / 64u;
  u64 uncol = 0ull;                                        // lane c holds positions [64c, 64c + 64)
  if (l < nchunks) uncol = (l * 64u + 64u <= r) ? ~0ull : ((1ull << (r - l * 64u)) - 1ull);
  uint32_t k = 1, outpos = 0;
  while (__ballot(uncol != 0ull) != 0ull) {
    u64 Q = uncol;
    while (true) {
      const u64 balQ = __ballot(Q != 0ull);
      if (balQ == 0ull) break;
      const uint32_t ll = (uint32_t)__ffsll((long long)balQ) - 1u;
      const u64 wq = shfl64(Q, ll);
      const uint32_t bit = uni((uint32_t)__ffsll((long long)wq) - 1u);
      const uint32_t g = uni(list[ll * 64u + bit]);
      if (l == 0) { L.tmp[outpos] = (uint16_t)g; L.C[outpos] = k; }
      ++outpos;
      if (l == ll) { uncol &= ~(1ull << bit); Q &= ~(1ull << bit); }
      const u64* grow = L.adjc + (size_t)g * MW;
      for (uint32_t c = 0; c < nchunks; ++c) {
        if (!((balQ >> c) & 1ull)) continue;               // wave-uniform
        const uint32_t pos = c * 64u + l;
        bool adj = false;
        if (pos < r) { const uint32_t h = list[pos]; adj = (grow[h >> 6] >> (h & 63u)) & 1ull; }
        const u64 bal = __ballot(adj);
        if (l == c) Q &= ~bal;                             // neighbours cannot join this class
      }
    }
    ++k;
  }
  __syncthreads();
  for (uint32_t i = l; i < r; i += 64u) list[i] = L.tmp[i];
  __syncthreads();
}

// FindClique + MaxCliqueDyn (maximum_clique.cpp:286-369) as an explicit state machine over one wave.
// Returns QMax.size(); *err != 0 when the per-wave stack is too small.
// vertices: the graph's vertex numbers in ascending order (nullptr: 0 .. m - 1). The search only ever compares vertex numbers and
// uses them as row / bit indices, so a graph whose m vertices keep larger, ascending numbers (all below 64 MW) behaves exactly
// like its renumbered copy. L.deg[i] = degree of the i-th vertex on entry.
// kGate: the caller only asks whether the clique FindClique(minimal_size) returns is LARGER than minimal_size
// (sac_model_registration_graph.h:260-262). FindClique stops at the first leaf with |Q| >= minimal_size, and that leaf's size is
// decided long before it is reached: once Q holds minimal_size vertices and their common neighbourhood Rp is not empty, the
// recursion can only go down -- the child's first candidate always passes |Q| + c > |QMax| (|QMax| < minimal_size <= |Q|, or the
// search had returned), so an (minimal_size + 1)-th vertex is pushed, and from there every path ends in a leaf of at least that
// size before anything is popped. (While |QMax| < minimal_size, that is: after the first such leaf the reference unwinds through
// the ancestors' remaining candidates, which can no longer change QMax -- the gate stops there too.) The only other exit is the step cap (:318), at most |Rp| + 1 steps away: if the cap cannot be
// reached within them the answer is known and the rest of the descent (typically 20-45 more levels, each colouring a list of
// several hundred vertices) is not walked; the returned size is then a lower bound, |Q| + 1. Otherwise: the search as it is.
template <bool kWide, bool kGate>
__device__ __forceinline__ uint32_t clique_search(GateLds L, uint32_t m, uint32_t minimal_size, uint16_t* gstack, uint32_t stack_cap,
                                  int* err, uint32_t* steps_out, uint32_t* prof = nullptr, const uint16_t* vertices = nullptr) {
  const uint32_t l = lane_id();
  const LevelStack stack = {L.lstack, L.lstack_cap, gstack};
  stack_cap += L.lstack_cap;
  const uint32_t MW = (m + 63u) / 64u;
  // R = all vertices, DegreeSort(R); L.deg holds the degree of the i-th vertex at index i
  uint32_t dmax = 0;
  for (uint32_t i = l; i < m; i += 64u) { L.cur[i] = vertices ? vertices[i] : (uint16_t)i; dmax = max(dmax, L.deg[i]); }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) dmax = max(dmax, (uint32_t)__shfl_xor((int)dmax, o));
  const uint32_t max_degree = uni(dmax);                 // = the degree of the sorted list's head (:352-355)
  __syncthreads();
  rank_sort_desc<kWide>(L.cur, L.tmp, L.deg, m, L.keys);
  __syncthreads();
  for (uint32_t i = l; i < m; i += 64u) L.C[i] = i < max_degree ? i + 1u : max_degree + 1u;     // :356-361
  for (uint32_t i = l; i < m + 2u; i += 64u) { L.S[i] = 0u; L.SOld[i] = 0u; }
  for (uint32_t i = l; i < m; i += 64u) stk_put(stack, i, L.cur[i]);
  if (l == 0) { L.lbase[1] = 0u; L.lsize[1] = m; L.lcap[1] = m; }
  __syncthreads();

  uint32_t level = 1, qsz = 0, qmax = 0, top = m;
  int all_steps = 1;
  // optional phase profile (diagnostics): cycles in intersection / degree re-sort / colouring, and their counts
  uint32_t pf_isect = 0, pf_sort = 0, pf_col = 0, pf_vfull = 0, pf_big = 0, pf_vbig = 0;
  // A level's list lives where it will be kept: on the LDS part of the level stack when it fits there whole (the usual case:
  // the lists of one root-to-leaf path add up to a few thousand entries) -- the child is built in place right behind its
  // parent's list, nothing is copied down when the search descends and nothing is restored when it returns. A list that does not
  // fit (the stack continues in global memory) is worked on in one of two LDS buffers and copied to / from the stack as before.
  auto in_lds = [&](uint32_t b, uint32_t n) { return b + n <= stack.lds_cap; };
  uint16_t* cur = in_lds(0u, m) ? stack.lds : L.cur;
  // The current level's frame -- list size, S[level], and where its list sits on the stack -- lives in scalar registers;
  // the LDS arrays are only touched when the level changes, and then all of a frame's words come back in ONE LDS round trip
  // (S, SOld, lbase, lsize, lcap are consecutive arrays of `ma` words: lane j reads array j at [level]). A lone wave pays
  // ~130 cycles per dependent LDS read, and the per-word form of this bookkeeping cost a dozen of them per step.
  const uint32_t ma = (m + 7u) & ~3u;
  auto frame_word = [&](uint32_t lvl) -> uint32_t { return l < 5u ? (L.S + (size_t)l * ma)[lvl] : 0u; };
  uint32_t sz = m, S_cur = 0u, base_cur = 0u, cap_cur = m;     // level 1: S[1] = S[1] + S[0] - SOld[1] = 0, SOld[1] = S[0] = 0 (:300-301)
  while (true) {
    bool ret = false;
    if (sz == 0u) {
      ret = true;                                          // while (!R.empty()) falls through, function returns
    } else {
      const uint32_t pv = cur[sz - 1u];
      const uint32_t cv = top > 0u ? L.C[top - 1u] : 0u;   // C.back(), decision D3
      const uint32_t p = uni(pv), c = uni(cv);
      if (qsz + c > qmax) {                                // :307
        ++qsz;                                             // Q.push_back(p)
        // Intersection(p, R, Rp), :209-217 -- order preserving compaction
        const u64* prow = L.adjc + (size_t)p * MW;
        const long long pt0 = prof ? clock64() : 0;
        const uint32_t nb = base_cur + cap_cur;            // where the child's list goes on the stack
        const bool in_place = in_lds(nb, sz);              // rp <= sz
        uint16_t* const nxt = in_place ? stack.lds + nb : (cur == L.cur ? L.nxt : L.cur);
        uint32_t rp = 0;
        for (uint32_t i0 = 0; i0 < sz; i0 += 64u) {
          const uint32_t i = i0 + l;
          uint32_t h = 0;
          bool adj = false;
          if (i < sz) { h = cur[i]; adj = (prow[h >> 6] >> (h & 63u)) & 1ull; }
          const u64 bal = __ballot(adj);
          if (adj) nxt[rp + (uint32_t)__popcll(bal & ((1ull << l) - 1ull))] = (uint16_t)h;
          rp += (uint32_t)__popcll(bal);
        }
        rp = uni(rp);
        __syncthreads();
        const long long pt1 = prof ? clock64() : 0;
        pf_isect += (uint32_t)(pt1 - pt0);
        if constexpr (kGate) {
          if (rp > 0u && qmax < minimal_size && qsz >= minimal_size && (uint32_t)all_steps + rp + 1u <= (uint32_t)kStepCap) { qmax = qsz + 1u; break; }
        }
        if (rp > 0u) {
          // :313 is (double)S[level] / all_steps_ < 0.025. With all_steps <= 100001 a quotient other than 1/40
          // differs from 1/40 by more than 1e-7, and 1/40 itself rounds to the literal: the test is 40 S < all_steps
          if ((uint64_t)S_cur * 40ull < (uint64_t)all_steps) {
            degrees_in_list(L, nxt, rp, MW);
            rank_sort_desc<kWide>(nxt, L.tmp, L.deg, rp, L.keys);
          }
          const long long pt2 = prof ? clock64() : 0;
          pf_sort += (uint32_t)(pt2 - pt1);
          colour_sort<kWide>(L, nxt, rp, MW, qmax, qsz);
          if (prof) {
            const uint32_t dt = (uint32_t)(clock64() - pt2);
            pf_col += dt;
            if ((int)qmax - (int)qsz + 1 < 2) { pf_vfull += rp; if (rp > 64u) { pf_big += dt; pf_vbig += rp; } }
          }
          S_cur += 1u;
          ++all_steps;
          if (all_steps > kStepCap) {
            ret = true;                                    // :318-319: returns without popping Q
          } else {
            if (nb + rp > stack_cap) { *err = 1; break; }
            if (!in_place)
              for (uint32_t i = l; i < rp; i += 64u) stk_put(stack, nb + i, nxt[i]);
            // leave this level: its frame goes to LDS; read the child's S / SOld in the same round trip
            const uint32_t child = frame_word(level + 1u);
            if (l == 0) { L.S[level] = S_cur; L.lsize[level] = sz; L.lbase[level] = base_cur; L.lcap[level] = cap_cur; }
            const uint32_t s_child = rdlane(child, 0u), sold_child = rdlane(child, 1u);
            ++level;
            cur = nxt;
            if (qmax >= minimal_size) {                    // :290-291 at the entry of the child: it returns at once; its S and
              // SOld stay as they were. Its frame must still be readable when the common return path below stores S
              if (l == 0) L.S[level] = s_child;
              S_cur = s_child; sz = rp; base_cur = nb; cap_cur = rp;
              ret = true;
            } else {
              if (l == 0) L.SOld[level] = S_cur;           // :300-301: S[level] += S[level - 1] - SOld[level]; SOld[level] = S[level - 1]
              S_cur = s_child + S_cur - sold_child;
              sz = rp; base_cur = nb; cap_cur = rp;
              __syncthreads();
              continue;
            }
          }
        } else {
          if (qsz > qmax) {                                // :322-326
            qmax = qsz;
            if (qmax >= minimal_size) {
              // (what follows in the reference is the unwinding: every ancestor still expands its remaining candidates, whose
              // children return at once (:290); a leaf there has |Q| < |QMax|, so QMax is final -- the gate needs no more)
              if constexpr (kGate) break;
              ret = true;
            }
          }
          if (!ret) --qsz;                                 // Q.pop_back(), :329
        }
      } else {
        ret = true;                                        // :331-332
      }
      if (!ret) {                                          // R.pop_back(); C.pop_back(), :333-334
        --sz;
        if (top > 0u) --top;
        continue;
      }
    }
    // the current level's function returns; its caller continues after the recursive call (:320)
    if (level == 1u) break;
    if (l == 0) L.S[level] = S_cur;                        // a later sibling re-enters this level and reads it (:300)
    --level;
    --qsz;                                                 // Q.pop_back()
    if (top > 0u) --top;                                   // C.pop_back()
    __syncthreads();
    const uint32_t fw = frame_word(level);                 // S, -, lbase, lsize, lcap of the caller: one round trip
    S_cur = rdlane(fw, 0u); base_cur = rdlane(fw, 2u); sz = rdlane(fw, 3u) - 1u; cap_cur = rdlane(fw, 4u);   // R.pop_back()
    if (in_lds(base_cur, cap_cur)) {
      cur = stack.lds + base_cur;                          // the caller's list is where it was built
    } else {
      cur = L.cur;
      for (uint32_t i = l; i < sz; i += 64u) cur[i] = stk_get(stack, base_cur + i);
    }
    __syncthreads();
  }
  if (steps_out) *steps_out = (uint32_t)all_steps;
  if (prof && l == 0) { prof[0] = pf_isect; prof[1] = pf_sort; prof[2] = pf_col; prof[3] = pf_big; prof[4] = pf_vbig; prof[5] = pf_vfull; }
  return qmax;
}

constexpr uint32_t kAdjcScratchWords = 256u * 1024u;   // 2 MB per deferred hypothesis: m * ceil(m/64) <= 262144 -> m <= 4064

struct EvalArgs {
  ObjJob job;
  const uint32_t* iter_samples;   // 3 per iteration
  uint32_t it_begin, it_end;      // iterations of this batch
  int32_t* counts;                // consensus size per iteration (0 = rejected by the gate)
  uint32_t* gate_m;               // per iteration: |F| when the gate ran (diagnostics), else 0
  uint32_t* work;                 // atomic work counter (zeroed by the host)
  uint32_t* status;               // [0] error flag, [1] gate calls, [2] deferred count
  uint32_t* deferred;             // iteration indices that need the big-LDS pass
  uint16_t* stacks;               // per resident wave: stack_cap entries
  uint32_t stack_cap;
  uint32_t lds_bytes;
  uint32_t from_deferred;         // 1: the work list is `deferred`
  uint32_t n_deferred;
  u64* adjc_scratch;              // deferred pass only: kAdjcScratchWords u64 per block for graphs beyond the LDS
  uint32_t* dbg;                  // optional: per iteration dbg_stride words {cnt, m, F members...}
  uint32_t dbg_stride;
  uint32_t stop_level;            // 0 = full evaluation, 1 = stop before the clique search (diagnostics)
  const uint32_t* n_items_dev;    // optional: only the first *n_items_dev items exist (an evaluation launched in the tick of the walk
                                  // that draws its iterations: ChainOut::n_done)
};

// The gate of one hypothesis (sac_model_registration_graph.h:219-265): induced sample sub-graph of F, degree test,
// maximum clique. Returns the consensus count to report (cnt, 0 = rejected, INT_MIN = error).
template <bool kExt, bool kWide>
__device__ __forceinline__ int32_t gate_eval(const EvalArgs& A, const WaveBits& F, uint32_t m, uint32_t it, uint32_t cnt,
                                             unsigned char* lds_raw, uint16_t* stack) {
  const uint32_t l = lane_id();
  const ObjJob& job = A.job;
  const uint32_t W = job.W;
  int32_t result = (int32_t)cnt;
  const uint32_t MW = (m + 63u) / 64u;
  // An object of up to 1024 matches whose consensus list needs as many 64-bit words as the object itself (MW == W: the usual
  // case when the object is really there) keeps the object's vertex numbers: the induced graph is then the object's sample rows
  // masked with F, a copy, instead of a column compaction that costs ~200 k cycles for 264 vertices; vertex numbers only ever
  // serve as row / bit indices and in comparisons, and F is ascending, so the search cannot tell the difference.
  const bool ident = !kExt && W <= (kWide ? 16u : 8u) && MW == W && gate_lds_bytes(m) + 8u * (job.n - m) * MW <= A.lds_bytes;
  GateLds L = gate_carve<kExt>(lds_raw, m, A.lds_bytes, kExt ? A.adjc_scratch + (size_t)blockIdx.x * kAdjcScratchWords : nullptr,
                               ident ? job.n : m);
  const long long t_start = A.dbg ? clock64() : 0;   // phase stamps (diagnostics builds of the call only)
  // F in ascending order (:219) -> graph index = rank (:241-243)
  uint32_t base = 0;
#pragma unroll
  for (int j = 0; j < kWPL; ++j) {
    const uint32_t c = (uint32_t)__popcll(F.w[j]);
    const uint32_t incl = wave_incl_scan(c);
    u64 w = F.w[j];
    uint32_t o = base + incl - c;
    while (w) {
      const uint32_t bit = (uint32_t)__ffsll((long long)w) - 1u;
      L.flist[o++] = (uint16_t)((j * 64u + l) * 64u + bit);
      w &= w - 1ull;
    }
    base += uni(__shfl(incl, 63));
  }
  __syncthreads();
  // induced sample sub-graph (:245-256) as an m x MW bit matrix in LDS, plus vertex degrees
  const long long t_flist = A.dbg ? clock64() : 0;
  bool bad_index = false;
  for (uint32_t g = l; g < m; g += 64u) bad_index = bad_index || L.flist[g] >= job.n;
  bad_index = __ballot(bad_index) != 0ull;           // never dereference an unchecked index
  if (bad_index && l == 0) { atomicExch(&A.status[0], 4u); A.status[6] = m; A.status[7] = it; }
  if (!bad_index && A.stop_level != 3u) {
    if (ident) {
      if (l < W) L.mask[l] = F.w[0];                       // lane l holds word l of F (W <= 16 < 64)
      __syncthreads();
      for (uint32_t i = l; i < job.n * W; i += 64u) {
        const uint32_t v = i / W, w = i - v * W;
        const bool member = (L.mask[v >> 6] >> (v & 63u)) & 1ull;
        L.adjc[i] = member ? (job.samp[i] & L.mask[w]) : 0ull;
      }
      __syncthreads();
      for (uint32_t g = l; g < m; g += 64u) {
        const u64* row = L.adjc + (size_t)L.flist[g] * MW;
        uint32_t d = 0;
        for (uint32_t w = 0; w < MW; ++w) d += (uint32_t)__popcll(row[w]);
        L.deg[g] = d;
      }
    } else if (W <= 8u) {
      // n <= 512: lane = one row of the induced graph, its whole sample row (<= 16 dwords) in registers;
      // the members of F are walked once per 64 rows, one v_readlane + bit-field extract + shift-or each.
      // F is ascending, so the source dword only ever moves forward.
      uint32_t fl[kRegChunks];
#pragma unroll
      for (uint32_t c = 0; c < kRegChunks; ++c) fl[c] = (c * 64u + l) < m ? L.flist[c * 64u + l] : 0u;
      for (uint32_t rc = 0; rc < MW; ++rc) {
        const uint32_t grow = rc * 64u + l;
        const bool have = grow < m;
        const uint32_t myv = have ? (uint32_t)L.flist[grow] : 0u;
        uint32_t rw[16];
        {
          const uint32_t* src = reinterpret_cast<const uint32_t*>(job.samp + (size_t)myv * W);
#pragma unroll
          for (uint32_t w = 0; w < 16u; ++w) rw[w] = (have && w < 2u * W) ? src[w] : 0u;
        }
        uint32_t out[2u * kRegChunks];
#pragma unroll
        for (uint32_t c = 0; c < 2u * kRegChunks; ++c) out[c] = 0u;
        uint32_t wcur = 0xFFFFFFFFu, word = 0u;
#pragma unroll
        for (uint32_t cj = 0; cj < 2u * kRegChunks; ++cj) {      // 32 positions per step: the target dword is static
          if (cj * 32u < m) {                                     // wave-uniform
            const uint32_t cnt = min(32u, m - cj * 32u);
            for (uint32_t lj = 0; lj < cnt; ++lj) {
              const uint32_t h = rdlane(fl[cj >> 1], (cj & 1u) * 32u + lj);
              if ((h >> 5) != wcur) {                             // wave-uniform, at most 2 W times per 64 rows
                wcur = h >> 5;
#pragma unroll
                for (uint32_t w = 0; w < 16u; ++w) if (wcur == w) word = rw[w];
              }
              out[cj] |= ((word >> (h & 31u)) & 1u) << lj;
            }
          }
        }
        if (have) {
          uint32_t d = 0;
#pragma unroll
          for (uint32_t c = 0; c < kRegChunks; ++c) {
            if (c < MW) {
              const u64 wv = ((u64)out[2u * c + 1u] << 32) | out[2u * c];
              L.adjc[(size_t)grow * MW + c] = wv;
              d += (uint32_t)__popcll(wv);
            }
          }
          L.deg[grow] = d;
        }
      }
    } else if (W <= 64u) {
      // lane l holds word l of a row; kRows rows are in flight so the global latency is paid once per group
      constexpr uint32_t kRows = 8;
      for (uint32_t g0 = 0; g0 < m; g0 += kRows) {
        u64 rw[kRows];
#pragma unroll
        for (uint32_t j = 0; j < kRows; ++j) {
          const uint32_t g = g0 + j;
          rw[j] = (g < m && l < W) ? job.samp[(size_t)uni(L.flist[g < m ? g : 0u]) * W + l] : 0ull;
        }
#pragma unroll
        for (uint32_t j = 0; j < kRows; ++j) {
          const uint32_t g = g0 + j;
          if (g < m) {                               // wave-uniform
            uint32_t d = 0;
            for (uint32_t c = 0; c < MW; ++c) {
              const uint32_t pos = c * 64u + l;
              const uint32_t h = pos < m ? L.flist[pos] : 0u;
              const bool adj = row_test(rw[j], h) && pos < m;
              const u64 bal = __ballot(adj);
              if (l == 0) L.adjc[(size_t)g * MW + c] = bal;
              d += (uint32_t)__popcll(bal);
            }
            if (l == 0) L.deg[g] = d;
          }
        }
      }
    } else {
      for (uint32_t g = 0; g < m; ++g) {
        const u64* row = job.samp + (size_t)uni(L.flist[g]) * W;
        uint32_t d = 0;
        for (uint32_t c = 0; c < MW; ++c) {
          const uint32_t pos = c * 64u + l;
          bool adj = false;
          if (pos < m) { const uint32_t h = L.flist[pos]; adj = (row[h >> 6] >> (h & 63u)) & 1ull; }
          const u64 bal = __ballot(adj);
          if (l == 0) L.adjc[(size_t)g * MW + c] = bal;
          d += (uint32_t)__popcll(bal);
        }
        if (l == 0) L.deg[g] = d;
      }
    }
  }
  __syncthreads();
  const long long t_adjc = A.dbg ? clock64() : 0;
  if (A.dbg) {
    uint32_t* d = A.dbg + (size_t)it * A.dbg_stride;
    if (l == 0) { d[0] = cnt; d[1] = m; }
    for (uint32_t g = l; g < m && 2u + 2u * g + 1u < A.dbg_stride; g += 64u) { d[2 + 2 * g] = L.flist[g]; d[3 + 2 * g] = L.deg[g]; }
  }
  // "make sure that those inliers have enough neighbors within the inliers themselves" (:221-238)
  bool any = false;
  for (uint32_t g = l; g < m; g += 64u) any = any || L.deg[g] > kGateMinimal;
  if (bad_index) {
    result = INT_MIN;
  } else if (A.stop_level != 0u) {
    result = -(int32_t)m;
  } else if (__ballot(any) == 0ull) {
    result = 0;
  } else {
    int err = 0;
    uint32_t steps = 0;
    uint32_t* prof = (A.dbg && A.dbg_stride >= 16u) ? A.dbg + (size_t)it * A.dbg_stride + (A.dbg_stride - 12u) : nullptr;
    const uint32_t q = clique_search<kWide, true>(L, m, kGateMinimal, stack, A.stack_cap, &err, &steps, prof, ident ? L.flist : nullptr);
    if (A.dbg && l == 0 && A.dbg_stride >= 8u) {
      uint32_t* d = A.dbg + (size_t)it * A.dbg_stride + (A.dbg_stride - 6u);
      d[0] = (uint32_t)(t_flist - t_start); d[1] = (uint32_t)(t_adjc - t_flist);
      d[2] = (uint32_t)(clock64() - t_adjc); d[3] = steps; d[4] = q;
    }
    if (err) {
      if (l == 0) atomicExch(&A.status[0], 1u);
      result = INT_MIN;
    } else if (q <= kGateMinimal) {
      result = 0;                                    // :260-265
    }
    if (l == 0) atomicAdd(&A.status[1], 1u);
  }
  __syncthreads();
  return result;
}

// launched with 64 threads; the bound is deliberately larger so that hipcc keeps __syncthreads() as a real,
// convergent s_barrier (with a 64-thread bound it drops the barrier and may split the lanes of the wave)
// kWide = false: objects of up to 512 matches (job.W <= 8), the launch's every slot; true: any size
template <bool kWide, class H = Slots<EvalArgs>>
__global__ __launch_bounds__(128) void eval_kernel(H SL) {
  TOD_LATENCY_PRIO();   // latency-bound: win issue arbitration against the VALU-saturating matcher
  const EvalArgs& A = SL.a[blockIdx.y];
  extern __shared__ __align__(16) unsigned char lds_raw[];
  const uint32_t l = lane_id();
  const ObjJob& job = A.job;
  const uint32_t W = job.W;
  uint16_t* stack = A.stacks + (size_t)blockIdx.x * A.stack_cap;
  uint32_t n_items = A.from_deferred ? A.n_deferred : (A.it_end - A.it_begin);
  if (A.n_items_dev) n_items = min(n_items, uni(*A.n_items_dev));
  {                                                        // one block = one hypothesis
    const uint32_t item = blockIdx.x;
    if (item >= n_items) return;
    const uint32_t it = uni(A.from_deferred ? A.deferred[item] : (A.it_begin + item));
    const uint32_t s0 = uni(A.iter_samples[3 * it]), s1 = uni(A.iter_samples[3 * it + 1]),
                   s2 = uni(A.iter_samples[3 * it + 2]);
    if (s0 >= job.n || s1 >= job.n || s2 >= job.n) {       // corrupt draw table: report, never dereference
      if (l == 0) { atomicExch(&A.status[0], 3u); A.counts[it] = INT_MIN; }
      return;
    }
    // common physical neighbours of the three samples (:178-184); the geometric test of :197 is
    // `finite < +inf` because threshold_ is DBL_MAX (D2), i.e. a finiteness test
    WaveBits P;
    wb_load(P, job.phys + (size_t)s0 * W, W);
    wb_and(P, job.phys + (size_t)s1 * W, W);
    wb_and(P, job.phys + (size_t)s2 * W, W);
    wb_and(P, job.valid, W);
    wb_and(P, job.finite, W);
    const uint32_t cnt = wb_count(P) + 3u;                 // + the samples themselves (:185-186)
    int32_t result = (int32_t)cnt;
    uint32_t m_diag = 0;
    if (cnt > kGateMinimal && A.stop_level != 2u) {        // :203-205
      WaveBits F = P;
      wb_set(F, s0); wb_set(F, s1); wb_set(F, s2);
      wb_and(F, job.deg7, W);                              // :211-213
      const uint32_t m = wb_count(F);
      m_diag = m;
      if (m <= kGateMinimal) {
        result = 0;                                        // :214-218
      } else if (gate_lds_bytes(m) > A.lds_bytes &&
                 !(A.from_deferred && A.adjc_scratch && gate_small_bytes(m) <= A.lds_bytes &&
                   m * ((m + 63u) / 64u) <= kAdjcScratchWords)) {
        if (A.from_deferred) {
          if (l == 0) atomicExch(&A.status[0], 2u);        // graph too large even with the adjacency in global memory
          result = INT_MIN;
        } else {
          if (l == 0) A.deferred[atomicAdd(&A.status[2], 1u)] = it;
          result = INT_MIN + 1;                            // filled in by the second pass
        }
      } else {
        // third tier: adjacency matrix in global scratch
        result = gate_lds_bytes(m) > A.lds_bytes ? gate_eval<true, kWide>(A, F, m, it, cnt, lds_raw, stack)
                                                 : gate_eval<false, kWide>(A, F, m, it, cnt, lds_raw, stack);
      }
    }
    if (l == 0) {
      if (result != INT_MIN + 1) A.counts[it] = result;
      if (A.gate_m) A.gate_m[it] = m_diag;
    }
  }
}

// stand-alone clique search on an explicit graph (the reference's test/test_maximum_clique.cpp shape):
// adj = m x MW bit matrix in global memory. One block of 64 threads.
template <bool kGate>
__global__ __launch_bounds__(128) void clique_test_kernel(const u64* adj, uint32_t m, uint32_t minimal_size,
                                                         uint16_t* stack, uint32_t stack_cap, uint32_t lds_bytes,
                                                         uint32_t* out) {
  extern __shared__ __align__(16) unsigned char lds_raw[];
  const uint32_t l = lane_id();
  GateLds L = gate_carve<false>(lds_raw, m, lds_bytes);
  const uint32_t MW = (m + 63u) / 64u;
  for (uint32_t i = l; i < m * MW; i += 64u) L.adjc[i] = adj[i];
  __syncthreads();
  for (uint32_t g = l; g < m; g += 64u) {
    uint32_t d = 0;
    for (uint32_t w = 0; w < MW; ++w) d += (uint32_t)__popcll(L.adjc[(size_t)g * MW + w]);
    L.deg[g] = d;
  }
  __syncthreads();
  int err = 0;
  uint32_t steps = 0;
  const uint32_t q = clique_search<true, kGate>(L, m, minimal_size, stack, stack_cap, &err, &steps);
  if (l == 0) { out[0] = q; out[1] = (uint32_t)err; out[2] = steps; }
}

// ------------------------------------------------------------------------------------------------ K9
// 3x3 one-sided Jacobi SVD in float, A = U diag(w) Vt, w descending. cv::SVD on a CV_32F 3x3
// (sac_model_registration_graph.h:333) is third-party arithmetic that the reference tree does not contain.
__device__ void svd3(const float Ain[3][3], float U[3][3], float w[3], float Vt[3][3]) {
  float A[3][3], V[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) A[i][j] = Ain[i][j];
  for (int sweep = 0; sweep < 30; ++sweep) {
    bool rotated = false;
    for (int p = 0; p < 2; ++p)
      for (int q = p + 1; q < 3; ++q) {
        float alpha = 0, beta = 0, gamma = 0;
        for (int i = 0; i < 3; ++i) { alpha += A[i][p] * A[i][p]; beta += A[i][q] * A[i][q]; gamma += A[i][p] * A[i][q]; }
        if (fabsf(gamma) <= 1.1920929e-07f * sqrtf(alpha * beta) || gamma == 0.f) continue;
        rotated = true;
        const float zeta = (beta - alpha) / (2.f * gamma);
        const float t = (zeta >= 0.f ? 1.f : -1.f) / (fabsf(zeta) + sqrtf(1.f + zeta * zeta));
        const float c = 1.f / sqrtf(1.f + t * t), s = c * t;
        for (int i = 0; i < 3; ++i) {
          const float ap = A[i][p], aq = A[i][q];
          A[i][p] = c * ap - s * aq; A[i][q] = s * ap + c * aq;
          const float vp = V[i][p], vq = V[i][q];
          V[i][p] = c * vp - s * vq; V[i][q] = s * vp + c * vq;
        }
      }
    if (!rotated) break;
  }
  float nrm[3];
  int order[3] = {0, 1, 2};
  for (int j = 0; j < 3; ++j) nrm[j] = sqrtf(A[0][j] * A[0][j] + A[1][j] * A[1][j] + A[2][j] * A[2][j]);
  for (int a = 0; a < 2; ++a)
    for (int b = 0; b < 2 - a; ++b)
      if (nrm[order[b]] < nrm[order[b + 1]]) { int t = order[b]; order[b] = order[b + 1]; order[b + 1] = t; }
  for (int jj = 0; jj < 3; ++jj) {
    const int j = order[jj];
    w[jj] = nrm[j];
    for (int i = 0; i < 3; ++i) { Vt[jj][i] = V[i][j]; U[i][jj] = nrm[j] > 0.f ? A[i][j] / nrm[j] : 0.f; }
  }
  const float tiny = 1.1920929e-07f * (w[0] > 0.f ? w[0] : 1.f);
  if (w[0] <= 0.f) { for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) U[i][j] = (i == j) ? 1.f : 0.f; return; }
  if (w[1] <= tiny) {
    int ax = 0;
    for (int i = 1; i < 3; ++i) if (fabsf(U[i][0]) < fabsf(U[ax][0])) ax = i;
    float e[3] = {0, 0, 0};
    e[ax] = 1.f;
    float c1[3] = {U[1][0] * e[2] - U[2][0] * e[1], U[2][0] * e[0] - U[0][0] * e[2], U[0][0] * e[1] - U[1][0] * e[0]};
    const float n1 = sqrtf(c1[0] * c1[0] + c1[1] * c1[1] + c1[2] * c1[2]);
    for (int i = 0; i < 3; ++i) U[i][1] = c1[i] / n1;
  }
  if (w[2] <= tiny) {
    U[0][2] = U[1][0] * U[2][1] - U[2][0] * U[1][1];
    U[1][2] = U[2][0] * U[0][1] - U[0][0] * U[2][1];
    U[2][2] = U[0][0] * U[1][1] - U[1][0] * U[0][1];
  }
}
__device__ inline float det3f(const float m[3][3]) {
  return m[0][0] * (m[1][1] * m[2][2] - m[1][2] * m[2][1]) - m[0][1] * (m[1][0] * m[2][2] - m[1][2] * m[2][0]) +
         m[0][2] * (m[1][0] * m[2][1] - m[1][1] * m[2][0]);
}

// The serial tail of estimateRigidTransformationSVD (sac_model_registration_graph.h:330-346): H (double sums, rounded to float), SVD,
// reflection fix, R = U Vt (double accumulation), T = c_train - R c_query. C = {c_train, c_query}. One lane's work; shared by the
// block form (growth_kernel) and the single-wave form (sprint_kernel) so that both execute the same arithmetic.
__device__ inline void kabsch_solve(const double Hd[9], const float C[6], float R[9], float T[3]) {
  float H[3][3], U[3][3], wv[3], Vt[3][3], Rm[3][3];
  for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) H[r][c] = (float)Hd[3 * r + c];
  svd3(H, U, wv, Vt);
  if (det3f(U) * det3f(Vt) < 0)
    for (int x = 0; x < 3; ++x) Vt[2][x] *= -1;
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) {
      double s = 0;
      for (int k = 0; k < 3; ++k) s += (double)U[r][k] * (double)Vt[k][c];
      Rm[r][c] = (float)s;
    }
  for (int r = 0; r < 3; ++r) {
    float s = 0;
    for (int k = 0; k < 3; ++k) s += Rm[r][k] * C[3 + k];
    T[r] = C[r] - s;
    for (int c = 0; c < 3; ++c) R[3 * r + c] = Rm[r][c];
  }
}
// adjacency_ransac.cpp:275-283: norm(R q + T - t)^2 < thresh, the norm in double
__device__ __forceinline__ bool growth_admits(const float R[9], const float T[3], const float* q, const float* t, double thresh) {
  float p[3];
  for (int r = 0; r < 3; ++r) {
    float s = 0;
    for (int k = 0; k < 3; ++k) s += R[3 * r + k] * q[k];
    p[r] = s + T[r];
  }
  const double nn = norm3d(p[0] - t[0], p[1] - t[1], p[2] - t[2]);
  return nn * nn < thresh;
}
// adjacency_ransac.cpp:304-305: R = R^T, T = -R T
__device__ inline void pose_invert(const float R[9], const float T[3], float Rout[9], float Tout[3]) {
  float Rt[3][3];
  for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) Rt[r][c] = R[3 * c + r];
  for (int r = 0; r < 3; ++r) {
    float s = 0;
    for (int k = 0; k < 3; ++k) s += (-Rt[r][k]) * T[k];
    Tout[r] = s;
    for (int c = 0; c < 3; ++c) Rout[3 * r + c] = Rt[r][c];
  }
}

constexpr uint32_t kGrowthLdsPoints = 2048;   // inlier points staged in LDS per Kabsch pass (48 KB)

struct GrowthOut {
  float R[9], T[3];             // inverted pose (training -> camera), adjacency_ransac.cpp:304-305
  uint32_t n_match_inliers;     // match indices in the grown set
  uint32_t n_kp_inliers;        // unique keypoint indices (:306-308)
  uint32_t passes;
  uint32_t n_model_inliers;
};

// One block. inl/rest/extra are W-word bitsets in global scratch. Sums that the reference accumulates
// sequentially (centroids in float, the correlation matrix in double) are accumulated sequentially here too,
// each by one lane, so that the admitted sets are reproducible bit for bit against a sequential CPU evaluation.
struct GrowthArgs {
  ObjJob job; const uint32_t* triple; float err;          // triple: the winning iteration's samples (device)
  u64 *inl, *rest, *extra; uint32_t* kp_list; u64* kp_bits; uint32_t kp_words; GrowthOut* out;
};
__global__ __launch_bounds__(256) void growth_kernel(Slots<GrowthArgs> SL) {
  TOD_LATENCY_PRIO();   // latency-bound: win issue arbitration against the VALU-saturating matcher
  const GrowthArgs& ga = SL.a[blockIdx.x];
  const ObjJob& job = ga.job;
  const uint32_t s0 = ga.triple[0], s1 = ga.triple[1], s2 = ga.triple[2];
  const float err = ga.err;
  u64* const inl = ga.inl; u64* const rest = ga.rest; u64* const extra = ga.extra;
  uint32_t* const kp_list = ga.kp_list; u64* const kp_bits = ga.kp_bits; const uint32_t kp_words = ga.kp_words;
  GrowthOut* const out = ga.out;
  __shared__ float sR[9], sT[3];
  __shared__ double sAcc[16];
  __shared__ float sC[6];
  __shared__ uint32_t sFlag, sCount;
  __shared__ uint32_t sPre[kMaxWords];
  __shared__ float sPts[kGrowthLdsPoints * 6];
  const uint32_t tid = threadIdx.x, W = job.W, n = job.n;
  // consensus set of the winning iteration: common physical neighbours + the samples
  for (uint32_t w = tid; w < W; w += 256u) {
    u64 v = job.phys[(size_t)s0 * W + w] & job.phys[(size_t)s1 * W + w] & job.phys[(size_t)s2 * W + w] &
            job.valid[w] & job.finite[w];
    if ((s0 >> 6) == w) v |= 1ull << (s0 & 63u);
    if ((s1 >> 6) == w) v |= 1ull << (s1 & 63u);
    if ((s2 >> 6) == w) v |= 1ull << (s2 & 63u);
    inl[w] = v;
    rest[w] = job.valid[w] & ~v;                           // :260-264
  }
  for (uint32_t w = tid; w < kp_words; w += 256u) kp_bits[w] = 0ull;
  __syncthreads();
  if (tid == 0) {
    uint32_t c = 0;
    for (uint32_t w = 0; w < W; ++w) c += (uint32_t)__popcll(inl[w]);
    out->n_model_inliers = c;
  }
  bool do_final = false;
  double thresh = (double)(err * err);                     // float product widened, :267
  uint32_t passes = 0;
  while (true) {
    // ---- estimateRigidTransformationSVD (sac_model_registration_graph.h:304-347) on the current inliers
    // ordered compaction of the inlier points into LDS (ascending match index = the reference's list order)
    for (uint32_t w = tid; w < W; w += 256u) sPre[w] = (uint32_t)__popcll(inl[w]);
    __syncthreads();
    if (tid == 0) {
      uint32_t acc = 0;
      for (uint32_t w = 0; w < W; ++w) { const uint32_t c = sPre[w]; sPre[w] = acc; acc += c; }
      sCount = acc;
    }
    __syncthreads();
    const uint32_t cnt = sCount;
    const bool staged = cnt <= kGrowthLdsPoints;
    if (staged) {                                          // one thread per match: its slot = inliers below it
      for (uint32_t v = tid; v < W * 64u; v += 256u) {
        const u64 bits = inl[v >> 6];
        if ((bits >> (v & 63u)) & 1ull) {
          const uint32_t o = sPre[v >> 6] + (uint32_t)__popcll(bits & ((1ull << (v & 63u)) - 1ull));
          for (int c = 0; c < 3; ++c) { sPts[o * 6u + c] = job.train[3 * v + c]; sPts[o * 6u + 3 + c] = job.query[3 * v + c]; }
        }
      }
    }
    __syncthreads();
    if (tid < 6) {                                         // 6 sequential float sums: centroids
      float s = 0.f;
      if (staged) {
        // 8 LDS reads in flight at once; the additions stay in list order
        uint32_t i = 0;
        for (; i + 8u <= cnt; i += 8u) {
          float v[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] = sPts[(i + j) * 6u + tid];
#pragma unroll
          for (int j = 0; j < 8; ++j) s += v[j];
        }
        for (; i < cnt; ++i) s += sPts[i * 6u + tid];
      } else {
        const float* src = tid < 3 ? job.train : job.query;
        const uint32_t c = tid % 3u;
        for (uint32_t w = 0; w < W; ++w) {
          u64 bits = inl[w];
          while (bits) {
            const uint32_t v = w * 64u + (uint32_t)__ffsll((long long)bits) - 1u;
            s += src[3 * v + c];
            bits &= bits - 1ull;
          }
        }
      }
      const double inv = 1. / (float)cnt;                  // Vec /= float: times the double reciprocal
      sC[tid] = (float)(s * inv);
    }
    __syncthreads();
    if (tid < 9) {                                         // H = sub_training^T * sub_query, double accumulation
      const uint32_t r = tid / 3u, c = tid % 3u;
      const float ct = sC[r], cq = sC[3 + c];
      double h = 0.0;
      if (staged) {
        uint32_t i = 0;
        for (; i + 8u <= cnt; i += 8u) {
          float va[8], vb[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) { va[j] = sPts[(i + j) * 6u + r]; vb[j] = sPts[(i + j) * 6u + 3 + c]; }
#pragma unroll
          for (int j = 0; j < 8; ++j) h += (double)(va[j] - ct) * (double)(vb[j] - cq);
        }
        for (; i < cnt; ++i) {
          const float a = sPts[i * 6u + r] - ct, b = sPts[i * 6u + 3 + c] - cq;
          h += (double)a * (double)b;
        }
      } else {
        for (uint32_t w = 0; w < W; ++w) {
          u64 bits = inl[w];
          while (bits) {
            const uint32_t v = w * 64u + (uint32_t)__ffsll((long long)bits) - 1u;
            const float a = job.train[3 * v + r] - ct, b = job.query[3 * v + c] - cq;
            h += (double)a * (double)b;
            bits &= bits - 1ull;
          }
        }
      }
      sAcc[tid] = h;
    }
    __syncthreads();
    if (tid == 0) {
      kabsch_solve(sAcc, sC, sR, sT);
      sFlag = 0u;
    }
    __syncthreads();
    ++passes;
    // ---- admit every valid non-inlier within thresh (adjacency_ransac.cpp:275-283)
    for (uint32_t w0 = 0; w0 < W; w0 += 4u) {
      const uint32_t w = w0 + (tid >> 6);
      bool pass = false;
      if (w < W) {
        const uint32_t v = w * 64u + (tid & 63u);
        if (v < n && ((rest[w] >> (v & 63u)) & 1ull)) {
          pass = growth_admits(sR, sT, job.query + 3 * v, job.train + 3 * v, thresh);
        }
      }
      const u64 bal = __ballot(pass);
      if ((tid & 63u) == 0 && w < W) {
        extra[w] = bal;
        if (bal) atomicOr(&sFlag, 1u);
      }
    }
    __syncthreads();
    for (uint32_t w = tid; w < W; w += 256u) { inl[w] |= extra[w]; rest[w] &= ~extra[w]; }
    const bool any_extra = sFlag != 0u;
    __syncthreads();
    if (do_final) break;
    if (!any_extra) { do_final = true; thresh *= 4; }      // :295-301
  }
  // ---- pose inversion (:304-305) and unique keypoint indices (:306-308)
  if (tid == 0) pose_invert(sR, sT, out->R, out->T);
  // unique keypoint indices in ascending match order (:306-308). qidx is non-decreasing in the match index (App. A Q4),
  // so an inlier starts a new keypoint iff the inlier before it has another qidx: one wave per 64-match word, the word
  // boundaries are stitched by one lane.
  uint32_t* const sFirstQ = sPre;                          // sPre is free after the last pass
  __shared__ uint32_t sLastQ[kMaxWords], sNewIn[kMaxWords], sOff[kMaxWords];
  const uint32_t lane = tid & 63u;
  for (uint32_t w = tid >> 6; w < W; w += 4u) {
    const u64 bits = inl[w];
    const bool in = (bits >> lane) & 1ull;
    const uint32_t q = in ? job.qidx[w * 64u + lane] : 0u;
    const u64 lower = bits & ((1ull << lane) - 1ull);
    const uint32_t pq = __shfl(q, lower ? 63u - (uint32_t)__clzll((long long)lower) : 0u);
    const u64 fresh = __ballot(in && lower != 0ull && q != pq);      // new keypoint, previous inlier in the same word
    if (lane == 0) sNewIn[w] = (uint32_t)__popcll(fresh);
    if (bits) {
      const uint32_t lo = (uint32_t)__ffsll((long long)bits) - 1u, hi = 63u - (uint32_t)__clzll((long long)bits);
      if (lane == lo) sFirstQ[w] = q;
      if (lane == hi) sLastQ[w] = q;
    }
  }
  __syncthreads();
  if (tid == 0) {
    uint32_t nm = 0, nk = 0, last = 0xFFFFFFFFu;
    for (uint32_t w = 0; w < W; ++w) {
      const u64 bits = inl[w];
      uint32_t first_new = 0;
      if (bits) { first_new = sFirstQ[w] != last ? 1u : 0u; last = sLastQ[w]; }
      sOff[w] = nk | (first_new << 31);
      nk += first_new + sNewIn[w];
      nm += (uint32_t)__popcll(bits);
    }
    out->n_match_inliers = nm;
    out->n_kp_inliers = nk;
    out->passes = passes;
  }
  __syncthreads();
  for (uint32_t w = tid >> 6; w < W; w += 4u) {
    const u64 bits = inl[w];
    if (!bits) continue;                                   // wave-uniform
    const bool in = (bits >> lane) & 1ull;
    const uint32_t q = in ? job.qidx[w * 64u + lane] : 0u;
    const u64 lower = bits & ((1ull << lane) - 1ull);
    const uint32_t pq = __shfl(q, lower ? 63u - (uint32_t)__clzll((long long)lower) : 0u);
    const bool first_new = (sOff[w] >> 31) != 0u;
    const bool is_new = in && (lower != 0ull ? q != pq : first_new);
    const u64 bal = __ballot(is_new);
    if (is_new) {
      kp_list[(sOff[w] & 0x7FFFFFFFu) + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull))] = q;
      atomicOr(&kp_bits[q >> 6], 1ull << (q & 63u));
    }
  }
}

// ------------------------------------------------------------------------------------------------ K11
// InvalidateQueryIndices (adjacency_ransac.cpp:93-123): drop every valid match whose keypoint is an inlier
// keypoint, then InvalidateIndices (:63-89): repeatedly drop valid matches whose sample degree is < 3.
struct InvArgs { ObjJob job; const u64* kp_bits; u64* scratch; const uint32_t* gate; uint32_t gate_min; };   // gate: as PrepArgs
// 256 threads (one wave per SIMD): a block this size still finds wave slots on a CU whose other slots are held by the
// matcher's resident grid; a 1024-thread block had to wait for a whole matcher launch to end (1.4 ms on average)
__global__ __launch_bounds__(256) void invalidate_kernel(Slots<InvArgs, kWideSlots> SL) {
  TOD_LATENCY_PRIO();   // latency-bound: win issue arbitration against the VALU-saturating matcher
  const ObjJob& job = SL.a[blockIdx.x].job;
  const u64* const kp_bits = SL.a[blockIdx.x].kp_bits; u64* const scratch = SL.a[blockIdx.x].scratch;
  if (SL.a[blockIdx.x].gate && *SL.a[blockIdx.x].gate < SL.a[blockIdx.x].gate_min) return;   // block-uniform
  __shared__ uint32_t sAny;
  const uint32_t tid = threadIdx.x, W = job.W, n = job.n;
  if (tid == 0) sAny = 0u;
  __syncthreads();
  for (uint32_t w = tid; w < W; w += 256u) {
    u64 gone = 0ull, val = job.valid[w];
    u64 bits = val;
    while (bits) {
      const uint32_t b = (uint32_t)__ffsll((long long)bits) - 1u;
      const uint32_t qi = job.qidx[w * 64u + b];
      if ((kp_bits[qi >> 6] >> (qi & 63u)) & 1ull) gone |= 1ull << b;
      bits &= bits - 1ull;
    }
    if (gone) { job.valid[w] = val & ~gone; atomicOr(&sAny, 1u); }
  }
  __syncthreads();
  if (sAny == 0u) return;                                  // InvalidateIndices(empty) does nothing (:68)
  while (true) {
    __syncthreads();
    if (tid == 0) sAny = 0u;
    __syncthreads();
    for (uint32_t w = tid; w < W; w += 256u) scratch[w] = 0ull;
    __syncthreads();
    for (uint32_t v = tid; v < n; v += 256u) {
      if ((job.valid[v >> 6] >> (v & 63u)) & 1ull) {
        uint32_t d = 0;
        for (uint32_t w = 0; w < W; ++w) d += (uint32_t)__popcll(job.samp[(size_t)v * W + w] & job.valid[w]);
        if (d < 3u) { atomicOr(&scratch[v >> 6], 1ull << (v & 63u)); atomicOr(&sAny, 1u); }   // min_sample_size_
      }
    }
    __syncthreads();
    if (sAny == 0u) break;
    for (uint32_t w = tid; w < W; w += 256u) job.valid[w] &= ~scratch[w];
  }
}

// finite_kernel + adjacency_kernel + round_prep_kernel for an object of at most 64 matches, by ONE wave: lane j = match j, row i of
// both bit matrices is one ballot, and the first round's statistics come from the rows in registers. A frame of self-similar texture
// has ~190 such objects and one big one: three dependent launches of ~2000 mostly empty blocks per frame become one launch of one
// wave per object (inside the pipeline, beside the matcher's resident grid, every dependent launch and every block costs a multiple
// of what it costs alone).
struct PrepSmallArgs { ObjJob job; uint32_t* stats; float span, err; };
template <class H>
__global__ __launch_bounds__(64) void small_prep_kernel(H S) {
  TOD_LATENCY_PRIO();
  const PrepSmallArgs& A = S.a[blockIdx.y];
  const ObjJob& job = A.job;
  const uint32_t n = job.n, l = lane_id();
  const float span = A.span, err = A.err;
  float q[3] = {0.f, 0.f, 0.f}, t[3] = {0.f, 0.f, 0.f}, kp[2] = {0.f, 0.f};
  if (l < n) {
    for (int c = 0; c < 3; ++c) { q[c] = job.query[3 * l + c]; t[c] = job.train[3 * l + c]; }
    kp[0] = job.kpxy[2 * l]; kp[1] = job.kpxy[2 * l + 1];
  }
  bool fin = l < n;
  for (int c = 0; c < 3; ++c) fin = fin && isfinite(t[c]) && isfinite(q[c]);
  const u64 finite = __ballot(fin);
  const u64 valid = n >= 64u ? ~0ull : ((1ull << n) - 1ull);
  u64 my_phys = 0ull, my_samp = 0ull;
  for (uint32_t i = 0; i < n; ++i) {                       // row i: the pair (i, lane)
    float qi[3], ti[3], ki[2];
    for (int c = 0; c < 3; ++c) { qi[c] = __shfl(q[c], (int)i); ti[c] = __shfl(t[c], (int)i); }
    ki[0] = __shfl(kp[0], (int)i); ki[1] = __shfl(kp[1], (int)i);
    bool ph = false, sa = false;
    if (l < n && l != i) {
      if (i < l) pair_test(qi, q, ti, t, ki, kp, span, err, ph, sa);   // the reference visits each pair once with i < j
      else pair_test(q, qi, t, ti, kp, ki, span, err, ph, sa);
    }
    const u64 pb = __ballot(ph), sb = __ballot(sa);
    if (l == i) { my_phys = pb; my_samp = sb; }
  }
  const bool isv = l < n;
  const uint32_t d = isv ? (uint32_t)__popcll(my_samp & valid) : 0u;
  const u64 deg7 = __ballot(isv && d >= kGateMinimal);
  const uint32_t degsum = wave_sum(d);
  bool on_tri = false;                                     // (wave-uniform loop: cross-lane reads need every lane active)
  for (uint32_t o = 0; o < n; ++o) {
    const u64 ro = rdlane64(my_samp, o);
    on_tri = on_tri || (((my_samp >> o) & 1ull) && (my_samp & ro & valid) != 0ull);
  }
  const bool triangle = __ballot(on_tri) != 0ull;
  if (l < n) { job.phys[l] = my_phys; job.samp[l] = my_samp; job.sampdeg[l] = d; }
  if (l == 0u) {
    job.finite[0] = finite; job.valid[0] = valid; job.deg7[0] = deg7;
    A.stats[0] = n; A.stats[1] = degsum; A.stats[2] = triangle ? 1u : 0u;
  }
}

// ------------------------------------------------------------------------------------------------ K_c
// ClusterPerObject (adjacency_ransac.cpp:176-205) for device-resident inputs. Matches arrive in the matcher's
// fixed-stride layout (k slots per query, counts[q] used); the flat order (query asc, rank asc) is what the
// reference's push_back order produces, and grouping by object is stable, so query_indices_ stays
// non-decreasing per object (App. A Q4).
struct LookupArgs {
  const float* kp_xy; uint32_t nq; const float* cloud; const void* depth; int depth_is_u16; uint32_t H, Wimg;
  float fx, fy, cx, cy; const uint32_t* counts; uint32_t* kept; float* qpt; uint32_t* err;
};
// cloud != nullptr: the query point is read from the H x W x 3 cloud (adjacency_ransac.cpp:184-185).
// Otherwise N3: the reference back-projects the WHOLE registered depth image to an H x W x 3 cloud
// (ecto_opencv DepthTo3d, python/object_recognition_tod/detector.py:26,62,66-69) and then reads Q points of it.
// Here the Q points are computed directly: same pixel truncation, same pinhole back-projection as cv::depthTo3d
// (x = (u - cx) z / fx, y = (v - cy) z / fy), uint16 depth in millimetres with 0 = no measurement -> NaN as
// cv::rescaleDepth does (third-party conventions, recalled; parity unpinned).
__global__ __launch_bounds__(256) void cluster_lookup_kernel(Slots<LookupArgs> SL) {
  TOD_LATENCY_PRIO();   // latency-bound: win issue arbitration against the VALU-saturating matcher
  const LookupArgs& a = SL.a[blockIdx.y];
  const uint32_t q = blockIdx.x * 256u + threadIdx.x;
  if (q >= a.nq) return;
  const int row = (int)a.kp_xy[2 * q + 1], col = (int)a.kp_xy[2 * q];    // float -> int truncation (:185)
  if (row < 0 || col < 0 || (uint32_t)row >= a.H || (uint32_t)col >= a.Wimg) {
    atomicExch(a.err, 1u);
    a.kept[q] = 0;
    return;
  }
  float x, y, z;
  if (a.cloud) {
    const float* p = a.cloud + 3 * ((size_t)row * a.Wimg + col);
    x = p[0]; y = p[1]; z = p[2];
  } else {
    if (a.depth_is_u16) {
      const uint16_t d = reinterpret_cast<const uint16_t*>(a.depth)[(size_t)row * a.Wimg + col];
      z = d == 0 ? __builtin_nanf("") : (float)d * 0.001f;
    } else {
      z = reinterpret_cast<const float*>(a.depth)[(size_t)row * a.Wimg + col];
    }
    x = ((float)col - a.cx) * z / a.fx; y = ((float)row - a.cy) * z / a.fy;
  }
  a.qpt[3 * q] = x; a.qpt[3 * q + 1] = y; a.qpt[3 * q + 2] = z;
  a.kept[q] = isnan(x) ? 0u : a.counts[q];                               // only .x is tested (:189)
}

// exclusive scan of kept[0..nq) into offs[0..nq], one block
struct ScanArgs { const uint32_t* kept; uint32_t nq; uint32_t* offs; };
__global__ __launch_bounds__(256) void cluster_scan_kernel(Slots<ScanArgs> SL) {
  TOD_LATENCY_PRIO();   // latency-bound: win issue arbitration against the VALU-saturating matcher
  const uint32_t* __restrict__ kept = SL.a[blockIdx.x].kept;
  const uint32_t nq = SL.a[blockIdx.x].nq;
  uint32_t* const offs = SL.a[blockIdx.x].offs;
  __shared__ uint32_t part[256];
  const uint32_t tid = threadIdx.x;
  const uint32_t chunk = (nq + 255u) / 256u;
  const uint32_t lo = min(nq, tid * chunk), hi = min(nq, lo + chunk);
  uint32_t s = 0;
  for (uint32_t i = lo; i < hi; ++i) s += kept[i];
  part[tid] = s;
  __syncthreads();
  if (tid == 0) {
    uint32_t acc = 0;
    for (uint32_t i = 0; i < 256u; ++i) { const uint32_t c = part[i]; part[i] = acc; acc += c; }
    offs[nq] = acc;
  }
  __syncthreads();
  uint32_t acc = part[tid];
  for (uint32_t i = lo; i < hi; ++i) { offs[i] = acc; acc += kept[i]; }
}

struct ScatterArgs {
  const float* kp_xy; uint32_t nq, k; const todhip_dmatch* matches; const float* mxyz; const uint32_t* kept;
  const uint32_t* offs; const float* qpt; uint32_t n_objs; uint32_t* obj_of; uint32_t* hist; float* ftrain; float* fquery;
  uint32_t* fqidx; float* fkp; uint32_t* err;
};
__global__ __launch_bounds__(256) void cluster_scatter_kernel(Slots<ScatterArgs> SL) {
  TOD_LATENCY_PRIO();   // latency-bound: win issue arbitration against the VALU-saturating matcher
  const ScatterArgs& a = SL.a[blockIdx.y];
  const float* __restrict__ kp_xy = a.kp_xy; const uint32_t nq = a.nq, k = a.k, n_objs = a.n_objs;
  const todhip_dmatch* __restrict__ matches = a.matches; const float* __restrict__ mxyz = a.mxyz;
  const uint32_t* __restrict__ kept = a.kept; const uint32_t* __restrict__ offs = a.offs; const float* __restrict__ qpt = a.qpt;
  uint32_t* const obj_of = a.obj_of; uint32_t* const hist = a.hist; float* const ftrain = a.ftrain; float* const fquery = a.fquery;
  uint32_t* const fqidx = a.fqidx; float* const fkp = a.fkp; uint32_t* const err = a.err;
  const uint32_t t = blockIdx.x * 256u + threadIdx.x;
  const uint32_t q = t / k, j = t % k;
  if (q >= nq || j >= kept[q]) return;
  const uint32_t f = offs[q] + j;
  const todhip_dmatch m = matches[(size_t)q * k + j];
  uint32_t o = (uint32_t)m.imgIdx;
  if (m.imgIdx < 0 || o >= n_objs) { atomicExch(err, 2u); o = 0; }
  obj_of[f] = o;
  atomicAdd(&hist[o], 1u);
  for (int c = 0; c < 3; ++c) { ftrain[3 * f + c] = mxyz[((size_t)q * k + j) * 3 + c]; fquery[3 * f + c] = qpt[3 * q + c]; }
  fqidx[f] = q;
  fkp[2 * f] = kp_xy[2 * q]; fkp[2 * f + 1] = kp_xy[2 * q + 1];
}

// stable grouping by object: destination = group offset + number of earlier matches of the same object
struct GroupArgs {
  uint32_t n_all; const uint32_t* obj_of; const uint32_t* goff; const float* ftrain; const float* fquery;
  const uint32_t* fqidx; const float* fkp; float* train; float* query; uint32_t* qidx; float* kpxy;
};
__global__ __launch_bounds__(256) void cluster_group_kernel(Slots<GroupArgs> SL) {
  TOD_LATENCY_PRIO();   // latency-bound: win issue arbitration against the VALU-saturating matcher
  const GroupArgs& a = SL.a[blockIdx.y];
  const uint32_t n_all = a.n_all;
  const uint32_t* __restrict__ obj_of = a.obj_of; const uint32_t* __restrict__ goff = a.goff;
  const float* __restrict__ ftrain = a.ftrain; const float* __restrict__ fquery = a.fquery;
  const uint32_t* __restrict__ fqidx = a.fqidx; const float* __restrict__ fkp = a.fkp;
  float* const train = a.train; float* const query = a.query; uint32_t* const qidx = a.qidx; float* const kpxy = a.kpxy;
  const uint32_t f = blockIdx.x * 256u + threadIdx.x;
  if (f >= n_all) return;
  const uint32_t o = obj_of[f];
  uint32_t rank = 0;
  for (uint32_t g = 0; g < f; ++g) rank += obj_of[g] == o;
  const uint32_t d = goff[o] + rank;
  for (int c = 0; c < 3; ++c) { train[3 * d + c] = ftrain[3 * f + c]; query[3 * d + c] = fquery[3 * f + c]; }
  qidx[d] = fqidx[f];
  kpxy[2 * d] = fkp[2 * f]; kpxy[2 * d + 1] = fkp[2 * f + 1];
}

// ClusterPerObject of one frame in ONE launch of one block (the four kernels above are its parts, kept for the launch-group of a
// frame whose match list does not fit one block's patience: none today): lookup -> scan -> scatter + histogram -> object offsets
// -> stable grouping, with the histogram and the frame's totals written straight into the slot's mailbox. Inside the pipeline
// every dependent launch waits for wave slots beside the matcher's resident grid: five dependent launches and a host round trip
// between the scatter and the grouping were 0.4-0.5 ms per batch there (45 us alone).
struct ClusterArgs {
  const float* kp_xy; const float* cloud; const void* depth; const uint32_t* counts; const todhip_dmatch* matches; const float* mxyz;
  uint32_t nq, k, H, Wimg, n_objs, qidx_add; int depth_is_u16; float fx, fy, cx, cy;   // qidx_add: added to the keypoint index stored per match
  uint32_t *kept, *offs, *obj_of, *src, *hist, *goff, *cnt; float* qpt;          // device scratch
  float *train, *query, *kpxy; uint32_t* qidx;                                    // grouped outputs
  uint32_t *m_hist, *m_ctl;                                                       // mailbox (pinned): histogram; [0] error, [4] n_all
};
__global__ __launch_bounds__(256) void cluster_frame_kernel(Slots<ClusterArgs> SL) {
  TOD_LATENCY_PRIO();
  const ClusterArgs& a = SL.a[blockIdx.x];
  __shared__ uint32_t part[256], s_o[256], s_err, s_total;
  const uint32_t tid = threadIdx.x, nq = a.nq, k = a.k, n_objs = a.n_objs;
  if (tid == 0) s_err = 0u;
  for (uint32_t o = tid; o < n_objs; o += 256u) { a.hist[o] = 0u; a.cnt[o] = 0u; }
  __syncthreads();
  // ---- the keypoint's 3D point (adjacency_ransac.cpp:184-189), see cluster_lookup_kernel
  for (uint32_t q = tid; q < nq; q += 256u) {
    const int row = (int)a.kp_xy[2 * q + 1], col = (int)a.kp_xy[2 * q];    // float -> int truncation (:185)
    const bool lookup = a.cloud || a.depth;                                // neither: the 2D-only branch (GuessGenerator.cpp:147-152), no 3D point
    if (lookup && (row < 0 || col < 0 || (uint32_t)row >= a.H || (uint32_t)col >= a.Wimg)) { atomicExch(&s_err, 1u); a.kept[q] = 0; continue; }
    float x, y, z;
    if (!lookup) {
      x = y = z = 0.f;
    } else if (a.cloud) {
      const float* p = a.cloud + 3 * ((size_t)row * a.Wimg + col);
      x = p[0]; y = p[1]; z = p[2];
    } else {
      if (a.depth_is_u16) {
        const uint16_t d = reinterpret_cast<const uint16_t*>(a.depth)[(size_t)row * a.Wimg + col];
        z = d == 0 ? __builtin_nanf("") : (float)d * 0.001f;
      } else {
        z = reinterpret_cast<const float*>(a.depth)[(size_t)row * a.Wimg + col];
      }
      x = ((float)col - a.cx) * z / a.fx; y = ((float)row - a.cy) * z / a.fy;
    }
    a.qpt[3 * q] = x; a.qpt[3 * q + 1] = y; a.qpt[3 * q + 2] = z;
    uint32_t c_q = a.counts[q];
    if (c_q > k) { atomicExch(&s_err, 3u); c_q = k; }                      // (a count beyond the fixed stride: refused)
    a.kept[q] = isnan(x) ? 0u : c_q;                                       // only .x is tested (:189)
  }
  __syncthreads();
  // ---- exclusive scan of kept -> offs: the flat order (query asc, rank asc) of the reference's push_back
  {
    const uint32_t chunk = (nq + 255u) / 256u;
    const uint32_t lo = min(nq, tid * chunk), hi = min(nq, lo + chunk);
    uint32_t sum = 0;
    for (uint32_t i = lo; i < hi; ++i) sum += a.kept[i];
    part[tid] = sum;
    __syncthreads();
    if (tid == 0) {
      uint32_t acc = 0;
      for (uint32_t i = 0; i < 256u; ++i) { const uint32_t c = part[i]; part[i] = acc; acc += c; }
      s_total = acc;
    }
    __syncthreads();
    uint32_t acc = part[tid];
    for (uint32_t i = lo; i < hi; ++i) { a.offs[i] = acc; acc += a.kept[i]; }
  }
  const uint32_t n_all = s_total;
  __syncthreads();
  // ---- every match's flat slot, object and source; histogram per object
  for (uint32_t t = tid; t < nq * k; t += 256u) {
    const uint32_t q = t / k, j = t % k;
    if (j >= a.kept[q]) continue;
    const uint32_t f = a.offs[q] + j;
    const todhip_dmatch m = a.matches[t];
    uint32_t o = (uint32_t)m.imgIdx;
    if (m.imgIdx < 0 || o >= n_objs) { atomicExch(&s_err, 2u); o = 0; }
    a.obj_of[f] = o; a.src[f] = t;
    atomicAdd(&a.hist[o], 1u);
  }
  __threadfence();
  __syncthreads();
  // ---- object offsets = exclusive scan of the histogram; the histogram goes to the host
  {
    const uint32_t chunk = (n_objs + 255u) / 256u;
    const uint32_t lo = min(n_objs, tid * chunk), hi = min(n_objs, lo + chunk);
    uint32_t sum = 0;
    for (uint32_t i = lo; i < hi; ++i) sum += __hip_atomic_load(a.hist + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    part[tid] = sum;
    __syncthreads();
    if (tid == 0) {
      uint32_t acc = 0;
      for (uint32_t i = 0; i < 256u; ++i) { const uint32_t c = part[i]; part[i] = acc; acc += c; }
    }
    __syncthreads();
    uint32_t acc = part[tid];
    for (uint32_t i = lo; i < hi; ++i) {
      const uint32_t h = __hip_atomic_load(a.hist + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      a.goff[i] = acc; a.m_hist[i] = h;
      acc += h;
    }
  }
  __syncthreads();
  // ---- stable grouping by object, 256 flat slots at a time: destination = object offset + matches of the object in earlier
  // chunks (cnt) + earlier matches of the object in this chunk
  for (uint32_t base = 0; base < n_all; base += 256u) {
    const uint32_t f = base + tid;
    const bool have = f < n_all;
    const uint32_t o = have ? a.obj_of[f] : 0xFFFFFFFFu;
    s_o[tid] = o;
    // the object's matches in earlier chunks: read by EVERY thread before any thread of this chunk updates it (the barrier below) --
    // the last match of an object in the chunk may sit in a wave that runs ahead of the waves holding its earlier ones, and a count
    // read after that update sends the earlier matches to the wrong places (found as one batch result in five differing from the
    // frame-by-frame call: tools/verify_repeat_frame.py)
    const uint32_t seen = have ? a.cnt[o] : 0u;
    __syncthreads();
    if (have) {
      uint32_t before = 0, after = 0;
      for (uint32_t u = 0; u < tid; ++u) before += s_o[u] == o;
      for (uint32_t u = tid + 1u; u < 256u; ++u) after += s_o[u] == o;
      const uint32_t d = a.goff[o] + seen + before;
      const uint32_t t = a.src[f], q = t / k;
      for (int c = 0; c < 3; ++c) { a.train[3 * d + c] = a.mxyz[(size_t)t * 3 + c]; a.query[3 * d + c] = a.qpt[3 * q + c]; }
      a.qidx[d] = q + a.qidx_add;
      a.kpxy[2 * d] = a.kp_xy[2 * q]; a.kpxy[2 * d + 1] = a.kp_xy[2 * q + 1];
      if (after == 0u) a.cnt[o] = seen + before + 1u;       // the object's last match of the chunk
    }
    __syncthreads();
  }
  if (tid == 0) { a.m_ctl[0] = s_err; a.m_ctl[4] = n_all; }
}

#include "verify_sprint.h"

// ------------------------------------------------------------------------------------------------ host side
struct VerifyWs {
  DevBuf train, query, qidx, kpxy, phys, samp, bits, sampdeg, nvalid, rnd, table, iter_samples, counts, gate_m,
      small, deferred, stacks, kp_bits, clique_adj, adjc_scratch, c_kept, c_offs, c_qpt, c_obj, c_hist, c_goff, f_train,
      f_query, f_qidx, f_kp, sprint_status, sprint_stack, c_src, c_cnt;
  // the slot's mailbox: pinned host memory that kernels read and write directly (see copy_words_kernel)
  HostBuf m_small, m_hist, m_goff, m_rnd, m_pos, m_counts, m_kp, m_nvalid;
  HostBuf m_sprint, m_sprint_out, m_sprint_kp;              // sprint_kernel: the object list in, records and keypoint lists out
  HostBuf h_small;                                          // staging of the test hooks
  void release() {
    DevBuf* bufs[] = {&train, &query, &qidx, &kpxy, &phys, &samp, &bits, &sampdeg, &nvalid, &rnd, &table, &iter_samples, &counts, &gate_m,
                      &small, &deferred, &stacks, &kp_bits, &clique_adj, &adjc_scratch, &c_kept, &c_offs, &c_qpt, &c_obj, &c_hist,
                      &c_goff, &f_train, &f_query, &f_qidx, &f_kp, &sprint_status, &sprint_stack, &c_src, &c_cnt};
    for (DevBuf* b : bufs) b->release();
    HostBuf* hb[] = {&m_small, &m_hist, &m_goff, &m_rnd, &m_pos, &m_counts, &m_kp, &m_nvalid, &h_small, &m_sprint, &m_sprint_out, &m_sprint_kp};
    for (HostBuf* b : hb) b->release();
  }
};

// one workspace per frame slot of a batch; slot 0 also serves the single-frame entry points and the test hooks
struct StreamCache;
struct VerifyPool {
  std::vector<VerifyWs*> slots; std::vector<StreamCache*> streams;
  std::vector<hipEvent_t> side_ev;                          // one per lane of a batch (Engine::run_ticks), created on first use
  // argument sets of a launch group's long lists (launch_many): one pair per lane, so that groups in flight on different streams
  // never share staging memory, and a pair only grows while its lane is idle
  static constexpr size_t kMaxLanes = 17;
  HostBuf args_stage[kMaxLanes]; DevBuf args_dev[kMaxLanes];
  DevBuf kceil; bool kceil_ready = false;                   // sprint_kernel: ceil(k) of ransac.h:130 per (|valid|, n_best), 65 x 65
};
// The flights' streams belong to the process, not to a context: a process has eight hardware queues for all of its streams
// (DESIGN 7), the runtime deals streams onto them round robin, and every further stream -- busy or not -- makes it likelier that two
// busy ones share a queue. Contexts driven from several host threads share these few: a flight takes a stream for itself
// (`taken`) and gives it back when it has landed; a context that finds none free keeps its heavy phase in the lock-step -- with
// several batches in flight on contexts of their own the batches overlap each other anyway.
struct SideStreams {
  struct PerDevice {
    std::vector<hipStream_t> st;
    std::atomic<bool> taken[16];
    uint32_t partition = 0;                                 // todhip_set_cu_partition's value the streams were created under
    PerDevice() { for (auto& t : taken) t.store(false); }
  };
  std::mutex mu;
  std::map<int, PerDevice> by_device;
  hipError_t get(int device, uint32_t n, std::vector<hipStream_t>& out, PerDevice** pd) {
    std::lock_guard<std::mutex> g(mu);
    PerDevice& d = by_device[device];
    const uint32_t part = tod_cu_partition();
    if (d.partition != part) {                              // the CU partition changed: new streams, once nobody is on the old ones
      bool in_use = false;
      for (size_t i = 0; i < d.st.size(); ++i) in_use = in_use || d.taken[i].load();
      if (!in_use) {
        for (hipStream_t s0 : d.st) { (void)hipStreamSynchronize(s0); (void)hipStreamDestroy(s0); }
        d.st.clear();
        d.partition = part;
      }
    }
    while (d.st.size() < n) {
      hipStream_t s2;                                       // latency-bound work: the highest priority there is, or the CU partition's
      const hipError_t e = tod_stream_create(&s2, device, TODHIP_STREAM_LATENCY);
      if (e != hipSuccess) return e;
      d.st.push_back(s2);
    }
    out.assign(d.st.begin(), d.st.begin() + n);
    *pd = &d;
    return hipSuccess;
  }
};
SideStreams g_side_streams;
// batches being verified right now, over all contexts of the process: with three or more in the air the hardware queues are
// already kept busy by each other's ticks and a flight only adds a stream to wait behind (bench `chained`, 4 workers: 4060
// frames/s in lock-step, 3740 with flights; 1 worker: 1490 / 2230, 2 workers: 2400 / 2900)
std::atomic<int> g_batches_in_air{0};

constexpr uint32_t kEvalLdsSmall = 48u * 1024u;
constexpr uint32_t kEvalLdsBig = 160u * 1024u - 512u;
constexpr uint32_t kStackCap = 128u * 1024u;       // u16 entries per wave beyond the LDS part of the stack (256 KB)
constexpr uint32_t kMaxEvalWaves = 4096u;          // hypotheses per evaluation batch
constexpr uint32_t kMailSmallWords = 128u;         // [0, 64) = the slot's device control words, [64] = n_all

VerifyPool* pool_of(todhip_ctx* ctx) {
  if (!ctx->verify_ws) ctx->verify_ws = new VerifyPool();
  return reinterpret_cast<VerifyPool*>(ctx->verify_ws);
}
VerifyWs* ws_of(todhip_ctx* ctx, size_t slot = 0) {
  VerifyPool* p = pool_of(ctx);
  while (p->slots.size() <= slot) p->slots.push_back(new VerifyWs());
  return p->slots[slot];
}

// glibc random_r TYPE_3 (see include/todhip.h, decision D4)
inline uint32_t rng_next(todhip_rng& r) {
  r.s[r.f] += r.s[r.b];
  const uint32_t out = r.s[r.f] >> 1;
  r.f = r.f == 30u ? 0u : r.f + 1u; r.b = r.b == 30u ? 0u : r.b + 1u;
  ++r.draws;
  return out;
}

// The rand() stream of a frame, generated once and shared: every round of every object reads a window of it, and
// frames that start from the same generator state (a harness restarting rand() per frame, decision D4) share one
// copy. Snapshots of the generator every kSnap draws give the state at any position without replaying the stream.
struct StreamCache {
  static constexpr uint64_t kSnap = 4096;
  todhip_rng init, gen;
  std::vector<uint32_t> vals;                              // vals[i] = i-th draw after `init`
  std::vector<todhip_rng> snaps;                           // snaps[j] = generator state before draw j * kSnap
  explicit StreamCache(const todhip_rng& r) : init(r), gen(r) {}
  bool same_start(const todhip_rng& r) const {
    return r.f == init.f && r.b == init.b && std::memcmp(r.s, init.s, sizeof(init.s)) == 0;
  }
  void extend_to(uint64_t n) {
    while (vals.size() < n) {
      if (vals.size() % kSnap == 0) snaps.push_back(gen);
      vals.push_back(rng_next(gen));
    }
  }
  // device copy of the stream (append-only): a round's window is an offset into it, nothing is copied per tick
  DevBuf dev;
  uint64_t dev_valid = 0;
  hipError_t ensure_device(uint64_t n, hipStream_t st) {
    extend_to(n);
    if (n <= dev_valid) return hipSuccess;
    const uint64_t want = std::max<uint64_t>(n, 2 * dev_valid);
    extend_to(want);
    if (dev.cap < want * sizeof(uint32_t)) {                // grow: DevBuf::reserve drops the old contents, upload all again
      hipError_t e = hipDeviceSynchronize();                // kernels of earlier ticks and of flights on other streams may still read it
      if (e != hipSuccess) return e;
      e = dev.reserve((size_t)want * 2 * sizeof(uint32_t));
      if (e != hipSuccess) return e;
      dev_valid = 0;
    }
    hipError_t e = hipMemcpyAsync(dev.as<uint32_t>() + dev_valid, vals.data() + dev_valid, (size_t)(want - dev_valid) * sizeof(uint32_t),
                                  hipMemcpyHostToDevice, st);
    // complete before anyone is told the stream reaches this far: slots that share the cache may read it from another stream
    // (the copy doubles the valid length, so a context in steady state never gets here)
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e == hipSuccess) dev_valid = want;
    return e;
  }
  todhip_rng state_at(uint64_t pos) {                      // generator after `pos` draws (draw counter relative to init)
    extend_to(pos + 1);
    todhip_rng g = snaps[pos / kSnap];
    for (uint64_t i = (pos / kSnap) * kSnap; i < pos; ++i) (void)rng_next(g);
    return g;
  }
};

int set_big_lds_once(todhip_ctx* ctx) {
  static std::atomic<bool> done{false};                   // contexts may be driven from several host threads
  if (!done.load(std::memory_order_acquire)) {
    TOD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(eval_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)kEvalLdsBig));
    TOD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(eval_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)kEvalLdsBig));
    TOD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(eval_kernel<false, SlotsPtr<EvalArgs>>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)kEvalLdsBig));
    TOD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(eval_kernel<true, SlotsPtr<EvalArgs>>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)kEvalLdsBig));
    TOD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(clique_test_kernel<false>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)kEvalLdsBig));
    TOD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(clique_test_kernel<true>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)kEvalLdsBig));
    TOD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(chain_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)kEvalLdsBig));
    TOD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(sprint_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)kSprintLds));
    done.store(true, std::memory_order_release);
  }
  return TODHIP_OK;
}

struct ObjSpan {
  uint32_t obj, offset, n;
  // this object's slices of the slot's adjacency / bitset / degree buffers (all objects of a frame are prepared in
  // one tick, so each needs its own), and its first round's |valid|
  uint64_t adj_off = 0; uint32_t bits_off = 0, deg_off = 0, nvalid = 0, degsum = 0, triangle = 1;
  bool resume = false, counted = false;                    // sprint_kernel left it unfinished (its first-round statistics are stale) / in the counters
};
struct DepthInput { const void* d_depth; int is_u16; float fx, fy, cx, cy; };

// ---------------------------------------------------------------------------------------------- the batch engine
// One Slot = one frame = GuessGenerator::process after matching (GuessGenerator.cpp:127-250): ClusterPerObject, then
// per object (ascending imgIdx) AdjacencyRansac::Ransac rounds (adjacency_ransac.cpp:234-309) until one fails. The
// recursion of the reference becomes an explicit phase machine per slot so that the slots of a batch can be
// advanced together: a TICK lets every live slot issue the kernels of its next phase into per-kernel lists, launches
// each non-empty list once (all slots in one grid), synchronizes once, and lets every slot consume its results
// (the ransac.h:95-135 bookkeeping is replayed on the host so that pow/log are libm's).
enum Phase { PH_CLUSTER, PH_CLUSTER_WAIT, PH_GROUP, PH_PREPALL, PH_PREPALL_WAIT, PH_OBJECT, PH_ROUND, PH_PREP_WAIT, PH_DRAW,
             PH_DRAW_WAIT, PH_EVAL2, PH_EVAL2_WAIT, PH_GROWTH, PH_GROWTH_WAIT, PH_SPRINT_WAIT, PH_DONE };

struct RoundState {                                       // computeModel (ransac.h:80-143) in flight
  uint64_t consumed = 0;                                  // draws used by completed getSamples calls of this round
  uint32_t it_drawn = 0, attempts_carry = 0;
  bool selection_empty = false, loop_done = false;
  int iterations = 0, n_best = -INT_MAX;
  double k = 1.0;
  uint32_t best_it = 0;
  uint64_t pos_after_stop = 0;
  uint32_t batch = 16, lookahead = 4096;   // evaluation batches: 16, 64, 256, 1024 (easy scenes stop within the first)
  uint32_t nvalid = 0, total_iters = 0;
  uint32_t it_begin = 0, want = 0, got = 0, S = 0, window_len = 0;   // the evaluation batch being drawn
  uint32_t n_def = 0;
  uint32_t s_floor = 0;                                   // grows x4 whenever a window ran out before the request was served
};

// TODHIP_SPRINT_MARGIN=n (diagnostics): the first look-ahead of a sprint, so that tests can put the stream's end -- and with it the
// kernel's stop-and-resume path -- anywhere in a frame's rounds (default 2^17 draws; it quadruples on every stop)
inline uint64_t sprint_margin0() {
  static const uint64_t v = [] { const char* e = getenv("TODHIP_SPRINT_MARGIN"); const long long x = e ? atoll(e) : 0; return x > 0 ? (uint64_t)x : (uint64_t)(1u << 17); }();
  return v;
}
struct Slot {
  VerifyWs* ws = nullptr;
  // inputs (device-resident form)
  const float* d_kp_xy = nullptr; const float* d_cloud = nullptr; DepthInput dep = {}; bool use_depth = false;
  const uint32_t* d_counts = nullptr; const todhip_dmatch* d_matches = nullptr; const float* d_mxyz = nullptr;
  todhip_rng* rng = nullptr;                              // caller's generator: set to the final state when the slot is done
  StreamCache* stream = nullptr;                          // shared with the slots that start from the same state
  uint64_t start_draws = 0, abs_pos = 0;                  // rng->draws at entry; draws consumed by the completed rounds
  // results
  std::vector<todhip_pose> poses;
  std::vector<uint32_t> inliers;
  std::vector<todhip_round_trace> traces;
  int rc = TODHIP_OK;
  // progress
  Phase ph = PH_DONE;
  std::vector<ObjSpan> objs;
  size_t oi = 0;
  ObjJob job = {};
  uint32_t n_all = 0;
  bool pending_invalidate = false;
  bool in_flight = false;                                 // its current tick runs on a side stream (run_ticks)
  // window-size hint for the next object's first draw window: what the previous objects of this frame consumed when their
  // getSamples gave up (1000 failing attempts, ~4-9k draws). Frames with many stray matches hold runs of such objects; a
  // first window sized for a healthy object (576) made each of them crawl through three windows = three ticks
  uint32_t s_hint = 0;
  std::vector<size_t> sprint_members;                      // indices into objs of the sprint in flight (sprint_kernel)
  uint64_t sprint_margin = sprint_margin0();               // rand() words the device copy of the stream reaches beyond the sprint's start
  todhip_round_trace tr = {};
  RoundState r;
};

struct Launches {
  std::vector<CopyArgs> copy_in, zero, copy_out;
  std::vector<LookupArgs> lookup; std::vector<ScanArgs> scan; std::vector<ScatterArgs> scatter; std::vector<GroupArgs> group;
  std::vector<InvArgs> inval, inval_after; std::vector<JobArgs> finite; std::vector<AdjArgs> adj; std::vector<PrepArgs> prep, prep_after;   // *_after: behind the growth kernels
  std::vector<DrawArgs> draw, draw_small; std::vector<ChainArgs> chain;
  // the rnd pointers of the draw lists are resolved at launch time: a later slot of the same tick may grow (move)
  // the shared stream buffer
  std::vector<std::pair<StreamCache*, uint64_t>> draw_src, draw_small_src; std::vector<EvalArgs> eval_small, eval_big, eval_direct;
  std::vector<GrowthArgs> growth;
  std::vector<SprintArgs> sprint; std::vector<StreamCache*> sprint_src;
  std::vector<ClusterArgs> cluster; std::vector<PrepSmallArgs> prep_small;
};

// launch `kern` over the argument sets of v, kMaxSlots at a time; extent(a) = blocks one set needs in x (and y)
template <uint32_t N = kMaxSlots, class A, class Kern, class Extent>
void launch_list(hipStream_t st, Kern kern, const std::vector<A>& v, uint32_t block, uint32_t lds, int slot_dim, Extent extent) {
  static_assert(sizeof(Slots<A, N>) <= 4096, "kernel arguments are limited to 4 KB");
  for (size_t i0 = 0; i0 < v.size(); i0 += N) {
    const uint32_t n = (uint32_t)std::min<size_t>(N, v.size() - i0);
    Slots<A, N> S;
    std::memset(&S, 0, sizeof(S));
    uint32_t gx = 1, gy = 1;
    for (uint32_t i = 0; i < n; ++i) {
      S.a[i] = v[i0 + i];
      const dim3 e = extent(v[i0 + i]);
      gx = std::max(gx, e.x); gy = std::max(gy, e.y);
    }
    dim3 grid;
    if (slot_dim == 0) grid = dim3(n);
    else if (slot_dim == 1) grid = dim3(gx, n);
    else grid = dim3(gx, gy, n);
    hipLaunchKernelGGL(kern, grid, dim3(block), lds, st, S);
  }
}

struct Engine {
  todhip_ctx* ctx;
  hipStream_t st;
  uint32_t nq, H, Wimg, k, n_objs;
  const float* spans;
  const todhip_verify_params* prm;
  Launches L;
  size_t lane_now = 0;                                      // the lane launch_all is filling (its staging pair)

  static uint32_t* mail(const Slot& s) { return s.ws->m_small.as<uint32_t>(); }
  void export_small(Slot& s) { L.copy_out.push_back({s.ws->small.as<uint32_t>(), mail(s), 64u}); }
  void fail(Slot& s, int rc) { s.rc = rc; s.ph = PH_DONE; }
#define SLOT_HIP(expr) do { if ((expr) != hipSuccess) { fail(s, TODHIP_EHIP); return; } } while (0)

  static ObjJob make_job(const Slot& s, const ObjSpan& o) {
    VerifyWs* ws = s.ws;
    ObjJob job;
    const uint32_t n = o.n, W = (n + 63u) / 64u;
    job.n = n; job.W = W;
    job.train = ws->train.as<float>() + 3 * (size_t)o.offset; job.query = ws->query.as<float>() + 3 * (size_t)o.offset;
    job.qidx = ws->qidx.as<uint32_t>() + o.offset; job.kpxy = ws->kpxy.as<float>() + 2 * (size_t)o.offset;
    job.phys = ws->phys.as<u64>() + o.adj_off; job.samp = ws->samp.as<u64>() + o.adj_off;
    u64* bits = ws->bits.as<u64>() + o.bits_off;            // finite | valid | deg7 | inl | rest | extra | scratch
    job.finite = bits; job.valid = bits + W; job.deg7 = bits + 2 * W;
    job.sampdeg = ws->sampdeg.as<uint32_t>() + o.deg_off;
    return job;
  }
  u64* obj_bits(const Slot& s) const { return s.ws->bits.as<u64>() + s.objs[s.oi].bits_off; }

  // one AdjacencyRansac::Ransac call starts with |valid| known (adjacency_ransac.cpp:234-241)
  void start_round(Slot& s, uint32_t nvalid, uint32_t degsum, uint32_t triangle) {
    VerifyWs* ws = s.ws;
    RoundState& r = s.r;
    TOD_DBG2("round: n=%u W=%u nvalid=%u edges=%u triangle=%u", s.job.n, s.job.W, nvalid, degsum / 2u, triangle);
    if (nvalid < 3) { round_done(s, false); return; }      // :238-241
    if (!triangle) {
      // no three mutually sample-adjacent valid matches: getSamples fails 1000 times, each attempt consuming exactly
      // |valid| + |E| draws whatever their values (round_prep_kernel), selection.empty() ends computeModel at
      // iterations_ == 0 (ransac.h:100-101) and Ransac returns nothing. No kernel, no tick.
      r = RoundState();
      s.abs_pos += (uint64_t)kMaxSampleChecks * ((uint64_t)nvalid + degsum / 2u);
      s.tr.iterations = 0; s.tr.best_iteration = 0; s.tr.best_count = -INT_MAX;
      round_done(s, false);
      return;
    }
    r = RoundState();
    r.s_floor = s.s_hint;
    r.nvalid = nvalid;
    // First evaluation batch: an object with many valid matches is expensive to evaluate (its clique gate walks a graph of
    // about that many vertices, one wave per hypothesis, and a tick lasts as long as its slowest hypothesis), and when it is
    // real its first hypotheses end the loop: with w = consensus / valid, k = log(0.01) / log(1 - w^3) (ransac.h:123-130) is
    // <= 2 from w = 0.966 on (<= 1 only from 0.9967 on). So two hypotheses, not 16; the replay asks for more if k says so.
    // Small objects keep the batch of 16 (cheap evaluations, usually needing many).
    if (nvalid >= 64u) r.batch = 2;
    r.total_iters = prm->n_ransac_iterations + 1u;          // iterations_ runs 0 .. max_iterations (ransac.h:132-134)
    SLOT_HIP(ws->iter_samples.reserve((size_t)(r.total_iters + 1) * 3 * sizeof(uint32_t)));
    SLOT_HIP(ws->gate_m.reserve((size_t)(r.total_iters + 1) * sizeof(uint32_t)));
    SLOT_HIP(ws->deferred.reserve((size_t)(r.total_iters + 1) * sizeof(uint32_t)));
    SLOT_HIP(ws->m_counts.reserve((size_t)(r.total_iters + 1) * sizeof(int32_t)));
    SLOT_HIP(ws->m_pos.reserve((size_t)(r.total_iters + 1) * sizeof(uint32_t)));
    begin_batch(s);
  }

  // the evaluation of iterations [it_lo, it_hi) (first pass), or of the deferred ones (second pass: graphs that need the whole
  // LDS of a CU, or global scratch). zero_status: first evaluation launch of the batch (the deferred list and the counters
  // accumulate over the windows of one batch)
  void push_eval(Slot& s, bool second, uint32_t it_lo, uint32_t it_hi, const uint32_t* n_items_dev, bool zero_status) {
    VerifyWs* ws = s.ws;
    uint32_t* d_small = ws->small.as<uint32_t>();
    RoundState& r = s.r;
    EvalArgs A;
    A.job = s.job; A.iter_samples = ws->iter_samples.as<uint32_t>(); A.it_begin = it_lo; A.it_end = it_hi;
    A.counts = ws->m_counts.as<int32_t>(); A.gate_m = ws->gate_m.as<uint32_t>(); A.work = d_small + 8;
    A.status = d_small + 12; A.deferred = ws->deferred.as<uint32_t>();
    A.stack_cap = kStackCap; A.lds_bytes = second ? kEvalLdsBig : eval_lds_small(s.job.n); A.from_deferred = second ? 1u : 0u;
    A.n_deferred = second ? r.n_def : 0u;
    SLOT_HIP(ws->stacks.reserve((size_t)std::max(std::max(it_hi - it_lo, r.n_def), 64u) * kStackCap * sizeof(uint16_t)));
    A.stacks = ws->stacks.as<uint16_t>();
    A.adjc_scratch = nullptr; A.dbg = nullptr; A.dbg_stride = 0; A.stop_level = 0; A.n_items_dev = n_items_dev;
    if (second) {
      SLOT_HIP(ws->adjc_scratch.reserve((size_t)r.n_def * kAdjcScratchWords * sizeof(u64)));
      A.adjc_scratch = ws->adjc_scratch.as<u64>();
      L.zero.push_back({nullptr, d_small + 8, 1u});
      L.eval_big.push_back(A);
      return;
    }
    if (zero_status) L.zero.push_back({nullptr, d_small + 8, 12u});
    // A few hypotheses of an object whose consensus lists (about all of its valid matches when the object is really there)
    // will not fit the 48 KB carve: straight to a whole CU's LDS instead of a first pass that only finds that out
    if (it_hi - it_lo <= 16u && gate_lds_bytes(r.nvalid) + 4096u > kEvalLdsSmall) {
      A.lds_bytes = kEvalLdsBig;
      L.eval_direct.push_back(A);
    } else {
      L.eval_small.push_back(A);
    }
  }

  // ---- issue: queue the kernels of the slot's next phase
  void issue(Slot& s) {
    VerifyWs* ws = s.ws;
    uint32_t* d_small = ws->small.as<uint32_t>();
    if (s.ph == PH_CLUSTER) {                               // ClusterPerObject, one launch (cluster_frame_kernel)
      ClusterArgs ca;
      ca.kp_xy = s.d_kp_xy; ca.cloud = s.use_depth ? nullptr : s.d_cloud; ca.depth = s.dep.d_depth; ca.counts = s.d_counts;
      ca.matches = s.d_matches; ca.mxyz = s.d_mxyz; ca.nq = nq; ca.k = k; ca.H = H; ca.Wimg = Wimg; ca.n_objs = n_objs; ca.qidx_add = 0u;
      ca.depth_is_u16 = s.dep.is_u16; ca.fx = s.dep.fx; ca.fy = s.dep.fy; ca.cx = s.dep.cx; ca.cy = s.dep.cy;
      ca.kept = ws->c_kept.as<uint32_t>(); ca.offs = ws->c_offs.as<uint32_t>(); ca.obj_of = ws->c_obj.as<uint32_t>();
      ca.src = ws->c_src.as<uint32_t>(); ca.hist = ws->c_hist.as<uint32_t>(); ca.goff = ws->c_goff.as<uint32_t>();
      ca.cnt = ws->c_cnt.as<uint32_t>(); ca.qpt = ws->c_qpt.as<float>();
      ca.train = ws->train.as<float>(); ca.query = ws->query.as<float>(); ca.kpxy = ws->kpxy.as<float>(); ca.qidx = ws->qidx.as<uint32_t>();
      ca.m_hist = ws->m_hist.as<uint32_t>(); ca.m_ctl = mail(s) + 60;
      L.cluster.push_back(ca);
      s.ph = PH_CLUSTER_WAIT;
      return;
    }
    if (s.ph == PH_GROUP) {
      L.copy_in.push_back({ws->m_goff.as<uint32_t>(), ws->c_goff.as<uint32_t>(), n_objs});
      GroupArgs ga = {s.n_all, ws->c_obj.as<uint32_t>(), ws->c_goff.as<uint32_t>(), ws->f_train.as<float>(), ws->f_query.as<float>(),
                      ws->f_qidx.as<uint32_t>(), ws->f_kp.as<float>(), ws->train.as<float>(), ws->query.as<float>(),
                      ws->qidx.as<uint32_t>(), ws->kpxy.as<float>()};
      L.group.push_back(ga);
      s.ph = PH_PREPALL;
    }
    if (s.ph == PH_PREPALL) {
      // FillAdjacency and the first round's validity/degree pass of EVERY object of the frame in this one tick: they
      // do not depend on the rand() stream, and an object with < 3 valid matches then costs no tick at all
      L.zero.push_back({nullptr, ws->nvalid.as<uint32_t>(), 4u * (uint32_t)std::max<size_t>(s.objs.size(), 1)});
      for (size_t i = 0; i < s.objs.size(); ++i) {
        if (s.objs[i].n < 3) continue;
        const ObjJob job = make_job(s, s.objs[i]);
        if (job.n <= 64u) {                                  // finite + adjacency + statistics by one wave
          PrepSmallArgs pa;
          pa.job = job; pa.stats = ws->nvalid.as<uint32_t>() + 4 * i; pa.span = spans[s.objs[i].obj]; pa.err = prm->sensor_error;
          L.prep_small.push_back(pa);
          continue;
        }
        L.finite.push_back({job});
        L.adj.push_back({job, spans[s.objs[i].obj], prm->sensor_error});
        L.prep.push_back({job, ws->nvalid.as<uint32_t>() + 4 * i, nullptr, 0u});
      }
      L.copy_out.push_back({ws->nvalid.as<uint32_t>(), ws->m_nvalid.as<uint32_t>(), 4u * (uint32_t)std::max<size_t>(s.objs.size(), 1)});
      s.ph = PH_PREPALL_WAIT;
      return;
    }
    while (s.ph == PH_OBJECT) {
      // Ransac returns no inliers for < 3 valid matches and draws nothing (:238-241)
      while (s.oi < s.objs.size() && s.objs[s.oi].n < 3) ++s.oi;
      if (s.oi >= s.objs.size()) { s.ph = PH_DONE; return; }
      if (sprint_on() && sprint_live(s.objs[s.oi])) { issue_sprint(s); return; }
      const ObjSpan& o = s.objs[s.oi];
      s.job = make_job(s, o);
      ctx->counters.last_objects_verified += 1;
      s.pending_invalidate = false;
      s.tr = todhip_round_trace();
      s.tr.object = o.obj; s.tr.draws_before = s.start_draws + s.abs_pos; s.tr.best_count = -INT_MAX;
      start_round(s, o.nvalid, o.degsum, o.triangle);       // -> PH_DRAW, or straight on to the next object
    }
    if (s.ph == PH_ROUND) {                                 // one AdjacencyRansac::Ransac call (GuessGenerator.cpp:192-231)
      if (s.pending_invalidate) {
        L.inval.push_back({s.job, ws->kp_bits.as<u64>(), obj_bits(s) + 6 * s.job.W, nullptr, 0u});
        s.pending_invalidate = false;
      }
      L.zero.push_back({nullptr, d_small, 64u});
      L.prep.push_back({s.job, d_small + 5, nullptr, 0u});   // words 5..7: |valid|, degree sum, triangle (1..4 = ChainOut)
      export_small(s);
      s.tr = todhip_round_trace();
      s.tr.object = s.objs[s.oi].obj; s.tr.draws_before = s.start_draws + s.abs_pos; s.tr.best_count = -INT_MAX;
      s.ph = PH_PREP_WAIT;
      return;
    }
    if (s.ph == PH_DRAW) {
      RoundState& r = s.r;
      // window = stream positions the requested iterations are expected to consume: 4 per iteration for a start
      // (3 draws + the odd failed attempt), then 1.5 x what this round's iterations consumed so far -- objects without
      // a consistent subset burn hundreds of draws per iteration in failed attempts, and a window sized for 4 made
      // them crawl through dozens of ticks
      const uint64_t seen_it = (uint64_t)r.it_begin + r.got;
      const uint64_t per_it = seen_it ? std::max<uint64_t>(4u, (3u * r.consumed / seen_it + 1u) / 2u + 1u) : 4u;
      r.S = (uint32_t)std::min<uint64_t>(std::max<uint64_t>(per_it * (r.want - r.got) + 512u, r.s_floor), 1u << 20);
      r.window_len = r.S + r.lookahead;
      SLOT_HIP(s.stream->ensure_device(s.abs_pos + r.consumed + r.window_len, st));
      SLOT_HIP(ws->table.reserve((size_t)r.S * sizeof(DrawEntry)));
      (s.job.W <= 2u ? L.draw_small : L.draw).push_back({s.job, nullptr, r.window_len, r.S, ws->table.as<DrawEntry>()});
      (s.job.W <= 2u ? L.draw_small_src : L.draw_src).push_back({s.stream, s.abs_pos + r.consumed});
      ChainArgs ca = {ws->table.as<DrawEntry>(), r.S, r.want - r.got, r.attempts_carry, r.it_begin + r.got,
                      ws->iter_samples.as<uint32_t>(), ws->m_pos.as<uint32_t>(), reinterpret_cast<ChainOut*>(d_small + 1)};
      L.chain.push_back(ca);
      // the evaluation of the iterations this walk draws rides in the same tick: its grid covers everything still wanted, and
      // the kernel takes the number that really exist from the walk's ChainOut. One host round trip less per evaluation batch.
      push_eval(s, false, r.it_begin + r.got, r.it_begin + r.want, d_small + 1, r.got == 0u);
      export_small(s);
      s.ph = PH_DRAW_WAIT;
      return;
    }
    if (s.ph == PH_EVAL2) {
      push_eval(s, true, 0u, 0u, nullptr, true);
      export_small(s);
      s.ph = PH_EVAL2_WAIT;
      return;
    }
    if (s.ph == PH_GROWTH) {                                // growth (adjacency_ransac.cpp:255-308)
      const uint32_t kp_words = (nq + 63u) / 64u, W = s.job.W;
      u64* d_bits = obj_bits(s);
      GrowthArgs ga = {s.job, ws->iter_samples.as<uint32_t>() + 3 * (size_t)s.r.best_it, prm->sensor_error, d_bits + 3 * W,
                       d_bits + 4 * W, d_bits + 5 * W, ws->m_kp.as<uint32_t>(), ws->kp_bits.as<u64>(), kp_words,
                       reinterpret_cast<GrowthOut*>(d_small + 32)};
      L.growth.push_back(ga);
      // If the pose is accepted (enough inlier keypoints, GuessGenerator.cpp:205-206) the object gets another round, which starts
      // with InvalidateQueryIndices and the validity / degree pass: both ride in this tick behind the growth, gated on the device
      // by the count the growth kernel writes, so an accepted pose costs no tick of its own. (Not accepted: they do nothing.)
      const uint32_t* gate = d_small + 32 + offsetof(GrowthOut, n_kp_inliers) / sizeof(uint32_t);
      L.zero.push_back({nullptr, d_small, 32u});           // words 5..7 (the pass's counters) among them; GrowthOut starts at 32
      L.inval_after.push_back({s.job, ws->kp_bits.as<u64>(), obj_bits(s) + 6 * s.job.W, gate, prm->min_inliers});
      L.prep_after.push_back({s.job, d_small + 5, gate, prm->min_inliers});
      export_small(s);
      s.ph = PH_GROWTH_WAIT;
      return;
    }
  }

  // ---- sprint_kernel: the live objects of at most kSprintN matches from s.oi up to the next live big object, in one launch
  static bool sprint_on() {
    static const bool on = [] { const char* e = getenv("TODHIP_VERIFY_SPRINT"); return !(e && e[0] == '0'); }();   // read once
    return on;
  }
  static bool sprint_live(const ObjSpan& o) {
    return o.n >= 3u && o.n <= kSprintN && (o.resume || (o.nvalid >= 3u && o.triangle != 0u));
  }
  void issue_sprint(Slot& s) {
    VerifyWs* ws = s.ws;
    SLOT_HIP(ws->m_sprint.reserve(kSprintMaxObjs * sizeof(SprintObj)));
    SLOT_HIP(ws->m_sprint_out.reserve((kSprintHdrWords + (size_t)kSprintMaxRecs * kSprintRecWords) * sizeof(uint32_t)));
    SLOT_HIP(ws->m_sprint_kp.reserve((size_t)kSprintMaxRecs * kSprintN * sizeof(uint32_t)));
    SLOT_HIP(ws->sprint_status.reserve(16 * sizeof(uint32_t)));
    SLOT_HIP(ws->sprint_stack.reserve((size_t)kSprintWaves * kSprintStackCap * sizeof(uint16_t)));
    SprintObj* list = ws->m_sprint.as<SprintObj>();
    s.sprint_members.clear();
    uint64_t skip = 0, total_skip = 0;
    for (size_t i = s.oi; i < s.objs.size() && s.sprint_members.size() < kSprintMaxObjs; ++i) {
      const ObjSpan& o = s.objs[i];
      if (o.n < 3u) continue;
      if (sprint_live(o)) {
        SprintObj so;
        so.job = make_job(s, o); so.skip = skip; so.index = (uint32_t)i; so.pad = 0;
        list[s.sprint_members.size()] = so;
        s.sprint_members.push_back(i);
        total_skip += skip; skip = 0;
        continue;
      }
      if (o.nvalid < 3u) continue;                           // no round, no draw (:238-241)
      if (o.triangle) break;                                 // a live big object: the sprint ends before it
      skip += (uint64_t)kMaxSampleChecks * ((uint64_t)o.nvalid + o.degsum / 2u);   // triangle-free: start_round
    }
    SLOT_HIP(s.stream->ensure_device(s.abs_pos + total_skip + s.sprint_margin, st));
    SprintArgs a;
    a.objs = list; a.rnd = nullptr; a.rnd_len = 0; a.pos0 = s.abs_pos; a.kceil = pool_of(ctx)->kceil.as<uint32_t>();
    a.out = ws->m_sprint_out.as<uint32_t>(); a.kp_out = ws->m_sprint_kp.as<uint32_t>();
    a.status = ws->sprint_status.as<uint32_t>(); a.stack = ws->sprint_stack.as<uint16_t>();
    a.n_objs = (uint32_t)s.sprint_members.size(); a.max_iterations = prm->n_ransac_iterations; a.min_inliers = prm->min_inliers;
    a.err = prm->sensor_error; a.rec_cap = kSprintMaxRecs; a.kp_cap = kSprintMaxRecs * kSprintN;
    a.out[0] = 0u; a.out[1] = SPRINT_ERROR; a.out[2] = 0u;   // (overwritten by the kernel)
    L.zero.push_back({nullptr, a.status, 8u});
    L.sprint.push_back(a);
    L.sprint_src.push_back(s.stream);                        // rnd / rnd_len are resolved at launch time (the shared stream may move)
    s.ph = PH_SPRINT_WAIT;
  }
  void consume_sprint(Slot& s) {
    VerifyWs* ws = s.ws;
    const uint32_t* out = ws->m_sprint_out.as<uint32_t>();
    const uint32_t* kp_out = ws->m_sprint_kp.as<uint32_t>();
    const uint32_t n_rec = out[0], reason = out[1], n_done = out[2];
    TOD_DBG("sprint: %zu objects, %u done, %u records, reason %u, gate calls %u, hypotheses %u, windows %u; ticks: ring %u attempt %u walk %u eval %u "
            "book %u growth %u all %u", s.sprint_members.size(), n_done, n_rec, reason, out[5], out[6], out[8], out[9], out[10], out[11], out[12],
            out[13], out[14], out[15]);
    if (reason == SPRINT_ERROR || n_rec > kSprintMaxRecs || n_done > s.sprint_members.size()) {
      if (tod_debug()) fprintf(stderr, "[todhip] sprint error: detail %u status %u\n", out[7], out[1]);
      fail(s, TODHIP_ESCRATCH);
      return;
    }
    ctx->counters.last_gate_calls += out[5];
    ctx->counters.last_hypotheses += out[6];
    ctx->counters.last_sprint_launches += 1;
    ctx->counters.last_sprint_rounds += n_rec;
    uint32_t ri = 0;
    for (uint32_t m = 0; m < s.sprint_members.size(); ++m) {
      const size_t idx = s.sprint_members[m];
      const bool complete = m < n_done;
      const bool touched = complete || (ri < n_rec && out[kSprintHdrWords + (size_t)ri * kSprintRecWords] == m);
      if (!touched) break;                                   // the wave stopped before this object: PH_OBJECT takes it from here
      // the objects the host decides without a kernel on the way (first round: fewer than 3 valid matches, or triangle-free)
      while (s.oi < idx) {
        const ObjSpan& t = s.objs[s.oi];
        if (t.n < 3u) { ++s.oi; continue; }
        s.job = make_job(s, t);
        ctx->counters.last_objects_verified += 1;
        s.pending_invalidate = false;
        s.tr = todhip_round_trace();
        s.tr.object = t.obj; s.tr.draws_before = s.start_draws + s.abs_pos; s.tr.best_count = -INT_MAX;
        start_round(s, t.nvalid, t.degsum, t.triangle);      // -> round_done: ++s.oi
        if (s.ph == PH_DONE) return;                         // (a failure)
      }
      ObjSpan& o = s.objs[idx];
      if (!o.counted) { ctx->counters.last_objects_verified += 1; o.counted = true; }
      for (; ri < n_rec; ++ri) {
        const uint32_t* rec = out + kSprintHdrWords + (size_t)ri * kSprintRecWords;
        if (rec[0] != m) break;
        const uint64_t consumed = ((uint64_t)rec[5] << 32) | rec[4];
        todhip_round_trace tr = todhip_round_trace();
        tr.object = o.obj; tr.draws_before = s.start_draws + s.abs_pos;
        tr.iterations = rec[1]; tr.best_iteration = rec[2]; tr.best_count = (int32_t)rec[3];
        s.abs_pos += consumed;
        const uint32_t n_kp = rec[7] ? rec[6] : 0u;
        ctx->counters.last_rounds += 1;
        tr.draws_after = s.start_draws + s.abs_pos; tr.n_inlier_kp = n_kp; tr.accepted = n_kp >= prm->min_inliers;
        s.traces.push_back(tr);
        if (n_kp >= prm->min_inliers) {                      // GuessGenerator.cpp:205-230
          if (rec[20] + n_kp > kSprintMaxRecs * kSprintN) { fail(s, TODHIP_ESCRATCH); return; }
          todhip_pose p;
          std::memset(&p, 0, sizeof(p));
          p.object = o.obj;
          std::memcpy(p.R, rec + 8, sizeof(p.R));
          std::memcpy(p.t, rec + 17, sizeof(p.t));
          p.inlier_begin = (uint32_t)s.inliers.size();
          s.inliers.insert(s.inliers.end(), kp_out + rec[20], kp_out + rec[20] + n_kp);
          p.inlier_end = (uint32_t)s.inliers.size();
          s.poses.push_back(p);
          ctx->counters.last_poses += 1;
        }
      }
      if (!complete) {                                       // relaunch from this object; the kernel recomputes its statistics
        o.resume = true;
        s.oi = idx;
        break;
      }
      s.oi = idx + 1;
    }
    if (reason == SPRINT_NEED_STREAM) {
      if (s.sprint_margin >= (1ull << 28)) { fail(s, TODHIP_ESCRATCH); return; }
      s.sprint_margin *= 4u;
    }
    s.sprint_members.clear();
    s.ph = PH_OBJECT;
  }

  // ---- one evaluation batch of computeModel starts: draw `want` iterations (as many windows as it takes)
  void begin_batch(Slot& s) {
    RoundState& r = s.r;
    r.it_begin = r.it_drawn;
    r.want = std::min(r.batch, r.total_iters - r.it_begin);
    r.got = 0;
    if (r.got < r.want && !r.selection_empty) s.ph = PH_DRAW; else after_draw(s);
  }
  void after_draw(Slot& s) {
    RoundState& r = s.r;
    r.it_drawn = r.it_begin + r.got;
    if (r.got > 0) eval_done(s, false); else replay(s);    // the iterations were evaluated in the ticks that drew them
  }
  // the first evaluation pass of a batch is complete (its status words are in the mailbox), or the second one
  void eval_done(Slot& s, bool second) {
    RoundState& r = s.r;
    const uint32_t* m = mail(s);
    if (!second) {
      ctx->counters.last_gate_calls += m[13];
      r.n_def = m[14];
      TOD_DBG2("  eval done: gate calls=%u deferred=%u", m[13], r.n_def);
      if (r.n_def > 0) { s.ph = PH_EVAL2; return; }
    }
    ctx->counters.last_hypotheses += r.got;
    replay(s);
  }
  bool eval_failed(Slot& s) {
    const uint32_t* m = mail(s);
    if (m[12] == 0) return false;
    if (tod_debug())
      fprintf(stderr, "[todhip] eval status %u: g=%u value=%u m=%u it=%u (n=%u W=%u)\n", m[12], m[16], m[17], m[18], m[19],
              s.job.n, s.job.W);
    fail(s, TODHIP_ESCRATCH);
    return true;
  }
  // ---- ransac.h:95-135 over the iterations known so far
  void replay(Slot& s) {
    RoundState& r = s.r;
    const int32_t* hc = s.ws->m_counts.as<int32_t>();
    const uint32_t* hp = s.ws->m_pos.as<uint32_t>();
    while (!r.loop_done) {
      if (!(r.iterations < r.k)) { r.loop_done = true; r.pos_after_stop = r.iterations > 0 ? hp[r.iterations - 1] : 0; break; }
      if ((uint32_t)r.iterations >= r.it_drawn) {
        if (r.selection_empty) { r.loop_done = true; r.pos_after_stop = r.consumed; }   // selection.empty() -> break (:100-101)
        break;                                              // need more iterations
      }
      const int n_count = hc[r.iterations];
      if (n_count > r.n_best) {
        r.n_best = n_count;
        r.best_it = (uint32_t)r.iterations;
        const double w = (double)r.n_best / (double)r.nvalid;
        double p_no_outliers = 1.0 - std::pow(w, 3.0);
        p_no_outliers = std::max(std::numeric_limits<double>::epsilon(), p_no_outliers);
        p_no_outliers = std::min(1.0 - std::numeric_limits<double>::epsilon(), p_no_outliers);
        r.k = std::log(1.0 - 0.99) / std::log(p_no_outliers);
      }
      ++r.iterations;
      if (r.iterations > (int)prm->n_ransac_iterations) { r.loop_done = true; r.pos_after_stop = hp[r.iterations - 1]; }
    }
    // next batch = what the loop still needs given the best model so far (k of :130): a hopeless object (k >> the
    // iteration budget) gets all of its remaining iterations evaluated at once instead of in 16/64/256/1024 steps
    {
      const double need = r.k - (double)r.iterations;
      const uint32_t want = need >= (double)kMaxEvalWaves ? kMaxEvalWaves : (uint32_t)std::max(1.0, std::ceil(need));
      r.batch = std::max(std::min<uint32_t>(r.batch * 4u, kMaxEvalWaves), std::min(want, kMaxEvalWaves));
    }
    if (!r.loop_done) { begin_batch(s); return; }
    // advance the caller's generator by exactly the draws the reference would have consumed
    s.abs_pos += r.pos_after_stop;
    s.tr.iterations = (uint32_t)r.iterations; s.tr.best_iteration = r.best_it; s.tr.best_count = r.n_best;
    if (r.n_best <= 0) { round_done(s, false); return; }   // inliers_.empty(): computeModel() == false (:137-138)
    s.ph = PH_GROWTH;
  }
  void round_done(Slot& s, bool have_pose) {
    const GrowthOut* go = reinterpret_cast<const GrowthOut*>(mail(s) + 32);
    const uint32_t n_kp = have_pose ? go->n_kp_inliers : 0u;
    ctx->counters.last_rounds += 1;
    s.tr.draws_after = s.start_draws + s.abs_pos; s.tr.n_inlier_kp = n_kp; s.tr.accepted = n_kp >= prm->min_inliers;
    s.traces.push_back(s.tr);
    if (n_kp < prm->min_inliers) { ++s.oi; s.ph = PH_OBJECT; return; }   // GuessGenerator.cpp:205-206
    todhip_pose p;
    std::memset(&p, 0, sizeof(p));
    p.object = s.objs[s.oi].obj;
    std::memcpy(p.R, go->R, sizeof(p.R));
    std::memcpy(p.t, go->T, sizeof(p.t));
    p.inlier_begin = (uint32_t)s.inliers.size();
    const uint32_t* list = s.ws->m_kp.as<uint32_t>();
    s.inliers.insert(s.inliers.end(), list, list + n_kp);
    p.inlier_end = (uint32_t)s.inliers.size();
    s.poses.push_back(p);
    ctx->counters.last_poses += 1;
    // InvalidateQueryIndices and the next round's validity pass (:207-230) ran behind the growth kernel in this tick
    s.pending_invalidate = false;
    s.tr = todhip_round_trace();
    s.tr.object = s.objs[s.oi].obj; s.tr.draws_before = s.start_draws + s.abs_pos; s.tr.best_count = -INT_MAX;
    const uint32_t* m = mail(s);
    start_round(s, m[5], m[6], m[7]);
  }

  // ---- consume: the tick's results are in the mailbox
  void consume(Slot& s) {
    VerifyWs* ws = s.ws;
    const uint32_t* m = mail(s);
    RoundState& r = s.r;
    if (s.ph == PH_CLUSTER_WAIT) {
      if (m[60] != 0) { fail(s, TODHIP_ERANGE); return; }
      s.n_all = m[64];
      if (s.n_all == 0) { s.ph = PH_DONE; return; }
      const uint32_t* hist = ws->m_hist.as<uint32_t>();
      uint32_t* goff = ws->m_goff.as<uint32_t>();
      uint32_t total = 0, max_n = 0;
      s.objs.clear();
      for (uint32_t o = 0; o < n_objs; ++o) {
        goff[o] = total;
        if (hist[o]) s.objs.push_back({o, total, hist[o]});
        total += hist[o];
        max_n = std::max(max_n, hist[o]);
      }
      s.oi = 0;
      if (!reserve_objects(s, max_n)) return;
      s.ph = PH_PREPALL;                                     // (the grouping by object rode in the cluster launch)
      return;
    }
    if (s.ph == PH_PREPALL_WAIT) {
      const uint32_t* nv = ws->m_nvalid.as<uint32_t>();
      for (size_t i = 0; i < s.objs.size(); ++i) {
        s.objs[i].nvalid = nv[4 * i]; s.objs[i].degsum = nv[4 * i + 1]; s.objs[i].triangle = nv[4 * i + 2];
      }
      s.ph = PH_OBJECT;
      return;
    }
    if (s.ph == PH_PREP_WAIT) {                             // a further round of the same object, after an accepted pose
      start_round(s, m[5], m[6], m[7]);
      return;
    }
    if (s.ph == PH_DRAW_WAIT) {
      if (eval_failed(s)) return;                           // (the evaluation of this walk's iterations ran in the same tick)
      const ChainOut co = *reinterpret_cast<const ChainOut*>(m + 1);
      TOD_DBG2("  draw window: S=%u len=%u -> done=%u pos_end=%u attempts=%u flag=%u", r.S, r.window_len, co.n_done, co.pos_end,
              co.attempts, co.flag);
      uint32_t* hp = ws->m_pos.as<uint32_t>();             // positions of this walk are relative to the window start
      for (uint32_t i = 0; i < co.n_done; ++i) hp[r.it_begin + r.got + i] += (uint32_t)r.consumed;
      r.got += co.n_done;
      r.consumed += co.pos_end;
      r.attempts_carry = co.attempts;
      if (co.flag == 2) {
        r.selection_empty = true;
        // the walk that just gave up consumed r.consumed draws in all: the next object's first window covers that much
        uint32_t hint = 1024u;
        while (hint < r.consumed + 512u && hint < (1u << 20)) hint <<= 1;
        s.s_hint = std::max(s.s_hint, hint);
      }
      if (co.flag == 1) r.s_floor = std::min<uint32_t>(std::max(r.S, 1024u) * 4u, 1u << 20);   // e.g. 1000 failing attempts in a row
      if (co.flag == 1 && co.n_done == 0 && co.pos_end == 0) {
        // a single attempt longer than the window: enlarge the look-ahead, give up beyond 64M draws
        if (r.lookahead >= (1u << 26)) { fail(s, TODHIP_ESCRATCH); return; }
        r.lookahead *= 4u;
      }
      if (r.got < r.want && !r.selection_empty) s.ph = PH_DRAW; else after_draw(s);
      return;
    }
    if (s.ph == PH_EVAL2_WAIT) {
      if (eval_failed(s)) return;
      eval_done(s, true);
      return;
    }
    if (s.ph == PH_SPRINT_WAIT) { consume_sprint(s); return; }
    if (s.ph == PH_GROWTH_WAIT) {
      const GrowthOut* go = reinterpret_cast<const GrowthOut*>(m + 32);
      TOD_DBG2("  growth: model=%u matches=%u kps=%u passes=%u", go->n_model_inliers, go->n_match_inliers, go->n_kp_inliers,
              go->passes);
      round_done(s, true);
      return;
    }
  }

  bool reserve_objects(Slot& s, uint32_t max_n) {
    VerifyWs* ws = s.ws;
    if (max_n > (uint32_t)kMaxWords * 64u) { fail(s, TODHIP_ESCRATCH); return false; }
    uint64_t adj = 0, bits = 0, deg = 0;
    for (ObjSpan& o : s.objs) {
      if (o.n < 3) continue;
      const uint64_t W = (o.n + 63u) / 64u;
      o.adj_off = adj; o.bits_off = (uint32_t)bits; o.deg_off = (uint32_t)deg;
      adj += (uint64_t)o.n * W; bits += 8u * W; deg += o.n;
    }
    const size_t n_objs_here = std::max<size_t>(s.objs.size(), 1);
    if (ws->phys.reserve((size_t)std::max<uint64_t>(adj, 1) * 8) != hipSuccess || ws->samp.reserve((size_t)std::max<uint64_t>(adj, 1) * 8) != hipSuccess ||
        ws->bits.reserve((size_t)std::max<uint64_t>(bits, 8) * 8) != hipSuccess || ws->sampdeg.reserve((size_t)std::max<uint64_t>(deg, 1) * 4) != hipSuccess ||
        ws->nvalid.reserve(n_objs_here * 16) != hipSuccess || ws->m_nvalid.reserve(n_objs_here * 16) != hipSuccess) {
      fail(s, TODHIP_EHIP);
      return false;
    }
    return true;
  }

  int reserve_common(Slot& s) {
    VerifyWs* ws = s.ws;
    const uint32_t kp_words = (nq + 63u) / 64u;
    TOD_HIP(ws->small.reserve(256 * sizeof(uint32_t)));
    TOD_HIP(ws->m_small.reserve(kMailSmallWords * sizeof(uint32_t)));
    TOD_HIP(ws->kp_bits.reserve((size_t)(kp_words + 1) * sizeof(u64)));
    TOD_HIP(ws->m_kp.reserve((size_t)std::max(nq, 1u) * sizeof(uint32_t)));
    VerifyPool* pool = pool_of(ctx);
    if (!pool->kceil_ready) {
      // ransac.h:121-130 with libm, once: k = log(1 - 0.99) / log(1 - w^3), w = n_best / |valid|; `iterations_ < k` (:95) is
      // `iterations_ < ceil(k)` for an integer iterations_, so the device replays the loop test exactly from this table
      std::vector<uint32_t> tab(65u * 65u, 1u);
      for (uint32_t nv = 1; nv <= 64u; ++nv)
        for (uint32_t nb = 0; nb <= 64u; ++nb) {
          const double w = (double)(int)nb / (double)nv;
          double p_no_outliers = 1.0 - std::pow(w, 3.0);
          p_no_outliers = std::max(std::numeric_limits<double>::epsilon(), p_no_outliers);
          p_no_outliers = std::min(1.0 - std::numeric_limits<double>::epsilon(), p_no_outliers);
          const double k = std::log(1.0 - 0.99) / std::log(p_no_outliers);
          const double c = std::ceil(k);
          tab[nv * 65u + nb] = c >= 2147483647.0 ? 0x7FFFFFFFu : (uint32_t)c;
        }
      TOD_HIP(pool->kceil.reserve(tab.size() * sizeof(uint32_t)));
      TOD_HIP(hipMemcpy(pool->kceil.p, tab.data(), tab.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
      pool->kceil_ready = true;
    }
    return TODHIP_OK;
  }
  int reserve_cluster(Slot& s) {
    VerifyWs* ws = s.ws;
    const size_t cap = (size_t)nq * k;
    TOD_HIP(ws->c_kept.reserve((size_t)nq * 4)); TOD_HIP(ws->c_offs.reserve(((size_t)nq + 1) * 4));
    TOD_HIP(ws->c_qpt.reserve((size_t)nq * 12)); TOD_HIP(ws->c_obj.reserve(cap * 4));
    TOD_HIP(ws->c_hist.reserve((size_t)n_objs * 4)); TOD_HIP(ws->c_goff.reserve((size_t)n_objs * 4));
    TOD_HIP(ws->c_src.reserve(cap * 4)); TOD_HIP(ws->c_cnt.reserve((size_t)n_objs * 4));
    TOD_HIP(ws->m_hist.reserve((size_t)n_objs * 4)); TOD_HIP(ws->m_goff.reserve((size_t)n_objs * 4));
    TOD_HIP(ws->f_train.reserve(cap * 12)); TOD_HIP(ws->f_query.reserve(cap * 12));
    TOD_HIP(ws->f_qidx.reserve(cap * 4)); TOD_HIP(ws->f_kp.reserve(cap * 8));
    TOD_HIP(ws->train.reserve(cap * 12)); TOD_HIP(ws->query.reserve(cap * 12));
    TOD_HIP(ws->qidx.reserve(cap * 4)); TOD_HIP(ws->kpxy.reserve(cap * 8));
    return TODHIP_OK;
  }

  // first-pass LDS per hypothesis: the induced graph has at most n vertices, so a small object does not need the
  // whole 48 KB carve (adjacency + colouring scratch + 4 KB of level stack) and more hypotheses fit a CU at once
  static uint32_t eval_lds_small(uint32_t n) {
    const uint32_t W = (n + 63u) / 64u;
    const uint32_t want = gate_lds_bytes(n) + 8u * n * W + 4096u;
    return std::min(kEvalLdsSmall, std::max(8192u, (want + 1023u) & ~1023u));
  }

  // A list of thousands of argument sets (the all-objects preparation tick of a batch: ~190 objects per frame): the sets go to
  // device memory in one copy, ordered by size class so that a launch's grid -- the extent of its largest member times the
  // count -- stays tight, and each class is ONE launch (252 launches of <= 36 sets each became 3-4 per kernel). `used` = bytes
  // of the staging area already taken in this tick.
  template <class A, class KernP, class Extent>
  bool launch_many(hipStream_t st, KernP kern, const std::vector<A>& v, int slot_dim, Extent extent, size_t& used, uint32_t block = 256u,
                   uint32_t lds = 0u, bool by_class = true) {
    VerifyPool* pool = pool_of(ctx);
    HostBuf& args_stage = pool->args_stage[lane_now];
    DevBuf& args_dev = pool->args_dev[lane_now];
    const size_t bytes = v.size() * sizeof(A);
    used = (used + 255u) & ~(size_t)255u;
    if (used + bytes > args_stage.cap || used + bytes > args_dev.cap) return false;
    static const uint32_t kClassBy[] = {16u, 64u, 256u, 0xFFFFFFFFu}, kClassOne[] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
    const uint32_t* kClass = by_class ? kClassBy : kClassOne;
    A* host = reinterpret_cast<A*>(reinterpret_cast<unsigned char*>(args_stage.p) + used);
    const A* dev = reinterpret_cast<const A*>(reinterpret_cast<const unsigned char*>(args_dev.p) + used);
    size_t n_cls[4] = {0, 0, 0, 0}, at = 0;
    auto cls_of = [&](const A& a) { uint32_t c = 0; while (a.job.n > kClass[c]) ++c; return c; };
    for (const A& a : v) ++n_cls[cls_of(a)];
    size_t start[4], fill[4];
    for (int c = 0; c < 4; ++c) { start[c] = fill[c] = at; at += n_cls[c]; }
    for (const A& a : v) host[fill[cls_of(a)]++] = a;
    if (hipMemcpyAsync(const_cast<A*>(dev), host, bytes, hipMemcpyHostToDevice, st) != hipSuccess) return false;
    used += bytes;
    for (int c = 0; c < 4; ++c) {
      for (size_t i0 = 0; i0 < n_cls[c]; i0 += 32768u) {    // (grid z / y limit)
        const uint32_t n = (uint32_t)std::min<size_t>(32768u, n_cls[c] - i0);
        uint32_t gx = 1, gy = 1;
        for (uint32_t i = 0; i < n; ++i) { const dim3 e = extent(host[start[c] + i0 + i]); gx = std::max(gx, e.x); gy = std::max(gy, e.y); }
        const dim3 grid = slot_dim == 1 ? dim3(gx, n) : dim3(gx, gy, n);
        SlotsPtr<A> S = {dev + start[c] + i0};
        hipLaunchKernelGGL(kern, grid, dim3(block), lds, st, S);
      }
    }
    return true;
  }

  template <class F>
  static void split_evals(const std::vector<EvalArgs>& v, uint32_t lds, bool deferred, F& launch_evals) {
    bool any_wide = false, any_narrow = false;
    for (const EvalArgs& a : v) (a.job.W <= 8u ? any_narrow : any_wide) = true;
    if (any_wide && any_narrow) { launch_evals(v, true, lds, deferred); return; }
    launch_evals(v, any_wide, lds, deferred);
  }
  void launch_all(hipStream_t st) {
    // the staging area for argument sets in device memory (launch_many) is sized once per tick, before anything reads it
    size_t used = 0;
    bool stage_ok = false;
    {
      const size_t n_eval = L.eval_small.size() + L.eval_direct.size() + L.eval_big.size();
      if (L.adj.size() > 4u * kManySlots || n_eval > kMaxSlots || L.prep_small.size() > kManySlots) {
        VerifyPool* pool = pool_of(ctx);
        const size_t need = L.finite.size() * sizeof(JobArgs) + L.adj.size() * sizeof(AdjArgs) + L.prep.size() * sizeof(PrepArgs) +
                            L.prep_small.size() * sizeof(PrepSmallArgs) + n_eval * sizeof(EvalArgs) + 4096u;
        stage_ok = pool->args_stage[lane_now].reserve(need) == hipSuccess && pool->args_dev[lane_now].reserve(need) == hipSuccess;   // (the lane is idle)
      }
    }
    // more hypothesis evaluations than fit one launch's arguments (a batch of more than 16 frames): one launch all the same
    auto launch_evals = [&](const std::vector<EvalArgs>& v, bool wide, uint32_t lds, bool deferred) {
      auto ext_it = [](const EvalArgs& a) { return dim3(a.it_end - a.it_begin); };
      auto ext_def = [](const EvalArgs& a) { return dim3(a.n_deferred); };
      if (v.size() > kMaxSlots && stage_ok) {
        const bool ok = wide ? (deferred ? launch_many(st, eval_kernel<true, SlotsPtr<EvalArgs>>, v, 1, ext_def, used, 64u, lds, false)
                                         : launch_many(st, eval_kernel<true, SlotsPtr<EvalArgs>>, v, 1, ext_it, used, 64u, lds, false))
                             : (deferred ? launch_many(st, eval_kernel<false, SlotsPtr<EvalArgs>>, v, 1, ext_def, used, 64u, lds, false)
                                         : launch_many(st, eval_kernel<false, SlotsPtr<EvalArgs>>, v, 1, ext_it, used, 64u, lds, false));
        if (ok) return;
      }
      if (wide) { if (deferred) launch_list(st, eval_kernel<true>, v, 64, lds, 1, ext_def); else launch_list(st, eval_kernel<true>, v, 64, lds, 1, ext_it); }
      else { if (deferred) launch_list(st, eval_kernel<false>, v, 64, lds, 1, ext_def); else launch_list(st, eval_kernel<false>, v, 64, lds, 1, ext_it); }
    };
    auto words = [](const CopyArgs& a) { return dim3(std::max(1u, std::min(64u, (a.n + 255u) / 256u))); };
    L.copy_in.insert(L.copy_in.end(), L.zero.begin(), L.zero.end());   // both precede every other kernel of the tick: one launch
    launch_list<kCopySlots>(st, copy_words_kernel, L.copy_in, 256, 0, 1, words);
    for (size_t i = 0; i < L.sprint.size(); ++i) { L.sprint[i].rnd = L.sprint_src[i]->dev.as<uint32_t>(); L.sprint[i].rnd_len = L.sprint_src[i]->dev_valid; }
    launch_list<kWideSlots>(st, sprint_kernel, L.sprint, kSprintThreads, kSprintLds, 0, [](const SprintArgs&) { return dim3(1); });
    launch_list(st, cluster_frame_kernel, L.cluster, 256, 0, 0, [](const ClusterArgs&) { return dim3(1); });
    launch_list(st, cluster_lookup_kernel, L.lookup, 256, 0, 1, [](const LookupArgs& a) { return dim3((a.nq + 255u) / 256u); });
    launch_list(st, cluster_scan_kernel, L.scan, 256, 0, 0, [](const ScanArgs&) { return dim3(1); });
    launch_list(st, cluster_scatter_kernel, L.scatter, 256, 0, 1,
                [](const ScatterArgs& a) { return dim3((uint32_t)(((size_t)a.nq * a.k + 255u) / 256u)); });
    launch_list(st, cluster_group_kernel, L.group, 256, 0, 1, [](const GroupArgs& a) { return dim3((a.n_all + 255u) / 256u); });
    launch_list<kWideSlots>(st, invalidate_kernel, L.inval, 256, 0, 0, [](const InvArgs&) { return dim3(1); });
    {
      auto one = [](const PrepSmallArgs&) { return dim3(1); };
      if (!(L.prep_small.size() > kManySlots && stage_ok &&
            launch_many(st, small_prep_kernel<SlotsPtr<PrepSmallArgs>>, L.prep_small, 1, one, used, 64u, 0u, false)))
        launch_list<kManySlots>(st, small_prep_kernel<Slots<PrepSmallArgs, kManySlots>>, L.prep_small, 64, 0, 1, one);
    }
    {
      auto ext_rows = [](const auto& a) { return dim3((a.job.n + 255u) / 256u); };
      auto ext_adj = [](const AdjArgs& a) { return dim3(a.job.n, (a.job.W + 3u) / 4u); };
      bool many = L.adj.size() > 4u * kManySlots;
      if (many) many = stage_ok;
      if (!many || !launch_many(st, finite_kernel<SlotsPtr<JobArgs>>, L.finite, 1, ext_rows, used))
        launch_list<kManySlots>(st, finite_kernel<Slots<JobArgs, kManySlots>>, L.finite, 256, 0, 1, ext_rows);
      if (!many || !launch_many(st, adjacency_kernel<SlotsPtr<AdjArgs>>, L.adj, 2, ext_adj, used))
        launch_list<kManySlots>(st, adjacency_kernel<Slots<AdjArgs, kManySlots>>, L.adj, 256, 0, 2, ext_adj);
      if (!many || !launch_many(st, round_prep_kernel<SlotsPtr<PrepArgs>>, L.prep, 1, ext_rows, used))
        launch_list<kManySlots>(st, round_prep_kernel<Slots<PrepArgs, kManySlots>>, L.prep, 256, 0, 1, ext_rows);
    }
    for (size_t i = 0; i < L.draw.size(); ++i) L.draw[i].rnd = L.draw_src[i].first->dev.as<uint32_t>() + L.draw_src[i].second;
    for (size_t i = 0; i < L.draw_small.size(); ++i)
      L.draw_small[i].rnd = L.draw_small_src[i].first->dev.as<uint32_t>() + L.draw_small_src[i].second;
    launch_list<kWideSlots>(st, draw_table_kernel, L.draw, 256, 0, 1, [](const DrawArgs& a) { return dim3((a.S + 3u) / 4u); });
    launch_list<kWideSlots>(st, draw_table_small_kernel, L.draw_small, 256, 0, 1, [](const DrawArgs& a) { return dim3((a.S + 255u) / 256u); });
    {
      // dynamic LDS: the packed hop words of the largest window of the launch + the per-iteration start positions
      uint32_t lds = 0;
      for (const ChainArgs& c : L.chain) lds = std::max(lds, (std::min(c.S, kChainLdsEntries) + std::min(c.n_req, kChainMaxReq)) * 4u);
      launch_list<kWideSlots>(st, chain_kernel, L.chain, 256, lds, 0, [](const ChainArgs&) { return dim3(1); });
    }
    {
      // one dynamic LDS size per launch: the largest any slot of the launch wants; every slot carves that much
      uint32_t lds = 8192u;
      for (const EvalArgs& a : L.eval_small) lds = std::max(lds, a.lds_bytes);
      for (EvalArgs& a : L.eval_small) a.lds_bytes = lds;
      // objects of more than 512 matches need the kernel's wide instantiation. It serves the smaller ones as well (6 % slower than
      // their own instantiation): when a tick holds both kinds -- the frames of a batch reach objects of 340 and of 590 matches
      // together -- one launch for all of them instead of two in a row, each as long as its slowest clique search
      split_evals(L.eval_small, lds, false, launch_evals);
    }
    split_evals(L.eval_direct, kEvalLdsBig, false, launch_evals);
    split_evals(L.eval_big, kEvalLdsBig, true, launch_evals);
    launch_list(st, growth_kernel, L.growth, 256, 0, 0, [](const GrowthArgs&) { return dim3(1); });
    launch_list<kWideSlots>(st, invalidate_kernel, L.inval_after, 256, 0, 0, [](const InvArgs&) { return dim3(1); });
    launch_list<kManySlots>(st, round_prep_kernel<Slots<PrepArgs, kManySlots>>, L.prep_after, 256, 0, 1, [](const PrepArgs& a) { return dim3((a.job.n + 255u) / 256u); });
    launch_list<kCopySlots>(st, copy_words_kernel, L.copy_out, 256, 0, 1, words);
    L = Launches();
  }

  // slots: live frames (phase set by the caller). Returns the first slot error, if any.
  int run(std::vector<Slot*>& slots) {
    // stream caches live in the context (a harness that restarts rand() per frame reuses one stream for ever); a
    // few of the most recent start states are kept
    std::vector<StreamCache*>& caches = pool_of(ctx)->streams;
    for (Slot* s : slots) {
      s->start_draws = s->rng->draws; s->abs_pos = 0; s->stream = nullptr;
      for (StreamCache* c : caches) if (c->same_start(*s->rng)) { s->stream = c; break; }
      if (!s->stream) {
        if (caches.size() >= 64) {                          // none of the live slots can be using the oldest ones
          bool in_use = false;
          for (Slot* t : slots) in_use = in_use || t->stream == caches.front();
          if (!in_use) { TOD_HIP(hipStreamSynchronize(st)); caches.front()->dev.release(); delete caches.front(); caches.erase(caches.begin()); }
        }
        caches.push_back(new StreamCache(*s->rng));
        s->stream = caches.back();
      }
    }
    const int rc_run = run_ticks(slots);
    for (Slot* s : slots) {                                 // the caller's generator ends where the reference's would
      if (s->abs_pos) { const uint64_t d0 = s->start_draws; *s->rng = s->stream->state_at(s->abs_pos); s->rng->draws = d0 + s->abs_pos; }
      s->stream = nullptr;
    }
    return rc_run;
  }
  // A launch group = the slots that are ready at one moment and in the same kind of phase: each issues the kernels of its next
  // phase, the lists are launched once for all of them on an idle LANE (the context's stream, or one of the process's few side
  // streams), an event is recorded, and when it has fired every slot of the group consumes its results. The host never blocks on one
  // group while another could be consumed or launched: the frames of a batch reach their sprints, their big object's clique gates
  // (a single wave for most of a millisecond) and its growth at different moments, and a kernel of one kind queued behind a long one
  // of another kind on the same stream would wait for it. With one lane (a lone frame, or more than two batches in the air in this
  // process: their contexts' streams already overlap each other) this is the plain lock-step tick: everything ready in one launch,
  // one wait. Nothing here changes what a slot computes or in which order it consumes it.
  static constexpr uint32_t kHeavyN = 96;                  // matches of an object from which its evaluation / growth is a kind of its own
  enum Kind { K_LIGHT = 0, K_GROWTH = 1, K_SPRINT = 2, K_EVAL = 3, K_COUNT = 4 };
  static Kind kind_of(const Slot& s) {
    if (s.ph == PH_OBJECT) {                                // its next launch: a sprint, or the first evaluation of a big object
      size_t i = s.oi;                                      // (the objects PH_OBJECT decides without a kernel are skipped)
      while (i < s.objs.size() && !s.objs[i].resume && (s.objs[i].n < 3u || s.objs[i].nvalid < 3u || !s.objs[i].triangle)) ++i;
      if (i >= s.objs.size()) return K_LIGHT;
      if (sprint_on() && sprint_live(s.objs[i])) return K_SPRINT;
      return s.objs[i].n >= kHeavyN ? K_EVAL : K_LIGHT;
    }
    if ((s.ph == PH_DRAW || s.ph == PH_EVAL2) && s.job.n >= kHeavyN) return K_EVAL;
    if (s.ph == PH_GROWTH && s.job.n >= kHeavyN) return K_GROWTH;
    return K_LIGHT;
  }
  static uint32_t n_side_streams() {
    static const uint32_t n = [] {
      const char* e = getenv("TODHIP_VERIFY_FLIGHTS");
      const long v = e ? strtol(e, nullptr, 10) : 2;
      return (uint32_t)std::min<long>(std::max<long>(v, 0), 16);
    }();
    return n;
  }
  void describe(char* what, size_t cap) const {
    snprintf(what, cap, "lookup %zu adj %zu prep %zu draw %zu+%zu chain %zu eval %zu+%zu growth %zu inval %zu sprint %zu", L.lookup.size() + L.cluster.size(),
             L.adj.size() + L.prep_small.size(), L.prep.size() + L.prep_small.size(), L.draw.size(), L.draw_small.size(), L.chain.size(), L.eval_small.size() + L.eval_direct.size(),
             L.eval_big.size(), L.growth.size(), L.inval.size(), L.sprint.size());
  }
  struct Lane {
    hipStream_t st; hipEvent_t ev; std::atomic<bool>* taken;   // taken: a side stream of the process, claimed while a group is on it
    std::vector<Slot*> slots; bool busy = false; std::chrono::steady_clock::time_point t0; char what[128];
  };
  int run_ticks(std::vector<Slot*>& slots) {
    struct InAir { InAir() { n = g_batches_in_air.fetch_add(1) + 1; } ~InAir() { g_batches_in_air.fetch_sub(1); } int n; } in_air;
    VerifyPool* pool = pool_of(ctx);
    std::vector<hipStream_t> side;
    SideStreams::PerDevice* pd = nullptr;
    if (slots.size() > 1 && in_air.n <= 2 && n_side_streams() > 0) TOD_HIP(g_side_streams.get(ctx->device, n_side_streams(), side, &pd));
    while (pool->side_ev.size() < side.size() + 1) {
      hipEvent_t e2;
      TOD_HIP(hipEventCreateWithFlags(&e2, hipEventDisableTiming));
      pool->side_ev.push_back(e2);
    }
    std::vector<Lane> lanes(1 + side.size());
    lanes[0].st = st; lanes[0].ev = pool->side_ev[0]; lanes[0].taken = nullptr;
    for (size_t i = 0; i < side.size(); ++i) { lanes[1 + i].st = side[i]; lanes[1 + i].ev = pool->side_ev[1 + i]; lanes[1 + i].taken = &pd->taken[i]; }
    auto drain = [&]() {
      for (Lane& ln : lanes) if (ln.busy) { (void)hipEventSynchronize(ln.ev); ln.busy = false; if (ln.taken) ln.taken->store(false); }
    };
#define LOOP_HIP(expr) do { if ((expr) != hipSuccess) { drain(); return TODHIP_EHIP; } } while (0)
    std::vector<Slot*> ready[K_COUNT];
    while (true) {
      // groups that have landed: their slots consume and are ready again
      bool any_busy = false;
      for (Lane& ln : lanes) {
        if (!ln.busy) continue;
        const hipError_t q = hipEventQuery(ln.ev);
        if (q == hipErrorNotReady) { any_busy = true; continue; }
        LOOP_HIP(q);
        ln.busy = false;
        if (ln.taken) ln.taken->store(false);
        if (tod_debug())
          TOD_DBG("tick %.1f us: %s (lane %zu, %zu slots)", std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - ln.t0).count(),
                  ln.what, (size_t)(&ln - lanes.data()), ln.slots.size());
        for (Slot* s : ln.slots) { s->in_flight = false; if (s->ph != PH_DONE) consume(*s); }
        ln.slots.clear();
        TOD_DBG("consumed");
      }
      for (auto& r : ready) r.clear();
      size_t n_ready = 0;
      for (Slot* s : slots) {
        if (s->ph == PH_DONE || s->in_flight) continue;
        ready[lanes.size() > 1 ? kind_of(*s) : K_LIGHT].push_back(s);
        ++n_ready;
      }
      if (n_ready == 0 && !any_busy) break;
      // every kind of ready slots takes an idle lane of its own; a kind that finds none waits for the next landing
      bool launched = false;
      for (int k = 0; k < K_COUNT && n_ready; ++k) {
        if (ready[k].empty()) continue;
        Lane* ln = nullptr;
        for (Lane& c : lanes) {
          if (c.busy) continue;
          if (c.taken) { bool expected = false; if (!c.taken->compare_exchange_strong(expected, true)) continue; }   // another context has it
          ln = &c;
          break;
        }
        if (!ln) break;
        lane_now = (size_t)(ln - lanes.data());
        for (Slot* s : ready[k]) {
          issue(*s);
          if (s->ph != PH_DONE) { s->in_flight = true; ln->slots.push_back(s); }
        }
        if (ln->slots.empty()) { L = Launches(); if (ln->taken) ln->taken->store(false); launched = true; continue; }   // (they finished without a kernel)
        if (tod_debug()) { describe(ln->what, sizeof(ln->what)); ln->t0 = std::chrono::steady_clock::now(); }
        launch_all(ln->st);
        const hipError_t e1 = hipGetLastError();
        const hipError_t e2 = e1 == hipSuccess ? hipEventRecord(ln->ev, ln->st) : e1;
        if (e2 != hipSuccess) { if (ln->taken) ln->taken->store(false); for (Slot* s : ln->slots) s->in_flight = false; ln->slots.clear(); drain(); return TODHIP_EHIP; }
        ln->busy = true;
        launched = true;
        ctx->counters.last_verify_ticks += 1;
      }
      if (launched) continue;                                // (slots that finished without a kernel may be ready for more)
      // nothing could be launched: wait for the first group to land
      bool landed = false;
      while (!landed) {
        bool busy_now = false;
        for (Lane& ln : lanes) {
          if (!ln.busy) continue;
          busy_now = true;
          const hipError_t q = hipEventQuery(ln.ev);
          if (q == hipSuccess) { landed = true; break; }
          if (q != hipErrorNotReady) LOOP_HIP(q);
        }
        if (!busy_now) break;                                // every lane is idle (another context holds the side streams): try again
        if (!landed) std::this_thread::yield();
      }
    }
#undef LOOP_HIP
    for (Slot* s : slots)
      if (s->rc != TODHIP_OK) return s->rc;
    return TODHIP_OK;
  }
#undef SLOT_HIP
};

}  // namespace

// ClusterPerObject of F frames without a cloud, for the 2D-only branch (pnp.hip): per frame f the matches grouped by object -- model
// points into d_X and keypoint indices f nq + q into d_qidx, both at [f nq k, ...) -- and per object its count and offset
// (d_hist, d_goff: F x n_objs). One launch, nothing comes to the host; d_err[f] != 0: the frame's inputs were refused.
int tod_cluster_frames_nocloud(todhip_ctx* ctx, uint32_t F, const float* d_kp_xy, uint32_t nq, const uint32_t* d_counts,
                               const todhip_dmatch* d_matches, const float* d_mxyz, uint32_t k, uint32_t n_objs, float* d_X,
                               uint32_t* d_qidx, uint32_t* d_hist, uint32_t* d_goff, uint32_t* d_err) {
  Engine E = {ctx, ctx->stream, nq, 0xFFFFFFFFu, 0xFFFFFFFFu, k, n_objs, nullptr, nullptr, {}};
  std::vector<ClusterArgs> v;
  const size_t per = (size_t)nq * k;
  for (uint32_t f = 0; f < F; ++f) {
    Slot s;
    s.ws = ws_of(ctx, f);
    int rc = E.reserve_common(s);
    if (rc == TODHIP_OK) rc = E.reserve_cluster(s);
    if (rc != TODHIP_OK) return rc;
    VerifyWs* ws = s.ws;
    ClusterArgs ca;
    ca.kp_xy = d_kp_xy + 2 * (size_t)f * nq; ca.cloud = nullptr; ca.depth = nullptr; ca.counts = d_counts + (size_t)f * nq;
    ca.matches = d_matches + f * per; ca.mxyz = d_mxyz + 3 * f * per; ca.nq = nq; ca.k = k; ca.H = ca.Wimg = 0xFFFFFFFFu; ca.n_objs = n_objs;
    ca.qidx_add = f * nq; ca.depth_is_u16 = 0; ca.fx = ca.fy = 1.f; ca.cx = ca.cy = 0.f;
    ca.kept = ws->c_kept.as<uint32_t>(); ca.offs = ws->c_offs.as<uint32_t>(); ca.obj_of = ws->c_obj.as<uint32_t>();
    ca.src = ws->c_src.as<uint32_t>(); ca.hist = d_hist + (size_t)f * n_objs; ca.goff = d_goff + (size_t)f * n_objs;
    ca.cnt = ws->c_cnt.as<uint32_t>(); ca.qpt = ws->c_qpt.as<float>();
    ca.train = d_X + 3 * f * per; ca.query = ws->query.as<float>(); ca.kpxy = ws->kpxy.as<float>(); ca.qidx = d_qidx + f * per;
    ca.m_hist = ws->m_hist.as<uint32_t>(); ca.m_ctl = d_err + 8 * (size_t)f;       // [0] error, [4] matches kept (device words here)
    v.push_back(ca);
  }
  launch_list(ctx->stream, cluster_frame_kernel, v, 256, 0, 0, [](const ClusterArgs&) { return dim3(1); });
  TOD_HIP(hipGetLastError());
  return TODHIP_OK;
}

void tod_verify_ws_free(todhip_ctx* ctx) {
  if (!ctx->verify_ws) return;
  VerifyPool* p = reinterpret_cast<VerifyPool*>(ctx->verify_ws);
  for (VerifyWs* ws : p->slots) { ws->release(); delete ws; }
  for (StreamCache* c : p->streams) { c->dev.release(); delete c; }
  for (hipEvent_t e : p->side_ev) (void)hipEventDestroy(e);
  for (size_t i = 0; i < VerifyPool::kMaxLanes; ++i) { p->args_stage[i].release(); p->args_dev[i].release(); }
  p->kceil.release();
  delete p;
  ctx->verify_ws = nullptr;
}

extern "C" {

void todhip_rng_seed(todhip_rng* r, uint32_t seed) {
  if (!r) return;
  if (seed == 0) seed = 1;
  int32_t word = (int32_t)seed;
  r->s[0] = seed;
  for (int i = 1; i < 31; ++i) {                          // srandom_r: Lehmer LCG, Schrage's method
    const long hi = word / 127773, lo = word % 127773;
    word = (int32_t)(16807 * lo - 2836 * hi);
    if (word < 0) word += 2147483647;
    r->s[i] = (uint32_t)word;
  }
  r->f = 3; r->b = 0;
  for (int i = 0; i < 310; ++i) {
    r->s[r->f] += r->s[r->b];
    r->f = (r->f + 1) % 31; r->b = (r->b + 1) % 31;
  }
  r->draws = 0;
}

static int verify_prologue(todhip_ctx* ctx, const todhip_verify_params* prm) {
  if (!ctx || !prm) return TODHIP_EINVAL;
  ctx->counters.last_objects_verified = ctx->counters.last_rounds = ctx->counters.last_hypotheses = 0;
  ctx->counters.last_gate_calls = ctx->counters.last_poses = 0;
  ctx->counters.last_sprint_launches = ctx->counters.last_sprint_rounds = ctx->counters.last_verify_ticks = 0;
  ctx->traces.clear();
  TOD_HIP(hipSetDevice(ctx->device));
  return set_big_lds_once(ctx);
}

// poses and inlier lists of the slots, concatenated in slot order; pose_ptr (optional) = CSR offsets per slot
static int collect(todhip_ctx* ctx, std::vector<Slot>& slots, todhip_pose* poses, uint32_t pose_cap, uint32_t* n_poses,
                   uint32_t* pose_ptr, uint32_t* inlier_kp, uint32_t kp_cap, uint32_t* n_inlier_kp) {
  uint32_t np = 0, nk = 0;
  for (size_t f = 0; f < slots.size(); ++f) {
    Slot& s = slots[f];
    if (pose_ptr) pose_ptr[f] = np;
    for (const todhip_round_trace& t : s.traces) ctx->traces.push_back(t);
    if (np + s.poses.size() > pose_cap || nk + s.inliers.size() > kp_cap) return TODHIP_ECAPACITY;
    for (todhip_pose p : s.poses) {
      p.inlier_begin += nk; p.inlier_end += nk;
      poses[np++] = p;
    }
    for (uint32_t v : s.inliers) inlier_kp[nk++] = v;
  }
  if (pose_ptr) pose_ptr[slots.size()] = np;
  *n_poses = np; *n_inlier_kp = nk;
  return TODHIP_OK;
}

int todhip_verify(todhip_ctx* ctx, const float* kp_xy, uint32_t nq, const float* cloud, uint32_t H, uint32_t Wimg,
                  const uint32_t* row_ptr, const todhip_dmatch* matches, const float* mxyz, const float* spans,
                  uint32_t n_objs, const todhip_verify_params* prm, todhip_rng* rng, todhip_pose* poses,
                  uint32_t* n_poses, uint32_t* inlier_kp, uint32_t* n_inlier_kp) {
  if (!ctx || !prm || !rng || !n_poses || !n_inlier_kp || (*n_poses && !poses) || (*n_inlier_kp && !inlier_kp))
    return TODHIP_EINVAL;
  int rc = verify_prologue(ctx, prm);
  if (rc != TODHIP_OK) return rc;
  const uint32_t pose_cap = *n_poses, kp_cap = *n_inlier_kp;
  *n_poses = 0; *n_inlier_kp = 0;
  if (!cloud || H == 0 || Wimg == 0) return TODHIP_OK;   // 2D-only input is an empty TODO (GuessGenerator.cpp:147-152)
  if (nq && (!kp_xy || !row_ptr || !spans)) return TODHIP_EINVAL;
  VerifyWs* ws = ws_of(ctx);

  // ---- ClusterPerObject (adjacency_ransac.cpp:176-205) on the host buffers the caller handed over: a gather of
  // one cloud point per keypoint and a bucket by imgIdx; per object, matches stay in (query asc, rank asc) order
  struct HostCluster { std::vector<float> train, query, kpxy; std::vector<uint32_t> qidx; };
  std::map<uint32_t, HostCluster> objects;
  for (uint32_t qi = 0; qi < nq; ++qi) {
    const int row = (int)kp_xy[2 * qi + 1], col = (int)kp_xy[2 * qi];   // float -> int truncation (:185)
    if (row < 0 || col < 0 || (uint32_t)row >= H || (uint32_t)col >= Wimg) return TODHIP_ERANGE;
    const float* qp = cloud + 3 * ((size_t)row * Wimg + col);
    if (std::isnan(qp[0])) continue;                                     // only .x is tested (:189)
    for (uint32_t m = row_ptr[qi]; m < row_ptr[qi + 1]; ++m) {
      if (matches[m].imgIdx < 0 || (uint32_t)matches[m].imgIdx >= n_objs) return TODHIP_ERANGE;
      HostCluster& c = objects[(uint32_t)matches[m].imgIdx];
      for (int k = 0; k < 3; ++k) { c.train.push_back(mxyz[3 * (size_t)m + k]); c.query.push_back(qp[k]); }
      c.kpxy.push_back(kp_xy[2 * qi]); c.kpxy.push_back(kp_xy[2 * qi + 1]);
      c.qidx.push_back(qi);
    }
  }
  hipStream_t st = ctx->stream;
  std::vector<Slot> slots(1);
  Slot& s = slots[0];
  s.ws = ws; s.rng = rng;
  uint32_t total = 0, max_n = 0;
  for (auto& kv : objects) {
    s.objs.push_back({kv.first, total, (uint32_t)kv.second.qidx.size()});
    total += (uint32_t)kv.second.qidx.size();
    max_n = std::max(max_n, (uint32_t)kv.second.qidx.size());
  }
  if (total == 0) return TODHIP_OK;
  TOD_HIP(ws->train.reserve((size_t)total * 12)); TOD_HIP(ws->query.reserve((size_t)total * 12));
  TOD_HIP(ws->qidx.reserve((size_t)total * 4)); TOD_HIP(ws->kpxy.reserve((size_t)total * 8));
  // the objects' slices are contiguous on the device (object order): four uploads for the frame, not four per object (a frame of
  // self-similar texture spreads its matches over a couple of hundred objects)
  std::vector<float> h_train((size_t)total * 3), h_query((size_t)total * 3), h_kpxy((size_t)total * 2);
  std::vector<uint32_t> h_qidx(total);
  size_t i = 0;
  for (auto& kv : objects) {
    HostCluster& c = kv.second;
    const ObjSpan& o = s.objs[i++];
    if (o.n == 0) continue;
    std::memcpy(h_train.data() + 3 * (size_t)o.offset, c.train.data(), (size_t)o.n * 12);
    std::memcpy(h_query.data() + 3 * (size_t)o.offset, c.query.data(), (size_t)o.n * 12);
    std::memcpy(h_qidx.data() + o.offset, c.qidx.data(), (size_t)o.n * 4);
    std::memcpy(h_kpxy.data() + 2 * (size_t)o.offset, c.kpxy.data(), (size_t)o.n * 8);
  }
  TOD_HIP(hipMemcpyAsync(ws->train.p, h_train.data(), (size_t)total * 12, hipMemcpyHostToDevice, st));
  TOD_HIP(hipMemcpyAsync(ws->query.p, h_query.data(), (size_t)total * 12, hipMemcpyHostToDevice, st));
  TOD_HIP(hipMemcpyAsync(ws->qidx.p, h_qidx.data(), (size_t)total * 4, hipMemcpyHostToDevice, st));
  TOD_HIP(hipMemcpyAsync(ws->kpxy.p, h_kpxy.data(), (size_t)total * 8, hipMemcpyHostToDevice, st));
  TOD_HIP(hipStreamSynchronize(st));   // the host vectors die with this scope
  Engine E = {ctx, st, nq, H, Wimg, 0u, n_objs, spans, prm, {}};
  rc = E.reserve_common(s);
  if (rc != TODHIP_OK) return rc;
  if (!E.reserve_objects(s, max_n)) return s.rc;
  s.ph = PH_PREPALL;
  std::vector<Slot*> live = {&s};
  rc = E.run(live);
  const int rc2 = collect(ctx, slots, poses, pose_cap, n_poses, nullptr, inlier_kp, kp_cap, n_inlier_kp);
  return rc != TODHIP_OK ? rc : rc2;
}

// Device-resident form for a batch of F frames (F = 1 for the single-frame entry points): keypoints, cloud or
// depth, and the matcher's fixed-stride outputs (counts[nq], matches[nq*k], matches_xyz[nq*k*3], see
// todhip_match_device) of frame f start at f times their per-frame size; only poses come back to the host.
static int verify_batch_impl(todhip_ctx* ctx, uint32_t F, const void* d_kp_xy, uint32_t nq, const void* d_cloud,
                             const DepthInput* dep, uint32_t H, uint32_t Wimg, const void* d_counts, const void* d_matches,
                             const void* d_mxyz, uint32_t k, const float* spans, uint32_t n_objs,
                             const todhip_verify_params* prm, todhip_rng* rng, todhip_pose* poses, uint32_t* n_poses,
                             uint32_t* pose_ptr, uint32_t* inlier_kp, uint32_t* n_inlier_kp) {
  if (!ctx || !prm || !rng || !n_poses || !n_inlier_kp || (*n_poses && !poses) || (*n_inlier_kp && !inlier_kp) || F == 0)
    return TODHIP_EINVAL;
  int rc = verify_prologue(ctx, prm);
  if (rc != TODHIP_OK) return rc;
  const uint32_t pose_cap = *n_poses, kp_cap = *n_inlier_kp;
  *n_poses = 0; *n_inlier_kp = 0;
  if (pose_ptr) for (uint32_t f = 0; f <= F; ++f) pose_ptr[f] = 0;
  if ((!d_cloud && !dep) || H == 0 || Wimg == 0) return TODHIP_OK;
  if (nq == 0 || n_objs == 0) return TODHIP_OK;
  if (!d_kp_xy || !d_counts || !d_matches || !d_mxyz || !spans || k == 0) return TODHIP_EINVAL;
  Engine E = {ctx, ctx->stream, nq, H, Wimg, k, n_objs, spans, prm, {}};
  std::vector<Slot> slots(F);
  std::vector<Slot*> live;
  const size_t px = (size_t)H * Wimg;
  for (uint32_t f = 0; f < F; ++f) {
    Slot& s = slots[f];
    s.ws = ws_of(ctx, f);
    s.rng = rng + f;
    s.d_kp_xy = reinterpret_cast<const float*>(d_kp_xy) + (size_t)f * nq * 2;
    s.d_counts = reinterpret_cast<const uint32_t*>(d_counts) + (size_t)f * nq;
    s.d_matches = reinterpret_cast<const todhip_dmatch*>(d_matches) + (size_t)f * nq * k;
    s.d_mxyz = reinterpret_cast<const float*>(d_mxyz) + (size_t)f * nq * k * 3;
    if (dep) {
      s.use_depth = true;
      s.dep = *dep;
      s.dep.d_depth = reinterpret_cast<const unsigned char*>(dep->d_depth) + (size_t)f * px * (dep->is_u16 ? 2 : 4);
    } else {
      s.d_cloud = reinterpret_cast<const float*>(d_cloud) + (size_t)f * px * 3;
    }
    rc = E.reserve_common(s);
    if (rc != TODHIP_OK) return rc;
    rc = E.reserve_cluster(s);
    if (rc != TODHIP_OK) return rc;
    s.ph = PH_CLUSTER;
    live.push_back(&s);
  }
  rc = E.run(live);
  const int rc2 = collect(ctx, slots, poses, pose_cap, n_poses, pose_ptr, inlier_kp, kp_cap, n_inlier_kp);
  return rc != TODHIP_OK ? rc : rc2;
}

int todhip_verify_device(todhip_ctx* ctx, const void* d_kp_xy, uint32_t nq, const void* d_cloud, uint32_t H,
                         uint32_t Wimg, const void* d_counts, const void* d_matches, const void* d_mxyz, uint32_t k,
                         const float* spans, uint32_t n_objs, const todhip_verify_params* prm, todhip_rng* rng,
                         todhip_pose* poses, uint32_t* n_poses, uint32_t* inlier_kp, uint32_t* n_inlier_kp) {
  return verify_batch_impl(ctx, 1, d_kp_xy, nq, d_cloud, nullptr, H, Wimg, d_counts, d_matches, d_mxyz, k, spans, n_objs, prm, rng,
                           poses, n_poses, nullptr, inlier_kp, n_inlier_kp);
}

int todhip_verify_device_depth(todhip_ctx* ctx, const void* d_kp_xy, uint32_t nq, const void* d_depth, int depth_is_u16,
                               uint32_t H, uint32_t Wimg, const float* K9, const void* d_counts, const void* d_matches,
                               const void* d_mxyz, uint32_t k, const float* spans, uint32_t n_objs,
                               const todhip_verify_params* prm, todhip_rng* rng, todhip_pose* poses, uint32_t* n_poses,
                               uint32_t* inlier_kp, uint32_t* n_inlier_kp) {
  if (!d_depth || !K9) return TODHIP_EINVAL;
  DepthInput dep = {d_depth, depth_is_u16, K9[0], K9[4], K9[2], K9[5]};
  return verify_batch_impl(ctx, 1, d_kp_xy, nq, nullptr, &dep, H, Wimg, d_counts, d_matches, d_mxyz, k, spans, n_objs, prm, rng,
                           poses, n_poses, nullptr, inlier_kp, n_inlier_kp);
}

int todhip_verify_batch_device(todhip_ctx* ctx, uint32_t n_frames, const void* d_kp_xy, uint32_t nq, const void* d_cloud,
                               uint32_t H, uint32_t Wimg, const void* d_counts, const void* d_matches, const void* d_mxyz,
                               uint32_t k, const float* spans, uint32_t n_objs, const todhip_verify_params* prm,
                               todhip_rng* rng, todhip_pose* poses, uint32_t* n_poses, uint32_t* pose_ptr,
                               uint32_t* inlier_kp, uint32_t* n_inlier_kp) {
  if (!pose_ptr) return TODHIP_EINVAL;
  return verify_batch_impl(ctx, n_frames, d_kp_xy, nq, d_cloud, nullptr, H, Wimg, d_counts, d_matches, d_mxyz, k, spans, n_objs,
                           prm, rng, poses, n_poses, pose_ptr, inlier_kp, n_inlier_kp);
}

int todhip_verify_batch_device_depth(todhip_ctx* ctx, uint32_t n_frames, const void* d_kp_xy, uint32_t nq, const void* d_depth,
                                     int depth_is_u16, uint32_t H, uint32_t Wimg, const float* K9, const void* d_counts,
                                     const void* d_matches, const void* d_mxyz, uint32_t k, const float* spans, uint32_t n_objs,
                                     const todhip_verify_params* prm, todhip_rng* rng, todhip_pose* poses, uint32_t* n_poses,
                                     uint32_t* pose_ptr, uint32_t* inlier_kp, uint32_t* n_inlier_kp) {
  if (!d_depth || !K9 || !pose_ptr) return TODHIP_EINVAL;
  DepthInput dep = {d_depth, depth_is_u16, K9[0], K9[4], K9[2], K9[5]};
  return verify_batch_impl(ctx, n_frames, d_kp_xy, nq, nullptr, &dep, H, Wimg, d_counts, d_matches, d_mxyz, k, spans, n_objs, prm,
                           rng, poses, n_poses, pose_ptr, inlier_kp, n_inlier_kp);
}

int todhip_verify_trace(const todhip_ctx* ctx, todhip_round_trace* out, uint32_t* n) {
  if (!ctx || !n || (*n && !out)) return TODHIP_EINVAL;
  const uint32_t cap = *n;
  *n = (uint32_t)ctx->traces.size();
  for (uint32_t i = 0; i < cap && i < ctx->traces.size(); ++i) out[i] = ctx->traces[i];
  return ctx->traces.size() > cap ? TODHIP_ECAPACITY : TODHIP_OK;
}

int todhip_test_adjacency(todhip_ctx* ctx, const float* train, const float* query, const float* kpxy, uint32_t n,
                          float span, float err, uint64_t* phys, uint64_t* samp) {
  if (!ctx || !train || !query || !kpxy || !phys || !samp || n == 0 || n > (uint32_t)kMaxWords * 64u) return TODHIP_EINVAL;
  TOD_HIP(hipSetDevice(ctx->device));
  VerifyWs* ws = ws_of(ctx);
  const uint32_t W = (n + 63u) / 64u;
  hipStream_t st = ctx->stream;
  TOD_HIP(ws->train.reserve((size_t)n * 12)); TOD_HIP(ws->query.reserve((size_t)n * 12));
  TOD_HIP(ws->kpxy.reserve((size_t)n * 8));
  TOD_HIP(ws->phys.reserve((size_t)n * W * 8)); TOD_HIP(ws->samp.reserve((size_t)n * W * 8));
  TOD_HIP(ws->bits.reserve((size_t)8 * W * 8));
  TOD_HIP(hipMemcpyAsync(ws->train.p, train, (size_t)n * 12, hipMemcpyHostToDevice, st));
  TOD_HIP(hipMemcpyAsync(ws->query.p, query, (size_t)n * 12, hipMemcpyHostToDevice, st));
  TOD_HIP(hipMemcpyAsync(ws->kpxy.p, kpxy, (size_t)n * 8, hipMemcpyHostToDevice, st));
  ObjJob job;
  std::memset(&job, 0, sizeof(job));
  job.n = n; job.W = W;
  job.train = ws->train.as<float>(); job.query = ws->query.as<float>(); job.kpxy = ws->kpxy.as<float>();
  job.phys = ws->phys.as<u64>(); job.samp = ws->samp.as<u64>();
  launch_list<kManySlots>(st, adjacency_kernel<Slots<AdjArgs, kManySlots>>, std::vector<AdjArgs>{{job, span, err}}, 256, 0, 2,
              [](const AdjArgs& a) { return dim3(a.job.n, (a.job.W + 3u) / 4u); });
  TOD_HIP(hipGetLastError());
  TOD_HIP(hipMemcpyAsync(phys, ws->phys.p, (size_t)n * W * 8, hipMemcpyDeviceToHost, st));
  TOD_HIP(hipMemcpyAsync(samp, ws->samp.p, (size_t)n * W * 8, hipMemcpyDeviceToHost, st));
  TOD_HIP(hipStreamSynchronize(st));
  return TODHIP_OK;
}

// Test hook: FillAdjacency + selectWithinDistance (sac_model_registration_graph.h:171-269) for given sample
// triples (samples_ order). counts[t] = consensus size (0 = rejected by the gate). With stop_level 1 the clique
// search is skipped and counts[t] = -|F| for hypotheses that reach it. dbg (optional): dbg_stride words per triple.
int todhip_test_consensus(todhip_ctx* ctx, const float* train, const float* query, const float* kpxy, uint32_t n,
                          float span, float err, const uint32_t* triples, uint32_t n_triples, uint32_t stop_level,
                          int32_t* counts, uint32_t* dbg, uint32_t dbg_stride) {
  if (!ctx || !train || !query || !kpxy || !triples || !counts || n < 3 || n > (uint32_t)kMaxWords * 64u || n_triples == 0)
    return TODHIP_EINVAL;
  TOD_HIP(hipSetDevice(ctx->device));
  int rc = set_big_lds_once(ctx);
  if (rc != TODHIP_OK) return rc;
  VerifyWs* ws = ws_of(ctx);
  const uint32_t W = (n + 63u) / 64u;
  hipStream_t st = ctx->stream;
  TOD_HIP(ws->train.reserve((size_t)n * 12)); TOD_HIP(ws->query.reserve((size_t)n * 12));
  TOD_HIP(ws->qidx.reserve((size_t)n * 4)); TOD_HIP(ws->kpxy.reserve((size_t)n * 8));
  TOD_HIP(ws->phys.reserve((size_t)n * W * 8)); TOD_HIP(ws->samp.reserve((size_t)n * W * 8));
  TOD_HIP(ws->bits.reserve((size_t)8 * W * 8)); TOD_HIP(ws->sampdeg.reserve((size_t)n * 4));
  TOD_HIP(ws->small.reserve(256 * sizeof(uint32_t))); TOD_HIP(ws->h_small.reserve(256 * sizeof(uint32_t)));
  TOD_HIP(ws->iter_samples.reserve((size_t)n_triples * 12)); TOD_HIP(ws->counts.reserve((size_t)n_triples * 4));
  TOD_HIP(ws->gate_m.reserve((size_t)n_triples * 4)); TOD_HIP(ws->deferred.reserve((size_t)n_triples * 4));
  TOD_HIP(ws->stacks.reserve((size_t)kMaxEvalWaves * kStackCap * sizeof(uint16_t)));
  if (dbg) TOD_HIP(ws->table.reserve((size_t)n_triples * dbg_stride * 4));
  TOD_HIP(hipMemcpyAsync(ws->train.p, train, (size_t)n * 12, hipMemcpyHostToDevice, st));
  TOD_HIP(hipMemcpyAsync(ws->query.p, query, (size_t)n * 12, hipMemcpyHostToDevice, st));
  TOD_HIP(hipMemcpyAsync(ws->kpxy.p, kpxy, (size_t)n * 8, hipMemcpyHostToDevice, st));
  TOD_HIP(hipMemcpyAsync(ws->iter_samples.p, triples, (size_t)n_triples * 12, hipMemcpyHostToDevice, st));
  TOD_HIP(hipMemsetAsync(ws->qidx.p, 0, (size_t)n * 4, st));
  if (dbg) TOD_HIP(hipMemsetAsync(ws->table.p, 0, (size_t)n_triples * dbg_stride * 4, st));
  ObjJob job;
  job.n = n; job.W = W;
  job.train = ws->train.as<float>(); job.query = ws->query.as<float>(); job.qidx = ws->qidx.as<uint32_t>();
  job.kpxy = ws->kpxy.as<float>(); job.phys = ws->phys.as<u64>(); job.samp = ws->samp.as<u64>();
  u64* bits = ws->bits.as<u64>();
  job.finite = bits; job.valid = bits + W; job.deg7 = bits + 2 * W; job.sampdeg = ws->sampdeg.as<uint32_t>();
  uint32_t* d_small = ws->small.as<uint32_t>();
  uint32_t* h_small = ws->h_small.as<uint32_t>();
  TOD_HIP(hipMemsetAsync(d_small, 0, 64 * sizeof(uint32_t), st));
  launch_list<kManySlots>(st, finite_kernel<Slots<JobArgs, kManySlots>>, std::vector<JobArgs>{{job}}, 256, 0, 1, [](const JobArgs& a) { return dim3((a.job.n + 255u) / 256u); });
  launch_list<kManySlots>(st, adjacency_kernel<Slots<AdjArgs, kManySlots>>, std::vector<AdjArgs>{{job, span, err}}, 256, 0, 2,
              [](const AdjArgs& a) { return dim3(a.job.n, (a.job.W + 3u) / 4u); });
  launch_list<kManySlots>(st, round_prep_kernel<Slots<PrepArgs, kManySlots>>, std::vector<PrepArgs>{{job, d_small + 5, nullptr, 0u}}, 256, 0, 1,
              [](const PrepArgs& a) { return dim3((a.job.n + 255u) / 256u); });
  EvalArgs A;
  A.job = job; A.iter_samples = ws->iter_samples.as<uint32_t>(); A.it_begin = 0; A.it_end = n_triples;
  A.counts = ws->counts.as<int32_t>(); A.gate_m = ws->gate_m.as<uint32_t>(); A.work = d_small + 8;
  A.status = d_small + 12; A.deferred = ws->deferred.as<uint32_t>(); A.stacks = ws->stacks.as<uint16_t>();
  A.stack_cap = kStackCap; A.lds_bytes = kEvalLdsSmall; A.from_deferred = 0; A.n_deferred = 0;
  A.adjc_scratch = nullptr; A.n_items_dev = nullptr;
  A.dbg = dbg ? ws->table.as<uint32_t>() : nullptr; A.dbg_stride = dbg_stride; A.stop_level = stop_level;
  if (n_triples > kMaxEvalWaves) return TODHIP_EINVAL;
  if (W <= 8u) launch_list(st, eval_kernel<false>, std::vector<EvalArgs>{A}, 64, kEvalLdsSmall, 1, [](const EvalArgs& a) { return dim3(a.it_end - a.it_begin); });
  else launch_list(st, eval_kernel<true>, std::vector<EvalArgs>{A}, 64, kEvalLdsSmall, 1, [](const EvalArgs& a) { return dim3(a.it_end - a.it_begin); });
  TOD_HIP(hipGetLastError());
  TOD_HIP(hipMemcpyAsync(h_small + 12, d_small + 12, 8 * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
  TOD_HIP(hipStreamSynchronize(st));
  const uint32_t status = h_small[12], n_def = h_small[14];
  if (status == 0 && n_def > 0) {
    A.lds_bytes = kEvalLdsBig; A.from_deferred = 1; A.n_deferred = n_def;
    TOD_HIP(ws->adjc_scratch.reserve((size_t)n_def * kAdjcScratchWords * sizeof(u64)));
    A.adjc_scratch = ws->adjc_scratch.as<u64>();
    TOD_HIP(hipMemsetAsync(d_small + 8, 0, sizeof(uint32_t), st));
    if (W <= 8u) launch_list(st, eval_kernel<false>, std::vector<EvalArgs>{A}, 64, kEvalLdsBig, 1, [](const EvalArgs& a) { return dim3(a.n_deferred); });
    else launch_list(st, eval_kernel<true>, std::vector<EvalArgs>{A}, 64, kEvalLdsBig, 1, [](const EvalArgs& a) { return dim3(a.n_deferred); });
    TOD_HIP(hipGetLastError());
    TOD_HIP(hipMemcpyAsync(h_small + 12, d_small + 12, 8 * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    TOD_HIP(hipStreamSynchronize(st));
  }
  TOD_HIP(hipMemcpyAsync(counts, ws->counts.p, (size_t)n_triples * 4, hipMemcpyDeviceToHost, st));
  if (dbg) TOD_HIP(hipMemcpyAsync(dbg, ws->table.p, (size_t)n_triples * dbg_stride * 4, hipMemcpyDeviceToHost, st));
  TOD_HIP(hipStreamSynchronize(st));
  if (h_small[12] != 0) {
    TOD_DBG("consensus status %u: g=%u value=%u m=%u it=%u", h_small[12], h_small[16], h_small[17], h_small[18], h_small[19]);
    return TODHIP_ESCRATCH;
  }
  return TODHIP_OK;
}

// Test hook: the clique search on an explicit graph (edges as pairs), FindClique(minimal_size).
// out3 = {clique size, error flag, steps}. Mirrors the reference's gtest shape (test/test_maximum_clique.cpp).
static int test_clique_impl(todhip_ctx* ctx, uint32_t m, const uint32_t* edges, uint32_t n_edges, uint32_t minimal_size,
                            uint32_t* out3, bool gate) {
  if (!ctx || !out3 || m == 0 || m > 1024 || (n_edges && !edges)) return TODHIP_EINVAL;
  TOD_HIP(hipSetDevice(ctx->device));
  int rc = set_big_lds_once(ctx);
  if (rc != TODHIP_OK) return rc;
  VerifyWs* ws = ws_of(ctx);
  const uint32_t MW = (m + 63u) / 64u;
  std::vector<u64> adj((size_t)m * MW, 0ull);
  for (uint32_t e = 0; e < n_edges; ++e) {
    const uint32_t a = edges[2 * e], b = edges[2 * e + 1];
    if (a >= m || b >= m || a == b) return TODHIP_EINVAL;
    adj[(size_t)a * MW + (b >> 6)] |= 1ull << (b & 63u);
    adj[(size_t)b * MW + (a >> 6)] |= 1ull << (a & 63u);
  }
  TOD_HIP(ws->clique_adj.reserve(adj.size() * 8 + 64));
  TOD_HIP(ws->stacks.reserve((size_t)kStackCap * sizeof(uint16_t)));
  TOD_HIP(ws->small.reserve(256 * sizeof(uint32_t)));
  TOD_HIP(hipMemcpyAsync(ws->clique_adj.p, adj.data(), adj.size() * 8, hipMemcpyHostToDevice, ctx->stream));
  if (gate_lds_bytes(m) > kEvalLdsBig) return TODHIP_ESCRATCH;
  const uint32_t lds = gate_lds_bytes(m) <= kEvalLdsSmall ? kEvalLdsSmall : kEvalLdsBig;   // the two LDS tiers of eval_kernel
  if (gate)
    hipLaunchKernelGGL(clique_test_kernel<true>, dim3(1), dim3(64), lds, ctx->stream, ws->clique_adj.as<u64>(), m, minimal_size,
                       ws->stacks.as<uint16_t>(), kStackCap, lds, ws->small.as<uint32_t>());
  else
    hipLaunchKernelGGL(clique_test_kernel<false>, dim3(1), dim3(64), lds, ctx->stream, ws->clique_adj.as<u64>(), m, minimal_size,
                       ws->stacks.as<uint16_t>(), kStackCap, lds, ws->small.as<uint32_t>());
  TOD_HIP(hipGetLastError());
  TOD_HIP(hipMemcpyAsync(out3, ws->small.p, 3 * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
  TOD_HIP(hipStreamSynchronize(ctx->stream));
  return TODHIP_OK;
}

int todhip_test_clique(todhip_ctx* ctx, uint32_t m, const uint32_t* edges, uint32_t n_edges, uint32_t minimal_size,
                       uint32_t* out3) {
  return test_clique_impl(ctx, m, edges, n_edges, minimal_size, out3, false);
}

// The same graph through the form of the search the verifier's gate runs (clique_search<., kGate = true>): it stops as soon as
// "is the clique FindClique(minimal_size) returns larger than minimal_size" is decided, so out3[0] is that clique's size only
// when it is <= minimal_size, and a lower bound > minimal_size otherwise; out3[2] counts the steps actually walked.
int todhip_test_clique_gate(todhip_ctx* ctx, uint32_t m, const uint32_t* edges, uint32_t n_edges, uint32_t minimal_size,
                            uint32_t* out3) {
  return test_clique_impl(ctx, m, edges, n_edges, minimal_size, out3, true);
}

}  // extern "C"
