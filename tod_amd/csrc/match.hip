// Stage B on gfx950: exact brute-force Hamming k-NN of the frame descriptors against the
// device-resident object database, radius truncation, (imgIdx, trainIdx) resolution and the
// match -> 3D gather.  Replaces DescriptorMatcher::process (reference
// src/detection/DescriptorMatcher.cpp:195-252; the k-NN call at :211 is FLANN-LSH there,
// an exact search with the order (distance asc, global row asc) here -- decision D1).
//
// K4  hamming_topk_tiles   one lane = one query descriptor held in 8 VGPRs; the DB rows of the
//                          block's tile are wave-uniform, so they are fetched with scalar loads
//                          (s_load_dwordx8 through the scalar cache) and every xor takes its DB
//                          word straight from an SGPR: 8 v_xor + 8 accumulating v_bcnt per pair,
//                          no LDS traffic and no cross-lane work in the inner loop. Each lane keeps
//                          its k best (distance, row) keys in registers; four rows share one
//                          min3/min/compare so the top-k test costs 0.75 VALU op per pair.
//                          Partial-distance elimination: the distance over the first 128 bits is a lower
//                          bound of the full one, so when it already reaches the running limit (k-th best so
//                          far, a bound published by another tile, or radius + 1 -- rows beyond the radius
//                          never survive DescriptorMatcher.cpp:212-220) for all 4 rows x 64 queries of a
//                          group, the second half of those rows is never touched. Exact for any data; how
//                          often it fires depends on the data (always, but for ~1e-4 of the groups, on
//                          descriptors with independent bits and radius 35).
// K4m merge_tiles_kernel   per query: merge the per-tile lists into k global keys.
// K4f finalize_kernel      per query: merge shard lists, radius cut, object lookup, 3D gather.
#include <algorithm>
#include <cstdlib>

#include <cstdio>
#include <cstring>

#include "ctx.h"

namespace {

constexpr int kWords = 8;          // 256-bit descriptors (ORB / rBRIEF), 32 bytes per row
constexpr int kGroupRows = 4;      // DB rows per SGPR group (two s_load_dwordx16)
constexpr int kLocalBits = 22;     // tile-local row index bits in a partial key (tile <= 4M rows)
constexpr uint32_t kLocalMask = (1u << kLocalBits) - 1u;
constexpr int kBlock = 256;
constexpr int kWavesPerBlock = kBlock / 64;
constexpr int kWavesPerCU = 24;    // waves a CU holds at once: 6 blocks of 4 waves (<= 112 SGPRs); the grid is ~3x that
constexpr uint32_t kSharePeriod = 128;   // groups between two exchanges of the per-query distance bound
constexpr int kMergeGroups = 16;   // stage-1 merge fan-in

template <int K>
__device__ __forceinline__ void topk_insert(uint32_t (&best)[K], uint32_t key) {
  // branch-free sorted insertion: key falls through the list, each slot keeps the smaller one
#pragma unroll
  for (int j = 0; j < K; ++j) {
    uint32_t lo = min(best[j], key);
    key = max(best[j], key);
    best[j] = lo;
  }
}

// popcount with accumulate: v_bcnt_u32_b32 D = countbits(S0) + S1. Written as asm because the compiler
// otherwise re-associates the chain into bcnt(x, 0) + v_add3 trees (3 extra VALU ops per row).
__device__ __forceinline__ uint32_t bcnt_acc(uint32_t x, uint32_t acc) {
  uint32_t d;
  asm("v_bcnt_u32_b32 %0, %1, %2" : "=v"(d) : "v"(x), "v"(acc));
  return d;
}

typedef uint32_t u32x16 __attribute__((ext_vector_type(16)));

// One SGPR group = kGroupRows (4) DB rows = two s_load_dwordx16. The loads are issued and waited for by
// hand (asm): hipcc otherwise sinks a prefetch below its consumer. Scalar loads return out of order, so the
// only usable wait is lgkmcnt(0); the ping-pong below always has exactly one group in flight when it waits.
struct RowGroup { u32x16 lo, hi; };

__device__ __forceinline__ void issue_rows(RowGroup& g, const uint32_t* p) {
  asm volatile("s_load_dwordx16 %0, %2, 0x0\n\ts_load_dwordx16 %1, %2, 0x40"
               : "=&s"(g.lo), "=&s"(g.hi) : "s"(p) : "memory");
  __builtin_amdgcn_sched_barrier(0);   // keep the prefetch ahead of the compute it overlaps with
}
__device__ __forceinline__ void wait_rows(RowGroup& g) {
  asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(g.lo), "+s"(g.hi));
}

// bits [32 W0, 32 W0 + 128) of row HALF of a 2-row SGPR block, accumulated onto acc
template <int HALF, int W0>
__device__ __forceinline__ uint32_t hamming128(const uint32_t (&q)[kWords], const u32x16& rows, uint32_t acc) {
#pragma unroll
  for (int w = W0; w < W0 + 4; ++w) acc = bcnt_acc(q[w] ^ rows[HALF * kWords + w], acc);
  return acc;
}

__device__ __forceinline__ uint32_t hamming256_mem(const uint32_t (&q)[kWords], const uint32_t* row) {
  uint32_t d = 0;
#pragma unroll
  for (int w = 0; w < kWords; ++w) d = bcnt_acc(q[w] ^ row[w], d);
  return d;
}

// `limit` = min(own k-th best distance, 1 + the smallest k-th best distance any tile has published for this
// query): a row at or above it cannot be among the k nearest of the whole DB (ties with a foreign bound are kept
// because a smaller row index could still win them), so skipping it keeps the merged result exact.
// MODE picks the elimination schedule by how tight the initial bound (radius + 1) is; every schedule is exact.
//   2: test after 96 bits per row, then after 128, after 192, then the rest   (pays for cut <= 38: a row passes the first test
//      in some lane with probability ~0.3 on independent bits at cut 36, ~0.9 at cut 40)
//   1: test after 128 bits per 4 rows, then the rest                  (38 < cut <= 48)
//   3: test after 192 bits per 4 rows, then the rest                  (48 < cut <= 80, e.g. radius 55 of conf/detection.ros.ork:60)
//   0: full distances, one test per 4 rows                            (larger radii: no lower bound prunes anything)
template <int K, int MODE>
__device__ __forceinline__ void consume_group(const uint32_t (&qd)[kWords], const RowGroup& g, uint32_t r,
                                              uint32_t (&best)[K], uint32_t& worst_d, uint32_t& limit, uint32_t foreign) {
  if (MODE != 2) {
    // the four rows' accumulate chains are interleaved word by word: no instruction depends on its predecessor
    constexpr int kFirst = MODE == 1 ? 4 : (MODE == 3 ? 6 : kWords);   // words before the group test
    uint32_t d0 = 0u, d1 = 0u, d2 = 0u, d3 = 0u;
#pragma unroll
    for (int w = 0; w < kFirst; ++w) {
      const uint32_t x0 = qd[w] ^ g.lo[w], x1 = qd[w] ^ g.lo[kWords + w], x2 = qd[w] ^ g.hi[w], x3 = qd[w] ^ g.hi[kWords + w];
      d0 = bcnt_acc(x0, d0); d1 = bcnt_acc(x1, d1); d2 = bcnt_acc(x2, d2); d3 = bcnt_acc(x3, d3);
    }
    uint32_t dmin = min(min(d0, d1), min(d2, d3));
    if (kFirst < kWords) {
      if (__builtin_amdgcn_ballot_w64(dmin < limit) == 0ull) return;    // lower bounds already out: skip the rest
#pragma unroll
      for (int w = kFirst; w < kWords; ++w) {
        const uint32_t x0 = qd[w] ^ g.lo[w], x1 = qd[w] ^ g.lo[kWords + w], x2 = qd[w] ^ g.hi[w], x3 = qd[w] ^ g.hi[kWords + w];
        d0 = bcnt_acc(x0, d0); d1 = bcnt_acc(x1, d1); d2 = bcnt_acc(x2, d2); d3 = bcnt_acc(x3, d3);
      }
      dmin = min(min(d0, d1), min(d2, d3));
    }
    if (__builtin_amdgcn_ballot_w64(dmin < limit) != 0ull) {
      // rows are visited in ascending order, so a later row never displaces an equal distance:
      // "key < best[K-1]" is exactly "d < worst_d" and insertion order inside the group is free.
      topk_insert<K>(best, (d0 << kLocalBits) | r);
      topk_insert<K>(best, (d1 << kLocalBits) | (r + 1));
      topk_insert<K>(best, (d2 << kLocalBits) | (r + 2));
      topk_insert<K>(best, (d3 << kLocalBits) | (r + 3));
      worst_d = best[K - 1] >> kLocalBits;
      limit = min(worst_d, foreign);
    }
    return;
  }
  // Three-stage partial-distance elimination. Stage A: 96 bits of each of the four rows (chains interleaved word by
  // word: no instruction depends on its predecessor) and one ballot per row; a row whose lower bound reaches the limit
  // in all 64 queries is finished. Stage B, per surviving row (~29 % of the rows on independent bits at radius 35):
  // the 4th word, test again; stage C (rare on independent bits, common on correlated ones): words 5-6, test; stage D: the
  // last 64 bits, test, insert.
  uint32_t d0 = 0u, d1 = 0u, d2 = 0u, d3 = 0u;
#pragma unroll
  for (int w = 0; w < 3; ++w) {
    const uint32_t x0 = qd[w] ^ g.lo[w], x1 = qd[w] ^ g.lo[kWords + w], x2 = qd[w] ^ g.hi[w], x3 = qd[w] ^ g.hi[kWords + w];
    d0 = bcnt_acc(x0, d0); d1 = bcnt_acc(x1, d1); d2 = bcnt_acc(x2, d2); d3 = bcnt_acc(x3, d3);
  }
  const unsigned long long b0 = __builtin_amdgcn_ballot_w64(d0 < limit), b1 = __builtin_amdgcn_ballot_w64(d1 < limit),
                           b2 = __builtin_amdgcn_ballot_w64(d2 < limit), b3 = __builtin_amdgcn_ballot_w64(d3 < limit);
  if ((b0 | b1 | b2 | b3) == 0ull) return;
  // rows are visited in ascending order, so a later row never displaces an equal distance:
  // "key < best[K-1]" is exactly "d < worst_d"
#define TOD_ROW_STAGES(bal_, d_, rows_, half_, idx_)                                                               \
  if ((bal_) != 0ull) {                                                                                            \
    d_ = bcnt_acc(qd[3] ^ rows_[half_ * kWords + 3], d_);                                                         \
    if (__builtin_amdgcn_ballot_w64(d_ < limit) != 0ull) {                                                         \
      d_ = bcnt_acc(qd[4] ^ rows_[half_ * kWords + 4], d_);                                                       \
      d_ = bcnt_acc(qd[5] ^ rows_[half_ * kWords + 5], d_);                                                       \
      if (__builtin_amdgcn_ballot_w64(d_ < limit) != 0ull) {                                                       \
        d_ = bcnt_acc(qd[6] ^ rows_[half_ * kWords + 6], d_);                                                     \
        d_ = bcnt_acc(qd[7] ^ rows_[half_ * kWords + 7], d_);                                                     \
        if (__builtin_amdgcn_ballot_w64(d_ < limit) != 0ull) {                                                     \
          topk_insert<K>(best, (d_ << kLocalBits) | (r + idx_));                                                  \
          worst_d = best[K - 1] >> kLocalBits;                                                                    \
          limit = min(worst_d, foreign);                                                                          \
        }                                                                                                         \
      }                                                                                                           \
    }                                                                                                             \
  }
  TOD_ROW_STAGES(b0, d0, g.lo, 0, 0u)
  TOD_ROW_STAGES(b1, d1, g.lo, 1, 1u)
  TOD_ROW_STAGES(b2, d2, g.hi, 0, 2u)
  TOD_ROW_STAGES(b3, d3, g.hi, 1, 3u)
#undef TOD_ROW_STAGES
}

// One WAVE = one work item (DB tile, group of 64 queries). Work items are numbered tile-major so the waves
// of a block share a tile (scalar-cache / L2 locality); blocks b and b+8 share an XCD, and the decode below
// gives each XCD a contiguous run of tiles.
template <int K, int MODE>
__global__ __launch_bounds__(kBlock) void hamming_topk_tiles(const uint32_t* __restrict__ db,
                                                             const uint32_t* __restrict__ q, uint32_t n_rows,
                                                             uint32_t nq, uint32_t nq_pad, uint32_t rows_per_tile,
                                                             uint32_t n_tiles, uint32_t n_qw,
                                                             uint32_t blocks_per_xcd, uint32_t tiles_per_xcd, uint32_t cut,
                                                             uint32_t* __restrict__ part, uint32_t* bound,
                                                             uint8_t* __restrict__ stored) {
  const uint32_t xcd = blockIdx.x & 7u, slot = blockIdx.x >> 3;
  uint32_t tile, qw;
  if (tiles_per_xcd) {
    // every XCD owns whole DB tiles (its L2 then holds one contiguous slice of the DB, read by all its query waves)
    const uint32_t local = __builtin_amdgcn_readfirstlane(slot * kWavesPerBlock + (threadIdx.x >> 6));
    if (local >= tiles_per_xcd * n_qw) return;
    tile = xcd * tiles_per_xcd + local / n_qw; qw = local % n_qw;
  } else {
    const uint32_t vblock = xcd * blocks_per_xcd + slot;               // XCD-contiguous virtual block id
    const uint32_t item = __builtin_amdgcn_readfirstlane(vblock * kWavesPerBlock + (threadIdx.x >> 6));
    tile = item / n_qw; qw = item % n_qw;
  }
  if (tile >= n_tiles) return;
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t qi = qw * 64u + lane;
  const uint32_t qi_ld = qi < nq ? qi : (nq - 1);

  uint32_t qd[kWords];
  {
    const uint4* qp = reinterpret_cast<const uint4*>(q + (size_t)qi_ld * kWords);
    uint4 a = qp[0], b = qp[1];
    qd[0] = a.x; qd[1] = a.y; qd[2] = a.z; qd[3] = a.w;
    qd[4] = b.x; qd[5] = b.y; qd[6] = b.z; qd[7] = b.w;
  }
  uint32_t best[K];
#pragma unroll
  for (int j = 0; j < K; ++j) best[j] = 0xFFFFFFFFu;
  uint32_t worst_d = 0xFFFFFFFFu >> kLocalBits;

  const uint32_t row0 = tile * rows_per_tile;
  const uint32_t row_end = min(n_rows, row0 + rows_per_tile);
  const uint32_t n_local = row_end > row0 ? row_end - row0 : 0u;
  const uint32_t* __restrict__ base = db + (size_t)row0 * kWords;

  // ping-pong SGPR groups: the load of group g+1 is in flight while group g is consumed
  const uint32_t n_groups = n_local / kGroupRows;
  uint32_t r = 0;
  // foreign = min(1 + smallest published k-th best distance, cut); cut = radius + 1: a row at distance > radius is
  // dropped by the radius truncation whatever its rank, so the search may drop it as well
  uint32_t foreign = cut, limit = min(worst_d, foreign);
  uint32_t* my_bound = bound + (qi < nq ? qi : nq - 1);
  if (n_groups > 0) {
    constexpr uint32_t kStride = kGroupRows * kWords;
    RowGroup ga, gb;
    issue_rows(ga, base);
    wait_rows(ga);
    uint32_t g = 0, next_share = 16;                        // first exchange early: the tile's own list is full by then
    for (; g + 2 <= n_groups; g += 2) {
      issue_rows(gb, base + (size_t)(g + 1) * kStride);
      consume_group<K, MODE>(qd, ga, r, best, worst_d, limit, foreign);
      wait_rows(gb);
      const uint32_t gn = (g + 2 < n_groups) ? g + 2 : g;      // the last pair re-reads an in-bounds group
      issue_rows(ga, base + (size_t)gn * kStride);
      consume_group<K, MODE>(qd, gb, r + kGroupRows, best, worst_d, limit, foreign);
      wait_rows(ga);
      r += 2 * kGroupRows;
      if (g >= next_share) {                                // wave-uniform
        next_share += kSharePeriod;
        // publish this tile's bound (only once its list is full), pick up the smallest bound published so far
        uint32_t seen = __hip_atomic_load(my_bound, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (worst_d < seen) { atomicMin(my_bound, worst_d); seen = worst_d; }   // publish only a real improvement
        foreign = seen == 0xFFFFFFFFu ? cut : min(cut, seen + 1u);
        limit = min(worst_d, foreign);
      }
    }
    if (g < n_groups) {                                         // odd group count: ga holds group g
      consume_group<K, MODE>(qd, ga, r, best, worst_d, limit, foreign);
      r += kGroupRows;
    }
  }
  for (; r < n_local; ++r) {
    uint32_t d = hamming256_mem(qd, base + (size_t)r * kWords);
    if (d < cut) topk_insert<K>(best, (d << kLocalBits) | r);
  }
  // A (tile, 64 queries) pair that found nothing below the limit -- the rule once a radius is set: 96 % of them on the
  // benchmark's data -- stores nothing; the merge skips it by its flag byte (0xFF from the launch's memset = nothing stored)
  if (__builtin_amdgcn_ballot_w64(qi < nq && best[0] != 0xFFFFFFFFu) != 0ull) {
    if (qi < nq) {
#pragma unroll
      for (int j = 0; j < K; ++j) part[((size_t)tile * K + j) * nq_pad + qi] = best[j];
    }
    if (lane == 0) stored[(size_t)tile * n_qw + qw] = 0;
  }
}

// ---------------------------------------------------------------------------------------------------------
// K4x  hamming_topk_mfma   the same exact search on the matrix cores. With every descriptor bit b written as the
//                          MX-fp4 (E2M1) value 1 - 2b, the dot product of two descriptors is 256 - 2 * hamming: products
//                          are +-1, the f32 accumulator holds integers <= 256, so the result is EXACT.
//                          v_mfma_f32_32x32x64_f8f6f4 (fp4 x fp4, unit scales) takes 64 bit positions of 32 DB rows x 32
//                          queries per issue: 4 MFMAs = 1024 complete distances in 128 matrix-pipe cycles (8 pairs per
//                          clock and SIMD; the VALU form above peaks at 1, or ~2 when its elimination fires) and the
//                          rate does not depend on the data.
//                          One WAVE = (DB tile, 32 QT queries). The query fragments stay in registers (16 VGPRs per
//                          32 queries); every lane loads 16 packed bytes of one DB row per 32-row step (a wave load = 32
//                          rows = 1 KB contiguous, served by L2: all query waves of a tile read the same lines), expands
//                          them to fp4 with 7 VALU ops per 32 bits, no LDS, no barrier. A and B use the same
//                          (lane, register, nibble) -> bit assignment, so the sum runs over matching bit positions whatever
//                          the hardware's internal k order is.
//                          Accumulator layout (dtype independent): lane = query column (l & 31), 16 registers = 16 DB
//                          rows (i & 3) + 8 (i >> 2) + 4 (l >> 5). A lane keeps its k best keys in registers exactly as
//                          K4 does; the test per 32 x 32 block is max over the 16 registers > threshold (8 v_max3 + 1
//                          compare, in the shadow of the next block's MFMAs) and only a block with a hit walks its registers.
//                          Bounds are exchanged between tiles through the same per-query word as K4 (loaded one period
//                          ahead, so the latency of the load is never waited for). Output = K4's partial-key layout.
typedef int mfma_i32x8 __attribute__((ext_vector_type(8)));
typedef float mfma_f32x16 __attribute__((ext_vector_type(16)));

// 32 descriptor bits -> 32 fp4 values (4 dwords): nibble i of out[j] = 0x2 | (bit (4 i + j) << 3)  (+1.0 / -1.0 in E2M1).
// The two constants live in registers (gfx9 VOP3 takes no literal), so each dword is one shift + one v_and_or_b32.
struct Fp4Consts { uint32_t sign, one; };
__device__ __forceinline__ Fp4Consts fp4_consts() {
  Fp4Consts k;
  asm volatile("s_mov_b32 %0, 0x88888888" : "=s"(k.sign));
  asm volatile("v_mov_b32 %0, 0x22222222" : "=v"(k.one));
  return k;
}
__device__ __forceinline__ mfma_i32x8 expand_word(uint32_t x, const Fp4Consts& k) {
  const int a = (int)(((x << 3) & k.sign) | k.one), b = (int)(((x << 2) & k.sign) | k.one),
            c = (int)(((x << 1) & k.sign) | k.one), d = (int)((x & k.sign) | k.one);
  return mfma_i32x8{a, b, c, d, 0, 0, 0, 0};
}

struct Fp4Row { mfma_i32x8 s[4]; };   // the lane's 128 bits of one row: 4 MFMA steps x 4 dwords (upper halves unused by fp4)

__device__ __forceinline__ void expand_row(const uint4& p, Fp4Row& f, const Fp4Consts& k) {
  f.s[0] = expand_word(p.x, k); f.s[1] = expand_word(p.y, k); f.s[2] = expand_word(p.z, k); f.s[3] = expand_word(p.w, k);
}

__device__ __forceinline__ float thr_of_limit(uint32_t limit) { return 256.f - 2.f * (float)limit; }   // dot > thr <=> d < limit

// 256 bit positions of 32 DB rows (A) x 32 queries (B): acc[i] of lane l = dot(row (i & 3) + 8 (i >> 2) + 4 (l >> 5), query l & 31)
__device__ __forceinline__ mfma_f32x16 dot_block(const Fp4Row& a, const Fp4Row& b) {
  mfma_f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a.s[s], b.s[s], acc, 4, 4, 0, 0, 0, 0);
  return acc;
}

// The test of one accumulator block: nothing to do unless some lane's best dot product beats its threshold (rare: the
// thresholds follow the k-th best distance found so far, anywhere in the DB); then walk the block's 16 rows. MASK: rows at
// or beyond n_lim do not exist (the last, partial step of the DB). IMAX: thresholds are >= 0 (radius < 128; they only rise),
// so the 16-way maximum may be taken on the raw bits as integers -- among non-negative floats the order is the same, and a
// negative dot product can never beat a non-negative threshold -- which spares the float maximum's NaN-quieting moves.
#ifdef TOD_K4X_COUNT_WALKS                                   // diagnostics build only (tools/k4x_walks.sh): blocks tested / blocks that walked
__device__ unsigned long long g_k4x_blocks[2];
#endif
template <int K, bool MASK, bool IMAX>
__device__ __forceinline__ void mfma_block_test(const mfma_f32x16& acc, float& thr, uint32_t r_lane, uint32_t n_lim,
                                                uint32_t (&best)[K]) {
#ifdef TOD_K4X_COUNT_WALKS
  if (!MASK && (threadIdx.x & 63u) == 0u) atomicAdd(&g_k4x_blocks[0], 1ull);
#endif
  if (!MASK) {
    bool any;
    if (IMAX) {                                            // (a tree: see mfma_block_test_part)
      int g[5];
#pragma unroll
      for (int j = 0; j < 5; ++j) g[j] = max(max(__float_as_int(acc[3 * j]), __float_as_int(acc[3 * j + 1])), __float_as_int(acc[3 * j + 2]));
      const int m = max(max(max(g[0], g[1]), g[2]), max(max(g[3], g[4]), __float_as_int(acc[15])));
      any = m > __float_as_int(thr);
    } else {
      float m = fmaxf(fmaxf(acc[0], acc[1]), acc[2]);
#pragma unroll
      for (int i = 3; i < 15; i += 2) m = fmaxf(fmaxf(m, acc[i]), acc[i + 1]);
      m = fmaxf(m, acc[15]);
      any = m > thr;
    }
    if (__builtin_amdgcn_ballot_w64(any) == 0ull) return;
#ifdef TOD_K4X_COUNT_WALKS
    if ((threadIdx.x & 63u) == 0u) atomicAdd(&g_k4x_blocks[1], 1ull);
#endif
  }
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    // the block's row i = r_lane | ((i & 3) + 8 (i >> 2)): r_lane = 32 step + 4 (lane >> 5) leaves bits 0, 1, 3, 4 free
    const uint32_t ri = (uint32_t)((i & 3) + 8 * (i >> 2));
    const bool hit = MASK ? (acc[i] > thr && (r_lane | ri) < n_lim) : (acc[i] > thr);
    if (__builtin_amdgcn_ballot_w64(hit) != 0ull) {
      // key = distance << 22 | row: (256 - dot) * 2^21 is an exact integer below 2^31. A stale (looser) threshold only
      // lets more rows try: the list keeps its k smallest keys whatever is offered
      const uint32_t key = (uint32_t)((256.f - acc[i]) * 2097152.f) | r_lane | ri;
      topk_insert<K>(best, hit ? key : 0xFFFFFFFFu);
    }
  }
  thr = fmaxf(thr, thr_of_limit(best[K - 1] >> kLocalBits));       // thresholds only ever tighten
}

// The same in two parts: the first SPLIT (2 or 3) of the block's 4 MFMAs, the rest behind a test (mfma_block_test_part)
template <int SPLIT>
__device__ __forceinline__ mfma_f32x16 dot_part0(const Fp4Row& a, const Fp4Row& b) {
  mfma_f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int s = 0; s < SPLIT; ++s) acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a.s[s], b.s[s], acc, 4, 4, 0, 0, 0, 0);
  return acc;
}
// Partial-distance elimination on the matrix cores (K4's idea, a block at a time): after P = 64 SPLIT of the 256 bit positions the
// accumulator holds P - 2 dP with dP <= d, so a pair whose partial dot product is not above thr - (256 - P) (dP >= limit) cannot
// be a hit whatever the other positions say -- exact for any data. On independent bits d128 of a non-match is 64 +- 5.7 and
// the radius 35: one block in five thousand goes on to its other two MFMAs (SPLIT 2). Real rBRIEF bits are biased and correlated
// (mean distance ~100 of 256 on this library's ORB descriptors of rendered views): there almost every block survives 128 positions
// and SPLIT 3 is the form that prunes (d192 ~ 75 +- 9.5). Needs thr - (256 - P) >= 0 for the integer maximum (limits up to 64 for
// SPLIT 2, up to 96 for SPLIT 3; thresholds only tighten). rows / q: the fragments the block's first part was computed from.
template <int K, int SPLIT>
__device__ __forceinline__ bool mfma_block_test_part(mfma_f32x16& acc, const Fp4Row& rows, const Fp4Row& q, float& thr, float& thrp,
                                                     uint32_t r_lane, uint32_t n_lim, uint32_t (&best)[K]) {
  // the 16-way maximum as a tree (5 independent max3, then 2 + 1): in this form the kernel is bound by vector issue, not by the matrix
  // pipe (tools/mfma_valu_overlap.hip: 2 MFMAs + chain + expansion 113 cycles per block and SIMD, + tree 103), and the tree's
  // independent operations fill the issue slots a chain leaves to its own latency. The part thresholds (thrp = thr - 64 (4 - SPLIT))
  // live in registers of their own beside the whole ones: one instruction less per block (1.82 -> 1.73 ms in the pipeline) for six
  // registers, 218 -> 224, still inside the budget that lets the other stages' kernels start beside the matcher's waves
  // (launch_topk_mfma; tests/test_build_checks.py holds the line). Keeping ONLY the part form -- no extra registers on paper -- made
  // hipcc allocate 243.
  int g[5];
#pragma unroll
  for (int j = 0; j < 5; ++j) g[j] = max(max(__float_as_int(acc[3 * j]), __float_as_int(acc[3 * j + 1])), __float_as_int(acc[3 * j + 2]));
  const int m = max(max(max(g[0], g[1]), g[2]), max(max(g[3], g[4]), __float_as_int(acc[15])));
  if (__builtin_amdgcn_ballot_w64(m > __float_as_int(thrp)) == 0ull) return false;
#pragma unroll
  for (int s = SPLIT; s < 4; ++s) acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(rows.s[s], q.s[s], acc, 4, 4, 0, 0, 0, 0);
  mfma_block_test<K, false, true>(acc, thr, r_lane, n_lim, best);
  thrp = thr - 64.f * (float)(4 - SPLIT);
  return true;
}

// One 32-row step: QT x 4 MFMAs against the resident query fragments. The test of block t-1 (and, in the first four
// blocks, the fp4 expansion of the NEXT step's packed rows) sits in the same basic block as the MFMAs of block t, so the
// vector ALU works in the matrix pipe's shadow; the last block's test is carried into the next step: QT is even, so it
// waits in acc_odd while block 0 of the next step fills acc_even.
// SPLIT 2 / 3 (never with MASK; 0 = whole blocks): every block starts with its first SPLIT MFMAs (dot_part0) and only completes
// behind mfma_block_test_part; the block carried in from the previous step (t == 0) completes with that step's rows, which are
// a_next's registers 2 and 3 until this step's expansion overwrites them at t == 2, 3.
template <int K, int QT, bool MASK, bool IMAX, int SPLIT = 0>
__device__ __forceinline__ uint32_t mfma_step(const Fp4Row& a, Fp4Row& a_next, const uint4& p_next, const Fp4Row (&qb)[QT],
                                              float (&thr)[QT], float (&thrp)[QT], uint32_t (&best)[QT][K], mfma_f32x16& acc_even,
                                              mfma_f32x16& acc_odd, uint32_t r_lane, uint32_t n_lim, const Fp4Consts& kc) {
  constexpr bool HALF = SPLIT != 0;
  uint32_t n_pass = 0;                                     // SPLIT: blocks that went on to their second part (wave-uniform)
  static_assert(!(HALF && MASK) && !(HALF && !IMAX) && !(HALF && QT < 4), "split blocks: unmasked steps, integer maximum, >= 4 query blocks");
  static_assert(SPLIT == 0 || SPLIT == 2 || SPLIT == 3, "2 or 3 of the 4 MFMAs first");
#pragma unroll
  for (int t = 0; t < QT; ++t) {
    if (HALF) { if (t & 1) acc_odd = dot_part0<HALF ? SPLIT : 2>(a, qb[t]); else acc_even = dot_part0<HALF ? SPLIT : 2>(a, qb[t]); }
    else { if (t & 1) acc_odd = dot_block(a, qb[t]); else acc_even = dot_block(a, qb[t]); }
#if defined(TOD_K4X_ABLATE) && TOD_K4X_ABLATE == 3           // diagnostics build only: no fp4 expansion (the packed words are "used")
    if (t == 0) { a_next = a; asm volatile("" :: "v"(p_next.x), "v"(p_next.y), "v"(p_next.z), "v"(p_next.w)); }
#else
    if (QT >= 4) {
      if (t == 0) a_next.s[0] = expand_word(p_next.x, kc);
      if (t == 1) a_next.s[1] = expand_word(p_next.y, kc);
      if (t == 2) a_next.s[2] = expand_word(p_next.z, kc);
      if (t == 3) a_next.s[3] = expand_word(p_next.w, kc);
    } else {                                             // two blocks per step: two words each
      if (t == 0) { a_next.s[0] = expand_word(p_next.x, kc); a_next.s[1] = expand_word(p_next.y, kc); }
      if (t == 1) { a_next.s[2] = expand_word(p_next.z, kc); a_next.s[3] = expand_word(p_next.w, kc); }
    }
#endif
#if defined(TOD_K4X_ABLATE) && TOD_K4X_ABLATE == 2           // diagnostics build only: no block test (the MFMAs stay: their results are "used")
    if (t == 0) asm volatile("" :: "v"(acc_odd)); else if (t & 1) asm volatile("" :: "v"(acc_even)); else asm volatile("" :: "v"(acc_odd));
#else
    if (HALF) {
      if (t == 0) n_pass += mfma_block_test_part<K, HALF ? SPLIT : 2>(acc_odd, a_next, qb[QT - 1], thr[QT - 1], thrp[QT - 1], r_lane - 32u, n_lim, best[QT - 1]) ? 1u : 0u;   // previous step's last block, its rows
      else n_pass += mfma_block_test_part<K, HALF ? SPLIT : 2>((t & 1) ? acc_even : acc_odd, a, qb[t - 1], thr[t - 1], thrp[t - 1], r_lane, n_lim, best[t - 1]) ? 1u : 0u;
    } else {
      if (t == 0) mfma_block_test<K, MASK, IMAX>(acc_odd, thr[QT - 1], r_lane - 32u, n_lim, best[QT - 1]);   // previous step's last block
      else mfma_block_test<K, MASK, IMAX>((t & 1) ? acc_even : acc_odd, thr[t - 1], r_lane, n_lim, best[t - 1]);
    }
#endif
  }
  return n_pass;
}

// MODE 0: float maximum in the block test (any radius); 1: integer maximum (cut <= 128: thresholds >= 0); 2 / 3: integer maximum and
// blocks split after 2 / 3 of their 4 MFMAs (cut <= 64 / <= 96: mfma_block_test_part)
template <int K, int QT, int MODE, bool PF2>
__global__ __launch_bounds__(kBlock, 2) void hamming_topk_mfma(const uint32_t* __restrict__ db,
                                                               const uint32_t* __restrict__ q, uint32_t n_rows,
                                                               uint32_t nq, uint32_t nq_pad, uint32_t rows_per_tile,
                                                               uint32_t n_tiles, uint32_t n_qw, uint32_t n_qw64,
                                                               uint32_t blocks_per_xcd, uint32_t tiles_per_xcd, uint32_t cut,
                                                               uint32_t share_period,
                                                               uint32_t* __restrict__ part, uint32_t* bound,
                                                               uint8_t* __restrict__ stored, uint32_t* half_stats) {
  static_assert(QT % 2 == 0 && QT >= 2, "two query blocks share a 64-query flag byte");
  constexpr bool IMAX = MODE >= 1;
  constexpr int HALF = (MODE >= 2 && QT >= 4) ? MODE : 0;           // the split (0: whole blocks)
  constexpr float kPartOff = HALF ? 64.f * (float)(4 - HALF) : 0.f;   // what the positions behind the split can still add to a dot product
  const uint32_t xcd = blockIdx.x & 7u, slot = blockIdx.x >> 3;
  uint32_t tile, qw;
  if (tiles_per_xcd) {
    const uint32_t local = __builtin_amdgcn_readfirstlane(slot * kWavesPerBlock + (threadIdx.x >> 6));
    if (local >= tiles_per_xcd * n_qw) return;
    tile = xcd * tiles_per_xcd + local / n_qw; qw = local % n_qw;
  } else {
    const uint32_t vblock = xcd * blocks_per_xcd + slot;
    const uint32_t item = __builtin_amdgcn_readfirstlane(vblock * kWavesPerBlock + (threadIdx.x >> 6));
    tile = item / n_qw; qw = item % n_qw;
  }
  if (tile >= n_tiles) return;
  const uint32_t lane = threadIdx.x & 63u, c = lane & 31u, h = lane >> 5;
  const uint32_t q0 = qw * (32u * QT);

  const Fp4Consts kc = fp4_consts();
  // query blocks beyond nq repeat the last query: their results are never stored
  Fp4Row qb[QT];
  uint32_t best[QT][K];
  float thr[QT], thrp[QT];                                         // thrp: the part thresholds of the split blocks (mfma_block_test_part)
#pragma unroll
  for (int t = 0; t < QT; ++t) {
    const uint32_t qi = q0 + 32u * t + c;
    const uint4 p = *reinterpret_cast<const uint4*>(q + (size_t)(qi < nq ? qi : nq - 1u) * kWords + 4u * h);
    expand_row(p, qb[t], kc);
#pragma unroll
    for (int j = 0; j < K; ++j) best[t][j] = 0xFFFFFFFFu;
    // cut = radius + 1: a row beyond the radius is dropped by the truncation (DescriptorMatcher.cpp:212-220) whatever its rank
    thr[t] = thr_of_limit(cut);
    thrp[t] = thr[t] - kPartOff;
  }

  const uint32_t row0 = tile * rows_per_tile;
  const uint32_t row_end = min(n_rows, row0 + rows_per_tile);
  const uint32_t n_local = row_end - row0;                          // > 0: tile < n_tiles
  const uint32_t n_full = n_local / 32u, n_steps = (n_local + 31u) / 32u;   // the DB's last step may be partial
  // this lane's 16 bytes of DB row (row0 + 32 step + c). No per-lane clamp: the DB's last step may reach up to 31 rows past its end
  // -- into the slack todhip_db_load leaves behind the descriptors (kDbSlackBytes), rows that are masked, never used -- and the
  // address stays a wave-uniform base plus a constant lane offset (no vector instruction per load: the kernel is bound by those)
  const uint32_t lane_off = (c * kWords + 4u * h) * 4u;              // bytes from the step's first row
  auto load_step = [&](uint32_t step) -> uint4 {
    const uint32_t first = row0 + 32u * min(step, n_steps - 1u);    // wave-uniform
#if defined(TOD_K4X_ABLATE) && TOD_K4X_ABLATE == 1           // diagnostics build only (tools/k4x_ablate.sh): no DB loads
    const uint32_t r = first + c;
    return uint4{r * 2654435761u, r ^ step, r + h, r * 40503u};
#else
    const char* base = reinterpret_cast<const char*>(db) + (size_t)first * (kWords * 4u);
    return *reinterpret_cast<const uint4*>(base + lane_off);
#endif
  };
  Fp4Row a0, a1;
  {
    const uint4 p = load_step(0);
    expand_row(p, a0, kc);
  }
  // packed rows in flight: of steps + 1 and + 2 (PF2), or of step + 1 only (4 registers less: what lets QT = 8 fit)
  uint4 pa = load_step(1), pb = PF2 ? load_step(2) : pa;
  mfma_f32x16 acc_even, acc_odd;                                    // acc_odd: pending block of the previous step -- none yet
#pragma unroll
  for (int i = 0; i < 16; ++i) acc_odd[i] = -4096.f;                      // below every threshold (256 - 2 * 1023 at the least)
  uint32_t seen[QT];
#pragma unroll
  for (int t = 0; t < QT; ++t) seen[t] = 0xFFFFFFFFu;              // "nothing published"
  uint32_t next_share = 2u;                                         // first exchange after 64 rows, as K4

  uint32_t step = 0;
  // HALF: every block of the unmasked steps starts as a half; the wave counts the blocks that went on to their second half, and
  // the HOST decides from the launch's totals whether the next launches use this mode at all (launch_topk_mfma_qt: on self-similar
  // texture most blocks go on and the half test only adds work). An in-kernel switch between the two loop bodies was tried: 79
  // spilled registers at the 256 this kernel lives on, 2.1 ms instead of 1.64.
  uint32_t n_pass = 0;
  for (; step + 2u <= n_full; step += 2u) {
    // two steps per trip: the expanded rows ping-pong between a0 and a1, the packed ones between pa and pb
    n_pass += mfma_step<K, QT, false, IMAX, HALF>(a0, a1, pa, qb, thr, thrp, best, acc_even, acc_odd, 32u * step + 4u * h, n_local, kc);
    if (PF2) {
      pa = load_step(step + 3u);
      n_pass += mfma_step<K, QT, false, IMAX, HALF>(a1, a0, pb, qb, thr, thrp, best, acc_even, acc_odd, 32u * step + 32u + 4u * h, n_local, kc);
      pb = load_step(step + 4u);
    } else {
      pa = load_step(step + 2u);
      n_pass += mfma_step<K, QT, false, IMAX, HALF>(a1, a0, pa, qb, thr, thrp, best, acc_even, acc_odd, 32u * step + 32u + 4u * h, n_local, kc);
      pa = load_step(step + 3u);
    }
    if (step + 2u >= next_share) {                                  // wave-uniform
      next_share += share_period;
      // take the bounds loaded one period ago (a published bound stays valid: bounds only fall), publish a full list's
      // bound if it improves on what was seen, start the loads of the next period
#pragma unroll
      for (int t = 0; t < QT; ++t) {
        const uint32_t qi = q0 + 32u * t + c;
        uint32_t* my_bound = bound + (qi < nq ? qi : nq - 1u);
        const uint32_t worst_d = best[t][K - 1] >> kLocalBits;
        if (worst_d < (0xFFFFFFFFu >> kLocalBits) && worst_d < seen[t]) atomicMin(my_bound, worst_d);
        // a foreign bound is applied with <=: a smaller row index elsewhere may still win a tie
        if (seen[t] != 0xFFFFFFFFu) { thr[t] = fmaxf(thr[t], thr_of_limit(seen[t] + 1u)); thrp[t] = thr[t] - kPartOff; }
        seen[t] = __hip_atomic_load(my_bound, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  }
  // the last unmasked step's last block is still a half: it completes here, with that step's rows (a1: the second step of the
  // loop's last trip ran on them) -- the masked steps and the drain below work on whole blocks
  if (HALF && step > 0u) {
    n_pass += mfma_block_test_part<K, HALF ? HALF : 2>(acc_odd, a1, qb[QT - 1], thr[QT - 1], thrp[QT - 1], 32u * (step - 1u) + 4u * h, n_local, best[QT - 1]) ? 1u : 0u;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc_odd[i] = -4096.f;
    if (lane == 0 && half_stats) { atomicAdd(half_stats, n_pass); atomicAdd(half_stats + 1, step * (uint32_t)QT); }
  }
  // at most one full and one partial step are left: the masked form serves both
  for (; step < n_steps; ++step) {
    mfma_step<K, QT, true, IMAX>(a0, a1, pa, qb, thr, thrp, best, acc_even, acc_odd, 32u * step + 4u * h, n_local, kc);
    a0 = a1;
    pa = PF2 ? pb : load_step(step + 2u);
  }
  mfma_block_test<K, true, IMAX>(acc_odd, thr[QT - 1], 32u * (n_steps - 1u) + 4u * h, n_local, best[QT - 1]);   // drain

  // lanes l and l + 32 hold the two halves of a query's rows: merge the partner's list, then K4's output format
  // (partial keys + one flag byte per (tile, 64 queries); two query blocks share a flag, so both are stored when
  // either kept something)
#pragma unroll
  for (int t = 0; t < QT; ++t) {
    uint32_t other[K];
#pragma unroll
    for (int j = 0; j < K; ++j) other[j] = __shfl_xor(best[t][j], 32);
#pragma unroll
    for (int j = 0; j < K; ++j) topk_insert<K>(best[t], other[j]);
  }
#pragma unroll
  for (int u = 0; u < QT / 2; ++u) {
    const uint32_t qa = q0 + 64u * u + c, qb2 = qa + 32u;
    const bool any_a = qa < nq && best[2 * u][0] != 0xFFFFFFFFu, any_b = qb2 < nq && best[2 * u + 1][0] != 0xFFFFFFFFu;
    if (__builtin_amdgcn_ballot_w64(any_a || any_b) != 0ull) {
      if (h == 0u) {
#pragma unroll
        for (int j = 0; j < K; ++j) {
          if (qa < nq) part[((size_t)tile * K + j) * nq_pad + qa] = best[2 * u][j];
          if (qb2 < nq) part[((size_t)tile * K + j) * nq_pad + qb2] = best[2 * u + 1][j];
        }
      }
      if (lane == 0) stored[(size_t)tile * n_qw64 + (q0 >> 6) + u] = 0;
    }
  }
}

// K4x for at most 32 queries: ONE query block per wave, so a 32-row step (1 KB of the DB) costs 4 MFMAs -- the matrix pipe
// could take 16 TB/s of rows at that rate, and the pass is bound by HBM alone (BASELINE.json's "achieved HBM GB/s on
// BF-matcher"; tools/k4_small_q.py). Same exact arithmetic, same per-lane lists, same output format as hamming_topk_mfma; the
// accumulators of consecutive steps alternate so that the test of step s runs beside the MFMAs of step s + 1, and four
// steps' packed rows are in flight per wave.
template <int K, bool IMAX>
__global__ __launch_bounds__(kBlock) void hamming_topk_mfma_q32(const uint32_t* __restrict__ db, const uint32_t* __restrict__ q,
                                                                uint32_t n_rows, uint32_t nq, uint32_t nq_pad, uint32_t rows_per_tile,
                                                                uint32_t n_tiles, uint32_t n_qw64, uint32_t cut, uint32_t share_period,
                                                                uint32_t* __restrict__ part, uint32_t* bound,
                                                                uint8_t* __restrict__ stored) {
  const uint32_t tile = __builtin_amdgcn_readfirstlane(blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6));
  if (tile >= n_tiles) return;
  const uint32_t lane = threadIdx.x & 63u, c = lane & 31u, h = lane >> 5;
  const Fp4Consts kc = fp4_consts();
  Fp4Row qb;
  {
    const uint4 p = *reinterpret_cast<const uint4*>(q + (size_t)(c < nq ? c : nq - 1u) * kWords + 4u * h);
    expand_row(p, qb, kc);
  }
  uint32_t best[K];
#pragma unroll
  for (int j = 0; j < K; ++j) best[j] = 0xFFFFFFFFu;
  float thr = thr_of_limit(cut);
  const uint32_t row0 = tile * rows_per_tile;
  const uint32_t n_local = min(n_rows, row0 + rows_per_tile) - row0;
  const uint32_t n_full = n_local / 32u, n_steps = (n_local + 31u) / 32u;
  const uint32_t last_row = n_rows - 1u;
  auto load_step = [&](uint32_t step) -> uint4 {
    const uint32_t r = min(row0 + 32u * min(step, n_steps - 1u) + c, last_row);
    return *reinterpret_cast<const uint4*>(db + (size_t)r * kWords + 4u * h);
  };
  uint4 p0 = load_step(0), p1 = load_step(1), p2 = load_step(2), p3 = load_step(3);
  mfma_f32x16 acc_a, acc_b;                                          // acc_b: pending block of the previous step -- none yet
#pragma unroll
  for (int i = 0; i < 16; ++i) acc_b[i] = -4096.f;                        // below every threshold (256 - 2 * 1023 at the least)
  uint32_t* my_bound = bound + (c < nq ? c : nq - 1u);
  uint32_t seen = 0xFFFFFFFFu, next_share = 2u, step = 0;
  for (; step + 4u <= n_full; step += 4u) {                          // four steps per trip: p0..p3 rotate by name, nothing is copied
    Fp4Row a;
    expand_row(p0, a, kc); p0 = load_step(step + 4u);
    acc_a = dot_block(a, qb);
    mfma_block_test<K, false, IMAX>(acc_b, thr, 32u * step - 32u + 4u * h, n_local, best);
    expand_row(p1, a, kc); p1 = load_step(step + 5u);
    acc_b = dot_block(a, qb);
    mfma_block_test<K, false, IMAX>(acc_a, thr, 32u * step + 4u * h, n_local, best);
    expand_row(p2, a, kc); p2 = load_step(step + 6u);
    acc_a = dot_block(a, qb);
    mfma_block_test<K, false, IMAX>(acc_b, thr, 32u * step + 32u + 4u * h, n_local, best);
    expand_row(p3, a, kc); p3 = load_step(step + 7u);
    acc_b = dot_block(a, qb);
    mfma_block_test<K, false, IMAX>(acc_a, thr, 32u * step + 64u + 4u * h, n_local, best);
    if (step + 4u >= next_share) {                                   // wave-uniform; as hamming_topk_mfma
      next_share += share_period;
      const uint32_t worst_d = best[K - 1] >> kLocalBits;
      if (worst_d < (0xFFFFFFFFu >> kLocalBits) && worst_d < seen) atomicMin(my_bound, worst_d);
      if (seen != 0xFFFFFFFFu) thr = fmaxf(thr, thr_of_limit(seen + 1u));
      seen = __hip_atomic_load(my_bound, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  // the pending block of the last full trip, then up to three full and one partial step, one at a time (masked form)
  mfma_block_test<K, true, IMAX>(acc_b, thr, 32u * step - 32u + 4u * h, step ? n_local : 0u, best);
  for (; step < n_steps; ++step) {
    Fp4Row a;
    expand_row(p0, a, kc);
    p0 = p1; p1 = p2; p2 = p3; p3 = load_step(step + 4u);
    acc_a = dot_block(a, qb);
    mfma_block_test<K, true, IMAX>(acc_a, thr, 32u * step + 4u * h, n_local, best);
  }
  uint32_t other[K];
#pragma unroll
  for (int j = 0; j < K; ++j) other[j] = __shfl_xor(best[j], 32);
#pragma unroll
  for (int j = 0; j < K; ++j) topk_insert<K>(best, other[j]);
  // nq <= 32: this block is the only one of its 64-query group, so the flag byte is this wave's alone
  if (__builtin_amdgcn_ballot_w64(c < nq && best[0] != 0xFFFFFFFFu) != 0ull) {
    if (h == 0u && c < nq) {
#pragma unroll
      for (int j = 0; j < K; ++j) part[((size_t)tile * K + j) * nq_pad + c] = best[j];
    }
    if (lane == 0) stored[(size_t)tile * n_qw64] = 0;
  }
}

// K4m: thread (query, group) merges the tiles t = group, group + G, ... ; keys become
// (distance << 32 | global_row), unique per row, so any merge order gives the same k smallest.
// Output layout [group][nq][K] == the [shard][nq][k] layout finalize_kernel consumes.
template <int K>
__global__ __launch_bounds__(kBlock) void merge_tiles_kernel(const uint32_t* __restrict__ part, uint32_t nq,
                                                             uint32_t nq_pad, uint32_t n_tiles,
                                                             uint32_t rows_per_tile, uint64_t first_global_row,
                                                             uint32_t n_groups, const uint8_t* __restrict__ stored,
                                                             uint32_t n_qw, uint64_t* __restrict__ keys,
                                                             const uint32_t* stat_src = nullptr, uint32_t* stat_dst = nullptr,
                                                             uint32_t stat_seq = 0) {
  TOD_LATENCY_PRIO();   // latency-bound: win issue arbitration against the VALU-saturating matcher
  const uint32_t qi = blockIdx.x * kBlock + threadIdx.x;
  const uint32_t grp = blockIdx.y;
  if (stat_dst && qi == 0u && grp == 0u) {                  // the DB pass's half-block counters -> pinned host memory (launch_topk_mfma_qt)
    for (int i = 0; i < 4; ++i) stat_dst[i] = __hip_atomic_load(stat_src + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __threadfence_system();
    stat_dst[4] = stat_seq;
  }
  if (qi >= nq) return;
  uint64_t best[K];
#pragma unroll
  for (int j = 0; j < K; ++j) best[j] = ~0ull;
#pragma unroll 4
  for (uint32_t t = grp; t < n_tiles; t += n_groups) {
    if (stored[(size_t)t * n_qw + (qi >> 6)] != 0) continue;   // this tile kept nothing for the 64 queries around qi
    uint32_t pk[K];
#pragma unroll
    for (int j = 0; j < K; ++j) pk[j] = part[((size_t)t * K + j) * nq_pad + qi];
#pragma unroll
    for (int j = 0; j < K; ++j) {
      uint64_t key = pk[j] == 0xFFFFFFFFu
                         ? ~0ull
                         : (((uint64_t)(pk[j] >> kLocalBits) << 32) |
                            (first_global_row + (uint64_t)t * rows_per_tile + (pk[j] & kLocalMask)));
#pragma unroll
      for (int s2 = 0; s2 < K; ++s2) {
        uint64_t lo = key < best[s2] ? key : best[s2];
        key = key < best[s2] ? best[s2] : key;
        best[s2] = lo;
      }
    }
  }
#pragma unroll
  for (int j = 0; j < K; ++j) keys[((size_t)grp * nq + qi) * K + j] = best[j];
}

// K4m for a handful of queries (the <= 32-query regime: 8192 tiles, and merge_tiles_kernel's 16 queries x 16 groups = 256 threads walk
// 512 tiles each, 71 us behind a 217 us DB pass): one WAVE per (query, group), the group's tiles spread over its lanes, the lanes'
// lists merged by a butterfly of shuffles. Same keys, same output layout.
template <int K>
__global__ __launch_bounds__(kBlock) void merge_tiles_wave_kernel(const uint32_t* __restrict__ part, uint32_t nq, uint32_t nq_pad,
                                                                  uint32_t n_tiles, uint32_t rows_per_tile, uint64_t first_global_row,
                                                                  uint32_t n_groups, const uint8_t* __restrict__ stored, uint32_t n_qw,
                                                                  uint64_t* __restrict__ keys) {
  TOD_LATENCY_PRIO();
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t qi = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6), grp = blockIdx.y;
  if (qi >= nq) return;                                     // (wave-uniform)
  uint64_t best[K];
#pragma unroll
  for (int j = 0; j < K; ++j) best[j] = ~0ull;
  for (uint32_t t = grp + n_groups * lane; t < n_tiles; t += n_groups * 64u) {
    if (stored[(size_t)t * n_qw + (qi >> 6)] != 0) continue;
#pragma unroll
    for (int j = 0; j < K; ++j) {
      const uint32_t pk = part[((size_t)t * K + j) * nq_pad + qi];
      uint64_t key = pk == 0xFFFFFFFFu ? ~0ull
                                       : (((uint64_t)(pk >> kLocalBits) << 32) | (first_global_row + (uint64_t)t * rows_per_tile + (pk & kLocalMask)));
#pragma unroll
      for (int s2 = 0; s2 < K; ++s2) { const uint64_t lo = key < best[s2] ? key : best[s2]; key = key < best[s2] ? best[s2] : key; best[s2] = lo; }
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    uint64_t other[K];
#pragma unroll
    for (int j = 0; j < K; ++j) other[j] = __shfl_xor(best[j], off);
#pragma unroll
    for (int j = 0; j < K; ++j) {
      uint64_t key = other[j];
#pragma unroll
      for (int s2 = 0; s2 < K; ++s2) { const uint64_t lo = key < best[s2] ? key : best[s2]; key = key < best[s2] ? best[s2] : key; best[s2] = lo; }
    }
  }
  if (lane == 0) {
#pragma unroll
    for (int j = 0; j < K; ++j) keys[((size_t)grp * nq + qi) * K + j] = best[j];
  }
}

// K4s: per query, the k smallest of n_lists ascending lists (layout [list][nq][k]) -> keys[nq][k].
__global__ __launch_bounds__(kBlock) void select_keys_kernel(const uint64_t* __restrict__ lists, uint32_t n_lists,
                                                             uint32_t nq, uint32_t k, uint64_t* __restrict__ keys) {
  TOD_LATENCY_PRIO();   // latency-bound: win issue arbitration against the VALU-saturating matcher
  const uint32_t qi = blockIdx.x * kBlock + threadIdx.x;
  if (qi >= nq) return;
  uint64_t last = 0;
  bool have_last = false;
  for (uint32_t j = 0; j < k; ++j) {
    uint64_t nxt = ~0ull;
    for (uint32_t s = 0; s < n_lists; ++s) {
      const uint64_t* lst = lists + ((size_t)s * nq + qi) * k;
      for (uint32_t i = 0; i < k; ++i) {
        uint64_t v = lst[i];
        if (have_last && v <= last) continue;
        if (v < nxt) nxt = v;
        break;
      }
    }
    keys[(size_t)qi * k + j] = nxt;
    if (nxt == ~0ull) {
      for (uint32_t jj = j + 1; jj < k; ++jj) keys[(size_t)qi * k + jj] = ~0ull;
      break;
    }
    last = nxt;
    have_last = true;
  }
}

// K4f: merge the shard lists (layout [shard][nq][k]), truncate at the first distance > radius
// (DescriptorMatcher.cpp:212-220), map the global row to (imgIdx, trainIdx) through the object prefix
// sums (DB load order, :60-129) and gather the model point of every kept match (:231-244).
__global__ __launch_bounds__(kBlock) void finalize_kernel(const uint64_t* __restrict__ keys_all, uint32_t n_shards,
                                                          uint32_t nq, uint32_t k_in, uint32_t k_out, uint32_t radius, float ratio,
                                                          const uint32_t* __restrict__ obj_off, uint32_t n_objs,
                                                          const float* __restrict__ pts,
                                                          uint32_t* __restrict__ counts,
                                                          todhip_dmatch* __restrict__ matches,
                                                          float* __restrict__ xyz) {
  TOD_LATENCY_PRIO();   // latency-bound: win issue arbitration against the VALU-saturating matcher
  const uint32_t qi = blockIdx.x * kBlock + threadIdx.x;
  if (qi >= nq) return;
  // the k_out (for the ratio test: at least 2) smallest keys over all lists: every list is ascending and keys are unique
  // (the row is part of the key), so each list's candidate is its first key greater than the last one taken
  uint64_t picked[9];
  const uint32_t want = ratio > 0.f ? max(k_out, 2u) : k_out;
  uint32_t n_picked = 0;
  uint64_t last = 0;
  bool have_last = false;
  for (uint32_t j = 0; j < want; ++j) {
    uint64_t nxt = ~0ull;
    for (uint32_t s = 0; s < n_shards; ++s) {
      const uint64_t* lst = keys_all + ((size_t)s * nq + qi) * k_in;
      for (uint32_t i = 0; i < k_in; ++i) {
        uint64_t v = lst[i];
        if (have_last && v <= last) continue;
        if (v < nxt) nxt = v;
        break;
      }
    }
    if (nxt == ~0ull) break;
    last = nxt;
    have_last = true;
    picked[n_picked++] = nxt;
  }
  // Lowe's ratio test on the two nearest neighbours (the block the reference leaves empty, DescriptorMatcher.cpp:223-227;
  // definition: include/todhip.h, todhip_set_ratio_test): an ambiguous query keeps nothing
  if (ratio > 0.f && n_picked >= 2u && !((float)(uint32_t)(picked[0] >> 32) < ratio * (float)(uint32_t)(picked[1] >> 32))) n_picked = 0;
  uint32_t kept = 0;
  for (uint32_t j = 0; j < n_picked && j < k_out; ++j) {
    const uint32_t d = (uint32_t)(picked[j] >> 32);
    if ((float)d > (float)radius) break;                    // radius truncation, :212-220 (float vs unsigned compare)
    const uint32_t row = (uint32_t)picked[j];
    uint32_t lo = 0, hi = n_objs;            // last object whose first row is <= row
    while (hi - lo > 1) {
      uint32_t mid = (lo + hi) >> 1;
      if (obj_off[mid] <= row) lo = mid; else hi = mid;
    }
    todhip_dmatch m;
    m.queryIdx = (int32_t)qi;
    m.trainIdx = (int32_t)(row - obj_off[lo]);
    m.imgIdx = (int32_t)lo;
    m.distance = (float)d;
    matches[(size_t)qi * k_out + kept] = m;
    float* o = xyz + ((size_t)qi * k_out + kept) * 3;
    o[0] = pts[(size_t)row * 3 + 0];
    o[1] = pts[(size_t)row * 3 + 1];
    o[2] = pts[(size_t)row * 3 + 2];
    ++kept;
  }
  counts[qi] = kept;
}

// 0: K4 on the VALU, 1: K4x on the matrix cores. todhip_set_matcher_engine() decides; while it says "auto" the
// environment variable TODHIP_K4_ENGINE=valu|mfma does (whole test suites can be run on either engine that way).
int k4_engine(const todhip_ctx* ctx, uint32_t nq) {
  if (ctx->matcher_engine == TODHIP_ENGINE_VALU) return 0;
  if (ctx->matcher_engine == TODHIP_ENGINE_MFMA) return 1;
  static const char* env = getenv("TODHIP_K4_ENGINE");
  if (env && env[0] == 'v') return 0;
  if (env && env[0] == 'm') return 1;
  // Measured (tools/k4_engines.py, ms per launch K4 | K4x): 16 000 x 1M 3.2 | 1.1 on independent bits and 5.7 | 1.35 on this
  // repo's ORB descriptors; 1000 x 1M 0.25 | 0.085; 1000 x 100k 0.035 | 0.021; 500 x 5000 0.008 | 0.013. The matrix form
  // pays from ~2^24 pairs on; below, a wave's fixed start-up costs more than it saves. (Few queries over a big DB are
  // matrix-engine work too: both engines pad to 64 query columns, and 8 MFMAs per KB of rows keep up with HBM.)
  return (uint64_t)nq * ctx->shard_rows >= (1ull << 24) ? 1 : 0;
}

template <int K, int QT>
int launch_topk_mfma_qt(todhip_ctx* ctx, const uint32_t* d_q, uint32_t nq, uint32_t radius, uint64_t* d_lists, uint32_t* n_lists) {
  constexpr bool PF2 = QT < 8;                                        // register budget: see hamming_topk_mfma
  const uint32_t cut = radius >= 256u ? 0xFFFFFFFFu >> kLocalBits : radius + 1u;   // distances are <= 256: no cut beyond that
  const uint32_t n_rows = (uint32_t)ctx->shard_rows;
  const uint32_t n_qw = (nq + 32u * QT - 1u) / (32u * QT), n_qw64 = (nq + 63u) / 64u;
  const uint32_t nq_pad = n_qw64 * 64u;
  // Rounds of waves: the chip holds 8 of these waves per CU (2 per SIMD, by registers). A launch whose wave count is just
  // under a whole number of rounds has no straggling last round (measured, tools/k4x_sweep.py, 16 000 x 1M: 16 waves per
  // CU = 2 rounds 1.13 ms, 14 = 1.6 rounds 1.37 ms, 8 = all resident 1.36 ms, 32 .. 128 1.11 ms); four rounds while a tile
  // then still has >= 48 steps (a tile starts with empty lists), two otherwise. Tiles are whole 32-row steps.
  static const int env_wpc = getenv("TODHIP_K4X_WAVES_PER_CU") ? atoi(getenv("TODHIP_K4X_WAVES_PER_CU")) : 0;   // tuning knobs, read once per process
  uint32_t wpc = 32;
  if ((uint64_t)n_rows * n_qw < (uint64_t)ctx->n_cu * wpc * 1536u) wpc = 16;
  // A tile starts with empty lists and the radius as its threshold, and every row inside the threshold costs a walk of its block
  // until the list's k-th entry tightens it. On independent bits almost no row is; on self-similar texture (rendered views of
  // rectangle patterns: a median of 1300 rows of 1M within 35 bits of a query, tools/count_close_rows.py) tiles of a few hundred
  // rows spend their life in that walk. One frame's launch therefore gets at most 8 waves per CU (tiles of >= ~4000 rows)
  // and 4 query blocks per wave (launch_topk_mfma): 0.32 -> 0.19 ms on such a frame, 0.096 -> 0.095 ms on independent bits
  // (tools/k4x_chained_frame.sh, tools/k4x_synth_frame.sh).
  if ((uint64_t)n_rows * n_qw < (uint64_t)ctx->n_cu * 16u * 4096u) wpc = 8;
  if (env_wpc > 0) wpc = (uint32_t)env_wpc;
  uint32_t n_tiles = std::max(1u, (uint32_t)ctx->n_cu * wpc / n_qw);
  n_tiles = std::min(n_tiles, std::max(1u, n_rows / 256u));
  n_tiles = std::min(n_tiles, 8192u);
  if (n_tiles >= 16) n_tiles &= ~7u;                                  // whole tiles per XCD, never more waves than asked for
  uint32_t rows_per_tile = (n_rows + n_tiles - 1) / n_tiles;
  rows_per_tile = (rows_per_tile + 31u) & ~31u;
  if (rows_per_tile > kLocalMask) return TODHIP_EINVAL;
  n_tiles = (n_rows + rows_per_tile - 1) / rows_per_tile;
  const uint32_t items = n_tiles * n_qw;
  const uint32_t blocks = (items + kWavesPerBlock - 1) / kWavesPerBlock;
  uint32_t blocks_per_xcd = (blocks + 7u) / 8u;
  uint32_t tiles_per_xcd = 0;
  if (n_tiles >= 8 && n_tiles % 8u == 0) {
    tiles_per_xcd = n_tiles / 8u;
    blocks_per_xcd = (tiles_per_xcd * n_qw + kWavesPerBlock - 1) / kWavesPerBlock;
  }
  static const int env_share = getenv("TODHIP_K4X_SHARE") ? atoi(getenv("TODHIP_K4X_SHARE")) : 16;
  const uint32_t groups = n_tiles < (uint32_t)kMergeGroups ? n_tiles : (uint32_t)kMergeGroups;
  TOD_HIP(ctx->m_part.reserve((size_t)n_tiles * K * nq_pad * sizeof(uint32_t)));
  const size_t bound_bytes = (size_t)nq_pad * sizeof(uint32_t), flag_bytes = (size_t)n_tiles * n_qw64;
  TOD_HIP(ctx->m_bound.reserve(bound_bytes + flag_bytes));
  TOD_HIP(hipMemsetAsync(ctx->m_bound.p, 0xFF, bound_bytes + flag_bytes, ctx->stream));
  uint8_t* const d_stored = ctx->m_bound.as<uint8_t>() + bound_bytes;
  int slot = -1;
  if (ctx->time_kernels) { int rc = tod_timing_begin(ctx, &slot); if (rc != TODHIP_OK) return rc; }
  // radius < 128: every threshold is >= 0 and the block test may compare raw bits (see mfma_block_test). cut <= 64 / <= 96: the
  // thresholds minus 128 / 64 are >= 0 as well and a block may be split after 2 / 3 of its 4 MFMAs (mfma_block_test_part). Which
  // split pays is a property of the DATA (independent bits: 2; this library's ORB descriptors of rendered views: 3, since nearly
  // every block survives 128 positions there and the 2-split then costs +30 %), so the launch adapts: a split launch counts the blocks
  // that went on to their second part, the merge kernel behind it leaves the totals in pinned memory, and the context moves one
  // level up (2 -> 3 -> whole blocks) when more than a quarter of the blocks went on, and probes one level down every k4x_hold_len
  // launches (32, doubling to 256 while the probes keep failing). todhip_set_matcher_block_split, or TODHIP_K4X_HALF=0 / 2 / 3 as the
  // process's default: never / always that split (1 = 2).
  static const int env_default = getenv("TODHIP_K4X_HALF") ? atoi(getenv("TODHIP_K4X_HALF")) : -1;
  const int env_half = ctx->k4x_force >= 0 ? ctx->k4x_force : env_default;                 // todhip_set_matcher_block_split wins
  const uint32_t min_split = QT < 4 ? 4u : (cut <= 64u ? 2u : (cut <= 96u ? 3u : 4u));   // the lowest split the thresholds allow
  uint32_t split = 4;                                                                   // 4 = whole blocks
  const bool adaptive = env_half < 0;
  if (env_half == 0 || min_split == 4u) split = 4;
  else {
    if (!ctx->k4x_stats_host.p) {
      TOD_HIP(ctx->k4x_stats_host.reserve(64));
      std::memset(ctx->k4x_stats_host.p, 0, 64);
      TOD_HIP(ctx->k4x_stats_dev.reserve(64));
      TOD_HIP(hipMemsetAsync(ctx->k4x_stats_dev.p, 0, 64, ctx->stream));
    }
    if (ctx->k4x_split < min_split) ctx->k4x_split = min_split;
    volatile uint32_t* hs = ctx->k4x_stats_host.as<uint32_t>();   // [0..1] split 2: blocks that went on, blocks; [2..3] split 3; [4] launches reported
    const uint32_t seq_now = hs[4];
    if (seq_now != ctx->k4x_seq_seen) {                                              // a split launch has reported since the last look
      ctx->k4x_seq_seen = seq_now;
      static const bool dbg = getenv("TODHIP_K4X_HALF_DEBUG") != nullptr;
      for (uint32_t m = 2; m <= 3; ++m) {
        const uint32_t pass = hs[2 * (m - 2)] - ctx->k4x_last[2 * (m - 2)], blocks = hs[2 * (m - 2) + 1] - ctx->k4x_last[2 * (m - 2) + 1];
        if (!blocks) continue;
        ctx->k4x_last[2 * (m - 2)] += pass; ctx->k4x_last[2 * (m - 2) + 1] += blocks;
        ctx->counters.k4x_half_blocks += blocks; ctx->counters.k4x_half_blocks_completed += pass;
        const bool pays = (uint64_t)pass * 4u <= blocks;     // (measured on the rendered-view DB: 48 % going on at split 3 = 2.71 ms, whole blocks 2.60)
        if (dbg) fprintf(stderr, "[todhip] K4x blocks split after %u MFMAs: %u of %u went on (%.3f)%s, split in use %u\n", m, pass, blocks,
                         (double)pass / blocks, pays ? "" : ": does not pay", ctx->k4x_split);
        if (!adaptive) continue;
        if (m == ctx->k4x_split && !pays) {                  // the level in use stopped paying: one level up
          ctx->k4x_split = m + 1; ctx->k4x_hold = ctx->k4x_hold_len = 32;
        } else if (m + 1 == ctx->k4x_split) {                // a probe's report
          if (pays) { ctx->k4x_split = m; ctx->k4x_hold_len = 32; }
          else ctx->k4x_hold_len = std::min(256u, ctx->k4x_hold_len * 2u);
        }
      }
    }
    split = adaptive ? ctx->k4x_split : std::max<uint32_t>(min_split, env_half == 1 ? 2u : (uint32_t)std::min(env_half, 3));
    if (adaptive && split > min_split) {
      if (ctx->k4x_hold == 0) { split -= 1; ctx->k4x_hold = ctx->k4x_hold_len ? ctx->k4x_hold_len : 32; }   // probe one level down
      else --ctx->k4x_hold;
    }
  }
  auto kern = split == 2 ? hamming_topk_mfma<K, QT, 2, PF2>
              : split == 3 ? hamming_topk_mfma<K, QT, 3, PF2>
              : (cut <= 128u ? hamming_topk_mfma<K, QT, 1, PF2> : hamming_topk_mfma<K, QT, 0, PF2>);
  const bool report = split < 4;
  ctx->counters.last_block_split = split;
  uint32_t* const d_stats = report ? ctx->k4x_stats_dev.as<uint32_t>() : nullptr;
  hipLaunchKernelGGL(kern, dim3(blocks_per_xcd * 8u), dim3(kBlock), 0, ctx->stream,
                     ctx->db_desc.as<uint32_t>(), d_q, n_rows, nq, nq_pad, rows_per_tile, n_tiles, n_qw, n_qw64,
                     blocks_per_xcd, tiles_per_xcd, cut, (uint32_t)std::max(2, env_share), ctx->m_part.as<uint32_t>(),
                     ctx->m_bound.as<uint32_t>(), d_stored, d_stats ? d_stats + 2u * (split - 2u) : nullptr);
  if (slot >= 0) { int rc = tod_timing_end(ctx, slot); if (rc != TODHIP_OK) return rc; }
  if (d_stats) ++ctx->k4x_seq_sent;
  if (nq <= 64u && !d_stats && n_tiles >= 256u)              // a handful of queries over thousands of tiles: a wave per (query, group)
    hipLaunchKernelGGL(merge_tiles_wave_kernel<K>, dim3((nq + kWavesPerBlock - 1) / kWavesPerBlock, groups), dim3(kBlock), 0, ctx->stream,
                       ctx->m_part.as<uint32_t>(), nq, nq_pad, n_tiles, rows_per_tile, ctx->shard_first, groups, d_stored, n_qw64, d_lists);
  else
    hipLaunchKernelGGL(merge_tiles_kernel<K>, dim3((nq + kBlock - 1) / kBlock, groups), dim3(kBlock), 0, ctx->stream,
                       ctx->m_part.as<uint32_t>(), nq, nq_pad, n_tiles, rows_per_tile, ctx->shard_first, groups,
                       d_stored, n_qw64, d_lists, (const uint32_t*)d_stats, d_stats ? ctx->k4x_stats_host.as<uint32_t>() : (uint32_t*)nullptr,
                       ctx->k4x_seq_sent);
  TOD_HIP(hipGetLastError());
  *n_lists = groups;
  return TODHIP_OK;
}

template <int K>
int launch_topk_mfma_q32(todhip_ctx* ctx, const uint32_t* d_q, uint32_t nq, uint32_t radius, uint64_t* d_lists, uint32_t* n_lists) {
  const uint32_t cut = radius >= 256u ? 0xFFFFFFFFu >> kLocalBits : radius + 1u;
  const uint32_t n_rows = (uint32_t)ctx->shard_rows, n_qw64 = 1u, nq_pad = 64u;
  // one wave per tile; about 32 waves per CU in all (each holds four 1 KB loads in flight), tiles of >= 2048 rows
  uint32_t n_tiles = std::max(1u, std::min<uint32_t>((uint32_t)ctx->n_cu * 32u, n_rows / 2048u));
  n_tiles = std::min(n_tiles, 8192u);
  uint32_t rows_per_tile = ((n_rows + n_tiles - 1) / n_tiles + 31u) & ~31u;
  if (rows_per_tile > kLocalMask) return TODHIP_EINVAL;
  n_tiles = (n_rows + rows_per_tile - 1) / rows_per_tile;
  static const int env_share = getenv("TODHIP_K4X_SHARE") ? atoi(getenv("TODHIP_K4X_SHARE")) : 16;
  const uint32_t groups = n_tiles < (uint32_t)kMergeGroups ? n_tiles : (uint32_t)kMergeGroups;
  TOD_HIP(ctx->m_part.reserve((size_t)n_tiles * K * nq_pad * sizeof(uint32_t)));
  const size_t bound_bytes = (size_t)nq_pad * sizeof(uint32_t), flag_bytes = (size_t)n_tiles * n_qw64;
  TOD_HIP(ctx->m_bound.reserve(bound_bytes + flag_bytes));
  TOD_HIP(hipMemsetAsync(ctx->m_bound.p, 0xFF, bound_bytes + flag_bytes, ctx->stream));
  uint8_t* const d_stored = ctx->m_bound.as<uint8_t>() + bound_bytes;
  int slot = -1;
  if (ctx->time_kernels) { int rc = tod_timing_begin(ctx, &slot); if (rc != TODHIP_OK) return rc; }
  auto kern = cut <= 128u ? hamming_topk_mfma_q32<K, true> : hamming_topk_mfma_q32<K, false>;
  hipLaunchKernelGGL(kern, dim3((n_tiles + kWavesPerBlock - 1) / kWavesPerBlock), dim3(kBlock), 0, ctx->stream,
                     ctx->db_desc.as<uint32_t>(), d_q, n_rows, nq, nq_pad, rows_per_tile, n_tiles, n_qw64, cut,
                     (uint32_t)std::max(4, env_share), ctx->m_part.as<uint32_t>(), ctx->m_bound.as<uint32_t>(), d_stored);
  if (slot >= 0) { int rc = tod_timing_end(ctx, slot); if (rc != TODHIP_OK) return rc; }
  hipLaunchKernelGGL(merge_tiles_wave_kernel<K>, dim3((nq + kWavesPerBlock - 1) / kWavesPerBlock, groups), dim3(kBlock), 0, ctx->stream,
                     ctx->m_part.as<uint32_t>(), nq, nq_pad, n_tiles, rows_per_tile, ctx->shard_first, groups,
                     d_stored, n_qw64, d_lists);
  TOD_HIP(hipGetLastError());
  *n_lists = groups;
  return TODHIP_OK;
}

template <int K>
int launch_topk_mfma(todhip_ctx* ctx, const uint32_t* d_q, uint32_t nq, uint32_t radius, uint64_t* d_lists, uint32_t* n_lists) {
  static const int env_qt = getenv("TODHIP_K4X_QT") ? atoi(getenv("TODHIP_K4X_QT")) : 0;     // tuning knob, read once per process
  if (nq <= 32u && !(env_qt > 0))
    return launch_topk_mfma_q32<K>(ctx, d_q, nq, radius, d_lists, n_lists);
  // Query blocks of 32 per wave (QT): the registers hold 8 beside the k-entry lists for k <= 2, 6 for k <= 5 (the reference's k,
  // DescriptorMatcher.cpp:211), 4 beyond. Six is the default even where eight fit: with eight a wave takes 256 registers, two waves
  // fill a SIMD's file, and every kernel of the other stages (ORB, verifier) then waits for a matcher workgroup to retire before
  // one of its own can start -- a quarter of a DB pass, 25 dependent launches per ORB batch. With six (213-221 registers, allocated
  // in eights: 64-80 of a SIMD's 512 stay free) those kernels run beside the matcher's waves: alone the pass is 2 % slower (2.36 vs 2.32 ms whole blocks), in the
  // pipeline ORB's stage falls from 1.9 to 1.2 ms, the verifier's from 2.05 to 1.5, and the matcher's own launch is no slower
  // (tools/ab_k4x_residency.sh: headline 16.2k -> 16.6k frames/s, chained 8.7k -> 9.9k). Among the candidates the one that pads nq
  // the least wins when that saves more than 3 % (a wave computes all its blocks; 1000 queries are 4 x 256 but 6 x 192).
  // TODHIP_K4X_QT forces one (experiments).
  // With at most 64 queries a wave holds two blocks (QT = 2): 8 MFMAs per 1 KB of rows -- the pass is then bound by HBM,
  // not by the matrix pipe (BASELINE.json's "achieved HBM GB/s on BF-matcher" regime; tools/k4_small_q.py).
  constexpr int kMaxQT = K <= 2 ? 8 : (K <= 5 ? 6 : 4);
  auto padded = [&](uint32_t qt) { return (uint64_t)((nq + 32u * qt - 1u) / (32u * qt) * (32u * qt)); };
  int qt = kMaxQT >= 6 ? 6 : 4;
  if (kMaxQT >= 8 && padded(8) * 103u < padded(6) * 100u) qt = 8;
  if (qt > 4 && padded(4) * 103u < padded((uint32_t)qt) * 100u) qt = 4;
  if (padded(2) * 103u < padded((uint32_t)qt) * 100u) qt = 2;
  if (nq <= 2048u && qt > 4) qt = 4;                                  // a frame or two: longer tiles, see launch_topk_mfma_qt
  if ((env_qt == 2 || env_qt == 4 || env_qt == 6 || env_qt == 8) && env_qt <= kMaxQT) qt = env_qt;
  if (qt == 8) return launch_topk_mfma_qt<K, (kMaxQT >= 8 ? 8 : 4)>(ctx, d_q, nq, radius, d_lists, n_lists);
  if (qt == 6) return launch_topk_mfma_qt<K, (kMaxQT >= 6 ? 6 : 4)>(ctx, d_q, nq, radius, d_lists, n_lists);
  if (qt == 2) return launch_topk_mfma_qt<K, 2>(ctx, d_q, nq, radius, d_lists, n_lists);
  return launch_topk_mfma_qt<K, 4>(ctx, d_q, nq, radius, d_lists, n_lists);
}

template <int K>
int launch_topk(todhip_ctx* ctx, const uint32_t* d_q, uint32_t nq, uint32_t radius, uint64_t* d_lists, uint32_t* n_lists) {
  if (k4_engine(ctx, nq) == 1) return launch_topk_mfma<K>(ctx, d_q, nq, radius, d_lists, n_lists);
  const uint32_t cut = radius >= 256u ? 0xFFFFFFFFu : radius + 1u;   // distances are <= 256: no cut beyond that
  const uint32_t n_rows = (uint32_t)ctx->shard_rows;
  const uint32_t n_qw = (nq + 63u) / 64u;
  const uint32_t nq_pad = n_qw * 64u;
  // Tiling (measured, tools/time_k4.py): about three times more waves than fit the chip at once and tiles of at most
  // ~2048 rows (16 000 queries x 1M rows: 3.27 ms with 4096-row tiles, 3.20 ms with 2048; 10M rows: 35.6 -> 34.3 ms).
  // An exactly-resident grid of long-running waves (the first design) lost 15-20 %: the hardware does not spread
  // blocks evenly over the CUs, and the query waves of a large tile drift apart in it (scalar-cache and L2 misses);
  // short blocks rebalance by themselves and keep a tile's readers together.
  static const int env_wpc = getenv("TODHIP_K4_WAVES_PER_CU") ? atoi(getenv("TODHIP_K4_WAVES_PER_CU")) : 0;   // tuning knob
  uint32_t n_tiles;
  if (env_wpc > 0) {
    n_tiles = (uint32_t)ctx->n_cu * (uint32_t)env_wpc / n_qw;
  } else {
    n_tiles = std::max(3u * (uint32_t)ctx->n_cu * (uint32_t)kWavesPerCU / n_qw, (n_rows + 2047u) / 2048u);
    n_tiles = std::min(n_tiles, 8192u);
  }
  if (n_tiles < 1) n_tiles = 1;
  if (n_tiles >= 8) n_tiles = (n_tiles + 7u) & ~7u;      // whole tiles per XCD (8 XCDs)
  uint32_t rows_per_tile = (n_rows + n_tiles - 1) / n_tiles;
  rows_per_tile = ((rows_per_tile + 2 * kGroupRows - 1) / (2 * kGroupRows)) * (2 * kGroupRows);
  if (rows_per_tile < 64) rows_per_tile = 64;
  if (rows_per_tile > kLocalMask) return TODHIP_EINVAL;
  n_tiles = (n_rows + rows_per_tile - 1) / rows_per_tile;
  const uint32_t items = n_tiles * n_qw;
  const uint32_t blocks = (items + kWavesPerBlock - 1) / kWavesPerBlock;
  uint32_t blocks_per_xcd = (blocks + 7u) / 8u;
  uint32_t tiles_per_xcd = 0;
  if (n_tiles >= 8 && n_tiles % 8u == 0) {
    tiles_per_xcd = n_tiles / 8u;
    blocks_per_xcd = (tiles_per_xcd * n_qw + kWavesPerBlock - 1) / kWavesPerBlock;
  }
  const uint32_t groups = n_tiles < (uint32_t)kMergeGroups ? n_tiles : (uint32_t)kMergeGroups;
  TOD_HIP(ctx->m_part.reserve((size_t)n_tiles * K * nq_pad * sizeof(uint32_t)));
  // one buffer, one memset: the per-query bound words (0xFFFFFFFF = none published) and the per-(tile, 64 queries) flag bytes
  const size_t bound_bytes = (size_t)nq_pad * sizeof(uint32_t), flag_bytes = (size_t)n_tiles * n_qw;
  TOD_HIP(ctx->m_bound.reserve(bound_bytes + flag_bytes));
  TOD_HIP(hipMemsetAsync(ctx->m_bound.p, 0xFF, bound_bytes + flag_bytes, ctx->stream));
  uint8_t* const d_stored = ctx->m_bound.as<uint8_t>() + bound_bytes;
  int slot = -1;
  if (ctx->time_kernels) { int rc = tod_timing_begin(ctx, &slot); if (rc != TODHIP_OK) return rc; }
  // every schedule is exact; TODHIP_K4_MODE=0/1/2/3 overrides the choice (diagnostics: tools/k4_on_correlated_descriptors.py)
  static const int env_mode = getenv("TODHIP_K4_MODE") ? atoi(getenv("TODHIP_K4_MODE")) : -1;
  const int mode = (env_mode >= 0 && env_mode <= 3 && cut <= 256u) ? env_mode
                                                                   : (cut <= 38u ? 2 : (cut <= 48u ? 1 : (cut <= 80u ? 3 : 0)));
  auto kern = mode == 2 ? hamming_topk_tiles<K, 2>
                        : (mode == 1 ? hamming_topk_tiles<K, 1> : (mode == 3 ? hamming_topk_tiles<K, 3> : hamming_topk_tiles<K, 0>));
  hipLaunchKernelGGL(kern, dim3(blocks_per_xcd * 8u), dim3(kBlock), 0, ctx->stream,
                     ctx->db_desc.as<uint32_t>(), d_q, n_rows, nq, nq_pad, rows_per_tile, n_tiles, n_qw,
                     blocks_per_xcd, tiles_per_xcd, cut, ctx->m_part.as<uint32_t>(), ctx->m_bound.as<uint32_t>(), d_stored);
  if (slot >= 0) { int rc = tod_timing_end(ctx, slot); if (rc != TODHIP_OK) return rc; }
  hipLaunchKernelGGL(merge_tiles_kernel<K>, dim3((nq + kBlock - 1) / kBlock, groups), dim3(kBlock), 0, ctx->stream,
                     ctx->m_part.as<uint32_t>(), nq, nq_pad, n_tiles, rows_per_tile, ctx->shard_first, groups,
                     d_stored, n_qw, d_lists);
  TOD_HIP(hipGetLastError());
  *n_lists = groups;
  return TODHIP_OK;
}

}  // namespace

int tod_timing_drain(todhip_ctx* ctx, uint64_t keep) {
  while (ctx->ev_head - ctx->ev_tail > keep) {
    const int i = (int)(ctx->ev_tail % todhip_ctx::kEvPairs);
    TOD_HIP(hipEventSynchronize(ctx->evp[2 * i + 1]));
    float ms = 0.f;
    TOD_HIP(hipEventElapsedTime(&ms, ctx->evp[2 * i], ctx->evp[2 * i + 1]));
    ctx->counters.last_match_kernel_ms = ms;
    ctx->counters.sum_match_kernel_ms += ms;
    ctx->counters.n_match_kernel_launches += 1;
    ++ctx->ev_tail;
  }
  return TODHIP_OK;
}
int tod_timing_begin(todhip_ctx* ctx, int* slot) {
  int rc = tod_timing_drain(ctx, todhip_ctx::kEvPairs - 1);
  if (rc != TODHIP_OK) return rc;
  *slot = (int)(ctx->ev_head % todhip_ctx::kEvPairs);
  TOD_HIP(hipEventRecord(ctx->evp[2 * *slot], ctx->stream));
  return TODHIP_OK;
}
int tod_timing_end(todhip_ctx* ctx, int slot) {
  TOD_HIP(hipEventRecord(ctx->evp[2 * slot + 1], ctx->stream));
  ++ctx->ev_head;
  return TODHIP_OK;
}

// Per-query candidate lists of this shard: d_lists[n_lists][nq][k] (each ascending). n_lists <= kMergeGroups.
int tod_match_lists(todhip_ctx* ctx, const void* d_q, uint32_t nq, uint32_t k, uint32_t radius, uint64_t* d_lists,
                    uint32_t* n_lists) {
  if (ctx->desc_bytes != 32) return TODHIP_EINVAL;
  if (tod_lsh_enabled(ctx)) return tod_lsh_lists(ctx, d_q, nq, k, d_lists, n_lists);   // todhip_set_lsh: candidates from the index only
  if (ctx->ratio > 0.f) radius = 256u;   // the ratio test needs the true second neighbour, however far: no radius bound in the search
  const uint32_t* q = reinterpret_cast<const uint32_t*>(d_q);
  switch (k) {
    case 1: return launch_topk<1>(ctx, q, nq, radius, d_lists, n_lists);
    case 2: return launch_topk<2>(ctx, q, nq, radius, d_lists, n_lists);
    case 3: return launch_topk<3>(ctx, q, nq, radius, d_lists, n_lists);
    case 4: return launch_topk<4>(ctx, q, nq, radius, d_lists, n_lists);
    case 5: return launch_topk<5>(ctx, q, nq, radius, d_lists, n_lists);
    case 6: return launch_topk<6>(ctx, q, nq, radius, d_lists, n_lists);
    case 7: return launch_topk<7>(ctx, q, nq, radius, d_lists, n_lists);
    case 8: return launch_topk<8>(ctx, q, nq, radius, d_lists, n_lists);
    default: return TODHIP_EINVAL;
  }
}

size_t tod_match_lists_bytes(uint32_t nq, uint32_t k) { return (size_t)kMergeGroups * nq * k * sizeof(uint64_t); }

int tod_match_shard_keys(todhip_ctx* ctx, const void* d_q, uint32_t nq, uint32_t k, uint32_t radius, uint64_t* d_keys) {
  if (nq == 0) return TODHIP_OK;
  if (ctx->shard_rows == 0) {   // an empty shard contributes only padding keys
    TOD_HIP(hipMemsetAsync(d_keys, 0xFF, (size_t)nq * k * sizeof(uint64_t), ctx->stream));
    return TODHIP_OK;
  }
  TOD_HIP(ctx->m_keys.reserve(tod_match_lists_bytes(nq, k)));
  uint32_t n_lists = 0;
  int rc = tod_match_lists(ctx, d_q, nq, k, radius, ctx->m_keys.as<uint64_t>(), &n_lists);
  if (rc != TODHIP_OK) return rc;
  hipLaunchKernelGGL(select_keys_kernel, dim3((nq + kBlock - 1) / kBlock), dim3(kBlock), 0, ctx->stream,
                     ctx->m_keys.as<uint64_t>(), n_lists, nq, k, d_keys);
  TOD_HIP(hipGetLastError());
  return TODHIP_OK;
}

int tod_match_finalize(todhip_ctx* ctx, const uint64_t* d_keys_all, uint32_t n_shards, uint32_t nq, uint32_t k_in, uint32_t k_out,
                       uint32_t radius, uint32_t* d_counts, todhip_dmatch* d_matches, float* d_xyz, hipStream_t stream) {
  if (nq == 0) return TODHIP_OK;
  const uint32_t blocks = (nq + kBlock - 1) / kBlock;
  hipLaunchKernelGGL(finalize_kernel, dim3(blocks), dim3(kBlock), 0, stream ? stream : ctx->stream, d_keys_all, n_shards, nq, k_in, k_out,
                     radius, ctx->ratio, ctx->db_obj_off.as<uint32_t>(), ctx->n_objs, ctx->db_pts.as<float>(), d_counts, d_matches,
                     d_xyz);
  TOD_HIP(hipGetLastError());
  return TODHIP_OK;
}

#ifdef TOD_K4X_COUNT_WALKS
// diagnostics build only: {32 x 32 blocks tested, blocks whose best dot product beat a threshold (16 rows walked)} since the last reset
extern "C" int todhip_debug_k4x_walks(unsigned long long out[2], int reset) {
  unsigned long long zero[2] = {0ull, 0ull};
  if (hipDeviceSynchronize() != hipSuccess) return TODHIP_EHIP;
  if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(g_k4x_blocks), sizeof(zero)) != hipSuccess) return TODHIP_EHIP;
  if (reset && hipMemcpyToSymbol(HIP_SYMBOL(g_k4x_blocks), zero, sizeof(zero)) != hipSuccess) return TODHIP_EHIP;
  return TODHIP_OK;
}
#endif
