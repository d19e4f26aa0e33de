// Float-descriptor matcher on gfx950 (BASELINE.json configs[3]: "SIFT-128 float descriptors, L2 brute force as an
// MFMA bf16 GEMM"). Not a reference feature: DescriptorMatcher serves binary descriptors through FLANN-LSH only and
// throws for other index types (src/detection/DescriptorMatcher.cpp:154-188). What is kept from the reference is the
// result shape of DescriptorMatcher::process (:195-252): k nearest rows per query over the concatenated DB, truncated
// at the first distance > radius (:212-220), (imgIdx, trainIdx) by object prefix sums, the 3D point of each match.
//
// Exact result from an approximate GEMM -- filter and refine:
//   score(q, r) = |r|^2 - 2 <q^, r^>          q^, r^ = bf16 roundings, product on the matrix cores (f32 accumulate)
//   |score + |q|^2 - d2(q, r)| <= eps(q)      eps(q) = 2^-7 (1 + 2^-6) |q| Rmax + 2^-14 (|q| + Rmax)^2, Rmax = max |r|
//   (bf16 rounding moves each vector by <= 2^-9 of its norm, Cauchy-Schwarz on the two error terms, the f32
//   accumulation of 128 products and of the norms is below 2^-16 of |q||r|; the second term covers the rounding
//   of the sequential f32 sum that DEFINES d2, see include/todhip.h).
//   pass 0  seed(q) >= A_k from a sample of the DB                  (the k-th smallest of per-partition minima, or of per-lane lists)
//   pass 1  A_k(q) = k-th smallest score over the DB                (GEMM + per-lane top-8 in registers; only scores
//                                                                    below the seed are ever inserted)
//   pass 2  candidates = { r : score <= A_k + 2 eps }                (same GEMM; a lane's candidates go to its private slot)
//           every row of the true top-k is a candidate: its d2 <= T_k <= A_k + |q|^2 + eps
//   DBs of >= 64k rows skip pass 1: the sample is an evenly spaced quarter of the DB (at most 64k rows) and seed + 2 eps
//   is used as the threshold of pass 2 -- seed >= A_k, so the candidates are a superset (rows of rank <= ~k N / n_sample
//   plus the eps band) and pass 3 returns the same result from it
//   pass 3  exact d2 of the candidates in the defining order (sequential f32, no fma), k smallest by (d2, row);
//           a query whose candidate lists overflowed is redone by an exact scan of the whole DB
// L2G  l2_gemm_kernel<PASS, KT>  wave = (DB chunk, 4 query tiles of 32): the queries are 128 VGPRs of B fragments; the DB is stored
//                              in A-fragment order and streams through a per-wave two-slot ring in LDS that global_load_lds fills
//                              two tiles ahead (no barrier anywhere); per 32 DB rows and query tile 8 v_mfma_f32_32x32x16_bf16
//                              with the row norms as the C operand and the queries pre-scaled by -2 (exact in bf16), so the
//                              accumulator IS the score
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "ctx.h"

namespace {

constexpr uint32_t kDim = 128;
constexpr uint32_t kTileRows = 32;                        // DB rows of one MFMA A tile (8 KB of bf16)
constexpr uint32_t kRingSlotBytes = kTileRows * kDim * 2u + 256u;   // a wave's ring slot in LDS: one tile's fragments + its 32 norms, twice
constexpr uint32_t kQTilesPerWave = 4;
constexpr uint32_t kWaves = 4;
constexpr uint32_t kBlockQueries = kWaves * kQTilesPerWave * 32u;   // 512
constexpr uint32_t kTop = 8;                              // per-lane list length of pass 1 (k <= 8)
constexpr uint32_t kSlot = 8;                             // candidate rows a (query, chunk, lane half) keeps in its private slot
constexpr uint32_t kCandCap = 1024;                       // shared list per query for what the slots cannot hold; beyond it the exact scan
constexpr uint32_t kListCap = 2048;                       // candidates per query all told (pass 3's list in LDS); beyond it the exact scan

typedef __attribute__((ext_vector_type(8))) short bf16x8;  // 8 bf16 = 4 VGPRs: one MFMA A/B fragment
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

struct L2Ws {
  DevBuf db_bf16, db_norm, q_bf16, q_eps, part, thr, slots, slot_cnt, cand, cand_cnt, keys, scal;
  uint32_t n_pad = 0;
};

L2Ws* l2ws_of(todhip_ctx* ctx) {
  if (!ctx->l2_ws) ctx->l2_ws = new L2Ws();
  return reinterpret_cast<L2Ws*>(ctx->l2_ws);
}

__device__ __forceinline__ uint16_t bf16_rne(float x) {
  const uint32_t b = __float_as_uint(x);
  return (uint16_t)((b + 0x7FFFu + ((b >> 16) & 1u)) >> 16);
}

// one wave per row: bf16 image (scaled by `scale`, a power of two) and |row|^2; rows >= n are padding
// frag != 0 (the DB image): a 32-row tile is stored in MFMA A-fragment order -- 8 segments (k-steps s) of 64 lanes x 16 bytes,
// lane (h = l >> 5, r = l & 31) of segment s holding A[row r][k = 16 s + 8 h + j], j = 0..7 -- so that one wave-load of a
// segment is 1 KB contiguous AND already the register image v_mfma_f32_32x32x16_bf16 wants: no LDS staging, no swizzle
// query side (rmax2_bits != nullptr): eps(q) from |q|^2 and Rmax^2 is written too, and the query's shared candidate counter is zeroed
// (two launches less per call)
__global__ __launch_bounds__(256) void l2_prepare_kernel(const float* __restrict__ src, uint32_t n, uint32_t n_pad, float scale,
                                                         uint16_t* __restrict__ dst, float* __restrict__ norm2,
                                                         uint32_t* __restrict__ max_norm2_bits, int frag,
                                                         const uint32_t* __restrict__ rmax2_bits = nullptr, float* __restrict__ eps = nullptr,
                                                         uint32_t* __restrict__ zero_cnt = nullptr) {
  const uint32_t row = blockIdx.x * 4u + (threadIdx.x >> 6), l = threadIdx.x & 63u;
  if (row >= n_pad) return;
  float a = 0.f, b = 0.f;
  if (row < n) { a = src[(size_t)row * kDim + 2u * l]; b = src[(size_t)row * kDim + 2u * l + 1u]; }
  const uint32_t k0 = 2u * l;                               // this lane's two dimensions k0, k0 + 1
  const size_t at = frag ? ((size_t)(row >> 5) * (kTileRows * kDim) + (k0 >> 4) * 512u + (((k0 >> 3) & 1u) * 32u + (row & 31u)) * 8u + (k0 & 7u)) / 2u
                         : (size_t)row * (kDim / 2u) + l;
  reinterpret_cast<uint32_t*>(dst)[at] =
      (uint32_t)bf16_rne(a * scale) | ((uint32_t)bf16_rne(b * scale) << 16);
  float s = a * a + b * b;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
  if (l == 0) {
    const float n2 = row < n ? s : 3.0e38f;                 // a padding row can never be a candidate
    norm2[row] = n2;
    if (row < n && max_norm2_bits) atomicMax(max_norm2_bits, __float_as_uint(s));
    if (rmax2_bits) {
      const float nq = sqrtf(n2), rmax = sqrtf(__uint_as_float(*rmax2_bits));
      eps[row] = 0.0079345703125f * nq * rmax + 6.103515625e-05f * (nq + rmax) * (nq + rmax);   // 2^-7 (1 + 2^-6), 2^-14
      zero_cnt[row] = 0u;
    }
  }
}

// Where pass 2 puts a lane's candidates. A lane is the only writer for (query, chunk, lane half): its first kSlot rows go to a
// private slot with plain stores and a count in a register -- nothing to wait for. (An atomic append to one list per query, the
// round-1 form, costs a round trip to L2 per candidate with every outstanding load drained first, and with some hundreds
// of candidates per query at the eps band of bf16 that was most of the pass.) Rows beyond kSlot go to the shared list.
struct CandSink {
  uint32_t* __restrict__ slots;        // [nq_pad][n_parts][kSlot]
  uint32_t* __restrict__ shared;       // [nq_pad][kCandCap]
  uint32_t* __restrict__ shared_cnt;   // [nq_pad]
  uint32_t n_parts, part;              // this lane's partition = 2 chunk + h
};

// What becomes of the 16 scores a lane holds for query q (register i <-> DB row row0 + (i & 3) + 8 (i >> 2), row0 including the
// lane half's 4 h).
//   pass 1, KT > 1: the KT smallest below `limit` in `best` (ascending), and the limit follows best[KT - 1]
//   pass 1, KT = 1: the minimum only -- one instruction behind the min tree, no branch (the seed pass: the k-th smallest of
//                   the partitions' minima is the score of k distinct rows, hence an upper bound of A_k like the exact k-th)
//   pass 2: rows with score <= limit are appended to the lane's slot (`cnt` rows so far)
template <int PASS, uint32_t KT>
__device__ __forceinline__ void l2_epilogue(const f32x16& sc, float& limit, float (&best)[KT], uint32_t& cnt, uint32_t q, uint32_t row0,
                                            const CandSink& sink) {
  if (PASS == 2) {
    // two levels, so that a tile with one candidate among its 64 x 16 scores (the usual case when there is any) pays four
    // group tests and four leaf tests rather than sixteen: minima of the four register groups (= row groups 8 g + 4 h ..+3)
    float gm[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) gm[g] = fminf(fminf(fminf(sc[4 * g], sc[4 * g + 1]), sc[4 * g + 2]), sc[4 * g + 3]);
    const float m = fminf(fminf(fminf(gm[0], gm[1]), gm[2]), gm[3]);
    if (__builtin_amdgcn_ballot_w64(m <= limit) != 0ull) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        if (__builtin_amdgcn_ballot_w64(gm[g] <= limit) != 0ull) {   // wave-uniform
          asm volatile("");                                           // a scalar branch of its own, not folded into the lanes' tests below
#pragma unroll
          for (int i = 4 * g; i < 4 * g + 4; ++i) {
            if (sc[i] <= limit) {
              uint32_t base = row0;
              asm volatile("" : "+v"(base));                    // keeps the 16 row numbers from being hoisted into 16 registers of the main path
              const uint32_t row = base + (uint32_t)(i & 3) + 8u * (uint32_t)(i >> 2);
              if (cnt < kSlot) {
                sink.slots[((size_t)q * sink.n_parts + sink.part) * kSlot + cnt] = row;
              } else {
                const uint32_t at = atomicAdd(&sink.shared_cnt[q], 1u);
                if (at < kCandCap) sink.shared[(size_t)q * kCandCap + at] = row;
              }
              ++cnt;
            }
          }
        }
      }
    }
    return;
  }
  float m = fminf(fminf(sc[0], sc[1]), sc[2]);
#pragma unroll
  for (int i = 3; i < 15; i += 2) m = fminf(fminf(m, sc[i]), sc[i + 1]);
  m = fminf(m, sc[15]);
  if (KT == 1) {
    best[0] = fminf(best[0], m);
  } else {
    if (__builtin_amdgcn_ballot_w64(m < limit) != 0ull) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        if (__builtin_amdgcn_ballot_w64(sc[i] < limit) != 0ull) {    // wave-uniform: usually one value of the 16
          float v = sc[i] < limit ? sc[i] : FLT_MAX;                  // FLT_MAX falls through the list unchanged
#pragma unroll
          for (uint32_t j = 0; j < KT; ++j) { const float lo = fminf(best[j], v); v = fmaxf(best[j], v); best[j] = lo; }
        }
      }
      limit = fminf(limit, best[KT - 1]);
    }
  }
}

// L2G. One WAVE = (DB chunk, 128 queries): 4 query tiles of 32 as B fragments in registers (128 VGPRs); the DB streams through
// in tiles of 32 rows that are stored in A-fragment order (l2_prepare_kernel), so a tile is 8 contiguous 1 KB wave-loads whose
// lane l part is exactly what lane l feeds the MFMA -- LDS is only a FIFO between the load and the register, never a
// transpose. No barrier, no dependence between the waves of a block (they load the same tiles at about the same time, so
// three of four loads hit the CU's L1). The row norms enter as the C operand of a tile's first MFMA, so the accumulator IS
// the score; the epilogue of query tile t sits behind the MFMAs of tile t + 1 (two accumulators).
// KT: per-lane list length of pass 1 (1: minimum only, the seed pass; 4 when k <= 4: 16 VGPRs less), unused in pass 2
template <int PASS, uint32_t KT>
__global__ __launch_bounds__(256, 2) void l2_gemm_kernel(const uint16_t* __restrict__ db, const float* __restrict__ dbn,
                                                         uint32_t n_tiles, uint32_t tiles_per_chunk, uint32_t tile_stride,
                                                         const uint16_t* __restrict__ qh, uint32_t nq_pad,
                                                         float* __restrict__ part, const float* __restrict__ thr,
                                                         uint32_t* __restrict__ slots, uint32_t* __restrict__ slot_cnt,
                                                         uint32_t* __restrict__ cand, uint32_t* __restrict__ cand_cnt) {
  // tile_stride: tile t of the launch is DB tile t * tile_stride (the seed pass samples the DB evenly; 1 otherwise)
  // thr: PASS 1 -- optional per-query seed (only scores below it can matter), PASS 2 -- the candidate threshold
  const uint32_t tid = threadIdx.x, wave = tid >> 6, l = tid & 63u, r = l & 31u, h = l >> 5;
  const uint32_t chunk = blockIdx.x;
  const uint32_t q_base = blockIdx.y * kBlockQueries + wave * (kQTilesPerWave * 32u);
  const uint32_t t_begin = chunk * tiles_per_chunk, t_end = min(n_tiles, t_begin + tiles_per_chunk);
  if (t_begin >= t_end) return;
  const uint64_t t_start = (PASS == 2 && part) ? __builtin_amdgcn_s_memrealtime() : 0ull;

  // this wave's query fragments: B[k = 16 s + 8 h + j][col r] = q^[q_base + 32 t + r][16 s + 8 h + j]
  bf16x8 bq[kQTilesPerWave][8];
#pragma unroll
  for (uint32_t t = 0; t < kQTilesPerWave; ++t) {
    const uint16_t* qp = qh + (size_t)(q_base + 32u * t + r) * kDim + 8u * h;
#pragma unroll
    for (uint32_t s = 0; s < 8; ++s) bq[t][s] = *reinterpret_cast<const bf16x8*>(qp + 16u * s);
  }
  float best[kQTilesPerWave][KT];
  float limit[kQTilesPerWave];
  uint32_t n_cand[kQTilesPerWave];                          // pass 2: rows this lane has appended per query
#pragma unroll
  for (uint32_t t = 0; t < kQTilesPerWave; ++t) {
#pragma unroll
    for (uint32_t j = 0; j < KT; ++j) best[t][j] = FLT_MAX;
    limit[t] = (PASS == 1 && !thr) ? FLT_MAX : thr[q_base + 32u * t + r];
    n_cand[t] = 0u;
  }
  const CandSink sink{slots, cand, cand_cnt, 2u * gridDim.x, 2u * chunk + h};
  // everything loaded so far is waited for HERE: the compiler does not see the ring's loads below, and a wait of its own for one
  // of these, placed lazily inside the loop, would be counted without them and drain the ring
#pragma unroll
  for (uint32_t t = 0; t < kQTilesPerWave; ++t) {
#pragma unroll
    for (uint32_t s = 0; s < 8; ++s) asm volatile("" :: "v"(bq[t][s]));
    asm volatile("" :: "v"(limit[t]));
  }
  // The wave's private ring in LDS: two slots of one DB tile (8 KB of fragments, already in register order) + its 32 row norms
  // (twice: the 64 lanes of the load bring them as lanes l & 31). A slot is filled by 9 loads that write LDS directly
  // (global_load_lds: no destination registers, so their depth costs nothing) and read back with ds_read_b128 -- lane l
  // gets exactly the 16 bytes lane l loaded. Written as asm, with the vmcnt waits counted by hand: loads retire in order, each
  // tile is exactly 9 of them, and "at most 9 outstanding" therefore means the older of the two tiles in flight has landed
  // (the compiler would wait for everything before any LDS read, which is the depth this ring exists for).
  __shared__ __attribute__((aligned(16))) uint8_t ring[kWaves][2][kRingSlotBytes];
  typedef __attribute__((address_space(3))) void* lptr_t;
  const uint32_t ring_lds = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(size_t)(lptr_t)(&ring[wave][0][0]));
  const uint32_t frag_off = l * 16u, norm_off = r * 4u;
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
#define L2_DMA(width_, voff_, gaddr_, lds_)                                                                     \
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_" width_ " %0, %1" :: "v"(voff_), "s"(gaddr_), "s"(lds_) : "memory", "m0")
  auto issue = [&](uint32_t tile, uint32_t slot) {
    tile = min(tile, t_end - 1u);                           // past the chunk: a harmless reload, the count per tile stays 9
    // (the addresses are wave-uniform; readfirstlane says so in a way the "s" constraints below can rely on)
    auto uniform64 = [](uint64_t v) {
      return ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32)) << 32) |
             (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
    };
    const uint64_t g = uniform64((uint64_t)(db + (size_t)tile * tile_stride * (kTileRows * kDim)));
    const uint64_t gn = uniform64((uint64_t)(dbn + (size_t)tile * tile_stride * kTileRows));
    const uint32_t lds = ring_lds + slot * kRingSlotBytes;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the slot's previous tile has been read out
#pragma unroll
    for (uint32_t s = 0; s < 8; ++s) L2_DMA("dwordx4", frag_off, g + s * 1024u, lds + s * 1024u);
    L2_DMA("dword", norm_off, gn, lds + 8192u);
  };
  // One DB tile against the wave's 4 query tiles. The tile's fragments live in ONE register set: the last query tile's
  // MFMA of k-step s is the last reader of a[s], and the next tile's a[s] is read from the ring right behind it (a second
  // set would not fit beside the 128 query registers at two waves per SIMD; LDS latency is what that distance covers).
  bf16x8 a[8];
  // |row|^2 of the rows this lane's accumulators belong to (register 4 g + i <-> row 8 g + 4 h + i): the C operand of a query
  // tile's first MFMA. Like a[], ONE register set, reloaded behind its last reader.
  f32x16 nrm;
  auto read_frag = [&](uint32_t slot, uint32_t s) { return *reinterpret_cast<const bf16x8*>(&ring[wave][slot][s * 1024u + frag_off]); };
  auto read_norms = [&](uint32_t slot) {
    const float* np = reinterpret_cast<const float*>(&ring[wave][slot][8192u]) + 4u * h;
#pragma unroll
    for (uint32_t gg = 0; gg < 4; ++gg) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(np + 8u * gg);
      nrm[4 * gg + 0] = v[0]; nrm[4 * gg + 1] = v[1]; nrm[4 * gg + 2] = v[2]; nrm[4 * gg + 3] = v[3];
    }
  };
  issue(t_begin, 0u);
  issue(t_begin + 1u, 1u);
  asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
#pragma unroll
  for (uint32_t s = 0; s < 8; ++s) a[s] = read_frag(0u, s);
  read_norms(0u);
  uint32_t cur = 0;
  for (uint32_t tile = t_begin; tile < t_end; ++tile, cur ^= 1u) {
#if !defined(TOD_L2_ABLATE) || (TOD_L2_ABLATE != 1 && TOD_L2_ABLATE != 3)   // diagnostics builds (tools/l2_ablate.sh): 1 no DMA, 2 no LDS reads, 3 neither
    issue(tile + 2u, cur);                                  // in flight now: tile + 1 (the other slot) and tile + 2
#endif
    f32x16 acc[2];
    auto mfma_tile = [&](uint32_t t) {
#if !defined(TOD_L2_ABLATE) || (TOD_L2_ABLATE != 1 && TOD_L2_ABLATE != 3)
      if (t + 1 == kQTilesPerWave) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");     // tile + 1 has landed
#endif
      f32x16 c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], bq[t][0], nrm, 0, 0, 0);
#if !defined(TOD_L2_ABLATE) || TOD_L2_ABLATE < 2
      if (t + 1 == kQTilesPerWave) { a[0] = read_frag(cur ^ 1u, 0u); read_norms(cur ^ 1u); }
#endif
#pragma unroll
      for (uint32_t s = 1; s < 8; ++s) {
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[s], bq[t][s], c, 0, 0, 0);
#if !defined(TOD_L2_ABLATE) || TOD_L2_ABLATE < 2
        if (t + 1 == kQTilesPerWave) a[s] = read_frag(cur ^ 1u, s);
#endif
      }
      return c;
    };
    acc[0] = mfma_tile(0);
#pragma unroll
    for (uint32_t t = 0; t < kQTilesPerWave; ++t) {
      if (t + 1 < kQTilesPerWave) acc[(t + 1) & 1u] = mfma_tile(t + 1);
      l2_epilogue<PASS, KT>(acc[t & 1u], limit[t], best[t], n_cand[t], q_base + 32u * t + r, tile * tile_stride * kTileRows + 4u * h, sink);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // nothing may still be on its way into this block's LDS when it is handed on
#undef L2_DMA
#pragma clang diagnostic pop
  if (PASS == 2) {
#pragma unroll
    for (uint32_t t = 0; t < kQTilesPerWave; ++t) slot_cnt[(size_t)(q_base + 32u * t + r) * sink.n_parts + sink.part] = n_cand[t];
    if (part && l == 0) {                                   // diagnostics (TODHIP_L2_TRACE): when each wave started and ended, 100 MHz ticks
      uint64_t* tr = reinterpret_cast<uint64_t*>(part) + 2u * (((size_t)blockIdx.y * gridDim.x + chunk) * kWaves + wave);
      tr[0] = t_start; tr[1] = __builtin_amdgcn_s_memrealtime();
    }
  }
  if (PASS == 1) {
    // partition (chunk, lane half): kTop ascending scores per query
#pragma unroll
    for (uint32_t t = 0; t < kQTilesPerWave; ++t) {
      float* dst = part + ((size_t)(chunk * 2u + h) * nq_pad + (q_base + 32u * t + r)) * kTop;
#pragma unroll
      for (uint32_t j = 0; j < kTop; ++j) dst[j] = j < KT ? best[t][j] : FLT_MAX;
    }
  }
}

// per query (one wave): k-th smallest score over the partitions' lists; scores at or above the seed were never
// recorded, so fewer than k entries mean A_k == seed. out = A_k + margin * eps
__global__ __launch_bounds__(256) void l2_threshold_kernel(const float* __restrict__ part, uint32_t n_parts, uint32_t nq,
                                                           uint32_t nq_pad, uint32_t k, const float* __restrict__ seed,
                                                           const float* __restrict__ eps, float margin, float* __restrict__ out) {
  const uint32_t q = blockIdx.x * 4u + (threadIdx.x >> 6), l = threadIdx.x & 63u;
  if (q >= nq_pad) return;
  if (q >= nq) { if (l == 0) out[q] = -FLT_MAX; return; }   // padding queries collect nothing
  float best[kTop];
#pragma unroll
  for (uint32_t j = 0; j < kTop; ++j) best[j] = FLT_MAX;
  for (uint32_t p = l; p < n_parts; p += 64u) {
    const float* src = part + ((size_t)p * nq_pad + q) * kTop;
    for (uint32_t i = 0; i < kTop; ++i) {
      float v = src[i];
      if (v >= best[kTop - 1]) break;                       // lists are ascending
#pragma unroll
      for (uint32_t j = 0; j < kTop; ++j) { const float lo = fminf(best[j], v); v = fmaxf(best[j], v); best[j] = lo; }
    }
  }
  // k rounds of wave-min over the lanes' heads (values may repeat: pop one lane per round)
  float ak = FLT_MAX;
  for (uint32_t j = 0; j < k; ++j) {
    float v = best[0];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fminf(v, __shfl_xor(v, off));
    ak = v;
    const unsigned long long owners = __builtin_amdgcn_ballot_w64(best[0] == v);
    if (v != FLT_MAX && l == (uint32_t)__ffsll((long long)owners) - 1u) {
#pragma unroll
      for (uint32_t i = 0; i + 1 < kTop; ++i) best[i] = best[i + 1];
      best[kTop - 1] = FLT_MAX;
    }
  }
  if (seed) ak = fminf(ak, seed[q]);
  if (l == 0) out[q] = ak + margin * eps[q];
}

// the defining distance (include/todhip.h): sequential f32, no fused multiply-add (-ffp-contract=off)
__device__ __forceinline__ float d2_exact(const float* __restrict__ q, const float* __restrict__ r) {
  float acc = 0.f;
  for (uint32_t i = 0; i < kDim; i += 4) {
    const float4 a = *reinterpret_cast<const float4*>(q + i), b = *reinterpret_cast<const float4*>(r + i);
    float t = a.x - b.x; acc = acc + t * t;
    t = a.y - b.y; acc = acc + t * t;
    t = a.z - b.z; acc = acc + t * t;
    t = a.w - b.w; acc = acc + t * t;
  }
  return acc;
}

__device__ __forceinline__ void keys_insert(uint64_t (&best)[kTop], uint64_t key) {
#pragma unroll
  for (uint32_t j = 0; j < kTop; ++j) { const uint64_t lo = key < best[j] ? key : best[j]; key = key < best[j] ? best[j] : key; best[j] = lo; }
}

// the k smallest keys of the wave's lanes' ascending lists -> out[0..k) (lane 0 writes)
__device__ __forceinline__ void wave_select(uint64_t (&best)[kTop], uint32_t k, uint64_t* out) {
  const uint32_t l = threadIdx.x & 63u;
  for (uint32_t j = 0; j < k; ++j) {
    uint64_t v = best[0];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      const uint64_t o = ((uint64_t)__shfl_xor((uint32_t)(v >> 32), off) << 32) | __shfl_xor((uint32_t)v, off);
      v = o < v ? o : v;
    }
    if (l == 0) out[j] = v;
    if (best[0] == v && v != ~0ull) {                        // keys are unique: exactly one lane pops
#pragma unroll
      for (uint32_t i = 0; i + 1 < kTop; ++i) best[i] = best[i + 1];
      best[kTop - 1] = ~0ull;
    }
  }
}

// pass 3: one wave per query. keys[q][k] = (f32 bits of d2) << 32 | row, ascending, ~0 padded
__global__ __launch_bounds__(256) void l2_rerank_kernel(const float* __restrict__ q, uint32_t nq, const float* __restrict__ db,
                                                        const uint32_t* __restrict__ slots, const uint32_t* __restrict__ slot_cnt,
                                                        uint32_t n_parts, const uint32_t* __restrict__ cand,
                                                        const uint32_t* __restrict__ cand_cnt, uint32_t k, uint64_t* __restrict__ keys,
                                                        uint32_t* __restrict__ overflow) {
  const uint32_t qi = blockIdx.x * 4u + (threadIdx.x >> 6), l = threadIdx.x & 63u;
  if (qi >= nq) return;
  const uint32_t cnt = cand_cnt[qi], w = threadIdx.x >> 6;
  // the query's candidate rows, gathered into one list in LDS first (the slots of 64 partitions at a time: counts, a wave
  // prefix sum, 32 bytes of rows per lane), so that the exact distances below are spread evenly over the lanes
  __shared__ uint32_t s_rows[4][kListCap];
  uint32_t total = 0;
  for (uint32_t p0 = 0; p0 < n_parts; p0 += 64u) {
    const uint32_t p = p0 + l;
    const uint32_t c = p < n_parts ? min(slot_cnt[(size_t)qi * n_parts + p], kSlot) : 0u;
    uint32_t incl = c;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { const uint32_t t = __shfl_up(incl, off); if (l >= (uint32_t)off) incl += t; }
    const uint32_t at = total + incl - c;
    if (c) {
      const uint4* src = reinterpret_cast<const uint4*>(slots + ((size_t)qi * n_parts + p) * kSlot);
      const uint4 lo = src[0], hi = src[1];
      const uint32_t rows[kSlot] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
#pragma unroll
      for (uint32_t j = 0; j < kSlot; ++j) if (j < c && at + j < kListCap) s_rows[w][at + j] = rows[j];
    }
    total += __shfl(incl, 63);
  }
  for (uint32_t c = l; c < min(cnt, kCandCap); c += 64u)    // what did not fit a slot
    if (total + c < kListCap) s_rows[w][total + c] = cand[(size_t)qi * kCandCap + c];
  total += cnt;
  if (cnt > kCandCap || total > kListCap) { if (l == 0) overflow[qi] = 1u; return; }   // redone by l2_exact_scan_kernel
  if (l == 0) overflow[qi] = 0u;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  uint64_t best[kTop];
#pragma unroll
  for (uint32_t j = 0; j < kTop; ++j) best[j] = ~0ull;
  for (uint32_t c = l; c < total; c += 64u) {
    const uint32_t row = s_rows[w][c];
    const float d2 = d2_exact(q + (size_t)qi * kDim, db + (size_t)row * kDim);
    keys_insert(best, ((uint64_t)__float_as_uint(d2) << 32) | row);
  }
  wave_select(best, k, keys + (size_t)qi * k);
}

// fallback for queries whose candidate list overflowed (and the test hook's exact GPU reference): one block per
// query scans every DB row with the defining distance
__global__ __launch_bounds__(256) void l2_exact_scan_kernel(const float* __restrict__ q, uint32_t nq, const float* __restrict__ db,
                                                            uint32_t n, uint32_t k, const uint32_t* __restrict__ only_flagged,
                                                            uint64_t* __restrict__ keys) {
  __shared__ uint64_t s_keys[4][kTop];
  const uint32_t qi = blockIdx.x, tid = threadIdx.x, l = tid & 63u, w = tid >> 6;
  if (qi >= nq || (only_flagged && only_flagged[qi] == 0u)) return;
  uint64_t best[kTop];
#pragma unroll
  for (uint32_t j = 0; j < kTop; ++j) best[j] = ~0ull;
  for (uint32_t row = tid; row < n; row += 256u) {
    const float d2 = d2_exact(q + (size_t)qi * kDim, db + (size_t)row * kDim);
    keys_insert(best, ((uint64_t)__float_as_uint(d2) << 32) | row);
  }
  wave_select(best, kTop, s_keys[w]);
  __syncthreads();
  if (w == 0) {
#pragma unroll
    for (uint32_t j = 0; j < kTop; ++j) best[j] = ~0ull;
    if (l < 4u * kTop) keys_insert(best, s_keys[l / kTop][l % kTop]);
    wave_select(best, k, keys + (size_t)qi * k);
  }
}

// radius cut (DescriptorMatcher.cpp:212-220 shape), object lookup (:60-129 load order), 3D gather (:231-244)
__global__ __launch_bounds__(256) void l2_finalize_kernel(const uint64_t* __restrict__ keys, uint32_t nq, uint32_t k, float radius,
                                                          const uint32_t* __restrict__ obj_off, uint32_t n_objs,
                                                          const float* __restrict__ pts, uint32_t* __restrict__ counts,
                                                          todhip_dmatch* __restrict__ matches, float* __restrict__ xyz) {
  const uint32_t qi = blockIdx.x * 256u + threadIdx.x;
  if (qi >= nq) return;
  uint32_t kept = 0;
  for (uint32_t j = 0; j < k; ++j) {
    const uint64_t key = keys[(size_t)qi * k + j];
    if (key == ~0ull) break;
    const float dist = sqrtf(__uint_as_float((uint32_t)(key >> 32)));
    if (dist > radius) break;
    const uint32_t row = (uint32_t)key;
    uint32_t lo = 0, hi = n_objs;                           // last object whose first row is <= row
    while (hi - lo > 1u) { const uint32_t mid = (lo + hi) >> 1; if (obj_off[mid] <= row) lo = mid; else hi = mid; }
    todhip_dmatch m;
    m.queryIdx = (int)qi; m.trainIdx = (int)(row - obj_off[lo]); m.imgIdx = (int)lo; m.distance = dist;
    matches[(size_t)qi * k + kept] = m;
    float* o = xyz + ((size_t)qi * k + kept) * 3;
    o[0] = pts[(size_t)row * 3 + 0]; o[1] = pts[(size_t)row * 3 + 1]; o[2] = pts[(size_t)row * 3 + 2];
    ++kept;
  }
  counts[qi] = kept;
}

}  // namespace

void tod_l2_ws_free(todhip_ctx* ctx) {
  if (!ctx->l2_ws) return;
  L2Ws* ws = reinterpret_cast<L2Ws*>(ctx->l2_ws);
  DevBuf* bufs[] = {&ws->db_bf16, &ws->db_norm, &ws->q_bf16, &ws->q_eps, &ws->part, &ws->thr, &ws->slots, &ws->slot_cnt, &ws->cand, &ws->cand_cnt, &ws->keys,
                    &ws->scal};
  for (DevBuf* b : bufs) b->release();
  delete ws;
  ctx->l2_ws = nullptr;
}

// called by todhip_db_load for 128 x f32 rows (already resident in ctx->db_desc): bf16 image, norms, Rmax
int tod_l2_db_prepare(todhip_ctx* ctx) {
  L2Ws* ws = l2ws_of(ctx);
  const uint32_t n = (uint32_t)ctx->shard_rows;
  ws->n_pad = ((n + kTileRows - 1u) / kTileRows) * kTileRows;
  TOD_HIP(ws->db_bf16.reserve((size_t)std::max(ws->n_pad, kTileRows) * kDim * 2));
  TOD_HIP(ws->db_norm.reserve((size_t)std::max(ws->n_pad, kTileRows) * 4));
  TOD_HIP(ws->scal.reserve(64));
  TOD_HIP(hipMemsetAsync(ws->scal.p, 0, 64, ctx->stream));
  if (ws->n_pad)
    hipLaunchKernelGGL(l2_prepare_kernel, dim3((ws->n_pad + 3u) / 4u), dim3(256), 0, ctx->stream, ctx->db_desc.as<float>(), n,
                       ws->n_pad, 1.0f, ws->db_bf16.as<uint16_t>(), ws->db_norm.as<float>(), ws->scal.as<uint32_t>(), 1);
  TOD_HIP(hipGetLastError());
  return TODHIP_OK;
}

static int l2_keys(todhip_ctx* ctx, const float* d_q, uint32_t nq, uint32_t k, int exact_only) {
  L2Ws* ws = l2ws_of(ctx);
  hipStream_t st = ctx->stream;
  const uint32_t n = (uint32_t)ctx->shard_rows;
  TOD_HIP(ws->keys.reserve((size_t)nq * k * 8));
  TOD_HIP(ws->cand_cnt.reserve(((size_t)nq + kBlockQueries) * 4 * 2));
  uint32_t* overflow = ws->cand_cnt.as<uint32_t>() + nq + kBlockQueries;
  if (exact_only) {
    hipLaunchKernelGGL(l2_exact_scan_kernel, dim3(nq), dim3(256), 0, st, d_q, nq, ctx->db_desc.as<float>(), n, k,
                       (const uint32_t*)nullptr, ws->keys.as<uint64_t>());
    TOD_HIP(hipGetLastError());
    return TODHIP_OK;
  }
  const uint32_t nq_pad = ((nq + kBlockQueries - 1u) / kBlockQueries) * kBlockQueries;
  const uint32_t q_blocks = nq_pad / kBlockQueries;
  const uint32_t n_tiles = ws->n_pad / kTileRows;
  // 2 blocks per CU in flight; the DB is cut into as many chunks as that allows for this many query blocks
  static const uint32_t env_cm = getenv("TODHIP_L2_CHUNK_ROUNDS") ? (uint32_t)atoi(getenv("TODHIP_L2_CHUNK_ROUNDS")) : 0u;    // tuning knob
  uint32_t n_chunks = std::max(1u, (std::max(env_cm, 1u) * 2u * (uint32_t)ctx->n_cu) / q_blocks);
  n_chunks = std::min(n_chunks, std::max(1u, n_tiles));
  const uint32_t tiles_per_chunk = (n_tiles + n_chunks - 1u) / n_chunks;
  n_chunks = (n_tiles + tiles_per_chunk - 1u) / tiles_per_chunk;
  const uint32_t n_parts = 2u * n_chunks;
  TOD_HIP(ws->q_bf16.reserve((size_t)nq_pad * kDim * 2));
  TOD_HIP(ws->q_eps.reserve((size_t)nq_pad * 4 * 2));
  TOD_HIP(ws->part.reserve((size_t)std::max(n_parts, 64u) * nq_pad * kTop * 4));   // the seed pass writes up to 2 x 32 partitions
  TOD_HIP(ws->thr.reserve((size_t)nq_pad * 4 * 2));
  TOD_HIP(ws->cand.reserve((size_t)nq_pad * kCandCap * 4));
  TOD_HIP(ws->slots.reserve((size_t)nq_pad * n_parts * kSlot * 4));        // 2 x 512 partitions x 8 rows x 4 B x 512 queries per query block: 16 MB whatever nq
  TOD_HIP(ws->slot_cnt.reserve((size_t)nq_pad * n_parts * 4));
  float* qnorm = ws->q_eps.as<float>() + nq_pad;
  // queries: bf16 image pre-scaled by -2 (a power of two: no extra rounding), |q|^2, eps
  hipLaunchKernelGGL(l2_prepare_kernel, dim3((nq_pad + 3u) / 4u), dim3(256), 0, st, d_q, nq, nq_pad, -2.0f,
                     ws->q_bf16.as<uint16_t>(), qnorm, (uint32_t*)nullptr, 0, (const uint32_t*)ws->scal.as<uint32_t>(), ws->q_eps.as<float>(),
                     ws->cand_cnt.as<uint32_t>());
  const uint32_t k_eff = std::min(k, std::max(1u, n));
  auto gemm1 = k_eff <= 4u ? l2_gemm_kernel<1, 4> : l2_gemm_kernel<1, 8>;
  float* const d_seed = ws->thr.as<float>() + nq_pad;
  // pass 0: the k-th smallest score over a sample of the DB bounds A_k from above.
  //  * one-GEMM path (DBs of >= 64k rows): the sample is every (n_tiles / sample)-th tile, a quarter of the DB at most
  //    2048 tiles, so it is representative whatever the object order; seed + 2 eps IS the candidate threshold of pass 2 --
  //    a superset of what the exact A_k would admit (rows of rank <= ~k N / n_sample plus the eps band) -- and the pass
  //    that computes A_k over the whole DB is skipped.
  //  * classic path (smaller DBs): the sample is the first rows and the seed only limits what enters pass 1's lists.
  const bool fast = n_tiles >= 2048u;
  static const uint32_t env_st = getenv("TODHIP_L2_SAMPLE_TILES") ? (uint32_t)atoi(getenv("TODHIP_L2_SAMPLE_TILES")) : 0u;   // tuning knob
  const uint32_t sample_tiles = fast ? std::min(env_st ? env_st : 2048u, n_tiles / 4u) : std::min(n_tiles, 256u);
  const uint32_t sample_stride = fast ? n_tiles / sample_tiles : 1u;
  const float* seed = nullptr;
  bool one_gemm = false;
  // diagnostics: a threshold below every score, so that pass 2 runs its GEMM and finds nothing (timing the bare pass; results are garbage)
  static const bool no_candidates = getenv("TODHIP_L2_NO_CANDIDATES") != nullptr;
  // diagnostics: pass 2 leaves (start, end) of every wave in the partition buffer, see tools/l2_wave_times.py
  static const bool trace = getenv("TODHIP_L2_TRACE") != nullptr;
  const float margin = no_candidates ? -1e30f : 2.f;
  if ((fast || n_tiles >= 8u * sample_tiles) && sample_tiles * kTileRows >= k_eff) {
    static const uint32_t env_sc = getenv("TODHIP_L2_SEED_CHUNKS") ? (uint32_t)atoi(getenv("TODHIP_L2_SEED_CHUNKS")) : 0u;   // tuning knob
    const uint32_t s_chunks = std::min(sample_tiles, fast ? (env_sc ? std::min(env_sc, n_chunks) : n_chunks) : 32u), s_tpc = (sample_tiles + s_chunks - 1u) / s_chunks;
    const uint32_t s_used = (sample_tiles + s_tpc - 1u) / s_tpc;
    // the one-GEMM path's seed needs no exact k-th smallest of the sample: the k-th smallest of the partitions' minima bounds A_k
    // just as well and costs the GEMM's epilogue one instruction (needs k partitions with a real row: 2 s_used >= 512 here)
    auto gemm0 = fast ? l2_gemm_kernel<1, 1> : gemm1;
    hipLaunchKernelGGL(gemm0, dim3(s_used, q_blocks), dim3(256), 0, st, ws->db_bf16.as<uint16_t>(),
                       ws->db_norm.as<float>(), sample_tiles, s_tpc, sample_stride, ws->q_bf16.as<uint16_t>(), nq_pad, ws->part.as<float>(),
                       (const float*)nullptr, (uint32_t*)nullptr, (uint32_t*)nullptr, (uint32_t*)nullptr, (uint32_t*)nullptr);
    if (fast) {
      hipLaunchKernelGGL(l2_threshold_kernel, dim3((nq_pad + 3u) / 4u), dim3(256), 0, st, ws->part.as<float>(), 2u * s_used, nq,
                         nq_pad, k_eff, (const float*)nullptr, ws->q_eps.as<float>(), margin, ws->thr.as<float>());
      one_gemm = true;
    } else {
      hipLaunchKernelGGL(l2_threshold_kernel, dim3((nq_pad + 3u) / 4u), dim3(256), 0, st, ws->part.as<float>(), 2u * s_used, nq_pad,
                         nq_pad, k_eff, (const float*)nullptr, ws->q_eps.as<float>(), 0.f, d_seed);
      seed = d_seed;
    }
  }
  int slot = -1;
  if (!one_gemm) {
    if (ctx->time_kernels) { int rc = tod_timing_begin(ctx, &slot); if (rc != TODHIP_OK) return rc; }
    hipLaunchKernelGGL(gemm1, dim3(n_chunks, q_blocks), dim3(256), 0, st, ws->db_bf16.as<uint16_t>(), ws->db_norm.as<float>(), n_tiles,
                       tiles_per_chunk, 1u, ws->q_bf16.as<uint16_t>(), nq_pad, ws->part.as<float>(), seed, (uint32_t*)nullptr,
                       (uint32_t*)nullptr, (uint32_t*)nullptr, (uint32_t*)nullptr);
    if (slot >= 0) { int rc = tod_timing_end(ctx, slot); if (rc != TODHIP_OK) return rc; slot = -1; }
    hipLaunchKernelGGL(l2_threshold_kernel, dim3((nq_pad + 3u) / 4u), dim3(256), 0, st, ws->part.as<float>(), n_parts, nq, nq_pad,
                       k_eff, seed, ws->q_eps.as<float>(), margin, ws->thr.as<float>());
  } else if (ctx->time_kernels) {
    int rc = tod_timing_begin(ctx, &slot); if (rc != TODHIP_OK) return rc;
  }
  hipLaunchKernelGGL((l2_gemm_kernel<2, 4>), dim3(n_chunks, q_blocks), dim3(256), 0, st, ws->db_bf16.as<uint16_t>(),
                     ws->db_norm.as<float>(), n_tiles, tiles_per_chunk, 1u, ws->q_bf16.as<uint16_t>(), nq_pad, trace ? ws->part.as<float>() : (float*)nullptr,
                     ws->thr.as<float>(), ws->slots.as<uint32_t>(), ws->slot_cnt.as<uint32_t>(), ws->cand.as<uint32_t>(),
                     ws->cand_cnt.as<uint32_t>());
  if (slot >= 0) { int rc = tod_timing_end(ctx, slot); if (rc != TODHIP_OK) return rc; }
  if (trace) {
    const size_t n_waves = (size_t)q_blocks * n_chunks * kWaves;
    std::vector<uint64_t> tr(2 * n_waves);
    TOD_HIP(hipStreamSynchronize(st));
    TOD_HIP(hipMemcpy(tr.data(), ws->part.p, tr.size() * 8, hipMemcpyDeviceToHost));
    uint64_t t0 = ~0ull, t1 = 0;
    for (size_t i = 0; i < n_waves; ++i) { t0 = std::min(t0, tr[2 * i]); t1 = std::max(t1, tr[2 * i + 1]); }
    std::vector<double> start(n_waves), end(n_waves);
    for (size_t i = 0; i < n_waves; ++i) { start[i] = (tr[2 * i] - t0) * 0.01; end[i] = (tr[2 * i + 1] - t0) * 0.01; }
    std::sort(start.begin(), start.end()); std::sort(end.begin(), end.end());
    auto pct = [&](const std::vector<double>& v, double p) { return v[(size_t)(p * (v.size() - 1))]; };
    for (uint32_t y = 0; y < q_blocks; ++y) {
      double sum = 0; size_t cnt = 0; double byx[8] = {0}; size_t nx[8] = {0}; double byw[4] = {0};
      for (uint32_t x = 0; x + 1 < n_chunks; ++x) for (uint32_t w = 0; w < kWaves; ++w) {
        const double e = (tr[2 * (((size_t)y * n_chunks + x) * kWaves + w) + 1] - t0) * 0.01;
        sum += e; ++cnt; byx[x & 7] += e; ++nx[x & 7]; byw[w] += e;
      }
      fprintf(stderr, "[todhip l2 trace]   query block %u: mean end %.1f us; by chunk %% 8:", y, sum / cnt);
      for (int i = 0; i < 8; ++i) fprintf(stderr, " %.1f", byx[i] / nx[i]);
      fprintf(stderr, "; by wave:");
      for (int i = 0; i < 4; ++i) fprintf(stderr, " %.1f", byw[i] / (cnt / 4));
      fprintf(stderr, "\n");
    }
    fprintf(stderr, "[todhip l2 trace] pass 2: %zu waves, span %.1f us; wave start p50 %.1f p99 %.1f max %.1f us; wave end min %.1f p10 %.1f p50 %.1f p90 %.1f max %.1f us\n",
            n_waves, (t1 - t0) * 0.01, pct(start, 0.5), pct(start, 0.99), start.back(), end.front(), pct(end, 0.1), pct(end, 0.5), pct(end, 0.9), end.back());
  }
  hipLaunchKernelGGL(l2_rerank_kernel, dim3((nq + 3u) / 4u), dim3(256), 0, st, d_q, nq, ctx->db_desc.as<float>(),
                     ws->slots.as<uint32_t>(), ws->slot_cnt.as<uint32_t>(), n_parts, ws->cand.as<uint32_t>(),
                     ws->cand_cnt.as<uint32_t>(), k, ws->keys.as<uint64_t>(), overflow);
  hipLaunchKernelGGL(l2_exact_scan_kernel, dim3(nq), dim3(256), 0, st, d_q, nq, ctx->db_desc.as<float>(), n, k,
                     (const uint32_t*)overflow, ws->keys.as<uint64_t>());
  TOD_HIP(hipGetLastError());
  return TODHIP_OK;
}

extern "C" {

int todhip_match_l2_device(todhip_ctx* ctx, const void* d_q_desc, uint32_t nq, uint32_t k, float radius, void* d_counts,
                           void* d_matches, void* d_matches_xyz) {
  if (!ctx || !d_q_desc || !d_counts || !d_matches || !d_matches_xyz) return TODHIP_EINVAL;
  if (k == 0 || k > kTop || !(radius > 0.f)) return TODHIP_EINVAL;
  if (ctx->total_rows == 0) return TODHIP_ENODB;
  if (ctx->desc_bytes != kDim * 4u || ctx->shard_rows != ctx->total_rows) return TODHIP_EINVAL;   // needs a 128 x f32 DB
  if (nq == 0) return TODHIP_OK;
  TOD_HIP(hipSetDevice(ctx->device));
  static const bool exact_only = getenv("TODHIP_L2_EXACT_SCAN") != nullptr;      // diagnostics: skip the GEMM filter
  int rc = l2_keys(ctx, reinterpret_cast<const float*>(d_q_desc), nq, k, exact_only ? 1 : 0);
  if (rc != TODHIP_OK) return rc;
  L2Ws* ws = l2ws_of(ctx);
  hipLaunchKernelGGL(l2_finalize_kernel, dim3((nq + 255u) / 256u), dim3(256), 0, ctx->stream, ws->keys.as<uint64_t>(), nq, k, radius,
                     ctx->db_obj_off.as<uint32_t>(), ctx->n_objs, ctx->db_pts.as<float>(), reinterpret_cast<uint32_t*>(d_counts),
                     reinterpret_cast<todhip_dmatch*>(d_matches), reinterpret_cast<float*>(d_matches_xyz));
  TOD_HIP(hipGetLastError());
  ctx->counters.last_nq = nq; ctx->counters.last_k = k;
  return TODHIP_OK;
}

int todhip_match_l2(todhip_ctx* ctx, const float* q_desc, uint32_t nq, uint32_t k, float radius, uint32_t* row_ptr,
                    todhip_dmatch* matches, float* matches_xyz) {
  if (!ctx || (!q_desc && nq) || !row_ptr || (nq && (!matches || !matches_xyz))) return TODHIP_EINVAL;
  if (k == 0 || k > kTop || !(radius > 0.f)) return TODHIP_EINVAL;
  if (ctx->total_rows == 0) return TODHIP_ENODB;
  row_ptr[0] = 0;
  if (nq == 0) return TODHIP_OK;
  TOD_HIP(hipSetDevice(ctx->device));
  const size_t cap = (size_t)nq * k;
  TOD_HIP(ctx->m_q.reserve((size_t)nq * kDim * 4));
  TOD_HIP(ctx->m_counts.reserve((size_t)nq * 4));
  TOD_HIP(ctx->m_matches.reserve(cap * sizeof(todhip_dmatch)));
  TOD_HIP(ctx->m_xyz.reserve(cap * 12));
  TOD_HIP(hipMemcpyAsync(ctx->m_q.p, q_desc, (size_t)nq * kDim * 4, hipMemcpyHostToDevice, ctx->stream));
  int rc = todhip_match_l2_device(ctx, ctx->m_q.p, nq, k, radius, ctx->m_counts.p, ctx->m_matches.p, ctx->m_xyz.p);
  if (rc != TODHIP_OK) return rc;
  std::vector<uint32_t> counts(nq);
  std::vector<todhip_dmatch> m(cap);
  std::vector<float> x(cap * 3);
  TOD_HIP(hipMemcpyAsync(counts.data(), ctx->m_counts.p, (size_t)nq * 4, hipMemcpyDeviceToHost, ctx->stream));
  TOD_HIP(hipMemcpyAsync(m.data(), ctx->m_matches.p, cap * sizeof(todhip_dmatch), hipMemcpyDeviceToHost, ctx->stream));
  TOD_HIP(hipMemcpyAsync(x.data(), ctx->m_xyz.p, cap * 12, hipMemcpyDeviceToHost, ctx->stream));
  TOD_HIP(hipStreamSynchronize(ctx->stream));
  uint32_t out = 0;
  for (uint32_t qi = 0; qi < nq; ++qi) {                    // fixed-stride slots -> CSR
    for (uint32_t j = 0; j < counts[qi]; ++j) {
      matches[out] = m[(size_t)qi * k + j];
      std::memcpy(matches_xyz + 3 * (size_t)out, &x[((size_t)qi * k + j) * 3], 12);
      ++out;
    }
    row_ptr[qi + 1] = out;
  }
  return TODHIP_OK;
}

}  // extern "C"
