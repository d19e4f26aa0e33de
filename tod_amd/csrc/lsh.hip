// Optional LSH-approximate mode of the Hamming matcher (todhip_set_lsh). The reference's matcher IS an LSH index --
// cv::FlannBasedMatcher(cv::flann::LshIndexParams(n_tables, key_size, multi_probe_level)), DescriptorMatcher.cpp:175-180, parameters
// conf/detection.ork:32-38 -- and this library answers the same configuration with an EXACT search by default (a superset of
// anything an LSH index can return, and on this GPU not slower: the whole 1M-row DB costs a 1000-descriptor frame ~0.1 ms on
// the matrix cores). The mode exists for callers that want the recall/time trade-off of the reference's index at DB sizes
// where it matters (SURVEY 8(f) N4 c). It follows FLANN's published scheme -- a table's key = key_size bits of the descriptor;
// a query visits the buckets of every key within multi_probe_level flipped bits of its own (lsh_index.h fill_xor_mask); the rows
// found are ranked by exact Hamming distance -- with this library's own, seeded choice of key bits (FLANN draws them from
// rand()): parity unpinned, the definition is include/todhip.h's and the tests' CPU checker's.
//
//   index  per table: bucket offsets (2^key_size + 1 u32, a counting sort's prefix sums) + the shard's rows grouped by key
//   LSHQ   lsh_query_kernel<K>   wave = query: per table the key by ballot over the key bits; 64 probe buckets at a time (lane =
//                                probe: its bucket's begin/length), their rows flattened over the lanes (prefix sum + a 6-step
//                                search through cross-lane reads); per row 32 B of the DB, xor + popcount, a sorted per-lane
//                                list of K distinct keys; K rounds of wave-min at the end. Output = the same
//                                (distance << 32 | global row) lists the exact engines produce, so radius cut, ratio test,
//                                shard merge and the 3D gather downstream are shared.
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <vector>

#include "ctx.h"

namespace {

constexpr uint32_t kMaxTables = 32, kMaxKeyBits = 24, kMaxLevel = 3;

struct LshWs {
  uint32_t n_tables = 0, key_size = 0, level = 0;
  uint32_t n_masks = 0;
  uint64_t built_rows = ~0ull;          // shard_rows the index was built for (~0: not built)
  DevBuf pos, masks, off, rows, cursor, scan_tmp;
};

LshWs* lshws_of(todhip_ctx* ctx) {
  if (!ctx->lsh_ws) ctx->lsh_ws = new LshWs();
  return reinterpret_cast<LshWs*>(ctx->lsh_ws);
}

inline uint32_t mix32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }

// bit positions of table t's key: the first key_size entries of a Fisher-Yates shuffle of 0..255 driven by mix32(table, step)
void key_bits(uint32_t table, uint32_t key_size, uint8_t* pos) {
  uint8_t idx[256];
  for (int i = 0; i < 256; ++i) idx[i] = (uint8_t)i;
  for (uint32_t i = 0; i < key_size; ++i) {
    const uint32_t j = i + mix32(table * 0x9E3779B9U + i + 0xABCDU) % (256u - i);
    std::swap(idx[i], idx[j]);
    pos[i] = idx[i];
  }
}

__device__ __forceinline__ uint32_t key_of(const uint8_t* d, const uint8_t* pos, uint32_t key_size) {
  uint32_t k = 0;
  for (uint32_t b = 0; b < key_size; ++b) k |= (uint32_t)((d[pos[b] >> 3] >> (pos[b] & 7)) & 1u) << b;
  return k;
}

// thread = row: its key in every table, counted into the table's histogram (off[key + 1])
__global__ __launch_bounds__(256) void lsh_count_kernel(const uint8_t* __restrict__ db, uint32_t n_rows, const uint8_t* __restrict__ pos,
                                                        uint32_t n_tables, uint32_t key_size, uint32_t* __restrict__ off) {
  const uint32_t row = blockIdx.x * 256u + threadIdx.x;
  if (row >= n_rows) return;
  alignas(16) uint8_t d[32];
  const uint4* src = reinterpret_cast<const uint4*>(db + (size_t)row * 32);
  *reinterpret_cast<uint4*>(d) = src[0]; *reinterpret_cast<uint4*>(d + 16) = src[1];
  const size_t stride = ((size_t)1 << key_size) + 1;
  for (uint32_t t = 0; t < n_tables; ++t) atomicAdd(&off[t * stride + key_of(d, pos + t * key_size, key_size) + 1], 1u);
}

// thread = row: drop it into its bucket of table t (any order inside a bucket: the result is a top-k over a SET)
__global__ __launch_bounds__(256) void lsh_scatter_kernel(const uint8_t* __restrict__ db, uint32_t n_rows, const uint8_t* __restrict__ pos,
                                                          uint32_t key_size, const uint32_t* __restrict__ off, uint32_t* __restrict__ cursor,
                                                          uint32_t* __restrict__ rows) {
  const uint32_t row = blockIdx.x * 256u + threadIdx.x;
  if (row >= n_rows) return;
  alignas(16) uint8_t d[32];
  const uint4* src = reinterpret_cast<const uint4*>(db + (size_t)row * 32);
  *reinterpret_cast<uint4*>(d) = src[0]; *reinterpret_cast<uint4*>(d + 16) = src[1];
  const uint32_t key = key_of(d, pos, key_size);
  rows[off[key] + atomicAdd(&cursor[key], 1u)] = row;
}

template <uint32_t K>
__global__ __launch_bounds__(256) void lsh_query_kernel(const uint8_t* __restrict__ db, uint64_t shard_first, const uint8_t* __restrict__ q,
                                                        uint32_t nq, const uint8_t* __restrict__ pos, uint32_t n_tables, uint32_t key_size,
                                                        const uint32_t* __restrict__ masks, uint32_t n_masks,
                                                        const uint32_t* __restrict__ off, const uint32_t* __restrict__ rows, uint32_t n_rows,
                                                        uint64_t* __restrict__ keys) {
  const uint32_t qi = blockIdx.x * 4u + (threadIdx.x >> 6), l = threadIdx.x & 63u;
  if (qi >= nq) return;                                               // whole waves leave together
  const uint8_t* qd = q + (size_t)qi * 32;
  uint32_t qw[8];
  {
    const uint4 a = reinterpret_cast<const uint4*>(qd)[0], b = reinterpret_cast<const uint4*>(qd)[1];
    qw[0] = a.x; qw[1] = a.y; qw[2] = a.z; qw[3] = a.w; qw[4] = b.x; qw[5] = b.y; qw[6] = b.z; qw[7] = b.w;
  }
  uint64_t best[K];
#pragma unroll
  for (uint32_t j = 0; j < K; ++j) best[j] = ~0ull;
  const size_t stride = ((size_t)1 << key_size) + 1;
  for (uint32_t t = 0; t < n_tables; ++t) {
    const uint8_t p = l < key_size ? pos[t * key_size + l] : (uint8_t)0;
    const bool bit = l < key_size && ((qd[p >> 3] >> (p & 7)) & 1u);
    const uint32_t qkey = (uint32_t)__builtin_amdgcn_ballot_w64(bit);   // lane b = key bit b
    const uint32_t* off_t = off + t * stride;
    const uint32_t* rows_t = rows + (size_t)t * n_rows;
    for (uint32_t p0 = 0; p0 < n_masks; p0 += 64u) {
      uint32_t beg = 0, len = 0;
      if (p0 + l < n_masks) { const uint32_t kk = qkey ^ masks[p0 + l]; beg = off_t[kk]; len = off_t[kk + 1] - beg; }
      uint32_t incl = len;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) { const uint32_t v = __shfl_up(incl, o); if (l >= (uint32_t)o) incl += v; }
      const uint32_t total = __shfl(incl, 63), excl = incl - len;
      for (uint32_t base = 0; base < total; base += 64u) {             // uniform trip count: every lane takes part in the cross-lane reads
        const uint32_t i = base + l;
        const bool valid = i < total;
        const uint32_t ii = valid ? i : total - 1u;
        uint32_t j = 0;                                                // the last lane whose exclusive prefix is <= ii: the bucket row ii falls into
#pragma unroll
        for (uint32_t step = 32; step > 0; step >>= 1) { const uint32_t e = __shfl(excl, (int)(j + step)); if (e <= ii) j += step; }
        const uint32_t row = rows_t[__shfl(beg, (int)j) + (ii - __shfl(excl, (int)j))];
        if (valid) {
          const uint4 a = reinterpret_cast<const uint4*>(db + (size_t)row * 32)[0], b = reinterpret_cast<const uint4*>(db + (size_t)row * 32)[1];
          const uint32_t d = __popc(a.x ^ qw[0]) + __popc(a.y ^ qw[1]) + __popc(a.z ^ qw[2]) + __popc(a.w ^ qw[3]) + __popc(b.x ^ qw[4]) +
                             __popc(b.y ^ qw[5]) + __popc(b.z ^ qw[6]) + __popc(b.w ^ qw[7]);
          const uint64_t key = ((uint64_t)d << 32) | (shard_first + row);
          if (key < best[K - 1]) {
            bool dup = false;
#pragma unroll
            for (uint32_t jj = 0; jj < K; ++jj) dup |= best[jj] == key;   // the same row comes up in more than one table
            if (!dup) {
              uint64_t v = key;
#pragma unroll
              for (uint32_t jj = 0; jj < K; ++jj) { const uint64_t lo = best[jj] < v ? best[jj] : v; v = best[jj] < v ? v : best[jj]; best[jj] = lo; }
            }
          }
        }
      }
    }
  }
  // K rounds of wave-min over the lanes' heads; every lane holding the minimum pops it (the lanes' lists overlap)
  uint64_t* out = keys + (size_t)qi * K;
  for (uint32_t j = 0; j < K; ++j) {
    uint64_t m = best[0];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const uint64_t v = __shfl_xor(m, o); m = v < m ? v : m; }
    if (l == 0) out[j] = m;
    if (m != ~0ull && best[0] == m) {
#pragma unroll
      for (uint32_t jj = 0; jj + 1 < K; ++jj) best[jj] = best[jj + 1];
      best[K - 1] = ~0ull;
    }
  }
}

template <uint32_t K>
int launch_query(todhip_ctx* ctx, LshWs* ws, const void* d_q, uint32_t nq, uint64_t* d_lists) {
  hipLaunchKernelGGL((lsh_query_kernel<K>), dim3((nq + 3u) / 4u), dim3(256), 0, ctx->stream, ctx->db_desc.as<uint8_t>(), ctx->shard_first,
                     reinterpret_cast<const uint8_t*>(d_q), nq, ws->pos.as<uint8_t>(), ws->n_tables, ws->key_size, ws->masks.as<uint32_t>(),
                     ws->n_masks, ws->off.as<uint32_t>(), ws->rows.as<uint32_t>(), (uint32_t)ctx->shard_rows, d_lists);
  TOD_HIP(hipGetLastError());
  return TODHIP_OK;
}

}  // namespace

void tod_lsh_ws_free(todhip_ctx* ctx) {
  LshWs* ws = reinterpret_cast<LshWs*>(ctx->lsh_ws);
  if (!ws) return;
  DevBuf* bufs[] = {&ws->pos, &ws->masks, &ws->off, &ws->rows, &ws->cursor, &ws->scan_tmp};
  for (DevBuf* b : bufs) b->release();
  delete ws;
  ctx->lsh_ws = nullptr;
}

bool tod_lsh_enabled(const todhip_ctx* ctx) {
  const LshWs* ws = reinterpret_cast<const LshWs*>(ctx->lsh_ws);
  return ws && ws->n_tables > 0;
}

// (re)build the index over the shard's rows resident in ctx->db_desc; called by todhip_db_load and todhip_set_lsh
int tod_lsh_build(todhip_ctx* ctx) {
  LshWs* ws = lshws_of(ctx);
  ws->built_rows = ~0ull;
  if (ws->n_tables == 0 || ctx->desc_bytes != 32 || ctx->shard_rows == 0) return TODHIP_OK;
  if (ctx->shard_rows > 0xFFFFFFFFull) return TODHIP_EINVAL;
  hipStream_t st = ctx->stream;
  const uint32_t n = (uint32_t)ctx->shard_rows, T = ws->n_tables, ks = ws->key_size;
  const size_t stride = ((size_t)1 << ks) + 1;
  std::vector<uint8_t> pos((size_t)T * ks);
  for (uint32_t t = 0; t < T; ++t) key_bits(t, ks, &pos[(size_t)t * ks]);
  std::vector<uint32_t> masks;                                         // every xor mask of at most `level` key bits, the query's own bucket first
  masks.push_back(0u);
  for (uint32_t a = 0; a < ks && ws->level >= 1; ++a) masks.push_back(1u << a);
  for (uint32_t a = 0; a < ks && ws->level >= 2; ++a)
    for (uint32_t b = a + 1; b < ks; ++b) masks.push_back((1u << a) | (1u << b));
  for (uint32_t a = 0; a < ks && ws->level >= 3; ++a)
    for (uint32_t b = a + 1; b < ks; ++b)
      for (uint32_t c = b + 1; c < ks; ++c) masks.push_back((1u << a) | (1u << b) | (1u << c));
  ws->n_masks = (uint32_t)masks.size();
  TOD_HIP(ws->pos.reserve(pos.size()));
  TOD_HIP(ws->masks.reserve(masks.size() * 4));
  TOD_HIP(ws->off.reserve(T * stride * 4));
  TOD_HIP(ws->rows.reserve((size_t)T * n * 4));
  TOD_HIP(ws->cursor.reserve(stride * 4));
  TOD_HIP(hipMemcpyAsync(ws->pos.p, pos.data(), pos.size(), hipMemcpyHostToDevice, st));
  TOD_HIP(hipMemcpyAsync(ws->masks.p, masks.data(), masks.size() * 4, hipMemcpyHostToDevice, st));
  TOD_HIP(hipMemsetAsync(ws->off.p, 0, T * stride * 4, st));
  TOD_HIP(hipStreamSynchronize(st));                                   // pos / masks leave scope below
  hipLaunchKernelGGL(lsh_count_kernel, dim3((n + 255u) / 256u), dim3(256), 0, st, ctx->db_desc.as<uint8_t>(), n, ws->pos.as<uint8_t>(), T, ks,
                     ws->off.as<uint32_t>());
  size_t tmp_bytes = 0;
  if (hipcub::DeviceScan::InclusiveSum(nullptr, tmp_bytes, ws->off.as<uint32_t>(), ws->off.as<uint32_t>(), (int)stride, st) != hipSuccess) return TODHIP_EHIP;
  TOD_HIP(ws->scan_tmp.reserve(tmp_bytes));
  for (uint32_t t = 0; t < T; ++t) {
    uint32_t* off_t = ws->off.as<uint32_t>() + t * stride;
    TOD_HIP(hipcub::DeviceScan::InclusiveSum(ws->scan_tmp.p, tmp_bytes, off_t, off_t, (int)stride, st));
    TOD_HIP(hipMemsetAsync(ws->cursor.p, 0, stride * 4, st));
    hipLaunchKernelGGL(lsh_scatter_kernel, dim3((n + 255u) / 256u), dim3(256), 0, st, ctx->db_desc.as<uint8_t>(), n,
                       ws->pos.as<uint8_t>() + (size_t)t * ks, ks, off_t, ws->cursor.as<uint32_t>(), ws->rows.as<uint32_t>() + (size_t)t * n);
  }
  TOD_HIP(hipGetLastError());
  ws->built_rows = ctx->shard_rows;
  return TODHIP_OK;
}

// one list of k keys per query, the format of tod_match_lists
int tod_lsh_lists(todhip_ctx* ctx, const void* d_q, uint32_t nq, uint32_t k, uint64_t* d_lists, uint32_t* n_lists) {
  LshWs* ws = lshws_of(ctx);
  if (ws->built_rows != ctx->shard_rows) return TODHIP_EINVAL;
  *n_lists = 1;
  switch (k) {
    case 1: return launch_query<1>(ctx, ws, d_q, nq, d_lists);
    case 2: return launch_query<2>(ctx, ws, d_q, nq, d_lists);
    case 3: return launch_query<3>(ctx, ws, d_q, nq, d_lists);
    case 4: return launch_query<4>(ctx, ws, d_q, nq, d_lists);
    case 5: return launch_query<5>(ctx, ws, d_q, nq, d_lists);
    case 6: return launch_query<6>(ctx, ws, d_q, nq, d_lists);
    case 7: return launch_query<7>(ctx, ws, d_q, nq, d_lists);
    case 8: return launch_query<8>(ctx, ws, d_q, nq, d_lists);
    default: return TODHIP_EINVAL;
  }
}

extern "C" int todhip_set_lsh(todhip_ctx* ctx, uint32_t n_tables, uint32_t key_size, uint32_t multi_probe_level) {
  if (!ctx) return TODHIP_EINVAL;
  if (n_tables > kMaxTables || (n_tables && (key_size == 0 || key_size > kMaxKeyBits)) || multi_probe_level > kMaxLevel) return TODHIP_EINVAL;
  if (n_tables && multi_probe_level > key_size) return TODHIP_EINVAL;
  TOD_HIP(hipSetDevice(ctx->device));
  LshWs* ws = lshws_of(ctx);
  ws->n_tables = n_tables; ws->key_size = key_size; ws->level = multi_probe_level;
  return tod_lsh_build(ctx);
}
