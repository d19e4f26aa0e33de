// C ABI entry points of libtodhip (include/todhip.h): context lifetime, DB ingest (stage B1),
// and the host-buffer / device-buffer forms of the matcher. The verifier and ORB entry points live
// in verify.hip and orb.hip.
#include <algorithm>
#include <atomic>
#include <cfloat>
#include <cmath>
#include <cstring>
#include <cstdlib>
#include <new>
#include <vector>

#include "ctx.h"

extern "C" {

int todhip_version(void) { return TODHIP_VERSION; }

int todhip_create(int device, void* hip_stream, todhip_ctx** out) {
  if (!out) return TODHIP_EINVAL;
  *out = nullptr;
  todhip_ctx* ctx = new (std::nothrow) todhip_ctx();
  if (!ctx) return TODHIP_ENOMEM;
  ctx->device = device;
  hipError_t e = hipSetDevice(device);
  if (e == hipSuccess) {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0)
      ctx->n_cu = prop.multiProcessorCount;
    if (hip_stream) {
      ctx->stream = reinterpret_cast<hipStream_t>(hip_stream);
    } else {
      e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
      ctx->own_stream = (e == hipSuccess);
    }
  }
  for (int i = 0; i < 2 * todhip_ctx::kEvPairs && e == hipSuccess; ++i) e = hipEventCreate(&ctx->evp[i]);
  if (e != hipSuccess) {
    // the product path fails loudly when there is no usable HIP device
    for (int i = 0; i < 2 * todhip_ctx::kEvPairs; ++i)
      if (ctx->evp[i]) (void)hipEventDestroy(ctx->evp[i]);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    return TODHIP_EHIP;
  }
  *out = ctx;
  return TODHIP_OK;
}

void todhip_destroy(todhip_ctx* ctx) {
  if (!ctx) return;
  (void)hipSetDevice(ctx->device);
  (void)hipStreamSynchronize(ctx->stream);
  tod_verify_ws_free(ctx);
  tod_orb_ws_free(ctx);
  tod_l2_ws_free(ctx);
  tod_pnp_ws_free(ctx);
  tod_lsh_ws_free(ctx);
  ctx->db_desc.release(); ctx->db_pts.release(); ctx->db_obj_off.release();
  ctx->m_q.release(); ctx->m_part.release(); ctx->m_keys.release(); ctx->m_counts.release();
  ctx->m_matches.release(); ctx->m_xyz.release(); ctx->m_bound.release(); ctx->h_stage.release();
  ctx->k4x_stats_host.release(); ctx->k4x_stats_dev.release();
  for (int i = 0; i < 2 * todhip_ctx::kEvPairs; ++i)
    if (ctx->evp[i]) (void)hipEventDestroy(ctx->evp[i]);
  if (ctx->own_stream) (void)hipStreamDestroy(ctx->stream);
  delete ctx;
}

// ---- CU partition: the short, latency-bound kernels (ORB, verifier, merges) beside the matcher's chip-filling DB pass.
// A wave of theirs shares its SIMD with resident matcher waves and its kernels run several times stretched; with a partition the
// last `latency_cus` compute units take no matcher waves at all (hipExtStreamCreateWithCUMask: a queue's waves only go to the CUs
// of its mask; the runtime deals the mask's bits round robin over the XCDs, so a multiple of 8 takes the same share of each).
static std::atomic<uint32_t> g_latency_cus{0xFFFFFFFFu};     // 0xFFFFFFFF: not set yet (environment TODHIP_LATENCY_CUS, else 0)
static uint32_t latency_cus_now() {
  uint32_t v = g_latency_cus.load();
  if (v == 0xFFFFFFFFu) {
    const char* e = getenv("TODHIP_LATENCY_CUS");
    v = e ? (uint32_t)strtoul(e, nullptr, 10) : 0u;
    g_latency_cus.store(v);
  }
  return v;
}
int todhip_set_cu_partition(uint32_t latency_cus) { g_latency_cus.store(latency_cus); return TODHIP_OK; }
uint32_t tod_cu_partition() { return latency_cus_now(); }

hipError_t tod_stream_create(hipStream_t* out, int device, int kind) {
  const uint32_t lat = latency_cus_now();
  int n_cu = 0;
  hipError_t e = hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, device);
  if (e != hipSuccess) return e;
  if (lat == 0u || lat >= (uint32_t)n_cu) {
    if (kind == TODHIP_STREAM_LATENCY) {
      int least = 0, greatest = 0;                          // latency-bound work: the highest priority there is
      if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) greatest = 0;
      return hipStreamCreateWithPriority(out, hipStreamNonBlocking, greatest);
    }
    return hipStreamCreateWithFlags(out, hipStreamNonBlocking);
  }
  const uint32_t words = ((uint32_t)n_cu + 31u) / 32u;
  std::vector<uint32_t> mask(words, 0u);
  const uint32_t lo = kind == TODHIP_STREAM_LATENCY ? (uint32_t)n_cu - lat : 0u, hi = kind == TODHIP_STREAM_LATENCY ? (uint32_t)n_cu : (uint32_t)n_cu - lat;
  for (uint32_t b = lo; b < hi; ++b) mask[b >> 5] |= 1u << (b & 31u);
  return hipExtStreamCreateWithCUMask(out, words, mask.data());
}

int todhip_stream_create(int device, int kind, void** stream_out) {
  if (!stream_out || (kind != TODHIP_STREAM_THROUGHPUT && kind != TODHIP_STREAM_LATENCY)) return TODHIP_EINVAL;
  if (hipSetDevice(device) != hipSuccess) return TODHIP_EHIP;
  hipStream_t st = nullptr;
  if (tod_stream_create(&st, device, kind) != hipSuccess) return TODHIP_EHIP;
  *stream_out = (void*)st;
  return TODHIP_OK;
}
int todhip_stream_destroy(void* stream) {
  if (!stream) return TODHIP_EINVAL;
  return hipStreamDestroy((hipStream_t)stream) == hipSuccess ? TODHIP_OK : TODHIP_EHIP;
}

void* todhip_stream(todhip_ctx* ctx) { return ctx ? (void*)ctx->stream : nullptr; }
int todhip_last_hip_error(const todhip_ctx* ctx) { return ctx ? ctx->last_hip_error : 0; }

int todhip_synchronize(todhip_ctx* ctx) {
  if (!ctx) return TODHIP_EINVAL;
  TOD_HIP(hipStreamSynchronize(ctx->stream));
  return TODHIP_OK;
}

int todhip_get_counters(todhip_ctx* ctx, todhip_counters* out) {
  if (!ctx || !out) return TODHIP_EINVAL;
  int rc = tod_timing_drain(ctx, 0);   // waits for the bracketed launches that are still in flight
  if (rc != TODHIP_OK) return rc;
  *out = ctx->counters;
  return TODHIP_OK;
}

int todhip_set_kernel_timing(todhip_ctx* ctx, int enable) {
  if (!ctx) return TODHIP_EINVAL;
  ctx->time_kernels = enable != 0;
  return TODHIP_OK;
}

int todhip_set_matcher_engine(todhip_ctx* ctx, int engine) {
  if (!ctx || engine < TODHIP_ENGINE_AUTO || engine > TODHIP_ENGINE_MFMA) return TODHIP_EINVAL;
  ctx->matcher_engine = engine;
  return TODHIP_OK;
}

int todhip_set_matcher_block_split(todhip_ctx* ctx, int split) {
  if (!ctx || !(split == -1 || split == 0 || split == 2 || split == 3)) return TODHIP_EINVAL;
  ctx->k4x_force = split;
  return TODHIP_OK;
}

int todhip_set_ratio_test(todhip_ctx* ctx, float ratio) {
  if (!ctx || !(ratio >= 0.f) || ratio > 1.f) return TODHIP_EINVAL;
  ctx->ratio = ratio;
  return TODHIP_OK;
}

// Object-aligned contiguous shards: object o belongs to the shard whose row range contains its first row
// when the rows are cut into shard_count equal pieces (an object never straddles two devices).
static void shard_bounds(const std::vector<uint32_t>& off, uint32_t n_objs, uint32_t rank, uint32_t count,
                         uint32_t* obj_lo, uint32_t* obj_hi) {
  const uint64_t total = off[n_objs];
  auto owner = [&](uint32_t o) -> uint32_t {
    if (total == 0) return 0;
    uint64_t s = (uint64_t)off[o] * count / total;
    return (uint32_t)std::min<uint64_t>(s, count - 1);
  };
  uint32_t lo = 0;
  while (lo < n_objs && owner(lo) < rank) ++lo;
  uint32_t hi = lo;
  while (hi < n_objs && owner(hi) == rank) ++hi;
  *obj_lo = lo;
  *obj_hi = hi;
}

// span of every object on the device (DescriptorMatcher.cpp:104-121): min / max are exact, the sum and the root are the host's
__global__ __launch_bounds__(256) void spans_kernel(const float* __restrict__ pts, const uint32_t* __restrict__ obj_off, float* __restrict__ spans) {
  __shared__ float s_mn[3][256], s_mx[3][256];
  const uint32_t o = blockIdx.x, lo = obj_off[o], hi = obj_off[o + 1], tid = threadIdx.x;
  float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
  for (uint32_t i = lo + tid; i < hi; i += 256u)
    for (int c = 0; c < 3; ++c) { const float v = pts[3 * (size_t)i + c]; mn[c] = std::min(mn[c], v); mx[c] = std::max(mx[c], v); }
  for (int c = 0; c < 3; ++c) { s_mn[c][tid] = mn[c]; s_mx[c][tid] = mx[c]; }
  __syncthreads();
  for (uint32_t st = 128; st > 0; st >>= 1) {
    if (tid < st)
      for (int c = 0; c < 3; ++c) { s_mn[c][tid] = std::min(s_mn[c][tid], s_mn[c][tid + st]); s_mx[c][tid] = std::max(s_mx[c][tid], s_mx[c][tid + st]); }
    __syncthreads();
  }
  if (tid == 0) {
    const float s = (s_mx[0][0] - s_mn[0][0]) * (s_mx[0][0] - s_mn[0][0]) + (s_mx[1][0] - s_mn[1][0]) * (s_mx[1][0] - s_mn[1][0]) +
                    (s_mx[2][0] - s_mn[2][0]) * (s_mx[2][0] - s_mn[2][0]);
    spans[o] = sqrtf(s);
  }
}

static int db_load_impl(todhip_ctx* ctx, const todhip_object* objs, uint32_t n_objs, uint32_t desc_bytes,
                        uint32_t shard_rank, uint32_t shard_count, float* spans_out, bool device_src) {
  if (!ctx || (!objs && n_objs) || shard_count == 0 || shard_rank >= shard_count) return TODHIP_EINVAL;
  // 32: 256-bit binary descriptors (ORB), Hamming; 512: 128 x f32 (SIFT-like), L2 -- one device only
  if (desc_bytes != 32 && !(desc_bytes == 512 && shard_count == 1)) return TODHIP_EINVAL;
  TOD_HIP(hipSetDevice(ctx->device));
  ctx->h_obj_off.assign(n_objs + 1, 0u);
  uint64_t total = 0;
  for (uint32_t o = 0; o < n_objs; ++o) {
    if (objs[o].n && (!objs[o].desc || !objs[o].pts_xyz)) return TODHIP_EINVAL;
    ctx->h_obj_off[o] = (uint32_t)total;
    total += objs[o].n;
    if (total > 0xFFFFFFF0ull) return TODHIP_EINVAL;
  }
  ctx->h_obj_off[n_objs] = (uint32_t)total;
  // spans: diagonal of the axis-aligned bounding box of the model points, DescriptorMatcher.cpp:104-121
  ctx->h_spans.assign(n_objs, 0.f);
  for (uint32_t o = 0; o < n_objs && !device_src; ++o) {
    float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    const float* p = objs[o].pts_xyz;
    for (uint32_t i = 0; i < objs[o].n; ++i)
      for (int c = 0; c < 3; ++c) {
        mn[c] = std::min(mn[c], p[3 * (size_t)i + c]);
        mx[c] = std::max(mx[c], p[3 * (size_t)i + c]);
      }
    float s = (mx[0] - mn[0]) * (mx[0] - mn[0]) + (mx[1] - mn[1]) * (mx[1] - mn[1]) +
              (mx[2] - mn[2]) * (mx[2] - mn[2]);
    ctx->h_spans[o] = std::sqrt(s);
    if (spans_out) spans_out[o] = ctx->h_spans[o];
  }
  uint32_t obj_lo = 0, obj_hi = n_objs;
  shard_bounds(ctx->h_obj_off, n_objs, shard_rank, shard_count, &obj_lo, &obj_hi);
  ctx->desc_bytes = desc_bytes;
  ctx->n_objs = n_objs;
  ctx->total_rows = total;
  ctx->shard_first = ctx->h_obj_off[obj_lo];
  ctx->shard_rows = (uint64_t)ctx->h_obj_off[obj_hi] - ctx->h_obj_off[obj_lo];
  ctx->counters.db_rows = total;
  ctx->counters.db_objects = n_objs;

  TOD_HIP(hipStreamSynchronize(ctx->stream));
  TOD_HIP(ctx->db_desc.reserve((size_t)ctx->shard_rows * desc_bytes + kDbSlackBytes));   // (the matrix-core matcher's last step reads into the slack)
  TOD_HIP(ctx->db_pts.reserve((size_t)total * 3 * sizeof(float) + 16));
  TOD_HIP(ctx->db_obj_off.reserve((size_t)(n_objs + 1) * sizeof(uint32_t)));
  size_t row = 0;
  for (uint32_t o = obj_lo; o < obj_hi; ++o) {
    if (objs[o].n)
      TOD_HIP(hipMemcpyAsync(ctx->db_desc.as<uint8_t>() + row * desc_bytes, objs[o].desc,
                             (size_t)objs[o].n * desc_bytes, device_src ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, ctx->stream));
    row += objs[o].n;
  }
  for (uint32_t o = 0; o < n_objs; ++o)
    if (objs[o].n)
      TOD_HIP(hipMemcpyAsync(ctx->db_pts.as<float>() + (size_t)ctx->h_obj_off[o] * 3, objs[o].pts_xyz,
                             (size_t)objs[o].n * 3 * sizeof(float), device_src ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, ctx->stream));
  TOD_HIP(hipMemcpyAsync(ctx->db_obj_off.p, ctx->h_obj_off.data(), (size_t)(n_objs + 1) * sizeof(uint32_t),
                         hipMemcpyHostToDevice, ctx->stream));
  if (device_src && n_objs) {                               // the model points never left the device: spans there, n_objs floats back
    DevBuf d_spans;
    TOD_HIP(d_spans.reserve((size_t)n_objs * sizeof(float)));
    hipLaunchKernelGGL(spans_kernel, dim3(n_objs), dim3(256), 0, ctx->stream, ctx->db_pts.as<float>(), ctx->db_obj_off.as<uint32_t>(),
                       d_spans.as<float>());
    hipError_t e = hipMemcpyAsync(ctx->h_spans.data(), d_spans.p, (size_t)n_objs * sizeof(float), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    d_spans.release();
    if (e != hipSuccess) { ctx->last_hip_error = (int)e; return TODHIP_EHIP; }
    for (uint32_t o = 0; o < n_objs; ++o) {
      if (objs[o].n == 0) ctx->h_spans[o] = std::sqrt(3.f * (-FLT_MAX - FLT_MAX) * (-FLT_MAX - FLT_MAX));   // the host loop's result for no points
      if (spans_out) spans_out[o] = ctx->h_spans[o];
    }
  }
  if (desc_bytes == 512) {
    int rc = tod_l2_db_prepare(ctx);
    if (rc != TODHIP_OK) return rc;
  }
  if (tod_lsh_enabled(ctx)) {                               // todhip_set_lsh before the load: index this shard
    int rc = tod_lsh_build(ctx);
    if (rc != TODHIP_OK) return rc;
  }
  TOD_HIP(hipStreamSynchronize(ctx->stream));
  return TODHIP_OK;
}

int todhip_db_load(todhip_ctx* ctx, const todhip_object* objs, uint32_t n_objs, uint32_t desc_bytes,
                   uint32_t shard_rank, uint32_t shard_count, float* spans_out) {
  return db_load_impl(ctx, objs, n_objs, desc_bytes, shard_rank, shard_count, spans_out, false);
}

int todhip_db_load_device(todhip_ctx* ctx, const todhip_object* objs, uint32_t n_objs, uint32_t desc_bytes,
                          uint32_t shard_rank, uint32_t shard_count, float* spans_out) {
  return db_load_impl(ctx, objs, n_objs, desc_bytes, shard_rank, shard_count, spans_out, true);
}

int todhip_db_info(const todhip_ctx* ctx, uint64_t* total_rows, uint64_t* shard_first_row, uint64_t* shard_rows,
                   uint32_t* n_objs) {
  if (!ctx) return TODHIP_EINVAL;
  if (total_rows) *total_rows = ctx->total_rows;
  if (shard_first_row) *shard_first_row = ctx->shard_first;
  if (shard_rows) *shard_rows = ctx->shard_rows;
  if (n_objs) *n_objs = ctx->n_objs;
  return TODHIP_OK;
}

int todhip_match_shard_device(todhip_ctx* ctx, const void* d_q_desc, uint32_t nq, uint32_t k, uint32_t radius,
                              void* d_keys) {
  if (!ctx || !d_q_desc || !d_keys || k == 0 || k > 8 || radius == 0) return TODHIP_EINVAL;
  if (ctx->ratio > 0.f && k < 2) return TODHIP_EINVAL;        // the ratio test needs every shard's two nearest
  if (ctx->total_rows == 0) return TODHIP_ENODB;
  return tod_match_shard_keys(ctx, d_q_desc, nq, k, radius, reinterpret_cast<uint64_t*>(d_keys));
}

int todhip_merge_shards_device(todhip_ctx* ctx, const void* d_keys_all, uint32_t n_shards, uint32_t nq, uint32_t k,
                               uint32_t radius, void* d_counts, void* d_matches, void* d_matches_xyz) {
  if (!ctx || !d_keys_all || !d_counts || !d_matches || !d_matches_xyz) return TODHIP_EINVAL;
  if (k == 0 || k > 8 || radius == 0 || n_shards == 0) return TODHIP_EINVAL;
  if (ctx->total_rows == 0) return TODHIP_ENODB;
  if (ctx->ratio > 0.f && k < 2) return TODHIP_EINVAL;
  return tod_match_finalize(ctx, reinterpret_cast<const uint64_t*>(d_keys_all), n_shards, nq, k, k, radius,
                            reinterpret_cast<uint32_t*>(d_counts), reinterpret_cast<todhip_dmatch*>(d_matches),
                            reinterpret_cast<float*>(d_matches_xyz));
}

int todhip_merge_shards_device_on(todhip_ctx* ctx, void* hip_stream, const void* d_keys_all, uint32_t n_shards, uint32_t nq,
                                  uint32_t k, uint32_t radius, void* d_counts, void* d_matches, void* d_matches_xyz) {
  if (!ctx || !hip_stream || !d_keys_all || !d_counts || !d_matches || !d_matches_xyz) return TODHIP_EINVAL;
  if (k == 0 || k > 8 || radius == 0 || n_shards == 0) return TODHIP_EINVAL;
  if (ctx->total_rows == 0) return TODHIP_ENODB;
  if (ctx->ratio > 0.f && k < 2) return TODHIP_EINVAL;
  return tod_match_finalize(ctx, reinterpret_cast<const uint64_t*>(d_keys_all), n_shards, nq, k, k, radius,
                            reinterpret_cast<uint32_t*>(d_counts), reinterpret_cast<todhip_dmatch*>(d_matches),
                            reinterpret_cast<float*>(d_matches_xyz), reinterpret_cast<hipStream_t>(hip_stream));
}

int todhip_match_device(todhip_ctx* ctx, const void* d_q_desc, uint32_t nq, uint32_t k, uint32_t radius,
                        void* d_counts, void* d_matches, void* d_matches_xyz) {
  if (!ctx || !d_q_desc || !d_counts || !d_matches || !d_matches_xyz) return TODHIP_EINVAL;
  if (k == 0 || k > 8 || radius == 0) return TODHIP_EINVAL;   // radius 0: DescriptorMatcher.cpp:237 is UB there
  if (ctx->total_rows == 0) return TODHIP_ENODB;
  if (ctx->shard_rows != ctx->total_rows) return TODHIP_EINVAL; // sharded DBs use the two-step form
  if (nq == 0) return TODHIP_OK;
  // single device: the stage-1 merge lists go straight into the finalize kernel (lists == "shards")
  const uint32_t k_in = (ctx->ratio > 0.f && k < 2) ? 2u : k;          // the ratio test looks at the two nearest
  TOD_HIP(ctx->m_keys.reserve(tod_match_lists_bytes(nq, k_in)));
  uint32_t n_lists = 0;
  int rc = tod_match_lists(ctx, d_q_desc, nq, k_in, radius, ctx->m_keys.as<uint64_t>(), &n_lists);
  if (rc != TODHIP_OK) return rc;
  rc = tod_match_finalize(ctx, ctx->m_keys.as<uint64_t>(), n_lists, nq, k_in, k, radius,
                          reinterpret_cast<uint32_t*>(d_counts), reinterpret_cast<todhip_dmatch*>(d_matches),
                          reinterpret_cast<float*>(d_matches_xyz));
  if (rc == TODHIP_OK) { ctx->counters.last_nq = nq; ctx->counters.last_k = k; }
  return rc;
}

int todhip_match(todhip_ctx* ctx, const uint8_t* q_desc, uint32_t nq, uint32_t k, uint32_t radius,
                 uint32_t* row_ptr, todhip_dmatch* matches, float* matches_xyz) {
  if (!ctx || !row_ptr || (nq && (!q_desc || !matches || !matches_xyz))) return TODHIP_EINVAL;
  if (k == 0 || k > 8 || radius == 0) return TODHIP_EINVAL;
  if (ctx->total_rows == 0) return TODHIP_ENODB;
  if (nq == 0) { row_ptr[0] = 0; return TODHIP_OK; }
  TOD_HIP(hipSetDevice(ctx->device));
  const size_t nm = (size_t)nq * k;
  TOD_HIP(ctx->m_q.reserve((size_t)nq * ctx->desc_bytes));
  TOD_HIP(ctx->m_counts.reserve((size_t)nq * sizeof(uint32_t)));
  TOD_HIP(ctx->m_matches.reserve(nm * sizeof(todhip_dmatch)));
  TOD_HIP(ctx->m_xyz.reserve(nm * 3 * sizeof(float)));
  const size_t stage_bytes = (size_t)nq * sizeof(uint32_t) + nm * sizeof(todhip_dmatch) + nm * 3 * sizeof(float);
  TOD_HIP(ctx->h_stage.reserve(stage_bytes));
  TOD_HIP(hipMemcpyAsync(ctx->m_q.p, q_desc, (size_t)nq * ctx->desc_bytes, hipMemcpyHostToDevice, ctx->stream));
  // the finalize kernel writes its three outputs straight into pinned host memory (a few tens of KB over PCIe): no device-to-host
  // copies behind it, one synchronization
  uint32_t* h_counts = ctx->h_stage.as<uint32_t>();
  todhip_dmatch* h_m = reinterpret_cast<todhip_dmatch*>(h_counts + nq);
  float* h_xyz = reinterpret_cast<float*>(h_m + nm);
  int rc = todhip_match_device(ctx, ctx->m_q.p, nq, k, radius, h_counts, h_m, h_xyz);
  if (rc != TODHIP_OK) return rc;
  TOD_HIP(hipStreamSynchronize(ctx->stream));
  // fixed stride k -> CSR (the cell's vector<vector<DMatch>> / vector<Mat> shapes)
  uint32_t out = 0;
  for (uint32_t qi = 0; qi < nq; ++qi) {
    row_ptr[qi] = out;
    const uint32_t c = h_counts[qi];
    std::memcpy(matches + out, h_m + (size_t)qi * k, (size_t)c * sizeof(todhip_dmatch));
    std::memcpy(matches_xyz + (size_t)out * 3, h_xyz + (size_t)qi * k * 3, (size_t)c * 3 * sizeof(float));
    out += c;
  }
  row_ptr[nq] = out;
  ctx->counters.last_matches = out;
  return TODHIP_OK;
}

}  // extern "C"
