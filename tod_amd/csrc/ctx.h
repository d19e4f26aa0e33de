// Internal context of libtodhip (not part of the C ABI).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/todhip.h"

#define TOD_HIP(call)                                  \
  do {                                                 \
    hipError_t e_ = (call);                            \
    if (e_ != hipSuccess) {                            \
      ctx->last_hip_error = (int)e_;                   \
      return TODHIP_EHIP;                              \
    }                                                  \
  } while (0)

// Growable device buffer: the hot path never calls hipMalloc once sizes have been seen.
// The short, latency-bound kernels (ORB, verifier, merges) raise their wave priority: beside the matcher's
// VALU-saturating waves a wave at default priority gets ~1/5 of a SIMD's issue slots.
#ifndef TOD_LATENCY_PRIO_LEVEL
#define TOD_LATENCY_PRIO_LEVEL 3
#endif
#define TOD_LATENCY_PRIO() __builtin_amdgcn_s_setprio(TOD_LATENCY_PRIO_LEVEL)

// bytes todhip_db_load allocates behind the last descriptor row: hamming_topk_mfma loads whole 32-row steps without a per-lane clamp
constexpr size_t kDbSlackBytes = 2048;

struct DevBuf {
  void* p = nullptr;
  size_t cap = 0;
  hipError_t reserve(size_t bytes) {
    if (bytes <= cap) return hipSuccess;
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
    size_t want = bytes + bytes / 4 + 256;
    hipError_t e = hipMalloc(&p, want);
    if (e == hipSuccess) cap = want;
    // diagnostics (tests/test_verify_gpu.py's poisoned run): fresh allocations start as 0xCD bytes instead of whatever the
    // allocator hands back, so that a read of memory nobody wrote shows up on every run, not on one in five
    static const bool poison = getenv("TODHIP_POISON_ALLOC") != nullptr;
    if (e == hipSuccess && poison) { e = hipMemset(p, 0xCD, want); if (e == hipSuccess) e = hipDeviceSynchronize(); }
    return e;
  }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
  }
  template <typename T> T* as() const { return reinterpret_cast<T*>(p); }
};

// Pinned host staging buffer.
struct HostBuf {
  void* p = nullptr;
  size_t cap = 0;
  hipError_t reserve(size_t bytes) {
    if (bytes <= cap) return hipSuccess;
    if (p) (void)hipHostFree(p);
    p = nullptr;
    cap = 0;
    size_t want = bytes + bytes / 4 + 256;
    hipError_t e = hipHostMalloc(&p, want, hipHostMallocDefault);
    if (e == hipSuccess) cap = want;
    static const bool poison = getenv("TODHIP_POISON_ALLOC") != nullptr;   // as DevBuf
    if (e == hipSuccess && poison) std::memset(p, 0xCD, want);
    return e;
  }
  void release() {
    if (p) (void)hipHostFree(p);
    p = nullptr;
    cap = 0;
  }
  template <typename T> T* as() const { return reinterpret_cast<T*>(p); }
};

struct todhip_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  int last_hip_error = 0;
  int n_cu = 256;

  // ---- object database (stage B1). Rows of this shard only; points + object table complete.
  uint32_t desc_bytes = 0;
  uint64_t total_rows = 0, shard_first = 0, shard_rows = 0;
  uint32_t n_objs = 0;
  DevBuf db_desc;          // shard_rows (+pad) x desc_bytes
  DevBuf db_pts;           // total_rows x 3 f32
  DevBuf db_obj_off;       // n_objs + 1 u32
  std::vector<uint32_t> h_obj_off;
  std::vector<float> h_spans;

  // ---- matcher workspaces
  DevBuf m_q, m_part, m_keys, m_counts, m_matches, m_xyz, m_bound;
  HostBuf h_stage;
  // optional HIP-event bracketing of the dominant matcher kernel: a ring of event pairs that is
  // drained lazily, so timing never adds a host sync inside the timed region
  bool time_kernels = false;
  static constexpr int kEvPairs = 64;
  hipEvent_t evp[2 * kEvPairs] = {};
  uint64_t ev_head = 0, ev_tail = 0;   // pairs [ev_tail, ev_head) are recorded and not yet read
  todhip_counters counters = {};
  int matcher_engine = TODHIP_ENGINE_AUTO;   // todhip_set_matcher_engine
  // K4x's half-block mode (match.hip, launch_topk_mfma_qt): cumulative {blocks that went on, blocks} of the launches in that mode,
  // as the merge kernel leaves them in pinned memory; launches left before the next probe
  HostBuf k4x_stats_host; DevBuf k4x_stats_dev;
  int k4x_force = -1;                         // todhip_set_matcher_block_split
  uint32_t k4x_seq_sent = 0, k4x_seq_seen = 0, k4x_split = 2, k4x_hold = 0, k4x_hold_len = 32, k4x_last[4] = {0, 0, 0, 0};   // launch_topk_mfma_qt
  float ratio = 0.f;                         // todhip_set_ratio_test (0 = off)

  std::vector<todhip_round_trace> traces;

  // ---- verifier / ORB workspaces live in their own translation units
  void* verify_ws = nullptr;
  void* orb_ws = nullptr;
  void* l2_ws = nullptr;
  void* pnp_ws = nullptr;
  void* lsh_ws = nullptr;
};

// capi.hip: a stream of the given kind (todhip_stream_create), honouring the process's CU partition
extern "C" hipError_t tod_stream_create(hipStream_t* out, int device, int kind);
extern "C" uint32_t tod_cu_partition();                     // todhip_set_cu_partition's current value
// match.hip
int tod_timing_begin(todhip_ctx* ctx, int* slot);
int tod_timing_end(todhip_ctx* ctx, int slot);
int tod_timing_drain(todhip_ctx* ctx, uint64_t keep);
int tod_match_lists(todhip_ctx* ctx, const void* d_q, uint32_t nq, uint32_t k, uint32_t radius, uint64_t* d_lists,
                    uint32_t* n_lists);
size_t tod_match_lists_bytes(uint32_t nq, uint32_t k);
int tod_match_shard_keys(todhip_ctx* ctx, const void* d_q, uint32_t nq, uint32_t k, uint32_t radius, uint64_t* d_keys);
// k_in: entries per list; k_out: matches per query in the outputs (k_in > k_out only for the ratio test with k == 1)
int tod_match_finalize(todhip_ctx* ctx, const uint64_t* d_keys_all, uint32_t n_shards, uint32_t nq, uint32_t k_in, uint32_t k_out,
                       uint32_t radius, uint32_t* d_counts, todhip_dmatch* d_matches, float* d_xyz,
                       hipStream_t stream = nullptr);   // nullptr: the context's stream
// verify.hip / orb.hip
void tod_verify_ws_free(todhip_ctx* ctx);
// verify.hip: ClusterPerObject of F frames on the device without a cloud (pnp.hip's 2D-only branch); d_err: 8 words per frame
int tod_cluster_frames_nocloud(todhip_ctx* ctx, uint32_t F, const float* d_kp_xy, uint32_t nq, const uint32_t* d_counts,
                               const todhip_dmatch* d_matches, const float* d_mxyz, uint32_t k, uint32_t n_objs, float* d_X,
                               uint32_t* d_qidx, uint32_t* d_hist, uint32_t* d_goff, uint32_t* d_err);
void tod_l2_ws_free(todhip_ctx* ctx);
void tod_pnp_ws_free(todhip_ctx* ctx);      // pnp.hip
// lsh.hip: the optional LSH-approximate mode (todhip_set_lsh)
void tod_lsh_ws_free(todhip_ctx* ctx);
bool tod_lsh_enabled(const todhip_ctx* ctx);
int tod_lsh_build(todhip_ctx* ctx);
int tod_lsh_lists(todhip_ctx* ctx, const void* d_q, uint32_t nq, uint32_t k, uint64_t* d_lists, uint32_t* n_lists);
int tod_l2_db_prepare(todhip_ctx* ctx);      // l2.hip: bf16 image + norms of a 128 x f32 DB resident in db_desc
void tod_orb_ws_free(todhip_ctx* ctx);
int tod_orb_device(todhip_ctx* ctx, const uint8_t* d_gray, const uint8_t* d_mask, uint32_t H, uint32_t W, uint32_t stride,
                   uint32_t n_features, uint32_t n_levels, float scale_factor, const int8_t* pattern, float* d_kp_xy,
                   float* d_kp_aux, uint8_t* d_desc, uint32_t cap, uint32_t* n_out);
