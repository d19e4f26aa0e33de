// Stage A on gfx950: ORB keypoints (FAST-9/16 + Harris ranking + intensity-centroid orientation) and 256-bit
// rBRIEF descriptors. In the reference this stage is third-party code outside its tree
// (ecto_opencv.features2d.FeatureDescriptor -> cv::ORB; python/object_recognition_tod/detector.py:10,27,
// src/training/Trainer.cpp:144-150), so there is nothing to be bit-exact with except the published algorithm;
// the free choices (fixed-point resize and blur, score definition, tie-breaks, rotation without angle
// quantisation, output order) are the ones stated at the top of this repo's CPU restatement and are followed
// here operation for operation (integer moments and gradients; float expressions without contraction).
//
// All images of a 640x480 pyramid are a few hundred KB: every kernel is latency/launch bound, not bandwidth
// bound; the per-level work is a fixed sequence of small launches with no host round trip until the final count.
// A batch of frames AND the pyramid levels ride in the last grid dimension of every launch: (level l, frame f) is "virtual
// frame" l F + f and owns slice l F + f of every workspace buffer and its own row of control words, the per-level geometry
// comes from a small table in the kernel arguments (LevelTab). A batch of any size and any number of levels therefore costs
// 13 + (levels - 1) launches (round 1 and most of round 2: 2 + 11 per level), blocks beyond a smaller level's extent leave at once.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

// ORB's kernels are wide (a thread per pixel, a wave per keypoint) and their stage has slack in the pipeline: at the raised wave
// priority of the verifier's single-wave kernels they took issue slots from the matcher's DB pass -- with the default priority the pass
// is 6 % faster in the pipeline (1.91 instead of 2.03 ms per 32 000 x 1M launch) and the headline 4 % (tools/ab_prio.sh), ORB's own
// stage time unchanged within the spread
#define TOD_LATENCY_PRIO_LEVEL 0
#include "ctx.h"

namespace {

constexpr int kEdge = 31;
constexpr int kHalfPatch = 15;
constexpr int kFastThr = 20;
constexpr int kMaxLevels = 16;

struct Cand { int x, y, score; float harris; };

// control words of one level and one frame (device): what the selection kernels hand to each other without a
// host round trip; [8 + l] = keypoints of level l
enum { W_NCAND = 0, W_NSEL1 = 1, W_THR = 2, W_NEED_EQ = 3, W_NEQ = 4, W_NGT = 5, W_WANT = 6, W_HIST = 32 };
constexpr uint32_t kCtlWords = 512;

// same generator as the CPU restatement: seeded xorshift, points inside radius 13
void default_pattern(int8_t* pat) {
  uint32_t s = 0x9E3779B9u;
  for (int i = 0; i < 256 * 2; ++i) {
    int x, y;
    do {
      s ^= s << 13; s ^= s >> 17; s ^= s << 5;
      x = (int)(s % 27u) - 13;
      s ^= s << 13; s ^= s >> 17; s ^= s << 5;
      y = (int)(s % 27u) - 13;
    } while (x * x + y * y > 13 * 13);
    pat[2 * i] = (int8_t)x;
    pat[2 * i + 1] = (int8_t)y;
  }
}

void disc_umax(int* umax) {
  const int hp = kHalfPatch;
  const int vmax = (int)std::floor(hp * std::sqrt(2.0) / 2 + 1), vmin = (int)std::ceil(hp * std::sqrt(2.0) / 2);
  for (int v = 0; v <= vmax; ++v) umax[v] = (int)std::rint(std::sqrt((double)hp * hp - (double)v * v));
  for (int v = hp, v0 = 0; v >= vmin; --v) {
    while (umax[v0] == umax[v0 + 1]) ++v0;
    umax[v] = v0;
    ++v0;
  }
}

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// geometry of the pyramid levels; want[l] == 0: the level yields nothing (too small, or no features asked of it)
struct LevelTab {
  uint32_t n_levels, F;
  const uint8_t* img[kMaxLevels];                          // level l of frame 0 (frame stride: the level-0 pixel count)
  uint32_t h[kMaxLevels], w[kMaxLevels], want[kMaxLevels];
  float scale[kMaxLevels];
};

__global__ __launch_bounds__(256) void copy_rows_kernel(const uint8_t* __restrict__ src, uint32_t stride, uint8_t* dst,
                                                        uint32_t h, uint32_t w, size_t src_fs, size_t dst_fs) {
  TOD_LATENCY_PRIO();   // latency-bound: win issue arbitration against the VALU-saturating matcher
  src += blockIdx.y * src_fs; dst += blockIdx.y * dst_fs;   // frame
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= h * w) return;
  dst[i] = src[(size_t)(i / w) * stride + (i % w)];
}

// 11-bit fixed-point bilinear resize, sample positions (x + 0.5) * sx - 0.5, replicate border
__global__ __launch_bounds__(256) void resize_kernel(const uint8_t* __restrict__ src, uint32_t sh, uint32_t sw, uint8_t* dst,
                                                     uint32_t dh, uint32_t dw, size_t fs) {
  TOD_LATENCY_PRIO();   // latency-bound: win issue arbitration against the VALU-saturating matcher
  src += blockIdx.y * fs; dst += blockIdx.y * fs;
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= dh * dw) return;
  const uint32_t x = i % dw, y = i / dw;
  const float sx = (float)sw / (float)dw, sy = (float)sh / (float)dh;
  const float fy = ((float)y + 0.5f) * sy - 0.5f;
  const int y0 = (int)floorf(fy);
  const int wy = (int)rintf((fy - (float)y0) * 2048.f);
  const int ya = clampi(y0, 0, (int)sh - 1), yb = clampi(y0 + 1, 0, (int)sh - 1);
  const float fx = ((float)x + 0.5f) * sx - 0.5f;
  const int x0 = (int)floorf(fx);
  const int wx = (int)rintf((fx - (float)x0) * 2048.f);
  const int xa = clampi(x0, 0, (int)sw - 1), xb = clampi(x0 + 1, 0, (int)sw - 1);
  const int top = src[ya * sw + xa] * (2048 - wx) + src[ya * sw + xb] * wx;
  const int bot = src[yb * sw + xa] * (2048 - wx) + src[yb * sw + xb] * wx;
  dst[i] = (uint8_t)((top * (2048 - wy) + bot * wy + (1 << 21)) >> 22);
}

// FAST-9/16 score: max over the 16 arcs of 9 contiguous circle pixels, both polarities, of the minimum difference.
// Sliding minima over the (cyclic) circle: windows of 2, 4, 8, then 9.
__device__ __forceinline__ int arc9_max(const int (&d)[16]) {
  int m2[16], m4[16], best = -(1 << 30);
#pragma unroll
  for (int i = 0; i < 16; ++i) m2[i] = min(d[i], d[(i + 1) & 15]);
#pragma unroll
  for (int i = 0; i < 16; ++i) m4[i] = min(m2[i], m2[(i + 2) & 15]);
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int m8 = min(m4[i], m4[(i + 4) & 15]);
    best = max(best, min(m8, d[(i + 8) & 15]));
  }
  return best;
}

__device__ __forceinline__ int fast_score_at(const uint8_t* __restrict__ img, uint32_t h, uint32_t w, int x, int y) {
  if (x < kEdge - 1 || x >= (int)w - kEdge + 1 || y < kEdge - 1 || y >= (int)h - kEdge + 1) return 0;
  const uint8_t* c = img + (size_t)y * w + x;
  const int p = c[0];
  const int W = (int)w;
  int d[16], nd[16];
  d[0] = c[3 * W] - p;       d[1] = c[3 * W + 1] - p;   d[2] = c[2 * W + 2] - p;   d[3] = c[W + 3] - p;
  d[4] = c[3] - p;           d[5] = c[-W + 3] - p;      d[6] = c[-2 * W + 2] - p;  d[7] = c[-3 * W + 1] - p;
  d[8] = c[-3 * W] - p;      d[9] = c[-3 * W - 1] - p;  d[10] = c[-2 * W - 2] - p; d[11] = c[-W - 3] - p;
  d[12] = c[-3] - p;         d[13] = c[W - 3] - p;      d[14] = c[2 * W - 2] - p;  d[15] = c[3 * W - 1] - p;
#pragma unroll
  for (int i = 0; i < 16; ++i) nd[i] = -d[i];
  const int best = max(0, max(arc9_max(d), arc9_max(nd)));
  return best > kFastThr ? best : 0;
}

// FAST score + 3x3 strict non-maximum suppression + compaction in one pass: a block scores a 64 x 16 pixel tile and
// its 1-pixel halo into LDS (the score image never goes to memory), then suppresses and appends. The candidate order
// is fixed later by the ranking kernels, so appends are aggregated per wave (one counter atomic per wave).
constexpr int kNmsTileW = 64, kNmsTileH = 16;
__global__ __launch_bounds__(256) void fast_nms_kernel(LevelTab T, Cand* cand, uint32_t cap, uint32_t* counter, uint32_t* hist,
                                                       const uint8_t* __restrict__ mask, uint32_t H0, uint32_t W0, size_t fs) {
  TOD_LATENCY_PRIO();   // latency-bound: win issue arbitration against the VALU-saturating matcher
  const uint32_t v = blockIdx.z, lvl = v / T.F, f = v - lvl * T.F;
  const uint32_t h = T.h[lvl], w = T.w[lvl];
  if (T.want[lvl] == 0u || blockIdx.x * (uint32_t)kNmsTileW >= w || blockIdx.y * (uint32_t)kNmsTileH >= h) return;   // block-uniform
  const uint8_t* __restrict__ img = T.img[lvl] + f * fs;
  cand += (size_t)v * cap; counter += v * kCtlWords; hist += v * kCtlWords;
  if (mask) mask += f * fs;
  __shared__ int s_score[(kNmsTileH + 2) * (kNmsTileW + 2)];
  __shared__ uint32_t s_hist[256];
  const int x0 = (int)blockIdx.x * kNmsTileW, y0 = (int)blockIdx.y * kNmsTileH;
  s_hist[threadIdx.x] = 0u;
  for (int p = (int)threadIdx.x; p < (kNmsTileH + 2) * (kNmsTileW + 2); p += 256)
    s_score[p] = fast_score_at(img, h, w, x0 - 1 + p % (kNmsTileW + 2), y0 - 1 + p / (kNmsTileW + 2));
  __syncthreads();
  const uint32_t lane = threadIdx.x & 63u;
  const int tx = (int)lane;
#pragma unroll
  for (int j = 0; j < kNmsTileH / 4; ++j) {
    const int ty = (int)(threadIdx.x >> 6) + 4 * j;
    const uint32_t x = (uint32_t)(x0 + tx), y = (uint32_t)(y0 + ty);
    const int* c = &s_score[(ty + 1) * (kNmsTileW + 2) + tx + 1];
    const int s = c[0];
    bool keep = s != 0 && !(x < (uint32_t)kEdge || x >= w - kEdge || y < (uint32_t)kEdge || y >= h - kEdge);
    if (keep) {
      constexpr int R = kNmsTileW + 2;
      keep = c[-R - 1] < s && c[-R] < s && c[-R + 1] < s && c[-1] < s && c[1] < s && c[R - 1] < s && c[R] < s && c[R + 1] < s;
    }
    if (keep && mask) {   // level-0 mask sampled at the nearest pixel (the training cell passes obs.mask to the detector, Trainer.cpp:144-150)
      uint32_t my = (uint32_t)floorf(((float)y + 0.5f) * (float)H0 / (float)h), mx = (uint32_t)floorf(((float)x + 0.5f) * (float)W0 / (float)w);
      my = min(my, H0 - 1u); mx = min(mx, W0 - 1u);
      keep = mask[(size_t)my * W0 + mx] != 0;
    }
    const unsigned long long bal = __builtin_amdgcn_ballot_w64(keep);
    if (bal != 0ull) {
      const uint32_t first = (uint32_t)__ffsll((long long)bal) - 1u;
      uint32_t base = 0;
      if (lane == first) base = atomicAdd(counter, (uint32_t)__popcll(bal));
      base = (uint32_t)__builtin_amdgcn_readlane((int)base, (int)first);
      const uint32_t i = base + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull));
      if (keep && i < cap) {
        Cand cd; cd.x = (int)x; cd.y = (int)y; cd.score = s; cd.harris = 0.f; cand[i] = cd;
        atomicAdd(&s_hist[s & 255], 1u);
      }
    }
  }
  __syncthreads();
  if (s_hist[threadIdx.x]) atomicAdd(&hist[threadIdx.x], s_hist[threadIdx.x]);
}

__device__ __forceinline__ unsigned long long key_of(const Cand& c, bool by_harris) {
  uint32_t hi;
  if (by_harris) {
    const uint32_t b = __float_as_uint(c.harris);
    const uint32_t asc = (b & 0x80000000u) ? ~b : (b | 0x80000000u);   // order-preserving map of float to uint
    hi = ~asc;                                                          // descending
  } else {
    hi = 0xFFFFFFFFu - (uint32_t)c.score;
  }
  return ((unsigned long long)hi << 32) | ((unsigned long long)(uint32_t)c.y << 16) | (uint32_t)c.x;
}

// "keep the 2n best by FAST score" only defines a SET (the Harris ranking re-orders it), so a 256-bin histogram
// gives the score threshold T: everything above T is kept, and of the candidates at exactly T the first
// keep - count(> T) in (y, x) order.
__global__ __launch_bounds__(64) void fast_threshold_kernel(LevelTab T, uint32_t* ctl, uint32_t cand_cap) {
  TOD_LATENCY_PRIO();   // latency-bound: win issue arbitration against the VALU-saturating matcher
  const uint32_t want = T.want[blockIdx.x / T.F], keep = 2u * want;
  if (want == 0u) return;
  ctl += blockIdx.x * kCtlWords;
  const uint32_t l = threadIdx.x;                          // one wave per frame: lane l owns score bins 4 l .. 4 l + 3
  const uint32_t n = min(ctl[W_NCAND], cand_cap);
  uint32_t b[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) b[i] = ctl[W_HIST + 4u * l + i];
  const uint32_t mine = b[0] + b[1] + b[2] + b[3];
  uint32_t suf = mine;                                     // candidates in this lane's bins and all higher ones
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) { const uint32_t t = __shfl_down(suf, d); if (l + d < 64u) suf += t; }
  uint32_t above = suf - mine, thr = 0, need = 0;
  bool hit = false;
  if (n > keep) {
#pragma unroll
    for (int i = 3; i >= 0; --i) {                         // the highest score at which the count reaches `keep`
      if (!hit && above + b[i] >= keep) { hit = true; thr = 4u * l + (uint32_t)i; need = keep - above; }
      above += b[i];
    }
  }
  const unsigned long long bal = __builtin_amdgcn_ballot_w64(hit);
  const uint32_t winner = bal ? 63u - (uint32_t)__clzll((long long)bal) : 0u;
  if (l == winner) {
    ctl[W_NCAND] = n;
    ctl[W_THR] = thr; ctl[W_NEED_EQ] = (n > keep) ? need : 0u;
    ctl[W_NSEL1] = min(n, keep);
    ctl[W_NGT] = 0u; ctl[W_NEQ] = 0u;
    ctl[W_WANT] = want;                                    // what the Harris ranking keeps (rank_tiled_kernel's keep_ptr)
  }
}

__global__ __launch_bounds__(256) void split_kernel(LevelTab T, const Cand* __restrict__ cand, uint32_t* ctl, Cand* sel1,
                                                    Cand* eq, uint32_t cand_cap, uint32_t sel1_cap) {
  TOD_LATENCY_PRIO();   // latency-bound: win issue arbitration against the VALU-saturating matcher
  const uint32_t keep = 2u * T.want[blockIdx.y / T.F];
  if (keep == 0u) return;
  cand += (size_t)blockIdx.y * cand_cap; eq += (size_t)blockIdx.y * cand_cap; sel1 += (size_t)blockIdx.y * sel1_cap;
  ctl += blockIdx.y * kCtlWords;
  const uint32_t n = ctl[W_NCAND];
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (blockIdx.x * 256u >= n) return;                       // block-uniform
  Cand c = {};
  bool gt = false, eqv = false;
  if (i < n) {
    c = cand[i];
    gt = n <= keep || (uint32_t)c.score > ctl[W_THR];
    eqv = !gt && (uint32_t)c.score == ctl[W_THR];
  }
  // one atomic per wave and list (the order inside both lists is free)
  const uint32_t lane = threadIdx.x & 63u;
  const unsigned long long lt = (1ull << lane) - 1ull;
  const unsigned long long bg = __builtin_amdgcn_ballot_w64(gt), be = __builtin_amdgcn_ballot_w64(eqv);
  if (bg != 0ull) {
    const uint32_t first = (uint32_t)__ffsll((long long)bg) - 1u;
    uint32_t base = 0;
    if (lane == first) base = atomicAdd(&ctl[W_NGT], (uint32_t)__popcll(bg));
    base = (uint32_t)__builtin_amdgcn_readlane((int)base, (int)first);
    if (gt) sel1[base + (uint32_t)__popcll(bg & lt)] = c;
  }
  if (be != 0ull) {
    const uint32_t first = (uint32_t)__ffsll((long long)be) - 1u;
    uint32_t base = 0;
    if (lane == first) base = atomicAdd(&ctl[W_NEQ], (uint32_t)__popcll(be));
    base = (uint32_t)__builtin_amdgcn_readlane((int)base, (int)first);
    if (eqv) eq[base + (uint32_t)__popcll(be & lt)] = c;
  }
}

// keep the `keep` best of in[0..n) at out[out_off + rank]: rank by counting against LDS-staged key tiles
// (keys are unique: they contain the position).
constexpr uint32_t kRankTile = 2048;
__global__ __launch_bounds__(256) void rank_tiled_kernel(const Cand* __restrict__ in, const uint32_t* __restrict__ n_ptr,
                                                         const uint32_t* __restrict__ keep_ptr, uint32_t keep_val,
                                                         int by_harris, Cand* out, const uint32_t* __restrict__ off_ptr,
                                                         uint32_t* n_out, uint32_t in_fs, uint32_t out_fs, uint32_t n_out_F) {
  TOD_LATENCY_PRIO();   // latency-bound: win issue arbitration against the VALU-saturating matcher
  __shared__ unsigned long long keys[kRankTile];
  in += (size_t)blockIdx.y * in_fs; out += (size_t)blockIdx.y * out_fs;      // frame: buffers by their strides,
  n_ptr += blockIdx.y * kCtlWords;                                           // control words by kCtlWords
  if (keep_ptr) keep_ptr += blockIdx.y * kCtlWords;
  if (off_ptr) off_ptr += blockIdx.y * kCtlWords;
  // n_out_F != 0: the counts go to row (frame) of n_out, word (level): blockIdx.y = level * n_out_F + frame
  if (n_out) n_out += n_out_F ? (blockIdx.y % n_out_F) * kCtlWords + blockIdx.y / n_out_F : blockIdx.y * kCtlWords;
  const uint32_t n = *n_ptr;
  const uint32_t keep = keep_ptr ? *keep_ptr : keep_val;
  const uint32_t off = off_ptr ? *off_ptr : 0u;
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (n_out && i == 0) *n_out = min(n, keep);
  if (blockIdx.x * 256u >= n) return;                       // block-uniform
  Cand me;
  unsigned long long mine = ~0ull;
  if (i < n) { me = in[i]; mine = key_of(me, by_harris != 0); }
  uint32_t rank = 0;
  for (uint32_t t0 = 0; t0 < n; t0 += kRankTile) {
    const uint32_t tn = min(kRankTile, n - t0);
    __syncthreads();
    for (uint32_t j = threadIdx.x; j < tn; j += 256u) keys[j] = key_of(in[t0 + j], by_harris != 0);
    __syncthreads();
    for (uint32_t j = 0; j < tn; ++j) rank += keys[j] < mine;
  }
  if (i < n && rank < keep) out[off + rank] = me;
}

__global__ __launch_bounds__(256) void harris_kernel(LevelTab T, Cand* cand, const uint32_t* __restrict__ n_ptr, size_t fs,
                                                     uint32_t cand_fs) {
  TOD_LATENCY_PRIO();   // latency-bound: win issue arbitration against the VALU-saturating matcher
  const uint32_t lvl = blockIdx.y / T.F, f = blockIdx.y - lvl * T.F, w = T.w[lvl];
  if (T.want[lvl] == 0u) return;
  const uint8_t* __restrict__ img = T.img[lvl] + f * fs;
  cand += (size_t)blockIdx.y * cand_fs; n_ptr += blockIdx.y * kCtlWords;
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= *n_ptr) return;
  const int x = cand[i].x, y = cand[i].y, W = (int)w;
  int a = 0, b = 0, c = 0;
  for (int dy = -3; dy <= 3; ++dy)
    for (int dx = -3; dx <= 3; ++dx) {
      const uint8_t* p = img + (size_t)(y + dy) * w + (x + dx);
      const int ix = ((int)p[1] - (int)p[-1]) * 2 + ((int)p[-W + 1] - (int)p[-W - 1]) + ((int)p[W + 1] - (int)p[W - 1]);
      const int iy = ((int)p[W] - (int)p[-W]) * 2 + ((int)p[W - 1] - (int)p[-W - 1]) + ((int)p[W + 1] - (int)p[-W + 1]);
      a += ix * ix; b += iy * iy; c += ix * iy;
    }
  const float scale = 1.f / (4.f * 7 * 255.f);
  const float s4 = (scale * scale) * (scale * scale);
  const float fa = (float)a, fb = (float)b, fc = (float)c;
  cand[i].harris = (fa * fb - fc * fc - 0.04f * ((fa + fb) * (fa + fb))) * s4;
}

__constant__ int c_gauss7[7] = {18, 33, 49, 56, 49, 33, 18};

__global__ __launch_bounds__(256) void blur_h_kernel(LevelTab T, uint8_t* dst, size_t fs) {
  TOD_LATENCY_PRIO();   // latency-bound: win issue arbitration against the VALU-saturating matcher
  const uint32_t lvl = blockIdx.y / T.F, f = blockIdx.y - lvl * T.F, h = T.h[lvl], w = T.w[lvl];
  if (T.want[lvl] == 0u) return;
  const uint8_t* __restrict__ src = T.img[lvl] + f * fs;
  dst += blockIdx.y * fs;
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= h * w) return;
  const int x = (int)(i % w), y = (int)(i / w);
  int s = 0;
#pragma unroll
  for (int k = -3; k <= 3; ++k) s += c_gauss7[k + 3] * src[(size_t)y * w + clampi(x + k, 0, (int)w - 1)];
  dst[i] = (uint8_t)((s + 128) >> 8);
}
__global__ __launch_bounds__(256) void blur_v_kernel(LevelTab T, const uint8_t* __restrict__ src, uint8_t* dst, size_t fs) {
  TOD_LATENCY_PRIO();   // latency-bound: win issue arbitration against the VALU-saturating matcher
  const uint32_t lvl = blockIdx.y / T.F, h = T.h[lvl], w = T.w[lvl];
  if (T.want[lvl] == 0u) return;
  src += blockIdx.y * fs; dst += blockIdx.y * fs;
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= h * w) return;
  const int x = (int)(i % w), y = (int)(i / w);
  int s = 0;
#pragma unroll
  for (int k = -3; k <= 3; ++k) s += c_gauss7[k + 3] * src[(size_t)clampi(y + k, 0, (int)h - 1) * w + x];
  dst[i] = (uint8_t)((s + 128) >> 8);
}

struct DescribeArgs {
  const uint8_t* img; const uint8_t* blur; uint32_t w;   // img, w, level, scale: filled in from the level table by the kernel
  const Cand* sel;                              // the levels' selections (virtual-frame slices)
  const uint32_t* level_counts; uint32_t level; // per frame: word l = keypoints of level l; output base = sum of the earlier levels' counts
  float scale; uint32_t cap;
  const int8_t* pattern; int umax[kHalfPatch + 2];
  float* kp_xy; float* kp_aux; uint8_t* desc;
  size_t fs; uint32_t sel_fs;                   // frame strides: images; selection (outputs: cap rows, controls: kCtlWords)
};

// one wave per keypoint: integer moments over the radius-15 disc (lane = row), then 4 tests per lane
__global__ __launch_bounds__(256) void describe_kernel(LevelTab T, DescribeArgs A) {
  TOD_LATENCY_PRIO();   // latency-bound: win issue arbitration against the VALU-saturating matcher
  {
    const uint32_t v = blockIdx.y, lvl = v / T.F, f = v - lvl * T.F;
    if (T.want[lvl] == 0u) return;
    A.img = T.img[lvl] + f * A.fs; A.w = T.w[lvl]; A.level = lvl; A.scale = T.scale[lvl];
    A.blur += v * A.fs; A.sel += (size_t)v * A.sel_fs;
    A.level_counts += f * kCtlWords;
    A.kp_xy += (size_t)f * A.cap * 2; A.kp_aux += (size_t)f * A.cap * 4; A.desc += (size_t)f * A.cap * 32;
  }
  const uint32_t i = blockIdx.x * 4u + (threadIdx.x >> 6), l = threadIdx.x & 63u;
  if (i >= A.level_counts[A.level]) return;
  uint32_t base = 0;
  for (uint32_t j = 0; j < A.level; ++j) base += A.level_counts[j];
  const uint32_t o = base + i;
  if (o >= A.cap) return;
  const Cand c = A.sel[i];
  const int x = c.x, y = c.y, W = (int)A.w;
  int m10 = 0, m01 = 0;
  if (l < 31u) {
    const int v = (int)l - kHalfPatch;
    const int d = v == 0 ? kHalfPatch : A.umax[v < 0 ? -v : v];
    int rs = 0;
    for (int u = -d; u <= d; ++u) {
      const int px = A.img[(size_t)(y + v) * W + x + u];
      m10 += u * px;
      rs += px;
    }
    m01 = v * rs;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) { m10 += __shfl_xor(m10, off); m01 += __shfl_xor(m01, off); }
  const float fm10 = (float)m10, fm01 = (float)m01;
  const float nrm = sqrtf(fm10 * fm10 + fm01 * fm01);
  const float ca = nrm > 0.f ? fm10 / nrm : 1.f, sa = nrm > 0.f ? fm01 / nrm : 0.f;
  uint32_t nib = 0;
#pragma unroll
  for (int b = 0; b < 4; ++b) {
    const int8_t* pp = A.pattern + 4 * (4 * (int)l + b);
    const int x0 = (int)rintf((float)pp[0] * ca - (float)pp[1] * sa), y0 = (int)rintf((float)pp[0] * sa + (float)pp[1] * ca);
    const int x1 = (int)rintf((float)pp[2] * ca - (float)pp[3] * sa), y1 = (int)rintf((float)pp[2] * sa + (float)pp[3] * ca);
    const int t0 = A.blur[(size_t)(y + y0) * W + (x + x0)], t1 = A.blur[(size_t)(y + y1) * W + (x + x1)];
    nib |= (uint32_t)(t0 < t1) << b;
  }
  const uint32_t hi = __shfl_down(nib, 1);
  if ((l & 1u) == 0u) A.desc[(size_t)o * 32 + (l >> 1)] = (uint8_t)(nib | (hi << 4));
  if (l == 0) {
    float ang = atan2f(fm01, fm10) * 57.29577951308232f;
    if (ang < 0.f) ang += 360.f;
    A.kp_xy[2 * o] = (float)x * A.scale; A.kp_xy[2 * o + 1] = (float)y * A.scale;
    A.kp_aux[4 * o] = 31.f * A.scale; A.kp_aux[4 * o + 1] = ang; A.kp_aux[4 * o + 2] = c.harris; A.kp_aux[4 * o + 3] = (float)A.level;
  }
}

// the graph writes keypoints into workspace buffers; this kernel hands the first min(total, cap) of them to the caller
__global__ __launch_bounds__(256) void copy_out_kernel(const uint32_t* __restrict__ level_counts, uint32_t n_levels, uint32_t cap,
                                                       const float* __restrict__ s_xy, const float* __restrict__ s_aux,
                                                       const uint32_t* __restrict__ s_desc, float* __restrict__ d_xy,
                                                       float* __restrict__ d_aux, uint32_t* __restrict__ d_desc,
                                                       uint32_t* __restrict__ totals) {
  const uint32_t f = blockIdx.y;
  level_counts += f * kCtlWords;
  s_xy += (size_t)f * cap * 2; s_aux += (size_t)f * cap * 4; s_desc += (size_t)f * cap * 8;
  d_xy += (size_t)f * cap * 2; d_aux += (size_t)f * cap * 4; d_desc += (size_t)f * cap * 8;
  uint32_t total = 0;
  for (uint32_t l = 0; l < n_levels; ++l) total += level_counts[l];
  const uint32_t n = min(total, cap);
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;      // one dword each: 2 + 4 + 8 dwords per keypoint
  if (i == 0) totals[f] = total;
  if (i < 2u * n) d_xy[i] = s_xy[i];
  if (i < 4u * n) d_aux[i] = s_aux[i];
  if (i < 8u * n) d_desc[i] = s_desc[i];
}

struct OrbWs {
  DevBuf pyr, blur, tmp, score, cand, eq, sel1, sel2, small, pattern, in_img, kp_xy, kp_aux, desc, o_xy, o_aux, o_desc, maskbuf;
  HostBuf h_out, h_kp;                                     // h_kp: the host form's keypoints, written by copy_out_kernel itself
  bool pattern_is_default = false;
  // the per-frame launch sequence is static for a given geometry: captured once per context, replayed. It reads
  // and writes workspace buffers only (image and mask are copied in, keypoints copied out), so the caller's
  // pointers may change from call to call without a re-capture.
  hipGraph_t graph = nullptr;
  hipGraphExec_t graph_exec = nullptr;
  struct Key { uint32_t H, W, n_features, n_levels, cap, F; float sf; const void *kp, *aux, *desc, *img0, *mask; } key = {};
};

OrbWs* ows_of(todhip_ctx* ctx) {
  if (!ctx->orb_ws) ctx->orb_ws = new OrbWs();
  return reinterpret_cast<OrbWs*>(ctx->orb_ws);
}

void features_per_level(uint32_t n_features, uint32_t n_levels, float scale_factor, uint32_t* out) {
  const float factor = 1.0f / scale_factor;
  float n_desired = (float)n_features * (1.f - factor) / (1.f - powf(factor, (float)n_levels));
  int sum = 0;
  for (uint32_t l = 0; l + 1 < n_levels; ++l) {
    out[l] = (uint32_t)rintf(n_desired);
    sum += (int)out[l];
    n_desired *= factor;
  }
  const int rest = (int)n_features - sum;
  out[n_levels - 1] = rest > 0 ? (uint32_t)rest : 0u;
}

// d_gray: F frames of H x W u8 on the device (row stride `stride`, frame stride gray_fs bytes). Results stay on the
// device: frame f's keypoints at row f * cap of the three outputs; n_out[f] is read back.
int orb_device(todhip_ctx* ctx, const uint8_t* d_gray, size_t gray_fs, const uint8_t* d_mask, uint32_t F, uint32_t H, uint32_t W,
               uint32_t stride, uint32_t n_features, uint32_t n_levels, float scale_factor, const int8_t* pattern,
               float* d_kp_xy, float* d_kp_aux, uint8_t* d_desc, uint32_t cap, uint32_t* n_out) {
  if (n_levels == 0 || n_levels > (uint32_t)kMaxLevels || scale_factor <= 1.f || H < 8 || W < 8 || n_features == 0 || F == 0 ||
      F > 65535u)
    return TODHIP_EINVAL;
  if (cap == 0) return TODHIP_ECAPACITY;
  OrbWs* ws = ows_of(ctx);
  hipStream_t st = ctx->stream;
  const size_t px = (size_t)H * W;
  const uint32_t cand_cap = (uint32_t)(px / 4 + 64);
  const uint32_t sel1_cap = 2u * n_features + 16u, sel2_cap = n_features + 16u;
  const size_t V = (size_t)n_levels * F;                    // virtual frames: (level, frame) pairs, level major
  if (V > 65535u) return TODHIP_EINVAL;
  TOD_HIP(ws->pyr.reserve(V * px));
  TOD_HIP(ws->blur.reserve(V * px)); TOD_HIP(ws->tmp.reserve(V * px));
  TOD_HIP(ws->cand.reserve(V * cand_cap * sizeof(Cand)));
  TOD_HIP(ws->eq.reserve(V * cand_cap * sizeof(Cand)));
  TOD_HIP(ws->sel1.reserve(V * sel1_cap * sizeof(Cand)));
  TOD_HIP(ws->sel2.reserve(V * sel2_cap * sizeof(Cand)));
  // control words: one row per virtual frame, then one row per frame for the level counts (word 8 + l), then the totals
  TOD_HIP(ws->small.reserve((V + F + 1) * kCtlWords * sizeof(uint32_t)));
  TOD_HIP(ws->pattern.reserve(1024));
  TOD_HIP(ws->h_out.reserve((size_t)F * sizeof(uint32_t) + 64));
  TOD_HIP(ws->o_xy.reserve((size_t)F * cap * 8)); TOD_HIP(ws->o_aux.reserve((size_t)F * cap * 16));
  TOD_HIP(ws->o_desc.reserve((size_t)F * cap * 32));
  if (d_mask) TOD_HIP(ws->maskbuf.reserve(F * px));
  float* const u_kp_xy = d_kp_xy; float* const u_kp_aux = d_kp_aux; uint8_t* const u_desc = d_desc;   // the caller's
  d_kp_xy = ws->o_xy.as<float>(); d_kp_aux = ws->o_aux.as<float>(); d_desc = ws->o_desc.as<uint8_t>();
  int8_t hpat[1024];
  if (pattern) { std::memcpy(hpat, pattern, 1024); ws->pattern_is_default = false; }
  if (pattern || !ws->pattern_is_default) {
    if (!pattern) { default_pattern(hpat); ws->pattern_is_default = true; }
    TOD_HIP(hipMemcpyAsync(ws->pattern.p, hpat, 1024, hipMemcpyHostToDevice, st));
    TOD_HIP(hipStreamSynchronize(st));                   // hpat lives on this stack frame
  }
  uint32_t per_level[kMaxLevels];
  features_per_level(n_features, n_levels, scale_factor, per_level);
  uint32_t* d_small = ws->small.as<uint32_t>();           // per virtual frame: the level's control words (W_*)
  uint32_t* d_cnt = d_small + V * kCtlWords;              // per frame: [8 + l] = keypoints of level l
  uint32_t* d_totals = d_cnt + (size_t)F * kCtlWords;
  TOD_HIP(hipMemsetAsync(d_small, 0, (V + F + 1) * kCtlWords * sizeof(uint32_t), st));
  const uint32_t px_blocks = (uint32_t)((px + 255) / 256);
  if (stride == W && (F == 1 || gray_fs == px))            // densely packed input: one device-to-device copy
    TOD_HIP(hipMemcpyAsync(ws->pyr.p, d_gray, (size_t)F * px, hipMemcpyDeviceToDevice, st));
  else
    hipLaunchKernelGGL(copy_rows_kernel, dim3(px_blocks, F), dim3(256), 0, st, d_gray, stride, ws->pyr.as<uint8_t>(), H, W,
                       gray_fs, px);
  if (d_mask) {
    hipLaunchKernelGGL(copy_rows_kernel, dim3(px_blocks, F), dim3(256), 0, st, d_mask, W, ws->maskbuf.as<uint8_t>(), H, W, px, px);
    d_mask = ws->maskbuf.as<uint8_t>();
  }
  OrbWs::Key key = {H, W, n_features, n_levels, cap, F, scale_factor, d_kp_xy, d_kp_aux, d_desc, ws->pyr.p, d_mask};
  static const bool use_graph = getenv("TODHIP_ORB_NO_GRAPH") == nullptr;
  const bool reuse = use_graph && ws->graph_exec && std::memcmp(&key, &ws->key, sizeof(key)) == 0;
  if (!reuse) {
    if (ws->graph_exec) { (void)hipGraphExecDestroy(ws->graph_exec); ws->graph_exec = nullptr; }
    if (ws->graph) { (void)hipGraphDestroy(ws->graph); ws->graph = nullptr; }
    // thread-local capture: other host threads keep using the runtime while this one records
    if (use_graph) TOD_HIP(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    {
      // the pyramid first (level l from level l - 1), then every kernel once for all levels and frames
      LevelTab T;
      std::memset(&T, 0, sizeof(T));
      T.n_levels = n_levels; T.F = F;
      uint32_t ph = H, pw = W, want_max = 0;
      for (uint32_t lvl = 0; lvl < n_levels; ++lvl) {
        float scale = 1.f;
        for (uint32_t i = 0; i < lvl; ++i) scale = scale * scale_factor;
        const uint32_t w = (uint32_t)rintf((float)W / scale), h = (uint32_t)rintf((float)H / scale);
        uint8_t* img = ws->pyr.as<uint8_t>() + (size_t)lvl * F * px;
        if (lvl > 0) {
          hipLaunchKernelGGL(resize_kernel, dim3((h * w + 255u) / 256u, F), dim3(256), 0, st, img - (size_t)F * px, ph, pw, img, h, w, px);
          ph = h; pw = w;
        }
        T.img[lvl] = img; T.h[lvl] = h; T.w[lvl] = w; T.scale[lvl] = scale;
        T.want[lvl] = (h <= 2u * kEdge || w <= 2u * kEdge) ? 0u : per_level[lvl];     // too small: the level count stays 0
        want_max = std::max(want_max, T.want[lvl]);
      }
      if (want_max > 0) {
        const uint32_t Vg = (uint32_t)V;
        hipLaunchKernelGGL(fast_nms_kernel, dim3((W + kNmsTileW - 1) / kNmsTileW, (H + kNmsTileH - 1) / kNmsTileH, Vg), dim3(256), 0, st,
                           T, ws->cand.as<Cand>(), cand_cap, d_small, d_small + W_HIST, d_mask, H, W, px);
        hipLaunchKernelGGL(fast_threshold_kernel, dim3(Vg), dim3(64), 0, st, T, d_small, cand_cap);
        hipLaunchKernelGGL(split_kernel, dim3((cand_cap + 255u) / 256u, Vg), dim3(256), 0, st, T, ws->cand.as<Cand>(), d_small,
                           ws->sel1.as<Cand>(), ws->eq.as<Cand>(), cand_cap, sel1_cap);
        // ties at the threshold: the first need_eq of them in (y, x) order, placed behind the count(> T) sure ones
        hipLaunchKernelGGL(rank_tiled_kernel, dim3((cand_cap + 255u) / 256u, Vg), dim3(256), 0, st, ws->eq.as<Cand>(), d_small + W_NEQ,
                           d_small + W_NEED_EQ, 0u, 0, ws->sel1.as<Cand>(), d_small + W_NGT, (uint32_t*)nullptr, cand_cap, sel1_cap, 0u);
        hipLaunchKernelGGL(harris_kernel, dim3((2u * want_max + 255u) / 256u, Vg), dim3(256), 0, st, T, ws->sel1.as<Cand>(),
                           d_small + W_NSEL1, px, sel1_cap);
        hipLaunchKernelGGL(rank_tiled_kernel, dim3((2u * want_max + 255u) / 256u, Vg), dim3(256), 0, st, ws->sel1.as<Cand>(),
                           d_small + W_NSEL1, d_small + W_WANT, 0u, 1, ws->sel2.as<Cand>(), (const uint32_t*)nullptr, d_cnt + 8,
                           sel1_cap, sel2_cap, F);
        hipLaunchKernelGGL(blur_h_kernel, dim3(px_blocks, Vg), dim3(256), 0, st, T, ws->tmp.as<uint8_t>(), px);
        hipLaunchKernelGGL(blur_v_kernel, dim3(px_blocks, Vg), dim3(256), 0, st, T, ws->tmp.as<uint8_t>(), ws->blur.as<uint8_t>(), px);
        DescribeArgs D;
        std::memset(&D, 0, sizeof(D));
        disc_umax(D.umax);
        D.blur = ws->blur.as<uint8_t>(); D.sel = ws->sel2.as<Cand>();
        D.level_counts = d_cnt + 8; D.cap = cap; D.pattern = ws->pattern.as<int8_t>();
        D.kp_xy = d_kp_xy; D.kp_aux = d_kp_aux; D.desc = d_desc; D.fs = px; D.sel_fs = sel2_cap;
        hipLaunchKernelGGL(describe_kernel, dim3((want_max + 3u) / 4u, Vg), dim3(256), 0, st, T, D);
      }
    }
    if (use_graph) {
      hipGraph_t g = nullptr;
      const hipError_t ee = hipStreamEndCapture(st, &g);
      if (ee != hipSuccess || !g) { ctx->last_hip_error = (int)ee; return TODHIP_EHIP; }
      ws->graph = g;
      TOD_HIP(hipGraphInstantiate(&ws->graph_exec, ws->graph, nullptr, nullptr, 0));
      ws->key = key;
    }
  }
  if (use_graph) TOD_HIP(hipGraphLaunch(ws->graph_exec, st));
  hipLaunchKernelGGL(copy_out_kernel, dim3((8u * cap + 255u) / 256u, F), dim3(256), 0, st, d_cnt + 8, n_levels, cap, d_kp_xy, d_kp_aux,
                     reinterpret_cast<const uint32_t*>(d_desc), u_kp_xy, u_kp_aux, reinterpret_cast<uint32_t*>(u_desc), d_totals);
  TOD_HIP(hipGetLastError());
  uint32_t* h_totals = ws->h_out.as<uint32_t>();
  TOD_HIP(hipMemcpyAsync(h_totals, d_totals, (size_t)F * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
  TOD_HIP(hipStreamSynchronize(st));
  for (uint32_t f = 0; f < F; ++f) n_out[f] = std::min(h_totals[f], cap);
  return TODHIP_OK;
}

}  // namespace

// used by train.hip: ORB with a level-0 mask on device-resident inputs
int tod_orb_device(todhip_ctx* ctx, const uint8_t* d_gray, const uint8_t* d_mask, uint32_t H, uint32_t W, uint32_t stride,
                   uint32_t n_features, uint32_t n_levels, float scale_factor, const int8_t* pattern, float* d_kp_xy,
                   float* d_kp_aux, uint8_t* d_desc, uint32_t cap, uint32_t* n_out) {
  return orb_device(ctx, d_gray, 0, d_mask, 1, H, W, stride, n_features, n_levels, scale_factor, pattern, d_kp_xy, d_kp_aux, d_desc,
                    cap, n_out);
}

void tod_orb_ws_free(todhip_ctx* ctx) {
  if (!ctx->orb_ws) return;
  OrbWs* ws = reinterpret_cast<OrbWs*>(ctx->orb_ws);
  DevBuf* bufs[] = {&ws->pyr, &ws->blur, &ws->tmp, &ws->score, &ws->cand, &ws->eq, &ws->sel1, &ws->sel2,
                    &ws->small, &ws->pattern, &ws->in_img, &ws->kp_xy, &ws->kp_aux, &ws->desc, &ws->o_xy, &ws->o_aux, &ws->o_desc,
                    &ws->maskbuf};
  for (DevBuf* b : bufs) b->release();
  if (ws->graph_exec) (void)hipGraphExecDestroy(ws->graph_exec);
  if (ws->graph) (void)hipGraphDestroy(ws->graph);
  ws->h_out.release(); ws->h_kp.release();
  delete ws;
  ctx->orb_ws = nullptr;
}

extern "C" {

int todhip_orb_device(todhip_ctx* ctx, const void* d_gray, uint32_t H, uint32_t W, uint32_t stride, uint32_t n_features,
                      uint32_t n_levels, float scale_factor, const int8_t* pattern, void* d_kp_xy, void* d_kp_aux,
                      void* d_desc, uint32_t* n_out) {
  if (!ctx || !d_gray || !d_kp_xy || !d_kp_aux || !d_desc || !n_out || stride < W) return TODHIP_EINVAL;
  TOD_HIP(hipSetDevice(ctx->device));
  const uint32_t cap = *n_out;
  *n_out = 0;
  return orb_device(ctx, reinterpret_cast<const uint8_t*>(d_gray), 0, nullptr, 1, H, W, stride, n_features, n_levels, scale_factor,
                    pattern, reinterpret_cast<float*>(d_kp_xy), reinterpret_cast<float*>(d_kp_aux),
                    reinterpret_cast<uint8_t*>(d_desc), cap, n_out);
}

int todhip_orb_batch_device(todhip_ctx* ctx, const void* d_gray, uint32_t n_frames, uint64_t frame_stride, uint32_t H, uint32_t W,
                            uint32_t stride, uint32_t n_features, uint32_t n_levels, float scale_factor, const int8_t* pattern,
                            void* d_kp_xy, void* d_kp_aux, void* d_desc, uint32_t cap, uint32_t* n_out) {
  if (!ctx || !d_gray || !d_kp_xy || !d_kp_aux || !d_desc || !n_out || stride < W || n_frames == 0) return TODHIP_EINVAL;
  if (n_frames > 1 && frame_stride < (uint64_t)H * stride) return TODHIP_EINVAL;
  TOD_HIP(hipSetDevice(ctx->device));
  for (uint32_t f = 0; f < n_frames; ++f) n_out[f] = 0;
  return orb_device(ctx, reinterpret_cast<const uint8_t*>(d_gray), (size_t)frame_stride, nullptr, n_frames, H, W, stride, n_features,
                    n_levels, scale_factor, pattern, reinterpret_cast<float*>(d_kp_xy), reinterpret_cast<float*>(d_kp_aux),
                    reinterpret_cast<uint8_t*>(d_desc), cap, n_out);
}

int todhip_orb_masked(todhip_ctx* ctx, const uint8_t* gray, const uint8_t* mask, uint32_t H, uint32_t W, uint32_t stride,
                      uint32_t n_features, uint32_t n_levels, float scale_factor, const int8_t* pattern, float* kp_xy, float* kp_aux,
                      uint8_t* desc, uint32_t* n_out) {
  if (!ctx || !gray || !kp_xy || !kp_aux || !desc || !n_out || stride < W) return TODHIP_EINVAL;
  TOD_HIP(hipSetDevice(ctx->device));
  OrbWs* ws = ows_of(ctx);
  const uint32_t cap = *n_out;
  *n_out = 0;
  if (cap == 0) return TODHIP_ECAPACITY;
  const size_t img_bytes = (size_t)H * stride;
  TOD_HIP(ws->in_img.reserve(mask ? 2 * img_bytes : img_bytes));       // the mask rides behind the image
  // the outputs land in pinned host memory, written by the last kernel itself (56 B per keypoint over PCIe): no device-to-host
  // copies and no second synchronization behind orb_device's own
  TOD_HIP(ws->h_kp.reserve((size_t)cap * 56 + 64));
  float* const p_xy = ws->h_kp.as<float>(); float* const p_aux = p_xy + (size_t)cap * 2;
  uint8_t* const p_desc = reinterpret_cast<uint8_t*>(p_aux + (size_t)cap * 4);
  TOD_HIP(hipMemcpyAsync(ws->in_img.p, gray, img_bytes, hipMemcpyHostToDevice, ctx->stream));
  const uint8_t* d_mask = nullptr;
  if (mask) {
    // orb_device reads the mask with row pitch W: repack rows when the caller's stride is larger
    uint8_t* dm = ws->in_img.as<uint8_t>() + img_bytes;
    TOD_HIP(hipMemcpy2DAsync(dm, W, mask, stride, W, H, hipMemcpyHostToDevice, ctx->stream));
    d_mask = dm;
  }
  uint32_t n = 0;
  const int rc = orb_device(ctx, ws->in_img.as<uint8_t>(), 0, d_mask, 1, H, W, stride, n_features, n_levels, scale_factor, pattern,
                            p_xy, p_aux, p_desc, cap, &n);           // (synchronizes the stream before it returns)
  if (rc != TODHIP_OK) return rc;
  if (n) {
    std::memcpy(kp_xy, p_xy, (size_t)n * 8); std::memcpy(kp_aux, p_aux, (size_t)n * 16); std::memcpy(desc, p_desc, (size_t)n * 32);
  }
  *n_out = n;
  return TODHIP_OK;
}

int todhip_orb(todhip_ctx* ctx, const uint8_t* gray, uint32_t H, uint32_t W, uint32_t stride, uint32_t n_features,
               uint32_t n_levels, float scale_factor, const int8_t* pattern, float* kp_xy, float* kp_aux, uint8_t* desc,
               uint32_t* n_out) {
  return todhip_orb_masked(ctx, gray, nullptr, H, W, stride, n_features, n_levels, scale_factor, pattern, kp_xy, kp_aux, desc, n_out);
}

}  // extern "C"
