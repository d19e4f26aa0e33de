// the per-observation arithmetic of the reference's training cell on the GPU
// (src/training/Trainer.cpp:121-187, src/training/training.cpp:57-195): ORB on the masked view, depth rescale
// (equal-size case), validateKeyPoints, depthTo3dSparse, cameraToWorld, mergePoints. A model is accumulated in HBM
// observation by observation and read back once; its rows (32-byte descriptors + object-frame points) are exactly
// what todhip_db_load ingests, so DBs built here match this repo's ORB.
#include <cfloat>
#include <cstring>
#include <new>

#include "ctx.h"

struct todhip_model {
  DevBuf desc, pts, kp_xy, kp_aux, kp_desc, img, mask, er_tmp, er, depth, flags, offs, small;
  uint32_t cap = 0;
};

namespace {

// erosion by the 3x3 element, 4 iterations == minimum over the 9x9 window restricted to the image
// (cv::erode's border value is +inf), done separably: rows, then columns (training.cpp:69-71)
__global__ __launch_bounds__(256) void erode_rows_kernel(const uint8_t* __restrict__ src, uint32_t H, uint32_t W, uint8_t* dst) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= H * W) return;
  const int x = (int)(i % W), y = (int)(i / W);
  int all = 1;
  for (int dx = -4; dx <= 4; ++dx) { const int xx = x + dx; if (xx >= 0 && xx < (int)W && !src[(size_t)y * W + xx]) all = 0; }
  dst[i] = all ? 255 : 0;
}
__global__ __launch_bounds__(256) void erode_cols_kernel(const uint8_t* __restrict__ src, uint32_t H, uint32_t W, uint8_t* dst) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= H * W) return;
  const int x = (int)(i % W), y = (int)(i / W);
  int all = 1;
  for (int dy = -4; dy <= 4; ++dy) { const int yy = y + dy; if (yy >= 0 && yy < (int)H && !src[(size_t)yy * W + x]) all = 0; }
  dst[i] = all ? 255 : 0;
}

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

struct Cam { float fx, fy, cx, cy, R[9], T[3]; };

// rescale_depth (Trainer.cpp:62-81; the detection side does the same in ecto_opencv's RescaledRegisteredDepth,
// detector.py:26,62): cv::rescaleDepth to float metres, then -- when the depth image is not of the image size --
// a resize into the top int(dH * factor) rows of an H x W image, the rest NaN. The reference's call
// cv::resize(depth, subregion, subregion.size(), CV_INTER_NN) (:78) hands CV_INTER_NN (== 0) to the `fx` parameter,
// so what it EXECUTES is cv::resize's default, bilinear interpolation; `nearest` selects what its comment (:77)
// intends. Both follow cv::resize's float path (third-party arithmetic, recalled; parity unpinned):
//   nearest : source = floor(x * (1 / (dst / src))) clamped to the last column / row, in double
//   bilinear: f = (float)((x + 0.5) * (src / dst) - 0.5), s = floor(f), f -= s; s < 0 -> s = 0, f = 0; s + 1 >= src
//             width -> the single tap S[src - 1] * 1; rows: the two row indices are clamped, the weights are kept;
//             out = (S00 * (1 - fx) + S01 * fx) * (1 - fy) + (S10 * (1 - fx) + S11 * fx) * fy in float, no fma
//             (a NaN tap poisons the pixel even under a zero weight, as in the reference).
//   exact 2x shrink: cv::resize turns INTER_LINEAR into its INTER_AREA fast path: (S00 + S01 + S10 + S11) * 0.25f.
__device__ __forceinline__ float depth_metres(const void* src, int is_u16, size_t i) {
  if (!is_u16) return reinterpret_cast<const float*>(src)[i];
  const uint16_t d = reinterpret_cast<const uint16_t*>(src)[i];
  return d ? (float)d * 0.001f : __builtin_nanf("");                 // cv::rescaleDepth
}
__global__ __launch_bounds__(256) void rescale_depth_kernel(const void* __restrict__ src, int is_u16, uint32_t dH, uint32_t dW,
                                                            uint32_t sub_rows, int mode, double sx_scale, double sy_scale,
                                                            float* __restrict__ dst, uint32_t H, uint32_t W) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= H * W) return;
  const uint32_t x = i % W, y = i / W;
  float z = __builtin_nanf("");
  if (y < sub_rows) {
    if (mode == 0) {                                                 // sizes equal: conversion only (:68-71)
      z = depth_metres(src, is_u16, i);
    } else if (mode == 1) {                                          // nearest
      const uint32_t sx = min((uint32_t)floor((double)x * sx_scale), dW - 1u), sy = min((uint32_t)floor((double)y * sy_scale), dH - 1u);
      z = depth_metres(src, is_u16, (size_t)sy * dW + sx);
    } else if (mode == 3) {                                          // exact 2x shrink
      const size_t o = (size_t)(2u * y) * dW + 2u * x;
      z = (depth_metres(src, is_u16, o) + depth_metres(src, is_u16, o + 1) + depth_metres(src, is_u16, o + dW) +
           depth_metres(src, is_u16, o + dW + 1)) * 0.25f;
    } else {                                                         // bilinear
      float fx = (float)(((double)x + 0.5) * sx_scale - 0.5), fy = (float)(((double)y + 0.5) * sy_scale - 0.5);
      int sx = (int)floorf(fx), sy = (int)floorf(fy);
      fx -= (float)sx; fy -= (float)sy;
      if (sx < 0) { sx = 0; fx = 0.f; }
      const bool one_tap = sx + 1 >= (int)dW;
      if (one_tap) sx = (int)dW - 1;
      const int sy0 = clampi(sy, 0, (int)dH - 1), sy1 = clampi(sy + 1, 0, (int)dH - 1);
      float h0, h1;
      if (one_tap) {
        h0 = depth_metres(src, is_u16, (size_t)sy0 * dW + sx) * 1.f;
        h1 = depth_metres(src, is_u16, (size_t)sy1 * dW + sx) * 1.f;
      } else {
        const float a0 = 1.f - fx, a1 = fx;
        h0 = depth_metres(src, is_u16, (size_t)sy0 * dW + sx) * a0 + depth_metres(src, is_u16, (size_t)sy0 * dW + sx + 1) * a1;
        h1 = depth_metres(src, is_u16, (size_t)sy1 * dW + sx) * a0 + depth_metres(src, is_u16, (size_t)sy1 * dW + sx + 1) * a1;
      }
      z = h0 * (1.f - fy) + h1 * fy;
    }
  }
  dst[i] = z;
}

// host part of rescale_depth: the geometry of the resize. Returns false when cv::Mat::rowRange / cv::resize would throw.
bool rescale_geometry(uint32_t dH, uint32_t dW, uint32_t H, uint32_t W, int nearest, uint32_t* sub_rows, int* mode, double* sx,
                      double* sy) {
  if (dH == H && dW == W) { *sub_rows = H; *mode = 0; *sx = 1.0; *sy = 1.0; return true; }
  const float factor = (float)W / (float)dW;                     // :73
  const int rows = (int)((float)dH * factor);                    // rowRange(0, dsize.height * factor), :76
  if (rows <= 0 || (uint32_t)rows > H) return false;
  *sub_rows = (uint32_t)rows;
  const double inv_x = (double)W / (double)dW, inv_y = (double)rows / (double)dH;   // cv::resize: inv_scale = dst / src
  *mode = nearest ? 1 : ((dW == 2u * W && dH == 2u * (uint32_t)rows) ? 3 : 2);
  *sx = 1.0 / inv_x; *sy = 1.0 / inv_y;                          // ifx (nearest) == scale_x (bilinear) == 1 / inv_scale
  return true;
}


// validateKeyPoints (training.cpp:57-145) + depthTo3dSparse + cameraToWorld (:175-195) for one keypoint per thread.
// roundWithinBounds clamps to [0, width] in the reference (:53-55), which can index one column past the image;
// here the clamp is to width-1 / height-1.
__global__ __launch_bounds__(256) void validate_kernel(const float* __restrict__ kp_xy, const uint32_t* __restrict__ n_ptr,
                                                       const uint8_t* __restrict__ er, const void* __restrict__ depth,
                                                       int depth_is_u16, uint32_t H, uint32_t W, Cam cam, uint32_t* flags,
                                                       float* pts_tmp) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= *n_ptr) return;
  const float px = kp_xy[2 * i], py = kp_xy[2 * i + 1];
  int x = clampi((int)rintf(px), 0, (int)W - 1), y = clampi((int)rintf(py), 0, (int)H - 1);
  bool good = er[(size_t)y * W + x] != 0;
  if (!good) {
    float best = FLT_MAX;
    int bx = x, by = y;
    for (int ii = clampi(x - 2, 0, (int)W - 1); ii <= clampi(x + 2, 0, (int)W - 1); ++ii)
      for (int jj = clampi(y - 2, 0, (int)H - 1); jj <= clampi(y + 2, 0, (int)H - 1); ++jj)
        if (er[(size_t)jj * W + ii]) {
          const float d = ((float)ii - px) * ((float)ii - px) + ((float)jj - py) * ((float)jj - py);
          if (d < best) { best = d; bx = ii; by = jj; good = true; }
        }
    x = bx; y = by;
  }
  float z = 0.f;
  if (good) {
    if (depth_is_u16) {                                   // cv::rescaleDepth: millimetres -> metres, 0 -> NaN
      const uint16_t d = reinterpret_cast<const uint16_t*>(depth)[(size_t)y * W + x];
      z = d == 0 ? __builtin_nanf("") : (float)d * 0.001f;
    } else {
      z = reinterpret_cast<const float*>(depth)[(size_t)y * W + x];
    }
    if (!(z == z) || z == FLT_MAX || z == -FLT_MAX || z == FLT_MIN) good = false;      // cv::isValidDepth(float)
  }
  flags[i] = good ? 1u : 0u;
  if (good) {
    const float p[3] = {((float)x - cam.cx) * z / cam.fx, ((float)y - cam.cy) * z / cam.fy, z};
    const float q[3] = {p[0] - cam.T[0], p[1] - cam.T[1], p[2] - cam.T[2]};
    for (int c = 0; c < 3; ++c) {                         // (p - T) * R, double accumulation like cv::gemm on CV_32F
      double s = 0;
      for (int k = 0; k < 3; ++k) s += (double)q[k] * (double)cam.R[3 * k + c];
      pts_tmp[3 * i + c] = (float)s;
    }
  }
}

// stable compaction of the accepted keypoints behind the rows the model already has (mergePoints, :147-173)
__global__ __launch_bounds__(1024) void append_kernel(const uint32_t* __restrict__ flags, const uint32_t* __restrict__ n_ptr,
                                                      const uint8_t* __restrict__ kp_desc, const float* __restrict__ pts_tmp,
                                                      uint32_t* counters /* [0] model rows, [1] added by this call */,
                                                      uint32_t cap, uint8_t* desc, float* pts) {
  __shared__ uint32_t part[1024];
  const uint32_t n = *n_ptr, tid = threadIdx.x;
  const uint32_t chunk = (n + 1023u) / 1024u;
  const uint32_t lo = min(n, tid * chunk), hi = min(n, lo + chunk);
  uint32_t s = 0;
  for (uint32_t i = lo; i < hi; ++i) s += flags[i];
  part[tid] = s;
  __syncthreads();
  __shared__ uint32_t base, total;
  if (tid == 0) {
    uint32_t acc = 0;
    for (uint32_t i = 0; i < 1024u; ++i) { const uint32_t c = part[i]; part[i] = acc; acc += c; }
    base = counters[0];
    total = acc;
  }
  __syncthreads();
  uint32_t o = base + part[tid];
  for (uint32_t i = lo; i < hi; ++i) {
    if (!flags[i]) continue;
    if (o < cap) {
      for (int b = 0; b < 8; ++b) reinterpret_cast<uint32_t*>(desc)[(size_t)o * 8 + b] = reinterpret_cast<const uint32_t*>(kp_desc)[(size_t)i * 8 + b];
      pts[3 * (size_t)o] = pts_tmp[3 * i]; pts[3 * (size_t)o + 1] = pts_tmp[3 * i + 1]; pts[3 * (size_t)o + 2] = pts_tmp[3 * i + 2];
    }
    ++o;
  }
  __syncthreads();
  if (tid == 0) { counters[1] = min(total, cap > base ? cap - base : 0u); counters[0] = min(base + total, cap); }
}

}  // namespace

extern "C" {

int todhip_model_begin(todhip_ctx* ctx, uint32_t capacity_rows, todhip_model** out) {
  if (!ctx || !out || capacity_rows == 0) return TODHIP_EINVAL;
  TOD_HIP(hipSetDevice(ctx->device));
  todhip_model* m = new (std::nothrow) todhip_model();
  if (!m) return TODHIP_ENOMEM;
  m->cap = capacity_rows;
  hipError_t e = m->desc.reserve((size_t)capacity_rows * 32);
  if (e == hipSuccess) e = m->pts.reserve((size_t)capacity_rows * 12);
  if (e == hipSuccess) e = m->small.reserve(64 * sizeof(uint32_t));
  if (e == hipSuccess) e = hipMemsetAsync(m->small.p, 0, 64 * sizeof(uint32_t), ctx->stream);
  if (e != hipSuccess) { ctx->last_hip_error = (int)e; delete m; return TODHIP_EHIP; }
  *out = m;
  return TODHIP_OK;
}

void todhip_model_free(todhip_ctx* ctx, todhip_model* m) {
  if (!m) return;
  if (ctx) { (void)hipSetDevice(ctx->device); (void)hipStreamSynchronize(ctx->stream); }
  DevBuf* bufs[] = {&m->desc, &m->pts, &m->kp_xy, &m->kp_aux, &m->kp_desc, &m->img, &m->mask, &m->er_tmp, &m->er, &m->depth,
                    &m->flags, &m->offs, &m->small};
  for (DevBuf* b : bufs) b->release();
  delete m;
}

// Trainer::process for one observation (Trainer.cpp:136-171). depth has the image size (rescale_depth's equal-size
// branch, :63-72): float metres or uint16 millimetres. K9/R9 row-major, T3. *n_added = rows appended to the model.
int todhip_model_add_observation(todhip_ctx* ctx, todhip_model* m, const uint8_t* gray, const uint8_t* mask, const void* depth,
                                 int depth_is_u16, uint32_t H, uint32_t W, const float* K9, const float* R9, const float* T3,
                                 uint32_t n_features, uint32_t n_levels, float scale_factor, const int8_t* pattern,
                                 uint32_t* n_added) {
  if (!ctx || !m || !gray || !mask || !depth || !K9 || !R9 || !T3 || H < 8 || W < 8 || n_features == 0) return TODHIP_EINVAL;
  TOD_HIP(hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  const size_t px = (size_t)H * W, dbytes = px * (depth_is_u16 ? 2 : 4);
  TOD_HIP(m->img.reserve(px)); TOD_HIP(m->mask.reserve(px)); TOD_HIP(m->er_tmp.reserve(px)); TOD_HIP(m->er.reserve(px));
  TOD_HIP(m->depth.reserve(dbytes));
  TOD_HIP(m->kp_xy.reserve((size_t)n_features * 8)); TOD_HIP(m->kp_aux.reserve((size_t)n_features * 16));
  TOD_HIP(m->kp_desc.reserve((size_t)n_features * 32)); TOD_HIP(m->flags.reserve((size_t)n_features * 4));
  TOD_HIP(m->offs.reserve((size_t)n_features * 12));
  TOD_HIP(hipMemcpyAsync(m->img.p, gray, px, hipMemcpyHostToDevice, st));
  TOD_HIP(hipMemcpyAsync(m->mask.p, mask, px, hipMemcpyHostToDevice, st));
  TOD_HIP(hipMemcpyAsync(m->depth.p, depth, dbytes, hipMemcpyHostToDevice, st));
  uint32_t n_kp = 0;
  int rc = tod_orb_device(ctx, m->img.as<uint8_t>(), m->mask.as<uint8_t>(), H, W, W, n_features, n_levels, scale_factor, pattern,
                          m->kp_xy.as<float>(), m->kp_aux.as<float>(), m->kp_desc.as<uint8_t>(), n_features, &n_kp);
  if (rc != TODHIP_OK) return rc;
  if (n_added) *n_added = 0;
  if (n_kp == 0) return TODHIP_OK;
  uint32_t* d_small = m->small.as<uint32_t>();
  TOD_HIP(hipMemcpyAsync(d_small + 2, &n_kp, sizeof(uint32_t), hipMemcpyHostToDevice, st));
  hipLaunchKernelGGL(erode_rows_kernel, dim3((uint32_t)((px + 255) / 256)), dim3(256), 0, st, m->mask.as<uint8_t>(), H, W, m->er_tmp.as<uint8_t>());
  hipLaunchKernelGGL(erode_cols_kernel, dim3((uint32_t)((px + 255) / 256)), dim3(256), 0, st, m->er_tmp.as<uint8_t>(), H, W, m->er.as<uint8_t>());
  Cam cam;
  cam.fx = K9[0]; cam.fy = K9[4]; cam.cx = K9[2]; cam.cy = K9[5];
  std::memcpy(cam.R, R9, sizeof(cam.R)); std::memcpy(cam.T, T3, sizeof(cam.T));
  hipLaunchKernelGGL(validate_kernel, dim3((n_kp + 255u) / 256u), dim3(256), 0, st, m->kp_xy.as<float>(), d_small + 2,
                     m->er.as<uint8_t>(), m->depth.p, depth_is_u16, H, W, cam, m->flags.as<uint32_t>(), m->offs.as<float>());
  hipLaunchKernelGGL(append_kernel, dim3(1), dim3(1024), 0, st, m->flags.as<uint32_t>(), d_small + 2, m->kp_desc.as<uint8_t>(),
                     m->offs.as<float>(), d_small, m->cap, m->desc.as<uint8_t>(), m->pts.as<float>());
  TOD_HIP(hipGetLastError());
  uint32_t h[2] = {0, 0};
  TOD_HIP(hipMemcpyAsync(h, d_small, sizeof(h), hipMemcpyDeviceToHost, st));
  TOD_HIP(hipStreamSynchronize(st));                     // also keeps gray/mask/depth/n_kp alive until the copies are done
  if (n_added) *n_added = h[1];
  return TODHIP_OK;
}

// rescale_depth (Trainer.cpp:62-81) on device-resident images: d_depth_in dH x dW (float metres or uint16 mm),
// d_depth_out H x W float metres. Asynchronous on the context's stream.
int todhip_rescale_depth_device(todhip_ctx* ctx, const void* d_depth_in, int depth_is_u16, uint32_t dH, uint32_t dW,
                                void* d_depth_out, uint32_t H, uint32_t W, int nearest) {
  if (!ctx || !d_depth_in || !d_depth_out || !dH || !dW || !H || !W || (uint64_t)H * W > 0xFFFFFFFFull) return TODHIP_EINVAL;
  uint32_t sub_rows; int mode; double sx, sy;
  if (!rescale_geometry(dH, dW, H, W, nearest, &sub_rows, &mode, &sx, &sy)) return TODHIP_EINVAL;
  TOD_HIP(hipSetDevice(ctx->device));
  hipLaunchKernelGGL(rescale_depth_kernel, dim3((uint32_t)(((size_t)H * W + 255) / 256)), dim3(256), 0, ctx->stream, d_depth_in,
                     depth_is_u16, dH, dW, sub_rows, mode, sx, sy, static_cast<float*>(d_depth_out), H, W);
  TOD_HIP(hipGetLastError());
  return TODHIP_OK;
}

// Host-buffer form (what Trainer::process does per observation before validateKeyPoints, :142-143).
int todhip_rescale_depth(todhip_ctx* ctx, const void* depth_in, int depth_is_u16, uint32_t dH, uint32_t dW, float* depth_out,
                         uint32_t H, uint32_t W, int nearest) {
  if (!ctx || !depth_in || !depth_out || !dH || !dW || !H || !W) return TODHIP_EINVAL;
  TOD_HIP(hipSetDevice(ctx->device));
  const size_t in_bytes = (size_t)dH * dW * (depth_is_u16 ? 2 : 4), out_bytes = (size_t)H * W * 4;
  DevBuf in, out;
  int rc = TODHIP_OK;
  if (in.reserve(in_bytes) != hipSuccess || out.reserve(out_bytes) != hipSuccess) rc = TODHIP_EHIP;
  if (rc == TODHIP_OK && hipMemcpyAsync(in.p, depth_in, in_bytes, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) rc = TODHIP_EHIP;
  if (rc == TODHIP_OK) rc = todhip_rescale_depth_device(ctx, in.p, depth_is_u16, dH, dW, out.p, H, W, nearest);
  if (rc == TODHIP_OK && hipMemcpyAsync(depth_out, out.p, out_bytes, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess) rc = TODHIP_EHIP;
  (void)hipStreamSynchronize(ctx->stream);
  in.release(); out.release();
  return rc;
}

// mergePoints (training.cpp:147-173) + ModelFiller (ModelFiller.cpp:20-26): the stacked descriptors and points.
// the model where it is: device pointers of its descriptors (n x 32) and points (n x 3 f32), valid until todhip_model_free --
// todhip_db_load_device takes them as a todhip_object, so a freshly trained model reaches the matcher without visiting the host
int todhip_model_device(todhip_ctx* ctx, todhip_model* m, const void** d_desc, const void** d_pts_xyz, uint32_t* n) {
  if (!ctx || !m || !n) return TODHIP_EINVAL;
  TOD_HIP(hipSetDevice(ctx->device));
  uint32_t rows = 0;
  TOD_HIP(hipMemcpyAsync(&rows, m->small.p, sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
  TOD_HIP(hipStreamSynchronize(ctx->stream));                 // (every observation's kernels have run: the buffers are complete)
  *n = rows;
  if (d_desc) *d_desc = m->desc.p;
  if (d_pts_xyz) *d_pts_xyz = m->pts.p;
  return TODHIP_OK;
}

int todhip_model_finish(todhip_ctx* ctx, todhip_model* m, uint8_t* desc, float* pts, uint32_t* n) {
  if (!ctx || !m || !n) return TODHIP_EINVAL;
  TOD_HIP(hipSetDevice(ctx->device));
  uint32_t rows = 0;
  TOD_HIP(hipMemcpyAsync(&rows, m->small.p, sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
  TOD_HIP(hipStreamSynchronize(ctx->stream));
  const uint32_t cap = *n;
  *n = rows;
  if (rows > cap) return TODHIP_ECAPACITY;
  if (rows && (!desc || !pts)) return TODHIP_EINVAL;
  if (rows) {
    TOD_HIP(hipMemcpyAsync(desc, m->desc.p, (size_t)rows * 32, hipMemcpyDeviceToHost, ctx->stream));
    TOD_HIP(hipMemcpyAsync(pts, m->pts.p, (size_t)rows * 12, hipMemcpyDeviceToHost, ctx->stream));
    TOD_HIP(hipStreamSynchronize(ctx->stream));
  }
  return TODHIP_OK;
}

}  // extern "C"
