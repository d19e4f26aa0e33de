/*
 * oracle/train_oracle.c -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement of the per-observation arithmetic of the reference's training cell
 * (src/training/Trainer.cpp:121-187 and src/training/training.cpp:57-195), SURVEY 8(f) row N2:
 *   rescale_depth        Trainer.cpp:63-81   (cv::rescaleDepth to float metres, uint16 0 -> NaN; resize when sizes differ)
 *   validateKeyPoints    training.cpp:57-145 (mask eroded 4x with the 3x3 element == 9x9 minimum inside the image;
 *                                             nearest masked pixel in a +-2 window; depth validity)
 *   depthTo3dSparse      Trainer.cpp:168     (third-party cv::depthTo3dSparse: x = (u-cx) z / fx, y = (v-cy) z / fy)
 *   cameraToWorld        training.cpp:175-195  ((p - T) * R, row vector times matrix; cv::gemm accumulates in double)
 *   mergePoints          training.cpp:147-173  (concatenation in observation order)
 * PARITY UNPINNED: the reference holds no fixture for this path, and cv::erode / rescaleDepth / depthTo3dSparse /
 * isValidDepth are third-party (recalled). Two reference quirks are not reproduced: roundWithinBounds clamps to
 * [0, width] and can index one past the last column (training.cpp:53-55,77) -- clamped to width-1 here. When depth
 * and image sizes differ the reference's cv::resize call passes CV_INTER_NN as `fx` (Trainer.cpp:78), i.e. it
 * interpolates bilinearly: train_rescale_depth() below restates that (and the nearest-neighbour intent as an option).
 * The keypoints come from this repo's ORB restatement with a mask (level-i mask = nearest-neighbour sample of the
 * level-0 mask; a candidate needs mask != 0), because cv::ORB is third-party.
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

void train_erode4(const uint8_t* mask, uint32_t H, uint32_t W, uint8_t* out) {
  for (uint32_t y = 0; y < H; ++y)
    for (uint32_t x = 0; x < W; ++x) {
      int all = 1;
      for (int dy = -4; dy <= 4 && all; ++dy)
        for (int dx = -4; dx <= 4; ++dx) {
          int yy = (int)y + dy, xx = (int)x + dx;
          if (yy < 0 || xx < 0 || yy >= (int)H || xx >= (int)W) continue;   /* border value of an erosion is +inf */
          if (!mask[yy * W + xx]) { all = 0; break; }
        }
      out[y * W + x] = all ? 255 : 0;
    }
}

/* one observation. kp_xy: n_kp keypoints (level-0 pixels), desc: n_kp x 32. depth_m: H x W float metres (NaN = none).
 * Outputs appended at out_desc/out_pts (world frame); returns the number of accepted keypoints. */
uint32_t train_observation(const float* kp_xy, const uint8_t* desc, uint32_t n_kp, const uint8_t* mask, const float* depth_m,
                           uint32_t H, uint32_t W, const float* K9, const float* R9, const float* T3, uint8_t* out_desc,
                           float* out_pts, uint32_t* out_src /* optional: index of the source keypoint */) {
  uint8_t* er = (uint8_t*)malloc((size_t)H * W);
  train_erode4(mask, H, W, er);
  const float fx = K9[0], fy = K9[4], cx = K9[2], cy = K9[5];
  uint32_t n = 0;
  for (uint32_t i = 0; i < n_kp; ++i) {
    const float px = kp_xy[2 * i], py = kp_xy[2 * i + 1];
    int x = clampi((int)lrintf(px), 0, (int)W - 1), y = clampi((int)lrintf(py), 0, (int)H - 1);
    int good = 0;
    if (er[y * W + x]) good = 1;
    else {
      float best = FLT_MAX;
      int bx = x, by = y;
      for (int ii = clampi(x - 2, 0, (int)W - 1); ii <= clampi(x + 2, 0, (int)W - 1); ++ii)
        for (int jj = clampi(y - 2, 0, (int)H - 1); jj <= clampi(y + 2, 0, (int)H - 1); ++jj)
          if (er[jj * W + ii]) {
            float d = ((float)ii - px) * ((float)ii - px) + ((float)jj - py) * ((float)jj - py);
            if (d < best) { best = d; bx = ii; by = jj; good = 1; }
          }
      x = bx; y = by;
    }
    if (!good) continue;
    const float z = depth_m[y * W + x];
    if (!(z == z) || z == FLT_MAX || z == -FLT_MAX || z == FLT_MIN) continue;          /* cv::isValidDepth(float) */
    /* depthTo3dSparse at the integer pixel (x, y), then cameraToWorld: (p - T) * R */
    const float p[3] = {((float)x - cx) * z / fx, ((float)y - cy) * z / fy, z};
    const float q[3] = {p[0] - T3[0], p[1] - T3[1], p[2] - T3[2]};
    for (int c = 0; c < 3; ++c) {
      double s = 0;
      for (int k = 0; k < 3; ++k) s += (double)q[k] * (double)R9[3 * k + c];
      out_pts[3 * n + c] = (float)s;
    }
    memcpy(out_desc + (size_t)n * 32, desc + (size_t)i * 32, 32);
    if (out_src) out_src[n] = i;
    ++n;
  }
  free(er);
  return n;
}

/* rescale_depth (src/training/Trainer.cpp:62-81). in: dH x dW, float metres (is_u16 = 0) or uint16 millimetres
 * (cv::rescaleDepth: value / 1000, 0 -> NaN). out: H x W float metres. Equal sizes: conversion only (:68-71).
 * Otherwise (:73-80) a resize into the top (int)(dH * factor) rows, factor = (float)W / dW, NaN below. The reference
 * calls cv::resize(depth, subregion, subregion.size(), CV_INTER_NN): the constant lands in the `fx` parameter, so
 * the interpolation that RUNS is cv::resize's default, bilinear (nearest = 0 here); nearest = 1 is what the
 * comment at :77 intends. cv::resize's float paths, third-party and recalled -- PARITY UNPINNED:
 *   nearest : src = min(floor(x * ifx), src_w - 1), ifx = 1 / (dst_w / src_w), double
 *   bilinear: fx = (float)((x + .5) * scale - .5); sx = floor(fx); fx -= sx; sx < 0 -> sx = 0, fx = 0;
 *             sx + 1 >= src_w -> single tap S[src_w - 1] * 1; row indices sy, sy + 1 clamped, weights kept;
 *             horizontal pass then vertical pass, float, no fma
 *   2x shrink in both directions: INTER_LINEAR is replaced by the INTER_AREA fast path (S00 + S01 + S10 + S11) * .25f
 * Returns 0, or -1 where cv::Mat::rowRange / cv::resize would throw. */
static float depth_metres(const void* in, int is_u16, size_t i) {
  if (!is_u16) return ((const float*)in)[i];
  const uint16_t d = ((const uint16_t*)in)[i];
  return d ? (float)d * 0.001f : NAN;
}
int train_rescale_depth(const void* in, int is_u16, uint32_t dH, uint32_t dW, float* out, uint32_t H, uint32_t W, int nearest) {
  if (dH == H && dW == W) {
    for (size_t i = 0; i < (size_t)H * W; ++i) out[i] = depth_metres(in, is_u16, i);
    return 0;
  }
  const float factor = (float)W / (float)dW;
  const int rows = (int)((float)dH * factor);
  if (rows <= 0 || (uint32_t)rows > H) return -1;
  const double scale_x = 1.0 / ((double)W / (double)dW), scale_y = 1.0 / ((double)rows / (double)dH);
  const int area2 = !nearest && dW == 2u * W && dH == 2u * (uint32_t)rows;
  for (uint32_t y = 0; y < H; ++y)
    for (uint32_t x = 0; x < W; ++x) {
      volatile float z = NAN;
      if (y < (uint32_t)rows) {
        if (nearest) {
          uint32_t sx = (uint32_t)floor((double)x * scale_x), sy = (uint32_t)floor((double)y * scale_y);
          if (sx > dW - 1) sx = dW - 1;
          if (sy > dH - 1) sy = dH - 1;
          z = depth_metres(in, is_u16, (size_t)sy * dW + sx);
        } else if (area2) {
          const size_t o = (size_t)(2u * y) * dW + 2u * x;
          volatile float s = depth_metres(in, is_u16, o) + depth_metres(in, is_u16, o + 1);
          s = s + depth_metres(in, is_u16, o + dW);
          s = s + depth_metres(in, is_u16, o + dW + 1);
          z = s * 0.25f;
        } else {
          float fx = (float)(((double)x + 0.5) * scale_x - 0.5), fy = (float)(((double)y + 0.5) * scale_y - 0.5);
          int sx = (int)floorf(fx), sy = (int)floorf(fy);
          fx -= (float)sx; fy -= (float)sy;
          if (sx < 0) { sx = 0; fx = 0.f; }
          const int one_tap = sx + 1 >= (int)dW;
          if (one_tap) sx = (int)dW - 1;
          int sy0 = sy < 0 ? 0 : (sy > (int)dH - 1 ? (int)dH - 1 : sy);
          int sy1 = sy + 1 < 0 ? 0 : (sy + 1 > (int)dH - 1 ? (int)dH - 1 : sy + 1);
          volatile float h0, h1, p0, p1;
          if (one_tap) {
            h0 = depth_metres(in, is_u16, (size_t)sy0 * dW + sx);
            h1 = depth_metres(in, is_u16, (size_t)sy1 * dW + sx);
          } else {
            const float a0 = 1.f - fx, a1 = fx;
            p0 = depth_metres(in, is_u16, (size_t)sy0 * dW + sx) * a0; p1 = depth_metres(in, is_u16, (size_t)sy0 * dW + sx + 1) * a1;
            h0 = p0 + p1;
            p0 = depth_metres(in, is_u16, (size_t)sy1 * dW + sx) * a0; p1 = depth_metres(in, is_u16, (size_t)sy1 * dW + sx + 1) * a1;
            h1 = p0 + p1;
          }
          p0 = h0 * (1.f - fy); p1 = h1 * fy;
          z = p0 + p1;
        }
      }
      out[(size_t)y * W + x] = z;
    }
  return 0;
}
