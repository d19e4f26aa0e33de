/*
 * oracle/orb_oracle.c -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement of stage A (ORB keypoints + rBRIEF descriptors) of the detection hot path.
 * PARITY UNPINNED: in the reference this stage is third-party code that is not in its tree --
 * `ecto_opencv.features2d.FeatureDescriptor` -> cv::ORB (python/object_recognition_tod/detector.py:10,27;
 * src/training/Trainer.cpp:144-150) -- and OpenCV is not in this image, nor is its learned 256-pair
 * pattern table. This file restates the PUBLISHED algorithm (Rublee et al., "ORB: an efficient
 * alternative to SIFT or SURF", ICCV 2011; FAST-9/16 of Rosten & Drummond; Harris corner measure)
 * with the structure of OpenCV 2.4's ORB as recalled, and pins every free choice so that the HIP
 * implementation can be checked bit for bit against it:
 *   - pyramid: level i is a fixed-point (11 bit) bilinear resize of level i-1 to
 *     round(W / f^i) x round(H / f^i), sample positions (x + 0.5) * sx - 0.5, replicate border
 *   - FAST-9/16: score = max over the 16 arcs of 9 contiguous circle pixels and both polarities of the
 *     minimum |difference|; a corner needs score > threshold (20); 3x3 strict non-maximum suppression;
 *     candidates closer than 31 px to the border are dropped
 *   - per level keep 2 n_i by (FAST score desc, y asc, x asc), then n_i by (Harris desc, y asc, x asc);
 *     n_i is OpenCV's geometric split of n_features over the levels
 *   - Harris: 7x7 block of the 3x3 Sobel-like integer gradients, k = 0.04, float evaluation
 *     a*c - b*b - k*(a+c)^2 scaled by (1/(4*7*255))^4
 *   - orientation: intensity centroid over the radius-15 disc (integer moments); the rotation uses
 *     cos = m10/|m|, sin = m01/|m| directly (no angle quantisation); angle in degrees is output only
 *   - descriptor: 7x7 Gaussian (sigma 2) blur with 8-bit fixed-point separable weights, then 256
 *     intensity tests at the rotated, rint-rounded pattern positions; bit i of byte j = test 8j+i
 *   - the test pattern is an argument (256 x 4 int8); orb_default_pattern() generates the built-in one
 *     (seeded xorshift, isotropic, inside radius 13 so every rotation stays inside the 31x31 patch)
 * Output order: level ascending, then rank of the Harris selection.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORB_EDGE 31
#define ORB_HALF_PATCH 15
#define ORB_FAST_THR 20
#define ORB_HARRIS_BLOCK 7

typedef struct { int x, y; int score; float harris; } orb_cand;

void orb_default_pattern(int8_t* pat /* 256*4 */) {
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

void orb_level_size(uint32_t H, uint32_t W, float scale_factor, uint32_t level, uint32_t* h, uint32_t* w, float* scale) {
  float sc = 1.f;
  for (uint32_t i = 0; i < level; ++i) sc = sc * scale_factor;
  *scale = sc;
  *w = (uint32_t)rintf((float)W / sc);
  *h = (uint32_t)rintf((float)H / sc);
}

void orb_features_per_level(uint32_t n_features, uint32_t n_levels, float scale_factor, uint32_t* out) {
  /* OpenCV 2.4 ORB: geometric distribution, the last level takes what is left */
  float factor = 1.0f / scale_factor;
  float n_desired = (float)n_features * (1.f - factor) / (1.f - powf(factor, (float)n_levels));
  int sum = 0;
  for (uint32_t l = 0; l + 1 < n_levels; ++l) {
    out[l] = (uint32_t)rintf(n_desired);
    sum += (int)out[l];
    n_desired *= factor;
  }
  int rest = (int)n_features - sum;
  out[n_levels - 1] = rest > 0 ? (uint32_t)rest : 0u;
}

static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

void orb_resize(const uint8_t* src, uint32_t sh, uint32_t sw, uint8_t* dst, uint32_t dh, uint32_t dw) {
  const float sx = (float)sw / (float)dw, sy = (float)sh / (float)dh;
  for (uint32_t y = 0; y < dh; ++y) {
    float fy = ((float)y + 0.5f) * sy - 0.5f;
    int y0 = (int)floorf(fy);
    int wy = (int)rintf((fy - (float)y0) * 2048.f);
    int ya = clampi(y0, 0, (int)sh - 1), yb = clampi(y0 + 1, 0, (int)sh - 1);
    for (uint32_t x = 0; x < dw; ++x) {
      float fx = ((float)x + 0.5f) * sx - 0.5f;
      int x0 = (int)floorf(fx);
      int wx = (int)rintf((fx - (float)x0) * 2048.f);
      int xa = clampi(x0, 0, (int)sw - 1), xb = clampi(x0 + 1, 0, (int)sw - 1);
      int top = src[ya * sw + xa] * (2048 - wx) + src[ya * sw + xb] * wx;
      int bot = src[yb * sw + xa] * (2048 - wx) + src[yb * sw + xb] * wx;
      int v = (top * (2048 - wy) + bot * wy + (1 << 21)) >> 22;
      dst[y * dw + x] = (uint8_t)v;
    }
  }
}

static const int kCircle[16][2] = {{0, 3}, {1, 3}, {2, 2}, {3, 1}, {3, 0}, {3, -1}, {2, -2}, {1, -3},
                                   {0, -3}, {-1, -3}, {-2, -2}, {-3, -1}, {-3, 0}, {-3, 1}, {-2, 2}, {-1, 3}};

int orb_fast_score(const uint8_t* img, uint32_t w, int x, int y) {
  int p = img[y * w + x], d[16], best = 0;
  for (int i = 0; i < 16; ++i) d[i] = (int)img[(y + kCircle[i][1]) * w + (x + kCircle[i][0])] - p;
  for (int pol = 0; pol < 2; ++pol) {
    for (int s = 0; s < 16; ++s) {
      int m = 1 << 30;
      for (int j = 0; j < 9; ++j) { int v = d[(s + j) & 15]; v = pol ? -v : v; if (v < m) m = v; }
      if (m > best) best = m;
    }
  }
  return best;
}

static const int kGauss7[7] = {18, 33, 49, 56, 49, 33, 18};   /* round(256 * exp(-x^2/8) / sum), sums to 256 */

void orb_blur(const uint8_t* src, uint32_t h, uint32_t w, uint8_t* dst) {
  uint8_t* tmp = (uint8_t*)malloc((size_t)h * w);
  for (uint32_t y = 0; y < h; ++y)
    for (uint32_t x = 0; x < w; ++x) {
      int s = 0;
      for (int k = -3; k <= 3; ++k) s += kGauss7[k + 3] * src[y * w + clampi((int)x + k, 0, (int)w - 1)];
      tmp[y * w + x] = (uint8_t)((s + 128) >> 8);
    }
  for (uint32_t y = 0; y < h; ++y)
    for (uint32_t x = 0; x < w; ++x) {
      int s = 0;
      for (int k = -3; k <= 3; ++k) s += kGauss7[k + 3] * tmp[clampi((int)y + k, 0, (int)h - 1) * w + x];
      dst[y * w + x] = (uint8_t)((s + 128) >> 8);
    }
  free(tmp);
}

float orb_harris(const uint8_t* img, uint32_t w, int x, int y) {
  int a = 0, b = 0, c = 0;
  const int r = ORB_HARRIS_BLOCK / 2;
  for (int dy = -r; dy <= r; ++dy)
    for (int dx = -r; dx <= r; ++dx) {
      const uint8_t* p = img + (y + dy) * w + (x + dx);
      int ix = ((int)p[1] - (int)p[-1]) * 2 + ((int)p[-(int)w + 1] - (int)p[-(int)w - 1]) + ((int)p[w + 1] - (int)p[w - 1]);
      int iy = ((int)p[w] - (int)p[-(int)w]) * 2 + ((int)p[w - 1] - (int)p[-(int)w - 1]) + ((int)p[w + 1] - (int)p[-(int)w + 1]);
      a += ix * ix; b += iy * iy; c += ix * iy;
    }
  const float scale = 1.f / (4.f * ORB_HARRIS_BLOCK * 255.f);
  const float s4 = (scale * scale) * (scale * scale);
  const float fa = (float)a, fb = (float)b, fc = (float)c;
  return (fa * fb - fc * fc - 0.04f * ((fa + fb) * (fa + fb))) * s4;
}

static int cmp_score(const void* pa, const void* pb) {
  const orb_cand* a = (const orb_cand*)pa; const orb_cand* b = (const orb_cand*)pb;
  if (a->score != b->score) return a->score > b->score ? -1 : 1;
  if (a->y != b->y) return a->y < b->y ? -1 : 1;
  return a->x < b->x ? -1 : (a->x > b->x ? 1 : 0);
}
static int cmp_harris(const void* pa, const void* pb) {
  const orb_cand* a = (const orb_cand*)pa; const orb_cand* b = (const orb_cand*)pb;
  if (a->harris != b->harris) return a->harris > b->harris ? -1 : 1;
  if (a->y != b->y) return a->y < b->y ? -1 : 1;
  return a->x < b->x ? -1 : (a->x > b->x ? 1 : 0);
}

/* umax[v] = half-width of row v of the radius-15 disc (OpenCV's construction) */
static void disc_umax(int* umax) {
  const int hp = ORB_HALF_PATCH;
  int vmax = (int)floor(hp * sqrt(2.0) / 2 + 1), vmin = (int)ceil(hp * sqrt(2.0) / 2);
  for (int v = 0; v <= vmax; ++v) umax[v] = (int)rint(sqrt((double)hp * hp - (double)v * v));
  for (int v = hp, v0 = 0; v >= vmin; --v) {
    while (umax[v0] == umax[v0 + 1]) ++v0;
    umax[v] = v0;
    ++v0;
  }
}

/* kp_xy: (x,y) level-0 pixels; kp_aux: (size, angle_deg, response, octave); desc: 32 bytes each.
 * kp_lvl_xy (optional): integer coordinates inside the level. Returns the number of keypoints. */
uint32_t orb_detect_masked(const uint8_t* gray, const uint8_t* mask, uint32_t H, uint32_t W, uint32_t stride,
                           uint32_t n_features, uint32_t n_levels, float scale_factor, const int8_t* pattern, uint32_t cap,
                           float* kp_xy, float* kp_aux, uint8_t* desc, int32_t* kp_lvl_xy);

uint32_t orb_detect(const uint8_t* gray, uint32_t H, uint32_t W, uint32_t stride, uint32_t n_features, uint32_t n_levels,
                    float scale_factor, const int8_t* pattern, uint32_t cap, float* kp_xy, float* kp_aux, uint8_t* desc,
                    int32_t* kp_lvl_xy) {
  return orb_detect_masked(gray, 0, H, W, stride, n_features, n_levels, scale_factor, pattern, cap, kp_xy, kp_aux, desc,
                           kp_lvl_xy);
}

/* mask (optional, H x W u8, row stride W): a candidate at level pixel (x, y) needs
 * mask[min(H-1, floor((y + 0.5) * H / h))][min(W-1, floor((x + 0.5) * W / w))] != 0 */
uint32_t orb_detect_masked(const uint8_t* gray, const uint8_t* mask, uint32_t H, uint32_t W, uint32_t stride,
                           uint32_t n_features, uint32_t n_levels, float scale_factor, const int8_t* pattern, uint32_t cap,
                           float* kp_xy, float* kp_aux, uint8_t* desc, int32_t* kp_lvl_xy) {
  int8_t defpat[1024];
  if (!pattern) { orb_default_pattern(defpat); pattern = defpat; }
  int umax[ORB_HALF_PATCH + 2];
  disc_umax(umax);
  uint32_t per_level[64];
  if (n_levels > 64) n_levels = 64;
  orb_features_per_level(n_features, n_levels, scale_factor, per_level);
  uint8_t* prev = (uint8_t*)malloc((size_t)H * W);
  for (uint32_t y = 0; y < H; ++y) memcpy(prev + (size_t)y * W, gray + (size_t)y * stride, W);
  uint32_t ph = H, pw = W, n_out = 0;
  for (uint32_t lvl = 0; lvl < n_levels; ++lvl) {
    uint32_t h, w; float scale;
    orb_level_size(H, W, scale_factor, lvl, &h, &w, &scale);
    uint8_t* img = prev;
    if (lvl > 0) {
      img = (uint8_t*)malloc((size_t)h * w);
      orb_resize(prev, ph, pw, img, h, w);
      free(prev);
      prev = img; ph = h; pw = w;
    }
    if (h <= 2 * ORB_EDGE || w <= 2 * ORB_EDGE) continue;
    /* FAST score map + 3x3 strict non-maximum suppression */
    int* score = (int*)calloc((size_t)h * w, sizeof(int));
    for (uint32_t y = ORB_EDGE - 1; y < h - ORB_EDGE + 1; ++y)
      for (uint32_t x = ORB_EDGE - 1; x < w - ORB_EDGE + 1; ++x) {
        int s = orb_fast_score(img, w, (int)x, (int)y);
        score[y * w + x] = s > ORB_FAST_THR ? s : 0;
      }
    orb_cand* cand = (orb_cand*)malloc(sizeof(orb_cand) * (size_t)h * w / 4 + 64);
    uint32_t nc = 0;
    for (uint32_t y = ORB_EDGE; y < h - ORB_EDGE; ++y)
      for (uint32_t x = ORB_EDGE; x < w - ORB_EDGE; ++x) {
        int s = score[y * w + x];
        if (s == 0) continue;
        int ismax = 1;
        for (int dy = -1; dy <= 1 && ismax; ++dy)
          for (int dx = -1; dx <= 1; ++dx)
            if ((dx || dy) && score[(y + dy) * w + (x + dx)] >= s) { ismax = 0; break; }
        if (ismax && mask) {
          uint32_t my = (uint32_t)floorf(((float)y + 0.5f) * (float)H / (float)h), mx = (uint32_t)floorf(((float)x + 0.5f) * (float)W / (float)w);
          if (my > H - 1) my = H - 1;
          if (mx > W - 1) mx = W - 1;
          if (!mask[(size_t)my * W + mx]) ismax = 0;
        }
        if (ismax) { cand[nc].x = (int)x; cand[nc].y = (int)y; cand[nc].score = s; cand[nc].harris = 0.f; ++nc; }
      }
    free(score);
    uint32_t want = per_level[lvl];
    qsort(cand, nc, sizeof(orb_cand), cmp_score);
    if (nc > 2 * want) nc = 2 * want;
    for (uint32_t i = 0; i < nc; ++i) cand[i].harris = orb_harris(img, w, cand[i].x, cand[i].y);
    qsort(cand, nc, sizeof(orb_cand), cmp_harris);
    if (nc > want) nc = want;
    uint8_t* blur = (uint8_t*)malloc((size_t)h * w);
    orb_blur(img, h, w, blur);
    for (uint32_t i = 0; i < nc && n_out < cap; ++i) {
      const int x = cand[i].x, y = cand[i].y;
      /* intensity centroid (IC_Angle) */
      int m01 = 0, m10 = 0;
      for (int u = -ORB_HALF_PATCH; u <= ORB_HALF_PATCH; ++u) m10 += u * img[y * w + x + u];
      for (int v = 1; v <= ORB_HALF_PATCH; ++v) {
        int vsum = 0, d = umax[v];
        for (int u = -d; u <= d; ++u) {
          int below = img[(y + v) * w + x + u], above = img[(y - v) * w + x + u];
          vsum += below - above;
          m10 += u * (below + above);
        }
        m01 += v * vsum;
      }
      float fm10 = (float)m10, fm01 = (float)m01;
      float nrm = sqrtf(fm10 * fm10 + fm01 * fm01);
      float ca = nrm > 0.f ? fm10 / nrm : 1.f, sa = nrm > 0.f ? fm01 / nrm : 0.f;
      float ang = atan2f(fm01, fm10) * 57.29577951308232f;
      if (ang < 0.f) ang += 360.f;
      uint8_t* dd = desc + (size_t)n_out * 32;
      for (int byte = 0; byte < 32; ++byte) {
        int val = 0;
        for (int bit = 0; bit < 8; ++bit) {
          const int8_t* pp = pattern + 4 * (byte * 8 + bit);
          int x0 = (int)rintf((float)pp[0] * ca - (float)pp[1] * sa), y0 = (int)rintf((float)pp[0] * sa + (float)pp[1] * ca);
          int x1 = (int)rintf((float)pp[2] * ca - (float)pp[3] * sa), y1 = (int)rintf((float)pp[2] * sa + (float)pp[3] * ca);
          int t0 = blur[(y + y0) * w + (x + x0)], t1 = blur[(y + y1) * w + (x + x1)];
          val |= (t0 < t1) << bit;
        }
        dd[byte] = (uint8_t)val;
      }
      kp_xy[2 * n_out] = (float)x * scale; kp_xy[2 * n_out + 1] = (float)y * scale;
      kp_aux[4 * n_out] = 31.f * scale; kp_aux[4 * n_out + 1] = ang; kp_aux[4 * n_out + 2] = cand[i].harris;
      kp_aux[4 * n_out + 3] = (float)lvl;
      if (kp_lvl_xy) { kp_lvl_xy[2 * n_out] = x; kp_lvl_xy[2 * n_out + 1] = y; }
      ++n_out;
    }
    free(blur);
    free(cand);
  }
  free(prev);
  return n_out;
}
