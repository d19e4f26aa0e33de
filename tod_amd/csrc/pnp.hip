// 2D-only geometric verification on gfx950: poses from keypoints and model points alone, no cloud, no depth image.
// The reference leaves this branch as a TODO (src/detection/GuessGenerator.cpp:147-152 "Only use 2d to 3d matching";
// doc/source/index.rst:36-46 "if the input is only 2d, it's a PnP problem (for which we have not plugged the solvePnP
// from OpenCV yet)"), so there is no reference result to match: the definition is stated in include/todhip.h
// (todhip_verify_2d) and checked bit for bit by the tests' CPU definition. What it keeps of the 3D branch: the per-object
// clustering in CSR order (adjacency_ransac.cpp:176-205), the sample-adjacency conditions that do not need a measured 3D point
// (keypoints > 20 px apart :484, model points within the object's span :477), keep-the-largest-consensus-set with the first one
// on ties (ransac.h:112-121), the min_inliers gate (GuessGenerator.cpp:211).
//
// Shape on the GPU: hypotheses are independent by construction (a counter-based hash of (seed, object, hypothesis,
// attempt) picks the sample; the rand() stream only supplies the seed), so they are evaluated side by side:
//   PNPH  pnp_hypotheses_kernel   thread = hypothesis: sample, Grunert P3P in f64 (quartic by polynomial arithmetic, roots
//                                 by derivative bracketing + 80 bisections -- only + - * / sqrt, so the order of operations
//                                 is the result), up to 4 poses, each scored over the object's matches; block max, then
//                                 one atomicMax per block on the object's (count, first (hypothesis, root)) key
//   PNPR  pnp_refine_kernel       wave = object: recompute the winner, flag its consensus set (lane-parallel), 5 Gauss-Newton
//                                 steps whose normal equations are 64 interleaved partial sums (lane = partial) added up
//                                 in lane order, consensus set under the refined pose
// f64 vector arithmetic is cheap on this part and the frame's matches are a few KB: neither kernel is near any roof; the
// point of the GPU form is that it sits behind the matcher's device buffers and costs the frame ~0.1 ms.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

#include "ctx.h"

namespace {

struct Cam { double fx, fy, cx, cy; };
struct PoseD { double R[9], t[3]; };
struct ObjSpanP { uint32_t begin, n, object; float span; uint32_t seed, frame; };   // one object of one frame of the batch
struct ObjResult { uint32_t valid, n_inliers; float R[9], t[3]; };

__device__ __forceinline__ uint32_t mix32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }

__device__ __forceinline__ double horner(const double* c, int deg, double x) {
  double y = c[deg];
  for (int i = deg - 1; i >= 0; --i) y = y * x + c[i];
  return y;
}

// roots of a quadratic, ascending
__device__ __forceinline__ int roots2(const double* c, double* out) {
  const double disc = c[1] * c[1] - 4.0 * c[2] * c[0];
  if (disc < 0.0) return 0;
  const double s = sqrt(disc), r0 = (-c[1] - s) / (2.0 * c[2]), r1 = (-c[1] + s) / (2.0 * c[2]);
  out[0] = r0 < r1 ? r0 : r1; out[1] = r0 < r1 ? r1 : r0;
  return 2;
}

// real roots of a polynomial of degree DEG (3 or 4), ascending, given the ascending real roots of its derivative: one root in
// every interval between consecutive critical points (and out to the Cauchy bound) over which the sign changes
template <int DEG>
__device__ int roots_between(const double* c, const double* crit, int nc, double* out) {
  double bound = 0.0;
  for (int i = 0; i < DEG; ++i) { const double a = fabs(c[i] / c[DEG]); if (a > bound) bound = a; }
  bound += 1.0;
  double edge[DEG + 1];
  int ne = 0;
  edge[ne++] = -bound;
  for (int i = 0; i < nc; ++i) if (crit[i] > -bound && crit[i] < bound) edge[ne++] = crit[i];
  edge[ne++] = bound;
  int n = 0;
  for (int i = 0; i + 1 < ne; ++i) {
    double lo = edge[i], hi = edge[i + 1];
    const double flo = horner(c, DEG, lo), fhi = horner(c, DEG, hi);
    if (flo == 0.0) { if (n == 0 || out[n - 1] != lo) out[n++] = lo; continue; }
    if ((flo < 0.0) == (fhi < 0.0) || fhi == 0.0) continue;
    for (int it = 0; it < 80; ++it) {
      const double mid = 0.5 * (lo + hi), fm = horner(c, DEG, mid);
      if ((fm < 0.0) == (flo < 0.0)) lo = mid; else hi = mid;
    }
    out[n++] = 0.5 * (lo + hi);
  }
  if (n < DEG && horner(c, DEG, bound) == 0.0) out[n++] = bound;
  return n;
}

__device__ int real_roots4(const double* c, double* out) {
  double d3[4], d2[3], r2[2], r3[3];
  for (int i = 1; i <= 4; ++i) d3[i - 1] = (double)i * c[i];        // the cubic derivative
  for (int i = 1; i <= 3; ++i) d2[i - 1] = (double)i * d3[i];       // its quadratic derivative
  const int n2 = roots2(d2, r2);
  const int n3 = roots_between<3>(d3, r2, n2, r3);
  return roots_between<4>(c, r3, n3, out);
}

__device__ __forceinline__ void cross3(const double* a, const double* b, double* o) {
  o[0] = a[1] * b[2] - a[2] * b[1]; o[1] = a[2] * b[0] - a[0] * b[2]; o[2] = a[0] * b[1] - a[1] * b[0];
}
__device__ __forceinline__ double dot3(const double* a, const double* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }

// orthonormal frame of a triangle: e1 along P2 - P1, e3 its normal, e2 = e3 x e1; false if the triangle is degenerate
__device__ bool frame3(const double* P1, const double* P2, const double* P3, double* e1, double* e2, double* e3) {
  double a[3] = {P2[0] - P1[0], P2[1] - P1[1], P2[2] - P1[2]}, b[3] = {P3[0] - P1[0], P3[1] - P1[1], P3[2] - P1[2]}, n[3];
  const double la = sqrt(dot3(a, a));
  if (!(la > 0.0)) return false;
  for (int i = 0; i < 3; ++i) e1[i] = a[i] / la;
  cross3(e1, b, n);
  const double ln = sqrt(dot3(n, n));
  if (!(ln > 1e-12 * sqrt(dot3(b, b)))) return false;
  for (int i = 0; i < 3; ++i) e3[i] = n[i] / ln;
  cross3(e3, e1, e2);
  return true;
}

// Grunert's P3P: s2 = u s1, s3 = v s1; u = N(v) / D(v) from the difference of the two ratio equations, the quartic in v from
// substituting it back: D^2 + N^2 - 2 cos(gamma) N D - (c^2 / b^2) W D^2 = 0 with W = 1 - 2 cos(beta) v + v^2
__device__ int p3p(const double X[3][3], const double f[3][3], PoseD* out) {
  double d[3];
  for (int i = 0; i < 3; ++i) d[i] = X[1][i] - X[2][i];
  const double a2 = dot3(d, d);
  for (int i = 0; i < 3; ++i) d[i] = X[0][i] - X[2][i];
  const double b2 = dot3(d, d);
  for (int i = 0; i < 3; ++i) d[i] = X[0][i] - X[1][i];
  const double c2 = dot3(d, d);
  if (!(a2 > 0.0) || !(b2 > 0.0) || !(c2 > 0.0)) return 0;
  double e1[3], e2[3], e3[3];
  if (!frame3(X[0], X[1], X[2], e1, e2, e3)) return 0;
  const double ca = dot3(f[1], f[2]), cb = dot3(f[0], f[2]), cg = dot3(f[0], f[1]);
  const double q = (a2 - c2) / b2, cb2 = c2 / b2;
  const double N[3] = {1.0 + q, -2.0 * q * cb, q - 1.0}, D[2] = {2.0 * cg, -2.0 * ca}, W[3] = {1.0, -2.0 * cb, 1.0};
  const double DD[3] = {D[0] * D[0], 2.0 * D[0] * D[1], D[1] * D[1]};
  double NN[5] = {0, 0, 0, 0, 0}, ND[4] = {0, 0, 0, 0}, WDD[5] = {0, 0, 0, 0, 0}, P[5];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) NN[i + j] += N[i] * N[j];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) ND[i + j] += N[i] * D[j];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) WDD[i + j] += W[i] * DD[j];
#pragma unroll
  for (int i = 0; i < 5; ++i) P[i] = (i < 3 ? DD[i] : 0.0) + NN[i] - 2.0 * cg * (i < 4 ? ND[i] : 0.0) - cb2 * WDD[i];
  double big = 0.0;
  for (int i = 0; i < 5; ++i) if (fabs(P[i]) > big) big = fabs(P[i]);
  if (!(fabs(P[4]) > 1e-12 * big)) return 0;
  double roots[4];
  const int nr = real_roots4(P, roots);
  int n = 0;
  for (int r = 0; r < nr; ++r) {
    const double v = roots[r];
    if (!(v > 0.0)) continue;
    const double den = D[0] + D[1] * v;
    if (den == 0.0) continue;
    const double u = ((N[2] * v + N[1]) * v + N[0]) / den;
    if (!(u > 0.0)) continue;
    const double w = 1.0 + u * u - 2.0 * u * cg;
    if (!(w > 0.0)) continue;
    const double s1 = sqrt(c2 / w), s2 = u * s1, s3 = v * s1;
    double Q[3][3], g1[3], g2[3], g3[3];
    for (int i = 0; i < 3; ++i) { Q[0][i] = s1 * f[0][i]; Q[1][i] = s2 * f[1][i]; Q[2][i] = s3 * f[2][i]; }
    if (!frame3(Q[0], Q[1], Q[2], g1, g2, g3)) continue;
    PoseD* p = &out[n++];
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) p->R[3 * i + j] = g1[i] * e1[j] + g2[i] * e2[j] + g3[i] * e3[j];
    for (int i = 0; i < 3; ++i) p->t[i] = Q[0][i] - (p->R[3 * i] * X[0][0] + p->R[3 * i + 1] * X[0][1] + p->R[3 * i + 2] * X[0][2]);
  }
  return n;
}

// squared reprojection error of model point X under (R, t); negative if the point is not in front of the camera
__device__ __forceinline__ double reproj2(const PoseD& p, const Cam& k, const float* X, const float* uv) {
  const double x = X[0], y = X[1], z = X[2];
  const double xc = p.R[0] * x + p.R[1] * y + p.R[2] * z + p.t[0];
  const double yc = p.R[3] * x + p.R[4] * y + p.R[5] * z + p.t[1];
  const double zc = p.R[6] * x + p.R[7] * y + p.R[8] * z + p.t[2];
  if (!(zc > 1e-9)) return -1.0;
  const double du = k.fx * (xc / zc) + k.cx - (double)uv[0], dv = k.fy * (yc / zc) + k.cy - (double)uv[1];
  return du * du + dv * dv;
}

// sample adjacency of two matches: keypoints more than 20 px apart (adjacency_ransac.cpp:484), model points distinct and within
// the object's span (:477 without the sensor-error slack)
__device__ __forceinline__ bool pair_ok(const float* kp_a, const float* kp_b, const float* Xa, const float* Xb, float span) {
  const float du = kp_a[0] - kp_b[0], dv = kp_a[1] - kp_b[1];
  if (!((du * du + dv * dv) > 20.f * 20.f)) return false;
  const float dx = Xa[0] - Xb[0], dy = Xa[1] - Xb[1], dz = Xa[2] - Xb[2];
  const float d2 = dx * dx + dy * dy + dz * dz;
  return d2 > 0.f && d2 <= span * span;
}

// hypothesis `hyp` of object `obj`: three pairwise sample-adjacent matches, at most 16 attempts
__device__ bool draw_sample(uint32_t seed, uint32_t obj, uint32_t hyp, uint32_t n, const float* kp_xy, const uint32_t* q_idx, const float* X,
                            float span, uint32_t* s3) {
  const uint32_t base = seed ^ mix32(obj * 0x9E3779B9U + hyp);
  for (uint32_t a = 0; a < 16; ++a) {
    const uint32_t h0 = mix32(base + a * 0x85EBCA6BU), h1 = mix32(h0 + 0x68E31DA4U), h2 = mix32(h1 + 0xB5297A4DU);
    const uint32_t i = h0 % n, j = h1 % n, k = h2 % n;
    if (i == j || i == k || j == k) continue;
    if (!pair_ok(kp_xy + 2 * q_idx[i], kp_xy + 2 * q_idx[j], X + 3 * i, X + 3 * j, span)) continue;
    if (!pair_ok(kp_xy + 2 * q_idx[i], kp_xy + 2 * q_idx[k], X + 3 * i, X + 3 * k, span)) continue;
    if (!pair_ok(kp_xy + 2 * q_idx[j], kp_xy + 2 * q_idx[k], X + 3 * j, X + 3 * k, span)) continue;
    s3[0] = i; s3[1] = j; s3[2] = k;
    return true;
  }
  return false;
}

__device__ int sample_poses(const Cam& cam, const float* kp_xy, const uint32_t* q_idx, const float* X, const uint32_t* s3, PoseD* out) {
  double Xd[3][3], f[3][3];
  for (int a = 0; a < 3; ++a) {
    for (int i = 0; i < 3; ++i) Xd[a][i] = X[3 * s3[a] + i];
    const float* uv = kp_xy + 2 * q_idx[s3[a]];
    const double x = ((double)uv[0] - cam.cx) / cam.fx, y = ((double)uv[1] - cam.cy) / cam.fy;
    const double l = sqrt(x * x + y * y + 1.0);
    f[a][0] = x / l; f[a][1] = y / l; f[a][2] = 1.0 / l;
  }
  return p3p(Xd, f, out);
}

// PNPH. grid (ceil(n_hyp / 256), active objects). best[obj]: (consensus size << 32) | ~(4 hypothesis + root), so that the maximum
// is the largest consensus set and among equals the first (hypothesis, root) -- what a sequential walk with `>` keeps
__global__ __launch_bounds__(256) void pnp_hypotheses_kernel(const ObjSpanP* __restrict__ objs, const float* __restrict__ kp_xy,
                                                             const uint32_t* __restrict__ q_idx_all, const float* __restrict__ X_all, Cam cam,
                                                             uint32_t n_hyp, double err2, unsigned long long* __restrict__ best) {
  const ObjSpanP o = objs[blockIdx.y];
  const uint32_t hyp = blockIdx.x * 256u + threadIdx.x;
  const uint32_t* q_idx = q_idx_all + o.begin;
  const float* X = X_all + 3 * (size_t)o.begin;
  unsigned long long key = 0ull;
  uint32_t s3[3];
  if (hyp < n_hyp && draw_sample(o.seed, o.object, hyp, o.n, kp_xy, q_idx, X, o.span, s3)) {
    PoseD sol[4];
    const int ns = sample_poses(cam, kp_xy, q_idx, X, s3, sol);
    for (int s = 0; s < ns; ++s) {
      uint32_t c = 0;
      for (uint32_t m = 0; m < o.n; ++m) {
        const double e = reproj2(sol[s], cam, X + 3 * m, kp_xy + 2 * q_idx[m]);
        c += (e >= 0.0 && e < err2) ? 1u : 0u;
      }
      const unsigned long long k2 = ((unsigned long long)c << 32) | (unsigned long long)(0xFFFFFFFFu - (4u * hyp + (uint32_t)s));
      if (c > 0u && k2 > key) key = k2;
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) { const unsigned long long other = __shfl_xor(key, off); if (other > key) key = other; }
  __shared__ unsigned long long s_key[4];
  if ((threadIdx.x & 63u) == 0u) s_key[threadIdx.x >> 6] = key;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 4; ++w) if (s_key[w] > key) key = s_key[w];
    if (key) atomicMax(&best[blockIdx.y], key);
  }
}

// PNPR. One block of 64 lanes per active object (declared for 128 so that __syncthreads stays a real barrier).
__global__ __launch_bounds__(128) void pnp_refine_kernel(const ObjSpanP* __restrict__ objs, const float* __restrict__ kp_xy,
                                                         const uint32_t* __restrict__ q_idx_all, const float* __restrict__ X_all, Cam cam,
                                                         double err2, uint32_t min_inliers,
                                                         const unsigned long long* __restrict__ best, uint8_t* __restrict__ flags_all,
                                                         ObjResult* __restrict__ results) {
  const ObjSpanP o = objs[blockIdx.x];
  const uint32_t l = threadIdx.x;
  const uint32_t* q_idx = q_idx_all + o.begin;
  const float* X = X_all + 3 * (size_t)o.begin;
  uint8_t* flags = flags_all + o.begin;
  __shared__ PoseD s_pose;
  __shared__ double s_part[64][42];
  __shared__ int s_ok;
  const unsigned long long key = best[blockIdx.x];
  const uint32_t bc = (uint32_t)(key >> 32), code = 0xFFFFFFFFu - (uint32_t)key;
  if (key == 0ull || bc < min_inliers || bc < 3u) {
    if (l == 0) { results[blockIdx.x].valid = 0u; results[blockIdx.x].n_inliers = 0u; }
    return;
  }
  if (l == 0) {
    uint32_t s3[3];
    PoseD sol[4];
    draw_sample(o.seed, o.object, code >> 2, o.n, kp_xy, q_idx, X, o.span, s3);
    sample_poses(cam, kp_xy, q_idx, X, s3, sol);
    s_pose = sol[code & 3u];
  }
  __syncthreads();
  PoseD p = s_pose;
  for (uint32_t m = l; m < o.n; m += 64u) {
    const double e = reproj2(p, cam, X + 3 * m, kp_xy + 2 * q_idx[m]);
    flags[m] = (e >= 0.0 && e < err2) ? 1 : 0;
  }
  for (int it = 0; it < 5; ++it) {
    double H[6][6], g[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) { g[i] = 0.0;
#pragma unroll
      for (int j = 0; j < 6; ++j) H[i][j] = 0.0; }
    for (uint32_t m = l; m < o.n; m += 64u) {                        // partial sum number l: matches l, l + 64, ...
      if (!flags[m]) continue;
      const double x = X[3 * m], y = X[3 * m + 1], z = X[3 * m + 2];
      const double xc = p.R[0] * x + p.R[1] * y + p.R[2] * z + p.t[0];
      const double yc = p.R[3] * x + p.R[4] * y + p.R[5] * z + p.t[1];
      const double zc = p.R[6] * x + p.R[7] * y + p.R[8] * z + p.t[2];
      if (!(zc > 1e-9)) continue;
      const double iz = 1.0 / zc, xn = xc * iz, yn = yc * iz;
      const double ru = cam.fx * xn + cam.cx - (double)kp_xy[2 * q_idx[m]], rv = cam.fy * yn + cam.cy - (double)kp_xy[2 * q_idx[m] + 1];
      const double a0 = cam.fx * iz, a2 = -cam.fx * xn * iz, b1 = cam.fy * iz, b2 = -cam.fy * yn * iz;
      const double Ju[6] = {a2 * yc, a0 * zc - a2 * xc, -a0 * yc, a0, 0.0, a2};
      const double Jv[6] = {-b1 * zc + b2 * yc, -b2 * xc, b1 * xc, 0.0, b1, b2};
#pragma unroll
      for (int i = 0; i < 6; ++i) {
#pragma unroll
        for (int j = 0; j < 6; ++j) H[i][j] += Ju[i] * Ju[j] + Jv[i] * Jv[j];
        g[i] += Ju[i] * ru + Jv[i] * rv;
      }
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) {
#pragma unroll
      for (int j = 0; j < 6; ++j) s_part[l][6 * i + j] = H[i][j];
      s_part[l][36 + i] = g[i];
    }
    __syncthreads();
    if (l == 0) {
      double A[6][6], b[6], dlt[6];
      for (int i = 0; i < 6; ++i) { b[i] = 0.0; for (int j = 0; j < 6; ++j) A[i][j] = 0.0; }
      for (int w = 0; w < 64; ++w)
        for (int i = 0; i < 6; ++i) {
          for (int j = 0; j < 6; ++j) A[i][j] += s_part[w][6 * i + j];
          b[i] += s_part[w][36 + i];
        }
      for (int i = 0; i < 6; ++i) b[i] = -b[i];
      // 6 x 6 Gaussian elimination with partial pivoting
      bool ok = true;
      for (int c = 0; c < 6 && ok; ++c) {
        int piv = c;
        for (int r = c + 1; r < 6; ++r) if (fabs(A[r][c]) > fabs(A[piv][c])) piv = r;
        if (!(fabs(A[piv][c]) > 1e-300)) { ok = false; break; }
        if (piv != c) { for (int j = 0; j < 6; ++j) { const double t = A[c][j]; A[c][j] = A[piv][j]; A[piv][j] = t; } const double t = b[c]; b[c] = b[piv]; b[piv] = t; }
        for (int r = c + 1; r < 6; ++r) {
          const double mlt = A[r][c] / A[c][c];
          for (int j = c; j < 6; ++j) A[r][j] -= mlt * A[c][j];
          b[r] -= mlt * b[c];
        }
      }
      if (ok) {
        for (int r = 5; r >= 0; --r) {
          double s = b[r];
          for (int j = r + 1; j < 6; ++j) s -= A[r][j] * dlt[j];
          dlt[r] = s / A[r][r];
        }
        // R <- (I + [w]x) R, t <- (I + [w]x) t + dt, Gram-Schmidt on the rows of R
        double Rn[9], tn[3];
        const double wx = dlt[0], wy = dlt[1], wz = dlt[2];
        for (int j = 0; j < 3; ++j) {
          Rn[j]     = p.R[j]     - wz * p.R[3 + j] + wy * p.R[6 + j];
          Rn[3 + j] = p.R[3 + j] + wz * p.R[j]     - wx * p.R[6 + j];
          Rn[6 + j] = p.R[6 + j] - wy * p.R[j]     + wx * p.R[3 + j];
        }
        tn[0] = p.t[0] - wz * p.t[1] + wy * p.t[2] + dlt[3];
        tn[1] = p.t[1] + wz * p.t[0] - wx * p.t[2] + dlt[4];
        tn[2] = p.t[2] - wy * p.t[0] + wx * p.t[1] + dlt[5];
        const double l0 = sqrt(dot3(Rn, Rn));
        for (int j = 0; j < 3; ++j) Rn[j] /= l0;
        const double d01 = dot3(Rn, Rn + 3);
        for (int j = 0; j < 3; ++j) Rn[3 + j] -= d01 * Rn[j];
        const double l1 = sqrt(dot3(Rn + 3, Rn + 3));
        for (int j = 0; j < 3; ++j) Rn[3 + j] /= l1;
        cross3(Rn, Rn + 3, Rn + 6);
        for (int j = 0; j < 9; ++j) s_pose.R[j] = Rn[j];
        for (int j = 0; j < 3; ++j) s_pose.t[j] = tn[j];
      }
      s_ok = ok ? 1 : 0;
    }
    __syncthreads();
    const int ok = s_ok;
    p = s_pose;
    __syncthreads();
    if (!ok) break;                                                  // a singular system ends the refinement (uniform)
  }
  uint32_t c = 0;
  for (uint32_t m = l; m < o.n; m += 64u) {
    const double e = reproj2(p, cam, X + 3 * m, kp_xy + 2 * q_idx[m]);
    const bool in = e >= 0.0 && e < err2;
    flags[m] = in ? 1 : 0;
    c += in ? 1u : 0u;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) c += __shfl_xor(c, off);
  if (l == 0) {
    ObjResult& r = results[blockIdx.x];
    r.valid = c >= min_inliers ? 1u : 0u;
    r.n_inliers = c;
    for (int j = 0; j < 9; ++j) r.R[j] = (float)p.R[j];
    for (int j = 0; j < 3; ++j) r.t[j] = (float)p.t[j];
  }
}

// The objects worth a RANSAC (>= 3 matches and >= min_inliers of them), in the order the host form visits them (frame major,
// object ascending), from the per-frame histograms and offsets of the device-side ClusterPerObject. One block.
__global__ __launch_bounds__(256) void pnp_active_kernel(const uint32_t* __restrict__ hist, const uint32_t* __restrict__ goff,
                                                         const float* __restrict__ spans, const uint32_t* __restrict__ seeds, uint32_t F,
                                                         uint32_t n_objs, uint32_t per_frame, uint32_t min_inliers, uint32_t cap,
                                                         ObjSpanP* __restrict__ objs, uint32_t* __restrict__ n_active) {
  __shared__ uint32_t part[256], s_base;
  const uint32_t tid = threadIdx.x;
  if (tid == 0) s_base = 0u;
  __syncthreads();
  const uint32_t chunk = (n_objs + 255u) / 256u;
  for (uint32_t f = 0; f < F; ++f) {
    const uint32_t lo = min(n_objs, tid * chunk), hi = min(n_objs, lo + chunk);
    uint32_t c = 0;
    for (uint32_t o = lo; o < hi; ++o) { const uint32_t n = hist[(size_t)f * n_objs + o]; c += (n >= 3u && n >= min_inliers) ? 1u : 0u; }
    part[tid] = c;
    __syncthreads();
    if (tid == 0) {
      uint32_t acc = s_base;
      for (uint32_t i = 0; i < 256u; ++i) { const uint32_t v = part[i]; part[i] = acc; acc += v; }
      s_base = acc;
    }
    __syncthreads();
    uint32_t at = part[tid];
    for (uint32_t o = lo; o < hi; ++o) {
      const uint32_t n = hist[(size_t)f * n_objs + o];
      if (n >= 3u && n >= min_inliers) {
        if (at < cap) objs[at] = ObjSpanP{f * per_frame + goff[(size_t)f * n_objs + o], n, o, spans[o], seeds[f], f};
        ++at;
      }
    }
    __syncthreads();
  }
  if (tid == 0) *n_active = s_base;
}

struct PnpWs { DevBuf objs, kp, q_idx, X, best, flags, results, hist, goff, err, spans, seeds, na; };

inline uint32_t rng_next(todhip_rng& r) {                            // glibc random_r TYPE_3, as in verify.hip
  r.s[r.f] += r.s[r.b];
  const uint32_t out = r.s[r.f] >> 1;
  r.f = r.f == 30u ? 0u : r.f + 1u; r.b = r.b == 30u ? 0u : r.b + 1u;
  ++r.draws;
  return out;
}

}  // namespace

void tod_pnp_ws_free(todhip_ctx* ctx) {
  PnpWs* ws = reinterpret_cast<PnpWs*>(ctx->pnp_ws);
  if (!ws) return;
  DevBuf* bufs[] = {&ws->objs, &ws->kp, &ws->q_idx, &ws->X, &ws->best, &ws->flags, &ws->results, &ws->hist, &ws->goff, &ws->err, &ws->spans,
                    &ws->seeds, &ws->na};
  for (DevBuf* b : bufs) b->release();
  delete ws;
  ctx->pnp_ws = nullptr;
}

// One batch: per frame host arrays (keypoints, CSR matches), one generator per frame. The objects of all frames go through the two
// kernels together (a frame's keypoints sit at frame * nq in the device array, q_idx carries that offset); poses come back per frame.
struct Frame2d { const float* kp_xy; const uint32_t* row_ptr; const todhip_dmatch* matches; const float* matches_xyz; };

static int verify_2d_core(todhip_ctx* ctx, uint32_t F, const Frame2d* fr, uint32_t nq, const float* K9, const float* spans, uint32_t n_objs,
                          const todhip_verify_params* prm, todhip_rng* rngs, todhip_pose* poses, uint32_t* n_poses, uint32_t* pose_ptr,
                          uint32_t* inlier_kp, uint32_t* n_inlier_kp) {
  if (!(prm->sensor_error > 0.f) || !(K9[0] > 0.f) || !(K9[4] > 0.f)) return TODHIP_EINVAL;
  if ((*n_poses && !poses) || (*n_inlier_kp && !inlier_kp)) return TODHIP_EINVAL;
  size_t n_all = 0;
  for (uint32_t f = 0; f < F; ++f) {
    const uint32_t nm = fr[f].row_ptr[nq];
    if (nm && (!fr[f].matches || !fr[f].matches_xyz || !spans)) return TODHIP_EINVAL;
    for (uint32_t m = 0; m < nm; ++m)
      if (fr[f].matches[m].imgIdx < 0 || (uint32_t)fr[f].matches[m].imgIdx >= n_objs) return TODHIP_EINVAL;
    n_all += nm;
  }
  if (n_all > 0xFFFFFFFFull || (uint64_t)F * nq > 0xFFFFFFFFull) return TODHIP_EINVAL;
  TOD_HIP(hipSetDevice(ctx->device));
  const uint32_t cap_poses = *n_poses, cap_inl = *n_inlier_kp;
  *n_poses = 0; *n_inlier_kp = 0;
  if (pose_ptr) for (uint32_t f = 0; f <= F; ++f) pose_ptr[f] = 0;
  // ClusterPerObject (adjacency_ransac.cpp:176-205) per frame: per object, matches in CSR order
  std::vector<uint32_t> q_idx(n_all);
  std::vector<float> X(3 * n_all), kp_all(2 * (size_t)F * nq);
  std::vector<ObjSpanP> active;
  std::vector<uint32_t> cnt(n_objs + 1), fill(n_objs);
  size_t base = 0;
  for (uint32_t f = 0; f < F; ++f) {
    const uint32_t seed = rng_next(rngs[f]);                           // the one draw a frame costs
    if (nq) std::memcpy(&kp_all[2 * (size_t)f * nq], fr[f].kp_xy, (size_t)nq * 8);
    const uint32_t nm = fr[f].row_ptr[nq];
    std::fill(cnt.begin(), cnt.end(), 0u);
    for (uint32_t m = 0; m < nm; ++m) cnt[(uint32_t)fr[f].matches[m].imgIdx + 1]++;
    for (uint32_t o = 0; o < n_objs; ++o) cnt[o + 1] += cnt[o];
    std::copy(cnt.begin(), cnt.end() - 1, fill.begin());
    for (uint32_t q = 0; q < nq; ++q)
      for (uint32_t m = fr[f].row_ptr[q]; m < fr[f].row_ptr[q + 1]; ++m) {
        const size_t at = base + fill[(uint32_t)fr[f].matches[m].imgIdx]++;
        q_idx[at] = f * nq + q; std::memcpy(&X[3 * at], fr[f].matches_xyz + 3 * (size_t)m, 3 * sizeof(float));
      }
    for (uint32_t o = 0; o < n_objs; ++o) {
      const uint32_t n = cnt[o + 1] - cnt[o];
      if (n >= 3u && n >= prm->min_inliers) active.push_back(ObjSpanP{(uint32_t)(base + cnt[o]), n, o, spans[o], seed, f});
    }
    base += nm;
  }
  if (active.empty() || prm->n_ransac_iterations == 0) return TODHIP_OK;
  if (!ctx->pnp_ws) ctx->pnp_ws = new PnpWs();
  PnpWs* ws = reinterpret_cast<PnpWs*>(ctx->pnp_ws);
  hipStream_t st = ctx->stream;
  const uint32_t na = (uint32_t)active.size();
  TOD_HIP(ws->objs.reserve(na * sizeof(ObjSpanP)));
  TOD_HIP(ws->kp.reserve(kp_all.size() * 4));
  TOD_HIP(ws->q_idx.reserve(n_all * 4));
  TOD_HIP(ws->X.reserve(n_all * 12));
  TOD_HIP(ws->best.reserve((size_t)na * 8));
  TOD_HIP(ws->flags.reserve(n_all));
  TOD_HIP(ws->results.reserve((size_t)na * sizeof(ObjResult)));
  TOD_HIP(hipMemcpyAsync(ws->objs.p, active.data(), na * sizeof(ObjSpanP), hipMemcpyHostToDevice, st));
  TOD_HIP(hipMemcpyAsync(ws->kp.p, kp_all.data(), kp_all.size() * 4, hipMemcpyHostToDevice, st));
  TOD_HIP(hipMemcpyAsync(ws->q_idx.p, q_idx.data(), n_all * 4, hipMemcpyHostToDevice, st));
  TOD_HIP(hipMemcpyAsync(ws->X.p, X.data(), n_all * 12, hipMemcpyHostToDevice, st));
  TOD_HIP(hipMemsetAsync(ws->best.p, 0, (size_t)na * 8, st));
  const Cam cam{(double)K9[0], (double)K9[4], (double)K9[2], (double)K9[5]};
  const double err2 = (double)prm->sensor_error * (double)prm->sensor_error;
  hipLaunchKernelGGL(pnp_hypotheses_kernel, dim3((prm->n_ransac_iterations + 255u) / 256u, na), dim3(256), 0, st, ws->objs.as<ObjSpanP>(),
                     ws->kp.as<float>(), ws->q_idx.as<uint32_t>(), ws->X.as<float>(), cam, prm->n_ransac_iterations, err2,
                     ws->best.as<unsigned long long>());
  hipLaunchKernelGGL(pnp_refine_kernel, dim3(na), dim3(64), 0, st, ws->objs.as<ObjSpanP>(), ws->kp.as<float>(), ws->q_idx.as<uint32_t>(),
                     ws->X.as<float>(), cam, err2, prm->min_inliers, ws->best.as<unsigned long long>(), ws->flags.as<uint8_t>(),
                     ws->results.as<ObjResult>());
  TOD_HIP(hipGetLastError());
  std::vector<ObjResult> res(na);
  std::vector<uint8_t> flags(n_all);
  TOD_HIP(hipMemcpyAsync(res.data(), ws->results.p, (size_t)na * sizeof(ObjResult), hipMemcpyDeviceToHost, st));
  TOD_HIP(hipMemcpyAsync(flags.data(), ws->flags.p, n_all, hipMemcpyDeviceToHost, st));
  TOD_HIP(hipStreamSynchronize(st));
  uint32_t np = 0, ni = 0;
  for (uint32_t a = 0; a < na; ++a) {                                   // frame major, objects ascending inside a frame
    if (!res[a].valid) continue;
    if (np >= cap_poses) return TODHIP_ECAPACITY;
    todhip_pose& out = poses[np];
    out.object = active[a].object;
    std::memcpy(out.R, res[a].R, sizeof(out.R)); std::memcpy(out.t, res[a].t, sizeof(out.t));
    out.inlier_begin = ni;
    uint32_t last = 0xFFFFFFFFu;
    for (uint32_t m = 0; m < active[a].n; ++m) {                       // keypoint indices of the frame, ascending, each once
      const uint32_t q = q_idx[active[a].begin + m] - active[a].frame * nq;
      if (!flags[active[a].begin + m] || q == last) continue;
      if (ni >= cap_inl) return TODHIP_ECAPACITY;
      inlier_kp[ni++] = last = q;
    }
    out.inlier_end = ni;
    ++np;
    if (pose_ptr) pose_ptr[active[a].frame + 1] = np;
  }
  if (pose_ptr) for (uint32_t f = 0; f < F; ++f) pose_ptr[f + 1] = std::max(pose_ptr[f + 1], pose_ptr[f]);   // frames without a pose
  *n_poses = np; *n_inlier_kp = ni;
  ctx->counters.last_poses = np;
  return TODHIP_OK;
}

extern "C" int todhip_verify_2d(todhip_ctx* ctx, const float* kp_xy, uint32_t nq, const float* K9, const uint32_t* row_ptr,
                                const todhip_dmatch* matches, const float* matches_xyz, const float* spans, uint32_t n_objs,
                                const todhip_verify_params* prm, todhip_rng* rng, todhip_pose* poses, uint32_t* n_poses,
                                uint32_t* inlier_kp, uint32_t* n_inlier_kp) {
  if (!ctx || !K9 || !row_ptr || !prm || !rng || !n_poses || !n_inlier_kp || (nq && !kp_xy)) return TODHIP_EINVAL;
  const Frame2d fr = {kp_xy, row_ptr, matches, matches_xyz};
  return verify_2d_core(ctx, 1, &fr, nq, K9, spans, n_objs, prm, rng, poses, n_poses, nullptr, inlier_kp, n_inlier_kp);
}

// Device-resident forms: keypoints and the matcher's fixed-stride outputs (exactly what todhip_match_device / todhip_merge_shards_device
// produced, for n_frames * nq queries) are in HBM and stay there: ClusterPerObject runs on the device (verify.hip's cluster_frame_kernel
// in its cloudless mode, one block per frame), pnp_active_kernel lists the objects worth a RANSAC, the host reads ONE word (their
// number, for the grids) and launches hypotheses + refinement; the flags, the results and the keypoint indices of the matches come
// back once, at the end. Each frame's result is the host-buffer call's (tests/test_pnp_gpu.py).
extern "C" int todhip_verify_2d_batch_device(todhip_ctx* ctx, uint32_t n_frames, const void* d_kp_xy, uint32_t nq, const float* K9,
                                             const void* d_counts, const void* d_matches, const void* d_matches_xyz, uint32_t k,
                                             const float* spans, uint32_t n_objs, const todhip_verify_params* prm, todhip_rng* rng,
                                             todhip_pose* poses, uint32_t* n_poses, uint32_t* pose_ptr, uint32_t* inlier_kp,
                                             uint32_t* n_inlier_kp) {
  if (!ctx || !K9 || !prm || !rng || !n_poses || !n_inlier_kp || k == 0 || n_frames == 0) return TODHIP_EINVAL;
  if (nq && (!d_kp_xy || !d_counts || !d_matches || !d_matches_xyz)) return TODHIP_EINVAL;
  if (!(prm->sensor_error > 0.f) || !(K9[0] > 0.f) || !(K9[4] > 0.f)) return TODHIP_EINVAL;
  if ((*n_poses && !poses) || (*n_inlier_kp && !inlier_kp)) return TODHIP_EINVAL;
  const uint32_t F = n_frames;
  const uint64_t per64 = (uint64_t)nq * k;
  if (per64 * F > 0xFFFFFFFFull || (uint64_t)F * nq > 0xFFFFFFFFull) return TODHIP_EINVAL;
  TOD_HIP(hipSetDevice(ctx->device));
  const uint32_t cap_poses = *n_poses, cap_inl = *n_inlier_kp;
  *n_poses = 0; *n_inlier_kp = 0;
  if (pose_ptr) for (uint32_t f = 0; f <= F; ++f) pose_ptr[f] = 0;
  std::vector<uint32_t> seeds(F);
  for (uint32_t f = 0; f < F; ++f) seeds[f] = rng_next(rng[f]);        // the one draw a frame costs (as the host form)
  if (nq == 0 || n_objs == 0 || !spans) return nq && n_objs && !spans ? TODHIP_EINVAL : TODHIP_OK;
  const uint32_t per = (uint32_t)per64;
  const size_t n_slots = (size_t)F * per;
  // an object needs max(3, min_inliers) matches of its frame's nq k: that bounds the list of active objects
  const uint32_t cap = F * std::min<uint32_t>(n_objs, per / std::max(3u, prm->min_inliers)) + 1u;
  if (!ctx->pnp_ws) ctx->pnp_ws = new PnpWs();
  PnpWs* ws = reinterpret_cast<PnpWs*>(ctx->pnp_ws);
  hipStream_t st = ctx->stream;
  TOD_HIP(ws->X.reserve(n_slots * 12)); TOD_HIP(ws->q_idx.reserve(n_slots * 4));
  TOD_HIP(ws->hist.reserve((size_t)F * n_objs * 4)); TOD_HIP(ws->goff.reserve((size_t)F * n_objs * 4));
  TOD_HIP(ws->err.reserve((size_t)F * 8 * 4)); TOD_HIP(ws->spans.reserve((size_t)n_objs * 4)); TOD_HIP(ws->seeds.reserve((size_t)F * 4));
  TOD_HIP(ws->objs.reserve((size_t)cap * sizeof(ObjSpanP))); TOD_HIP(ws->na.reserve(64));
  TOD_HIP(ws->best.reserve((size_t)cap * 8)); TOD_HIP(ws->flags.reserve(n_slots)); TOD_HIP(ws->results.reserve((size_t)cap * sizeof(ObjResult)));
  TOD_HIP(hipMemcpyAsync(ws->spans.p, spans, (size_t)n_objs * 4, hipMemcpyHostToDevice, st));
  TOD_HIP(hipMemcpyAsync(ws->seeds.p, seeds.data(), (size_t)F * 4, hipMemcpyHostToDevice, st));
  int rc = tod_cluster_frames_nocloud(ctx, F, reinterpret_cast<const float*>(d_kp_xy), nq, reinterpret_cast<const uint32_t*>(d_counts),
                                      reinterpret_cast<const todhip_dmatch*>(d_matches), reinterpret_cast<const float*>(d_matches_xyz), k,
                                      n_objs, ws->X.as<float>(), ws->q_idx.as<uint32_t>(), ws->hist.as<uint32_t>(), ws->goff.as<uint32_t>(),
                                      ws->err.as<uint32_t>());
  if (rc != TODHIP_OK) return rc;
  hipLaunchKernelGGL(pnp_active_kernel, dim3(1), dim3(256), 0, st, ws->hist.as<uint32_t>(), ws->goff.as<uint32_t>(), ws->spans.as<float>(),
                     ws->seeds.as<uint32_t>(), F, n_objs, per, prm->min_inliers, cap, ws->objs.as<ObjSpanP>(), ws->na.as<uint32_t>());
  TOD_HIP(hipGetLastError());
  uint32_t na = 0;
  std::vector<uint32_t> err(8 * (size_t)F);
  TOD_HIP(hipMemcpyAsync(&na, ws->na.p, 4, hipMemcpyDeviceToHost, st));
  TOD_HIP(hipMemcpyAsync(err.data(), ws->err.p, err.size() * 4, hipMemcpyDeviceToHost, st));
  TOD_HIP(hipStreamSynchronize(st));
  for (uint32_t f = 0; f < F; ++f)
    if (err[8 * (size_t)f] != 0u) return TODHIP_EINVAL;                 // an object index outside the DB, a count beyond k
  if (na > cap - 1u) return TODHIP_ESCRATCH;                            // (cannot happen: cap bounds the list)
  if (na == 0 || prm->n_ransac_iterations == 0) return TODHIP_OK;
  TOD_HIP(hipMemsetAsync(ws->best.p, 0, (size_t)na * 8, st));
  const Cam cam{(double)K9[0], (double)K9[4], (double)K9[2], (double)K9[5]};
  const double err2 = (double)prm->sensor_error * (double)prm->sensor_error;
  const float* kp = reinterpret_cast<const float*>(d_kp_xy);
  hipLaunchKernelGGL(pnp_hypotheses_kernel, dim3((prm->n_ransac_iterations + 255u) / 256u, na), dim3(256), 0, st, ws->objs.as<ObjSpanP>(),
                     kp, ws->q_idx.as<uint32_t>(), ws->X.as<float>(), cam, prm->n_ransac_iterations, err2, ws->best.as<unsigned long long>());
  hipLaunchKernelGGL(pnp_refine_kernel, dim3(na), dim3(64), 0, st, ws->objs.as<ObjSpanP>(), kp, ws->q_idx.as<uint32_t>(), ws->X.as<float>(),
                     cam, err2, prm->min_inliers, ws->best.as<unsigned long long>(), ws->flags.as<uint8_t>(), ws->results.as<ObjResult>());
  TOD_HIP(hipGetLastError());
  std::vector<ObjResult> res(na);
  std::vector<ObjSpanP> active(na);
  std::vector<uint8_t> flags(n_slots);
  std::vector<uint32_t> q_idx(n_slots);
  TOD_HIP(hipMemcpyAsync(res.data(), ws->results.p, (size_t)na * sizeof(ObjResult), hipMemcpyDeviceToHost, st));
  TOD_HIP(hipMemcpyAsync(active.data(), ws->objs.p, (size_t)na * sizeof(ObjSpanP), hipMemcpyDeviceToHost, st));
  TOD_HIP(hipMemcpyAsync(flags.data(), ws->flags.p, n_slots, hipMemcpyDeviceToHost, st));
  TOD_HIP(hipMemcpyAsync(q_idx.data(), ws->q_idx.p, n_slots * 4, hipMemcpyDeviceToHost, st));
  TOD_HIP(hipStreamSynchronize(st));
  uint32_t np = 0, ni = 0;
  for (uint32_t a = 0; a < na; ++a) {                                   // frame major, objects ascending inside a frame
    if (!res[a].valid) continue;
    if (np >= cap_poses) return TODHIP_ECAPACITY;
    todhip_pose& out = poses[np];
    out.object = active[a].object;
    std::memcpy(out.R, res[a].R, sizeof(out.R)); std::memcpy(out.t, res[a].t, sizeof(out.t));
    out.inlier_begin = ni;
    uint32_t last = 0xFFFFFFFFu;
    for (uint32_t m = 0; m < active[a].n; ++m) {                       // keypoint indices of the frame, ascending, each once
      const uint32_t q = q_idx[active[a].begin + m] - active[a].frame * nq;
      if (!flags[active[a].begin + m] || q == last) continue;
      if (ni >= cap_inl) return TODHIP_ECAPACITY;
      inlier_kp[ni++] = last = q;
    }
    out.inlier_end = ni;
    ++np;
    if (pose_ptr) pose_ptr[active[a].frame + 1] = np;
  }
  if (pose_ptr) for (uint32_t f = 0; f < F; ++f) pose_ptr[f + 1] = std::max(pose_ptr[f + 1], pose_ptr[f]);   // frames without a pose
  *n_poses = np; *n_inlier_kp = ni;
  ctx->counters.last_poses = np;
  return TODHIP_OK;
}

extern "C" int todhip_verify_2d_device(todhip_ctx* ctx, const void* d_kp_xy, uint32_t nq, const float* K9, const void* d_counts,
                                       const void* d_matches, const void* d_matches_xyz, uint32_t k, const float* spans, uint32_t n_objs,
                                       const todhip_verify_params* prm, todhip_rng* rng, todhip_pose* poses, uint32_t* n_poses,
                                       uint32_t* inlier_kp, uint32_t* n_inlier_kp) {
  return todhip_verify_2d_batch_device(ctx, 1, d_kp_xy, nq, K9, d_counts, d_matches, d_matches_xyz, k, spans, n_objs, prm, rng, poses, n_poses,
                                       nullptr, inlier_kp, n_inlier_kp);
}
