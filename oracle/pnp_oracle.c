/*
 * oracle/pnp_oracle.c -- TEST INFRASTRUCTURE ONLY (see tod_oracle.h).
 *
 * 2D-only geometric verification: the branch the reference leaves as a TODO
 * (src/detection/GuessGenerator.cpp:147-152 "Only use 2d to 3d matching // TODO"; doc/source/index.rst:36-46: "if the
 * input is only 2d, it's a PnP problem (for which we have not plugged the solvePnP from OpenCV yet)").
 * PARITY UNPINNED BY CONSTRUCTION: the reference has no code for it, so this file DEFINES the result that
 * todhip_verify_2d (tod_amd/csrc/pnp.hip) must reproduce. What is kept from the reference's 3D branch:
 *   - matches are clustered per object in CSR order (ClusterPerObject, adjacency_ransac.cpp:176-205), objects in ascending index
 *   - a sample is three matches that are pairwise "sample adjacent": keypoints more than 20 px apart
 *     (adjacency_ransac.cpp:484) and model points no farther apart than the object's span (:477, without the sensor-error
 *     slack: there is no measured 3D on the query side to compare with)
 *   - RANSAC keeps the hypothesis with the largest consensus set, first one on ties (ransac.h:112-121); a pose is
 *     reported if its consensus set has at least min_inliers matches (GuessGenerator.cpp:211)
 * and what is this file's own:
 *   - hypotheses: Grunert's P3P on the sample (depth ratios u = s2/s1, v = s3/s1; the quartic in v is assembled by polynomial
 *     arithmetic from the two ratio equations, its real roots found by bracketing between the roots of its derivative and 80
 *     bisection steps), up to four poses per sample, each scored by reprojection error < err_px
 *   - samples are drawn by a counter-based hash of (seed, object, hypothesis, attempt), not by a sequential rand() walk:
 *     one rand() call per frame supplies the seed, every hypothesis is independent of every other (that is what lets
 *     the GPU evaluate them side by side), at most 16 attempts per hypothesis to hit an admissible triple
 *   - a fixed number of hypotheses (n_ransac_iterations), no probabilistic early exit
 *   - the winner is refined by 5 Gauss-Newton steps on its consensus set (left-multiplied small rotation + translation,
 *     re-orthonormalised by Gram-Schmidt), the consensus set is recomputed with the refined pose, one pose per object
 * All arithmetic is IEEE f64 with +, -, *, /, sqrt only and a fixed order of operations (-ffp-contract=off), so the GPU
 * can match it bit for bit.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "tod_oracle.h"

static uint32_t mix32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }

static double horner(const double* c, int deg, double x) {
  double y = c[deg];
  for (int i = deg - 1; i >= 0; --i) y = y * x + c[i];
  return y;
}

/* real roots of c[0] + c[1] x + ... + c[deg] x^deg, ascending, deg <= 4, c[deg] != 0. A root is found in every interval
 * between consecutive roots of the derivative (and out to the Cauchy bound) over which the polynomial changes sign. */
static int real_roots(const double* c, int deg, double* out) {
  if (deg == 1) { out[0] = -c[0] / c[1]; return 1; }
  if (deg == 2) {
    const double disc = c[1] * c[1] - 4.0 * c[2] * c[0];
    if (disc < 0.0) return 0;
    const double s = sqrt(disc), r0 = (-c[1] - s) / (2.0 * c[2]), r1 = (-c[1] + s) / (2.0 * c[2]);
    out[0] = r0 < r1 ? r0 : r1; out[1] = r0 < r1 ? r1 : r0;
    return 2;
  }
  double d[4] = {0, 0, 0, 0}, crit[3];
  for (int i = 1; i <= deg; ++i) d[i - 1] = (double)i * c[i];
  const int nc = real_roots(d, deg - 1, crit);
  double bound = 0.0;
  for (int i = 0; i < deg; ++i) { const double a = fabs(c[i] / c[deg]); if (a > bound) bound = a; }
  bound += 1.0;
  double edge[5];
  int ne = 0;
  edge[ne++] = -bound;
  for (int i = 0; i < nc; ++i) if (crit[i] > -bound && crit[i] < bound) edge[ne++] = crit[i];
  edge[ne++] = bound;
  int n = 0;
  for (int i = 0; i + 1 < ne; ++i) {
    double lo = edge[i], hi = edge[i + 1];
    const double flo = horner(c, deg, lo), fhi = horner(c, deg, hi);
    if (flo == 0.0) { if (n == 0 || out[n - 1] != lo) out[n++] = lo; continue; }
    if ((flo < 0.0) == (fhi < 0.0) || fhi == 0.0) continue;        /* a zero at hi is picked up as the next interval's lo */
    for (int it = 0; it < 80; ++it) {
      const double mid = 0.5 * (lo + hi), fm = horner(c, deg, mid);
      if ((fm < 0.0) == (flo < 0.0)) lo = mid; else hi = mid;
    }
    out[n++] = 0.5 * (lo + hi);
  }
  if (n < deg && horner(c, deg, bound) == 0.0) out[n++] = bound;
  return n;
}

typedef struct { double R[9], t[3]; } pose_d;

static void cross3(const double* a, const double* b, double* o) {
  o[0] = a[1] * b[2] - a[2] * b[1]; o[1] = a[2] * b[0] - a[0] * b[2]; o[2] = a[0] * b[1] - a[1] * b[0];
}
static double dot3(const double* a, const double* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }

/* orthonormal frame of a triangle: e1 along P2 - P1, e3 its normal, e2 = e3 x e1; 0 if the triangle is degenerate */
static int frame3(const double* P1, const double* P2, const double* P3, double* e1, double* e2, double* e3) {
  double a[3] = {P2[0] - P1[0], P2[1] - P1[1], P2[2] - P1[2]}, b[3] = {P3[0] - P1[0], P3[1] - P1[1], P3[2] - P1[2]}, n[3];
  const double la = sqrt(dot3(a, a));
  if (!(la > 0.0)) return 0;
  for (int i = 0; i < 3; ++i) e1[i] = a[i] / la;
  cross3(e1, b, n);
  const double ln = sqrt(dot3(n, n));
  if (!(ln > 1e-12 * sqrt(dot3(b, b)))) return 0;
  for (int i = 0; i < 3; ++i) e3[i] = n[i] / ln;
  cross3(e3, e1, e2);
  return 1;
}

/* Grunert's P3P. X: three model points, f: three unit bearings (camera frame). Up to 4 poses (model -> camera). */
static int p3p(const double X[3][3], const double f[3][3], pose_d* out) {
  double d[3] = {0, 0, 0};
  double a2 = 0, b2 = 0, c2 = 0;
  for (int i = 0; i < 3; ++i) { d[i] = X[1][i] - X[2][i]; } a2 = dot3(d, d);
  for (int i = 0; i < 3; ++i) { d[i] = X[0][i] - X[2][i]; } b2 = dot3(d, d);
  for (int i = 0; i < 3; ++i) { d[i] = X[0][i] - X[1][i]; } c2 = dot3(d, d);
  if (!(a2 > 0.0) || !(b2 > 0.0) || !(c2 > 0.0)) return 0;
  double e1[3], e2[3], e3[3];
  if (!frame3(X[0], X[1], X[2], e1, e2, e3)) return 0;
  const double ca = dot3(f[1], f[2]), cb = dot3(f[0], f[2]), cg = dot3(f[0], f[1]);
  const double q = (a2 - c2) / b2, cb2 = c2 / b2;
  /* u = N(v) / D(v):  N = (q - 1) v^2 - 2 q cb v + (1 + q),  D = 2 (cg - v ca);  W = 1 - 2 cb v + v^2
   * quartic: D^2 + N^2 - 2 cg N D - (c2 / b2) W D^2 = 0 */
  const double N[3] = {1.0 + q, -2.0 * q * cb, q - 1.0}, D[2] = {2.0 * cg, -2.0 * ca}, W[3] = {1.0, -2.0 * cb, 1.0};
  double DD[3] = {D[0] * D[0], 2.0 * D[0] * D[1], D[1] * D[1]};
  double NN[5] = {0, 0, 0, 0, 0}, ND[4] = {0, 0, 0, 0}, WDD[5] = {0, 0, 0, 0, 0}, P[5];
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) NN[i + j] += N[i] * N[j];
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 2; ++j) ND[i + j] += N[i] * D[j];
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) WDD[i + j] += W[i] * DD[j];
  for (int i = 0; i < 5; ++i) P[i] = (i < 3 ? DD[i] : 0.0) + NN[i] - 2.0 * cg * (i < 4 ? ND[i] : 0.0) - cb2 * WDD[i];
  double big = 0.0;
  for (int i = 0; i < 5; ++i) if (fabs(P[i]) > big) big = fabs(P[i]);
  if (!(fabs(P[4]) > 1e-12 * big)) return 0;                      /* degenerate configuration: no hypothesis */
  double roots[4];
  const int nr = real_roots(P, 4, roots);
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
    pose_d* p = &out[n++];
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) p->R[3 * i + j] = g1[i] * e1[j] + g2[i] * e2[j] + g3[i] * e3[j];
    for (int i = 0; i < 3; ++i) p->t[i] = Q[0][i] - (p->R[3 * i] * X[0][0] + p->R[3 * i + 1] * X[0][1] + p->R[3 * i + 2] * X[0][2]);
  }
  return n;
}

typedef struct { double fx, fy, cx, cy; } cam_d;

/* squared reprojection error of model point X under (R, t); negative if the point is not in front of the camera */
static double reproj2(const pose_d* p, const cam_d* k, const float* X, const float* uv) {
  const double x = X[0], y = X[1], z = X[2];
  const double xc = p->R[0] * x + p->R[1] * y + p->R[2] * z + p->t[0];
  const double yc = p->R[3] * x + p->R[4] * y + p->R[5] * z + p->t[1];
  const double zc = p->R[6] * x + p->R[7] * y + p->R[8] * z + p->t[2];
  if (!(zc > 1e-9)) return -1.0;
  const double du = k->fx * (xc / zc) + k->cx - (double)uv[0], dv = k->fy * (yc / zc) + k->cy - (double)uv[1];
  return du * du + dv * dv;
}

/* the three sample-adjacency conditions of a pair of matches (file header) */
static int pair_ok(const float* kp_a, const float* kp_b, const float* Xa, const float* Xb, float span) {
  const float du = kp_a[0] - kp_b[0], dv = kp_a[1] - kp_b[1];
  if (!((du * du + dv * dv) > 20.f * 20.f)) return 0;
  const float dx = Xa[0] - Xb[0], dy = Xa[1] - Xb[1], dz = Xa[2] - Xb[2];
  const float d2 = dx * dx + dy * dy + dz * dz;
  return d2 > 0.f && d2 <= span * span;
}

/* hypothesis `hyp` of object `obj`: the sample (match indices within the object's cluster), 0 if no admissible triple came up */
static int draw_sample(uint32_t seed, uint32_t obj, uint32_t hyp, uint32_t n, const float* kp_xy, const uint32_t* q_idx,
                       const float* X, float span, uint32_t* s3) {
  const uint32_t base = seed ^ mix32(obj * 0x9E3779B9U + hyp);
  for (uint32_t a = 0; a < 16; ++a) {
    const uint32_t h0 = mix32(base + a * 0x85EBCA6BU), h1 = mix32(h0 + 0x68E31DA4U), h2 = mix32(h1 + 0xB5297A4DU);
    const uint32_t i = h0 % n, j = h1 % n, k = h2 % n;
    if (i == j || i == k || j == k) continue;
    if (!pair_ok(kp_xy + 2 * q_idx[i], kp_xy + 2 * q_idx[j], X + 3 * i, X + 3 * j, span)) continue;
    if (!pair_ok(kp_xy + 2 * q_idx[i], kp_xy + 2 * q_idx[k], X + 3 * i, X + 3 * k, span)) continue;
    if (!pair_ok(kp_xy + 2 * q_idx[j], kp_xy + 2 * q_idx[k], X + 3 * j, X + 3 * k, span)) continue;
    s3[0] = i; s3[1] = j; s3[2] = k;
    return 1;
  }
  return 0;
}

static void bearing(const cam_d* k, const float* uv, double* f) {
  const double x = ((double)uv[0] - k->cx) / k->fx, y = ((double)uv[1] - k->cy) / k->fy;
  const double l = sqrt(x * x + y * y + 1.0);
  f[0] = x / l; f[1] = y / l; f[2] = 1.0 / l;
}

/* the poses of one sample, in root order */
static int sample_poses(const cam_d* cam, const float* kp_xy, const uint32_t* q_idx, const float* X, const uint32_t* s3, pose_d* out) {
  double Xd[3][3], f[3][3];
  for (int a = 0; a < 3; ++a) {
    for (int i = 0; i < 3; ++i) Xd[a][i] = X[3 * s3[a] + i];
    bearing(cam, kp_xy + 2 * q_idx[s3[a]], f[a]);
  }
  return p3p(Xd, f, out);
}

static uint32_t count_inliers(const pose_d* p, const cam_d* cam, uint32_t n, const float* kp_xy, const uint32_t* q_idx, const float* X,
                              double err2, uint8_t* flags) {
  uint32_t c = 0;
  for (uint32_t m = 0; m < n; ++m) {
    const double e = reproj2(p, cam, X + 3 * m, kp_xy + 2 * q_idx[m]);
    const int in = e >= 0.0 && e < err2;
    if (flags) flags[m] = (uint8_t)in;
    c += (uint32_t)in;
  }
  return c;
}

/* 6 x 6 solve by Gaussian elimination with partial pivoting; 0 if singular */
static int solve6(double A[6][6], double* b, double* x) {
  for (int c = 0; c < 6; ++c) {
    int piv = c;
    for (int r = c + 1; r < 6; ++r) if (fabs(A[r][c]) > fabs(A[piv][c])) piv = r;
    if (!(fabs(A[piv][c]) > 1e-300)) return 0;
    if (piv != c) { for (int j = 0; j < 6; ++j) { const double t = A[c][j]; A[c][j] = A[piv][j]; A[piv][j] = t; } const double t = b[c]; b[c] = b[piv]; b[piv] = t; }
    for (int r = c + 1; r < 6; ++r) {
      const double m = A[r][c] / A[c][c];
      for (int j = c; j < 6; ++j) A[r][j] -= m * A[c][j];
      b[r] -= m * b[c];
    }
  }
  for (int r = 5; r >= 0; --r) {
    double s = b[r];
    for (int j = r + 1; j < 6; ++j) s -= A[r][j] * x[j];
    x[r] = s / A[r][r];
  }
  return 1;
}

/* 5 Gauss-Newton steps on the flagged matches. The normal equations are summed as 64 interleaved partial sums (match m of the
 * object's cluster goes to partial m mod 64, each partial in ascending m) that are then added up in ascending order: a
 * fixed summation tree that 64 lanes can execute side by side. */
static void refine(pose_d* p, const cam_d* cam, uint32_t n, const float* kp_xy, const uint32_t* q_idx, const float* X, const uint8_t* flags) {
  for (int it = 0; it < 5; ++it) {
    static double Hp[64][6][6], gp[64][6];                        /* (the oracle is single-threaded test infrastructure) */
    double H[6][6], g[6], dlt[6];
    memset(Hp, 0, sizeof(Hp)); memset(gp, 0, sizeof(gp));
    for (uint32_t m = 0; m < n; ++m) {
      if (!flags[m]) continue;
      const double x = X[3 * m], y = X[3 * m + 1], z = X[3 * m + 2];
      const double xc = p->R[0] * x + p->R[1] * y + p->R[2] * z + p->t[0];
      const double yc = p->R[3] * x + p->R[4] * y + p->R[5] * z + p->t[1];
      const double zc = p->R[6] * x + p->R[7] * y + p->R[8] * z + p->t[2];
      if (!(zc > 1e-9)) continue;
      const double iz = 1.0 / zc, xn = xc * iz, yn = yc * iz;
      const double ru = cam->fx * xn + cam->cx - (double)kp_xy[2 * q_idx[m]], rv = cam->fy * yn + cam->cy - (double)kp_xy[2 * q_idx[m] + 1];
      /* d(projection)/d(Xc) */
      const double a0 = cam->fx * iz, a2 = -cam->fx * xn * iz, b1 = cam->fy * iz, b2 = -cam->fy * yn * iz;
      /* d(Xc)/d(omega) = -[Xc]x, d(Xc)/d(dt) = I */
      const double Ju[6] = {a2 * yc, a0 * zc - a2 * xc, -a0 * yc, a0, 0.0, a2};
      const double Jv[6] = {-b1 * zc + b2 * yc, -b2 * xc, b1 * xc, 0.0, b1, b2};
      for (int i = 0; i < 6; ++i) {
        for (int j = 0; j < 6; ++j) Hp[m & 63][i][j] += Ju[i] * Ju[j] + Jv[i] * Jv[j];
        gp[m & 63][i] += Ju[i] * ru + Jv[i] * rv;
      }
    }
    memset(H, 0, sizeof(H)); memset(g, 0, sizeof(g));
    for (int l = 0; l < 64; ++l)
      for (int i = 0; i < 6; ++i) {
        for (int j = 0; j < 6; ++j) H[i][j] += Hp[l][i][j];
        g[i] += gp[l][i];
      }
    for (int i = 0; i < 6; ++i) g[i] = -g[i];
    if (!solve6(H, g, dlt)) return;
    /* R <- (I + [w]x) R, t <- (I + [w]x) t + dt, then Gram-Schmidt on the rows of R */
    double Rn[9], tn[3];
    const double wx = dlt[0], wy = dlt[1], wz = dlt[2];
    for (int j = 0; j < 3; ++j) {
      Rn[j]     = p->R[j]     - wz * p->R[3 + j] + wy * p->R[6 + j];
      Rn[3 + j] = p->R[3 + j] + wz * p->R[j]     - wx * p->R[6 + j];
      Rn[6 + j] = p->R[6 + j] - wy * p->R[j]     + wx * p->R[3 + j];
    }
    tn[0] = p->t[0] - wz * p->t[1] + wy * p->t[2] + dlt[3];
    tn[1] = p->t[1] + wz * p->t[0] - wx * p->t[2] + dlt[4];
    tn[2] = p->t[2] - wy * p->t[0] + wx * p->t[1] + dlt[5];
    double l0 = sqrt(dot3(Rn, Rn));
    for (int j = 0; j < 3; ++j) Rn[j] /= l0;
    const double d01 = dot3(Rn, Rn + 3);
    for (int j = 0; j < 3; ++j) Rn[3 + j] -= d01 * Rn[j];
    double l1 = sqrt(dot3(Rn + 3, Rn + 3));
    for (int j = 0; j < 3; ++j) Rn[3 + j] /= l1;
    cross3(Rn, Rn + 3, Rn + 6);
    memcpy(p->R, Rn, sizeof(Rn)); memcpy(p->t, tn, sizeof(tn));
  }
}

/* Whole frame. kp_xy: nq x 2 pixels; K9 row-major intrinsics; matches in CSR form as orc_match produces them (imgIdx = object,
 * matches_xyz = the model point of each match); spans per object. prm->sensor_error is the reprojection threshold in PIXELS here.
 * Advances rng by one draw. best_hyp/best_count (optional, per object index, n_obj entries): the winning hypothesis and its
 * consensus size before refinement (0xFFFFFFFF / 0 if the object had no admissible hypothesis). */
int orc_verify_2d(const float* kp_xy, uint32_t nq, const float* K9, const uint32_t* row_ptr, const orc_dmatch* matches,
                  const float* matches_xyz, const float* spans, uint32_t n_obj, const orc_verify_params* prm, orc_rng* rng,
                  orc_pose* poses, uint32_t* n_poses, uint32_t* inlier_kp, uint32_t* n_inlier_kp, uint32_t* best_hyp,
                  uint32_t* best_count) {
  const cam_d cam = {K9[0], K9[4], K9[2], K9[5]};
  const uint32_t seed = (uint32_t)orc_rng_next(rng);
  const uint32_t n_matches = row_ptr[nq];
  const double err2 = (double)prm->sensor_error * (double)prm->sensor_error;
  const uint32_t cap_poses = *n_poses, cap_inl = *n_inlier_kp;
  uint32_t np = 0, ni = 0;
  /* ClusterPerObject: per object, matches in CSR order */
  uint32_t* cnt = (uint32_t*)calloc(n_obj + 1, sizeof(uint32_t));
  for (uint32_t m = 0; m < n_matches; ++m) cnt[matches[m].imgIdx + 1]++;
  for (uint32_t o = 0; o < n_obj; ++o) cnt[o + 1] += cnt[o];
  uint32_t* fill = (uint32_t*)malloc((n_obj + 1) * sizeof(uint32_t));
  memcpy(fill, cnt, (n_obj + 1) * sizeof(uint32_t));
  uint32_t* q_idx = (uint32_t*)malloc((n_matches + 1) * sizeof(uint32_t));
  float* X = (float*)malloc((3 * (size_t)n_matches + 3) * sizeof(float));
  for (uint32_t q = 0; q < nq; ++q)
    for (uint32_t m = row_ptr[q]; m < row_ptr[q + 1]; ++m) {
      const uint32_t at = fill[matches[m].imgIdx]++;
      q_idx[at] = q; memcpy(X + 3 * at, matches_xyz + 3 * m, 3 * sizeof(float));
    }
  uint8_t* flags = (uint8_t*)malloc(n_matches + 1);
  int rc = 0;
  for (uint32_t o = 0; o < n_obj; ++o) {
    if (best_hyp) { best_hyp[o] = 0xFFFFFFFFU; best_count[o] = 0; }
    const uint32_t n = cnt[o + 1] - cnt[o];
    if (n < 3 || n < prm->min_inliers) continue;
    const uint32_t* qi = q_idx + cnt[o];
    const float* Xo = X + 3 * (size_t)cnt[o];
    uint32_t bh = 0xFFFFFFFFU, bs = 0, bc = 0;
    for (uint32_t h = 0; h < prm->n_ransac_iterations; ++h) {
      uint32_t s3[3];
      if (!draw_sample(seed, o, h, n, kp_xy, qi, Xo, spans[o], s3)) continue;
      pose_d sol[4];
      const int ns = sample_poses(&cam, kp_xy, qi, Xo, s3, sol);
      for (int s = 0; s < ns; ++s) {
        const uint32_t c = count_inliers(&sol[s], &cam, n, kp_xy, qi, Xo, err2, NULL);
        if (c > bc) { bc = c; bh = h; bs = (uint32_t)s; }
      }
    }
    if (best_hyp) { best_hyp[o] = bh; best_count[o] = bc; }
    if (bh == 0xFFFFFFFFU || bc < prm->min_inliers || bc < 3) continue;
    uint32_t s3[3];
    pose_d sol[4];
    draw_sample(seed, o, bh, n, kp_xy, qi, Xo, spans[o], s3);
    sample_poses(&cam, kp_xy, qi, Xo, s3, sol);
    pose_d p = sol[bs];
    count_inliers(&p, &cam, n, kp_xy, qi, Xo, err2, flags);
    refine(&p, &cam, n, kp_xy, qi, Xo, flags);
    const uint32_t c = count_inliers(&p, &cam, n, kp_xy, qi, Xo, err2, flags);
    if (c < prm->min_inliers) continue;
    if (np >= cap_poses) { rc = -2; break; }
    orc_pose* out = &poses[np];
    out->object = o;
    for (int i = 0; i < 9; ++i) out->R[i] = (float)p.R[i];
    for (int i = 0; i < 3; ++i) out->t[i] = (float)p.t[i];
    out->inlier_begin = ni;
    uint32_t last = 0xFFFFFFFFU;
    for (uint32_t m = 0; m < n; ++m) {
      if (!flags[m] || qi[m] == last) continue;                    /* keypoint indices, ascending, each once */
      if (ni >= cap_inl) { rc = -2; break; }
      inlier_kp[ni++] = last = qi[m];
    }
    if (rc) break;
    out->inlier_end = ni;
    ++np;
  }
  free(cnt); free(fill); free(q_idx); free(X); free(flags);
  *n_poses = np; *n_inlier_kp = ni;
  return rc;
}
