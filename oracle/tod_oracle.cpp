/*
 * oracle/tod_oracle.cpp -- TEST INFRASTRUCTURE ONLY (see tod_oracle.h).
 *
 * CPU restatement of the detection hot path of wg-perception/tod, written from the
 * behaviour of the reference sources (cited as file:line relative to the reference root).
 * It deliberately keeps the reference's data structures (sorted neighbour lists, binary
 * search adjacency tests, recursive branch and bound) so that it doubles as the timed
 * "port" CPU baseline. Nothing here is used by the product path.
 *
 * Declared decisions:
 *   D1 exact Hamming k-NN, ties by ascending global row (replaces FLANN-LSH, DescriptorMatcher.cpp:211)
 *   D2 consensus ignores R,T: threshold_ is DBL_MAX (sac.h:69-70), so threshold^2 = +inf
 *   D3 colour stack = fixed-capacity array + top; read at top==0 -> 0, pop at top==0 -> no-op
 *   D4 rand() = explicit glibc TYPE_3 state passed in/out
 */
#include "tod_oracle.h"

#include <algorithm>
#include <cfloat>
#include <climits>
#include <cmath>
#include <cstring>
#include <limits>
#include <map>
#include <unordered_set>
#include <utility>
#include <vector>

namespace orc {

typedef unsigned int Index;
typedef std::vector<Index> IndexVector;
struct P3 { float v[3]; };

/* ======================================================================================== */
/* glibc rand(): random_r TYPE_3, degree 31, separation 3; srandom_r LCG seeding + 310 discards.
 * Call site: sac_model_registration_graph.h:111; never seeded: sac.h:71.                     */
static void rng_seed(orc_rng& r, uint32_t seed) {
  if (seed == 0) seed = 1;
  int32_t word = (int32_t)seed;
  r.s[0] = seed;
  for (int i = 1; i < 31; ++i) {
    long hi = word / 127773, lo = word % 127773;
    word = (int32_t)(16807 * lo - 2836 * hi);
    if (word < 0) word += 2147483647;
    r.s[i] = (uint32_t)word;
  }
  r.f = 3; r.b = 0; r.draws = 0;
  for (int i = 0; i < 310; ++i) {
    r.s[r.f] += r.s[r.b];
    r.f = (r.f + 1) % 31; r.b = (r.b + 1) % 31;
  }
}
static inline int32_t rng_next(orc_rng& r) {
  r.s[r.f] += r.s[r.b];
  uint32_t out = r.s[r.f] >> 1;
  r.f = (r.f + 1) % 31; r.b = (r.b + 1) % 31;
  ++r.draws;
  return (int32_t)out;
}

/* ======================================================================================== */
/* maximum_clique.h:52-148, maximum_clique.cpp:49-145 -- sorted neighbour lists               */
class SortedAdjacency {
 public:
  SortedAdjacency() {}
  explicit SortedAdjacency(Index n) : rows_(n) {}
  size_t size() const { return rows_.size(); }
  bool empty() const { return rows_.empty(); }
  /* set_sorted, maximum_clique.h:85-90 */
  void append_pair(Index i, Index j) { rows_[i].push_back(j); rows_[j].push_back(i); }
  /* set + SetOneWay, maximum_clique.cpp:128-133, maximum_clique.h:120-145 */
  void insert_pair(Index i, Index j) { insert_one(i, j); insert_one(j, i); }
  /* invalidate(i,j) + InvalidateOneWay, maximum_clique.cpp:115-120, maximum_clique.h:111-118 */
  void erase_pair(Index i, Index j) { erase_one(i, j); erase_one(j, i); }
  /* test, maximum_clique.cpp:122-126 */
  bool has(Index i, Index j) const { return std::binary_search(rows_[i].begin(), rows_[i].end(), j); }
  size_t count(Index i) const { return rows_[i].size(); }
  const IndexVector& row(Index i) const { return rows_[i]; }
  /* InvalidateCluster, maximum_clique.cpp:59-86 */
  void drop_cluster(const IndexVector& gone) {
    std::unordered_set<Index> done;
    for (size_t a = 0; a < gone.size(); ++a) {
      IndexVector& r = rows_[gone[a]];
      r.resize(std::set_difference(r.begin(), r.end(), gone.begin(), gone.end(), r.begin()) - r.begin());
      for (size_t b = 0; b < r.size(); ++b) {
        Index sub = r[b];
        if (done.count(sub)) continue;
        IndexVector& sr = rows_[sub];
        sr.resize(std::set_difference(sr.begin(), sr.end(), gone.begin(), gone.end(), sr.begin()) - sr.begin());
        done.insert(sub);
      }
      rows_[gone[a]].clear();
    }
  }

 private:
  void insert_one(Index i, Index j) {
    IndexVector& r = rows_[i];
    IndexVector::iterator it = std::lower_bound(r.begin(), r.end(), j);
    if (it != r.end() && *it == j) return;
    r.insert(it, j);
  }
  void erase_one(Index i, Index j) {
    IndexVector& r = rows_[i];
    IndexVector::iterator it = std::lower_bound(r.begin(), r.end(), j);
    if (it != r.end() && *it == j) r.erase(it); /* the reference erases unconditionally (UB if absent) */
  }
  std::vector<IndexVector> rows_;
};

/* ======================================================================================== */
/* Decision D3: `Colors C` is one vector shared by reference across all recursion levels
 * (maximum_clique.cpp:287,315,320). Children pop the parent's entries (:334), ColorSort writes by
 * absolute position (:247-260). Modelled as capacity-n storage plus a moving top.            */
struct ColourStack {
  std::vector<unsigned> cell;
  size_t top;
  unsigned underruns;
  explicit ColourStack(size_t n) : cell(n, 0u), top(n), underruns(0) {}
  unsigned back() { if (top == 0) { ++underruns; return 0u; } return cell[top - 1]; }
  void pop() { if (top == 0) { ++underruns; return; } --top; }
};

/* maximum_clique.h:152-274, maximum_clique.cpp:202-375 */
class CliqueGraph {
 public:
  explicit CliqueGraph(Index n) : adj_(n), all_steps_(0), t_limit_(0), underruns(0) {}
  SortedAdjacency adj_;
  int all_steps_;
  double t_limit_;
  unsigned underruns;

  /* FindClique, maximum_clique.cpp:343-369 */
  void find_clique(IndexVector& QMax, unsigned minimal_size) {
    if (adj_.empty()) return;
    all_steps_ = 1;
    t_limit_ = 0.025;
    const Index n = (Index)adj_.size();
    IndexVector R(n);
    for (Index i = 0; i < n; ++i) R[i] = i;
    degree_sort(R);
    unsigned max_degree = (unsigned)adj_.count(R[0]);
    ColourStack C(n);
    for (unsigned i = 0; i < max_degree; ++i) C.cell[i] = i + 1;
    for (unsigned i = max_degree; i < n; ++i) C.cell[i] = max_degree + 1;
    IndexVector Q;
    QMax.clear();
    std::vector<unsigned> S(n + 1, 0u), SOld(n + 1, 0u);
    expand(R, C, 1, minimal_size, QMax, Q, S, SOld);
    underruns = C.underruns;
  }

 private:
  /* IsIntersecting, maximum_clique.h:234-257 (the #else branch) */
  bool touches(Index p, const IndexVector& cls) const {
    for (size_t i = 0; i < cls.size(); ++i)
      if (adj_.has(p, cls[i])) return true;
    return false;
  }
  /* Intersection, maximum_clique.cpp:209-217 */
  bool neighbours_in(Index p, const IndexVector& R, IndexVector& out) const {
    out.clear();
    for (size_t i = 0; i < R.size(); ++i)
      if (adj_.has(p, R[i])) out.push_back(R[i]);
    return !out.empty();
  }
  /* DegreeSort, maximum_clique.cpp:263-284: (degree within R, vertex) ascending, then reversed */
  void degree_sort(IndexVector& R) const {
    const unsigned n = (unsigned)R.size();
    std::vector<std::pair<unsigned, Index> > deg(n);
    for (unsigned i = 0; i < n; ++i) {
      deg[i] = std::make_pair(0u, R[i]);
      for (unsigned j = 0; j < i; ++j)
        if (adj_.has(R[i], R[j])) { ++deg[i].first; ++deg[j].first; }
    }
    std::sort(deg.begin(), deg.end());
    for (unsigned i = 0; i < n; ++i) R[i] = deg[n - 1 - i].second;
  }
  /* ColorSort, maximum_clique.cpp:219-261. Note a vertex whose class k < min_k is NOT added to class k. */
  void colour_sort(IndexVector& R, ColourStack& C, const IndexVector& QMax, const IndexVector& Q) const {
    unsigned min_k = (unsigned)std::max(1, int(QMax.size()) - int(Q.size()) + 1);
    std::vector<IndexVector> cls(2);
    unsigned j = 0;
    unsigned maxno = (unsigned)cls.size();
    for (size_t i = 0; i < R.size(); ++i) {
      Index p = R[i];
      unsigned k = 1;
      while (touches(p, cls[k])) {
        ++k;
        if (k >= maxno) { ++maxno; cls.resize(maxno); break; }
      }
      if (k < min_k) R[j++] = p; else cls[k].push_back(p);
    }
    if (j > 0) C.cell[j - 1] = 0;
    size_t pos = j;
    for (unsigned k = min_k; k < maxno; ++k)
      for (size_t i = 0; i < cls[k].size(); ++i) { R[pos] = cls[k][i]; C.cell[pos] = k; ++pos; }
  }
  /* MaxCliqueDyn, maximum_clique.cpp:286-336 */
  void expand(IndexVector& R, ColourStack& C, unsigned level, unsigned minimal_size, IndexVector& QMax,
              IndexVector& Q, std::vector<unsigned>& S, std::vector<unsigned>& SOld) {
    if (QMax.size() >= minimal_size) return;
    if (level >= S.size()) { S.resize(S.size() + 1); SOld.resize(SOld.size() + 1); }
    S[level] = S[level] + S[level - 1] - SOld[level];
    SOld[level] = S[level - 1];
    while (!R.empty()) {
      Index p = R.back();
      unsigned c = C.back();
      if (Q.size() + c > QMax.size()) {
        Q.push_back(p);
        IndexVector Rp;
        if (neighbours_in(p, R, Rp)) {
          if ((double)S[level] / all_steps_ < t_limit_) degree_sort(Rp);
          colour_sort(Rp, C, QMax, Q);
          ++S[level];
          ++all_steps_;
          if (all_steps_ > 100000) return;          /* :318-319 returns without popping Q */
          expand(Rp, C, level + 1, minimal_size, QMax, Q, S, SOld);
        } else if (Q.size() > QMax.size()) {
          QMax = Q;
          if (QMax.size() >= minimal_size) return;  /* :325-326 */
        }
        Q.pop_back();
      } else {
        return;
      }
      R.pop_back();
      C.pop();
    }
  }
};

/* ======================================================================================== */
/* float helpers with the reference's precision (App. A Q5, Q12)                              */
static inline float dist_sq(const P3& a, const P3& b) { /* sac_model_registration_graph.h:52-58 */
  float t0 = a.v[0] - b.v[0], t1 = a.v[1] - b.v[1], t2 = a.v[2] - b.v[2];
  return t0 * t0 + t1 * t1 + t2 * t2;
}
static inline double norm3(const P3& a) { /* cv::norm(Vec3f): squares accumulated in double */
  return std::sqrt((double)a.v[0] * a.v[0] + (double)a.v[1] * a.v[1] + (double)a.v[2] * a.v[2]);
}
static inline P3 sub3(const P3& a, const P3& b) { P3 r = {{a.v[0] - b.v[0], a.v[1] - b.v[1], a.v[2] - b.v[2]}}; return r; }
struct M33 { float m[3][3]; };
static inline P3 mulMV(const M33& R, const P3& p) { /* Matx33f * Vec3f: float accumulate, k ascending */
  P3 r;
  for (int i = 0; i < 3; ++i) {
    float s = 0;
    for (int k = 0; k < 3; ++k) s += R.m[i][k] * p.v[k];
    r.v[i] = s;
  }
  return r;
}

/* 3x3 one-sided Jacobi SVD in float: A = U diag(w) Vt, w descending.
 * Stand-in for cv::SVD on a CV_32F 3x3 (sac_model_registration_graph.h:333), which is third-party
 * arithmetic absent from the reference tree: PARITY UNPINNED, absorbed by the 1e-3 pose tolerance. */
static void svd3(const float Ain[3][3], float U[3][3], float w[3], float Vt[3][3]) {
  float A[3][3], V[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
  std::memcpy(A, Ain, sizeof(A));
  for (int sweep = 0; sweep < 30; ++sweep) {
    bool rotated = false;
    for (int p = 0; p < 2; ++p)
      for (int q = p + 1; q < 3; ++q) {
        float alpha = 0, beta = 0, gamma = 0;
        for (int i = 0; i < 3; ++i) { alpha += A[i][p] * A[i][p]; beta += A[i][q] * A[i][q]; gamma += A[i][p] * A[i][q]; }
        if (std::fabs(gamma) <= FLT_EPSILON * std::sqrt(alpha * beta) || gamma == 0.f) continue;
        rotated = true;
        float zeta = (beta - alpha) / (2.f * gamma);
        float t = (zeta >= 0.f ? 1.f : -1.f) / (std::fabs(zeta) + std::sqrt(1.f + zeta * zeta));
        float c = 1.f / std::sqrt(1.f + t * t), s = c * t;
        for (int i = 0; i < 3; ++i) {
          float ap = A[i][p], aq = A[i][q];
          A[i][p] = c * ap - s * aq; A[i][q] = s * ap + c * aq;
          float vp = V[i][p], vq = V[i][q];
          V[i][p] = c * vp - s * vq; V[i][q] = s * vp + c * vq;
        }
      }
    if (!rotated) break;
  }
  float nrm[3];
  int order[3] = {0, 1, 2};
  for (int j = 0; j < 3; ++j) nrm[j] = std::sqrt(A[0][j] * A[0][j] + A[1][j] * A[1][j] + A[2][j] * A[2][j]);
  for (int a = 0; a < 2; ++a)
    for (int b = 0; b < 2 - a; ++b)
      if (nrm[order[b]] < nrm[order[b + 1]]) std::swap(order[b], order[b + 1]);
  for (int jj = 0; jj < 3; ++jj) {
    int j = order[jj];
    w[jj] = nrm[j];
    for (int i = 0; i < 3; ++i) { Vt[jj][i] = V[i][j]; U[i][jj] = nrm[j] > 0.f ? A[i][j] / nrm[j] : 0.f; }
  }
  /* complete U for (numerically) zero singular values so that U stays orthonormal */
  const float tiny = FLT_EPSILON * (w[0] > 0.f ? w[0] : 1.f);
  if (w[0] <= 0.f) { for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) U[i][j] = (i == j); return; }
  if (w[1] <= tiny) {
    /* any unit vector orthogonal to u0: cross with the axis of u0's smallest component */
    int ax = 0; for (int i = 1; i < 3; ++i) if (std::fabs(U[i][0]) < std::fabs(U[ax][0])) ax = i;
    float e[3] = {0, 0, 0}; e[ax] = 1.f;
    float c1[3] = {U[1][0] * e[2] - U[2][0] * e[1], U[2][0] * e[0] - U[0][0] * e[2], U[0][0] * e[1] - U[1][0] * e[0]};
    float n1 = std::sqrt(c1[0] * c1[0] + c1[1] * c1[1] + c1[2] * c1[2]);
    for (int i = 0; i < 3; ++i) U[i][1] = c1[i] / n1;
  }
  if (w[2] <= tiny) {
    U[0][2] = U[1][0] * U[2][1] - U[2][0] * U[1][1];
    U[1][2] = U[2][0] * U[0][1] - U[0][0] * U[2][1];
    U[2][2] = U[0][0] * U[1][1] - U[1][0] * U[0][1];
  }
}
static inline float det3f(const float m[3][3]) { /* cv::determinant on a 3x3 CV_32F: float arithmetic */
  return m[0][0] * (m[1][1] * m[2][2] - m[1][2] * m[2][1]) - m[0][1] * (m[1][0] * m[2][2] - m[1][2] * m[2][0]) +
         m[0][2] * (m[1][0] * m[2][1] - m[1][1] * m[2][0]);
}

/* ======================================================================================== */
/* SampleConsensusModelRegistrationGraph, sac_model_registration_graph.h:67-367               */
class GraphRegistrationModel {
 public:
  GraphRegistrationModel(const std::vector<P3>& q, const std::vector<P3>& t, const IndexVector& indices,
                         const SortedAdjacency& phys, const SortedAdjacency& samp, orc_rng* rng)
      : gate_calls(0), last_gate_called(0), last_gate_size(0), phys_(phys), samp_(samp), best_inlier_number_(8),
        query_(q), train_(t), indices_(indices), rng_(rng) {}

  const IndexVector& indices() const { return indices_; }
  IndexVector samples_;
  unsigned gate_calls, last_gate_called, last_gate_size;

  /* drawIndexSampleHelper, :102-132 */
  bool draw_helper(IndexVector& valid, unsigned n_samples) {
    if (n_samples == 0) return true;
    if (valid.empty()) return false;
    while (true) {
      Index sample = valid[(size_t)rng_next(*rng_) % valid.size()];
      IndexVector next(valid.size());
      const IndexVector& nb = samp_.row(sample);
      next.resize(std::set_intersection(valid.begin(), valid.end(), nb.begin(), nb.end(), next.begin()) - next.begin());
      if (draw_helper(next, n_samples - 1)) {
        samples_.push_back(sample);
        return true;
      }
      valid.resize(std::remove(valid.begin(), valid.end(), sample) - valid.begin());
      if (valid.empty()) return false;
    }
  }
  /* getSamples, :141-168 */
  void get_samples(int& iterations, IndexVector& samples) {
    if (indices_.size() < 3) { samples.clear(); iterations = INT_MAX - 1; return; }
    samples.resize(3);
    for (unsigned iter = 0; iter < 1000; ++iter) {   /* max_sample_checks_, :366 */
      IndexVector valid = indices_;
      samples_.clear();
      if (draw_helper(valid, 3)) { samples = samples_; return; }
    }
    samples.clear();
  }
  /* selectWithinDistance, :171-269, with D2: `distSq(R*p+T, t) < threshold*threshold` (:197) is
   * `finite < +inf`; it is false only for non-finite operands, which is kept here as a finiteness test. */
  void select_within_distance(IndexVector& inliers) {
    last_gate_called = 0; last_gate_size = 0;
    if (samples_.empty()) return;
    IndexVector possible = phys_.row(samples_[0]);
    for (size_t i = 1; i < samples_.size(); ++i) {
      const IndexVector& nb = phys_.row(samples_[i]);
      possible.resize(std::set_intersection(possible.begin(), possible.end(), nb.begin(), nb.end(), possible.begin()) -
                      possible.begin());
    }
    for (size_t i = 0; i < samples_.size(); ++i) possible.push_back(samples_[i]);
    bool model_finite = true;
    for (size_t i = 0; i < samples_.size(); ++i) model_finite = model_finite && finite_pt(samples_[i]);
    inliers.resize(possible.size());
    int nr = 0;
    for (size_t i = 0; i < possible.size(); ++i)
      if (model_finite && finite_pt(possible[i])) inliers[nr++] = possible[i];
    inliers.resize(nr);

    size_t minimal_size = std::min(best_inlier_number_, size_t(7));   /* always 7: :85,:203,:268 */
    if (inliers.size() <= minimal_size) return;
    IndexVector filtered;
    for (size_t j = 0; j < inliers.size(); ++j)
      if (samp_.row(inliers[j]).size() >= minimal_size) filtered.push_back(inliers[j]);
    if (filtered.size() <= minimal_size) { inliers.clear(); return; }
    std::sort(filtered.begin(), filtered.end());
    size_t max_possible = 0;
    IndexVector nbuf(filtered.size());
    for (size_t a = 0; a < filtered.size(); ++a) {
      const IndexVector& nb = samp_.row(filtered[a]);
      max_possible = size_t(std::set_intersection(nb.begin(), nb.end(), filtered.begin(), filtered.end(), nbuf.begin()) -
                            nbuf.begin());
      if (max_possible > minimal_size) break;
    }
    if (max_possible <= minimal_size) { inliers.clear(); return; }
    std::map<unsigned, unsigned> to_graph;
    for (unsigned j = 0; j < filtered.size(); ++j) to_graph[filtered[j]] = j;
    CliqueGraph graph((Index)filtered.size());
    for (unsigned j = 0; j + 1 < filtered.size(); ++j) {
      const IndexVector& nb = samp_.row(filtered[j]);
      nbuf.resize(filtered.size());
      nbuf.resize(std::set_intersection(nb.begin(), nb.end(), filtered.begin() + j + 1, filtered.end(), nbuf.begin()) -
                  nbuf.begin());
      for (size_t b = 0; b < nbuf.size(); ++b) graph.adj_.append_pair(j, to_graph[nbuf[b]]);
    }
    IndexVector clique;
    graph.find_clique(clique, (unsigned)minimal_size);
    ++gate_calls; last_gate_called = 1; last_gate_size = (unsigned)clique.size();
    if (clique.size() <= minimal_size) { inliers.clear(); return; }
    std::sort(inliers.begin(), inliers.end());
    best_inlier_number_ = std::max(inliers.size(), best_inlier_number_);
  }
  /* estimateRigidTransformationSVD, :304-347 (maps query -> training) */
  bool kabsch(const IndexVector& idx, M33& R, P3& T) const {
    if (idx.size() < 3) return false;
    P3 ct = {{0, 0, 0}}, cq = {{0, 0, 0}};
    for (size_t i = 0; i < idx.size(); ++i)
      for (int c = 0; c < 3; ++c) { ct.v[c] += train_[idx[i]].v[c]; cq.v[c] += query_[idx[i]].v[c]; }
    /* Vec /= float: multiply by the double reciprocal, round to float (OpenCV 2.4 operations.hpp; recalled) */
    double inv = 1. / float(idx.size());
    for (int c = 0; c < 3; ++c) { ct.v[c] = (float)(ct.v[c] * inv); cq.v[c] = (float)(cq.v[c] * inv); }
    /* H = sub_training^T * sub_query: cv::gemm on CV_32F accumulates each entry in double, k ascending */
    double Hd[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
    for (size_t i = 0; i < idx.size(); ++i) {
      P3 a = sub3(train_[idx[i]], ct), b = sub3(query_[idx[i]], cq);
      for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) Hd[r][c] += (double)a.v[r] * (double)b.v[c];
    }
    float H[3][3], U[3][3], w[3], Vt[3][3];
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) H[r][c] = (float)Hd[r][c];
    svd3(H, U, w, Vt);
    if (det3f(U) * det3f(Vt) < 0)
      for (int x = 0; x < 3; ++x) Vt[2][x] *= -1;
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) {
        double s = 0;
        for (int k = 0; k < 3; ++k) s += (double)U[r][k] * (double)Vt[k][c];
        R.m[r][c] = (float)s;
      }
    P3 rc = mulMV(R, cq);
    T = sub3(ct, rc);
    return true;
  }


 private:
  bool finite_pt(Index i) const {
    for (int c = 0; c < 3; ++c)
      if (!std::isfinite(query_[i].v[c]) || !std::isfinite(train_[i].v[c])) return false;
    return true;
  }
  const SortedAdjacency& phys_;
  const SortedAdjacency& samp_;
  size_t best_inlier_number_;
  const std::vector<P3>& query_;
  const std::vector<P3>& train_;
  IndexVector indices_;
  orc_rng* rng_;
};

/* ======================================================================================== */
/* tod::AdjacencyRansac, adjacency_ransac.h:48-133, adjacency_ransac.cpp:51-172,234-309        */
class MatchCluster {
 public:
  std::vector<P3> query_, train_;
  IndexVector query_idx_, valid_;
  SortedAdjacency phys_, samp_;

  /* AddPoints, adjacency_ransac.cpp:51-59 */
  void add(const P3& t, const P3& q, unsigned qi) {
    valid_.push_back((Index)query_idx_.size());
    train_.push_back(t); query_.push_back(q); query_idx_.push_back(qi);
  }
  /* FillAdjacency, adjacency_ransac.cpp:127-172 */
  void fill(const float* kp_xy, float span, float err) {
    const unsigned n = (unsigned)train_.size();
    phys_ = SortedAdjacency(n);
    samp_ = SortedAdjacency(n);
    for (unsigned i = 0; i < n; ++i)
      for (unsigned j = i + 1; j < n; ++j) {
        float dq = dist_sq(query_[i], query_[j]);
        if (dq > (span + 2 * err) * (span + 2 * err)) continue;
        dq = std::sqrt(dq);
        float dt = (float)norm3(sub3(train_[i], train_[j]));
        if (std::abs(dt - dq) > 4 * err) continue;
        phys_.append_pair(i, j);
        const float* k1 = kp_xy + 2 * query_idx_[i];
        const float* k2 = kp_xy + 2 * query_idx_[j];
        if ((((k1[0] - k2[0]) * (k1[0] - k2[0]) + (k1[1] - k2[1]) * (k1[1] - k2[1])) > 20 * 20) &&
            (std::abs(dt - dq) < 2 * err))
          samp_.append_pair(i, j);
      }
    invalidate(IndexVector());   /* :169-171 -- a no-op because of the `while (!empty)` at :68 */
  }
  /* InvalidateIndices, adjacency_ransac.cpp:63-89 */
  void invalidate(const IndexVector& in) {
    IndexVector gone = in;
    while (!gone.empty()) {
      std::sort(gone.begin(), gone.end());
      gone.resize(std::unique(gone.begin(), gone.end()) - gone.begin());
      valid_.resize(std::set_difference(valid_.begin(), valid_.end(), gone.begin(), gone.end(), valid_.begin()) -
                    valid_.begin());
      phys_.drop_cluster(gone);
      samp_.drop_cluster(gone);
      gone.clear();
      for (size_t i = 0; i < valid_.size(); ++i)
        if (samp_.row(valid_[i]).size() < 3) gone.push_back(valid_[i]);   /* min_sample_size_ = 3 */
    }
  }
  /* InvalidateQueryIndices, adjacency_ransac.cpp:93-123 (Q15: the read at iter==end is "no match") */
  void invalidate_keypoints(IndexVector& kp) {
    if (kp.empty()) return;
    std::sort(kp.begin(), kp.end());
    kp.resize(std::unique(kp.begin(), kp.end()) - kp.begin());
    IndexVector to_remove;
    size_t it = 0;
    const size_t end = kp.size();
    for (size_t a = 0; a < valid_.size(); ++a) {
      Index index = valid_[a];
      unsigned qi = query_idx_[index];
      if (qi < kp[it]) continue;
      while (it != end && qi > kp[it]) ++it;
      if (it != end && qi == kp[it]) { to_remove.push_back(index); continue; }
      if (it == end) break;
    }
    invalidate(to_remove);
  }
  /* Ransac, adjacency_ransac.cpp:234-309, with pcl::RandomSampleConsensus::computeModel (ransac.h:80-143) inlined */
  void ransac(float err, unsigned n_iter, orc_rng* rng, IndexVector& inliers_in, M33& R, P3& T, orc_round_trace* tr,
              IndexVector* model_inliers_out, int32_t* iter_counts, uint32_t* iter_samples) {
    if (tr) { std::memset(tr, 0, sizeof(*tr)); tr->draws_before = tr->draws_after = rng->draws; }
    inliers_in.clear();
    if (valid_.size() < 3) return;
    GraphRegistrationModel model(query_, train_, valid_, phys_, samp_, rng);

    /* ---- computeModel, ransac.h:80-143; max_iterations_ = n_iter (adjacency_ransac.cpp:249), probability_ 0.99 */
    int iterations = 0;
    int n_best = -INT_MAX;
    double k = 1.0;
    IndexVector inliers, selection, best;
    unsigned best_iteration = 0;
    const int max_iterations = (int)n_iter;
    while (iterations < k) {
      model.get_samples(iterations, selection);
      if (selection.empty()) break;
      /* computeModelCoefficients (:271-288) always succeeds for 3 samples; its R,T are dead (D2) */
      model.select_within_distance(inliers);
      int n_count = (int)inliers.size();
      if (iterations >= 0 && iterations <= max_iterations + 1) {
        if (iter_counts) iter_counts[iterations] = n_count;
        if (iter_samples) for (int s = 0; s < 3; ++s) iter_samples[3 * iterations + s] = selection[s];
      }
      if (n_count > n_best) {
        n_best = n_count;
        best = inliers;
        best_iteration = (unsigned)iterations;
        double w = (double)n_best / (double)model.indices().size();
        double p_no_outliers = 1.0 - std::pow(w, (double)selection.size());
        p_no_outliers = std::max(std::numeric_limits<double>::epsilon(), p_no_outliers);
        p_no_outliers = std::min(1.0 - std::numeric_limits<double>::epsilon(), p_no_outliers);
        k = std::log(1.0 - 0.99) / std::log(p_no_outliers);
      }
      ++iterations;
      if (iterations > max_iterations) break;
    }
    if (tr) {
      tr->iterations = (uint32_t)iterations; tr->best_iteration = best_iteration; tr->best_count = n_best;
      tr->draws_after = rng->draws; tr->n_model_inliers = (uint32_t)best.size();
    }
    if (best.empty()) return;                       /* computeModel() == false, adjacency_ransac.cpp:252 */

    /* ---- growth, adjacency_ransac.cpp:255-303 */
    IndexVector grown = best;
    std::sort(grown.begin(), grown.end());
    if (model_inliers_out) *model_inliers_out = grown;
    IndexVector rest = valid_;
    rest.resize(std::set_difference(rest.begin(), rest.end(), grown.begin(), grown.end(), rest.begin()) - rest.begin());
    bool do_final = false;
    double thresh = err * err;                      /* float product widened, :267 */
    unsigned passes = 0;
    while (true) {
      model.kabsch(grown, R, T);
      ++passes;
      IndexVector extra;
      for (size_t a = 0; a < rest.size(); ++a) {
        Index index = rest[a];
        P3 ptr = mulMV(R, query_[index]);
        for (int c = 0; c < 3; ++c) ptr.v[c] = ptr.v[c] + T.v[c];
        double nn = norm3(sub3(ptr, train_[index]));
        if (nn * nn < thresh) extra.push_back(index);
      }
      IndexVector merged(grown.size() + extra.size());
      std::merge(grown.begin(), grown.end(), extra.begin(), extra.end(), merged.begin());
      grown.swap(merged);
      rest.resize(std::set_difference(rest.begin(), rest.end(), extra.begin(), extra.end(), rest.begin()) - rest.begin());
      if (do_final) break;
      if (extra.empty()) { do_final = true; thresh *= 4; }
    }
    /* pose inversion, :304-305 */
    M33 Rt;
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) Rt.m[r][c] = R.m[c][r];
    M33 negRt;
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) negRt.m[r][c] = -Rt.m[r][c];
    T = mulMV(negRt, T);
    R = Rt;
    for (size_t i = 0; i < grown.size(); ++i) inliers_in.push_back(query_idx_[grown[i]]);
    std::sort(inliers_in.begin(), inliers_in.end());
    inliers_in.resize(std::unique(inliers_in.begin(), inliers_in.end()) - inliers_in.begin());
    if (tr) { tr->n_final_inliers = (uint32_t)grown.size(); tr->growth_passes = passes; }
  }
};

}  // namespace orc

/* ========================================================================================== */
/* C interface                                                                                 */
using namespace orc;

struct orc_cluster { MatchCluster c; };

extern "C" {

void orc_rng_seed(orc_rng* r, uint32_t seed) { rng_seed(*r, seed); }
int32_t orc_rng_next(orc_rng* r) { return rng_next(*r); }

/* span of each object = diagonal of the axis-aligned bounding box, DescriptorMatcher.cpp:104-121 */
void orc_spans(const float* pts, const uint32_t* obj_off, uint32_t n_obj, float* spans) {
  for (uint32_t o = 0; o < n_obj; ++o) {
    float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    for (uint32_t i = obj_off[o]; i < obj_off[o + 1]; ++i)
      for (int c = 0; c < 3; ++c) { mn[c] = std::min(mn[c], pts[3 * i + c]); mx[c] = std::max(mx[c], pts[3 * i + c]); }
    float s = (mx[0] - mn[0]) * (mx[0] - mn[0]) + (mx[1] - mn[1]) * (mx[1] - mn[1]) + (mx[2] - mn[2]) * (mx[2] - mn[2]);
    spans[o] = std::sqrt(s);
  }
}

static inline uint32_t hamming(const uint8_t* a, const uint8_t* b, uint32_t nbytes) {
  uint32_t d = 0, i = 0;
  for (; i + 8 <= nbytes; i += 8) {
    uint64_t x, y;
    std::memcpy(&x, a + i, 8); std::memcpy(&y, b + i, 8);
    d += (uint32_t)__builtin_popcountll(x ^ y);
  }
  for (; i < nbytes; ++i) d += (uint32_t)__builtin_popcount((unsigned)(a[i] ^ b[i]));
  return d;
}

/* D1: exact Hamming k-NN over the concatenated DB, order (distance asc, global row asc) */
void orc_knn_keys(const uint8_t* db, uint64_t n_db, uint32_t desc_bytes, const uint8_t* q, uint32_t nq, uint32_t k,
                  uint64_t* keys) {
  std::vector<uint64_t> best(k);
  for (uint32_t qi = 0; qi < nq; ++qi) {
    uint32_t have = 0;
    const uint8_t* qd = q + (size_t)qi * desc_bytes;
    for (uint64_t r = 0; r < n_db; ++r) {
      uint64_t key = ((uint64_t)hamming(qd, db + r * desc_bytes, desc_bytes) << 32) | r;
      if (have < k) {
        uint32_t p = have++;
        while (p > 0 && best[p - 1] > key) { best[p] = best[p - 1]; --p; }
        best[p] = key;
      } else if (key < best[k - 1]) {
        uint32_t p = k - 1;
        while (p > 0 && best[p - 1] > key) { best[p] = best[p - 1]; --p; }
        best[p] = key;
      }
    }
    for (uint32_t j = 0; j < k; ++j) keys[(size_t)qi * k + j] = j < have ? best[j] : UINT64_MAX;
  }
}

/* DescriptorMatcher::process, DescriptorMatcher.cpp:195-252 with D1 replacing :211.
 * radius is `unsigned int radius_` (:257); radius == 0 skips matching and then indexes an empty
 * vector (:237) -> reported as an error here. */
int orc_match(const uint8_t* db, const uint32_t* obj_off, uint32_t n_obj, const float* db_pts, uint32_t desc_bytes,
              const uint8_t* q, uint32_t nq, uint32_t k, uint32_t radius, uint32_t* row_ptr, orc_dmatch* matches,
              float* xyz) {
  return orc_match_ratio(db, obj_off, n_obj, db_pts, desc_bytes, q, nq, k, radius, 0.f, row_ptr, matches, xyz);
}

/* The ratio test the reference announces and leaves empty (DescriptorMatcher.cpp:223-227, "TODO Perform ratio testing if
 * necessary"; the shipped configs ask for ratio 0.8, conf/detection.ork:39). DEFINITION used here (Lowe's test on the exact
 * neighbours, there is no reference behaviour to follow): with d1 <= d2 the distances of a query's two nearest DB rows
 * (order: distance, then global row), the query keeps its matches iff (float)d1 < ratio * (float)d2; a DB with a single
 * row passes. The test looks at the true two nearest neighbours, whatever the radius; the radius cut (:212-220) then
 * applies to the survivors as before. ratio == 0 switches the test off (the reference's `unsigned int ratio_` turns the
 * shipped 0.8 into exactly that). */
int orc_match_ratio(const uint8_t* db, const uint32_t* obj_off, uint32_t n_obj, const float* db_pts, uint32_t desc_bytes,
                    const uint8_t* q, uint32_t nq, uint32_t k, uint32_t radius, float ratio, uint32_t* row_ptr,
                    orc_dmatch* matches, float* xyz) {
  if (radius == 0 || k == 0) return -1;
  const uint64_t n_db = obj_off[n_obj];
  if (n_db == 0) return -2;                               /* "No descriptors loaded", :204-208 */
  const uint32_t k_all = ratio > 0.f && k < 2 ? 2 : k;
  std::vector<uint64_t> keys_all((size_t)nq * k_all);
  orc_knn_keys(db, n_db, desc_bytes, q, nq, k_all, keys_all.data());
  uint32_t out = 0;
  for (uint32_t qi = 0; qi < nq; ++qi) {
    row_ptr[qi] = out;
    const uint64_t* keys = keys_all.data() + (size_t)qi * k_all - (size_t)qi * k;   /* so that keys[qi * k + j] is this query's j-th */
    if (ratio > 0.f && keys[(size_t)qi * k + 1] != UINT64_MAX) {
      const float d1 = (float)(uint32_t)(keys[(size_t)qi * k] >> 32), d2 = (float)(uint32_t)(keys[(size_t)qi * k + 1] >> 32);
      if (!(d1 < ratio * d2)) continue;                   /* ambiguous: the query keeps nothing */
    }
    for (uint32_t j = 0; j < k; ++j) {
      uint64_t key = keys[(size_t)qi * k + j];
      if (key == UINT64_MAX) break;
      float dist = (float)(uint32_t)(key >> 32);
      if (dist > (float)radius) break;                    /* :212-220, float vs unsigned compare */
      uint32_t row = (uint32_t)key;
      uint32_t obj = (uint32_t)(std::upper_bound(obj_off, obj_off + n_obj + 1, row) - obj_off) - 1;
      orc_dmatch m = {(int32_t)qi, (int32_t)(row - obj_off[obj]), (int32_t)obj, dist};
      matches[out] = m;
      for (int c = 0; c < 3; ++c) xyz[3 * (size_t)out + c] = db_pts[3 * (size_t)row + c];   /* :231-244 */
      ++out;
    }
  }
  row_ptr[nq] = out;
  return 0;
}

uint32_t orc_clique(uint32_t n, const uint32_t* added, uint32_t n_added, const uint32_t* deleted, uint32_t n_deleted,
                    uint32_t minimal_size, uint32_t* out, uint32_t* underruns, uint32_t* steps) {
  CliqueGraph g(n);
  for (uint32_t e = 0; e < n_added; ++e) g.adj_.insert_pair(added[2 * e], added[2 * e + 1]);     /* AddEdge */
  for (uint32_t e = 0; e < n_deleted; ++e) g.adj_.erase_pair(deleted[2 * e], deleted[2 * e + 1]); /* DeleteEdge */
  IndexVector q;
  g.find_clique(q, minimal_size);
  for (size_t i = 0; i < q.size() && i < n; ++i) out[i] = q[i];
  if (underruns) *underruns = g.underruns;
  if (steps) *steps = (uint32_t)g.all_steps_;
  return (uint32_t)q.size();
}

orc_cluster* orc_cluster_new(const float* t, const float* q, const uint32_t* qi, uint32_t n) {
  orc_cluster* h = new orc_cluster();
  for (uint32_t i = 0; i < n; ++i) {
    P3 tp = {{t[3 * i], t[3 * i + 1], t[3 * i + 2]}}, qp = {{q[3 * i], q[3 * i + 1], q[3 * i + 2]}};
    h->c.add(tp, qp, qi[i]);
  }
  return h;
}
void orc_cluster_free(orc_cluster* h) { delete h; }
void orc_cluster_fill(orc_cluster* h, const float* kp_xy, uint32_t, float span, float err) { h->c.fill(kp_xy, span, err); }
uint32_t orc_cluster_size(const orc_cluster* h) { return (uint32_t)h->c.train_.size(); }
void orc_cluster_bits(const orc_cluster* h, int which, uint64_t* bits, uint32_t wpr) {
  const SortedAdjacency& a = which ? h->c.samp_ : h->c.phys_;
  const uint32_t n = (uint32_t)a.size();
  std::memset(bits, 0, sizeof(uint64_t) * (size_t)n * wpr);
  for (uint32_t i = 0; i < n; ++i)
    for (size_t b = 0; b < a.row(i).size(); ++b) {
      uint32_t j = a.row(i)[b];
      bits[(size_t)i * wpr + (j >> 6)] |= 1ull << (j & 63);
    }
}
uint32_t orc_cluster_valid(const orc_cluster* h, uint32_t* out) {
  for (size_t i = 0; i < h->c.valid_.size(); ++i) out[i] = h->c.valid_[i];
  return (uint32_t)h->c.valid_.size();
}
uint32_t orc_cluster_draw(orc_cluster* h, orc_rng* rng, uint32_t* s3) {
  GraphRegistrationModel m(h->c.query_, h->c.train_, h->c.valid_, h->c.phys_, h->c.samp_, rng);
  int it = 0;
  IndexVector sel;
  m.get_samples(it, sel);
  for (size_t i = 0; i < sel.size(); ++i) s3[i] = sel[i];
  return (uint32_t)sel.size();
}
uint32_t orc_cluster_consensus(orc_cluster* h, const uint32_t* s3, uint32_t* out, uint32_t* gate_called,
                               uint32_t* gate_size) {
  orc_rng dummy;
  rng_seed(dummy, 1);
  GraphRegistrationModel m(h->c.query_, h->c.train_, h->c.valid_, h->c.phys_, h->c.samp_, &dummy);
  m.samples_.assign(s3, s3 + 3);
  IndexVector inl;
  m.select_within_distance(inl);
  for (size_t i = 0; i < inl.size(); ++i) out[i] = inl[i];
  if (gate_called) *gate_called = m.last_gate_called;
  if (gate_size) *gate_size = m.last_gate_size;
  return (uint32_t)inl.size();
}
uint32_t orc_cluster_ransac(orc_cluster* h, float err, uint32_t n_iter, orc_rng* rng, uint32_t* inlier_kp, float* R9,
                            float* T3, orc_round_trace* tr, uint32_t* model_inliers, int32_t* iter_counts,
                            uint32_t* iter_samples) {
  IndexVector inl, mi;
  M33 R; P3 T;
  std::memset(&R, 0, sizeof(R)); std::memset(&T, 0, sizeof(T));
  h->c.ransac(err, n_iter, rng, inl, R, T, tr, model_inliers ? &mi : 0, iter_counts, iter_samples);
  for (size_t i = 0; i < inl.size(); ++i) inlier_kp[i] = inl[i];
  if (model_inliers) for (size_t i = 0; i < mi.size(); ++i) model_inliers[i] = mi[i];
  for (int r = 0; r < 3; ++r) { for (int c = 0; c < 3; ++c) R9[3 * r + c] = R.m[r][c]; T3[r] = T.v[r]; }
  return (uint32_t)inl.size();
}
void orc_cluster_invalidate_kp(orc_cluster* h, const uint32_t* kp, uint32_t n) {
  IndexVector v(kp, kp + n);
  h->c.invalidate_keypoints(v);
}
int orc_cluster_kabsch(const orc_cluster* h, const uint32_t* idx, uint32_t n, float* R9, float* T3) {
  orc_rng dummy;
  rng_seed(dummy, 1);
  GraphRegistrationModel m(h->c.query_, h->c.train_, h->c.valid_, h->c.phys_, h->c.samp_, &dummy);
  M33 R; P3 T;
  IndexVector v(idx, idx + n);
  if (!m.kabsch(v, R, T)) return -1;
  for (int r = 0; r < 3; ++r) { for (int c = 0; c < 3; ++c) R9[3 * r + c] = R.m[r][c]; T3[r] = T.v[r]; }
  return 0;
}

/* GuessGenerator::process, GuessGenerator.cpp:127-250 (+ ClusterPerObject, adjacency_ransac.cpp:176-205) */
int orc_verify(const float* kp_xy, uint32_t nq, const float* cloud, uint32_t H, uint32_t W, const uint32_t* row_ptr,
               const orc_dmatch* matches, const float* mxyz, const float* spans, uint32_t n_obj,
               const orc_verify_params* prm, orc_rng* rng, orc_pose* poses, uint32_t* n_poses, uint32_t* inlier_kp,
               uint32_t* n_inlier_kp, orc_round_trace* rounds, uint32_t* n_rounds) {
  const uint32_t pose_cap = *n_poses, kp_cap = *n_inlier_kp, round_cap = n_rounds ? *n_rounds : 0;
  *n_poses = 0; *n_inlier_kp = 0;
  if (n_rounds) *n_rounds = 0;
  if (!cloud || H == 0 || W == 0) return 0;                 /* 2D-only branch is an empty TODO, :147-152 */
  std::map<size_t, MatchCluster> objects;
  for (uint32_t qi = 0; qi < nq; ++qi) {
    int row = (int)kp_xy[2 * qi + 1], col = (int)kp_xy[2 * qi];   /* float -> int truncation, :185 */
    if (row < 0 || col < 0 || (uint32_t)row >= H || (uint32_t)col >= W) return -1;
    const float* qp = cloud + 3 * ((size_t)row * W + col);
    if (std::isnan(qp[0])) continue;                              /* only .x is tested, :189 */
    P3 q = {{qp[0], qp[1], qp[2]}};
    for (uint32_t m = row_ptr[qi]; m < row_ptr[qi + 1]; ++m) {
      if (matches[m].imgIdx < 0 || (uint32_t)matches[m].imgIdx >= n_obj) return -2;
      P3 t = {{mxyz[3 * m], mxyz[3 * m + 1], mxyz[3 * m + 2]}};
      objects[(size_t)matches[m].imgIdx].add(t, q, qi);
    }
  }
  while (!objects.empty()) {
    size_t obj = objects.begin()->first;
    MatchCluster& cl = objects.begin()->second;
    cl.fill(kp_xy, spans[obj], prm->sensor_error);
    while (true) {
      IndexVector inl;
      M33 R; P3 T;
      std::memset(&R, 0, sizeof(R)); std::memset(&T, 0, sizeof(T));
      orc_round_trace tr;
      cl.ransac(prm->sensor_error, prm->n_ransac_iterations, rng, inl, R, T, &tr, 0, 0, 0);
      if (n_rounds && *n_rounds < round_cap) rounds[(*n_rounds)++] = tr;
      if (inl.size() < prm->min_inliers) break;                   /* :205-206 */
      cl.invalidate_keypoints(inl);
      if (*n_poses >= pose_cap || *n_inlier_kp + inl.size() > kp_cap) return -3;
      orc_pose& p = poses[(*n_poses)++];
      p.object = (uint32_t)obj;
      for (int r = 0; r < 3; ++r) { for (int c = 0; c < 3; ++c) p.R[3 * r + c] = R.m[r][c]; p.t[r] = T.v[r]; }
      p.inlier_begin = *n_inlier_kp;
      for (size_t i = 0; i < inl.size(); ++i) inlier_kp[(*n_inlier_kp)++] = inl[i];
      p.inlier_end = *n_inlier_kp;
    }
    objects.erase(obj);
  }
  return 0;
}

}  // extern "C"
