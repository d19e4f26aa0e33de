/* TEST INFRASTRUCTURE ONLY (see oracle/README.md): CPU definition of the float-descriptor matcher of BASELINE.json
 * configs[3] ("SIFT-128 float descriptors, L2 brute force"). This is NOT a restatement of reference code: the
 * reference's DescriptorMatcher serves binary descriptors through FLANN-LSH only and throws for any other index
 * type (src/detection/DescriptorMatcher.cpp:154-188), so there is nothing in the reference to be equal to -- PARITY
 * UNPINNED. What is kept from the reference is the shape of the result (DescriptorMatcher.cpp:195-252): per query the
 * k nearest rows of the concatenated DB, truncated at the first distance > radius (strict, :215), (imgIdx, trainIdx)
 * by object prefix sums, and the 3D point of every kept match (:231-244).
 *
 * Definition of the distance (what the GPU path has to reproduce bit for bit): d2 = sum over i = 0..dim-1, in index
 * order, of (q[i] - r[i]) * (q[i] - r[i]) in IEEE binary32 without fused multiply-add; distance = sqrtf(d2).
 * Order: (d2 ascending, global row ascending).  Build with -ffp-contract=off and without -ffast-math. */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct { int queryIdx, trainIdx, imgIdx; float distance; } l2_dmatch;

static float l2_d2(const float* q, const float* r, uint32_t dim) {
  float acc = 0.f;                          /* float addition is not associative: without -ffast-math the order stays */
  for (uint32_t i = 0; i < dim; ++i) {
    const float t = q[i] - r[i];
    const float p = t * t;
    acc = acc + p;
  }
  return acc;
}

/* keys[nq*k]: (float bits of d2) << 32 | global row, ascending; UINT64_MAX padding */
void l2_knn_keys(const float* db, uint64_t n_db, uint32_t dim, const float* q, uint32_t nq, uint32_t k, uint64_t* keys) {
  for (uint32_t qi = 0; qi < nq; ++qi) {
    uint64_t* best = keys + (size_t)qi * k;
    for (uint32_t j = 0; j < k; ++j) best[j] = UINT64_MAX;
    for (uint64_t r = 0; r < n_db; ++r) {
      const float d2 = l2_d2(q + (size_t)qi * dim, db + (size_t)r * dim, dim);
      uint32_t bits;
      memcpy(&bits, &d2, 4);
      uint64_t key = ((uint64_t)bits << 32) | r;
      if (key >= best[k - 1]) continue;
      uint32_t j = k - 1;
      while (j > 0 && best[j - 1] > key) { best[j] = best[j - 1]; --j; }
      best[j] = key;
    }
  }
}

int l2_match(const float* db, const uint32_t* obj_off, uint32_t n_obj, const float* db_pts_xyz, uint32_t dim, const float* q,
             uint32_t nq, uint32_t k, float radius, uint32_t* row_ptr, l2_dmatch* matches, float* xyz) {
  if (k == 0 || !(radius > 0.f)) return -1;
  const uint64_t n_db = obj_off[n_obj];
  uint64_t* keys = (uint64_t*)malloc((size_t)nq * k * sizeof(uint64_t));
  l2_knn_keys(db, n_db, dim, q, nq, k, keys);
  uint32_t out = 0;
  for (uint32_t qi = 0; qi < nq; ++qi) {
    row_ptr[qi] = out;
    for (uint32_t j = 0; j < k; ++j) {
      const uint64_t key = keys[(size_t)qi * k + j];
      if (key == UINT64_MAX) break;
      const uint32_t bits = (uint32_t)(key >> 32), row = (uint32_t)key;
      float d2;
      memcpy(&d2, &bits, 4);
      const float dist = sqrtf(d2);
      if (dist > radius) break;
      uint32_t o = 0;
      while (o + 1 < n_obj && obj_off[o + 1] <= row) ++o;
      while (obj_off[o + 1] <= row) ++o;                 /* skips empty objects */
      matches[out].queryIdx = (int)qi; matches[out].trainIdx = (int)(row - obj_off[o]); matches[out].imgIdx = (int)o;
      matches[out].distance = dist;
      memcpy(xyz + 3 * (size_t)out, db_pts_xyz + 3 * (size_t)row, 12);
      ++out;
    }
  }
  row_ptr[nq] = out;
  free(keys);
  return 0;
}
