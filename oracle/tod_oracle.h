/*
 * oracle/tod_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * C interface of the CPU restatement ("oracle") of the textured-object-detection
 * hot path of wg-perception/tod (stages B = matching, C = geometric verification).
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library; nothing under tod_amd/ links, imports or executes it.
 *
 * Parity pin status (see oracle/README.md and DESIGN.md):
 *   - clique search: pinned by the reference's own gtests (test/test_maximum_clique.cpp:7-53)
 *   - rand(): pinned against this container's libc
 *   - everything else: PARITY UNPINNED (the reference holds no fixture for it and its
 *     sources cannot be built here without stand-in Boost/OpenCV headers, which the
 *     rules forbid) -- it is a line-by-line restatement, cited per function.
 */
#ifndef TOD_ORACLE_H_
#define TOD_ORACLE_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* same layout as cv::DMatch (queryIdx, trainIdx, imgIdx, distance) */
typedef struct { int32_t queryIdx, trainIdx, imgIdx; float distance; } orc_dmatch;

/* glibc random_r TYPE_3 state (31-word additive feedback ring, front/back cursors) */
typedef struct { uint32_t s[31]; uint32_t f, b; uint64_t draws; } orc_rng;

typedef struct { uint32_t min_inliers, n_ransac_iterations; float sensor_error; } orc_verify_params;

typedef struct { uint32_t object; float R[9]; float t[3]; uint32_t inlier_begin, inlier_end; } orc_pose;

/* per-RANSAC-round trace (optional, for fine-grained parity tests) */
typedef struct {
  uint32_t iterations;     /* iterations_ at loop exit (ransac.h:95-135)                    */
  uint32_t best_iteration; /* index of the iteration whose consensus set was kept           */
  int32_t  best_count;     /* n_best_inliers_count                                          */
  uint64_t draws_before, draws_after; /* rand() calls consumed before/after the round       */
  uint32_t n_model_inliers;/* size of the consensus set handed to the growth loop           */
  uint32_t n_final_inliers;/* match indices after growth (adjacency_ransac.cpp:270-303)     */
  uint32_t growth_passes;
} orc_round_trace;

/* ---- stage B ------------------------------------------------------------------------- */
void orc_spans(const float* pts_xyz, const uint32_t* obj_off, uint32_t n_obj, float* spans);
int orc_match(const uint8_t* db_desc, const uint32_t* obj_off, uint32_t n_obj, const float* db_pts_xyz,
              uint32_t desc_bytes, const uint8_t* q_desc, uint32_t nq, uint32_t k, uint32_t radius,
              uint32_t* row_ptr, orc_dmatch* matches, float* matches_xyz);
int orc_match_ratio(const uint8_t* db_desc, const uint32_t* obj_off, uint32_t n_obj, const float* db_pts_xyz,
              uint32_t desc_bytes, const uint8_t* q_desc, uint32_t nq, uint32_t k, uint32_t radius, float ratio,
              uint32_t* row_ptr, orc_dmatch* matches, float* matches_xyz);
/* exact k-NN keys only: key = (distance << 32) | global_row, ascending, k per query (UINT64_MAX padded) */
void orc_knn_keys(const uint8_t* db_desc, uint64_t n_db, uint32_t desc_bytes, const uint8_t* q_desc,
                  uint32_t nq, uint32_t k, uint64_t* keys);

/* ---- rand() -------------------------------------------------------------------------- */
void orc_rng_seed(orc_rng* r, uint32_t seed);
int32_t orc_rng_next(orc_rng* r);

/* ---- clique -------------------------------------------------------------------------- */
/* Builds a graph with AddEdge(added) then DeleteEdge(deleted) and runs FindClique(minimal_size).
 * minimal_size = 0xFFFFFFFF is FindMaximumClique. Returns clique size; vertices in out (cap n).
 * *underruns = number of reads/pops of the colour stack at top==0 (decision D3). */
uint32_t orc_clique(uint32_t n, const uint32_t* added, uint32_t n_added, const uint32_t* deleted,
                    uint32_t n_deleted, uint32_t minimal_size, uint32_t* out, uint32_t* underruns,
                    uint32_t* steps);

/* ---- stage C, stepwise (one object = one tod::AdjacencyRansac) ------------------------- */
typedef struct orc_cluster orc_cluster;
orc_cluster* orc_cluster_new(const float* train_xyz, const float* query_xyz, const uint32_t* query_idx, uint32_t n);
void orc_cluster_free(orc_cluster*);
void orc_cluster_fill(orc_cluster*, const float* kp_xy, uint32_t n_kp, float span, float sensor_error);
uint32_t orc_cluster_size(const orc_cluster*);
/* which: 0 = physical, 1 = sample. bits: n rows of words_per_row u64, bit j of row i set iff (i,j) adjacent */
void orc_cluster_bits(const orc_cluster*, int which, uint64_t* bits, uint32_t words_per_row);
uint32_t orc_cluster_valid(const orc_cluster*, uint32_t* out);
/* one getSamples() call: returns number of samples (0 or 3), samples in samples_ order */
uint32_t orc_cluster_draw(orc_cluster*, orc_rng*, uint32_t* samples3);
/* selectWithinDistance() for a given sample triple (samples_ order); returns consensus size, list in out (cap n) */
uint32_t orc_cluster_consensus(orc_cluster*, const uint32_t* samples3, uint32_t* out, uint32_t* gate_called,
                               uint32_t* gate_clique_size);
/* AdjacencyRansac::Ransac */
uint32_t orc_cluster_ransac(orc_cluster*, float sensor_error, uint32_t n_iterations, orc_rng*, uint32_t* inlier_kp,
                            float* R9, float* T3, orc_round_trace* trace, uint32_t* model_inliers /* cap n or NULL */,
                            int32_t* iter_counts /* cap n_iterations+2 or NULL */,
                            uint32_t* iter_samples /* cap 3*(n_iterations+2) or NULL */);
void orc_cluster_invalidate_kp(orc_cluster*, const uint32_t* kp, uint32_t n_kp);
/* Kabsch on a list of match indices (estimateRigidTransformationSVD); query->training */
int orc_cluster_kabsch(const orc_cluster*, const uint32_t* idx, uint32_t n, float* R9, float* T3);

/* ---- stage C, whole frame (GuessGenerator::process) ------------------------------------ */
int orc_verify(const float* kp_xy, uint32_t nq, const float* cloud_xyz, uint32_t H, uint32_t W,
               const uint32_t* row_ptr, const orc_dmatch* matches, const float* matches_xyz,
               const float* spans, uint32_t n_obj, const orc_verify_params* prm, orc_rng* rng,
               orc_pose* poses, uint32_t* n_poses, uint32_t* inlier_kp, uint32_t* n_inlier_kp,
               orc_round_trace* rounds, uint32_t* n_rounds);

/* ---- the optional LSH-approximate matcher mode (oracle/lsh_oracle.c): FLANN's published scheme with this repo's own choice of
 * key bits; parity unpinned (OpenCV's FLANN is not in the reference tree) */
void orc_lsh_key_bits(uint32_t table, uint32_t key_size, uint8_t* pos);
void orc_lsh_knn_keys(const uint8_t* db_desc, uint64_t n_db, const uint8_t* q_desc, uint32_t nq, uint32_t k, uint32_t n_tables,
                      uint32_t key_size, uint32_t multi_probe_level, uint64_t* keys, uint32_t* n_candidates);

/* ---- stage C without depth (oracle/pnp_oracle.c): the reference's TODO branch, DEFINED there; parity unpinned by construction.
 * prm->sensor_error is the reprojection threshold in pixels; best_hyp / best_count: optional, n_obj entries each. */
int orc_verify_2d(const float* kp_xy, uint32_t nq, const float* K9, const uint32_t* row_ptr, const orc_dmatch* matches,
                  const float* matches_xyz, const float* spans, uint32_t n_obj, const orc_verify_params* prm, orc_rng* rng,
                  orc_pose* poses, uint32_t* n_poses, uint32_t* inlier_kp, uint32_t* n_inlier_kp, uint32_t* best_hyp,
                  uint32_t* best_count);

#ifdef __cplusplus
}
#endif
#endif
