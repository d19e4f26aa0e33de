/*
 * todhip.h -- C ABI of the MI355X-native textured-object-detection hot path.
 *
 * This is the drop-in boundary for wg-perception/tod's detection cells: a thin C++ adapter
 * (adapter/ecto_cells.hpp) keeps the ecto cell, parameter and tendril names of
 * src/detection/DescriptorMatcher.cpp and src/detection/GuessGenerator.cpp and converts the
 * tendril types to the flat buffers below. Citations are file:line in the reference tree.
 *
 * Conventions: every function returns a todhip_status (0 = ok, negative = error); nothing throws,
 * nothing prints; all buffers are caller-owned with explicit sizes; one call in flight per
 * context; a context is bound to one HIP device and one HIP stream. Pointers named d_* are device
 * pointers on the context's device, everything else is host memory.
 */
#ifndef TODHIP_H_
#define TODHIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TODHIP_VERSION 1

typedef enum {
  TODHIP_OK = 0,
  TODHIP_EINVAL = -1,      /* bad argument (null pointer, k == 0, radius == 0, unsupported descriptor size ...) */
  TODHIP_ENODB = -2,       /* no descriptors loaded: DescriptorMatcher.cpp:204-208 logs and returns            */
  TODHIP_EHIP = -3,        /* a HIP runtime call failed; todhip_last_hip_error() has the code                   */
  TODHIP_ECAPACITY = -4,   /* caller-provided output capacity too small                                         */
  TODHIP_ERANGE = -5,      /* keypoint outside the cloud / imgIdx outside the object table                      */
  TODHIP_ENOMEM = -6,
  TODHIP_ESCRATCH = -7     /* verifier scratch budget exceeded (never silently wrong)                           */
} todhip_status;

typedef struct todhip_ctx todhip_ctx;

/* One trained object as the DB hands it over (DescriptorMatcher.cpp:72-86):
 * `descriptors` attachment = n x desc_bytes CV_8U, `points` attachment = n x 3 f32 (object frame). */
typedef struct { const uint8_t* desc; const float* pts_xyz; uint32_t n; } todhip_object;

/* Same layout as cv::DMatch: the adapter reinterprets, no conversion. */
typedef struct { int32_t queryIdx, trainIdx, imgIdx; float distance; } todhip_dmatch;

/* glibc rand() state (random_r TYPE_3). The reference calls the process-global, never-seeded
 * rand() (sac_model_registration_graph.h:111, sac.h:71); here the stream is explicit (decision D4). */
typedef struct { uint32_t s[31]; uint32_t f, b; uint64_t draws; } todhip_rng;

/* GuessGenerator parameters, GuessGenerator.cpp:71-81 (defaults 15 / 1000 / 0.01). */
typedef struct { uint32_t min_inliers, n_ransac_iterations; float sensor_error; } todhip_verify_params;

/* One PoseResult (GuessGenerator.cpp:223-230): R row-major, pose maps object -> camera frame. */
typedef struct { uint32_t object; float R[9]; float t[3]; uint32_t inlier_begin, inlier_end; } todhip_pose;

/* Counters a cell would have printed to stdout (DescriptorMatcher.cpp:68-122, GuessGenerator.cpp:177-243). */
typedef struct {
  uint64_t db_rows, db_objects;
  uint32_t last_nq, last_k, last_matches;
  uint32_t last_objects_verified, last_rounds, last_hypotheses, last_gate_calls, last_poses;
  double   last_match_kernel_ms;      /* HIP-event time of the dominant matcher kernel (if timing enabled) */
  double   sum_match_kernel_ms;
  uint64_t n_match_kernel_launches;
  uint32_t last_sprint_launches;      /* verifier: launches of the single-wave small-object kernel in the last call ... */
  uint32_t last_sprint_rounds;        /* ... and the Ransac rounds (adjacency_ransac.cpp:234-309) it ran without the host */
  uint32_t last_verify_ticks;         /* host round trips (launch + synchronize) of the last verify call */
  uint32_t last_block_split;          /* matrix-core matcher: the block form of the last launch (4 = whole, 2 / 3: todhip_set_matcher_block_split) */
  /* matrix-core matcher, launches with split blocks (partial-distance elimination): accumulator blocks started as
   * parts, and those that went on to their second part -- cumulative over the context's life, as of the last report read */
  uint64_t k4x_half_blocks, k4x_half_blocks_completed;
} todhip_counters;

/* ---- lifetime ---------------------------------------------------------------------------------- */
int  todhip_version(void);
/* stream == NULL: the context creates and owns a non-blocking stream. */
int  todhip_create(int device, void* hip_stream, todhip_ctx** out);
void todhip_destroy(todhip_ctx*);
void* todhip_stream(todhip_ctx*);
int  todhip_last_hip_error(const todhip_ctx*);
int  todhip_synchronize(todhip_ctx*);
int  todhip_get_counters(todhip_ctx*, todhip_counters* out);
/* Streams for contexts that run side by side (a pipeline: ORB, matcher and verifier contexts, each on a stream of its own, as
 * bench.py and tod_amd/pipeline.py drive them). todhip_set_cu_partition(n) reserves the device's last n compute units for
 * TODHIP_STREAM_LATENCY streams (the short, latency-bound kernels of ORB and of the verifier -- a single wave for hundreds of
 * microseconds -- which otherwise share every SIMD with the matcher's chip-filling DB pass); TODHIP_STREAM_THROUGHPUT streams get
 * the other compute units. n = 0 (the default, or the environment's TODHIP_LATENCY_CUS): no partition, latency streams are plain
 * high-priority streams. Process-wide; applies to streams created afterwards (the verifier's own side streams -- the lanes its
 * single-wave launch groups run on: sprints, clique gates, growth -- are latency streams and are re-created on their next use).
 * A multiple of 8 takes the same share of every XCD. Measured: a pipeline whose matcher is the bound loses with any
 * partition; a verifier-bound one (frames of ~190 objects) gains 15 % with the matcher alone confined to 160 of 256 CUs. */
enum { TODHIP_STREAM_THROUGHPUT = 0, TODHIP_STREAM_LATENCY = 1 };
int  todhip_set_cu_partition(uint32_t latency_cus);
int  todhip_stream_create(int device, int kind, void** stream_out);
int  todhip_stream_destroy(void* stream);
int  todhip_set_kernel_timing(todhip_ctx*, int enable);   /* bracket the matcher kernel with HIP events */
/* The exact Hamming search of todhip_match* exists twice, with identical results: on the vector ALU (xor + popcount,
 * partial-distance elimination: data dependent) and on the matrix cores (bits as +-1 MX-fp4 values, dot = 256 - 2 d:
 * data independent). AUTO picks by launch shape. */
enum { TODHIP_ENGINE_AUTO = 0, TODHIP_ENGINE_VALU = 1, TODHIP_ENGINE_MFMA = 2 };
int  todhip_set_matcher_engine(todhip_ctx*, int engine);
/* The matrix-core engine's partial-distance elimination: a 32 x 32 block's 256 bit positions are 4 matrix instructions, and the
 * block may stop after the first `split` of them when no partial sum can still reach a threshold (exact for any data). Which split
 * pays depends on the data, so the default (-1) adapts per context from the launches' own statistics; 0 = whole blocks, 2 / 3 = always
 * that split where the radius allows it (2: radius < 64, 3: radius < 96). Identical results in every setting. */
int  todhip_set_matcher_block_split(todhip_ctx*, int split);

/* ---- stage B: DescriptorMatcher ---------------------------------------------------------------- */
/* Replaces DescriptorMatcher::parameter_callback (DescriptorMatcher.cpp:60-129): ingest every object,
 * compute spans (:104-121), build the device-resident DB. With shard_count > 1 the descriptor rows are
 * split into object-aligned contiguous shards and only shard `shard_rank` is kept on this device; model
 * points and the object table are always complete. spans_out (n_objs floats) may be NULL. */
int todhip_db_load(todhip_ctx*, const todhip_object* objs, uint32_t n_objs, uint32_t desc_bytes,
                   uint32_t shard_rank, uint32_t shard_count, float* spans_out);
/* The same ingest from attachments that are already in this device's memory (objs[i].desc / .pts_xyz are device pointers, e.g.
 * todhip_model_device's): device-to-device copies, the spans computed there (identical values), n_objs floats come back. */
int todhip_db_load_device(todhip_ctx*, const todhip_object* objs, uint32_t n_objs, uint32_t desc_bytes,
                          uint32_t shard_rank, uint32_t shard_count, float* spans_out);
int todhip_db_info(const todhip_ctx*, uint64_t* total_rows, uint64_t* shard_first_row, uint64_t* shard_rows,
                   uint32_t* n_objs);

/* Replaces DescriptorMatcher::process (DescriptorMatcher.cpp:195-252): exact Hamming k-NN (decision D1,
 * instead of FLANN-LSH knnMatch(k=5) at :211), radius truncation (:212-220; radius is the reference's
 * `unsigned int radius_`, 0 is rejected because the reference then indexes an empty vector at :237) and
 * the match -> 3D gather (:231-244). Outputs in CSR form: row_ptr[nq+1], matches/matches_xyz capacity nq*k. */
int todhip_match(todhip_ctx*, const uint8_t* q_desc, uint32_t nq, uint32_t k, uint32_t radius,
                 uint32_t* row_ptr, todhip_dmatch* matches, float* matches_xyz);

/* Lowe's ratio test, the step the reference announces and leaves empty (DescriptorMatcher.cpp:223-227; its shipped configs
 * ask for ratio 0.8, conf/detection.ork:39, which its `unsigned int ratio_` turns into 0 = off). Off by default; when set
 * (0 < ratio <= 1), every todhip_match* form drops all matches of a query whose two nearest DB rows (d1 <= d2, exact, over
 * the whole DB whatever the radius) do not satisfy (float)d1 < ratio * (float)d2; a one-row DB passes. The radius cut then
 * applies to the survivors. The sharded forms need k >= 2 while it is on. Definition = oracle/tod_oracle.cpp
 * orc_match_ratio (there is no reference behaviour to be identical to). */
int todhip_set_ratio_test(todhip_ctx*, float ratio);

/* Optional LSH-approximate mode (SURVEY 8(f) N4 c). The reference's matcher is cv::FlannBasedMatcher over
 * cv::flann::LshIndexParams(n_tables, key_size, multi_probe_level) (DescriptorMatcher.cpp:175-180; conf/detection.ork:32-38:
 * 10 tables, 16-bit keys, level 1); this library answers that configuration with the EXACT search by default (a superset of
 * what any LSH index returns). With n_tables > 0 every todhip_match* form on 32-byte descriptors ranks, instead of the whole
 * shard, only the rows an index of FLANN's published scheme turns up: per table a key of key_size descriptor bits (table t uses
 * the first key_size entries of a Fisher-Yates shuffle of 0..255 driven by a 32-bit mix of (t, step): FLANN draws them from
 * rand(), which cannot be reproduced -- parity unpinned, the CPU checker is oracle/lsh_oracle.c); a query's candidates are the
 * rows whose key differs from its own in at most multi_probe_level bits in at least one table; the k nearest of those by exact
 * distance, ties by row. Radius cut, ratio test (on the candidates' two nearest) and the sharded forms work as before.
 * n_tables = 0 switches back to the exact search. Limits: n_tables <= 32, key_size <= 24, multi_probe_level <= 3.
 * May be called before or after todhip_db_load (the index is built for the resident shard either way). */
int todhip_set_lsh(todhip_ctx*, uint32_t n_tables, uint32_t key_size, uint32_t multi_probe_level);

/* Device-resident form of the same call (inputs already in HBM, outputs stay in HBM):
 * d_counts[nq] (matches kept per query), d_matches[nq*k], d_matches_xyz[nq*k*3], fixed stride k. */
int todhip_match_device(todhip_ctx*, const void* d_q_desc, uint32_t nq, uint32_t k, uint32_t radius,
                        void* d_counts, void* d_matches, void* d_matches_xyz);

/* Float descriptors (BASELINE.json configs[3]; not a reference feature -- the reference's matcher throws for anything
 * but FLANN-LSH on binary descriptors, DescriptorMatcher.cpp:154-188): a DB loaded with desc_bytes == 512 (128 x f32 per
 * row, one device) is searched by exact L2 k-NN, k <= 8. Result shape as todhip_match[_device] (DescriptorMatcher.cpp:
 * 195-252): order (distance asc, global row asc), truncated at the first distance > radius; DMatch.distance =
 * sqrtf(d2), d2 = sum over i = 0..127 in index order of (q[i] - r[i])^2 in IEEE binary32 without fused multiply-add
 * (the CPU checker of this definition is oracle/l2_oracle.c). The candidates come from a
 * bf16 MFMA GEMM with a proven error bound, the final order from the exact distances.
 * Batch form: the nq queries may be those of F frames (F x Q rows, frame f's query q at row f Q + q, which is also its
 * queryIdx): they share ONE pass over the DB, and every query's matches are those of a call that carried its frame alone
 * (16 frames of 1000 queries against 500k rows: 0.12 ms per frame instead of 0.19). */
int todhip_match_l2(todhip_ctx*, const float* q_desc, uint32_t nq, uint32_t k, float radius, uint32_t* row_ptr,
                    todhip_dmatch* matches, float* matches_xyz);
int todhip_match_l2_device(todhip_ctx*, const void* d_q_desc, uint32_t nq, uint32_t k, float radius, void* d_counts,
                           void* d_matches, void* d_matches_xyz);

/* Sharded form, step 1: this shard's top-k per query as keys (distance << 32 | global_row), ascending,
 * UINT64_MAX padded; d_keys[nq*k]. The ranks exchange these with one RCCL all-gather. `radius` is the radius of
 * step 2: rows farther away never survive the truncation of DescriptorMatcher.cpp:212-220, so the search is
 * allowed to leave them out of the keys (it is not obliged to); pass >= 256 to get the plain top-k. */
int todhip_match_shard_device(todhip_ctx*, const void* d_q_desc, uint32_t nq, uint32_t k, uint32_t radius, void* d_keys);
/* Sharded form, step 2: merge n_shards key sets (layout [shard][nq][k]) with the total order
 * (distance asc, global row asc), apply the radius cut, resolve (imgIdx, trainIdx), gather 3D. */
int todhip_merge_shards_device(todhip_ctx*, const void* d_keys_all, uint32_t n_shards, uint32_t nq, uint32_t k,
                               uint32_t radius, void* d_counts, void* d_matches, void* d_matches_xyz);

/* The same merge launched on a stream of the caller's instead of the context's: it only READS the context's immutable
 * object table and model points, so it may run beside a todhip_match_shard_device call that is in flight on the context's
 * own stream (the multi-GPU step puts collectives + merge on a second stream, tod_amd/sharded.py). The caller orders the
 * stream against the producer of d_keys_all and the consumers of the outputs. */
int todhip_merge_shards_device_on(todhip_ctx*, void* hip_stream, const void* d_keys_all, uint32_t n_shards, uint32_t nq,
                                  uint32_t k, uint32_t radius, void* d_counts, void* d_matches, void* d_matches_xyz);

/* ---- stage C: GuessGenerator ------------------------------------------------------------------- */
void todhip_rng_seed(todhip_rng*, uint32_t seed);   /* srand(seed); the reference never seeds => seed 1 */
/* Replaces GuessGenerator::process (GuessGenerator.cpp:127-250) and everything under src/common.
 * kp_xy: nq x 2 keypoint pixels (cv::KeyPoint::pt); cloud_xyz: H x W x 3 f32 organised cloud (NaN = no depth);
 * matches in CSR form as produced by todhip_match; spans: per object index (imgIdx).
 * poses: capacity *n_poses in, count out. inlier_kp: capacity *n_inlier_kp in, count out. */
int todhip_verify(todhip_ctx*, const float* kp_xy, uint32_t nq, const float* cloud_xyz, uint32_t H, uint32_t W,
                  const uint32_t* row_ptr, const todhip_dmatch* matches, const float* matches_xyz,
                  const float* spans, uint32_t n_objs, const todhip_verify_params*, todhip_rng* rng,
                  todhip_pose* poses, uint32_t* n_poses, uint32_t* inlier_kp, uint32_t* n_inlier_kp);

/* Device-resident form: d_kp_xy[nq*2] f32, d_cloud_xyz[H*W*3] f32 and the matcher's fixed-stride outputs
 * (d_counts[nq] u32, d_matches[nq*k], d_matches_xyz[nq*k*3], exactly what todhip_match_device produced) are in
 * HBM already; ClusterPerObject (adjacency_ransac.cpp:176-205) runs as kernels. Poses come back to the host. */
int todhip_verify_device(todhip_ctx*, const void* d_kp_xy, uint32_t nq, const void* d_cloud_xyz, uint32_t H, uint32_t W,
                         const void* d_counts, const void* d_matches, const void* d_matches_xyz, uint32_t k,
                         const float* spans, uint32_t n_objs, const todhip_verify_params*, todhip_rng* rng,
                         todhip_pose* poses, uint32_t* n_poses, uint32_t* inlier_kp, uint32_t* n_inlier_kp);

/* Same, but the query points are back-projected from the registered depth image instead of being read from a
 * materialised H x W x 3 cloud (replaces RescaledRegisteredDepth -> DepthTo3d ->
 * GuessGenerator's cloud lookup, detector.py:26,62,66-69 + adjacency_ransac.cpp:184-185). d_depth: H x W, float
 * metres (NaN = no depth) or, with depth_is_u16, uint16 millimetres (0 = no depth). K9: row-major 3x3 intrinsics. */
int todhip_verify_device_depth(todhip_ctx*, const void* d_kp_xy, uint32_t nq, const void* d_depth, int depth_is_u16,
                               uint32_t H, uint32_t W, const float* K9, const void* d_counts, const void* d_matches,
                               const void* d_matches_xyz, uint32_t k, const float* spans, uint32_t n_objs,
                               const todhip_verify_params*, todhip_rng* rng, todhip_pose* poses, uint32_t* n_poses,
                               uint32_t* inlier_kp, uint32_t* n_inlier_kp);

/* A batch of n_frames frames through GuessGenerator::process in the launches and host round trips of one frame.
 * Inputs are the per-frame arrays of todhip_verify_device[_depth] back to back (frame f at f times the per-frame size:
 * d_kp_xy[n_frames*nq*2], cloud [n_frames*H*W*3] or depth [n_frames*H*W], d_counts[n_frames*nq], d_matches[n_frames*nq*k],
 * d_matches_xyz[n_frames*nq*k*3] -- what todhip_match_device yields for n_frames*nq queries); a frame with fewer
 * than nq keypoints pads with counts 0. rng[n_frames]: one generator per frame (decision D4), each advanced as
 * the single-frame call would. Poses of frame f: poses[pose_ptr[f] .. pose_ptr[f+1]) (pose_ptr[n_frames+1]);
 * inlier_begin/end index the shared inlier_kp array. Each frame's result equals the single-frame call's.
 * Frames of a batch that are out of step have their heavy phases (clique gate, growth of objects with >= 96 matches) launched on
 * up to two process-wide side streams (TODHIP_VERIFY_FLIGHTS, 0 = none); the call returns when everything has finished. */
int todhip_verify_batch_device(todhip_ctx*, uint32_t n_frames, const void* d_kp_xy, uint32_t nq, const void* d_cloud_xyz,
                               uint32_t H, uint32_t W, const void* d_counts, const void* d_matches, const void* d_matches_xyz,
                               uint32_t k, const float* spans, uint32_t n_objs, const todhip_verify_params*, todhip_rng* rng,
                               todhip_pose* poses, uint32_t* n_poses, uint32_t* pose_ptr, uint32_t* inlier_kp,
                               uint32_t* n_inlier_kp);
int todhip_verify_batch_device_depth(todhip_ctx*, uint32_t n_frames, const void* d_kp_xy, uint32_t nq, const void* d_depth,
                                     int depth_is_u16, uint32_t H, uint32_t W, const float* K9, const void* d_counts,
                                     const void* d_matches, const void* d_matches_xyz, uint32_t k, const float* spans,
                                     uint32_t n_objs, const todhip_verify_params*, todhip_rng* rng, todhip_pose* poses,
                                     uint32_t* n_poses, uint32_t* pose_ptr, uint32_t* inlier_kp, uint32_t* n_inlier_kp);

/* ---- stage A: ORB features (ecto_opencv FeatureDescriptor -> cv::ORB; detector.py:10,27) --------- */
/* gray: H x W u8, row stride `stride`. Outputs up to n_features keypoints: kp_xy (x,y level-0 pixels),
 * kp_aux (size, angle_deg, response, octave) and 32-byte rBRIEF descriptors. *n_out: capacity in, count out.
 * pattern: 256 x 4 int8 (x0,y0,x1,y1) test pairs, or NULL for the built-in seeded pattern. */
int todhip_orb(todhip_ctx*, const uint8_t* gray, uint32_t H, uint32_t W, uint32_t stride, uint32_t n_features,
               uint32_t n_levels, float scale_factor, const int8_t* pattern, float* kp_xy, float* kp_aux,
               uint8_t* desc, uint32_t* n_out);

/* The same with the cell's `mask` input (detector.py:41 forwards it to FeatureDescriptor; cv::ORB's second argument): only
 * pixels with mask != 0 (H x W u8, row stride `stride`, level 0) can become keypoints. mask == NULL: todhip_orb. */
int todhip_orb_masked(todhip_ctx*, const uint8_t* gray, const uint8_t* mask, uint32_t H, uint32_t W, uint32_t stride,
                      uint32_t n_features, uint32_t n_levels, float scale_factor, const int8_t* pattern, float* kp_xy,
                      float* kp_aux, uint8_t* desc, uint32_t* n_out);

/* Device-resident form: d_gray (H x W u8, row stride `stride`) is in HBM, keypoints and descriptors stay in HBM
 * (d_kp_xy[cap*2] f32, d_kp_aux[cap*4] f32, d_desc[cap*32] u8); only the count comes back. *n_out: capacity in. */
int todhip_orb_device(todhip_ctx*, const void* d_gray, uint32_t H, uint32_t W, uint32_t stride, uint32_t n_features,
                      uint32_t n_levels, float scale_factor, const int8_t* pattern, void* d_kp_xy, void* d_kp_aux,
                      void* d_desc, uint32_t* n_out);

/* A batch of n_frames device-resident frames (frame f at d_gray + f * frame_stride bytes) in the launches of one:
 * frame f's keypoints land at row f * cap of d_kp_xy[n_frames*cap*2], d_kp_aux[n_frames*cap*4], d_desc[n_frames*cap*32];
 * n_out[n_frames] comes back. Each frame's result equals todhip_orb_device's on that frame. */
int todhip_orb_batch_device(todhip_ctx*, const void* d_gray, uint32_t n_frames, uint64_t frame_stride, uint32_t H, uint32_t W,
                            uint32_t stride, uint32_t n_features, uint32_t n_levels, float scale_factor, const int8_t* pattern,
                            void* d_kp_xy, void* d_kp_aux, void* d_desc, uint32_t cap, uint32_t* n_out);

/* ---- stage C without depth (SURVEY 8(f) row N4 b) -------------------------------------------------------- */
/* The branch GuessGenerator::process leaves empty when `points3d` is empty (GuessGenerator.cpp:147-152 "Only use 2d to 3d
 * matching // TODO"; doc/source/index.rst:36-46: a PnP problem the reference never plugged in). NOT a reference result:
 * defined here, checked bit for bit by the tests' CPU definition (oracle/pnp_oracle.c), parity unpinned by construction.
 *   inputs   kp_xy nq x 2 keypoint pixels; K9 the camera matrix, row-major; matches in CSR form as produced by todhip_match
 *            (matches_xyz = the model point of each match); spans per object index. prm->sensor_error is the reprojection
 *            threshold in PIXELS here, prm->n_ransac_iterations the (fixed) number of hypotheses per object.
 *   per object (ascending index, matches clustered in CSR order, at least max(3, min_inliers) of them):
 *     hypothesis h = 0 .. n - 1: a sample of three matches, pairwise with keypoints > 20 px apart and distinct model points
 *       within the object's span, picked by a counter-based hash of (seed, object, h, attempt <= 16); Grunert's P3P in f64 on
 *       it (up to 4 poses); consensus set of a pose = matches whose model point reprojects within the threshold, in front of
 *       the camera. The largest consensus set wins, the first (h, root) on ties; it must reach min_inliers.
 *     the winner is refined by 5 Gauss-Newton steps on its consensus set and the consensus set recomputed; the pose is
 *       reported if that still reaches min_inliers. ONE pose per object, by design: the 3D branch finds further instances by
 *       taking a pose's inlier keypoints out and trying the object again (GuessGenerator.cpp:192-231), guarded by its clique
 *       gate; reprojection alone is too weak a constraint for that -- at the reference's min_inliers of 8 a second round
 *       assembles chance poses from the leftovers of repeating texture (measured: 7 poses instead of 1 on the rendered-view
 *       test). A caller that expects several instances of an object supplies depth.
 *   rng: ONE draw per call supplies the seed (hypotheses do not walk the stream, which is what lets them run side by side).
 *   outputs as todhip_verify: poses (object -> camera), inlier_kp = keypoint indices of each pose's consensus set, ascending. */
int todhip_verify_2d(todhip_ctx*, const float* kp_xy, uint32_t nq, const float* K9, const uint32_t* row_ptr,
                     const todhip_dmatch* matches, const float* matches_xyz, const float* spans, uint32_t n_objs,
                     const todhip_verify_params*, todhip_rng* rng, todhip_pose* poses, uint32_t* n_poses, uint32_t* inlier_kp,
                     uint32_t* n_inlier_kp);

/* The same with the keypoints and the matcher's fixed-stride outputs in HBM (what todhip_match_device / todhip_merge_shards_device
 * left there): d_kp_xy[nq*2] f32, d_counts[nq] u32, d_matches[nq*k], d_matches_xyz[nq*k*3]. Same result as the call above. The
 * matches never come to the host: ClusterPerObject runs on the device, the host reads the number of objects worth a RANSAC
 * (one word) and, at the end, the poses and their consensus flags. */
int todhip_verify_2d_device(todhip_ctx*, const void* d_kp_xy, uint32_t nq, const float* K9, const void* d_counts, const void* d_matches,
                            const void* d_matches_xyz, uint32_t k, const float* spans, uint32_t n_objs, const todhip_verify_params*,
                            todhip_rng* rng, todhip_pose* poses, uint32_t* n_poses, uint32_t* inlier_kp, uint32_t* n_inlier_kp);

/* A batch of n_frames frames in the launches of one (the layout of todhip_verify_batch_device: per-frame arrays back to back,
 * rng[n_frames], poses of frame f at poses[pose_ptr[f] .. pose_ptr[f+1]), inlier_kp frame-local). Each frame's result equals the
 * single-frame call's. */
int todhip_verify_2d_batch_device(todhip_ctx*, uint32_t n_frames, const void* d_kp_xy, uint32_t nq, const float* K9, const void* d_counts,
                                  const void* d_matches, const void* d_matches_xyz, uint32_t k, const float* spans, uint32_t n_objs,
                                  const todhip_verify_params*, todhip_rng* rng, todhip_pose* poses, uint32_t* n_poses,
                                  uint32_t* pose_ptr, uint32_t* inlier_kp, uint32_t* n_inlier_kp);

/* ---- training ----------------------------------------------------------------- */
/* Per-observation arithmetic of the reference's Trainer cell (src/training/Trainer.cpp:121-187, training.cpp:57-195):
 * ORB on the masked view (the reference uses cv::ORB defaults: 500 features, 8 levels, scale 1.2 -- :148-149),
 * validateKeyPoints (mask eroded 4x, +-2 pixel rescue, depth validity), depthTo3dSparse, cameraToWorld, mergePoints.
 * A model is accumulated on the device; its rows are what todhip_db_load ingests (ModelFiller.cpp:23-24). */
typedef struct todhip_model todhip_model;
int  todhip_model_begin(todhip_ctx*, uint32_t capacity_rows, todhip_model** out);
/* gray/mask: H x W u8 (mask != 0 = object); depth: H x W float metres (NaN = none) or uint16 millimetres (0 = none),
 * of the image size (other sizes: todhip_rescale_depth first, as Trainer.cpp:142-143 does); K9, R9 row-major; T3. */
int  todhip_model_add_observation(todhip_ctx*, todhip_model*, const uint8_t* gray, const uint8_t* mask, const void* depth,
                                  int depth_is_u16, uint32_t H, uint32_t W, const float* K9, const float* R9, const float* T3,
                                  uint32_t n_features, uint32_t n_levels, float scale_factor, const int8_t* pattern,
                                  uint32_t* n_added);
/* rescale_depth (Trainer.cpp:62-81; on the detection side ecto_opencv's RescaledRegisteredDepth cell, detector.py:26,62):
 * depth_in dH x dW, float metres or uint16 millimetres -> depth_out H x W float metres (NaN = none). Equal sizes: the
 * unit conversion only (:68-71). Otherwise a resize into the top int(dH * (W / dW)) rows, NaN below (:73-80):
 * nearest == 0 is what the reference executes -- its cv::resize call passes CV_INTER_NN in the `fx` position (:78), so
 * cv::resize's default bilinear interpolation runs; nearest != 0 is the nearest-neighbour resize its comment intends.
 * TODHIP_EINVAL where the reference's rowRange/resize would throw (that row count is 0 or exceeds H). The output feeds
 * todhip_model_add_observation / todhip_verify_device_depth (depth_is_u16 = 0). */
int  todhip_rescale_depth(todhip_ctx*, const void* depth_in, int depth_is_u16, uint32_t dH, uint32_t dW, float* depth_out,
                          uint32_t H, uint32_t W, int nearest);
int  todhip_rescale_depth_device(todhip_ctx*, const void* d_depth_in, int depth_is_u16, uint32_t dH, uint32_t dW,
                                 void* d_depth_out, uint32_t H, uint32_t W, int nearest);
/* *n: capacity in rows in, rows out. desc: rows x 32, pts_xyz: rows x 3 (object/world frame). */
int  todhip_model_finish(todhip_ctx*, todhip_model*, uint8_t* desc, float* pts_xyz, uint32_t* n);
/* The model where it is: device pointers of its descriptors (n x 32 u8) and points (n x 3 f32), valid until todhip_model_free.
 * With todhip_db_load_device a freshly trained model reaches the matcher without visiting the host (the reference writes it to
 * CouchDB, ModelFiller.cpp:23-24, and DescriptorMatcher::parameter_callback reads it back, DescriptorMatcher.cpp:60-129). */
int  todhip_model_device(todhip_ctx*, todhip_model*, const void** d_desc, const void** d_pts_xyz, uint32_t* n);
void todhip_model_free(todhip_ctx*, todhip_model*);

/* ---- diagnostics --------------------------------------------------------------------------------- */
/* Per-RANSAC-round trace of the last todhip_verify call (what GuessGenerator.cpp:202 prints, plus the
 * rand() stream position): `iterations` = iterations_ at loop exit (ransac.h:95-135). */
typedef struct {
  uint32_t object, iterations, best_iteration;
  int32_t  best_count;
  uint64_t draws_before, draws_after;
  uint32_t n_inlier_kp, accepted;
} todhip_round_trace;
int todhip_verify_trace(const todhip_ctx*, todhip_round_trace* out, uint32_t* n /* capacity in, count out */);
/* FillAdjacency (adjacency_ransac.cpp:127-172) alone: n matches (train/query n x 3, per-match keypoint pixel
 * n x 2) -> physical and sample bit matrices, n rows of ceil(n/64) u64 words each. */
int todhip_test_adjacency(todhip_ctx*, const float* train_xyz, const float* query_xyz, const float* kp_xy_per_match,
                          uint32_t n, float span, float sensor_error, uint64_t* phys, uint64_t* samp);
/* The verifier's clique search (maximum_clique.cpp:343-369, FindClique(minimal_size); 0xFFFFFFFF =
 * FindMaximumClique) on an explicit graph of m <= 1024 vertices, as the reference's own gtests call it
 * (test/test_maximum_clique.cpp:7-53). out3 = {clique size, internal error flag, search steps}. */
int todhip_test_clique(todhip_ctx*, uint32_t m, const uint32_t* edges, uint32_t n_edges, uint32_t minimal_size,
                       uint32_t* out3);
/* The same graph through the form of the search the verifier's gate runs: sac_model_registration_graph.h:260-262 only asks whether
 * the clique FindClique(minimal_size) returns is larger than minimal_size, which is decided once Q holds minimal_size vertices
 * with a common neighbour (or at the first leaf of at least that size) -- the search stops there. out3[0] = that clique's size
 * when it is <= minimal_size, a lower bound > minimal_size otherwise; out3[2] = steps actually walked. */
int todhip_test_clique_gate(todhip_ctx*, uint32_t m, const uint32_t* edges, uint32_t n_edges, uint32_t minimal_size,
                            uint32_t* out3);

/* FillAdjacency + selectWithinDistance (sac_model_registration_graph.h:171-269) for given sample triples
 * (samples_ order): counts[t] = consensus size, 0 when the clique gate rejects. stop_level 1 skips the clique
 * search (counts[t] = -|F| where it would have run). dbg (optional, dbg_stride u32 per triple): {count, |F|,
 * then (F member, its degree in the induced graph) pairs}. */
int todhip_test_consensus(todhip_ctx*, const float* train_xyz, const float* query_xyz, const float* kp_xy_per_match,
                          uint32_t n, float span, float sensor_error, const uint32_t* triples, uint32_t n_triples,
                          uint32_t stop_level, int32_t* counts, uint32_t* dbg, uint32_t dbg_stride);

#ifdef __cplusplus
}
#endif
#endif /* TODHIP_H_ */
