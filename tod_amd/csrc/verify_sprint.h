// Included by verify.hip inside its anonymous namespace, after eval_kernel / growth_kernel / invalidate_kernel.
//
// sprint_kernel: the small objects of a frame, verified from the first draw to the last invalidation by ONE wave, with no host
// round trip in between. A frame of self-similar texture spreads its matches over a couple of hundred objects
// (GuessGenerator.cpp:170-235 walks them in ascending imgIdx, all through one rand() stream): nearly all of them are decided by
// arithmetic on their first round's statistics (fewer than 3 valid matches, or no triangle in the sample graph: the host skips
// them without a kernel, Engine::start_round), a handful are small live RANSAC problems, and one is the object that is really
// there. Lock-step ticks made every live small object cost one to three host round trips, and -- worse -- made the frames of a
// batch reach their big object at different ticks. Here the host hands the wave the list of a frame's live objects of at most 64
// matches between two big ones (with the number of draws the skipped objects in between consume); the wave runs, per object,
// AdjacencyRansac::Ransac rounds (adjacency_ransac.cpp:234-309) until one fails:
//   round statistics      lane v = match v: sample degree inside the valid set, the ">= 7" mask, |valid|, triangle test
//   computeModel          ransac.h:80-143, serial semantics: per window of 64 rand() positions every lane runs ONE
//                         drawIndexSampleHelper attempt from its own position (sac_model_registration_graph.h:102-132, adjacency rows
//                         are single 64-bit words in LDS), the wave walks position -> position + consumed with v_readlane (getSamples,
//                         :141-168, incl. the 1000-attempt give-up), the iterations found are evaluated one per lane (3-row AND +
//                         popcount, degree filter, :171-238), the rare hypothesis that reaches the clique search runs it at wave level
//                         (gate_eval: the code of eval_kernel), and the strictly-better / adaptive-k bookkeeping of ransac.h:95-135 is
//                         replayed in iteration order. The one host-only piece is k = log(0.01) / log(1 - w^3): the loop test
//                         `iterations_ < k` equals `iterations_ < ceil(k)`, so a 65 x 65 table of ceil(k) per (|valid|, n_best),
//                         computed with libm on the host, makes the replay exact.
//   growth                adjacency_ransac.cpp:255-308 with the sequential sums of growth_kernel (lanes 0..5 / 0..8 walk the inliers in
//                         list order), kabsch_solve / growth_admits / pose_invert shared with it
//   InvalidateQueryIndices + InvalidateIndices (:63-123) on the valid word
// and writes one record per round (and the inlier keypoints of accepted poses) into device-visible pinned memory.
// The wave returns early -- the host relaunches from the unfinished object, nothing is lost -- when the device copy of the rand()
// stream ends or the record buffer is full.

constexpr uint32_t kSprintN = 64;              // matches per object: one 64-bit word per adjacency row
constexpr uint32_t kSprintWin = 256;           // rand() words of a window staged in LDS: 64 start positions + 192 of look-ahead
constexpr uint32_t kSprintGateLds = 8192;      // gate_eval's carve (4.1 KB for 64 vertices) + the first part of its level stack
constexpr uint32_t kSprintOwnLds = 512 + 512 + 4 * kSprintWin + 4 * 6 * 64;
constexpr uint32_t kSprintLds = kSprintOwnLds + kSprintGateLds;
constexpr uint32_t kSprintStackCap = 32u * 1024u;   // u16 entries of clique level stack beyond the LDS part
constexpr uint32_t kSprintHdrWords = 16;
constexpr uint32_t kSprintRecWords = 32;
constexpr uint32_t kSprintMaxRecs = 96;        // rounds per launch
constexpr uint32_t kSprintMaxObjs = 48;        // live objects per launch
enum { SPRINT_DONE = 0, SPRINT_NEED_STREAM = 1, SPRINT_FULL = 2, SPRINT_ERROR = 3 };

// header (words): [0] records written, [1] exit reason, [2] objects completed, [3..4] stream position at exit (of the last completed
// round), [5] gate calls, [6] hypotheses evaluated, [7] error detail
// record (words): [0] index of the object in the launch's list, [1] iterations, [2] best iteration, [3] best count (int), [4..5] draws
// consumed by the round, [6] inlier keypoints, [7] 1 = growth ran (n_best > 0), [8..16] R, [17..19] T (inverted pose), [20] offset of
// the keypoint list in kp_out, [21] match inliers, [22] growth passes, [23] model inliers
struct SprintObj { ObjJob job; uint64_t skip; uint32_t index, pad; };   // skip: draws consumed by the objects the host skipped before it
struct SprintArgs {
  const SprintObj* objs;          // pinned host memory, read directly
  const uint32_t* rnd;            // the frame's rand() stream (StreamCache::dev), position 0 = the slot's start state
  uint64_t rnd_len;               // valid words
  uint64_t pos0;                  // stream position when the launch starts
  const uint32_t* kceil;          // 65 x 65: ceil(k) per (|valid|, n_best)
  uint32_t* out;                  // header + records (pinned host memory)
  uint32_t* kp_out;               // inlier keypoint lists (pinned host memory)
  uint32_t* status;               // gate_eval's status words (device)
  uint16_t* stack;                // gate_eval's global level stack
  uint32_t n_objs, max_iterations, min_inliers;
  float err;
  uint32_t rec_cap, kp_cap;
};
static_assert(sizeof(SprintArgs) <= 124, "32 argument sets per launch");

__device__ __forceinline__ u64 uni64(u64 v) { return ((u64)uni((uint32_t)(v >> 32)) << 32) | uni((uint32_t)v); }

__device__ __forceinline__ uint32_t nth_set_bit64(u64 w, uint32_t n) {   // n-th set bit (ascending), n < popc(w)
  uint32_t pos = 0;
#pragma unroll
  for (uint32_t shift = 32; shift > 0; shift >>= 1) {
    const uint32_t cnt = (uint32_t)__popcll((w >> pos) & ((1ull << shift) - 1ull));
    if (n >= cnt) { n -= cnt; pos += shift; }
  }
  return pos;
}

// one drawIndexSampleHelper attempt (sac_model_registration_graph.h:102-132) from window position `at`, by one lane.
// returns status | consumed << 2 | s0 << 14 | s1 << 20 | s2 << 26 (samples_ order: deepest pick first, :118-121)
__device__ __forceinline__ uint32_t sprint_attempt(const uint32_t* s_rnd, const u64* s_samp, const uint32_t* __restrict__ rnd, u64 pos,
                                                  u64 rnd_len, uint32_t at, u64 valid, uint32_t nvalid) {
  u64 a_mask = valid;
  uint32_t nA = nvalid, i = at, status = DRAW_FAIL, s0 = 0, s1 = 0, s2 = 0;
  bool over = false;
  const u64 left = rnd_len > pos ? rnd_len - pos : 0ull;   // stream words from the window's start on
  const uint32_t lim = left > 0x7FFFFFFFull ? 0x7FFFFFFFu : (uint32_t)left;
  auto next = [&](uint32_t& r) {                           // the draw at window position i
    if (i >= lim) { over = true; r = 0; }
    else if (i < kSprintWin) { r = s_rnd[i]; }
    else { r = rnd[pos + i]; }
    ++i;
  };
  while (nA > 0) {                                          // level "3 samples left"
    uint32_t r;
    next(r);
    if (over) break;
    const uint32_t a = nth_set_bit64(a_mask, r % nA);       // valid_samples[rand() % size], :111
    u64 b_mask = a_mask & s_samp[a];                        // set_intersection with the sample neighbours, :113-117
    uint32_t nB = (uint32_t)__popcll(b_mask);
    bool ok = false;
    uint32_t b = 0, c = 0;
    while (nB > 0) {                                        // level "2 samples left"
      next(r);
      if (over) break;
      b = nth_set_bit64(b_mask, r % nB);
      const u64 c_mask = b_mask & s_samp[b];
      const uint32_t nC = (uint32_t)__popcll(c_mask);
      if (nC > 0) {                                         // level "1 sample left": any pick succeeds
        next(r);
        if (over) break;
        c = nth_set_bit64(c_mask, r % nC);
        ok = true;
        break;
      }
      b_mask &= ~(1ull << b);                               // std::remove of the failed pick, :125-128
      --nB;
    }
    if (over) break;
    if (ok) { status = DRAW_OK; s0 = c; s1 = b; s2 = a; break; }
    a_mask &= ~(1ull << a);
    --nA;
  }
  if (over) return DRAW_OVERFLOW;
  return status | ((i - at) << 2) | (s0 << 14) | (s1 << 20) | (s2 << 26);
}

__global__ __launch_bounds__(128) void sprint_kernel(Slots<SprintArgs, kWideSlots> SL) {
  TOD_LATENCY_PRIO();
  const SprintArgs& A = SL.a[blockIdx.x];
  extern __shared__ __align__(16) unsigned char lds_raw[];
  u64* const s_phys = reinterpret_cast<u64*>(lds_raw);                 // 64 rows
  u64* const s_samp = s_phys + 64;                                     // 64 rows
  uint32_t* const s_rnd = reinterpret_cast<uint32_t*>(s_samp + 64);    // kSprintWin
  float* const s_pts = reinterpret_cast<float*>(s_rnd + kSprintWin);   // 64 x {train xyz, query xyz}
  unsigned char* const lds_gate = lds_raw + kSprintOwnLds;
  const uint32_t l = lane_id();
  const uint32_t* __restrict__ rnd = A.rnd;
  const u64 rnd_len = A.rnd_len;
  u64 pos = A.pos0;                                                    // stream position after the last completed round
  uint32_t n_rec = 0, kp_used = 0, reason = SPRINT_DONE, hyps = 0, n_done_objs = 0, err_detail = 0;
  const uint32_t max_it = A.max_iterations;

  for (uint32_t j = 0; j < A.n_objs && reason == SPRINT_DONE; ++j) {
    const SprintObj* so = A.objs + j;
    const ObjJob job = so->job;
    const uint32_t n = uni(job.n);
    if (n > kSprintN || uni(job.W) != 1u) { reason = SPRINT_ERROR; err_detail = 1; break; }
    u64 obj_pos = pos + uni64(so->skip);                               // the draws of the skipped objects before this one
    u64 valid = uni64(job.valid[0]);
    const u64 finite = uni64(job.finite[0]);
    const u64 my_phys = l < n ? job.phys[l] : 0ull;
    const u64 my_samp = l < n ? job.samp[l] : 0ull;
    const uint32_t my_q = l < n ? job.qidx[l] : 0xFFFFFFFFu;
    float my_t[3], my_qp[3];
    for (int c = 0; c < 3; ++c) { my_t[c] = l < n ? job.train[3 * l + c] : 0.f; my_qp[c] = l < n ? job.query[3 * l + c] : 0.f; }
    __syncthreads();                                                   // the previous object's readers are done
    s_phys[l] = my_phys; s_samp[l] = my_samp;
    for (int c = 0; c < 3; ++c) { s_pts[l * 6u + c] = my_t[c]; s_pts[l * 6u + 3 + c] = my_qp[c]; }
    __syncthreads();

    bool obj_done = false;
    while (!obj_done) {                                                // rounds: GuessGenerator.cpp:192-231
      if (n_rec >= A.rec_cap || n_rec >= kSprintMaxRecs || kp_used + kSprintN > A.kp_cap) { reason = SPRINT_FULL; break; }
      // ---- round statistics (round_prep_kernel)
      const bool isv = (valid >> l) & 1ull;
      const uint32_t d = isv ? (uint32_t)__popcll(my_samp & valid) : 0u;
      const u64 deg7 = __ballot(isv && d >= kGateMinimal);
      const uint32_t nvalid = (uint32_t)__popcll(valid);
      const uint32_t degsum = wave_sum(d);
      bool tri = false;
      if (isv && d >= 2u) {
        u64 nb = my_samp & valid & (l == 63u ? 0ull : (~0ull << (l + 1u)));
        while (nb && !tri) {
          const uint32_t o = (uint32_t)__ffsll((long long)nb) - 1u;
          nb &= nb - 1ull;
          tri = (my_samp & s_samp[o] & valid) != 0ull;
        }
      }
      const bool triangle = __ballot(tri) != 0ull;
      const u64 round_pos = obj_pos;                                   // the round's first draw
      uint32_t iterations = 0, best_it = 0, n_kp = 0, grew = 0, n_match = 0, passes = 0, n_model = 0;
      int32_t n_best = -INT_MAX;
      uint32_t best_tri = 0;
      bool round_ok = true;                                            // false: the round could not be completed (stream ended)
      if (nvalid < 3u) {
        obj_done = true;                                               // Ransac returns nothing and draws nothing (:238-241)
      } else if (!triangle) {
        obj_pos += (u64)kMaxSampleChecks * ((u64)nvalid + degsum / 2u);   // 1000 failing attempts of |valid| + |E| draws each
        obj_done = true;
      } else {
        // ---- computeModel (ransac.h:80-143)
        uint32_t k_ceil = 1u, attempts = 0u;
        u64 p_win = obj_pos;                                           // stream position of the window's lane 0
        bool stop = false;
        while (!stop) {
          __syncthreads();
          for (uint32_t w = l; w < kSprintWin; w += 64u) s_rnd[w] = (p_win + w) < rnd_len ? rnd[p_win + w] : 0u;
          __syncthreads();
          const uint32_t entry = sprint_attempt(s_rnd, s_samp, rnd, p_win, rnd_len, l, valid, nvalid);
          // getSamples x iterations (:141-168): position -> position + consumed
          uint32_t p = 0, cnt = 0, it_start = 0, it_end = 0;           // lane i: window position where iteration i's attempt starts / ends
          bool sel_empty = false;
          while (p < 64u) {
            const uint32_t e = rdlane(entry, p);
            const uint32_t st_e = e & 3u;
            if (st_e == DRAW_OVERFLOW) { round_ok = false; break; }
            const uint32_t nx = p + ((e >> 2) & 0xFFFu);
            if (st_e == DRAW_OK) {
              if (l == cnt) { it_start = p; it_end = nx; }
              ++cnt;
              attempts = 0u;
            } else if (++attempts >= kMaxSampleChecks) {               // getSamples gives up: samples.clear(), :167
              p = nx;
              sel_empty = true;
              break;
            }
            p = nx;
          }
          if (!round_ok) break;
          // selectWithinDistance (:171-238) of the window's iterations, one per lane
          const bool have = l < cnt;
          const uint32_t my_e = (uint32_t)__shfl((int)entry, (int)(have ? it_start : 0u));
          const uint32_t t0 = (my_e >> 14) & 63u, t1 = (my_e >> 20) & 63u, t2 = (my_e >> 26) & 63u;
          int32_t c_mine = 0;
          u64 F_mine = 0ull;
          bool pending = false;
          if (have) {
            const u64 P = s_phys[t0] & s_phys[t1] & s_phys[t2] & valid & finite;   // common physical neighbours (:178-184), D2
            c_mine = (int32_t)__popcll(P) + 3;                                     // + the samples themselves (:185-186)
            if (c_mine > (int32_t)kGateMinimal) {                                  // :203-205
              F_mine = (P | (1ull << t0) | (1ull << t1) | (1ull << t2)) & deg7;    // :211-213
              if ((uint32_t)__popcll(F_mine) <= kGateMinimal) {
                c_mine = 0;                                                        // :214-218
              } else {
                bool any = false;                                                  // :221-238
                u64 f = F_mine;
                while (f && !any) {
                  const uint32_t v = (uint32_t)__ffsll((long long)f) - 1u;
                  f &= f - 1ull;
                  any = (uint32_t)__popcll(s_samp[v] & F_mine) > kGateMinimal;
                }
                if (any) pending = true; else c_mine = 0;
              }
            }
          }
          u64 pend = __ballot(pending);
          // ransac.h:95-135 in iteration order; a hypothesis that needs the clique search gets it when the loop reaches it
          uint32_t i = 0;
          while (i < cnt && !stop) {
            if ((pend >> i) & 1ull) {
              const u64 Fu = ((u64)rdlane((uint32_t)(F_mine >> 32), i) << 32) | rdlane((uint32_t)F_mine, i);
              const uint32_t cu = rdlane((uint32_t)c_mine, i);
              WaveBits F;
#pragma unroll
              for (int q = 0; q < kWPL; ++q) F.w[q] = 0ull;
              if (l == 0u) F.w[0] = Fu;
              EvalArgs E;
              E.job = job; E.iter_samples = nullptr; E.it_begin = 0; E.it_end = 0; E.counts = nullptr; E.gate_m = nullptr;
              E.work = nullptr; E.status = A.status; E.deferred = nullptr; E.stacks = A.stack; E.stack_cap = kSprintStackCap;
              E.lds_bytes = kSprintGateLds; E.from_deferred = 0; E.n_deferred = 0; E.adjc_scratch = nullptr; E.dbg = nullptr;
              E.dbg_stride = 0; E.stop_level = 0; E.n_items_dev = nullptr;
              const int32_t res = gate_eval<false, false>(E, F, (uint32_t)__popcll(Fu), iterations, cu, lds_gate, A.stack);
              if (res == INT_MIN) { reason = SPRINT_ERROR; err_detail = 2; round_ok = false; stop = true; break; }
              if (l == i) c_mine = res;
              pend &= ~(1ull << i);
            }
            const int32_t c_i = (int32_t)rdlane((uint32_t)c_mine, i);
            ++hyps;
            if (c_i > n_best) {                                        // strictly better (:115)
              n_best = c_i;
              best_it = iterations;
              best_tri = rdlane(my_e, i);
              const uint32_t nb_idx = (uint32_t)(n_best < 0 ? 0 : n_best);
              k_ceil = uni(A.kceil[nvalid * 65u + min(nb_idx, 64u)]);  // ceil(log(0.01) / log(1 - w^3)), :121-130
            }
            ++iterations;
            const uint32_t end_i = rdlane(it_end, i);
            if (iterations > max_it || !(iterations < k_ceil)) { stop = true; obj_pos = p_win + end_i; }   // :95, :132-134
            ++i;
          }
          if (stop) break;
          if (sel_empty) { stop = true; obj_pos = p_win + p; break; }  // selection.empty() -> break (:100-101)
          p_win += p;
        }
        if (!round_ok) {
          if (reason == SPRINT_DONE) reason = SPRINT_NEED_STREAM;
          break;                                                       // nothing of this round has been recorded: it starts again
        }
        if (n_best > 0) {
          // ---- growth (adjacency_ransac.cpp:255-303)
          grew = 1u;
          const uint32_t b0 = (best_tri >> 14) & 63u, b1 = (best_tri >> 20) & 63u, b2 = (best_tri >> 26) & 63u;
          u64 inl = uni64((s_phys[b0] & s_phys[b1] & s_phys[b2] & valid & finite) | (1ull << b0) | (1ull << b1) | (1ull << b2));
          u64 rest = valid & ~inl;                                     // :260-264
          n_model = (uint32_t)__popcll(inl);
          bool do_final = false;
          double thresh = (double)(A.err * A.err);                     // float product widened, :267
          float R[9], T[3];
          while (true) {
            const uint32_t cnt = (uint32_t)__popcll(inl);
            float csum = 0.f;                                          // lanes 0..5: sequential float sums in list order (centroids)
            if (l < 6u) {
              u64 b = inl;
              while (b) { const uint32_t v = (uint32_t)__ffsll((long long)b) - 1u; b &= b - 1ull; csum += s_pts[v * 6u + l]; }
            }
            const double inv = 1. / (float)cnt;                        // Vec /= float: times the double reciprocal
            const float cmine = (float)(csum * inv);
            float C[6];
#pragma unroll
            for (int c = 0; c < 6; ++c) C[c] = __shfl(cmine, c);
            double h = 0.0;                                            // lanes 0..8: H = sub_training^T * sub_query, double accumulation
            const uint32_t r = min(l / 3u, 2u), c = l % 3u;
            const float ct = __shfl(cmine, (int)r), cq = __shfl(cmine, (int)(3u + c));
            if (l < 9u) {
              u64 b = inl;
              while (b) {
                const uint32_t v = (uint32_t)__ffsll((long long)b) - 1u;
                b &= b - 1ull;
                const float a = s_pts[v * 6u + r] - ct, bb = s_pts[v * 6u + 3u + c] - cq;
                h += (double)a * (double)bb;
              }
            }
            double Hd[9];
#pragma unroll
            for (int e = 0; e < 9; ++e) Hd[e] = __shfl(h, e);
            kabsch_solve(Hd, C, R, T);                                 // every lane computes the same R, T
            ++passes;
            const bool in_rest = (rest >> l) & 1ull;
            const bool pass = in_rest && growth_admits(R, T, my_qp, my_t, thresh);   // :275-283
            const u64 extra = __ballot(pass);
            inl |= extra; rest &= ~extra;
            if (do_final) break;
            if (!extra) { do_final = true; thresh *= 4; }              // :295-301
          }
          n_match = (uint32_t)__popcll(inl);
          // unique keypoint indices in ascending match order (:306-308); qidx is non-decreasing in the match index
          const bool in = (inl >> l) & 1ull;
          const u64 lower = inl & ((1ull << l) - 1ull);
          const uint32_t pq = (uint32_t)__shfl((int)my_q, (int)(lower ? 63u - (uint32_t)__clzll((long long)lower) : 0u));
          const bool is_new = in && (lower == 0ull || my_q != pq);
          const u64 newb = __ballot(is_new);
          n_kp = (uint32_t)__popcll(newb);
          if (is_new) A.kp_out[kp_used + (uint32_t)__popcll(newb & ((1ull << l) - 1ull))] = my_q;
          float Ro[9], To[3];
          pose_invert(R, T, Ro, To);                                   // :304-305
          if (l == 0u) {
            uint32_t* rec = A.out + kSprintHdrWords + (size_t)n_rec * kSprintRecWords;
            for (int e = 0; e < 9; ++e) rec[8 + e] = __float_as_uint(Ro[e]);
            for (int e = 0; e < 3; ++e) rec[17 + e] = __float_as_uint(To[e]);
          }
          if (n_kp >= A.min_inliers) {
            // ---- InvalidateQueryIndices (:93-123): every valid match whose keypoint is an inlier keypoint, then
            // InvalidateIndices (:63-89): sample degree < 3 until nothing changes (only if something was removed, :68)
            bool hit = false;
            u64 g = inl;
            while (g) {
              const uint32_t u = (uint32_t)__ffsll((long long)g) - 1u;
              g &= g - 1ull;
              hit = hit || my_q == rdlane(my_q, u);
            }
            const u64 gone = __ballot(hit && ((valid >> l) & 1ull));
            if (gone) {
              valid &= ~gone;
              while (true) {
                const bool v_now = (valid >> l) & 1ull;
                const u64 low = __ballot(v_now && (uint32_t)__popcll(my_samp & valid) < 3u);   // min_sample_size_
                if (!low) break;
                valid &= ~low;
              }
              if (l == 0u) job.valid[0] = valid;                       // a relaunch of this object starts from here
            }
          } else {
            obj_done = true;                                           // GuessGenerator.cpp:205-206
          }
        } else {
          obj_done = true;                                             // inliers_.empty(): computeModel() == false (:137-138)
        }
      }
      // ---- the round's record
      if (l == 0u) {
        uint32_t* rec = A.out + kSprintHdrWords + (size_t)n_rec * kSprintRecWords;
        const u64 consumed = obj_pos - round_pos;
        rec[0] = j; rec[1] = iterations; rec[2] = best_it; rec[3] = (uint32_t)n_best;
        rec[4] = (uint32_t)consumed; rec[5] = (uint32_t)(consumed >> 32);
        rec[6] = n_kp; rec[7] = grew; rec[20] = kp_used; rec[21] = n_match; rec[22] = passes; rec[23] = n_model;
      }
      kp_used += n_kp;
      ++n_rec;
      pos = obj_pos;                                                   // (includes the object's skip once its first round is recorded)
    }
    if (reason != SPRINT_DONE) break;
    ++n_done_objs;
  }
  if (l == 0u) {
    A.out[0] = n_rec; A.out[1] = reason; A.out[2] = n_done_objs;
    A.out[3] = (uint32_t)pos; A.out[4] = (uint32_t)(pos >> 32);
    A.out[5] = __hip_atomic_load(A.status + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); A.out[6] = hyps; A.out[7] = err_detail;
  }
}
