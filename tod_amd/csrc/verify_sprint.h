// Included by verify.hip inside its anonymous namespace, after eval_kernel / growth_kernel / invalidate_kernel.
//
// sprint_kernel: the small objects of a frame, verified from the first draw to the last invalidation by ONE workgroup, with no host
// round trip in between. A frame of self-similar texture spreads its matches over a couple of hundred objects
// (GuessGenerator.cpp:170-235 walks them in ascending imgIdx, all through one rand() stream): nearly all of them are decided by
// arithmetic on their first round's statistics (fewer than 3 valid matches, or no triangle in the sample graph: the host skips
// them without a kernel, Engine::start_round), a handful are small live RANSAC problems, and one is the object that is really
// there. Lock-step ticks made every live small object cost one to three host round trips, and -- worse -- made the frames of a
// batch reach their big object at different ticks. Here the host hands the wave the list of a frame's live objects of at most 64
// matches between two big ones (with the number of draws the skipped objects in between consume); the workgroup runs, per object,
// AdjacencyRansac::Ransac rounds (adjacency_ransac.cpp:234-309) until one fails. Its four waves execute the SAME wave-uniform
// control flow on the same data (every decision is taken from LDS / global values all of them read), so block barriers are always
// reached together; they differ only in the stream positions their lanes speculate on, and wave 0 writes the results:
//   round statistics      lane v = match v: sample degree inside the valid set, the ">= 7" mask, |valid|, triangle test
//   computeModel          ransac.h:80-143, serial semantics: per window of 256 rand() positions every thread runs ONE
//                         drawIndexSampleHelper attempt from its own position (sac_model_registration_graph.h:102-132, adjacency rows
//                         are single 64-bit words in LDS), position -> position + consumed is walked with v_readlane (getSamples,
//                         :141-168, incl. the 1000-attempt give-up), the iterations found are evaluated one per thread (3-row AND +
//                         popcount, degree filter, :171-238), the rare hypothesis that reaches the clique search runs it at wave level
//                         (gate_eval: the code of eval_kernel), and the strictly-better / adaptive-k bookkeeping of ransac.h:95-135 is
//                         replayed in iteration order (a prefix maximum over the window's iterations). The one host-only piece is
//                         k = log(0.01) / log(1 - w^3): the loop test `iterations_ < k` equals `iterations_ < ceil(k)`, so a 65 x 65
//                         table of ceil(k) per (|valid|, n_best), computed with libm on the host, makes the replay exact.
//   growth                adjacency_ransac.cpp:255-308 with the sequential sums of growth_kernel (lanes 0..5 / 0..8 walk the inliers in
//                         list order), kabsch_solve / growth_admits / pose_invert shared with it
//   InvalidateQueryIndices + InvalidateIndices (:63-123) on the valid word
// and writes one record per round (and the inlier keypoints of accepted poses) into device-visible pinned memory.
// The wave returns early -- the host relaunches from the unfinished object, nothing is lost -- when the device copy of the rand()
// stream ends or the record buffer is full.

constexpr uint32_t kSprintN = 64;              // matches per object: one 64-bit word per adjacency row
constexpr uint32_t kSprintWaves = 4;
constexpr uint32_t kSprintThreads = 64 * kSprintWaves;
constexpr uint32_t kSprintPerWave = 256;       // rand() positions a wave speculates on per window (its lanes take them one after the other)
constexpr uint32_t kSprintPos = kSprintPerWave * kSprintWaves;   // positions per window
constexpr uint32_t kSprintRing = 4096;         // rand() words staged in LDS: a window's kSprintPos positions + look-ahead; the longest
                                               // attempt there is (64 top-level draws + 2016 edges) fits behind the ring's first position
constexpr uint32_t kSprintGateLds = 8192;      // per wave: gate_eval's carve (4.1 KB for 64 vertices) + the first part of its level stack
constexpr uint32_t kSprintMaxIt = 384;         // iterations one window can hold (1024 positions, >= 3 draws each: 342)
// LDS: phys rows, samp rows, points, stream ring, attempt entries, modulo magics, ceil(k) row, iteration records, counts, the walk's
// jump table and reach flags (kSprintPos + 64 each), control words, gate areas
constexpr uint32_t kSprintOwnLds = 512 + 512 + 4 * 6 * 64 + 4 * kSprintRing + 4 * kSprintPos + 4 * 80 + 4 * 80 + 4 * kSprintMaxIt + 4 * kSprintMaxIt +
                                   2 * 4 * (kSprintPos + 64) + 128;
constexpr uint32_t kSprintLds = kSprintOwnLds + kSprintWaves * kSprintGateLds;
constexpr uint32_t kSprintStackCap = 32u * 1024u;   // u16 entries of clique level stack per wave beyond the LDS part
constexpr uint32_t kSprintHdrWords = 16;
constexpr uint32_t kSprintRecWords = 32;
constexpr uint32_t kSprintMaxRecs = 96;        // rounds per launch
constexpr uint32_t kSprintMaxObjs = 48;        // live objects per launch
enum { SPRINT_DONE = 0, SPRINT_NEED_STREAM = 1, SPRINT_FULL = 2, SPRINT_ERROR = 3 };
constexpr uint32_t DRAW_LONG = 3;              // the attempt ran past the staged part of the stream: the window ends in front of it

// header (words): [0] records written, [1] exit reason, [2] objects completed, [3..4] stream position at exit (of the last completed
// round), [5] gate calls, [6] hypotheses evaluated, [7] error detail, [8] windows, [9..15] s_memtime ticks (100 MHz) spent staging the
// stream / in the attempts / the walk / the evaluation / the bookkeeping / the growth / in all
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
  uint16_t* stack;                // gate_eval's global level stacks (kSprintWaves x kSprintStackCap)
  uint32_t n_objs, max_iterations, min_inliers;
  float err;
  uint32_t rec_cap, kp_cap;
};
static_assert(sizeof(SprintArgs) <= 124, "32 argument sets per launch");

__device__ __forceinline__ u64 uni64(u64 v) { return ((u64)uni((uint32_t)(v >> 32)) << 32) | uni((uint32_t)v); }

// n-th set bit (ascending), n < popc(w): pick the half, then five bit-field steps on 32 bits
__device__ __forceinline__ uint32_t nth_set_bit64(u64 w, uint32_t n) {
  const uint32_t lo = (uint32_t)w, hi = (uint32_t)(w >> 32);
  const uint32_t c0 = (uint32_t)__popc(lo);
  const bool up = n >= c0;
  const uint32_t x = up ? hi : lo;
  n = up ? n - c0 : n;
  uint32_t pos = 0;
#pragma unroll
  for (uint32_t sh = 16; sh > 0; sh >>= 1) {
    const uint32_t cnt = (uint32_t)__popc(__builtin_amdgcn_ubfe(x, pos, sh));
    const bool mv = n >= cnt;
    n = mv ? n - cnt : n;
    pos = mv ? pos + sh : pos;
  }
  return pos + (up ? 32u : 0u);
}
// r % d for d in 1..64 with magic[d] = floor((2^32 - 1) / d): the quotient estimate is exact or one short
__device__ __forceinline__ uint32_t sprint_mod(uint32_t r, uint32_t d, const uint32_t* s_magic) {
  const uint32_t q = __umulhi(r, s_magic[d]);
  uint32_t rem = r - q * d;
  return rem >= d ? rem - d : rem;
}

// drawIndexSampleHelper (sac_model_registration_graph.h:102-132) from every position [first, first + kSprintPerWave) of the window,
// by one wave: ONE loop whose every pass performs one draw of whatever attempt a lane is working on (level 3 / 2 / 1 samples left
// is lane state), and a lane that finishes an attempt takes the wave's next position at once -- a wave's time is the sum of its
// positions' draws / 64, not 64 times the longest attempt (a top-level pick fails 3 times out of 4 on these graphs, 13 times in a row
// on the unluckiest of 256 positions).
// s_win = the LDS ring at the window's first position, n_ring = words of it that are staged (an attempt that needs more ends as
// DRAW_LONG: the stream is read from LDS only -- with a fallback to global memory in the same expression hipcc turns every read
// into a flat load), lim = stream words that exist from the window's first position on.
// tri = the valid matches that lie on a triangle of the sample graph (inside valid). A top-level pick a outside tri cannot succeed:
// its level-2 loop draws and erases every member of valid' & N(a) once (one draw each, whatever the values), then a is erased (:125-128)
// -- so the pick costs its own draw + that count, in O(1). Erasing such picks never breaks a triangle, so a pick inside tri still has
// one among the remaining indices and its level-2 loop ends in a success.
// entry = status | consumed << 2 | s0 << 14 | s1 << 20 | s2 << 26 (samples_ order: deepest pick first, :118-121)
__device__ __forceinline__ void sprint_attempts(uint32_t* s_entry, const uint32_t* s_win, uint32_t n_ring, const u64* s_samp,
                                                const uint32_t* s_magic, uint32_t lim, uint32_t first, u64 valid, uint32_t nvalid, u64 tri) {
  const uint32_t l = lane_id();
  const uint32_t end = min(lim, n_ring), range_end = first + kSprintPerWave;
  uint32_t wave_next = first + 64u;                        // next position nobody works on yet (wave-uniform)
  uint32_t at = first + l, i = at;                         // the attempt's first draw, the next draw
  uint32_t level = 0, nA = nvalid, nB = 0, a = 0, b = 0;
  u64 a_mask = valid, b_mask = 0ull, c_mask = 0ull;
  while (true) {
    const bool active = at < range_end;
    if (__ballot(active) == 0ull) break;
    uint32_t entry = 0;
    bool fin = false;
    if (active) {
      const u64 m = level == 0u ? a_mask : (level == 1u ? b_mask : c_mask);
      const uint32_t cnt = (uint32_t)__popcll(m);          // == nA / nB / nC
      if (i >= end) {
        fin = true; entry = end < lim ? DRAW_LONG : (uint32_t)DRAW_OVERFLOW;
      } else {
        const uint32_t r = s_win[i];
        ++i;
        const uint32_t v = nth_set_bit64(m, sprint_mod(r, cnt, s_magic));   // valid_samples[rand() % size], :111
        const u64 row = s_samp[v];
        if (level == 0u) {
          a = v;
          b_mask = a_mask & row;                           // set_intersection with the sample neighbours, :113-117
          nB = (uint32_t)__popcll(b_mask);
          if ((tri >> a) & 1ull) {
            level = 1u;
          } else {                                         // no triangle through a: nB failing level-2 picks, one draw each, then a is erased
            i += nB;
            a_mask &= ~(1ull << a);
            --nA;
            if (i > end) { fin = true; entry = end < lim ? DRAW_LONG : (uint32_t)DRAW_OVERFLOW; }
            else if (nA == 0u) { fin = true; entry = DRAW_FAIL | ((i - at) << 2); }
          }
        } else if (level == 1u) {
          b = v;
          c_mask = b_mask & row;
          if (c_mask) {                                    // level "1 sample left": any pick succeeds
            level = 2u;
          } else {
            b_mask &= ~(1ull << b);                        // std::remove of the failed pick, :125-128
            --nB;
            if (nB == 0u) {                                // (only if a were on no triangle) back to the top level without a
              a_mask &= ~(1ull << a);
              --nA;
              level = 0u;
              if (nA == 0u) { fin = true; entry = DRAW_FAIL | ((i - at) << 2); }
            }
          }
        } else {
          fin = true;
          entry = DRAW_OK | ((i - at) << 2) | (v << 14) | (b << 20) | (a << 26);
        }
      }
      if (fin) s_entry[at] = entry;
    }
    const u64 fb = __ballot(fin);
    if (fin) {                                             // the wave's next positions, in lane order
      at = wave_next + (uint32_t)__popcll(fb & ((1ull << l) - 1ull));
      i = at; level = 0u; nA = nvalid; a_mask = valid;
    }
    wave_next += (uint32_t)__popcll(fb);
  }
}

__device__ __forceinline__ int32_t wave_incl_scan_max(int32_t v) {
  const uint32_t l = lane_id();
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int32_t t = __shfl_up(v, d);
    if (l >= (uint32_t)d) v = max(v, t);
  }
  return v;
}

__global__ __launch_bounds__(kSprintThreads) void sprint_kernel(Slots<SprintArgs, kWideSlots> SL) {
  TOD_LATENCY_PRIO();
  const SprintArgs& A = SL.a[blockIdx.x];
  extern __shared__ __align__(16) unsigned char lds_raw[];
  u64* const s_phys = reinterpret_cast<u64*>(lds_raw);                 // 64 rows
  u64* const s_samp = s_phys + 64;                                     // 64 rows
  float* const s_pts = reinterpret_cast<float*>(s_samp + 64);          // 64 x {train xyz, query xyz}
  uint32_t* const s_rnd = reinterpret_cast<uint32_t*>(s_pts + 6 * 64); // kSprintRing
  uint32_t* const s_entry = s_rnd + kSprintRing;                       // kSprintPos: the window's attempts
  uint32_t* const s_magic = s_entry + kSprintPos;                      // 80 (65 used)
  uint32_t* const s_kceil = s_magic + 80;                              // 80 (65 used): ceil(k) per n_best for the round's |valid|
  uint32_t* const s_it = s_kceil + 80;                                 // kSprintMaxIt: start | end << 10 of the window's iterations
  uint32_t* const s_c = s_it + kSprintMaxIt;                           // kSprintMaxIt: consensus count | gate pending << 31
  uint32_t* const s_jump = s_c + kSprintMaxIt;                         // kSprintPos + 64: the walk's jump table (entry kSprintPos = end)
  uint32_t* const s_reach = s_jump + kSprintPos + 64;                  // kSprintPos + 64: position is on the chain
  uint32_t* const s_ctl = s_reach + kSprintPos + 64;                   // 32 control words
  const uint32_t tid = threadIdx.x, l = lane_id(), wave = uni(tid >> 6);
  unsigned char* const lds_gate = lds_raw + kSprintOwnLds + wave * kSprintGateLds;
  uint16_t* const gstack = A.stack + (size_t)wave * kSprintStackCap;
  const bool writer = tid < 64u;                                       // wave 0 writes the results
  const uint32_t* __restrict__ rnd = A.rnd;
  const u64 rnd_len = A.rnd_len;
  u64 pos = A.pos0;                                                    // stream position after the last completed round
  u64 ring_base = ~0ull;                                               // stream position of s_rnd[0] (nothing staged yet)
  uint32_t n_rec = 0, kp_used = 0, reason = SPRINT_DONE, hyps = 0, n_done_objs = 0, err_detail = 0, n_windows = 0;
  // phase clocks (s_memtime ticks, diagnostics in the header): ring, attempt, walk, evaluate, bookkeeping, growth, everything
  long long t_ring = 0, t_att = 0, t_walk = 0, t_eval = 0, t_book = 0, t_grow = 0;
  const long long t_begin = clock64();
  const uint32_t max_it = A.max_iterations;
  if (tid < 65u) s_magic[tid] = tid ? 0xFFFFFFFFu / tid : 0u;

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
    if (writer) {
      s_phys[l] = my_phys; s_samp[l] = my_samp;
      for (int c = 0; c < 3; ++c) { s_pts[l * 6u + c] = my_t[c]; s_pts[l * 6u + 3 + c] = my_qp[c]; }
    }
    __syncthreads();

    bool obj_done = false;
    while (!obj_done) {                                                // rounds: GuessGenerator.cpp:192-231
      if (n_rec >= A.rec_cap || kp_used + kSprintN > A.kp_cap) { reason = SPRINT_FULL; break; }
      // ---- round statistics (round_prep_kernel); tri: the valid matches on a triangle of the sample graph
      const bool isv = (valid >> l) & 1ull;
      const uint32_t d = isv ? (uint32_t)__popcll(my_samp & valid) : 0u;
      const u64 deg7 = __ballot(isv && d >= kGateMinimal);
      const uint32_t nvalid = (uint32_t)__popcll(valid);
      const uint32_t degsum = wave_sum(d);
      bool on_tri = false;
      if (isv && d >= 2u) {
        u64 nb = my_samp & valid;
        while (nb && !on_tri) {
          const uint32_t o = (uint32_t)__ffsll((long long)nb) - 1u;
          nb &= nb - 1ull;
          on_tri = (my_samp & s_samp[o] & valid) != 0ull;
        }
      }
      const u64 tri = __ballot(on_tri);
      const u64 round_pos = obj_pos;                                   // the round's first draw
      uint32_t iterations = 0, best_it = 0, n_kp = 0, grew = 0, n_match = 0, passes = 0, n_model = 0;
      int32_t n_best = -INT_MAX;
      uint32_t best_tri = 0;
      bool round_ok = true;                                            // false: the round could not be completed (stream ended)
      if (nvalid < 3u) {
        obj_done = true;                                               // Ransac returns nothing and draws nothing (:238-241)
      } else if (tri == 0ull) {
        obj_pos += (u64)kMaxSampleChecks * ((u64)nvalid + degsum / 2u);   // 1000 failing attempts of |valid| + |E| draws each
        obj_done = true;
      } else {
        // ---- computeModel (ransac.h:80-143)
        __syncthreads();
        if (tid < 65u) s_kceil[tid] = A.kceil[nvalid * 65u + tid];    // ceil(log(0.01) / log(1 - w^3)), :121-130
        uint32_t k_ceil = 1u;
        u64 p_win = obj_pos;                                           // stream position of the window's thread 0
        bool stop = false, fresh_ring = false;
        while (!stop) {
          ++n_windows;
          long long tc = clock64();
          // the stream ring holds the window's positions and a look-ahead of at least as many words
          if (fresh_ring || ring_base == ~0ull || p_win < ring_base || p_win + kSprintPos + 512u > ring_base + kSprintRing) {
            __syncthreads();
            ring_base = p_win;
            for (uint32_t w = tid; w < kSprintRing; w += kSprintThreads) s_rnd[w] = (ring_base + w) < rnd_len ? rnd[ring_base + w] : 0u;
            fresh_ring = false;
          }
          __syncthreads();
          { const long long t2 = clock64(); t_ring += t2 - tc; tc = t2; }
          const uint32_t rel = (uint32_t)(p_win - ring_base);
          const u64 left = rnd_len > p_win ? rnd_len - p_win : 0ull;   // stream words from the window's start on
          const uint32_t lim = left > 0x7FFFFFFFull ? 0x7FFFFFFFu : (uint32_t)left;
          sprint_attempts(s_entry, s_rnd + rel, kSprintRing - rel, s_samp, s_magic, lim, wave * kSprintPerWave, valid, nvalid, tri);
          __syncthreads();
          { const long long t2 = clock64(); t_att += t2 - tc; tc = t2; }
          // getSamples x iterations (:141-168): position -> position + consumed, by pointer doubling over the window's positions.
          // With a triangle among the valid matches every attempt ends in a success (sprint_attempts), so getSamples never retries
          // and every position on the chain is one iteration; a position that is not DRAW_OK ends the chain.
          uint32_t ent[kSprintPerWave / 64u];                            // thread t: positions t, t + 256, ...
#pragma unroll
          for (uint32_t q = 0; q < kSprintPerWave / 64u; ++q) {
            const uint32_t ps = q * kSprintThreads + tid;
            ent[q] = s_entry[ps];
            const uint32_t nx = ps + ((ent[q] >> 2) & 0xFFFu);
            s_jump[ps] = (ent[q] & 3u) == DRAW_OK ? min(nx, kSprintPos) : kSprintPos;
            s_reach[ps] = ps == 0u ? 1u : 0u;
          }
          if (tid < 64u) { s_jump[kSprintPos + tid] = kSprintPos; s_reach[kSprintPos + tid] = 0u; }
          if (tid == 0u) { s_ctl[0] = 0u; s_ctl[1] = DRAW_OK; }
          __syncthreads();
#pragma unroll 1
          for (uint32_t r = 0; r < 9u; ++r) {                          // 2^9 hops >= the 342 iterations a window can hold
            uint32_t jmp[kSprintPerWave / 64u], on[kSprintPerWave / 64u], jj[kSprintPerWave / 64u];
#pragma unroll
            for (uint32_t q = 0; q < kSprintPerWave / 64u; ++q) {
              const uint32_t ps = q * kSprintThreads + tid;
              jmp[q] = s_jump[ps]; on[q] = s_reach[ps]; jj[q] = s_jump[jmp[q]];
            }
            __syncthreads();
#pragma unroll
            for (uint32_t q = 0; q < kSprintPerWave / 64u; ++q) {
              const uint32_t ps = q * kSprintThreads + tid;
              if (on[q]) s_reach[jmp[q]] = 1u;                         // (entry kSprintPos collects the marks that leave the window)
              s_jump[ps] = jj[q];
            }
            __syncthreads();
          }
          // the iterations in chain order = ascending position: rank by ballots, positions q * 256 + tid in (q, wave, lane) order
          u64 itb[kSprintPerWave / 64u];
#pragma unroll
          for (uint32_t q = 0; q < kSprintPerWave / 64u; ++q) {
            const uint32_t ps = q * kSprintThreads + tid;
            const uint32_t st_e = ent[q] & 3u, nx_true = ps + ((ent[q] >> 2) & 0xFFFu);
            const bool on_chain = s_reach[ps] != 0u;
            const bool is_it = on_chain && st_e == DRAW_OK;
            itb[q] = __ballot(is_it);
            if (l == 0u) s_ctl[4 + q * kSprintWaves + wave] = (uint32_t)__popcll(itb[q]);
            if (is_it && nx_true >= kSprintPos) s_ctl[0] = nx_true;    // the chain leaves the window here (one thread at most) ...
            if (on_chain && st_e != DRAW_OK) { s_ctl[0] = ps; s_ctl[1] = st_e; }   // ... or ends in front of this position (one at most)
          }
          __syncthreads();
          uint32_t cnt = 0;
#pragma unroll
          for (uint32_t q = 0; q < kSprintPerWave / 64u; ++q) {
            uint32_t rank = cnt + (uint32_t)__popcll(itb[q] & ((1ull << l) - 1ull));
#pragma unroll
            for (uint32_t w2 = 0; w2 < kSprintWaves; ++w2) { const uint32_t c = s_ctl[4 + q * kSprintWaves + w2]; cnt += c; if (w2 < wave) rank += c; }
            const uint32_t ps = q * kSprintThreads + tid;
            if (((itb[q] >> l) & 1ull) && rank < kSprintMaxIt) s_it[rank] = ps | ((ps + ((ent[q] >> 2) & 0xFFFu)) << 10);
          }
          cnt = uni(cnt);
          const uint32_t p = uni(s_ctl[0]), end_kind = uni(s_ctl[1]);  // window positions consumed; why the chain ended
          if (end_kind == DRAW_OVERFLOW) { round_ok = false; break; }  // the device copy of the stream ends here
          if (end_kind == DRAW_FAIL || cnt > kSprintMaxIt || (end_kind == DRAW_LONG && p == 0u)) {
            reason = SPRINT_ERROR; err_detail = 3; round_ok = false; break;   // (cannot happen)
          }
          if (end_kind == DRAW_LONG) fresh_ring = true;                // the next window starts at that position with the whole ring ahead
          __syncthreads();
          { const long long t2 = clock64(); t_walk += t2 - tc; tc = t2; }
          // selectWithinDistance (:171-238) of the window's iterations, one per thread (threads 0 .. cnt-1)
          for (uint32_t it = tid; it < cnt; it += kSprintThreads) {
            const uint32_t my_e = s_entry[s_it[it] & 1023u];
            const uint32_t t0 = (my_e >> 14) & 63u, t1 = (my_e >> 20) & 63u, t2 = (my_e >> 26) & 63u;
            const u64 P = s_phys[t0] & s_phys[t1] & s_phys[t2] & valid & finite;   // common physical neighbours (:178-184), D2
            uint32_t c_mine = (uint32_t)__popcll(P) + 3u;                          // + the samples themselves (:185-186)
            if (c_mine > kGateMinimal) {                                           // :203-205
              const u64 F = (P | (1ull << t0) | (1ull << t1) | (1ull << t2)) & deg7;   // :211-213
              if ((uint32_t)__popcll(F) <= kGateMinimal) {
                c_mine = 0u;                                                       // :214-218
              } else {
                bool any = false;                                                  // :221-238
                u64 f = F;
                while (f && !any) {
                  const uint32_t v = (uint32_t)__ffsll((long long)f) - 1u;
                  f &= f - 1ull;
                  any = (uint32_t)__popcll(s_samp[v] & F) > kGateMinimal;
                }
                c_mine = any ? (c_mine | 0x80000000u) : 0u;                        // pending: the clique search decides
              }
            }
            s_c[it] = c_mine;
          }
          __syncthreads();
          { const long long t2 = clock64(); t_eval += t2 - tc; tc = t2; }
          // ransac.h:95-135 in iteration order, 64 iterations at a time: lane i <-> iteration base + i of the window
          for (uint32_t base = 0; base < cnt && !stop; base += 64u) {
            const uint32_t nchunk = min(64u, cnt - base);
            const uint32_t it_rec = s_it[base + l];
            uint32_t cw = l < nchunk ? s_c[base + l] : 0u;
            uint32_t done = 0;                                         // iterations of the chunk already accounted for
            while (done < nchunk && !stop) {
              const u64 pend = __ballot(l >= done && l < nchunk && (cw >> 31));
              const uint32_t lim_l = pend ? (uint32_t)__ffsll((long long)pend) - 1u : nchunk;   // the known prefix ends here
              if (lim_l > done) {
                const bool known = l >= done && l < lim_l;
                const int32_t c_i = known ? (int32_t)cw : INT_MIN;
                const int32_t best_after = max(n_best, wave_incl_scan_max(c_i));
                const uint32_t k_i = best_after > n_best ? s_kceil[min((uint32_t)max(best_after, 0), 64u)] : k_ceil;
                const uint32_t it_after = iterations + (l - done) + 1u;
                const u64 sb = __ballot(known && (it_after > max_it || !(it_after < k_i)));      // :95, :132-134
                const uint32_t last = sb ? (uint32_t)__ffsll((long long)sb) - 1u : lim_l - 1u;   // last iteration that runs
                const int32_t n_new = (int32_t)rdlane((uint32_t)best_after, last);
                if (n_new > n_best) {                                  // strictly better (:115): the first iteration that reaches it
                  const u64 eq = __ballot(known && l <= last && c_i == n_new);
                  const uint32_t bl = (uint32_t)__ffsll((long long)eq) - 1u;
                  best_it = iterations + (bl - done);
                  best_tri = uni(s_entry[rdlane(it_rec, bl) & 1023u]);
                  n_best = n_new;
                  k_ceil = uni(s_kceil[min((uint32_t)max(n_new, 0), 64u)]);
                }
                iterations += last - done + 1u;
                hyps += last - done + 1u;
                if (sb) { stop = true; obj_pos = p_win + (rdlane(it_rec, last) >> 10); break; }
                done = lim_l;
              }
              if (done < nchunk) {                                     // iteration `done` of the chunk needs the clique search
                const uint32_t e_g = uni(s_entry[rdlane(it_rec, done) & 1023u]);
                const uint32_t t0 = (e_g >> 14) & 63u, t1 = (e_g >> 20) & 63u, t2 = (e_g >> 26) & 63u;
                const u64 P = uni64(s_phys[t0] & s_phys[t1] & s_phys[t2] & valid & finite);
                const u64 Fu = (P | (1ull << t0) | (1ull << t1) | (1ull << t2)) & deg7;
                const uint32_t cu = (uint32_t)__popcll(P) + 3u;
                WaveBits F;
#pragma unroll
                for (int q = 0; q < kWPL; ++q) F.w[q] = 0ull;
                if (l == 0u) F.w[0] = Fu;
                EvalArgs E;
                E.job = job; E.iter_samples = nullptr; E.it_begin = 0; E.it_end = 0; E.counts = nullptr; E.gate_m = nullptr;
                E.work = nullptr; E.status = A.status; E.deferred = nullptr; E.stacks = gstack; E.stack_cap = kSprintStackCap;
                E.lds_bytes = kSprintGateLds; E.from_deferred = 0; E.n_deferred = 0; E.adjc_scratch = nullptr; E.dbg = nullptr;
                E.dbg_stride = 0; E.stop_level = 0; E.n_items_dev = nullptr;
                const int32_t res = gate_eval<false, false>(E, F, (uint32_t)__popcll(Fu), iterations, cu, lds_gate, gstack);
                if (res == INT_MIN) { reason = SPRINT_ERROR; err_detail = 2; round_ok = false; stop = true; break; }
                if (l == done) cw = (uint32_t)res;
              }
            }
          }
          t_book += clock64() - tc;
          if (stop) break;
          p_win += p;
        }
        if (!round_ok) {
          if (reason == SPRINT_DONE) reason = SPRINT_NEED_STREAM;
          break;                                                       // nothing of this round has been recorded: it starts again
        }
        if (n_best > 0) {
          // ---- growth (adjacency_ransac.cpp:255-303)
          grew = 1u;
          const long long tg = clock64();
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
          t_grow += clock64() - tg;
          // unique keypoint indices in ascending match order (:306-308); qidx is non-decreasing in the match index
          const bool in = (inl >> l) & 1ull;
          const u64 lower = inl & ((1ull << l) - 1ull);
          const uint32_t pq = (uint32_t)__shfl((int)my_q, (int)(lower ? 63u - (uint32_t)__clzll((long long)lower) : 0u));
          const bool is_new = in && (lower == 0ull || my_q != pq);
          const u64 newb = __ballot(is_new);
          n_kp = (uint32_t)__popcll(newb);
          if (is_new && writer) A.kp_out[kp_used + (uint32_t)__popcll(newb & ((1ull << l) - 1ull))] = my_q;
          float Ro[9], To[3];
          pose_invert(R, T, Ro, To);                                   // :304-305
          if (tid == 0u) {
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
              if (tid == 0u) job.valid[0] = valid;                     // a relaunch of this object starts from here
            }
          } else {
            obj_done = true;                                           // GuessGenerator.cpp:205-206
          }
        } else {
          obj_done = true;                                             // inliers_.empty(): computeModel() == false (:137-138)
        }
      }
      // ---- the round's record
      if (tid == 0u) {
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
  if (tid == 0u) {
    A.out[0] = n_rec; A.out[1] = reason; A.out[2] = n_done_objs;
    A.out[3] = (uint32_t)pos; A.out[4] = (uint32_t)(pos >> 32);
    A.out[5] = __hip_atomic_load(A.status + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) / kSprintWaves;   // every wave ran every gate
    A.out[6] = hyps; A.out[7] = err_detail; A.out[8] = n_windows;
    A.out[9] = (uint32_t)t_ring; A.out[10] = (uint32_t)t_att; A.out[11] = (uint32_t)t_walk; A.out[12] = (uint32_t)t_eval;
    A.out[13] = (uint32_t)t_book; A.out[14] = (uint32_t)t_grow; A.out[15] = (uint32_t)(clock64() - t_begin);
  }
}
