/*
 * oracle/lsh_oracle.c -- TEST INFRASTRUCTURE ONLY (see tod_oracle.h).
 *
 * Definition of the optional LSH-approximate mode of the matcher (todhip_set_lsh, tod_amd/csrc/lsh.hip). The reference's
 * matcher IS an LSH index -- cv::FlannBasedMatcher over cv::flann::LshIndexParams(n_tables, key_size, multi_probe_level),
 * src/detection/DescriptorMatcher.cpp:175-180, parameters conf/detection.ork:32-38 (key_size 16, multi_probe_level 1,
 * n_tables 10) -- but that code is OpenCV's (third party, absent from /root/reference and from this image), so its choice of key
 * bits cannot be reproduced: PARITY UNPINNED. What is restated here is the published scheme of FLANN's lsh_index / lsh_table:
 *   - a table's key of a binary descriptor = key_size of its bits
 *   - a query visits, in every table, the bucket of its own key and the buckets of all keys within multi_probe_level
 *     flipped bits of it (lsh_index.h fill_xor_mask: every xor mask of <= level bits)
 *   - the rows found there are ranked by their exact Hamming distance; k nearest, ties by row
 * and what is this file's own: WHICH bits a table uses -- the first key_size entries of a Fisher-Yates shuffle of 0..255 driven
 * by a 32-bit mix of (table, step) -- since FLANN draws them from rand().
 * Result = top-k by (distance, global row) over the candidate set: independent of any processing order.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "tod_oracle.h"

static uint32_t mix32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }

/* bit positions (0..255, bit b of byte b >> 3 is (byte >> (b & 7)) & 1) of table t's key, least significant key bit first */
void orc_lsh_key_bits(uint32_t table, uint32_t key_size, uint8_t* pos) {
  uint8_t idx[256];
  for (int i = 0; i < 256; ++i) idx[i] = (uint8_t)i;
  for (uint32_t i = 0; i < key_size; ++i) {
    const uint32_t j = i + mix32(table * 0x9E3779B9U + i + 0xABCDU) % (256u - i);
    const uint8_t t = idx[i]; idx[i] = idx[j]; idx[j] = t;
    pos[i] = idx[i];
  }
}

static uint32_t key_of(const uint8_t* d, const uint8_t* pos, uint32_t key_size) {
  uint32_t k = 0;
  for (uint32_t b = 0; b < key_size; ++b) k |= (uint32_t)((d[pos[b] >> 3] >> (pos[b] & 7)) & 1u) << b;
  return k;
}

/* keys[nq][k] = (distance << 32) | row over the LSH candidate set of each query, ascending, UINT64_MAX padded.
 * n_candidates (optional, nq entries): size of each query's candidate set (distinct rows). 32-byte descriptors. */
void orc_lsh_knn_keys(const uint8_t* db, uint64_t n_db, const uint8_t* q, uint32_t nq, uint32_t k, uint32_t n_tables,
                      uint32_t key_size, uint32_t level, uint64_t* keys, uint32_t* n_candidates) {
  uint8_t* pos = (uint8_t*)malloc((size_t)n_tables * key_size + 1);
  uint32_t* rk = (uint32_t*)malloc(((size_t)n_tables * n_db + 1) * sizeof(uint32_t));
  for (uint32_t t = 0; t < n_tables; ++t) {
    orc_lsh_key_bits(t, key_size, pos + (size_t)t * key_size);
    for (uint64_t r = 0; r < n_db; ++r) rk[(size_t)t * n_db + r] = key_of(db + 32 * r, pos + (size_t)t * key_size, key_size);
  }
  uint64_t* best = (uint64_t*)malloc((k + 1) * sizeof(uint64_t));
  uint32_t qk[64];
  for (uint32_t qi = 0; qi < nq; ++qi) {
    const uint8_t* qd = q + 32 * (size_t)qi;
    for (uint32_t t = 0; t < n_tables; ++t) qk[t] = key_of(qd, pos + (size_t)t * key_size, key_size);
    uint32_t have = 0, n_cand = 0;
    for (uint64_t r = 0; r < n_db; ++r) {
      int cand = 0;
      for (uint32_t t = 0; t < n_tables && !cand; ++t) cand = (uint32_t)__builtin_popcount(rk[(size_t)t * n_db + r] ^ qk[t]) <= level;
      if (!cand) continue;
      ++n_cand;
      uint32_t d = 0;
      for (int i = 0; i < 32; ++i) d += (uint32_t)__builtin_popcount((unsigned)(qd[i] ^ db[32 * r + i]));
      const uint64_t key = ((uint64_t)d << 32) | r;
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
    if (n_candidates) n_candidates[qi] = n_cand;
  }
  free(pos); free(rk); free(best);
}
