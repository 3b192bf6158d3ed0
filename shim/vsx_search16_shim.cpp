// vsx_search16_shim.cpp -- the reference-side binding of libvsx: a drop-in replacement for the ONE translation unit
// src/core/align_simd.cpp of torognes/vsearch.  It defines exactly the four symbols that unit exports
// (core/align_simd.hpp:76-108) on top of include/vsx.h, so every caller in the reference (search.cpp:147,
// searchcore.cpp:768,892, cluster.cpp:217,743, allpairs_global.cpp:351,420,422, chimera.cpp) runs unchanged on the GPU.
//
// Built against the reference's own headers where they lie (oracle/Makefile target `ref_shim`, test infrastructure:
// the result, oracle/_ref/vsearch_vsx, is the reference CLI with only this unit swapped; tests/test_gpu_shim.py compares
// its output files with the unmodified CLI's).  This is the COMPATIBILITY binding -- one query x <= 8 targets per call,
// far too small to feed a GPU; the throughput path is include/vsx_search.h.
#include "vsearch.h"
#include "core/align_simd.hpp"
#include "core/db.hpp"
#include "utils/fatal.hpp"
#include "utils/string_alloc.hpp"

#include "vsx.h"

#include <cstdint>
#include <vector>

struct s16info_s {                     // opaque to every other unit (core/searchcore.hpp:151)
  vsx_ctx * ctx = nullptr;
  char * qseq = nullptr;               // borrowed between search16_qprep and search16 (align_simd.cpp:1406-1428)
  int qlen = 0;
  vsx_seqset * db_set = nullptr;       // device mirror of the Database this context last saw
  Database const * db_seen = nullptr;
  uint64_t db_count = 0;
  uint64_t db_symbols = 0;
};

static void die(char const * where) { fatal("libvsx: %s", vsx_last_error()); (void) where; }

auto search16_init(int64_t score_match, int64_t score_mismatch,
                   int64_t penalty_gap_open_query_left, int64_t penalty_gap_open_target_left,
                   int64_t penalty_gap_open_query_interior, int64_t penalty_gap_open_target_interior,
                   int64_t penalty_gap_open_query_right, int64_t penalty_gap_open_target_right,
                   int64_t penalty_gap_extension_query_left, int64_t penalty_gap_extension_target_left,
                   int64_t penalty_gap_extension_query_interior, int64_t penalty_gap_extension_target_interior,
                   int64_t penalty_gap_extension_query_right, int64_t penalty_gap_extension_target_right,
                   bool score_n_mismatch) -> struct s16info_s *
{
  vsx_scoring sc = {score_match, score_mismatch,
                    penalty_gap_open_query_left, penalty_gap_open_target_left,
                    penalty_gap_open_query_interior, penalty_gap_open_target_interior,
                    penalty_gap_open_query_right, penalty_gap_open_target_right,
                    penalty_gap_extension_query_left, penalty_gap_extension_target_left,
                    penalty_gap_extension_query_interior, penalty_gap_extension_target_interior,
                    penalty_gap_extension_query_right, penalty_gap_extension_target_right,
                    score_n_mismatch ? 1 : 0};
  auto * s = new s16info_s();
  if (vsx_create(&s->ctx, &sc, /*device*/ 0) != VSX_OK) die("vsx_create");
  return s;
}

auto search16_exit(s16info_s * s) -> void
{
  if (s == nullptr) return;
  vsx_seqset_destroy(s->db_set);
  vsx_destroy(s->ctx);
  delete s;
}

auto search16_qprep(s16info_s * s, char * qseq, int qlen) -> void
{
  s->qseq = qseq;
  s->qlen = qlen;
}

// The Database is read-only once indexed for searching, but clustering keeps adding to what is visible: re-mirror when the
// object, the sequence count or the symbol count changed.
static void mirror_db(s16info_s * s, Database const & db)
{
  uint64_t const n = db.getsequencecount();
  if (s->db_seen == &db && s->db_count == n && s->db_set != nullptr) return;
  std::vector<uint64_t> off(n);
  std::vector<uint32_t> len(n);
  uint64_t total = 0;
  for (uint64_t i = 0; i < n; ++i) { off[i] = total; len[i] = (uint32_t) db.getsequencelen(i); total += len[i]; }
  std::vector<char> blob(total + 1);
  for (uint64_t i = 0; i < n; ++i)
    {
      char const * p = db.getsequence(i);
      for (uint32_t k = 0; k < len[i]; ++k) blob[off[i] + k] = p[k];
    }
  vsx_seqset_destroy(s->db_set);
  s->db_set = nullptr;
  if (vsx_seqset_create(s->ctx, &s->db_set, n, blob.data(), total, off.data(), len.data()) != VSX_OK) die("vsx_seqset_create");
  s->db_seen = &db;
  s->db_count = n;
  s->db_symbols = total;
}

auto search16(s16info_s * s, unsigned int sequences, unsigned int const * seqnos, CELL * pscores,
              unsigned short * paligned, unsigned short * pmatches, unsigned short * pmismatches,
              unsigned short * pgaps, char * * pcigar, struct Database const & db) -> void
{
  if (sequences == 0) return;
  mirror_db(s, db);
  vsx_seqset * q = nullptr;
  uint64_t const zero = 0;
  uint32_t const ql = (uint32_t) s->qlen;
  if (vsx_seqset_create(s->ctx, &q, 1, s->qseq, ql, &zero, &ql) != VSX_OK) die("vsx_seqset_create(query)");
  std::vector<uint32_t> qi(sequences, 0);
  vsx_results r;
  if (vsx_align_pairs(s->ctx, q, s->db_set, sequences, qi.data(), seqnos, &r) != VSX_OK) die("vsx_align_pairs");
  for (unsigned int k = 0; k < sequences; ++k)
    {
      pscores[k] = r.score[k];
      paligned[k] = r.aligned[k];
      pmatches[k] = r.matches[k];
      pmismatches[k] = r.mismatches[k];
      pgaps[k] = r.gaps[k];
      pcigar[k] = xstrdup(r.cigar_blob + r.cigar_off[k]);          // the caller xfree()s it (align_simd.hpp:99-108)
    }
  vsx_results_free(&r);
  vsx_seqset_destroy(q);
}
