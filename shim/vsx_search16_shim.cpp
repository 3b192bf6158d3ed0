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

#include <atomic>
#include <cstdint>
#include <cstdlib>
#include <map>
#include <mutex>
#include <vector>

// One device mirror of a Database per (process, device), shared by every aligner context on that device: the reference runs
// one s16info_s per worker thread and strand (search.cpp:241-249) over ONE read-only Database (db.hpp:91-99), so T threads
// must not hold T copies of it in HBM.  The mirror belongs to a holder context that lives as long as the process.
struct DbMirror {
  vsx_ctx * holder = nullptr;
  vsx_seqset * set = nullptr;
  // fingerprint of what was mirrored: object, sequence count, first / last sequence address and the last length -- a Database
  // reloaded or grown at the same address changes at least one of them (clustering appends; O(1) per search16 call)
  Database const * db = nullptr;
  uint64_t count = 0;
  char const * first_seq = nullptr, * last_seq = nullptr;
  uint64_t last_len = 0;
};
static std::mutex g_mirror_mu;
static std::map<int, DbMirror> g_mirror;                 // by device

struct s16info_s {                     // opaque to every other unit (core/searchcore.hpp:151)
  vsx_ctx * ctx = nullptr;
  int device = 0;
  vsx_scoring sc {};
  char * qseq = nullptr;               // borrowed between search16_qprep and search16 (align_simd.cpp:1406-1428)
  int qlen = 0;
  vsx_seqset * qset = nullptr;         // the bound query on the device: made by the first search16 after a qprep, reused by the rest
  vsx_seqset * db_set = nullptr;       // the shared mirror this context last used (not owned)
};

static void die(char const * where) { fatal("libvsx: %s", vsx_last_error()); (void) where; }

// VSX_DEVICE pins every context to one GPU; otherwise contexts are dealt round robin over the usable devices (the reference's
// worker threads each create their own contexts)
static int pick_device()
{
  static std::atomic<unsigned> next {0};
  if (char const * e = std::getenv("VSX_DEVICE")) return std::atoi(e);
  int const n = vsx_device_count();
  return n > 0 ? (int) (next.fetch_add(1) % (unsigned) n) : 0;
}

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
  s->device = pick_device();
  s->sc = sc;
  if (vsx_create(&s->ctx, &sc, s->device) != VSX_OK) die("vsx_create");
  return s;
}

auto search16_exit(s16info_s * s) -> void
{
  if (s == nullptr) return;
  vsx_seqset_destroy(s->qset);
  vsx_destroy(s->ctx);                 // the Database mirror stays with its holder context
  delete s;
}

auto search16_qprep(s16info_s * s, char * qseq, int qlen) -> void
{
  s->qseq = qseq;
  s->qlen = qlen;
  vsx_seqset_destroy(s->qset);         // a new query: the device copy is made when it is first aligned
  s->qset = nullptr;
}

// The Database is read-only once indexed for searching, but clustering keeps adding to what is visible: re-mirror when the
// fingerprint changed.  One mirror per device, whichever context asks first builds it.
static vsx_seqset * mirror_db(s16info_s * s, Database const & db)
{
  uint64_t const n = db.getsequencecount();
  char const * const f0 = n ? db.getsequence(0) : nullptr;
  char const * const fl = n ? db.getsequence(n - 1) : nullptr;
  uint64_t const ll = n ? db.getsequencelen(n - 1) : 0;
  std::lock_guard<std::mutex> lk(g_mirror_mu);
  DbMirror & m = g_mirror[s->device];
  if (m.set != nullptr && m.db == &db && m.count == n && m.first_seq == f0 && m.last_seq == fl && m.last_len == ll) return m.set;
  if (m.holder == nullptr && vsx_create(&m.holder, &s->sc, s->device) != VSX_OK) die("vsx_create(mirror)");
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
  // (a superseded mirror is released: the reference's callers finish their search16 calls before the Database changes --
  //  clustering adds centroids between rounds, under its own lock)
  vsx_seqset_destroy(m.set);
  m.set = nullptr;
  if (vsx_seqset_create(m.holder, &m.set, n, blob.data(), total, off.data(), len.data()) != VSX_OK) die("vsx_seqset_create");
  m.db = &db; m.count = n; m.first_seq = f0; m.last_seq = fl; m.last_len = ll;
  return m.set;
}

auto search16(s16info_s * s, unsigned int sequences, unsigned int const * seqnos, CELL * pscores,
              unsigned short * paligned, unsigned short * pmatches, unsigned short * pmismatches,
              unsigned short * pgaps, char * * pcigar, struct Database const & db) -> void
{
  if (sequences == 0) return;
  vsx_seqset * const targets = mirror_db(s, db);
  if (s->qset == nullptr)
    {
      uint64_t const zero = 0;
      uint32_t const ql = (uint32_t) s->qlen;
      if (vsx_seqset_create(s->ctx, &s->qset, 1, s->qseq, ql, &zero, &ql) != VSX_OK) die("vsx_seqset_create(query)");
    }
  std::vector<uint32_t> qi(sequences, 0);
  vsx_results r;
  if (vsx_align_pairs(s->ctx, s->qset, targets, sequences, qi.data(), seqnos, &r) != VSX_OK) die("vsx_align_pairs");
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
}
