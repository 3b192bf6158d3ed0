/*
  oracle/ref_driver.cc -- TEST INFRASTRUCTURE ONLY (never linked into the product).

  A thin C-ABI driver around the *real* reference aligner, compiled from the
  reference's own translation units where they lie under /root/reference/src
  (see oracle/Makefile: core/align_simd.cpp, core/linmemalign.cpp,
  utils/maps.cpp, utils/string_alloc.cpp, utils/fatal.cpp, os/posix/system.cc).
  No reference source is copied into this repository; this file only *calls*
  the reference's public functions:

    search16_init / search16_exit / search16_qprep / search16
        (core/align_simd.hpp:76-108)
    LinearMemoryAligner::align / alignstats   (core/linmemalign.hpp:167-179)

  The reference's `Database` (core/db.hpp:100-225) is only declared here; the
  few members the driver needs (init/add/clear/dtor) are defined below by us so
  that core/db.cpp (which drags in the FASTA/FASTQ/UDB readers) need not be
  compiled.  getsequence()/getsequencelen() are inline in the reference header.

  Output goes to oracle/_ref/libvsref.so (git-ignored, travels with gpurun).
*/

#include "vsearch.h"
#include "core/align_simd.hpp"
#include "core/linmemalign.hpp"
#include "core/db.hpp"
#include "utils/logfile.hpp"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <cstring>
#include <limits>
#include <thread>
#include <vector>

/* ---- minimal definitions of the Database members we use (ours, not copied) ---- */

Database::~Database() { clear(); }

auto Database::clear() -> void
{
  data_.clear();
  seqindex_.clear();
  sequences = 0;
  nucleotides = 0;
  longest = 0;
  shortest = 0;
  longestheader = 0;
}

auto Database::init() -> void
{
  clear();
  fastq_format = false;
}

auto Database::add(bool const, char const * header, char const * sequence,
                   char const *, std::size_t const headerlength,
                   std::size_t const sequencelength, int64_t const abundance) -> void
{
  seqinfo_t rec;
  rec.header_p = data_.size();
  data_.insert(data_.end(), header, header + headerlength);
  data_.push_back('\0');
  rec.seq_p = data_.size();
  data_.insert(data_.end(), sequence, sequence + sequencelength);
  data_.push_back('\0');
  rec.qual_p = data_.size();
  rec.headerlen = static_cast<unsigned int>(headerlength);
  rec.seqlen = static_cast<unsigned int>(sequencelength);
  rec.size = static_cast<uint64_t>(abundance);
  seqindex_.push_back(rec);
  ++sequences;
  nucleotides += sequencelength;
  longest = std::max<uint64_t>(longest, sequencelength);
}

/* fatal() in utils/fatal.cpp consults the --log handle; there is none here */
namespace log_file
{
  auto handle() noexcept -> std::FILE * { return nullptr; }
  auto set_handle(std::FILE *) noexcept -> void {}
}

/* ------------------------------------------------------------------------- */

namespace {

struct RefCtx
{
  int64_t P[14];
  bool nmm;
  s16info_s * s;
};

auto make_s16(const int64_t * P, bool nmm) -> s16info_s *
{
  return search16_init(P[0], P[1], P[2], P[3], P[4], P[5], P[6], P[7],
                       P[8], P[9], P[10], P[11], P[12], P[13], nmm);
}

auto make_scoring(const int64_t * P, bool nmm) -> Scoring
{
  Scoring sc;
  sc.match = P[0];
  sc.mismatch = P[1];
  sc.gap_open_query_left = P[2];
  sc.gap_open_target_left = P[3];
  sc.gap_open_query_interior = P[4];
  sc.gap_open_target_interior = P[5];
  sc.gap_open_query_right = P[6];
  sc.gap_open_target_right = P[7];
  sc.gap_extension_query_left = P[8];
  sc.gap_extension_target_left = P[9];
  sc.gap_extension_query_interior = P[10];
  sc.gap_extension_target_interior = P[11];
  sc.gap_extension_query_right = P[12];
  sc.gap_extension_target_right = P[13];
  sc.n_mismatch = nmm;
  return sc;
}

}  // namespace

extern "C" {

/* P = (match, mismatch, go_q_l, go_t_l, go_q_i, go_t_i, go_q_r, go_t_r,
        ge_q_l, ge_t_l, ge_q_i, ge_t_i, ge_q_r, ge_t_r), post-fixup values --
   exactly the argument order of search16_init (core/align_simd.hpp:76-90). */
void * vsref_create(const int64_t * P, int n_mismatch)
{
  auto * c = new RefCtx;
  std::memcpy(c->P, P, sizeof(c->P));
  c->nmm = (n_mismatch != 0);
  c->s = make_s16(P, c->nmm);
  return c;
}

void vsref_destroy(void * ctx)
{
  auto * c = static_cast<RefCtx *>(ctx);
  search16_exit(c->s);
  delete c;
}

void vsref_free(void * p) { std::free(p); }

/* One query against n targets in ONE search16 call (the reference's own batch
   shape).  cigars[i] receives a malloc'ed NUL-terminated string (vsref_free). */
void vsref_search16(void * ctx, const char * q, int qlen, int n,
                    const char * const * tseqs, const int * tlens,
                    int16_t * scores, uint16_t * aligned, uint16_t * matches,
                    uint16_t * mismatches, uint16_t * gaps, char ** cigars)
{
  auto * c = static_cast<RefCtx *>(ctx);
  Database db;
  db.init();
  std::vector<unsigned int> seqnos(static_cast<size_t>(n));
  for (int i = 0; i < n; ++i)
    {
      /* add() copies [seq, seq+len) and terminates it itself */
      std::vector<char> tmp(tseqs[i], tseqs[i] + tlens[i]);
      tmp.push_back('\0');
      db.add(false, "t", tmp.data(), nullptr, 1, static_cast<size_t>(tlens[i]), 1);
      seqnos[static_cast<size_t>(i)] = static_cast<unsigned int>(i);
    }
  std::vector<char> qbuf(q, q + qlen);
  qbuf.push_back('\0');
  search16_qprep(c->s, qbuf.data(), qlen);
  search16(c->s, static_cast<unsigned int>(n), seqnos.data(), scores, aligned,
           matches, mismatches, gaps, cigars, db);
}

/* The scalar fallback the callers use on the SHRT_MAX sentinel
   (core/searchcore.cpp:806-832): LinearMemoryAligner::align + alignstats. */
void vsref_lma(void * ctx, const char * q, int qlen, const char * t, int tlen,
               int64_t * score, int64_t * alnlen, int64_t * matches,
               int64_t * mismatches, int64_t * gaps, char ** cigar)
{
  auto * c = static_cast<RefCtx *>(ctx);
  Scoring sc = make_scoring(c->P, c->nmm);
  LinearMemoryAligner lma(sc);
  std::vector<char> qb(q, q + qlen); qb.push_back('\0');
  std::vector<char> tb(t, t + tlen); tb.push_back('\0');
  char * cg = lma.align(qb.data(), tb.data(), qlen, tlen);
  char * dup = static_cast<char *>(std::malloc(std::strlen(cg) + 1));
  std::strcpy(dup, cg);
  lma.alignstats(dup, qb.data(), tb.data(), score, alnlen, matches, mismatches, gaps);
  *cigar = dup;
}

/*
  CPU baseline: time the reference SSE2 aligner (search16 incl. its scalar
  backtrack) on a pair list grouped by query, `threads` std::threads, each with
  its own s16info_s (the reference's threading contract, core/search.cpp:128).
  Thread creation, search16_init and one warm-up group per thread are outside the timed region.

  Sequences are given as one ASCII blob + offsets/lengths.  Groups: query g has
  targets tidx[goff[g] .. goff[g+1]).  Returns wall seconds; *cells receives
  sum(Q*D) over all pairs, *checksum a simple sum of scores (defeats DCE),
  *digest an order-independent hash of EVERY field of every pair -- pair number,
  score, aligned, matches, mismatches, gaps and the CIGAR text -- that the caller
  compares with vsref_digest_results() over the GPU path's results for the same pairs.
*/
static inline uint64_t pair_digest(uint64_t pair, int16_t score, uint16_t aligned, uint16_t matches, uint16_t mismatches,
                                   uint16_t gaps, char const * cigar)
{
  uint64_t h = 14695981039346656037ull;
  auto mix = [&h](void const * p, size_t n) {
    auto const * c = static_cast<unsigned char const *>(p);
    for (size_t i = 0; i < n; ++i) { h ^= c[i]; h *= 1099511628211ull; }
  };
  mix(&pair, 8); mix(&score, 2); mix(&aligned, 2); mix(&matches, 2); mix(&mismatches, 2); mix(&gaps, 2);
  if (cigar != nullptr) mix(cigar, std::strlen(cigar));
  return h;
}

/* the same digest over result arrays as libvsx hands them out (pairs first_pair .. first_pair + n - 1; cigar k at blob + off[k]) */
uint64_t vsref_digest_results(uint64_t n, uint64_t first_pair, const int16_t * score, const uint16_t * aligned, const uint16_t * matches,
                              const uint16_t * mismatches, const uint16_t * gaps, const char * blob, const uint64_t * off)
{
  uint64_t d = 0;
  for (uint64_t k = 0; k < n; ++k)
    d += pair_digest(first_pair + k, score[k], aligned[k], matches[k], mismatches[k], gaps[k], blob + off[k]);
  return d;
}

double vsref_time_groups(const int64_t * P, int n_mismatch,
                         const char * qblob, const uint64_t * qoff, const uint32_t * qlen,
                         const char * tblob, const uint64_t * toff, const uint32_t * tlen,
                         uint32_t n_targets_total,
                         uint32_t n_groups, const uint32_t * gq, const uint64_t * goff,
                         const uint32_t * tidx, int threads,
                         uint64_t * cells, int64_t * checksum, uint64_t * digest)
{
  /* one shared read-only Database holding every target (as the reference does) */
  Database db;
  db.init();
  for (uint32_t i = 0; i < n_targets_total; ++i)
    {
      std::vector<char> tmp(tblob + toff[i], tblob + toff[i] + tlen[i]);
      tmp.push_back('\0');
      db.add(false, "t", tmp.data(), nullptr, 1, tlen[i], 1);
    }
  std::atomic<uint32_t> next {0};
  std::atomic<uint64_t> tot_cells {0};
  std::atomic<int64_t> tot_sum {0};
  std::atomic<uint64_t> tot_digest {0};
  std::atomic<int> ready {0};
  std::atomic<bool> go {false};
  bool const nmm = (n_mismatch != 0);
  Database const & cdb = db;

  auto run_group = [&](s16info_s * s, uint32_t gi, std::vector<char> & qbuf, std::vector<int16_t> & sc,
                       std::vector<uint16_t> & a, std::vector<uint16_t> & m, std::vector<uint16_t> & mm,
                       std::vector<uint16_t> & g, std::vector<char *> & cg, uint64_t & my_cells, int64_t & my_sum, uint64_t & my_digest) {
    uint32_t const qi = gq[gi];
    uint64_t const b = goff[gi];
    uint64_t const e = goff[gi + 1];
    auto const n = static_cast<unsigned int>(e - b);
    qbuf.assign(qblob + qoff[qi], qblob + qoff[qi] + qlen[qi]);
    qbuf.push_back('\0');
    sc.resize(n); a.resize(n); m.resize(n); mm.resize(n); g.resize(n); cg.resize(n);
    search16_qprep(s, qbuf.data(), static_cast<int>(qlen[qi]));
    search16(s, n, tidx + b, sc.data(), a.data(), m.data(), mm.data(), g.data(), cg.data(), cdb);
    for (unsigned int k = 0; k < n; ++k)
      {
        my_cells += static_cast<uint64_t>(qlen[qi]) * tlen[tidx[b + k]];
        my_sum += sc[k] + a[k] + m[k];
        my_digest += pair_digest(b + k, sc[k], a[k], m[k], mm[k], g[k], cg[k]);
        std::free(cg[k]);
      }
  };

  auto worker = [&]() {
    s16info_s * s = make_s16(P, nmm);
    std::vector<char> qbuf;
    std::vector<int16_t> sc;
    std::vector<uint16_t> a, m, mm, g;
    std::vector<char *> cg;
    uint64_t my_cells = 0, my_digest = 0;
    int64_t my_sum = 0;
    if (n_groups > 0)
      {
        /* untimed warm-up: sizes this thread's direction/cigar buffers (first-touch page faults) */
        uint64_t wc = 0, wd = 0; int64_t ws = 0;
        run_group(s, 0, qbuf, sc, a, m, mm, g, cg, wc, ws, wd);
      }
    ready.fetch_add(1);
    while (!go.load(std::memory_order_acquire)) { std::this_thread::yield(); }
    while (true)
      {
        uint32_t const gi = next.fetch_add(1);
        if (gi >= n_groups) { break; }
        run_group(s, gi, qbuf, sc, a, m, mm, g, cg, my_cells, my_sum, my_digest);
      }
    search16_exit(s);
    tot_cells += my_cells;
    tot_sum += my_sum;
    tot_digest += my_digest;
  };

  std::vector<std::thread> pool;
  for (int t = 0; t < threads; ++t) { pool.emplace_back(worker); }
  while (ready.load() < threads) { std::this_thread::yield(); }
  auto const t0 = std::chrono::steady_clock::now();
  go.store(true, std::memory_order_release);
  for (auto & th : pool) { th.join(); }
  auto const t1 = std::chrono::steady_clock::now();
  *cells = tot_cells.load();
  *checksum = tot_sum.load();
  if (digest != nullptr) { *digest = tot_digest.load(); }
  return std::chrono::duration<double>(t1 - t0).count();
}

}  // extern "C"
