// vsx_search16_shim.cpp -- the reference-side binding of libvsx: a drop-in replacement for the ONE translation unit
// src/core/align_simd.cpp of torognes/vsearch.  It defines exactly the four symbols that unit exports
// (core/align_simd.hpp:76-108) on top of include/vsx.h, so every caller in the reference (search.cpp:147,
// searchcore.cpp:768,892, cluster.cpp:217,743, allpairs_global.cpp:351,420,422, chimera.cpp) runs unchanged on the GPU.
//
// Built against the reference's own headers where they lie (oracle/Makefile target `ref_shim`, test infrastructure:
// the result, oracle/_ref/vsearch_vsx, is the reference CLI with only this unit swapped; tests/test_gpu_shim.py compares
// its output files with the unmodified CLI's).  This is the COMPATIBILITY binding -- the reference hands over one query x <= 8
// targets per call, far too small to feed a GPU; the throughput path is include/vsx_search.h.
//
// r05, the COMBINER: the reference's worker threads (utils/threads.hpp:85) block in search16 one candidate batch at a time
// (core/searchcore.cpp:757-778).  Their calls are merged "group-commit" style: the first caller that finds no batch in flight becomes
// the leader, takes every request queued so far -- its own and those of the threads that arrived while the previous batch was on the
// GPU -- aligns them as ONE vsx_align_pairs call (one query set, one plan, one fetch) on a context shared by all threads of that
// (device, scoring), and hands the slices back; a thread that finds a leader at work just queues.  A single thread never waits for
// anybody; T threads settle at ~T - 1 requests per round trip.  One aligner context per (device, scoring) instead of one per thread
// and strand (search.cpp:241-249): the start-up cost of the relinked CLI no longer grows with --threads.
#include "vsearch.h"
#include "core/align_simd.hpp"
#include "core/db.hpp"
#include "utils/fatal.hpp"
#include "utils/string_alloc.hpp"

#include "vsx.h"

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <cstdint>
#include <cstdlib>
#include <map>
#include <mutex>
#include <vector>

// One device mirror of a Database per (process, device), shared by every caller on that device: the reference runs
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

static void die(char const * where) { fatal("libvsx: %s", vsx_last_error()); (void) where; }

// one search16 call waiting for its results
struct Request {
  char const * qseq; uint32_t qlen;
  unsigned int n; unsigned int const * seqnos;
  CELL * pscores; unsigned short * paligned, * pmatches, * pmismatches, * pgaps; char * * pcigar;
  Database const * db;
  bool done;
};

// the shared aligner of one (device, scoring): context + the queue of the group commit
struct Combiner;
static void run_batch(Combiner & c, vsx_ctx * ctx, std::vector<Request *> & batch);
struct Combiner {
  int device = 0;
  vsx_scoring sc {};
  // LANES batches may be in flight, each on a context of its own.  Two were measured (profiles/r05/r05h_shim_two_lanes.txt): the calls of
  // 16 threads split into 375 - 412 batches of ~5 instead of 250 of 8 and the relinked CLI took 0.81 - 0.99 s instead of 0.79 -- a round
  // trip is latency, not occupancy, and a second one in flight only halves what each carries.  One lane.
  static constexpr int LANES = 1;
  vsx_ctx * ctx[LANES] = {nullptr};
  bool lane_busy[LANES] = {false};
  std::mutex mu;
  std::condition_variable cv;
  std::vector<Request *> pending;
  uint64_t calls = 0, batches = 0, pairs = 0;           // VSX_SHIM_STATS=1: printed at exit
  void submit(Request & rq)
  {
    std::unique_lock<std::mutex> lk(mu);
    pending.push_back(&rq);
    ++calls;
    while (!rq.done)
      {
        // (this request is either still queued -- then this thread may lead it -- or already part of somebody's batch: then it only waits)
        const bool queued = std::find(pending.begin(), pending.end(), &rq) != pending.end();
        int lane = -1;
        if (queued)
          for (int k = 0; k < LANES; ++k) if (!lane_busy[k]) { lane = k; break; }
        if (lane < 0) { cv.wait(lk); continue; }
        lane_busy[lane] = true;                          // a free context: this thread leads everything queued so far
        std::vector<Request *> batch;
        batch.swap(pending);
        ++batches;
        if (ctx[lane] == nullptr && vsx_create(&ctx[lane], &sc, device) != VSX_OK) die("vsx_create");
        vsx_ctx * const cx = ctx[lane];
        lk.unlock();
        run_batch(*this, cx, batch);
        lk.lock();
        for (Request * r : batch) r->done = true;
        lane_busy[lane] = false;
        cv.notify_all();
      }
  }
};
static std::mutex g_comb_mu;
static std::vector<Combiner *> g_comb;

struct s16info_s {                     // opaque to every other unit (core/searchcore.hpp:151)
  Combiner * comb = nullptr;           // shared, never freed (contexts live as long as the process, like the Database mirror)
  char * qseq = nullptr;               // borrowed between search16_qprep and search16 (align_simd.cpp:1406-1428)
  int qlen = 0;
};

// VSX_DEVICE pins every caller to one GPU; otherwise the (scoring) combiners are dealt round robin over the usable devices
static int pick_device()
{
  static std::atomic<unsigned> next {0};
  if (char const * e = std::getenv("VSX_DEVICE")) return std::atoi(e);
  int const n = vsx_device_count();
  return n > 0 ? (int) (next.fetch_add(1) % (unsigned) n) : 0;
}

static void shim_stats()
{
  for (Combiner * c : g_comb)
    std::fprintf(stderr, "libvsx shim: device %d: %llu search16 calls in %llu batches (%.1f calls, %.1f pairs per batch)\n", c->device,
                 (unsigned long long) c->calls, (unsigned long long) c->batches, c->batches ? (double) c->calls / (double) c->batches : 0.0,
                 c->batches ? (double) c->pairs / (double) c->batches : 0.0);
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
  // every s16info_s of one scoring shares one combiner (the reference creates one per worker thread and strand with the same 15 values)
  std::lock_guard<std::mutex> lk(g_comb_mu);
  static bool const pin = std::getenv("VSX_DEVICE") != nullptr || vsx_device_count() <= 1;
  for (Combiner * c : g_comb)
    if (std::memcmp(&c->sc, &sc, sizeof sc) == 0 && pin) { s->comb = c; return s; }
  // (several devices and no VSX_DEVICE: a combiner per device and scoring, callers dealt round robin)
  int const dev = pick_device();
  for (Combiner * c : g_comb)
    if (std::memcmp(&c->sc, &sc, sizeof sc) == 0 && c->device == dev) { s->comb = c; return s; }
  auto * c = new Combiner();
  c->device = dev;
  c->sc = sc;
  if (vsx_create(&c->ctx[0], &sc, dev) != VSX_OK) die("vsx_create");      // (the second lane's context is created by its first leader)
  if (g_comb.empty() && std::getenv("VSX_SHIM_STATS") != nullptr) std::atexit(shim_stats);
  g_comb.push_back(c);
  s->comb = c;
  return s;
}

auto search16_exit(s16info_s * s) -> void
{
  delete s;                            // the shared context and the Database mirror stay (other threads' handles use them)
}

auto search16_qprep(s16info_s * s, char * qseq, int qlen) -> void
{
  s->qseq = qseq;
  s->qlen = qlen;
}

// The Database is read-only once indexed for searching, but clustering keeps adding to what is visible: re-mirror when the
// fingerprint changed.  One mirror per device, whichever batch asks first builds it.
static vsx_seqset * mirror_db(Combiner * c, Database const & db)
{
  uint64_t const n = db.getsequencecount();
  char const * const f0 = n ? db.getsequence(0) : nullptr;
  char const * const fl = n ? db.getsequence(n - 1) : nullptr;
  uint64_t const ll = n ? db.getsequencelen(n - 1) : 0;
  std::lock_guard<std::mutex> lk(g_mirror_mu);
  DbMirror & m = g_mirror[c->device];
  if (m.set != nullptr && m.db == &db && m.count == n && m.first_seq == f0 && m.last_seq == fl && m.last_len == ll) return m.set;
  if (m.holder == nullptr && vsx_create(&m.holder, &c->sc, c->device) != VSX_OK) die("vsx_create(mirror)");
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

// the leader's part: every request of the batch (same Database: a batch is cut where it changes) as one pair list
static void run_batch(Combiner & c, vsx_ctx * ctx, std::vector<Request *> & batch)
{
  size_t b0 = 0;
  while (b0 < batch.size())
    {
      size_t b1 = b0 + 1;
      while (b1 < batch.size() && batch[b1]->db == batch[b0]->db) ++b1;
      vsx_seqset * const targets = mirror_db(&c, *batch[b0]->db);
      // the queries of the batch as one sequence set (request k = query k), the pairs request after request
      std::vector<uint64_t> qoff(b1 - b0);
      std::vector<uint32_t> qlen(b1 - b0);
      uint64_t qtotal = 0, npairs = 0;
      for (size_t k = b0; k < b1; ++k) { qoff[k - b0] = qtotal; qlen[k - b0] = batch[k]->qlen; qtotal += batch[k]->qlen; npairs += batch[k]->n; }
      std::vector<char> qblob(qtotal + 1);
      for (size_t k = b0; k < b1; ++k) std::memcpy(qblob.data() + qoff[k - b0], batch[k]->qseq, batch[k]->qlen);
      std::vector<uint32_t> qi(npairs), ti(npairs);
      uint64_t at = 0;
      for (size_t k = b0; k < b1; ++k)
        for (unsigned int x = 0; x < batch[k]->n; ++x) { qi[at] = (uint32_t) (k - b0); ti[at] = batch[k]->seqnos[x]; ++at; }
      vsx_seqset * qset = nullptr;
      if (vsx_seqset_create(ctx, &qset, b1 - b0, qblob.data(), qtotal, qoff.data(), qlen.data()) != VSX_OK) die("vsx_seqset_create(queries)");
      vsx_results r;
      if (vsx_align_pairs(ctx, qset, targets, npairs, qi.data(), ti.data(), &r) != VSX_OK) die("vsx_align_pairs");
      at = 0;
      for (size_t k = b0; k < b1; ++k)
        {
          Request & rq = *batch[k];
          for (unsigned int x = 0; x < rq.n; ++x, ++at)
            {
              rq.pscores[x] = r.score[at];
              rq.paligned[x] = r.aligned[at];
              rq.pmatches[x] = r.matches[at];
              rq.pmismatches[x] = r.mismatches[at];
              rq.pgaps[x] = r.gaps[at];
              rq.pcigar[x] = xstrdup(r.cigar_blob + r.cigar_off[at]);          // the caller xfree()s it (align_simd.hpp:99-108)
            }
        }
      vsx_results_free(&r);
      vsx_seqset_destroy(qset);
      { std::lock_guard<std::mutex> lk(c.mu); c.pairs += npairs; }
      b0 = b1;
    }
}

auto search16(s16info_s * s, unsigned int sequences, unsigned int const * seqnos, CELL * pscores,
              unsigned short * paligned, unsigned short * pmatches, unsigned short * pmismatches,
              unsigned short * pgaps, char * * pcigar, struct Database const & db) -> void
{
  if (sequences == 0) return;
  Request rq {s->qseq, (uint32_t) s->qlen, sequences, seqnos, pscores, paligned, pmatches, pmismatches, pgaps, pcigar, &db, false};
  s->comb->submit(rq);
}
