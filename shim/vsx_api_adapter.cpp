// vsx_api_adapter.cpp -- the reference's LIBRARY API (src/vsearch_api.h) on the throughput path of libvsx: the two batch entry
// points an embedder calls -- search_batch (core/search.hpp:139-150) and cluster_assign_batch (core/cluster.hpp:112-115) --
// keep their reference signatures and result structs (search_result_s, cluster_result_s with char cigar[4096] +
// cigar_truncated) but run on include/vsx_search.h: device k-mer counting, the accept/reject replay, pipelined GPU alignment.
// Unlike shim/vsx_search16_shim.cpp (one query x <= 8 targets per call) this is the fast path behind the reference's API.
//
// Built against the reference's own headers where they lie (oracle/Makefile target `ref_api`, test infrastructure): the
// reference objects are linked with the originals of the overridden functions RENAMED (objcopy --redefine-sym ... vsxref_*),
// so that
//   * the sequential entry points (search_session_single, cluster_assign_single) stay the reference's own code, and the
//     reference's api_examples -- which assert batch == sequential field by field (api_examples/example_search.cc:131-240,
//     example_cluster.cc:121-219) -- become a check of the GPU path against the reference inside the reference's own test;
//   * a configuration this path does not cover is forwarded to the renamed original instead of being approximated.
//
// Contract differences an embedder must know (also in INTEGRATION.md):
//   * cluster_assign_batch clusters the WHOLE database on its first call (the result of greedy clustering does not depend on
//     the batch boundaries -- that is what the reference's intra-batch fix-up guarantees) and serves every later range from
//     that result; the caller's Dbindex is not grown.  A session that started with cluster_assign_single stays on the
//     reference's code for its whole life.
//   * opt_strand with clustering takes the reference's code (--hardmask is covered since r06).  maxaccepts / maxrejects == 0 are NOT "unlimited" in the
//     library (only the CLI rewrites them, search.cpp:521-529): search_onequery's loop condition accepts < maxaccepts /
//     rejects < maxrejects (searchcore.cpp:915-918) is false at once, so search_batch reports no hit for any query -- answered
//     here directly; clustering with such a configuration takes the reference's code.
//   * a RUNTIME failure of the fast path (no device, out of memory, a libvsx error) is logged and the call is answered by the
//     reference's code; nothing aborts the embedding process.
//   * devices: VSX_DEVICES=0,1,2,... (one database replica per device, queries sharded over them: include/vsx_search.h multi-device
//     form), else VSX_DEVICE=n, else device 0.  Clustering runs on the first device.
//   * the Database must not be modified while a session uses it; a cheap best-effort fingerprint (count, sampled sequence and
//     header bytes; lengths and abundances of all up to 65 536 sequences) catches in-place changes between calls (dust_all /
//     hardmask_all after the first batch) and rebuilds the searcher.  VSX_API_FINGERPRINT=full hashes every byte on every call;
//     vsx_api_invalidate() (extern "C") drops the replica explicitly.
#include "vsearch_api.h"

#include "vsx.h"
#include "vsx_search.h"

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

// the reference's own definitions, renamed at link time
extern "C" {
void vsxref_search_batch(struct Parameters const &, struct Dbindex const &, struct Database const &, const char **, const char **,
                         const int *, const int64_t *, int, struct search_result_s *, int, int *);
struct cluster_session_s * vsxref_cluster_session_alloc();
void vsxref_cluster_session_free(struct cluster_session_s *);
void vsxref_cluster_session_init(struct cluster_session_s *, struct Parameters const &, struct Dbindex &, struct Database const &);
void vsxref_cluster_assign_single(struct cluster_session_s *, int, struct cluster_result_s *);
void vsxref_cluster_assign_batch(struct cluster_session_s *, int, int, struct cluster_result_s *);
void vsxref_cluster_session_cleanup(struct cluster_session_s *);
}

namespace {

// a failure of the fast path is reported and the caller falls back to the reference's code (never abort the embedder)
bool complain(char const * where)
{
  std::fprintf(stderr, "libvsx adapter: %s failed: %s -- answering with the reference's code\n", where, vsx_last_error());
  return false;
}

// VSX_DEVICES=0,1,2 | VSX_DEVICE=n | 0
std::vector<int32_t> wanted_devices()
{
  std::vector<int32_t> d;
  if (char const * list = std::getenv("VSX_DEVICES"))
    {
      char const * p = list;
      while (*p)
        {
          char * end = nullptr;
          long const v = std::strtol(p, &end, 10);
          if (end == p) break;
          d.push_back((int32_t) v);
          p = (*end == ',') ? end + 1 : end;
        }
    }
  if (d.empty()) d.push_back(std::getenv("VSX_DEVICE") ? (int32_t) std::atoi(std::getenv("VSX_DEVICE")) : 0);
  return d;
}

// VSX_ADAPTER_TRACE=1: say on stderr which code answered (tests assert that the fast path really ran)
void trace(char const * what, long n)
{
  static bool const on = std::getenv("VSX_ADAPTER_TRACE") != nullptr;
  if (on) std::fprintf(stderr, "libvsx adapter: %s (%ld)\n", what, n);
}

vsx_scoring scoring_of(struct Parameters const & p)           // the arguments of search16_init, core/search.cpp:147-161
{
  return vsx_scoring {p.opt_match, p.opt_mismatch,
                      p.opt_gap_open_query_left, p.opt_gap_open_target_left,
                      p.opt_gap_open_query_interior, p.opt_gap_open_target_interior,
                      p.opt_gap_open_query_right, p.opt_gap_open_target_right,
                      p.opt_gap_extension_query_left, p.opt_gap_extension_target_left,
                      p.opt_gap_extension_query_interior, p.opt_gap_extension_target_interior,
                      p.opt_gap_extension_query_right, p.opt_gap_extension_target_right,
                      p.opt_n_mismatch ? 1 : 0};
}

int mask_mode(Masking m) { return m == Masking::none ? 0 : (m == Masking::soft ? 1 : 2); }

// Parameters after vsearch_apply_defaults_fixups (src/vsearch.cc:186-278) -> the fields this path reads
vsx_search_opts opts_of(struct Parameters const & p, bool clustering)
{
  vsx_search_opts o;
  vsx_search_opts_default(&o);
  o.id = p.opt_id;
  o.weak_id = p.opt_weak_id;
  o.maxaccepts = p.opt_maxaccepts;
  o.maxrejects = p.opt_maxrejects;
  o.wordlength = p.opt_wordlength;
  o.minwordmatches = p.opt_minwordmatches;
  o.iddef = (int32_t) p.opt_iddef;
  // the Database handed over is ALREADY masked by the caller (dust_all / hardmask_all before indexing, usearch_global.cpp:
  // 577-583; api_examples): any mode but "none" means "its lower case is masked".  Queries arrive raw.
  if (clustering) o.soft_mask = p.opt_qmask == Masking::none ? 0 : 1;      // clustering masks and indexes by --qmask (cluster.cpp:1192-1212)
  else
    {
      o.soft_mask = p.opt_dbmask == Masking::none ? 0 : 1;
      o.qmask = 1 + mask_mode(p.opt_qmask);
    }
  o.maxsubs = p.opt_maxsubs; o.maxgaps = p.opt_maxgaps; o.mincols = p.opt_mincols; o.maxdiffs = p.opt_maxdiffs;
  o.query_cov = p.opt_query_cov; o.target_cov = p.opt_target_cov; o.maxid = p.opt_maxid; o.mid = p.opt_mid;
  o.leftjust = (int32_t) p.opt_leftjust; o.rightjust = (int32_t) p.opt_rightjust;
  o.minqt = p.opt_minqt; o.maxqt = p.opt_maxqt; o.minsl = p.opt_minsl; o.maxsl = p.opt_maxsl;
  o.idprefix = p.opt_idprefix; o.idsuffix = p.opt_idsuffix;
  o.selfid = (int32_t) p.opt_selfid;
  o.self = (int32_t) p.opt_self;
  o.threads = (int32_t) p.opt_threads;
  o.strand_both = p.opt_strand ? 1u : 0u;
  o.maxqsize = p.opt_maxqsize; o.mintsize = p.opt_mintsize;
  o.minsizeratio = p.opt_minsizeratio; o.maxsizeratio = p.opt_maxsizeratio;
  o.sizeorder = p.opt_sizeorder ? 1 : 0;
  o.cluster_unoise = p.opt_cluster_unoise != nullptr ? 1 : 0;
  o.unoise_alpha = p.opt_unoise_alpha;
  o.hardmask = p.opt_hardmask ? 2 : 0;               // bit 1: the queries (search.cpp:294-303); the Database was masked by the caller
  bool const inf[12] = {p.opt_gap_open_query_left_infinite, p.opt_gap_open_target_left_infinite,
                        p.opt_gap_open_query_interior_infinite, p.opt_gap_open_target_interior_infinite,
                        p.opt_gap_open_query_right_infinite, p.opt_gap_open_target_right_infinite,
                        p.opt_gap_extension_query_left_infinite, p.opt_gap_extension_target_left_infinite,
                        p.opt_gap_extension_query_interior_infinite, p.opt_gap_extension_target_interior_infinite,
                        p.opt_gap_extension_query_right_infinite, p.opt_gap_extension_target_right_infinite};
  for (int k = 0; k < 12; ++k) if (inf[k]) o.gap_infinite |= 1u << k;
  return o;
}

bool covered(struct Parameters const & p, bool clustering)
{
  // (r06: --hardmask is covered -- the caller's Database arrives hard-masked already, the queries are hard-masked here: vsx_search_opts::hardmask = 2)
  if (clustering && (p.opt_maxaccepts == 0 || p.opt_maxrejects == 0)) return false;
  if (clustering && p.opt_strand) return false;
  return true;
}

// A searcher over the caller's Database, rebuilt when the Database or the configuration changed
struct Fast {
  vsx_multi_searcher * M = nullptr;             // one replica per device; replica 0 serves clustering
  struct Database const * db = nullptr;
  uint64_t count = 0, mark = 0;
  vsx_search_opts o {};
  vsx_scoring sc {};
  void drop() { vsx_multi_searcher_destroy(M); M = nullptr; }
  vsx_searcher * first() { return vsx_multi_searcher_replica(M, 0); }
  // Best-effort detection of an in-place change of the Database between calls (the reference has no generation counter to ask):
  // the count, and length + header + sequence bytes of a spread of sequences (256; 1 024 above 65 536 sequences) plus the last
  // one.  The lengths and abundances of ALL sequences are folded in as well -- up to 65 536 sequences always, above that whenever the
  // call's own work dwarfs the walk (`work` >= n / 64 queries: a search of q queries costs O(q n) counter increments, the walk O(n);
  // clustering always) -- so only a SMALL batch against a multi-million-sequence database skips it (no O(n) per tiny call;
  // an in-place change of one unsampled length / abundance must not go unseen by every large call).  dust_all /
  // hardmask_all rewrite most sequences and are caught by the sample; an edit of the BYTES of one unsampled sequence is not -- an
  // embedder that edits in place calls vsx_api_invalidate() (below), or sets VSX_API_FINGERPRINT=full (every byte hashed on every call).
  static int fingerprint_mode()
  {
    static int const mode = [] {
      char const * e = std::getenv("VSX_API_FINGERPRINT");
      if (e == nullptr || std::strcmp(e, "sample") == 0) return 0;
      if (std::strcmp(e, "full") == 0) return 1;
      std::fprintf(stderr, "vsx adapter: VSX_API_FINGERPRINT=%s not understood (sample | full); using sample\n", e);
      return 0;
    }();
    return mode;
  }
  static uint64_t fingerprint(struct Database const & d, bool walk_all)
  {
    uint64_t const n = d.getsequencecount();
    uint64_t h = 1469598103934665603ull;
    auto mix = [&h](void const * p, size_t bytes) {
      auto const * c = static_cast<unsigned char const *>(p);
      size_t i = 0;
      for (; i + 8 <= bytes; i += 8) { uint64_t w; std::memcpy(&w, c + i, 8); h ^= w; h *= 1099511628211ull; h ^= h >> 29; }
      for (; i < bytes; ++i) { h ^= c[i]; h *= 1099511628211ull; }
    };
    bool const full = fingerprint_mode() == 1;
    mix(&n, sizeof n);
    if (full || n <= 65536 || walk_all)
      {
        uint64_t total = 0, sizes = 0;
        for (uint64_t i = 0; i < n; ++i) { total += d.getsequencelen(i); sizes += (uint64_t) d.getabundance(i); }
        mix(&total, sizeof total); mix(&sizes, sizeof sizes);
      }
    uint64_t const spread = n > 65536 ? 1024 : 256;
    uint64_t const step = full ? 1 : (n > spread ? n / spread : 1);
    for (uint64_t i = 0; i < n; i += step)
      {
        uint64_t const len = d.getsequencelen(i);
        uint64_t const size = (uint64_t) d.getabundance(i);
        mix(&len, sizeof len); mix(&size, sizeof size);
        mix(d.getsequence(i), (size_t) len);
        mix(d.getheader(i), (size_t) d.getheaderlen(i));
      }
    if (n) { uint64_t const len = d.getsequencelen(n - 1); mix(&len, sizeof len); mix(d.getsequence(n - 1), (size_t) len); }
    return h;
  }
  bool walked = false;              // whether `mark` includes the totals over all sequences
  bool ensure(struct Parameters const & p, struct Database const & d, bool clustering, uint64_t work /* queries of this call */)
  {
    uint64_t const n = d.getsequencecount();
    vsx_search_opts const want = opts_of(p, clustering);
    vsx_scoring const wsc = scoring_of(p);
    bool const walk_all = clustering || work >= n / 64;
    // a mark is comparable only with a mark of the same kind: a searcher built by a small call is re-marked (not rebuilt) by the
    // first large one, whose totals then guard the later large calls
    uint64_t fp = fingerprint(d, walk_all);
    if (M != nullptr && db == &d && count == n && walked != walk_all && n > 65536 && fingerprint_mode() == 0)
      {
        uint64_t const same_kind = fingerprint(d, walked);
        if (same_kind == mark) { mark = fp; walked = walk_all; }
      }
    if (M != nullptr && db == &d && count == n && mark == fp &&
        std::memcmp(&want, &o, sizeof o) == 0 && std::memcmp(&wsc, &sc, sizeof sc) == 0)
      return true;
    drop();
    std::vector<uint64_t> off(n), size(n);
    std::vector<uint32_t> len(n);
    std::vector<char const *> label(n);
    uint64_t total = 0;
    for (uint64_t i = 0; i < n; ++i) { off[i] = total; len[i] = (uint32_t) d.getsequencelen(i); total += len[i]; }
    std::string blob(total, '\0');
    for (uint64_t i = 0; i < n; ++i)
      {
        std::memcpy(&blob[off[i]], d.getsequence(i), len[i]);
        size[i] = d.getabundance(i);
        label[i] = d.getheader(i);
      }
    vsx_seq_meta const meta = {size.data(), label.data()};
    std::vector<int32_t> dev = wanted_devices();
    if (clustering) dev.resize(1);               // cluster_* does not shard (sequential centroid dependency)
    if (vsx_multi_searcher_create(&M, &wsc, dev.data(), (int32_t) dev.size(), &want, n, blob.data(), total, off.data(), len.data(), &meta) != VSX_OK)
      { M = nullptr; return complain("vsx_multi_searcher_create"); }
    db = &d; count = n; mark = fp; walked = walk_all; o = want; sc = wsc;
    return true;
  }
  ~Fast() { drop(); }
};

Fast g_search;            // search_batch is not re-entrant in the reference either (core/search.hpp:128)

}  // namespace

// for an embedder that edits the Database in place between batches: the next call rebuilds the device replica
extern "C" void vsx_api_invalidate(void) { g_search.drop(); }


auto search_batch(struct Parameters const & parameters, struct Dbindex const & dbindex, struct Database const & db,
                  const char ** query_seqs, const char ** query_heads, const int * query_lens, const int64_t * query_sizes,
                  int query_count, struct search_result_s * results, int max_results_per_query, int * result_counts) -> void
{
  if (!covered(parameters, false))
    {
      trace("search_batch -> reference code", query_count);
      vsxref_search_batch(parameters, dbindex, db, query_seqs, query_heads, query_lens, query_sizes, query_count, results,
                          max_results_per_query, result_counts);
      return;
    }
  if (parameters.opt_maxaccepts == 0 || parameters.opt_maxrejects == 0)
    {
      // the library does not rewrite 0 into "unlimited" (search.cpp:521-529): the candidate loop of search_onequery never runs
      trace("search_batch -> no candidates are examined (maxaccepts / maxrejects == 0)", query_count);
      for (int k = 0; k < query_count; ++k) result_counts[k] = 0;
      return;
    }
  auto reference = [&]() {
    trace("search_batch -> reference code (fast path failed)", query_count);
    vsxref_search_batch(parameters, dbindex, db, query_seqs, query_heads, query_lens, query_sizes, query_count, results,
                        max_results_per_query, result_counts);
  };
  if (!g_search.ensure(parameters, db, false, (uint64_t) query_count)) { reference(); return; }
  uint64_t const n = (uint64_t) query_count;
  std::vector<uint64_t> off(n), size(n);
  std::vector<uint32_t> len(n);
  uint64_t total = 0;
  for (uint64_t k = 0; k < n; ++k) { off[k] = total; len[k] = (uint32_t) query_lens[k]; total += len[k]; }
  std::string blob(total, '\0');
  for (uint64_t k = 0; k < n; ++k)
    {
      std::memcpy(&blob[off[k]], query_seqs[k], len[k]);
      size[k] = (uint64_t) query_sizes[k];
    }
  vsx_seq_meta const qmeta = {size.data(), query_heads};
  vsx_hits H;
  if (vsx_multi_search_batch(g_search.M, n, blob.data(), total, off.data(), len.data(), &qmeta, &H) != VSX_OK)
    {
      complain("vsx_multi_search_batch");
      reference();
      return;
    }
  trace("search_batch -> vsx_multi_search_batch", query_count);          // only once the fast path HAS answered (the tests key on it)
  // search_joinhits order (accepted / weak hits of both strands, best first), the first max_results of it (search.cpp:463-488)
  for (uint64_t k = 0; k < n; ++k)
    {
      int count = 0;
      for (uint64_t x = H.first[k]; x < H.first[k + 1] && count < max_results_per_query; ++x, ++count)
        {
          vsx_hit const & h = H.hit[x];
          struct search_result_s & r = results[k * (uint64_t) max_results_per_query + (uint64_t) count];
          r.target = (int) h.target;
          r.id = h.id;
          r.matches = h.matches;
          r.mismatches = h.mismatches;
          r.gaps = h.nwgaps;
          r.alignment_length = h.nwalignmentlength;
          r.query_length = query_lens[k];
          r.target_length = (int) db.getsequencelen(h.target);
          r.accepted = h.accepted != 0;
          r.strand = h.strand;
        }
      result_counts[k] = count;
    }
  vsx_hits_free(&H);
}


// ---- clustering: the opaque session of the API is ours; it owns a reference session for the sequential entry point
namespace {

struct FastCluster {
  struct cluster_session_s * ref = nullptr;
  struct Parameters const * parameters = nullptr;
  struct Database const * db = nullptr;
  enum { undecided, reference_code, fast_path } mode = undecided;
  Fast fast;
  vsx_cluster_out out {};                           // fast path: the whole database's clustering (compact; results are
  bool have = false;                                //  materialised per request -- a cluster_result_s is 5 KB)
  void forget() { if (have) vsx_cluster_out_free(&out); have = false; }
};

FastCluster * mine(struct cluster_session_s * cs) { return reinterpret_cast<FastCluster *>(cs); }

bool run_fast(FastCluster & c)
{
  if (!c.fast.ensure(*c.parameters, *c.db, true, 0)) return false;
  if (vsx_cluster_fast(c.fast.first(), 0, &c.out) != VSX_OK) return complain("vsx_cluster_fast");
  c.have = true;
  return true;
}

// cluster_assign_single's result record (core/cluster.cpp:1719-1750) for sequence s
void materialise(FastCluster const & c, uint64_t s, struct cluster_result_s & r)
{
  struct Database const & db = *c.db;
  vsx_cluster_out const & out = c.out;
  if (s >= out.n) { std::fprintf(stderr, "libvsx adapter: sequence %llu is not in the database\n", (unsigned long long) s); std::abort(); }
  r = cluster_result_s {};
  r.cluster_id = (int) out.clusterno[s];
  bool const centroid = out.hits.first[s] == out.hits.first[s + 1];
  uint64_t const cen = centroid ? s : out.hits.hit[out.hits.first[s]].target;
  r.is_centroid = centroid;
  r.centroid_seqno = (int) cen;
  std::snprintf(r.centroid_label, sizeof r.centroid_label, "%.*s", (int) db.getheaderlen(cen), db.getheader(cen));
  if (centroid) { r.identity = 100.0; return; }
  vsx_hit const & h = out.hits.hit[out.hits.first[s]];
  r.identity = h.id;
  int const nch = std::snprintf(r.cigar, sizeof r.cigar, "%s", out.hits.cigar_blob + h.cigar_off);
  r.cigar_truncated = nch >= (int) sizeof r.cigar;
}

}  // namespace

auto cluster_session_alloc() -> struct cluster_session_s *
{
  auto * c = new FastCluster;
  c->ref = vsxref_cluster_session_alloc();
  return reinterpret_cast<struct cluster_session_s *>(c);
}

auto cluster_session_free(struct cluster_session_s * cs) -> void
{
  if (cs == nullptr) return;
  vsxref_cluster_session_free(mine(cs)->ref);
  mine(cs)->forget();
  delete mine(cs);
}

auto cluster_session_init(struct cluster_session_s * cs, struct Parameters const & parameters, struct Dbindex & dbindex,
                          struct Database const & db) -> void
{
  FastCluster & c = *mine(cs);
  c.parameters = &parameters;
  c.db = &db;
  c.mode = FastCluster::undecided;
  c.forget();
  vsxref_cluster_session_init(c.ref, parameters, dbindex, db);
}

auto cluster_assign_single(struct cluster_session_s * cs, int seqno, struct cluster_result_s * result) -> void
{
  FastCluster & c = *mine(cs);
  if (c.mode == FastCluster::undecided) c.mode = FastCluster::reference_code;
  if (c.mode == FastCluster::reference_code) { vsxref_cluster_assign_single(c.ref, seqno, result); return; }
  materialise(c, (uint64_t) seqno, *result);
}

auto cluster_assign_batch(struct cluster_session_s * cs, int start_seqno, int count, struct cluster_result_s * results) -> void
{
  FastCluster & c = *mine(cs);
  if (c.mode == FastCluster::undecided) c.mode = covered(*c.parameters, true) ? FastCluster::fast_path : FastCluster::reference_code;
  if (c.mode == FastCluster::reference_code) { vsxref_cluster_assign_batch(c.ref, start_seqno, count, results); return; }
  if (!c.have)
    {
      if (!run_fast(c))
        {
          trace("cluster_assign_batch -> reference code (fast path failed)", (long) c.db->getsequencecount());
          // nothing has been answered from the fast path yet (this is the session's first batch): the reference's session, which
          // cluster_session_init prepared alongside, takes over for the rest of the session
          c.mode = FastCluster::reference_code;
          vsxref_cluster_assign_batch(c.ref, start_seqno, count, results);
          return;
        }
      trace("cluster_assign_batch -> vsx_cluster_fast", (long) c.db->getsequencecount());          // the fast path has answered
    }
  for (int k = 0; k < count; ++k) materialise(c, (uint64_t) (start_seqno + k), results[k]);
}

auto cluster_session_cleanup(struct cluster_session_s * cs) -> void
{
  FastCluster & c = *mine(cs);
  vsxref_cluster_session_cleanup(c.ref);
  c.fast.drop();
  c.forget();
}
