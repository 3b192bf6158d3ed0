// vsx_multi.cpp -- several GPUs behind ONE C handle (include/vsx_search.h, "multi-device form").
//
// The reference is one process with a pool of worker threads that share the database and the index and pull queries off a
// common reader (commands/usearch_global.cpp:500-535, core/search.cpp:397-508 search_batch; allpairs: commands/
// allpairs_global.cpp:394-527).  The MI355X form of that pool: one aligner context + database replica + k-mer index per
// device (the path shards embarrassingly, nothing is exchanged before the final merge), one host thread per device
// driving vsx_search_batch_meta / vsx_allpairs_rows on its share, and a merge on the host in the caller's query order --
// no collective is needed inside one process; the multi-PROCESS form (torch.distributed, RCCL gather) is vsearch_amd/sharding.py.
//
//   usearch_global: contiguous blocks of queries, one per device (equal sizes: the per-query work of a read set is uniform);
//   allpairs_global: the rows of the triangular pair space dealt out boustrophedon (row r has n - 1 - r pairs; dealing
//                    0 1 .. D-1 D-1 .. 1 0 balances pairs and cells to < 0.5 % at 50 000 sequences), every device runs its
//                    rows against its replica;
//   cluster_*:      does not shard (sequential centroid dependency): use one searcher.
//
// The same device may be listed more than once (two replicas on one GPU): that is how the suite checks this file on a one-GPU box.
#include "../../include/vsx_search.h"

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <thread>
#include <vector>

extern "C" void vsx_internal_set_error(const char * msg);

struct vsx_multi_searcher {
  struct Replica {
    int device = 0;
    vsx_ctx * ctx = nullptr;
    vsx_searcher * S = nullptr;
  };
  std::vector<Replica> rep;
  uint64_t n_db = 0;
  ~vsx_multi_searcher()
  {
    for (Replica & r : rep)
      {
        if (r.S) vsx_searcher_destroy(r.S);
        if (r.ctx) vsx_destroy(r.ctx);
      }
  }
};

extern "C" int vsx_internal_usable_cpus(void);

namespace {

int mfail(int code, const std::string & msg) { vsx_internal_set_error(msg.c_str()); return code; }

// run fn(d) for every replica on a thread of its own; the first failure (by device order) is reported
template <typename F>
int on_all(vsx_multi_searcher * m, F fn)
{
  const size_t D = m->rep.size();
  std::vector<int> rc(D, VSX_OK);
  std::vector<std::string> msg(D);
  auto run = [&](size_t d) {
    rc[d] = fn(d);
    if (rc[d] != VSX_OK) msg[d] = vsx_last_error();          // the error text is thread-local
  };
  std::vector<std::thread> th;
  for (size_t d = 1; d < D; ++d) th.emplace_back(run, d);
  run(0);
  for (auto & t : th) t.join();
  for (size_t d = 0; d < D; ++d)
    if (rc[d] != VSX_OK) return mfail(rc[d], "device " + std::to_string(m->rep[d].device) + ": " + msg[d]);
  return VSX_OK;
}

// `part` hit lists -> one, in the order given by (part, local query) per output query
struct Source { uint32_t part; uint64_t local; };
int merge_hits(const std::vector<vsx_hits> & H, const std::vector<Source> & src, const std::vector<uint64_t> & query_shift, vsx_hits * out)
{
  const uint64_t nq = src.size();
  uint64_t n_hits = 0, bytes = 0;
  for (const vsx_hits & h : H) { n_hits += h.n_hits; bytes += h.cigar_bytes; }
  std::memset(out, 0, sizeof *out);
  out->n_queries = nq;
  out->first = (uint64_t *) std::malloc((nq + 1) * sizeof(uint64_t));
  out->hit = (vsx_hit *) std::malloc(std::max<uint64_t>(n_hits, 1) * sizeof(vsx_hit));
  out->cigar_blob = (char *) std::malloc(std::max<uint64_t>(bytes, 1));
  if (!out->first || !out->hit || !out->cigar_blob) { vsx_hits_free(out); return mfail(VSX_ENOMEM, "vsx_multi: out of memory"); }
  // the CIGAR blobs are concatenated as they are: offsets of part p shift by the bytes of the parts before it
  std::vector<uint64_t> blob_shift(H.size(), 0);
  uint64_t at = 0;
  for (size_t p = 0; p < H.size(); ++p)
    {
      blob_shift[p] = at;
      if (H[p].cigar_bytes) std::memcpy(out->cigar_blob + at, H[p].cigar_blob, H[p].cigar_bytes);
      at += H[p].cigar_bytes;
    }
  out->cigar_bytes = at;
  uint64_t w = 0;
  for (uint64_t q = 0; q < nq; ++q)
    {
      out->first[q] = w;
      const vsx_hits & h = H[src[q].part];
      for (uint64_t x = h.first[src[q].local]; x < h.first[src[q].local + 1]; ++x)
        {
          vsx_hit v = h.hit[x];
          v.query = (uint32_t) (v.query + query_shift[src[q].part]);
          v.cigar_off += blob_shift[src[q].part];
          out->hit[w++] = v;
        }
    }
  out->first[nq] = w;
  out->n_hits = w;
  for (const vsx_hits & h : H)
    {
      out->pairs_aligned += h.pairs_aligned; out->cells_aligned += h.cells_aligned; out->sentinel_pairs += h.sentinel_pairs;
      out->stages = std::max(out->stages, h.stages);
      out->seconds_kmer = std::max(out->seconds_kmer, h.seconds_kmer);
      out->seconds_align = std::max(out->seconds_align, h.seconds_align);
      out->seconds_total = std::max(out->seconds_total, h.seconds_total);
    }
  return VSX_OK;
}

}  // namespace

extern "C" {

int vsx_multi_searcher_create(vsx_multi_searcher ** out, const vsx_scoring * scoring, const int32_t * devices, int32_t n_devices,
                              const vsx_search_opts * opts, uint64_t n, const char * blob, uint64_t blob_bytes,
                              const uint64_t * offsets, const uint32_t * lengths, const vsx_seq_meta * meta)
{
  if (!out || !scoring || !opts || n_devices < 1 || !devices) return mfail(VSX_EINVAL, "vsx_multi_searcher_create: null argument");
  *out = nullptr;
  std::unique_ptr<vsx_multi_searcher> m(new vsx_multi_searcher);
  m->rep.resize((size_t) n_devices);
  m->n_db = n;
  for (int32_t d = 0; d < n_devices; ++d) m->rep[(size_t) d].device = devices[d];
  // every replica starts its own host workers (word stage, rank workers, consumers): D replicas with the caller's
  // whole thread budget each oversubscribe the host D times, so the budget is shared out (at least 2 per replica).  Listing one
  // device twice still doubles that device's context memory (three aligner contexts + counting scratch per replica).
  vsx_search_opts ropts = *opts;
  {
    //  the floor of 2 applies to the auto-detected budget only: an explicit `threads` is the caller's limit and is never raised
    const bool explicit_budget = ropts.threads > 0;
    int budget = explicit_budget ? ropts.threads : vsx_internal_usable_cpus();
    if (budget <= 0) budget = 2;
    ropts.threads = std::max(explicit_budget ? 1 : 2, budget / n_devices);
  }
  // every replica uploads and indexes on its own device at the same time
  const int rc = on_all(m.get(), [&](size_t d) -> int {
    vsx_multi_searcher::Replica & r = m->rep[d];
    int e = vsx_create(&r.ctx, scoring, r.device);
    if (e != VSX_OK) return e;
    e = vsx_searcher_create(r.ctx, &r.S, &ropts, n, blob, blob_bytes, offsets, lengths);
    if (e != VSX_OK) return e;
    return meta ? vsx_searcher_set_meta(r.S, meta) : VSX_OK;
  });
  if (rc != VSX_OK) return rc;
  *out = m.release();
  return VSX_OK;
}

void vsx_multi_searcher_destroy(vsx_multi_searcher * m) { delete m; }

int32_t vsx_multi_searcher_devices(const vsx_multi_searcher * m) { return m ? (int32_t) m->rep.size() : 0; }

vsx_searcher * vsx_multi_searcher_replica(vsx_multi_searcher * m, int32_t k)
{
  return (m && k >= 0 && (size_t) k < m->rep.size()) ? m->rep[(size_t) k].S : nullptr;
}

int vsx_multi_search_batch(vsx_multi_searcher * m, uint64_t nq, const char * qblob, uint64_t qbytes, const uint64_t * qoff,
                           const uint32_t * qlen, const vsx_seq_meta * qmeta, vsx_hits * out)
{
  if (!m || !out || (nq && (!qblob || !qoff || !qlen))) return mfail(VSX_EINVAL, "vsx_multi_search_batch: null argument");
  std::memset(out, 0, sizeof *out);
  if (nq > 0xFFFFFFFFull) return mfail(VSX_EINVAL, "vsx_multi_search_batch: more than 2^32 - 1 queries (vsx_hit.query is 32 bits wide)");
  const size_t D = m->rep.size();
  // contiguous blocks: block d = [lo[d], lo[d + 1])
  std::vector<uint64_t> lo(D + 1, 0);
  for (size_t d = 0; d <= D; ++d) lo[d] = nq * d / D;
  std::vector<vsx_hits> H(D);
  for (vsx_hits & h : H) std::memset(&h, 0, sizeof h);
  const int rc = on_all(m, [&](size_t d) -> int {
    const uint64_t a = lo[d], cnt = lo[d + 1] - lo[d];
    // the block's offsets still point into the caller's blob: nothing is copied
    vsx_seq_meta part {nullptr, nullptr};
    if (qmeta) { part.abundance = qmeta->abundance ? qmeta->abundance + a : nullptr; part.label = qmeta->label ? qmeta->label + a : nullptr; }
    return vsx_search_batch_meta(m->rep[d].S, cnt, qblob, qbytes, qoff + a, qlen + a, qmeta ? &part : nullptr, &H[d]);
  });
  int mrc = rc;
  if (rc == VSX_OK)
    {
      std::vector<Source> src(nq);
      std::vector<uint64_t> shift(D);
      for (size_t d = 0; d < D; ++d)
        {
          shift[d] = lo[d];
          for (uint64_t k = lo[d]; k < lo[d + 1]; ++k) src[k] = Source {(uint32_t) d, k - lo[d]};
        }
      mrc = merge_hits(H, src, shift, out);
    }
  for (vsx_hits & h : H) vsx_hits_free(&h);
  return mrc;
}

int vsx_multi_allpairs(vsx_multi_searcher * m, int32_t acceptall, uint64_t first, uint64_t count, vsx_hits * out)
{
  if (!m || !out) return mfail(VSX_EINVAL, "vsx_multi_allpairs: null argument");
  std::memset(out, 0, sizeof *out);
  if (first > m->n_db || count > m->n_db - first) return mfail(VSX_EINVAL, "vsx_multi_allpairs: rows exceed the database");
  const size_t D = m->rep.size();
  // boustrophedon deal of the rows first .. first + count - 1 (ascending within a device, as vsx_allpairs_rows wants them)
  std::vector<std::vector<uint32_t>> rows(D);
  std::vector<Source> src(count);
  for (uint64_t k = 0; k < count; ++k)
    {
      const uint64_t lap = k / D, pos = k % D;
      const size_t d = (size_t) ((lap & 1) ? D - 1 - pos : pos);
      src[k] = Source {(uint32_t) d, rows[d].size()};
      rows[d].push_back((uint32_t) (first + k));
    }
  std::vector<vsx_hits> H(D);
  for (vsx_hits & h : H) std::memset(&h, 0, sizeof h);
  const int rc = on_all(m, [&](size_t d) -> int {
    return vsx_allpairs_rows(m->rep[d].S, acceptall, rows[d].data(), rows[d].size(), &H[d]);
  });
  int mrc = rc;
  if (rc == VSX_OK)
    {
      const std::vector<uint64_t> shift(D, 0);               // vsx_hit.query already is the database sequence number
      mrc = merge_hits(H, src, shift, out);
    }
  for (vsx_hits & h : H) vsx_hits_free(&h);
  return mrc;
}

}  // extern "C"
