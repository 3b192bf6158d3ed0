// vsx_kmer.h -- host-side handle of the device k-mer index (vsx_kmer.hip / vsx_kmer_host.cpp); internal C++ API used by
// the dispatch layer (vsx_search.cpp).
#ifndef VSX_KMER_H
#define VSX_KMER_H

#include <stdint.h>
#include <vector>
#include "../../include/vsx.h"

// candidates of a counting batch, grouped by query: query q owns rec[off[q] .. off[q] + cnt[q]), each (target, count) packed
// as target | (uint64_t) count << 32; order inside a query arbitrary
struct VsxKmerResult {
  std::vector<uint64_t> rec;
  std::vector<uint64_t> off;
  std::vector<uint32_t> cnt;
};
struct VsxKmerIndex;

struct VsxKmerStats {
  double build_ms = 0;            // index build (two sweeps + prefix sum), device time + the small D2H/H2D of the bucket table
  double count_ms = 0;            // last vsx_kmer_count_batch: kernel time (hipEvents)
  uint64_t postings = 0;          // entries in the index
  uint64_t increments = 0;        // last batch: counter updates = postings streamed
  uint64_t streamed_bytes = 0;    // last batch: bytes of postings the count kernel read (2 per posting; tagged 4; packed 16 per unit of <= 15)
  uint64_t records = 0;           // last batch: candidates emitted
  uint64_t index_bytes = 0;
};

// words of length w (3..8) over the sequence set's 4-bit codes; ambiguous symbols poison the words that cover them, and so do
// lower-case symbols when the set was made with its case bitmap (soft masking, vsx_internal_seqset_create_cased)
int vsx_kmer_index_create(vsx_ctx * ctx, const vsx_seqset * db, int w, VsxKmerIndex ** out);
// an index with nothing in it yet; vsx_kmer_index_rebuild(ix, list, n) (re)builds it over the listed sequences of `db`
// (index position i = sequence list[i]; targets in the records are POSITIONS) or, with list == nullptr, over all of them.
int vsx_kmer_index_create_empty(vsx_ctx * ctx, const vsx_seqset * db, int w, VsxKmerIndex ** out);
int vsx_kmer_index_rebuild(VsxKmerIndex * ix, const uint32_t * list, uint64_t n_list);
void vsx_kmer_index_destroy(VsxKmerIndex * ix);
// qk_start[nq + 1] / qk[]: each query's unique words; minmatch[q] = threshold, 0xffffffff = skip the query.
// out: per query the (target, count) records with count >= minmatch[query] that survive the device selection.
// keep = size of the reference's heap (tophits): per query the device keeps every record whose count is >= the
// keep-th largest count (a superset of the heap under any tie-break); the caller applies the total order.
// Thread-safe against other count batches on the same index (each call leases its own scratch set and stream; at most two run
// at once, further callers wait), NOT against vsx_kmer_index_rebuild.  stats_out: this call's kernel time / increments / records.
int vsx_kmer_count_batch(VsxKmerIndex * ix, uint64_t nq, const uint64_t * qk_start, const uint32_t * qk,
                         const uint32_t * minmatch, uint32_t keep, VsxKmerResult & out, uint32_t cap_hint = 0,
                         VsxKmerStats * stats_out = nullptr, bool want_increments = false /* stats.increments costs a serial pass over qk */);
const VsxKmerStats * vsx_kmer_stats(const VsxKmerIndex * ix);

#endif
