/*
  vsx_search.h -- C-ABI of the candidate-batch dispatch layer of libvsx: the part of vsearch's search core
  that decides WHICH (query, target) pairs reach the aligner and what happens to the results.

  Replaces (all citations relative to the reference's src/):
    search_onequery        core/searchcore.cpp:884-957   candidate loop, batches of MAXDELAYED = 8
    align_delayed          core/searchcore.cpp:740-881   batch alignment + sequential accept/reject bookkeeping
    search_acceptable_unaligned / _aligned   :541-609 / :664-737
    align_trim             core/searchcore.cpp:343-464   terminal-gap trimming, id0..id4
    search_joinhits / hit_compare_byid       :1028-1052 / :133-179
    search_topscores + unique_count + Dbindex + minheap order   (device: vsx_kmer.hip; host restatement kept as checker)
                           core/searchcore.cpp:260-340, core/unique.cpp:155-352, core/dbindex.cpp:163-255,
                           core/minheap.cpp:82-146
  and mirrors the reference's library entry points search_session_* / search_batch
  (core/search.hpp:88-145): one searcher per (database, options), batches of queries in, hit lists out.

  Control flow: instead of one search16 call of <= 8 targets per query, a WINDOW of queries advances in
  lock step -- every open query contributes its next delayed batch (exactly the targets the reference's
  align_delayed would pass to search16), all batches of the window go to the GPU as one vsx_plan, and the
  accept/reject counters are then replayed per query in the reference's order.  Every hit field is identical to
  the reference's; the set of aligned pairs is the reference's or (vsx_search_batch, r05: lazy first batches) a subset
  of it that leaves out only alignments the reference computes and frees unread.
*/
#ifndef VSX_SEARCH_H
#define VSX_SEARCH_H

#include "vsx.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct vsx_searcher vsx_searcher;

/* The Parameters fields this path reads (src/vsearch.h:224-540), with the reference's defaults after
   vsearch_apply_defaults_fixups (src/vsearch.cc:186-278).  Fill with vsx_search_opts_default() first. */
typedef struct vsx_search_opts {
  double  id;               /* --id (fraction, e.g. 0.9); required                               */
  double  weak_id;          /* --weak_id; default 10.0, clamped to id                             */
  int64_t maxaccepts;       /* 1;  0 = all                                                        */
  int64_t maxrejects;       /* 32; 0 = all                                                        */
  int64_t wordlength;       /* 8   (3..15)                                                        */
  int64_t minwordmatches;   /* -1 = table minwordmatches_defaults (core/searchcore.hpp:75-76)     */
  int32_t iddef;            /* 2                                                                  */
  int32_t soft_mask;        /* masking before the k-mer stage, of the DATABASE and -- unless qmask says otherwise -- of the queries
                               (--dbmask / --qmask; clustering masks everything by --qmask: pass it here; unique_count masks lower case for
                               every mode but "none", core/unique.cpp:198-199):
                                 0  none  lower case searchable
                                 1  soft  lower-case symbols are left out of the k-mers
                                 2  dust  the reference's DEFAULT: the database is DUST-masked on the device when the
                                          searcher is created (vsx_mask.hip), every query -- each strand on its own -- on the
                                          host threads (core/mask.cpp:79-199, core/search.cpp:294-303), then as 1.
                               All three stay on the device k-mer path (a per-symbol case bitmap beside the 4-bit codes);
                               the alignment itself never reads the case.  See `hardmask` below. */
  int64_t maxsubs, maxgaps, mincols, maxdiffs;
  double  query_cov, target_cov, maxid, mid;
  int32_t leftjust, rightjust;
  double  minqt, maxqt, minsl, maxsl;
  int64_t idprefix, idsuffix;
  int32_t selfid;
  int32_t threads;          /* host threads for the k-mer heuristic; 0 = all usable CPUs          */
  int64_t window;           /* queries advanced together; 0 = default (65536)                     */
  uint32_t gap_infinite;    /* '*' gap penalties (cli.cc:307-384): bit k = penalty k is infinite, k in the order of
                               vsx_scoring after match/mismatch (0..5 open q_l t_l q_i t_i q_r t_r, 6..11 extension);
                               an alignment using a forbidden gap class is rejected (searchcore.cpp:621-660) */
  uint32_t strand_both;     /* 0: --strand plus (default); 1: --strand both -- the reverse complement of every query is
                               searched as well (search.cpp:200-214), hits of both strands are joined (searchcore.cpp:1028-1052) */
  /* abundance-aware part of search_acceptable_unaligned (searchcore.cpp:541-609); abundances arrive through
     struct vsx_seq_meta; the caller resolves --sizein (the ';size=' annotation, or 1) */
  int64_t maxqsize;         /* --maxqsize, default INT64_MAX: query abundance above it -> every target rejected           */
  int64_t mintsize;         /* --mintsize, default 0                                                                    */
  double  minsizeratio;     /* --minsizeratio, default 0: query size >= ratio * target size (abundance_ratio_cmp :480)  */
  double  maxsizeratio;     /* --maxsizeratio, default DBL_MAX                                                          */
  int32_t self;             /* --self: a target whose label equals the query's label is rejected (needs labels)         */
  int32_t sizeorder;        /* --sizeorder (cluster_*): the member joins the most ABUNDANT accepted centroid
                               (hit_compare_bysize / search_findbest2_bysize, searchcore.cpp:182-243, :994-1025)        */
  int32_t cluster_unoise;   /* --cluster_unoise: UNOISE skew rule instead of the --id threshold in
                               search_acceptable_aligned (searchcore.cpp:701-718); weak_id is forced to 0.90 (cli.cc:4153) */
  int32_t qmask;            /* --qmask when it differs from --dbmask: 0 = the queries are masked as soft_mask says (default),
                               otherwise 1 + mode (1 none, 2 soft, 3 dust).  Searching only; clustering uses soft_mask */
  double  unoise_alpha;     /* --unoise_alpha, default 2.0                                                              */
  int32_t hardmask;         /* --hardmask (core/mask.cpp:137-191,248-271; core/search.cpp:294-303; commands/usearch_global.cpp,
                               core/cluster.cpp:1192-1197): masked symbols become 'N' IN THE SEQUENCE -- the k-mer stage and the
                               alignment both see them.  Bit 0: the database text handed to vsx_searcher_create (mode dust: the DUST
                               intervals become 'N' and the text keeps its case; mode soft: every lower-case symbol becomes 'N'; none:
                               nothing); bit 1: the queries, each strand on its own, by the query mask mode.  3 = the reference's command
                               line; 2 = a caller whose database is already hard-masked (the library API, shim/vsx_api_adapter.cpp).
                               Default 0.  Hits, %id and CIGARs then refer to the masked sequences, as in the reference. */
} vsx_search_opts;

void vsx_search_opts_default(vsx_search_opts * o);

/* One reported hit: the fields of `struct hit` (core/searchcore.hpp:78-126) that survive search_joinhits. */
typedef struct vsx_hit {
  uint32_t query;           /* index into the batch                                               */
  uint32_t target;          /* database sequence number                                           */
  uint32_t count;           /* shared unique k-mers (candidate rank key)                          */
  uint8_t  accepted, weak, used_fallback, strand;   /* strand: 0 = plus, 1 = minus (the query's reverse complement matched) */
  int32_t  nwscore, nwdiff, nwgaps, nwindels, nwalignmentlength;
  int32_t  matches, mismatches;
  int32_t  internal_alignmentlength, internal_gaps, internal_indels;
  int32_t  trim_q_left, trim_q_right, trim_t_left, trim_t_right;
  int32_t  shortest, longest;
  double   nwid, id, id0, id1, id2, id3, id4;
  uint64_t cigar_off;       /* into vsx_hits.cigar_blob (NUL-terminated)                          */
} vsx_hit;

typedef struct vsx_hits {
  uint64_t   n_queries;
  uint64_t   n_hits;
  uint64_t * first;         /* n_queries + 1: hits of query q are hit[first[q] .. first[q+1]), best first */
  vsx_hit  * hit;
  char     * cigar_blob;
  uint64_t   cigar_bytes;
  /* accounting.  pairs_aligned / cells_aligned: what reached the aligner.  vsx_search_batch aligns LAZILY since r05 -- a query's first batch is
     as many candidates as it still needs accepts (min(8, maxaccepts)) instead of the reference's 8, later batches are eights -- so the number
     is <= the pairs the reference passes to search16, with identical hits; VSX_SEARCH_LAZY=0 restores the reference's batches.
     allpairs and clustering align exactly the reference's pairs. */
  uint64_t   pairs_aligned, cells_aligned, stages, sentinel_pairs;
  double     seconds_kmer, seconds_align, seconds_total;
} vsx_hits;

/* Database + k-mer index (Database::add + Dbindex::prepare/add_all_sequences): n ASCII sequences.
   The sequences are also mirrored in HBM (vsx_seqset) for the aligner. */
int vsx_searcher_create(vsx_ctx * ctx, vsx_searcher ** out, const vsx_search_opts * opts,
                        uint64_t n, const char * blob, uint64_t blob_bytes,
                        const uint64_t * offsets, const uint32_t * lengths);
void vsx_searcher_destroy(vsx_searcher * s);
/* The database text AS INDEXED AND ALIGNED: the caller's blob with the searcher's masking applied (DUST: upper case with the masked
   A C G T U in lower case -- a masked ambiguity code stays upper case, no step reads its case; --hardmask: the masked symbols as 'N', core/mask.cpp:137-191,248-271).  What the reference prints as the target
   rows of --alnout after dust_all / hardmask_all rewrote the Database (db.hpp:179).  Copies min(cap, blob_bytes) bytes; returns blob_bytes. */
uint64_t vsx_searcher_db_text(const vsx_searcher * s, char * dst, uint64_t cap);

/* Per-sequence annotations the filters above read: Database::getabundance / getheader (core/db.hpp).  Either member may
   be NULL (abundances all 1, no labels).  The arrays are copied. */
typedef struct vsx_seq_meta {
  const uint64_t * abundance;       /* n values                                   */
  const char * const * label;       /* n NUL-terminated headers                   */
} vsx_seq_meta;
/* annotations of the searcher's database sequences (targets; in allpairs / clustering also the queries) */
int vsx_searcher_set_meta(vsx_searcher * s, const vsx_seq_meta * meta);

/* search_batch (core/search.hpp:131-145): every query against the database, --strand plus or both (opts.strand_both).
   Batches above 32 768 queries run as a pipeline of windows -- unique words (host threads) -> device k-mer counting + ranking (two
   workers) -> alignment, accept / reject replay and hit joining (two consumers; the second one on an aligner context of its own
   that the searcher creates on the same device and keeps) -- so one call uses several host threads and two contexts' worth of
   device scratch.  Results do not depend on the window size or the number of workers.  Not re-entrant per searcher. */
int vsx_search_batch(vsx_searcher * s, uint64_t n_queries, const char * qblob, uint64_t qblob_bytes,
                     const uint64_t * qoffsets, const uint32_t * qlengths, vsx_hits * out);
/* the same with the queries' annotations (si->qsize, si->query_head: core/search.cpp:88-94); qmeta may be NULL */
int vsx_search_batch_meta(vsx_searcher * s, uint64_t n_queries, const char * qblob, uint64_t qblob_bytes,
                          const uint64_t * qoffsets, const uint32_t * qlengths, const vsx_seq_meta * qmeta, vsx_hits * out);
void vsx_hits_free(vsx_hits * h);

/* DUST-mask sequences in place, as dust() of the reference does (core/mask.cpp:127-199): everything upper case, the
   low-complexity intervals lower case.  Host threads (threads <= 0: all usable CPUs); sequences must not overlap.  What
   soft_mask = 2 applies to the queries; exported for callers that prepare their own text. */
int vsx_dust_mask(char * blob, uint64_t n, const uint64_t * offsets, const uint32_t * lengths, int32_t threads);

/* Sign of (value - ratio * reference) as the abundance filters compare it (--minsizeratio / --maxsizeratio / abskew;
   core/searchcore.cpp:480-537): the rounded double product while both abundances are below 2^53, exact integer arithmetic on
   the double's stored value beyond.  Host only; exported so that callers and tests can pin the boundary behaviour. */
int vsx_abundance_ratio_cmp(int64_t value, double ratio, int64_t reference);

/* allpairs_global (commands/allpairs_global.cpp:394-527): database sequences [first, first+count) as
   queries, each against every LATER sequence (unaligned filters applied unless acceptall); kept hits are
   those accepted (or all with acceptall), ordered id desc, target asc.  vsx_hit.query is the database
   sequence number.  Callers walk the database in blocks to bound the result size. */
int vsx_allpairs_block(vsx_searcher * s, int32_t acceptall, uint64_t first, uint64_t count, vsx_hits * out);
/* the same for an arbitrary ascending list of query rows -- the multi-GPU form: the triangular pair space is sharded by
   interleaved rows (vsearch_amd/sharding.py shard_allpairs_rows), every rank runs its rows against its DB replica.
   hits.first has count + 1 entries (entry k = rows[k]); vsx_hit.query is the database sequence number. */
int vsx_allpairs_rows(vsx_searcher * s, int32_t acceptall, const uint32_t * rows, uint64_t count, vsx_hits * out);
/* the whole command as one call (r05): rows first .. first + count - 1 in blocks of `block` queries (0 = 1 000), the stages of consecutive
   blocks overlapped -- while block i is on the GPU, block i + 1's pair list is enumerated and block i - 1's hits are completed on host
   threads.  `sink` receives each block's hits (those of vsx_allpairs_block(first + i * block, ...)), in order, from a helper thread, one
   call at a time; the hits belong to the library and are released when the sink returns.  A non-zero return of the sink stops the run
   and becomes the function's result.  The reference's worker threads report each query as it finishes (commands/allpairs_global.cpp:394-527). */
typedef int (*vsx_hits_sink)(void * user, uint64_t first, uint64_t count, const vsx_hits * hits);
int vsx_allpairs_stream(vsx_searcher * s, int32_t acceptall, uint64_t first, uint64_t count, uint64_t block, vsx_hits_sink sink, void * user);

/* ---- multi-device form: several GPUs of one node behind ONE handle (vsx_multi.cpp) ----------------------------------------
   The reference is one process with a pool of worker threads over a shared database (commands/usearch_global.cpp:500-535,
   core/search.cpp:397-508); here the pool is one aligner context + database replica + k-mer index per listed device and one
   host thread per device.  vsx_multi_search_batch cuts the batch into contiguous blocks of queries (one per device), runs
   vsx_search_batch_meta on each concurrently and merges the hit lists in the caller's query order: the result is what a single
   vsx_search_batch_meta of all queries returns, hit for hit (the accounting fields are summed, the times are the slowest
   device's).  vsx_multi_allpairs deals the rows first .. first + count - 1 of the triangular pair space out boustrophedon (balanced
   pairs and cells) and merges per row: the result of vsx_allpairs_block(first, count).  Clustering does not shard (sequential
   centroid dependency): use vsx_multi_searcher_replica(m, 0) with vsx_cluster_fast.  No collective is involved: the multi-PROCESS
   form with an RCCL gather is vsearch_amd/sharding.py.  A device may be listed more than once (several replicas on one GPU). */
typedef struct vsx_multi_searcher vsx_multi_searcher;
int vsx_multi_searcher_create(vsx_multi_searcher ** out, const vsx_scoring * scoring, const int32_t * devices, int32_t n_devices,
                              const vsx_search_opts * opts, uint64_t n, const char * blob, uint64_t blob_bytes,
                              const uint64_t * offsets, const uint32_t * lengths, const vsx_seq_meta * meta /* may be NULL */);
void vsx_multi_searcher_destroy(vsx_multi_searcher * m);
int32_t vsx_multi_searcher_devices(const vsx_multi_searcher * m);
vsx_searcher * vsx_multi_searcher_replica(vsx_multi_searcher * m, int32_t k);       /* owned by m */
int vsx_multi_search_batch(vsx_multi_searcher * m, uint64_t n_queries, const char * qblob, uint64_t qblob_bytes,
                           const uint64_t * qoffsets, const uint32_t * qlengths, const vsx_seq_meta * qmeta /* may be NULL */, vsx_hits * out);
int vsx_multi_allpairs(vsx_multi_searcher * m, int32_t acceptall, uint64_t first, uint64_t count, vsx_hits * out);

/* Candidate list of ONE query exactly as search_topscores + minheap_sort produce it (best first):
   fills up to `cap` (target, count) pairs, returns the number of candidates. For tests / tooling. */
int64_t vsx_search_candidates(vsx_searcher * s, const char * q, uint32_t qlen,
                              uint32_t * targets, uint32_t * counts, uint64_t cap);

/* Candidate lists of a BATCH of queries (search_topscores + heap order, core/searchcore.cpp:260-340,
   core/minheap.cpp:82-146): best first, at most maxaccepts + maxrejects + 8 per query.  device != 0 counts on the GPU
   (vsx_kmer.hip: tiled LDS counters over a device-resident index), 0 on host threads; both give identical lists.
   All arrays are malloc()'d; release with vsx_candidates_free. */
typedef struct vsx_candidates {
  uint64_t   n_queries;
  uint64_t * start;            /* n_queries + 1 */
  uint32_t * target;
  uint32_t * count;
  double     seconds;          /* wall time of the call (host k-mer extraction + counting + ranking) */
  double     kernel_ms;        /* device counting kernel(s), hipEvents; 0 on the host path */
  double     index_build_ms;   /* device index build, paid on first use */
  uint64_t   index_postings;   /* entries of the index */
  uint64_t   postings_streamed;/* counter increments of this call = postings read */
  uint64_t   bytes_streamed;   /* bytes of postings the count kernel read (the index format decides: 2 per posting, 4 tagged, 16 per packed unit) */
} vsx_candidates;
int vsx_search_candidates_batch(vsx_searcher * s, int32_t device, uint64_t n, const char * qblob, uint64_t qbytes,
                                const uint64_t * qoff, const uint32_t * qlen, vsx_candidates * out);
void vsx_candidates_free(vsx_candidates * c);

/* Greedy centroid clustering, --cluster_fast / --cluster_smallmem semantics (core/cluster.cpp:877-1125 with the
   intra-round fix-up evaluate_extra_hits :601-856): the searcher's sequences are processed IN THE GIVEN ORDER
   (sort them first: cluster_fast = length descending, core/db.cpp:433-450); each joins the cluster of its best
   accepted centroid hit or founds a new cluster.  `round` sequences are searched per GPU stage (0 = 16384); the
   result is independent of `round`.  hits.first[s]..first[s+1] holds the one hit of a member (target = its
   centroid) and is empty for centroids; clusterno[s] is the 0-based cluster number in creation order. */
typedef struct vsx_cluster_out {
  uint64_t   n;
  uint64_t   n_clusters;
  uint32_t * clusterno;
  vsx_hits   hits;
} vsx_cluster_out;
int vsx_cluster_fast(vsx_searcher * s, uint64_t round, vsx_cluster_out * out);
void vsx_cluster_out_free(vsx_cluster_out * o);

/* Star multiple alignment / profile / consensus of ONE cluster from the members' CIGARs (core/msa.cpp:569-613;
   no DP).  seqs[0] is the centroid (cigars[0] ignored), members follow with the CIGAR of (query = member,
   target = centroid) exactly as vsx_cluster_fast reports it.  rows: n_rows = n + 1 NUL-terminated strings of alnlen
   columns each (centroid, members, then the consensus row with '+' outside the centroid and '-' where gaps win);
   profile: alnlen x 6 counts in the order A C G T N gap (msa.cpp:92-141); consensus: the un-gapped majority sequence. */
typedef struct vsx_msa_out {
  uint64_t   alnlen, n_rows, conslen;
  char     * rows;
  char     * consensus;
  uint64_t * profile;
} vsx_msa_out;
int vsx_msa(uint32_t n, const char * const * seqs, const uint32_t * lens, const char * const * cigars,
            const uint64_t * abundances /* NULL = all 1 */, vsx_msa_out * out);
void vsx_msa_out_free(vsx_msa_out * o);
/* The same on the device (vsx_msa.hip), for one cluster or for many in one pass: cluster c owns entries
   cluster_start[c] .. cluster_start[c+1]-1 of seqs/lens/cigars/abundances (first entry = its centroid, whose cigar is
   ignored); outs[c] receives what vsx_msa would return for it (byte-identical; free each with vsx_msa_out_free).
   The CIGAR text is parsed and max-insertions/column offsets are computed on the host (O(runs)); the O(rows x alnlen)
   work -- row fill, profile, consensus row -- runs in three kernels.  Fails (VSX_EINVAL) on what vsx_msa rejects and on
   two adjacent 'D' runs; no CPU fallback. */
int vsx_msa_device(vsx_ctx * ctx, uint32_t n, const char * const * seqs, const uint32_t * lens, const char * const * cigars,
                   const uint64_t * abundances /* NULL = all 1 */, vsx_msa_out * out);
int vsx_msa_device_batch(vsx_ctx * ctx, uint32_t n_clusters, const uint64_t * cluster_start /* n_clusters + 1 */,
                         const char * const * seqs, const uint32_t * lens, const char * const * cigars,
                         const uint64_t * abundances /* NULL = all 1 */, vsx_msa_out * outs /* n_clusters */);

/* The scalar fallback the callers run on the SHRT_MAX sentinel: LinearMemoryAligner::align + alignstats
   (core/linmemalign.cpp:694-808; call sites core/searchcore.cpp:806-832, commands/allpairs_global.cpp:447-473).
   Host CPU, int64 arithmetic, linear memory, the reference's tie-breaks; uses the UNclamped scoring values
   (scoring_from_options, linmemalign.cpp:99-118).  *cigar is malloc'ed. */
int vsx_lma_align(const vsx_scoring * scoring, const char * q, uint64_t qlen, const char * t, uint64_t tlen,
                  int64_t * score, int64_t * alnlen, int64_t * matches, int64_t * mismatches,
                  int64_t * gaps, char ** cigar);

#ifdef __cplusplus
}
#endif
#endif
