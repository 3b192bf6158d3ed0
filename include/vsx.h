/*
  vsx.h -- C-ABI of libvsx: MI355X-native (gfx950) replacement for the global
  pairwise alignment hot path of vsearch (search16 / align_simd).

  The reference has no FFI layer; its seam for this path is the translation
  unit src/core/align_simd.cpp, whose whole external surface is four C++
  functions (src/core/align_simd.hpp:76-108).  Every entry point below cites
  the reference interface it replaces.  Plain pointers and sizes only; all
  memory handed out is malloc-compatible (the reference frees CIGARs with
  xfree == free, src/os/posix/system.cc:123).

  Error convention: functions return VSX_OK (0) or a negative VSX_E* code and
  leave a message retrievable with vsx_last_error().  A *pair* that the 16-bit
  aligner cannot represent is NOT an error: it is reported exactly as the
  reference does, score == VSX_SCORE_SENTINEL (SHRT_MAX), statistics 0, CIGAR ""
  (src/core/align_simd.cpp:1463-1479, :1867-1882, :1774-1786), and the caller
  falls back to its linear-memory aligner (src/core/searchcore.cpp:806-832).
  There is no CPU fallback inside libvsx: without a usable gfx950 device every
  compute entry point fails with VSX_ENODEVICE.
*/
#ifndef VSX_H
#define VSX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VSX_API_VERSION 1000          /* MAJOR*1000000 + MINOR*1000 + PATCH */
#define VSX_SCORE_SENTINEL 32767      /* SHRT_MAX: "use the scalar fallback" */

#define VSX_OK         0
#define VSX_EINVAL    -1
#define VSX_ENODEVICE -2
#define VSX_ENOMEM    -3
#define VSX_EHIP      -4

typedef struct vsx_ctx vsx_ctx;         /* replaces s16info_s (align_simd.cpp:411-443); opaque */
typedef struct vsx_seqset vsx_seqset;   /* device mirror of Database (core/db.hpp:100-107)     */
typedef struct vsx_plan vsx_plan;       /* a batch of (query, target) pairs bound to buffers   */

/* The 14 post-fixup scores/penalties + flag that search16_init receives
   (align_simd.hpp:76-90; call site core/search.cpp:147-161).  Same order. */
typedef struct vsx_scoring {
  int64_t match;
  int64_t mismatch;
  int64_t gap_open_query_left;
  int64_t gap_open_target_left;
  int64_t gap_open_query_interior;
  int64_t gap_open_target_interior;
  int64_t gap_open_query_right;
  int64_t gap_open_target_right;
  int64_t gap_ext_query_left;
  int64_t gap_ext_target_left;
  int64_t gap_ext_query_interior;
  int64_t gap_ext_target_interior;
  int64_t gap_ext_query_right;
  int64_t gap_ext_target_right;
  int32_t n_mismatch;                   /* opt_n_mismatch */
} vsx_scoring;

/* Per-pair results, exactly the five output arrays + CIGAR of search16
   (align_simd.hpp:99-108).  cigar_blob holds the NUL-terminated run-length
   strings back to back; cigar_off[k] is the start of pair k's string. */
typedef struct vsx_results {
  uint64_t   n_pairs;
  int16_t  * score;        /* pscores[]      (CELL)            */
  uint16_t * aligned;      /* paligned[]     alignment columns */
  uint16_t * matches;      /* pmatches[]                       */
  uint16_t * mismatches;   /* pmismatches[]                    */
  uint16_t * gaps;         /* pgaps[]        gap runs          */
  uint64_t * cigar_off;
  char     * cigar_blob;
  uint64_t   cigar_bytes;
  uint8_t  * verdict;      /* NULL unless the plan has a filter: VSX_VERDICT_* per pair */
} vsx_results;

/* Device-side accept filter: align_trim (core/searchcore.cpp:343-464) + search_acceptable_aligned (:664-737) evaluated by
   the traceback kernel on the finished alignment, with the reference's double expressions.  A REJECTED pair keeps its
   statistics but gets no CIGAR (its runs never leave the device): all-vs-all and large candidate batches return only
   what the caller will keep.  Pairs the 16-bit aligner refuses (sentinel) stay UNDECIDED for the caller's fallback. */
#define VSX_VERDICT_UNDECIDED 0
#define VSX_VERDICT_ACCEPTED  1
#define VSX_VERDICT_WEAK      2   /* rejected, but id >= weak_id and all other tests passed: reported as a weak hit */
#define VSX_VERDICT_REJECTED  3
typedef struct vsx_filter {
  int32_t iddef;            /* 0..4, --iddef */
  int32_t leftjust, rightjust;
  int32_t pad;
  double  id, weak_id, maxid, mid, query_cov, target_cov;
  int64_t maxsubs, maxgaps, mincols, maxdiffs;
} vsx_filter;
/* before vsx_plan_run; NULL removes the filter.  Needs the default (checkpoint) traceback. */
int vsx_plan_set_filter(vsx_plan * plan, const vsx_filter * f);

/* Timing of the last vsx_plan_run, measured with hipEvents on the plan's stream. */
typedef struct vsx_timing {
  float    forward_ms;      /* DP (score + direction) kernel(s)        */
  float    traceback_ms;    /* traceback / statistics / CIGAR kernel    */
  float    total_ms;        /* first launch -> last kernel done         */
  uint32_t forward_launches;
  uint32_t traceback_launches;
  uint64_t cells;           /* sum over pairs of qlen*tlen (DP cells)   */
  uint64_t dir_bytes;       /* direction bytes written by the DP kernel */
} vsx_timing;

/* How a plan was mapped onto kernel classes (for measurement code: which row body does the dominant launch issue?). */
typedef struct vsx_plan_info {
  uint64_t tasks;           /* wavefront tasks (one query x <= 8 targets)                                  */
  uint64_t tasks_tilted;    /* ... in the TILT class (tilted coordinates, compressed checkpoints)           */
  uint64_t tasks_tracked;   /* ... in the TRACK class (overflow rule evaluated, saturating arithmetic)      */
  uint32_t rows_dominant;   /* query rows per lane (R) of the launch with the most tasks                   */
  uint32_t chunks;          /* checkpoint-buffer chunks the plan runs in                                    */
  uint64_t tasks_max3;      /* ... of the tilted ones in the MAX3 sub-class (15-bit range, v_pk_maximum3_f16)       */
  uint64_t tasks_sparse;    /* ... in a sparse-task class: tasks of <= 4 / <= 2 targets that share a wave with 1 / 3 others */
  uint64_t waves;           /* wavefronts the DP launches of one run start (= tasks without sparse-task classes)            */
  uint64_t tasks_pair;      /* ... in a pair-profile class: groups of four tasks of one pure-ACGT query as one workgroup    */
} vsx_plan_info;

const char * vsx_version_string(void);
int vsx_device_count(void);                       /* usable gfx950 devices, 0 if none */
const char * vsx_last_error(void);                /* thread-local message of the last failure */

/* search16_init (align_simd.cpp:1282-1376): build aligner state on `device`. */
int vsx_create(vsx_ctx ** out, const vsx_scoring * scoring, int device);
/* search16_exit (align_simd.cpp:1379-1403). */
void vsx_destroy(vsx_ctx * ctx);

/* Threading (as s16info_s, LIBRARY_API.md:962-1000): one vsx_ctx per host thread -- a context's entry points are not
   re-entrant; any number of contexts per GPU.  A vsx_seqset is read-only device memory once created: contexts of the SAME
   device may share it (e.g. one Database mirror for all worker threads); it must outlive every plan that uses it and the
   context that created it must outlive it. */

/* Database::add / getsequence / getsequencelen (core/db.hpp:146-152,172,198):
   n ASCII sequences (any case, IUPAC; blob + offsets + lengths) are encoded to
   4-bit codes (utils/maps.cpp:75-117) ON THE DEVICE and kept resident in HBM.
   The same call serves the query side (search16_qprep, align_simd.cpp:1406-1428). */
int vsx_seqset_create(vsx_ctx * ctx, vsx_seqset ** out, uint64_t n,
                      const char * blob, uint64_t blob_bytes,
                      const uint64_t * offsets, const uint32_t * lengths);
/* Same, but `d_blob` is ASCII already resident in device memory (bench path). */
int vsx_seqset_create_from_device(vsx_ctx * ctx, vsx_seqset ** out, uint64_t n,
                                  const void * d_blob, uint64_t blob_bytes,
                                  const uint64_t * offsets, const uint32_t * lengths);
/* Both strands of n sequences (--strand both: core/search.cpp:200-214 searches every query and its reverse complement,
   utils/reverse_complement.cpp:70-82): a set of 2n sequences, k < n as given, n + k = the reverse complement of k,
   computed ON THE DEVICE from the 4-bit codes -- only the plus strands cross PCIe. */
int vsx_seqset_create_both_strands(vsx_ctx * ctx, vsx_seqset ** out, uint64_t n,
                                   const char * blob, uint64_t blob_bytes,
                                   const uint64_t * offsets, const uint32_t * lengths);
void vsx_seqset_destroy(vsx_seqset * s);
uint64_t vsx_seqset_count(const vsx_seqset * s);

/* search16 (align_simd.cpp:1447-2060), generalised from "one query, n targets"
   to an arbitrary pair list: pair k aligns queries[qidx[k]] with targets[tidx[k]].
   A pair's result is a pure function of (query, target, scoring) in the reference
   (lanes are independent), so batching across queries changes nothing.
   vsx_plan_create groups pairs by query into wavefront tasks (<= 8 targets each),
   uploads the task list and sizes the device buffers; vsx_plan_run launches the
   DP + traceback kernels (asynchronously, in chunks that fit `dir_budget_bytes`
   of direction storage; 0 = default) and, behind them, the kernel that formats the
   CIGAR text and the output arrays in HBM (pushop/finishop, align_simd.cpp:1013-1049);
   vsx_plan_fetch waits and copies them to the host (pinned staging -> malloc'd arrays;
   strings are 4-byte aligned inside cigar_blob, in no particular order). */
int vsx_plan_create(vsx_ctx * ctx, vsx_plan ** out,
                    const vsx_seqset * queries, const vsx_seqset * targets,
                    uint64_t n_pairs, const uint32_t * qidx, const uint32_t * tidx,
                    uint64_t dir_budget_bytes);
int vsx_plan_run(vsx_plan * plan);
int vsx_plan_sync(vsx_plan * plan, vsx_timing * timing /* may be NULL */);
int vsx_plan_fetch(vsx_plan * plan, vsx_results * out);
int vsx_plan_describe(const vsx_plan * plan, vsx_plan_info * info);
/* Device-side hit records of the last run, for a multi-GPU gather without a host round trip:
   copies n_pairs records of VSX_HIT_RECORD_BYTES each {int16 score; uint16 aligned, matches,
   mismatches, gaps, pad; uint32 n_cigar_runs; uint64 cigar_run_offset} into device memory `d_dst`
   (asynchronously on the plan's stream, then waits).  Records of pairs resolved on the host
   (sentinels / empty query) are filled from the host copy. */
#define VSX_HIT_RECORD_BYTES 24
int vsx_plan_export_hits(vsx_plan * plan, void * d_dst, uint64_t dst_bytes);
/* The dense run buffer those records point into (cigar_run_offset, n_cigar_runs): n_runs words of (length << 2) | op,
   op 0 = M, 1 = I, 2 = D, each pair's runs in traceback order (last alignment column first).  d_dst == NULL only reports
   *n_runs.  Together with vsx_plan_export_hits this is what a rank contributes to the multi-GPU gather (SURVEY 8e:
   "{query, target, score, 4 stats, cigar_off} + CIGAR bytes to rank 0"); the receiver rebases cigar_run_offset by the
   run counts of the lower ranks and formats text with vsx_cigar_from_runs. */
int vsx_plan_export_runs(vsx_plan * plan, void * d_dst, uint64_t dst_bytes, uint64_t * n_runs);
/* pushop / finishop (align_simd.cpp:1013-1049) on the host: run words -> NUL-terminated CIGAR text (count omitted when 1).
   Returns the bytes needed including the NUL; writes only if cap suffices.  No device needed. */
int64_t vsx_cigar_from_runs(const uint32_t * runs, uint32_t n, char * dst, uint64_t cap);
void vsx_plan_destroy(vsx_plan * plan);

/* Convenience: create + run + fetch + destroy. */
int vsx_align_pairs_filtered(vsx_ctx * ctx, const vsx_seqset * queries, const vsx_seqset * targets,
                             uint64_t n_pairs, const uint32_t * qidx, const uint32_t * tidx,
                             const vsx_filter * filter, vsx_results * out);
int vsx_align_pairs(vsx_ctx * ctx, const vsx_seqset * queries, const vsx_seqset * targets,
                    uint64_t n_pairs, const uint32_t * qidx, const uint32_t * tidx,
                    vsx_results * out);
void vsx_results_free(vsx_results * r);

/* Ranking and compaction on the device: of a pair list grouped by query (all pairs of a query contiguous)
   only the pairs the filter KEEPS come back -- verdict ACCEPTED, plus WEAK with keep_weak -- already in report order:
   queries in list order, inside a query identity descending, then list order (= target ascending when the caller lists
   targets ascending): hit_compare_byid (core/searchcore.cpp:133-179), allpairs_hit_compare (commands/allpairs_global.cpp:
   116-138).  `id` is the identity the filter compared (iddef of the filter).  Pairs the 16-bit aligner refused (sentinel)
   are listed in `undecided` for the caller's fallback.  The filter is required. */
typedef struct vsx_ranked {
  uint64_t   n_pairs, n_hits;
  uint32_t * pair;          /* n_hits indices into the caller's pair list */
  int16_t  * score;
  uint16_t * aligned, * matches, * mismatches, * gaps;
  uint8_t  * verdict;
  double   * id;
  uint64_t * cigar_off;
  char     * cigar_blob;
  uint64_t   cigar_bytes;
  uint64_t   n_undecided;
  uint32_t * undecided;
} vsx_ranked;
int vsx_align_pairs_ranked(vsx_ctx * ctx, const vsx_seqset * queries, const vsx_seqset * targets,
                           uint64_t n_pairs, const uint32_t * qidx, const uint32_t * tidx,
                           const vsx_filter * filter, int keep_weak, vsx_ranked * out);
void vsx_ranked_free(vsx_ranked * r);

#ifdef __cplusplus
}
#endif
#endif /* VSX_H */
