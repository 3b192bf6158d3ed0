// vsx_internal.h -- shared between the HIP kernels (vsx_device.hip) and the host layer (vsx_host.cpp).
#ifndef VSX_INTERNAL_H
#define VSX_INTERNAL_H

#include <stdint.h>
#include <hip/hip_runtime_api.h>

#define VSX_TASK_SLOTS 8          // targets per wavefront task: 4 lane groups x 2 packed int16 halves
#define VSX_GROUP_LANES 16        // lanes per group == DPP row
#define VSX_MAX_SEQLEN_SUM 65535LL        // reference core/align_simd.cpp:89
#define VSX_MAX_SEQLEN_PRODUCT 25000000LL // reference core/align_simd.cpp:88
#define VSX_TABLE_LEN (65536 + 64)
#define VSX_CODE_SLACK 64          // readable bytes before and after a sequence set's 4-bit codes
#define VSX_CK_SLACK_DW 4096        // readable dwords before the first and after the last task's checkpoints of a chunk (vsx_traceback_tilt_kernel
                                   // loads the row-checkpoint pairs p0 - 1 .. p0 + 8 around a tile unclamped: affine addresses, immediate offsets)
// TILT class: checkpoints leave the DP kernel through an LDS transposition, so that in HBM every pipeline lane owns CONTIGUOUS
// 48-byte segments (8 steps of row checkpoints; a third of a column checkpoint at R = 16) while every store instruction still
// writes 1 KB of full lines -- the traceback then reads whole segments instead of 12-16 bytes out of fifteen 128-byte lines per
// tile (vsx_device.hip).  The step count of such a task is a multiple of 8.  MEASURED (r02, profiles/r02_ckt_ab.txt): the
// traceback fetches 35 % fewer bytes (FETCH_SIZE 9.68e6 -> 6.26e6 KiB per launch) but runs only 1.7 % faster -- it is bound by
// latency at 2.5 waves per SIMD, not by HBM -- while the DP kernel pays 5 % for the barriers and the extra LDS traffic:
// 6 634 -> 6 384 GCUPS.  Kept as an A/B build (-DVSX_CKT=1), default OFF.
#ifndef VSX_CKT
#define VSX_CKT 0
#endif
// TILT class: byte query profile in LDS (primed scores are 0 .. 255) read ONE STEP AHEAD: a lane's column symbol of step t+1 is
// its neighbour's symbol of step t, so the profile rows of the next step are requested while the current step's rows are computed
// and the step never starts with an LDS round trip.  0 = int16 profile read at the top of the step (r01).
#ifndef VSX_QPL
#define VSX_QPL 1
#endif
// TILT class: the FEED2 record of the look-ahead kernels (vsx_device.hip); 0 = the r04 record (A/B)
#ifndef VSX_FEED2
#define VSX_FEED2 1
#endif
// TILT class, R >= VSX_MID_MIN_ROWS rows per lane: the DP kernel stores a SECOND row checkpoint per step, after the middle row of
// every pipeline position, and lays the column checkpoints out per half; the traceback's tiles are then R/2 rows high -- half the
// recompute area per crossing and half the register state per lane (the R >= 18 tracebacks sat at 2 waves per SIMD on registers).
// +6 B per lane-step of checkpoint stores in those classes only (square-ish shapes, where the traceback was 40 % of the step).
#ifndef VSX_MID_MIN_ROWS
#define VSX_MID_MIN_ROWS 18
#endif
#define VSX_MID(R_, TILT_) ((TILT_) && (R_) >= VSX_MID_MIN_ROWS && !VSX_CKT)

// Device-side constants derived from the 14 post-fixup penalties (reference search16_init,
// core/align_simd.cpp:1282-1376 and the QR/R vectors at :1629-1649).  "pk" = the int16 value
// replicated in both halves of a dword.
struct VsxDevParams {
  uint32_t match_pk;
  uint32_t qrq_i_pk, rq_i_pk;     // gap in target ("E"/left moves): query-row penalties, interior rows
  uint32_t qrq_r_pk, rq_r_pk;     //                                  last query row
  int32_t  qrt_i, rt_i;           // gap in query  ("F"/up moves):  target-column penalties, interior columns
  int32_t  qrt_r, rt_r;           //                                  last target column and padding
  int32_t  match, mismatch;
  int32_t  smin;                  // overflow threshold, compute_score_min (:1432-1444)
  int32_t  n_mismatch;
  int32_t  share_sub;             // 1: QR_q(interior) == QR_t(interior): the DP kernel shares one H - QR per row (see `rows`)
  int32_t  top_open, top_step;    // go / ge of a query-left terminal gap: Htop(j) = -(go + (j + 1) ge); the dummy rows of TOPPAD
  int32_t  tilt;                  // g > 0: TILTED coordinates X* = X + (i + j) g (vsx_forward_kernel TILT): every penalty above is
                                  // the original minus g, matrix = original + 2g, htop[j] / hleft[i] = original + (j - 1) g / (i - 1) g;
                                  // top_open / top_step stay the originals.  0: plain coordinates
  int32_t  max3;                  // TILT only: 1 = the MAX3 sub-class (values biased into [0, 0x7BFF]: H = max(h0, F, E) is ONE v_pk_maximum3_f16)
  const int16_t * htop;           // H(-1, j), j >= 0: top border chain (:1895-1910, :2043-2051)
  const int16_t * hleft;          // H(i, -1), i >= 0: left border chain (:844-859, :881-887)
  const int16_t * matrix;         // 16x16 score matrix S[target code][query code] (:1319-1342)
};

// One wavefront task: one query against up to 8 targets (the reference's own batch shape,
// core/searchcore.cpp:757-778).  Lane group g (16 lanes) aligns targets 2g (low int16 halves)
// and 2g+1 (high halves).
struct VsxTask {
  uint64_t qoff;                      // query codes offset
  uint64_t toff[VSX_TASK_SLOTS];      // target codes offsets
  uint64_t dir_off;                   // first dword of this task's direction block (chunk-relative)
  uint64_t strip_off;                 // first uint2 of this task's strip hand-over scratch (multi-strip only)
  uint32_t tlen[VSX_TASK_SLOTS];      // 0 = empty slot
  uint32_t qlen;
  uint32_t steps;                     // per strip: max padded target length + 15
  uint32_t rows;                      // R: query rows per lane for this task's kernel variant
  uint32_t group0;                    // first lane group of the DP wave that works on this task: 0 for a whole-wave task; a sparse task
                                      // (vsx_forward_kernel NQ) shares a wave -- and the wave's checkpoint block, dir_off -- with NQ - 1 others
};

// Accept filter evaluated in the traceback epilogue (include/vsx.h vsx_filter); enabled == 0: every pair is kept.
struct VsxFilterDev {
  int32_t enabled, iddef, leftjust, rightjust;
  double  id, weak_id, maxid, mid, query_cov, target_cov;
  int64_t maxsubs, maxgaps, mincols, maxdiffs;
  // r06, ranked plans (vsx_align_pairs_ranked): the traceback's epilogue appends every accepted / weak pair (plan-local pair index + the
  // identity the filter compared) and every pair the 16-bit DP refused at run time to two lists -- the hand-written compaction that
  // replaces the flag kernel + rocPRIM scan / scatter / segmented sort over ALL pairs of a plan (vsx_rank.hip).  rank_counts == nullptr: off.
  uint32_t * rank_counts;      // [0] kept, [1] refused
  uint32_t * kept_pair;
  double   * kept_id;
  uint32_t * refused_pair;
};

// Per (task, slot) output of the DP kernel.
struct VsxSlotOut {
  int16_t  score;
  uint16_t overflow;      // 1 = the reference's 16-bit overflow rule fired -> sentinel
  uint16_t leave;         // checkpoint kernels: the column where the traceback leaves the last query row (the run of 'I'
                          // moves from (Q-1, D-1) ends there; 0xFFFF = it runs off the left edge); D-1 = no such run
  uint16_t pad;
};

// Per pair output of the traceback kernel.
struct VsxPairOut {
  int16_t  score;
  uint16_t aligned, matches, mismatches, gaps;
  uint16_t pad;           // verdict of the accept filter (VSX_VERDICT_*), 0 without a filter
  uint32_t nruns;
  uint64_t run_off;       // offset into the dense run buffer (unordered allocation)
};

// The reference's five output arrays (align_simd.hpp:99-108) + verdict + text offset, indexed by pair id, in HBM:
// written by vsx_cigar_text_kernel, copied to the host as they are.
struct VsxSoaOut {
  int16_t  * score;
  uint16_t * aligned, * matches, * mismatches, * gaps;
  uint8_t  * verdict;
  uint64_t * text_off;
};

// Ranked, compacted hits in HBM (vsx_rank.hip): entry j = the j-th kept pair in report order
struct VsxRankedOut {
  uint32_t * pair;
  int16_t  * score;
  uint16_t * aligned, * matches, * mismatches, * gaps;
  uint8_t  * verdict;
  double   * id;
  uint64_t * text_off;
};

#ifdef __cplusplus
extern "C" {
#endif

// vsx_rank.hip: keep flags + identities + exclusive scan (d_temp == NULL: only *temp_bytes is set) ...
hipError_t vsx_rank_gather_list(const uint32_t * d_kept_pair, const double * d_kept_id, uint32_t kept, const VsxPairOut * d_out,
                                const uint64_t * d_text_off, VsxRankedOut r, hipStream_t st);
hipError_t vsx_rank_flag_scan(VsxFilterDev F, int keep_weak, const VsxPairOut * d_out, const uint32_t * d_pair_ids,
                              const uint32_t * d_pair_slot, const VsxTask * d_tasks, uint32_t ngpu_pairs, uint32_t n_pairs,
                              const uint32_t * d_runs, uint64_t runs_capacity, uint32_t * d_flag, uint32_t * d_pos, double * d_id,
                              uint32_t * d_refused /* [0] = count, then the pairs the DP refused at run time */,
                              void * d_temp, size_t * temp_bytes, hipStream_t st);
// ... then compaction, per-query stable sort by identity (descending) and the gather of the kept pairs' fields
hipError_t vsx_rank_sort_gather(const uint32_t * d_flag, const uint32_t * d_pos, const double * d_id, uint32_t n_pairs, uint32_t kept,
                                const uint32_t * d_qstart, uint32_t ngroups, double * d_key_in, double * d_key_out,
                                uint32_t * d_val_in, uint32_t * d_val_out, uint32_t * d_seg, const VsxPairOut * d_out,
                                const uint64_t * d_text_off, VsxRankedOut r, void * d_temp, size_t * temp_bytes, hipStream_t st);

// vsx_tbtext.hip: run lists -> CIGAR text (pushop / finishop, align_simd.cpp:1013-1049) + records -> output arrays
hipError_t vsx_launch_cigar_text(const VsxPairOut * d_out, const uint32_t * d_pair_ids, uint32_t npairs,
                                 const uint32_t * d_runs, uint64_t runs_capacity,
                                 uint8_t * d_text, uint64_t text_capacity, unsigned long long * d_text_cursor,
                                 VsxSoaOut soa, hipStream_t st);

// reverse complement of sequences [0, nseq) in the code domain, written to d_dst_off inside the same code buffer
hipError_t vsx_launch_revcomp(uint8_t * d_codes, const uint64_t * d_src_off, const uint64_t * d_dst_off,
                              const uint32_t * d_len, uint64_t nseq, hipStream_t st);

// launchers implemented in vsx_device.hip (host code compiled by hipcc)
hipError_t vsx_launch_encode(const uint8_t * d_ascii, uint8_t * d_codes, uint64_t nbytes, hipStream_t st);
hipError_t vsx_launch_purity(const uint8_t * d_codes, const uint64_t * d_off, const uint32_t * d_len,
                             uint64_t nseq, uint8_t * d_impure, hipStream_t st);
// rows must be one of vsx_supported_rows(); generic != 0 selects the LDS score-table variant;
// track == 0 selects the variant without overflow (H min/max) tracking
// nq = tasks per wave (1; 2 / 4 = the sparse-task classes of the TILT family: tasks of <= 4 / <= 2 targets in their first slots)
hipError_t vsx_launch_forward(int rows, int generic, int track, int ckpt, int nq, int one /* every task single-strip */, VsxDevParams P, const VsxTask * d_tasks, uint32_t ntasks,
                              const uint8_t * d_qcodes, const uint8_t * d_tcodes,
                              uint32_t * d_dir, uint2 * d_strip, VsxSlotOut * d_slot, hipStream_t st);
hipError_t vsx_launch_traceback(VsxDevParams P, const VsxTask * d_tasks, const uint32_t * d_pair_slot,
                                const uint32_t * d_pair_ids, uint32_t npairs,
                                const uint8_t * d_qcodes, const uint8_t * d_tcodes,
                                const uint32_t * d_dir, const VsxSlotOut * d_slot,
                                uint32_t * d_slab, const uint64_t * d_slab_off,
                                uint32_t * d_runs, uint64_t runs_capacity, unsigned long long * d_cursor,
                                VsxPairOut * d_out, hipStream_t st);
// fast16: the tasks never saturate (planner's TRACK = 0 class) -> biased-u16 VOP2 arithmetic in the tile recompute
hipError_t vsx_launch_traceback_ck(int rows, int fast16, VsxDevParams P, VsxFilterDev F, const VsxTask * d_tasks, const uint32_t * d_pair_slot,
                                   const uint32_t * d_pair_ids, uint32_t npairs,
                                   const uint8_t * d_qcodes, const uint8_t * d_tcodes,
                                   const uint32_t * d_ck, const VsxSlotOut * d_slot,
                                   uint32_t * d_slab, const uint64_t * d_slab_off,
                                   uint32_t * d_runs, uint64_t runs_capacity, unsigned long long * d_cursor,
                                   VsxPairOut * d_out, hipStream_t st);
// bad[0] += disagreements of v_pk_maximum3_f16 with the integer maximum over the MAX3 class's value range (must stay 0)
hipError_t vsx_launch_max3_selftest(uint32_t * d_bad /* [2]: max3, v_perm sign selectors */, hipStream_t st);
void vsx_internal_set_tb_v2(int device, int on);      // on == 0: the first traceback kernel for the TILT class on that device
uint64_t vsx_ckpt_dwords(uint64_t nstrips, uint64_t steps, uint64_t rows, int tilt /* compressed layout of the TILT class */);
const int * vsx_supported_rows(int * count);

// launchers implemented in vsx_kmer.hip (k-mer candidate counting)
hipError_t vsx_kmer_launch_sweep(int fill, const uint8_t * codes, const uint64_t * off, const uint32_t * len,
                                 const uint32_t * seq_list, uint32_t nseq, int w, uint32_t ntiles, uint32_t * bucket_count,
                                 const uint64_t * bucket_start, uint32_t * postings, const uint8_t * lower_bits, hipStream_t st);
hipError_t vsx_kmer_launch_case_bits(const uint8_t * d_ascii, uint64_t nbytes, uint8_t * d_bits /* (nbytes + 7) / 8 */, int fold_case,
                                     hipStream_t st);
// vsx_mask.hip: DUST intervals of every sequence OR-ed into the case bitmap (dword-aligned, zero-padded to a dword)
hipError_t vsx_launch_dust(const uint8_t * d_codes, const uint64_t * d_off, const uint32_t * d_len, uint64_t nseq, uint8_t * d_bits,
                           hipStream_t st);
// bucket ranges of the 8-bit class in the order the count kernel's waves read them: ranges[(tile * nslots + slot) * 256 + ..] (uint2)
hipError_t vsx_kmer_launch_ranges(const uint64_t * bucket_start, uint32_t ntiles, const uint64_t * qk_start, const uint32_t * qk,
                                  const uint32_t * minmatch, const uint32_t * qlist, uint32_t nslots, void * ranges, hipStream_t st);
// bits = 8 | 16: counter width of the kernel (queries with <= 255 unique words take 8); slots [slot_base, slot_base + nslots) of the
// batch (query = qlist ? qlist[slot] : slot); ranges (8-bit class, may be NULL) / rec / tile_count belong to THESE slots:
// (slot, tile) owns rec[(slot * ntiles + tile) * subcap ..) and tile_count[slot * ntiles + tile]
// tagged != 0: the index of a word length 9..15 (postings = tag << 16 | sequence, buckets by the word's low 16 bits)
hipError_t vsx_kmer_launch_count(int bits, int tagged, const uint32_t * postings, const uint64_t * bucket_start, const void * ranges,
                                 uint32_t ntiles, uint32_t nseq, uint32_t nslots, uint32_t slot_base, const uint64_t * qk_start,
                                 const uint32_t * qk, const uint32_t * minmatch, const uint32_t * qlist, void * rec, uint32_t subcap,
                                 uint32_t * tile_count, hipStream_t st);
hipError_t vsx_kmer_tagged_tile(int fill, const uint8_t * codes, const uint64_t * off, const uint32_t * len, uint32_t first_seq,
                                uint32_t nseq_tile, int w, const uint8_t * lower_bits, const uint64_t * slot_of /* running sum of the lengths */, uint64_t n_slots,
                                uint64_t * keys_a, uint64_t * keys_b, void * temp, size_t * temp_bytes, uint32_t tile, uint32_t ntiles,
                                uint32_t * bucket_count, const uint64_t * bucket_start, uint32_t * postings, hipStream_t st);
hipError_t vsx_kmer_launch_select(const void * rec, uint32_t subcap, uint32_t ntiles, const uint32_t * tile_count, uint32_t nslots,
                                  uint32_t keep, void * dense, unsigned long long * cursor, uint64_t capacity,
                                  void * sel_m_n, uint64_t * sel_off, hipStream_t st);
uint32_t vsx_kmer_tile_shift(void);

#ifdef __cplusplus
}
#endif
#endif
