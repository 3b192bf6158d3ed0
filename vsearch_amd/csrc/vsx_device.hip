// vsx_device.hip -- hand-written gfx950 (CDNA4) kernels for the vsearch global-alignment hot path.
//
// Replaces, bit for bit, the arithmetic of the reference's 8-lane SSE2 aligner
// (src/core/align_simd.cpp: onestep :752-781, aligncolumns_first/rest :783-1010, the search16
// driver :1447-2060 and backtrack16 :1052-1245) with a wavefront design:
//
//   * one wavefront (64 lanes) = one TASK = one query x up to 8 targets;
//   * a 16-lane group (one DPP row) sweeps TWO targets at once: target 2g lives in the low
//     int16 half of every VGPR, target 2g+1 in the high half (v_pk_*_i16, saturating = `clamp`);
//   * lane l of a group owns R consecutive query rows; at step t it processes target column
//     j = t - l (systolic skew), so H/F cross lanes through one v_mov_b32_dpp row_shr:1 per step
//     and E / the left-neighbour H never leave VGPRs;
//   * the column descriptor (target symbol pair, column penalties, score delta) travels down the
//     same DPP pipeline; lane 0 is fed from a 16-column block parked in LDS once per 16 steps;
//   * default (CKPT): no direction bits are stored.  Every step each lane stores the (H, F) pair it hands to the next pipeline
//     position (row checkpoints) and every 16 steps its column state hprev[R], E[R] (column checkpoints); the traceback kernel
//     (one lane per pair: vsx_traceback_tilt_kernel for the tilted class -- r04: position-synchronous iterations, two rows of a pair
//     per register -- vsx_traceback_ck_kernel for the plain classes) recomputes the direction bits only for the <= R x 16 tiles
//     the path crosses and walks them with backtrack16's rules, emitting statistics and the run-length CIGAR;
//   * default arithmetic (TILT, tasks whose score range the planner can bound): tilted coordinates X* = X + (i + j) g remove
//     F - R and E - R from the interior (7 instructions per lane-row); values biased into unsigned halves so that the add and
//     the subtraction are 32-bit ops over both halves.  Exact: every maximum compares two values of the same cell;
//   * VSX_TRACEBACK=dirs keeps the first design: the 4 direction bits per cell are the SIGN bits of four saturating
//     subtractions (a > b  <=>  ssub(b, a) < 0, exact under saturation), funnelled into 16-bit fields with v_lshrrev_b32 +
//     v_bfi_b32, R/4 dwords per lane and step, walked by vsx_traceback_kernel;
//   * no MFMA: this is a max-plus recurrence on int16, bound by VALU issue.
#include <hip/hip_runtime.h>
#include <type_traits>
#include <cstdlib>
#include <atomic>
#include "vsx_internal.h"

typedef unsigned int u32;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x3 __attribute__((ext_vector_type(3)));
typedef short s2 __attribute__((ext_vector_type(2)));
typedef unsigned short us2 __attribute__((ext_vector_type(2)));

#define DEV __device__ __forceinline__
struct __attribute__((aligned(4))) Pair2 { unsigned int x, y; };   // 8 bytes at dword alignment (ds_read2_b32)

#define SIGN2 0x80008000u

DEV s2 S2(u32 x) { return __builtin_bit_cast(s2, x); }
DEV us2 U2(u32 x) { return __builtin_bit_cast(us2, x); }
DEV u32 UI(s2 x) { return __builtin_bit_cast(u32, x); }
DEV u32 UI(us2 x) { return __builtin_bit_cast(u32, x); }

DEV u32 sadd(u32 a, u32 b) { return UI(__builtin_elementwise_add_sat(S2(a), S2(b))); }   // v_pk_add_i16 clamp
DEV u32 ssub(u32 a, u32 b) { return UI(__builtin_elementwise_sub_sat(S2(a), S2(b))); }   // v_pk_sub_i16 clamp
DEV u32 pmax(u32 a, u32 b) { return UI(__builtin_elementwise_max(S2(a), S2(b))); }       // v_pk_max_i16
DEV u32 pmin(u32 a, u32 b) { return UI(__builtin_elementwise_min(S2(a), S2(b))); }       // v_pk_min_i16
DEV u32 pminu(u32 a, u32 b) { return UI(__builtin_elementwise_min(U2(a), U2(b))); }      // v_pk_min_u16
DEV u32 pmaxu(u32 a, u32 b) { return UI(__builtin_elementwise_max(U2(a), U2(b))); }      // v_pk_max_u16
DEV u32 psubw(u32 a, u32 b) { return UI((us2) (U2(a) - U2(b))); }                        // v_pk_sub_u16 (wraps per half)
DEV u32 pmad(u32 a, u32 b, u32 c) { return UI((us2) (U2(a) * U2(b) + U2(c))); }          // v_pk_mad_u16
DEV u32 pashr15(u32 a) { return UI((s2) (S2(a) >> (s2){15, 15})); }                      // v_pk_ashrrev_i16
DEV u32 bfi(u32 mask, u32 a, u32 b) { return (a & mask) | (b & ~mask); }                 // v_bfi_b32
// Pinned encodings: hipcc otherwise rewrites these idioms into v_cmp/v_cndmask/v_perm sequences.
DEV u32 a_pk_minu(u32 a, u32 b) { u32 r; asm("v_pk_min_u16 %0, %1, %2" : "=v"(r) : "v"(a), "s"(b)); return r; }
DEV u32 a_pk_mad(u32 a, u32 b, u32 c) { u32 r; asm("v_pk_mad_u16 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "s"(c)); return r; }
DEV u32 a_bfi(u32 mask, u32 a, u32 b) { u32 r; asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(r) : "s"(mask), "v"(a), "v"(b)); return r; }
DEV u32 a_bfi_v(u32 mask, u32 a, u32 b) { u32 r; asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(r) : "v"(mask), "v"(a), "v"(b)); return r; }
DEV u32 a_lshr1(u32 a) { u32 r; asm("v_lshrrev_b32 %0, 1, %1" : "=v"(r) : "v"(a)); return r; }
DEV u32 a_pk_ashr15(u32 a) { u32 r; asm("v_pk_ashrrev_i16 %0, 15, %1 op_sel_hi:[0,1]" : "=v"(r) : "v"(a)); return r; }
// direction funnel: shift the accumulator right by one, drop the sign bits of d into bits 15 / 31
DEV u32 funnel(u32 acc, u32 d) { return a_bfi(SIGN2, d, a_lshr1(acc)); }
DEV u32 pack16(int v) { const u32 x = (u32) v & 0xffffu; return x | (x << 16); }

// lane l <- lane l-1 within each 16-lane row; lane 0 of a row keeps `feed`
DEV u32 dpp_shr1(u32 feed, u32 src) { return (u32) __builtin_amdgcn_update_dpp((int) feed, (int) src, 0x111, 0xF, 0xF, false); }
// rotate left by one within each 16-lane row (row_ror:15): lane l <- lane (l+1) mod 16
DEV u32 dpp_rol1(u32 src) { return (u32) __builtin_amdgcn_update_dpp((int) src, (int) src, 0x12F, 0xF, 0xF, false); }


// ---------------------------------------------------------------------------------------------
// DP kernel: scores + direction bits.  R = query rows per lane.  GENERIC = score lookup through a
// 16 KB LDS table (any IUPAC / unknown symbol in the QUERY); the fast variant requires a pure
// A/C/G/T(U) query and derives the score from an XOR of the 4-bit codes (targets may contain
// anything: an ambiguous target symbol scores a per-column constant against unambiguous rows).
// ---------------------------------------------------------------------------------------------
// TRACK = false drops the running H min/max (the overflow rule): the planner selects it only for tasks
// whose score range provably stays inside (SHRT_MIN + max(go+ge), SHRT_MAX) -- see vsx_host.cpp no_overflow_possible().
// CKPT = true replaces the 4 direction bits per cell by CHECKPOINTS: every step each lane stores the (H, F) pair it
// hands to the next pipeline position (row checkpoints, 8 B) and every 16 steps its whole column state hprev[R], E[R]
// (column checkpoints); the traceback kernel recomputes the direction bits only for the <= R x 16 tiles the path
// crosses.  Same bytes to HBM, ~40 % fewer VALU cycles per cell (the sign-bit funnel is gone).
// column checkpoint of one (strip, 16-step block): [group of G lanes][block of 4 rows: hprev blocks, then E blocks][lane in
// group][4] dwords -- a store instruction (one block, all lanes) writes G x 16 B runs of complete lines
#define VSX_COLCK_G 64      // lanes per group.  64 = [block][lane][4]: a store instruction writes 1 KB of full lines (DP kernel 32.1 ms, traceback 7.8 ms);
                            // 4 = one 64 B line per (quad, block): a lane's blocks and its neighbours' share L2 lines (32.8 / 7.0 ms).  Same total.
#define VSX_COLCK_DW(R_, lane_, block_) ((((size_t) ((lane_) / VSX_COLCK_G) * (size_t) ((2 * (R_)) / 4) + (size_t) (block_)) * VSX_COLCK_G + (size_t) ((lane_) % VSX_COLCK_G)) * 4)
#define VSX_RB 1            // row checkpoints: [2^RB-step block][lane][step in block] uint2
// TILT class: COMPRESSED checkpoints.  In tilted coordinates the value a lane hands on differs from its H by a bounded amount:
// F(i+1,j) = max(F - R', H - QR') with H >= F gives d = H - F(i+1,j) in [min(R', QR'), QR'], likewise H - E(i,j+1) (the planner
// admits the class only if that interval fits a signed byte).  Row checkpoints of a two-step pair: 3 dwords {H_t, H_t+1,
// bytes d_t(lo) d_t(hi) d_t+1(lo) d_t+1(hi)} instead of 4; column checkpoints: hprev[R] followed by R/2 dwords of the bytes
// d_r(lo) d_r(hi) d_r+1(lo) d_r+1(hi), padded to blocks of 4 dwords: 12 instead of 16 bytes per lane-step.
#define VSX_ROWCK_PAIR_DW(TILT_) ((TILT_) ? 192 : 256)                       // dwords of one two-step pair of a wave
#define VSX_COLCK_NBH(RT_) (((RT_) + ((RT_) + 1) / 2 + 3) / 4)                                  // ... of ONE half of RT rows (mid-row class)
#define VSX_COLCK_NB(R_, TILT_) ((TILT_) ? (VSX_MID(R_, true) ? 2 * VSX_COLCK_NBH((R_) / 2) : ((R_) + ((R_) + 1) / 2 + 3) / 4) : (2 * (R_)) / 4)   // 4-dword blocks per lane and column checkpoint
// Slot of pipeline lane (g, l) inside a wave's checkpoint chunk (TILT class).  1: l * 4 + g -- the four lane groups of a task
// (= its 8 targets, whose tracebacks move through the same tiles most of the time) sit next to each other, so the lanes of
// a traceback wave that work on one task read the same lines; 0: g * 16 + l (lane order).
#ifndef VSX_CK_TASKMAJOR
#define VSX_CK_TASKMAJOR 1
#endif
#define VSX_CK_SLOT(TILT_, g_, l_) (((TILT_) && VSX_CK_TASKMAJOR) ? ((l_) * 4 + (g_)) : ((g_) * 16 + (l_)))
#ifndef VSX_COLCK_CG
#define VSX_COLCK_CG 64     // lanes per group of the compressed layout: [group][block][lane in group][4] dwords
#endif
// transposed layout (VSX_CKT, vsx_internal.h): one block = 64 slots x 12 dwords = 3 KB, written as three 1 KB stores
#define VSX_CKT_BLOCK_DW 768
#define VSX_COLCK_NCHUNK(R_) ((VSX_COLCK_NB(R_, true) + 2) / 3)      // 48-byte chunks per lane and column checkpoint
#define VSX_COLCK_CDW(NB_, lane_, block_) ((((size_t) ((lane_) / VSX_COLCK_CG) * (size_t) (NB_) + (size_t) (block_)) * VSX_COLCK_CG + (size_t) ((lane_) % VSX_COLCK_CG)) * 4)
// TILT = true (a sub-class of TOPPAD: checkpoints, LDS profile, no tracking): the kernel runs in TILTED coordinates,
//   X*(i, j) = X(i, j) + (i + j) g   for X in {H, E, F},   g = the interior gap extension (both sides equal),
// in which the recurrence is the same max-plus recurrence with score' = score + 2g, every QR' = QR - g, every R' = R - g:
//   H*(i,j) = max(H*(i-1,j-1) + S + 2g, F*(i,j), E*(i,j)),  F*(i+1,j) = max(F*(i,j) - (Rt_j - g), H*(i,j) - (QRt_j - g)),  E* alike.
// Both sides of every comparison sit on the same cell, so every maximum picks the same operand and every direction bit is
// unchanged; the planner proves the shifted range (vsx_host.cpp tilt_possible()).  The host hands over the primed constants
// and tables (VsxDevParams with tilt = g); what this kernel adds is that in the interior R' = 0, so F - R and E - R are not
// computed at all: 7 instead of 9 packed instructions per lane-row.  Borders: Htop*(j) = Htop(j) + (j-1)g, Hleft* alike,
// H*(-1,-1) = -2g; the score is un-tilted in the epilogue; checkpoints hold tilted values (the traceback recomputes with the
// same primed constants).
// occupancy floor of the DP kernel (waves per SIMD the register allocator must make room for): 4 up to R = 16 (128 VGPRs, a few
// spills: measured better than 3 without), 3 up to R = 24 (168 VGPRs; r03: the allocator left to itself took 244 = 2 waves,
// 300 x 300: 12.23 -> 11.17 ms, 360 x 360: 15.9 -> 15.0 ms; 4 waves at R = 18 / 20: 12.7 ms), 2 beyond (256 VGPRs; 3 there
// spills into the loop: 500 x 500 25.5 -> 30.8 ms) -- profiles/r03/r03c_occupancy_ab.txt.  A/B builds override VSX_FWD_WAVES
// FEED2 (r05, prepared in r04 as experiments/dp_feed2.patch): the feed record of the look-ahead classes, see vsx_forward_kernel
// r05 same-box A/B (profiles/r05/r05b_feed2_ab.txt, two runs each): DP 250 x 1000 22.73 -> 22.41 ms, 150 x 1000 16.65 -> 16.40, 150 x 300
// 5.58 -> 5.45, 300 x 300 10.24 -> 10.20, 400 x 400 17.06 -> 17.12 (R = 26: level); parity suite + aligner soak green on the build.  Default ON.
// (the default of VSX_FEED2 lives in vsx_internal.h: the planner needs it for the pair-profile classes)
#ifndef VSX_FWD_WAVES
#define VSX_FWD_WAVES(R_, TILT_) ((R_) <= 16 ? 4 : ((R_) <= 24 ? 3 : 2))
#endif
// r06: the ONE variants need fewer registers, and R = 26 (400-bp queries: BASELINE configs[3]) now pays at three waves: 168 VGPRs + 124 B of
// scratch, 400 x 400 DP 16.06 -> 15.68 ms, with 32 candidates per query (the PAIR class) 13.74 -> 13.37; R = 28 is level (18.39 / 18.44),
// R = 32 loses (500 x 500: 21.9 -> 23.6 ms) -- profiles/r06/r06n_waves3_ab.txt
#ifndef VSX_FWD_WAVES_ONE
#define VSX_FWD_WAVES_ONE(R_, TILT_) ((R_) == 26 ? 3 : VSX_FWD_WAVES(R_, TILT_))
#endif
// MAX3 (r03, a sub-class of TILT for tasks whose tilted range fits 15 bits): the values are biased by 0x3E00 instead of 0x8000, i.e.
// every H / E / F lies in [0, 0x7BFF] -- the bit patterns of the non-negative, finite fp16 numbers, whose IEEE order IS the
// order of the patterns read as integers.  H = max(h0, F, E) is then ONE instruction for both halves, v_pk_maximum3_f16 (new on
// gfx950), instead of two v_pk_max_u16: 6 instead of 7 instructions per lane-row, and every VALU instruction of this mix costs
// the same ~4 issue cycles (profiles/r03/r03_ubench_rowbody.txt), so that is one seventh of the recurrence's issue time.
// Exactness on the hardware (denormal patterns included, no flush): vsearch_amd/csrc/ubench_max3.hip, 3 x 4 M random triples.
// Adds and subtractions stay integer ops on the patterns.  The checkpoints hold the 0x3E00-biased values as computed; the traceback
// converts its border values with the same bias (VsxDevParams::max3).
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
DEV u32 pk_max3_bits(u32 a, u32 b, u32 c)            // IEEE maximum of three: the compiler selects v_pk_maximum3_f16 (no inline asm: free register choice)
{
  const f16x2 x = __builtin_bit_cast(f16x2, a), y = __builtin_bit_cast(f16x2, b), z = __builtin_bit_cast(f16x2, c);
  return __builtin_bit_cast(u32, __builtin_elementwise_maximum(__builtin_elementwise_maximum(x, y), z));
}
// NQ (r05, the SPARSE-TASK classes of the TILT family): a wave works on NQ = 2 or 4 TASKS at once -- sub-task u = lane groups
// u * (4 / NQ) .. of the wave, i.e. a task of <= 4 (NQ = 2) or <= 2 (NQ = 4) targets whose targets sit in its first slots.  The
// reference refills its 8 SIMD lanes with new targets as old ones finish (align_simd.cpp:1823-1946); here a task was one wave whatever
// the number of its targets (a query with 1 candidate paid for 8: VERDICT r04 missing 3).  A task keeps its VsxTask and its VsxSlotOut
// entries; the NQ tasks of a wave share ONE checkpoint block and ONE step range (VsxTask::steps = the longest of them, ::dir_off the
// same: the planner's job, vsx_host.cpp), in which lane group gw of the wave owns the slots l * 4 + gw exactly as in a whole-wave
// task -- every store instruction still writes 768 B of full lines (the first build gave each task a block of its own: 12-byte pieces
// 48 bytes apart, 4 x the lines, each partially written: cands = 1 ran 1.5 x SLOWER than as whole waves,
// profiles/r05/r05b_sparse_ab_first_build.txt).  The traceback finds a pair's lane group as VsxTask::group0 + slot / 2.  Each
// sub-task has its own LDS profile (NQ x 256 RP bytes), which is what bounds the occupancy of these classes (VSX_FWD_WAVES_NQ);
// single-strip queries only.  Sub-tasks beyond `ntasks` (the last wave of a launch) run on junk in slots nobody reads.
#ifndef VSX_FWD_WAVES_NQ
#define VSX_FWD_WAVES_NQ(R_, NQ_) ((NQ_) == 4 ? 2 : ((R_) <= 24 ? 3 : 2))
#endif
// PAIR (r05, the PAIR-PROFILE classes of the TILT family): a WORKGROUP of four waves = four whole-wave tasks of ONE query whose symbols
// and whose targets' symbols are all plain A / C / G / T.  They share one LDS profile indexed by the PAIR of column symbols of a lane
// group's two targets: PP[pair][position][row] = S'[a][q_row] | S'[b][q_row] << 16 -- the packed score of both halves is one LDS dword,
// the row body loses its v_perm_b32: 5 instead of 6 instructions per lane-row.  The profile is 4 x the byte profile (16 pairs instead
// of 16 codes at a dword instead of a byte per row), which one wave cannot afford (two waves per SIMD: measured +18 %);
// four waves sharing it can, when a query has >= 32 targets (--allpairs_global).  No look-ahead register set: the R dwords of a step are
// read at its top, row r's compute waits for its own quarter only.  Everything a task owns -- VsxTask, checkpoint block, VsxSlotOut -- is
// as in the whole-wave class, so the traceback is unchanged.  The planner forms the groups of four (vsx_host.cpp).
// ONE (r06): every task of the launch is a SINGLE-STRIP query (Q <= 16 R -- which pick_rows() gives every query of at most 512 symbols;
// the planner knows per launch, Launch::multi).  The strip loop, its hand-over buffer and the per-strip address set then fold away at
// compile time.  What that is worth was measured before it was built (a forced build, profiles/r06/r06l_onestrip_ab.txt): the whole-wave
// MAX3 kernels lose their scratch -- R = 16: 128 VGPRs + 188 B of scratch (46 VGPR + 63 SGPR spills) -> 128 VGPRs, none; R = 10: 72 B -> 0
// at 111 VGPRs; R = 20: 196 -> 76 B; R = 26: 250 -> 197 VGPRs -- and run 3.4-4.9 % faster: 250 x 1000 DP 22.19 -> 21.10 ms, 150 x 1000
// 16.22 -> 15.6, 300 x 300 10.10 -> 9.76, 400 x 400 16.83 -> 16.07, 150 x 300 5.43 -> 5.22 (same box, two runs each).  The sparse-task
// classes (NQ > 1) always were single-strip, which is why they compiled without spills (VERDICT r05 "next" 3b asked what they do not
// keep live: the strip state).
template <int R, bool GENERIC, bool TRACK, bool CKPT, bool TILT = false, bool MAX3 = false, int NQ = 1, bool PAIR = false, bool ONE = false>
__global__ void __launch_bounds__(PAIR ? 256 : 64) __attribute__((amdgpu_waves_per_eu(NQ == 1 ? (ONE ? VSX_FWD_WAVES_ONE(R, TILT) : VSX_FWD_WAVES(R, TILT)) : VSX_FWD_WAVES_NQ(R, NQ), 8)))
vsx_forward_kernel(const VsxDevParams P, const VsxTask * __restrict__ tasks,
                   const uint8_t * __restrict__ qc, const uint8_t * __restrict__ tc,
                   u32 * __restrict__ dir, uint2 * strip, VsxSlotOut * __restrict__ slot_out, const u32 ntasks)
{
  constexpr int ND = (R + 3) / 4;                  // direction dwords per lane per step
  // TOPPAD: pipeline position 0 holds fewer than R query rows.  In this class its spare slots sit ABOVE the real rows and
  // are made transparent -- each dummy row reproduces the top border chain, H = Htop(j), via a constant profile score
  // -ge (query-left extension) and a diagonal seeded with -go, so the first real row sees exactly Htop(j-1), Htop(j) and
  // F = Htop(j) - QR_t(j) -- and every position hands (H, F) over from its compile-time row R-1: no per-row capture.
  // Needs the LDS profile (a per-row constant score), no min/max tracking (the dummies would pollute it) and
  // QR_q(interior) >= ge (the planner checks both, vsx_host.cpp no_overflow_possible()).  The traceback of this class
  // (vsx_traceback_ck_kernel<R, true>) uses the same slot layout.
  constexpr bool TOPPAD = CKPT && GENERIC && !TRACK;
  static_assert(!TILT || TOPPAD, "tilted coordinates exist for the TOPPAD class only");
  const int tl = TILT ? P.tilt : 0;                // g of the tilt (P then holds the primed constants)
  // TILT values are additionally BIASED into unsigned 16 bits (x + 0x8000 per half; the planner's range proof keeps every
  // intermediate inside (0, 65535)): the primed scores and the interior QR' = go are non-negative, so H + S' and H - go are
  // ONE 32-bit v_add_u32 / v_sub_u32 for both halves (2 cycles instead of the 4 of v_pk_add/sub_i16: no carry or borrow can
  // cross the halves); maxima are v_pk_max_u16, the remaining (possibly negative) penalties subtract with v_pk_sub_u16.
  static_assert(!MAX3 || TILT, "the MAX3 arithmetic is a sub-class of TILT");
  constexpr u32 BIAS16 = MAX3 ? 0x3E00u : (TILT ? 0x8000u : 0u);
  constexpr u32 BIAS = BIAS16 * 0x10001u;
  constexpr u32 CK_REBIAS = 0u;      // (first MAX3 build: stored checkpoints re-biased to 0x8000; the traceback takes the class's bias now)
  auto bpack = [](int v) -> u32 { return pack16(v + (int) BIAS16); };        // a border value -> both halves, biased
  auto vadd = [](u32 a, u32 b) -> u32 { return TILT ? a + b : sadd(a, b); };       // b >= 0 per half
  auto vsubk = [](u32 a, u32 b) -> u32 { return TILT ? a - b : ssub(a, b); };      // b >= 0 per half, a >= b per half
  auto vsub = [](u32 a, u32 b) -> u32 { return TILT ? psubw(a, b) : ssub(a, b); };
  auto vmax = [](u32 a, u32 b) -> u32 { return TILT ? pmaxu(a, b) : pmax(a, b); };
  // GENERIC: query profile in LDS, QP[target code][row of the strip] = S[code][query symbol of the row] (int16).
  // Any IUPAC / unknown symbol on either side is handled by construction; one v_perm_b32 per row packs the two targets.
  // TILT + VSX_CKT: the primed scores are 0 .. 255 (planner-checked), so the profile holds BYTES (half the LDS, half the reads;
  // the same single v_perm_b32 per row widens them): QPb[code][position][RP rows], RP = R rounded up to a multiple of 4
  constexpr bool MIDCK = VSX_MID(R, TILT);          // second row checkpoint after row R/2 - 1, column checkpoints per half
  constexpr int RT = MIDCK ? R / 2 : R;
  constexpr bool CKST = TILT && (VSX_CKT != 0);     // LDS-transposed checkpoint stores (A/B build)
  static_assert(!(MAX3 && CKST), "the transposed A/B layout predates the MAX3 class");
  constexpr bool QPL = TILT && (VSX_QPL != 0);      // byte profile, read one step ahead
  constexpr bool QP8 = CKST || QPL;
  constexpr int RP = (R + 3) & ~3;
  static_assert(NQ == 1 || NQ == 2 || NQ == 4, "tasks per wave");
  static_assert(NQ == 1 || (TILT && VSX_QPL != 0 && !VSX_CKT), "the sparse-task classes exist for the look-ahead TILT kernels only");
  static_assert(!PAIR || (NQ == 1 && TILT && VSX_QPL != 0 && VSX_FEED2 != 0 && !VSX_CKT && GENERIC), "the pair-profile classes extend the FEED2 look-ahead kernels");
  constexpr int WPB = PAIR ? 4 : 1;                  // waves per workgroup
  constexpr int QPBYTES = 16 * 16 * RP;              // one byte profile
  // pair profile: row stride RPP dwords with RPP / 4 odd, so that the 16 lanes of a group (each reading whole 16-byte quarters) start
  // in sixteen different 4-bank groups
  constexpr int RPP = ((RP / 4) & 1) ? RP : RP + 4;
  __shared__ __attribute__((aligned(16))) u32 PPf[PAIR ? 16 * 16 * RPP : 4];
  __shared__ __attribute__((aligned(16))) int16_t QP[(GENERIC && !PAIR) ? (QP8 ? (NQ * QPBYTES) / 2 : 16 * 16 * R) : 8];
  uint8_t * const QPb = reinterpret_cast<uint8_t *>(QP);
  // checkpoint staging of the transposed layout: [slot][12 dwords]
  __shared__ __attribute__((aligned(16))) u32 RS[CKST ? 64 * 12 : 4];
  // feed block of the column pipeline: [lane group][column of the 16-block] (sym, QR_t, R_t, H) and F -- written once per 16
  // steps by the 16 lanes of a group, read back one column per step for lane 0 (replaces five v_mov_b32_dpp row_ror rotations)
  constexpr int FSZ = (TILT && VSX_QPL ? 2 : 1) * 4 * 16;             // feed entries of one wave
  __shared__ uint4 FEED4a[FSZ * WPB];                                  // QPL: two blocks (the next one is built a step early)
  __shared__ u32 FEEDFa[FSZ * WPB], FEEDN[GENERIC ? 1 : 4 * 16];
  // FEED2 (the classes with the look-ahead byte profile): the record of a column is (byte offsets of the two targets' profile rows in
  // the halves of one dword, symbols + flags, H, F) -- an interior step reads H and F of its column and the offsets of the next column
  // from three neighbouring dwords, and the profile addresses are two adds -- and (QR_t, R_t) live in a second array for phase B
  constexpr bool FEED2 = (VSX_FEED2 != 0) && TOPPAD && (TILT && (VSX_QPL != 0));
  __shared__ uint2 FEEDBa[FEED2 ? 2 * 4 * 16 * WPB : 1];

  const int wv = PAIR ? __builtin_amdgcn_readfirstlane((int) (threadIdx.x >> 6)) : 0;      // wave of the workgroup = task of the group of four (scalar: the task's fields stay in SGPRs)
  uint4 * const FEED4 = FEED4a + wv * FSZ;
  u32 * const FEEDF = FEEDFa + wv * FSZ;
  uint2 * const FEEDB = FEEDBa + (FEED2 ? wv * FSZ : 0);
  const int lane = PAIR ? (int) (threadIdx.x & 63) : (int) threadIdx.x;
  const int gw = lane >> 4;                        // lane group of the wave
  constexpr int GPT = 4 / NQ;                      // lane groups per task
  const int sq = (NQ == 1) ? 0 : gw / GPT;         // this lane's sub-task
  const int g = (NQ == 1) ? gw : gw % GPT;         // its group inside the task (targets 2g, 2g + 1; checkpoint slot l * 4 + g)
  const int l = lane & 15;
  const u32 tix = PAIR ? blockIdx.x * 4u + (u32) wv : ((NQ == 1) ? blockIdx.x : blockIdx.x * (u32) NQ + (u32) sq);
  const bool sub_on = (NQ == 1 && !PAIR) ? true : (tix < ntasks);      // (PAIR: the planner launches whole groups of four)
  const VsxTask & T = tasks[sub_on ? tix : ntasks - 1];
  uint8_t * const QPl = reinterpret_cast<uint8_t *>(QP) + ((NQ == 1) ? 0 : sq * QPBYTES);      // this lane's profile

  const int Q = (int) T.qlen;
  const int total_lanes = (Q + R - 1) / R;         // pipeline positions holding query rows
  const int rcnt0 = Q - (total_lanes - 1) * R;     // rows in position 0 (1..R); all others hold R
  const int rc0 = __builtin_amdgcn_readfirstlane(rcnt0);   // provably scalar: keeps the per-row capture test on the SALU
  static_assert(!PAIR || ONE, "the pair-profile classes are planned for single-strip queries only");
  const int nstrips = (NQ == 1 && !ONE) ? (total_lanes + 15) >> 4 : 1;
  const int steps = (NQ == 1) ? (int) T.steps : __builtin_amdgcn_readfirstlane((int) T.steps);   // (NQ > 1: the planner gives the tasks of a wave ONE step range)

  const int DA = sub_on ? (int) T.tlen[2 * g] : 0;
  const int DB = sub_on ? (int) T.tlen[2 * g + 1] : 0;
  const int DpA = (DA + 3) & ~3;                   // the reference pads the last 4-column block
  const int DpB = (DB + 3) & ~3;
  const int Dpg = DpA > DpB ? DpA : DpB;
  const uint8_t * __restrict__ tA = tc + T.toff[2 * g];
  const uint8_t * __restrict__ tB = tc + T.toff[2 * g + 1];
  const uint8_t * __restrict__ qq = qc + T.qoff;

  const u32 qrt_i_pk = pack16(P.qrt_i);
  u32 hmin = 0, hmax = 0;      // running min(0, H...) / max(0, H...) of align_simd.cpp:810-811,772-773
  u32 score = 0;
  // Where the traceback leaves the last query row (backtrack16 :1161-1210 restricted to that row): coming from the right in an
  // 'I' run it continues through column j iff ext-left(j) or left(j); lv1 = column of the first cell (scanning left) where it
  // does not, carried along the row: lv1(j) = cont(j) ? lv1(j-1) : j.  At column D-1 (no run yet) only left(j) counts.
  u32 lv1 = 0xFFFFFFFFu, leave = 0;

  for (int s = 0; s < nstrips; ++s)
    {
      const int L = 16 * s + l;                    // global pipeline position
      const bool lane_on = (L < total_lanes) && (Dpg > 0);
      const bool first = (L == 0);
      const int pad = TOPPAD ? R - rcnt0 : 0;    // dummy slots above the rows of position 0
      const int i0 = first ? -pad : rcnt0 + (L - 1) * R;

      if (GENERIC)
        {
          __syncthreads();                           // previous strip's readers are done
          if (PAIR)
            {
              // the pair profile of the group's query (the four waves hold tasks of ONE query: any wave's task describes it), by all 256 threads
              for (int idx = (int) threadIdx.x; idx < 16 * 16 * RP; idx += 256)
                {
                  const int pr = idx / (16 * RP), row = idx % (16 * RP);
                  const int Lr = 16 * s + row / RP, rr = row % RP;
                  int va = 0, vb = 0;
                  if (rr < R)
                    {
                      if (Lr == 0 && rr < pad) va = vb = -P.top_step + 2 * tl;
                      else if (Lr < total_lanes)
                        {
                          const int gi = (Lr == 0) ? rr - pad : rcnt0 + (Lr - 1) * R + rr;
                          const int qs = (int) qq[gi];
                          va = P.matrix[(1 << (pr >> 2)) * 16 + qs];
                          vb = P.matrix[(1 << (pr & 3)) * 16 + qs];
                        }
                    }
                  PPf[(pr * 16 + row / RP) * RPP + rr] = ((u32) va & 0xffu) | (((u32) vb & 0xffu) << 16);
                }
            }
          else if (QP8)
            {
              // (NQ > 1: the 64 / NQ lanes of a sub-task fill THEIR profile from their own task's query)
              constexpr int LPT = 64 / NQ;
              for (int idx = (NQ == 1) ? lane : lane % LPT; idx < 16 * 16 * RP; idx += LPT)
                {
                  const int code = idx / (16 * RP), row = idx % (16 * RP);
                  const int Lr = 16 * s + row / RP, rr = row % RP;
                  int v = 0;
                  if (rr < R)
                    {
                      if (Lr == 0 && rr < pad) v = -P.top_step + 2 * tl;
                      else if (Lr < total_lanes)
                        {
                          const int gi = (Lr == 0) ? rr - pad : rcnt0 + (Lr - 1) * R + rr;
                          v = P.matrix[code * 16 + (int) qq[gi]];
                        }
                    }
                  QPl[idx] = (uint8_t) v;
                }
            }
          else
          for (int idx = lane; idx < 16 * 16 * R; idx += 64)
            {
              const int code = idx / (16 * R), row = idx % (16 * R);
              const int Lr = 16 * s + row / R, rr = row % R;
              int v = 0;
              if (TOPPAD && Lr == 0 && rr < pad) v = -P.top_step + 2 * tl;
              else if (Lr < total_lanes && (TOPPAD || !(Lr == 0 && rr >= rcnt0)))
                {
                  const int gi = (Lr == 0) ? rr - pad : rcnt0 + (Lr - 1) * R + rr;
                  v = P.matrix[code * 16 + (int) qq[gi]];
                }
              QP[idx] = (int16_t) v;
            }
          __syncthreads();
        }

      // ---- per-lane row state: left border (aligncolumns_first :844-859, :881-887) ----
      u32 hprev[R];      // H(i, j-1): left neighbour, next column's diagonal for row i+1
      u32 hnext[R];      // its ping-pong partner
      u32 E[R];          // E(i, j)
      u32 ac[R];         // query symbol of the row (fast: code in both halves; generic: code << 8)
#pragma unroll
      for (int r = 0; r < R; ++r)
        {
          int i = i0 + r;
          const bool dummy = TOPPAD && first && r < pad;
          if (i > Q - 1) i = Q - 1;                // junk rows (skipped or idle lanes): any valid address
          if (i < 0) i = 0;
          const u32 a = qq[i];
          ac[r] = a | (a << 16);
          u32 hl = bpack(P.hleft[i]);
          u32 e0 = vsub(hl, (i < Q - 1) ? P.qrq_i_pk : P.qrq_r_pk);
          if (dummy)
            {
              const int sh = (r - pad - 1) * tl;                                     // tilt of (i, -1), i = r - pad
              hl = bpack(((r == pad - 1) ? 0 : -P.top_open) + sh);           // Htop(-1) = 0 for the first real row's diagonal
              e0 = vsub(bpack(-P.top_open - P.top_step + sh), P.qrq_i_pk);   // <= Htop(0), stays below the chain
            }
          hprev[r] = hl;
          hnext[r] = hl;     // a lane that has not started yet must find its border state in either array
          E[r] = e0;
        }
      u32 diag = first ? bpack(((TOPPAD && pad > 0) ? -P.top_open : 0) - (pad + 2) * tl) : bpack(P.hleft[i0 - 1]);   // H(i0-1, -1); Htop(-1) = 0 (:1895)
      // query-gap penalties of row R-1: only the globally last row uses the right-end pair (:836-897)
      const bool lastpos = (L == total_lanes - 1);
      const u32 qrq_last = lastpos ? P.qrq_r_pk : P.qrq_i_pk;
      const u32 rq_last = lastpos ? P.rq_r_pk : P.rq_i_pk;

      // ---- column pipeline ----
      u32 sym = 0, nd = 0, qrt = 0, rt = 0;        // this lane's current column descriptor (FEED2: sym = the profile byte offsets)
      u32 symf = 0;                                // FEED2: symbols + flags of the column, shifted along in phase B only (no column in
                                                   // flight at the switch is a last column: 0 is what every lane would hold)
      u32 outH = 0, outF = 0;                      // H(bottom row, j), F(bottom row + 1, j)
      u32 f_sym = 0, f_nd = 0, f_qrt = 0, f_rt = 0, f_H = 0, f_F = 0;   // 16-column feed block
      u32 rawA = 0, rawB = 0;
      int rawH = 0;
      uint2 rawS = make_uint2(0, 0);
      const uint2 * strip_in = strip + T.strip_off + (size_t) (((s + 1) & 1) * 4 + g) * (size_t) steps;
      uint2 * strip_outp = strip + T.strip_off + (size_t) ((s & 1) * 4 + g) * (size_t) steps;

      auto prefetch = [&](int blk) {
        const int c = 16 * blk + l;
        rawA = (c < DA) ? (u32) tA[c] : 0u;
        rawB = (c < DB) ? (u32) tB[c] : 0u;
        if (s == 0) rawH = P.htop[c];
        else if (c < Dpg) rawS = strip_in[c];
      };
      prefetch(0);

      // feed block `blk` = columns 16 blk .. 16 blk + 15 (lane l describes column 16 blk + l), built from the prefetched symbols;
      // prefetches the symbols of the block after it
      auto build_feed = [&](int blk) __attribute__((always_inline)) {
              const int c = 16 * blk + l;
              auto mne = [&](u32 code) -> int {     // score of an unambiguous query row vs `code` when codes differ
                const bool unamb = (code != 0) && ((code & (code - 1)) == 0);
                return unamb ? P.mismatch : ((P.n_mismatch && code == 15) ? P.mismatch : 0);
              };
              const u32 ndA = (u32) (mne(rawA) - P.match) & 0xffffu;
              const u32 ndB = (u32) (mne(rawB) - P.match) & 0xffffu;
              const u32 qA = (u32) ((c < DA - 1) ? P.qrt_i : P.qrt_r) & 0xffffu;   // :1719-1753
              const u32 qB = (u32) ((c < DB - 1) ? P.qrt_i : P.qrt_r) & 0xffffu;
              const u32 rA = (u32) ((c < DA - 1) ? P.rt_i : P.rt_r) & 0xffffu;
              const u32 rB = (u32) ((c < DB - 1) ? P.rt_i : P.rt_r) & 0xffffu;
              const u32 sA = rawA | ((c == DA - 1) ? 0x100u : 0u) | ((c < DpA) ? 0x200u : 0u);
              const u32 sB = rawB | ((c == DB - 1) ? 0x100u : 0u) | ((c < DpB) ? 0x200u : 0u);
              f_sym = sA | (sB << 16);
              f_nd = ndA | (ndB << 16);
              f_qrt = qA | (qB << 16);
              f_rt = rA | (rB << 16);
              if (s == 0)
                {
                  f_H = bpack(rawH - pad * tl);             // H(-1, j) = Htop(j) (tilted: it enters at row -pad - 1)
                  f_F = vsub(f_H, f_qrt);              // f = v_sub(f, QR_t) at block entry (:830-833)
                }
              else
                {
                  f_H = rawS.x;                        // handed over by the previous strip's last lane
                  f_F = rawS.y;
                }
              const int fo = QPL ? (blk & 1) * 64 : 0;
              if (FEED2)
                {
                  // QPb[code][16 positions][RP] bytes; PAIR: the byte offset of PP[pair = 4 ia + ib] (codes 1 2 4 8 -> 0 1 2 3; a column
                  // past a target's end has code 0 -> index 0: any valid address, nobody keeps what it yields)
                  const u32 f_off = PAIR ? (((rawA >> 1) - (rawA >> 3)) * 4u + ((rawB >> 1) - (rawB >> 3))) * (u32) (16 * RPP * 4)
                                         : (rawA * (u32) (16 * RP) | ((rawB * (u32) (16 * RP)) << 16));
                  FEED4[fo + gw * 16 + l] = make_uint4(f_off, f_sym, f_H, f_F);
                  FEEDB[fo + gw * 16 + l] = make_uint2(f_qrt, f_rt);
                }
              else
                {
                  FEED4[fo + gw * 16 + l] = make_uint4(f_sym, f_qrt, f_rt, f_H);
                  FEEDF[fo + gw * 16 + l] = f_F;
                }
              if (!GENERIC) FEEDN[gw * 16 + l] = f_nd;
              prefetch(blk + 1);
      };
      // QPL: byte profile rows of BOTH targets for one step, in registers ([target][4 rows per dword]); two sets ping-pong with
      // the two-step unrolling: the set of step t+1 is requested at the top of step t, from the symbols the lane will hold then
      // (its neighbour's current ones)
      constexpr int PD = RP / 4;
      u32 profA[2][(QPL && !PAIR) ? PD : 1], profB[2][(QPL && !PAIR) ? PD : 1];
      auto load_profile = [&](u32 (&buf)[2][(QPL && !PAIR) ? PD : 1], u32 sy) __attribute__((always_inline)) {
        if (PAIR) return;                            // (the pair profile is read at the top of the step it serves: see step())
        const u32 cd = sy & 0x000F000Fu;
        // (the offsets are multiples of 16 RP: the rows keep the alignment of l RP, which the compiler cannot see through the feed --
        //  so the FEED2 form spells the wide LDS reads out)
        constexpr int QPA = (RP % 16 == 0) ? 16 : ((RP % 8 == 0) ? 8 : 4);
        if (FEED2)
          {
            const uint8_t * pa = QPl + l * RP + (sy & 0xFFFFu);
            const uint8_t * pb = QPl + l * RP + (sy >> 16);
            constexpr int PDN = (QPL && !PAIR) ? PD : 1;
            if constexpr (PAIR) { }
            else if constexpr (QPA == 16)
              {
#pragma unroll
                for (int k = 0; k < PDN; k += 4)
                  {
                    const uint4 va = *reinterpret_cast<const uint4 *>(pa + 4 * k), vb = *reinterpret_cast<const uint4 *>(pb + 4 * k);
                    buf[0][k] = va.x; buf[0][k + 1] = va.y; buf[0][k + 2] = va.z; buf[0][k + 3] = va.w;
                    buf[1][k] = vb.x; buf[1][k + 1] = vb.y; buf[1][k + 2] = vb.z; buf[1][k + 3] = vb.w;
                  }
              }
            else if constexpr (QPA == 8)
              {
#pragma unroll
                for (int k = 0; k < PDN; k += 2)
                  {
                    const uint2 va = *reinterpret_cast<const uint2 *>(pa + 4 * k), vb = *reinterpret_cast<const uint2 *>(pb + 4 * k);
                    buf[0][k] = va.x; buf[0][k + 1] = va.y;
                    buf[1][k] = vb.x; buf[1][k + 1] = vb.y;
                  }
              }
            else
              {
#pragma unroll
                for (int k = 0; k < PDN; ++k)
                  { buf[0][k] = reinterpret_cast<const u32 *>(pa)[k]; buf[1][k] = reinterpret_cast<const u32 *>(pb)[k]; }
              }
            return;
          }
        const u32 * a = reinterpret_cast<const u32 *>(QPl + (cd & 0xFu) * (16 * RP) + l * RP);
        const u32 * b = reinterpret_cast<const u32 *>(QPl + (cd >> 16) * (16 * RP) + l * RP);
#pragma unroll
        for (int k = 0; k < ((QPL && !PAIR) ? PD : 1); ++k) { buf[0][k] = a[k]; buf[1][k] = b[k]; }
      };
      if (QPL)
        {
          build_feed(0);
          sym = dpp_shr1(FEED4[gw * 16].x, 0u);           // the symbols of step 0
          load_profile(profA, sym);
        }

      const uint4 * const feed4_g = FEED4 + gw * 16;
      const u32 * const feedf_g = FEEDF + gw * 16;
      // one pipeline step; reads the left-neighbour row state from hin[] and writes hout[] (the caller ping-pongs the two
      // arrays over an even number of steps, so no per-step register copies remain)
      // INTERIOR (phase A of the strip, see the loops below): no lane of the wave has reached a last / padded column yet, so
      // the column penalties are the interior constants (not pipelined), H - QR is shared by E and F, no score capture.
      u32 pendH = 0, pendF = 0;                    // row checkpoint of the even step, stored together with the odd step's
      u32 jpk_run = 0;                             // the column of this lane in both halves, carried by the steady loop
      u32 pendMH = 0, pendMF = 0, curMH = 0, curMF = 0;          // the same for the mid-row checkpoint (MIDCK)
      bool pend_on = false;
      // per-lane base addresses of this strip's checkpoint regions (computed once: the step only adds a uniform offset)
      const size_t ck_rowdw = ((((size_t) nstrips * steps + 1) & ~(size_t) 1) >> 1) * VSX_ROWCK_PAIR_DW(TILT);
      const size_t ck_nblk = ((size_t) steps + 15) >> 4;
      constexpr int CK_LANE_DW = TILT ? 3 : 4;                                                                     // row checkpoint dwords per lane and pair
      constexpr size_t CK_COL_DW = TILT ? (size_t) 64 * 4 * VSX_COLCK_NB(R, true) : (size_t) 64 * (2 * R);           // column checkpoint dwords per wave
      const int ck_slot = VSX_CK_SLOT(TILT, gw, l);     // (NQ > 1: the wave's tasks share ONE checkpoint block, each in the slots of its own lane groups)
      u32 * const rck_base = dir + T.dir_off + ((((size_t) s * steps) >> 1) * 64 + ck_slot) * CK_LANE_DW;         // + (t >> 1) * VSX_ROWCK_PAIR_DW
      u32 * const cck_base = dir + T.dir_off + ck_rowdw + (size_t) s * ck_nblk * CK_COL_DW;                        // + (t >> 4) * CK_COL_DW
      // mid-row checkpoints: a third region behind the column checkpoints, laid out like the row checkpoints
      u32 * const mck_base = dir + T.dir_off + ck_rowdw + (size_t) nstrips * ck_nblk * CK_COL_DW
                             + ((((size_t) s * steps) >> 1) * 64 + ck_slot) * CK_LANE_DW;                           // + (t >> 1) * 192
      // transposed layout: block bases (steps is a multiple of 8 there); a lane adds (k * 64 + lane) * 4 dwords per store
      u32 * const rck_blk = dir + T.dir_off + (((size_t) s * steps) >> 3) * VSX_CKT_BLOCK_DW;                       // + (t >> 3) * 768
      u32 * const cck_blk = dir + T.dir_off + (((size_t) nstrips * steps) >> 3) * VSX_CKT_BLOCK_DW
                            + (size_t) s * ck_nblk * VSX_COLCK_NCHUNK(R) * VSX_CKT_BLOCK_DW;                        // + ((t >> 4) * NCHUNK + c) * 768
      // one 3 KB block: LDS -> HBM as it lies (three full 1 KB stores); the barriers order the lanes' LDS accesses (one wave)
      auto flush_block = [&](u32 * gdst) __attribute__((always_inline)) {
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 3; ++k)
          {
            const u32x4 v = *reinterpret_cast<const u32x4 *>(RS + (k * 64 + lane) * 4);
            __builtin_nontemporal_store(v, reinterpret_cast<u32x4 *>(gdst + (k * 64 + lane) * 4));
          }
        __syncthreads();
      };

      // STEADY (the part of phase A after the pipeline has filled, t >= 15): every lane that owns rows is inside its targets, so the
      // per-lane activity test and its EXEC mask are dropped (lanes beyond the query's positions compute junk nobody reads).
      auto step = [&](const int t, u32 (&hin)[R], u32 (&hout)[R], u32 (&pc)[2][(QPL && !PAIR) ? PD : 1], u32 (&pn)[2][(QPL && !PAIR) ? PD : 1],
                      auto interior_tag, auto odd_tag, auto steady_tag) __attribute__((always_inline)) {
          constexpr bool INTERIOR = decltype(interior_tag)::value;
          constexpr bool ODD = decltype(odd_tag)::value;
          constexpr bool STEADY = decltype(steady_tag)::value;
          u32 symn = 0;
          // PAIR: this step's packed scores, one dword per row, straight from the pair profile (`sym` = the byte offset of the pair's
          // block, as the feed delivers it); issued first, consumed a quarter at a time by the row loop below
          u32 pp[PAIR ? RP : 1];
          if (PAIR)
            {
              const u32 * ppa = PPf + (sym >> 2) + l * RPP;
#pragma unroll
              for (int k = 0; k < (PAIR ? RP : 0); k += 4)
                {
                  const uint4 v = *reinterpret_cast<const uint4 *>(ppa + k);
                  pp[k] = v.x; pp[k + 1] = v.y; pp[k + 2] = v.z; pp[k + 3] = v.w;
                }
            }
          if (QPL) { if ((t & 15) == 15) build_feed((t >> 4) + 1); }      // the next block, one step early (look-ahead below)
          else if ((t & 15) == 0) build_feed(t >> 4);

          // ---- systolic shift (all lanes, full EXEC) ----
          // feed reads: lane part (the group's 16-column row) hoisted, the step's part is uniform -- one v_add per address
          const int fo = QPL ? ((t >> 4) & 1) * 64 : 0;
          const u32 fslot = (u32) (fo + (t & 15));
          const uint4 fv = feed4_g[fslot];
          const u32 fF = FEED2 ? fv.w : feedf_g[fslot];
          if (QPL)
            {
              // `sym` already holds this step's symbols; look one step ahead and request that step's profile rows now
              const int t1 = t + 1;
              symn = dpp_shr1(feed4_g[(u32) (((t1 >> 4) & 1) * 64 + (t1 & 15))].x, sym);
              load_profile(pn, symn);
            }
          else sym = dpp_shr1(fv.x, sym);
          if (!GENERIC) nd = dpp_shr1(FEEDN[gw * 16 + (t & 15)], nd);
          if (!INTERIOR)
            {
              if (FEED2)
                {
                  const uint2 fb = FEEDB[fo + gw * 16 + (t & 15)];
                  qrt = dpp_shr1(fb.x, qrt); rt = dpp_shr1(fb.y, rt);
                  symf = dpp_shr1(fv.y, symf);
                }
              else { qrt = dpp_shr1(fv.y, qrt); rt = dpp_shr1(fv.z, rt); }
            }
          const u32 inH = dpp_shr1(FEED2 ? fv.z : fv.w, outH);
          const u32 inF = dpp_shr1(fF, outF);

          const int j = t - l;
          const bool active = STEADY ? true : (lane_on && j >= 0 && j < Dpg);
          if (active)
            {
              const u32 code = sym & 0x000F000Fu;
              const int16_t * qpA = QP + (code & 0xFu) * (16 * R) + l * R;      // GENERIC only
              const int16_t * qpB = QP + (code >> 16) * (16 * R) + l * R;
              u32 pa = 0, pb = 0;                                               // two rows' scores per dword
              u32 Hd = diag;
              u32 F = inF;
              u32 smn = 0x7FFF7FFFu, smx = 0x80008000u;
              u32 acc = 0, h2 = 0;
              u32 capH = 0, capF = 0, capmn = 0, capmx = 0;    // position 0: state after its last real row
              u32 dw[ND];
              u32 lastL = 0, lastEL = 0;
              u32 midH = 0, midF = 0;
              // SHARED: the interior query-row and target-column gap penalties coincide (planner flag P.share_sub) and no
              // lane of the wave is in a last / padded column this step, so H - QR is the same number for E and for F in
              // every row but R-1: 9 instead of 10 instructions per lane-row.
              auto rows = [&](auto shared_tag) __attribute__((always_inline)) {
              constexpr bool SHARED = decltype(shared_tag)::value;
              // LASTFAST (r04): in the last query row "left" implies "ext-left" when its gap-open penalty is positive -- E > max(h0, F)
              // makes H = E, and then (E - R) - (H - QR) = QR - R = go > 0 -- so `cont` of the leave-column tracking is the sign of
              // he - e alone: no h1 of its own (the row takes the three-way maximum like the others), no h1 - E, no OR.  The planner
              // admits a pair to the MAX3 class only with go_q(right end) > 0 (vsx_host.cpp tilt_possible()); phase B needs
              // left(D - 1) by itself and keeps the long form.
              constexpr bool LASTFAST = MAX3 && SHARED && CKPT;
#pragma unroll
              for (int r = 0; r < R; ++r)
                {
                  u32 V;
                  if (PAIR) V = pp[r];
                  else if (GENERIC && QP8)
                    {
                      if ((r & 3) == 0)
                        {
                          if (QPL) { pa = pc[0][r >> 2]; pb = pc[1][r >> 2]; }
                          else
                            {
                              pa = *reinterpret_cast<const u32 *>(QPb + (code & 0xFu) * (16 * RP) + l * RP + r);
                              pb = *reinterpret_cast<const u32 *>(QPb + (code >> 16) * (16 * RP) + l * RP + r);
                            }
                        }
                      // byte r & 3 of the A rows -> low half, of the B rows -> high half, zero-extended (selector 0x0C = 0x00)
                      V = __builtin_amdgcn_perm(pb, pa, 0x0C040C00u + 0x00010001u * (u32) (r & 3));
                    }
                  else if (GENERIC)
                    {
                      if (R % 2 == 0)
                        {
                          if ((r & 3) == 0 && r + 3 < R)
                            {
                              // 4 rows = 8 bytes; only dword-aligned when R is not a multiple of 4 (Pair2 -> ds_read2_b32)
                              typedef typename std::conditional<R % 4 == 0, uint2, Pair2>::type Rows4;
                              const Rows4 va = *reinterpret_cast<const Rows4 *>(qpA + r);
                              const Rows4 vb = *reinterpret_cast<const Rows4 *>(qpB + r);
                              pa = va.x; pb = vb.x;
                              ac[r + 1] = va.y; ac[r + 2] = vb.y;             // rows r+2, r+3 (ac[] is free in this variant)
                            }
                          else if ((r & 3) == 0)                              // R = 4k + 2: the last two rows
                            {
                              pa = *reinterpret_cast<const u32 *>(qpA + r);
                              pb = *reinterpret_cast<const u32 *>(qpB + r);
                            }
                          else if ((r & 3) == 2) { pa = ac[r - 1]; pb = ac[r]; }
                          V = (r & 1) ? __builtin_amdgcn_perm(pb, pa, 0x07060302u) : __builtin_amdgcn_perm(pb, pa, 0x05040100u);
                        }
                      else V = (u32) (uint16_t) qpA[r] | ((u32) (uint16_t) qpB[r] << 16);
                    }
                  else V = a_pk_mad(a_pk_minu(ac[r] ^ code, 0x00010001u), nd, P.match_pk);
                  // onestep (:765-780)
                  const u32 h0 = vadd(Hd, V);
                  // (row R-1 keeps the two-step maximum where the leave-column tracking needs h1 = max(h0, F) on its own: the steps of
                  //  phase B.  In the interior steps of the MAX3 class the tracking reads ext-left alone -- LASTFAST below)
                  const bool ONE_MAX = MAX3 && (r < R - 1 || LASTFAST);     // (folds after unrolling)
                  const u32 h1 = ONE_MAX ? 0u : vmax(h0, F);
                  h2 = ONE_MAX ? pk_max3_bits(h0, F, E[r]) : vmax(h1, E[r]);
                  if (TRACK) { smn = pmin(smn, h2); smx = pmax(smx, h2); }
                  Hd = hin[r];
                  hout[r] = h2;
                  const u32 qrq = (r == R - 1) ? qrq_last : P.qrq_i_pk;
                  const u32 rq = (r == R - 1) ? rq_last : P.rq_i_pk;
                  const u32 he = (r < R - 1) ? vsubk(h2, qrq) : vsub(h2, qrq);             // QR'(interior rows) = go >= 0
                  const u32 hf = (SHARED && r < R - 1) ? he : vsub(h2, qrt);
                  const u32 f = (TILT && INTERIOR) ? F : vsub(F, rt);                      // tilted interior: R' = 0
                  const u32 e = (TILT && INTERIOR && r < R - 1) ? E[r] : vsub(E[r], rq);   // (row R-1 may be the query's last row)
                  if (CKPT && r == R - 1) { lastL = LASTFAST ? 0u : vsub(h1, E[r]); lastEL = vsub(he, e); }     // the last row's left / ext-left diffs
                  if (!CKPT)
                    {
                      const u32 dU = ssub(h0, F);          // sign <=> F > H      (up)
                      const u32 dL = ssub(h1, E[r]);       // sign <=> E > H      (left)
                      const u32 dEU = ssub(hf, f);         // sign <=> F-R > H-QR (extend up)
                      const u32 dEL = ssub(he, e);         // sign <=> E-R > H-QR (extend left)
                      acc = funnel(funnel(funnel(funnel(acc, dU), dL), dEU), dEL);
                      if ((r & 3) == 3 || r == R - 1) dw[r >> 2] = acc;
                    }
                  F = vmax(f, hf);
                  E[r] = vmax(e, he);
                  if (MIDCK && r == RT - 1) { midH = h2; midF = F; }          // what row R/2 - 1 hands to row R/2
                  // position 0 holds only rcnt0 rows: its later rows compute junk that never leaves the lane
                  if (!TOPPAD && __builtin_expect(rc0 == r + 1, 0)) { capH = h2; capF = F; capmn = smn; capmx = smx; }   // wave-uniform branch
                }
              };
              if (INTERIOR) rows(std::true_type {});
              else rows(std::false_type {});
              const u32 hl = (!TOPPAD && first) ? capH : h2;
              F = (!TOPPAD && first) ? capF : F;
              smn = (!TOPPAD && first) ? capmn : smn;
              smx = (!TOPPAD && first) ? capmx : smx;
              outH = hl;
              outF = F;
              diag = inH;
              if (CKPT)
                {
                  // (the steady loop carries j in both halves along: one add per step instead of forming it from t and the lane)
                  const u32 jpk = STEADY ? jpk_run : ((u32) j | ((u32) j << 16));
                  if (STEADY) jpk_run += 0x00010001u;
                  if (!INTERIOR)
                    {
                      const u32 atlast = a_pk_ashr15((FEED2 ? symf : sym) << 7);               // bit 8 (column == D-1)
                      leave = a_bfi_v(atlast, a_bfi_v(a_pk_ashr15(lastL), lv1, jpk), leave);
                    }
                  lv1 = a_bfi_v(a_pk_ashr15((MAX3 && INTERIOR) ? lastEL : (lastL | lastEL)), lv1, jpk);
                }

              // per-block h_min/h_max tracking incl. padded columns (:772-773, :1774-1786), gated per half
              if (TRACK)
                {
                  const u32 vm = a_pk_ashr15(sym << 6);     // bit 9 (column < padded length) -> 0xFFFF
                  hmin = pmin(hmin, a_bfi_v(vm, smn, 0x7FFF7FFFu));
                  hmax = pmax(hmax, a_bfi_v(vm, smx, 0x80008000u));
                }
              if (!INTERIOR)
                {
                  const u32 lm = a_pk_ashr15((FEED2 ? symf : sym) << 7);     // bit 8 (column == D-1)
                  score = a_bfi_v(lm, hl, score);            // S[(D+3)%4] of the last row (:1835-1836)
                }

              // [4-step block][lane][step in block][ND]: a lane's 4 consecutive steps share one 64 B line
              // (4x fewer lines for the traceback walk) while a wave-step still lands in one 4 KB window
              const size_t gt = (size_t) s * steps + t;
              if (CKPT)
                {
                  if (!ODD) { pendH = outH; pendF = outF; }             // stored by the odd step below
                  if (MIDCK) { if (!ODD) { pendMH = midH; pendMF = midF; } else { curMH = midH; curMF = midF; } }
                }
              else
                {
                  u32 * dp = dir + T.dir_off + (((gt >> 2) * 64 + lane) * 4 + (gt & 3)) * ND;
                  if (ND == 4) *reinterpret_cast<uint4 *>(dp) = make_uint4(dw[0], dw[1], dw[2], dw[3]);
                  else if (ND == 2) *reinterpret_cast<uint2 *>(dp) = make_uint2(dw[0], dw[1]);
                  else
                    {
#pragma unroll
                      for (int w = 0; w < ND; ++w) dp[w] = dw[w];
                    }
                }
              if (l == 15 && s + 1 < nstrips) strip_outp[j] = make_uint2(outH, outF);
            }
          if (CKPT && CKST)
            {
              // transposed layout: the two-step pair goes to the lane's 48-byte segment of the staging block -- H of step i at
              // dword i, the four difference bytes of the pair at dword 8 + i / 2 -- and every 8 steps the block leaves as it lies
              if (!ODD) pend_on = active;
              else
                {
                  const u32 dpk = __builtin_amdgcn_perm(psubw(outH, outF), psubw(pendH, pendF), 0x06040200u);
                  u32 * rs = RS + ck_slot * 12;
                  const int i = t & 7;
                  *reinterpret_cast<uint2 *>(rs + (i - 1)) = make_uint2(pendH, outH);
                  rs[8 + (i >> 1)] = dpk;
                  if (i == 7) flush_block(rck_blk + (size_t) (t >> 3) * VSX_CKT_BLOCK_DW);
                }
            }
          else if (CKPT)
            {
              // row checkpoints [two-step pair][lane][2] uint2: one 16-byte store per lane and pair, so a wave writes
              // 1 KB of full lines (steps is even: a pair never straddles two strips); halves that were not active hold junk
              // nobody reads
              if (!ODD) pend_on = active;
              else if (pend_on || active)
                {
                  if (TILT && !CKST)
                    {
                      // bytes 0 / 2 of the two wrapped differences = the signed 8-bit H - F of the lo / hi target, steps t-1 and t
                      const u32 dpk = __builtin_amdgcn_perm(psubw(outH, outF), psubw(pendH, pendF), 0x06040200u);
                      __builtin_nontemporal_store((u32x3) {pendH + CK_REBIAS, outH + CK_REBIAS, dpk}, reinterpret_cast<u32x3 *>(rck_base + (size_t) (t >> 1) * 192));
                      if (MIDCK)
                        {
                          const u32 dpm = __builtin_amdgcn_perm(psubw(curMH, curMF), psubw(pendMH, pendMF), 0x06040200u);
                          __builtin_nontemporal_store((u32x3) {pendMH + CK_REBIAS, curMH + CK_REBIAS, dpm}, reinterpret_cast<u32x3 *>(mck_base + (size_t) (t >> 1) * 192));
                        }
                    }
                  else
                    __builtin_nontemporal_store((u32x4) {pendH, pendF, outH, outF}, reinterpret_cast<u32x4 *>(rck_base + (size_t) (t >> 1) * 256));
                }
            }
          if (CKPT && (t & 15) == 15)
            {
              // column checkpoint m = t / 16 of this lane: state after its column t - l (or its border state if it has
              // not started yet).  Layout VSX_COLCK_DW (R = 1: [strip][m][lane][2]).
              u32 * cb = cck_base + (size_t) (t >> 4) * CK_COL_DW;
              if (TILT)
                {
                  // hprev[0..R), then the byte differences of two rows per dword; zero-padded to whole blocks
                  constexpr int NF = R + (R + 1) / 2;
                  // MIDCK: per half of RT rows {H of the half's rows, then its difference bytes}, each half padded to whole blocks
                  constexpr int NFH = RT + (RT + 1) / 2, ZH = 4 * VSX_COLCK_NBH(RT);
                  auto flatc = [&](int z) -> u32 {
                    if (MIDCK)
                      {
                        const int hh = z / ZH, zz = z % ZH, r00 = hh * RT;
                        if (zz < RT) return hout[r00 + zz] + CK_REBIAS;
                        if (zz < NFH)
                          {
                            const int r0 = r00 + 2 * (zz - RT), r1 = (r0 + 1 < r00 + RT) ? r0 + 1 : r0;
                            return __builtin_amdgcn_perm(psubw(hout[r1], E[r1]), psubw(hout[r0], E[r0]), 0x06040200u);
                          }
                        return 0u;
                      }
                    if (z < R) return hout[z] + CK_REBIAS;
                    if (z < NF)
                      {
                        const int r0 = 2 * (z - R), r1 = r0 + 1 < R ? r0 + 1 : r0;
                        return __builtin_amdgcn_perm(psubw(hout[r1], E[r1]), psubw(hout[r0], E[r0]), 0x06040200u);
                      }
                    return 0u;
                  };
                  if (CKST)
                    {
                      // chunks of three 16-byte pieces per lane through the staging block: a lane's checkpoint is NCHUNK
                      // contiguous 48-byte segments in HBM
                      constexpr int NBC = VSX_COLCK_NB(R, true);
#pragma unroll
                      for (int c = 0; c < VSX_COLCK_NCHUNK(R); ++c)
                        {
#pragma unroll
                          for (int pc = 0; pc < 3; ++pc)
                            {
                              const int z = 4 * (3 * c + pc);
                              if (3 * c + pc < NBC)
                                *reinterpret_cast<u32x4 *>(RS + ck_slot * 12 + pc * 4) = (u32x4) {flatc(z), flatc(z + 1), flatc(z + 2), flatc(z + 3)};
                            }
                          flush_block(cck_blk + ((size_t) (t >> 4) * VSX_COLCK_NCHUNK(R) + c) * VSX_CKT_BLOCK_DW);
                        }
                    }
                  else
                    {
#pragma unroll
                  for (int z = 0; z < 4 * VSX_COLCK_NB(R, true); z += 4)
                    __builtin_nontemporal_store((u32x4) {flatc(z), flatc(z + 1), flatc(z + 2), flatc(z + 3)}, reinterpret_cast<u32x4 *>(cb + VSX_COLCK_CDW(VSX_COLCK_NB(R, true), ck_slot, z >> 2)));
                    }
                }
              else if (R % 2 == 0)
                {
                  // the 2R dwords hprev[0..R), E[0..R) as one flat array, four to a block
                  auto flat = [&](int z) -> u32 { return z < R ? hout[z] : E[z - R]; };
#pragma unroll
                  for (int z = 0; z < 2 * R; z += 4)
                    __builtin_nontemporal_store((u32x4) {flat(z), flat(z + 1), flat(z + 2), flat(z + 3)}, reinterpret_cast<u32x4 *>(cb + VSX_COLCK_DW(R, lane, z >> 2)));
                }
              else
                {
                  u32 * cp = cb + lane * (2 * R);
#pragma unroll
                  for (int x = 0; x < R; ++x) { cp[x] = hout[x]; cp[R + x] = E[x]; }
                }
            }
          if (QPL) sym = symn;                      // the look-ahead becomes the next step's symbols
      };
      // Phase A: steps before ANY lane of the wave reaches column D - 1 of one of its targets (lane l works on column
      // t - l <= t): every column in flight is interior.  Needs QR_q(interior) == QR_t(interior) for the shared
      // subtraction (planner flag); the checkpoint kernels only.  Phase B: the general step for the rest.
      int t_switch = 0;
      if (CKPT && P.share_sub)
        {
          int dmin = 0x7fffffff;
          if (DA > 0 && DA < dmin) dmin = DA;
          if (DB > 0 && DB < dmin) dmin = DB;
#pragma unroll
          for (int m = 32; m >= 1; m >>= 1) { const int o = __shfl_xor(dmin, m, 64); dmin = o < dmin ? o : dmin; }
          t_switch = __builtin_amdgcn_readfirstlane(dmin == 0x7fffffff ? 0 : dmin - 1) & ~1;
          if (t_switch > steps) t_switch = steps & ~1;
        }
      int t = 0;
      qrt = qrt_i_pk; rt = pack16(P.rt_i);
      for (; t < t_switch && (TRACK || t < 16); t += 2)       // the pipeline fills (TRACK: the whole interior phase)
        {
          step(t, hprev, hnext, profA, profB, std::true_type {}, std::false_type {}, std::false_type {});
          step(t + 1, hnext, hprev, profB, profA, std::true_type {}, std::true_type {}, std::false_type {});
        }
      if (!TRACK)                                             // (with overflow tracking the junk of idle lanes would reach the min/max.
                                                              //  The lanes of a group WITHOUT targets run along on junk -- r04: skipping
                                                              //  the loop for them made the step counter divergent, and those lanes then
                                                              //  walked phase B's loop over the whole step range on their own after the
                                                              //  others had finished: a task with fewer than 7 of its 8 targets took 2.3 x
                                                              //  the time of a full one, profiles/r04/r04w_uniform_steps_ab.txt.  Their
                                                              //  stores land in their own, unread checkpoint slots)
        for (jpk_run = (u32) (t - l) * 0x00010001u; t < t_switch; t += 2)
          {
            step(t, hprev, hnext, profA, profB, std::true_type {}, std::false_type {}, std::true_type {});
            step(t + 1, hnext, hprev, profB, profA, std::true_type {}, std::true_type {}, std::true_type {});
          }
      for (; t < steps; t += 2)
        {
          step(t, hprev, hnext, profA, profB, std::false_type {}, std::false_type {}, std::false_type {});
          step(t + 1, hnext, hprev, profB, profA, std::false_type {}, std::true_type {}, std::false_type {});
        }

      if (s + 1 < nstrips)
        {
          // hand-over rows were written with global stores by this wave: make them visible to its own loads
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
    }

  // ---- epilogue: reduce min/max over the group, fetch the score from the last pipeline position ----
#pragma unroll
  for (int m = 8; m >= 1; m >>= 1)
    {
      hmin = pmin(hmin, (u32) __shfl_xor((int) hmin, m, 16));
      hmax = pmax(hmax, (u32) __shfl_xor((int) hmax, m, 16));
    }
  const int l_last = (total_lanes - 1) & 15;
  score = (u32) __shfl((int) score, l_last, 16);
  leave = (u32) __shfl((int) leave, l_last, 16);
  if (l == 0)
    {
      const int mnA = (int16_t) (hmin & 0xffff), mnB = (int16_t) (hmin >> 16);
      const int mxA = (int16_t) (hmax & 0xffff), mxB = (int16_t) (hmax >> 16);
      VsxSlotOut oA, oB;
      score = psubw(score, BIAS);                 // (for the 0x8000 bias the same as the former xor)
      oA.score = (int16_t) ((int) (int16_t) (score & 0xffff) - (DA > 0 ? (Q + DA - 2) * tl : 0));     // H = H* - (i + j) g
      oB.score = (int16_t) ((int) (int16_t) (score >> 16) - (DB > 0 ? (Q + DB - 2) * tl : 0));
      oA.leave = (uint16_t) (leave & 0xffff); oB.leave = (uint16_t) (leave >> 16); oA.pad = 0; oB.pad = 0;
      oA.overflow = (mnA <= P.smin || mxA >= 32767) ? 1 : 0;     // :1774-1786
      oB.overflow = (mnB <= P.smin || mxB >= 32767) ? 1 : 0;
      if (sub_on)
        {
          slot_out[(size_t) tix * VSX_TASK_SLOTS + 2 * g] = oA;
          slot_out[(size_t) tix * VSX_TASK_SLOTS + 2 * g + 1] = oB;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Traceback: one lane per pair; follows backtrack16 (align_simd.cpp:1137-1235) over the direction
// bits written above.  Emits statistics and the run-length op list (reverse order: last column first).
// run word = (length << 2) | op,  op: 0 = M, 1 = I (gap in query, consumes target), 2 = D.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
vsx_traceback_kernel(const VsxDevParams P, const VsxTask * __restrict__ tasks,
                     const u32 * __restrict__ pair_slot, const u32 * __restrict__ pair_ids, u32 npairs,
                     const uint8_t * __restrict__ qc, const uint8_t * __restrict__ tc,
                     const u32 * __restrict__ dir, const VsxSlotOut * __restrict__ slot,
                     u32 * __restrict__ slab, const uint64_t * __restrict__ slab_off,
                     u32 * __restrict__ runs, uint64_t runs_capacity, unsigned long long * cursor,
                     VsxPairOut * __restrict__ out)
{
  const u32 k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= npairs) return;
  const u32 ts = pair_slot[k];
  const u32 task = ts >> 3, sl = ts & 7;
  const VsxTask & T = tasks[task];
  const VsxSlotOut so = slot[ts];

  VsxPairOut o;
  o.pad = 0;
  o.nruns = 0;
  o.run_off = 0;
  if (so.overflow)
    {
      o.score = 32767; o.aligned = 0; o.matches = 0; o.mismatches = 0; o.gaps = 0;
      out[pair_ids[k]] = o;
      return;
    }

  const int R = (int) T.rows;
  const int ND = (R + 3) >> 2;
  const int Q = (int) T.qlen;
  const int D = (int) T.tlen[sl];
  const int total_lanes = (Q + R - 1) / R;
  const int rcnt0 = Q - (total_lanes - 1) * R;
  const size_t steps = T.steps;
  const int g = (int) (sl >> 1) + (int) T.group0;      // lane group of the DP wave (group0 > 0: a sparse task in a shared wave)
  const bool hi = (sl & 1) != 0;
  const uint8_t * __restrict__ q = qc + T.qoff;
  const uint8_t * __restrict__ d = tc + T.toff[sl];
  u32 * __restrict__ my = slab + slab_off[k];

  int i = Q - 1, j = D - 1;
  int L = total_lanes - 1;                 // pipeline position of row i
  int r = (L == 0 ? rcnt0 : R) - 1;        // row inside that position
  int op = -1;                             // -1 none, 0 M, 1 I, 2 D
  u32 runlen = 0, nruns = 0;
  u32 al = 0, ma = 0, mi = 0, ga = 0;

  auto push = [&](int newop) {             // pushop (:1013-1030), run-length
    if (newop == op) { ++runlen; return; }
    if (op >= 0) my[nruns++] = (runlen << 2) | (u32) op;
    op = newop;
    runlen = 1;
  };
  auto row_up = [&]() {
    --i;
    if (--r < 0) { --L; r = (L == 0 ? rcnt0 : R) - 1; }
  };

  while (i >= 0 && j >= 0)
    {
      const int s = L >> 4, l = L & 15;
      const size_t t = (size_t) j + (size_t) l;
      const size_t gt = (size_t) s * steps + t;
      const u32 w = dir[T.dir_off + (((gt >> 2) * 64 + (size_t) (g * 16 + l)) * 4 + (gt & 3)) * ND + (size_t) (r >> 2)];
      const u32 hw = hi ? (w >> 16) : (w & 0xffffu);
      int rid = R - 4 * (r >> 2);                          // every position funnels all R rows (dummy rows incl.)
      if (rid > 4) rid = 4;                                // rows sharing this dword
      const int pos = 16 - 4 * rid + 4 * (r & 3);
      const u32 b = (hw >> pos) & 15u;                     // bit0 up, bit1 left, bit2 ext-up, bit3 ext-left
      ++al;
      if (op == 1 && (b & 8u)) { --j; push(1); }
      else if (op == 2 && (b & 4u)) { row_up(); push(2); }
      else if (b & 2u) { if (op != 1) ++ga; --j; push(1); }
      else if (b & 1u) { if (op != 2) ++ga; row_up(); push(2); }
      else
        {
          const u32 a = q[i], c = d[j];
          if ((a & c) != 0 && !(P.n_mismatch && (a == 15 || c == 15))) ++ma; else ++mi;
          row_up(); --j; push(0);
        }
    }
  while (i >= 0) { ++al; if (op != 2) ++ga; --i; push(2); }
  while (j >= 0) { ++al; if (op != 1) ++ga; --j; push(1); }
  if (op >= 0) my[nruns++] = (runlen << 2) | (u32) op;     // finishop (:1033-1049)

  // dense (unordered) allocation of the run list
  const unsigned long long base = atomicAdd(cursor, (unsigned long long) nruns);
  if (base + nruns <= runs_capacity)
    for (u32 x = 0; x < nruns; ++x) runs[base + x] = my[x];

  o.score = so.score;
  o.aligned = (uint16_t) al; o.matches = (uint16_t) ma; o.mismatches = (uint16_t) mi; o.gaps = (uint16_t) ga;
  o.nruns = nruns;
  o.run_off = base;
  out[pair_ids[k]] = o;
}

// ---------------------------------------------------------------------------------------------
// Traceback for the CKPT layout: one lane per pair.  The path is followed tile by tile; a tile is the
// part of one pipeline position (<= R rows) between two column checkpoints (<= 16 columns).  For the tile
// under the cursor the lane recomputes the direction bits of the sub-rectangle above/left of the cursor
// from the stored boundaries (left: column checkpoint or the left border; top: the row checkpoints of the
// position above, or the top border) with the SAME saturating int16 arithmetic as the DP kernel, then
// walks inside it with backtrack16's rules.  Bits live in LDS ([row][2][64 lanes] dwords).
// ---------------------------------------------------------------------------------------------
DEV u32 half_lo(u32 w, bool hi) { return hi ? (w >> 16) : w; }      // this pair's int16 in the LOW half (high half: don't care)
DEV int wave_max_i32(int v)
{
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) { const int o = __shfl_xor(v, m, 64); v = o > v ? o : v; }
  return v;
}

// Arithmetic of the tile recompute.  PACKED = the DP kernel's saturating v_pk_*_i16 (this pair in the LOW int16 half,
// sign bits funnelled to the right); FAST16 = for tasks whose scores provably never saturate (the planner's TRACK = 0
// class, vsx_host.cpp no_overflow_possible()): values biased by 0x8000 into unsigned 16 bits, so the half-rate-free VOP2
// forms v_add_u16 / v_sub_u16 / v_max_u16 (2 cycles per wave, measured: profiles/r01_ubench_valu.txt) give the same
// numbers as the clamped packed ops (4 cycles), the comparison a > b is the sign of an exact 32-bit v_sub_u32, and one
// v_alignbit_b32 shifts that sign into the direction word.
// r06: the compaction of a ranked plan (VsxFilterDev::rank_counts): a kept pair (accepted or weak) or a refused one (score sentinel, no
// verdict: the overflow rule fired at run time) takes the next slot of its list.  Kept pairs are a fraction of a per cent of an
// --allpairs_global plan, so the atomic is rare; the lists come out in arrival order and the host orders the few entries.
DEV void rank_note(const VsxFilterDev & FL, u32 pid, u32 verdict, int score, double idv)
{
  if (FL.rank_counts == nullptr) return;
  if (verdict == 1u || verdict == 2u)
    {
      const u32 pos = atomicAdd(FL.rank_counts, 1u);
      FL.kept_pair[pos] = pid;
      FL.kept_id[pos] = idv;
    }
  else if (verdict == 0u && score == 32767)
    FL.refused_pair[atomicAdd(FL.rank_counts + 1, 1u)] = pid;
}

template <bool FAST> struct TbOps;
template <> struct TbOps<false>
{
  static DEV u32 in(u32 lo16) { return lo16; }
  static DEV u32 score(int16_t v) { return (u32) (uint16_t) v; }
  static DEV u32 add(u32 a, u32 b) { return sadd(a, b); }
  static DEV u32 sub(u32 a, u32 b) { return ssub(a, b); }
  static DEV u32 max(u32 a, u32 b) { return pmax(a, b); }
  static DEV u32 dif(u32 a, u32 b) { return ssub(a, b); }                 // sign of a - b in bit 15
  static DEV u32 neg(u32 d) { return (d >> 15) & 1u; }
  static DEV u32 fun(u32 acc, u32 d) { return funnel(acc, d); }           // first row of a group ends in the LOW nibble
  static DEV u32 one_row_word(u32 acc) { return acc >> 12; }
  static DEV u32 nibble(u32 w, int rows_in_group, int y) { return (w >> (16 - 4 * rows_in_group + 4 * y)) & 15u; }
  static constexpr u32 UP = 1, LEFT = 2, EXT_UP = 4, EXT_LEFT = 8;
};
template <> struct TbOps<true>
{
  // no value ever leaves (0, 65535) in this class, so plain 32-bit add/sub ARE the 16-bit results (v_add_u32 / v_sub_u32,
  // 2 cycles); only the maximum needs the 16-bit form (v_max_u16 is full rate, v_max_u32 is not)
  static DEV u32 in(u32 lo16) { return (lo16 & 0xffffu) ^ 0x8000u; }
  static DEV u32 score(int16_t v) { return (u32) (int) v; }              // sign-extended addend
  static DEV u32 add(u32 a, u32 b) { return a + b; }
  static DEV u32 sub(u32 a, u32 b) { return a - b; }
  static DEV u32 max(u32 a, u32 b) { return (u32) __builtin_elementwise_max((unsigned short) a, (unsigned short) b); }
  static DEV u32 dif(u32 a, u32 b) { return a - b; }                      // exact sign in bit 31
  static DEV u32 neg(u32 d) { return d >> 31; }
  static DEV u32 fun(u32 acc, u32 d) { return __builtin_amdgcn_alignbit(acc, d, 31); }   // (acc << 1) | sign(d)
  static DEV u32 one_row_word(u32 acc) { return acc << 12; }
  static DEV u32 nibble(u32 w, int rows_in_group, int y) { return (w >> (4 * (rows_in_group - 1 - y))) & 15u; }
  static constexpr u32 UP = 8, LEFT = 4, EXT_UP = 2, EXT_LEFT = 1;
};

typedef u32 u32_unaligned __attribute__((aligned(1)));
typedef unsigned long long u64_unaligned __attribute__((aligned(1)));
struct __attribute__((aligned(4))) Quad { u32 x, y, z, w; };
struct __attribute__((aligned(4))) Trio { u32 x, y, z; };

#include "vsx_accept.h"

// -DVSX_TB_STATS=1 (A/B build, measurements only): what the checkpoint traceback does per launch -- [0] wave iterations of the
// tile loop, [1] busy (pair, tile) visits, [2] distinct tiles among the <= 8 lanes of a task summed over the iterations (= how
// many separate sets of checkpoint lines a task's lanes ask for), [3] cells walked, [4] walk-loop trips (wave level), [5] pairs;
// vsx_traceback_tilt_kernel also: [8 ..] wave cycles (s_memtime) per phase of an iteration: loads of tile 1 until they have landed,
// recompute 1, walk 1, loads of tile 2, recompute 2, walk 2, the rest; [15] = the whole kernel per wave (out16: 16 values)
#ifndef VSX_TB_STATS
#define VSX_TB_STATS 0
#endif
#if VSX_TB_STATS
__device__ unsigned long long vsx_tb_stats_dev[16];
extern "C" hipError_t vsx_internal_tb_stats(unsigned long long * out8, int reset)
{
  hipError_t e = hipMemcpyFromSymbol(out8, HIP_SYMBOL(vsx_tb_stats_dev), sizeof(unsigned long long) * 16);
  if (e == hipSuccess && reset) { unsigned long long z[16] = {0}; e = hipMemcpyToSymbol(HIP_SYMBOL(vsx_tb_stats_dev), z, sizeof z); }
  return e;
}
#endif

// CK8 = the compressed checkpoint layout of the TILT class (VSX_ROWCK_PAIR_DW / VSX_COLCK_NB): FAST arithmetic, tilted constants
// Occupancy (r02 PMC: the kernel is latency bound -- 40 % of the wave cycles in s_waitcnt at 10 waves per CU, profiles/r02_ckt_ab.txt):
// a workgroup is TWO independent waves (no barrier after the table set-up) that share the score table, the symbols of a tile are
// packed two to a byte, the boundary keeps only the 17 entries that are read and the tilted class stores its scores (0 .. 255) as
// bytes: 13.0 KB of LDS per wave at R = 16 instead of 14.75 -> 12 waves per CU = the 3 per SIMD the 168 VGPRs allow.
// occupancy floor of the traceback (latency bound, see below): 4 waves per SIMD (128 VGPRs) for the half-height tiles up to R = 24
// and the short tiles up to R = 12, 3 (168) up to R = 16 and for the taller half-height tiles, 2 beyond.  r03 A/B
// (profiles/r03/r03j_tb_occupancy_ab.txt): 150 x 300 2.71 -> 2.50 ms, 300 x 300 6.88 -> 6.61 ms; the LDS (11 KB per wave)
// then caps the CU at 14 waves
#ifndef VSX_TB_WAVES
#define VSX_TB_WAVES(R_, MID_) ((MID_) ? ((R_) <= 24 ? 4 : 3) : ((R_) <= 12 ? 4 : ((R_) <= 16 ? 3 : 2)))
#endif
template <int R, bool FAST, bool CK8 = false, bool MID = false>
__global__ void __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(VSX_TB_WAVES(R, MID), 8)))
vsx_traceback_ck_kernel(const VsxDevParams P, const VsxFilterDev FL, const VsxTask * __restrict__ tasks,
                        const u32 * __restrict__ pair_slot, const u32 * __restrict__ pair_ids, u32 npairs,
                        const uint8_t * __restrict__ qc, const uint8_t * __restrict__ tc,
                        const u32 * __restrict__ ck, const VsxSlotOut * __restrict__ slot,
                        u32 * __restrict__ slab, const uint64_t * __restrict__ slab_off,
                        u32 * __restrict__ runs, uint64_t runs_capacity, unsigned long long * cursor,
                        VsxPairOut * __restrict__ out)
{
  typedef TbOps<FAST> A;
  static_assert(VSX_RB == 1, "the tile staging below reads row checkpoints as two-step (16-byte) pairs");
  static_assert(!CK8 || FAST, "compressed checkpoints belong to the TILT class");
  // MID (vsx_internal.h VSX_MID): the DP kernel also stored a row checkpoint after row R/2 - 1 of every position and laid the
  // column checkpoints out per half, so a tile is one HALF of a position: RT = R/2 rows.  The cursor keeps its position-level
  // row index r; hh = r / RT names the half, the tile's rows are slots hh * RT .. hh * RT + RT - 1 of the position.
  static_assert(!MID || (CK8 && R % 2 == 0 && !VSX_CKT), "half-height tiles belong to the TILT class");
  constexpr int RT = MID ? R / 2 : R;
  constexpr int ND = (RT + 3) / 4;
  constexpr bool TOPPAD = FAST;                    // slot layout of position 0, see vsx_forward_kernel
  typedef typename std::conditional<CK8, uint8_t, int16_t>::type SshT;      // tilted scores are 0 .. 255 (planner-checked)
  __shared__ SshT Ssh[512];                        // S[target code][query code], row stride 32; query "code" 16 = a dummy row
  __shared__ uint16_t bitsL_all[2 * 16 * ND * 64]; // per wave: [column in tile][4-row group][lane]: this pair's 16 direction bits
  __shared__ u32 tbL_all[2 * 17 * 64];             // per wave: top boundary of the tile, H (low 16) | F (high 16), entries 1 .. 17 (entry cc + 1 = column c0 - 1 + cc)
  __shared__ uint8_t symL_all[2 * 8 * 64];         // per wave: target symbols of the tile's columns, two columns per byte
  const int tid = (int) (threadIdx.x & 63);        // lane; the two waves of a workgroup never synchronise after the set-up
  const int wv = (int) (threadIdx.x >> 6);
  uint16_t * const bitsL = bitsL_all + wv * (16 * ND * 64);
  u32 * const tbL = tbL_all + wv * (17 * 64) - 64;         // indexed 1 .. 17
  uint8_t * const symL = symL_all + wv * (8 * 64);
  for (int x = (int) threadIdx.x; x < 512; x += 128)
    Ssh[x] = (SshT) (((x & 31) < 16) ? P.matrix[(x >> 5) * 16 + (x & 15)] : (int16_t) (-P.top_step + 2 * P.tilt));
  __syncthreads();

  // every lane of the wave stays in the tile loop until all are done (wave-uniform bounds, shuffles)
  const u32 k = blockIdx.x * 128 + (u32) threadIdx.x;
  const bool valid = k < npairs;
  const u32 ts = pair_slot[valid ? k : 0];
  const u32 task = ts >> 3, sl = ts & 7;
  const VsxTask & T = tasks[task];
  const VsxSlotOut so = slot[ts];
  const bool live = valid && !so.overflow;

  const int Q = (int) T.qlen;
  const int D = (int) T.tlen[sl];
  const int total_lanes = (Q + R - 1) / R;
  const int rcnt0 = Q - (total_lanes - 1) * R;
  const int pad = TOPPAD ? R - rcnt0 : 0;
  const int rtop0 = TOPPAD ? R : rcnt0;            // slots of position 0
  const int nstrips = (total_lanes + 15) >> 4;
  const size_t steps = T.steps;
  const size_t nblk = (steps + 15) >> 4;
  const size_t rowsteps = ((size_t) nstrips * steps + 1) & ~(size_t) 1;
  constexpr bool CKT = CK8 && (VSX_CKT != 0);      // transposed checkpoint layout (vsx_internal.h)
  const size_t rowck_dw = CKT ? (((size_t) nstrips * steps) >> 3) * VSX_CKT_BLOCK_DW : (rowsteps >> 1) * VSX_ROWCK_PAIR_DW(CK8);
  constexpr size_t COL_DW = CK8 ? (size_t) 64 * 4 * VSX_COLCK_NB(R, true) : (size_t) 64 * (2 * R);     // column checkpoint dwords per wave
  const u32 * __restrict__ rowck = ck + T.dir_off;
  const u32 * __restrict__ colck = ck + T.dir_off + rowck_dw;
  const u32 * __restrict__ midck = colck + (size_t) nstrips * nblk * COL_DW;      // MID: the mid-row checkpoints, laid out like rowck
  // column checkpoint element x (0 .. 2R-1: hprev[R], E[R]) of pipeline lane `lanepos`, strip sp, 16-step block mb
  auto colck_at = [&](int sp, int mb, int lanepos, int x) -> u32 {
    const u32 * cb = colck + ((size_t) sp * nblk + (size_t) mb) * COL_DW;
    return (R % 2 == 0) ? cb[VSX_COLCK_DW(R, lanepos, x >> 2) + (x & 3)] : cb[(size_t) lanepos * (2 * R) + x];
  };
  const int g = (int) (sl >> 1) + (int) T.group0;      // lane group of the DP wave (group0 > 0: a sparse task in a shared wave)
  const bool hi = (sl & 1) != 0;
  const u32 half_sel = hi ? 0x07060302u : 0x05040100u;           // v_perm_b32(F, H, sel) = this pair's H | F << 16
  // Border values (tables, corner seeds) enter the recompute through vin(): the FAST domain is the int16 value plus the class's bias --
  // 0x8000, or 0x3E00 when the DP kernel ran its MAX3 arithmetic (VsxDevParams::max3): the checkpoints then hold 0x3E00-biased values
  // as the DP kernel computes them (r03: re-biasing them at store time cost the DP kernel ~2 instructions per step)
  const u32 fast_bias = (CK8 && P.max3) ? 0x3E00u : 0x8000u;
  auto vin = [&](u32 lo16) -> u32 { return FAST ? ((lo16 + fast_bias) & 0xffffu) : A::in(lo16); };
  const u32 bias2 = (FAST && P.tilt == 0) ? 0x80008000u : 0u;      // checkpoints of the TILT class are stored biased
  const u32 ckb = (FAST && P.tilt == 0) ? 0x8000u : 0u;
  // penalties: FAST subtracts them in 32 bits, so they are sign-extended there (tilted penalties can be negative)
  auto pen = [](int v) -> u32 { return FAST ? (u32) v : (u32) (uint16_t) v; };
  auto pen_pk = [](u32 pk) -> u32 { return FAST ? (u32) (int) (int16_t) (pk & 0xffffu) : pk; };
  const int tl = FAST ? P.tilt : 0;                // tilted coordinates, see vsx_forward_kernel
  const u32 qrt_i16 = pen(P.qrt_i), qrt_r16 = pen(P.qrt_r);     // column penalties (target-side gaps)
  const u32 rt_i16 = pen(P.rt_i), rt_r16 = pen(P.rt_r);
  const uint8_t * __restrict__ q = qc + T.qoff;
  const uint8_t * __restrict__ d = tc + T.toff[sl];
  u32 * __restrict__ my = slab + slab_off[valid ? k : 0];

  // 18 consecutive checkpoint steps, stored as two-step pairs `stride` dwords apart, land in tbL[1 ..]: entry cc + 1 is
  // step gstart + cc (cc = 0 .. 16).  Pairs outside [0, maxpair] are clamped (their entries are never used).
  auto stage_top = [&](const u32 * base, size_t stride, long gstart, long maxpair) {
    const long p0 = gstart >> 1;                               // floor, also for gstart = -1
    const int par = (int) (gstart - 2 * p0);
    if (CK8)
      {
        // {H_t, H_t+1, bytes d_t(lo) d_t(hi) d_t+1(lo) d_t+1(hi)}: F = H - d
        Trio v3[9];
#pragma unroll
        for (int e = 0; e < 9; ++e)
          {
            long p = p0 + e;
            p = p < 0 ? 0 : (p > maxpair ? maxpair : p);
            v3[e] = *reinterpret_cast<const Trio *>(base + (size_t) p * stride);
          }
        u32 * dst = tbL + (1 - par) * 64 + tid;
#pragma unroll
        for (int e = 0; e < 9; ++e)
          {
            const u32 h0 = half_lo(v3[e].x, hi) & 0xffffu, h1 = half_lo(v3[e].y, hi) & 0xffffu;
            const u32 dd = hi ? (v3[e].z >> 8) : v3[e].z;
            const int d0 = (int) (int8_t) (dd & 0xffu), d1 = (int) (int8_t) ((dd >> 16) & 0xffu);
            if (e > 0 || par == 0) dst[(2 * e) * 64] = h0 | ((h0 - (u32) d0) << 16);          // entry 0 does not exist
            if (e < 8 || par == 1) dst[(2 * e + 1) * 64] = h1 | ((h1 - (u32) d1) << 16);      // nor entry 18
          }
        return;
      }
    Quad v[9];
#pragma unroll
    for (int e = 0; e < 9; ++e)
      {
        long p = p0 + e;
        p = p < 0 ? 0 : (p > maxpair ? maxpair : p);
        v[e] = *reinterpret_cast<const Quad *>(base + (size_t) p * stride);
      }
    u32 * dst = tbL + (1 - par) * 64 + tid;
#pragma unroll
    for (int e = 0; e < 9; ++e)
      {
        if (e > 0 || par == 0) dst[(2 * e) * 64] = __builtin_amdgcn_perm(v[e].y, v[e].x, half_sel) ^ bias2;
        if (e < 8 || par == 1) dst[(2 * e + 1) * 64] = __builtin_amdgcn_perm(v[e].w, v[e].z, half_sel) ^ bias2;
      }
  };
  // transposed layout: the 17 steps gstart .. gstart + 16 of pipeline slot `slot` lie in (at most) three consecutive 8-step
  // blocks, 48 contiguous bytes each: {H of the 8 steps, the 8 x 2 difference bytes}.  Nine 16-byte loads from <= 3 segments;
  // entry cc + 1 of tbL = step gstart + cc.  Blocks outside [0, maxblk] are clamped (their entries are never used).
  auto stage_top_t = [&](const u32 * rowbase, int slot, long gstart, long maxblk) {
    const long b0 = gstart >> 3;                                 // floor, also for gstart = -1
    const int o = (int) (gstart - 8 * b0);                       // 0 .. 7
    Quad q[3][3];
#pragma unroll
    for (int e = 0; e < 3; ++e)
      {
        long b = b0 + e;
        b = b < 0 ? 0 : (b > maxblk ? maxblk : b);
        const u32 * src = rowbase + (size_t) b * VSX_CKT_BLOCK_DW + (size_t) slot * 12;
#pragma unroll
        for (int pc = 0; pc < 3; ++pc) q[e][pc] = *reinterpret_cast<const Quad *>(src + pc * 4);
      }
#pragma unroll
    for (int e = 0; e < 3; ++e)
#pragma unroll
      for (int i = 0; i < 8; ++i)
        {
          const Quad & hq = q[e][i >> 2];
          const u32 hw = (i & 3) == 0 ? hq.x : (i & 3) == 1 ? hq.y : (i & 3) == 2 ? hq.z : hq.w;
          const Quad & dq = q[e][2];
          const u32 dw2 = (i >> 1) == 0 ? dq.x : (i >> 1) == 1 ? dq.y : (i >> 1) == 2 ? dq.z : dq.w;
          const u32 h = half_lo(hw, hi) & 0xffffu;
          const int dd = (int) (int8_t) ((dw2 >> (8 * ((i & 1) * 2 + (hi ? 1 : 0)))) & 0xffu);
          const int cc = 8 * e + i - o;
          if (cc >= 0 && cc <= 16) tbL[(cc + 1) * 64 + tid] = h | ((h - (u32) dd) << 16);
        }
  };
  auto inck = [&](u32 lo16) -> u32 { return FAST ? ((lo16 & 0xffffu) ^ ckb) : lo16; };     // a checkpoint value -> A's domain
  auto stage_symbols = [&](int c0) {
    u32 w[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) w[e] = *reinterpret_cast<const u32_unaligned *>(d + c0 + 4 * e);   // VSX_CODE_SLACK bytes follow the codes
#pragma unroll
    for (int c2 = 0; c2 < 8; ++c2)
      {
        const u32 lo = (w[c2 >> 1] >> (16 * (c2 & 1))) & 15u, hi4 = (w[c2 >> 1] >> (16 * (c2 & 1) + 8)) & 15u;
        symL[c2 * 64 + tid] = (uint8_t) (lo | (hi4 << 4));
      }
  };

  int i = live ? Q - 1 : -1, j = live ? D - 1 : -1;
  int L = total_lanes - 1;
  int r = (L == 0 ? rtop0 : R) - 1;
  int op = -1;
  u32 runlen = 0, nruns = 0;
  u32 al = 0, ga = 0;
  auto push_n = [&](int newop, u32 n) {
    if (newop == op) { runlen += n; return; }
    if (op >= 0) my[nruns++] = (runlen << 2) | (u32) op;
    op = newop;
    runlen = n;
  };
  auto push = [&](int newop) { push_n(newop, 1u); };

  // ---- the run of 'I' moves along the LAST query row (right-terminal gap when the target is longer than the query): the DP
  // kernel already found where it ends (VsxSlotOut::leave), so it is one CIGAR run here.  Within the run op stays 1, hence
  // exactly one gap is opened (backtrack16 :1161-1210).  All lanes of a wave enter the tile loop in the same phase.
  if (live)
    {
      const int lv = (so.leave == 0xFFFFu) ? -1 : (int) so.leave;
      if (lv < D - 1)
        {
          const u32 n = (u32) (D - 1 - lv);
          al += n; ++ga;
          push_n(1, n);
          j = lv;
        }
    }

  for (;;)
    {
      const bool busy = (i >= 0) && (j >= 0);
      if (!__any(busy)) break;

      // ---- the tile under this lane's cursor (idle lanes recompute a harmless dummy) ----
      const int s = L >> 4, l = L & 15;
      const int jj = busy ? j : 0;
      const int m = (jj + l) >> 4;
      int c0 = 16 * m - l;
      if (c0 < 0) c0 = 0;
      const int i0 = (L == 0) ? -pad : rcnt0 + (L - 1) * R;
      const bool lastpos = (L == total_lanes - 1);
      const int hh = (MID && busy && r >= RT) ? 1 : 0;            // which half of the position (per lane)
      const int r0h = hh * RT;                                    // first slot of the tile
      const int rr = busy ? r - r0h : 0;
      const int rmax_raw = __builtin_amdgcn_readfirstlane(wave_max_i32(rr));
      const bool one_row = (rmax_raw == 0) && (RT >= 4);          // e.g. the left-terminal run in query row 0
      const int rmax = one_row ? 0 : (rmax_raw | 3);              // rows are funnelled four to a dword
      const int cmax = __builtin_amdgcn_readfirstlane(wave_max_i32(jj - c0));     // columns c0 .. c0 + cmax
#if VSX_TB_STATS
      {
        const unsigned key = busy ? (((unsigned) task & 0x3FFu) << 22) | ((unsigned) L << 12) | ((unsigned) m << 1) | (unsigned) hh : 0xFFFFFFFFu;
        bool leader = busy;
        for (int o = 1; o < 8; ++o)
          {
            const unsigned other = (unsigned) __shfl((int) key, (tid & ~7) | ((tid - o) & 7), 64);
            if ((tid & 7) >= o && other == key) leader = false;
          }
        const unsigned long long nb = __popcll(__ballot(busy)), nl = __popcll(__ballot(leader));
        if (tid == 0) { atomicAdd(&vsx_tb_stats_dev[0], 1ull); atomicAdd(&vsx_tb_stats_dev[1], nb); atomicAdd(&vsx_tb_stats_dev[2], nl); }
      }
#endif

      // top boundary for columns c0-1 .. c0+15 (entry 1 is the corner column c0 - 1) and the target symbols
      if (MID && hh == 1)
        {
          // lower half: the boundary is the mid-row checkpoint of THIS position (column c at step c + l)
          stage_top(midck + (size_t) VSX_CK_SLOT(true, g, l) * 3, VSX_ROWCK_PAIR_DW(true), (long) ((size_t) s * steps) + (long) (c0 - 1 + l),
                    (long) (rowsteps >> 1) - 1);
          if (c0 == 0)
            {
              // corner H(slot RT - 1, -1): the left border of the row above the tile, as the DP kernel seeded it
              const int xs = RT - 1;
              int ii = i0 + xs; if (ii > Q - 1) ii = Q - 1;
              u32 cv = vin((u32) (uint16_t) P.hleft[ii < 0 ? 0 : ii]);
              if (TOPPAD && L == 0 && xs < pad) cv = vin((u32) (uint16_t) (((xs == pad - 1) ? 0 : -P.top_open) + (xs - pad - 1) * tl));
              tbL[64 + tid] = cv;
            }
        }
      else if (L == 0)
        {
#pragma unroll
          for (int cc = 0; cc < 17; ++cc)
            {
              int c = c0 - 1 + cc;
              if (c > jj) c = jj;
              const u32 corner = (u32) (uint16_t) (((TOPPAD && pad > 0) ? -P.top_open : 0) - (pad + 2) * tl);   // see the DP kernel's diag seed
              tbL[(cc + 1) * 64 + tid] = (c < 0) ? vin(corner) : vin((u32) (uint16_t) (P.htop[c] - pad * tl));   // F is derived from H below
            }
        }
      else
        {
          const int Lp = L - 1;
          const int sp = Lp >> 4, lp = Lp & 15;
          if (CKT)
            stage_top_t(rowck, VSX_CK_SLOT(true, g, lp), (long) ((size_t) sp * steps) + (long) (c0 - 1 + lp), (long) ((((size_t) nstrips * steps) >> 3)) - 1);
          else
          stage_top(rowck + (size_t) VSX_CK_SLOT(CK8, g, lp) * (CK8 ? 3 : 4), VSX_ROWCK_PAIR_DW(CK8), (long) ((size_t) sp * steps) + (long) (c0 - 1 + lp),
                    (long) (rowsteps >> 1) - 1);
          if (c0 == 0) tbL[64 + tid] = vin((u32) (uint16_t) P.hleft[i0 - 1]);     // corner H(i0-1, -1)
        }
      stage_symbols(c0);

      // left boundary (state after column c0 - 1)
      u32 hp[RT], ee[RT], qa[RT];
      const u32 qrq_i = pen_pk(P.qrq_i_pk);
      const u32 rq_i = pen_pk(P.rq_i_pk);
      const bool lastrow_here = lastpos && (!MID || hh == 1);    // slot RT - 1 of this tile is the query's last row
      const u32 qrq_last = lastrow_here ? pen_pk(P.qrq_r_pk) : qrq_i;
      const u32 rq_last = lastrow_here ? pen_pk(P.rq_r_pk) : rq_i;
      if (m == 0)
        {
#pragma unroll
          for (int x = 0; x < RT; ++x)
            {
              int ii = i0 + r0h + x; if (ii > Q - 1) ii = Q - 1;
              if (ii < 0) ii = 0;
              const u32 hl = vin((u32) (uint16_t) P.hleft[ii]);
              hp[x] = hl;
              ee[x] = A::sub(hl, (ii < Q - 1) ? qrq_i : pen_pk(P.qrq_r_pk));
            }
          if (TOPPAD && L == 0 && pad > 0)                       // border state of the dummy rows, as seeded by the DP kernel
            {
#pragma unroll
              for (int x = 0; x < RT; ++x)
                if (r0h + x < pad)                                  // (pad <= R - 1: slot R - 1 is always a real row)
                  {
                    const int xs = r0h + x;
                    const int sh = (xs - pad - 1) * tl;                 // tilt of (i, -1), i = xs - pad
                    hp[x] = vin((u32) (uint16_t) (((xs == pad - 1) ? 0 : -P.top_open) + sh));
                    ee[x] = A::sub(vin((u32) (uint16_t) (-P.top_open - P.top_step + sh)), qrq_i);
                  }
            }
        }
      else
        {
          if (CK8)
            {
              constexpr int NBQ = MID ? VSX_COLCK_NBH(RT) : VSX_COLCK_NB(R, true);      // MID: the blocks of this tile's half only
              Quad fq[NBQ];
              if (MID)
                {
                  const u32 * cb = colck + ((size_t) s * nblk + (size_t) (m - 1)) * COL_DW + VSX_COLCK_CDW(VSX_COLCK_NB(R, true), VSX_CK_SLOT(true, g, l), 0)
                                   + (size_t) (hh * NBQ) * (4 * VSX_COLCK_CG);
#pragma unroll
                  for (int b = 0; b < NBQ; ++b) fq[b] = *reinterpret_cast<const Quad *>(cb + (size_t) b * (4 * VSX_COLCK_CG));
                }
              else if (CKT)
                {
                  // piece b of the lane's checkpoint = piece b % 3 of its 48-byte segment in chunk b / 3
                  const u32 * cb = colck + ((size_t) s * nblk + (size_t) (m - 1)) * VSX_COLCK_NCHUNK(R) * VSX_CKT_BLOCK_DW + (size_t) VSX_CK_SLOT(true, g, l) * 12;
#pragma unroll
                  for (int b = 0; b < NBQ; ++b) fq[b] = *reinterpret_cast<const Quad *>(cb + (size_t) (b / 3) * VSX_CKT_BLOCK_DW + (b % 3) * 4);
                }
              else
                {
                  const u32 * cb = colck + ((size_t) s * nblk + (size_t) (m - 1)) * COL_DW + VSX_COLCK_CDW(VSX_COLCK_NB(R, true), VSX_CK_SLOT(true, g, l), 0);
#pragma unroll
                  for (int b = 0; b < NBQ; ++b) fq[b] = *reinterpret_cast<const Quad *>(cb + (size_t) b * (4 * VSX_COLCK_CG));
                }
              auto flat = [&](int z) -> u32 {
                const Quad & qd = fq[z >> 2];
                return (z & 3) == 0 ? qd.x : (z & 3) == 1 ? qd.y : (z & 3) == 2 ? qd.z : qd.w;
              };
#pragma unroll
              for (int x = 0; x < RT; ++x)
                {
                  const u32 h = half_lo(flat(x), hi) & 0xffffu;                       // stored biased: already in A's domain
                  const u32 dd = flat(RT + (x >> 1)) >> (8 * ((x & 1) * 2 + (hi ? 1 : 0)));
                  hp[x] = h;
                  ee[x] = h - (u32) (int) (int8_t) (dd & 0xffu);
                }
            }
          else if (R % 2 == 0)
            {
              const u32 * cb = colck + ((size_t) s * nblk + (size_t) (m - 1)) * COL_DW + VSX_COLCK_DW(R, g * 16 + l, 0);
              constexpr int NBQ = (2 * R) / 4 ? (2 * R) / 4 : 1;        // blocks of the flat hprev[R], E[R] array
              Quad fq[NBQ];
#pragma unroll
              for (int b = 0; b < NBQ; ++b) fq[b] = *reinterpret_cast<const Quad *>(cb + (size_t) b * (4 * VSX_COLCK_G));
              auto flat = [&](int z) -> u32 {
                const Quad & qd = fq[z >> 2];
                return (z & 3) == 0 ? qd.x : (z & 3) == 1 ? qd.y : (z & 3) == 2 ? qd.z : qd.w;
              };
#pragma unroll
              for (int x = 0; x < R; ++x) { hp[x] = inck(half_lo(flat(x), hi)); ee[x] = inck(half_lo(flat(R + x), hi)); }
            }
          else
            {
#pragma unroll
              for (int x = 0; x < R; ++x)
                {
                  hp[x] = inck(half_lo(colck_at(s, m - 1, g * 16 + l, x), hi));
                  ee[x] = inck(half_lo(colck_at(s, m - 1, g * 16 + l, R + x), hi));
                }
            }
        }
      {
        u32 w[(RT + 3) / 4];
#pragma unroll
        for (int e = 0; e < (RT + 3) / 4; ++e) w[e] = *reinterpret_cast<const u32_unaligned *>(q + i0 + r0h + 4 * e);   // VSX_CODE_SLACK on both sides
#pragma unroll
        for (int x = 0; x < RT; ++x)
          qa[x] = (w[x >> 2] >> (8 * (x & 3))) & 15u;
        if (TOPPAD && L == 0)
          {
#pragma unroll
            for (int x = 0; x < RT; ++x)
              if (r0h + x < pad) qa[x] = 16u;                                             // dummy row: score -ge
          }
      }
      u32 diag = tbL[64 + tid] & 0xffffu;

      // ---- recompute: the DP kernel's row body, directions funnelled into bitsL ----
      // INT (tilted class only): every column of the tile is an interior column of every lane's target, so R_t' = 0 and, for the
      // rows below R-1, R_q' = 0 and QR_q' = QR_t': F - R, E - R and the second H - QR are not computed (14 instead of 17
      // instructions per cell).  Wave-uniform choice per tile.
      const bool tile_int = CK8 && RT > 1 && RT <= 16 && !__any(busy && (c0 + cmax >= D - 1));     // (tiles higher than 16 rows: the second row body costs more registers than it saves)
      const bool top_is_border = (L == 0) && (hh == 0);           // F of the tile's first row comes from Htop, not from a checkpoint
      for (int cc = 0; cc <= cmax; ++cc)
        {
          const int c = c0 + cc;
          const u32 qrt = (c < D - 1) ? qrt_i16 : qrt_r16;
          const u32 rt = (c < D - 1) ? rt_i16 : rt_r16;
          const u32 tbv = tbL[(cc + 2) * 64 + tid];
          const u32 topH = tbv & 0xffffu;
          u32 F = top_is_border ? A::sub(topH, qrt) : (tbv >> 16);
          const u32 b16 = (((u32) symL[(cc >> 1) * 64 + tid] >> (4 * (cc & 1))) & 15u) * 32u;
          u32 Hd = diag;
          u32 acc = 0;
          auto row = [&](int x, auto int_tag) __attribute__((always_inline)) {
            constexpr bool INT = decltype(int_tag)::value;
            const u32 V = A::score((int16_t) Ssh[b16 + qa[x]]);
            const u32 h0 = A::add(Hd, V);
            const u32 dU = A::dif(h0, F);
            const u32 h1 = A::max(h0, F);
            const u32 dL = A::dif(h1, ee[x]);
            const u32 h2 = A::max(h1, ee[x]);
            Hd = hp[x];
            hp[x] = h2;
            const bool plain = INT && x < RT - 1;                // compile-time per unrolled row
            const u32 qrq = (x == RT - 1) ? qrq_last : qrq_i;
            const u32 rq = (x == RT - 1) ? rq_last : rq_i;
            const u32 hf = A::sub(h2, qrt);
            const u32 f = INT ? F : A::sub(F, rt);
            const u32 dEU = A::dif(hf, f);
            F = A::max(f, hf);
            const u32 he = plain ? hf : A::sub(h2, qrq);
            const u32 e = plain ? ee[x] : A::sub(ee[x], rq);
            const u32 dEL = A::dif(he, e);
            ee[x] = A::max(e, he);
            acc = A::fun(A::fun(A::fun(A::fun(acc, dU), dL), dEU), dEL);
          };
          if (one_row)                                           // wave-uniform (SGPR) branches throughout
            {
              row(0, std::false_type {});
              bitsL[(cc * ND) * 64 + tid] = (uint16_t) A::one_row_word(acc);
            }
          else
            {
#pragma unroll
              for (int x4 = 0; x4 < RT; x4 += 4)
                if (x4 <= rmax)                                  // skip the rows no lane needs
                  {
                    if (tile_int)
                      {
#pragma unroll
                        for (int y = 0; y < 4; ++y)
                          if (x4 + y < RT) row(x4 + y, std::true_type {});
                      }
                    else
                      {
#pragma unroll
                        for (int y = 0; y < 4; ++y)
                          if (x4 + y < RT) row(x4 + y, std::false_type {});
                      }
                    bitsL[(cc * ND + (x4 >> 2)) * 64 + tid] = (uint16_t) acc;
                  }
            }
          diag = topH;
        }

      // ---- walk inside the tile (backtrack16 :1137-1211); matches are counted from the finished CIGAR below ----
      if (busy)
        {
          while (r >= r0h && j >= c0 && (!TOPPAD || i >= 0))
            {
              const int cw = j - c0;
              const int rv = r - r0h;                            // row inside the tile
              const u32 w = bitsL[(cw * ND + (rv >> 2)) * 64 + tid];
              int rid = RT - 4 * (rv >> 2);
              if (rid > 4) rid = 4;
              const u32 bts = A::nibble(w, rid, rv & 3);
              ++al;
              if (op == 1 && (bts & A::EXT_LEFT)) { --j; push(1); }
              else if (op == 2 && (bts & A::EXT_UP)) { --i; --r; push(2); }
              else if (bts & A::LEFT) { if (op != 1) ++ga; --j; push(1); }
              else if (bts & A::UP) { if (op != 2) ++ga; --i; --r; push(2); }
              else { --i; --r; --j; push(0); }
#if VSX_TB_STATS
              atomicAdd(&vsx_tb_stats_dev[3], 1ull);
#endif
            }
          if (r < 0 && L > 0) { --L; r = (L == 0 ? rtop0 : R) - 1; }
        }
    }
#if VSX_TB_STATS
  if (valid) atomicAdd(&vsx_tb_stats_dev[5], 1ull);
#endif
  if (!valid) return;

  VsxPairOut o;
  o.pad = 0; o.nruns = 0; o.run_off = 0;
  if (so.overflow)
    {
      o.score = 32767; o.aligned = 0; o.matches = 0; o.mismatches = 0; o.gaps = 0;
      out[pair_ids[k]] = o;
      rank_note(FL, pair_ids[k], 0u, 32767, 0.0);
      return;
    }
  if (i >= 0) { al += (u32) (i + 1); if (op != 2) ++ga; push_n(2, (u32) (i + 1)); }     // left-terminal runs
  if (j >= 0) { al += (u32) (j + 1); if (op != 1) ++ga; push_n(1, (u32) (j + 1)); }
  if (op >= 0) my[nruns++] = (runlen << 2) | (u32) op;

  // matches / mismatches of the M runs (the walk above makes no dependent global loads): replay the runs from the end
  u32 ma = 0, mi = 0;
  {
    int qi = Q - 1, tj = D - 1;
    for (u32 x = 0; x < nruns; ++x)
      {
        const u32 w = my[x];
        const int len = (int) (w >> 2);
        const u32 o2 = w & 3u;
        if (o2 == 0)
          {
            // eight cells per trip: the symbols are the 8 bytes ENDING at q[qi - b] / d[tj - b] (unaligned 64-bit loads, the
            // code buffers have slack in front); byte u of a chunk is cell kk = 7 - u of the trip
            for (int b = 0; b < len; b += 8)
              {
                const int rem = (len - b < 8) ? len - b : 8;
                const unsigned long long aw = *reinterpret_cast<const u64_unaligned *>(q + (qi - b - 7));
                const unsigned long long cw = *reinterpret_cast<const u64_unaligned *>(d + (tj - b - 7));
                const unsigned long long ones = 0x0101010101010101ull;
                const unsigned long long x = aw & cw;
                unsigned long long t = (x | (x >> 1) | (x >> 2) | (x >> 3)) & ones;            // codes share a nucleotide
                if (P.n_mismatch)
                  {
                    const unsigned long long a15 = aw & (aw >> 1) & (aw >> 2) & (aw >> 3);
                    const unsigned long long c15 = cw & (cw >> 1) & (cw >> 2) & (cw >> 3);
                    t &= ~(a15 | c15);                                                         // N never matches (:1191-1206)
                  }
                t &= ~0ull << (8 * (8 - rem));
                const u32 m8 = (u32) __builtin_popcountll(t);
                ma += m8;
                mi += (u32) rem - m8;
              }
            qi -= len; tj -= len;
          }
        else if (o2 == 1) tj -= len;
        else qi -= len;
      }
  }

  u32 verdict = 0;
  double idv = 0.0;
  if (FL.enabled && nruns > 0) verdict = accept_verdict(FL, Q, D, (int) al, (int) ma, (int) mi, (int) ga, my[nruns - 1], my[0], &idv);
  if (verdict == 3u) nruns = 0;                          // rejected: the runs never leave the device
  rank_note(FL, pair_ids[k], verdict, (int) so.score, idv);

  const unsigned long long base = atomicAdd(cursor, (unsigned long long) nruns);
  if (base + nruns <= runs_capacity)
    for (u32 x = 0; x < nruns; ++x) runs[base + x] = my[x];

  o.score = so.score;
  o.aligned = (uint16_t) al; o.matches = (uint16_t) ma; o.mismatches = (uint16_t) mi; o.gaps = (uint16_t) ga;
  o.pad = (uint16_t) verdict;
  o.nruns = nruns;
  o.run_off = base;
  out[pair_ids[k]] = o;
}


// ---------------------------------------------------------------------------------------------------------------------------------
// Traceback of the TILT class, second design (r04): vsx_traceback_tilt_kernel<R, MID>.
//
// What r03's kernel (above, still used for the plain classes) was bound by, measured (profiles/r04/r04a_fetch_calibration.txt,
// r04b_tb_stats.txt): (1) every 12- or 16-byte checkpoint read costs HBM a whole 128-byte line (a sparse one-read-per-line kernel runs
// at the same 45 G lines/s as a coalesced stream, and FETCH_SIZE tallies 64 B per line either way), and the 8 lanes of a task shared
// their lines only 2.95 : 1 -- they drift apart by a tile step whenever one of them leaves a tile through the top instead of the
// left edge -- so a launch fetched 1.55e8 lines = 19.8 GB: a floor of 3.4 ms under a 4.7 ms kernel; (2) 14 unpacked instructions per
// recomputed cell, for the whole 16 x 16 tile whatever part of it a lane needs.
//
// This kernel: * an iteration of the wave is one POSITION (MID: half position) of every lane, not one tile: the lane stages the tile
//   under its cursor AND the tile left of it (column blocks m and m - 1) with one batch of loads, then recomputes / walks them one
//   after the other.  A lane spends exactly one iteration per position unless its path spans more than two column blocks there, so
//   the lanes of a task stay together and read the same lines; half as many dependent round trips to HBM.
//   * the recompute packs TWO ROWS OF THE SAME PAIR into the halves of a register: rows x (low half) and x + HR (high half, HR =
//   half the tile height), the high half one column behind (it needs the low half's bottom row of that column, handed over in
//   registers).  All values are plain int16 (checkpoints minus the class's bias), every instruction is a packed one:
//   add, 4 saturating subtractions whose signs ARE the direction bits (exact for any int16 pair), 4 maxima, H - QR = 10 per row
//   pair in an interior tile; the eight sign bits of a row pair are collected by two v_perm_b32 (selectors 8 .. 11 replicate a sign
//   bit over a byte) and two v_bfi_b32 into one dword per four row pairs = 4 instead of 8; the substitution scores of both rows
//   come from ONE v_perm_b32 over the two columns' score dwords when the tile's query symbols are plain A C G T (else two byte reads).
//   15 instructions per two cells instead of 28.
//   Bit layout of a bits dword (column slot `it`, group of four row pairs): byte 0 = rows x (column it): bits 0-3 UP of the four
//   rows, bits 4-7 EXT_UP; byte 1 = the same for rows x + HR (column it - 1); byte 2 / 3 = LEFT | EXT_LEFT of the two halves.
// Same inputs, same outputs and the same walk rules (backtrack16, align_simd.cpp:1137-1235) as the kernel above.
// ---------------------------------------------------------------------------------------------------------------------------------
DEV u32 pk_add(u32 a, u32 b) { return UI((us2) (U2(a) + U2(b))); }                                 // v_pk_add_u16 (wraps; the planner proves that no int16 overflows)
DEV u32 pk_sub(u32 a, u32 b) { return psubw(a, b); }                                                // v_pk_sub_u16
#ifdef VSX_TB2_WAVES_N
#define VSX_TB2_WAVES(R_, MID_) VSX_TB2_WAVES_N
#endif
// occupancy floor: the 14.7 KB of LDS per wave cap a CU at 11 waves whatever the registers.  First build (r04k_tb2_occupancy_ab.txt): asking
// for 3 waves per SIMD (168 VGPRs) spilled 276 B per lane at R = 16 and LOST to 2 waves without spills (4.53 vs 4.06 ms).  After the
// register diet (rare-path table addresses behind barriers, invariants re-derived per iteration, one rolled two-tile loop: 250 -> 211
// VGPRs unconstrained, 160 B of spills at 168, none of them in the column loop) 3 waves are level or ahead on every shape
// (r04v_tb2_waves_ab.txt: 400 x 400 6.54 -> 6.35 ms)
#ifndef VSX_TB2_WAVES
#define VSX_TB2_WAVES(R_, MID_) 3
#endif
// VSX_TB_LDSSTAGE (r05): the nine row-checkpoint pairs above a tile -- 27 of the 56 staging VGPRs of a tile -- go from HBM straight into
// LDS (global_load_lds_dwordx3: 12 bytes per lane, lane-linear) instead of through registers; the landing zone is the direction-bit array
// of the tile before (dead between a tile's walk and the next tile's column loop), so no LDS is added.  Only where that array is large
// enough (two bit words per column slot: every variant that spills).
// Same-box A/B, two runs each (profiles/r05/r05f_ldsst_ab.txt; scratch per lane from the ISA): R = 16 (168 -> 64 B) 3.825 -> 3.675 ms,
// R = 26 (140 -> 52 B) 6.284 -> 6.203; R = 10 (44 -> 0 B) 2.43 -> 2.44 and 2.455 -> 2.455; R = 20 (52 -> 0 B) 4.82 -> 4.88: the DMA pays
// where it removes most of a large spill area and costs a little where there was next to none.  Hence on for R = 14, 16 and R >= 26.
#ifndef VSX_TB_LDSSTAGE
#define VSX_TB_LDSSTAGE 1
#endif
#ifndef VSX_TB_LDSSTAGE_FOR
#define VSX_TB_LDSSTAGE_FOR(R_) ((R_) == 14 || (R_) == 16 || (R_) >= 26)
#endif
template <int R, bool MID>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(VSX_TB2_WAVES(R, MID), 8)))
vsx_traceback_tilt_kernel(const VsxDevParams P, const VsxFilterDev FL, const VsxTask * __restrict__ tasks,
                          const u32 * __restrict__ pair_slot, const u32 * __restrict__ pair_ids, u32 npairs,
                          const uint8_t * __restrict__ qc, const uint8_t * __restrict__ tc,
                          const u32 * __restrict__ ck, const VsxSlotOut * __restrict__ slot,
                          u32 * __restrict__ slab, const uint64_t * __restrict__ slab_off,
                          u32 * __restrict__ runs, uint64_t runs_capacity, unsigned long long * cursor,
                          VsxPairOut * __restrict__ out)
{
  static_assert(R >= 4 && (!MID || R % 2 == 0), "R = 1 keeps the first kernel");
  constexpr int RT = MID ? R / 2 : R;              // rows of a tile
  constexpr int HR = (RT + 1) / 2;                 // row pairs: rows x and x + HR share a register (row RT of an odd tile is junk)
  constexpr int NG = (HR + 3) / 4;                 // bits dwords per column slot
  constexpr int XL = RT - 1 - HR;                  // the row pair whose HIGH half is the tile's last row
  constexpr int NBQ = MID ? VSX_COLCK_NBH(RT) : VSX_COLCK_NB(R, true);      // 16-byte blocks of one (half-)column checkpoint
  // One wave per workgroup: 14.5 KB of LDS each = 11 waves per CU (the wave's own arrays are lane-minor: no bank conflicts)
  __shared__ uint8_t Ssh[512];                     // S'[target code][query code], row stride 32; query "code" 16 = a dummy row of position 0
  __shared__ u32 Ssh4[16];                         // S'[target code][A | C << 8 | G << 16 | T << 24]: the fast score pick
  // one arena: [bitsL | symL | tbL].  bitsL[17 * NG * 64]: [column slot 0 .. 16][group of four row pairs][lane]; symL[10 * 64] bytes: target
  // symbols of columns cst .. cst + 19, two to a byte; tbL[19 * 64]: top boundary of the tile in work, H (low 16) | F (high 16), int16
  // values, entry e = column cst - 1 + e.
  // LSTAGE: the nine raw row-checkpoint pairs of a tile land at the start of the arena as [pair e][lane][4 dwords] -- a
  // global_load_lds_dwordx3 writes lane l's 12 bytes at base + 16 l (measured: ubench_ldsdma.hip, profiles/r05/r05e_ubench_ldsdma.txt;
  // NOT at 12 l) -- = 9 216 B: the direction bits of the tile before (dead between its walk and this tile's column loop) and the first
  // 512 B of its symbols (dead since its column loop; rewritten only after the raw pairs have been decoded into tbL)
  constexpr int BITS_DW = 17 * NG * 64, SYM_DW = 10 * 64 / 4;
  __shared__ __attribute__((aligned(16))) u32 arenaL[BITS_DW + SYM_DW + 19 * 64];
  u32 * const bitsL = arenaL;
  uint8_t * const symL = reinterpret_cast<uint8_t *>(arenaL + BITS_DW);
  u32 * const tbL = arenaL + BITS_DW + SYM_DW;
  constexpr bool LSTAGE = (VSX_TB_LDSSTAGE != 0) && VSX_TB_LDSSTAGE_FOR(R) && (BITS_DW + SYM_DW >= 9 * 64 * 4);
  const int tid = (int) threadIdx.x;
  for (int x = tid; x < 512; x += 64)
    Ssh[x] = (uint8_t) (((x & 31) < 16) ? P.matrix[(x >> 5) * 16 + (x & 15)] : (int16_t) (-P.top_step + 2 * P.tilt));
  if (tid < 16)
    {
      const int16_t * mr = P.matrix + tid * 16;
      Ssh4[tid] = (u32) (uint8_t) mr[1] | ((u32) (uint8_t) mr[2] << 8) | ((u32) (uint8_t) mr[4] << 16) | ((u32) (uint8_t) mr[8] << 24);
    }
  __syncthreads();

  const u32 k = blockIdx.x * 64 + (u32) threadIdx.x;
  const bool valid = k < npairs;
  const u32 ts = pair_slot[valid ? k : 0];
  const u32 task = ts >> 3, sl = ts & 7;
  const VsxTask & T = tasks[task];
  const VsxSlotOut so = slot[ts];
  const bool live = valid && !so.overflow;

  const int Q = (int) T.qlen;
  const int D = (int) T.tlen[sl];
  // (what follows from Q, the task's step count and its block offset -- positions, strips, the three checkpoint regions -- is derived
  //  again at the top of every iteration, behind a barrier: a dozen cheap instructions per iteration instead of ~20 VGPRs held for the
  //  whole kernel)
  const u32 steps32 = T.steps;
  const unsigned long long dir_off = T.dir_off;
  constexpr size_t COL_DW = (size_t) 64 * 4 * VSX_COLCK_NB(R, true);
  const int g = (int) (sl >> 1) + (int) T.group0;      // lane group of the DP wave (group0 > 0: a sparse task in a shared wave)
  const bool hi = (sl & 1) != 0;
  const u32 half_sel = hi ? 0x07060302u : 0x05040100u;           // v_perm_b32(b, a, sel) = this pair's half of a | of b << 16
  const u32 bias = P.max3 ? 0x3E00u : 0x8000u;                   // what the DP kernel's class added to every stored value
  const u32 bias_pk = bias * 0x10001u;
  const int tl = P.tilt;
  auto pk16 = [](int v) -> u32 { const u32 x = (u32) v & 0xffffu; return x | (x << 16); };
  auto pk2 = [](int lo, int hi_) -> u32 { return ((u32) lo & 0xffffu) | ((u32) hi_ << 16); };
  const int qrt_i = P.qrt_i, qrt_r = P.qrt_r, rt_i = P.rt_i, rt_r = P.rt_r;
  const int qrq_i = (int) (int16_t) (P.qrq_i_pk & 0xffffu), rq_i = (int) (int16_t) (P.rq_i_pk & 0xffffu);
  const int qrq_r = (int) (int16_t) (P.qrq_r_pk & 0xffffu), rq_r = (int) (int16_t) (P.rq_r_pk & 0xffffu);
  const uint8_t * __restrict__ q = qc + T.qoff;
  const uint8_t * __restrict__ d = tc + T.toff[sl];
  u32 * __restrict__ my = slab + slab_off[valid ? k : 0];

  int i = live ? Q - 1 : -1, j = live ? D - 1 : -1;
  int L = (Q + R - 1) / R - 1;
  int r = R - 1;
  int op = -1;
  u32 runlen = 0, nruns = 0;
  u32 al = 0, ga = 0;
  auto push_n = [&](int newop, u32 n) {
    if (newop == op) { runlen += n; return; }
    if (op >= 0) my[nruns++] = (runlen << 2) | (u32) op;
    op = newop;
    runlen = n;
  };
  auto push = [&](int newop) { push_n(newop, 1u); };

  // the run of 'I' moves along the last query row: one CIGAR run (VsxSlotOut::leave, see the first kernel)
  if (live)
    {
      const int lv = (so.leave == 0xFFFFu) ? -1 : (int) so.leave;
      if (lv < D - 1)
        {
          const u32 n = (u32) (D - 1 - lv);
          al += n; ++ga;
          push_n(1, n);
          j = lv;
        }
    }

  // masks of the bit collection: row pair k of a group owns bit k (UP / LEFT) and bit 4 + k (EXT_UP / EXT_LEFT) of every byte
  const u32 mA[4] = {0x01010101u, 0x02020202u, 0x04040404u, 0x08080808u};
  const u32 mB[4] = {0x10101010u, 0x20202020u, 0x40404040u, 0x80808080u};

#if VSX_TB_STATS
  const unsigned long long t_kernel0 = __builtin_readcyclecounter();
  unsigned n_iter = 0;
  unsigned long long tphase[6] = {0, 0, 0, 0, 0, 0};
#endif
  for (;;)
    {
      const bool busy = (i >= 0) && (j >= 0);
      if (!__any(busy)) break;
#if VSX_TB_STATS
      ++n_iter;
#endif

      // ---- this iteration: position L (MID: its half hh), column blocks m (tile 1, under the cursor) and m - 1 (tile 2) ----
      u32 Qb = (u32) Q, stb = steps32;
      unsigned long long dob = dir_off;
      asm volatile("" : "+v"(Qb), "+v"(stb), "+v"(dob));
      const int total_lanes = ((int) Qb + R - 1) / R;
      const int rcnt0 = (int) Qb - (total_lanes - 1) * R;
      const int pad = R - rcnt0;                   // dummy slots above the rows of position 0 (TOPPAD layout)
#ifdef VSX_TB_FORCE_ONE
      const int nstrips = 1;                     // (A/B build only: every query single-strip; profiles/r06/r06t_tb_onestrip_ab.txt)
#else
      const int nstrips = (total_lanes + 15) >> 4;
#endif
      const size_t steps = stb;
      const size_t nblk = (steps + 15) >> 4;
      const size_t rowsteps = ((size_t) nstrips * steps + 1) & ~(size_t) 1;
      const size_t rowck_dw = (rowsteps >> 1) * VSX_ROWCK_PAIR_DW(true);
      const u32 * __restrict__ rowck = ck + dob;
      const u32 * __restrict__ colck = rowck + rowck_dw;
      const u32 * __restrict__ midck = colck + (size_t) nstrips * nblk * COL_DW;
#ifdef VSX_TB_FORCE_ONE
      const int s = 0, l = L;
#else
      const int s = L >> 4, l = L & 15;
#endif
      const int jj = busy ? j : 0;
      const int m = (jj + l) >> 4;
      const int c0 = (16 * m - l) > 0 ? 16 * m - l : 0;                     // tile 1 = columns c0 .. c0 + 15
      const int c0b = (m >= 1 && 16 * (m - 1) - l > 0) ? 16 * (m - 1) - l : 0; // tile 2 = columns c0b .. c0 - 1 (m >= 1)
      const int i0 = (L == 0) ? -pad : rcnt0 + (L - 1) * R;
      const bool lastpos = (L == total_lanes - 1);
      const int hh = (MID && busy && r >= RT) ? 1 : 0;
      const int r0h = hh * RT;
      const bool lastrow_here = lastpos && (!MID || hh == 1);
      const bool top_is_border = (L == 0) && (hh == 0);
#if VSX_TB_STATS & 1
      {
        const unsigned key = busy ? (((unsigned) task & 0x3FFu) << 22) | ((unsigned) L << 12) | ((unsigned) m << 1) | (unsigned) hh : 0xFFFFFFFFu;
        bool leader = busy;
        for (int o = 1; o < 8; ++o)
          {
            const unsigned other = (unsigned) __shfl((int) key, (tid & ~7) | ((tid - o) & 7), 64);
            if ((tid & 7) >= o && other == key) leader = false;
          }
        const unsigned long long nb = __popcll(__ballot(busy)), nl = __popcll(__ballot(leader));
        if (tid == 0) { atomicAdd(&vsx_tb_stats_dev[1], nb); atomicAdd(&vsx_tb_stats_dev[2], nl); }
      }
#endif

      // ---- what a tile reads from HBM: nine two-step pairs of the row (or mid-row) checkpoints above it, the column checkpoint left of
      // it, 20 target symbols.  Issued as one batch, at the top of the tile's trip of the two-trip loop below -- i.e. tile 2's batch is
      // issued AFTER tile 1's walk.  LSTAGE depends on that order: its LDS-DMA lands on the direction-bit array and the first 512 B of the
      // symbol array of the tile before, which are dead only once that tile has been walked (an earlier form of this comment
      // still described the r04 order, batch 2 between tile 1's recompute and its walk -- restoring it would corrupt the walk silently) ----
      struct TileIn { Trio v3[LSTAGE ? 1 : 9]; int par; Quad fq[NBQ]; u32 sw[5]; };
      auto load_tile = [&](TileIn & in, const int cst, const int mleft) __attribute__((always_inline)) {
        in.par = 0;
        if (!top_is_border)
          {
            const u32 * base;
            long gstart;
            if (MID && hh == 1)
              {
                base = midck + (size_t) VSX_CK_SLOT(true, g, l) * 3;         // the mid-row checkpoint of THIS position (column c at step c + l)
                gstart = (long) ((size_t) s * steps) + (long) (cst - 1 + l);
              }
            else
              {
                
#ifdef VSX_TB_FORCE_ONE
                const int Lp = L - 1, sp = 0, lp = Lp;
#else
                const int Lp = L - 1, sp = Lp >> 4, lp = Lp & 15;
#endif

                base = rowck + (size_t) VSX_CK_SLOT(true, g, lp) * 3;
                gstart = (long) ((size_t) sp * steps) + (long) (cst - 1 + lp);
              }
            const long p0 = gstart >> 1;                                     // floor, also for gstart = -1
            in.par = (int) (gstart - 2 * p0);
            // pairs p0 .. p0 + 8, unclamped: one base address and immediate offsets.  Pairs outside the region (p0 = -1, or past the
            // last pair) feed entries nobody reads; the chunk's block has VSX_CK_SLACK_DW readable dwords at both ends (vsx_host.cpp)
            const u32 * bp = base + p0 * (long) VSX_ROWCK_PAIR_DW(true);
            if (LSTAGE)
              {
#pragma unroll
                for (int e = 0; e < 9; ++e)
                  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *) (bp + e * VSX_ROWCK_PAIR_DW(true)),
                                                   (__attribute__((address_space(3))) void *) (arenaL + e * 256), 12, 0, 0);
              }
            else
              {
#pragma unroll
            for (int e = 0; e < 9; ++e) in.v3[LSTAGE ? 0 : e] = *reinterpret_cast<const Trio *>(bp + e * VSX_ROWCK_PAIR_DW(true));
              }
          }
        const u32 * cb = colck + ((size_t) s * nblk + (size_t) (mleft > 0 ? mleft : 0)) * COL_DW + VSX_COLCK_CDW(VSX_COLCK_NB(R, true), VSX_CK_SLOT(true, g, l), 0)
                         + (MID ? (size_t) (hh * NBQ) * (4 * VSX_COLCK_CG) : 0);
#pragma unroll
        for (int b = 0; b < NBQ; ++b) in.fq[b] = *reinterpret_cast<const Quad *>(cb + (size_t) b * (4 * VSX_COLCK_CG));
#pragma unroll
        for (int e = 0; e < 5; ++e) in.sw[e] = *reinterpret_cast<const u32_unaligned *>(d + cst + 4 * e);     // VSX_CODE_SLACK bytes follow the codes
      };

      // ---- the tile's query rows: codes of both halves, fast-pick selectors (both tiles of the iteration share them) ----
      u32 qcd[HR], qsel[HR];
      bool plain_acgt = true;
      {
        u32 w[(RT + 3) / 4];
#pragma unroll
        for (int e = 0; e < (RT + 3) / 4; ++e) w[e] = *reinterpret_cast<const u32_unaligned *>(q + i0 + r0h + 4 * e);
        auto code_of = [&](int x) -> u32 {
          const int xx = x < RT ? x : RT - 1;                                // the junk row of an odd tile repeats the last one
          u32 c = (w[xx >> 2] >> (8 * (xx & 3))) & 15u;
          if (L == 0 && r0h + xx < pad) c = 16u;                             // dummy row of position 0: score -ge
          return c;
        };
#pragma unroll
        for (int x = 0; x < HR; ++x)
          {
            const u32 a = code_of(x), b = code_of(x + HR);
            qcd[x] = a | (b << 16);
            const u32 ia = (a >> 1) - (a >> 3), ib = (b >> 1) - (b >> 3);    // 1 2 4 8 -> 0 1 2 3
            qsel[x] = ia | 0x0C000C00u | ((4u + ib) << 16);
            plain_acgt = plain_acgt && (a == 1 || a == 2 || a == 4 || a == 8) && (b == 1 || b == 2 || b == 4 || b == 8);
          }
      }
      const bool fastV = !__any(busy && !plain_acgt);

      // row penalties, packed per half: only the row pair XL differs (its high half may be the query's last row)
      const u32 qrq_pk_i = pk16(qrq_i), rq_pk_i = pk16(rq_i);
      const u32 qrq_pk_x = pk2(qrq_i, lastrow_here ? qrq_r : qrq_i), rq_pk_x = pk2(rq_i, lastrow_here ? rq_r : rq_i);

      // ---- a tile's staged inputs -> LDS / registers, then the recompute with the direction bits into bitsL ----
      auto recompute = [&](const TileIn & in, const bool border_left, const int cst, const int rmax_raw, const int cmax, const bool tile_int)
                       __attribute__((always_inline)) {
        // top boundary: entry e = column cst - 1 + e (e = 0 .. 16; entry 17 is read by the junk column only)
        if (top_is_border)
          {
            const u32 corner = (u32) (uint16_t) (((pad > 0) ? -P.top_open : 0) - (pad + 2) * tl);      // the DP kernel's diag seed
#pragma unroll
            for (int e = 0; e < 17; ++e)
              {
                int c = cst - 1 + e;
                if (c > jj) c = jj;
                tbL[e * 64 + tid] = (c < 0) ? (corner & 0xffffu) : (u32) (uint16_t) (P.htop[c] - pad * tl);   // F is derived from H in the column loop
              }
          }
        else
          {
            u32 * dst = tbL + tid - in.par * 64;                              // entry of step 2 (p0 + e) is 2 e - par
            if (LSTAGE) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the LDS-DMA of load_tile has landed (one wave per workgroup: no barrier)
#pragma unroll
            for (int e = 0; e < 9; ++e)
              {
                // {H_t, H_t+1, bytes d_t(lo) d_t(hi) d_t+1(lo) d_t+1(hi)}: F = H - d; everything minus the class's bias
                Trio raw;
                if (LSTAGE) raw = *reinterpret_cast<const Trio *>(arenaL + e * 256 + tid * 4);
                else raw = in.v3[LSTAGE ? 0 : e];
                const u32 h0 = (half_lo(raw.x, hi) - bias) & 0xffffu, h1 = (half_lo(raw.y, hi) - bias) & 0xffffu;
                const u32 dd = hi ? (raw.z >> 8) : raw.z;
                const int d0 = (int) (int8_t) (dd & 0xffu), d1 = (int) (int8_t) ((dd >> 16) & 0xffu);
                if (e > 0 || in.par == 0) dst[(2 * e) * 64] = h0 | ((h0 - (u32) d0) << 16);
                dst[(2 * e + 1) * 64] = h1 | ((h1 - (u32) d1) << 16);        // (entries 0 .. 17)
              }
            if (cst == 0)
              {
                // corner H(row above the tile, -1)
                u32 cv;
                if (MID && hh == 1)
                  {
                    const int xs = RT - 1;                                    // the slot above the tile inside this position
                    int ii = i0 + xs; if (ii > Q - 1) ii = Q - 1;
                    cv = (u32) (uint16_t) P.hleft[ii < 0 ? 0 : ii];
                    if (L == 0 && xs < pad) cv = (u32) (uint16_t) (((xs == pad - 1) ? 0 : -P.top_open) + (xs - pad - 1) * tl);
                  }
                else cv = (u32) (uint16_t) P.hleft[i0 - 1];
                tbL[tid] = cv;
              }
          }
#pragma unroll
        for (int c2 = 0; c2 < 10; ++c2)
          {
            const u32 lo = (in.sw[c2 >> 1] >> (16 * (c2 & 1))) & 15u, hi4 = (in.sw[c2 >> 1] >> (16 * (c2 & 1) + 8)) & 15u;
            symL[c2 * 64 + tid] = (uint8_t) (lo | (hi4 << 4));
          }
        u32 hp[HR], ee[HR];
        if (border_left)
          {
#pragma unroll
            for (int x = 0; x < HR; ++x)
              {
                u32 hv[2], ev[2];
#pragma unroll
                for (int hf_ = 0; hf_ < 2; ++hf_)
                  {
                    const int xs = r0h + x + hf_ * HR;                        // slot inside the position
                    int ii = i0 + xs; if (ii > Q - 1) ii = Q - 1;
                    if (ii < 0) ii = 0;
                    asm volatile("" : "+v"(ii));                              // (this path runs in the first column block only: without the barrier its 2 HR table
                                                                              //  addresses are hoisted out of the tile loop and held in 4 HR VGPRs for the whole iteration)
                    int hl = P.hleft[ii];
                    int e0 = hl - ((ii < Q - 1) ? qrq_i : qrq_r);
                    if (L == 0 && xs < pad)                                   // border state of the dummy rows, as the DP kernel seeds it
                      {
                        const int sh = (xs - pad - 1) * tl;
                        hl = ((xs == pad - 1) ? 0 : -P.top_open) + sh;
                        e0 = -P.top_open - P.top_step + sh - qrq_i;
                      }
                    hv[hf_] = (u32) hl & 0xffffu; ev[hf_] = (u32) e0 & 0xffffu;
                  }
                hp[x] = hv[0] | (hv[1] << 16);
                ee[x] = ev[0] | (ev[1] << 16);
              }
          }
        else
          {
            auto flat = [&](int z) -> u32 {
              const Quad & qd = in.fq[z >> 2];
              return (z & 3) == 0 ? qd.x : (z & 3) == 1 ? qd.y : (z & 3) == 2 ? qd.z : qd.w;
            };
#pragma unroll
            for (int x = 0; x < HR; ++x)
              {
                const int xa = x, xb = (x + HR < RT) ? x + HR : RT - 1;
                const u32 h = pk_sub(__builtin_amdgcn_perm(flat(xb), flat(xa), half_sel), bias_pk);
                const int da = (int) (int8_t) ((flat(RT + (xa >> 1)) >> (8 * ((xa & 1) * 2 + (hi ? 1 : 0)))) & 0xffu);
                const int db = (int) (int8_t) ((flat(RT + (xb >> 1)) >> (8 * ((xb & 1) * 2 + (hi ? 1 : 0)))) & 0xffu);
                hp[x] = h;
                ee[x] = pk_sub(h, ((u32) da & 0xffffu) | ((u32) db << 16));
              }
          }
        const int xmax = rmax_raw >= HR ? HR - 1 : rmax_raw;                  // row pairs somebody needs (wave-uniform)
        // column loop state: the high half runs one column behind; the LDS reads of column it + 1 are issued during column it
        u32 tb_prev = tbL[tid];                                               // column cst - 1: the low half's first diagonal
        u32 Hd_end = 0, F_end = 0;                                            // what the low half's last row hands to the high half
        u32 cr_prev = 0, qrt_prev = 0, rt_prev = 0;
        auto sym_at = [&](int it) -> u32 { return ((u32) symL[(it >> 1) * 64 + tid] >> (4 * (it & 1))) & 15u; };
        const bool FASTV_any = fastV;
        u32 tbv_n = tbL[64 + tid];                                            // column cst
        u32 cr_n = FASTV_any ? Ssh4[sym_at(0)] : sym_at(0) * 32u;             // (two-deep: the symbol of column it + 2 is read during column it,
        u32 sy_n = sym_at(1);                                                 //  the score dword of column it + 1 from the symbol read a column earlier)
        auto column = [&](const int it, auto int_tag, auto fast_tag, auto first_tag) __attribute__((always_inline)) {
          constexpr bool INT = decltype(int_tag)::value;
          constexpr bool FASTV = decltype(fast_tag)::value;
          constexpr bool FIRST = decltype(first_tag)::value;
          const int c = cst + it;
          const u32 tbv = tbv_n;
          const u32 cr = cr_n;                                                // fast: the column's four scores; slow: its table row
          cr_n = FASTV ? Ssh4[sy_n] : sy_n * 32u;
          tbv_n = tbL[(it + 2) * 64 + tid];                                   // (column it + 1: lands while this column computes)
          sy_n = sym_at(it + 2);
          u32 qrt_pk = 0, rt_pk = 0;
          if (!INT)
            {
              const u32 ql = (u32) ((c < D - 1) ? qrt_i : qrt_r) & 0xffffu, rl = (u32) ((c < D - 1) ? rt_i : rt_r) & 0xffffu;
              qrt_pk = ql | (qrt_prev << 16); rt_pk = rl | (rt_prev << 16);
              qrt_prev = ql; rt_prev = rl;
            }
          else qrt_pk = pk16(qrt_i);
          // low half: this column's top boundary (H of column c - 1 as the diagonal, F of column c); high half: what the low half's last
          // row left behind one column ago -- one v_perm_b32 each (Hd_end / F_end keep their junk high halves)
          u32 Hd = __builtin_amdgcn_perm(Hd_end, tb_prev, 0x05040100u);
          u32 F = __builtin_amdgcn_perm(F_end, tbv, 0x05040302u);
          if (top_is_border)                                                  // F = Htop - QR_t (align_simd.cpp:830-833)
            F = __builtin_amdgcn_perm(F_end, ((tbv & 0xffffu) - (INT ? (u32) qrt_i : (qrt_pk & 0xffffu))), 0x05040100u);
#pragma unroll
          for (int g4 = 0; g4 < NG; ++g4)
            if (4 * g4 <= xmax)                                               // (wave-uniform: ONE branch per group of four row pairs; groups nobody needs are skipped)
              {
                u32 acc = 0;
#pragma unroll
                for (int x = 4 * g4; x < 4 * g4 + 4; ++x)
                  if (x < HR)
                    {
                      u32 V;
                      if (FASTV) V = __builtin_amdgcn_perm(cr_prev, cr, qsel[x]);
                      else
                        {
                          u32 qc2 = qcd[x];
                          asm volatile("" : "+v"(qc2));                       // (keeps the two table addresses of this rare path from being hoisted into 2 more VGPRs per row pair)
                          V = (u32) Ssh[cr + (qc2 & 0xffffu)] | ((u32) Ssh[cr_prev + (qc2 >> 16)] << 16);
                        }
                      const u32 h0 = pk_add(Hd, V);
                      const u32 dU = ssub(h0, F);                             // sign <=> F > h0   (up)
                      const u32 h1 = pmax(h0, F);
                      const u32 dL = ssub(h1, ee[x]);                         // sign <=> E > h1   (left)
                      const u32 h2 = pmax(h1, ee[x]);
                      Hd = hp[x];
                      hp[x] = FIRST ? bfi(0x0000FFFFu, h2, hp[x]) : h2;       // (first column: the high half has not started, its boundary stays)
                      const u32 hf = pk_sub(h2, qrt_pk);
                      const u32 f = INT ? F : pk_sub(F, rt_pk);
                      const u32 dEU = ssub(hf, f);                            // sign <=> F - R > H - QR (extend up)
                      F = pmax(f, hf);
                      const bool plain = INT && x != XL;
                      const u32 he = plain ? hf : pk_sub(h2, x == XL ? qrq_pk_x : qrq_pk_i);
                      const u32 e = plain ? ee[x] : pk_sub(ee[x], x == XL ? rq_pk_x : rq_pk_i);
                      const u32 dEL = ssub(he, e);                            // sign <=> E - R > H - QR (extend left)
                      const u32 en = pmax(e, he);
                      ee[x] = FIRST ? bfi(0x0000FFFFu, en, ee[x]) : en;
                      acc = a_bfi(mA[x & 3], __builtin_amdgcn_perm(dL, dU, 0x0B0A0908u), acc);
                      acc = a_bfi(mB[x & 3], __builtin_amdgcn_perm(dEL, dEU, 0x0B0A0908u), acc);
                    }
                bitsL[(it * NG + g4) * 64 + tid] = acc;
              }
          Hd_end = Hd;                                                        // low half = H(row HR - 1, column c - 1): the high half's next diagonal
          F_end = F;                                                          // low half = F leaving row HR - 1 in column c
          tb_prev = tbv;
          cr_prev = cr;
        };
        auto sweep = [&](auto int_tag, auto fast_tag) __attribute__((always_inline)) {
          column(0, int_tag, fast_tag, std::true_type {});
          for (int it = 1; it <= cmax + 1; ++it) column(it, int_tag, fast_tag, std::false_type {});
        };
        if (tile_int) { if (fastV) sweep(std::true_type {}, std::true_type {}); else sweep(std::true_type {}, std::false_type {}); }
        else          { if (fastV) sweep(std::false_type {}, std::true_type {}); else sweep(std::false_type {}, std::false_type {}); }
      };
      // ---- walk inside the tile (backtrack16 :1137-1211); matches are counted from the finished CIGAR below ----
      auto walk = [&](const int cst) __attribute__((always_inline)) {
        while (r >= r0h && j >= cst && i >= 0)
          {
            const int cw = j - cst;
            const int rv = r - r0h;
            const bool up_half = rv >= HR;
            const int rx = up_half ? rv - HR : rv;
            const u32 w = bitsL[((cw + (up_half ? 1 : 0)) * NG + (rx >> 2)) * 64 + tid] >> ((rx & 3) + (up_half ? 8 : 0));
            ++al;
            // backtrack16's five-way choice without divergence (r03: ~50 instructions per cell walked as nested branches): continue-I on
            // ext-left > continue-D on ext-up > left > up > diagonal; only the run hand-over to the slab is a branch
            const u32 inI = (op == 1) ? 1u : 0u, inD = (op == 2) ? 1u : 0u;
            const u32 c1 = inI & (w >> 20);                                   // EXT_LEFT
            const u32 c2 = (c1 ^ 1u) & inD & (w >> 4);                        // EXT_UP
            const u32 c12 = c1 | c2;
            const u32 c3 = (c12 ^ 1u) & (w >> 16);                            // LEFT
            const u32 c4 = ((c12 | c3) ^ 1u) & w;                             // UP
            const u32 isI = (c1 | c3) & 1u, isD = (c2 | c4) & 1u;
            const int newop = (int) (isI + 2u * isD);                         // 0 M, 1 I (consumes a target column), 2 D
            ga += ((c3 & (inI ^ 1u)) | (c4 & (inD ^ 1u))) & 1u;               // a gap opens
            j -= (int) (isD ^ 1u);
            const int di = (int) (isI ^ 1u);
            i -= di; r -= di;
            if (newop != op)
              {
                if (op >= 0) my[nruns++] = (runlen << 2) | (u32) op;
                op = newop;
                runlen = 0;
              }
            ++runlen;
#if VSX_TB_STATS & 1
            atomicAdd(&vsx_tb_stats_dev[3], 1ull);
#endif
          }
      };

#if VSX_TB_STATS & 2
#define TBMARK(k_) do { __builtin_amdgcn_s_waitcnt(0); const unsigned long long now_ = __builtin_readcyclecounter(); tphase[k_] += now_ - tmark; tmark = now_; } while (0)
      unsigned long long tmark = __builtin_readcyclecounter();
#else
#define TBMARK(k_) do { } while (0)
#endif
      // tile 1 (under the cursor), then tile 2 (left of it) for the lanes that left tile 1 through its left edge and are still inside
      // this (half) position.  Every lane of the wave that has a second tile asks for its inputs in the same iteration, whichever way
      // it left tile 1: the lanes of a task stay on the same lines.  ONE copy of the code (a two-trip loop, not unrolled): half the
      // instruction footprint and the register allocation of a single tile.
#pragma unroll 1
      for (int tno = 0; tno < 2; ++tno)
        {
          const bool need = (tno == 0) ? busy : (busy && (m >= 1) && (r >= r0h) && (i >= 0) && (j >= 0) && (j < c0));
          if (tno == 1 && !__any(need)) break;
          const int cst = tno ? c0b : c0;
          TileIn in;
          load_tile(in, cst, tno ? m - 2 : m - 1);
          TBMARK(3 * tno);
          const int rmax_raw = __builtin_amdgcn_readfirstlane(wave_max_i32(need ? r - r0h : 0));
          const int cmax = __builtin_amdgcn_readfirstlane(wave_max_i32(need ? j - cst : 0));
          const bool tile_int = tno ? true : !__any(busy && (c0 + cmax >= D - 1));
          recompute(in, tno ? (m <= 1) : (m == 0), cst, rmax_raw, cmax, tile_int);
          TBMARK(3 * tno + 1);
          if (need) walk(cst);
          TBMARK(3 * tno + 2);
        }
      if (busy && r < 0 && L > 0) { --L; r = R - 1; }
    }
#if VSX_TB_STATS
  if (valid) atomicAdd(&vsx_tb_stats_dev[5], 1ull);
  if (tid == 0)
    {
      atomicAdd(&vsx_tb_stats_dev[15], __builtin_readcyclecounter() - t_kernel0); atomicAdd(&vsx_tb_stats_dev[0], (unsigned long long) n_iter);
      for (int k2 = 0; k2 < 6; ++k2) atomicAdd(&vsx_tb_stats_dev[8 + k2], tphase[k2]);
    }
#endif
  if (!valid) return;

  VsxPairOut o;
  o.pad = 0; o.nruns = 0; o.run_off = 0;
  if (so.overflow)
    {
      o.score = 32767; o.aligned = 0; o.matches = 0; o.mismatches = 0; o.gaps = 0;
      out[pair_ids[k]] = o;
      rank_note(FL, pair_ids[k], 0u, 32767, 0.0);
      return;
    }
  if (i >= 0) { al += (u32) (i + 1); if (op != 2) ++ga; push_n(2, (u32) (i + 1)); }     // left-terminal runs
  if (j >= 0) { al += (u32) (j + 1); if (op != 1) ++ga; push_n(1, (u32) (j + 1)); }
  if (op >= 0) my[nruns++] = (runlen << 2) | (u32) op;

  // matches / mismatches of the M runs: replay the runs from the end, eight symbols a trip (see the first kernel)
  u32 ma = 0, mi = 0;
  {
    int qi = Q - 1, tj = D - 1;
    for (u32 x = 0; x < nruns; ++x)
      {
        const u32 w = my[x];
        const int len = (int) (w >> 2);
        const u32 o2 = w & 3u;
        if (o2 == 0)
          {
            for (int b = 0; b < len; b += 8)
              {
                const int rem = (len - b < 8) ? len - b : 8;
                const unsigned long long aw = *reinterpret_cast<const u64_unaligned *>(q + (qi - b - 7));
                const unsigned long long cw = *reinterpret_cast<const u64_unaligned *>(d + (tj - b - 7));
                const unsigned long long ones = 0x0101010101010101ull;
                const unsigned long long x2 = aw & cw;
                unsigned long long t = (x2 | (x2 >> 1) | (x2 >> 2) | (x2 >> 3)) & ones;
                if (P.n_mismatch)
                  {
                    const unsigned long long a15 = aw & (aw >> 1) & (aw >> 2) & (aw >> 3);
                    const unsigned long long c15 = cw & (cw >> 1) & (cw >> 2) & (cw >> 3);
                    t &= ~(a15 | c15);
                  }
                t &= ~0ull << (8 * (8 - rem));
                const u32 m8 = (u32) __builtin_popcountll(t);
                ma += m8;
                mi += (u32) rem - m8;
              }
            qi -= len; tj -= len;
          }
        else if (o2 == 1) tj -= len;
        else qi -= len;
      }
  }

  u32 verdict = 0;
  double idv = 0.0;
  if (FL.enabled && nruns > 0) verdict = accept_verdict(FL, Q, D, (int) al, (int) ma, (int) mi, (int) ga, my[nruns - 1], my[0], &idv);
  if (verdict == 3u) nruns = 0;
  rank_note(FL, pair_ids[k], verdict, (int) so.score, idv);

  const unsigned long long base = atomicAdd(cursor, (unsigned long long) nruns);
  if (base + nruns <= runs_capacity)
    for (u32 x = 0; x < nruns; ++x) runs[base + x] = my[x];

  o.score = so.score;
  o.aligned = (uint16_t) al; o.matches = (uint16_t) ma; o.mismatches = (uint16_t) mi; o.gaps = (uint16_t) ga;
  o.pad = (uint16_t) verdict;
  o.nruns = nruns;
  o.run_off = base;
  out[pair_ids[k]] = o;
}

// ---------------------------------------------------------------------------------------------
// ASCII -> 4-bit IUPAC set code (utils/maps.cpp:75-117), 16 bytes per lane; HBM-bound.
// ---------------------------------------------------------------------------------------------
DEV u32 map4(u32 ch)
{
  const u32 c = ch | 0x20u;                 // fold case; non-letters never match a letter below
  if (((ch | 0x20u) < 'a') || ((ch | 0x20u) > 'z') || ((ch & 0xC0u) != 0x40u)) return 0;
  switch (c)
    {
    case 'a': return 1;  case 'b': return 14; case 'c': return 2;  case 'd': return 13;
    case 'g': return 4;  case 'h': return 11; case 'k': return 12; case 'm': return 3;
    case 'n': return 15; case 'r': return 5;  case 's': return 6;  case 't': return 8;
    case 'u': return 8;  case 'v': return 7;  case 'w': return 9;  case 'y': return 10;
    default: return 0;
    }
}

__global__ void __launch_bounds__(256)
vsx_encode_kernel(const uint8_t * __restrict__ ascii, uint8_t * __restrict__ codes, uint64_t nbytes)
{
  __shared__ uint8_t lut[256];
  lut[threadIdx.x] = (uint8_t) map4(threadIdx.x);
  __syncthreads();
  const uint64_t nvec = nbytes >> 4;
  const uint64_t stride = (uint64_t) gridDim.x * blockDim.x;
  for (uint64_t v = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x; v < nvec; v += stride)
    {
      const uint4 in = reinterpret_cast<const uint4 *>(ascii)[v];
      u32 w[4] = {in.x, in.y, in.z, in.w};
      u32 o[4];
#pragma unroll
      for (int x = 0; x < 4; ++x)
        o[x] = (u32) lut[w[x] & 0xff] | ((u32) lut[(w[x] >> 8) & 0xff] << 8) |
               ((u32) lut[(w[x] >> 16) & 0xff] << 16) | ((u32) lut[w[x] >> 24] << 24);
      reinterpret_cast<uint4 *>(codes)[v] = make_uint4(o[0], o[1], o[2], o[3]);
    }
  // tail (< 16 bytes)
  if (blockIdx.x == 0 && threadIdx.x < (nbytes & 15))
    {
      const uint64_t p = (nvec << 4) + threadIdx.x;
      codes[p] = lut[ascii[p]];
    }
}

// one wavefront per sequence: impure[s] = 1 if any symbol is not an unambiguous A/C/G/T(U) code
__global__ void __launch_bounds__(256)
vsx_purity_kernel(const uint8_t * __restrict__ codes, const uint64_t * __restrict__ off,
                  const uint32_t * __restrict__ len, uint64_t nseq, uint8_t * __restrict__ impure)
{
  const uint64_t s = (uint64_t) blockIdx.x * 4 + (threadIdx.x >> 6);
  if (s >= nseq) return;
  const int lane = threadIdx.x & 63;
  const uint8_t * p = codes + off[s];
  const uint32_t n = len[s];
  int bad = 0;
  for (uint32_t x = lane; x < n; x += 64)
    {
      const u32 c = p[x];
      bad |= !((c != 0) && ((c & (c - 1)) == 0));
    }
  const unsigned long long any = __ballot(bad);
  if (lane == 0) impure[s] = any ? 1 : 0;
}

// the MAX3 class rests on v_pk_maximum3_f16 being an INTEGER maximum on the bit patterns 0x0000 .. 0x7BFF (the
// non-negative finite fp16 numbers, denormals included, no flush).  One-off self-test per device (vsx_create): every pattern of
// that range against two others drawn from the boundaries and a hash, both halves; bad[0] counts disagreements.
__global__ void __launch_bounds__(256)
vsx_max3_selftest_kernel(u32 * bad)
{
  const u32 x = blockIdx.x * 256u + threadIdx.x;              // 0 .. 0x7BFF
  if (x > 0x7BFFu) return;
  const u32 edge[8] = {0u, 1u, 0x03FFu, 0x0400u, 0x3C00u, 0x3E00u, 0x7BFEu, 0x7BFFu};
  u32 wrong = 0;
  for (u32 k = 0; k < 64; ++k)
    {
      const u32 h = (x * 2654435761u) ^ (k * 0x9E3779B9u);
      const u32 y = (k < 8) ? edge[k] : (h >> 7) % 0x7C00u;
      const u32 z = (k < 16) ? edge[(k + 3) & 7] : (h >> 17) % 0x7C00u;
      const u32 a = x | (y << 16), b = y | (z << 16), c = z | (x << 16);
      const u32 got = pk_max3_bits(a, b, c);
      const u32 m = x > y ? (x > z ? x : z) : (y > z ? y : z);
      if (got != (m | (m << 16))) ++wrong;
      // vsx_traceback_tilt_kernel collects direction bits with v_perm_b32 selectors 8 .. 11 (a byte of 0x00 / 0xFF from the sign of
      // bytes 1, 3 of the second and 1, 3 of the first source): bad[1] counts disagreements with that reading
      const u32 pa = h ^ (x << 13), pb = (h * 0x85EBCA6Bu) ^ k;
      const u32 pg = __builtin_amdgcn_perm(pa, pb, 0x0B0A0908u);
      const u32 pe = ((pb >> 15) & 1u) * 0xFFu | (((pb >> 31) & 1u) * 0xFFu) << 8 | (((pa >> 15) & 1u) * 0xFFu) << 16 | (((pa >> 31) & 1u) * 0xFFu) << 24;
      if (pg != pe) atomicAdd(bad + 1, 1u);
    }
  if (wrong) atomicAdd(bad, wrong);
}
// per device (one failing device must not decide for the others; contexts of several devices are created concurrently):
// 1 = the TILT class keeps the first traceback kernel there (self-test failed or could not run).  Devices >= 64 share the last entry.
static std::atomic<int> g_tb_v1_dev[65];
extern "C" void vsx_internal_set_tb_v2(int device, int on) { g_tb_v1_dev[device < 0 || device > 64 ? 64 : device].store(on ? 0 : 1); }
extern "C" hipError_t vsx_launch_max3_selftest(u32 * d_bad, hipStream_t st)
{
  hipLaunchKernelGGL(vsx_max3_selftest_kernel, dim3((0x7C00 + 255) / 256), dim3(256), 0, st, d_bad);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------
static const int kRows[] = {1, 4, 8, 10, 12, 14, 16, 18, 20, 22, 24, 26, 28, 32};

extern "C" const int * vsx_supported_rows(int * count)
{
  *count = (int) (sizeof(kRows) / sizeof(kRows[0]));
  return kRows;
}

extern "C" hipError_t vsx_launch_encode(const uint8_t * d_ascii, uint8_t * d_codes, uint64_t nbytes, hipStream_t st)
{
  if (nbytes == 0) return hipSuccess;
  uint64_t blocks = ((nbytes >> 4) + 255) / 256;
  if (blocks < 1) blocks = 1;
  if (blocks > 256 * 8) blocks = 256 * 8;
  hipLaunchKernelGGL(vsx_encode_kernel, dim3((unsigned) blocks), dim3(256), 0, st, d_ascii, d_codes, nbytes);
  return hipGetLastError();
}

extern "C" hipError_t vsx_launch_purity(const uint8_t * d_codes, const uint64_t * d_off, const uint32_t * d_len,
                                        uint64_t nseq, uint8_t * d_impure, hipStream_t st)
{
  if (nseq == 0) return hipSuccess;
  hipLaunchKernelGGL(vsx_purity_kernel, dim3((unsigned) ((nseq + 3) / 4)), dim3(256), 0, st,
                     d_codes, d_off, d_len, nseq, d_impure);
  return hipGetLastError();
}

#define VSX_FWD_GO(GRID_, ...) hipLaunchKernelGGL((vsx_forward_kernel<__VA_ARGS__>), dim3(GRID_), dim3(64), 0, st, P, d_tasks, q, t, dir, strip, slot, ntasks)
#define VSX_FWD_GO4(GRID_, ...) hipLaunchKernelGGL((vsx_forward_kernel<__VA_ARGS__>), dim3(GRID_), dim3(256), 0, st, P, d_tasks, q, t, dir, strip, slot, ntasks)
template <int R, bool CK>
static hipError_t launch_fwd2(int generic, int track, int nq, int one, const VsxDevParams & P, const VsxTask * d_tasks, uint32_t ntasks,
                              const uint8_t * q, const uint8_t * t, uint32_t * dir, uint2 * strip,
                              VsxSlotOut * slot, hipStream_t st)
{
  if (nq == 8)
    {
      // the pair-profile classes: workgroups of four whole-wave tasks of one query (the planner launches whole groups)
      if (!(P.tilt != 0 && CK && generic && !track) || (ntasks & 3u)) return hipErrorInvalidValue;
      if constexpr (CK && R >= 4 && VSX_QPL != 0 && VSX_FEED2 != 0 && !VSX_CKT)
        {
          if (!one) return hipErrorInvalidValue;
          if (P.max3) VSX_FWD_GO4(ntasks / 4, R, true, false, true, true, true, 1, true, true); else VSX_FWD_GO4(ntasks / 4, R, true, false, true, true, false, 1, true, true);
          return hipGetLastError();
        }
      else return hipErrorInvalidValue;
    }
  if (nq != 1)
    {
      // the sparse-task classes (NQ tasks per wave): TILT family only, R >= 4 (a one-row-per-lane wave has nothing to share)
      if (!(P.tilt != 0 && CK && generic && !track && (nq == 2 || nq == 4))) return hipErrorInvalidValue;
      if constexpr (CK && R >= 4 && VSX_QPL != 0 && !VSX_CKT)
        {
          const uint32_t grid = (ntasks + (uint32_t) nq - 1) / (uint32_t) nq;
          if (nq == 2) { if (P.max3) VSX_FWD_GO(grid, R, true, false, true, true, true, 2); else VSX_FWD_GO(grid, R, true, false, true, true, false, 2); }
          else         { if (P.max3) VSX_FWD_GO(grid, R, true, false, true, true, true, 4); else VSX_FWD_GO(grid, R, true, false, true, true, false, 4); }
          return hipGetLastError();
        }
      else return hipErrorInvalidValue;
    }
  if (P.tilt != 0)
    {
      if (!(CK && generic && !track)) return hipErrorInvalidValue;
      if constexpr (CK)
        {
          if constexpr (R >= 4 && VSX_QPL != 0 && !VSX_CKT)
            {
              if (one)
                {
                  if (P.max3) VSX_FWD_GO(ntasks, R, true, false, true, true, true, 1, false, true);
                  else VSX_FWD_GO(ntasks, R, true, false, true, true, false, 1, false, true);
                  return hipGetLastError();
                }
            }
          if (P.max3 && !VSX_CKT) VSX_FWD_GO(ntasks, R, true, false, true, true, (VSX_CKT == 0));
          else VSX_FWD_GO(ntasks, R, true, false, true, true);
        }
    }
  else if (generic && track) VSX_FWD_GO(ntasks, R, true, true, CK);
  else if (generic) VSX_FWD_GO(ntasks, R, true, false, CK);
  else if (track) VSX_FWD_GO(ntasks, R, false, true, CK);
  else VSX_FWD_GO(ntasks, R, false, false, CK);
  return hipGetLastError();
}
#undef VSX_FWD_GO
#undef VSX_FWD_GO4

// nq = tasks per wave: 1, or 2 / 4 for the sparse-task classes (tasks of <= 4 / <= 2 targets, TILT family, single-strip queries);
// 8 = the pair-profile class: workgroups of four whole-wave tasks of one pure-ACGT query (ntasks a multiple of 4)
// one != 0: every task of the launch is a single-strip query (ONE)
extern "C" hipError_t vsx_launch_forward(int rows, int generic, int track, int ckpt, int nq, int one, VsxDevParams P, const VsxTask * d_tasks, uint32_t ntasks,
                                         const uint8_t * q, const uint8_t * t, uint32_t * dir, uint2 * strip,
                                         VsxSlotOut * slot, hipStream_t st)
{
  if (ntasks == 0) return hipSuccess;
#define FWDR(RR) case RR: return ckpt ? launch_fwd2<RR, true>(generic, track, nq, one, P, d_tasks, ntasks, q, t, dir, strip, slot, st) \
                                       : launch_fwd2<RR, false>(generic, track, nq, one, P, d_tasks, ntasks, q, t, dir, strip, slot, st)
  switch (rows)
    {
    FWDR(1); FWDR(4); FWDR(8); FWDR(10); FWDR(12); FWDR(14); FWDR(16); FWDR(18); FWDR(20); FWDR(22); FWDR(24); FWDR(26); FWDR(28); FWDR(32);
    default: return hipErrorInvalidValue;
    }
#undef FWDR
}

extern "C" hipError_t vsx_launch_traceback(VsxDevParams P, const VsxTask * d_tasks, const uint32_t * d_pair_slot,
                                           const uint32_t * d_pair_ids, uint32_t npairs,
                                           const uint8_t * q, const uint8_t * t,
                                           const uint32_t * dir, const VsxSlotOut * slot,
                                           uint32_t * slab, const uint64_t * slab_off,
                                           uint32_t * runs, uint64_t runs_capacity, unsigned long long * cursor,
                                           VsxPairOut * out, hipStream_t st)
{
  if (npairs == 0) return hipSuccess;
  hipLaunchKernelGGL(vsx_traceback_kernel, dim3((npairs + 255) / 256), dim3(256), 0, st,
                     P, d_tasks, d_pair_slot, d_pair_ids, npairs, q, t, dir, slot, slab, slab_off,
                     runs, runs_capacity, cursor, out);
  return hipGetLastError();
}

template <int R, bool MID>
static hipError_t launch_tbtilt(const VsxDevParams & P, const VsxFilterDev & F, const VsxTask * d_tasks, const uint32_t * d_pair_slot,
                                const uint32_t * d_pair_ids, uint32_t npairs, const uint8_t * q, const uint8_t * t,
                                const uint32_t * ck, const VsxSlotOut * slot, uint32_t * slab, const uint64_t * slab_off,
                                uint32_t * runs, uint64_t cap, unsigned long long * cursor, VsxPairOut * out, hipStream_t st)
{
  if constexpr (R >= 4)
    hipLaunchKernelGGL((vsx_traceback_tilt_kernel<R, MID>), dim3((npairs + 63) / 64), dim3(64), 0, st,
                       P, F, d_tasks, d_pair_slot, d_pair_ids, npairs, q, t, ck, slot, slab, slab_off, runs, cap, cursor, out);
  return hipGetLastError();
}

template <int R, bool FAST, bool CK8 = false, bool MID = false>
static hipError_t launch_tbck(const VsxDevParams & P, const VsxFilterDev & F, const VsxTask * d_tasks, const uint32_t * d_pair_slot,
                              const uint32_t * d_pair_ids, uint32_t npairs, const uint8_t * q, const uint8_t * t,
                              const uint32_t * ck, const VsxSlotOut * slot, uint32_t * slab, const uint64_t * slab_off,
                              uint32_t * runs, uint64_t cap, unsigned long long * cursor, VsxPairOut * out, hipStream_t st)
{
  hipLaunchKernelGGL((vsx_traceback_ck_kernel<R, FAST, CK8, MID>), dim3((npairs + 127) / 128), dim3(128), 0, st,
                     P, F, d_tasks, d_pair_slot, d_pair_ids, npairs, q, t, ck, slot, slab, slab_off, runs, cap, cursor, out);
  return hipGetLastError();
}

extern "C" hipError_t vsx_launch_traceback_ck(int rows, int fast16, VsxDevParams P, VsxFilterDev F, const VsxTask * d_tasks, const uint32_t * d_pair_slot,
                                              const uint32_t * d_pair_ids, uint32_t npairs,
                                              const uint8_t * q, const uint8_t * t,
                                              const uint32_t * ck, const VsxSlotOut * slot,
                                              uint32_t * slab, const uint64_t * slab_off,
                                              uint32_t * runs, uint64_t runs_capacity, unsigned long long * cursor,
                                              VsxPairOut * out, hipStream_t st)
{
  if (npairs == 0) return hipSuccess;
  static const bool v1_env = std::getenv("VSX_TB_V1") != nullptr;          // A/B: the first kernel for the TILT class too
  int dev_now = 0;
  if (hipGetDevice(&dev_now) != hipSuccess) { (void) hipGetLastError(); dev_now = 64; }
  const bool v2 = g_tb_v1_dev[dev_now < 0 || dev_now > 64 ? 64 : dev_now].load(std::memory_order_relaxed) == 0 && !v1_env && !VSX_CKT;
#define TBCK(RR) case RR: return (fast16 && P.tilt != 0 && v2 && RR >= 4) ? launch_tbtilt<RR, VSX_MID(RR, true)>(P, F, d_tasks, d_pair_slot, d_pair_ids, npairs, q, t, ck, slot, slab, slab_off, runs, runs_capacity, cursor, out, st) \
                                 : (fast16 && P.tilt != 0) ? launch_tbck<RR, true, true, VSX_MID(RR, true)>(P, F, d_tasks, d_pair_slot, d_pair_ids, npairs, q, t, ck, slot, slab, slab_off, runs, runs_capacity, cursor, out, st) \
                                 : fast16 ? launch_tbck<RR, true>(P, F, d_tasks, d_pair_slot, d_pair_ids, npairs, q, t, ck, slot, slab, slab_off, runs, runs_capacity, cursor, out, st) \
                                        : launch_tbck<RR, false>(P, F, d_tasks, d_pair_slot, d_pair_ids, npairs, q, t, ck, slot, slab, slab_off, runs, runs_capacity, cursor, out, st)
  switch (rows)
    {
    TBCK(1); TBCK(4); TBCK(8); TBCK(10); TBCK(12); TBCK(14); TBCK(16); TBCK(18); TBCK(20); TBCK(22); TBCK(24); TBCK(26); TBCK(28); TBCK(32);
    default: return hipErrorInvalidValue;
    }
#undef TBCK
}

// dwords of checkpoint storage one task needs (row + column checkpoints)
extern "C" uint64_t vsx_ckpt_dwords(uint64_t nstrips, uint64_t steps, uint64_t rows, int tilt)
{
  const uint64_t nblk = (steps + 15) >> 4;
  if (tilt && VSX_CKT)      // transposed layout: 3 KB blocks; steps is a multiple of 8
    return ((nstrips * steps) >> 3) * VSX_CKT_BLOCK_DW + nstrips * nblk * (uint64_t) VSX_COLCK_NCHUNK(rows) * VSX_CKT_BLOCK_DW;
  const uint64_t rowck = (((nstrips * steps) + 1) >> 1) * VSX_ROWCK_PAIR_DW(tilt != 0);
  if (tilt)
    {
      // (VSX_COLCK_NB needs a compile-time row count inside the kernels; here the same arithmetic at run time)
      const bool mid = VSX_MID(rows, true);
      const uint64_t rt = mid ? rows / 2 : rows;
      const uint64_t nbh = (rt + (rt + 1) / 2 + 3) / 4;
      const uint64_t nb = mid ? 2 * nbh : nbh;
      return rowck + nstrips * nblk * 64 * 4 * nb + (mid ? rowck : 0);          // mid-row checkpoints: a second row-checkpoint region
    }
  return rowck + nstrips * nblk * 64 * 2 * rows;
}
