// vsx_kmer.hip -- k-mer candidate counting on gfx950 (SURVEY.md 8f "next" #1).
//
// Replaces the reference's search_topscores (core/searchcore.cpp:260-340: one counter per database sequence, bumped
// once per (unique query word, sequence containing the word), then the sequences with count >= minmatches go to the
// top-N heap) and the index it walks (core/dbindex.cpp:125-255, core/unique.cpp:155-352) by HBM/LDS-bound integer
// kernels.  No MFMA: this is a sparse count, not a contraction.
//
//   index   = postings grouped by (word, TILE of 2^15 consecutive sequence numbers): a CSR over 4^w x ntiles buckets of
//             16-BIT tile-local sequence indices, two per dword (an odd bucket ends in the 0xFFFF sentinel) -- half the
//             bytes of sequence numbers, and the count kernel is bound by exactly this stream.
//             Built on the device from the 4-bit codes in two sweeps (count, fill); the order inside a bucket is
//             arbitrary (counting commutes), so no sort is needed.
//   count   = one 1024-thread block per (query, tile): 2^15 16-bit counters live in 64 KB of LDS; the query's unique words
//             select one bucket each, whose postings are streamed (coalesced) into ds_add_u32 on the packed halves;
//             a final LDS sweep appends (sequence, count) for count >= minmatches to the QUERY's own record region
//             (one counter per query: a single global cursor serialises at ~40 ns per atomic and cost 7x the counting).
//   select  = one wave per query: threshold c* = the largest count with at least `tophits` records at or above it
//             (15 counting passes over the <= cap records); records >= c* -- a superset of the heap's content under any
//             tie-break -- go to a dense buffer for the host.
//             Blocks are ordered tile-major (blockIdx.x = query), so the postings of one tile (index bytes / ntiles) are
//             re-read from L2 / Infinity Cache by all queries before the next tile is touched.
//   Counts are exact integers, the host applies the reference's total order (count desc, length asc, seqno asc) and the
//   heap size, so candidate lists are bit-identical to the host restatement in vsx_search.cpp.
#include <hip/hip_runtime.h>
#include "vsx_internal.h"

typedef unsigned int u32;
typedef unsigned long long u64;
typedef u32 u32_una __attribute__((aligned(1)));

#define KM_TILE_SHIFT 15
#define KM_TILE (1u << KM_TILE_SHIFT)

// word starting at position p of a sequence of 4-bit codes: valid iff all w symbols are A/C/G/T(U) (one-hot codes),
// i.e. chrmap_mask_ambig == 0 over the window (core/unique.cpp:199-238); value as the reference builds it, first symbol
// in the most significant bit pair (2-bit map A0 C1 G2 T3, utils/maps.cpp:156-186)
// Soft masking (--qmask / --dbmask soft or dust: maskmap = map_mask_lower for every mode but "none", core/unique.cpp:198-199):
// `lower` = one bit per symbol of the set's blob, set where the ASCII symbol was anything but an upper-case A C G T U
// (vsx_kmer_case_bits_kernel); a word is valid iff none of its w bits is set.  lower == nullptr: hard masking.
__device__ __forceinline__ bool word_at(const uint8_t * __restrict__ s, int p, int w, u32 & word,
                                        const uint8_t * __restrict__ lower = nullptr, u64 base = 0)
{
  // 8 symbols per pair of unaligned dword loads (VSX_CODE_SLACK bytes follow the codes); w <= 8 here
  const u32 lo = *reinterpret_cast<const u32_una *>(s + p);
  const u32 hi = *reinterpret_cast<const u32_una *>(s + p + 4);
  u32 v = 0;
  bool ok = true;
#pragma unroll
  for (int x = 0; x < 8; ++x)
    if (x < w)
      {
        const u32 c = ((x < 4 ? lo : hi) >> (8 * (x & 3))) & 0xffu;
        ok = ok && (c == 1u || c == 2u || c == 4u || c == 8u);
        v = (v << 2) | ((c >> 1) - (c >> 3));
      }
  word = v;
  if (lower)
    {
      const u64 i = base + (u64) p;
      const uint8_t * b = lower + (i >> 3);
      const u32 b0 = *reinterpret_cast<const u32_una *>(b);
      const u32 wnd = (b0 >> (u32) (i & 7)) & 0xffffu;        // 16 bits >= the 8 of a word, whatever the phase
      ok = ok && ((wnd & ((1u << w) - 1u)) == 0u);
    }
  return ok;
}

// one lane per 8 symbols of the ASCII blob -> one byte of the case bitmap
__global__ void __launch_bounds__(256)
vsx_kmer_case_bits_kernel(const uint8_t * __restrict__ ascii, u64 nbytes, uint8_t * __restrict__ bits, int fold)
{
  const u64 g = (u64) blockIdx.x * blockDim.x + threadIdx.x;
  if (g * 8 >= nbytes) return;
  u32 m = 0;
#pragma unroll
  for (int x = 0; x < 8; ++x)
    {
      const u64 i = g * 8 + (u64) x;
      u32 c = i < nbytes ? (u32) ascii[i] : (u32) 'A';
      if (fold && c >= 'a' && c <= 'z') c -= 32u;                 // DUST upper-cases the sequence first (core/mask.cpp:137-144)
      const bool upper_nt = (c == 'A') || (c == 'C') || (c == 'G') || (c == 'T') || (c == 'U');
      m |= (upper_nt ? 0u : 1u) << x;
    }
  bits[g] = (uint8_t) m;
}

// One wave per sequence.  FILL = false: bucket histogram; FILL = true: scatter the sequence number into its buckets.
// A word counts once per sequence (unique_count): first setter of the word's bit in a per-wave LDS bitmap emits.
template <bool FILL>
__global__ void __launch_bounds__(256)
vsx_kmer_sweep_kernel(const uint8_t * __restrict__ codes, const u64 * __restrict__ off, const u32 * __restrict__ len,
                      const u32 * __restrict__ seq_list, u32 nseq,
                      int w, u32 ntiles, u32 * __restrict__ bucket_count, const u64 * __restrict__ bucket_start,
                      uint16_t * __restrict__ postings, const uint8_t * __restrict__ lower)
{
  extern __shared__ u32 bm_all[];                       // 4 waves x (4^w / 32) words
  const int lane = (int) (threadIdx.x & 63), wave = (int) (threadIdx.x >> 6);
  const u32 words = (1u << (2 * w)) >> 5;
  u32 * bm = bm_all + (size_t) wave * (words ? words : 1);
  for (u32 x = lane; x < (words ? words : 1); x += 64) bm[x] = 0;
  const u32 seq = blockIdx.x * 4 + wave;                 // position in the index
  if (seq >= nseq) return;
  const u32 sid = seq_list ? seq_list[seq] : seq;        // sequence of the set it stands for (subset index: clustering)
  const u64 base = off[sid];
  const uint8_t * __restrict__ s = codes + base;
  const int L = (int) len[sid];
  const u32 tile = seq >> KM_TILE_SHIFT;
  for (int p0 = 0; p0 + w <= L; p0 += 64)
    {
      const int p = p0 + lane;
      u32 word = 0;
      if (p + w <= L && word_at(s, p, w, word, lower, base))
        {
          const u32 bit = 1u << (word & 31);
          const u32 old = atomicOr(&bm[word >> 5], bit);
          if (!(old & bit))
            {
              const size_t b = (size_t) word * ntiles + tile;
              const u32 slot = atomicAdd(&bucket_count[b], 1u);
              if (FILL) postings[2 * bucket_start[b] + slot] = (uint16_t) (seq & (KM_TILE - 1));   // bucket_start counts dwords
            }
        }
    }
  // leave the bitmap clean for nobody (one sequence per wave): nothing to do
}

// counters of one (query, tile) in LDS; see the header comment
#define KM_UNROLL 8                    // independent postings loads per lane and trip
#define KM_COUNT_THREADS 1024          // 16 waves share one 64 KB counter tile: two blocks = 32 waves per CU keep the postings stream deep
__global__ void __launch_bounds__(KM_COUNT_THREADS)
vsx_kmer_count_kernel(const u32 * __restrict__ postings, const u64 * __restrict__ bucket_start, u32 ntiles, u32 nseq,
                      const u64 * __restrict__ qk_start, const u32 * __restrict__ qk, const u32 * __restrict__ minmatch,
                      const u32 * __restrict__ qlist, uint2 * __restrict__ rec, u32 cap, u32 * __restrict__ qcount)
{
  __shared__ u32 cnt[KM_TILE / 2];                      // two 16-bit counters per dword: 64 KB
  __shared__ u64 rs[256];                                // first posting of each selected bucket
  __shared__ u32 pre[257];                               // exclusive prefix sum of the bucket sizes; pre[ne ..] = total
  const int tid = (int) threadIdx.x, lane = tid & 63;
  const u32 slot = blockIdx.x, tile = blockIdx.y;
  const u32 q = qlist ? qlist[slot] : slot;             // second pass: only the queries whose region overflowed
  const u32 mm = minmatch[q];
  if (mm == 0xffffffffu) return;                         // this query is answered on the host (minmatches == 0, > 32767 words)
  const u32 base = tile << KM_TILE_SHIFT;
  {
    uint4 * c4 = reinterpret_cast<uint4 *>(cnt);
    for (int x = tid; x < (int) (KM_TILE / 8); x += KM_COUNT_THREADS) c4[x] = make_uint4(0, 0, 0, 0);
  }
  const u64 k0 = qk_start[q];
  const int nk = (int) (qk_start[q + 1] - k0);
  for (int chunk = 0; chunk < nk; chunk += 256)
    {
      __syncthreads();
      // the buckets of up to 256 words, flattened: 256 threads fetch one range each, wave 0 scans the sizes (4 per lane)
      if (tid < 256)
        {
          u32 n = 0;
          if (chunk + tid < nk)
            {
              const size_t b = (size_t) qk[k0 + chunk + tid] * ntiles + tile;
              const u64 first = bucket_start[b];
              rs[tid] = first;
              n = (u32) (bucket_start[b + 1] - first);
            }
          pre[tid] = n;
        }
      __syncthreads();
      if (tid < 64)
        {
          u32 sz[4], run = 0;
#pragma unroll
          for (int u = 0; u < 4; ++u) { sz[u] = run; run += pre[4 * tid + u]; }     // exclusive within the lane
          u32 incl = run;
#pragma unroll
          for (int d = 1; d < 64; d <<= 1)
            {
              const u32 up = (u32) __shfl_up((int) incl, d, 64);
              if (tid >= d) incl += up;
            }
          const u32 excl = incl - run;
#pragma unroll
          for (int u = 0; u < 4; ++u) pre[4 * tid + u] = excl + sz[u];
          if (tid == 63) pre[256] = incl;
        }
      __syncthreads();
      const u32 total = pre[256];
      // each wave streams one contiguous sixteenth of the flattened postings: the bucket of a lane's index is found by
      // binary search once, afterwards it only ever steps forward; four independent loads in flight per lane
      const u32 per_wave = (total + KM_COUNT_THREADS / 64 - 1) / (KM_COUNT_THREADS / 64);
      const u32 wbeg = ((u32) (tid >> 6) * per_wave < total) ? (u32) (tid >> 6) * per_wave : total;
      const u32 wend = (wbeg + per_wave < total) ? wbeg + per_wave : total;
      if (wbeg + (u32) lane < wend)
        {
          int e = 0;
          {
            const u32 i = wbeg + (u32) lane;
#pragma unroll
            for (int bstep = 128; bstep >= 1; bstep >>= 1)
              if (pre[e + bstep] <= i) e += bstep;              // entries past the last word hold `total`
          }
          u32 lo = pre[e], hi = pre[e + 1];
          u64 r0 = rs[e];
          // software pipeline: the loads of trip k + 1 are issued before the LDS atomics of trip k
          auto fetch = [&](u32 i0, u32 (&s)[KM_UNROLL]) {
#pragma unroll
            for (int u = 0; u < KM_UNROLL; ++u)
              {
                const u32 i = i0 + 64u * (u32) u;
                s[u] = 0xffffffffu;
                if (i < wend)
                  {
                    while (i >= hi) { ++e; lo = hi; hi = pre[e + 1]; r0 = rs[e]; }
                    s[u] = postings[r0 + (i - lo)];               // two tile-local indices
                  }
              }
          };
          u32 cur[KM_UNROLL], nxt[KM_UNROLL];
          u32 i0 = wbeg + (u32) lane;
          fetch(i0, cur);
          for (;;)
            {
              i0 += 64 * KM_UNROLL;
              const bool more = i0 < wend;
              if (more) fetch(i0, nxt);
#pragma unroll
              for (int u = 0; u < KM_UNROLL; ++u)
                {
                  const u32 x0 = cur[u] & 0xffffu, x1 = cur[u] >> 16;
                  if (x0 != 0xffffu) atomicAdd(&cnt[x0 >> 1], 1u << ((x0 & 1u) * 16));
                  if (x1 != 0xffffu) atomicAdd(&cnt[x1 >> 1], 1u << ((x1 & 1u) * 16));
                }
              if (!more) break;
#pragma unroll
              for (int u = 0; u < KM_UNROLL; ++u) cur[u] = nxt[u];
            }
        }
    }
  __syncthreads();
  const u32 top = (nseq - base < KM_TILE) ? nseq - base : KM_TILE;
  for (u32 x = (u32) tid; x < KM_TILE / 2; x += KM_COUNT_THREADS)     // same trip count for every lane: the ballots below are wave-wide
    {
      const u32 v = cnt[x];
#pragma unroll
      for (int h = 0; h < 2; ++h)
        {
          const u32 c = h ? (v >> 16) : (v & 0xffffu);
          const u32 sidx = 2 * x + h;
          const bool hit = (c >= mm) && (sidx < top);
          const u64 ballot = __ballot(hit);
          if (ballot)
            {
              u32 first = 0;
              if (lane == 0) first = atomicAdd(&qcount[slot], (u32) __popcll(ballot));
              first = (u32) __shfl((int) first, 0, 64);
              if (hit)
                {
                  const u32 pos = first + (u32) __popcll(ballot & ((1ull << lane) - 1ull));
                  if (pos < cap) rec[(size_t) slot * cap + pos] = make_uint2(base + sidx, c);
                }
            }
        }
    }
}

// One wave per query slot: keep the records that can still reach the top `keep` (see the header comment).
// sel[slot] = (offset into dense, number kept); a slot whose region overflowed (qcount > cap) is left for the second pass.
__global__ void __launch_bounds__(256)
vsx_kmer_select_kernel(const uint2 * __restrict__ rec, u32 cap, const u32 * __restrict__ qcount, u32 nslots, u32 keep,
                       uint2 * __restrict__ dense, u64 * cursor, u64 capacity, uint2 * __restrict__ sel_off_n, u64 * __restrict__ sel_off)
{
  const int lane = (int) (threadIdx.x & 63);
  const u32 slot = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (slot >= nslots) return;
  const u32 n = qcount[slot];
  if (n > cap) { if (lane == 0) { sel_off[slot] = 0; sel_off_n[slot] = make_uint2(0xffffffffu, n); } return; }
  const uint2 * __restrict__ r = rec + (size_t) slot * cap;
  u32 thr = 0;
  if (n > keep)
    {
      // largest c with #(count >= c) >= keep: binary search on the 15-bit count
      u32 lo = 0, hi = 32768;                            // invariant: #(>= lo) >= keep, #(>= hi) < keep
      while (hi - lo > 1)
        {
          const u32 mid = (lo + hi) >> 1;
          u32 c = 0;
          for (u32 x = (u32) lane; x < n; x += 64) c += (r[x].y >= mid) ? 1u : 0u;
#pragma unroll
          for (int m = 32; m >= 1; m >>= 1) c += (u32) __shfl_xor((int) c, m, 64);
          if (c >= keep) lo = mid; else hi = mid;
        }
      thr = lo;
    }
  u32 m = 0;
  for (u32 x = (u32) lane; x < n; x += 64) m += (r[x].y >= thr) ? 1u : 0u;
#pragma unroll
  for (int k = 32; k >= 1; k >>= 1) m += (u32) __shfl_xor((int) m, k, 64);
  u64 first = 0;
  if (lane == 0) first = atomicAdd(cursor, (u64) m);
  first = (u64) __shfl((long long) first, 0, 64);
  if (lane == 0) { sel_off[slot] = first; sel_off_n[slot] = make_uint2(m, n); }
  if (first + m > capacity) return;                      // the host re-runs the selection with the exact size
  u32 done = 0;
  for (u32 x0 = 0; x0 < n; x0 += 64)
    {
      const u32 x = x0 + (u32) lane;
      const bool take = (x < n) && (r[x].y >= thr);
      const u64 ballot = __ballot(take);
      if (take) dense[first + done + (u32) __popcll(ballot & ((1ull << lane) - 1ull))] = r[x];
      done += (u32) __popcll(ballot);
    }
}

extern "C" hipError_t vsx_kmer_launch_sweep(int fill, const uint8_t * codes, const uint64_t * off, const uint32_t * len,
                                            const uint32_t * seq_list, uint32_t nseq, int w, uint32_t ntiles, uint32_t * bucket_count,
                                            const uint64_t * bucket_start, uint32_t * postings, const uint8_t * lower_bits,
                                            hipStream_t st)   // postings: dwords of two 16-bit indices; lower_bits: soft masking or NULL
{
  if (nseq == 0) return hipSuccess;
  const size_t lds = (size_t) 4 * (((1u << (2 * w)) >> 5) ? ((1u << (2 * w)) >> 5) : 1) * 4;
  const dim3 grid((nseq + 3) / 4), block(256);
  if (fill)
    hipLaunchKernelGGL(vsx_kmer_sweep_kernel<true>, grid, block, lds, st, codes, (const u64 *) off, len, seq_list, nseq, w, ntiles,
                       bucket_count, (const u64 *) bucket_start, (uint16_t *) postings, lower_bits);
  else
    hipLaunchKernelGGL(vsx_kmer_sweep_kernel<false>, grid, block, lds, st, codes, (const u64 *) off, len, seq_list, nseq, w, ntiles,
                       bucket_count, (const u64 *) bucket_start, (uint16_t *) postings, lower_bits);
  return hipGetLastError();
}

extern "C" hipError_t vsx_kmer_launch_case_bits(const uint8_t * d_ascii, uint64_t nbytes, uint8_t * d_bits, int fold_case, hipStream_t st)
{
  if (nbytes == 0) return hipSuccess;
  const u64 lanes = (nbytes + 7) / 8;
  hipLaunchKernelGGL(vsx_kmer_case_bits_kernel, dim3((unsigned) ((lanes + 255) / 256)), dim3(256), 0, st, d_ascii, (u64) nbytes, d_bits, fold_case);
  return hipGetLastError();
}

extern "C" hipError_t vsx_kmer_launch_count(const uint32_t * postings, const uint64_t * bucket_start, uint32_t ntiles,
                                            uint32_t nseq, uint32_t nslots, const uint64_t * qk_start, const uint32_t * qk,
                                            const uint32_t * minmatch, const uint32_t * qlist, void * rec, uint32_t cap,
                                            uint32_t * qcount, hipStream_t st)
{
  if (nslots == 0 || nseq == 0) return hipSuccess;
  hipLaunchKernelGGL(vsx_kmer_count_kernel, dim3(nslots, ntiles), dim3(KM_COUNT_THREADS), 0, st, postings,
                     (const u64 *) bucket_start, ntiles, nseq, (const u64 *) qk_start, qk, minmatch, qlist, (uint2 *) rec, cap, qcount);
  return hipGetLastError();
}

extern "C" hipError_t vsx_kmer_launch_select(const void * rec, uint32_t cap, const uint32_t * qcount, uint32_t nslots,
                                             uint32_t keep, void * dense, unsigned long long * cursor, uint64_t capacity,
                                             void * sel_m_n, uint64_t * sel_off, hipStream_t st)
{
  if (nslots == 0) return hipSuccess;
  hipLaunchKernelGGL(vsx_kmer_select_kernel, dim3((nslots + 3) / 4), dim3(256), 0, st, (const uint2 *) rec, cap, qcount, nslots,
                     keep, (uint2 *) dense, (u64 *) cursor, (u64) capacity, (uint2 *) sel_m_n, (u64 *) sel_off);
  return hipGetLastError();
}

extern "C" uint32_t vsx_kmer_tile_shift(void) { return KM_TILE_SHIFT; }
