// vsx_kmer.hip -- k-mer candidate counting on gfx950 (SURVEY.md 8f "next" #1).
//
// Replaces the reference's search_topscores (core/searchcore.cpp:260-340: one counter per database sequence, bumped
// once per (unique query word, sequence containing the word), then the sequences with count >= minmatches go to the
// top-N heap) and the index it walks (core/dbindex.cpp:125-255, core/unique.cpp:155-352) by HBM/LDS-bound integer
// kernels.  No MFMA: this is a sparse count, not a contraction.
//
//   index   = postings grouped by (word, TILE of consecutive sequence numbers): a CSR over 4^w x ntiles buckets, in one of three
//             formats (all in 16-byte units, so the count kernels stream whole units and never test for a sentinel):
//             * PACKED (the whole-set index of a search, word lengths 3..8; "PACKED postings" below): sorted counter indices as a
//               16-bit first value + 14 one-byte gaps per unit, tiles of 32 630 sequences -- 1.1 bytes per posting;
//             * 16-BIT (subset indexes, rebuilt every clustering round): tile-local sequence indices, eight per unit, tiles of 2^15,
//               the last unit of a bucket padded with the index of a spare counter; built in two sweeps (count, fill), the order
//               inside a bucket arbitrary (counting commutes), so no sort is needed;
//             * TAGGED (word lengths 9..15): dwords tag << 16 | index, four per unit.
//   count   = one block per (query, tile): 2^15 counters live in LDS -- 8 bits each (32 KB, 512 threads, four blocks per CU) for
//             queries with at most 255 unique words, 16 bits (64 KB, 1024 threads) otherwise; the query's unique words
//             select one bucket each, whose postings are streamed (coalesced 16-byte loads) into ds_add_u32 on the packed fields;
//             a final LDS sweep writes (sequence, count) for count >= minmatches to the (query, tile)'s own record sub-region
//             (no atomics: a single global cursor serialised at ~40 ns per atomic and cost 7x the counting, and even one
//             atomic with return per block is a latency the block cannot hide).
//   select  = one wave per query: threshold c* = the largest count with at least `tophits` records at or above it
//             (one histogram pass over the records); records >= c* -- a superset of the heap's content under any
//             tie-break -- go to a dense buffer for the host.
//             Blocks are ordered tile-major (blockIdx.x = query), so the postings of one tile (index bytes / ntiles) are
//             re-read from L2 / Infinity Cache by all queries before the next tile is touched.
//   Counts are exact integers, the host applies the reference's total order (count desc, length asc, seqno asc) and the
//   heap size, so candidate lists are bit-identical to the host restatement in vsx_search.cpp.
#include <hip/hip_runtime.h>
#include <cstdlib>
#include <rocprim/device/device_radix_sort.hpp>
#include "vsx_internal.h"
#include "vsx_kmer_pack.h"

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
              if (FILL) postings[8 * bucket_start[b] + slot] = (uint16_t) (seq & (KM_TILE - 1));   // bucket_start counts 16-byte units
            }
        }
    }
  // leave the bitmap clean for nobody (one sequence per wave): nothing to do
}

// ---- word lengths 9..15 (r03): TAGGED postings -------------------------------------------------------------------------------
// A bucket table over 4^w words stops being affordable at w = 12 (4.2 GB for 31 tiles) and impossible at 15.  The index of a
// longer word length keeps the 4^8 x ntiles table: a word goes to the bucket of its LAST EIGHT symbols (its low 16 bits) and its
// posting carries the remaining symbols as a tag -- one dword (tag << 16 | tile-local sequence), four per 16-byte unit, pad
// 0xFFFF8000 (a tag no word has).  The count kernel streams the bucket of a query word's low 16 bits and bumps the counter only
// where the tag equals the word's high part: exact counts, the same kernel structure, twice the bytes of the w = 8 index per
// posting.  Build: per tile, every position's (bucket, tag, sequence) key -> radix sort (rocPRIM: a library sort of ~32 M keys,
// off the search path) -> adjacent equal keys are the repeats of a word inside one sequence (unique_count, core/unique.cpp:155-352)
// -> first of each run counted / scattered.  Two passes (count, fill) like the short-word sweep.
typedef unsigned long long u64k;
#define KM_TAG_BUCKET_BITS 16
__device__ __forceinline__ bool long_word_at(const uint8_t * __restrict__ s, int p, int w, u32 & word,
                                             const uint8_t * __restrict__ lower, u64 base)
{
  u32 v = 0;
  bool ok = true;
  for (int x = 0; x < w; ++x)
    {
      const u32 c = s[p + x];
      ok = ok && (c == 1u || c == 2u || c == 4u || c == 8u);
      v = (v << 2) | ((c >> 1) - (c >> 3));
    }
  word = v;
  if (lower)
    {
      const u64 i = base + (u64) p;
      const uint8_t * b = lower + (i >> 3);
      const u32 b0 = *reinterpret_cast<const u32_una *>(b);
      const u32 wnd = (b0 >> (u32) (i & 7)) & 0xffffffu;        // 24 bits >= the 15 of a word, whatever the phase
      ok = ok && ((wnd & ((1u << w) - 1u)) == 0u);
    }
  return ok;
}

// one wave per sequence of the tile: key of every position -> keys[position in the tile's blob range]; invalid positions hold ~0
__global__ void __launch_bounds__(256)
vsx_kmer_keys_kernel(const uint8_t * __restrict__ codes, const u64 * __restrict__ off, const u32 * __restrict__ len, u32 first_seq,
                     u32 nseq_tile, int w, const uint8_t * __restrict__ lower, const u64 * __restrict__ slot_of, u64k * __restrict__ keys)
{
  const int lane = (int) (threadIdx.x & 63);
  const u32 local = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (local >= nseq_tile) return;
  const u32 sid = first_seq + local;
  const u64 base = off[sid];
  const uint8_t * __restrict__ s = codes + base;
  const int L = (int) len[sid];
  u64k * __restrict__ out = keys + (slot_of[sid] - slot_of[first_seq]);        // slot_of = running sum of the lengths: any blob layout works
  for (int p0 = 0; p0 < L; p0 += 64)
    {
      const int p = p0 + lane;
      if (p >= L) break;
      u32 word = 0;
      u64k key = ~0ull;
      if (p + w <= L && long_word_at(s, p, w, word, lower, base))
        key = ((u64k) (word & 0xffffu) << 32) | ((u64k) (word >> KM_TAG_BUCKET_BITS) << 16) | (u64k) local;      // bucket | tag | tile-local sequence
      out[p] = key;
    }
}

// sorted keys: the first of each run of equal keys is one (word, sequence) pair
template <bool FILL>
__global__ void __launch_bounds__(256)
vsx_kmer_runs_kernel(const u64k * __restrict__ keys, u64 n, u32 tile, u32 ntiles, u32 * __restrict__ bucket_count,
                     const u64 * __restrict__ bucket_start, u32 * __restrict__ postings)
{
  const u64 i = (u64) blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const u64k k = keys[i];
  if (k == ~0ull || (i > 0 && keys[i - 1] == k)) return;
  const size_t b = (size_t) (k >> 32) * ntiles + tile;
  const u32 slot = atomicAdd(&bucket_count[b], 1u);
  if (FILL) postings[4 * bucket_start[b] + slot] = (u32) (k & 0xffffffffu);       // tag << 16 | sequence; bucket_start counts 16-byte units
}

// ---- counting: the counters of one (query, tile) live in LDS; see the header comment -------------------------------------
// Postings format (r03): a bucket is a whole number of 16-BYTE units (bucket_start counts uint4), i.e. 8 tile-local
// indices per unit; the last unit of a bucket is padded with KM_PAD = 0x8000 -- the index of a SPARE counter one past the tile,
// so the hot loop carries no sentinel test: every posting is one unconditional ds_add_u32.
// Two counter widths, chosen per query by the host:
//   BITS = 8  (queries with <= 255 unique words, i.e. every BASELINE read length): four counters per dword, 32 KB per tile,
//             512 threads -> FOUR blocks per CU (32 waves).  A count never exceeds the number of words, so bytes cannot carry;
//   BITS = 16 (longer queries): two counters per dword, 64 KB, 1024 threads, two blocks per CU.
// r03 measurements that shaped it (profiles/r03/r03_ubench_lds.txt, r03e_kmer_probe.txt; 100 k x 250 bp vs 1 M x 1 kbp):
//   * random-address ds_add_u32 sustains 8.1 ops per clock and CU (13 conflict-free); the r02 kernel reached 3.2;
//   * r02's flattened equal split looked its bucket up in LDS for every load; a look-up waits for lgkmcnt(0), i.e. for all the
//     atomics queued before it, so a wave's atomics and its next loads took turns: 192 ms.  Per-wave buckets whose ranges are
//     broadcast with v_readlane (no LDS read in the streaming loop): 142 ms;
//   * of those 142 ms, 73 remained with neither loads nor atomics: a block's fixed latency chain -- query -> its words -> their
//     bucket ranges (three dependent global loads), barriers, and a global atomic with return in the sweep.  Now a pre-pass
//     (vsx_kmer_ranges_kernel) writes the ranges of every (query, tile) in the order the waves want them, so a block starts with
//     ONE coalesced load, and every (query, tile) owns a sub-region of the record buffer, so the sweep needs no atomic.
#define KM_PAD 0x8000u
#ifndef KM_LOADS
#define KM_LOADS 4                     // independent 16-byte loads per lane and trip (32 postings)
#endif

// Bucket ranges of the 8-bit class, one block per query slot, thread i = the query's i-th word.  R[(tile * nslots + slot) * 256 +
// (i % 8) * 32 + i / 8] = (first unit, units): wave w of the count kernel reads entries w * 32 .. w * 32 + 31 -- its buckets w,
// w + 8, ... -- with one coalesced load.  The ranges of one word over all tiles are contiguous in bucket_start (word-major).
__global__ void __launch_bounds__(256)
vsx_kmer_ranges_kernel(const u64 * __restrict__ bucket_start, u32 ntiles, const u64 * __restrict__ qk_start, const u32 * __restrict__ qk,
                       const u32 * __restrict__ minmatch, const u32 * __restrict__ qlist, u32 nslots, uint2 * __restrict__ R)
{
  const u32 slot = blockIdx.x, i = threadIdx.x;
  const u32 q = qlist ? qlist[slot] : slot;
  const u64 k0 = qk_start[q];
  const u32 nk = (minmatch[q] == 0xffffffffu) ? 0u : (u32) (qk_start[q + 1] - k0);
  const u32 p = (i & 7u) * 32u + (i >> 3);
  if (i < nk)
    {
      // eight tiles per trip: the nine loads are independent (one after the other they were 31 dependent latencies per thread)
      const u64 * __restrict__ row = bucket_start + (size_t) qk[k0 + i] * ntiles;
      for (u32 t0 = 0; t0 < ntiles; t0 += 8)
        {
          u64 v[9];
#pragma unroll
          for (int x = 0; x < 9; ++x) v[x] = row[t0 + (u32) x <= ntiles ? t0 + (u32) x : ntiles];
#pragma unroll
          for (int x = 0; x < 8; ++x)
            if (t0 + (u32) x < ntiles)
              R[((size_t) (t0 + (u32) x) * nslots + slot) * 256 + p] = make_uint2((u32) v[x], (u32) (v[x + 1] - v[x]));
        }
    }
  else
    for (u32 t = 0; t < ntiles; ++t) R[((size_t) t * nslots + slot) * 256 + p] = make_uint2(0u, 0u);
}

// PRE = the ranges come from vsx_kmer_ranges_kernel (8-bit class); otherwise the block looks them up itself, 256 words at a time.
// Records: (query slot, tile) owns rec[(slot * ntiles + tile) * subcap ..) and tile_count[slot * ntiles + tile] (the number of
// counters at or above the threshold, also when it exceeds subcap: the host then repeats the slot with larger sub-regions).
// TAG = tagged postings (word lengths 9..15, see above): one dword per posting, counted where its tag equals the query word's
template <int BITS, bool PRE, bool TAG = false>
__global__ void __launch_bounds__(BITS == 8 ? 512 : 1024) __attribute__((amdgpu_waves_per_eu(8, 8)))     // 64 VGPRs: 32 waves per CU in both widths
vsx_kmer_count_kernel(const uint4 * __restrict__ postings, const u64 * __restrict__ bucket_start, const uint2 * __restrict__ R,
                      u32 ntiles, u32 nseq, const u64 * __restrict__ qk_start, const u32 * __restrict__ qk, const u32 * __restrict__ minmatch,
                      const u32 * __restrict__ qlist, u32 slot_base, u32 nslots, uint2 * __restrict__ rec, u32 subcap,
                      u32 * __restrict__ tile_count, int probe)
{
  // probe (VSX_KMER_PROBE, measurements only -- results are wrong): bit 0 = no LDS atomics (the loads are still consumed),
  // bit 1 = no postings loads (addresses synthesised), bit 2 = no final sweep
  constexpr int THREADS = (BITS == 8) ? 512 : 1024;
  constexpr int WAVES = THREADS / 64;
  constexpr int PER = 32 / BITS;                                  // counters per dword
  constexpr int NDW = (int) (KM_TILE / PER);                      // dwords of real counters
  constexpr int BPW = 256 / WAVES;                                // buckets per wave and chunk of 256 words
  static_assert(!PRE || BITS == 8, "the range table is laid out for 8 waves");
  static_assert(!(PRE && TAG), "tagged indexes look their ranges up in the block");
  __shared__ __attribute__((aligned(16))) u32 cnt[NDW + 4];       // + the spare dword the pad index lands in
  __shared__ uint2 rng[PRE ? 1 : 256];                            // !PRE: (first unit, units) of each selected bucket
  __shared__ u32 tagL[TAG ? 256 : 1];                             // TAG: the high part of each selected word
  __shared__ u32 wave_hits[WAVES];
  const int tid = (int) threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const u32 srel = blockIdx.x, slot = srel + slot_base, tile = blockIdx.y;        // srel indexes R / rec / tile_count
  const u32 q = qlist ? qlist[slot] : slot;
  const u32 mm = minmatch[q];
  const size_t region = (size_t) srel * ntiles + tile;
  if (mm == 0xffffffffu)                                          // this query is answered on the host (minmatches == 0, > 32767 words)
    {
      if (tid == 0) tile_count[region] = 0;
      return;
    }
  // the wave's bucket ranges: lane j holds bucket wave + WAVES * j of the chunk (PRE: straight from the table)
  uint2 mine_rng = make_uint2(0u, 0u);
  u32 mine_tag = 0;
  if (PRE && lane < BPW) mine_rng = R[((size_t) tile * nslots + srel) * 256 + (u32) (wave * BPW + lane)];
  const u32 base = tile << KM_TILE_SHIFT;
  {
    uint4 * c4 = reinterpret_cast<uint4 *>(cnt);
    for (int x = tid; x < (NDW + 4) / 4; x += THREADS) c4[x] = make_uint4(0, 0, 0, 0);
  }
  const u64 k0 = PRE ? 0 : qk_start[q];
  const int nk = PRE ? 256 : (int) (qk_start[q + 1] - k0);        // PRE: absent words have empty ranges
  auto bump = [&](u32 x) __attribute__((always_inline)) {
    // counter x: dword x / PER, field x % PER
    if (BITS == 8) atomicAdd(&cnt[x >> 2], 1u << ((x & 3u) << 3));
    else atomicAdd(&cnt[x >> 1], 1u << ((x & 1u) << 4));
  };
  for (int chunk = 0; chunk < nk; chunk += 256)
    {
      __syncthreads();                                              // counters cleared / previous chunk's ranges consumed
      if (!PRE)
        {
          if (tid < 256)
            {
              uint2 r = make_uint2(0u, 0u);
              u32 tg = 0;
              if (chunk + tid < nk)
                {
                  const u32 word = qk[k0 + chunk + tid];
                  const size_t b = (size_t) (TAG ? (word & 0xffffu) : word) * ntiles + tile;
                  const u64 first = bucket_start[b];
                  r = make_uint2((u32) first, (u32) (bucket_start[b + 1] - first));
                  tg = word >> KM_TAG_BUCKET_BITS;
                }
              rng[tid] = r;
              if (TAG) tagL[tid] = tg;
            }
          __syncthreads();
          mine_rng = (lane < BPW) ? rng[wave + WAVES * lane] : make_uint2(0u, 0u);
          if (TAG) mine_tag = (lane < BPW) ? tagL[wave + WAVES * lane] : 0u;
        }
      // Wave w takes the buckets w, w + WAVES, ...; the loop below broadcasts lane j's range with v_readlane -- the streaming
      // loop touches the LDS with ds_add_u32 ONLY
      u32 sink = 0;
      auto consume = [&](const uint4 & v, u32 tag) __attribute__((always_inline)) {
        const u32 w4[4] = {v.x, v.y, v.z, v.w};
        if (probe & 1) sink ^= w4[0] + w4[1] + w4[2] + w4[3];
        else if (TAG)
          {
#pragma unroll
            for (int d = 0; d < 4; ++d) if ((w4[d] >> 16) == tag) bump(w4[d] & 0xffffu);
          }
        else
          {
#pragma unroll
            for (int d = 0; d < 4; ++d) { bump(w4[d] & 0xffffu); bump(w4[d] >> 16); }
          }
      };
      auto unit = [&](u32 idx) __attribute__((always_inline)) -> uint4 {
        if (probe & 2) { const u32 z = (idx * 2654435761u) & 0x7fff7fffu; return make_uint4(z, (z ^ 0x01230456u) & 0x7fff7fffu, (z ^ 0x10002000u) & 0x7fff7fffu, (z ^ 0x5a5a2b2bu) & 0x7fff7fffu); }
        return postings[idx];
      };
      // first 64 units of KM_LOADS buckets per trip (a bucket of the bench shape has ~61 units); the loads of trip t + 1 are
      // issued before the atomics of trip t.  Longer buckets: their remaining units follow in a tail loop.
      auto fetch = [&](int j0, uint4 (&s4)[KM_LOADS], bool (&on)[KM_LOADS]) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < KM_LOADS; ++u)
          {
            on[u] = false;
            if (j0 + u < BPW)
              {
                const u32 n = (u32) __builtin_amdgcn_readlane((int) mine_rng.y, j0 + u);
                const u32 r0 = (u32) __builtin_amdgcn_readlane((int) mine_rng.x, j0 + u);
                on[u] = (u32) lane < n;
                if (on[u]) s4[u] = unit(r0 + (u32) lane);
              }
          }
      };
      uint4 cur[KM_LOADS], nxt[KM_LOADS];
      bool con[KM_LOADS], non[KM_LOADS];
      fetch(0, cur, con);
      for (int j0 = 0; j0 < BPW; j0 += KM_LOADS)
        {
          const bool more = j0 + KM_LOADS < BPW;
          if (more) fetch(j0 + KM_LOADS, nxt, non);
#pragma unroll
          for (int u = 0; u < KM_LOADS; ++u)
            if (con[u]) consume(cur[u], TAG && j0 + u < BPW ? (u32) __builtin_amdgcn_readlane((int) mine_tag, j0 + u) : 0u);
          // tails of this trip's buckets (rare for sizes around one wave-load; wave-uniform trip counts)
#pragma unroll
          for (int u = 0; u < KM_LOADS; ++u)
            if (j0 + u < BPW)
              {
                const u32 n = (u32) __builtin_amdgcn_readlane((int) mine_rng.y, j0 + u);
                if (n > 64u)
                  {
                    const u32 r0 = (u32) __builtin_amdgcn_readlane((int) mine_rng.x, j0 + u);
                    const u32 tg = TAG ? (u32) __builtin_amdgcn_readlane((int) mine_tag, j0 + u) : 0u;
                    for (u32 i = 64u + (u32) lane; i < n; i += 64u) consume(unit(r0 + i), tg);
                  }
              }
          if (more)
            {
#pragma unroll
              for (int u = 0; u < KM_LOADS; ++u) { cur[u] = nxt[u]; con[u] = non[u]; }
            }
        }
      if ((probe & 1) && sink == 0x9e3779b9u) cnt[NDW] = sink;       // keeps the probe's loads alive
    }
  __syncthreads();
  if (probe & 4) { if (tid == 0) tile_count[region] = 0; return; }
  // ---- sweep: counters >= mm -> (sequence, count) records in the (query, tile) sub-region.  Hits are rare (a fraction of a per
  // cent of the counters), so a lane first tests whole dwords (SWAR for the byte counters); a wave-level prefix sum and one
  // exchange through LDS give every lane its position -- no atomic, the sub-region belongs to this block.
  const u32 top = (nseq - base < KM_TILE) ? nseq - base : KM_TILE;
  constexpr int DW_PER_THREAD = NDW / THREADS;                    // 16 in both configurations
  static_assert(DW_PER_THREAD * THREADS == NDW && DW_PER_THREAD % 4 == 0, "the sweep reads whole uint4s");
  auto field = [&](u32 v, int h) -> u32 { return (BITS == 8) ? ((v >> (8 * h)) & 0xffu) : ((v >> (16 * h)) & 0xffffu); };
  auto any_hit = [&](u32 v) -> bool {
    if (BITS == 16) return ((v & 0xffffu) >= mm) || ((v >> 16) >= mm);
    if (mm > 255u) return false;
    // bytes >= mm, for 1 <= mm <= 255: split at 128 so that the per-byte addition cannot carry into the next byte
    const u32 hi7 = v & 0x80808080u, lo7 = v & 0x7f7f7f7fu;
    if (mm <= 128u) return (hi7 | ((lo7 + (128u - mm) * 0x01010101u) & 0x80808080u)) != 0u;
    return (hi7 & (lo7 + (256u - mm) * 0x01010101u)) != 0u;
  };
  u32 mine = 0;
  const uint4 * c4 = reinterpret_cast<const uint4 *>(cnt);
  for (int g4 = 0; g4 < DW_PER_THREAD / 4; ++g4)
    {
      const uint4 v4 = c4[g4 * THREADS + tid];
      const u32 vv[4] = {v4.x, v4.y, v4.z, v4.w};
#pragma unroll
      for (int d = 0; d < 4; ++d)
        if (any_hit(vv[d]))
          {
            const u32 dw = (u32) (g4 * THREADS + tid) * 4u + (u32) d;
#pragma unroll
            for (int h = 0; h < PER; ++h)
              if (field(vv[d], h) >= mm && dw * PER + (u32) h < top) ++mine;
          }
    }
  u32 incl = mine;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1)
    {
      const u32 up = (u32) __shfl_up((int) incl, d, 64);
      if (lane >= d) incl += up;
    }
  if (lane == 63) wave_hits[wave] = incl;
  __syncthreads();
  u32 before = 0, total = 0;
#pragma unroll
  for (int w2 = 0; w2 < WAVES; ++w2) { const u32 t = wave_hits[w2]; before += (w2 < wave) ? t : 0u; total += t; }
  if (tid == 0) tile_count[region] = total;
  if (mine)
    {
      u32 pos = before + incl - mine;
      uint2 * __restrict__ out = rec + region * subcap;
      for (int g4 = 0; g4 < DW_PER_THREAD / 4; ++g4)
        {
          const uint4 v4 = c4[g4 * THREADS + tid];
          const u32 vv[4] = {v4.x, v4.y, v4.z, v4.w};
#pragma unroll
          for (int d = 0; d < 4; ++d)
            if (any_hit(vv[d]))
              {
                const u32 dw = (u32) (g4 * THREADS + tid) * 4u + (u32) d;
#pragma unroll
                for (int h = 0; h < PER; ++h)
                  {
                    const u32 c = field(vv[d], h);
                    const u32 sidx = dw * PER + (u32) h;
                    if (c >= mm && sidx < top)
                      {
                        if (pos < subcap) out[pos] = make_uint2(base + sidx, c);
                        ++pos;
                      }
                  }
              }
        }
    }
}

// ---- PACKED postings (r03; word lengths 3..8, the whole-set index) --------------------------------------------------------------
// The 16-bit format streams 2 bytes per posting and the count kernel is bound by exactly that stream (6.3 TB/s = what HBM
// delivers).  Packed format: a bucket's tile-local counter indices SORTED, one 16-byte unit = a 16-bit first index + 14 one-byte
// gaps = 15 postings (8.5 bits each).  Two things make every byte a plain increment -- no escape code, count field or predicate:
//   * every 252nd counter of a tile (index = 251 mod 252: an odd index in the TOP byte of its dword) is a DUMMY that stands for no
//     sequence.  A gap of more than 255 hops over dummies (one harmless increment each; a carry out of a top byte or top half
//     leaves the dword), and the unused tail of a bucket's last unit is gaps of 0 on a dummy.  A tile therefore holds
//     130 x 251 = 32 630 sequences;
//   * sequence s of the tile owns counter (s mod 130) * 252 + s / 130: NEIGHBOURS in the database sit 252 counters apart.  Related
//     sequences are usually adjacent, so the postings of a word come in clusters with thousands of counters in between -- ten hops
//     per cluster; transposed, the gaps of any word are spread evenly (bench database: 1.23 -> 1.1x bytes per posting);
//   * the sweep clears the dummies before it looks for hits.
// The units of one wave's buckets are dealt to its lanes as ONE run (lane = position in the concatenation mod 64): a bucket of
// the bench shape has ~35 units, and a trip per bucket would leave half the lanes idle while costing the same instructions.
// Build (per tile, like the tagged index): keys (word << 16 | counter index) of every position -> radix sort -> one thread per
// word walks its run of distinct keys and counts / writes the units.
// (the format's constants, the counter <-> sequence maps and the encoder: vsx_kmer_pack.h, shared with a host entry for the CPU suite)

__global__ void __launch_bounds__(256)
vsx_kmer_pk_keys_kernel(const uint8_t * __restrict__ codes, const u64 * __restrict__ off, const u32 * __restrict__ len, u32 first_seq,
                        u32 nseq_tile, int w, const uint8_t * __restrict__ lower, const u64 * __restrict__ slot_of, u32 * __restrict__ keys)
{
  const int lane = (int) (threadIdx.x & 63);
  const u32 local = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (local >= nseq_tile) return;
  const u32 sid = first_seq + local;
  const u64 base = off[sid];
  const uint8_t * __restrict__ s = codes + base;
  const int L = (int) len[sid];
  const u32 idx = km_pk_counter_of(local);                          // see the header comment: neighbours land 252 counters apart
  u32 * __restrict__ out = keys + (slot_of[sid] - slot_of[first_seq]);
  for (int p0 = 0; p0 < L; p0 += 64)
    {
      const int p = p0 + lane;
      if (p >= L) break;
      u32 word = 0;
      u32 key = 0xffffffffu;                                       // (no word has counter index 0xffff)
      if (p + w <= L && word_at(s, p, w, word, lower, base)) key = (word << 16) | idx;
      out[p] = key;
    }
}

// One thread per word: its distinct keys in the sorted array -> units.  FILL = false: bucket_count[b] = units | postings << 16.
template <bool FILL>
__global__ void __launch_bounds__(64)
vsx_kmer_pk_walk_kernel(const u32 * __restrict__ keys, u64 n, u32 tile, u32 ntiles, u32 nwords, u32 * __restrict__ bucket_count,
                        const u64 * __restrict__ bucket_start, uint4 * __restrict__ postings)
{
  const u32 word = blockIdx.x * 64 + threadIdx.x;
  if (word >= nwords) return;
  // first key of the word
  u64 lo = 0, hi = n;
  const u32 want = word << 16;
  while (lo < hi) { const u64 mid = (lo + hi) >> 1; if (keys[mid] < want) lo = mid + 1; else hi = mid; }
  const size_t b = (size_t) word * ntiles + tile;
  KmPkEncoder enc(FILL ? reinterpret_cast<KmPkUnit *>(postings + bucket_start[b]) : nullptr);
  u32 real = 0;
  u32 prev = 0xffffffffu;
  for (u64 i = lo; i < n; ++i)
    {
      const u32 k = keys[i];
      if ((k >> 16) != word || k == 0xffffffffu) break;
      if (k == prev) continue;                                      // the same word again in the same sequence (unique_count)
      prev = k;
      ++real;
      enc.push(k & 0xffffu);
    }
  enc.finish();
  const u32 units = enc.units;
  if (!FILL) bucket_count[b] = units | (real << 16);
}

// Counting on the packed index.  PRE as in vsx_kmer_count_kernel (the 8-bit class reads its ranges from the pre-pass table).
// PROBE (timing only, results are wrong): the 16-bit index read as if it were packed, 9/16 of every bucket's units.
template <int BITS, bool PRE, bool PROBE = false>
__global__ void __launch_bounds__(BITS == 8 ? 512 : 1024) __attribute__((amdgpu_waves_per_eu(8, 8)))
vsx_kmer_count_packed_kernel(const uint4 * __restrict__ postings, const u64 * __restrict__ bucket_start, const uint2 * __restrict__ R,
                             u32 ntiles, u32 nseq, const u64 * __restrict__ qk_start, const u32 * __restrict__ qk, const u32 * __restrict__ minmatch,
                             const u32 * __restrict__ qlist, u32 slot_base, u32 nslots, uint2 * __restrict__ rec, u32 subcap,
                             u32 * __restrict__ tile_count, int probe)
{
  constexpr int THREADS = (BITS == 8) ? 512 : 1024;
  constexpr int WAVES = THREADS / 64;
  constexpr int PER = 32 / BITS;
  constexpr int NDW = (int) (KM_TILE / PER);
  constexpr int BPW = 256 / WAVES;
  static_assert(!PRE || BITS == 8, "the range table is laid out for 8 waves");
  __shared__ __attribute__((aligned(16))) u32 cnt[NDW + 4];
  __shared__ uint2 rng[PRE ? 1 : 256];
  __shared__ u32 wave_hits[WAVES];
  const int tid = (int) threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const u32 srel = blockIdx.x, slot = srel + slot_base, tile = blockIdx.y;
  const u32 q = qlist ? qlist[slot] : slot;
  const u32 mm = minmatch[q];
  const size_t region = (size_t) srel * ntiles + tile;
  if (mm == 0xffffffffu)
    {
      if (tid == 0) tile_count[region] = 0;
      return;
    }
  uint2 mine = make_uint2(0u, 0u);
  if (PRE && lane < BPW) mine = R[((size_t) tile * nslots + srel) * 256 + (u32) (wave * BPW + lane)];
  {
    uint4 * c4 = reinterpret_cast<uint4 *>(cnt);
    for (int x = tid; x < (NDW + 4) / 4; x += THREADS) c4[x] = make_uint4(0, 0, 0, 0);
  }
  const u64 k0 = PRE ? 0 : qk_start[q];
  const int nk = PRE ? 256 : (int) (qk_start[q + 1] - k0);
  auto bump = [&](u32 x) __attribute__((always_inline)) {
    if (BITS == 8) atomicAdd(&cnt[x >> 2], 1u << ((x & 3u) << 3));
    else atomicAdd(&cnt[x >> 1], 1u << ((x & 1u) << 4));
  };
  u32 sink = 0;
  auto consume = [&](const uint4 & v) __attribute__((always_inline)) {
    if (probe & 1) { sink ^= v.x + v.y + v.z + v.w; return; }
    u32 acc = v.x & 0xffffu;
    if (PROBE) acc &= 0x7fffu;
    bump(acc);
    acc += (v.x >> 16) & 0xffu; if (PROBE) acc &= 0x7fffu; bump(acc);
    acc += v.x >> 24; if (PROBE) acc &= 0x7fffu; bump(acc);
    const u32 d3[3] = {v.y, v.z, v.w};
#pragma unroll
    for (int e = 0; e < 3; ++e)
#pragma unroll
      for (int b = 0; b < 4; ++b)
        {
          acc += (d3[e] >> (8 * b)) & 0xffu;
          if (PROBE) acc &= 0x7fffu;
          bump(acc);
        }
  };
  for (int chunk = 0; chunk < nk; chunk += 256)
    {
      __syncthreads();                                              // counters cleared / previous chunk's ranges consumed
      if (!PRE)
        {
          if (tid < 256)
            {
              uint2 r = make_uint2(0u, 0u);
              if (chunk + tid < nk)
                {
                  const size_t b = (size_t) qk[k0 + chunk + tid] * ntiles + tile;
                  const u64 first = bucket_start[b];
                  r = make_uint2((u32) first, (u32) (bucket_start[b + 1] - first));
                }
              rng[tid] = r;
            }
          __syncthreads();
          mine = (lane < BPW) ? rng[wave + WAVES * lane] : make_uint2(0u, 0u);
        }
      u32 n = mine.y;
      if (PROBE) n = (n * 9u + 15u) >> 4;
      // position of each of the wave's buckets in its run of units
      u32 incl = n;
#pragma unroll
      for (int d = 1; d < BPW; d <<= 1)
        {
          const u32 up = (u32) __shfl_up((int) incl, d, 64);
          if (lane >= d) incl += up;
        }
      const u32 cum = incl - n;
      const u32 U = (u32) __builtin_amdgcn_readlane((int) incl, BPW - 1);
      const u32 T = (U + 63u) >> 6;
      // unit of this lane in trip t: the bucket whose run covers position 64 t + lane (a wave-uniform walk over the buckets the
      // trip touches; jb = the first bucket that is not finished yet)
      int jb = 0;
      auto fetch = [&](u32 t, uint4 & v, bool & on) __attribute__((always_inline)) {
        const u32 g = t * 64u + (u32) lane, lim = (t + 1u) * 64u;
        u32 a = 0xffffffffu;
        int j = __builtin_amdgcn_readfirstlane(jb);
        while (j < BPW)
          {
            const u32 c = (u32) __builtin_amdgcn_readlane((int) cum, j);
            if (c >= lim) break;
            const u32 nn = (u32) __builtin_amdgcn_readlane((int) n, j);
            const u32 r0 = (u32) __builtin_amdgcn_readlane((int) mine.x, j);
            const u32 rel = g - c;
            if (rel < nn) a = r0 + rel;
            if (c + nn > lim) break;                                   // continues in the next trip
            ++j;
          }
        jb = j;
        on = a != 0xffffffffu;
        if (on) v = postings[a];
      };
      // three trips in flight per lane
      uint4 b0 = make_uint4(0, 0, 0, 0), b1 = b0, b2 = b0;
      bool o0 = false, o1 = false, o2 = false;
      if (T > 0) fetch(0, b0, o0);
      if (T > 1) fetch(1, b1, o1);
      for (u32 t = 0; t < T; t += 3)
        {
          if (t + 2 < T) fetch(t + 2, b2, o2); else o2 = false;
          if (o0) consume(b0);
          if (t + 1 >= T) break;
          if (t + 3 < T) fetch(t + 3, b0, o0); else o0 = false;
          if (o1) consume(b1);
          if (t + 2 >= T) break;
          if (t + 4 < T) fetch(t + 4, b1, o1); else o1 = false;
          if (o2) consume(b2);
        }
    }
  if ((probe & 1) && sink == 0x9e3779b9u) cnt[NDW] = sink;
  __syncthreads();
  if (probe & 4) { if (tid == 0) tile_count[region] = 0; return; }
  // the dummies have collected the hops and the padding (~20 increments each): cleared here, or nearly every dword that holds one
  // would look like a hit to the sweep's dword test (26 ms of 113 before this).  Dummy k = counter 251 + 252 k = the top byte of
  // dword 62 + 63 k (8-bit counters) / the top half of dword 125 + 126 k (16-bit)
  if (tid < 130) { if (BITS == 8) cnt[63 * tid + 62] &= 0x00ffffffu; else cnt[126 * tid + 125] &= 0x0000ffffu; }
  __syncthreads();
  // ---- sweep: counters >= mm -> (counter, count) records in the (query, tile) sub-region, as in vsx_kmer_count_kernel.  Here the
  // streaming part is short enough for the sweep to show (20 ms of 109 in its first form: ~0.5 % of the counters hit, so every wave
  // iteration had SOME lane in the per-field path).  Now: one flag bit per hit counter by SWAR (4 instructions per dword, no
  // branch), the hit count is a popcount, and the writing pass visits set bits only -- four loop heads per thread.
  // Dummies are cleared, counters of sequences past the last one were never touched.
  constexpr int DW_PER_THREAD = NDW / THREADS;                      // 16 in both configurations
  // flag = top bit of every field >= mm (1 <= mm; 8-bit class: a count never exceeds 255)
  constexpr u32 TOPS = (BITS == 8) ? 0x80808080u : 0x80008000u, LOWS = ~TOPS, ONES = (BITS == 8) ? 0x01010101u : 0x00010001u;
  constexpr u32 HALF = (BITS == 8) ? 128u : 32768u;
  const bool never = (BITS == 8) ? (mm > 255u) : (mm > 65535u);
  const bool low = mm <= HALF;
  const u32 addk = (low ? HALF - mm : 2u * HALF - mm) * ONES;
  auto flags = [&](u32 v) -> u32 {
    const u32 t = (v & LOWS) + addk;                                 // per field: top bit set iff its low bits >= mm (resp. mm - HALF)
    return low ? ((t | v) & TOPS) : (t & v & TOPS);
  };
  // the flags of a uint4's fields share one dword: dword j's flags (bits BITS * h + BITS - 1) move right by j
  u32 ms[DW_PER_THREAD / 4];
  u32 found = 0;
  const uint4 * c4 = reinterpret_cast<const uint4 *>(cnt);
#pragma unroll
  for (int g4 = 0; g4 < DW_PER_THREAD / 4; ++g4)
    {
      const uint4 v4 = c4[g4 * THREADS + tid];
      const u32 m = never ? 0u : (flags(v4.x) | (flags(v4.y) >> 1) | (flags(v4.z) >> 2) | (flags(v4.w) >> 3));
      ms[g4] = m;
      found += (u32) __builtin_popcount(m);
    }
  u32 inc2 = found;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1)
    {
      const u32 up = (u32) __shfl_up((int) inc2, d, 64);
      if (lane >= d) inc2 += up;
    }
  if (lane == 63) wave_hits[wave] = inc2;
  __syncthreads();
  u32 before = 0, total = 0;
#pragma unroll
  for (int w2 = 0; w2 < WAVES; ++w2) { const u32 t = wave_hits[w2]; before += (w2 < wave) ? t : 0u; total += t; }
  if (tid == 0) tile_count[region] = total;
  if (found)
    {
      // records carry tile << 15 | counter index; the selection kernel turns the ones it keeps into sequence numbers
      u32 pos = before + inc2 - found;
      uint2 * __restrict__ out = rec + region * subcap;
      const u32 tagx = tile << KM_TILE_SHIFT;
#pragma unroll
      for (int g4 = 0; g4 < DW_PER_THREAD / 4; ++g4)
        {
          u32 m = ms[g4];
          const u32 x0 = (u32) (g4 * THREADS + tid) * 4u * PER;      // first counter of the uint4
          while (m)
            {
              const u32 bit = (u32) __builtin_ctz(m);
              m &= m - 1u;
              const u32 j = (BITS - 1) - (bit & (BITS - 1)), h = bit / BITS;      // dword of the uint4, field of the dword
              const u32 x = x0 + j * PER + h;
              const u32 c = (BITS == 8) ? (u32) reinterpret_cast<const uint8_t *>(cnt)[x] : (u32) reinterpret_cast<const uint16_t *>(cnt)[x];
              if (pos < subcap) out[pos] = make_uint2(tagx | x, c);
              ++pos;
            }
        }
    }
}

// One wave per query slot: keep the records that can still reach the top `keep` (see the header comment).  The slot's records lie
// in ntiles sub-regions; ONE pass builds a histogram of their counts (clamped at 255), its suffix sums give the threshold -- the
// largest count with at least `keep` records at or above it -- and a second pass copies the records at or above the threshold to
// the dense buffer.  sel_off_n[slot] = (kept, seen); a slot with a sub-region that overflowed is marked (0xffffffff, the largest
// sub-region count) and left for the second pass.
__global__ void __launch_bounds__(256)
vsx_kmer_select_kernel(const uint2 * __restrict__ rec, u32 subcap, u32 ntiles, const u32 * __restrict__ tile_count, u32 nslots, u32 keep,
                       uint2 * __restrict__ dense, u64 * cursor, u64 capacity, uint2 * __restrict__ sel_off_n, u64 * __restrict__ sel_off, int packed)
{
  // packed != 0: the records of the packed index hold tile << 15 | counter index; counter x of a tile is sequence
  // tile * 32 630 + (x mod 252) * 130 + x / 252 (see the packed format above) -- converted here, for the kept records only
  __shared__ u32 hist_all[4][256];
  const int lane = (int) (threadIdx.x & 63), wv = (int) (threadIdx.x >> 6);
  const u32 slot = blockIdx.x * 4 + (u32) wv;
  if (slot >= nslots) return;                                      // (no barrier below: the four waves are independent)
  u32 * hist = hist_all[wv];
  const u32 * __restrict__ tc = tile_count + (size_t) slot * ntiles;
  u32 n = 0, worst = 0;
  for (u32 t = (u32) lane; t < ntiles; t += 64) { const u32 c = tc[t]; n += c; worst = c > worst ? c : worst; }
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1)
    {
      n += (u32) __shfl_xor((int) n, m, 64);
      const u32 o = (u32) __shfl_xor((int) worst, m, 64);
      worst = o > worst ? o : worst;
    }
  if (worst > subcap) { if (lane == 0) { sel_off[slot] = 0; sel_off_n[slot] = make_uint2(0xffffffffu, worst); } return; }
  const uint2 * __restrict__ r0 = rec + (size_t) slot * ntiles * subcap;
  u32 thr = 0;
  u32 m = n;                                                       // records at or above the threshold (thr == 0: all of them)
  if (n > keep)
    {
      for (int x = lane; x < 256; x += 64) hist[x] = 0;
      for (u32 t = 0; t < ntiles; ++t)
        {
          const u32 nt = tc[t];
          const uint2 * __restrict__ r = r0 + (size_t) t * subcap;
          for (u32 x = (u32) lane; x < nt; x += 64) { const u32 c = r[x].y; atomicAdd(&hist[c < 255u ? c : 255u], 1u); }
        }
      // suffix sums: lane l owns bins 4 l .. 4 l + 3
      u32 h4[4], own = 0;
#pragma unroll
      for (int u = 0; u < 4; ++u) { h4[u] = hist[4 * lane + u]; own += h4[u]; }
      u32 above = own;                                             // inclusive suffix over lanes l .. 63
#pragma unroll
      for (int d = 1; d < 64; d <<= 1)
        {
          const u32 dn = (u32) __shfl_down((int) above, d, 64);
          if (lane + d < 64) above += dn;
        }
      above -= own;                                                // records in the bins of the lanes above this one
      // the largest bin b with #(>= b) >= keep lies in exactly one lane; `at` = #(>= that bin)
      u32 cand = 0xffffffffu, run = above, at = 0;
#pragma unroll
      for (int u = 3; u >= 0; --u)
        {
          run += h4[u];
          if (cand == 0xffffffffu && run >= keep) { cand = (u32) (4 * lane + u); at = run; }
        }
      // the highest lane that found one wins
      const u64 found = __ballot(cand != 0xffffffffu);
      const int win = found ? 63 - __builtin_clzll(found) : 0;
      thr = (u32) __shfl((int) cand, win, 64);
      m = (u32) __shfl((int) at, win, 64);                         // (the histogram already knows how many records pass)
      if (thr == 255u)
        {
          // more than `keep` records sit in the clamped bin (16-bit class only): exact threshold by bisection over those
          u32 lo = 255, hi = 32768;                                // invariant: #(>= lo) >= keep, #(>= hi) < keep
          while (hi - lo > 1)
            {
              const u32 mid = (lo + hi) >> 1;
              u32 c = 0;
              for (u32 t = 0; t < ntiles; ++t)
                {
                  const u32 nt = tc[t];
                  const uint2 * __restrict__ r = r0 + (size_t) t * subcap;
                  for (u32 x = (u32) lane; x < nt; x += 64) c += (r[x].y >= mid) ? 1u : 0u;
                }
#pragma unroll
              for (int k = 32; k >= 1; k >>= 1) c += (u32) __shfl_xor((int) c, k, 64);
              if (c >= keep) { lo = mid; m = c; } else hi = mid;
            }
          thr = lo;
        }
    }
  u64 first = 0;
  if (lane == 0) first = atomicAdd(cursor, (u64) m);
  first = (u64) __shfl((long long) first, 0, 64);
  if (lane == 0) { sel_off[slot] = first; sel_off_n[slot] = make_uint2(m, n); }
  if (first + m > capacity) return;                      // the host re-runs the selection with the exact size
  u32 done = 0;
  for (u32 t = 0; t < ntiles; ++t)
    {
      const u32 nt = tc[t];
      const uint2 * __restrict__ r = r0 + (size_t) t * subcap;
      for (u32 x0 = 0; x0 < nt; x0 += 64)
        {
          const u32 x = x0 + (u32) lane;
          const bool take = (x < nt) && (r[x].y >= thr);
          const u64 ballot = __ballot(take);
          if (take)
            {
              uint2 v = r[x];
              if (packed)
                {
                  const u32 cx = v.x & (KM_TILE - 1u);
                  v.x = (v.x >> KM_TILE_SHIFT) * KM_PK_TILE_SEQS + km_pk_seq_of(cx);
                }
              dense[first + done + (u32) __popcll(ballot & ((1ull << lane) - 1ull))] = v;
            }
          done += (u32) __popcll(ballot);
        }
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

extern "C" hipError_t vsx_kmer_launch_ranges(const uint64_t * bucket_start, uint32_t ntiles, const uint64_t * qk_start, const uint32_t * qk,
                                             const uint32_t * minmatch, const uint32_t * qlist, uint32_t nslots, void * ranges, hipStream_t st)
{
  if (nslots == 0) return hipSuccess;
  hipLaunchKernelGGL(vsx_kmer_ranges_kernel, dim3(nslots), dim3(256), 0, st, (const u64 *) bucket_start, ntiles, (const u64 *) qk_start, qk,
                     minmatch, qlist, nslots, (uint2 *) ranges);
  return hipGetLastError();
}

extern "C" hipError_t vsx_kmer_launch_count(int bits, int tagged, const uint32_t * postings, const uint64_t * bucket_start, const void * ranges,
                                            uint32_t ntiles, uint32_t nseq, uint32_t nslots, uint32_t slot_base, const uint64_t * qk_start,
                                            const uint32_t * qk, const uint32_t * minmatch, const uint32_t * qlist, void * rec, uint32_t subcap,
                                            uint32_t * tile_count, hipStream_t st)
{
  // slots [slot_base, slot_base + nslots) of the batch; ranges / rec / tile_count belong to THESE slots (indexed from 0)
  // tagged: 0 = 16-bit postings, 1 = tagged postings (word lengths 9..15), 2 = packed postings
  if (nslots == 0 || nseq == 0) return hipSuccess;
  static const int probe = std::getenv("VSX_KMER_PROBE") ? std::atoi(std::getenv("VSX_KMER_PROBE")) : 0;
#define KM_ARGS (const uint4 *) postings, (const u64 *) bucket_start, (const uint2 *) ranges, ntiles, nseq, (const u64 *) qk_start, qk, minmatch, qlist, slot_base, \
                nslots, (uint2 *) rec, subcap, tile_count, probe
  if (tagged == 2 && bits == 8 && ranges) hipLaunchKernelGGL((vsx_kmer_count_packed_kernel<8, true>), dim3(nslots, ntiles), dim3(512), 0, st, KM_ARGS);
  else if (tagged == 2 && bits == 8) hipLaunchKernelGGL((vsx_kmer_count_packed_kernel<8, false>), dim3(nslots, ntiles), dim3(512), 0, st, KM_ARGS);
  else if (tagged == 2) hipLaunchKernelGGL((vsx_kmer_count_packed_kernel<16, false>), dim3(nslots, ntiles), dim3(1024), 0, st, KM_ARGS);
  else if (tagged && bits == 8) hipLaunchKernelGGL((vsx_kmer_count_kernel<8, false, true>), dim3(nslots, ntiles), dim3(512), 0, st, KM_ARGS);
  else if (tagged) hipLaunchKernelGGL((vsx_kmer_count_kernel<16, false, true>), dim3(nslots, ntiles), dim3(1024), 0, st, KM_ARGS);
  else if (bits == 8 && ranges && (probe & 8)) hipLaunchKernelGGL((vsx_kmer_count_packed_kernel<8, true, true>), dim3(nslots, ntiles), dim3(512), 0, st, KM_ARGS);
  else if (bits == 8 && ranges) hipLaunchKernelGGL((vsx_kmer_count_kernel<8, true>), dim3(nslots, ntiles), dim3(512), 0, st, KM_ARGS);
  else if (bits == 8) hipLaunchKernelGGL((vsx_kmer_count_kernel<8, false>), dim3(nslots, ntiles), dim3(512), 0, st, KM_ARGS);
  else hipLaunchKernelGGL((vsx_kmer_count_kernel<16, false>), dim3(nslots, ntiles), dim3(1024), 0, st, KM_ARGS);
#undef KM_ARGS
  return hipGetLastError();
}

// One tile of a packed index build (word lengths 3..8): keys of the tile's positions -> sorted -> one thread per word counts
// (fill == 0) or writes (fill != 0) its units.  temp == nullptr: only *temp_bytes (the sort's scratch for n_slots keys) is set.
extern "C" hipError_t vsx_kmer_packed_tile(int fill, const uint8_t * codes, const uint64_t * off, const uint32_t * len, uint32_t first_seq,
                                           uint32_t nseq_tile, int w, const uint8_t * lower_bits, const uint64_t * slot_of, uint64_t n_slots,
                                           uint32_t * keys_a, uint32_t * keys_b, void * temp, size_t * temp_bytes, uint32_t tile, uint32_t ntiles,
                                           uint32_t * bucket_count, const uint64_t * bucket_start, uint32_t * postings, hipStream_t st)
{
  if (!temp)
    return rocprim::radix_sort_keys(nullptr, *temp_bytes, keys_a, keys_b, (size_t) n_slots, 0, 32, st);
  if (nseq_tile == 0) return hipSuccess;
  const u32 nwords = 1u << (2 * w);
  if (n_slots)
    {
      hipLaunchKernelGGL(vsx_kmer_pk_keys_kernel, dim3((nseq_tile + 3) / 4), dim3(256), 0, st, codes, (const u64 *) off, len, first_seq, nseq_tile, w,
                         lower_bits, (const u64 *) slot_of, keys_a);
      hipError_t e = hipGetLastError();
      if (e != hipSuccess) return e;
      if ((e = rocprim::radix_sort_keys(temp, *temp_bytes, keys_a, keys_b, (size_t) n_slots, 0, 32, st)) != hipSuccess) return e;
    }
  const dim3 grid((nwords + 63) / 64);
  if (fill)
    hipLaunchKernelGGL(vsx_kmer_pk_walk_kernel<true>, grid, dim3(64), 0, st, (const u32 *) keys_b, (u64) n_slots, tile, ntiles, nwords, bucket_count,
                       (const u64 *) bucket_start, (uint4 *) postings);
  else
    hipLaunchKernelGGL(vsx_kmer_pk_walk_kernel<false>, grid, dim3(64), 0, st, (const u32 *) keys_b, (u64) n_slots, tile, ntiles, nwords, bucket_count,
                       (const u64 *) bucket_start, (uint4 *) postings);
  return hipGetLastError();
}

extern "C" uint32_t vsx_kmer_packed_tile_seqs(void) { return KM_PK_TILE_SEQS; }

// One tile of a tagged index build (word lengths 9..15): keys of the tile's positions -> sorted -> runs counted (fill == 0) or
// scattered (fill != 0).  temp == nullptr: only *temp_bytes (the sort's scratch for n_slots keys) is set.
extern "C" hipError_t vsx_kmer_tagged_tile(int fill, const uint8_t * codes, const uint64_t * off, const uint32_t * len, uint32_t first_seq,
                                           uint32_t nseq_tile, int w, const uint8_t * lower_bits, const uint64_t * slot_of, uint64_t n_slots,
                                           uint64_t * keys_a, uint64_t * keys_b, void * temp, size_t * temp_bytes, uint32_t tile, uint32_t ntiles,
                                           uint32_t * bucket_count, const uint64_t * bucket_start, uint32_t * postings, hipStream_t st)
{
  if (!temp)
    return rocprim::radix_sort_keys(nullptr, *temp_bytes, (u64k *) keys_a, (u64k *) keys_b, (size_t) n_slots, 0, 48, st);
  if (n_slots == 0 || nseq_tile == 0) return hipSuccess;
  hipLaunchKernelGGL(vsx_kmer_keys_kernel, dim3((nseq_tile + 3) / 4), dim3(256), 0, st, codes, (const u64 *) off, len, first_seq, nseq_tile, w,
                     lower_bits, (const u64 *) slot_of, (u64k *) keys_a);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  if ((e = rocprim::radix_sort_keys(temp, *temp_bytes, (u64k *) keys_a, (u64k *) keys_b, (size_t) n_slots, 0, 48, st)) != hipSuccess) return e;
  const dim3 grid((unsigned) ((n_slots + 255) / 256));
  if (fill)
    hipLaunchKernelGGL(vsx_kmer_runs_kernel<true>, grid, dim3(256), 0, st, (const u64k *) keys_b, (u64) n_slots, tile, ntiles, bucket_count,
                       (const u64 *) bucket_start, postings);
  else
    hipLaunchKernelGGL(vsx_kmer_runs_kernel<false>, grid, dim3(256), 0, st, (const u64k *) keys_b, (u64) n_slots, tile, ntiles, bucket_count,
                       (const u64 *) bucket_start, postings);
  return hipGetLastError();
}

extern "C" hipError_t vsx_kmer_launch_select(const void * rec, uint32_t subcap, uint32_t ntiles, const uint32_t * tile_count, uint32_t nslots,
                                             uint32_t keep, void * dense, unsigned long long * cursor, uint64_t capacity,
                                             void * sel_m_n, uint64_t * sel_off, hipStream_t st)
{
  if (nslots == 0) return hipSuccess;
  hipLaunchKernelGGL(vsx_kmer_select_kernel, dim3((nslots + 3) / 4), dim3(256), 0, st, (const uint2 *) rec, subcap, ntiles, tile_count, nslots,
                     keep, (uint2 *) dense, (u64 *) cursor, (u64) capacity, (uint2 *) sel_m_n, (u64 *) sel_off, 0);
  return hipGetLastError();
}

extern "C" hipError_t vsx_kmer_launch_select_packed(const void * rec, uint32_t subcap, uint32_t ntiles, const uint32_t * tile_count, uint32_t nslots,
                                                    uint32_t keep, void * dense, unsigned long long * cursor, uint64_t capacity,
                                                    void * sel_m_n, uint64_t * sel_off, hipStream_t st)
{
  if (nslots == 0) return hipSuccess;
  hipLaunchKernelGGL(vsx_kmer_select_kernel, dim3((nslots + 3) / 4), dim3(256), 0, st, (const uint2 *) rec, subcap, ntiles, tile_count, nslots,
                     keep, (uint2 *) dense, (u64 *) cursor, (u64) capacity, (uint2 *) sel_m_n, (u64 *) sel_off, 1);
  return hipGetLastError();
}

extern "C" uint32_t vsx_kmer_tile_shift(void) { return KM_TILE_SHIFT; }
