// vsx_tbtext.hip -- result marshalling on the device: run lists -> CIGAR text, records -> the five output arrays.
//
// The reference builds the CIGAR string inside its traceback (pushop / finishop, src/core/align_simd.cpp:1013-1049:
// run-length text written right to left, the count omitted when it is 1) and hands every caller five arrays
// (pscores / paligned / pmatches / pmismatches / pgaps, align_simd.hpp:99-108).  Here the traceback kernel leaves
// 24-byte records and a dense buffer of run words ((length << 2) | op, traceback order = last column first);
// this kernel turns them into exactly those outputs in HBM, so the host only copies (r01 formatted the text with
// host threads: 0.25 s per 800 k pairs, 8x the kernels).  HBM-bound byte work: ~80 B of runs read and ~60 B of text
// written per pair, one lane per pair, strings dword-aligned so every store is a full dword.
#include <hip/hip_runtime.h>
#include "vsx_internal.h"

typedef unsigned int u32;

#define DEV __device__ __forceinline__

DEV u32 ndigits(u32 v) { return v >= 10000u ? 5u : v >= 1000u ? 4u : v >= 100u ? 3u : v >= 10u ? 2u : 1u; }

// One lane per GPU pair (k-th entry of pair_ids).  Text bytes of a pair: for each run in TEXT order (= reverse run order)
// the decimal length if > 1, then 'M' / 'I' / 'D'; a terminating NUL; padded to a multiple of 4.  A wave allocates the
// strings of its 64 pairs with ONE atomic on the text cursor (they end up back to back in lane order).
__global__ void __launch_bounds__(256)
vsx_cigar_text_kernel(const VsxPairOut * __restrict__ out, const u32 * __restrict__ pair_ids, u32 npairs,
                      const u32 * __restrict__ runs, uint64_t runs_capacity,
                      uint8_t * __restrict__ text, uint64_t text_capacity, unsigned long long * text_cursor,
                      VsxSoaOut soa)
{
  const u32 k = blockIdx.x * blockDim.x + threadIdx.x;
  const bool valid = k < npairs;
  const u32 pid = valid ? pair_ids[k] : 0u;
  VsxPairOut o {};
  if (valid) o = out[pid];
  const bool have_runs = valid && (o.run_off + o.nruns <= runs_capacity);    // an overflowed run buffer is re-run by the host
  const u32 * __restrict__ my = runs + o.run_off;
  const u32 nruns = have_runs ? o.nruns : 0u;

  u32 len = 1;                                           // NUL
  for (u32 x = 0; x < nruns; ++x)
    {
      const u32 n = my[x] >> 2;
      len += 1u + (n > 1u ? ndigits(n) : 0u);
    }
  const u32 len4 = valid ? ((len + 3u) & ~3u) : 0u;

  // wave-level exclusive prefix of len4, one atomic per wave
  u32 incl = len4;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1)
    {
      const u32 v = (u32) __shfl_up((int) incl, d, 64);
      if ((int) (threadIdx.x & 63) >= d) incl += v;
    }
  const u32 total = (u32) __shfl((int) incl, 63, 64);
  unsigned long long base = 0;
  if ((threadIdx.x & 63) == 0 && total) base = atomicAdd(text_cursor, (unsigned long long) total);
  base = (unsigned long long) __shfl((long long) base, 0, 64);
  if (!valid) return;
  const unsigned long long off = base + (unsigned long long) (incl - len4);

  if (off + len4 <= text_capacity)
    {
      u32 * __restrict__ dst = reinterpret_cast<u32 *>(text + off);
      u32 acc = 0, have = 0;                             // bytes waiting to be stored, lowest byte first
      auto put = [&](u32 ch) {
        acc |= ch << (8u * have);
        if (++have == 4u) { *dst++ = acc; acc = 0; have = 0; }
      };
      for (u32 x = nruns; x-- > 0;)
        {
          const u32 w = my[x];
          const u32 n = w >> 2;
          if (n > 1u)
            {
              if (n >= 10000u) put('0' + n / 10000u);
              if (n >= 1000u) put('0' + (n / 1000u) % 10u);
              if (n >= 100u) put('0' + (n / 100u) % 10u);
              if (n >= 10u) put('0' + (n / 10u) % 10u);
              put('0' + n % 10u);
            }
          put((w & 3u) == 0u ? 'M' : (w & 3u) == 1u ? 'I' : 'D');
        }
      put(0u);
      if (have) *dst = acc;                        // padding bytes are zero
    }

  soa.score[pid] = o.score;
  soa.aligned[pid] = o.aligned;
  soa.matches[pid] = o.matches;
  soa.mismatches[pid] = o.mismatches;
  soa.gaps[pid] = o.gaps;
  soa.verdict[pid] = (uint8_t) o.pad;
  soa.text_off[pid] = off;
}

// ---------------------------------------------------------------------------------------------
// Reverse complement in the 4-bit code domain (utils/reverse_complement.cpp:70-82 + chrmap_complement, utils/maps.cpp:
// 121-150): the complement of an IUPAC set code is its bit reversal (A1 <-> T8, C2 <-> G4, R5 <-> Y10, ...); a symbol
// that is no IUPAC letter (code 0) complements to 'N' = 15, as the reference's table does.  One workgroup per sequence:
// sequence k of `src` is read backwards and written forwards as sequence k of `dst`.  HBM-bound, 1 B in / 1 B out.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
vsx_revcomp_kernel(const uint8_t * __restrict__ codes, const uint64_t * __restrict__ src_off, const uint64_t * __restrict__ dst_off,
                   const uint32_t * __restrict__ len)
{
  const uint64_t k = blockIdx.x;
  const uint8_t * __restrict__ s = codes + src_off[k];
  uint8_t * __restrict__ d = const_cast<uint8_t *>(codes) + dst_off[k];
  const u32 L = len[k];
  for (u32 x = threadIdx.x; x < L; x += 256)
    {
      const u32 c = s[L - 1 - x] & 15u;
      const u32 r = ((c & 1u) << 3) | ((c & 2u) << 1) | ((c & 4u) >> 1) | ((c & 8u) >> 3);
      d[x] = (uint8_t) (c == 0u ? 15u : r);
    }
}

extern "C" hipError_t vsx_launch_revcomp(uint8_t * d_codes, const uint64_t * d_src_off, const uint64_t * d_dst_off,
                                         const uint32_t * d_len, uint64_t nseq, hipStream_t st)
{
  if (nseq == 0) return hipSuccess;
  hipLaunchKernelGGL(vsx_revcomp_kernel, dim3((unsigned) nseq), dim3(256), 0, st, d_codes, d_src_off, d_dst_off, d_len);
  return hipGetLastError();
}

extern "C" hipError_t vsx_launch_cigar_text(const VsxPairOut * d_out, const uint32_t * d_pair_ids, uint32_t npairs,
                                            const uint32_t * d_runs, uint64_t runs_capacity,
                                            uint8_t * d_text, uint64_t text_capacity, unsigned long long * d_text_cursor,
                                            VsxSoaOut soa, hipStream_t st)
{
  if (npairs == 0) return hipSuccess;
  hipLaunchKernelGGL(vsx_cigar_text_kernel, dim3((npairs + 255) / 256), dim3(256), 0, st,
                     d_out, d_pair_ids, npairs, d_runs, runs_capacity, d_text, text_capacity, d_text_cursor, soa);
  return hipGetLastError();
}
