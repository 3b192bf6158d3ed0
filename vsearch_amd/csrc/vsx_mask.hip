// vsx_mask.hip -- DUST low-complexity masking of a whole sequence set on gfx950: the reference masks the database once before
// it is indexed (dust_all, core/mask.cpp:233-249 = 54 % of a small --usearch_global run) and this is the
// same computation over the set's 4-bit codes in HBM, its result OR-ed into the set's case bitmap (the k-mer index skips
// every word over a set bit, vsx_kmer.hip).  Host restatement with the algorithm spelled out: vsx_mask.cpp.
//
//   one wave per sequence; the windows of a sequence are visited in order (their positions depend on the previous window's
//   result, mask.cpp:146-170), inside a window lane i owns start offset i (at most 57 of them): it walks the end offsets j
//   with its own 64 one-byte 3-mer counters in LDS (64 x 64 B per wave, lane-minor so a wave's accesses spread over the
//   banks), keeps its first maximum of 10 * pairs / j, and a wave reduction picks the maximum with the smallest i -- the
//   interval the reference's nested loops find first.  The division is exact: __umulhi(x, ceil(2^32 / j)) == x / j for
//   x < 2^32 / 63 (x <= 18 910 here).
//   Integer / LDS work, ~3 k lane-iterations per 32 symbols; no MFMA shape in it.
#include <hip/hip_runtime.h>
#include "vsx_internal.h"

typedef unsigned int u32;
typedef unsigned long long u64;

#define DW 64
#define DH 32
#define DLEVEL 20

__global__ void __launch_bounds__(256)
vsx_dust_kernel(const uint8_t * __restrict__ codes, const u64 * __restrict__ off, const u32 * __restrict__ len, u32 nseq,
                u32 * __restrict__ bits /* the case bitmap, as dwords */)
{
  __shared__ uint8_t cnt_all[4][64 * 64];               // [wave][word * 64 + lane]
  __shared__ uint8_t tri_all[4][DW];
  __shared__ u32 magic[64];                              // ceil(2^32 / j)
  const int lane = (int) (threadIdx.x & 63), wave = (int) (threadIdx.x >> 6);
  if (threadIdx.x < 64)
    {
      const u32 j = threadIdx.x < 2 ? 2u : threadIdx.x;
      magic[threadIdx.x] = (u32) (0x100000000ull / j) + ((j & (j - 1u)) ? 1u : 0u);
    }
  __syncthreads();
  const u32 seq = blockIdx.x * 4 + wave;
  if (seq >= nseq) return;
  uint8_t * cnt = cnt_all[wave];
  uint8_t * tri = tri_all[wave];
  const u64 base = off[seq];
  const uint8_t * __restrict__ s = codes + base;
  const long L = (long) len[seq];
  for (long i0 = 0; i0 < L; i0 += DH)
    {
      const int n = (int) (L > i0 + DW ? DW : L - i0);
      const int starts = n - 7;
      if (starts <= 0) continue;
      // 3-mer ending at window position `lane` (2-bit map of the one-hot codes, anything else 0: map_2bit)
      {
        u32 t = 0;
#pragma unroll
        for (int x = 2; x >= 0; --x)
          {
            const long p = i0 + lane - x;
            u32 c = (lane - x >= 0 && lane < n) ? (u32) s[p] : 0u;
            c = (c == 1u || c == 2u || c == 4u || c == 8u) ? ((c >> 1) - (c >> 3)) : 0u;
            t = (t << 2) | c;
          }
        tri[lane] = (uint8_t) t;
      }
      // clear this lane's counters: 64 bytes at stride 64
#pragma unroll 8
      for (int w = 0; w < 64; ++w) cnt[w * 64 + lane] = 0;
      // tri[] is written by the other lanes of this wave: order the LDS accesses around a wave barrier
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      int best = 0, bj = 0;
      if (lane < starts)
        {
          int pairs = 0;
          const int jn = n - lane;
          for (int j = 2; j < jn; ++j)
            {
              const u32 w = tri[lane + j];
              uint8_t * c = cnt + w * 64 + lane;
              const u32 cv = *c;
              if (cv)
                {
                  pairs += (int) cv;
                  const int v = (int) __umulhi((u32) (10 * pairs), magic[j]);
                  if (v > best) { best = v; bj = j; }
                }
              *c = (uint8_t) (cv + 1u);
            }
        }
      // maximum score, smallest start offset
      u32 key = ((u32) best << 6) | (u32) (63 - lane);
#pragma unroll
      for (int d = 32; d >= 1; d >>= 1)
        {
          const u32 o = (u32) __shfl_xor((int) key, d, 64);
          key = o > key ? o : key;
        }
      const int wbest = (int) (key >> 6), wi = 63 - (int) (key & 63u);
      const int wj = __shfl(bj, wi, 64);
      if (wbest > DLEVEL)
        {
          const int a = wi, b = wi + wj;
          const long p = i0 + a + lane;
          if (a + lane <= b)
            {
              const u64 bit = base + (u64) p;
              atomicOr(&bits[bit >> 5], 1u << (u32) (bit & 31u));
            }
          if (b < DH) i0 += DH - b;
        }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");          // the next window rewrites tri[] and the counters
      __builtin_amdgcn_wave_barrier();
    }
}

extern "C" hipError_t vsx_launch_dust(const uint8_t * d_codes, const uint64_t * d_off, const uint32_t * d_len, uint64_t nseq,
                                      uint8_t * d_bits, hipStream_t st)
{
  if (nseq == 0) return hipSuccess;
  hipLaunchKernelGGL(vsx_dust_kernel, dim3((unsigned) ((nseq + 3) / 4)), dim3(256), 0, st, d_codes, (const u64 *) d_off, d_len, (u32) nseq,
                     reinterpret_cast<u32 *>(d_bits));
  return hipGetLastError();
}
