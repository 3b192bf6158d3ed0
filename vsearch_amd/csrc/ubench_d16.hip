// Micro-benchmark for the ADD3 row body of the DP kernel (r03): can the per-row v_perm_b32 go?
//   (1) semantics: on gfx950 with SRAM ECC, does ds_read_u16_d16_hi ZERO the low half of its destination (LLVM will not select d16
//       loads on sramecc targets because the unused half is not preserved)?  If so, score(A) = ds_read_u16 (zero-extended) and
//       score(B) << 16 = ds_read_u16_d16_hi come out of the int16 profile without any VALU work and H + S is ONE v_add3_u32.
//   (2) throughput: 16 rows per step, profile rows picked by a per-lane symbol pair, 64-thread blocks with the 8 KB profile:
//       mode 0 = today's body (ds_read_b64 per 4 rows and target, v_perm per row, add, max3, sub, 2 max),
//       mode 1 = two 16-bit loads per row, add3, max3, sub, 2 max.
// Build: hipcc --offload-arch=gfx950 -O3 ubench_d16.hip -o ubench_d16
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e)); exit(1); } } while (0)

__global__ void k_semantics(unsigned * out)
{
  __shared__ unsigned short tab[256];
  tab[threadIdx.x] = (unsigned short) (0x1100u + threadIdx.x);
  tab[threadIdx.x + 64] = (unsigned short) (0x2200u + threadIdx.x);
  __syncthreads();
  const unsigned addr = (unsigned) (size_t) tab + threadIdx.x * 2u;   // LDS byte address
  unsigned hi = 0xDEADBEEFu, lo = 0xDEADBEEFu, u16 = 0xDEADBEEFu;
  asm volatile("ds_read_u16_d16_hi %0, %3 offset:128\n"
               "ds_read_u16_d16 %1, %3\n"
               "ds_read_u16 %2, %3\n"
               "s_waitcnt lgkmcnt(0)\n" : "+v"(hi), "+v"(lo), "+v"(u16) : "v"(addr));
  out[threadIdx.x * 3 + 0] = hi;
  out[threadIdx.x * 3 + 1] = lo;
  out[threadIdx.x * 3 + 2] = u16;
}

typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef unsigned short us2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned max3(unsigned a, unsigned b, unsigned c)
{
  const f16x2 x = __builtin_bit_cast(f16x2, a), y = __builtin_bit_cast(f16x2, b), z = __builtin_bit_cast(f16x2, c);
  return __builtin_bit_cast(unsigned, __builtin_elementwise_maximum(__builtin_elementwise_maximum(x, y), z));
}
__device__ __forceinline__ unsigned pmaxu(unsigned a, unsigned b)
{
  return __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(us2, a), __builtin_bit_cast(us2, b)));
}

#define R 16
//       mode 2 = PAIR profile: one dword per (symbol pair of the two targets, position, row) = both halves ready; 16 pairs (pure
//       A/C/G/T targets) x 16 x 16 x 4 B = 16 KB per wave, i.e. at most 2 waves per SIMD; ds_read_b128 per 4 rows, add, max3, sub, 2 max.
template <int MODE>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(MODE == 2 ? 2 : 4, 8))) k_rows(int steps, unsigned * out, unsigned seed)
{
  __shared__ __attribute__((aligned(16))) short QP[(MODE == 2 ? 2 : 1) * 16 * 16 * R];            // [code][position][row]
  for (int i = threadIdx.x; i < (MODE == 2 ? 2 : 1) * 16 * 16 * R; i += 64) QP[i] = (short) ((i * 7 + (int) seed) & 0xff);
  __syncthreads();
  const int l = threadIdx.x & 15;
  unsigned H[R], H2[R], E[R];
  for (int r = 0; r < R; ++r) { H[r] = 0x3E003E00u + (unsigned) (r * 3 + (int) threadIdx.x); H2[r] = H[r]; E[r] = 0x3E003E00u - (unsigned) (r + (int) seed); }
  unsigned F = 0x3E003E00u, diag = 0x3E003E00u, sym = threadIdx.x * 2654435761u + seed;
  const unsigned go = 0x00120012u;
  auto step = [&](unsigned (&hin)[R], unsigned (&hout)[R]) __attribute__((always_inline)) {
    sym = sym * 1664525u + 1013904223u;
    const unsigned ca = (sym >> 8) & 15u, cb = (sym >> 20) & 15u;
    unsigned Hd = diag;
    if (MODE == 0)
      {
        const short * qa = QP + ca * (16 * R) + l * R;
        const short * qb = QP + cb * (16 * R) + l * R;
        unsigned pa = 0, pb = 0, pa2 = 0, pb2 = 0;
#pragma unroll
        for (int r = 0; r < R; ++r)
          {
            if ((r & 3) == 0)
              {
                const uint2 va = *reinterpret_cast<const uint2 *>(qa + r), vb = *reinterpret_cast<const uint2 *>(qb + r);
                pa = va.x; pb = vb.x; pa2 = va.y; pb2 = vb.y;
              }
            else if ((r & 3) == 2) { pa = pa2; pb = pb2; }
            const unsigned V = (r & 1) ? __builtin_amdgcn_perm(pb, pa, 0x07060302u) : __builtin_amdgcn_perm(pb, pa, 0x05040100u);
            const unsigned h0 = Hd + V;
            const unsigned h2 = max3(h0, F, E[r]);
            Hd = hin[r];
            hout[r] = h2;
            const unsigned he = h2 - go;
            F = pmaxu(F, he);
            E[r] = pmaxu(E[r], he);
          }
      }
    else if (MODE == 2)
      {
        const uint4 * q4 = reinterpret_cast<const uint4 *>(QP) + (((ca & 3u) * 4u + (cb & 3u)) * 16u + (unsigned) l) * (R / 4);
        uint4 v4 = make_uint4(0, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < R; ++r)
          {
            if ((r & 3) == 0) v4 = q4[r >> 2];
            const unsigned V = (r & 3) == 0 ? v4.x : ((r & 3) == 1 ? v4.y : ((r & 3) == 2 ? v4.z : v4.w));
            const unsigned h0 = Hd + V;
            const unsigned h2 = max3(h0, F, E[r]);
            Hd = hin[r];
            hout[r] = h2;
            const unsigned he = h2 - go;
            F = pmaxu(F, he);
            E[r] = pmaxu(E[r], he);
          }
      }
    else
      {
        const unsigned aA = (unsigned) (size_t) QP + (ca * (16 * R) + (unsigned) l * R) * 2u;
        const unsigned aB = (unsigned) (size_t) QP + (cb * (16 * R) + (unsigned) l * R) * 2u;
        unsigned lo[R], hi[R];
        // chunks of two rows: two chunks in flight
#define LOAD2(k) asm volatile("ds_read_u16 %0, %4 offset:%6\n ds_read_u16_d16_hi %1, %5 offset:%6\n ds_read_u16 %2, %4 offset:%7\n ds_read_u16_d16_hi %3, %5 offset:%7\n" \
                              : "=&v"(lo[2 * (k)]), "=&v"(hi[2 * (k)]), "=&v"(lo[2 * (k) + 1]), "=&v"(hi[2 * (k) + 1]) : "v"(aA), "v"(aB), "n"(4 * (k)), "n"(4 * (k) + 2))
#define WAIT2(k, n) asm volatile("s_waitcnt lgkmcnt(" #n ")" : "+v"(lo[2 * (k)]), "+v"(hi[2 * (k)]), "+v"(lo[2 * (k) + 1]), "+v"(hi[2 * (k) + 1]))
        LOAD2(0); LOAD2(1);
#pragma unroll
        for (int k = 0; k < R / 2; ++k)
          {
            if (k + 1 < R / 2) { WAIT2(k, 4); } else { WAIT2(k, 0); }
#pragma unroll
            for (int rr = 0; rr < 2; ++rr)
              {
                const int r = 2 * k + rr;
                const unsigned h0 = Hd + lo[r] + hi[r];
                const unsigned h2 = max3(h0, F, E[r]);
                Hd = hin[r];
                hout[r] = h2;
                const unsigned he = h2 - go;
                F = pmaxu(F, he);
                E[r] = pmaxu(E[r], he);
              }
            if (k + 2 < R / 2)
              {
                switch (k + 2) { case 2: LOAD2(2); break; case 3: LOAD2(3); break; case 4: LOAD2(4); break; case 5: LOAD2(5); break; case 6: LOAD2(6); break; case 7: LOAD2(7); break; }
              }
          }
      }
    diag = hout[R - 1] ^ sym;
  };
  for (int t = 0; t < steps; t += 2) { step(H, H2); step(H2, H); }
  unsigned acc = F;
  for (int r = 0; r < R; ++r) acc ^= H[r] ^ E[r];
  out[blockIdx.x * 64 + threadIdx.x] = acc;
}

template <int MODE> static void run(int waves_per_simd)
{
  const int blocks = 1024 * waves_per_simd, steps = 2000;
  unsigned * out; CK(hipMalloc(&out, (size_t) blocks * 64 * 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL(k_rows<MODE>, dim3(blocks), dim3(64), 0, 0, 20, out, 1u);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL(k_rows<MODE>, dim3(blocks), dim3(64), 0, 0, steps, out, 1u);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  unsigned h0 = 0;
  CK(hipMemcpy(&h0, out, 4, hipMemcpyDeviceToHost));
  const double cycles = ms * 1e-3 * 2.4e9;
  printf("mode %d (%s) waves/SIMD %d: %8.3f ms, %.1f cycles per lane-row and wave   [check %08x]\n", MODE,
         MODE == 0 ? "b64 loads + perm, 6 VALU per row" : (MODE == 1 ? "u16 loads, add3, 5 VALU per row" : "pair profile, b128 loads, 5 VALU per row"), waves_per_simd, ms,
         cycles / ((double) steps * R * waves_per_simd), h0);
  CK(hipFree(out));
}

int main()
{
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  printf("device %s  CUs %d  arch %s\n", p.name, p.multiProcessorCount, p.gcnArchName);
  unsigned * out; CK(hipMalloc(&out, 64 * 3 * 4));
  hipLaunchKernelGGL(k_semantics, dim3(1), dim3(64), 0, 0, out);
  unsigned h[12]; CK(hipMemcpy(h, out, sizeof h, hipMemcpyDeviceToHost));
  printf("lane 0: d16_hi into 0xDEADBEEF -> %08x, d16 -> %08x, u16 -> %08x;  lane 1: %08x %08x %08x\n", h[0], h[1], h[2], h[3], h[4], h[5]);
  printf("  (table: lane x holds 0x1100 + x at its address, 0x2200 + x at +128)  d16_hi zeroes the low half: %s\n", (h[0] & 0xffffu) == 0 ? "YES" : "no");
  for (int w = 1; w <= 4; w *= 2) { run<0>(w); run<1>(w); if (w <= 2) run<2>(w); }
  run<0>(3);
  return 0;
}
