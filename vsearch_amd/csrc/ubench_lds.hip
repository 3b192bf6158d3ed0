// ubench_lds.hip -- LDS atomic throughput on gfx950 under the address patterns the k-mer count kernel can produce.
//   hipcc --offload-arch=gfx950 -O3 -o ubench_lds ubench_lds.hip && ./ubench_lds
// Each block: 1024 threads, 64 KB of counters; every thread performs N ds_add_u32 (no return) at addresses from a
// pre-generated per-lane table in registers (so no global loads sit in the timed loop).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef unsigned int u32;

template <int MODE>
__global__ void __launch_bounds__(1024) k_lds(const u32 * __restrict__ addr, int iters, u32 * out)
{
  __shared__ u32 cnt[16384];
  for (int x = threadIdx.x; x < 16384; x += 1024) cnt[x] = 0;
  __syncthreads();
  u32 a[16];
#pragma unroll
  for (int u = 0; u < 16; ++u) a[u] = addr[(blockIdx.x * 16 + u) * 1024 + threadIdx.x];
  for (int it = 0; it < iters; ++it)
    {
#pragma unroll
      for (int u = 0; u < 16; ++u)
        {
          const u32 x = (a[u] + (u32) it * 33u) & 16383u;           // keeps the pattern class (mod 32 bank preserved for +33*it? no: shifts all lanes alike)
          if (MODE == 0) atomicAdd(&cnt[x], 1u);                     // ds_add_u32
          else if (MODE == 1) { u32 r = atomicAdd(&cnt[x], 1u); a[u] ^= (r >> 31); }   // ds_add_rtn_u32
          else if (MODE == 2) cnt[x] += 1u;                          // ds_read + ds_write (racy; throughput only)
          else if (MODE == 3) atomicAdd(reinterpret_cast<unsigned long long *>(&cnt[x & ~1u]), 1ull);   // ds_add_u64
        }
    }
  __syncthreads();
  u32 s = 0;
  for (int x = threadIdx.x; x < 16384; x += 1024) s += cnt[x];
  if (s == 0xdeadbeefu) out[0] = s + a[0];
}

int main()
{
  const int blocks = 512, iters = 200;
  std::vector<u32> h((size_t) blocks * 16 * 1024);
  u32 * d_addr; u32 * d_out;
  hipMalloc(&d_addr, h.size() * 4); hipMalloc(&d_out, 4);
  const char * names[] = {"random dword in 64 KB", "conflict-free (lane l -> bank l % 32, distinct dwords)", "2 lanes per dword (pairs share an address)",
                          "all lanes of a wave in ONE bank (32-way)", "random within a 4 KB window per wave", "sorted ascending per wave, stride ~33 dwords"};
  for (int pat = 0; pat < 6; ++pat)
    {
      srand(7);
      for (size_t i = 0; i < h.size(); ++i)
        {
          const u32 lane = (u32) (i & 63), wave = (u32) ((i >> 6) & 15);
          u32 x = 0;
          switch (pat)
            {
            case 0: x = (u32) rand() & 16383u; break;
            case 1: x = ((((u32) rand() & 255u) * 64u) + lane) & 16383u; break;      // dword index = k * 64 + lane: bank = lane % 32
            case 2: x = ((((u32) rand() & 255u) * 64u) + (lane >> 1) * 2u) & 16383u; break;
            case 3: x = (((u32) rand() & 511u) * 32u) & 16383u; break;
            case 4: x = (wave * 1024u + ((u32) rand() & 1023u)) & 16383u; break;
            case 5: x = (((u32) rand() & 7u) * 2048u + lane * 33u + ((u32) rand() & 15u)) & 16383u; break;
            }
          h[i] = x;
        }
      hipMemcpy(d_addr, h.data(), h.size() * 4, hipMemcpyHostToDevice);
      for (int mode = 0; mode < 4; ++mode)
        {
          if (pat > 1 && mode > 0 && pat != 4) continue;
          hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
          auto launch = [&]() {
            if (mode == 0) hipLaunchKernelGGL(k_lds<0>, dim3(blocks), dim3(1024), 0, 0, d_addr, iters, d_out);
            if (mode == 1) hipLaunchKernelGGL(k_lds<1>, dim3(blocks), dim3(1024), 0, 0, d_addr, iters, d_out);
            if (mode == 2) hipLaunchKernelGGL(k_lds<2>, dim3(blocks), dim3(1024), 0, 0, d_addr, iters, d_out);
            if (mode == 3) hipLaunchKernelGGL(k_lds<3>, dim3(blocks), dim3(1024), 0, 0, d_addr, iters, d_out);
          };
          launch(); hipDeviceSynchronize();
          hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
          float ms = 0; hipEventElapsedTime(&ms, e0, e1);
          const double ops = (double) blocks * 1024 * 16 * iters;
          const char * mn[] = {"ds_add_u32", "ds_add_rtn_u32", "read+add+write", "ds_add_u64"};
          printf("%-58s %-15s %8.3f ms  %7.2f G ops/s  %5.2f ops/clk/CU (256 CU, 2.4 GHz)\n", names[pat], mn[mode], ms, ops / ms / 1e6,
                 ops / (ms * 1e-3) / (256.0 * 2.4e9));
        }
    }
  return 0;
}
