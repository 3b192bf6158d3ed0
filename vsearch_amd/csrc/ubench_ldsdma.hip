// ubench_ldsdma.hip -- where do the bytes of a global_load_lds land?  (r05: the traceback's row checkpoints straight into LDS.)
// Every lane loads `size` bytes (4, 12 or 16) from its own global address -- dword k of lane l holds the value 0x10000 * k + l -- with
// __builtin_amdgcn_global_load_lds to a wave-uniform LDS base; the wave then dumps the LDS region, and the host prints for the first
// lanes at which LDS dword each (lane, k) was found.      hipcc --offload-arch=gfx950 -O3 -o ubench_ldsdma ubench_ldsdma.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned int u32;

template <int SIZE>
__global__ void __launch_bounds__(64) probe(const u32 * __restrict__ src, u32 * __restrict__ dump, int masked)
{
  __shared__ __attribute__((aligned(16))) u32 L[64 * 4 + 64];
  const int tid = (int) threadIdx.x;
  for (int x = tid; x < 64 * 4 + 64; x += 64) L[x] = 0xDEADBEEFu;
  __syncthreads();
  const u32 * g = src + tid * 4;                       // 16 bytes apart in global memory
  if (!masked || (tid & 1))
    {
      const __attribute__((address_space(1))) void * gp = (const __attribute__((address_space(1))) void *) g;
      __attribute__((address_space(3))) void * lp = (__attribute__((address_space(3))) void *) (L + 16);
      if constexpr (SIZE == 4) __builtin_amdgcn_global_load_lds(gp, lp, 4, 0, 0);
      else if constexpr (SIZE == 12) __builtin_amdgcn_global_load_lds(gp, lp, 12, 0, 0);
      else __builtin_amdgcn_global_load_lds(gp, lp, 16, 0, 0);
    }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int x = tid; x < 64 * 4 + 64; x += 64) dump[x] = L[x];
}

template <int SIZE> static void run(const u32 * d_src, u32 * d_dump, int masked)
{
  hipLaunchKernelGGL(probe<SIZE>, dim3(1), dim3(64), 0, 0, d_src, d_dump, masked);
  std::vector<u32> h(64 * 4 + 64);
  hipMemcpy(h.data(), d_dump, h.size() * 4, hipMemcpyDeviceToHost);
  std::printf("size %2d%s: ", SIZE, masked ? " (odd lanes only)" : "");
  for (int l = 0; l < 4; ++l)
    for (int k = 0; k < SIZE / 4; ++k)
      {
        const u32 want = 0x10000u * (u32) k + (u32) l;
        int at = -1;
        for (size_t x = 0; x < h.size(); ++x) if (h[x] == want) { at = (int) x - 16; break; }
        std::printf("(lane %d dw %d)->%d ", l, k, at);
      }
  int lane63 = -1;
  for (size_t x = 0; x < h.size(); ++x) if (h[x] == 63u) { lane63 = (int) x - 16; break; }
  int untouched = 0;
  for (size_t x = 0; x < h.size(); ++x) if (h[x] == 0xDEADBEEFu) ++untouched;
  std::printf("| lane 63 dw 0 -> %d | dwords untouched %d of %zu\n", lane63, untouched, h.size());
}

int main()
{
  std::vector<u32> src(64 * 4);
  for (int l = 0; l < 64; ++l) for (int k = 0; k < 4; ++k) src[l * 4 + k] = 0x10000u * (u32) k + (u32) l;
  u32 * d_src, * d_dump;
  hipMalloc(&d_src, src.size() * 4); hipMalloc(&d_dump, (64 * 4 + 64) * 4);
  hipMemcpy(d_src, src.data(), src.size() * 4, hipMemcpyHostToDevice);
  for (int m = 0; m < 2; ++m) { run<4>(d_src, d_dump, m); run<12>(d_src, d_dump, m); run<16>(d_src, d_dump, m); }
  return 0;
}
