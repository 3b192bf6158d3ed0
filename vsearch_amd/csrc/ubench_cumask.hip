// ubench_cumask.hip -- what hipExtStreamCreateWithCUMask does on this part (MI355X: 8 XCDs x 32 CUs): a kernel of many one-wave
// workgroups records where each one ran (XCC_ID and HW_ID: shader engine / CU); the host prints the set of (xcc, se, cu) that a
// mask admits and the time a fixed amount of ALU work takes on it.  Decides how vsx_align_pairs may give its traceback stream a
// fixed share of the device.
// build: hipcc --offload-arch=gfx950 -O3 -o ubench_cumask ubench_cumask.hip ;  run: ./ubench_cumask
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <map>
#include <set>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void __launch_bounds__(64) where(uint32_t * out, int spin)
{
  uint32_t hwid, xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  float a = (float) threadIdx.x;
  for (int k = 0; k < spin; ++k) a = a * 1.0001f + 0.5f;
  if (threadIdx.x == 0) out[blockIdx.x] = (hwid & 0xffffu) | ((xcc & 0xfu) << 16) | (a == 123.f ? 1u << 31 : 0u);
}

static int run(const char * name, const std::vector<uint32_t> & mask)
{
  hipStream_t st;
  if (mask.empty()) CHECK(hipStreamCreate(&st));
  else CHECK(hipExtStreamCreateWithCUMask(&st, (uint32_t) mask.size(), mask.data()));
  const int nblk = 65536;
  uint32_t * d = nullptr;
  CHECK(hipMalloc(&d, nblk * 4));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL(where, dim3(nblk), dim3(64), 0, st, d, 2000);
  CHECK(hipEventRecord(e0, st));
  hipLaunchKernelGGL(where, dim3(nblk), dim3(64), 0, st, d, 20000);
  CHECK(hipEventRecord(e1, st));
  CHECK(hipStreamSynchronize(st));
  float ms = 0;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<uint32_t> h(nblk);
  CHECK(hipMemcpy(h.data(), d, nblk * 4, hipMemcpyDeviceToHost));
  std::map<int, std::set<int>> cus;          // xcc -> {se << 8 | sh << 4 | cu}
  for (uint32_t v : h)
    {
      const int xcc = (v >> 16) & 0xf, cu = (v >> 8) & 0xf, sh = (v >> 12) & 1, se = (v >> 13) & 7;
      cus[xcc].insert(se << 8 | sh << 4 | cu);
    }
  size_t total = 0;
  std::printf("%-28s %7.2f ms |", name, ms);
  for (auto & kv : cus) { std::printf(" xcc%d:%zu", kv.first, kv.second.size()); total += kv.second.size(); }
  std::printf(" | %zu CUs\n", total);
  CHECK(hipFree(d));
  CHECK(hipStreamDestroy(st));
  return 0;
}

int main()
{
  run("no mask", {});
  std::vector<uint32_t> m(8, 0);
  for (int b = 0; b < 32; ++b) m[b >> 5] |= 1u << (b & 31);
  run("bits 0..31", m);
  m.assign(8, 0);
  for (int b = 0; b < 256; b += 8) m[b >> 5] |= 1u << (b & 31);
  run("every 8th bit of 256", m);
  m.assign(8, 0);
  for (int b = 0; b < 256; ++b) if ((b & 7) != 0) m[b >> 5] |= 1u << (b & 31);
  run("all but every 8th of 256", m);
  m.assign(8, 0xFFFFFFFFu);
  run("256 bits set", m);
  m.assign(1, 0x0000FFFFu);
  run("one word 0x0000FFFF", m);
  m.assign(2, 0xFFFFFFFFu);
  run("64 bits set", m);
  m.assign(8, 0);
  for (int b = 0; b < 40; ++b) { const int bit = (b * 256) / 40; m[bit >> 5] |= 1u << (bit & 31); }
  run("40 spread bits", m);
  return 0;
}
