#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef unsigned int u32;
__global__ void k(const u32 * a, const u32 * b, const u32 * c, u32 * o, int n)
{
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  u32 r;
  asm volatile("v_pk_maximum3_f16 %0, %1, %2, %3" : "=v"(r) : "v"(a[i]), "v"(b[i]), "v"(c[i]));
  o[i] = r;
}
int main()
{
  const int n = 1 << 22;
  std::vector<u32> a(n), b(n), c(n), o(n);
  auto rnd = [](u32 lo, u32 hi) { return lo + (u32) (rand() % (int) (hi - lo + 1)); };
  for (int pass = 0; pass < 3; ++pass)
    {
      u32 lo = pass == 0 ? 0x0400 : (pass == 1 ? 0x0000 : 0x0001), hi = pass == 0 ? 0x7BFF : (pass == 1 ? 0x03FF : 0x7BFF);
      for (int i = 0; i < n; ++i)
        {
          a[i] = rnd(lo, hi) | (rnd(lo, hi) << 16); b[i] = rnd(lo, hi) | (rnd(lo, hi) << 16); c[i] = rnd(lo, hi) | (rnd(lo, hi) << 16);
          if (i % 7 == 0) b[i] = a[i];
          if (i % 11 == 0) { a[i] = (a[i] & 0xffff0000u) | lo; c[i] = (c[i] & 0xffffu) | (hi << 16); }
        }
      u32 *da, *db, *dc, *dout;
      hipMalloc(&da, n * 4); hipMalloc(&db, n * 4); hipMalloc(&dc, n * 4); hipMalloc(&dout, n * 4);
      hipMemcpy(da, a.data(), n * 4, hipMemcpyHostToDevice); hipMemcpy(db, b.data(), n * 4, hipMemcpyHostToDevice); hipMemcpy(dc, c.data(), n * 4, hipMemcpyHostToDevice);
      hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, da, db, dc, dout, n);
      hipMemcpy(o.data(), dout, n * 4, hipMemcpyDeviceToHost);
      long bad = 0;
      for (int i = 0; i < n; ++i)
        {
          auto mx = [](u32 x, u32 y, u32 z) { u32 m = x > y ? x : y; return m > z ? m : z; };
          u32 e = mx(a[i] & 0xffff, b[i] & 0xffff, c[i] & 0xffff) | (mx(a[i] >> 16, b[i] >> 16, c[i] >> 16) << 16);
          if (e != o[i]) { if (bad < 5) printf("  mismatch: %08x %08x %08x -> %08x expected %08x\n", a[i], b[i], c[i], o[i], e); ++bad; }
        }
      printf("range [%04x, %04x]: %ld mismatches of %d\n", lo, hi, bad, n);
    }
  return 0;
}
