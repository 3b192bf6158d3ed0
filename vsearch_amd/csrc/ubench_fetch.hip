// ubench_fetch.hip -- calibration of rocprofv3's FETCH_SIZE on the traceback's own access pattern (VERDICT r03 "next" 1a).
//
// MI355X_MICROARCH.md: on gfx950 FETCH_SIZE = TCC_EA0_RDREQ x 64 B reports HALF the bytes of a wide coalesced read (128-byte
// requests tallied at 64) and "other access widths are uncalibrated".  vsx_traceback_ck_kernel reads 12 B (row checkpoints, pairs
// 768 B apart) and 16 B (column checkpoints, blocks 1 KB apart) per lane out of distinct lines, so whether its 9.7e6 KiB per
// launch mean 10 GB or 20 GB decides what bounds it.  Every mode below touches each 128-byte line of a large buffer AT MOST ONCE,
// so the true traffic is known: lines x 128 B (or x 64 B if the fabric fetches half lines for such requests).
//
//   mode 0  control: 16 B per lane, fully coalesced stream                                (bytes = n x 16)
//   mode 1  12 B per lane (global_load_dwordx3), every lane its own line, 9 loads 768 B apart in flight (the row checkpoints
//           as one lane sees them)                                                          (lines = threads x 9)
//   mode 2  the same, but the 8 lanes of a "task" share one 48-byte window (lo / hi partners read the same 12 B): what the
//           task-major slot order gives when the 8 pairs of a task travel together       (lines = threads / 8 x 9)
//   mode 3  16 B per lane, every lane its own line, 6 loads 1 KB apart (column checkpoints)  (lines = threads x 6)
//   mode 4  mode 3 with the 8 lanes of a task inside one 64-byte window                     (lines = threads / 8 x 6)
//
// build: hipcc --offload-arch=gfx950 -O3 -o ubench_fetch ubench_fetch.hip ;  run: ./ubench_fetch <mode> [GiB]
// under rocprofv3 --pmc FETCH_SIZE the per-dispatch counter divided by the printed `lines` gives the bytes the counter
// attributes to one partial-line read.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>

typedef unsigned int u32;
typedef unsigned int u32x3 __attribute__((ext_vector_type(3)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
struct __attribute__((aligned(4))) Trio { u32 x, y, z; };
struct __attribute__((aligned(4))) Quad { u32 x, y, z, w; };

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void __launch_bounds__(256) stream16(const u32x4 * __restrict__ p, size_t n, u32 * sink)
{
  u32 acc = 0;
  for (size_t i = (size_t) blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t) gridDim.x * 256)
    {
      const u32x4 v = __builtin_nontemporal_load(p + i);
      acc += v.x ^ v.y ^ v.z ^ v.w;
    }
  if (acc == 0x12345678u) *sink = acc;
}

// SHARE = lanes that read the same window (1 or 8); W = bytes per load (12 or 16); K loads per lane, STRIDE bytes apart;
// consecutive groups are K * STRIDE apart, so no line is touched twice
template <int SHARE, int W, int K, int STRIDE>
__global__ void __launch_bounds__(128) sparse(const unsigned char * __restrict__ base, size_t ngroups, u32 * sink)
{
  const size_t gid = (size_t) blockIdx.x * 128 + threadIdx.x;
  const size_t grp = gid / SHARE;
  if (grp >= ngroups) return;
  const int sub = (int) (gid % SHARE) / 2;                 // lo / hi partners read the same bytes
  const unsigned char * p = base + grp * (size_t) (K * STRIDE) + (size_t) sub * W;
  u32 acc = 0;
  if (W == 12)
    {
      Trio v[K];
#pragma unroll
      for (int e = 0; e < K; ++e) v[e] = *reinterpret_cast<const Trio *>(p + (size_t) e * STRIDE);
#pragma unroll
      for (int e = 0; e < K; ++e) acc += v[e].x ^ v[e].y ^ v[e].z;
    }
  else
    {
      Quad v[K];
#pragma unroll
      for (int e = 0; e < K; ++e) v[e] = *reinterpret_cast<const Quad *>(p + (size_t) e * STRIDE);
#pragma unroll
      for (int e = 0; e < K; ++e) acc += v[e].x ^ v[e].y ^ v[e].z ^ v[e].w;
    }
  if (acc == 0x12345678u) *sink = acc;
}

int main(int argc, char ** argv)
{
  const int mode = argc > 1 ? std::atoi(argv[1]) : 1;
  const double gib = argc > 2 ? std::atof(argv[2]) : 12.0;
  const size_t bytes = (size_t) (gib * (double) (1ull << 30));
  unsigned char * d = nullptr;
  u32 * sink = nullptr;
  CHECK(hipMalloc(&d, bytes + 4096));
  CHECK(hipMalloc(&sink, 4));
  CHECK(hipMemset(d, 1, bytes + 4096));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  double lines = 0, useful = 0;
  const char * what = "";
  for (int rep = 0; rep < 3; ++rep)              // (rocprofv3 sees three dispatches; all touch the same lines, the buffer is >> L2 + MALL)
    {
      CHECK(hipEventRecord(e0));
      if (mode == 0)
        {
          const size_t n = bytes / 16;
          hipLaunchKernelGGL(stream16, dim3(256 * 16), dim3(256), 0, 0, reinterpret_cast<const u32x4 *>(d), n, sink);
          lines = (double) n / 8; useful = (double) n * 16; what = "coalesced 16 B per lane";
        }
      else if (mode == 1 || mode == 2)
        {
          const size_t ngroups = bytes / (9 * 768);
          const size_t threads = ngroups * (mode == 1 ? 1 : 8);
          if (mode == 1) hipLaunchKernelGGL((sparse<1, 12, 9, 768>), dim3((unsigned) ((threads + 127) / 128)), dim3(128), 0, 0, d, ngroups, sink);
          else hipLaunchKernelGGL((sparse<8, 12, 9, 768>), dim3((unsigned) ((threads + 127) / 128)), dim3(128), 0, 0, d, ngroups, sink);
          lines = (double) ngroups * 9; useful = lines * (mode == 1 ? 12 : 48);
          what = mode == 1 ? "12 B per lane, one line each, 9 x 768 B apart" : "12 B per lane, 8 lanes per 48-byte window, 9 x 768 B apart";
        }
      else
        {
          const size_t ngroups = bytes / (6 * 1024);
          const size_t threads = ngroups * (mode == 3 ? 1 : 8);
          if (mode == 3) hipLaunchKernelGGL((sparse<1, 16, 6, 1024>), dim3((unsigned) ((threads + 127) / 128)), dim3(128), 0, 0, d, ngroups, sink);
          else hipLaunchKernelGGL((sparse<8, 16, 6, 1024>), dim3((unsigned) ((threads + 127) / 128)), dim3(128), 0, 0, d, ngroups, sink);
          lines = (double) ngroups * 6; useful = lines * (mode == 3 ? 16 : 64);
          what = mode == 3 ? "16 B per lane, one line each, 6 x 1 KB apart" : "16 B per lane, 8 lanes per 64-byte window, 6 x 1 KB apart";
        }
      CHECK(hipGetLastError());
      CHECK(hipEventRecord(e1));
      CHECK(hipEventSynchronize(e1));
      float ms = 0;
      CHECK(hipEventElapsedTime(&ms, e0, e1));
      std::printf("mode %d (%s): lines %.0f  useful_bytes %.4e  time %.3f ms  -> %.2f G lines/s = %.2f TB/s at 128 B per line, %.2f TB/s at 64 B\n",
                  mode, what, lines, useful, ms, lines / ms * 1e-6, lines * 128 / ms * 1e-9, lines * 64 / ms * 1e-9);
    }
  return 0;
}
