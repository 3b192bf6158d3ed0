// vsx_rank.hip -- hit ranking and compaction on the device: of all pairs of a plan only those the accept
// filter kept leave the GPU, already in the order the reference reports them.
//
// Reference: hits are kept if accepted (or weak) and ordered per query by identity descending, then target ascending
// (hit_compare_byid_typed, src/core/searchcore.cpp:133-179; allpairs_hit_compare_typed, src/commands/allpairs_global.cpp:
// 116-138; search_joinhits :1028-1052 and the qsort at allpairs_global.cpp:522).  The pair list of a plan is grouped by
// query with ascending targets inside a query, so "target ascending" is "pair order" and a STABLE sort by identity inside
// every query's segment reproduces the order.  The identity is the double the accept filter compares (vsx_accept.h), so
// ties are exactly the reference's ties.
//
// Steps (all HBM-bound integer / byte work over n pairs, no MFMA):
//   flag      one lane per aligned pair: keep = verdict accepted (| weak), id = the filter's identity        (24 B read / pair)
//   scan      exclusive prefix sum of the flags (rocPRIM) -> position of every kept pair                   (8 B / pair)
//   scatter   kept pairs in pair order: (id, pair index); segment starts = positions at the query boundaries
//   sort      rocPRIM segmented radix sort, descending id, stable                                            (kept pairs only)
//   gather    the kept pairs' statistics / verdict / id / text offset in ranked order -> compact arrays -> PCIe
#include <hip/hip_runtime.h>
#include <cstring>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/device/device_segmented_radix_sort.hpp>
#include "vsx_accept.h"
#include "vsx_internal.h"

typedef unsigned int u32;

__global__ void __launch_bounds__(256)
vsx_rank_flag_kernel(const VsxFilterDev F, int keep_weak, const VsxPairOut * __restrict__ out, const u32 * __restrict__ pair_ids,
                     const u32 * __restrict__ pair_slot, const VsxTask * __restrict__ tasks, u32 npairs,
                     const u32 * __restrict__ runs, uint64_t runs_capacity, u32 * __restrict__ flag, double * __restrict__ id,
                     u32 * __restrict__ refused /* [0] = count, then pair indices */)
{
  const u32 k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= npairs) return;
  const u32 pid = pair_ids[k];
  const VsxPairOut o = out[pid];
  const u32 v = o.pad;
  const bool keep = (v == 1u) || (keep_weak && v == 2u);
  double idv = 0.0;
  if (keep && o.nruns > 0 && o.run_off + o.nruns <= runs_capacity)
    {
      const u32 ts = pair_slot[k];
      const VsxTask & T = tasks[ts >> 3];
      (void) accept_verdict(F, (int) T.qlen, (int) T.tlen[ts & 7], (int) o.aligned, (int) o.matches, (int) o.mismatches, (int) o.gaps,
                            runs[o.run_off + o.nruns - 1], runs[o.run_off], &idv);
    }
  flag[pid] = keep ? 1u : 0u;
  id[pid] = idv;
  // a pair the 16-bit DP refused AT RUN TIME (overflow rule, align_simd.cpp:1774-1786: score SHRT_MAX, no statistics, no verdict)
  // is neither kept nor rejected: it goes on the list of pairs the caller's linear-memory fallback must decide
  // (searchcore.cpp:806-832).  Rare, so one atomic per such pair; the host sorts the list.
  if (o.score == 32767 /* VSX_SCORE_SENTINEL */ && v == 0u) refused[1 + atomicAdd(refused, 1u)] = pid;
}

// kept pairs in pair order: keys (id) and values (pair index) at their scanned positions
__global__ void __launch_bounds__(256)
vsx_rank_scatter_kernel(const u32 * __restrict__ flag, const u32 * __restrict__ pos, const double * __restrict__ id, u32 n,
                        double * __restrict__ key, u32 * __restrict__ val)
{
  const u32 p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n || !flag[p]) return;
  key[pos[p]] = id[p];
  val[pos[p]] = p;
}

// segment g of the compacted list = the kept pairs of query group g: [pos[qstart[g]], pos[qstart[g + 1]])  (pos[n] = total)
__global__ void __launch_bounds__(256)
vsx_rank_segments_kernel(const u32 * __restrict__ pos, const u32 * __restrict__ qstart, u32 ngroups, u32 * __restrict__ seg)
{
  const u32 g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g <= ngroups) seg[g] = pos[qstart[g]];
}

__global__ void __launch_bounds__(256)
vsx_rank_gather_kernel(const u32 * __restrict__ ranked, const double * __restrict__ key, u32 kept, const VsxPairOut * __restrict__ out,
                       const uint64_t * __restrict__ text_off, VsxRankedOut r)
{
  const u32 j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= kept) return;
  const u32 pid = ranked[j];
  const VsxPairOut o = out[pid];
  r.pair[j] = pid;
  r.score[j] = o.score; r.aligned[j] = o.aligned; r.matches[j] = o.matches; r.mismatches[j] = o.mismatches; r.gaps[j] = o.gaps;
  r.verdict[j] = (uint8_t) o.pad;
  r.id[j] = key[j];
  r.text_off[j] = text_off[pid];
}

// r06: the kept pairs of a ranked plan as the traceback's epilogue listed them (arrival order; VsxFilterDev::kept_pair / kept_id) ->
// the compact arrays; the host orders the few entries (fetch_ranked_core, vsx_host.cpp)
extern "C" hipError_t vsx_rank_gather_list(const uint32_t * d_kept_pair, const double * d_kept_id, uint32_t kept, const VsxPairOut * d_out,
                                           const uint64_t * d_text_off, VsxRankedOut r, hipStream_t st)
{
  if (kept)
    hipLaunchKernelGGL(vsx_rank_gather_kernel, dim3((kept + 255) / 256), dim3(256), 0, st, d_kept_pair, d_kept_id, kept, d_out, d_text_off, r);
  return hipGetLastError();
}

extern "C" hipError_t vsx_rank_flag_scan(VsxFilterDev F, int keep_weak, const VsxPairOut * d_out, const uint32_t * d_pair_ids,
                                         const uint32_t * d_pair_slot, const VsxTask * d_tasks, uint32_t ngpu_pairs, uint32_t n_pairs,
                                         const uint32_t * d_runs, uint64_t runs_capacity, uint32_t * d_flag /* n + 1 */,
                                         uint32_t * d_pos /* n + 1 */, double * d_id, uint32_t * d_refused /* ngpu_pairs + 1 */,
                                         void * d_temp, size_t * temp_bytes, hipStream_t st)
{
  // size query: d_temp == nullptr
  if (!d_temp)
    return rocprim::exclusive_scan(nullptr, *temp_bytes, d_flag, d_pos, 0u, (size_t) n_pairs + 1, rocprim::plus<u32>(), st);
  hipError_t e = hipMemsetAsync(d_flag, 0, ((size_t) n_pairs + 1) * 4, st);        // pairs answered on the host stay unflagged
  if (e != hipSuccess) return e;
  if ((e = hipMemsetAsync(d_refused, 0, 4, st)) != hipSuccess) return e;
  if (ngpu_pairs)
    hipLaunchKernelGGL(vsx_rank_flag_kernel, dim3((ngpu_pairs + 255) / 256), dim3(256), 0, st, F, keep_weak, d_out, d_pair_ids, d_pair_slot,
                       d_tasks, ngpu_pairs, d_runs, runs_capacity, d_flag, d_id, d_refused);
  if ((e = hipGetLastError()) != hipSuccess) return e;
  return rocprim::exclusive_scan(d_temp, *temp_bytes, d_flag, d_pos, 0u, (size_t) n_pairs + 1, rocprim::plus<u32>(), st);
}

extern "C" hipError_t vsx_rank_sort_gather(const uint32_t * d_flag, const uint32_t * d_pos, const double * d_id, uint32_t n_pairs, uint32_t kept,
                                           const uint32_t * d_qstart /* ngroups + 1, d_qstart[ngroups] = n_pairs */, uint32_t ngroups,
                                           double * d_key_in, double * d_key_out, uint32_t * d_val_in, uint32_t * d_val_out, uint32_t * d_seg,
                                           const VsxPairOut * d_out, const uint64_t * d_text_off, VsxRankedOut r,
                                           void * d_temp, size_t * temp_bytes, hipStream_t st)
{
  if (!d_temp)
    return rocprim::segmented_radix_sort_pairs_desc(nullptr, *temp_bytes, d_key_in, d_key_out, d_val_in, d_val_out, kept, ngroups,
                                                    d_seg, d_seg + 1, 0, 64, st);
  hipLaunchKernelGGL(vsx_rank_scatter_kernel, dim3((n_pairs + 255) / 256), dim3(256), 0, st, d_flag, d_pos, d_id, n_pairs, d_key_in, d_val_in);
  hipLaunchKernelGGL(vsx_rank_segments_kernel, dim3((ngroups + 1 + 255) / 256), dim3(256), 0, st, d_pos, d_qstart, ngroups, d_seg);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  e = rocprim::segmented_radix_sort_pairs_desc(d_temp, *temp_bytes, d_key_in, d_key_out, d_val_in, d_val_out, kept, ngroups,
                                               d_seg, d_seg + 1, 0, 64, st);
  if (e != hipSuccess) return e;
  if (kept)
    hipLaunchKernelGGL(vsx_rank_gather_kernel, dim3((kept + 255) / 256), dim3(256), 0, st, d_val_out, d_key_out, kept, d_out, d_text_off, r);
  return hipGetLastError();
}
