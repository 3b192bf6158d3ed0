// vsx_msa.hip -- star multiple alignment, profile and consensus of MANY clusters in one device pass
// (SURVEY.md 8f "next" #3; the host form of the same algorithm is vsx_msa.cpp, which is also this file's parity partner).
//
// Restates src/core/msa.cpp of the reference as a column-addressed fill instead of its row-serial print loop:
//   find_max_insertions_per_position (:154-189)  -> host, while the CIGAR text is parsed (O(runs))
//   column of centroid position p                 = sum_{k<p}(maxins[k]+1) + maxins[p]     (host prefix sum, O(clen))
//   process_and_print_centroid / compute_and_print_msa (:268-426): every row is '-' except
//       'M' run at (cpos,spos,n):  row[col[cpos+j]]              = seq[spos+j]
//       'D' run at (cpos,spos,n):  row[col[cpos]-maxins[cpos]+j] = seq[spos+j]   (inserted block, left-justified, :372-395)
//       'I' run: gaps (already '-')
//     -> vsx_msa_rows_kernel, one workgroup per row
//   update_profile (:92-141) -> vsx_msa_profile_kernel, tiles of 64 columns x <=4096 rows, 64-bit abundance sums
//   compute_and_print_consensus (:429-500) -> vsx_msa_consensus_kernel, one lane per column
// O(rows x alnlen) byte work, HBM/latency bound; nothing here is GEMM shaped.
#include "../../include/vsx_search.h"

#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

extern "C" void vsx_internal_set_error(const char * msg);
extern "C" int vsx_internal_device(const vsx_ctx * ctx);
extern "C" hipStream_t vsx_internal_stream(const vsx_ctx * ctx);
extern "C" int vsx_internal_usable_cpus(void);
extern "C" const char * vsx_last_error(void);

namespace {

struct MsaRow {            // one output row (centroid, member)
  uint64_t out;            // byte offset of the row in the rows blob
  uint64_t seq;            // byte offset of its sequence in the sequence blob
  uint64_t col;            // offset of its cluster's column table (clen+1 entries of {col, maxins})
  uint32_t run0, run1;     // its runs
  uint32_t alnlen;
  uint32_t pad;
};
struct MsaRun { uint32_t cpos, spos, n, op; };     // op: 0 = M, 1 = D (insert before cpos); 'I' runs are not stored
struct MsaTile {           // profile work item
  uint64_t rows;           // byte offset of the cluster's first row
  uint64_t prof;           // index of the cluster's first profile counter
  uint64_t ab;             // index of the cluster's first abundance
  uint32_t alnlen, col0, row0, row1;
};
struct MsaCluster {
  uint64_t rows, prof;
  uint32_t alnlen, nrows, leftc, rightc;
};

__device__ __forceinline__ int prof_slot(unsigned c)
{
  if (c == '-') return 5;
  c &= ~0x20u;                                 // toupper for the letters (nothing else lands on a letter)
  switch (c)
    {
    case 'A': return 0; case 'C': return 1; case 'G': return 2; case 'T': case 'U': return 3;
    case 'R': case 'Y': case 'S': case 'W': case 'K': case 'M': case 'B': case 'D': case 'H': case 'V': case 'N': return 4;
    default: return -1;
    }
}

__global__ __launch_bounds__(256) void vsx_msa_rows_kernel(const MsaRow * __restrict__ rows, const MsaRun * __restrict__ runs,
                                                           const uint2 * __restrict__ cols, const uint8_t * __restrict__ seqs,
                                                           uint8_t * __restrict__ out)
{
  const MsaRow r = rows[blockIdx.x];
  uint8_t * row = out + r.out;
  for (uint32_t i = threadIdx.x; i <= r.alnlen; i += 256) row[i] = i < r.alnlen ? (uint8_t) '-' : (uint8_t) 0;
  __syncthreads();
  const uint2 * col = cols + r.col;
  const uint8_t * s = seqs + r.seq;
  for (uint32_t k = r.run0; k < r.run1; ++k)
    {
      const MsaRun u = runs[k];
      if (u.op == 0)
        for (uint32_t j = threadIdx.x; j < u.n; j += 256) row[col[u.cpos + j].x] = s[u.spos + j];
      else
        {
          const uint2 c = col[u.cpos];
          const uint32_t base = c.x - c.y;
          for (uint32_t j = threadIdx.x; j < u.n; j += 256) row[base + j] = s[u.spos + j];
        }
    }
}

// 256 threads = 64 columns x 4 row lanes; per-thread 6 counters, LDS reduce over the 4 lanes, one atomic per (column, slot)
__global__ __launch_bounds__(256) void vsx_msa_profile_kernel(const MsaTile * __restrict__ tiles, const uint8_t * __restrict__ rows,
                                                              const uint64_t * __restrict__ ab, unsigned long long * __restrict__ prof)
{
  __shared__ unsigned long long red[4][64][6];
  const MsaTile t = tiles[blockIdx.x];
  const uint32_t cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
  const uint32_t c = t.col0 + cx;
  unsigned long long acc[6] = {0, 0, 0, 0, 0, 0};
  if (c < t.alnlen)
    {
      const uint64_t stride = (uint64_t) t.alnlen + 1;
      const uint8_t * p = rows + t.rows + c;
      for (uint32_t r = t.row0 + ry; r < t.row1; r += 4)
        {
          const int s = prof_slot(p[(uint64_t) r * stride]);
          const unsigned long long a = ab[t.ab + r];
#pragma unroll
          for (int k = 0; k < 6; ++k) acc[k] += (s == k) ? a : 0ull;
        }
    }
#pragma unroll
  for (int k = 0; k < 6; ++k) red[ry][cx][k] = acc[k];
  __syncthreads();
  if (ry == 0 && c < t.alnlen)
#pragma unroll
    for (int k = 0; k < 6; ++k)
      {
        const unsigned long long v = red[0][cx][k] + red[1][cx][k] + red[2][cx][k] + red[3][cx][k];
        if (v) atomicAdd(&prof[t.prof + (uint64_t) c * 6 + k], v);
      }
}

__global__ __launch_bounds__(256) void vsx_msa_consensus_kernel(const MsaCluster * __restrict__ cl, const uint32_t * __restrict__ colblk_cluster,
                                                                const uint32_t * __restrict__ colblk_col0,
                                                                const unsigned long long * __restrict__ prof, uint8_t * __restrict__ rows)
{
  const MsaCluster c = cl[colblk_cluster[blockIdx.x]];
  const uint32_t i = colblk_col0[blockIdx.x] + threadIdx.x;
  if (i > c.alnlen) return;
  uint8_t * crow = rows + c.rows + (uint64_t) (c.nrows - 1) * ((uint64_t) c.alnlen + 1);
  if (i == c.alnlen) { crow[i] = 0; return; }
  if (i < c.leftc || i >= c.alnlen - c.rightc) { crow[i] = '+'; return; }
  const unsigned long long * p = prof + c.prof + (uint64_t) i * 6;
  // bit k of best_sym = nucleotide k; index into "-ACMGRSVTWYHKDBN"
  unsigned best_sym = 0; unsigned long long best = 0;
#pragma unroll
  for (unsigned k = 0; k < 4; ++k)
    if (p[k] > best) { best = p[k]; best_sym = 1u << k; }
  if (best == 0 && p[4] > 0) { best = p[4]; best_sym = 15; }
  const char sym4[17] = "-ACMGRSVTWYHKDBN";
  crow[i] = best >= p[5] ? (uint8_t) sym4[best_sym] : (uint8_t) '-';
}

int mfail(int code, const char * what, hipError_t e)
{
  std::string m = std::string(what) + ": " + hipGetErrorString(e);
  vsx_internal_set_error(m.c_str());
  return code;
}
int minval(const char * what) { vsx_internal_set_error(what); return VSX_EINVAL; }

struct DevBuf {
  void * p = nullptr;
  ~DevBuf() { if (p) (void) hipFree(p); }
};

#define MCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return mfail(e_ == hipErrorOutOfMemory ? VSX_ENOMEM : VSX_EHIP, #x, e_); } while (0)

template <typename T> int upload(DevBuf & b, const std::vector<T> & v, hipStream_t st)
{
  MCHK(hipMalloc(&b.p, std::max<size_t>(v.size(), 1) * sizeof(T)));
  if (!v.empty()) MCHK(hipMemcpyAsync(b.p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice, st));
  return VSX_OK;
}

// one device pass over clusters [c0, c1)
int msa_chunk(hipStream_t st, uint32_t c0, uint32_t c1, const uint64_t * cstart, const char * const * seqs, const uint32_t * lens,
              const char * const * cigars, const uint64_t * abundances, vsx_msa_out * outs)
{
  const bool dbg = std::getenv("VSX_DEBUG_TIMING") != nullptr;
  auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  const double t_0 = now();
  std::vector<MsaRow> rows;
  std::vector<MsaRun> runs;
  std::vector<uint2> cols;
  std::vector<uint8_t> blob;
  std::vector<uint64_t> ab;
  std::vector<MsaTile> tiles;
  std::vector<MsaCluster> cls;
  std::vector<uint32_t> cb_cluster, cb_col0;
  std::vector<uint32_t> maxins;
  uint64_t rows_bytes = 0, prof_count = 0;

  for (uint32_t c = c0; c < c1; ++c)
    {
      const uint64_t m0 = cstart[c], m1 = cstart[c + 1];
      if (m1 <= m0) return minval("vsx_msa_device: empty cluster");
      const uint64_t n = m1 - m0;
      const uint32_t clen = lens[m0];
      const size_t run_first = runs.size();
      const size_t row_first = rows.size();
      maxins.assign((size_t) clen + 1, 0);
      // centroid row = one 'M' run over itself
      MsaRow cr {};
      cr.seq = blob.size(); cr.run0 = (uint32_t) runs.size();
      if (clen) runs.push_back(MsaRun {0, 0, clen, 0});
      cr.run1 = (uint32_t) runs.size();
      rows.push_back(cr);
      blob.insert(blob.end(), (const uint8_t *) seqs[m0], (const uint8_t *) seqs[m0] + clen);
      ab.push_back(abundances ? abundances[m0] : 1);
      for (uint64_t i = m0 + 1; i < m1; ++i)
        {
          const char * cg = cigars[i];
          if (!cg || !seqs[i]) return minval("vsx_msa_device: member without CIGAR or sequence");
          MsaRow mr {};
          mr.seq = blob.size(); mr.run0 = (uint32_t) runs.size();
          uint64_t cpos = 0, spos = 0;
          bool last_d = false;
          while (*cg)
            {
              uint64_t k = 0; bool any = false;
              while (*cg >= '0' && *cg <= '9') { k = k * 10 + (uint64_t) (*cg - '0'); ++cg; any = true; if (k > 0xffffffffull) return minval("vsx_msa_device: run too long"); }
              if (!*cg) break;
              const char op = *cg++;
              if (!any) k = 1;
              if (op == 'M')
                {
                  if (cpos + k > clen || spos + k > lens[i]) return minval("vsx_msa_device: CIGAR does not fit the sequences");
                  if (k) runs.push_back(MsaRun {(uint32_t) cpos, (uint32_t) spos, (uint32_t) k, 0});
                  cpos += k; spos += k; if (k) last_d = false;
                }
              else if (op == 'I') { if (cpos + k > clen) return minval("vsx_msa_device: CIGAR does not fit the sequences"); cpos += k; if (k) last_d = false; }
              else if (op == 'D')
                {
                  if (last_d) return minval("vsx_msa_device: adjacent 'D' runs");
                  if (spos + k > lens[i]) return minval("vsx_msa_device: CIGAR does not fit the sequences");
                  if (k) runs.push_back(MsaRun {(uint32_t) cpos, (uint32_t) spos, (uint32_t) k, 1});
                  maxins[cpos] = std::max<uint32_t>(maxins[cpos], (uint32_t) k);
                  spos += k; last_d = true;
                }
            }
          if (cpos != clen) return minval("vsx_msa_device: CIGAR does not span the centroid");
          mr.run1 = (uint32_t) runs.size();
          rows.push_back(mr);
          blob.insert(blob.end(), (const uint8_t *) seqs[i], (const uint8_t *) seqs[i] + lens[i]);
          ab.push_back(abundances ? abundances[i] : 1);
        }
      (void) run_first;
      // column table
      const uint64_t col_first = cols.size();
      uint64_t p = 0;
      for (uint32_t k = 0; k <= clen; ++k)
        {
          p += maxins[k];
          if (p > 0xfffffff0ull) return minval("vsx_msa_device: alignment too long");
          cols.push_back(make_uint2((uint32_t) p, maxins[k]));
          ++p;
        }
      const uint32_t alnlen = (uint32_t) (p - 1);       // the column "of position clen" is the end of the alignment
      MsaCluster mc {};
      mc.rows = rows_bytes; mc.prof = prof_count; mc.alnlen = alnlen; mc.nrows = (uint32_t) n + 1;
      mc.leftc = maxins[0]; mc.rightc = maxins[clen];
      if (clen == 0) { mc.leftc = alnlen; mc.rightc = 0; }
      cls.push_back(mc);
      for (size_t r = row_first; r < rows.size(); ++r)
        {
          rows[r].out = rows_bytes + (uint64_t) (r - row_first) * ((uint64_t) alnlen + 1);
          rows[r].col = col_first; rows[r].alnlen = alnlen;
        }
      const uint64_t ab_first = ab.size() - n;
      for (uint32_t col0 = 0; col0 < alnlen; col0 += 64)
        for (uint64_t r0 = 0; r0 < n; r0 += 4096)
          tiles.push_back(MsaTile {rows_bytes, prof_count, ab_first, alnlen, col0, (uint32_t) r0, (uint32_t) std::min<uint64_t>(n, r0 + 4096)});
      for (uint32_t col0 = 0; col0 <= alnlen; col0 += 256) { cb_cluster.push_back(c - c0); cb_col0.push_back(col0); }
      rows_bytes += (n + 1) * ((uint64_t) alnlen + 1);
      prof_count += (uint64_t) alnlen * 6;
    }
  if (rows.size() > 0x7fffffffull || runs.size() > 0xffffffffull) return minval("vsx_msa_device: too many rows in one pass");

  const double t_1 = now();
  DevBuf d_rows, d_runs, d_cols, d_blob, d_ab, d_tiles, d_cls, d_cbc, d_cb0, d_out, d_prof;
  int rc;
  if ((rc = upload(d_rows, rows, st)) || (rc = upload(d_runs, runs, st)) || (rc = upload(d_cols, cols, st)) || (rc = upload(d_blob, blob, st))
      || (rc = upload(d_ab, ab, st)) || (rc = upload(d_tiles, tiles, st)) || (rc = upload(d_cls, cls, st)) || (rc = upload(d_cbc, cb_cluster, st))
      || (rc = upload(d_cb0, cb_col0, st)))
    return rc;
  MCHK(hipMalloc(&d_out.p, std::max<uint64_t>(rows_bytes, 1)));
  MCHK(hipMalloc(&d_prof.p, std::max<uint64_t>(prof_count, 1) * sizeof(uint64_t)));
  MCHK(hipMemsetAsync(d_prof.p, 0, std::max<uint64_t>(prof_count, 1) * sizeof(uint64_t), st));
  hipLaunchKernelGGL(vsx_msa_rows_kernel, dim3((unsigned) rows.size()), dim3(256), 0, st, (const MsaRow *) d_rows.p, (const MsaRun *) d_runs.p,
                     (const uint2 *) d_cols.p, (const uint8_t *) d_blob.p, (uint8_t *) d_out.p);
  if (!tiles.empty())
    hipLaunchKernelGGL(vsx_msa_profile_kernel, dim3((unsigned) tiles.size()), dim3(256), 0, st, (const MsaTile *) d_tiles.p, (const uint8_t *) d_out.p,
                       (const uint64_t *) d_ab.p, (unsigned long long *) d_prof.p);
  hipLaunchKernelGGL(vsx_msa_consensus_kernel, dim3((unsigned) cb_cluster.size()), dim3(256), 0, st, (const MsaCluster *) d_cls.p,
                     (const uint32_t *) d_cbc.p, (const uint32_t *) d_cb0.p, (const unsigned long long *) d_prof.p, (uint8_t *) d_out.p);
  MCHK(hipGetLastError());
  if (dbg) MCHK(hipStreamSynchronize(st));
  const double t_2 = now();
  std::vector<char> h_rows((size_t) rows_bytes);
  std::vector<uint64_t> h_prof((size_t) prof_count);
  if (rows_bytes) MCHK(hipMemcpyAsync(h_rows.data(), d_out.p, rows_bytes, hipMemcpyDeviceToHost, st));
  if (prof_count) MCHK(hipMemcpyAsync(h_prof.data(), d_prof.p, prof_count * sizeof(uint64_t), hipMemcpyDeviceToHost, st));
  MCHK(hipStreamSynchronize(st));
  const double t_3 = now();

  for (uint32_t c = c0; c < c1; ++c)
    {
      const MsaCluster & mc = cls[c - c0];
      vsx_msa_out * o = &outs[c];
      const size_t rb = (size_t) mc.nrows * ((size_t) mc.alnlen + 1);
      o->alnlen = mc.alnlen; o->n_rows = mc.nrows;
      o->rows = (char *) std::malloc(std::max<size_t>(rb, 1));
      o->profile = (uint64_t *) std::malloc(std::max<size_t>((size_t) mc.alnlen * 6, 1) * sizeof(uint64_t));
      o->consensus = (char *) std::malloc((size_t) mc.alnlen + 1);
      if (!o->rows || !o->profile || !o->consensus) { vsx_internal_set_error("vsx_msa_device: out of host memory"); return VSX_ENOMEM; }
      std::memcpy(o->rows, h_rows.data() + mc.rows, rb);
      if (mc.alnlen) std::memcpy(o->profile, h_prof.data() + mc.prof, (size_t) mc.alnlen * 6 * sizeof(uint64_t));
      // the consensus sequence = the uncensored columns whose best count >= gap count (msa.cpp:474-492): every non-'-'
      // column, plus the all-zero columns (best_sym 0 prints as '-' and IS appended by the reference)
      const char * crow = o->rows + (size_t) (mc.nrows - 1) * ((size_t) mc.alnlen + 1);
      size_t k = 0;
      for (uint32_t i = mc.leftc; i + mc.rightc < mc.alnlen; ++i)
        if (crow[i] != '-' || o->profile[(size_t) i * 6 + 5] == 0) o->consensus[k++] = crow[i];
      o->consensus[k] = 0;
      o->conslen = k;
    }
  if (dbg)
    std::fprintf(stderr, "[vsx msa] clusters %u rows %zu: parse %.3f s, upload+kernels %.3f s, download %.3f s (%.1f MB), copy-out %.3f s\n", c1 - c0,
                 rows.size(), t_1 - t_0, t_2 - t_1, t_3 - t_2, (double) (rows_bytes + prof_count * 8) / 1e6, now() - t_3);
  return VSX_OK;
}

}  // namespace

extern "C" int vsx_msa_device_batch(vsx_ctx * ctx, uint32_t n_clusters, const uint64_t * cluster_start, const char * const * seqs,
                                    const uint32_t * lens, const char * const * cigars, const uint64_t * abundances, vsx_msa_out * outs)
{
  if (!ctx || !cluster_start || !seqs || !lens || !cigars || !outs) return minval("vsx_msa_device_batch: null argument");
  for (uint32_t c = 0; c < n_clusters; ++c) std::memset(&outs[c], 0, sizeof outs[c]);
  const int device = vsx_internal_device(ctx);
  MCHK(hipSetDevice(device));
  for (uint32_t c = 0; c < n_clusters; ++c)
    if (cluster_start[c + 1] <= cluster_start[c]) return minval("vsx_msa_device_batch: empty cluster");
  // The kernels are a few per cent of the call; CIGAR parsing, the PCIe copies and the per-cluster result buffers are host
  // work, so the clusters are split (by rows) over host threads, each driving its own stream through passes of bounded size
  // (estimate per cluster: rows x (centroid + longest member)).
  const uint64_t total_rows = n_clusters ? cluster_start[n_clusters] - cluster_start[0] : 0;
  const int nth = (int) std::max<uint64_t>(1, std::min<uint64_t>({(uint64_t) vsx_internal_usable_cpus(), total_rows / 16384, (uint64_t) n_clusters, 32}));
  const uint64_t budget = nth > 1 ? (1ull << 28) : (1ull << 30);
  std::vector<int> rcs((size_t) nth, VSX_OK);
  std::vector<std::string> msgs((size_t) nth);
  auto part = [&](int t, hipStream_t st) {
    // clusters whose first row falls into the t-th share of the rows
    const uint64_t base = cluster_start[0];
    const uint64_t lo = base + total_rows * (uint64_t) t / (uint64_t) nth, hi = base + total_rows * (uint64_t) (t + 1) / (uint64_t) nth;
    uint32_t c0 = (uint32_t) (std::lower_bound(cluster_start, cluster_start + n_clusters, lo) - cluster_start);
    const uint32_t cend = (uint32_t) (std::lower_bound(cluster_start, cluster_start + n_clusters, hi) - cluster_start);
    int rc = VSX_OK;
    while (c0 < cend && rc == VSX_OK)
      {
        uint64_t est = 0;
        uint32_t c1 = c0;
        while (c1 < cend)
          {
            const uint64_t m0 = cluster_start[c1], m1 = cluster_start[c1 + 1];
            uint64_t longest = 0;
            for (uint64_t i = m0; i < m1; ++i) longest = std::max<uint64_t>(longest, lens[i]);
            const uint64_t e = (m1 - m0 + 1) * ((uint64_t) lens[m0] + longest + 1);
            if (c1 > c0 && est + e > budget) break;
            est += e; ++c1;
          }
        rc = msa_chunk(st, c0, c1, cluster_start, seqs, lens, cigars, abundances, outs);
        c0 = c1;
      }
    rcs[(size_t) t] = rc;
    if (rc != VSX_OK) msgs[(size_t) t] = vsx_last_error();       // thread-local: carry it to the caller's thread
  };
  int rc = VSX_OK;
  if (nth == 1)
    {
      part(0, vsx_internal_stream(ctx));
      rc = rcs[0];
    }
  else
    {
      std::vector<std::thread> th;
      for (int t = 0; t < nth; ++t)
        th.emplace_back([&, t] {
          hipStream_t st = nullptr;
          if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess)
            { rcs[(size_t) t] = VSX_EHIP; msgs[(size_t) t] = "vsx_msa_device_batch: cannot create a stream"; return; }
          part(t, st);
          (void) hipStreamDestroy(st);
        });
      for (auto & x : th) x.join();
      for (int t = 0; t < nth && rc == VSX_OK; ++t)
        if (rcs[(size_t) t] != VSX_OK) { rc = rcs[(size_t) t]; vsx_internal_set_error(msgs[(size_t) t].c_str()); }
    }
  if (rc != VSX_OK)
    for (uint32_t c = 0; c < n_clusters; ++c) vsx_msa_out_free(&outs[c]);
  return rc;
}

extern "C" int vsx_msa_device(vsx_ctx * ctx, uint32_t n, const char * const * seqs, const uint32_t * lens, const char * const * cigars,
                              const uint64_t * abundances, vsx_msa_out * out)
{
  if (!out || n == 0) return minval("vsx_msa_device: null argument");
  const uint64_t start[2] = {0, n};
  return vsx_msa_device_batch(ctx, 1, start, seqs, lens, cigars, abundances, out);
}
