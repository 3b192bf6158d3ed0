// vsx_host.cpp -- host side of libvsx: the C-ABI of include/vsx.h, batch planning (pairs -> wavefront
// tasks -> direction-buffer chunks), device memory, result marshalling.
//
// Reference seams replaced here (see include/vsx.h for per-function citations):
//   search16_init/exit/qprep/search16            src/core/align_simd.cpp:1282-2060
//   the "<= 8 targets per call" batch shape      src/core/searchcore.cpp:757-778
// Everything that touches sequence data runs on the GPU; the host only groups indices, evaluates the
// reference's closed-form special cases (empty query / empty target / size guard / forced fallback)
// and turns run lists into CIGAR text.
#include "../../include/vsx.h"
#include "vsx_internal.h"

#include <hip/hip_runtime.h>
#include <cstdlib>
// VSX_POISON=1 (debugging aid, r04): every device block this library hands out is filled with 0xA5 first, so a read of memory nobody
// has written gives the same junk in every run instead of whatever an earlier plan, index or process left there
extern "C" uint64_t vsx_internal_memory_pressure(int device);
extern "C" void vsx_internal_poison(void * p, size_t bytes)
{
  static const bool on = std::getenv("VSX_POISON") != nullptr;
  if (on && p && bytes) { (void) hipMemset(p, 0xA5, bytes); (void) hipDeviceSynchronize(); }
}

#include <algorithm>
#include <cinttypes>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <sched.h>
#include <thread>
#include <atomic>
#include <condition_variable>
#include <mutex>
#include <chrono>
#include <deque>
#include <functional>
#include <memory>
#include <numeric>
#include <string>
#include <vector>

namespace {

thread_local std::string g_err;

int fail(int code, const char * fmt, ...)
{
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}

#define HIPCHK(call)                                                                        \
  do {                                                                                      \
    hipError_t e_ = (call);                                                                 \
    if (e_ != hipSuccess)                                                                   \
      return fail(e_ == hipErrorOutOfMemory ? VSX_ENOMEM : VSX_EHIP, "%s failed: %s (%s:%d)", \
                  #call, hipGetErrorString(e_), __FILE__, __LINE__);                        \
  } while (0)

inline int sat16(int x) { return x > 32767 ? 32767 : (x < -32768 ? -32768 : x); }

inline bool fits(int64_t q, int64_t d)      // search16_fits, core/align_simd.cpp:130-134
{
  return (q + d <= VSX_MAX_SEQLEN_SUM) && (q * d <= VSX_MAX_SEQLEN_PRODUCT);
}

template <typename T>
struct DevBuf {
  T * p = nullptr;
  size_t n = 0;
  ~DevBuf() { release(); }
  void release() { if (p) { (void) hipFree(p); p = nullptr; n = 0; } }
  hipError_t alloc(size_t count)
  {
    release();
    if (count == 0) count = 1;
    hipError_t e = hipMalloc(reinterpret_cast<void **>(&p), count * sizeof(T));
    if (e == hipSuccess) { n = count; vsx_internal_poison(p, count * sizeof(T)); }
    return e;
  }
  hipError_t ensure(size_t count) { return (count <= n && p) ? hipSuccess : alloc(count); }
};


// Scratch pool: hipMalloc/hipFree of the multi-GB checkpoint, slab and run buffers cost seconds per plan (page-table
// set-up), far more than the kernels of a 500 k-pair stage.  A context keeps the blocks its finished plans give back and
// hands them to the next plan (best fit, but never a block more than 4x + 1 MB larger than the request: a retired multi-GB
// checkpoint block must not end up pinned under an 8-byte cursor).  VSX_POOL=0 disables it; vsx_destroy frees everything.
struct PoolBlock { void * p; size_t bytes; };
struct ScratchPool {
  std::vector<PoolBlock> free_blocks;
  std::mutex mu;
  bool enabled = true;
  bool retired = false;                 // the owning context is gone (sequence sets may outlive it): nothing is kept any more
  ~ScratchPool() { for (PoolBlock & b : free_blocks) (void) hipFree(b.p); }
  hipError_t get(size_t bytes, void ** out, size_t * got)
  {
    if (bytes == 0) bytes = 1;
    {
      std::lock_guard<std::mutex> lk(mu);
      size_t best = SIZE_MAX;
      const size_t limit = bytes > (SIZE_MAX >> 3) ? SIZE_MAX : 4 * bytes + (1u << 20);
      for (size_t k = 0; k < free_blocks.size(); ++k)
        if (free_blocks[k].bytes >= bytes && free_blocks[k].bytes <= limit &&
            (best == SIZE_MAX || free_blocks[k].bytes < free_blocks[best].bytes)) best = k;
      if (best != SIZE_MAX)
        {
          *out = free_blocks[best].p; *got = free_blocks[best].bytes;
          free_blocks.erase(free_blocks.begin() + (long) best);
          vsx_internal_poison(*out, std::min<size_t>(*got, (size_t) 256 << 20));
          return hipSuccess;
        }
    }
    if (bytes >= ((size_t) 16 << 20))
      {
        // (r05) never fill the device to the brim: the runtime needs room of its own (kernel scratch of every queue) and aborts the
        // process when it finds none -- profiles/r05/r05b_config5_share_abort.txt.  Below the reserve, idle blocks go first.
        // r06: only as much as restores the reserve -- this pool's largest idle blocks first, the other contexts of the device
        // only if that was not enough -- and said once: on a device that stays near full every plan would otherwise re-hipMalloc its
        // multi-GB blocks (seconds apiece) and the warm-call promise of the pools would vanish silently.
        size_t free_b = 0, total_b = 0;
        int dev = 0;
        const size_t need = bytes + ((size_t) 6 << 30);
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && free_b < need && hipGetDevice(&dev) == hipSuccess)
          {
            static std::atomic<bool> said {false};
            if (!said.exchange(true))
              std::fprintf(stderr, "vsx: device %d is nearly full (%.1f GB free, %.1f GB wanted + 6 GB reserve): idle scratch blocks are being "
                                   "returned; later plans re-allocate theirs (said once)\n", dev, free_b / 1e9, bytes / 1e9);
            if (!trim_until(need)) (void) vsx_internal_memory_pressure(dev);
          }
        (void) hipGetLastError();
      }
    hipError_t e = hipMalloc(out, bytes);
    if (e == hipErrorOutOfMemory) { trim(); (void) hipGetLastError(); e = hipMalloc(out, bytes); }
    if (e == hipErrorOutOfMemory)
      {
        // r05: the idle blocks of the OTHER contexts of this device (a searcher's consumers, a context that ran one huge plan earlier)
        int dev = 0;
        (void) hipGetLastError();
        if (hipGetDevice(&dev) == hipSuccess && vsx_internal_memory_pressure(dev) > 0) e = hipMalloc(out, bytes);
      }
    if (e == hipSuccess) { *got = bytes; vsx_internal_poison(*out, std::min<size_t>(bytes, (size_t) 256 << 20)); }
    return e;
  }
  void put(void * p, size_t bytes)
  {
    if (!p) return;
    if (!enabled) { (void) hipFree(p); return; }
    std::lock_guard<std::mutex> lk(mu);
    if (retired) { (void) hipFree(p); return; }
    free_blocks.push_back(PoolBlock {p, bytes});
    while (free_blocks.size() > 64)                       // bound the number of idle blocks: drop the smallest
      {
        size_t m = 0;
        for (size_t k = 1; k < free_blocks.size(); ++k) if (free_blocks[k].bytes < free_blocks[m].bytes) m = k;
        (void) hipFree(free_blocks[m].p);
        free_blocks.erase(free_blocks.begin() + (long) m);
      }
  }
  size_t idle_bytes()
  {
    std::lock_guard<std::mutex> lk(mu);
    size_t t = 0;
    for (const PoolBlock & b : free_blocks) t += b.bytes;
    return t;
  }
  void trim()
  {
    std::lock_guard<std::mutex> lk(mu);
    for (PoolBlock & b : free_blocks) (void) hipFree(b.p);
    free_blocks.clear();
  }
  // free idle blocks, largest first, until the device reports `want_free` bytes free; false: the pool ran dry first
  bool trim_until(size_t want_free)
  {
    std::lock_guard<std::mutex> lk(mu);
    for (;;)
      {
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && free_b >= want_free) return true;
        if (free_blocks.empty()) return false;
        size_t m = 0;
        for (size_t k = 1; k < free_blocks.size(); ++k) if (free_blocks[k].bytes > free_blocks[m].bytes) m = k;
        (void) hipFree(free_blocks[m].p);
        free_blocks.erase(free_blocks.begin() + (long) m);
      }
  }
  void retire() { trim(); std::lock_guard<std::mutex> lk(mu); retired = true; }
};

template <typename T>
struct PoolBuf {
  T * p = nullptr;
  size_t n = 0, bytes = 0;
  ScratchPool * pool = nullptr;
  ~PoolBuf() { release(); }
  void release() { if (p && pool) pool->put(p, bytes); else if (p) (void) hipFree(p); p = nullptr; n = 0; bytes = 0; }
  hipError_t alloc(ScratchPool * pl, size_t count)       // pl == nullptr: plain hipMalloc / hipFree
  {
    release();
    pool = pl;
    if (count == 0) count = 1;
    void * q = nullptr;
    size_t got = count * sizeof(T);
    hipError_t e = pl ? pl->get(count * sizeof(T), &q, &got) : hipMalloc(&q, count * sizeof(T));
    if (e == hipSuccess) { p = static_cast<T *>(q); n = count; bytes = got; }
    return e;
  }
};

// Stream-ordered scratch shared by the plans of one context: checkpoints and the traceback slab are dead once a plan's
// kernels have finished, and the plans of a context execute in order on its streams, so they can all use ONE block (the
// next plan's kernels are queued behind the previous plan's traceback).  A plan holds a reference; a larger request
// replaces the context's current block, the old one goes back to the pool when its last plan dies.
struct SharedBlock {
  void * p = nullptr;
  size_t bytes = 0;
  ScratchPool * pool = nullptr;
  ~SharedBlock() { if (p) { if (pool) pool->put(p, bytes); else (void) hipFree(p); } }
};
struct SharedSlot;
thread_local const SharedSlot * tl_acquiring_slot = nullptr;      // the slot whose lock THIS thread holds inside acquire() (try_reset)
struct SharedSlot {
  std::mutex mu;
  std::shared_ptr<SharedBlock> cur;
  hipError_t acquire(ScratchPool * pool, size_t bytes, std::shared_ptr<SharedBlock> & out, bool headroom = true)
  {
    std::lock_guard<std::mutex> lk(mu);
    struct Mark { const SharedSlot * prev; explicit Mark(const SharedSlot * s) : prev(tl_acquiring_slot) { tl_acquiring_slot = s; } ~Mark() { tl_acquiring_slot = prev; } } mark(this);
    if (cur && cur->bytes >= bytes) { out = cur; return hipSuccess; }
    cur.reset();                       // (r05) too small: back to the pool as soon as no plan holds it, where memory pressure can free it
    auto b = std::make_shared<SharedBlock>();
    static const bool dbg = std::getenv("VSX_DEBUG_TIMING") != nullptr;
    const auto t0 = std::chrono::steady_clock::now();
    // a little headroom: the slices of a pipeline differ by a few per cent, and replacing a multi-GB block costs a hipMalloc
    hipError_t e = pool->get(headroom ? bytes + bytes / 8 : bytes, &b->p, &b->bytes);
    if (dbg) std::fprintf(stderr, "  shared block: %zu bytes wanted (had %zu), got %zu in %.2f ms\n", bytes, cur ? cur->bytes : (size_t) 0, b->bytes,
                          std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() * 1e3);
    if (e == hipErrorOutOfMemory) { (void) hipGetLastError(); e = pool->get(bytes, &b->p, &b->bytes); }
    if (e != hipSuccess) { b->p = nullptr; return e; }
    b->pool = pool;
    cur = b;
    out = b;
    return hipSuccess;
  }
  void reset() { std::lock_guard<std::mutex> lk(mu); cur.reset(); }
  // (memory pressure may be raised from INSIDE an acquire of this very slot, whose lock is then held by the caller: skip it -- by the
  //  thread-local mark, not by try_lock on a mutex the caller owns, which is undefined; the try_lock that remains is for
  //  slots held by OTHER threads, which may themselves be waiting for a slot of ours)
  void try_reset() { if (tl_acquiring_slot == this) return; if (mu.try_lock()) { cur.reset(); mu.unlock(); } }
  size_t bytes() { std::lock_guard<std::mutex> lk(mu); return cur ? cur->bytes : 0; }
};
template <typename T>
struct SharedBuf {
  std::shared_ptr<SharedBlock> blk;
  T * p = nullptr;
  hipError_t alloc(SharedSlot & slot, ScratchPool * pool, size_t count)
  {
    hipError_t e = slot.acquire(pool, std::max<size_t>(count, 1) * sizeof(T), blk);
    p = (e == hipSuccess) ? static_cast<T *>(blk->p) : nullptr;
    return e;
  }
};

}  // namespace

struct vsx_ctx {
  int device = 0;
  hipStream_t stream = nullptr;      // DP kernels, copies
  hipStream_t stream2 = nullptr;     // traceback kernels (overlap with the next chunk's DP)
  hipStream_t stream_b = nullptr;    // DP kernels of every other pipelined slice (vsx_align_pairs): the next launch fills this one's tail
  hipStream_t stream_up = nullptr;   // plan uploads (a plan may be created while another one runs: vsx_align_pairs pipeline)
  hipStream_t stream_dn = nullptr;   // result downloads (vsx_plan_fetch of slice i-1 while slice i's kernels occupy `stream`)
  vsx_scoring sc {};
  bool force_fallback = false;      // a score/penalty left the 16-bit range: every pair -> sentinel
  bool tb_packed = false;           // VSX_TB_ARITH=packed: plan every task into the TRACK = 1 class (saturating packed ops, capture layout)
  bool ckpt = true;                 // checkpoint + tile-recompute traceback (VSX_TRACEBACK=dirs selects stored direction bits)
  int pen[12] {};                   // clamped CELL penalties: go_q_l, go_t_l, go_q_i, go_t_i, go_q_r, go_t_r, ge_*
  VsxDevParams P {};
  DevBuf<int16_t> d_htop, d_hleft, d_matrix;
  VsxDevParams Pt {};               // the same scoring in TILTED coordinates (VsxDevParams::tilt, vsx_forward_kernel TILT); Pt.tilt == 0: unavailable
  DevBuf<int16_t> d_htop_t, d_hleft_t, d_matrix_t;
  std::shared_ptr<ScratchPool> pool_owner = std::make_shared<ScratchPool>();   // sequence sets hold a reference (they may outlive the context)
  ScratchPool & pool = *pool_owner;
  // Two checkpoint blocks, used alternately by the plans of the context: the DP kernels of plan i+1 run while the traceback of
  // plan i still reads its block, so the launch tails of one fill with the other's waves (a pipeline of small plans lost 15 %
  // to tails when every plan waited for its predecessor's traceback: r02 timeline in DESIGN 5).  ev_tb[s] = "the last traceback
  // that read block s has finished".
  // r04: THREE blocks (VSX_CK_BLOCKS=2 restores two for A/B).  In a pipeline of slices the DP kernel of slice i + 2 used to wait for the
  // traceback of slice i, which shares the device with the DP kernel of slice i + 1 on bad terms (rocprofv3 trace of one call,
  // profiles/r03/r03v_e2e_trace_*.csv); with a third block only slice i + 3 waits for it.  A block is sized by the largest plan that
  // used it (<= 0.4 of what is free when it is made), 20 GB for the 200 k-pair slices of the bench shape.
  SharedSlot shared_dir[3], shared_slab;   // declared after the pool: released first
  std::atomic<uint64_t> req_hwm[2] {{0}, {0}};      // largest checkpoint block / slab a plan of this context asked for since the last reset (bytes)
  hipEvent_t ev_tb[3] = {nullptr, nullptr, nullptr};
  bool ev_tb_used[3] = {false, false, false};
  int last_tb_slot = -1;            // the slot whose traceback was queued last (VSX_ALIGN_SERIAL: the next plan's DP waits for it)
  std::atomic<unsigned> plan_seq {0};
  // r06: which block a new plan takes (vsx_plan_create).  slot_refs[s] = live plans on block s (a plan is destroyed after its traceback
  // has finished), slot_stamp[s] = plan_seq at the last acquisition.
  std::mutex slot_mu;
  int slot_refs[3] = {0, 0, 0};
  unsigned slot_stamp[3] = {0, 0, 0};
  // pinned host memory: results cross PCIe into it (vsx_plan_fetch), one fetch at a time; grow-only
  std::mutex stage_mu;
  uint8_t * stage = nullptr;
  size_t stage_bytes = 0;
  std::vector<unsigned long long *> cursor_slots;   // idle pinned {run cursor, text cursor} pairs of finished plans
  ~vsx_ctx()
  {
    for (hipEvent_t e : ev_tb) if (e) (void) hipEventDestroy(e);
    if (stage) (void) hipHostFree(stage);
    for (auto * c : cursor_slots) (void) hipHostFree(c);
  }
};

#define VSX_SEQSET_POOLED_BYTES (64ull << 20)
struct vsx_seqset {
  vsx_ctx * ctx = nullptr;
  int device = 0;
  uint64_t n = 0;
  uint64_t bytes = 0;
  std::vector<uint64_t> off;
  std::vector<uint32_t> len;
  // Device buffers of sets up to VSX_SEQSET_POOLED_BYTES come from the context's scratch pool and go back to it: a search creates
  // one query set per window, and hipMalloc / hipFree synchronise the WHOLE device -- with the k-mer counting kernel of the next
  // window running that was an 8-11 ms stall per window (r03 timeline); large sets (databases) use hipMalloc as before.
  std::shared_ptr<ScratchPool> pool_ref;   // keeps the pool alive; declared BEFORE the buffers, so it is destroyed after them
  PoolBuf<uint8_t> d_codes;         // VSX_CODE_SLACK bytes | codes | VSX_CODE_SLACK bytes: the traceback stages rows/columns with unaligned dword loads
  uint8_t * codes() const { return d_codes.p + VSX_CODE_SLACK; }
  PoolBuf<uint64_t> d_off;
  PoolBuf<uint32_t> d_len;
  // soft masking for the k-mer index only (vsx_internal_seqset_create_cased): one bit per blob byte, set where the symbol was
  // not an upper-case A C G T U.  The aligner never sees it: alignment is case-blind in the reference too (chrmap_4bit)
  PoolBuf<uint8_t> d_lower;
  uint64_t lower_bytes = 0;
  // VSX_SCORE=arith only: per-sequence "contains a non-ACGT symbol", computed on first use (vsx_purity_kernel) under the lock
  mutable std::mutex impure_mu;
  mutable std::vector<uint8_t> impure;
  mutable bool have_impure = false;
};

namespace {

struct Launch { int rows; int generic; int track; int tilt; int nq /* tasks per wave: 1, or 2 / 4 = a sparse-task class */; uint32_t first, count; uint32_t pair_first, pair_count;
                bool multi = false /* r06: some task of the launch spans several strips (Q > 16 rows): the general kernel; otherwise the ONE variants */; };

struct Chunk {
  uint32_t task_first = 0, task_count = 0;
  uint32_t pair_first = 0, pair_count = 0;      // into the gpu-pair arrays
  std::vector<Launch> launches;
  // (a chunk's checkpoint block starts and ends with VSX_CK_SLACK_DW dwords nobody writes: the traceback reads the nine row-checkpoint
  //  pairs around a tile without clamping their addresses -- vsx_internal.h)
  uint64_t dir_dwords = VSX_CK_SLACK_DW, strip_elems = 0, slab_words = 0;
  hipEvent_t e0 = nullptr, e1 = nullptr, e1b = nullptr, e2 = nullptr;   // DP begin/end (stream), traceback begin/end (stream2)
};

}  // namespace

extern "C" int vsx_internal_usable_cpus(void);
// per device: 0 = not tested yet, 1 = v_pk_maximum3_f16 is the integer maximum on [0, 0x7BFF] (vsx_create's self-test), 2 = it is not
// r05j same-box A/B, 25 000 queries x 32 candidates, two runs each (profiles/r05/r05j_pairprof_first_contact.txt): DP 400 x 400 16.53 -> 14.45 ms,
// 300 x 300 9.96 -> 8.68, 250 x 1000 22.12 -> 20.39, 150 x 1000 16.04 -> 14.50.  Default ON (it engages only where a query has >= 4 eligible tasks)
#ifndef VSX_PAIRPROF_DEFAULT
#define VSX_PAIRPROF_DEFAULT 1
#endif
static std::atomic<int> g_max3_state[64];
static std::mutex g_ctx_mu;                   // the live contexts of the process (vsx_internal_memory_pressure)
static std::vector<vsx_ctx *> g_ctxs;
template <typename T>
static T * dup_array(const std::vector<T> & v)
{
  T * p = (T *) std::malloc(std::max<size_t>(v.size(), 1) * sizeof(T));
  if (p && !v.empty()) std::memcpy(p, v.data(), v.size() * sizeof(T));
  return p;
}


// threads for a memory-bound host pass over `bytes` bytes
static int copy_threads(uint64_t bytes)
{
  return (int) std::max<uint64_t>(1, std::min<uint64_t>((uint64_t) vsx_internal_usable_cpus(), bytes >> 21));
}

// Host worker pool: a plan is built in a dozen short parallel passes (validation, classification, grouping, task tables; the
// fetch spreads its copies); spawning std::threads for each cost more than the passes themselves on 200 k-pair slices.  Workers
// are created once per process; a parallel region pushes its tasks, runs task 0 itself, helps with the queue and waits for its
// own tasks only -- regions of different host threads (the planner of a pipeline, the fetching caller) interleave freely.
namespace {
class WorkerPool {
 public:
  static WorkerPool & get() { static WorkerPool p; return p; }
  template <typename F>
  void run(int nth, F && f)
  {
    if (nth <= 1) { f(0); return; }
    // `left` is only touched under `m`: a worker's last access to the region is its unlock, and the caller can see left == 0 only
    // after locking behind it -- the region lives on the caller's stack and must not be touched once run() returns
    struct Region { int left; std::mutex m; std::condition_variable cv; };
    Region rg;
    rg.left = nth - 1;
    {
      std::lock_guard<std::mutex> lk(mu_);
      for (int t = 1; t < nth; ++t)
        queue_.push_back([&f, &rg, t]() {
          f(t);
          std::lock_guard<std::mutex> lk2(rg.m);
          if (--rg.left == 0) rg.cv.notify_all();
        });
    }
    cv_.notify_all();
    f(0);
    // help: run queued tasks (ours or another region's) instead of sleeping while ours are pending
    for (;;)
      {
        { std::lock_guard<std::mutex> lk(rg.m); if (rg.left == 0) break; }
        std::function<void()> job;
        {
          std::lock_guard<std::mutex> lk(mu_);
          if (!queue_.empty()) { job = std::move(queue_.front()); queue_.pop_front(); }
        }
        if (job) { job(); continue; }
        std::unique_lock<std::mutex> lk(rg.m);
        if (rg.cv.wait_for(lk, std::chrono::microseconds(50), [&] { return rg.left == 0; })) break;
      }
  }
 private:
  WorkerPool()
  {
    const int n = std::max(1, vsx_internal_usable_cpus() - 1);
    for (int i = 0; i < n; ++i)
      workers_.emplace_back([this]() {
        for (;;)
          {
            std::function<void()> job;
            {
              std::unique_lock<std::mutex> lk(mu_);
              cv_.wait(lk, [&] { return stop_ || !queue_.empty(); });
              if (stop_ && queue_.empty()) return;
              job = std::move(queue_.front());
              queue_.pop_front();
            }
            job();
          }
      });
  }
  ~WorkerPool()
  {
    { std::lock_guard<std::mutex> lk(mu_); stop_ = true; }
    cv_.notify_all();
    for (auto & w : workers_) w.join();
  }
  std::mutex mu_;
  std::condition_variable cv_;
  std::deque<std::function<void()>> queue_;
  std::vector<std::thread> workers_;
  bool stop_ = false;
};
}  // namespace

template <typename F>
static void run_threads(int nth, F && f)
{
  WorkerPool::get().run(nth, f);
}

struct vsx_plan {
  vsx_ctx * ctx = nullptr;
  const vsx_seqset * Q = nullptr;
  const vsx_seqset * T = nullptr;
  uint64_t n_pairs = 0;
  VsxFilterDev filter {};                // accept filter evaluated by the traceback kernel (enabled == 0: none)
  std::vector<VsxPairOut> host_out;      // answers of the closed-form / sentinel pairs, parallel to host_pairs
  std::vector<std::string> host_cigar;   // only for the Q == 0 closed form
  std::vector<uint32_t> host_cigar_pair;
  std::vector<uint32_t> host_cigar_run;  // the same alignments as run words ((D << 2) | I), for the exported run buffer

  std::vector<VsxTask> tasks;
  std::vector<uint32_t> pair_slot, pair_ids;
  std::vector<uint64_t> slab_off;
  std::vector<Chunk> chunks;
  uint64_t cells = 0, dir_bytes_total = 0;

  // every device buffer comes from the context's pool (hipFree synchronises the device: it would stall a pipeline of plans)
  PoolBuf<VsxTask> d_tasks;
  PoolBuf<uint32_t> d_pair_slot, d_pair_ids;
  SharedBuf<uint32_t> d_dir[1], d_slab;         // stream-ordered scratch shared by the context's plans
  int dir_slot = 0;                             // which of the context's checkpoint blocks
  bool slot_held = false;                       // counted in ctx->slot_refs[dir_slot]
  bool alt_fwd = false;                         // DP kernels on the context's second DP stream (pipelined slices, see vsx_align_pairs)
  PoolBuf<uint32_t> d_runs;
  PoolBuf<uint64_t> d_slab_off;
  PoolBuf<uint2> d_strip;
  PoolBuf<VsxSlotOut> d_slot;
  PoolBuf<VsxPairOut> d_out;
  PoolBuf<unsigned long long> d_cursor;  // [0] run words used, [1] text bytes used
  PoolBuf<uint8_t> d_text, d_soa;        // CIGAR text and the output arrays, written by vsx_cigar_text_kernel (vsx_tbtext.hip)
  // r06, ranked plans: the lists the traceback's epilogue fills (VsxFilterDev::rank_counts ...), see plan_set_ranked()
  PoolBuf<uint32_t> d_rank_counts, d_kept_pair, d_refused_pair;
  PoolBuf<double> d_kept_id;
  VsxSoaOut soa {};
  uint64_t soa_bytes = 0, soa_off[7] {};
  uint64_t text_capacity = 0;
  unsigned long long * h_cursor = nullptr;   // pinned copy of d_cursor, filled at the end of a run
  std::vector<uint32_t> host_pairs;      // pairs answered without DP (sentinels, empty query), ascending
  uint64_t runs_capacity = 0;
  hipEvent_t ev_begin = nullptr, ev_end = nullptr;
  bool ran = false;
  bool host_patched = false;

  ~vsx_plan()
  {
    for (auto & c : chunks)
      {
        if (c.e0) (void) hipEventDestroy(c.e0);
        if (c.e1) (void) hipEventDestroy(c.e1);
        if (c.e1b) (void) hipEventDestroy(c.e1b);
        if (c.e2) (void) hipEventDestroy(c.e2);
      }
    if (ev_begin) (void) hipEventDestroy(ev_begin);
    if (ev_end) (void) hipEventDestroy(ev_end);
    if (h_cursor) { std::lock_guard<std::mutex> lk(ctx->stage_mu); ctx->cursor_slots.push_back(h_cursor); }
    if (slot_held) { std::lock_guard<std::mutex> lk(ctx->slot_mu); --ctx->slot_refs[dir_slot]; }
  }
};

extern "C" {

// the host worker pool for vsx_search.cpp (fn(0) runs on the caller, which also helps with queued jobs while it waits)
void vsx_internal_run_threads(int nth, void (*fn)(int, void *), void * arg) { run_threads(nth, [&](int t) { fn(t, arg); }); }
// shared with vsx_search.cpp: one thread-local error slot for the whole library
void vsx_internal_set_error(const char * msg) { g_err = msg ? msg : ""; }
const vsx_scoring * vsx_internal_scoring(const vsx_ctx * ctx) { return &ctx->sc; }
int vsx_internal_usable_cpus(void)
{
  // CPUs this process may really use: affinity mask, capped by a cgroup v2 CPU quota
  int n = (int) std::thread::hardware_concurrency();
  cpu_set_t set;
  if (sched_getaffinity(0, sizeof set, &set) == 0) n = CPU_COUNT(&set);
  if (FILE * f = std::fopen("/sys/fs/cgroup/cpu.max", "r"))
    {
      char q[32]; long long period = 0;
      if (std::fscanf(f, "%31s %lld", q, &period) == 2 && std::strcmp(q, "max") != 0 && period > 0)
        n = std::min<int>(n, (int) std::max<long long>(1, std::atoll(q) / period));
      std::fclose(f);
    }
  return std::max(1, n);
}
// Self-test of the host worker pool (no device): `regions` short parallel regions of `width` tasks each, issued from TWO
// host threads at once (as the planner of a pipeline and the fetching caller do); every task adds its index into a per-region
// slot on the issuing thread's stack.  Returns 0 if every region saw exactly the sum of its task indices.
int vsx_internal_pool_selftest(int regions, int width)
{
  std::atomic<int> bad {0};
  auto driver = [&](int salt) {
    for (int r = 0; r < regions; ++r)
      {
        const int w = 1 + (r * 7 + salt) % std::max(1, width);
        std::vector<long long> part((size_t) w, 0);
        volatile long long guard_before = 0x1122334455667788ll;
        run_threads(w, [&](int t) { part[(size_t) t] += (long long) t + 1; });
        volatile long long guard_after = 0x1122334455667788ll;
        long long sum = 0;
        for (long long v : part) sum += v;
        if (sum != (long long) w * (w + 1) / 2 || guard_before != guard_after) bad.fetch_add(1);
      }
  };
  std::thread other(driver, 3);
  driver(0);
  other.join();
  return bad.load();
}
int vsx_internal_device(const vsx_ctx * ctx) { return ctx->device; }
// r05: device memory under pressure.  A context keeps its big stream-ordered blocks (three checkpoint blocks, the slab) and a pool of
// idle scratch between plans, so that a warm call never meets a multi-GB hipMalloc; after ONE very large plan (BASELINE config 5's
// per-GPU share: 3 x 62 GB of checkpoints) those blocks starved every other allocator of the device -- the k-mer scratch of a search on
// the same GPU failed with "out of memory" (profiles/r05/r05a_config5_share_oom.json).  Whoever meets hipErrorOutOfMemory calls this
// once and retries: every live context of the device drops the blocks no plan holds and frees its idle pool.  Returns the bytes released.
uint64_t vsx_internal_memory_pressure(int device)
{
  std::lock_guard<std::mutex> lk(g_ctx_mu);
  uint64_t freed = 0;
  for (vsx_ctx * c : g_ctxs)
    if (c->device == device)
      {
        SharedSlot * slot[4] = {&c->shared_dir[0], &c->shared_dir[1], &c->shared_dir[2], &c->shared_slab};
        for (SharedSlot * sl : slot) sl->try_reset();      // a block a plan still references lives until that plan dies
        freed += c->pool.idle_bytes();
        c->pool.trim();
      }
  return freed;
}
// The big stream-ordered scratch blocks of a context (three checkpoint blocks, the traceback slab): their sizes, and a reservation
// of at least those sizes.  A search runs its windows on several contexts of one device; a context that meets its first full
// window in the middle of a warm search would pay the multi-GB hipMalloc there (0.5-1 s in a 0.15 s call), so the searcher levels
// the contexts after a call (vsx_search.cpp).
// what the plans of a context asked for since the last reset: {checkpoint block, slab} bytes.  The searcher levels its contexts to
// THESE (r04: it used to level to the blocks the contexts hold, and a context that had also run somebody else's 80 GB plan -- the
// bench's aligner context is the searcher's first -- made every search call try, and fail, to give all contexts 88 GB blocks:
// 13 ms of refused hipMallocs per call)
void vsx_internal_scratch_requests(vsx_ctx * ctx, uint64_t out[2], int reset)
{
  for (int k = 0; k < 2; ++k) out[k] = reset ? ctx->req_hwm[k].exchange(0) : ctx->req_hwm[k].load();
}
void vsx_internal_scratch_sizes(vsx_ctx * ctx, uint64_t out[4])
{
  out[0] = ctx->shared_dir[0].bytes(); out[1] = ctx->shared_dir[1].bytes(); out[2] = ctx->shared_dir[2].bytes(); out[3] = ctx->shared_slab.bytes();
}
int vsx_internal_scratch_reserve(vsx_ctx * ctx, const uint64_t want[4])
{
  if (hipSetDevice(ctx->device) != hipSuccess) { (void) hipGetLastError(); return VSX_EHIP; }
  SharedSlot * slot[4] = {&ctx->shared_dir[0], &ctx->shared_dir[1], &ctx->shared_dir[2], &ctx->shared_slab};
  for (int k = 0; k < 4; ++k)
    if (want[k] > slot[k]->bytes())
      {
        std::shared_ptr<SharedBlock> hold;
        const hipError_t e = slot[k]->acquire(&ctx->pool, (size_t) want[k], hold, true);
        if (e != hipSuccess) { (void) hipGetLastError(); return e == hipErrorOutOfMemory ? VSX_ENOMEM : VSX_EHIP; }
      }
  return VSX_OK;
}
static int pick_rows(int Q);
// r06: what the checkpoint block of ONE plan of `ntasks` whole-wave tasks (a query of qlen rows against up to eight targets of tlen columns)
// will ask for -- a caller that knows the largest plan it is going to submit (vsx_cluster_fast: a round of queries, eight candidates each)
// reserves the block once instead of letting it grow over its first rounds; 0 when the context keeps direction bits instead of checkpoints
uint64_t vsx_internal_ckpt_bytes_estimate(const vsx_ctx * ctx, uint64_t ntasks, uint32_t qlen, uint32_t tlen)
{
  if (!ctx->ckpt || qlen == 0 || tlen == 0) return 0;
  const int rows = pick_rows((int) qlen);
  const uint64_t total_lanes = ((uint64_t) qlen + (uint64_t) rows - 1) / (uint64_t) rows, nstrips = (total_lanes + 15) / 16;
  const uint64_t steps = (((uint64_t) tlen + 3) & ~3ull) + 16;
  const int tilt = (ctx->Pt.tilt != 0 && rows >= 4) ? 1 : 0;
  return (ntasks * vsx_ckpt_dwords(nstrips, steps, (uint64_t) rows, tilt) + 2 * VSX_CK_SLACK_DW) * 4;
}
hipStream_t vsx_internal_stream(const vsx_ctx * ctx) { return ctx->stream; }
// host copies of the set's lengths (the k-mer index build of long words lays its key slots out by their running sum)
const uint32_t * vsx_internal_seqset_host_lengths(const vsx_seqset * s) { return s ? s->len.data() : nullptr; }
void vsx_internal_seqset_device(const vsx_seqset * s, const uint8_t ** codes, const uint64_t ** off, const uint32_t ** len, uint64_t * n)
{
  *codes = s->codes(); *off = s->d_off.p; *len = s->d_len.p; *n = s->n;
}

const char * vsx_version_string(void) { return "libvsx 0.1.0 (gfx950)"; }

const char * vsx_last_error(void) { return g_err.c_str(); }

int vsx_device_count(void)
{
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) { (void) hipGetLastError(); return 0; }
  int usable = 0;
  for (int d = 0; d < n; ++d)
    {
      hipDeviceProp_t p;
      if (hipGetDeviceProperties(&p, d) == hipSuccess && std::strncmp(p.gcnArchName, "gfx950", 6) == 0) ++usable;
    }
  return usable;
}

int vsx_create(vsx_ctx ** out, const vsx_scoring * s, int device)
{
  if (!out || !s) return fail(VSX_EINVAL, "vsx_create: null argument");
  *out = nullptr;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
    {
      (void) hipGetLastError();
      return fail(VSX_ENODEVICE, "vsx_create: no HIP device visible (libvsx has no CPU fallback)");
    }
  if (device < 0 || device >= ndev) return fail(VSX_EINVAL, "vsx_create: device %d out of range (0..%d)", device, ndev - 1);
  hipDeviceProp_t prop;
  HIPCHK(hipGetDeviceProperties(&prop, device));
  if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    return fail(VSX_ENODEVICE, "vsx_create: device %d is %s, libvsx is built for gfx950 only", device, prop.gcnArchName);
  HIPCHK(hipSetDevice(device));

  auto * c = new vsx_ctx;
  c->device = device;
  c->sc = *s;
  if (const char * mode = std::getenv("VSX_TRACEBACK")) c->ckpt = std::strcmp(mode, "dirs") != 0;
  if (const char * mode = std::getenv("VSX_POOL")) c->pool.enabled = std::strcmp(mode, "0") != 0;
  if (const char * mode = std::getenv("VSX_TB_ARITH")) c->tb_packed = std::strcmp(mode, "packed") == 0;

  // search16_init, align_simd.cpp:1282-1376: scores must fit a CELL, each penalty SHRT_MAX/(1+CDEPTH)
  auto clamp = [&](int64_t v, int64_t lim) -> int {
    if (v > lim) { c->force_fallback = true; return (int) lim; }
    if (v < -lim) { c->force_fallback = true; return (int) -lim; }
    return (int) v;
  };
  const int match = clamp(s->match, 32767);
  const int mism = clamp(s->mismatch, 32767);
  const int64_t raw[12] = {s->gap_open_query_left, s->gap_open_target_left, s->gap_open_query_interior,
                           s->gap_open_target_interior, s->gap_open_query_right, s->gap_open_target_right,
                           s->gap_ext_query_left, s->gap_ext_target_left, s->gap_ext_query_interior,
                           s->gap_ext_target_interior, s->gap_ext_query_right, s->gap_ext_target_right};
  for (int k = 0; k < 12; ++k) c->pen[k] = clamp(raw[k], 6553);
  const int goql = c->pen[0], gotl = c->pen[1], goqi = c->pen[2], goti = c->pen[3], goqr = c->pen[4], gotr = c->pen[5];
  const int geql = c->pen[6], getl = c->pen[7], geqi = c->pen[8], geti = c->pen[9], geqr = c->pen[10], getr = c->pen[11];

  auto pk = [](int v) -> uint32_t { uint32_t x = (uint32_t) v & 0xffffu; return x | (x << 16); };
  VsxDevParams & P = c->P;
  P.match_pk = pk(match);
  P.qrq_i_pk = pk(goqi + geqi); P.rq_i_pk = pk(geqi);
  P.qrq_r_pk = pk(goqr + geqr); P.rq_r_pk = pk(geqr);
  P.qrt_i = goti + geti; P.rt_i = geti;
  P.qrt_r = gotr + getr; P.rt_r = getr;
  P.match = match; P.mismatch = mism;
  P.n_mismatch = s->n_mismatch ? 1 : 0;
  P.top_open = goql; P.top_step = geql;
  P.share_sub = (goqi + geqi == goti + geti && !std::getenv("VSX_NO_SHARE_SUB")) ? 1 : 0;
  int pmax = 0;
  for (int v : {goql + geql, goqi + geqi, goqr + geqr, gotl + getl, goti + geti, gotr + getr}) pmax = std::max(pmax, v);
  P.smin = -32768 + pmax;                                       // compute_score_min :1432-1444

  // border chains, evaluated with the reference's saturating steps
  std::vector<int16_t> htop(VSX_TABLE_LEN), hleft(VSX_TABLE_LEN), matrix(256);
  {
    int h = -goql - geql;                                       // :1895-1910 (plain casts, in range by the 6553 limit)
    for (int j = 0; j < VSX_TABLE_LEN; ++j) { htop[j] = (int16_t) h; h = sat16(h - geql); }   // :2043-2051
    int m = gotl + getl;                                        // M_QR_target_left
    for (int i = 0; i < VSX_TABLE_LEN; ++i) { hleft[i] = (int16_t) sat16(0 - m); m = sat16(m + getl); }  // :844-859
  }
  auto amb = [](unsigned x) { return !(x == 1 || x == 2 || x == 4 || x == 8); };
  for (unsigned x = 0; x < 16; ++x)                              // :1319-1342
    for (unsigned y = 0; y < 16; ++y)
      {
        int v;
        if (P.n_mismatch && (x == 15 || y == 15)) v = mism;
        else if (amb(x) || amb(y)) v = 0;
        else v = (x == y) ? match : mism;
        matrix[x * 16 + y] = (int16_t) v;
      }

  auto cleanup = [&]() {
    if (c->stream) (void) hipStreamDestroy(c->stream);
    if (c->stream2) (void) hipStreamDestroy(c->stream2);
    if (c->stream_b) (void) hipStreamDestroy(c->stream_b);
    if (c->stream_up) (void) hipStreamDestroy(c->stream_up);
    if (c->stream_dn) (void) hipStreamDestroy(c->stream_dn);
    delete c;
  };
  hipError_t e;
  // the aligner's streams get the HIGHEST priority the device offers: in a search its short plans share the GPU with the k-mer
  // counting kernel of the next window (lowest priority, vsx_kmer_host.cpp), whose millions of 15-us blocks would otherwise keep
  // every CU busy and hold each alignment stage back until the counting is over (r03 timeline: windows aligned 100 ms late)
  // Between them the TRACEBACK stream ranks above the DP streams where the device has a level to spare: the DP kernel of slice
  // i+2 waits for the traceback of slice i (two checkpoint blocks).  Same-box A/B of vsx_align_pairs (800 k pairs): 36.7-37.3 ms
  // against 37.5-39.0 with equal priorities.  It does not cure the starvation itself: a traceback wave needs 168 VGPRs, a retiring
  // DP wave frees 128, so while a DP launch still has workgroups to dispatch the traceback only gets whole-SIMD gaps (rocprofv3
  // traces of one call: profiles/r03/r03v_e2e_trace*.csv).
  int prio_low = 0, prio_high = 0;
  (void) hipDeviceGetStreamPriorityRange(&prio_low, &prio_high);            // numerically: high < low
  static const bool flat_env = std::getenv("VSX_ALIGN_FLAT_PRIORITY") != nullptr;      // A/B
  const int prio_dp = (!flat_env && prio_low - prio_high >= 2) ? prio_high + 1 : prio_high;
  // (r04 measured CU-masked streams -- a fixed share of the CUs for the traceback -- and rejected them: the DP kernel ran 2 x slower,
  //  profiles/r04/r04p_e2e_cumask_ab.txt; the VSX_TB_CUS switch is gone: masked streams also lose hipStreamNonBlocking and the priorities)
  if ((e = hipStreamCreateWithPriority(&c->stream, hipStreamNonBlocking, prio_dp)) != hipSuccess ||
      (e = hipStreamCreateWithPriority(&c->stream2, hipStreamNonBlocking, prio_high)) != hipSuccess ||
      (e = hipStreamCreateWithPriority(&c->stream_b, hipStreamNonBlocking, prio_dp)) != hipSuccess ||
      (e = hipStreamCreateWithPriority(&c->stream_up, hipStreamNonBlocking, prio_high)) != hipSuccess ||
      (e = hipStreamCreateWithPriority(&c->stream_dn, hipStreamNonBlocking, prio_high)) != hipSuccess ||
      (e = hipEventCreateWithFlags(&c->ev_tb[0], hipEventDisableTiming)) != hipSuccess ||
      (e = hipEventCreateWithFlags(&c->ev_tb[1], hipEventDisableTiming)) != hipSuccess ||
      (e = hipEventCreateWithFlags(&c->ev_tb[2], hipEventDisableTiming)) != hipSuccess ||
      (e = c->d_htop.alloc(VSX_TABLE_LEN)) != hipSuccess || (e = c->d_hleft.alloc(VSX_TABLE_LEN)) != hipSuccess ||
      (e = c->d_matrix.alloc(256)) != hipSuccess ||
      (e = hipMemcpy(c->d_htop.p, htop.data(), VSX_TABLE_LEN * 2, hipMemcpyHostToDevice)) != hipSuccess ||
      (e = hipMemcpy(c->d_hleft.p, hleft.data(), VSX_TABLE_LEN * 2, hipMemcpyHostToDevice)) != hipSuccess ||
      (e = hipMemcpy(c->d_matrix.p, matrix.data(), 512, hipMemcpyHostToDevice)) != hipSuccess)
    {
      cleanup();
      return fail(VSX_EHIP, "vsx_create: %s", hipGetErrorString(e));
    }
  P.htop = c->d_htop.p;
  P.hleft = c->d_hleft.p;
  P.matrix = c->d_matrix.p;
  P.tilt = 0;
  // once per device, check on the hardware what the MAX3 class assumes (v_pk_maximum3_f16 == integer maximum on
  // the patterns 0 .. 0x7BFF, denormals included); a device that fails keeps the 16-bit TILT class (tilt_possible())
  if (device < 64 && g_max3_state[device].load() == 0)
    {
      DevBuf<uint32_t> d_bad;
      uint32_t bad2[2] = {1, 1};
      if (d_bad.alloc(2) == hipSuccess && hipMemsetAsync(d_bad.p, 0, 8, c->stream) == hipSuccess &&
          vsx_launch_max3_selftest(d_bad.p, c->stream) == hipSuccess &&
          hipMemcpyAsync(bad2, d_bad.p, 8, hipMemcpyDeviceToHost, c->stream) == hipSuccess && hipStreamSynchronize(c->stream) == hipSuccess)
        {
          const uint32_t bad = bad2[0];
          if (bad) std::fprintf(stderr, "libvsx: device %d: v_pk_maximum3_f16 disagrees with the integer maximum on %u probes -- MAX3 class disabled\n", device, bad);
          g_max3_state[device].store(bad ? 2 : 1);
          // the second traceback kernel reads sign bits with v_perm_b32 selectors 8 .. 11: same one-off check, same kind of fallback
          if (bad2[1])
            {
              std::fprintf(stderr, "libvsx: device %d: v_perm_b32 sign selectors disagree on %u probes -- the first traceback kernel is used\n", device, bad2[1]);
              vsx_internal_set_tb_v2(device, 0);
            }
        }
      else
        {
          //  a self-test that could not RUN proves nothing: fail closed -- the 16-bit TILT class and the first traceback
          // kernel, which rest on no hardware assumption -- and say so
          (void) hipGetLastError();
          std::fprintf(stderr, "libvsx: device %d: the MAX3 / v_perm_b32 self-test could not run -- MAX3 class and second traceback kernel disabled\n", device);
          g_max3_state[device].store(2);
          vsx_internal_set_tb_v2(device, 0);
        }
    }
  else if (device >= 64) vsx_internal_set_tb_v2(device, 0);      // (no per-device state beyond 64 devices: the conservative kernels)

  // Tilted coordinates X* = X + (i + j) g, g = the interior extension (VsxDevParams::tilt): available when both interior
  // extensions are g > 0 and the interior QR coincide (the DP kernel's shared H - QR); per task the planner still has to
  // prove the shifted range (tilt_possible()).  VSX_TILT=0 switches the class off (A/B measurements, tests).
  // The tilted kernel adds the primed scores with a 32-bit add over both int16 halves: they must be non-negative
  // (mismatch + 2g >= 0, the dummy rows' -ge(query left) + 2g >= 0).
  if (geqi == geti && geti > 0 && P.share_sub && c->ckpt && !c->tb_packed && !c->force_fallback &&
      std::min(match, mism) + 2 * geti >= 0 && 2 * geti - geql >= 0 &&
      std::max(match, mism) + 2 * geti <= 255 && 2 * geti - geql <= 255 &&       // the LDS query profile of the class holds bytes
      // compressed checkpoints: H - F(next row) and H - E(next column) lie in [-g, max QR'] (plus the dummy rows' seed
      // go_ql + ge_ql + QR'): they must fit a signed byte
      geti <= 127 && std::max({goqi, goti, goqr + geqr - geti, gotr + getr - geti, goql + geql + goqi}) <= 127 &&
      !(std::getenv("VSX_TILT") && std::strcmp(std::getenv("VSX_TILT"), "0") == 0))
    {
      const int g = geti;
      VsxDevParams & T = c->Pt;
      T = P;
      T.tilt = g;
      T.qrq_i_pk = pk(goqi + geqi - g); T.rq_i_pk = pk(geqi - g);
      T.qrq_r_pk = pk(goqr + geqr - g); T.rq_r_pk = pk(geqr - g);
      T.qrt_i = goti + geti - g; T.rt_i = geti - g;
      T.qrt_r = gotr + getr - g; T.rt_r = getr - g;
      std::vector<int16_t> htop_t(VSX_TABLE_LEN), hleft_t(VSX_TABLE_LEN), matrix_t(256);
      for (int j = 0; j < VSX_TABLE_LEN; ++j)              // entries beyond the range tilt_possible() admits are never read
        {
          htop_t[j] = (int16_t) sat16((int) htop[j] + (j - 1) * g);
          hleft_t[j] = (int16_t) sat16((int) hleft[j] + (j - 1) * g);
        }
      for (int x = 0; x < 256; ++x) matrix_t[x] = (int16_t) sat16(matrix[x] + 2 * g);
      if ((e = c->d_htop_t.alloc(VSX_TABLE_LEN)) != hipSuccess || (e = c->d_hleft_t.alloc(VSX_TABLE_LEN)) != hipSuccess ||
          (e = c->d_matrix_t.alloc(256)) != hipSuccess ||
          (e = hipMemcpy(c->d_htop_t.p, htop_t.data(), VSX_TABLE_LEN * 2, hipMemcpyHostToDevice)) != hipSuccess ||
          (e = hipMemcpy(c->d_hleft_t.p, hleft_t.data(), VSX_TABLE_LEN * 2, hipMemcpyHostToDevice)) != hipSuccess ||
          (e = hipMemcpy(c->d_matrix_t.p, matrix_t.data(), 512, hipMemcpyHostToDevice)) != hipSuccess)
        {
          cleanup();
          return fail(VSX_EHIP, "vsx_create: %s", hipGetErrorString(e));
        }
      T.htop = c->d_htop_t.p; T.hleft = c->d_hleft_t.p; T.matrix = c->d_matrix_t.p;
    }
  { std::lock_guard<std::mutex> lk(g_ctx_mu); g_ctxs.push_back(c); }
  *out = c;
  return VSX_OK;
}

void vsx_destroy(vsx_ctx * c)
{
  if (!c) return;
  { std::lock_guard<std::mutex> lk(g_ctx_mu); g_ctxs.erase(std::remove(g_ctxs.begin(), g_ctxs.end(), c), g_ctxs.end()); }
  (void) hipSetDevice(c->device);
  if (c->stream) { (void) hipStreamSynchronize(c->stream); (void) hipStreamDestroy(c->stream); }
  if (c->stream2) { (void) hipStreamSynchronize(c->stream2); (void) hipStreamDestroy(c->stream2); }
  if (c->stream_b) { (void) hipStreamSynchronize(c->stream_b); (void) hipStreamDestroy(c->stream_b); }
  if (c->stream_up) { (void) hipStreamSynchronize(c->stream_up); (void) hipStreamDestroy(c->stream_up); }
  if (c->stream_dn) { (void) hipStreamSynchronize(c->stream_dn); (void) hipStreamDestroy(c->stream_dn); }
  c->shared_dir[0].reset();
  c->shared_dir[1].reset();
  c->shared_dir[2].reset();
  c->shared_slab.reset();
  c->pool.retire();
  delete c;
}

static int seqset_common(vsx_ctx * ctx, vsx_seqset ** out, uint64_t n, const void * blob, bool blob_on_device,
                         uint64_t blob_bytes, const uint64_t * offsets, const uint32_t * lengths, bool both_strands = false,
                         int case_bits = 0 /* 1: soft masking, 2: DUST */)
{
  if (!ctx || !out || (n && (!offsets || !lengths)) || (blob_bytes && !blob))
    return fail(VSX_EINVAL, "vsx_seqset_create: null argument");
  *out = nullptr;
  for (uint64_t i = 0; i < n; ++i)
    if (offsets[i] + lengths[i] > blob_bytes)
      return fail(VSX_EINVAL, "vsx_seqset_create: sequence %" PRIu64 " exceeds the blob", i);
  HIPCHK(hipSetDevice(ctx->device));
  auto * s = new vsx_seqset;
  s->ctx = ctx;
  s->device = ctx->device;
  s->n = n;
  s->bytes = blob_bytes;
  s->off.assign(offsets, offsets + n);
  s->len.assign(lengths, lengths + n);
  uint64_t code_bytes = blob_bytes;
  if (both_strands)
    {
      // the minus strands follow the plus strands' codes, packed in sequence order
      s->n = 2 * n;
      s->off.resize(2 * n);
      s->len.resize(2 * n);
      for (uint64_t k = 0; k < n; ++k) { s->off[n + k] = code_bytes; s->len[n + k] = lengths[k]; code_bytes += lengths[k]; }
      s->bytes = code_bytes;
    }
  hipError_t e = hipSuccess;
  ScratchPool * pl = (code_bytes <= VSX_SEQSET_POOLED_BYTES) ? &ctx->pool : nullptr;
  if (pl) s->pool_ref = ctx->pool_owner;
  PoolBuf<uint8_t> staging;
  const uint8_t * d_ascii = static_cast<const uint8_t *>(blob);
  do {
    if ((e = s->d_codes.alloc(pl, code_bytes + 2 * VSX_CODE_SLACK)) != hipSuccess) break;
    if ((e = s->d_off.alloc(pl, s->n)) != hipSuccess) break;
    if ((e = s->d_len.alloc(pl, s->n)) != hipSuccess) break;
    if (n)
      {
        if ((e = hipMemcpyAsync(s->d_off.p, s->off.data(), s->n * 8, hipMemcpyHostToDevice, ctx->stream)) != hipSuccess) break;
        if ((e = hipMemcpyAsync(s->d_len.p, s->len.data(), s->n * 4, hipMemcpyHostToDevice, ctx->stream)) != hipSuccess) break;
      }
    if (!blob_on_device && blob_bytes)
      {
        if ((e = staging.alloc(pl, blob_bytes + 16)) != hipSuccess) break;
        if ((e = hipMemcpyAsync(staging.p, blob, blob_bytes, hipMemcpyHostToDevice, ctx->stream)) != hipSuccess) break;
        d_ascii = staging.p;
      }
    if ((e = vsx_launch_encode(d_ascii, s->codes(), blob_bytes, ctx->stream)) != hipSuccess) break;
    if (case_bits)
      {
        const uint64_t nb = (blob_bytes + 7) / 8 + 16;             // the sweep reads a dword at any byte of the map
        if ((e = s->d_lower.alloc(pl, nb)) != hipSuccess) break;
        if ((e = hipMemsetAsync(s->d_lower.p, 0, nb, ctx->stream)) != hipSuccess) break;
        if ((e = vsx_kmer_launch_case_bits(d_ascii, blob_bytes, s->d_lower.p, case_bits == 2, ctx->stream)) != hipSuccess) break;
        if (case_bits == 2 && (e = vsx_launch_dust(s->codes(), s->d_off.p, s->d_len.p, n, s->d_lower.p, ctx->stream)) != hipSuccess) break;
        s->lower_bytes = (blob_bytes + 7) / 8;
      }
    if (both_strands && (e = vsx_launch_revcomp(s->codes(), s->d_off.p, s->d_off.p + n, s->d_len.p, n, ctx->stream)) != hipSuccess) break;
    e = hipStreamSynchronize(ctx->stream);
  } while (false);
  if (e != hipSuccess)
    {
      delete s;
      return fail(e == hipErrorOutOfMemory ? VSX_ENOMEM : VSX_EHIP, "vsx_seqset_create: %s", hipGetErrorString(e));
    }
  *out = s;
  return VSX_OK;
}

int vsx_seqset_create(vsx_ctx * ctx, vsx_seqset ** out, uint64_t n, const char * blob, uint64_t blob_bytes,
                      const uint64_t * offsets, const uint32_t * lengths)
{
  return seqset_common(ctx, out, n, blob, false, blob_bytes, offsets, lengths);
}

int vsx_seqset_create_both_strands(vsx_ctx * ctx, vsx_seqset ** out, uint64_t n, const char * blob, uint64_t blob_bytes,
                                   const uint64_t * offsets, const uint32_t * lengths)
{
  return seqset_common(ctx, out, n, blob, false, blob_bytes, offsets, lengths, true);
}

// internal (vsx_search.cpp, soft_mask): the set also keeps the case bitmap its k-mer index honours
// (mode 1: the bitmap is the input's case; mode 2: the input is upper-cased and DUST-masked on the device, vsx_mask.hip)
int vsx_internal_seqset_create_cased(vsx_ctx * ctx, vsx_seqset ** out, uint64_t n, const char * blob, uint64_t blob_bytes,
                                     const uint64_t * offsets, const uint32_t * lengths, int mode)
{
  if (mode != 1 && mode != 2) return fail(VSX_EINVAL, "vsx_internal_seqset_create_cased: mode must be 1 (soft) or 2 (dust)");
  return seqset_common(ctx, out, n, blob, false, blob_bytes, offsets, lengths, false, mode);
}
const uint8_t * vsx_internal_seqset_lower(const vsx_seqset * s) { return s ? s->d_lower.p : nullptr; }
// the bitmap on the host: bit i of byte i / 8 = blob byte i is masked
int vsx_internal_seqset_lower_download(const vsx_seqset * s, uint8_t * dst, uint64_t nbytes)
{
  if (!s || !s->d_lower.p || nbytes > s->lower_bytes) return fail(VSX_EINVAL, "vsx_internal_seqset_lower_download: no case bitmap of that size");
  HIPCHK(hipSetDevice(s->device));
  if (nbytes) HIPCHK(hipMemcpy(dst, s->d_lower.p, nbytes, hipMemcpyDeviceToHost));
  return VSX_OK;
}

// test hook (tests/test_gpu_mask.py): the case bitmap a mode-1 / mode-2 set ends up with, (blob_bytes + 7) / 8 bytes
int vsx_internal_mask_bits(vsx_ctx * ctx, uint64_t n, const char * blob, uint64_t blob_bytes, const uint64_t * offsets,
                           const uint32_t * lengths, int mode, uint8_t * bits_out)
{
  vsx_seqset * s = nullptr;
  int rc = vsx_internal_seqset_create_cased(ctx, &s, n, blob, blob_bytes, offsets, lengths, mode);
  if (rc != VSX_OK) return rc;
  rc = vsx_internal_seqset_lower_download(s, bits_out, (blob_bytes + 7) / 8);
  vsx_seqset_destroy(s);
  return rc;
}

int vsx_seqset_create_from_device(vsx_ctx * ctx, vsx_seqset ** out, uint64_t n, const void * d_blob,
                                  uint64_t blob_bytes, const uint64_t * offsets, const uint32_t * lengths)
{
  return seqset_common(ctx, out, n, d_blob, true, blob_bytes, offsets, lengths);
}

void vsx_seqset_destroy(vsx_seqset * s)
{
  if (!s) return;
  (void) hipSetDevice(s->device);       // the owning context may already be gone
  delete s;
}

uint64_t vsx_seqset_count(const vsx_seqset * s) { return s ? s->n : 0; }

static int ensure_impure(const vsx_seqset * s)
{
  std::lock_guard<std::mutex> lk(s->impure_mu);
  if (s->have_impure) return VSX_OK;
  vsx_ctx * ctx = s->ctx;
  HIPCHK(hipSetDevice(ctx->device));
  // (r06, ADVICE r05: the flags come from the scratch pool -- hipMalloc / hipFree synchronise the WHOLE device, and a search's plans are
  //  made while the counting kernel of the next window runs: 15-42 ms per window, profiles/r06/r06a_search_timeline.txt)
  PoolBuf<uint8_t> d_flags;
  HIPCHK(d_flags.alloc(s->pool_ref ? s->pool_ref.get() : nullptr, s->n));
  HIPCHK(vsx_launch_purity(s->codes(), s->d_off.p, s->d_len.p, s->n, d_flags.p, ctx->stream));
  s->impure.assign(s->n, 0);
  if (s->n) HIPCHK(hipMemcpyAsync(s->impure.data(), d_flags.p, s->n, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  s->have_impure = true;
  return VSX_OK;
}

// Can any H/E/F value of a (Q x D) alignment reach the 16-bit limits?  With all 12 penalties >= 0,
//   B = max(|match|, |mismatch|, every gap extension), G = max(every gap open):
//   every H, E, F, border and intermediate of onestep lies in [-(3G + (Q+Dp+4)B) - (G+B), min(Q,Dp)*B]
//   (lower bound: the pure-extension chains E(i,j) >= E(i,0) - jB, F(i,j) >= F(0,j) - iB from the borders
//   -(go + k*ge); upper bound: <= one match per diagonal step).  If that interval is inside
//   (SHRT_MIN + max(go+ge), SHRT_MAX), no saturation occurs and the reference's overflow rule
//   (align_simd.cpp:1432-1444, :1774-1786) can never fire, so the kernel may skip the min/max tracking.
static bool no_overflow_possible(const vsx_ctx * ctx, int64_t Q, int64_t D)
{
  int64_t B = std::max<int64_t>(std::llabs(ctx->P.match), std::llabs(ctx->P.mismatch));
  int64_t G = 0;
  for (int k = 0; k < 12; ++k)
    {
      if (ctx->pen[k] < 0) return false;
      if (k < 6) G = std::max<int64_t>(G, ctx->pen[k]); else B = std::max<int64_t>(B, ctx->pen[k]);
    }
  // TOPPAD dummy rows (above query row 0) run the interior recurrence; a horizontal gap opened inside them must never beat the
  // true top border H(-1, j) = -(go_ql + j ge_ql): after n columns it has lost QR_q(interior) + (n - 1) R_q(interior) against
  // the border's n ge_ql, so both QR_q(interior) >= ge_ql and R_q(interior) >= ge_ql are needed (found by oracle/soak.py: with
  // ge_ql = 5 > ge_qi = 4 a 77-column terminal gap was priced through the dummy rows)
  if (ctx->pen[2] + ctx->pen[8] < ctx->pen[6] || ctx->pen[8] < ctx->pen[6]) return false;
  const int64_t Dp = (D + 3) & ~3ll;
  return 4 * G + (Q + Dp + 16) * B < 32000;
}

// The TILT class (vsx_forward_kernel): every value is shifted by (i + j) g with -R - 2 <= i < Q, -1 <= j < Dp + 16 and g <= B,
// scores grow by 2g: the interval of no_overflow_possible() widened by that shift must still fit.
// Returns 0 (not admitted), 1 (TILT: values biased by 0x8000 into unsigned 16 bits) or 2 (the MAX3 sub-class: the same bound
// must hold for HALF the range -- values biased by 0x3E00 have to stay inside [0, 0x7BFF], the finite non-negative fp16 patterns,
// for v_pk_maximum3_f16 to be an integer maximum; vsx_forward_kernel MAX3).  VSX_MAX3=0 switches the sub-class off (A/B, tests).
static int tilt_possible(const vsx_ctx * ctx, int64_t Q, int64_t D)
{
  if (ctx->Pt.tilt == 0) return 0;
  static const bool max3_off = std::getenv("VSX_MAX3") && std::strcmp(std::getenv("VSX_MAX3"), "0") == 0;
  int64_t B = std::max<int64_t>(std::llabs(ctx->P.match), std::llabs(ctx->P.mismatch));
  int64_t G = 0;
  for (int k = 0; k < 12; ++k)
    if (k < 6) G = std::max<int64_t>(G, ctx->pen[k]); else B = std::max<int64_t>(B, ctx->pen[k]);
  const int64_t Dp = (D + 3) & ~3ll;
  const int64_t reach = 4 * G + 2 * (Q + Dp + 64) * B;          // |value| of anything the kernel forms stays below this
  // (pen[4] = the query's right-end gap open: the MAX3 kernel's interior steps read "the last row continues to the left" off ext-left
  //  alone, which covers "left" only when that penalty is positive -- vsx_forward_kernel LASTFAST; the reference's default is 1)
  if (reach < 15800 && !max3_off && !VSX_CKT && ctx->pen[4] > 0 &&
      ctx->device < 64 && g_max3_state[ctx->device].load(std::memory_order_relaxed) == 1) return 2;         // 0x3E00 = 15872 each way inside [0, 0x7BFF]
  return reach < 32000 ? 1 : 0;
}

static int pick_rows(int Q)
{
  static const int forced = std::getenv("VSX_ROWS") ? std::atoi(std::getenv("VSX_ROWS")) : 0;      // A/B experiments: rows per lane
  if (forced > 0 && Q >= forced) return forced;
  int cnt = 0;
  const int * rows = vsx_supported_rows(&cnt);
  int idx = cnt - 1;
  for (int k = 0; k < cnt; ++k)
    if (16 * rows[k] >= Q) { idx = k; break; }
  // a single, partially filled pipeline position would put the last query row at r != R-1
  while (idx > 0 && Q < rows[idx]) --idx;
  return rows[idx];
}

int vsx_plan_create(vsx_ctx * ctx, vsx_plan ** out, const vsx_seqset * queries, const vsx_seqset * targets,
                    uint64_t n_pairs, const uint32_t * qidx, const uint32_t * tidx, uint64_t dir_budget_bytes)
{
  if (!ctx || !out || !queries || !targets || (n_pairs && (!qidx || !tidx)))
    return fail(VSX_EINVAL, "vsx_plan_create: null argument");
  *out = nullptr;
  // a sequence set is device memory: any context of the same GPU may align against it (one Database mirror serves all the
  // worker threads' contexts, shim/vsx_search16_shim.cpp); it must outlive the plans that use it
  if (queries->device != ctx->device || targets->device != ctx->device)
    return fail(VSX_EINVAL, "vsx_plan_create: seqset lives on another device");
  if (n_pairs > 0xffffffffull / 8) return fail(VSX_EINVAL, "vsx_plan_create: too many pairs for one plan");
  static const bool timing = std::getenv("VSX_DEBUG_TIMING") != nullptr;
  auto now = []() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  const double tc0 = now();
  const int nthc = (int) std::max<uint64_t>(1, std::min<uint64_t>((uint64_t) vsx_internal_usable_cpus(), n_pairs / 32768));
  {
    std::vector<uint64_t> bad((size_t) nthc, UINT64_MAX);
    auto check = [&](int t) {
      const uint64_t lo = n_pairs * (uint64_t) t / (uint64_t) nthc, hi = n_pairs * (uint64_t) (t + 1) / (uint64_t) nthc;
      for (uint64_t k = lo; k < hi; ++k)
        if (qidx[k] >= queries->n || tidx[k] >= targets->n) { bad[(size_t) t] = k; return; }
    };
    run_threads(nthc, check);
    for (int t = 0; t < nthc; ++t)
      if (bad[(size_t) t] != UINT64_MAX)
        return fail(VSX_EINVAL, "vsx_plan_create: pair %" PRIu64 " references a sequence out of range", bad[(size_t) t]);
  }
  HIPCHK(hipSetDevice(ctx->device));
  static const bool arith = std::getenv("VSX_SCORE") && std::strcmp(std::getenv("VSX_SCORE"), "arith") == 0;
  if (arith) { const int rc = ensure_impure(queries); if (rc != VSX_OK) return rc; }
  // r05, pair-profile classes (vsx_forward_kernel PAIR): groups of four whole-wave tasks of ONE pure-ACGT query with pure-ACGT targets
  // run as a workgroup that shares a pair-indexed dword profile (5 instead of 6 instructions per lane-row).  Worth it only for queries
  // with many targets (--allpairs_global); needs the purity of both sets.  VSX_PAIRPROF=0 / 1 overrides the default below
  // (r06, ADVICE r05: the purity of the two sets is asked for only once the grouping below has found a query with four whole-wave tasks --
  //  a search's windows never have one, and every new query set used to pay a purity pass with a device-wide synchronisation; only the
  //  environment and build parts of the switches are process-wide, ctx->ckpt is this context's)
  static const bool pair_env = (std::getenv("VSX_PAIRPROF") ? std::strcmp(std::getenv("VSX_PAIRPROF"), "0") != 0 : (VSX_PAIRPROF_DEFAULT != 0)) &&
                               !arith && VSX_QPL != 0 && VSX_FEED2 != 0 && !VSX_CKT;
  const bool pair_try = pair_env && ctx->ckpt && n_pairs >= 32;

  std::unique_ptr<vsx_plan> pl(new vsx_plan);
  pl->ctx = ctx; pl->Q = queries; pl->T = targets; pl->n_pairs = n_pairs;
  static const unsigned ck_blocks = (std::getenv("VSX_CK_BLOCKS") && std::atoi(std::getenv("VSX_CK_BLOCKS")) == 2) ? 2u : 3u;
  // r06: a block nobody uses, the LARGEST such -- plans that follow one another (the stages of a clustering round, of a search window:
  // each is destroyed, i.e. its traceback has finished, before the next is made) then keep ONE block warm instead of growing three in
  // turn: 2 M x 300 bp --cluster_fast spent 1.1 s of its 6-7 s in eight multi-GB hipMallocs (profiles/r06/r06e_cluster_blocks.txt).
  // A pipeline of slices (vsx_align_pairs) finds the earlier plans' blocks taken and so still alternates; when all are taken, the one
  // acquired longest ago (its traceback finishes first).  VSX_CK_ROTATE=1: the blind rotation of r04 / r05 (A/B).
  {
    static const bool rotate_env = std::getenv("VSX_CK_ROTATE") && std::strcmp(std::getenv("VSX_CK_ROTATE"), "1") == 0;
    std::lock_guard<std::mutex> lk(ctx->slot_mu);
    const unsigned seq = ctx->plan_seq.fetch_add(1);
    int pick = -1;
    if (rotate_env) pick = (int) (seq % ck_blocks);
    else
      {
        for (int k = 0; k < (int) ck_blocks; ++k)
          if (ctx->slot_refs[k] == 0 && (pick < 0 || ctx->shared_dir[k].bytes() > ctx->shared_dir[pick].bytes())) pick = k;
        if (pick < 0)
          for (int k = 0; k < (int) ck_blocks; ++k)
            if (pick < 0 || (int) (seq - ctx->slot_stamp[k]) > (int) (seq - ctx->slot_stamp[pick])) pick = k;
      }
    ++ctx->slot_refs[pick];
    ctx->slot_stamp[pick] = seq;
    pl->dir_slot = pick;
    pl->slot_held = true;
  }

  // ---- the reference's closed-form / sentinel cases (no DP) ----
  // host threads classify contiguous slices: the common case (a pair for the GPU) is handled in place, the rare closed-form
  // pairs are collected per slice and answered serially below (they append to shared CIGAR lists, in pair order)
  std::vector<uint32_t> gpu_pairs;
  {
    std::vector<std::vector<uint32_t>> gp((size_t) nthc), special((size_t) nthc);
    std::vector<uint64_t> pcells((size_t) nthc, 0);
    auto classify = [&](int t) {
      const uint64_t lo = n_pairs * (uint64_t) t / (uint64_t) nthc, hi = n_pairs * (uint64_t) (t + 1) / (uint64_t) nthc;
      gp[(size_t) t].reserve(hi - lo);
      uint64_t cells = 0;
      for (uint64_t k = lo; k < hi; ++k)
        {
          const int64_t Q = queries->len[qidx[k]], D = targets->len[tidx[k]];
          if (ctx->force_fallback || Q == 0 || D == 0 || !fits(Q, D)) { special[(size_t) t].push_back((uint32_t) k); continue; }
          gp[(size_t) t].push_back((uint32_t) k);
          cells += (uint64_t) Q * (uint64_t) D;
        }
      pcells[(size_t) t] = cells;
    };
    run_threads(nthc, classify);
    size_t total = 0;
    for (auto & v : gp) total += v.size();
    gpu_pairs.resize(total);
    {
      std::vector<size_t> base((size_t) nthc);
      size_t acc = 0;
      for (int t = 0; t < nthc; ++t) { base[(size_t) t] = acc; acc += gp[(size_t) t].size(); pl->cells += pcells[(size_t) t]; }
      auto place = [&](int t) { if (!gp[(size_t) t].empty()) std::memcpy(gpu_pairs.data() + base[(size_t) t], gp[(size_t) t].data(), gp[(size_t) t].size() * 4); };
      run_threads(nthc, place);
    }
    for (int t = 0; t < nthc; ++t)
      for (uint32_t k : special[(size_t) t])
        {
          const int64_t Q = queries->len[qidx[k]], D = targets->len[tidx[k]];
          pl->host_pairs.push_back(k);
          pl->host_out.emplace_back();
          VsxPairOut & o = pl->host_out.back();
          auto sentinel = [&]() { o = VsxPairOut {}; o.score = 32767; };
          if (ctx->force_fallback) { sentinel(); continue; }              // align_simd.cpp:1463-1479
          if (Q == 0)                                                      // :1481-1539
            {
              if (!fits(0, D)) { sentinel(); continue; }
              o = VsxPairOut {};
              o.aligned = (uint16_t) D; o.gaps = (uint16_t) D;
              if (D > 0)
                {
                  const int64_t a = -(int64_t) ctx->pen[1] - D * (int64_t) ctx->pen[7];
                  const int64_t b = -(int64_t) ctx->pen[5] - D * (int64_t) ctx->pen[11];
                  o.score = (int16_t) (uint16_t) (std::max(a, b) & 0xffff);   // plain narrowing cast :1515
                  pl->host_cigar.push_back(std::to_string(D) + "I");
                  pl->host_cigar_pair.push_back(k);
                  pl->host_cigar_run.push_back(((uint32_t) D << 2) | 1u);
                }
              continue;
            }
          sentinel();                                                      // D == 0 or the size guard (:1867-1882)
        }
  }
  const double tc1 = now();

  // ---- group by query -> tasks of <= 8 targets, similar lengths together (host threads over query groups) ----
  auto by_query = [&](uint32_t a, uint32_t b) { return qidx[a] < qidx[b]; };
  if (!std::is_sorted(gpu_pairs.begin(), gpu_pairs.end(), by_query)) std::stable_sort(gpu_pairs.begin(), gpu_pairs.end(), by_query);
  struct ProtoTask { uint32_t q; int rows; int generic; int track; int tilt; int nq; uint32_t n; uint32_t pair[8]; };
  // r05, sparse tasks: a task of <= 2 (<= 4) targets of the TILT family shares its wave with three (one) other tasks of its class
  // (vsx_forward_kernel NQ) instead of leaving three (two) of the four lane groups idle.  VSX_SPARSE=0: every task is a wave (A/B, tests)
  static const bool sparse_env = !(std::getenv("VSX_SPARSE") && std::strcmp(std::getenv("VSX_SPARSE"), "0") == 0) && VSX_QPL != 0 && !VSX_CKT;
  const bool sparse_on = sparse_env && ctx->ckpt;
  std::vector<size_t> group_begin;                       // start of every query's run in gpu_pairs, plus the end
  for (size_t b = 0; b < gpu_pairs.size();)
    {
      group_begin.push_back(b);
      const uint32_t q = qidx[gpu_pairs[b]];
      size_t e = b;
      while (e < gpu_pairs.size() && qidx[gpu_pairs[e]] == q) ++e;
      b = e;
    }
  group_begin.push_back(gpu_pairs.size());
  // substitution scores: LDS query profile by default (handles every symbol); VSX_SCORE=arith selects the XOR/min/mad
  // variant for queries made of A/C/G/T(U) only (kept for A/B measurements)
  const size_t ngroups = group_begin.size() - 1;
  const int nth = (int) std::max<size_t>(1, std::min<size_t>((size_t) vsx_internal_usable_cpus(), gpu_pairs.size() / 16384));
  std::vector<std::vector<ProtoTask>> part((size_t) nth);
  std::vector<std::vector<std::pair<size_t, size_t>>> pair_cand((size_t) nth);      // per slice: [first, end) task ranges of queries that may form PAIR groups
  auto work = [&](int t) {
    // contiguous ranges of groups with about the same number of pairs
    const size_t p_lo = gpu_pairs.size() * (size_t) t / (size_t) nth, p_hi = gpu_pairs.size() * (size_t) (t + 1) / (size_t) nth;
    const size_t g_lo = (size_t) (std::lower_bound(group_begin.begin(), group_begin.end() - 1, p_lo) - group_begin.begin());
    const size_t g_hi = (t == nth - 1) ? ngroups : (size_t) (std::lower_bound(group_begin.begin(), group_begin.end() - 1, p_hi) - group_begin.begin());
    std::vector<ProtoTask> & outp = part[(size_t) t];
    for (size_t g = g_lo; g < g_hi; ++g)
      {
        const size_t b = group_begin[g], e = group_begin[g + 1];
        const uint32_t q = qidx[gpu_pairs[b]];
        std::stable_sort(gpu_pairs.begin() + (long) b, gpu_pairs.begin() + (long) e,
                         [&](uint32_t x, uint32_t y) { return targets->len[tidx[x]] > targets->len[tidx[y]]; });
        const int rows = pick_rows((int) queries->len[q]);
        const int generic = (!arith || queries->impure[q]) ? 1 : 0;
        const size_t first_of_query = outp.size();
        for (size_t x = b; x < e; x += VSX_TASK_SLOTS)
          {
            ProtoTask pt {};
            pt.q = q; pt.rows = rows; pt.generic = generic;
            pt.n = (uint32_t) std::min<size_t>(VSX_TASK_SLOTS, e - x);
            uint32_t dmax = 0;
            for (uint32_t sidx = 0; sidx < pt.n; ++sidx)
              {
                pt.pair[sidx] = gpu_pairs[x + sidx];
                dmax = std::max(dmax, targets->len[tidx[pt.pair[sidx]]]);
              }
            pt.track = (!ctx->tb_packed && no_overflow_possible(ctx, queries->len[q], dmax)) ? 0 : 1;
            pt.tilt = (pt.track == 0 && generic && ctx->ckpt) ? tilt_possible(ctx, queries->len[q], dmax) : 0;
            pt.nq = 1;
            if (sparse_on && pt.tilt && rows >= 4 && pt.n <= 4 && (int64_t) queries->len[q] <= 16ll * rows)      // (one strip: the hand-over rows are per wave)
              pt.nq = pt.n <= 2 ? 4 : 2;
            outp.push_back(pt);
          }
        if (pair_try && rows >= 4 && outp.size() - first_of_query >= 4 && (int64_t) queries->len[q] <= 16ll * rows)     // (one strip: the PAIR kernels are ONE variants)
          {
            size_t eligible = 0;
            for (size_t k = first_of_query; k < outp.size(); ++k) eligible += (outp[k].nq == 1 && outp[k].tilt) ? 1 : 0;
            if (eligible >= 4) pair_cand[(size_t) t].emplace_back(first_of_query, outp.size());
          }
      }
  };
  run_threads(nth, work);
  bool any_pair_cand = false;
  for (const auto & v : pair_cand) any_pair_cand = any_pair_cand || !v.empty();
  if (any_pair_cand)
    {
      int rc = ensure_impure(queries);
      if (rc == VSX_OK) rc = ensure_impure(targets);
      if (rc != VSX_OK) return rc;
    }
  auto mark_pairs = [&](int t) {
    std::vector<ProtoTask> & outp = part[(size_t) t];
    for (const std::pair<size_t, size_t> & range : pair_cand[(size_t) t])
      {
        const size_t first_of_query = range.first, end_of_query = range.second;
        if (queries->impure[outp[first_of_query].q]) continue;
          {
            // whole groups of four eligible tasks of this query (same kernel class, every target plain ACGT) -> nq = 8; what is left over
            // stays in the whole-wave class.  The tasks of a query are consecutive and stay so through the stable class sort below.
            std::vector<size_t> ok;
            for (size_t k = first_of_query; k < end_of_query; ++k)
              {
                ProtoTask & pt = outp[k];
                if (pt.nq != 1 || !pt.tilt) continue;
                bool pure = true;
                for (uint32_t sidx = 0; sidx < pt.n && pure; ++sidx) pure = !targets->impure[tidx[pt.pair[sidx]]];
                if (pure) ok.push_back(k);
              }
            // (groups must not mix the MAX3 and the 16-bit TILT sub-class: take runs of equal tilt)
            size_t i0 = 0;
            while (i0 < ok.size())
              {
                size_t i1 = i0;
                while (i1 < ok.size() && outp[ok[i1]].tilt == outp[ok[i0]].tilt) ++i1;
                const size_t whole = (i1 - i0) / 4 * 4;
                for (size_t k = i0; k < i0 + whole; ++k) outp[ok[k]].nq = 8;
                i0 = i1;
              }
          }
      }
  };
  if (any_pair_cand) run_threads(nth, mark_pairs);
  std::vector<ProtoTask> protos;
  {
    size_t total = 0;
    for (auto & v : part) total += v.size();
    protos.reserve(total);
    for (auto & v : part) protos.insert(protos.end(), v.begin(), v.end());
  }
  // r04: a class with a handful of tasks still costs a launch of each kernel on the plan's streams, and a one-wave launch is a pure
  // latency chain (profiles/r04/r04z_shape_300x300_summary.txt: four tasks with 18 rows per lane beside 99 996 with 20 -- queries a few
  // symbols short of 289 -- took 0.15 ms of DP and 0.61 ms of traceback, 11 % of the step's traceback time).  Any row count with
  // 16 R >= Q is valid for a query, so the tasks of a sparse class join the next denser class of the same kind.
  {
    static const bool no_promote = std::getenv("VSX_NO_PROMOTE") != nullptr;      // A/B, tests
    struct Cls { int rows, generic, track, tilt, nq; size_t count; };
    std::vector<Cls> cls;
    for (const ProtoTask & pt : protos)
      {
        bool found = false;
        for (Cls & c : cls) if (c.rows == pt.rows && c.generic == pt.generic && c.track == pt.track && c.tilt == pt.tilt && c.nq == pt.nq) { ++c.count; found = true; break; }
        if (!found) cls.push_back(Cls {pt.rows, pt.generic, pt.track, pt.tilt, pt.nq, 1});
      }
    std::vector<std::pair<size_t, int>> move;                                 // class index -> new rows
    for (size_t a = 0; a < cls.size() && !no_promote; ++a)
      {
        int best = 0;
        for (const Cls & c : cls)
          if (c.generic == cls[a].generic && c.track == cls[a].track && c.tilt == cls[a].tilt && c.nq == cls[a].nq && c.rows > cls[a].rows &&
              c.count >= 32 * cls[a].count && (best == 0 || c.rows < best))
            best = c.rows;
        if (best && cls[a].count < 4096) move.emplace_back(a, best);
      }
    for (auto & mv : move)                                                    // (a target class that moves itself takes its newcomers along)
      for (int hop = 0; hop < 4; ++hop)
        for (const auto & other : move)
          {
            const Cls & a = cls[mv.first], & o = cls[other.first];
            if (o.rows == mv.second && o.generic == a.generic && o.track == a.track && o.tilt == a.tilt && o.nq == a.nq) { mv.second = other.second; break; }
          }
    if (!move.empty())
      for (ProtoTask & pt : protos)
        for (const auto & mv : move)
          {
            const Cls & c = cls[mv.first];
            if (pt.rows == c.rows && pt.generic == c.generic && pt.track == c.track && pt.tilt == c.tilt && pt.nq == c.nq)
              {
                // (only queries that still span two pipeline positions: a query shorter than one position of the new class would make
                //  position 0 the last position as well, a shape no row count chosen by pick_rows() ever produces)
                if ((int) queries->len[pt.q] > mv.second) pt.rows = mv.second;
                break;
              }
          }
  }
  // a SMALL sparse class is not worth its own launches: below one round of the chip (256 CUs x <= 16 waves) whole-wave tasks already
  // have the SIMDs largely to themselves -- packing four of them into one slower wave frees nothing -- and every extra class is one more
  // DP launch in the plan's stream (r05c: the speculative alignments of --cluster_fast, ~30 k pairs per round in three classes, ran
  // 0.44 -> 0.64 s with sparse classes of a few thousand tasks each).  VSX_SPARSE_MIN=n: the smallest class kept (tests: 1)
  if (sparse_on)
    {
      static const size_t sparse_min = std::getenv("VSX_SPARSE_MIN") ? (size_t) std::atoll(std::getenv("VSX_SPARSE_MIN")) : 4096;
      struct Key { int rows, generic, track, tilt, nq; size_t count; };
      std::vector<Key> ks;
      for (const ProtoTask & pt : protos)
        {
          if (pt.nq != 2 && pt.nq != 4) continue;
          bool found = false;
          for (Key & c : ks) if (c.rows == pt.rows && c.generic == pt.generic && c.track == pt.track && c.tilt == pt.tilt && c.nq == pt.nq) { ++c.count; found = true; break; }
          if (!found) ks.push_back(Key {pt.rows, pt.generic, pt.track, pt.tilt, pt.nq, 1});
        }
      bool any = false;
      for (const Key & c : ks) any = any || c.count < sparse_min;
      if (any)
        for (ProtoTask & pt : protos)
          if (pt.nq == 2 || pt.nq == 4)
            for (const Key & c : ks)
              if (c.count < sparse_min && pt.nq == c.nq && pt.rows == c.rows && pt.generic == c.generic && pt.track == c.track && pt.tilt == c.tilt) { pt.nq = 1; break; }
    }
  // kernel classes together (one launch per class and chunk)
  auto by_class = [](const ProtoTask & a, const ProtoTask & b) {
    if (a.rows != b.rows) return a.rows < b.rows;
    if (a.generic != b.generic) return a.generic < b.generic;
    if (a.track != b.track) return a.track < b.track;
    if (a.tilt != b.tilt) return a.tilt < b.tilt;
    return a.nq < b.nq;
  };
  if (!std::is_sorted(protos.begin(), protos.end(), by_class)) std::stable_sort(protos.begin(), protos.end(), by_class);

  // ---- direction-buffer budget ----
  double t_budget = 0;
  if (dir_budget_bytes == 0)
    {
      size_t free_b = 0, total_b = 0;
      const double tb0 = now();
      HIPCHK(hipMemGetInfo(&free_b, &total_b));
      t_budget = now() - tb0;
      free_b += ctx->pool.idle_bytes() + ctx->shared_dir[pl->dir_slot].bytes();   // blocks this context can hand straight back / already holds
      dir_budget_bytes = std::min<uint64_t>((uint64_t) (free_b * 0.4), 128ull << 30);
      // stay inside the context's current checkpoint block unless it is less than half of what could be had: a slightly
      // larger request would cost another multi-second hipMalloc
      const uint64_t have = ctx->shared_dir[pl->dir_slot].bytes();
      if (have >= dir_budget_bytes / 2) dir_budget_bytes = have;
    }
  const uint64_t budget_dwords = std::max<uint64_t>(dir_budget_bytes / 4, 1);

  const double tc2 = now();
  // tasks: (1) host threads fill the descriptors and sizes, (2) a serial scan assigns checkpoint / strip / slab offsets and cuts
  // chunks and launches, (3) host threads write the per-pair tables.  (1) and (3) are bound by random reads of the sequence
  // tables; the scan touches 3 numbers per task.
  const size_t NT = protos.size();
  pl->tasks.resize(NT);
  pl->pair_slot.resize(gpu_pairs.size());
  pl->pair_ids.resize(gpu_pairs.size());
  pl->slab_off.resize(gpu_pairs.size());
  std::vector<uint64_t> t_dwords(NT), t_slab(NT), t_pair0(NT);
  const int ntt = (int) std::max<size_t>(1, std::min<size_t>((size_t) vsx_internal_usable_cpus(), NT / 4096));
  run_threads(ntt, [&](int th) {
    const size_t lo = NT * (size_t) th / (size_t) ntt, hi = NT * (size_t) (th + 1) / (size_t) ntt;
    for (size_t x = lo; x < hi; ++x)
      {
        const ProtoTask & pt = protos[x];
        VsxTask t {};
        t.qoff = queries->off[pt.q];
        t.qlen = queries->len[pt.q];
        t.rows = (uint32_t) pt.rows;
        uint32_t dmax = 0;
        uint64_t slab = 0;
        for (uint32_t sl = 0; sl < pt.n; ++sl)
          {
            const uint32_t ti = tidx[pt.pair[sl]];
            t.toff[sl] = targets->off[ti];
            t.tlen[sl] = targets->len[ti];
            dmax = std::max(dmax, (targets->len[ti] + 3u) & ~3u);
            slab += (uint64_t) t.qlen + t.tlen[sl] + 1;
          }
        t.steps = dmax + 16;                                // pipeline drain (15) rounded to an even step count: the DP
                                                            // kernel stores row checkpoints two steps at a time
        if (pt.tilt && VSX_CKT) t.steps = (t.steps + 7u) & ~7u;   // ... and in blocks of 8 steps in the transposed layout
        const uint64_t total_lanes = (t.qlen + pt.rows - 1) / pt.rows;
        const uint64_t nstrips = (total_lanes + 15) / 16;
        const uint64_t nd = (uint64_t) (pt.rows + 3) / 4;
        t_dwords[x] = ctx->ckpt ? vsx_ckpt_dwords(nstrips, t.steps, (uint64_t) pt.rows, pt.tilt)
                                : ((nstrips * t.steps + 3) & ~3ull) * 64 * nd;     // [4-step block][lane][4][nd]
        t.strip_off = nstrips > 1 ? 2ull * 4 * t.steps : 0;                          // size for now, offset after the scan
        t_slab[x] = slab;
        pl->tasks[x] = t;
      }
  });
  size_t np_out = 0;
  Chunk cur;
  auto close_chunk = [&]() {
    if (cur.task_count) pl->chunks.push_back(cur);
    Chunk nc;
    nc.task_first = cur.task_first + cur.task_count;
    nc.pair_first = cur.pair_first + cur.pair_count;
    cur = nc;
  };
  // sparse-task classes: the nq tasks of a wave -- consecutive tasks of the launch, counted from its first -- share ONE checkpoint block
  // and one step range (the longest of them); `wave_left` tasks of the open wave are still to come, they take `wave_off` / `wave_steps`
  uint32_t wave_left = 0, wave_steps = 0, wave_group = 0, pair_left = 0;
  uint64_t wave_off = 0;
  auto same_class = [&](size_t a, size_t b) {
    return protos[a].rows == protos[b].rows && protos[a].generic == protos[b].generic && protos[a].track == protos[b].track &&
           protos[a].tilt == protos[b].tilt && protos[a].nq == protos[b].nq;
  };
  for (size_t x = 0; x < NT; ++x)
    {
      const ProtoTask & pt = protos[x];
      VsxTask & t = pl->tasks[x];
      uint64_t dwords = t_dwords[x];
      const uint64_t strip = t.strip_off;
      const bool sparse_task = pt.nq == 2 || pt.nq == 4;
      if (pt.nq == 8)
        {
          // a pair-profile group of four (consecutive tasks of one query): never cut a chunk inside it
          if (pair_left == 0)
            {
              pair_left = 4;
              // (the WHOLE group must fit: a chunk that overshoots the budget -- which is the size of the context's current block once one
              //  exists -- by three tasks asks for a slightly larger block while the old one is still held by the plans in flight:
              //  out of memory in the middle of an allpairs run, profiles/r05/r05j_pairprof_first_contact.txt)
              uint64_t gsum = 0;
              for (size_t y = x; y < std::min(NT, x + 4); ++y) gsum += t_dwords[y];
              if (cur.task_count && cur.dir_dwords + gsum + VSX_CK_SLACK_DW > budget_dwords) close_chunk();
            }
          --pair_left;
        }
      else if (sparse_task && wave_left == 0)
        {
          // a new wave: its tasks are x .. x + nq - 1 as far as the class reaches (a chunk is never cut inside a wave)
          uint32_t smax = t.steps, members = 1;
          while (members < (uint32_t) pt.nq && x + members < NT && same_class(x, x + members)) { smax = std::max(smax, pl->tasks[x + members].steps); ++members; }
          wave_left = members; wave_steps = smax; wave_group = 0;
          dwords = vsx_ckpt_dwords(1, smax, (uint64_t) pt.rows, pt.tilt);
          if (cur.task_count && cur.dir_dwords + dwords + VSX_CK_SLACK_DW > budget_dwords) close_chunk();
          wave_off = cur.dir_dwords;
        }
      else if (sparse_task) dwords = 0;                   // (the wave's block is already counted)
      else if (cur.task_count && cur.dir_dwords + dwords + VSX_CK_SLACK_DW > budget_dwords) close_chunk();
      if (sparse_task)
        {
          t.dir_off = wave_off;
          t.steps = wave_steps;
          t.group0 = wave_group;
          wave_group += 4u / (uint32_t) pt.nq;
          --wave_left;
        }
      else t.dir_off = cur.dir_dwords;
      t.strip_off = cur.strip_elems;
      cur.dir_dwords += dwords;
      cur.strip_elems += strip;
      pl->dir_bytes_total += dwords * 4;
      if (cur.launches.empty() || cur.launches.back().rows != pt.rows || cur.launches.back().generic != pt.generic ||
          cur.launches.back().track != pt.track || cur.launches.back().tilt != pt.tilt || cur.launches.back().nq != pt.nq)
        cur.launches.push_back(Launch {pt.rows, pt.generic, pt.track, pt.tilt, pt.nq, (uint32_t) x, 0, cur.pair_first + cur.pair_count, 0});
      cur.launches.back().count++;
      if (strip != 0) cur.launches.back().multi = true;           // (a task owns hand-over rows only when it has a second strip)
      cur.launches.back().pair_count += pt.n;
      t_pair0[x] = np_out;
      np_out += pt.n;
      const uint64_t slab = t_slab[x];
      t_slab[x] = cur.slab_words;                         // from here on: the task's first slab word (chunk-relative)
      cur.slab_words += slab;
      cur.pair_count += pt.n;
      cur.task_count++;
    }
  run_threads(ntt, [&](int th) {
    const size_t lo = NT * (size_t) th / (size_t) ntt, hi = NT * (size_t) (th + 1) / (size_t) ntt;
    for (size_t x = lo; x < hi; ++x)
      {
        const ProtoTask & pt = protos[x];
        const VsxTask & t = pl->tasks[x];
        uint64_t slab = t_slab[x];
        for (uint32_t sl = 0; sl < pt.n; ++sl)
          {
            const size_t o = (size_t) t_pair0[x] + sl;
            pl->pair_slot[o] = (uint32_t) x * VSX_TASK_SLOTS + sl;
            pl->pair_ids[o] = pt.pair[sl];
            pl->slab_off[o] = slab;
            slab += (uint64_t) t.qlen + t.tlen[sl] + 1;
          }
      }
  });
  close_chunk();
  const double tc3 = now();

  // ---- device buffers ----
  uint64_t max_dir = 1, max_strip = 1, max_slab = 1, worst_runs = 0;
  for (const Chunk & c : pl->chunks)
    {
      max_dir = std::max(max_dir, c.dir_dwords + VSX_CK_SLACK_DW);
      max_strip = std::max(max_strip, c.strip_elems);
      max_slab = std::max(max_slab, c.slab_words);
      worst_runs += c.slab_words;
    }
  const size_t ngp = pl->pair_ids.size();
  pl->runs_capacity = std::min<uint64_t>(worst_runs, std::max<uint64_t>(16ull << 20, 48ull * ngp)) + 1;
  HIPCHK(pl->d_tasks.alloc(&ctx->pool, pl->tasks.size()));
  HIPCHK(pl->d_pair_slot.alloc(&ctx->pool, ngp));
  HIPCHK(pl->d_pair_ids.alloc(&ctx->pool, ngp));
  HIPCHK(pl->d_slab_off.alloc(&ctx->pool, ngp));
  HIPCHK(pl->d_slot.alloc(&ctx->pool, pl->tasks.size() * VSX_TASK_SLOTS));
  HIPCHK(pl->d_out.alloc(&ctx->pool, n_pairs));
  HIPCHK(pl->d_cursor.alloc(&ctx->pool, 2));
  {
    // output arrays, 16-byte aligned sections of one block: score, aligned, matches, mismatches, gaps (2 B), verdict (1 B), text offset (8 B)
    const uint64_t np = std::max<uint64_t>(n_pairs, 1);
    const uint64_t width[7] = {2, 2, 2, 2, 2, 1, 8};
    uint64_t at = 0;
    for (int x = 0; x < 7; ++x) { pl->soa_off[x] = at; at += (np * width[x] + 15) & ~15ull; }
    pl->soa_bytes = at;
    HIPCHK(pl->d_soa.alloc(&ctx->pool, at));
    uint8_t * b = pl->d_soa.p;
    pl->soa.score = reinterpret_cast<int16_t *>(b + pl->soa_off[0]);
    pl->soa.aligned = reinterpret_cast<uint16_t *>(b + pl->soa_off[1]);
    pl->soa.matches = reinterpret_cast<uint16_t *>(b + pl->soa_off[2]);
    pl->soa.mismatches = reinterpret_cast<uint16_t *>(b + pl->soa_off[3]);
    pl->soa.gaps = reinterpret_cast<uint16_t *>(b + pl->soa_off[4]);
    pl->soa.verdict = b + pl->soa_off[5];
    pl->soa.text_off = reinterpret_cast<uint64_t *>(b + pl->soa_off[6]);
  }
  {
    std::lock_guard<std::mutex> lk(ctx->stage_mu);
    if (!ctx->cursor_slots.empty()) { pl->h_cursor = ctx->cursor_slots.back(); ctx->cursor_slots.pop_back(); }
  }
  if (!pl->h_cursor) HIPCHK(hipHostMalloc(reinterpret_cast<void **>(&pl->h_cursor), 64, hipHostMallocDefault));
  // one checkpoint buffer, reused chunk after chunk: overlapping chunk k's traceback with chunk k+1's DP bought nothing
  // (both are issue-bound), while a second buffer doubled a multi-second hipMalloc
  for (int k = 0; k < 2; ++k)
    {
      const uint64_t need = (k == 0 ? max_dir : max_slab) * 4;
      uint64_t seen = ctx->req_hwm[k].load();
      while (need > seen && !ctx->req_hwm[k].compare_exchange_weak(seen, need)) { }
    }
  HIPCHK(pl->d_dir[0].alloc(ctx->shared_dir[pl->dir_slot], &ctx->pool, max_dir));
  HIPCHK(pl->d_strip.alloc(&ctx->pool, max_strip));
  HIPCHK(pl->d_slab.alloc(ctx->shared_slab, &ctx->pool, max_slab));
  HIPCHK(pl->d_runs.alloc(&ctx->pool, pl->runs_capacity));
  // text: <= 6 bytes per run (5 digits + op) + NUL + padding per pair in the worst case; typical alignments need ~2 per run
  pl->text_capacity = std::min<uint64_t>(6 * worst_runs + 8 * (uint64_t) ngp, std::max<uint64_t>(32ull << 20, 128ull * ngp)) + 16;
  HIPCHK(pl->d_text.alloc(&ctx->pool, pl->text_capacity));
  if (!pl->tasks.empty())
    HIPCHK(hipMemcpyAsync(pl->d_tasks.p, pl->tasks.data(), pl->tasks.size() * sizeof(VsxTask), hipMemcpyHostToDevice, ctx->stream_up));
  if (ngp)
    {
      HIPCHK(hipMemcpyAsync(pl->d_pair_slot.p, pl->pair_slot.data(), ngp * 4, hipMemcpyHostToDevice, ctx->stream_up));
      HIPCHK(hipMemcpyAsync(pl->d_pair_ids.p, pl->pair_ids.data(), ngp * 4, hipMemcpyHostToDevice, ctx->stream_up));
      HIPCHK(hipMemcpyAsync(pl->d_slab_off.p, pl->slab_off.data(), ngp * 8, hipMemcpyHostToDevice, ctx->stream_up));
    }
  HIPCHK(hipEventCreate(&pl->ev_begin));
  HIPCHK(hipEventCreate(&pl->ev_end));
  for (Chunk & c : pl->chunks)
    {
      HIPCHK(hipEventCreate(&c.e0));
      HIPCHK(hipEventCreate(&c.e1));
      HIPCHK(hipEventCreate(&c.e1b));
      HIPCHK(hipEventCreate(&c.e2));
    }
  HIPCHK(hipStreamSynchronize(ctx->stream_up));
  if (timing)
    std::fprintf(stderr, "vsx_plan_create: %llu pairs, %zu tasks: classify %.2f ms, group %.2f ms (budget query %.2f), tasks %.2f ms, device buffers + upload %.2f ms\n",
                 (unsigned long long) n_pairs, pl->tasks.size(), (tc1 - tc0) * 1e3, (tc2 - tc1) * 1e3, t_budget * 1e3, (tc3 - tc2) * 1e3, (now() - tc3) * 1e3);
  *out = pl.release();
  return VSX_OK;
}

int vsx_plan_set_filter(vsx_plan * pl, const vsx_filter * f)
{
  if (!pl) return fail(VSX_EINVAL, "vsx_plan_set_filter: null plan");
  pl->filter = VsxFilterDev {};            // (also drops the lists of a ranked plan: plan_set_ranked() follows the filter)
  if (!f) return VSX_OK;
  if (!pl->ctx->ckpt) return fail(VSX_EINVAL, "vsx_plan_set_filter: needs the checkpoint traceback (VSX_TRACEBACK=dirs is set)");
  if (f->iddef < 0 || f->iddef > 4) return fail(VSX_EINVAL, "vsx_plan_set_filter: iddef must be 0..4");
  VsxFilterDev & d = pl->filter;
  d.enabled = 1; d.iddef = f->iddef; d.leftjust = f->leftjust; d.rightjust = f->rightjust;
  d.id = f->id; d.weak_id = f->weak_id; d.maxid = f->maxid; d.mid = f->mid; d.query_cov = f->query_cov; d.target_cov = f->target_cov;
  d.maxsubs = f->maxsubs; d.maxgaps = f->maxgaps; d.mincols = f->mincols; d.maxdiffs = f->maxdiffs;
  pl->ran = false;
  return VSX_OK;
}

// r06: make `pl` a RANKED plan -- its traceback lists the kept and the refused pairs (vsx_device.hip rank_note) for fetch_ranked_core.
// VSX_RANK_PRIM=1 keeps the r02-r05 path (flag kernel + rocPRIM scan / segmented sort over all pairs) for A/B.
static bool rank_lists_on()
{
  static const bool prim = std::getenv("VSX_RANK_PRIM") && std::strcmp(std::getenv("VSX_RANK_PRIM"), "1") == 0;
  return !prim;
}
static int plan_set_ranked(vsx_plan * pl)
{
  if (!rank_lists_on() || !pl->filter.enabled) return VSX_OK;
  vsx_ctx * ctx = pl->ctx;
  const size_t n = (size_t) pl->n_pairs;
  HIPCHK(hipSetDevice(ctx->device));
  HIPCHK(pl->d_rank_counts.alloc(&ctx->pool, 2));
  HIPCHK(pl->d_kept_pair.alloc(&ctx->pool, n + 1));
  HIPCHK(pl->d_kept_id.alloc(&ctx->pool, n + 1));
  HIPCHK(pl->d_refused_pair.alloc(&ctx->pool, n + 1));
  pl->filter.rank_counts = pl->d_rank_counts.p;
  pl->filter.kept_pair = pl->d_kept_pair.p;
  pl->filter.kept_id = pl->d_kept_id.p;
  pl->filter.refused_pair = pl->d_refused_pair.p;
  return VSX_OK;
}

int vsx_plan_run(vsx_plan * pl)
{
  if (!pl) return fail(VSX_EINVAL, "vsx_plan_run: null plan");
  vsx_ctx * ctx = pl->ctx;
  HIPCHK(hipSetDevice(ctx->device));
  hipStream_t st = pl->alt_fwd ? ctx->stream_b : ctx->stream, st2 = ctx->stream2;
  const int slot = pl->dir_slot;
  HIPCHK(hipMemsetAsync(pl->d_cursor.p, 0, 2 * sizeof(unsigned long long), st));
  if (pl->filter.rank_counts) HIPCHK(hipMemsetAsync(pl->d_rank_counts.p, 0, 2 * sizeof(uint32_t), st));      // (every run: settle() may run a plan twice)
  HIPCHK(hipEventRecord(pl->ev_begin, st));
  // DP kernels on `st`, tracebacks + text on `st2`.  A plan's DP waits only for the last traceback that read ITS checkpoint
  // block (the context alternates two), so it overlaps the previous plan's traceback; inside a plan the chunks share the block.
  for (size_t k = 0; k < pl->chunks.size(); ++k)
    {
      Chunk & c = pl->chunks[k];
      uint32_t * dir = pl->d_dir[0].p;
      if (k >= 1) HIPCHK(hipStreamWaitEvent(st, pl->chunks[k - 1].e2, 0));      // buffer reuse: the previous traceback is done
      else if (ctx->ev_tb_used[slot]) HIPCHK(hipStreamWaitEvent(st, ctx->ev_tb[slot], 0));
      // VSX_ALIGN_SERIAL=1 (A/B, r05): no overlap between a plan's DP and the previous plan's traceback.  The trace of an allpairs run
      // (profiles/r05/r05q_*) shows the R = 26 DP launches at 28.7 ms beside a traceback where the kernel alone takes 13.3 ms
      static const bool serial_env = std::getenv("VSX_ALIGN_SERIAL") && std::strcmp(std::getenv("VSX_ALIGN_SERIAL"), "1") == 0;
      if (k == 0 && serial_env && ctx->last_tb_slot >= 0 && ctx->last_tb_slot != slot) HIPCHK(hipStreamWaitEvent(st, ctx->ev_tb[ctx->last_tb_slot], 0));
      HIPCHK(hipEventRecord(c.e0, st));
      for (const Launch & L : c.launches)
        {
          VsxDevParams Pf = L.tilt ? ctx->Pt : ctx->P;
          Pf.max3 = (L.tilt == 2) ? 1 : 0;
          static const bool one_off = std::getenv("VSX_ONESTRIP") && std::strcmp(std::getenv("VSX_ONESTRIP"), "0") == 0;      // A/B, tests: the general kernels
          HIPCHK(vsx_launch_forward(L.rows, L.generic, L.track, ctx->ckpt ? 1 : 0, L.nq, (L.multi || (one_off && L.nq != 8)) ? 0 : 1, Pf, pl->d_tasks.p + L.first, L.count,
                                    pl->Q->codes(), pl->T->codes(), dir, pl->d_strip.p,
                                    pl->d_slot.p + (size_t) L.first * VSX_TASK_SLOTS, st));
        }
      HIPCHK(hipEventRecord(c.e1, st));
      HIPCHK(hipStreamWaitEvent(st2, c.e1, 0));
      HIPCHK(hipEventRecord(c.e1b, st2));
      if (ctx->ckpt)
        {
          for (size_t li = 0; li < c.launches.size();)      // the recompute traceback is specialised on R like the DP kernel
            {
              const Launch & L = c.launches[li];
              // (the sparse-task variants of one class -- nq = 1, 2, 4, adjacent in the class order -- share ONE traceback launch: the
              //  traceback finds a pair's lane group through its task and does not know about nq; their pairs are contiguous)
              uint32_t pair_count = L.pair_count;
              size_t lj = li + 1;
              while (lj < c.launches.size() && c.launches[lj].rows == L.rows && c.launches[lj].generic == L.generic &&
                     c.launches[lj].track == L.track && c.launches[lj].tilt == L.tilt) { pair_count += c.launches[lj].pair_count; ++lj; }
              VsxDevParams Pb = L.tilt ? ctx->Pt : ctx->P;
              Pb.max3 = (L.tilt == 2) ? 1 : 0;         // the checkpoints of the MAX3 class carry its bias
              HIPCHK(vsx_launch_traceback_ck(L.rows, (L.track == 0 && L.generic != 0) ? 1 : 0, Pb, pl->filter, pl->d_tasks.p, pl->d_pair_slot.p + L.pair_first,
                                             pl->d_pair_ids.p + L.pair_first, pair_count, pl->Q->codes(), pl->T->codes(),
                                             dir, pl->d_slot.p, pl->d_slab.p, pl->d_slab_off.p + L.pair_first,
                                             pl->d_runs.p, pl->runs_capacity, pl->d_cursor.p, pl->d_out.p, st2));
              li = lj;
            }
        }
      else
        HIPCHK(vsx_launch_traceback(ctx->P, pl->d_tasks.p, pl->d_pair_slot.p + c.pair_first, pl->d_pair_ids.p + c.pair_first,
                                    c.pair_count, pl->Q->codes(), pl->T->codes(), dir, pl->d_slot.p,
                                    pl->d_slab.p, pl->d_slab_off.p + c.pair_first, pl->d_runs.p, pl->runs_capacity,
                                    pl->d_cursor.p, pl->d_out.p, st2));
      HIPCHK(hipEventRecord(c.e2, st2));
    }
  if (pl->chunks.empty()) HIPCHK(hipStreamWaitEvent(st2, pl->ev_begin, 0));          // (the cursor memset precedes the text kernel)
  HIPCHK(hipEventRecord(ctx->ev_tb[slot], st2));
  ctx->ev_tb_used[slot] = true;
  ctx->last_tb_slot = slot;
  // run lists -> CIGAR text, records -> output arrays (pushop / finishop are part of the reference's timed path); st2 is in order
  HIPCHK(vsx_launch_cigar_text(pl->d_out.p, pl->d_pair_ids.p, (uint32_t) pl->pair_ids.size(), pl->d_runs.p, pl->runs_capacity,
                               pl->d_text.p, pl->text_capacity, pl->d_cursor.p + 1, pl->soa, st2));
  HIPCHK(hipMemcpyAsync(pl->h_cursor, pl->d_cursor.p, 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost, st2));
  HIPCHK(hipEventRecord(pl->ev_end, st2));
  pl->ran = true;
  pl->host_patched = false;            // the exported records of host-answered pairs point behind THIS run's device runs
  return VSX_OK;
}

int vsx_plan_sync(vsx_plan * pl, vsx_timing * tm)
{
  if (!pl) return fail(VSX_EINVAL, "vsx_plan_sync: null plan");
  if (!pl->ran) return fail(VSX_EINVAL, "vsx_plan_sync: plan has not been run");
  HIPCHK(hipSetDevice(pl->ctx->device));
  HIPCHK(hipEventSynchronize(pl->ev_end));
  if (tm)
    {
      *tm = vsx_timing {};
      for (Chunk & c : pl->chunks)
        {
          float a = 0, b = 0;
          HIPCHK(hipEventElapsedTime(&a, c.e0, c.e1));
          HIPCHK(hipEventElapsedTime(&b, c.e1b, c.e2));
          tm->forward_ms += a;
          tm->traceback_ms += b;
          tm->forward_launches += (uint32_t) c.launches.size();
          tm->traceback_launches += 1;
        }
      HIPCHK(hipEventElapsedTime(&tm->total_ms, pl->ev_begin, pl->ev_end));
      tm->cells = pl->cells;
      tm->dir_bytes = pl->dir_bytes_total;
    }
  return VSX_OK;
}

int vsx_plan_describe(const vsx_plan * pl, vsx_plan_info * info)
{
  if (!pl || !info) return fail(VSX_EINVAL, "vsx_plan_describe: null argument");
  *info = vsx_plan_info {};
  info->tasks = pl->tasks.size();
  info->chunks = (uint32_t) pl->chunks.size();
  uint64_t best = 0;
  for (const Chunk & c : pl->chunks)
    for (const Launch & L : c.launches)
      {
        if (L.nq == 2 || L.nq == 4) info->tasks_sparse += L.count;
        if (L.nq == 8) info->tasks_pair += L.count;
        info->waves += (L.nq == 2 || L.nq == 4) ? (L.count + (uint32_t) L.nq - 1) / (uint32_t) L.nq : L.count;
        if (L.tilt) info->tasks_tilted += L.count;
        if (L.tilt == 2) info->tasks_max3 += L.count;
        if (L.track) info->tasks_tracked += L.count;
        if (L.count > best) { best = L.count; info->rows_dominant = (uint32_t) L.rows; }
      }
  return VSX_OK;
}

static void append_cigar(std::string & s, const uint32_t * runs, uint32_t n)
{
  // runs are in traceback order (last column first); the text runs left to right,
  // count omitted when 1 (pushop/finishop, align_simd.cpp:1013-1049)
  static const char ops[4] = {'M', 'I', 'D', '?'};
  char buf[16];
  for (uint32_t k = n; k-- > 0;)
    {
      const uint32_t len = runs[k] >> 2;
      if (len > 1) { int w = snprintf(buf, sizeof buf, "%u", len); s.append(buf, (size_t) w); }
      s.push_back(ops[runs[k] & 3]);
    }
}

// Where a plan's results go: arrays of the caller (already offset to the plan's first pair), a growing text blob.
struct FetchDest {
  int16_t * score; uint16_t * aligned, * matches, * mismatches, * gaps; uint64_t * cigar_off; uint8_t * verdict;
  char ** blob; uint64_t * blob_used;      // text is appended at *blob_used (the blob is realloc'ed); offsets are rebased by it
};

// wait for a plan's kernels; re-run what overflowed (run buffer: the whole plan; text buffer: the formatting kernel)
static int settle(vsx_plan * pl, unsigned long long & used, unsigned long long & text_used)
{
  vsx_ctx * ctx = pl->ctx;
  if (!pl->ran) { int rc = vsx_plan_run(pl); if (rc != VSX_OK) return rc; }
  int rc = vsx_plan_sync(pl, nullptr);
  if (rc != VSX_OK) return rc;
  HIPCHK(hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  used = pl->h_cursor[0]; text_used = pl->h_cursor[1];
  if (used > pl->runs_capacity)
    {
      // the dense run buffer was sized for typical alignments; size it exactly and run again
      pl->runs_capacity = used + 1;
      HIPCHK(pl->d_runs.alloc(&pl->ctx->pool, pl->runs_capacity));
      if ((rc = vsx_plan_run(pl)) != VSX_OK) return rc;
      if ((rc = vsx_plan_sync(pl, nullptr)) != VSX_OK) return rc;
      used = pl->h_cursor[0]; text_used = pl->h_cursor[1];
      if (used > pl->runs_capacity) return fail(VSX_EHIP, "vsx_plan_fetch: run buffer overflow after resize");
    }
  if (text_used > pl->text_capacity)
    {
      // same for the text: only the formatting kernel runs again
      pl->text_capacity = text_used + 16;
      HIPCHK(pl->d_text.alloc(&pl->ctx->pool, pl->text_capacity));
      HIPCHK(hipMemsetAsync(pl->d_cursor.p + 1, 0, sizeof(unsigned long long), st));
      HIPCHK(vsx_launch_cigar_text(pl->d_out.p, pl->d_pair_ids.p, (uint32_t) pl->pair_ids.size(), pl->d_runs.p, pl->runs_capacity,
                                   pl->d_text.p, pl->text_capacity, pl->d_cursor.p + 1, pl->soa, st));
      HIPCHK(hipMemcpyAsync(pl->h_cursor, pl->d_cursor.p, 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
      HIPCHK(hipStreamSynchronize(st));
      text_used = pl->h_cursor[1];
      if (text_used > pl->text_capacity) return fail(VSX_EHIP, "vsx_plan_fetch: text buffer overflow after resize");
    }
  return VSX_OK;
}

// pinned staging of at least `need` bytes (caller holds ctx->stage_mu)
static int stage_reserve(vsx_ctx * ctx, uint64_t need)
{
  if (need <= ctx->stage_bytes) return VSX_OK;
  if (ctx->stage) { (void) hipHostFree(ctx->stage); ctx->stage = nullptr; ctx->stage_bytes = 0; }
  const uint64_t want = need + need / 4 + (1u << 20);
  if (hipHostMalloc(reinterpret_cast<void **>(&ctx->stage), want, hipHostMallocDefault) != hipSuccess)
    { (void) hipGetLastError(); return fail(VSX_ENOMEM, "vsx_plan_fetch: pinned staging allocation failed"); }
  ctx->stage_bytes = want;
  return VSX_OK;
}

static int fetch_core(vsx_plan * pl, const FetchDest & D)
{
  vsx_ctx * ctx = pl->ctx;
  unsigned long long used = 0, text_used = 0;
  int rc = settle(pl, used, text_used);
  if (rc != VSX_OK) return rc;

  const uint64_t n = pl->n_pairs;
  // the rare pairs answered on the host get their strings behind the device text
  uint64_t tail = 0;
  for (size_t h = 0, hc = 0; h < pl->host_pairs.size(); ++h)
    {
      ++tail;
      if (hc < pl->host_cigar_pair.size() && pl->host_cigar_pair[hc] == pl->host_pairs[h]) tail += pl->host_cigar[hc++].size();
    }
  const uint64_t base = *D.blob_used;
  {
    char * nb = (char *) std::realloc(*D.blob, std::max<uint64_t>(base + text_used + tail, 1));
    if (!nb) return fail(VSX_ENOMEM, "vsx_plan_fetch: host allocation failed");
    *D.blob = nb;
  }
  char * const blob = *D.blob;

  {
    // one PCIe crossing into pinned memory (arrays + text), then host threads spread it over the result arrays
    std::lock_guard<std::mutex> lk(ctx->stage_mu);
    if ((rc = stage_reserve(ctx, pl->soa_bytes + ((text_used + 15) & ~15ull))) != VSX_OK) return rc;
    hipError_t e = hipSuccess;
    // (the plan's kernels are done -- vsx_plan_sync above -- so the copies need no ordering against `stream`, where the next
    //  slice of a pipeline may already be running)
    if (n) e = hipMemcpyAsync(ctx->stage, pl->d_soa.p, pl->soa_bytes, hipMemcpyDeviceToHost, ctx->stream_dn);
    if (e == hipSuccess && text_used) e = hipMemcpyAsync(ctx->stage + pl->soa_bytes, pl->d_text.p, text_used, hipMemcpyDeviceToHost, ctx->stream_dn);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream_dn);
    if (e != hipSuccess) return fail(VSX_EHIP, "vsx_plan_fetch: %s", hipGetErrorString(e));
    const uint8_t * sg = ctx->stage;
    const int nth = copy_threads(n * 19 + text_used);
    run_threads(nth, [&](int t) {
      const uint64_t lo = n * (uint64_t) t / (uint64_t) nth, hi = n * (uint64_t) (t + 1) / (uint64_t) nth;
      if (hi > lo)
        {
          std::memcpy(D.score + lo, sg + pl->soa_off[0] + 2 * lo, 2 * (hi - lo));
          std::memcpy(D.aligned + lo, sg + pl->soa_off[1] + 2 * lo, 2 * (hi - lo));
          std::memcpy(D.matches + lo, sg + pl->soa_off[2] + 2 * lo, 2 * (hi - lo));
          std::memcpy(D.mismatches + lo, sg + pl->soa_off[3] + 2 * lo, 2 * (hi - lo));
          std::memcpy(D.gaps + lo, sg + pl->soa_off[4] + 2 * lo, 2 * (hi - lo));
          if (D.verdict) std::memcpy(D.verdict + lo, sg + pl->soa_off[5] + lo, hi - lo);
          const uint64_t * so = reinterpret_cast<const uint64_t *>(sg + pl->soa_off[6]);
          for (uint64_t k = lo; k < hi; ++k) D.cigar_off[k] = so[k] + base;
        }
      const uint64_t blo = text_used * (uint64_t) t / (uint64_t) nth, bhi = text_used * (uint64_t) (t + 1) / (uint64_t) nth;
      if (bhi > blo) std::memcpy(blob + base + blo, sg + pl->soa_bytes + blo, bhi - blo);
    });
  }
  uint64_t at = base + text_used;
  for (size_t h = 0, hc = 0; h < pl->host_pairs.size(); ++h)
    {
      const uint32_t k = pl->host_pairs[h];
      const VsxPairOut & o = pl->host_out[h];
      D.score[k] = o.score; D.aligned[k] = o.aligned; D.matches[k] = o.matches;
      D.mismatches[k] = o.mismatches; D.gaps[k] = o.gaps;
      if (D.verdict) D.verdict[k] = (uint8_t) VSX_VERDICT_UNDECIDED;
      D.cigar_off[k] = at;
      if (hc < pl->host_cigar_pair.size() && pl->host_cigar_pair[hc] == k)
        {
          std::memcpy(blob + at, pl->host_cigar[hc].data(), pl->host_cigar[hc].size());
          at += pl->host_cigar[hc++].size();
        }
      blob[at++] = '\0';
    }
  *D.blob_used = at;
  return VSX_OK;
}

// Ranked fetch (vsx_rank.hip): only the pairs the plan's filter kept, in report order, appended to a sink.
struct RankedSink {
  std::vector<uint32_t> pair, undecided;
  std::vector<int16_t> score;
  std::vector<uint16_t> aligned, matches, mismatches, gaps;
  std::vector<uint8_t> verdict;
  std::vector<double> id;
  std::vector<uint64_t> cigar_off;
  char * blob = nullptr;
  uint64_t blob_used = 0;
  int keep_weak = 0;
  ~RankedSink() { std::free(blob); }
};

// r06: the ranked fetch from the lists the traceback's epilogue wrote.  Kept pairs arrive in arbitrary order: their fields are gathered
// as they lie, cross PCIe, and the HOST puts the few of them in report order -- per query (queries ascending = pair index ascending
// across groups), identity descending, pair index ascending among equals: what the stable segmented sort produced.
static int fetch_ranked_lists(vsx_plan * pl, uint64_t first, const uint32_t * qkey, RankedSink & S, unsigned long long text_used)
{
  vsx_ctx * ctx = pl->ctx;
  hipStream_t st = ctx->stream_dn;
  const uint64_t n = pl->n_pairs;
  uint32_t counts[2] = {0, 0};
  HIPCHK(hipMemcpyAsync(counts, pl->d_rank_counts.p, sizeof counts, hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  const uint32_t kept = counts[0], n_refused = counts[1];
  if (kept > n || n_refused > n) return fail(VSX_EHIP, "vsx_align_pairs_ranked: the kept / refused lists are corrupt");
  if (n_refused)
    {
      // (include/vsx.h: a pair the 16-bit aligner refused is `undecided`, never silently dropped)
      std::vector<uint32_t> rf(n_refused);
      HIPCHK(hipMemcpy(rf.data(), pl->d_refused_pair.p, (size_t) n_refused * 4, hipMemcpyDeviceToHost));
      std::sort(rf.begin(), rf.end());
      for (uint32_t k : rf) S.undecided.push_back((uint32_t) (first + k));
    }
  if (!kept) return VSX_OK;
  const uint64_t width[9] = {4, 8, 8, 2, 2, 2, 2, 2, 1};
  uint64_t off[9], at = 0;
  for (int x = 0; x < 9; ++x) { off[x] = at; at += ((uint64_t) kept * width[x] + 15) & ~15ull; }
  const uint64_t rank_bytes = at;
  PoolBuf<uint8_t> d_rank;
  HIPCHK(d_rank.alloc(&ctx->pool, rank_bytes));
  VsxRankedOut R;
  R.pair = reinterpret_cast<uint32_t *>(d_rank.p + off[0]);
  R.id = reinterpret_cast<double *>(d_rank.p + off[1]);
  R.text_off = reinterpret_cast<uint64_t *>(d_rank.p + off[2]);
  R.score = reinterpret_cast<int16_t *>(d_rank.p + off[3]);
  R.aligned = reinterpret_cast<uint16_t *>(d_rank.p + off[4]);
  R.matches = reinterpret_cast<uint16_t *>(d_rank.p + off[5]);
  R.mismatches = reinterpret_cast<uint16_t *>(d_rank.p + off[6]);
  R.gaps = reinterpret_cast<uint16_t *>(d_rank.p + off[7]);
  R.verdict = d_rank.p + off[8];
  HIPCHK(vsx_rank_gather_list(pl->d_kept_pair.p, pl->d_kept_id.p, kept, pl->d_out.p, pl->soa.text_off, R, st));
  std::lock_guard<std::mutex> lk(ctx->stage_mu);
  int rc = stage_reserve(ctx, rank_bytes + ((text_used + 15) & ~15ull));
  if (rc != VSX_OK) return rc;
  HIPCHK(hipMemcpyAsync(ctx->stage, d_rank.p, rank_bytes, hipMemcpyDeviceToHost, st));
  if (text_used) HIPCHK(hipMemcpyAsync(ctx->stage + rank_bytes, pl->d_text.p, text_used, hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  const uint8_t * sg = ctx->stage;
  const char * text = reinterpret_cast<const char *>(sg + rank_bytes);
  const uint32_t * h_pair = reinterpret_cast<const uint32_t *>(sg + off[0]);
  const double * h_id = reinterpret_cast<const double *>(sg + off[1]);
  const uint64_t * h_toff = reinterpret_cast<const uint64_t *>(sg + off[2]);
  const int16_t * h_score = reinterpret_cast<const int16_t *>(sg + off[3]);
  const uint16_t * h_al = reinterpret_cast<const uint16_t *>(sg + off[4]), * h_ma = reinterpret_cast<const uint16_t *>(sg + off[5]);
  const uint16_t * h_mi = reinterpret_cast<const uint16_t *>(sg + off[6]), * h_ga = reinterpret_cast<const uint16_t *>(sg + off[7]);
  const uint8_t * h_vd = sg + off[8];
  std::vector<uint32_t> order;
  order.reserve(kept);
  for (uint32_t j = 0; j < kept; ++j)
    {
      if (h_pair[j] >= n) return fail(VSX_EHIP, "vsx_align_pairs_ranked: a kept pair is out of range");
      if (h_vd[j] == 1u || (S.keep_weak && h_vd[j] == 2u)) order.push_back(j);
    }
  std::sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) {
    const uint32_t pa = h_pair[a], pb = h_pair[b];
    if (qkey[pa] != qkey[pb]) return pa < pb;                 // (the pairs of a query are contiguous: another query = another segment)
    if (h_id[a] != h_id[b]) return h_id[a] > h_id[b];
    return pa < pb;
  });
  const size_t m = order.size(), base = S.pair.size();
  S.pair.resize(base + m); S.id.resize(base + m); S.cigar_off.resize(base + m);
  S.score.resize(base + m); S.aligned.resize(base + m); S.matches.resize(base + m);
  S.mismatches.resize(base + m); S.gaps.resize(base + m); S.verdict.resize(base + m);
  uint64_t need = 0;
  std::vector<uint32_t> slen(m);
  for (size_t x = 0; x < m; ++x) { slen[x] = (uint32_t) std::strlen(text + h_toff[order[x]]) + 1; need += slen[x]; }
  char * nb = (char *) std::realloc(S.blob, std::max<uint64_t>(S.blob_used + need, 1));
  if (!nb) return fail(VSX_ENOMEM, "vsx_align_pairs_ranked: host allocation failed");
  S.blob = nb;
  for (size_t x = 0; x < m; ++x)
    {
      const uint32_t j = order[x];
      S.pair[base + x] = (uint32_t) (first + h_pair[j]);
      S.id[base + x] = h_id[j];
      S.score[base + x] = h_score[j]; S.aligned[base + x] = h_al[j]; S.matches[base + x] = h_ma[j];
      S.mismatches[base + x] = h_mi[j]; S.gaps[base + x] = h_ga[j]; S.verdict[base + x] = h_vd[j];
      S.cigar_off[base + x] = S.blob_used;
      std::memcpy(S.blob + S.blob_used, text + h_toff[j], slen[x]);
      S.blob_used += slen[x];
    }
  return VSX_OK;
}

// `first` = index of the plan's pair 0 in the caller's list; `qkey` = the caller's query index per pair (plan-local view)
static int fetch_ranked_core(vsx_plan * pl, uint64_t first, const uint32_t * qkey, RankedSink & S)
{
  vsx_ctx * ctx = pl->ctx;
  unsigned long long used = 0, text_used = 0;
  int rc = settle(pl, used, text_used);
  if (rc != VSX_OK) return rc;
  const uint64_t n = pl->n_pairs;
  for (uint32_t k : pl->host_pairs) S.undecided.push_back((uint32_t) (first + k));
  if (n == 0) return VSX_OK;
  hipStream_t st = ctx->stream_dn;             // the plan's kernels are done; the next slice may own `stream`
  if (pl->filter.rank_counts) return fetch_ranked_lists(pl, first, qkey, S, text_used);

  // query groups of the pair list (pairs of one query are contiguous)
  std::vector<uint32_t> qstart;
  for (uint64_t k = 0; k < n; ++k) if (k == 0 || qkey[k] != qkey[k - 1]) qstart.push_back((uint32_t) k);
  const uint32_t G = (uint32_t) qstart.size();
  qstart.push_back((uint32_t) n);

  PoolBuf<uint32_t> d_flag, d_pos, d_qstart, d_seg, d_val_in, d_val_out;
  PoolBuf<double> d_id, d_key_in, d_key_out;
  PoolBuf<uint8_t> d_temp, d_rank;
  HIPCHK(d_flag.alloc(&ctx->pool, n + 1));
  HIPCHK(d_pos.alloc(&ctx->pool, n + 1));
  HIPCHK(d_id.alloc(&ctx->pool, n));
  HIPCHK(d_qstart.alloc(&ctx->pool, G + 1));
  HIPCHK(d_seg.alloc(&ctx->pool, G + 1));
  PoolBuf<uint32_t> d_refused;                 // [0] = count, then the GPU pairs whose 16-bit DP overflowed at run time
  const uint32_t ngpu = (uint32_t) pl->pair_ids.size();
  HIPCHK(d_refused.alloc(&ctx->pool, (size_t) ngpu + 1));
  HIPCHK(hipMemcpyAsync(d_qstart.p, qstart.data(), (G + 1) * 4, hipMemcpyHostToDevice, st));
  size_t tb = 0;
  HIPCHK(vsx_rank_flag_scan(pl->filter, S.keep_weak, pl->d_out.p, pl->d_pair_ids.p, pl->d_pair_slot.p, pl->d_tasks.p, (uint32_t) pl->pair_ids.size(),
                            (uint32_t) n, pl->d_runs.p, pl->runs_capacity, d_flag.p, d_pos.p, d_id.p, d_refused.p, nullptr, &tb, st));
  HIPCHK(d_temp.alloc(&ctx->pool, tb + 16));
  HIPCHK(vsx_rank_flag_scan(pl->filter, S.keep_weak, pl->d_out.p, pl->d_pair_ids.p, pl->d_pair_slot.p, pl->d_tasks.p, (uint32_t) pl->pair_ids.size(),
                            (uint32_t) n, pl->d_runs.p, pl->runs_capacity, d_flag.p, d_pos.p, d_id.p, d_refused.p, d_temp.p, &tb, st));
  uint32_t kept = 0, n_refused = 0;
  HIPCHK(hipMemcpyAsync(&kept, d_pos.p + n, 4, hipMemcpyDeviceToHost, st));
  HIPCHK(hipMemcpyAsync(&n_refused, d_refused.p, 4, hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  if (n_refused)
    {
      // the contract of vsx_ranked (include/vsx.h): a pair the 16-bit aligner refused -- before the launch (host_pairs above) or
      // by the overflow rule inside it -- is `undecided`, never silently dropped; the caller's fallback aligns and filters it
      if (n_refused > ngpu) return fail(VSX_EHIP, "vsx_align_pairs_ranked: refused-pair list is corrupt");
      std::vector<uint32_t> rf(n_refused);
      HIPCHK(hipMemcpy(rf.data(), d_refused.p + 1, (size_t) n_refused * 4, hipMemcpyDeviceToHost));
      std::sort(rf.begin(), rf.end());         // the device appends in arbitrary order
      for (uint32_t k : rf) S.undecided.push_back((uint32_t) (first + k));
    }

  const uint64_t k1 = std::max<uint32_t>(kept, 1);
  // compact output arrays in one block: pair u32, id f64, text_off u64, 5 x 16 bit, verdict u8 (16-byte aligned sections)
  const uint64_t width[9] = {4, 8, 8, 2, 2, 2, 2, 2, 1};
  uint64_t off[9], at = 0;
  for (int x = 0; x < 9; ++x) { off[x] = at; at += (k1 * width[x] + 15) & ~15ull; }
  const uint64_t rank_bytes = at;
  HIPCHK(d_rank.alloc(&ctx->pool, rank_bytes));
  VsxRankedOut R;
  R.pair = reinterpret_cast<uint32_t *>(d_rank.p + off[0]);
  R.id = reinterpret_cast<double *>(d_rank.p + off[1]);
  R.text_off = reinterpret_cast<uint64_t *>(d_rank.p + off[2]);
  R.score = reinterpret_cast<int16_t *>(d_rank.p + off[3]);
  R.aligned = reinterpret_cast<uint16_t *>(d_rank.p + off[4]);
  R.matches = reinterpret_cast<uint16_t *>(d_rank.p + off[5]);
  R.mismatches = reinterpret_cast<uint16_t *>(d_rank.p + off[6]);
  R.gaps = reinterpret_cast<uint16_t *>(d_rank.p + off[7]);
  R.verdict = d_rank.p + off[8];
  if (kept)
    {
      HIPCHK(d_key_in.alloc(&ctx->pool, kept)); HIPCHK(d_key_out.alloc(&ctx->pool, kept));
      HIPCHK(d_val_in.alloc(&ctx->pool, kept)); HIPCHK(d_val_out.alloc(&ctx->pool, kept));
      size_t sb = 0;
      HIPCHK(vsx_rank_sort_gather(d_flag.p, d_pos.p, d_id.p, (uint32_t) n, kept, d_qstart.p, G, d_key_in.p, d_key_out.p, d_val_in.p, d_val_out.p,
                                  d_seg.p, pl->d_out.p, pl->soa.text_off, R, nullptr, &sb, st));
      PoolBuf<uint8_t> d_temp2;
      HIPCHK(d_temp2.alloc(&ctx->pool, sb + 16));
      HIPCHK(vsx_rank_sort_gather(d_flag.p, d_pos.p, d_id.p, (uint32_t) n, kept, d_qstart.p, G, d_key_in.p, d_key_out.p, d_val_in.p, d_val_out.p,
                                  d_seg.p, pl->d_out.p, pl->soa.text_off, R, d_temp2.p, &sb, st));
      HIPCHK(hipStreamSynchronize(st));        // d_temp2 goes back to the pool at the end of this scope
    }

  // text: the whole device blob crosses (rejected pairs hold 4 bytes each), then only the kept strings are copied out
  std::lock_guard<std::mutex> lk(ctx->stage_mu);
  if ((rc = stage_reserve(ctx, rank_bytes + ((text_used + 15) & ~15ull))) != VSX_OK) return rc;
  if (kept) HIPCHK(hipMemcpyAsync(ctx->stage, d_rank.p, rank_bytes, hipMemcpyDeviceToHost, st));
  if (kept && text_used) HIPCHK(hipMemcpyAsync(ctx->stage + rank_bytes, pl->d_text.p, text_used, hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  if (!kept) return VSX_OK;
  const uint8_t * sg = ctx->stage;
  const char * text = reinterpret_cast<const char *>(sg + rank_bytes);
  const uint32_t * h_pair = reinterpret_cast<const uint32_t *>(sg + off[0]);
  const double * h_id = reinterpret_cast<const double *>(sg + off[1]);
  const uint64_t * h_toff = reinterpret_cast<const uint64_t *>(sg + off[2]);
  const size_t base = S.pair.size();
  S.pair.resize(base + kept); S.id.resize(base + kept); S.cigar_off.resize(base + kept);
  S.score.resize(base + kept); S.aligned.resize(base + kept); S.matches.resize(base + kept);
  S.mismatches.resize(base + kept); S.gaps.resize(base + kept); S.verdict.resize(base + kept);
  std::memcpy(S.id.data() + base, h_id, kept * 8);
  std::memcpy(S.score.data() + base, sg + off[3], kept * 2);
  std::memcpy(S.aligned.data() + base, sg + off[4], kept * 2);
  std::memcpy(S.matches.data() + base, sg + off[5], kept * 2);
  std::memcpy(S.mismatches.data() + base, sg + off[6], kept * 2);
  std::memcpy(S.gaps.data() + base, sg + off[7], kept * 2);
  std::memcpy(S.verdict.data() + base, sg + off[8], kept);
  // string lengths first (one realloc), then the copies
  uint64_t need = 0;
  std::vector<uint32_t> slen(kept);
  for (uint32_t j = 0; j < kept; ++j) { slen[j] = (uint32_t) std::strlen(text + h_toff[j]) + 1; need += slen[j]; }
  char * nb = (char *) std::realloc(S.blob, std::max<uint64_t>(S.blob_used + need, 1));
  if (!nb) return fail(VSX_ENOMEM, "vsx_align_pairs_ranked: host allocation failed");
  S.blob = nb;
  for (uint32_t j = 0; j < kept; ++j)
    {
      S.pair[base + j] = (uint32_t) (first + h_pair[j]);
      S.cigar_off[base + j] = S.blob_used;
      std::memcpy(S.blob + S.blob_used, text + h_toff[j], slen[j]);
      S.blob_used += slen[j];
    }
  return VSX_OK;
}

// result arrays for n pairs (the blob starts empty and grows with every fetched plan)
static int results_alloc(vsx_results * out, uint64_t n, bool with_verdict)
{
  std::memset(out, 0, sizeof *out);
  const uint64_t n1 = std::max<uint64_t>(n, 1);
  out->n_pairs = n;
  out->score = (int16_t *) std::malloc(n1 * 2);
  out->aligned = (uint16_t *) std::malloc(n1 * 2);
  out->matches = (uint16_t *) std::malloc(n1 * 2);
  out->mismatches = (uint16_t *) std::malloc(n1 * 2);
  out->gaps = (uint16_t *) std::malloc(n1 * 2);
  out->cigar_off = (uint64_t *) std::malloc(n1 * 8);
  out->verdict = with_verdict ? (uint8_t *) std::malloc(n1) : nullptr;
  out->cigar_blob = (char *) std::malloc(1);
  if (!out->score || !out->aligned || !out->matches || !out->mismatches || !out->gaps || !out->cigar_off || !out->cigar_blob ||
      (with_verdict && !out->verdict))
    { vsx_results_free(out); return fail(VSX_ENOMEM, "vsx_plan_fetch: host allocation failed"); }
  return VSX_OK;
}

static FetchDest dest_at(vsx_results * out, uint64_t first)
{
  return FetchDest {out->score + first, out->aligned + first, out->matches + first, out->mismatches + first, out->gaps + first,
                    out->cigar_off + first, out->verdict ? out->verdict + first : nullptr, &out->cigar_blob, &out->cigar_bytes};
}

int vsx_plan_fetch(vsx_plan * pl, vsx_results * out)
{
  if (!pl || !out) return fail(VSX_EINVAL, "vsx_plan_fetch: null argument");
  int rc = results_alloc(out, pl->n_pairs, pl->filter.enabled != 0);
  if (rc != VSX_OK) return rc;
  rc = fetch_core(pl, dest_at(out, 0));
  if (rc != VSX_OK) { const std::string keep = g_err; vsx_results_free(out); g_err = keep; }
  return rc;
}

int vsx_plan_export_hits(vsx_plan * pl, void * d_dst, uint64_t dst_bytes)
{
  static_assert(sizeof(VsxPairOut) == VSX_HIT_RECORD_BYTES, "hit record layout");
  if (!pl || !d_dst) return fail(VSX_EINVAL, "vsx_plan_export_hits: null argument");
  if (!pl->ran) return fail(VSX_EINVAL, "vsx_plan_export_hits: plan has not been run");
  if (dst_bytes < pl->n_pairs * sizeof(VsxPairOut)) return fail(VSX_EINVAL, "vsx_plan_export_hits: destination too small");
  HIPCHK(hipSetDevice(pl->ctx->device));
  hipStream_t st = pl->ctx->stream;
  HIPCHK(hipEventSynchronize(pl->ev_end));          // the records are written on the traceback stream
  if (!pl->host_patched)
    {
      // pairs answered without DP live on the host: patch them into the device array once; their CIGARs (the Q == 0 closed
      // form, one 'I' run) follow the device runs in the exported run buffer (vsx_plan_export_runs)
      HIPCHK(hipEventSynchronize(pl->ev_end));
      const uint64_t used = pl->h_cursor[0];
      for (size_t h = 0, hc = 0; h < pl->host_pairs.size(); ++h)
        {
          VsxPairOut o = pl->host_out[h];
          if (hc < pl->host_cigar_pair.size() && pl->host_cigar_pair[hc] == pl->host_pairs[h]) { o.nruns = 1; o.run_off = used + hc; ++hc; }
          HIPCHK(hipMemcpy(pl->d_out.p + pl->host_pairs[h], &o, sizeof(VsxPairOut), hipMemcpyHostToDevice));
        }
      pl->host_patched = true;
    }
  HIPCHK(hipMemcpyAsync(d_dst, pl->d_out.p, pl->n_pairs * sizeof(VsxPairOut), hipMemcpyDeviceToDevice, st));
  HIPCHK(hipStreamSynchronize(st));
  return VSX_OK;
}

int vsx_plan_export_runs(vsx_plan * pl, void * d_dst, uint64_t dst_bytes, uint64_t * n_runs)
{
  if (!pl || !n_runs) return fail(VSX_EINVAL, "vsx_plan_export_runs: null argument");
  if (!pl->ran) return fail(VSX_EINVAL, "vsx_plan_export_runs: plan has not been run");
  HIPCHK(hipSetDevice(pl->ctx->device));
  HIPCHK(hipEventSynchronize(pl->ev_end));
  const uint64_t used = pl->h_cursor[0];
  if (used > pl->runs_capacity) return fail(VSX_EINVAL, "vsx_plan_export_runs: the run buffer overflowed; call vsx_plan_fetch first (it resizes and re-runs)");
  const uint64_t extra = pl->host_cigar_run.size();          // run words of the pairs answered on the host, behind the device's
  *n_runs = used + extra;
  if (!d_dst) return VSX_OK;                     // size query
  if (dst_bytes < (used + extra) * 4) return fail(VSX_EINVAL, "vsx_plan_export_runs: destination too small");
  if (used) HIPCHK(hipMemcpyAsync(d_dst, pl->d_runs.p, used * 4, hipMemcpyDeviceToDevice, pl->ctx->stream));
  if (extra) HIPCHK(hipMemcpyAsync(static_cast<uint32_t *>(d_dst) + used, pl->host_cigar_run.data(), extra * 4, hipMemcpyHostToDevice, pl->ctx->stream));
  HIPCHK(hipStreamSynchronize(pl->ctx->stream));
  return VSX_OK;
}

int64_t vsx_cigar_from_runs(const uint32_t * runs, uint32_t n, char * dst, uint64_t cap)
{
  if (n && !runs) { (void) fail(VSX_EINVAL, "vsx_cigar_from_runs: null argument"); return VSX_EINVAL; }
  std::string s;
  append_cigar(s, runs, n);
  if (dst && cap > s.size()) std::memcpy(dst, s.c_str(), s.size() + 1);
  return (int64_t) s.size() + 1;
}

void vsx_plan_destroy(vsx_plan * pl)
{
  if (!pl) return;
  (void) hipSetDevice(pl->ctx->device);
  // all of a plan's device work precedes its ev_end (uploads were synchronised at creation): waiting for the context's
  // streams instead would also wait for the NEXT plan of a pipeline and leave the GPU idle between slices
  if (pl->ran) (void) hipEventSynchronize(pl->ev_end);
  delete pl;
}

int vsx_align_pairs(vsx_ctx * ctx, const vsx_seqset * queries, const vsx_seqset * targets, uint64_t n_pairs,
                    const uint32_t * qidx, const uint32_t * tidx, vsx_results * out)
{
  return vsx_align_pairs_filtered(ctx, queries, targets, n_pairs, qidx, tidx, nullptr, out);
}

// one plan: create, run, fetch, destroy
static int align_pairs_single(vsx_ctx * ctx, const vsx_seqset * queries, const vsx_seqset * targets, uint64_t n_pairs,
                              const uint32_t * qidx, const uint32_t * tidx, const vsx_filter * filter, vsx_results * out,
                              RankedSink * sink = nullptr)
{
  static const bool timing = std::getenv("VSX_DEBUG_TIMING") != nullptr;
  auto now = []() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  const double t0 = now();
  vsx_plan * pl = nullptr;
  int rc = vsx_plan_create(ctx, &pl, queries, targets, n_pairs, qidx, tidx, 0);
  if (rc != VSX_OK) return rc;
  if (filter && ctx->ckpt) { rc = vsx_plan_set_filter(pl, filter); if (rc != VSX_OK) { vsx_plan_destroy(pl); return rc; } }
  if (sink) { rc = plan_set_ranked(pl); if (rc != VSX_OK) { vsx_plan_destroy(pl); return rc; } }
  const double t1 = now();
  rc = vsx_plan_run(pl);
  double t2 = t1, t3 = t1;
  if (rc == VSX_OK) { rc = vsx_plan_sync(pl, nullptr); t2 = now(); }
  if (rc == VSX_OK) { rc = sink ? fetch_ranked_core(pl, 0, qidx, *sink) : vsx_plan_fetch(pl, out); t3 = now(); }
  vsx_plan_destroy(pl);
  if (timing)
    std::fprintf(stderr, "vsx_align_pairs: %llu pairs: plan %.3f s, run+sync %.3f s, fetch %.3f s, destroy %.3f s\n",
                 (unsigned long long) n_pairs, t1 - t0, t2 - t1, t3 - t2, now() - t3);
  return rc;
}

// Large pair lists run as a PIPELINE of plans over contiguous slices (cut at query boundaries): a helper thread plans slice
// i+1 (host grouping, task upload) while the GPU runs slice i and the caller's thread fetches slice i-1 (download, CIGAR
// text); the slices' results are concatenated.  Same results as one plan -- a pair's alignment does not depend on its
// batch.  VSX_PIPELINE=0 switches it off.
static int align_pairs_impl(vsx_ctx * ctx, const vsx_seqset * queries, const vsx_seqset * targets, uint64_t n_pairs,
                            const uint32_t * qidx, const uint32_t * tidx, const vsx_filter * filter, vsx_results * out, RankedSink * sink)
{
  static const bool pipeline_off = std::getenv("VSX_PIPELINE") && std::strcmp(std::getenv("VSX_PIPELINE"), "0") == 0;
  // slice size: VSX_PIPELINE_SLICE, else a quarter of the list within [128 k, 2 M] pairs -- mid-sized lists (one search
  // stage of 100 k queries = 800 k pairs) overlap planning, kernels and the fetch as well; below 256 k pairs one plan
  static const uint64_t forced_slice = std::getenv("VSX_PIPELINE_SLICE") ? std::max<uint64_t>(64, std::strtoull(std::getenv("VSX_PIPELINE_SLICE"), nullptr, 10)) : 0;
  const uint64_t slice_pairs = forced_slice ? forced_slice : std::min<uint64_t>(2ull << 20, std::max<uint64_t>(128ull << 10, n_pairs / 4));
  if (pipeline_off || !ctx || (!out && !sink) || !queries || !targets || !qidx || !tidx || n_pairs < 2 * slice_pairs)
    return align_pairs_single(ctx, queries, targets, n_pairs, qidx, tidx, filter, out, sink);
  static const bool timing = std::getenv("VSX_DEBUG_TIMING") != nullptr;
  auto now = []() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  const double t_begin = now();
  if (out) std::memset(out, 0, sizeof *out);
  if (hipSetDevice(ctx->device) != hipSuccess) return fail(VSX_EHIP, "vsx_align_pairs: hipSetDevice failed");

  // slices of about slice_pairs pairs, cut where the query changes (a query's pairs then share tasks as in one plan)
  // (the first and the last slice are half-size: the GPU starts after planning the first one, and the last one's fetch is
  //  the only one nothing overlaps)
  std::vector<uint64_t> cut {0};
  // r04: the slices RAMP -- a quarter and a half slice first, a half and a quarter last.  The timeline of one 800 k-pair call
  // (profiles/r04/r04m_e2e_timeline.txt) is GPU-bound from the first launch to the last kernel, so what the caller waits for beyond
  // the kernels is the planning of the first slice (2.7 ms for 100 k pairs) and the traceback + text + download of the last one;
  // both shrink with their slice, and the planner (15 ns per pair) outruns the GPU (35 ns per pair) by the second slice.
  // VSX_PIPELINE_RAMP=0: the r03 schedule (half, full ..., half).
  static const bool ramp = !(std::getenv("VSX_PIPELINE_RAMP") && std::strcmp(std::getenv("VSX_PIPELINE_RAMP"), "0") == 0);
  std::vector<uint64_t> sizes;
  if (ramp && n_pairs >= 3 * slice_pairs)
    {
      const uint64_t head[2] = {slice_pairs / 4, slice_pairs / 2};
      const uint64_t mid = n_pairs - 2 * (head[0] + head[1]);
      const uint64_t k = (mid + slice_pairs - 1) / slice_pairs;
      sizes.push_back(head[0]); sizes.push_back(head[1]);
      for (uint64_t x = 0; x < k; ++x) sizes.push_back(mid * (x + 1) / k - mid * x / k);
      sizes.push_back(head[1]); sizes.push_back(head[0]);
    }
  size_t next_size = 0;
  while (cut.back() < n_pairs)
    {
      const uint64_t left = n_pairs - cut.back();
      uint64_t want = slice_pairs;
      if (!sizes.empty()) want = next_size < sizes.size() ? sizes[next_size++] : left;
      else if (cut.size() == 1) want = slice_pairs / 2;
      else if (left <= slice_pairs / 2 + slice_pairs / 8) want = left;
      else if (left <= slice_pairs + slice_pairs / 2) want = left - slice_pairs / 2;
      uint64_t e = std::min<uint64_t>(n_pairs, cut.back() + want);
      const uint64_t limit = sink ? n_pairs : std::min<uint64_t>(n_pairs, e + want / 2);      // ranked: a query is never split
      while (e < limit && qidx[e] == qidx[e - 1]) ++e;
      if (n_pairs - e < slice_pairs / 16) e = n_pairs;
      cut.push_back(e);
    }
  const size_t S = cut.size() - 1;
  const uint64_t slice_budget = 0;       // the context's shared checkpoint block (the plans execute in stream order)
  std::vector<vsx_plan *> plans(S, nullptr);
  std::vector<int> plan_rc(S, VSX_OK);
  std::vector<std::string> plan_msg(S);
  if (!sink)
    {
      const int arc = results_alloc(out, n_pairs, filter && ctx->ckpt);
      if (arc != VSX_OK) return arc;
    }
  std::mutex mu;
  std::condition_variable cv;
  size_t consumed = 0;
  std::vector<char> is_ready(S, 0);
  bool stop = false;
  int plan_failed = VSX_OK;
  std::string plan_failed_msg;
  static const size_t depth = std::getenv("VSX_PIPELINE_DEPTH") ? (size_t) std::max(1, std::min(6, std::atoi(std::getenv("VSX_PIPELINE_DEPTH")))) : 2;
  // r06, measured and NOT adopted: VSX_PLANNERS=2 / 3 planner threads (slices i = k, k + n, ...; plans of one context may be made
  // concurrently: pool, block and slot choices are locked, uploads share stream_up).  A planner needs 2.1-2.8 ms per slice where the
  // GPU needs 5.3, so a second one only adds contention: 29.9 / 31.2 ms per 800 k-pair call with one, 30.6 / 35.4 with two, 30.8 with
  // three (profiles/r06/r06r_e2e_planners_ab.txt).  What the call loses against its 25.4 ms of kernels is on the device: the traceback of
  // slice i runs starved beside the DP kernels of slices i + 1, i + 2 (a retiring DP wave frees 128 VGPRs, a traceback wave needs 168),
  // and the slice that reuses its checkpoint block waits for it.
  static const size_t n_planners = std::getenv("VSX_PLANNERS") ? (size_t) std::max(1, std::min(4, std::atoi(std::getenv("VSX_PLANNERS")))) : 1;
  auto planner_body = [&](size_t first_slice) {
    (void) hipSetDevice(ctx->device);
    for (size_t i = first_slice; i < S; i += n_planners)
      {
        {
          std::unique_lock<std::mutex> lk(mu);
          cv.wait(lk, [&] { return stop || i < consumed + depth + 2; });      // at most depth + 2 plans alive beyond the fetched ones
          if (stop) return;
        }
        vsx_plan * pl = nullptr;
        int rc = vsx_plan_create(ctx, &pl, queries, targets, cut[i + 1] - cut[i], qidx + cut[i], tidx + cut[i], slice_budget);
        if (rc == VSX_OK && filter && ctx->ckpt) rc = vsx_plan_set_filter(pl, filter);
        if (rc == VSX_OK && sink) rc = plan_set_ranked(pl);
        if (rc != VSX_OK) { plan_msg[i] = vsx_last_error(); if (pl) vsx_plan_destroy(pl); pl = nullptr; }
        if (timing) std::fprintf(stderr, "  slice %zu (%llu pairs): planned at %.1f ms\n", i, (unsigned long long) (cut[i + 1] - cut[i]), (now() - t_begin) * 1e3);
        std::lock_guard<std::mutex> lk(mu);
        plans[i] = pl; plan_rc[i] = rc; is_ready[i] = 1;
        cv.notify_all();
        if (rc != VSX_OK)                                          // (the consumer stops at this slice or, if the other planner then
          {                                                        //  leaves an earlier one unplanned, at that one: plan_failed)
            if (plan_failed == VSX_OK) { plan_failed = rc; plan_failed_msg = plan_msg[i]; }
            stop = true;
            return;
          }
      }
  };
  std::vector<std::thread> planners;
  for (size_t k = 0; k < std::min(n_planners, S); ++k) planners.emplace_back(planner_body, k);
  int rc = VSX_OK;
  std::string msg;
  size_t launched = 0;
  auto finish = [&](size_t i) {           // fetch slice i straight into its part of the result arrays, release it
    const double tf0 = now();
    int frc = sink ? fetch_ranked_core(plans[i], cut[i], qidx + cut[i], *sink) : fetch_core(plans[i], dest_at(out, cut[i]));
    if (timing)
      {
        vsx_timing tm;
        if (vsx_plan_sync(plans[i], &tm) == VSX_OK)
          std::fprintf(stderr, "  slice %zu: fetch %.1f .. %.1f ms, kernels %.2f ms (fwd %.2f tb %.2f)\n", i, (tf0 - t_begin) * 1e3, (now() - t_begin) * 1e3,
                       tm.total_ms, tm.forward_ms, tm.traceback_ms);
      }
    vsx_plan_destroy(plans[i]);
    plans[i] = nullptr;
    { std::lock_guard<std::mutex> lk(mu); consumed = i + 1; }
    cv.notify_all();
    return frc;
  };
  for (size_t i = 0; i < S && rc == VSX_OK; ++i)
    {
      {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return is_ready[i] != 0 || plan_failed != VSX_OK; });
        if (!is_ready[i]) { rc = plan_failed; msg = plan_failed_msg; break; }
      }
      if (plan_rc[i] != VSX_OK) { rc = plan_rc[i]; msg = plan_msg[i]; break; }
      // odd slices launch their DP kernels on the second DP stream: a slice is ~3-6 rounds of resident waves, its last round drains
      // for about half a wave's lifetime, and a launch on the same stream would wait for that (inputs are complete: plan_create
      // has synchronised its uploads; the two checkpoint blocks alternate the same way)
      static const bool alt_env = !(std::getenv("VSX_ALIGN_ALT_STREAM") && std::strcmp(std::getenv("VSX_ALIGN_ALT_STREAM"), "0") == 0);
      plans[i]->alt_fwd = alt_env && (i & 1);
      rc = vsx_plan_run(plans[i]);                       // asynchronous: queued behind slice i-1 on the context's streams
      if (timing) std::fprintf(stderr, "  slice %zu: queued at %.1f ms\n", i, (now() - t_begin) * 1e3);
      if (rc != VSX_OK) { msg = vsx_last_error(); break; }
      launched = i + 1;
      // `depth` slices stay queued behind the one being fetched (VSX_PIPELINE_DEPTH, default 2): the DP kernels of slice i+1 overlap
      // the traceback of slice i, so the GPU needs the next launch in its queue before the previous slice has drained.  r04 A/B
      // (profiles/r04/r04q_e2e_depth_ab.txt): 3 or 4 queued slices are no better -- a traceback makes little progress beside the DP
      // kernels queued after it (its waves need 256 VGPRs, a retiring DP wave frees 128), so deeper queues only move the wait.
      if (i >= depth) { rc = finish(i - depth); if (rc != VSX_OK) msg = vsx_last_error(); }
    }
  for (size_t i = (launched >= depth ? launched - depth : 0); i < launched && rc == VSX_OK && launched == S; ++i)
    { rc = finish(i); if (rc != VSX_OK) msg = vsx_last_error(); }
  {
    std::lock_guard<std::mutex> lk(mu);
    stop = true;
  }
  cv.notify_all();
  for (std::thread & t : planners) t.join();
  for (size_t i = 0; i < S; ++i)
    if (plans[i]) { vsx_plan_destroy(plans[i]); plans[i] = nullptr; }
  if (rc != VSX_OK)
    {
      if (out) vsx_results_free(out);
      vsx_internal_set_error(msg.c_str());
      return rc;
    }
  if (timing)
    std::fprintf(stderr, "vsx_align_pairs: %llu pairs in %zu pipelined slices: %.3f s\n", (unsigned long long) n_pairs, S, now() - t_begin);
  return VSX_OK;
}

int vsx_align_pairs_filtered(vsx_ctx * ctx, const vsx_seqset * queries, const vsx_seqset * targets, uint64_t n_pairs,
                             const uint32_t * qidx, const uint32_t * tidx, const vsx_filter * filter, vsx_results * out)
{
  if (!out) return fail(VSX_EINVAL, "vsx_align_pairs: null argument");
  return align_pairs_impl(ctx, queries, targets, n_pairs, qidx, tidx, filter, out, nullptr);
}

int vsx_align_pairs_ranked(vsx_ctx * ctx, const vsx_seqset * queries, const vsx_seqset * targets, uint64_t n_pairs,
                           const uint32_t * qidx, const uint32_t * tidx, const vsx_filter * filter, int keep_weak, vsx_ranked * out)
{
  if (!ctx || !out || !queries || !targets || (n_pairs && (!qidx || !tidx))) return fail(VSX_EINVAL, "vsx_align_pairs_ranked: null argument");
  std::memset(out, 0, sizeof *out);
  if (!filter) return fail(VSX_EINVAL, "vsx_align_pairs_ranked: the ranking needs the accept filter");
  if (!ctx->ckpt) return fail(VSX_EINVAL, "vsx_align_pairs_ranked: needs the checkpoint traceback (VSX_TRACEBACK=dirs is set)");
  RankedSink sink;
  sink.keep_weak = keep_weak ? 1 : 0;
  const int rc = align_pairs_impl(ctx, queries, targets, n_pairs, qidx, tidx, filter, nullptr, &sink);
  if (rc != VSX_OK) return rc;
  out->n_pairs = n_pairs;
  out->n_hits = sink.pair.size();
  out->pair = dup_array(sink.pair); out->score = dup_array(sink.score); out->aligned = dup_array(sink.aligned);
  out->matches = dup_array(sink.matches); out->mismatches = dup_array(sink.mismatches); out->gaps = dup_array(sink.gaps);
  out->verdict = dup_array(sink.verdict); out->id = dup_array(sink.id); out->cigar_off = dup_array(sink.cigar_off);
  out->n_undecided = sink.undecided.size();
  out->undecided = dup_array(sink.undecided);
  out->cigar_blob = sink.blob ? sink.blob : (char *) std::malloc(1);
  out->cigar_bytes = sink.blob_used;
  sink.blob = nullptr;
  if (!out->pair || !out->score || !out->aligned || !out->matches || !out->mismatches || !out->gaps || !out->verdict || !out->id ||
      !out->cigar_off || !out->undecided || !out->cigar_blob)
    { vsx_ranked_free(out); return fail(VSX_ENOMEM, "vsx_align_pairs_ranked: host allocation failed"); }
  return VSX_OK;
}

void vsx_ranked_free(vsx_ranked * r)
{
  if (!r) return;
  std::free(r->pair); std::free(r->score); std::free(r->aligned); std::free(r->matches); std::free(r->mismatches); std::free(r->gaps);
  std::free(r->verdict); std::free(r->id); std::free(r->cigar_off); std::free(r->cigar_blob); std::free(r->undecided);
  std::memset(r, 0, sizeof *r);
}

void vsx_results_free(vsx_results * r)
{
  if (!r) return;
  std::free(r->score); std::free(r->aligned); std::free(r->matches); std::free(r->mismatches);
  std::free(r->gaps); std::free(r->cigar_off); std::free(r->cigar_blob); std::free(r->verdict);
  std::memset(r, 0, sizeof *r);
}

}  // extern "C"
