// vsx_kmer_host.cpp -- builds and drives the device k-mer index (kernels: vsx_kmer.hip).
#include "vsx_kmer.h"
#include "vsx_internal.h"

#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <memory>
#include <mutex>
#include <string>

extern "C" void vsx_internal_set_error(const char * msg);
extern "C" int vsx_internal_device(const vsx_ctx * ctx);
extern "C" hipStream_t vsx_internal_stream(const vsx_ctx * ctx);
extern "C" void vsx_internal_seqset_device(const vsx_seqset * s, const uint8_t ** codes, const uint64_t ** off,
                                           const uint32_t ** len, uint64_t * n);
// the set's case bitmap (soft masking) or NULL: an index over a set that has one leaves out every word over a lower-case symbol
extern "C" const uint8_t * vsx_internal_seqset_lower(const vsx_seqset * s);
extern "C" const uint32_t * vsx_internal_seqset_host_lengths(const vsx_seqset * s);

// vsx_kmer.hip, packed postings (see there)
extern "C" hipError_t vsx_kmer_packed_tile(int fill, const uint8_t * codes, const uint64_t * off, const uint32_t * len, uint32_t first_seq,
                                           uint32_t nseq_tile, int w, const uint8_t * lower_bits, const uint64_t * slot_of, uint64_t n_slots,
                                           uint32_t * keys_a, uint32_t * keys_b, void * temp, size_t * temp_bytes, uint32_t tile, uint32_t ntiles,
                                           uint32_t * bucket_count, const uint64_t * bucket_start, uint32_t * postings, hipStream_t st);
extern "C" uint32_t vsx_kmer_packed_tile_seqs(void);
#include "vsx_kmer_pack.h"
extern "C" hipError_t vsx_kmer_launch_select_packed(const void * rec, uint32_t subcap, uint32_t ntiles, const uint32_t * tile_count, uint32_t nslots,
                                                    uint32_t keep, void * dense, unsigned long long * cursor, uint64_t capacity,
                                                    void * sel_m_n, uint64_t * sel_off, hipStream_t st);

extern "C" void vsx_internal_poison(void * p, size_t bytes);
extern "C" uint64_t vsx_internal_memory_pressure(int device);
#ifndef VSX_DEVICE_RESERVE_BYTES
#define VSX_DEVICE_RESERVE_BYTES ((size_t) 6 << 30)      // what stays free for the runtime itself (kernel scratch of every queue, code objects)
#endif      // vsx_host.cpp: every context of the device frees what no plan holds
namespace {

int kfail(int code, const char * what, hipError_t e)
{
  std::string m = std::string(what) + ": " + hipGetErrorString(e);
  vsx_internal_set_error(m.c_str());
  return code;
}

template <typename T> struct Buf {
  T * p = nullptr;
  size_t n = 0;
  ~Buf() { if (p) (void) hipFree(p); }
  hipError_t alloc(size_t count)
  {
    if (p) { (void) hipFree(p); p = nullptr; n = 0; }
    const size_t want = std::max<size_t>(count, 1) * sizeof(T);
    if (want >= ((size_t) 16 << 20))
      {
        // a device filled to the brim fails LATER and worse than a refused hipMalloc: the runtime cannot allocate a queue's kernel scratch
        // and aborts the process (HSA_STATUS_ERROR_OUT_OF_RESOURCES, profiles/r05/r05b_config5_share_abort.txt).  Keep a reserve.
        size_t free_b = 0, total_b = 0;
        int dev = 0;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && free_b < want + VSX_DEVICE_RESERVE_BYTES && hipGetDevice(&dev) == hipSuccess)
          (void) vsx_internal_memory_pressure(dev);
        (void) hipGetLastError();
      }
    hipError_t e = hipMalloc(reinterpret_cast<void **>(&p), std::max<size_t>(count, 1) * sizeof(T));
    if (e == hipErrorOutOfMemory)
      {
        // (r05) the aligner contexts of this device may sit on idle checkpoint blocks of an earlier, much larger plan: ask for them
        int dev = 0;
        (void) hipGetLastError();
        if (hipGetDevice(&dev) == hipSuccess && vsx_internal_memory_pressure(dev) > 0)
          e = hipMalloc(reinterpret_cast<void **>(&p), std::max<size_t>(count, 1) * sizeof(T));
      }
    if (e == hipSuccess) { n = count; vsx_internal_poison(p, std::max<size_t>(count, 1) * sizeof(T)); }
    return e;
  }
  hipError_t ensure(size_t count) { return (p && count <= n) ? hipSuccess : alloc(count); }
};

}  // namespace

// Per-call state of a counting batch.  The index itself is read-only between rebuilds, so several batches may count against
// it at once, each with its own scratch and stream: the search runs two windows' k-mer stages concurrently, one's host work
// (CSR, uploads, record download, ranking) under the other's counting kernel.
struct KmerScratch {
  hipStream_t st = nullptr;
  Buf<uint64_t> d_qk_start;
  Buf<uint32_t> d_qk, d_minmatch;
  Buf<uint64_t> d_rec, d_dense, d_sel_mn, d_sel_off;    // uint2 records (target, count); (kept, seen) per slot
  Buf<uint64_t> d_ranges;                                // uint2 (first unit, units) per (tile, slot, word): 8-bit class
  Buf<uint32_t> d_tilecnt;                               // records per (slot, tile)
  Buf<unsigned long long> d_cursor;
  hipEvent_t e0 = nullptr, e1 = nullptr, e_turn = nullptr;   // timing; e_turn = this batch's counting kernels are done
  uint64_t records = 0;
  bool busy = false;
  ~KmerScratch()
  {
    if (e_turn) (void) hipEventDestroy(e_turn);
    if (e0) (void) hipEventDestroy(e0);
    if (e1) (void) hipEventDestroy(e1);
    if (st) { (void) hipStreamSynchronize(st); (void) hipStreamDestroy(st); }
  }
};

struct VsxKmerIndex {
  vsx_ctx * ctx = nullptr;
  const vsx_seqset * db = nullptr;
  int device = 0;
  hipStream_t st = nullptr;
  int w = 8;
  bool tagged = false;            // word lengths 9..15: buckets by the word's low 16 bits, postings carry the rest as a tag (vsx_kmer.hip)
  bool packed = false;            // word lengths 3..8, whole-set index: sorted byte-gap postings, tiles of 32 630 sequences (vsx_kmer.hip)
  bool prewarm = false;           // a search index: make the spare scratch sets behind the first batch (vsx_kmer_count_batch)
  uint32_t nseq = 0, ntiles = 0;
  uint64_t nbuckets = 0;
  Buf<uint64_t> d_start;          // nbuckets + 1
  Buf<uint32_t> d_post, d_count, d_list;
  std::vector<uint32_t> h_count;
  std::vector<uint64_t> h_start;
  // per-batch scratch sets (at most VSX_KMER_SCRATCH_MAX), handed out under the lock
  std::vector<std::unique_ptr<KmerScratch>> scratch;
  std::mutex mu;
  std::condition_variable cv;
  std::mutex turn_mu;                                    // count_pass: the counting kernels of concurrent batches run one after the other
  hipEvent_t turn_ev = nullptr;                          // (borrowed from the scratch set that launched last)
  hipEvent_t e0 = nullptr, e1 = nullptr;                 // build timing
  VsxKmerStats stats;
  std::vector<uint64_t> word_total;   // postings per word over all tiles
  std::vector<uint64_t> word_units;   // packed: 16-byte units per word over all tiles
  bool own_stream = false;
  ~VsxKmerIndex()
  {
    if (e0) (void) hipEventDestroy(e0);
    if (e1) (void) hipEventDestroy(e1);
    if (own_stream && st) { (void) hipStreamSynchronize(st); (void) hipStreamDestroy(st); }
  }
  // the index works on a stream of its own: its kernels read the (already complete) sequence codes only, and a search
  // counts the candidates of window i+1 while the aligner's streams run window i (vsx_search_batch)
  void make_stream()
  {
    hipStream_t s2 = nullptr;
    if (hipSetDevice(device) == hipSuccess && hipStreamCreateWithFlags(&s2, hipStreamNonBlocking) == hipSuccess) { st = s2; own_stream = true; }
  }
};

#define KCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return kfail(e_ == hipErrorOutOfMemory ? VSX_ENOMEM : VSX_EHIP, #x, e_); } while (0)

int vsx_kmer_index_create(vsx_ctx * ctx, const vsx_seqset * db, int w, VsxKmerIndex ** out)
{
  if (!ctx || !db || !out) { vsx_internal_set_error("vsx_kmer_index_create: null argument"); return VSX_EINVAL; }
  *out = nullptr;
  if (w < 3 || w > 15) { vsx_internal_set_error("vsx_kmer_index_create: device index supports word lengths 3..15"); return VSX_EINVAL; }
  std::unique_ptr<VsxKmerIndex> ix(new VsxKmerIndex);
  ix->ctx = ctx; ix->db = db; ix->device = vsx_internal_device(ctx); ix->st = vsx_internal_stream(ctx); ix->w = w;
  ix->tagged = w > 8;
  // the whole-set index of a short word length takes the packed format (VSX_KMER_PACKED=0: the 16-bit format, A/B and tests)
  static const bool packed_off = std::getenv("VSX_KMER_PACKED") && std::strcmp(std::getenv("VSX_KMER_PACKED"), "0") == 0;
  ix->packed = !ix->tagged && !packed_off;
  ix->prewarm = true;
  ix->make_stream();
  KCHK(hipSetDevice(ix->device));
  KCHK(hipEventCreate(&ix->e0));
  KCHK(hipEventCreate(&ix->e1));
  const int rc = vsx_kmer_index_rebuild(ix.get(), nullptr, 0);
  if (rc != VSX_OK) return rc;
  *out = ix.release();
  return VSX_OK;
}

int vsx_kmer_index_create_empty(vsx_ctx * ctx, const vsx_seqset * db, int w, VsxKmerIndex ** out)
{
  if (!ctx || !db || !out) { vsx_internal_set_error("vsx_kmer_index_create_empty: null argument"); return VSX_EINVAL; }
  *out = nullptr;
  if (w < 3 || w > 8) { vsx_internal_set_error("vsx_kmer_index_create_empty: device index supports word lengths 3..8"); return VSX_EINVAL; }
  std::unique_ptr<VsxKmerIndex> ix(new VsxKmerIndex);
  ix->ctx = ctx; ix->db = db; ix->device = vsx_internal_device(ctx); ix->st = vsx_internal_stream(ctx); ix->w = w;
  ix->make_stream();
  KCHK(hipSetDevice(ix->device));
  KCHK(hipEventCreate(&ix->e0));
  KCHK(hipEventCreate(&ix->e1));
  *out = ix.release();
  return VSX_OK;
}

// (Re)build the index over the whole sequence set (list == nullptr) or over the listed sequences: index position i stands
// for sequence list[i].  Buffers are reused and only ever grow (clustering rebuilds once per round).
int vsx_kmer_index_rebuild(VsxKmerIndex * ix, const uint32_t * list, uint64_t n_list)
{
  if (!ix) { vsx_internal_set_error("vsx_kmer_index_rebuild: null index"); return VSX_EINVAL; }
  const uint8_t * codes; const uint64_t * off; const uint32_t * len; uint64_t n_all;
  vsx_internal_seqset_device(ix->db, &codes, &off, &len, &n_all);
  const uint64_t n = list ? n_list : n_all;
  if (n >= (1ull << 31)) { vsx_internal_set_error("vsx_kmer_index_rebuild: too many sequences"); return VSX_EINVAL; }
  KCHK(hipSetDevice(ix->device));
  const int w = ix->w;
  const uint32_t shift = vsx_kmer_tile_shift();
  if (ix->tagged && list) { vsx_internal_set_error("vsx_kmer_index_rebuild: subset indexes exist for word lengths 3..8 only"); return VSX_EINVAL; }
  if (list) ix->packed = false;                                                    // subset indexes (clustering) are rebuilt every round: two sweeps, no sort
  const uint64_t nwords = ix->tagged ? (1ull << 16) : (1ull << (2 * w));          // rows of the bucket table
  const uint64_t tile_seqs = ix->packed ? vsx_kmer_packed_tile_seqs() : (1ull << shift);
  ix->nseq = (uint32_t) n;
  ix->ntiles = std::max<uint32_t>(1, (uint32_t) ((n + tile_seqs - 1) / tile_seqs));
  ix->nbuckets = nwords * ix->ntiles;
  ix->word_total.assign(nwords, 0);
  ix->stats.postings = 0;
  if (n == 0) return VSX_OK;
  const uint32_t * d_list = nullptr;
  if (list)
    {
      KCHK(ix->d_list.ensure(n));
      KCHK(hipMemcpyAsync(ix->d_list.p, list, n * 4, hipMemcpyHostToDevice, ix->st));
      d_list = ix->d_list.p;
    }
  KCHK(ix->d_count.ensure(ix->nbuckets));
  KCHK(ix->d_start.ensure(ix->nbuckets + 1));
  KCHK(hipEventRecord(ix->e0, ix->st));
  KCHK(hipMemsetAsync(ix->d_count.p, 0, ix->nbuckets * 4, ix->st));
  // tagged build (word lengths 9..15): per tile keys -> sort -> runs; scratch for the largest tile
  // (packed build, word lengths 3..8: the same per-tile scheme with 32-bit keys, vsx_kmer_packed_tile)
  Buf<uint64_t> d_slot, d_keys_a, d_keys_b;                       // packed: the key buffers hold 32-bit keys (half used)
  Buf<uint8_t> d_sort_temp;
  std::vector<uint64_t> slot_of;
  size_t sort_bytes = 0;
  const bool sorted_build = ix->tagged || ix->packed;
  auto tagged_pass = [&](int fill) -> int {
    for (uint32_t t = 0; t < ix->ntiles; ++t)
      {
        const uint64_t first = (uint64_t) t * tile_seqs, last = std::min<uint64_t>(n, first + tile_seqs);
        if (ix->packed)
          KCHK(vsx_kmer_packed_tile(fill, codes, off, len, (uint32_t) first, (uint32_t) (last - first), w, vsx_internal_seqset_lower(ix->db), d_slot.p,
                                    slot_of[last] - slot_of[first], reinterpret_cast<uint32_t *>(d_keys_a.p), reinterpret_cast<uint32_t *>(d_keys_b.p),
                                    d_sort_temp.p, &sort_bytes, t, ix->ntiles, ix->d_count.p, ix->d_start.p, ix->d_post.p, ix->st));
        else
        KCHK(vsx_kmer_tagged_tile(fill, codes, off, len, (uint32_t) first, (uint32_t) (last - first), w, vsx_internal_seqset_lower(ix->db), d_slot.p,
                                  slot_of[last] - slot_of[first], d_keys_a.p, d_keys_b.p, d_sort_temp.p, &sort_bytes, t, ix->ntiles, ix->d_count.p,
                                  ix->d_start.p, ix->d_post.p, ix->st));
      }
    return VSX_OK;
  };
  if (sorted_build)
    {
      const uint32_t * hl = vsx_internal_seqset_host_lengths(ix->db);
      slot_of.assign(n + 1, 0);
      uint64_t widest = 0;
      for (uint64_t i = 0; i < n; ++i) slot_of[i + 1] = slot_of[i] + hl[i];
      for (uint32_t t = 0; t < ix->ntiles; ++t)
        {
          const uint64_t first = (uint64_t) t * tile_seqs, last = std::min<uint64_t>(n, first + tile_seqs);
          widest = std::max(widest, slot_of[last] - slot_of[first]);
        }
      KCHK(d_slot.alloc(n + 1));
      KCHK(hipMemcpyAsync(d_slot.p, slot_of.data(), (n + 1) * 8, hipMemcpyHostToDevice, ix->st));
      KCHK(d_keys_a.alloc(std::max<uint64_t>(widest, 1)));
      KCHK(d_keys_b.alloc(std::max<uint64_t>(widest, 1)));
      if (ix->packed)
        KCHK(vsx_kmer_packed_tile(0, nullptr, nullptr, nullptr, 0, 0, w, nullptr, nullptr, widest, reinterpret_cast<uint32_t *>(d_keys_a.p),
                                  reinterpret_cast<uint32_t *>(d_keys_b.p), nullptr, &sort_bytes, 0, ix->ntiles, nullptr, nullptr, nullptr, ix->st));
      else
      KCHK(vsx_kmer_tagged_tile(0, nullptr, nullptr, nullptr, 0, 0, w, nullptr, nullptr, widest, d_keys_a.p, d_keys_b.p, nullptr, &sort_bytes, 0, ix->ntiles,
                                nullptr, nullptr, nullptr, ix->st));
      KCHK(d_sort_temp.alloc(sort_bytes + 16));
      const int prc = tagged_pass(0);
      if (prc != VSX_OK) return prc;
    }
  else
  KCHK(vsx_kmer_launch_sweep(0, codes, off, len, d_list, ix->nseq, w, ix->ntiles, ix->d_count.p, nullptr, nullptr, vsx_internal_seqset_lower(ix->db), ix->st));
  // bucket table: exclusive prefix sum on the host (4^w x ntiles entries: 8 MB for 1 M sequences, w = 8)
  std::vector<uint32_t> & cnt = ix->h_count;
  cnt.resize(ix->nbuckets);
  KCHK(hipMemcpyAsync(cnt.data(), ix->d_count.p, ix->nbuckets * 4, hipMemcpyDeviceToHost, ix->st));
  KCHK(hipStreamSynchronize(ix->st));
  std::vector<uint64_t> & start = ix->h_start;
  start.resize(ix->nbuckets + 1);
  uint64_t acc = 0;
  // a bucket holds 16-bit tile-local indices in 16-byte units of eight (tagged: dwords tag << 16 | index, units of four):
  // start[] counts UNITS, the last unit of a bucket is padded with 0x8000 = the spare counter past the tile (vsx_kmer.hip KM_PAD;
  // tagged: 0xFFFF8000, a tag no word has): the count kernel streams units without any sentinel test
  // (packed: the walk has already counted units, cnt = units | postings << 16)
  const uint32_t per_unit = ix->tagged ? 4u : 8u;
  uint64_t entries = 0;
  if (ix->packed) ix->word_units.assign(nwords, 0);
  {
    // buckets are word-major: b = word * ntiles + tile (nested loops: no division per bucket -- clustering rebuilds the index
    // once per round)
    const uint64_t nt = ix->ntiles;
    uint64_t b = 0;
    for (uint64_t word = 0; word < nwords; ++word)
      {
        uint64_t tot = 0, units = 0;
        for (uint64_t t = 0; t < nt; ++t, ++b)
          {
            start[b] = acc;
            if (ix->packed) { acc += cnt[b] & 0xffffu; units += cnt[b] & 0xffffu; tot += cnt[b] >> 16; }
            else { acc += (cnt[b] + per_unit - 1) / per_unit; tot += cnt[b]; }
          }
        ix->word_total[word] = tot;
        if (ix->packed) ix->word_units[word] = units;
        entries += tot;
      }
  }
  start[ix->nbuckets] = acc;
  if (acc >= (1ull << 32)) { vsx_internal_set_error("vsx_kmer_index_rebuild: more than 64 GB of postings (32-bit unit addresses)"); return VSX_EINVAL; }
  KCHK(ix->d_post.ensure(acc * 4));                    // dwords
  if (acc && !ix->packed)                                  // (the packed walk writes whole units)
    KCHK(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(ix->d_post.p), (int) (ix->tagged ? 0xFFFF8000u : 0x80008000u), acc * 4, ix->st));
  KCHK(hipMemcpyAsync(ix->d_start.p, start.data(), (ix->nbuckets + 1) * 8, hipMemcpyHostToDevice, ix->st));
  KCHK(hipMemsetAsync(ix->d_count.p, 0, ix->nbuckets * 4, ix->st));
  if (sorted_build)
    {
      const int prc = tagged_pass(1);
      if (prc != VSX_OK) return prc;
    }
  else
  KCHK(vsx_kmer_launch_sweep(1, codes, off, len, d_list, ix->nseq, w, ix->ntiles, ix->d_count.p, ix->d_start.p, ix->d_post.p, vsx_internal_seqset_lower(ix->db), ix->st));
  KCHK(hipEventRecord(ix->e1, ix->st));
  KCHK(hipStreamSynchronize(ix->st));
  float ms = 0;
  KCHK(hipEventElapsedTime(&ms, ix->e0, ix->e1));
  ix->stats.build_ms = ms;
  ix->stats.postings = entries;
  ix->stats.index_bytes = acc * 16 + (ix->nbuckets + 1) * 8;
  return VSX_OK;
}

void vsx_kmer_index_destroy(VsxKmerIndex * ix)
{
  if (!ix) return;
  (void) hipSetDevice(ix->device);
  delete ix;
}

const VsxKmerStats * vsx_kmer_stats(const VsxKmerIndex * ix) { return ix ? &ix->stats : nullptr; }

namespace {

// one counting + selection pass over `nslots` query slots (slot -> query through qlist, or identity); appends to recs
// Slots [0, n8) hold queries of the 8-bit counter class (<= 255 unique words), the rest take the 16-bit kernel.
// subcap = records per (slot, tile) sub-region.
int count_pass(VsxKmerIndex * ix, KmerScratch * sc, uint32_t nslots, uint32_t n8, const uint32_t * d_qlist, const std::vector<uint32_t> * h_qlist, uint32_t subcap,
               uint32_t keep, VsxKmerResult & out, std::vector<uint32_t> & overflow, uint32_t & overflow_max, float & ms_total)
{
  const uint32_t nt = ix->ntiles;
  KCHK(sc->d_rec.ensure((size_t) nslots * nt * subcap));
  KCHK(sc->d_tilecnt.ensure((size_t) nslots * nt));
  KCHK(sc->d_sel_mn.ensure(nslots));
  KCHK(sc->d_sel_off.ensure(nslots));
  static const bool no_pre_env = std::getenv("VSX_KMER_NO_RANGES") != nullptr;   // A/B: the blocks look their ranges up themselves
  const bool no_pre = no_pre_env || ix->tagged;                                   // tagged indexes (word lengths 9..15) always do
  const int tg = ix->tagged ? 1 : (ix->packed ? 2 : 0);                           // the postings format (vsx_kmer_launch_count)
  if (n8 && !no_pre) KCHK(sc->d_ranges.ensure((size_t) n8 * nt * 256));
  KCHK(hipEventRecord(sc->e0, sc->st));
  if (n8 && !no_pre)
    KCHK(vsx_kmer_launch_ranges(ix->d_start.p, nt, sc->d_qk_start.p, sc->d_qk.p, sc->d_minmatch.p, d_qlist, n8, sc->d_ranges.p, sc->st));
  // Counting kernels of concurrent batches take turns in arrival order: each fills the device on its own, so two at once only
  // finish BOTH late (the search's windows then reach the aligner in pairs and its last stage starts later); uploads, ranges,
  // selection and downloads of the other batch still overlap.  VSX_KMER_TURNS=0: free-running (A/B).
  static const bool turns = !(std::getenv("VSX_KMER_TURNS") && std::strcmp(std::getenv("VSX_KMER_TURNS"), "0") == 0);
  std::unique_lock<std::mutex> turn(ix->turn_mu, std::defer_lock);
  if (turns)
    {
      turn.lock();
      if (ix->turn_ev) KCHK(hipStreamWaitEvent(sc->st, ix->turn_ev, 0));
    }
  KCHK(vsx_kmer_launch_count(8, tg, ix->d_post.p, ix->d_start.p, (n8 && !no_pre) ? sc->d_ranges.p : nullptr, nt, ix->nseq, n8, 0, sc->d_qk_start.p, sc->d_qk.p,
                             sc->d_minmatch.p, d_qlist, sc->d_rec.p, subcap, sc->d_tilecnt.p, sc->st));
  KCHK(vsx_kmer_launch_count(16, tg, ix->d_post.p, ix->d_start.p, nullptr, nt, ix->nseq, nslots - n8, n8, sc->d_qk_start.p, sc->d_qk.p,
                             sc->d_minmatch.p, d_qlist, sc->d_rec.p + (size_t) n8 * nt * subcap, subcap, sc->d_tilecnt.p + (size_t) n8 * nt, sc->st));
  if (turns)
    {
      KCHK(hipEventRecord(sc->e_turn, sc->st));
      ix->turn_ev = sc->e_turn;
      turn.unlock();
    }
  uint64_t capacity = std::max<uint64_t>(sc->d_dense.n, std::max<uint64_t>(1u << 20, (uint64_t) nslots * 128));
  unsigned long long produced = 0;
  for (int attempt = 0; attempt < 2; ++attempt)
    {
      KCHK(sc->d_dense.ensure(capacity));
      KCHK(hipMemsetAsync(sc->d_cursor.p, 0, sizeof(unsigned long long), sc->st));
      KCHK((ix->packed ? vsx_kmer_launch_select_packed : vsx_kmer_launch_select)(sc->d_rec.p, subcap, nt, sc->d_tilecnt.p, nslots, keep, sc->d_dense.p,
                                                                                  sc->d_cursor.p, sc->d_dense.n, sc->d_sel_mn.p, sc->d_sel_off.p, sc->st));
      KCHK(hipEventRecord(sc->e1, sc->st));
      KCHK(hipMemcpyAsync(&produced, sc->d_cursor.p, sizeof produced, hipMemcpyDeviceToHost, sc->st));
      KCHK(hipStreamSynchronize(sc->st));
      if (produced <= sc->d_dense.n) break;
      capacity = produced;
      if (attempt == 1) { vsx_internal_set_error("vsx_kmer_count_batch: selection buffer overflow"); return VSX_EHIP; }
    }
  float ms = 0;
  KCHK(hipEventElapsedTime(&ms, sc->e0, sc->e1));
  ms_total += ms;
  std::vector<uint64_t> mn(nslots), off(nslots);
  KCHK(hipMemcpy(mn.data(), sc->d_sel_mn.p, (size_t) nslots * 8, hipMemcpyDeviceToHost));
  KCHK(hipMemcpy(off.data(), sc->d_sel_off.p, (size_t) nslots * 8, hipMemcpyDeviceToHost));
  // the dense buffer is already grouped by slot: append it wholesale and record each query's range
  const size_t before = out.rec.size();
  out.rec.resize(before + produced);
  if (produced) KCHK(hipMemcpy(out.rec.data() + before, sc->d_dense.p, produced * 8, hipMemcpyDeviceToHost));
  for (uint32_t s = 0; s < nslots; ++s)
    {
      const uint32_t m = (uint32_t) (mn[s] & 0xffffffffu), n = (uint32_t) (mn[s] >> 32);
      const uint32_t q = h_qlist ? (*h_qlist)[s] : s;
      if (m == 0xffffffffu) { overflow.push_back(q); overflow_max = std::max(overflow_max, n); out.cnt[q] = 0; continue; }
      out.off[q] = before + off[s];
      out.cnt[q] = m;
    }
  sc->records += produced;
  return VSX_OK;
}

}  // namespace

#define VSX_KMER_SCRATCH_MAX 3

namespace {
std::unique_ptr<KmerScratch> new_scratch()
{
  std::unique_ptr<KmerScratch> p(new KmerScratch);
  int prio_low = 0, prio_high = 0;               // counting runs BEHIND the aligner's plans (vsx_host.cpp vsx_create)
  (void) hipDeviceGetStreamPriorityRange(&prio_low, &prio_high);
  if (hipStreamCreateWithPriority(&p->st, hipStreamNonBlocking, prio_low) != hipSuccess || hipEventCreate(&p->e0) != hipSuccess ||
      hipEventCreate(&p->e1) != hipSuccess || hipEventCreateWithFlags(&p->e_turn, hipEventDisableTiming) != hipSuccess ||
      p->d_cursor.alloc(1) != hipSuccess)
    return nullptr;
  return p;
}
struct ScratchLease {
  VsxKmerIndex * ix; KmerScratch * sc;
  ~ScratchLease() { if (sc) { { std::lock_guard<std::mutex> lk(ix->mu); sc->busy = false; } ix->cv.notify_all(); } }
};
}  // namespace

int vsx_kmer_count_batch(VsxKmerIndex * ix, uint64_t nq, const uint64_t * qk_start, const uint32_t * qk,
                         const uint32_t * minmatch, uint32_t keep, VsxKmerResult & out, uint32_t cap_hint, VsxKmerStats * stats_out, bool want_increments)
{
  out.rec.clear();
  out.off.assign(nq, 0);
  out.cnt.assign(nq, 0);
  if (stats_out) *stats_out = VsxKmerStats {};
  if (!ix || (nq && (!qk_start || !minmatch))) { vsx_internal_set_error("vsx_kmer_count_batch: null argument"); return VSX_EINVAL; }
  if (nq == 0 || ix->nseq == 0) return VSX_OK;
  if (nq >= (1ull << 22)) { vsx_internal_set_error("vsx_kmer_count_batch: at most 4 M queries per batch"); return VSX_EINVAL; }
  KCHK(hipSetDevice(ix->device));
  // a free scratch set, or a new one, or wait for one
  ScratchLease lease {ix, nullptr};
  {
    std::unique_lock<std::mutex> lk(ix->mu);
    for (;;)
      {
        for (auto & p : ix->scratch) if (!p->busy) { lease.sc = p.get(); break; }
        if (lease.sc) break;
        if (ix->scratch.size() < VSX_KMER_SCRATCH_MAX)
          {
            std::unique_ptr<KmerScratch> p = new_scratch();
            if (!p) { (void) hipGetLastError(); vsx_internal_set_error("vsx_kmer_count_batch: scratch allocation failed"); return VSX_EHIP; }
            lease.sc = p.get();
            ix->scratch.push_back(std::move(p));
            break;
          }
        ix->cv.wait(lk);
      }
    lease.sc->busy = true;
  }
  KmerScratch * sc = lease.sc;
  const uint64_t nk = qk_start[nq];
  uint64_t increments = 0, streamed_bytes = 0;
  if (want_increments)
    {
      for (uint64_t x = 0; x < nk; ++x) increments += ix->word_total[ix->tagged ? (qk[x] & 0xffffu) : qk[x]];      // postings streamed
      if (ix->packed) for (uint64_t x = 0; x < nk; ++x) streamed_bytes += 16 * ix->word_units[qk[x]];
      else streamed_bytes = increments * (ix->tagged ? 4 : 2);
    }
  sc->records = 0;
  KCHK(sc->d_qk_start.ensure(nq + 1));
  KCHK(sc->d_qk.ensure(nk));
  KCHK(sc->d_minmatch.ensure(nq));
  KCHK(hipMemcpyAsync(sc->d_qk_start.p, qk_start, (nq + 1) * 8, hipMemcpyHostToDevice, sc->st));
  if (nk) KCHK(hipMemcpyAsync(sc->d_qk.p, qk, nk * 4, hipMemcpyHostToDevice, sc->st));
  KCHK(hipMemcpyAsync(sc->d_minmatch.p, minmatch, nq * 4, hipMemcpyHostToDevice, sc->st));
  float ms = 0;
  std::vector<uint32_t> overflow;
  uint32_t overflow_max = 0;
  // records per query region in the first pass: at 1 M x 1 kbp a 250-bp query has ~5 000 sequences with >= 12 shared
  // words (chance runs of shared 11..13-mers), so 8 192 x 8 B = 64 KB per query; HBM is plentiful (100 k queries = 6.5 GB)
  // (r03) every (query, tile) owns a sub-region: 512 records where a tile is one of many (a 250-bp query finds ~160 per tile of
  // 32 768 sequences), the whole 8 192 for single-tile databases (clustering rounds)
  uint32_t cap = cap_hint ? cap_hint : (nq <= (1u << 18) ? 8192 : 2048);
  bool forced = cap_hint != 0;
  if (const char * c = std::getenv("VSX_KMER_CAP")) { cap = (uint32_t) std::max(1, std::atoi(c)); forced = true; }      // tests: force the second pass
  cap = std::max<uint32_t>(1, forced ? cap / ix->ntiles : std::max<uint32_t>(512, cap / ix->ntiles));
  // counter class per query: at most 255 unique words -> byte counters (a count never exceeds the number of words)
  auto by_class = [&](const std::vector<uint32_t> * subset, std::vector<uint32_t> & ordered, uint32_t & n8) -> bool {
    const uint64_t n = subset ? subset->size() : nq;
    n8 = 0;
    bool all8 = true;
    for (uint64_t x = 0; x < n && all8; ++x)
      {
        const uint32_t q = subset ? (*subset)[x] : (uint32_t) x;
        all8 = (qk_start[q + 1] - qk_start[q]) <= 255;
      }
    if (all8) { n8 = (uint32_t) n; return false; }              // no reordering needed
    ordered.clear();
    ordered.reserve(n);
    for (int pass = 0; pass < 2; ++pass)
      {
        for (uint64_t x = 0; x < n; ++x)
          {
            const uint32_t q = subset ? (*subset)[x] : (uint32_t) x;
            const bool small = (qk_start[q + 1] - qk_start[q]) <= 255;
            if (small == (pass == 0)) ordered.push_back(q);
          }
        if (pass == 0) n8 = (uint32_t) ordered.size();
      }
    return true;
  };
  // One pass over a set of queries (NULL = the whole batch in its own order): class split, slot list, counting + selection.
  auto run_subset = [&](const std::vector<uint32_t> * subset, uint32_t subcap, std::vector<uint32_t> & over, uint32_t & over_max) -> int {
    std::vector<uint32_t> ordered;
    uint32_t n8 = 0;
    Buf<uint32_t> d_list;
    const uint32_t n = subset ? (uint32_t) subset->size() : (uint32_t) nq;
    const std::vector<uint32_t> * list = subset;
    if (by_class(subset, ordered, n8)) list = &ordered;
    if (list)
      {
        KCHK(d_list.alloc(list->size()));
        KCHK(hipMemcpyAsync(d_list.p, list->data(), list->size() * 4, hipMemcpyHostToDevice, sc->st));
      }
    return count_pass(ix, sc, n, n8, list ? d_list.p : nullptr, list, subcap, keep, out, over, over_max, ms);
  };
  // the per-(query, tile) record sub-regions and range tables grow with queries x tiles -- 10 M sequences are
  // ~307 tiles, i.e. ~30 GB per 16 k-query window and scratch set.  The scratch of one pass is bounded instead: a batch whose
  // regions would exceed the budget runs as several passes over slices of its queries through the same buffers (results are per
  // query, so the slicing is invisible); VSX_KMER_SCRATCH_BYTES overrides the budget (tests force the sliced path with it).
  const uint64_t scratch_budget = []() -> uint64_t {
    if (const char * c = std::getenv("VSX_KMER_SCRATCH_BYTES")) return (uint64_t) std::max<long long>(1 << 16, std::atoll(c));
    return 6ull << 30;
  }();
  auto run_sliced = [&](const std::vector<uint32_t> * subset, uint32_t subcap, std::vector<uint32_t> & over, uint32_t & over_max) -> int {
    const uint64_t n = subset ? subset->size() : nq;
    const uint64_t per_slot = (uint64_t) ix->ntiles * ((uint64_t) subcap * 8 + 256 * 8 + 4);
    const uint64_t gmax = std::max<uint64_t>(64, scratch_budget / per_slot);
    if (n <= gmax) return run_subset(subset, subcap, over, over_max);
    std::vector<uint32_t> slice;
    for (uint64_t b = 0; b < n; b += gmax)
      {
        const uint64_t e = std::min(n, b + gmax);
        slice.clear();
        for (uint64_t x = b; x < e; ++x) slice.push_back(subset ? (*subset)[x] : (uint32_t) x);
        const int r = run_subset(&slice, subcap, over, over_max);
        if (r != VSX_OK) return r;
      }
    return VSX_OK;
  };
  int rc = run_sliced(nullptr, cap, overflow, overflow_max);
  if (rc != VSX_OK) return rc;
  if (std::getenv("VSX_KMER_DEBUG")) std::fprintf(stderr, "kmer: %zu of %llu queries overflowed cap %u (max %u records), %.1f ms\n", overflow.size(), (unsigned long long) nq, cap, overflow_max, ms);
  if (!overflow.empty())
    {
      // queries with more than `cap` sequences at or above their threshold (low-complexity words): a second pass over
      // just these, with regions of the size the first pass measured
      std::vector<uint32_t> again;
      uint32_t again_max = 0;
      rc = run_sliced(&overflow, overflow_max, again, again_max);
      if (rc != VSX_OK) return rc;
      if (!again.empty()) { vsx_internal_set_error("vsx_kmer_count_batch: record region overflow in the second pass"); return VSX_EHIP; }
    }
  // A search index (vsx_kmer_index_create) is counted against by up to three windows at once.  The further scratch sets are made
  // HERE, behind the first batch and with its buffer sizes: created on demand, the first call that happens to overlap three
  // windows paid ~0.4 s of hipMalloc in the middle of a warm search (one in seven calls of the bench took 0.55 s instead of 0.15)
  // (... unless a spare set of this size would eat into the memory the aligner contexts need -- large databases keep ONE set)
  auto room_for_spare = [&]() -> bool {
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { (void) hipGetLastError(); return false; }
    const uint64_t set_bytes = (uint64_t) sc->d_rec.n * 8 + (uint64_t) sc->d_ranges.n * 8 + (uint64_t) sc->d_dense.n * 8 + (uint64_t) sc->d_qk.n * 4;
    return (uint64_t) free_b > 2 * set_bytes + (uint64_t) total_b / 4;
  };
  if (ix->prewarm && room_for_spare())
    {
      // ... and an idle set that has not met a batch of this size yet grows now rather than in the middle of a later call
      auto grow_like = [&](KmerScratch * o) -> bool {
        return o->d_qk_start.ensure(sc->d_qk_start.n) == hipSuccess && o->d_qk.ensure(sc->d_qk.n) == hipSuccess && o->d_minmatch.ensure(sc->d_minmatch.n) == hipSuccess &&
               o->d_rec.ensure(sc->d_rec.n) == hipSuccess && o->d_dense.ensure(sc->d_dense.n) == hipSuccess && o->d_sel_mn.ensure(sc->d_sel_mn.n) == hipSuccess &&
               o->d_sel_off.ensure(sc->d_sel_off.n) == hipSuccess && o->d_ranges.ensure(sc->d_ranges.n) == hipSuccess && o->d_tilecnt.ensure(sc->d_tilecnt.n) == hipSuccess;
      };
      for (;;)
        {
          KmerScratch * idle = nullptr;
          {
            std::lock_guard<std::mutex> lk(ix->mu);
            for (auto & q : ix->scratch)
              if (!q->busy && q.get() != sc && (q->d_rec.n < sc->d_rec.n || q->d_qk.n < sc->d_qk.n || q->d_ranges.n < sc->d_ranges.n || q->d_dense.n < sc->d_dense.n))
                { idle = q.get(); idle->busy = true; break; }
          }
          if (!idle) break;
          const bool ok = grow_like(idle);
          if (!ok) (void) hipGetLastError();
          { std::lock_guard<std::mutex> lk(ix->mu); idle->busy = false; }
          ix->cv.notify_all();
          if (!ok) break;
        }
      bool more = false;
      { std::lock_guard<std::mutex> lk(ix->mu); more = ix->scratch.size() < VSX_KMER_SCRATCH_MAX; }
      while (more)
        {
          std::unique_ptr<KmerScratch> p = new_scratch();
          if (!p) { (void) hipGetLastError(); break; }
          if (p->d_qk_start.ensure(sc->d_qk_start.n) != hipSuccess || p->d_qk.ensure(sc->d_qk.n) != hipSuccess || p->d_minmatch.ensure(sc->d_minmatch.n) != hipSuccess ||
              p->d_rec.ensure(sc->d_rec.n) != hipSuccess || p->d_dense.ensure(sc->d_dense.n) != hipSuccess || p->d_sel_mn.ensure(sc->d_sel_mn.n) != hipSuccess ||
              p->d_sel_off.ensure(sc->d_sel_off.n) != hipSuccess || p->d_ranges.ensure(sc->d_ranges.n) != hipSuccess || p->d_tilecnt.ensure(sc->d_tilecnt.n) != hipSuccess)
            { (void) hipGetLastError(); break; }                   // (not enough memory for a spare set: the batches will share)
          std::lock_guard<std::mutex> lk(ix->mu);
          if (ix->scratch.size() < VSX_KMER_SCRATCH_MAX) ix->scratch.push_back(std::move(p));
          more = ix->scratch.size() < VSX_KMER_SCRATCH_MAX;
          ix->cv.notify_all();
        }
    }
  {
    std::lock_guard<std::mutex> lk(ix->mu);
    ix->stats.count_ms = ms; ix->stats.increments = increments; ix->stats.streamed_bytes = streamed_bytes; ix->stats.records = sc->records;
    if (stats_out) { *stats_out = ix->stats; }
  }
  return VSX_OK;
}

// ---- the packed postings format on the host: what the CPU suite drives (tests/test_host_cpu.py) ------------------------------------
// counters[n] ascending, none of them a dummy -> units (units_out may be NULL: only the count); returns the number of units, or
// -1 for an input the format does not take
extern "C" int64_t vsx_internal_kmer_pack_encode(const uint32_t * counters, uint64_t n, uint32_t * units_out, uint64_t units_cap)
{
  for (uint64_t k = 0; k < n; ++k)
    if (counters[k] >= KM_PK_ROWS * KM_PK_PERIOD || km_pk_is_dummy(counters[k]) || (k && counters[k] <= counters[k - 1])) return -1;
  KmPkEncoder count(nullptr);
  for (uint64_t k = 0; k < n; ++k) count.push(counters[k]);
  count.finish();
  if (!units_out) return (int64_t) count.units;
  if (count.units > units_cap) return -1;
  KmPkEncoder enc(reinterpret_cast<KmPkUnit *>(units_out));
  for (uint64_t k = 0; k < n; ++k) enc.push(counters[k]);
  enc.finish();
  return (int64_t) enc.units;
}
// what the count kernel does with a bucket: every byte of every unit is one increment; hits[counter] += 1 (hits has 32 768 entries)
extern "C" void vsx_internal_kmer_pack_count(const uint32_t * units, uint64_t n_units, uint32_t * hits)
{
  for (uint64_t u = 0; u < n_units; ++u)
    {
      const uint32_t * w = units + 4 * u;
      uint32_t acc = w[0] & 0xffffu;
      ++hits[acc & 0x7fffu];
      for (int slot = 1; slot < KM_PK_SLOTS; ++slot)
        {
          const int bytepos = slot + 1;
          acc += (w[bytepos >> 2] >> (8 * (bytepos & 3))) & 0xffu;
          ++hits[acc & 0x7fffu];
        }
    }
}
extern "C" uint32_t vsx_internal_kmer_pack_counter_of(uint32_t local_seq) { return km_pk_counter_of(local_seq); }
extern "C" uint32_t vsx_internal_kmer_pack_seq_of(uint32_t counter) { return km_pk_seq_of(counter); }
