// vsx_mask.cpp -- DUST low-complexity masking on the host, as the reference applies it to the database and to every query
// BEFORE the k-mer stage of the path (commands/usearch_global.cpp:386-392,577-583; core/search.cpp:294-303; the default of
// --qmask and --dbmask is "dust").  Restated from core/mask.cpp:79-199:
//
//   the sequence is cut into windows of 64 symbols that advance by 32; in every window the best-scoring sub-interval
//   [first, last] is looked for among all start offsets i and end offsets j: score = 10 * (number of pairs of equal 3-mers
//   inside the interval, counted incrementally) / j (integer division); the FIRST interval reaching the maximum wins
//   (i ascending, then j ascending).  A window whose best score exceeds 20 has that interval masked (lower case; everything
//   else is upper case), and when the interval ends inside the window's first half the next window starts right behind it
//   plus one half window (the loop's own += 32 still applies).
//
// The result is TEXT with the reference's case convention, so the rest of the library treats a dust-masked set exactly like
// a soft-masked one (vsx_search_opts.soft_mask).  The device version (vsx_mask.hip) produces the same intervals as bits.
#include "../../include/vsx_search.h"
#include "vsx_internal.h"

#include <algorithm>
#include <atomic>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

extern "C" void vsx_internal_set_error(const char * msg);
extern "C" int vsx_internal_usable_cpus(void);

namespace {

constexpr int kWindow = 64, kHalf = 32, kLevel = 20;

inline unsigned two_bit(unsigned char c)        // map_2bit, utils/maps.cpp:156-186: A0 C1 G2 T3 U3, either case; everything else 0
{
  switch (c | 0x20u)
    {
    case 'c': return 1;
    case 'g': return 2;
    case 't': case 'u': return 3;
    default: return 0;
    }
}

// ceil(2^32 / j): x / j == (x * kMagic[j]) >> 32 for x < 2^32 / 63 (x <= 18 910 here; checked exhaustively in the tests' golden run)
struct Magic {
  uint32_t m[kWindow];
  Magic() { m[0] = m[1] = 0; for (uint32_t j = 2; j < (uint32_t) kWindow; ++j) m[j] = (uint32_t) (0x100000000ull / j) + ((j & (j - 1)) ? 1u : 0u); }
};
const Magic kMagic;

// best interval of one window of n <= 64 symbols; returns its score (0: none)
int window_best(const char * s, int n, int & first, int & last)
{
  first = last = 0;
  const int starts = n - 7;                     // the smallest region has 8 symbols (mask.cpp:84)
  if (starts <= 0) return 0;
  unsigned char tri[kWindow];                   // 3-mer ending at each position (the first two entries are partial and never read)
  unsigned acc = 0;
  for (int j = 0; j < n; ++j) { acc = (acc << 2) | two_bit((unsigned char) s[j]); tri[j] = (unsigned char) (acc & 63u); }
  // Only windows with a score ABOVE the level matter (dust_one masks nothing otherwise), and 10 pairs / j > 20 needs pairs >= 2.1 j
  // while no interval holds more pairs of equal 3-mers than the whole window does: end offsets beyond 10 total / 21 cannot exceed
  // the level, so they can neither be the answer nor change it.  One counting pass bounds the search; windows of ordinary sequence
  // (total ~ 30) walk a fifth of the end offsets, low-complexity windows all of them.
  int jlim;
  {
    unsigned char cnt[64];
    std::memset(cnt, 0, sizeof cnt);
    int total = 0;
    for (int j = 2; j < n; ++j) total += cnt[tri[j]]++;
    jlim = 10 * total / (kLevel + 1) + 1;       // exclusive
    if (jlim <= 2) return 0;
  }
  int best = 0, bi = 0, bj = 0;
  for (int i = 0; i < starts; ++i)
    {
      unsigned char seen[64];
      std::memset(seen, 0, sizeof seen);
      unsigned pairs = 0;
      const unsigned char * t = tri + i;
      const int m = std::min(n - i, jlim);
      for (int j = 2; j < m; ++j)
        {
          unsigned char & c = seen[t[j]];
          if (c)
            {
              pairs += c;
              const int v = (int) (((uint64_t) (10u * pairs) * kMagic.m[j]) >> 32);
              if (v > best) { best = v; bi = i; bj = j; }
            }
          ++c;
        }
    }
  first = bi;
  last = bi + bj;
  return best;
}

// hard (--hardmask, core/mask.cpp:137-191 with use_hardmask): the text keeps its case and every symbol of a masked interval becomes 'N'
void dust_one(char * seq, int64_t len, std::vector<char> & orig, bool hard = false)
{
  orig.assign(seq, seq + len);
  if (!hard)
    for (int64_t i = 0; i < len; ++i)
      {
        const unsigned char c = (unsigned char) seq[i];
        if (c >= 'a' && c <= 'z') seq[i] = (char) (c - 32);
      }
  for (int64_t i = 0; i < len; i += kHalf)
    {
      const int n = (int) (len > i + kWindow ? kWindow : len - i);
      int a = 0, b = 0;
      if (window_best(orig.data() + i, n, a, b) > kLevel)
        {
          if (hard) for (int64_t j = i + a; j <= i + b; ++j) seq[j] = 'N';
          else for (int64_t j = i + a; j <= i + b; ++j) seq[j] = (char) ((unsigned char) orig[(size_t) j] | 0x20u);
          if (b < kHalf) i += kHalf - b;
        }
    }
}

}  // namespace

extern "C" {

int vsx_dust_mask(char * blob, uint64_t n, const uint64_t * offsets, const uint32_t * lengths, int32_t threads)
{
  if (n && (!blob || !offsets || !lengths)) { vsx_internal_set_error("vsx_dust_mask: null argument"); return VSX_EINVAL; }
  int nth = threads > 0 ? threads : vsx_internal_usable_cpus();
  if ((uint64_t) nth > n / 64 + 1) nth = (int) (n / 64 + 1);
  std::atomic<uint64_t> next {0};
  auto work = [&]() {
    std::vector<char> orig;
    for (;;)
      {
        const uint64_t k0 = next.fetch_add(64);
        if (k0 >= n) break;
        for (uint64_t k = k0; k < std::min<uint64_t>(n, k0 + 64); ++k) dust_one(blob + offsets[k], (int64_t) lengths[k], orig);
      }
  };
  std::vector<std::thread> pool;
  for (int t = 1; t < nth; ++t) pool.emplace_back(work);
  work();
  for (auto & th : pool) th.join();
  return VSX_OK;
}

}  // extern "C"

// one sequence, for the dispatch layer's per-query masking (the caller owns the scratch copy)
void vsx_internal_dust_one(char * seq, int64_t len, std::vector<char> & scratch, bool hard) { dust_one(seq, len, scratch, hard); }
