// vsx_lma.cpp -- the scalar fallback for pairs the 16-bit aligner refuses (score == SHRT_MAX sentinel).
//
// In the reference the CALLER owns this step: LinearMemoryAligner::align + alignstats
// (src/core/linmemalign.cpp:311-808, call sites core/searchcore.cpp:806-832, commands/allpairs_global.cpp:447-473).
// It stays on the host CPU here as well (rare path, int64 arithmetic, linear memory).
// This is a restatement of the reference's divide-and-conquer (Hirschberg / Myers-Miller with 12
// position-specific gap penalties); every tie-break of the reference is kept so that CIGARs are identical:
//   * midpoint row I = a_len / 2; forward pass over the upper half, reverse pass over the lower half
//   * join type 0 (diagonal at the break) beats type 1 (gap in b across the break) on a strictly larger score;
//     equal scores: the smaller column wins, type 0 on equal columns
//   * within a type the FIRST maximal column wins
//   * a_len == 1: candidates in the order "D then I", "I then D", then the substitution columns left to right,
//     replaced only on a strictly larger score
#include "../../include/vsx_search.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <string>
#include <vector>

namespace {

inline unsigned map4(unsigned char c)
{
  switch (c)
    {
    case 'A': case 'a': return 1;  case 'B': case 'b': return 14; case 'C': case 'c': return 2;  case 'D': case 'd': return 13;
    case 'G': case 'g': return 4;  case 'H': case 'h': return 11; case 'K': case 'k': return 12; case 'M': case 'm': return 3;
    case 'N': case 'n': return 15; case 'R': case 'r': return 5;  case 'S': case 's': return 6;
    case 'T': case 't': case 'U': case 'u': return 8;
    case 'V': case 'v': return 7;  case 'W': case 'w': return 9;  case 'Y': case 'y': return 10;
    default: return 0;
    }
}

constexpr int64_t NEG = std::numeric_limits<int64_t>::min();

struct Lma {
  int64_t goql, gotl, goqi, goti, goqr, gotr, geql, getl, geqi, geti, geqr, getr;
  bool nmm;
  int64_t S[16][16];
  const char * a = nullptr;      // query
  const char * b = nullptr;      // target
  std::vector<int64_t> HH, EE, XX, YY;
  std::string cigar;
  char op = 0;
  int64_t run = 0;

  explicit Lma(const vsx_scoring & sc)
    : goql(sc.gap_open_query_left), gotl(sc.gap_open_target_left), goqi(sc.gap_open_query_interior),
      goti(sc.gap_open_target_interior), goqr(sc.gap_open_query_right), gotr(sc.gap_open_target_right),
      geql(sc.gap_ext_query_left), getl(sc.gap_ext_target_left), geqi(sc.gap_ext_query_interior),
      geti(sc.gap_ext_target_interior), geqr(sc.gap_ext_query_right), getr(sc.gap_ext_target_right),
      nmm(sc.n_mismatch != 0)
  {
    // scorematrix_fill, linmemalign.cpp:189-222
    auto amb = [](unsigned x) { return !(x == 1 || x == 2 || x == 4 || x == 8); };
    for (unsigned r = 0; r < 16; ++r)
      for (unsigned c = 0; c < 16; ++c)
        S[r][c] = (amb(r) || amb(c)) ? 0 : (r == c ? sc.match : sc.mismatch);
    if (nmm)
      for (unsigned k = 0; k < 16; ++k) { S[k][15] = sc.mismatch; S[15][k] = sc.mismatch; }
  }

  int64_t sub(char x, char y) const { return S[map4((unsigned char) y)][map4((unsigned char) x)]; }

  void flush()
  {
    if (run <= 0) return;
    if (run > 1) cigar += std::to_string(run);
    cigar.push_back(op);
  }
  void add(char o, int64_t n)
  {
    if (op == o) { run += n; return; }
    flush();
    op = o;
    run = n;
  }

  void diff(int64_t a0, int64_t b0, int64_t al, int64_t bl, bool gap_b_left, bool gap_b_right,
            bool a_left, bool a_right, bool b_left, bool b_right)
  {
    if (bl == 0) { if (al > 0) add('D', al); return; }
    if (al == 0) { add('I', bl); return; }
    if (al == 1)
      {
        // one query symbol against bl target symbols (linmemalign.cpp:338-451)
        int64_t score = 0;
        if (!gap_b_left) score -= b_left ? gotl : goti;
        score -= b_left ? getl : geti;
        score -= a_right ? goqr + bl * geqr : goqi + bl * geqi;
        int64_t best_score = score;
        int64_t best = -1;                                   // "D then I"
        // NB: the reference keeps accumulating into the same variable for the second candidate
        score -= a_left ? goql + bl * geql : goqi + bl * geqi;
        if (!gap_b_right) score -= b_right ? gotr : goti;
        score -= b_right ? getr : geti;
        if (score > best_score) { best_score = score; best = bl; }   // "I then D"
        for (int64_t i = 0; i < bl; ++i)
          {
            int64_t s = 0;
            if (i > 0) s -= a_left ? goql + i * geql : goqi + i * geqi;
            s += sub(a[a0], b[b0 + i]);
            if (i < bl - 1) s -= a_right ? goqr + (bl - 1 - i) * geqr : goqi + (bl - 1 - i) * geqi;
            if (s > best_score) { best_score = s; best = i; }
          }
        if (best == -1) { add('D', 1); add('I', bl); }
        else if (best == bl) { add('I', bl); add('D', 1); }
        else
          {
            if (best > 0) add('I', best);
            add('M', 1);
            if (best < bl - 1) add('I', bl - 1 - best);
          }
        return;
      }

    const int64_t I = al / 2;
    // forward pass over rows 1..I (linmemalign.cpp:463-513)
    HH[0] = 0; EE[0] = 0;
    for (int64_t j = 1; j <= bl; ++j) { HH[j] = -(a_left ? goql + j * geql : goqi + j * geqi); EE[j] = NEG; }
    for (int64_t i = 1; i <= I; ++i)
      {
        int64_t p = HH[0];
        int64_t h = -(b_left ? (gap_b_left ? 0 : gotl) + i * getl : (gap_b_left ? 0 : goti) + i * geti);
        HH[0] = h;
        int64_t f = NEG;
        for (int64_t j = 1; j <= bl; ++j)
          {
            f = std::max(f, h - goqi) - geqi;
            if (b_right && j == bl) EE[j] = std::max(EE[j], HH[j] - gotr) - getr;
            else EE[j] = std::max(EE[j], HH[j] - goti) - geti;
            h = p + sub(a[a0 + i - 1], b[b0 + j - 1]);
            h = std::max(f, h);
            h = std::max(EE[j], h);
            p = HH[j];
            HH[j] = h;
          }
      }
    EE[0] = HH[0];
    // reverse pass over the lower half (:515-569)
    XX[0] = 0; YY[0] = 0;
    for (int64_t j = 1; j <= bl; ++j) { XX[j] = -(a_right ? goqr + j * geqr : goqi + j * geqi); YY[j] = NEG; }
    for (int64_t i = 1; i <= al - I; ++i)
      {
        int64_t p = XX[0];
        int64_t h = -(b_right ? (gap_b_right ? 0 : gotr) + i * getr : (gap_b_right ? 0 : goti) + i * geti);
        XX[0] = h;
        int64_t f = NEG;
        for (int64_t j = 1; j <= bl; ++j)
          {
            f = std::max(f, h - goqi) - geqi;
            if (b_left && j == bl) YY[j] = std::max(YY[j], XX[j] - gotl) - getl;
            else YY[j] = std::max(YY[j], XX[j] - goti) - geti;
            h = p + sub(a[a0 + al - i], b[b0 + bl - j]);
            h = std::max(f, h);
            h = std::max(YY[j], h);
            p = XX[j];
            XX[j] = h;
          }
      }
    YY[0] = XX[0];
    // best join along the division line (:572-652)
    int64_t m0 = NEG, j0 = -1;
    for (int64_t j = 0; j <= bl; ++j)
      {
        const int64_t s = HH[j] + XX[bl - j];
        if (s > m0) { m0 = s; j0 = j; }
      }
    int64_t m1 = NEG, j1 = -1;
    for (int64_t j = 0; j <= bl; ++j)
      {
        const int64_t g = (b_left && j == 0) ? gotl : ((b_right && j == bl) ? gotr : goti);
        const int64_t s = EE[j] + YY[bl - j] + g;
        if (s > m1) { m1 = s; j1 = j; }
      }
    bool split;      // true: a gap in b spans the division line (two 'D' columns emitted here)
    int64_t best;
    if (m0 > m1) { split = false; best = j0; }
    else if (m1 > m0) { split = true; best = j1; }
    else if (j0 <= j1) { split = false; best = j0; }
    else { split = true; best = j1; }
    if (!split)
      {
        diff(a0, b0, I, best, gap_b_left, false, a_left, false, b_left, b_right && best == bl);
        diff(a0 + I, b0 + best, al - I, bl - best, false, gap_b_right, false, a_right, b_left && best == 0, b_right);
      }
    else
      {
        diff(a0, b0, I - 1, best, gap_b_left, true, a_left, false, b_left, b_right && best == bl);
        add('D', 2);
        diff(a0 + I + 1, b0 + best, al - I - 1, bl - best, true, gap_b_right, false, a_right, b_left && best == 0, b_right);
      }
  }

  void align(const char * q, int64_t ql, const char * t, int64_t tl)
  {
    a = q; b = t;
    cigar.clear(); op = 0; run = 0;
    HH.assign((size_t) tl + 1, 0); EE.assign((size_t) tl + 1, 0);
    XX.assign((size_t) tl + 1, 0); YY.assign((size_t) tl + 1, 0);
    diff(0, 0, ql, tl, false, false, true, true, true, true);
    flush();
  }

  // alignstats, linmemalign.cpp:722-808
  void stats(const char * q, const char * t, int64_t * score, int64_t * alnlen, int64_t * matches,
             int64_t * mismatches, int64_t * gaps) const
  {
    int64_t sc = 0, al = 0, ma = 0, mi = 0, ga = 0, ap = 0, bp = 0;
    const char * p = cigar.c_str();
    while (*p)
      {
        long long n = 1; int scan = 0;
        std::sscanf(p, "%lld%n", &n, &scan);
        p += scan;
        const char o = *p++;
        if (o == 'M')
          {
            al += n;
            for (long long k = 0; k < n; ++k)
              {
                const char x = q[ap], y = t[bp];
                sc += sub(x, y);
                const unsigned cx = map4((unsigned char) x), cy = map4((unsigned char) y);
                if (nmm && (cx == 15 || cy == 15)) ++mi;
                else if (cx & cy) ++ma;
                else ++mi;
                ++ap; ++bp;
              }
          }
        else if (o == 'I')
          {
            const int64_t g = (ap == 0 && bp == 0) ? goql + n * geql : (*p == 0 ? goqr + n * geqr : goqi + n * geqi);
            sc -= g; ++ga; al += n; bp += n;
          }
        else if (o == 'D')
          {
            const int64_t g = (ap == 0 && bp == 0) ? gotl + n * getl : (*p == 0 ? gotr + n * getr : goti + n * geti);
            sc -= g; ++ga; al += n; ap += n;
          }
      }
    *score = sc; *alnlen = al; *matches = ma; *mismatches = mi; *gaps = ga;
  }
};

}  // namespace

extern "C" int vsx_lma_align(const vsx_scoring * scoring, const char * q, uint64_t qlen, const char * t, uint64_t tlen,
                             int64_t * score, int64_t * alnlen, int64_t * matches, int64_t * mismatches,
                             int64_t * gaps, char ** cigar)
{
  if (!scoring || !cigar || (qlen && !q) || (tlen && !t)) return VSX_EINVAL;
  Lma lma(*scoring);
  lma.align(q, (int64_t) qlen, t, (int64_t) tlen);
  int64_t s, l, m, x, g;
  lma.stats(q, t, &s, &l, &m, &x, &g);
  if (score) *score = s;
  if (alnlen) *alnlen = l;
  if (matches) *matches = m;
  if (mismatches) *mismatches = x;
  if (gaps) *gaps = g;
  *cigar = (char *) std::malloc(lma.cigar.size() + 1);
  if (!*cigar) return VSX_ENOMEM;
  std::memcpy(*cigar, lma.cigar.c_str(), lma.cigar.size() + 1);
  return VSX_OK;
}
