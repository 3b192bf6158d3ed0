/*
  oracle/nw_oracle.c -- TEST INFRASTRUCTURE ONLY (see nw_oracle.h).

  Scalar restatement of the reference's 16-bit saturating Needleman-Wunsch
  (reference: src/core/align_simd.cpp).  The reference runs 8 targets per SSE2
  vector; every lane is independent, so one pair is a
  pure function of (query, target, 14 penalties, n_mismatch).  This file is that
  function, written cell by cell; nothing here is tuned for speed.

  Each block cites the reference lines it restates.
*/
#include "nw_oracle.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* utils/maps.cpp:75-117 (chrmap_4bit): A1 C2 G4 T/U8, IUPAC sets, others 0 */
unsigned char vsxo_map4(unsigned char c)
{
  switch (c)
    {
    case 'A': case 'a': return 1;
    case 'B': case 'b': return 14;
    case 'C': case 'c': return 2;
    case 'D': case 'd': return 13;
    case 'G': case 'g': return 4;
    case 'H': case 'h': return 11;
    case 'K': case 'k': return 12;
    case 'M': case 'm': return 3;
    case 'N': case 'n': return 15;
    case 'R': case 'r': return 5;
    case 'S': case 's': return 6;
    case 'T': case 't': case 'U': case 'u': return 8;
    case 'V': case 'v': return 7;
    case 'W': case 'w': return 9;
    case 'Y': case 'y': return 10;
    default: return 0;
    }
}

/* utils/maps.cpp:188-205 (chrmap_ambiguous_4bit): only 1,2,4,8 are unambiguous */
static int ambiguous4(unsigned x) { return !(x == 1 || x == 2 || x == 4 || x == 8); }

static inline int sat16(int x) { return x > 32767 ? 32767 : (x < -32768 ? -32768 : x); }
static inline int sadd(int a, int b) { return sat16(a + b); }   /* v_add, align_simd.cpp:350 */
static inline int ssub(int a, int b) { return sat16(a - b); }   /* v_sub, align_simd.cpp:353 */

/* clamp_to_cell, align_simd.cpp:1264-1278 */
static int clamp_cell(int64_t v, int64_t limit, int * fallback)
{
  if (v > limit) { *fallback = 1; return (int) limit; }
  if (v < -limit) { *fallback = 1; return (int) -limit; }
  return (int) v;
}

static void sentinel(int16_t * score, uint16_t * aligned, uint16_t * matches,
                     uint16_t * mismatches, uint16_t * gaps, char * cigar)
{
  *score = 32767; *aligned = 0; *matches = 0; *mismatches = 0; *gaps = 0; cigar[0] = 0;
}

/* search16_fits, align_simd.cpp:130-134 */
static int fits(int64_t qlen, int64_t dlen)
{
  return (qlen + dlen <= 65535) && (qlen * dlen <= 25000000LL);
}

int vsxo_search16_pair(const char * q, int64_t Q, const char * d, int64_t D,
                       const int64_t P[14], int nmm,
                       int16_t * score, uint16_t * aligned, uint16_t * matches,
                       uint16_t * mismatches, uint16_t * gaps, char * cigar)
{
  /* search16_init, align_simd.cpp:1282-1376: scores limited to 32767, each gap
     penalty to 32767/(1+CDEPTH) = 6553; anything beyond => every pair deferred */
  int fb = 0;
  const int match = clamp_cell(P[0], 32767, &fb);
  const int mism  = clamp_cell(P[1], 32767, &fb);
  int pen[12];
  for (int k = 0; k < 12; ++k) { pen[k] = clamp_cell(P[2 + k], 6553, &fb); }
  const int goql = pen[0], gotl = pen[1], goqi = pen[2], goti = pen[3], goqr = pen[4], gotr = pen[5];
  const int geql = pen[6], getl = pen[7], geqi = pen[8], geti = pen[9], geqr = pen[10], getr = pen[11];

  if (fb) { sentinel(score, aligned, matches, mismatches, gaps, cigar); return 0; }  /* :1463-1479 */

  if (Q == 0)                                                     /* :1481-1539 */
    {
      if (!fits(0, D)) { sentinel(score, aligned, matches, mismatches, gaps, cigar); return 0; }
      *aligned = (uint16_t) D; *matches = 0; *mismatches = 0; *gaps = (uint16_t) D;
      if (D == 0) { *score = 0; cigar[0] = 0; return 0; }
      int64_t a = -(int64_t) gotl - D * (int64_t) getl;
      int64_t b = -(int64_t) gotr - D * (int64_t) getr;
      int64_t x = a > b ? a : b;
      *score = (int16_t) (uint16_t) (x & 0xffff);                 /* plain narrowing cast :1515 */
      sprintf(cigar, "%lldI", (long long) D);
      return 0;
    }
  if (D == 0 || !fits(Q, D))                                      /* :1867-1882 */
    { sentinel(score, aligned, matches, mismatches, gaps, cigar); return 0; }

  /* score matrix, :1319-1342 */
  int S[16][16];
  for (unsigned x = 0; x < 16; ++x)
    for (unsigned y = 0; y < 16; ++y)
      {
        int v;
        if (nmm && (x == 15 || y == 15)) v = mism;
        else if (ambiguous4(x) || ambiguous4(y)) v = 0;
        else v = (x == y) ? match : mism;
        S[x][y] = v;
      }

  const int64_t Dp = 4 * ((D + 3) / 4);      /* columns incl. padding of the last 4-column block */
  const int QRqi = goqi + geqi, Rqi = geqi, QRqr = goqr + geqr, Rqr = geqr;
  const int QRti = goti + geti, Rti = geti, QRtr = gotr + getr, Rtr = getr;

  unsigned char * a = malloc((size_t) Q);
  unsigned char * b = malloc((size_t) Dp);
  int16_t * Hprev = malloc(sizeof(int16_t) * (size_t) Q);   /* H(i, j-1) */
  int16_t * E = malloc(sizeof(int16_t) * (size_t) Q);       /* E(i, j)   */
  unsigned char * B = malloc((size_t) Q * (size_t) Dp);     /* 4 direction bits per cell */
  if (!a || !b || !Hprev || !E || !B) { free(a); free(b); free(Hprev); free(E); free(B); return -1; }

  for (int64_t i = 0; i < Q; ++i) a[i] = vsxo_map4((unsigned char) q[i]);
  for (int64_t j = 0; j < Dp; ++j) b[j] = j < D ? vsxo_map4((unsigned char) d[j]) : 0;   /* :1695-1712 */

  /* left border, aligncolumns_first :844-859, :881-887 */
  {
    int m = gotl + getl;
    for (int64_t i = 0; i < Q; ++i)
      {
        Hprev[i] = (int16_t) ssub(0, m);
        E[i] = (int16_t) ssub(ssub(0, m), i < Q - 1 ? QRqi : QRqr);
        m = sadd(m, getl);
      }
  }

  /* overflow threshold, compute_score_min :1432-1444 */
  int pmax = 0;
  {
    int c[6] = { goql + geql, goqi + geqi, goqr + geqr, gotl + getl, goti + geti, gotr + getr };
    for (int k = 0; k < 6; ++k) if (c[k] > pmax) pmax = c[k];
  }
  const int smin = -32768 + pmax;

  int ovf = 0;
  int htop_prev = 0;      /* H(-1, j-1): Htop(-1) = 0                         (:1895) */
  int htop = 0;           /* H(-1, j), doubles as the F seed of column j       (:1902-1910, :2043-2051) */
  int hmin = 0, hmax = 0;
  int16_t final_score = 0;

  for (int64_t j = 0; j < Dp; ++j)
    {
      if ((j & 3) == 0) { hmin = 0; hmax = 0; }                   /* per 4-column block :810-811 */
      /* top border: plain casts for the first columns, then a saturating chain.
         -(go + (j+1)*ge) cannot leave int16 for j <= 3 (penalties <= 6553). */
      htop_prev = htop;
      if (j == 0) htop = -goql - geql;
      else if (j < 4) htop = -goql - (int) (j + 1) * geql;
      else htop = ssub(htop, geql);
      if (j == 0) htop_prev = 0;

      const int QRt = (j < D - 1) ? QRti : QRtr;                  /* :1719-1753 */
      const int Rt  = (j < D - 1) ? Rti : Rtr;
      int F = ssub(htop, QRt);                                    /* :830-833 */
      int hd = htop_prev;                                         /* diagonal for row 0 */
      const unsigned bj = b[j];

      for (int64_t i = 0; i < Q; ++i)
        {
          const int QRq = (i < Q - 1) ? QRqi : QRqr;              /* :836-897 */
          const int Rq  = (i < Q - 1) ? Rqi : Rqr;
          const int e_in = E[i];
          /* onestep :765-780 */
          int h = sadd(hd, S[bj][a[i]]);
          const int up = F > h;       if (F > h) h = F;
          const int left = e_in > h;  if (e_in > h) h = e_in;
          if (h < hmin) hmin = h;
          if (h > hmax) hmax = h;
          const int hf = ssub(h, QRt);
          const int f = ssub(F, Rt);
          const int eu = f > hf;
          F = f > hf ? f : hf;
          const int he = ssub(h, QRq);
          const int e = ssub(e_in, Rq);
          const int el = e > he;
          E[i] = (int16_t) (e > he ? e : he);
          hd = Hprev[i];              /* H(i, j-1) is the diagonal of row i+1 */
          Hprev[i] = (int16_t) h;
          B[(size_t) i * (size_t) Dp + (size_t) j] = (unsigned char) (up | (left << 1) | (eu << 2) | (el << 3));
        }
      if (j == D - 1) final_score = Hprev[Q - 1];                 /* :1835-1836 */
      if ((j & 3) == 3 && (hmin <= smin || hmax >= 32767)) ovf = 1;   /* :1774-1786 */
    }

  if (ovf)
    {
      sentinel(score, aligned, matches, mismatches, gaps, cigar);
      free(a); free(b); free(Hprev); free(E); free(B);
      return 0;
    }

  /* backtrack16 :1137-1235 */
  char * ops = malloc((size_t) (Q + D + 1));
  if (!ops) { free(a); free(b); free(Hprev); free(E); free(B); return -1; }
  int64_t n = 0;
  int64_t i = Q - 1, j = D - 1;
  char op = 0;
  unsigned al = 0, ma = 0, mi = 0, ga = 0;
  while (i >= 0 && j >= 0)
    {
      const unsigned char dd = B[(size_t) i * (size_t) Dp + (size_t) j];
      ++al;
      if (op == 'I' && (dd & 8)) { --j; ops[n++] = 'I'; }
      else if (op == 'D' && (dd & 4)) { --i; ops[n++] = 'D'; }
      else if (dd & 2) { if (op != 'I') ++ga; --j; op = 'I'; ops[n++] = 'I'; }
      else if (dd & 1) { if (op != 'D') ++ga; --i; op = 'D'; ops[n++] = 'D'; }
      else
        {
          if ((a[i] & b[j]) != 0 && !(nmm && (a[i] == 15 || b[j] == 15))) ++ma; else ++mi;
          --i; --j; op = 'M'; ops[n++] = 'M';
        }
    }
  while (i >= 0) { ++al; if (op != 'D') ++ga; --i; op = 'D'; ops[n++] = 'D'; }
  while (j >= 0) { ++al; if (op != 'I') ++ga; --j; op = 'I'; ops[n++] = 'I'; }

  /* run-length text, left to right; count omitted when 1 (pushop/finishop :1013-1049) */
  char * out = cigar;
  int64_t k = n - 1;
  while (k >= 0)
    {
      int64_t r = k;
      while (r >= 0 && ops[r] == ops[k]) --r;
      const int64_t run = k - r;
      if (run > 1) out += sprintf(out, "%lld", (long long) run);
      *out++ = ops[k];
      k = r;
    }
  *out = 0;

  *score = final_score;
  *aligned = (uint16_t) al; *matches = (uint16_t) ma; *mismatches = (uint16_t) mi; *gaps = (uint16_t) ga;
  free(ops); free(a); free(b); free(Hprev); free(E); free(B);
  return 0;
}

int64_t vsxo_search16_batch(const char * qblob, const uint64_t * qoff, const uint32_t * qlen,
                            const char * tblob, const uint64_t * toff, const uint32_t * tlen,
                            uint64_t npairs, const uint32_t * qi, const uint32_t * ti,
                            const int64_t P[14], int n_mismatch,
                            int16_t * score, uint16_t * aligned, uint16_t * matches,
                            uint16_t * mismatches, uint16_t * gaps,
                            char * cigar_blob, uint64_t * cigar_off, uint64_t capacity)
{
  uint64_t used = 0;
  for (uint64_t k = 0; k < npairs; ++k)
    {
      const uint32_t a = qi[k], b = ti[k];
      const uint64_t need = (uint64_t) qlen[a] + tlen[b] + 24;
      if (used + need > capacity) return -1;
      cigar_off[k] = used;
      if (vsxo_search16_pair(qblob + qoff[a], qlen[a], tblob + toff[b], tlen[b], P, n_mismatch,
                             score + k, aligned + k, matches + k, mismatches + k, gaps + k,
                             cigar_blob + used) != 0)
        return -1;
      used += strlen(cigar_blob + used) + 1;
    }
  return (int64_t) used;
}
