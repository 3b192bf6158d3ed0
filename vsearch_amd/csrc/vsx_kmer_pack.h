// vsx_kmer_pack.h -- the packed postings format of the device k-mer index, shared by the build kernel (vsx_kmer.hip), the count /
// selection kernels and a host entry that the CPU test suite drives (vsx_kmer_host.cpp: vsx_internal_kmer_pack_*).
//
// A bucket = the counters (tile-local indices 0 .. 32 759) of the sequences of one tile that contain one word, SORTED, as 16-byte
// units: bytes 0-1 the first counter of the unit, bytes 2 .. 15 fourteen gaps -- 15 postings.  Every 252nd counter (251 mod 252)
// is a DUMMY that stands for no sequence: a gap above 255 hops over dummies, the unused tail of a bucket's last unit stays on one
// (gaps of 0).  Sequence s of a tile (0 <= s < 130 * 251) owns counter (s mod 130) * 252 + s / 130.
#ifndef VSX_KMER_PACK_H
#define VSX_KMER_PACK_H

#include <stdint.h>

#if defined(__HIPCC__)
#define KM_HD __host__ __device__ __forceinline__
#else
#define KM_HD inline
#endif

#define KM_PK_PERIOD 252u
#define KM_PK_REAL 251u
#define KM_PK_ROWS 130u                                   /* periods of a tile */
#define KM_PK_TILE_SEQS (KM_PK_ROWS * KM_PK_REAL)        /* 32 630 sequences per tile */
#define KM_PK_SLOTS 15

struct KmPkUnit { uint32_t w[4]; };

KM_HD uint32_t km_pk_counter_of(uint32_t local_seq) { return (local_seq % KM_PK_ROWS) * KM_PK_PERIOD + local_seq / KM_PK_ROWS; }
KM_HD uint32_t km_pk_seq_of(uint32_t counter) { return (counter % KM_PK_PERIOD) * KM_PK_ROWS + counter / KM_PK_PERIOD; }
KM_HD bool km_pk_is_dummy(uint32_t counter) { return counter % KM_PK_PERIOD == KM_PK_REAL; }

// The encoder of one bucket: push() the counters in ascending order, then finish().  out == nullptr: only the unit count.
struct KmPkEncoder {
  KmPkUnit * out;
  uint32_t units, s, acc;
  uint32_t wv[4];
  KM_HD explicit KmPkEncoder(KmPkUnit * o) : out(o), units(0), s(0), acc(0) { wv[0] = wv[1] = wv[2] = wv[3] = 0; }
  KM_HD void put(uint32_t delta)                           // slot s (1 .. 14) of the open unit
  {
    const uint32_t bytepos = s + 1;                        // bytes 0-1 hold the first counter
    wv[bytepos >> 2] |= delta << (8 * (bytepos & 3));
    ++s;
  }
  KM_HD void close()
  {
    if (out) { KmPkUnit u; u.w[0] = wv[0]; u.w[1] = wv[1]; u.w[2] = wv[2]; u.w[3] = wv[3]; out[units] = u; }
    ++units;
    s = 0;
    wv[0] = wv[1] = wv[2] = wv[3] = 0;
  }
  KM_HD void push(uint32_t idx)
  {
    for (;;)
      {
        if (s == KM_PK_SLOTS) close();
        if (s == 0) { wv[0] = idx; acc = idx; s = 1; return; }
        if (idx - acc <= 255u) { put(idx - acc); acc = idx; return; }
        const uint32_t d = KM_PK_REAL + KM_PK_PERIOD * ((acc + 4u) / KM_PK_PERIOD);       // the farthest dummy within 255
        put(d - acc);
        acc = d;
      }
  }
  KM_HD void finish()
  {
    if (s == 0) return;
    if (s < KM_PK_SLOTS && !km_pk_is_dummy(acc))
      put(KM_PK_REAL + KM_PK_PERIOD * (acc / KM_PK_PERIOD) - acc);                        // the dummy of acc's own period
    close();                                                                              // the remaining gaps are 0: they stay on the dummy
  }
};

#endif
