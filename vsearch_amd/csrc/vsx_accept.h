// vsx_accept.h -- the accept filter evaluated on the device (shared by the traceback kernel and the ranking kernels).
#ifndef VSX_ACCEPT_H
#define VSX_ACCEPT_H
#include "vsx_internal.h"

// align_trim (core/searchcore.cpp:343-464) + search_acceptable_aligned (:664-737) on the finished alignment, with the
// reference's double expressions (no contraction: every product feeds a comparison or a quotient, never a sum).
// first_text / last_text: the first and last run of the CIGAR in TEXT order, (length << 2) | op, op 0 = M, 1 = I, 2 = D.
#pragma clang fp contract(off)
__device__ __forceinline__ unsigned int accept_verdict(const VsxFilterDev & F, int Q, int D, int al, int ma, int mi, int ga, unsigned int first_text, unsigned int last_text,
                                                       double * id_out = nullptr)
{
  int tql = 0, ttl = 0, tqr = 0, ttr = 0;
  if ((first_text & 3u) != 0u) { if ((first_text & 3u) == 2u) tql = (int) (first_text >> 2); else ttl = (int) (first_text >> 2); }
  if ((last_text & 3u) != 0u) { if ((last_text & 3u) == 2u) tqr = (int) (last_text >> 2); else ttr = (int) (last_text >> 2); }
  if (tql >= al) tqr = 0;
  if (ttl >= al) ttr = 0;
  const int indels = al - ma - mi;
  const int ial = al - tql - ttl - tqr - ttr;
  const int iindels = indels - tql - ttl - tqr - ttr;
  const int igaps = ga - ((tql + ttl) > 0 ? 1 : 0) - ((tqr + ttr) > 0 ? 1 : 0);
  const int shortest = Q < D ? Q : D, longest = Q < D ? D : Q;
  double id;
  switch (F.iddef)
    {
    case 0: id = shortest > 0 ? 100.0 * ma / shortest : 0.0; break;
    case 2: id = ial > 0 ? 100.0 * ma / ial : 0.0; break;
    case 3: { const double x = 100.0 * (1.0 - (1.0 * (mi + ga) / longest)); id = x > 0.0 ? x : 0.0; } break;
    default: id = al > 0 ? 100.0 * ma / al : 0.0; break;          // 1 and 4
    }
  if (id_out) *id_out = id;
  const bool pass = (id >= 100.0 * F.weak_id) && (mi <= F.maxsubs) && (igaps <= F.maxgaps) && (ial >= F.mincols) &&
                    ((F.leftjust == 0) || (tql + ttl == 0)) && ((F.rightjust == 0) || (tqr + ttr == 0)) &&
                    (ma + mi >= F.query_cov * Q) && (ma + mi >= F.target_cov * (double) D) && (id <= 100.0 * F.maxid) &&
                    (100.0 * ma / (ma + mi) >= F.mid) && (mi + iindels <= F.maxdiffs);
  if (!pass) return 3u;
  return (id >= 100.0 * F.id) ? 1u : 2u;
}

#endif
