// vsx_msa.cpp -- star multiple alignment, profile and consensus of one cluster from the stored CIGARs
// (the consumer that pins the aligner's CIGAR grammar: SURVEY.md 8a row 15).
//
// Restates src/core/msa.cpp of the reference: find_max_insertions_per_position (:154-189), process_and_print_centroid
// (:268-305), compute_and_print_msa (:324-426), compute_and_print_consensus (:429-500), print_alignment_profile (:527-566).
// CIGAR orientation (query = member, target = centroid): 'M' consumes both, 'I' is a gap in the member (consumes a
// centroid position), a 'D' run at centroid position p is a block of member symbols inserted BEFORE p.
// No DP happens here; host code (formatting of FASTA files stays with the caller).
#include "../../include/vsx_search.h"

#include <algorithm>
#include <cctype>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

namespace {

// update_profile (msa.cpp:92-141): A C G T/U | IUPAC incl. N | '-' ; anything else is not counted
inline int prof_slot(char c)
{
  switch (std::toupper((unsigned char) c))
    {
    case 'A': return 0; case 'C': return 1; case 'G': return 2; case 'T': case 'U': return 3;
    case 'R': case 'Y': case 'S': case 'W': case 'K': case 'M': case 'B': case 'D': case 'H': case 'V': case 'N': return 4;
    case '-': return 5;
    default: return -1;
    }
}

struct Run { char op; int64_t n; };

std::vector<Run> parse_cigar(const char * c)
{
  std::vector<Run> out;
  while (*c)
    {
      int64_t n = 0; bool any = false;
      while (*c >= '0' && *c <= '9') { n = n * 10 + (*c - '0'); ++c; any = true; }
      if (!*c) break;
      out.push_back(Run {*c, any ? n : 1});
      ++c;
    }
  return out;
}

}  // namespace

extern "C" int vsx_msa(uint32_t n, const char * const * seqs, const uint32_t * lens, const char * const * cigars,
                       const uint64_t * abundances, vsx_msa_out * out)
{
  if (!out || n == 0 || !seqs || !lens || !cigars) return VSX_EINVAL;
  std::memset(out, 0, sizeof *out);
  const int64_t clen = lens[0];
  // max insertions in front of each centroid position (:154-189)
  std::vector<int64_t> maxins((size_t) clen + 1, 0);
  std::vector<std::vector<Run>> runs(n);
  for (uint32_t i = 1; i < n; ++i)
    {
      if (!cigars[i]) return VSX_EINVAL;
      runs[i] = parse_cigar(cigars[i]);
      int64_t pos = 0;
      char prev = 0;
      for (const Run & r : runs[i])
        {
          if (r.op == 'M' || r.op == 'I') pos += r.n;
          else if (r.op == 'D')
            {
              // two adjacent 'D' runs would put more member symbols in front of one centroid position than maxins accounts
              // for (the row fill below would run past alnlen); no aligner emits them, vsx_msa_device rejects them too
              if (pos > clen || prev == 'D') return VSX_EINVAL;
              maxins[(size_t) pos] = std::max(maxins[(size_t) pos], r.n);
            }
          prev = r.op;
        }
      if (pos != clen) return VSX_EINVAL;          // the CIGAR must span the centroid
    }
  int64_t alnlen = clen;
  for (int64_t v : maxins) alnlen += v;

  std::vector<uint64_t> prof((size_t) alnlen * 6, 0);
  std::vector<char> rows((size_t) (n + 1) * (size_t) (alnlen + 1), 0);
  auto put = [&](char * row, int64_t & p, char c, uint64_t ab) {
    const int s = prof_slot(c);
    if (s >= 0) prof[(size_t) p * 6 + (size_t) s] += ab;
    row[p++] = c;
  };

  // centroid row (:268-305)
  {
    char * row = rows.data();
    const uint64_t ab = abundances ? abundances[0] : 1;
    int64_t p = 0;
    for (int64_t i = 0; i < clen; ++i)
      {
        for (int64_t j = 0; j < maxins[(size_t) i]; ++j) put(row, p, '-', ab);
        put(row, p, seqs[0][i], ab);
      }
    for (int64_t j = 0; j < maxins[(size_t) clen]; ++j) put(row, p, '-', ab);
  }
  // member rows (:344-425)
  for (uint32_t i = 1; i < n; ++i)
    {
      char * row = rows.data() + (size_t) i * (size_t) (alnlen + 1);
      const uint64_t ab = abundances ? abundances[i] : 1;
      int64_t p = 0, qpos = 0, tpos = 0;
      bool inserted = false;
      auto pad = [&]() { if (!inserted) for (int64_t j = 0; j < maxins[(size_t) qpos]; ++j) put(row, p, '-', ab); };
      for (const Run & r : runs[i])
        {
          if (r.op == 'D')
            {
              for (int64_t j = 0; j < r.n; ++j) { if (tpos >= lens[i]) return VSX_EINVAL; put(row, p, seqs[i][tpos++], ab); }
              for (int64_t j = r.n; j < maxins[(size_t) qpos]; ++j) put(row, p, '-', ab);
              inserted = true;
            }
          else if (r.op == 'M')
            for (int64_t j = 0; j < r.n; ++j)
              {
                pad();
                if (tpos >= lens[i]) return VSX_EINVAL;
                put(row, p, seqs[i][tpos++], ab);
                ++qpos; inserted = false;
              }
          else if (r.op == 'I')
            for (int64_t j = 0; j < r.n; ++j) { pad(); put(row, p, '-', ab); ++qpos; inserted = false; }
        }
      pad();
    }
  // consensus (:429-500)
  char * crow = rows.data() + (size_t) n * (size_t) (alnlen + 1);
  std::string cons;
  const int64_t leftc = maxins.front(), rightc = maxins.back();
  for (int64_t i = 0; i < leftc; ++i) crow[i] = '+';
  for (int64_t i = alnlen - rightc; i < alnlen; ++i) crow[i] = '+';
  static const char sym4[16] = {'-', 'A', 'C', 'M', 'G', 'R', 'S', 'V', 'T', 'W', 'Y', 'H', 'K', 'D', 'B', 'N'};
  for (int64_t i = leftc; i < alnlen - rightc; ++i)
    {
      unsigned best_sym = 0; uint64_t best = 0;
      for (unsigned k = 0; k < 4; ++k)
        {
          const uint64_t c = prof[(size_t) i * 6 + k];
          if (c > best) { best = c; best_sym = 1u << k; }
        }
      const uint64_t ncount = prof[(size_t) i * 6 + 4];
      if (best == 0 && ncount > 0) { best = ncount; best_sym = 15; }
      if (best >= prof[(size_t) i * 6 + 5]) { crow[i] = sym4[best_sym]; cons.push_back(sym4[best_sym]); }
      else crow[i] = '-';
    }

  out->alnlen = (uint64_t) alnlen;
  out->n_rows = n + 1;
  out->rows = (char *) std::malloc(rows.size());
  out->consensus = (char *) std::malloc(cons.size() + 1);
  out->profile = (uint64_t *) std::malloc(std::max<size_t>(prof.size(), 1) * sizeof(uint64_t));
  if (!out->rows || !out->consensus || !out->profile) { vsx_msa_out_free(out); return VSX_ENOMEM; }
  std::memcpy(out->rows, rows.data(), rows.size());
  std::memcpy(out->consensus, cons.c_str(), cons.size() + 1);
  out->conslen = cons.size();
  std::memcpy(out->profile, prof.data(), prof.size() * sizeof(uint64_t));
  return VSX_OK;
}

extern "C" void vsx_msa_out_free(vsx_msa_out * o)
{
  if (!o) return;
  std::free(o->rows); std::free(o->consensus); std::free(o->profile);
  std::memset(o, 0, sizeof *o);
}
