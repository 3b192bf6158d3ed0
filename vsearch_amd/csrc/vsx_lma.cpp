// vsx_lma.cpp -- the scalar fallback for pairs the 16-bit aligner refuses (score == SHRT_MAX sentinel).
//
// In the reference the CALLER owns this step: LinearMemoryAligner::align + alignstats
// (src/core/linmemalign.cpp:311-808, call sites core/searchcore.cpp:806-832, commands/allpairs_global.cpp:447-473).
// It stays on the host CPU here as well (rare path, int64 arithmetic, linear memory).
//
// What has to be reproduced is the reference's OUTPUT (the CIGAR it picks among co-optimal alignments), so this file is
// written against the following specification of that choice, not against the reference's code:
//
//   S1  divide and conquer on the query: a piece of q rows is cut after row floor(q / 2); the upper part is scored top-down,
//       the lower part bottom-up, both over all columns of the piece (linear memory: two score vectors per direction);
//   S2  two ways to cross the cut at column c:  THROUGH (the two halves simply meet at c) and HANGING (a gap in the target
//       spans the cut: the rows on both sides of the cut are deleted, and the opening penalty both halves charged is
//       refunded once).  Per way the FIRST column attaining the maximum counts.  THROUGH wins on a strictly larger score;
//       on equal scores the way with the smaller column wins, THROUGH on equal columns;
//   S3  a piece keeps six facts: whether it touches the head / tail of the query, the head / tail of the target (terminal
//       gap penalties apply there), and whether a target gap is already open on its left / right side (then the opening
//       penalty of a gap starting at that side is waived);
//   S4  gaps in the query INSIDE a sweep are always priced as interior gaps; only the sweep's starting row (an all-gap
//       prefix) uses the terminal penalties of S3.  Gaps in the target use the terminal penalties only in the column
//       farthest from the sweep's start, and in column 0;
//   S5  a piece of ONE query row is solved by enumeration, in this order and replaced only by a strictly better score:
//       "delete the row, then insert all columns", "insert all columns, then delete the row" (whose score is charged ON TOP
//       of the first candidate's -- a quirk of the reference that decides ties and is therefore kept), then "substitute at
//       column c" for c = 0, 1, ...;
//   S6  empty pieces: no columns -> delete the rows; no rows -> insert the columns.
//
// The recursion is an explicit work stack (no call depth proportional to log(query length) x stack frame), the two sweeps
// are ONE routine run in two orientations, and the statistics are computed by an own CIGAR tokenizer.
// Pinned by tests/golden/lma_golden.json and a live fuzz against the reference's class (tests/test_host_cpu.py).
#include "../../include/vsx_search.h"

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <string>
#include <vector>

namespace {

constexpr int64_t kNever = std::numeric_limits<int64_t>::min();

// 4-bit IUPAC set of a nucleotide symbol (utils/maps.cpp:75-117); 0 = not a nucleotide code
struct SymbolSets {
  unsigned char set[256];
  SymbolSets()
  {
    std::memset(set, 0, sizeof set);
    const char * letters = "ACMGRSVTWYHKDBN";            // the set value of letters[k] is k + 1 (A=1, C=2, G=4, T=8 and unions)
    for (int k = 0; letters[k]; ++k)
      {
        set[(unsigned char) letters[k]] = (unsigned char) (k + 1);
        set[(unsigned char) (letters[k] | 0x20)] = (unsigned char) (k + 1);
      }
    set[(unsigned char) 'U'] = set[(unsigned char) 'u'] = 8;
  }
};
const SymbolSets kSets;

enum Zone { HEAD = 0, INNER = 1, TAIL = 2 };

// facts a piece carries (S3)
enum : unsigned {
  Q_HEAD = 1u, Q_TAIL = 2u, T_HEAD = 4u, T_TAIL = 8u,       // the piece touches that end of the whole problem
  OPEN_L = 16u, OPEN_R = 32u                                // a target gap is already open on that side of the piece
};

struct Piece { int64_t q0, t0, qn, tn; unsigned facts; };

// the CIGAR under construction: equal neighbouring operations merge, a count of 1 is not written
class OpWriter {
public:
  void put(char op, int64_t n)
  {
    if (n <= 0) return;
    if (op == last_) { count_ += n; return; }
    close();
    last_ = op;
    count_ = n;
  }
  std::string finish() { close(); last_ = 0; count_ = 0; return std::move(text_); }
private:
  void close()
  {
    if (!last_) return;
    if (count_ > 1) text_ += std::to_string(count_);
    text_.push_back(last_);
  }
  std::string text_;
  char last_ = 0;
  int64_t count_ = 0;
};

class LinearAligner {
public:
  explicit LinearAligner(const vsx_scoring & sc) : count_n_as_mismatch_(sc.n_mismatch != 0)
  {
    q_open_[HEAD] = sc.gap_open_query_left;  q_open_[INNER] = sc.gap_open_query_interior;  q_open_[TAIL] = sc.gap_open_query_right;
    q_ext_[HEAD] = sc.gap_ext_query_left;    q_ext_[INNER] = sc.gap_ext_query_interior;    q_ext_[TAIL] = sc.gap_ext_query_right;
    t_open_[HEAD] = sc.gap_open_target_left; t_open_[INNER] = sc.gap_open_target_interior; t_open_[TAIL] = sc.gap_open_target_right;
    t_ext_[HEAD] = sc.gap_ext_target_left;   t_ext_[INNER] = sc.gap_ext_target_interior;   t_ext_[TAIL] = sc.gap_ext_target_right;
    // substitution scores by symbol set (linmemalign.cpp:189-222): unambiguous symbols score match / mismatch, anything
    // ambiguous scores 0, unless N-as-mismatch is on, which overrides every pairing with set 15
    for (unsigned x = 0; x < 16; ++x)
      for (unsigned y = 0; y < 16; ++y)
        {
          const bool plain = (x && !(x & (x - 1))) && (y && !(y & (y - 1)));
          pair_score_[x][y] = plain ? (x == y ? sc.match : sc.mismatch) : 0;
          if (count_n_as_mismatch_ && (x == 15 || y == 15)) pair_score_[x][y] = sc.mismatch;
        }
  }

  std::string align(const char * query, int64_t qlen, const char * target, int64_t tlen)
  {
    query_ = query; target_ = target;
    for (auto * v : {&down_.reach, &down_.hang, &up_.reach, &up_.hang}) v->assign((size_t) tlen + 1, 0);
    OpWriter out;
    // pending work, last in first out: a piece to solve, or (qn < 0) an operation to write when its turn comes
    std::vector<Piece> todo;
    todo.push_back(Piece {0, 0, qlen, tlen, Q_HEAD | Q_TAIL | T_HEAD | T_TAIL});
    while (!todo.empty())
      {
        const Piece p = todo.back();
        todo.pop_back();
        if (p.qn < 0) { out.put((char) p.q0, p.t0); continue; }                  // a deferred operation
        if (p.tn == 0) { out.put('D', p.qn); continue; }                         // S6
        if (p.qn == 0) { out.put('I', p.tn); continue; }
        if (p.qn == 1) { single_row(p, out); continue; }                         // S5
        const Cut c = best_cut(p);
        const int64_t upper_rows = p.qn / 2;
        const unsigned keep_left = p.facts & (Q_HEAD | T_HEAD | OPEN_L), keep_right = p.facts & (Q_TAIL | T_TAIL | OPEN_R);
        // what the halves know about the target's ends: only a half that still reaches the end inherits the fact
        const unsigned upper_facts = keep_left | ((c.column == p.tn) ? (p.facts & T_TAIL) : 0u);
        const unsigned lower_facts = keep_right | ((c.column == 0) ? (p.facts & T_HEAD) : 0u);
        if (!c.hanging)
          {
            todo.push_back(Piece {p.q0 + upper_rows, p.t0 + c.column, p.qn - upper_rows, p.tn - c.column, lower_facts});
            todo.push_back(Piece {p.q0, p.t0, upper_rows, c.column, upper_facts});
          }
        else
          {
            // the two rows around the cut are deleted; both neighbours see an open target gap on that side
            todo.push_back(Piece {p.q0 + upper_rows + 1, p.t0 + c.column, p.qn - upper_rows - 1, p.tn - c.column, lower_facts | OPEN_L});
            todo.push_back(Piece {(int64_t) 'D', 2, -1, 0, 0});
            todo.push_back(Piece {p.q0, p.t0, upper_rows - 1, c.column, upper_facts | OPEN_R});
          }
      }
    return out.finish();
  }

  // score and counts of a finished CIGAR (linmemalign.cpp:722-808): a gap that starts the alignment is a head gap, one that
  // ends it a tail gap (head wins when the whole alignment is one gap), all others are interior
  void measure(const std::string & cigar, const char * query, const char * target, int64_t * score, int64_t * columns,
               int64_t * matches, int64_t * mismatches, int64_t * gaps) const
  {
    int64_t total = 0, cols = 0, same = 0, differ = 0, opened = 0, qpos = 0, tpos = 0;
    size_t at = 0;
    while (at < cigar.size())
      {
        int64_t n = 0;
        bool counted = false;
        while (at < cigar.size() && cigar[at] >= '0' && cigar[at] <= '9') { n = 10 * n + (cigar[at++] - '0'); counted = true; }
        if (!counted) n = 1;
        const char op = cigar[at++];
        const bool first = (qpos == 0 && tpos == 0), last = (at == cigar.size());
        const Zone z = first ? HEAD : (last ? TAIL : INNER);
        cols += n;
        if (op == 'M')
          for (int64_t k = 0; k < n; ++k, ++qpos, ++tpos)
            {
              const unsigned x = kSets.set[(unsigned char) query[qpos]], y = kSets.set[(unsigned char) target[tpos]];
              total += pair_score_[y][x];
              if ((x & y) && !(count_n_as_mismatch_ && (x == 15 || y == 15))) ++same; else ++differ;
            }
        else if (op == 'I') { total -= q_open_[z] + n * q_ext_[z]; ++opened; tpos += n; }
        else if (op == 'D') { total -= t_open_[z] + n * t_ext_[z]; ++opened; qpos += n; }
      }
    *score = total; *columns = cols; *matches = same; *mismatches = differ; *gaps = opened;
  }

private:
  // scores of the best alignments of (rows swept so far) x (first j columns in sweep order): `reach` = ending anyhow,
  // `hang` = ending inside a gap in the target
  struct Frontier { std::vector<int64_t> reach, hang; };
  struct Cut { bool hanging; int64_t column; };

  int64_t substitution(int64_t qi, int64_t ti) const
  {
    return pair_score_[kSets.set[(unsigned char) target_[ti]]][kSets.set[(unsigned char) query_[qi]]];
  }
  int64_t query_gap(Zone z, int64_t n) const { return q_open_[z] + n * q_ext_[z]; }

  // One orientation of S1: sweep `rows` rows of piece p starting from its top (mirrored = false) or from its bottom, with the
  // columns running away from the matching side.  Orientation only changes which symbols meet and which facts are "near".
  void sweep(Frontier & f, const Piece & p, int64_t rows, bool mirrored) const
  {
    const unsigned near_q = mirrored ? Q_TAIL : Q_HEAD, near_t = mirrored ? T_TAIL : T_HEAD, far_t = mirrored ? T_HEAD : T_TAIL;
    const unsigned near_open = mirrored ? OPEN_R : OPEN_L;
    const Zone start_q = (p.facts & near_q) ? (mirrored ? TAIL : HEAD) : INNER;      // S4: only the all-gap starting row
    const Zone col0_t = (p.facts & near_t) ? (mirrored ? TAIL : HEAD) : INNER;
    const Zone last_t = (p.facts & far_t) ? (mirrored ? HEAD : TAIL) : INNER;
    const int64_t col0_open = (p.facts & near_open) ? 0 : t_open_[col0_t];

    f.reach[0] = 0;
    f.hang[0] = 0;
    for (int64_t j = 1; j <= p.tn; ++j) { f.reach[(size_t) j] = -query_gap(start_q, j); f.hang[(size_t) j] = kNever; }
    for (int64_t i = 1; i <= rows; ++i)
      {
        const int64_t qi = mirrored ? p.q0 + p.qn - i : p.q0 + i - 1;
        int64_t corner = f.reach[0];                               // the cell diagonally behind the one being filled
        int64_t here = -(col0_open + i * t_ext_[col0_t]);          // column 0: i rows deleted
        f.reach[0] = here;
        int64_t run = kNever;                                      // ending inside a gap in the query, along this row
        for (int64_t j = 1; j <= p.tn; ++j)
          {
            const size_t jx = (size_t) j;
            const Zone tz = (j == p.tn) ? last_t : INNER;
            run = std::max(run, here - q_open_[INNER]) - q_ext_[INNER];
            f.hang[jx] = std::max(f.hang[jx], f.reach[jx] - t_open_[tz]) - t_ext_[tz];
            const int64_t ti = mirrored ? p.t0 + p.tn - j : p.t0 + j - 1;
            here = std::max(std::max(corner + substitution(qi, ti), run), f.hang[jx]);
            corner = f.reach[jx];
            f.reach[jx] = here;
          }
      }
    f.hang[0] = f.reach[0];
  }

  // S2
  Cut best_cut(const Piece & p)
  {
    const int64_t upper_rows = p.qn / 2;
    sweep(down_, p, upper_rows, false);
    sweep(up_, p, p.qn - upper_rows, true);
    int64_t through = kNever, through_at = -1, hanging = kNever, hanging_at = -1;
    for (int64_t c = 0; c <= p.tn; ++c)
      {
        const size_t a = (size_t) c, b = (size_t) (p.tn - c);
        const int64_t meet = down_.reach[a] + up_.reach[b];
        if (meet > through) { through = meet; through_at = c; }
        // both halves opened the gap: one opening is refunded, priced where the gap lies
        const Zone z = ((p.facts & T_HEAD) && c == 0) ? HEAD : (((p.facts & T_TAIL) && c == p.tn) ? TAIL : INNER);
        const int64_t span = down_.hang[a] + up_.hang[b] + t_open_[z];
        if (span > hanging) { hanging = span; hanging_at = c; }
      }
    const bool take_hanging = (hanging > through) || (hanging == through && hanging_at < through_at);
    return take_hanging ? Cut {true, hanging_at} : Cut {false, through_at};
  }

  // S5
  void single_row(const Piece & p, OpWriter & out) const
  {
    const Zone qz_head = (p.facts & Q_HEAD) ? HEAD : INNER, qz_tail = (p.facts & Q_TAIL) ? TAIL : INNER;
    const Zone tz_head = (p.facts & T_HEAD) ? HEAD : INNER, tz_tail = (p.facts & T_TAIL) ? TAIL : INNER;
    enum { DELETE_THEN_INSERT = -1 };
    // "delete, then insert": the deleted row hangs on the left side, the inserted columns end the piece
    int64_t charged = -(((p.facts & OPEN_L) ? 0 : t_open_[tz_head]) + t_ext_[tz_head]) - query_gap(qz_tail, p.tn);
    int64_t best = charged;
    int64_t choice = DELETE_THEN_INSERT;
    // "insert, then delete": priced on top of what the first candidate was charged (S5)
    charged -= query_gap(qz_head, p.tn);
    charged -= ((p.facts & OPEN_R) ? 0 : t_open_[tz_tail]) + t_ext_[tz_tail];
    if (charged > best) { best = charged; choice = p.tn; }
    for (int64_t c = 0; c < p.tn; ++c)
      {
        const int64_t before = c, after = p.tn - 1 - c;
        int64_t s = substitution(p.q0, p.t0 + c);
        if (before > 0) s -= query_gap(qz_head, before);
        if (after > 0) s -= query_gap(qz_tail, after);
        if (s > best) { best = s; choice = c; }
      }
    if (choice == DELETE_THEN_INSERT) { out.put('D', 1); out.put('I', p.tn); }
    else if (choice == p.tn) { out.put('I', p.tn); out.put('D', 1); }
    else { out.put('I', choice); out.put('M', 1); out.put('I', p.tn - 1 - choice); }
  }

  int64_t q_open_[3], q_ext_[3], t_open_[3], t_ext_[3];
  int64_t pair_score_[16][16];
  bool count_n_as_mismatch_;
  const char * query_ = nullptr;
  const char * target_ = nullptr;
  Frontier down_, up_;
};

}  // namespace

extern "C" int vsx_lma_align(const vsx_scoring * scoring, const char * q, uint64_t qlen, const char * t, uint64_t tlen,
                             int64_t * score, int64_t * alnlen, int64_t * matches, int64_t * mismatches,
                             int64_t * gaps, char ** cigar)
{
  if (!scoring || !cigar || (qlen && !q) || (tlen && !t)) return VSX_EINVAL;
  LinearAligner aligner(*scoring);
  const std::string text = aligner.align(q, (int64_t) qlen, t, (int64_t) tlen);
  int64_t s, l, m, x, g;
  aligner.measure(text, q, t, &s, &l, &m, &x, &g);
  if (score) *score = s;
  if (alnlen) *alnlen = l;
  if (matches) *matches = m;
  if (mismatches) *mismatches = x;
  if (gaps) *gaps = g;
  *cigar = (char *) std::malloc(text.size() + 1);
  if (!*cigar) return VSX_ENOMEM;
  std::memcpy(*cigar, text.c_str(), text.size() + 1);
  return VSX_OK;
}
