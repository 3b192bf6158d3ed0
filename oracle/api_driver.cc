// api_driver.cc -- TEST INFRASTRUCTURE ONLY.  An embedder of the reference's library API (src/vsearch_api.h), larger than its
// api_examples: for a database / query FASTA pair it runs
//   search   the reference's sequential entry point (search_session_single = the reference's own code) for every query, then
//            search_batch -- which oracle/Makefile `ref_api` links to ../shim/vsx_api_adapter.cpp, i.e. the GPU path -- and
//            compares every field of every search_result_s;
//   cluster  cluster_assign_single over the length-sorted database (reference code) against cluster_assign_batch in several
//            ranges (GPU path), every field of every cluster_result_s.
// Usage: api_driver search  db.fa q.fa id maxaccepts maxrejects strand(0|1) qmask dbmask [key=value ...]  (masks: none|soft|dust)
//        api_driver cluster db.fa      id maxaccepts maxrejects batch_size qmask        [key=value ...]
//        key = a CLI option name (wordlength, iddef, maxgaps, ..., match, mismatch, gapopen_i/e, gapext_i/e; sizes=1 reads ";size=")
// Exit code 0 = identical; differences are listed on stderr.
#include "vsearch_api.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

static void read_fasta(char const * path, std::vector<std::string> & labels, std::vector<std::string> & seqs)
{
  std::FILE * f = std::fopen(path, "r");
  if (f == nullptr) { std::fprintf(stderr, "cannot open %s\n", path); std::exit(2); }
  std::string line, label, seq;
  bool have = false;
  int c;
  auto flush = [&]() { if (have) { labels.push_back(label); seqs.push_back(seq); } };
  while (true)
    {
      line.clear();
      while ((c = std::fgetc(f)) != EOF && c != '\n') if (c != '\r') line.push_back((char) c);
      if (!line.empty() && line[0] == '>') { flush(); label = line.substr(1); seq.clear(); have = true; }
      else seq += line;
      if (c == EOF) break;
    }
  flush();
  std::fclose(f);
}

static Masking mask_of(char const * s)
{
  if (std::strcmp(s, "none") == 0) return Masking::none;
  if (std::strcmp(s, "soft") == 0) return Masking::soft;
  return Masking::dust;
}

// trailing key=value arguments -> Parameters fields (the names of the CLI options; sizes = 1: abundances from ";size=" in the labels)
static bool g_sizes = false;
static void apply_options(struct Parameters & p, int argc, char ** argv, int first)
{
  for (int k = first; k < argc; ++k)
    {
      char const * eq = std::strchr(argv[k], '=');
      if (eq == nullptr) { std::fprintf(stderr, "bad option %s\n", argv[k]); std::exit(2); }
      std::string const key(argv[k], (size_t) (eq - argv[k]));
      double const v = std::atof(eq + 1);
      if (key == "wordlength") p.opt_wordlength = (int64_t) v;
      else if (key == "minwordmatches") p.opt_minwordmatches = (int64_t) v;
      else if (key == "iddef") p.opt_iddef = (int64_t) v;
      else if (key == "weak_id") p.opt_weak_id = v;
      else if (key == "maxgaps") p.opt_maxgaps = (int64_t) v;
      else if (key == "maxsubs") p.opt_maxsubs = (int64_t) v;
      else if (key == "maxdiffs") p.opt_maxdiffs = (int64_t) v;
      else if (key == "mincols") p.opt_mincols = (int64_t) v;
      else if (key == "query_cov") p.opt_query_cov = v;
      else if (key == "target_cov") p.opt_target_cov = v;
      else if (key == "maxid") p.opt_maxid = v;
      else if (key == "mid") p.opt_mid = v;
      else if (key == "minqt") p.opt_minqt = v;
      else if (key == "maxqt") p.opt_maxqt = v;
      else if (key == "minsl") p.opt_minsl = v;
      else if (key == "maxsl") p.opt_maxsl = v;
      else if (key == "idprefix") p.opt_idprefix = (int64_t) v;
      else if (key == "idsuffix") p.opt_idsuffix = (int64_t) v;
      else if (key == "leftjust") p.opt_leftjust = (int64_t) v;
      else if (key == "rightjust") p.opt_rightjust = (int64_t) v;
      else if (key == "selfid") p.opt_selfid = (int64_t) v;
      else if (key == "self") p.opt_self = (int64_t) v;
      else if (key == "maxqsize") p.opt_maxqsize = (int64_t) v;
      else if (key == "mintsize") p.opt_mintsize = (int64_t) v;
      else if (key == "minsizeratio") p.opt_minsizeratio = v;
      else if (key == "maxsizeratio") p.opt_maxsizeratio = v;
      else if (key == "sizeorder") p.opt_sizeorder = v != 0;
      else if (key == "hardmask") p.opt_hardmask = v != 0;
      else if (key == "sizes") g_sizes = v != 0;
      else if (key == "match") p.opt_match = (int64_t) v;
      else if (key == "mismatch") p.opt_mismatch = (int64_t) v;
      else if (key == "gapopen_i") { p.opt_gap_open_query_interior = (int) v; p.opt_gap_open_target_interior = (int) v; }
      else if (key == "gapopen_e") { p.opt_gap_open_query_left = p.opt_gap_open_target_left = p.opt_gap_open_query_right = p.opt_gap_open_target_right = (int) v; }
      else if (key == "gapext_i") { p.opt_gap_extension_query_interior = (int) v; p.opt_gap_extension_target_interior = (int) v; }
      else if (key == "gapext_e") { p.opt_gap_extension_query_left = p.opt_gap_extension_target_left = p.opt_gap_extension_query_right = p.opt_gap_extension_target_right = (int) v; }
      else { std::fprintf(stderr, "unknown option %s\n", key.c_str()); std::exit(2); }
    }
}

static int64_t size_of(std::string const & label)
{
  if (!g_sizes) return 1;
  size_t const at = label.find(";size=");
  return at == std::string::npos ? 1 : std::atoll(label.c_str() + at + 6);
}

static void load(Database & db, struct Parameters & parameters, char const * path, Masking mode)
{
  std::vector<std::string> labels, seqs;
  read_fasta(path, labels, seqs);
  db.init();
  for (size_t i = 0; i < labels.size(); ++i)
    db.add(false, labels[i].c_str(), seqs[i].c_str(), nullptr, labels[i].size(), seqs[i].size(), size_of(labels[i]));
  if (mode == Masking::dust) dust_all(db, parameters);
  else if (mode == Masking::soft && parameters.opt_hardmask) hardmask_all(db);       // (usearch_global.cpp / cluster.cpp:1192-1197)
}

static int run_search(int argc, char ** argv)
{
  if (argc < 10) return 2;
  struct Parameters parameters;
  parameters.opt_wordlength = 8;
  parameters.opt_id = std::atof(argv[4]);
  parameters.opt_maxaccepts = std::atol(argv[5]);
  parameters.opt_maxrejects = std::atol(argv[6]);
  parameters.opt_strand = std::atoi(argv[7]) != 0;
  parameters.opt_qmask = mask_of(argv[8]);
  parameters.opt_dbmask = mask_of(argv[9]);
  parameters.opt_threads = 4;
  apply_options(parameters, argc, argv, 10);
  vsearch_session_begin(parameters);

  Database db;
  load(db, parameters, argv[2], parameters.opt_dbmask);
  Dbindex dbindex;
  dbindex.prepare(1, parameters.opt_dbmask, db, parameters);
  dbindex.add_all_sequences(parameters.opt_dbmask, db, parameters);

  std::vector<std::string> qlabels, qseqs;
  read_fasta(argv[3], qlabels, qseqs);
  int const nq = (int) qlabels.size();
  int const per = (int) (parameters.opt_maxaccepts + 2);

  std::vector<struct search_result_s> seq_results((size_t) nq * per), batch_results((size_t) nq * per);
  std::vector<int> seq_counts(nq, 0), batch_counts(nq, 0);

  struct search_session_s * ss = search_session_alloc();
  search_session_init(ss, parameters, dbindex, db);
  for (int i = 0; i < nq; ++i)
    search_session_single(ss, qseqs[i].c_str(), qlabels[i].c_str(), (int) qseqs[i].size(), size_of(qlabels[i]), &seq_results[(size_t) i * per], per, &seq_counts[i]);
  search_session_cleanup(ss);
  search_session_free(ss);

  if (char const * dump = std::getenv("API_DRIVER_DUMP"))          // the reference's sequential results, for a look from outside
    {
      std::FILE * f = std::fopen(dump, "w");
      for (int i = 0; i < nq; ++i)
        for (int j = 0; j < seq_counts[i]; ++j)
          {
            struct search_result_s const & a = seq_results[(size_t) i * per + j];
            std::fprintf(f, "%s\t%s\t%.1f\t%d\t%d\t%d\n", qlabels[i].c_str(), db.getheader(a.target), a.id, a.alignment_length, a.mismatches, a.gaps);
          }
      std::fclose(f);
      if (std::getenv("API_DRIVER_SEQUENTIAL_ONLY")) return 0;
    }
  std::vector<char const *> qs(nq), qh(nq);
  std::vector<int> ql(nq);
  std::vector<int64_t> qz(nq, 1);
  for (int i = 0; i < nq; ++i) { qs[i] = qseqs[i].c_str(); qh[i] = qlabels[i].c_str(); ql[i] = (int) qseqs[i].size(); qz[i] = size_of(qlabels[i]); }
  search_batch(parameters, dbindex, db, qs.data(), qh.data(), ql.data(), qz.data(), nq, batch_results.data(), per, batch_counts.data());

  long bad = 0, hits = 0, minus = 0;
  for (int i = 0; i < nq; ++i)
    {
      if (seq_counts[i] != batch_counts[i])
        {
          if (bad++ < 10) std::fprintf(stderr, "query %d: %d hits in the batch, %d sequentially\n", i, batch_counts[i], seq_counts[i]);
          continue;
        }
      for (int j = 0; j < seq_counts[i]; ++j)
        {
          struct search_result_s const & a = seq_results[(size_t) i * per + j];
          struct search_result_s const & b = batch_results[(size_t) i * per + j];
          ++hits;
          minus += a.strand;
          if (a.target != b.target || a.id != b.id || a.matches != b.matches || a.mismatches != b.mismatches || a.gaps != b.gaps ||
              a.alignment_length != b.alignment_length || a.query_length != b.query_length || a.target_length != b.target_length ||
              a.accepted != b.accepted || a.strand != b.strand)
            if (bad++ < 10)
              std::fprintf(stderr, "query %d hit %d: batch t%d id %.4f m%d x%d g%d l%d acc%d s%d / sequential t%d id %.4f m%d x%d g%d l%d acc%d s%d\n", i, j,
                           b.target, b.id, b.matches, b.mismatches, b.gaps, b.alignment_length, b.accepted, b.strand,
                           a.target, a.id, a.matches, a.mismatches, a.gaps, a.alignment_length, a.accepted, a.strand);
        }
    }
  std::printf("search: %d queries, %ld hits (%ld on the minus strand), %ld differences\n", nq, hits, minus, bad);
  dbindex.clear();
  db.clear();
  vsearch_session_end();
  return bad == 0 ? 0 : 1;
}

static int run_cluster(int argc, char ** argv)
{
  if (argc < 8) return 2;
  struct Parameters parameters;
  parameters.opt_wordlength = 8;
  parameters.opt_id = std::atof(argv[3]);
  parameters.opt_maxaccepts = std::atol(argv[4]);
  parameters.opt_maxrejects = std::atol(argv[5]);
  int const batch = std::atoi(argv[6]);
  parameters.opt_qmask = mask_of(argv[7]);
  parameters.opt_threads = 4;
  apply_options(parameters, argc, argv, 8);
  vsearch_session_begin(parameters);

  Database db;
  load(db, parameters, argv[2], parameters.opt_qmask);
  db.sortbylength(parameters);
  int const sc = (int) db.getsequencecount();

  Dbindex dbindex;
  dbindex.prepare(1, parameters.opt_qmask, db, parameters);
  struct cluster_session_s * cs = cluster_session_alloc();
  cluster_session_init(cs, parameters, dbindex, db);
  std::vector<struct cluster_result_s> seq_results(sc), batch_results(sc);
  for (int i = 0; i < sc; ++i) cluster_assign_single(cs, i, &seq_results[i]);
  cluster_session_cleanup(cs);
  cluster_session_free(cs);
  dbindex.clear();

  dbindex.prepare(1, parameters.opt_qmask, db, parameters);
  cs = cluster_session_alloc();
  cluster_session_init(cs, parameters, dbindex, db);
  for (int i = 0; i < sc; i += batch) cluster_assign_batch(cs, i, std::min(batch, sc - i), &batch_results[i]);
  cluster_session_cleanup(cs);
  cluster_session_free(cs);
  dbindex.clear();

  long bad = 0, members = 0, clusters = 0;
  for (int i = 0; i < sc; ++i)
    {
      struct cluster_result_s const & a = seq_results[i];
      struct cluster_result_s const & b = batch_results[i];
      clusters += a.is_centroid;
      members += !a.is_centroid;
      bool diff = a.is_centroid != b.is_centroid || a.cluster_id != b.cluster_id || a.centroid_seqno != b.centroid_seqno ||
                  std::strcmp(a.centroid_label, b.centroid_label) != 0 || a.identity != b.identity ||
                  std::strcmp(a.cigar, b.cigar) != 0 || a.cigar_truncated != b.cigar_truncated;
      if (diff && bad++ < 10)
        std::fprintf(stderr, "sequence %d: batch cen%d cid%d cseq%d id %.4f %s / sequential cen%d cid%d cseq%d id %.4f %s\n", i,
                     b.is_centroid, b.cluster_id, b.centroid_seqno, b.identity, b.cigar, a.is_centroid, a.cluster_id, a.centroid_seqno, a.identity, a.cigar);
    }
  std::printf("cluster: %d sequences, %ld clusters, %ld members, %ld differences\n", sc, clusters, members, bad);
  db.clear();
  vsearch_session_end();
  return bad == 0 ? 0 : 1;
}

int main(int argc, char ** argv)
{
  if (argc >= 2 && std::strcmp(argv[1], "search") == 0) return run_search(argc, argv);
  if (argc >= 2 && std::strcmp(argv[1], "cluster") == 0) return run_cluster(argc, argv);
  std::fprintf(stderr, "usage: api_driver search|cluster ...\n");
  return 2;
}
