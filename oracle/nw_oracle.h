/*
  oracle/nw_oracle.h -- TEST INFRASTRUCTURE ONLY.

  Scalar CPU restatement of the reference's global aligner (search16 and its
  backtrack) and of the caller-side post-processing (align_trim).  Only tests/,
  __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this; the
  product path (vsearch_amd/, libvsx.so) never does.

  Parity status: PINNED -- tests/test_oracle.py checks this restatement against
  (a) fixtures generated from the reference's own compiled sources
      (oracle/_ref/libvsref.so, script oracle/gen_golden.py), committed under
      tests/golden/, and
  (b) the reference's in-tree golden files for this path
      (api_examples/data/expected_search.tsv / expected_cluster.uc values,
      copied as data into tests/golden/ref_api_examples.json by the script).
*/
#ifndef VSX_NW_ORACLE_H
#define VSX_NW_ORACLE_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* P = (match, mismatch, go_q_l, go_t_l, go_q_i, go_t_i, go_q_r, go_t_r,
        ge_q_l, ge_t_l, ge_q_i, ge_t_i, ge_q_r, ge_t_r): the post-fixup values in
   the argument order of search16_init (reference core/align_simd.hpp:76-90). */

/* ASCII -> 4-bit IUPAC set code (reference utils/maps.cpp:75-117). */
unsigned char vsxo_map4(unsigned char c);

/* One pair through the reference's search16 semantics.
   score == 32767 means "not aligned by the 16-bit path" (stats 0, cigar "").
   cigar must have room for qlen+dlen+2 bytes (run-length text, NUL-terminated).
   Returns 0, or -1 on allocation failure. */
int vsxo_search16_pair(const char * q, int64_t qlen, const char * d, int64_t dlen,
                       const int64_t P[14], int n_mismatch,
                       int16_t * score, uint16_t * aligned, uint16_t * matches,
                       uint16_t * mismatches, uint16_t * gaps, char * cigar);

/* Batch form over a blob: pair k aligns query qi[k] with target ti[k].
   cigar_blob receives the strings back to back (each NUL-terminated),
   cigar_off[k] the start of pair k; capacity in bytes; returns bytes used or -1. */
int64_t vsxo_search16_batch(const char * qblob, const uint64_t * qoff, const uint32_t * qlen,
                            const char * tblob, const uint64_t * toff, const uint32_t * tlen,
                            uint64_t npairs, const uint32_t * qi, const uint32_t * ti,
                            const int64_t P[14], int n_mismatch,
                            int16_t * score, uint16_t * aligned, uint16_t * matches,
                            uint16_t * mismatches, uint16_t * gaps,
                            char * cigar_blob, uint64_t * cigar_off, uint64_t capacity);

#ifdef __cplusplus
}
#endif
#endif
