"""ctypes binding of libvsx.so (the C-ABI declared in include/vsx.h).

The shared library is built IN-TREE (vsearch_amd/libvsx.so, `make -C vsearch_amd/csrc` or
__graft_entry__.build()).  There is no CPU fallback: if the library is missing, loading fails
loudly; if no gfx950 device is visible every compute entry point returns VSX_ENODEVICE.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# VSX_LIBRARY: another build of the same library (A/B builds of the kernels under build/variants/); default: the in-tree one
LIB_PATH = os.environ.get("VSX_LIBRARY") or os.path.join(HERE, "libvsx.so")

VSX_OK, VSX_EINVAL, VSX_ENODEVICE, VSX_ENOMEM, VSX_EHIP = 0, -1, -2, -3, -4
SENTINEL = 32767

# every symbol include/vsx.h declares
SYMBOLS = [
    "vsx_version_string", "vsx_device_count", "vsx_last_error", "vsx_create", "vsx_destroy",
    "vsx_seqset_create", "vsx_seqset_create_both_strands", "vsx_seqset_create_from_device", "vsx_seqset_destroy", "vsx_seqset_count",
    "vsx_plan_create", "vsx_plan_run", "vsx_plan_sync", "vsx_plan_fetch", "vsx_plan_export_hits", "vsx_plan_export_runs",
    "vsx_cigar_from_runs", "vsx_plan_destroy",
    "vsx_align_pairs", "vsx_align_pairs_filtered", "vsx_align_pairs_ranked", "vsx_ranked_free", "vsx_plan_set_filter", "vsx_results_free", "vsx_plan_describe",
]
# include/vsx_search.h
SEARCH_SYMBOLS = ["vsx_search_opts_default", "vsx_searcher_create", "vsx_searcher_destroy", "vsx_searcher_db_text", "vsx_search_batch",
                  "vsx_search_batch_meta", "vsx_searcher_set_meta",
                  "vsx_hits_free", "vsx_search_candidates", "vsx_search_candidates_batch", "vsx_candidates_free", "vsx_lma_align", "vsx_allpairs_block", "vsx_allpairs_rows", "vsx_allpairs_stream", "vsx_cluster_fast", "vsx_cluster_out_free", "vsx_msa", "vsx_msa_out_free",
                  "vsx_msa_device", "vsx_msa_device_batch", "vsx_dust_mask", "vsx_abundance_ratio_cmp",
                  "vsx_multi_searcher_create", "vsx_multi_searcher_destroy", "vsx_multi_searcher_devices", "vsx_multi_searcher_replica",
                  "vsx_multi_search_batch", "vsx_multi_allpairs"]


class Candidates(C.Structure):
    """vsx_candidates (include/vsx_search.h)"""
    _fields_ = [("n_queries", C.c_uint64), ("start", C.POINTER(C.c_uint64)), ("target", C.POINTER(C.c_uint32)),
                ("count", C.POINTER(C.c_uint32)), ("seconds", C.c_double), ("kernel_ms", C.c_double),
                ("index_build_ms", C.c_double), ("index_postings", C.c_uint64), ("postings_streamed", C.c_uint64),
                ("bytes_streamed", C.c_uint64)]


class SearchOpts(C.Structure):
    """vsx_search_opts (include/vsx_search.h): the Parameters fields the search path reads"""
    _fields_ = [("id", C.c_double), ("weak_id", C.c_double), ("maxaccepts", C.c_int64), ("maxrejects", C.c_int64),
                ("wordlength", C.c_int64), ("minwordmatches", C.c_int64), ("iddef", C.c_int32), ("soft_mask", C.c_int32),
                ("maxsubs", C.c_int64), ("maxgaps", C.c_int64), ("mincols", C.c_int64), ("maxdiffs", C.c_int64),
                ("query_cov", C.c_double), ("target_cov", C.c_double), ("maxid", C.c_double), ("mid", C.c_double),
                ("leftjust", C.c_int32), ("rightjust", C.c_int32),
                ("minqt", C.c_double), ("maxqt", C.c_double), ("minsl", C.c_double), ("maxsl", C.c_double),
                ("idprefix", C.c_int64), ("idsuffix", C.c_int64), ("selfid", C.c_int32), ("threads", C.c_int32),
                ("window", C.c_int64), ("gap_infinite", C.c_uint32), ("strand_both", C.c_uint32),
                ("maxqsize", C.c_int64), ("mintsize", C.c_int64), ("minsizeratio", C.c_double), ("maxsizeratio", C.c_double),
                ("self", C.c_int32), ("sizeorder", C.c_int32), ("cluster_unoise", C.c_int32), ("qmask", C.c_int32),
                ("unoise_alpha", C.c_double), ("hardmask", C.c_int32)]


class SeqMeta(C.Structure):
    """vsx_seq_meta (include/vsx_search.h): abundances / labels of a set of sequences"""
    _fields_ = [("abundance", C.POINTER(C.c_uint64)), ("label", C.POINTER(C.c_char_p))]


class Hit(C.Structure):
    _fields_ = [("query", C.c_uint32), ("target", C.c_uint32), ("count", C.c_uint32),
                ("accepted", C.c_uint8), ("weak", C.c_uint8), ("used_fallback", C.c_uint8), ("strand", C.c_uint8),
                ("nwscore", C.c_int32), ("nwdiff", C.c_int32), ("nwgaps", C.c_int32), ("nwindels", C.c_int32),
                ("nwalignmentlength", C.c_int32), ("matches", C.c_int32), ("mismatches", C.c_int32),
                ("internal_alignmentlength", C.c_int32), ("internal_gaps", C.c_int32), ("internal_indels", C.c_int32),
                ("trim_q_left", C.c_int32), ("trim_q_right", C.c_int32), ("trim_t_left", C.c_int32), ("trim_t_right", C.c_int32),
                ("shortest", C.c_int32), ("longest", C.c_int32),
                ("nwid", C.c_double), ("id", C.c_double), ("id0", C.c_double), ("id1", C.c_double), ("id2", C.c_double),
                ("id3", C.c_double), ("id4", C.c_double), ("cigar_off", C.c_uint64)]


class Hits(C.Structure):
    _fields_ = [("n_queries", C.c_uint64), ("n_hits", C.c_uint64), ("first", C.POINTER(C.c_uint64)),
                ("hit", C.POINTER(Hit)), ("cigar_blob", C.POINTER(C.c_char)), ("cigar_bytes", C.c_uint64),
                ("pairs_aligned", C.c_uint64), ("cells_aligned", C.c_uint64), ("stages", C.c_uint64),
                ("sentinel_pairs", C.c_uint64), ("seconds_kmer", C.c_double), ("seconds_align", C.c_double),
                ("seconds_total", C.c_double)]


class Scoring(C.Structure):
    """vsx_scoring: the 14 post-fixup values in search16_init order + n_mismatch."""
    _fields_ = [(n, C.c_int64) for n in (
        "match", "mismatch", "gap_open_query_left", "gap_open_target_left", "gap_open_query_interior",
        "gap_open_target_interior", "gap_open_query_right", "gap_open_target_right",
        "gap_ext_query_left", "gap_ext_target_left", "gap_ext_query_interior", "gap_ext_target_interior",
        "gap_ext_query_right", "gap_ext_target_right")] + [("n_mismatch", C.c_int32)]


class Results(C.Structure):
    _fields_ = [("n_pairs", C.c_uint64), ("score", C.POINTER(C.c_int16)), ("aligned", C.POINTER(C.c_uint16)),
                ("matches", C.POINTER(C.c_uint16)), ("mismatches", C.POINTER(C.c_uint16)),
                ("gaps", C.POINTER(C.c_uint16)), ("cigar_off", C.POINTER(C.c_uint64)),
                ("cigar_blob", C.POINTER(C.c_char)), ("cigar_bytes", C.c_uint64), ("verdict", C.POINTER(C.c_uint8))]


class Ranked(C.Structure):
    """vsx_ranked (include/vsx.h)"""
    _fields_ = [("n_pairs", C.c_uint64), ("n_hits", C.c_uint64), ("pair", C.POINTER(C.c_uint32)), ("score", C.POINTER(C.c_int16)),
                ("aligned", C.POINTER(C.c_uint16)), ("matches", C.POINTER(C.c_uint16)), ("mismatches", C.POINTER(C.c_uint16)),
                ("gaps", C.POINTER(C.c_uint16)), ("verdict", C.POINTER(C.c_uint8)), ("id", C.POINTER(C.c_double)),
                ("cigar_off", C.POINTER(C.c_uint64)), ("cigar_blob", C.POINTER(C.c_char)), ("cigar_bytes", C.c_uint64),
                ("n_undecided", C.c_uint64), ("undecided", C.POINTER(C.c_uint32))]


class Filter(C.Structure):
    """vsx_filter (include/vsx.h): device-side align_trim + search_acceptable_aligned"""
    _fields_ = [("iddef", C.c_int32), ("leftjust", C.c_int32), ("rightjust", C.c_int32), ("pad", C.c_int32),
                ("id", C.c_double), ("weak_id", C.c_double), ("maxid", C.c_double), ("mid", C.c_double),
                ("query_cov", C.c_double), ("target_cov", C.c_double),
                ("maxsubs", C.c_int64), ("maxgaps", C.c_int64), ("mincols", C.c_int64), ("maxdiffs", C.c_int64)]


class PlanInfo(C.Structure):
    """vsx_plan_info (include/vsx.h)"""
    _fields_ = [("tasks", C.c_uint64), ("tasks_tilted", C.c_uint64), ("tasks_tracked", C.c_uint64),
                ("rows_dominant", C.c_uint32), ("chunks", C.c_uint32), ("tasks_max3", C.c_uint64),
                ("tasks_sparse", C.c_uint64), ("waves", C.c_uint64), ("tasks_pair", C.c_uint64)]


class Timing(C.Structure):
    _fields_ = [("forward_ms", C.c_float), ("traceback_ms", C.c_float), ("total_ms", C.c_float),
                ("forward_launches", C.c_uint32), ("traceback_launches", C.c_uint32),
                ("cells", C.c_uint64), ("dir_bytes", C.c_uint64)]


class ClusterOut(C.Structure):
    _fields_ = [("n", C.c_uint64), ("n_clusters", C.c_uint64), ("clusterno", C.POINTER(C.c_uint32)), ("hits", Hits)]


class MsaOut(C.Structure):
    _fields_ = [("alnlen", C.c_uint64), ("n_rows", C.c_uint64), ("conslen", C.c_uint64), ("rows", C.POINTER(C.c_char)),
                ("consensus", C.POINTER(C.c_char)), ("profile", C.POINTER(C.c_uint64))]


_lib = None


def load():
    """dlopen libvsx.so and declare the prototypes.  Raises if the extension has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build the HIP extension first (python -c 'import __graft_entry__ as g; "
            "g.build()' or make -C vsearch_amd/csrc). vsearch_amd has no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    vp = C.c_void_p
    lib.vsx_version_string.restype = C.c_char_p
    lib.vsx_last_error.restype = C.c_char_p
    lib.vsx_device_count.restype = C.c_int
    lib.vsx_create.argtypes = [C.POINTER(vp), C.POINTER(Scoring), C.c_int]
    lib.vsx_destroy.argtypes = [vp]
    lib.vsx_destroy.restype = None
    lib.vsx_seqset_create.argtypes = [vp, C.POINTER(vp), C.c_uint64, vp, C.c_uint64, vp, vp]
    lib.vsx_seqset_create_both_strands.argtypes = [vp, C.POINTER(vp), C.c_uint64, vp, C.c_uint64, vp, vp]
    lib.vsx_seqset_create_from_device.argtypes = [vp, C.POINTER(vp), C.c_uint64, vp, C.c_uint64, vp, vp]
    lib.vsx_seqset_destroy.argtypes = [vp]
    lib.vsx_seqset_destroy.restype = None
    lib.vsx_seqset_count.argtypes = [vp]
    lib.vsx_seqset_count.restype = C.c_uint64
    lib.vsx_plan_create.argtypes = [vp, C.POINTER(vp), vp, vp, C.c_uint64, vp, vp, C.c_uint64]
    lib.vsx_plan_run.argtypes = [vp]
    lib.vsx_plan_sync.argtypes = [vp, C.POINTER(Timing)]
    lib.vsx_plan_fetch.argtypes = [vp, C.POINTER(Results)]
    lib.vsx_plan_export_hits.argtypes = [vp, vp, C.c_uint64]
    lib.vsx_plan_export_runs.argtypes = [vp, vp, C.c_uint64, C.POINTER(C.c_uint64)]
    lib.vsx_cigar_from_runs.argtypes = [vp, C.c_uint32, vp, C.c_uint64]
    lib.vsx_cigar_from_runs.restype = C.c_int64
    lib.vsx_plan_destroy.argtypes = [vp]
    lib.vsx_plan_destroy.restype = None
    lib.vsx_plan_describe.argtypes = [vp, C.POINTER(PlanInfo)]
    lib.vsx_align_pairs.argtypes = [vp, vp, vp, C.c_uint64, vp, vp, C.POINTER(Results)]
    lib.vsx_align_pairs_filtered.argtypes = [vp, vp, vp, C.c_uint64, vp, vp, C.POINTER(Filter), C.POINTER(Results)]
    lib.vsx_align_pairs_ranked.argtypes = [vp, vp, vp, C.c_uint64, vp, vp, C.POINTER(Filter), C.c_int, C.POINTER(Ranked)]
    lib.vsx_ranked_free.argtypes = [C.POINTER(Ranked)]
    lib.vsx_ranked_free.restype = None
    lib.vsx_plan_set_filter.argtypes = [vp, C.POINTER(Filter)]
    lib.vsx_results_free.argtypes = [C.POINTER(Results)]
    lib.vsx_results_free.restype = None
    lib.vsx_search_opts_default.argtypes = [C.POINTER(SearchOpts)]
    lib.vsx_search_opts_default.restype = None
    lib.vsx_searcher_create.argtypes = [vp, C.POINTER(vp), C.POINTER(SearchOpts), C.c_uint64, vp, C.c_uint64, vp, vp]
    lib.vsx_searcher_destroy.argtypes = [vp]
    lib.vsx_searcher_destroy.restype = None
    lib.vsx_search_batch.argtypes = [vp, C.c_uint64, vp, C.c_uint64, vp, vp, C.POINTER(Hits)]
    lib.vsx_search_batch_meta.argtypes = [vp, C.c_uint64, vp, C.c_uint64, vp, vp, C.POINTER(SeqMeta), C.POINTER(Hits)]
    lib.vsx_searcher_set_meta.argtypes = [vp, C.POINTER(SeqMeta)]
    lib.vsx_hits_free.argtypes = [C.POINTER(Hits)]
    lib.vsx_hits_free.restype = None
    lib.vsx_search_candidates.argtypes = [vp, C.c_char_p, C.c_uint32, vp, vp, C.c_uint64]
    lib.vsx_search_candidates.restype = C.c_int64
    lib.vsx_search_candidates_batch.argtypes = [vp, C.c_int32, C.c_uint64, vp, C.c_uint64, vp, vp, C.POINTER(Candidates)]
    lib.vsx_candidates_free.argtypes = [C.POINTER(Candidates)]
    lib.vsx_candidates_free.restype = None
    lib.vsx_allpairs_block.argtypes = [vp, C.c_int32, C.c_uint64, C.c_uint64, C.POINTER(Hits)]
    lib.vsx_allpairs_rows.argtypes = [vp, C.c_int32, vp, C.c_uint64, C.POINTER(Hits)]
    lib.vsx_allpairs_stream.argtypes = [vp, C.c_int32, C.c_uint64, C.c_uint64, C.c_uint64, vp, vp]     # (sink: a CFUNCTYPE object, passed as a pointer)
    lib.vsx_dust_mask.argtypes = [vp, C.c_uint64, vp, vp, C.c_int32]
    lib.vsx_multi_searcher_create.argtypes = [C.POINTER(vp), vp, vp, C.c_int32, vp, C.c_uint64, vp, C.c_uint64, vp, vp, vp]
    lib.vsx_multi_searcher_destroy.argtypes = [vp]
    lib.vsx_multi_searcher_destroy.restype = None
    lib.vsx_multi_searcher_devices.argtypes = [vp]
    lib.vsx_multi_searcher_devices.restype = C.c_int32
    lib.vsx_multi_searcher_replica.argtypes = [vp, C.c_int32]
    lib.vsx_multi_searcher_replica.restype = vp
    lib.vsx_multi_search_batch.argtypes = [vp, C.c_uint64, vp, C.c_uint64, vp, vp, vp, C.POINTER(Hits)]
    lib.vsx_multi_allpairs.argtypes = [vp, C.c_int32, C.c_uint64, C.c_uint64, C.POINTER(Hits)]
    lib.vsx_abundance_ratio_cmp.argtypes = [C.c_int64, C.c_double, C.c_int64]
    lib.vsx_abundance_ratio_cmp.restype = C.c_int
    lib.vsx_cluster_fast.argtypes = [vp, C.c_uint64, C.POINTER(ClusterOut)]
    lib.vsx_cluster_out_free.argtypes = [C.POINTER(ClusterOut)]
    lib.vsx_cluster_out_free.restype = None
    lib.vsx_msa.argtypes = [C.c_uint32, vp, vp, vp, vp, C.POINTER(MsaOut)]
    lib.vsx_msa_device.argtypes = [vp, C.c_uint32, vp, vp, vp, vp, C.POINTER(MsaOut)]
    lib.vsx_msa_device_batch.argtypes = [vp, C.c_uint32, vp, vp, vp, vp, vp, C.POINTER(MsaOut)]
    lib.vsx_msa_out_free.argtypes = [C.POINTER(MsaOut)]
    lib.vsx_msa_out_free.restype = None
    lib.vsx_lma_align.argtypes = [C.POINTER(Scoring), C.c_char_p, C.c_uint64, C.c_char_p, C.c_uint64] + \
        [C.POINTER(C.c_int64)] * 5 + [C.POINTER(vp)]
    _lib = lib
    return lib


class VsxError(RuntimeError):
    def __init__(self, code, where):
        msg = load().vsx_last_error().decode(errors="replace")
        super().__init__(f"{where}: error {code}: {msg}")
        self.code = code


def check(code, where):
    if code != VSX_OK:
        raise VsxError(code, where)
