"""vsearch_amd -- MI355X-native (gfx950) global pairwise alignment for vsearch workloads.

The package is a thin host-side mirror of the reference's aligner interface over libvsx's C-ABI
(include/vsx.h); all arithmetic runs in hand-written HIP kernels (vsearch_amd/csrc).
"""
from .aligner import (Aligner, SequenceSet, Plan, AlignmentResults, RawResults, DEFAULT_SCORING,  # noqa: F401
                      scoring_from_tuple, cigar_from_runs)
from .search import SearchSession, msa, msa_batch, dust_mask  # noqa: F401
from ._lib import SENTINEL, VsxError, load as load_library  # noqa: F401

__all__ = ["Aligner", "SequenceSet", "Plan", "AlignmentResults", "RawResults", "cigar_from_runs", "DEFAULT_SCORING", "scoring_from_tuple",
           "SearchSession", "SENTINEL", "VsxError", "load_library"]
