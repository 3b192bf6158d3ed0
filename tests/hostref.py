"""Small pure-Python restatements of caller-side post-processing used by the CPU tests
(reference: src/core/searchcore.cpp align_trim :343-464; iddef 2 = 100 * matches / internal alignment length)."""
import re

_RUN = re.compile(r"(\d*)([MID])")


def cigar_runs(cigar):
    return [(int(n) if n else 1, op) for n, op in _RUN.findall(cigar)]


def id_iddef2(cigar, matches):
    """%id as --iddef 2 (default): terminal gaps are trimmed from the alignment length (align_trim)"""
    runs = cigar_runs(cigar)
    total = sum(n for n, _ in runs)
    left = runs[0][0] if runs and runs[0][1] != "M" else 0
    right = runs[-1][0] if len(runs) > 1 and runs[-1][1] != "M" else 0
    internal = total - left - right
    return 100.0 * matches / internal if internal > 0 else 0.0
