"""Host-side mirror of the reference's aligner interface (src/core/align_simd.hpp:76-108) on top of
the libvsx C-ABI.  Names and argument meaning follow the reference:

    search16_init(...)   -> Aligner(...)            14 post-fixup penalties + n_mismatch
    search16_qprep(q)    -> Aligner.qprep(q)
    search16(seqnos, db) -> Aligner.search16(seqnos, db)  -> per-target (score, aligned, matches,
                                                             mismatches, gaps, cigar)
    Database             -> SequenceSet                    device-resident, 4-bit coded
    (new) multi-query batch: Aligner.align_pairs(queries, targets, qidx, tidx)

Unalignable pairs come back exactly as the reference reports them: score 32767, stats 0, cigar "".
"""
import ctypes as C
import weakref

import numpy as np

from . import _lib
from ._lib import Scoring, Results, Timing, check

# post-fixup defaults: src/vsearch.h:450-461 after vsearch_apply_defaults_fixups (src/vsearch.cc:250-259)
DEFAULT_SCORING = dict(match=2, mismatch=-4,
                       gap_open_query_left=1, gap_open_target_left=1,
                       gap_open_query_interior=18, gap_open_target_interior=18,
                       gap_open_query_right=1, gap_open_target_right=1,
                       gap_ext_query_left=1, gap_ext_target_left=1,
                       gap_ext_query_interior=2, gap_ext_target_interior=2,
                       gap_ext_query_right=1, gap_ext_target_right=1)
P_ORDER = [n for n, _ in Scoring._fields_[:14]]


def scoring_from_tuple(P, n_mismatch=False):
    """P in search16_init argument order (match, mismatch, go_q_l, go_t_l, go_q_i, go_t_i, go_q_r,
    go_t_r, ge_q_l, ge_t_l, ge_q_i, ge_t_i, ge_q_r, ge_t_r)."""
    s = Scoring()
    for n, v in zip(P_ORDER, P):
        setattr(s, n, int(v))
    s.n_mismatch = 1 if n_mismatch else 0
    return s


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


class AlignmentResults:
    """numpy view of vsx_results (copied out, then freed)."""

    def __init__(self, res):
        n = int(res.n_pairs)
        cp = lambda p, dt: np.ctypeslib.as_array(p, shape=(n,)).astype(dt, copy=True) if n else np.zeros(0, dt)
        self.score = cp(res.score, np.int16)
        self.aligned = cp(res.aligned, np.uint16)
        self.matches = cp(res.matches, np.uint16)
        self.mismatches = cp(res.mismatches, np.uint16)
        self.gaps = cp(res.gaps, np.uint16)
        off = cp(res.cigar_off, np.uint64)
        blob = C.string_at(res.cigar_blob, int(res.cigar_bytes)) if res.cigar_bytes else b""
        self.cigar = [blob[int(o):blob.index(b"\0", int(o))].decode() for o in off]
        # VSX_VERDICT_* per pair when the plan carried a filter (0 undecided, 1 accepted, 2 weak, 3 rejected)
        self.verdict = (np.ctypeslib.as_array(res.verdict, shape=(n,)).astype(np.uint8, copy=True)
                        if (n and bool(res.verdict)) else None)

    def __len__(self):
        return len(self.score)

    def row(self, k):
        return (int(self.score[k]), int(self.aligned[k]), int(self.matches[k]), int(self.mismatches[k]),
                int(self.gaps[k]), self.cigar[k])


class RawResults:
    """vsx_results as handed out by the library (malloc'd arrays, freed on close): zero-copy numpy views and CIGARs on
    demand -- for callers that consume a few fields of many pairs (bench.py's end-to-end loop, the sharded gather)."""

    def __init__(self, res):
        self._res = res
        n = self.n = int(res.n_pairs)
        view = lambda p: np.ctypeslib.as_array(p, shape=(n,)) if n else np.zeros(0, np.uint16)
        self.score, self.aligned, self.matches = view(res.score), view(res.aligned), view(res.matches)
        self.mismatches, self.gaps, self.cigar_off = view(res.mismatches), view(res.gaps), view(res.cigar_off)
        self.verdict = np.ctypeslib.as_array(res.verdict, shape=(n,)) if (n and bool(res.verdict)) else None
        self.cigar_bytes = int(res.cigar_bytes)

    def cigar(self, k):
        return C.string_at(C.addressof(self._res.cigar_blob.contents) + int(self.cigar_off[k])).decode()

    def row(self, k):
        return (int(self.score[k]), int(self.aligned[k]), int(self.matches[k]), int(self.mismatches[k]),
                int(self.gaps[k]), self.cigar(k))

    def __len__(self):
        return self.n

    def close(self):
        if self._res is not None:
            self.score = self.aligned = self.matches = self.mismatches = self.gaps = self.cigar_off = self.verdict = None
            _lib.load().vsx_results_free(C.byref(self._res))
            self._res = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class SequenceSet:
    """Device-resident sequences (mirror of Database: core/db.hpp getsequence/getsequencelen)."""

    def __init__(self, aligner, seqs=None, blob=None, offsets=None, lengths=None, device_ptr=None, both_strands=False):
        self.aligner = aligner
        aligner._children.add(self)
        lib = _lib.load()
        if seqs is not None:
            bs = [s.encode() if isinstance(s, str) else bytes(s) for s in seqs]
            lengths = np.array([len(b) for b in bs], np.uint32)
            offsets = np.zeros(len(bs), np.uint64)
            if len(bs):
                offsets[1:] = np.cumsum(lengths[:-1], dtype=np.uint64)
            blob = b"".join(bs)
        self.offsets = np.ascontiguousarray(offsets, np.uint64)
        self.lengths = np.ascontiguousarray(lengths, np.uint32)
        self.n = len(self.lengths)
        self.h = C.c_void_p()
        if device_ptr is not None:
            nbytes = int(blob)          # blob carries the byte count when the data is already in HBM
            check(lib.vsx_seqset_create_from_device(aligner.h, C.byref(self.h), self.n, C.c_void_p(device_ptr),
                                                    nbytes, _ptr(self.offsets), _ptr(self.lengths)),
                  "vsx_seqset_create_from_device")
        else:
            if isinstance(blob, np.ndarray):
                self._keep = np.ascontiguousarray(blob, np.uint8)
                p, nbytes = _ptr(self._keep), self._keep.size
            else:
                self._keep = bytes(blob)
                p, nbytes = C.cast(C.c_char_p(self._keep), C.c_void_p), len(self._keep)
            if both_strands:
                # sequences n .. 2n-1 = the reverse complements, computed on the device (vsx_seqset_create_both_strands)
                check(lib.vsx_seqset_create_both_strands(aligner.h, C.byref(self.h), self.n, p, nbytes,
                                                         _ptr(self.offsets), _ptr(self.lengths)), "vsx_seqset_create_both_strands")
                self.n *= 2
            else:
                check(lib.vsx_seqset_create(aligner.h, C.byref(self.h), self.n, p, nbytes,
                                            _ptr(self.offsets), _ptr(self.lengths)), "vsx_seqset_create")

    def close(self):
        if getattr(self, "h", None) and self.h.value:
            _lib.load().vsx_seqset_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __len__(self):
        return self.n


def cigar_from_runs(runs):
    """vsx_cigar_from_runs: run words ((length << 2) | op, traceback order) -> CIGAR text; host only"""
    r = np.ascontiguousarray(runs, np.uint32)
    buf = C.create_string_buffer(7 * r.size + 1)
    _lib.load().vsx_cigar_from_runs(_ptr(r), r.size, buf, len(buf))
    return buf.value.decode()


def make_filter(**kw):
    """a vsx_filter with defaults that accept everything, overridden by the keyword arguments"""
    f = _lib.Filter()
    f.iddef, f.id, f.weak_id, f.maxid = 2, 0.0, 0.0, 1.0
    f.maxsubs = f.maxgaps = f.maxdiffs = 2 ** 31 - 1
    for k, v in kw.items():
        if not hasattr(f, k):
            raise TypeError(f"unknown filter field {k}")
        setattr(f, k, v)
    return f


class Plan:
    """A batch of (query, target) pairs bound to device buffers (vsx_plan)."""

    def __init__(self, aligner, queries, targets, qidx, tidx, dir_budget_bytes=0):
        self.aligner, self.queries, self.targets = aligner, queries, targets
        aligner._children.add(self)
        self.qidx = np.ascontiguousarray(qidx, np.uint32)
        self.tidx = np.ascontiguousarray(tidx, np.uint32)
        if self.qidx.shape != self.tidx.shape:
            raise ValueError("qidx and tidx must have the same length")
        self.h = C.c_void_p()
        check(_lib.load().vsx_plan_create(aligner.h, C.byref(self.h), queries.h, targets.h, self.qidx.size,
                                          _ptr(self.qidx), _ptr(self.tidx), int(dir_budget_bytes)),
              "vsx_plan_create")

    def set_filter(self, **kw):
        """device-side accept filter (vsx_filter): iddef, id, weak_id, maxid, mid, query_cov, target_cov, maxsubs, maxgaps,
        mincols, maxdiffs, leftjust, rightjust; call before run().  No arguments = defaults that accept everything."""
        check(_lib.load().vsx_plan_set_filter(self.h, C.byref(make_filter(**kw))), "vsx_plan_set_filter")

    def run(self):
        check(_lib.load().vsx_plan_run(self.h), "vsx_plan_run")

    def sync(self):
        t = Timing()
        check(_lib.load().vsx_plan_sync(self.h, C.byref(t)), "vsx_plan_sync")
        return t

    def describe(self):
        """vsx_plan_describe: dict(tasks, tasks_tilted, tasks_tracked, rows_dominant, chunks)"""
        info = _lib.PlanInfo()
        check(_lib.load().vsx_plan_describe(self.h, C.byref(info)), "vsx_plan_describe")
        return {k: int(getattr(info, k)) for k, _ in info._fields_}

    def fetch(self):
        res = Results()
        lib = _lib.load()
        check(lib.vsx_plan_fetch(self.h, C.byref(res)), "vsx_plan_fetch")
        try:
            return AlignmentResults(res)
        finally:
            lib.vsx_results_free(C.byref(res))

    def export_hits(self, device_ptr, nbytes):
        """copy the 24-byte hit records of the last run into device memory (e.g. a torch tensor)"""
        check(_lib.load().vsx_plan_export_hits(self.h, C.c_void_p(device_ptr), int(nbytes)), "vsx_plan_export_hits")

    def export_runs(self, device_ptr=None, nbytes=0):
        """vsx_plan_export_runs: number of run words of the last run; with a destination, also copies them there (device memory)"""
        n = C.c_uint64(0)
        check(_lib.load().vsx_plan_export_runs(self.h, C.c_void_p(device_ptr) if device_ptr else None, int(nbytes), C.byref(n)),
              "vsx_plan_export_runs")
        return int(n.value)

    def close(self):
        if getattr(self, "h", None) and self.h.value:
            _lib.load().vsx_plan_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Aligner:
    """search16_init .. search16_exit.  One per (device, scoring); not thread-safe, like s16info_s."""

    def __init__(self, scoring=None, n_mismatch=False, device=0, **kw):
        lib = _lib.load()
        if scoring is None:
            d = dict(DEFAULT_SCORING)
            d.update(kw)
            s = Scoring()
            for k, v in d.items():
                setattr(s, k, int(v))
            s.n_mismatch = 1 if n_mismatch else 0
        elif isinstance(scoring, Scoring):
            s = scoring
        else:
            s = scoring_from_tuple(scoring, n_mismatch)
        self.scoring = s
        self._children = weakref.WeakSet()      # plans / sequence sets must die before the context
        self.h = C.c_void_p()
        check(lib.vsx_create(C.byref(self.h), C.byref(s), int(device)), "vsx_create")
        self._q = None

    def close(self):
        if getattr(self, "h", None) and self.h.value:
            kids = list(self._children)
            # plans first, then searchers / sequence sets
            for k in [k for k in kids if isinstance(k, Plan)] + [k for k in kids if not isinstance(k, Plan)]:
                k.close()
            self._q = None
            _lib.load().vsx_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    # -- the reference's per-query interface -------------------------------------------------
    def sequences(self, seqs, both_strands=False):
        return SequenceSet(self, seqs=seqs, both_strands=both_strands)

    def qprep(self, qseq):
        """search16_qprep (align_simd.cpp:1406-1428): bind the query of the following search16 calls."""
        if self._q is not None:
            self._q.close()
        self._q = SequenceSet(self, seqs=[qseq])

    def search16(self, seqnos, db):
        """search16 (align_simd.cpp:1447-2060): align the bound query against db[seqnos]."""
        if self._q is None:
            raise RuntimeError("search16 called before qprep")
        seqnos = np.ascontiguousarray(seqnos, np.uint32)
        return self.align_pairs(self._q, db, np.zeros(seqnos.size, np.uint32), seqnos)

    # -- the batched entry ---------------------------------------------------------------------
    def plan(self, queries, targets, qidx, tidx, dir_budget_bytes=0):
        return Plan(self, queries, targets, qidx, tidx, dir_budget_bytes)

    def align_pairs(self, queries, targets, qidx, tidx):
        p = self.plan(queries, targets, qidx, tidx)
        try:
            p.run()
            return p.fetch()
        finally:
            p.close()

    def align_pairs_oneshot(self, queries, targets, qidx, tidx, filter=None):
        """vsx_align_pairs / vsx_align_pairs_filtered: plan + run + fetch in one C call (large lists are pipelined as several
        plans inside the library); `filter` = dict of vsx_filter fields or None"""
        lib = _lib.load()
        qidx = np.ascontiguousarray(qidx, np.uint32)
        tidx = np.ascontiguousarray(tidx, np.uint32)
        res = Results()
        f = make_filter(**filter) if filter is not None else None
        check(lib.vsx_align_pairs_filtered(self.h, queries.h, targets.h, qidx.size, qidx.ctypes.data_as(C.c_void_p),
                                           tidx.ctypes.data_as(C.c_void_p), C.byref(f) if f is not None else None, C.byref(res)),
              "vsx_align_pairs_filtered")
        try:
            return AlignmentResults(res)
        finally:
            lib.vsx_results_free(C.byref(res))

    def align_pairs_raw(self, queries, targets, qidx, tidx, filter=None):
        """vsx_align_pairs[_filtered] without the Python-side copies: returns RawResults (close() it)"""
        lib = _lib.load()
        qidx = np.ascontiguousarray(qidx, np.uint32)
        tidx = np.ascontiguousarray(tidx, np.uint32)
        res = Results()
        f = make_filter(**filter) if filter is not None else None
        check(lib.vsx_align_pairs_filtered(self.h, queries.h, targets.h, qidx.size, qidx.ctypes.data_as(C.c_void_p),
                                           tidx.ctypes.data_as(C.c_void_p), C.byref(f) if f is not None else None, C.byref(res)),
              "vsx_align_pairs_filtered")
        return RawResults(res)

    def align_pairs_ranked(self, queries, targets, qidx, tidx, filter, keep_weak=False):
        """vsx_align_pairs_ranked: the pairs the filter keeps, ranked and compacted on the device (queries in list order, id
        descending, then list order).  -> dict of numpy arrays (pair, score, aligned, matches, mismatches, gaps, verdict, id),
        cigar list, undecided pair indices"""
        lib = _lib.load()
        qidx = np.ascontiguousarray(qidx, np.uint32)
        tidx = np.ascontiguousarray(tidx, np.uint32)
        res = _lib.Ranked()
        f = make_filter(**filter)
        check(lib.vsx_align_pairs_ranked(self.h, queries.h, targets.h, qidx.size, _ptr(qidx), _ptr(tidx), C.byref(f),
                                         1 if keep_weak else 0, C.byref(res)), "vsx_align_pairs_ranked")
        try:
            n = int(res.n_hits)
            cp = lambda p, dt: np.ctypeslib.as_array(p, shape=(n,)).astype(dt, copy=True) if n else np.zeros(0, dt)
            out = {"pair": cp(res.pair, np.uint32), "score": cp(res.score, np.int16), "aligned": cp(res.aligned, np.uint16),
                   "matches": cp(res.matches, np.uint16), "mismatches": cp(res.mismatches, np.uint16), "gaps": cp(res.gaps, np.uint16),
                   "verdict": cp(res.verdict, np.uint8), "id": cp(res.id, np.float64)}
            off = cp(res.cigar_off, np.uint64)
            blob = C.string_at(res.cigar_blob, int(res.cigar_bytes)) if res.cigar_bytes else b""
            out["cigar"] = [blob[int(o):blob.index(b"\0", int(o))].decode() for o in off]
            nu = int(res.n_undecided)
            out["undecided"] = np.ctypeslib.as_array(res.undecided, shape=(nu,)).astype(np.uint32, copy=True) if nu else np.zeros(0, np.uint32)
            return out
        finally:
            lib.vsx_ranked_free(C.byref(res))

    def align(self, q, t):
        """one pair -> (score, aligned, matches, mismatches, gaps, cigar)"""
        qs, ts = SequenceSet(self, seqs=[q]), SequenceSet(self, seqs=[t])
        try:
            return self.align_pairs(qs, ts, [0], [0]).row(0)
        finally:
            qs.close()
            ts.close()
