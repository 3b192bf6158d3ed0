"""Host-side mirror of the reference's library search API (src/core/search.hpp:88-145:
search_session_* / search_batch) over include/vsx_search.h.

    SearchSession(aligner, db_sequences, id=0.9, ...)   ~ Database + Dbindex + search_session_init
    .search_batch(queries) -> list of per-query hit lists (best first, accepted | weak hits only)
    .userout(queries, fields)                           ~ --userout lines of --usearch_global (for parity tests)
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import SearchOpts, Hits, check


def _blob(seqs):
    bs = [s.encode() if isinstance(s, str) else bytes(s) for s in seqs]
    lens = np.array([len(b) for b in bs], np.uint32)
    off = np.zeros(len(bs), np.uint64)
    if len(bs) > 1:
        off[1:] = np.cumsum(lens[:-1], dtype=np.uint64)
    return b"".join(bs), off, lens


HIT_FIELDS = [n for n, _ in _lib.Hit._fields_ if n not in ("pad", "cigar_off")]


def _meta(sizes, labels, n):
    """vsx_seq_meta + the buffers it points into (keep them alive for the call)"""
    m = _lib.SeqMeta()
    keep = []
    if sizes is not None:
        a = np.ascontiguousarray(sizes, np.uint64)
        if a.size != n:
            raise ValueError("sizes: one abundance per sequence")
        m.abundance = a.ctypes.data_as(C.POINTER(C.c_uint64))
        keep.append(a)
    if labels is not None:
        if len(labels) != n:
            raise ValueError("labels: one header per sequence")
        arr = (C.c_char_p * max(n, 1))(*[l.encode() if isinstance(l, str) else bytes(l) for l in labels])
        m.label = arr
        keep.append(arr)
    return m, keep


def dust_mask(seqs, threads=0):
    """DUST-masked copies of the sequences (bytes): upper case, low-complexity intervals lower case -- dust() of the
    reference (core/mask.cpp:127-199), what --qmask dust / --dbmask dust do before the k-mer stage.  Host threads."""
    lib = _lib.load()
    blob, off, lens = _blob(seqs)
    buf = C.create_string_buffer(blob, len(blob) + 1)
    check(lib.vsx_dust_mask(C.cast(buf, C.c_void_p), len(lens), off.ctypes.data_as(C.c_void_p), lens.ctypes.data_as(C.c_void_p),
                            int(threads)), "vsx_dust_mask")
    raw = buf.raw
    return [raw[int(o):int(o) + int(n)] for o, n in zip(off, lens)]


class SearchSession:
    def __init__(self, aligner, db, sizes=None, labels=None, **opts):
        lib = _lib.load()
        self.aligner = aligner
        aligner._children.add(self)
        o = SearchOpts()
        lib.vsx_search_opts_default(C.byref(o))
        for k, v in opts.items():
            k = "self" if k == "self_" else k             # --self (a Python keyword-ish name: pass self_=1)
            if not hasattr(o, k):
                raise TypeError(f"unknown search option {k}")
            setattr(o, k, v)
        self.opts = o
        self.db = list(db)
        blob, off, lens = _blob(self.db)
        self._keep = (blob, off, lens)
        self.h = C.c_void_p()
        check(lib.vsx_searcher_create(aligner.h, C.byref(self.h), C.byref(o), len(lens),
                                      C.cast(C.c_char_p(blob), C.c_void_p), len(blob),
                                      off.ctypes.data_as(C.c_void_p), lens.ctypes.data_as(C.c_void_p)),
              "vsx_searcher_create")
        if sizes is not None or labels is not None:
            # Database::getabundance / getheader of the targets (--sizein, --self; core/db.hpp)
            m, keep = _meta(sizes, labels, len(lens))
            check(lib.vsx_searcher_set_meta(self.h, C.byref(m)), "vsx_searcher_set_meta")
        self.stats = {}

    def close(self):
        if getattr(self, "h", None) and self.h.value:
            _lib.load().vsx_searcher_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def masked_db_text(self):
        """the database sequences as indexed and aligned (vsx_searcher_db_text): the searcher's masking applied to the text"""
        lib = _lib.load()
        lib.vsx_searcher_db_text.restype = C.c_uint64
        blob, off, lens = self._keep
        buf = C.create_string_buffer(max(1, len(blob)))
        n = lib.vsx_searcher_db_text(self.h, buf, C.c_uint64(len(blob)))
        raw = buf.raw[:int(n)]
        return [raw[int(o):int(o) + int(l)].decode("latin-1") for o, l in zip(off, lens)]

    def candidates(self, query, cap=4096):
        """(target, shared-kmer count) in the order search_topscores + minheap_sort hand them out"""
        q = query.encode() if isinstance(query, str) else bytes(query)
        t = np.zeros(cap, np.uint32)
        c = np.zeros(cap, np.uint32)
        n = _lib.load().vsx_search_candidates(self.h, q, len(q), t.ctypes.data_as(C.c_void_p),
                                              c.ctypes.data_as(C.c_void_p), cap)
        n = min(int(n), cap)
        return list(zip(t[:n].tolist(), c[:n].tolist()))

    def candidates_batch(self, queries, device=True):
        """per-query [(target, count)] lists, best first (search_topscores + heap order) -- counted on the GPU
        (device=True, vsx_kmer.hip) or by the host restatement; self.kmer_stats holds the timings of the call"""
        lib = _lib.load()
        blob, off, lens = _blob(queries)
        res = _lib.Candidates()
        check(lib.vsx_search_candidates_batch(self.h, 1 if device else 0, len(lens), C.cast(C.c_char_p(blob), C.c_void_p),
                                              len(blob), off.ctypes.data_as(C.c_void_p), lens.ctypes.data_as(C.c_void_p),
                                              C.byref(res)), "vsx_search_candidates_batch")
        try:
            n = int(res.n_queries)
            start = np.ctypeslib.as_array(res.start, shape=(n + 1,)).copy()
            tot = int(start[n])
            tg = np.ctypeslib.as_array(res.target, shape=(max(tot, 1),))[:tot].copy()
            ct = np.ctypeslib.as_array(res.count, shape=(max(tot, 1),))[:tot].copy()
            self.kmer_stats = {k: getattr(res, k) for k in ("seconds", "kernel_ms", "index_build_ms", "index_postings",
                                                            "postings_streamed", "bytes_streamed")}
            return [list(zip(tg[start[k]:start[k + 1]].tolist(), ct[start[k]:start[k + 1]].tolist())) for k in range(n)]
        finally:
            lib.vsx_candidates_free(C.byref(res))

    def allpairs(self, first=0, count=None, acceptall=False):
        """allpairs_global over database sequences [first, first+count): per-query hit lists (query = db index)"""
        lib = _lib.load()
        count = len(self.db) - first if count is None else count
        res = Hits()
        check(lib.vsx_allpairs_block(self.h, 1 if acceptall else 0, first, count, C.byref(res)), "vsx_allpairs_block")
        return self._unpack(res)

    def allpairs_rows(self, rows, acceptall=False):
        """allpairs_global for an ascending list of query rows (a rank's share under sharding.shard_allpairs_rows):
        per-row hit lists, entry k = rows[k]"""
        lib = _lib.load()
        r = np.ascontiguousarray(rows, np.uint32)
        res = Hits()
        check(lib.vsx_allpairs_rows(self.h, 1 if acceptall else 0, r.ctypes.data_as(C.c_void_p), r.size, C.byref(res)), "vsx_allpairs_rows")
        return self._unpack(res)

    def allpairs_stream(self, first=0, count=None, block=0, acceptall=False):
        """vsx_allpairs_stream: the rows in blocks, enumeration / alignment / completion of consecutive blocks overlapped inside the
        library; -> per-query hit lists as allpairs() returns them (the sink copies each block's hits)"""
        lib = _lib.load()
        count = len(self.db) - first if count is None else count
        out = []
        stats = []
        SINK = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_uint64, C.c_uint64, C.POINTER(Hits))

        def sink(_user, bfirst, bcount, hp):
            try:
                res = hp.contents
                cig = C.string_at(res.cigar_blob, int(res.cigar_bytes)) if res.cigar_bytes else b""
                assert int(res.n_queries) == int(bcount) and int(bfirst) == first + len(out)
                for q in range(int(res.n_queries)):
                    hs = []
                    for k in range(int(res.first[q]), int(res.first[q + 1])):
                        h = res.hit[k]
                        d = {n: getattr(h, n) for n in HIT_FIELDS}
                        o = int(h.cigar_off)
                        d["cigar"] = cig[o:cig.index(b"\0", o)].decode()
                        hs.append(d)
                    out.append(hs)
                stats.append((int(res.pairs_aligned), int(res.cells_aligned)))
                return 0
            except Exception:                    # (never let an exception cross the C boundary)
                import traceback
                traceback.print_exc()
                return -99

        cb = SINK(sink)
        check(lib.vsx_allpairs_stream(self.h, 1 if acceptall else 0, first, count, block, C.cast(cb, C.c_void_p), None), "vsx_allpairs_stream")
        self.stats = {"pairs_aligned": sum(p for p, _ in stats), "cells_aligned": sum(c for _, c in stats)}
        return out

    def cluster_fast(self, round=0):
        """greedy centroid clustering of the session's sequences in their given order (sort them first).
        -> (clusterno list, per-sequence hit dict or None, number of clusters)"""
        lib = _lib.load()
        res = _lib.ClusterOut()
        check(lib.vsx_cluster_fast(self.h, int(round), C.byref(res)), "vsx_cluster_fast")
        try:
            n = int(res.n)
            cno = [int(res.clusterno[k]) for k in range(n)]
            hits = res.hits
            cig = C.string_at(hits.cigar_blob, int(hits.cigar_bytes)) if hits.cigar_bytes else b""
            per = []
            for q in range(n):
                a, b = int(hits.first[q]), int(hits.first[q + 1])
                if a == b:
                    per.append(None)
                    continue
                h = hits.hit[a]
                d = {nm: getattr(h, nm) for nm in HIT_FIELDS}
                o = int(h.cigar_off)
                d["cigar"] = cig[o:cig.index(b"\0", o)].decode()
                per.append(d)
            self.stats = {nm: getattr(hits, nm) for nm in ("pairs_aligned", "cells_aligned", "stages", "sentinel_pairs",
                                                             "seconds_kmer", "seconds_align", "seconds_total")}
            return cno, per, int(res.n_clusters)
        finally:
            lib.vsx_cluster_out_free(C.byref(res))

    def uc_lines(self, names, round=0, sizes=None, command="cluster_fast"):
        """the --uc file of --cluster_fast: S/H records in processing order, then one C record per cluster
        (core/results.cpp:274-327, core/cluster.cpp:513-547, :1366-1378); with --sizein (`sizes`) a C record carries the
        cluster's total abundance instead of its member count (cluster.cpp:1268-1282)"""
        cno, per, ncl = self.cluster_fast(round)
        lines, size, centroid = [], [0] * ncl, [None] * ncl
        for s, (c, h) in enumerate(zip(cno, per)):
            size[c] += 1 if sizes is None else int(sizes[s])
            if h is None:
                centroid[c] = s
                lines.append(f"S\t{c}\t{len(self.db[s])}\t*\t*\t*\t*\t*\t{names[s]}\t*")
            else:
                # '=': identical ignoring terminal gaps for cluster_fast, strictly identical for the other commands
                # (check_if_perfect_match, core/results.cpp:84-95)
                full = h["internal_alignmentlength"] if command == "cluster_fast" else h["nwalignmentlength"]
                aln = "=" if h["matches"] == full else h["cigar"]
                lines.append(f"H\t{c}\t{len(self.db[s])}\t{h['id']:.1f}\t+\t0\t0\t{aln}\t{names[s]}\t{names[h['target']]}")
        for c in range(ncl):
            lines.append(f"C\t{c}\t{size[c]}\t*\t*\t*\t*\t*\t{names[centroid[c]]}\t*")
        return lines

    def search_batch(self, queries, sizes=None, labels=None):
        """search_batch (core/search.hpp:131-145); sizes / labels = the queries' abundances and headers (--sizein, --self)"""
        lib = _lib.load()
        blob, off, lens = _blob(queries)
        res = Hits()
        m, keep = _meta(sizes, labels, len(lens))
        check(lib.vsx_search_batch_meta(self.h, len(lens), C.cast(C.c_char_p(blob), C.c_void_p), len(blob),
                                        off.ctypes.data_as(C.c_void_p), lens.ctypes.data_as(C.c_void_p),
                                        C.byref(m) if keep else None, C.byref(res)),
              "vsx_search_batch_meta")
        return self._unpack(res)

    def search_batch_raw(self, queries, sizes=None, labels=None):
        """search_batch without the per-hit Python objects: (first[n + 1] uint64, hits = structured array of vsx_hit,
        cigar blob bytes) -- copies, the C result is released.  What sharding.sharded_search gathers."""
        lib = _lib.load()
        blob, off, lens = _blob(queries)
        res = Hits()
        m, keep = _meta(sizes, labels, len(lens))
        check(lib.vsx_search_batch_meta(self.h, len(lens), C.cast(C.c_char_p(blob), C.c_void_p), len(blob),
                                        off.ctypes.data_as(C.c_void_p), lens.ctypes.data_as(C.c_void_p),
                                        C.byref(m) if keep else None, C.byref(res)),
              "vsx_search_batch_meta")
        try:
            n, nh = int(res.n_queries), int(res.n_hits)
            first = np.ctypeslib.as_array(res.first, shape=(n + 1,)).copy()
            hits = np.ctypeslib.as_array(res.hit, shape=(max(nh, 1),))[:nh].copy()
            cig = C.string_at(res.cigar_blob, int(res.cigar_bytes)) if res.cigar_bytes else b""
            self.stats = {nm: getattr(res, nm) for nm in ("pairs_aligned", "cells_aligned", "stages", "sentinel_pairs",
                                                         "seconds_kmer", "seconds_align", "seconds_total")}
            return first, hits, cig
        finally:
            lib.vsx_hits_free(C.byref(res))

    @staticmethod
    def hits_as_lists(first, hits, cig):
        """(first, hits, cigar blob) -> the per-query lists of dicts search_batch returns"""
        out = []
        for q in range(len(first) - 1):
            hs = []
            for k in range(int(first[q]), int(first[q + 1])):
                h = hits[k]
                d = {n: h[n].item() for n in HIT_FIELDS}
                o = int(h["cigar_off"])
                d["cigar"] = cig[o:cig.index(b"\0", o)].decode()
                hs.append(d)
            out.append(hs)
        return out

    def _unpack(self, res):
        lib = _lib.load()
        try:
            cig = C.string_at(res.cigar_blob, int(res.cigar_bytes)) if res.cigar_bytes else b""
            out = []
            for q in range(int(res.n_queries)):
                hs = []
                for k in range(int(res.first[q]), int(res.first[q + 1])):
                    h = res.hit[k]
                    d = {n: getattr(h, n) for n in HIT_FIELDS}
                    o = int(h.cigar_off)
                    d["cigar"] = cig[o:cig.index(b"\0", o)].decode()
                    hs.append(d)
                out.append(hs)
            self.stats = {n: getattr(res, n) for n in ("pairs_aligned", "cells_aligned", "stages", "sentinel_pairs",
                                                        "seconds_kmer", "seconds_align", "seconds_total")}
            return out
        finally:
            lib.vsx_hits_free(C.byref(res))

    def userout(self, queries, qnames=None, tnames=None,
                fields=("query", "target", "id", "alnlen", "mism", "opens", "raw", "caln"), hits=None):
        """--userout lines exactly as results_show_userout_one prints these fields (core/results.cpp:330-470)"""
        if hits is None:
            hits = self.search_batch(queries)
        qnames = qnames or [f"q{i}" for i in range(len(queries))]
        tnames = tnames or [f"t{i}" for i in range(len(self.db))]
        fmt = {
            "query": lambda q, h: qnames[q], "target": lambda q, h: tnames[h["target"]],
            "id": lambda q, h: "%.1f" % h["id"], "alnlen": lambda q, h: "%d" % h["internal_alignmentlength"],
            "mism": lambda q, h: "%d" % h["mismatches"], "opens": lambda q, h: "%d" % h["internal_gaps"],
            "exts": lambda q, h: "%d" % (h["internal_indels"] - h["internal_gaps"]),
            "gaps": lambda q, h: "%d" % h["internal_indels"], "pairs": lambda q, h: "%d" % (h["matches"] + h["mismatches"]),
            "pv": lambda q, h: "%d" % h["matches"], "raw": lambda q, h: "%d" % h["nwscore"], "caln": lambda q, h: h["cigar"],
            "id0": lambda q, h: "%.1f" % h["id0"], "id1": lambda q, h: "%.1f" % h["id1"], "id2": lambda q, h: "%.1f" % h["id2"],
            "id3": lambda q, h: "%.1f" % h["id3"], "id4": lambda q, h: "%.1f" % h["id4"],
            "qstrand": lambda q, h: "-" if h.get("strand") else "+",
            "ql": lambda q, h: "%d" % len(queries[q]), "tl": lambda q, h: "%d" % len(self.db[h["target"]]),
        }
        lines = []
        for q, hs in enumerate(hits):
            for h in hs:
                lines.append("\t".join(fmt[f](q, h) for f in fields))
        return lines


def _msa_unpack(lib, out):
    L = int(out.alnlen)
    raw = C.string_at(out.rows, int(out.n_rows) * (L + 1))
    rows = [raw[k * (L + 1):k * (L + 1) + L].decode("latin-1") for k in range(int(out.n_rows))]
    prof = np.ctypeslib.as_array(out.profile, shape=(max(L, 1), 6))[:L].tolist()
    return dict(rows=rows, consensus=C.string_at(out.consensus, int(out.conslen)).decode("latin-1"), profile=prof)


def msa(seqs, cigars, abundances=None, aligner=None):
    """star MSA / profile / consensus of one cluster (core/msa.cpp): seqs[0] = centroid, cigars[i] = member i vs centroid.
    -> dict(rows=[centroid, members..., consensus row], consensus=str, profile=[[A,C,G,T,N,gap] per column]).
    With `aligner` (an Aligner = a device context) the rows, profile and consensus are computed on the GPU (vsx_msa_device);
    without, by the host form (vsx_msa).  Both return the same bytes."""
    return msa_batch([seqs], [cigars], [abundances] if abundances is not None else None, aligner)[0]


def msa_batch(clusters_seqs, clusters_cigars, clusters_abundances=None, aligner=None):
    """many clusters; with `aligner` they go to the device in ONE pass (vsx_msa_device_batch)."""
    lib = _lib.load()
    bs, cs, ab, start = [], [], [], [0]
    for k, (seqs, cigars) in enumerate(zip(clusters_seqs, clusters_cigars)):
        bs += [s.encode() if isinstance(s, str) else bytes(s) for s in seqs]
        cs += [(c.encode() if isinstance(c, str) else c) if c is not None else b"" for c in cigars]
        if clusters_abundances is not None:
            ab += [int(x) for x in clusters_abundances[k]]
        start.append(len(bs))
    n = len(bs)
    nc = len(start) - 1
    sp = (C.c_char_p * max(n, 1))(*bs)
    cp = (C.c_char_p * max(n, 1))(*cs)
    lens = (C.c_uint32 * max(n, 1))(*[len(b) for b in bs])
    abp = (C.c_uint64 * max(n, 1))(*ab) if clusters_abundances is not None else None
    outs = (_lib.MsaOut * max(nc, 1))()
    res = []
    if aligner is not None:
        st = (C.c_uint64 * (nc + 1))(*start)
        check(lib.vsx_msa_device_batch(aligner.h, nc, st, sp, lens, cp, abp, outs), "vsx_msa_device_batch")
        try:
            res = [_msa_unpack(lib, outs[k]) for k in range(nc)]
        finally:
            for k in range(nc):
                lib.vsx_msa_out_free(C.byref(outs[k]))
        return res
    for k in range(nc):
        a, b = start[k], start[k + 1]
        off = lambda arr, ty: C.cast(C.byref(arr, a * C.sizeof(ty)), C.c_void_p)
        check(lib.vsx_msa(b - a, off(sp, C.c_char_p), off(lens, C.c_uint32), off(cp, C.c_char_p),
                          off(abp, C.c_uint64) if abp is not None else None, C.byref(outs[k])), "vsx_msa")
        try:
            res.append(_msa_unpack(lib, outs[k]))
        finally:
            lib.vsx_msa_out_free(C.byref(outs[k]))
    return res


class MultiSearchSession:
    """Several GPUs of one node behind one handle (include/vsx_search.h, multi-device form; vsx_multi.cpp): the reference's
    worker pool over a shared database (commands/usearch_global.cpp:500-535) as one database replica + k-mer index + aligner
    context per listed device.  search_batch_raw / allpairs_raw return what SearchSession.search_batch_raw returns for the same
    call on ONE device -- hit for hit.  A device may be listed twice (two replicas on one GPU)."""

    def __init__(self, db, devices=(0,), scoring=None, sizes=None, labels=None, **opts):
        from .aligner import DEFAULT_SCORING
        lib = _lib.load()
        o = SearchOpts()
        lib.vsx_search_opts_default(C.byref(o))
        for k, v in opts.items():
            k = "self" if k == "self_" else k
            if not hasattr(o, k):
                raise TypeError(f"unknown search option {k}")
            setattr(o, k, v)
        sc = scoring
        if sc is None:
            sc = _lib.Scoring()
            for k, v in DEFAULT_SCORING.items():
                setattr(sc, k, int(v))
        self.db = list(db)
        blob, off, lens = _blob(self.db)
        dev = np.ascontiguousarray(devices, np.int32)
        m, keep = _meta(sizes, labels, len(lens))
        self.h = C.c_void_p()
        check(lib.vsx_multi_searcher_create(C.byref(self.h), C.byref(sc), dev.ctypes.data_as(C.c_void_p), dev.size, C.byref(o), len(lens),
                                            C.cast(C.c_char_p(blob), C.c_void_p), len(blob), off.ctypes.data_as(C.c_void_p),
                                            lens.ctypes.data_as(C.c_void_p), C.byref(m) if keep else None),
              "vsx_multi_searcher_create")
        self.n_devices = int(lib.vsx_multi_searcher_devices(self.h))

    def close(self):
        if getattr(self, "h", None) and self.h.value:
            _lib.load().vsx_multi_searcher_destroy(self.h)
            self.h = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @staticmethod
    def _raw(res):
        lib = _lib.load()
        try:
            n, nh = int(res.n_queries), int(res.n_hits)
            first = np.ctypeslib.as_array(res.first, shape=(n + 1,)).copy()
            hits = np.ctypeslib.as_array(res.hit, shape=(max(nh, 1),))[:nh].copy()
            cig = C.string_at(res.cigar_blob, int(res.cigar_bytes)) if res.cigar_bytes else b""
            stats = {nm: getattr(res, nm) for nm in ("pairs_aligned", "cells_aligned", "stages", "sentinel_pairs")}
            return first, hits, cig, stats
        finally:
            lib.vsx_hits_free(C.byref(res))

    def search_batch_raw(self, queries, sizes=None, labels=None):
        lib = _lib.load()
        blob, off, lens = _blob(queries)
        res = Hits()
        m, keep = _meta(sizes, labels, len(lens))
        check(lib.vsx_multi_search_batch(self.h, len(lens), C.cast(C.c_char_p(blob), C.c_void_p), len(blob), off.ctypes.data_as(C.c_void_p),
                                         lens.ctypes.data_as(C.c_void_p), C.byref(m) if keep else None, C.byref(res)),
              "vsx_multi_search_batch")
        first, hits, cig, self.stats = self._raw(res)
        return first, hits, cig

    def allpairs_raw(self, first=0, count=None, acceptall=False):
        lib = _lib.load()
        count = len(self.db) - first if count is None else count
        res = Hits()
        check(lib.vsx_multi_allpairs(self.h, 1 if acceptall else 0, first, count, C.byref(res)), "vsx_multi_allpairs")
        f, hits, cig, self.stats = self._raw(res)
        return f, hits, cig
