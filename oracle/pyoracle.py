"""ctypes bindings for the CHECKERS: oracle/liboracle.so (our scalar C restatement) and,
when built, oracle/_ref/libvsref.so (the reference's own aligner compiled from its sources).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py -- never by the product package vsearch_amd/.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_SO = os.path.join(HERE, "liboracle.so")
REF_SO = os.path.join(HERE, "_ref", "libvsref.so")
REFERENCE_ROOT = "/root/reference"

# post-fixup defaults (reference src/vsearch.h:450-461, fix-up src/vsearch.cc:250-259):
# match 2, mismatch -4, open (Q/T) L=1 I=18 R=1, ext L=1 I=2 R=1 -- search16_init argument order
DEFAULT_P = (2, -4, 1, 1, 18, 18, 1, 1, 1, 1, 2, 2, 1, 1)


def build(ref=True):
    """make liboracle.so; when the reference tree is present also _ref/libvsref.so (the reference's aligner), _ref/vsearch_ref
    (its whole CLI, the command-level oracle) and -- once ../vsearch_amd/libvsx.so exists -- _ref/vsearch_vsx (the same CLI
    with core/align_simd.cpp swapped for ../shim/vsx_search16_shim.cpp + libvsx: the drop-in proof, tests/test_gpu_shim.py) and
    the reference's api_examples + oracle/api_driver.cc with search_batch / cluster_assign_batch bound to
    ../shim/vsx_api_adapter.cpp (ref_api)."""
    subprocess.check_call(["make", "-s", "-C", HERE, "liboracle.so"])
    if ref and os.path.isdir(os.path.join(REFERENCE_ROOT, "src")):
        subprocess.check_call(["make", "-s", "-C", HERE, "ref", "ref_full"])
        if os.path.exists(os.path.join(os.path.dirname(HERE), "vsearch_amd", "libvsx.so")):
            subprocess.check_call(["make", "-s", "-C", HERE, "ref_shim", "ref_api"])


def _P(P):
    return (C.c_int64 * 14)(*[int(x) for x in P])


class Oracle:
    """Scalar restatement (oracle/nw_oracle.c)."""

    def __init__(self):
        if not os.path.exists(ORACLE_SO):
            build(ref=False)
        self.lib = C.CDLL(ORACLE_SO)
        self.lib.vsxo_search16_pair.restype = C.c_int
        self.lib.vsxo_search16_batch.restype = C.c_int64
        self.lib.vsxo_map4.restype = C.c_ubyte

    def map4(self, ch):
        return int(self.lib.vsxo_map4(C.c_ubyte(ch)))

    def align(self, q, t, P=DEFAULT_P, nmm=False):
        """-> (score, aligned, matches, mismatches, gaps, cigar)"""
        q = q.encode() if isinstance(q, str) else bytes(q)
        t = t.encode() if isinstance(t, str) else bytes(t)
        sc = C.c_int16()
        a, m, mm, g = C.c_uint16(), C.c_uint16(), C.c_uint16(), C.c_uint16()
        buf = C.create_string_buffer(len(q) + len(t) + 24)
        rc = self.lib.vsxo_search16_pair(q, C.c_int64(len(q)), t, C.c_int64(len(t)), _P(P), int(nmm),
                                         C.byref(sc), C.byref(a), C.byref(m), C.byref(mm), C.byref(g), buf)
        if rc != 0:
            raise MemoryError("oracle allocation failed")
        return (sc.value, a.value, m.value, mm.value, g.value, buf.value.decode())

    def align_batch(self, qblob, qoff, qlen, tblob, toff, tlen, qi, ti, P=DEFAULT_P, nmm=False):
        """numpy in / numpy out: (score i16, aligned, matches, mismatches, gaps u16, [cigar str])"""
        n = len(qi)
        qoff = np.ascontiguousarray(qoff, np.uint64); qlen = np.ascontiguousarray(qlen, np.uint32)
        toff = np.ascontiguousarray(toff, np.uint64); tlen = np.ascontiguousarray(tlen, np.uint32)
        qi = np.ascontiguousarray(qi, np.uint32); ti = np.ascontiguousarray(ti, np.uint32)
        sc = np.zeros(n, np.int16)
        a = np.zeros(n, np.uint16); m = np.zeros(n, np.uint16)
        mm = np.zeros(n, np.uint16); g = np.zeros(n, np.uint16)
        cap = int((qlen[qi].astype(np.int64) + tlen[ti].astype(np.int64) + 24).sum()) + 64
        blob = C.create_string_buffer(cap)
        off = np.zeros(n, np.uint64)
        p = lambda x: x.ctypes.data_as(C.c_void_p)
        used = self.lib.vsxo_search16_batch(
            C.c_char_p(bytes(qblob)), p(qoff), p(qlen), C.c_char_p(bytes(tblob)), p(toff), p(tlen),
            C.c_uint64(n), p(qi), p(ti), _P(P), int(nmm), p(sc), p(a), p(m), p(mm), p(g),
            blob, p(off), C.c_uint64(cap))
        if used < 0:
            raise MemoryError("oracle batch failed")
        raw = blob.raw
        cig = []
        for k in range(n):
            s = int(off[k]); e = raw.index(b"\0", s)
            cig.append(raw[s:e].decode())
        return sc, a, m, mm, g, cig


def have_ref():
    return os.path.exists(REF_SO)


class Reference:
    """The reference's own search16 / LinearMemoryAligner (oracle/ref_driver.cc)."""

    def __init__(self, P=DEFAULT_P, nmm=False):
        if not have_ref():
            raise FileNotFoundError(REF_SO)
        self.lib = C.CDLL(REF_SO)
        self.lib.vsref_create.restype = C.c_void_p
        self.lib.vsref_time_groups.restype = C.c_double
        self.P = tuple(int(x) for x in P)
        self.nmm = bool(nmm)
        self.ctx = C.c_void_p(self.lib.vsref_create(_P(P), int(nmm)))

    def close(self):
        if self.ctx:
            self.lib.vsref_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def search16(self, q, targets):
        """one query vs a list of targets in ONE reference search16 call"""
        q = q.encode() if isinstance(q, str) else bytes(q)
        ts = [t.encode() if isinstance(t, str) else bytes(t) for t in targets]
        n = len(ts)
        bufs = [C.create_string_buffer(t, len(t) + 1) for t in ts]
        arr = (C.c_void_p * n)(*[C.cast(b, C.c_void_p) for b in bufs])
        lens = (C.c_int * n)(*[len(t) for t in ts])
        sc = (C.c_int16 * n)()
        a = (C.c_uint16 * n)(); m = (C.c_uint16 * n)(); mm = (C.c_uint16 * n)(); g = (C.c_uint16 * n)()
        cg = (C.c_void_p * n)()
        self.lib.vsref_search16(self.ctx, q, len(q), n, arr, lens, sc, a, m, mm, g, cg)
        out = []
        for i in range(n):
            s = C.string_at(cg[i]).decode()
            self.lib.vsref_free(C.c_void_p(cg[i]))
            out.append((sc[i], a[i], m[i], mm[i], g[i], s))
        return out

    def align(self, q, t):
        return self.search16(q, [t])[0]

    def lma(self, q, t):
        """LinearMemoryAligner::align + alignstats -> (score, alnlen, matches, mismatches, gaps, cigar)"""
        q = q.encode() if isinstance(q, str) else bytes(q)
        t = t.encode() if isinstance(t, str) else bytes(t)
        v = [C.c_int64() for _ in range(5)]
        cg = C.c_void_p()
        self.lib.vsref_lma(self.ctx, q, len(q), t, len(t), *[C.byref(x) for x in v], C.byref(cg))
        s = C.string_at(cg).decode()
        self.lib.vsref_free(cg)
        return tuple(x.value for x in v) + (s,)

    def time_groups(self, qblob, qoff, qlen, tblob, toff, tlen, gq, goff, tidx, threads=1):
        """Time the reference SSE2 search16 over query groups -> (seconds, cells, checksum); self.last_digest = the
        order-independent hash over every field of every pair incl. the CIGAR text (ref_driver.cc pair_digest)."""
        qoff = np.ascontiguousarray(qoff, np.uint64); qlen = np.ascontiguousarray(qlen, np.uint32)
        toff = np.ascontiguousarray(toff, np.uint64); tlen = np.ascontiguousarray(tlen, np.uint32)
        gq = np.ascontiguousarray(gq, np.uint32); goff = np.ascontiguousarray(goff, np.uint64)
        tidx = np.ascontiguousarray(tidx, np.uint32)
        p = lambda x: x.ctypes.data_as(C.c_void_p)
        cells = C.c_uint64(); chk = C.c_int64(); dig = C.c_uint64()
        secs = self.lib.vsref_time_groups(
            _P(self.P), int(self.nmm), C.c_char_p(bytes(qblob)), p(qoff), p(qlen),
            C.c_char_p(bytes(tblob)), p(toff), p(tlen), C.c_uint32(len(tlen)),
            C.c_uint32(len(gq)), p(gq), p(goff), p(tidx), C.c_int(threads),
            C.byref(cells), C.byref(chk), C.byref(dig))
        self.last_digest = int(dig.value)
        return float(secs), int(cells.value), int(chk.value)

    def digest_results(self, first_pair, score, aligned, matches, mismatches, gaps, cigar_blob_ptr, cigar_off):
        """the same hash over a result block as libvsx hands it out (raw arrays + CIGAR blob pointer + offsets)"""
        n = len(score)
        a = lambda x, t: np.ascontiguousarray(x, t)
        sc, al, ma, mi, ga, of = a(score, np.int16), a(aligned, np.uint16), a(matches, np.uint16), a(mismatches, np.uint16), a(gaps, np.uint16), a(cigar_off, np.uint64)
        p = lambda x: x.ctypes.data_as(C.c_void_p)
        self.lib.vsref_digest_results.restype = C.c_uint64
        self.lib.vsref_digest_results.argtypes = [C.c_uint64, C.c_uint64] + [C.c_void_p] * 7
        return int(self.lib.vsref_digest_results(n, first_pair, p(sc), p(al), p(ma), p(mi), p(ga), cigar_blob_ptr, p(of)))
