"""ctypes binding of libvsx.so (the C-ABI declared in include/vsx.h).

The shared library is built IN-TREE (vsearch_amd/libvsx.so, `make -C vsearch_amd/csrc` or
__graft_entry__.build()).  There is no CPU fallback: if the library is missing, loading fails
loudly; if no gfx950 device is visible every compute entry point returns VSX_ENODEVICE.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libvsx.so")

VSX_OK, VSX_EINVAL, VSX_ENODEVICE, VSX_ENOMEM, VSX_EHIP = 0, -1, -2, -3, -4
SENTINEL = 32767

# every symbol include/vsx.h declares
SYMBOLS = [
    "vsx_version_string", "vsx_device_count", "vsx_last_error", "vsx_create", "vsx_destroy",
    "vsx_seqset_create", "vsx_seqset_create_from_device", "vsx_seqset_destroy", "vsx_seqset_count",
    "vsx_plan_create", "vsx_plan_run", "vsx_plan_sync", "vsx_plan_fetch", "vsx_plan_export_hits", "vsx_plan_destroy",
    "vsx_align_pairs", "vsx_results_free",
]


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
                ("cigar_blob", C.POINTER(C.c_char)), ("cigar_bytes", C.c_uint64)]


class Timing(C.Structure):
    _fields_ = [("forward_ms", C.c_float), ("traceback_ms", C.c_float), ("total_ms", C.c_float),
                ("forward_launches", C.c_uint32), ("traceback_launches", C.c_uint32),
                ("cells", C.c_uint64), ("dir_bytes", C.c_uint64)]


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
    lib.vsx_plan_destroy.argtypes = [vp]
    lib.vsx_plan_destroy.restype = None
    lib.vsx_align_pairs.argtypes = [vp, vp, vp, C.c_uint64, vp, vp, C.POINTER(Results)]
    lib.vsx_results_free.argtypes = [C.POINTER(Results)]
    lib.vsx_results_free.restype = None
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
