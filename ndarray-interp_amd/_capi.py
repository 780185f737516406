"""ctypes binding of include/ndinterp.h (libndinterp_hip.so).

The library is the product: there is no Python or CPU re-implementation behind it.  If the
shared object is missing, or no HIP device is usable, the calls fail loudly.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# NDI_LIB names another build of the same library (the bounds-checked one, `make debug`: libndinterp_hip_dbg.so;
# the tuning one, `make tune`); a bare file name is looked up next to this file.
LIB_PATH = os.environ.get("NDI_LIB") or os.path.join(_HERE, "libndinterp_hip.so")
if not os.path.isabs(LIB_PATH):
    LIB_PATH = os.path.join(_HERE, LIB_PATH)

# ndi_status
OK, NOT_ENOUGH_DATA, MONOTONIC, SHAPE, VALUE, OUT_OF_BOUNDS, NAN_QUERY, HIP_ERROR, BAD_ARG, UNSUPPORTED = range(10)
STATUS_NAMES = ["OK", "NOT_ENOUGH_DATA", "MONOTONIC", "SHAPE", "VALUE", "OUT_OF_BOUNDS", "NAN_QUERY",
                "HIP_ERROR", "BAD_ARG", "UNSUPPORTED"]
F32, F64 = 0, 1
MEM_HOST, MEM_DEVICE = 0, 1
LINEAR, CUBIC_SPLINE = 0, 1
BC_NOT_A_KNOT, BC_NATURAL, BC_CLAMPED, BC_FIRST_DERIV, BC_SECOND_DERIV = range(5)
BUILD_DEFAULT, BUILD_REFERENCE_ORDER = 0, 1
EVAL_DEFAULT, EVAL_FRESH_OUTPUT, EVAL_ROWS_AFTER_ERROR_UNSPECIFIED = 0, 1, 2
OUTPUT_ZEROED, OUTPUT_UNINITIALIZED = 0, 1   # ndi_output_flags
PATH_AUTO, PATH_GATHER, PATH_BUCKETED = 0, 1, 2
PATH_NAMES = {0: "auto", 1: "gather", 2: "bucketed"}


class Boundary(C.Structure):
    _fields_ = [("kind", C.c_int32), ("value", C.c_double)]


class Interp1DDesc(C.Structure):
    _fields_ = [
        ("dtype", C.c_int32), ("strategy", C.c_int32), ("extrapolate", C.c_int32), ("device", C.c_int32),
        ("n", C.c_uint64), ("lanes", C.c_uint64), ("x_len", C.c_uint64),
        ("x", C.c_void_p), ("data", C.c_void_p), ("memspace", C.c_int32), ("validate", C.c_int32),
        ("periodic", C.c_int32), ("build_flags", C.c_int32), ("left", Boundary), ("right", Boundary),
        ("lane_left_kind", C.c_void_p), ("lane_left_value", C.c_void_p),
        ("lane_right_kind", C.c_void_p), ("lane_right_value", C.c_void_p),
    ]


class Interp2DDesc(C.Structure):
    _fields_ = [
        ("dtype", C.c_int32), ("extrapolate", C.c_int32), ("device", C.c_int32), ("memspace", C.c_int32),
        ("nx", C.c_uint64), ("ny", C.c_uint64), ("lanes", C.c_uint64),
        ("x_len", C.c_uint64), ("y_len", C.c_uint64),
        ("x", C.c_void_p), ("y", C.c_void_p), ("data", C.c_void_p),
        ("validate", C.c_int32), ("reserved", C.c_int32),
    ]


class OobInfo(C.Structure):
    _fields_ = [("index", C.c_uint64), ("value", C.c_double), ("axis", C.c_int32), ("status", C.c_int32)]


class EvalOpts(C.Structure):
    _fields_ = [("q_memspace", C.c_int32), ("out_memspace", C.c_int32), ("stream", C.c_void_p),
                ("path", C.c_int32), ("async_launch", C.c_int32), ("flags", C.c_int32), ("reserved", C.c_int32)]


class Profile(C.Structure):
    _fields_ = [("eval_launches", C.c_uint64), ("eval_ms", C.c_double),
                ("locate_launches", C.c_uint64), ("locate_ms", C.c_double),
                ("group_launches", C.c_uint64), ("group_ms", C.c_double),
                ("last_path", C.c_int32), ("reserved", C.c_int32)]


class OutputInfo(C.Structure):
    _fields_ = [("tries", C.c_uint32), ("reserved", C.c_uint32), ("fill_tbps", C.c_double),
                ("worst_fill_tbps", C.c_double), ("alloc_ms", C.c_double)]


class RingChunk(C.Structure):
    _fields_ = [("index", C.c_uint64), ("q_begin", C.c_uint64), ("q_count", C.c_uint64), ("out", C.c_void_p),
                ("row_stride", C.c_uint64), ("slot", C.c_uint32), ("shard", C.c_uint32), ("stream", C.c_void_p)]


class RingDesc(C.Structure):
    _fields_ = [("slots", C.POINTER(C.c_void_p)), ("n_slots", C.c_uint32), ("reserved", C.c_uint32),
                ("chunk_queries", C.c_uint64), ("row_stride", C.c_uint64)]


class ShardIO(C.Structure):
    _fields_ = [("q", C.c_void_p), ("qy", C.c_void_p), ("out", C.c_void_p), ("stream", C.c_void_p)]


RING_CONSUMER = C.CFUNCTYPE(C.c_void_p, C.c_void_p, C.POINTER(RingChunk))

# every symbol include/ndinterp.h declares: name -> (restype, argtypes)
_P = C.c_void_p
SYMBOLS = {
    "ndi_interp1d_create": (C.c_int, [C.POINTER(Interp1DDesc), C.POINTER(_P)]),
    "ndi_interp1d_destroy": (None, [_P]),
    "ndi_interp2d_create": (C.c_int, [C.POINTER(Interp2DDesc), C.POINTER(_P)]),
    "ndi_interp2d_destroy": (None, [_P]),
    "ndi_interp1d_clone": (C.c_int, [_P, C.c_int32, C.POINTER(_P)]),
    "ndi_interp2d_clone": (C.c_int, [_P, C.c_int32, C.POINTER(_P)]),
    "ndi_interp1d_coefficients": (C.c_int, [_P, _P, _P, C.c_int32]),
    "ndi_interp1d_eval": (C.c_int, [_P, _P, C.c_uint64, _P, C.c_uint64, C.POINTER(EvalOpts), C.POINTER(OobInfo)]),
    "ndi_interp2d_eval": (C.c_int, [_P, _P, _P, C.c_uint64, _P, C.c_uint64, C.POINTER(EvalOpts), C.POINTER(OobInfo)]),
    "ndi_interp1d_finish": (C.c_int, [_P, _P, C.POINTER(OobInfo)]),
    "ndi_interp2d_finish": (C.c_int, [_P, _P, C.POINTER(OobInfo)]),
    "ndi_interp1d_eval_ring": (C.c_int, [_P, _P, C.c_uint64, C.POINTER(RingDesc), RING_CONSUMER, _P,
                                         C.POINTER(EvalOpts), C.POINTER(OobInfo)]),
    "ndi_interp2d_eval_ring": (C.c_int, [_P, _P, _P, C.c_uint64, C.POINTER(RingDesc), RING_CONSUMER, _P,
                                         C.POINTER(EvalOpts), C.POINTER(OobInfo)]),
    "ndi_shard_bounds": (None, [C.c_uint64, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "ndi_interp1d_eval_sharded": (C.c_int, [C.POINTER(_P), C.c_uint32, _P, C.c_uint64, C.POINTER(ShardIO), C.c_uint64,
                                            C.POINTER(EvalOpts), C.POINTER(OobInfo)]),
    "ndi_interp2d_eval_sharded": (C.c_int, [C.POINTER(_P), C.c_uint32, _P, _P, C.c_uint64, C.POINTER(ShardIO),
                                            C.c_uint64, C.POINTER(EvalOpts), C.POINTER(OobInfo)]),
    "ndi_interp1d_eval_ring_sharded": (C.c_int, [C.POINTER(_P), C.c_uint32, _P, C.c_uint64, C.POINTER(ShardIO),
                                                 C.POINTER(RingDesc), RING_CONSUMER, _P, C.POINTER(EvalOpts),
                                                 C.POINTER(OobInfo)]),
    "ndi_interp2d_eval_ring_sharded": (C.c_int, [C.POINTER(_P), C.c_uint32, _P, _P, C.c_uint64, C.POINTER(ShardIO),
                                                 C.POINTER(RingDesc), RING_CONSUMER, _P, C.POINTER(EvalOpts),
                                                 C.POINTER(OobInfo)]),
    "ndi_interp1d_trim": (C.c_int, [_P]),
    "ndi_interp2d_trim": (C.c_int, [_P]),
    "ndi_interp1d_scratch_sets": (C.c_uint64, [_P]),
    "ndi_locator_create": (C.c_int, [C.c_int32, C.c_int32, _P, C.c_uint64, C.c_int32, C.POINTER(_P)]),
    "ndi_locator_eval": (C.c_int, [_P, _P, C.c_uint64, _P, C.c_int32, _P]),
    "ndi_locator_destroy": (None, [_P]),
    "ndi_get_lower_index_batch": (C.c_int, [C.c_int32, C.c_int32, _P, C.c_uint64, _P, C.c_uint64, _P, C.c_int32]),
    "ndi_monotonic_prop": (C.c_int32, [C.c_int32, _P, C.c_uint64]),
    "ndi_validate1d": (C.c_int, [C.c_int32, _P, C.c_uint64, C.c_uint64, C.c_int32]),
    "ndi_validate2d": (C.c_int, [C.c_int32, _P, C.c_uint64, _P, C.c_uint64, C.c_uint64, C.c_uint64]),
    "ndi_output_alloc": (C.c_int, [C.c_int32, C.c_uint64, C.c_uint32, C.c_uint32, C.POINTER(_P), C.POINTER(OutputInfo)]),
    "ndi_output_free": (C.c_int, [_P]),
    "ndi_output_trim": (C.c_int, []),
    "ndi_device_count": (C.c_int32, []),
    "ndi_last_error_string": (C.c_char_p, []),
    "ndi_version": (C.c_uint32, []),
    "ndi_profile_enable": (None, [C.c_int32]),
    "ndi_profile_read": (C.c_int, [C.POINTER(Profile), C.c_int32]),
    "ndi_interp2d_probe_ceiling": (C.c_int, [_P, C.c_uint64, _P, C.c_uint64, _P, C.c_int32, C.POINTER(C.c_double)]),
}

_lib = None


class NativeLibraryMissing(RuntimeError):
    pass


def lib() -> C.CDLL:
    """Load libndinterp_hip.so (built by __graft_entry__.build() / csrc/Makefile)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise NativeLibraryMissing(
                f"{LIB_PATH} not found: build it with `make -C {os.path.join(_HERE, 'csrc')}` "
                "(hipcc, gfx950).  There is no CPU fallback.")
        handle = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(handle, name)  # AttributeError if the .so does not export it
            fn.restype = res
            fn.argtypes = args
        _lib = handle
    return _lib


def last_error() -> str:
    s = lib().ndi_last_error_string()
    return s.decode("utf-8", "replace") if s else ""
