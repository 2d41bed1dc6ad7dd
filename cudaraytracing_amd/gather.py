"""ctypes binding of librc_gather.so (include/rc_gather.h): row-band assembly over RCCL.

Plumbing for tests and bench.py; the product is the shared library.  Loading needs no GPU (the
partition helpers are pure host arithmetic); creating a communicator does.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "librc_gather.so")
ID_BYTES = 128

SYMBOLS = {
    "rc_band_rows": (None, [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "rc_sub_band": (None, [C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "rc_gather_unique_id": (C.c_int, [C.c_void_p]),
    "rc_gather_create": (C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_void_p]),
    "rc_gather_begin": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]),
    "rc_gather_piece": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]),
    "rc_gather_wait": (C.c_int, [C.c_void_p, C.c_void_p]),
    "rc_gather_destroy": (C.c_int, [C.c_void_p]),
    "rc_gather_last_error": (C.c_char_p, []),
}

_lib = None


class GatherError(RuntimeError):
    pass


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise GatherError(f"{LIB_PATH} is missing: build it with `python -m cudaraytracing_amd.build`")
    lib = C.CDLL(LIB_PATH)
    for name, (restype, argtypes) in SYMBOLS.items():
        fn = getattr(lib, name)
        fn.restype = restype
        fn.argtypes = argtypes
    _lib = lib
    return lib


def check(rc, what="rc_gather call"):
    if rc != 0:
        raise GatherError(f"{what} failed ({rc}): {load().rc_gather_last_error().decode(errors='replace')}")


def band_rows(height, world, rank):
    b, e = C.c_int(), C.c_int()
    load().rc_band_rows(height, world, rank, C.byref(b), C.byref(e))
    return b.value, e.value


def sub_band(begin, end, parts, k):
    b, e = C.c_int(), C.c_int()
    load().rc_sub_band(begin, end, parts, k, C.byref(b), C.byref(e))
    return b.value, e.value


def unique_id():
    buf = C.create_string_buffer(ID_BYTES)
    check(load().rc_gather_unique_id(buf), "rc_gather_unique_id")
    return buf.raw


class Gather:
    """One communicator on the current device; pieces of this rank's band go to rank 0."""

    def __init__(self, world, rank, ident):
        self.lib = load()
        self.handle = C.c_void_p()
        keep = C.create_string_buffer(ident, ID_BYTES)
        check(self.lib.rc_gather_create(C.byref(self.handle), world, rank, keep), "rc_gather_create")

    def begin(self, parts, full_ptr, row_bytes, height, stream):
        check(self.lib.rc_gather_begin(self.handle, parts, full_ptr, row_bytes, height, stream), "rc_gather_begin")

    def piece(self, k, parts, band_ptr, full_ptr, row_bytes, height, stream):
        check(self.lib.rc_gather_piece(self.handle, k, parts, band_ptr, full_ptr, row_bytes, height, stream), "rc_gather_piece")

    def wait(self, stream):
        check(self.lib.rc_gather_wait(self.handle, stream), "rc_gather_wait")

    def destroy(self):
        if self.handle:
            check(self.lib.rc_gather_destroy(self.handle), "rc_gather_destroy")
            self.handle = C.c_void_p()
