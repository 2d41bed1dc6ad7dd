"""ctypes binding of librc_hip.so (include/rc_abi.h).

Plumbing for tests and bench.py only: the product is the shared library.  There is no
CPU fallback -- if the library is missing, loading fails loudly.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "librc_hip.so")

ABI_VERSION = 3     # RC_ABI_VERSION of include/rc_abi.h this binding was written against

RC_OK = 0
RC_EINVAL = -100001

VARIANT_AUTO = 0
VARIANT_LITERAL = 1
VARIANT_TILED = 2
VARIANT_TILED2 = 3
VARIANT_TILED8 = 4
VARIANT_TILED4 = 5
VARIANT_CULLED = 6


class RcVec3(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float), ("z", C.c_float)]


class RcCamera(C.Structure):
    _fields_ = [("aspect", C.c_float), ("fov", C.c_float), ("position", RcVec3),
                ("forward", RcVec3), ("up", RcVec3), ("right", RcVec3)]


class RcScene(C.Structure):
    _fields_ = [("sphere_num", C.c_int), ("position", C.c_void_p), ("radius", C.c_void_p),
                ("color", C.c_void_p), ("cam", RcCamera)]


class RcScreen(C.Structure):
    _fields_ = [("surface", C.c_void_p), ("width", C.c_int), ("height", C.c_int),
                ("pitch", C.c_size_t)]


class RcExt(C.Structure):
    """EXT-mode additions (plane, point light, Phong, shadows); no counterpart in the reference."""
    _fields_ = [("plane_normal", RcVec3), ("plane_h", C.c_float), ("plane_color", RcVec3),
                ("light_pos", RcVec3), ("specular", C.c_float), ("shininess_log2", C.c_int),
                ("has_plane", C.c_int), ("shadows", C.c_int)]


class RcOpts(C.Structure):
    _fields_ = [("y_offset", C.c_int), ("full_height", C.c_int), ("variant", C.c_int),
                ("mode", C.c_int), ("stream", C.c_void_p), ("rgba8", C.c_void_p),
                ("rgba8_pitch", C.c_size_t), ("rgba8_flip", C.c_int), ("tuning", C.c_int),
                ("ext", C.POINTER(RcExt)), ("workspace", C.c_void_p), ("workspace_bytes", C.c_size_t)]


class RcInput(C.Structure):
    """One poll of the interactive controls, scripted (main.cpp:53-128)."""
    _fields_ = [("rbutton", C.c_int), ("lshift", C.c_int), ("key_q", C.c_int), ("key_z", C.c_int),
                ("key_w", C.c_int), ("key_a", C.c_int), ("key_s", C.c_int), ("key_d", C.c_int),
                ("cursor_dx", C.c_int), ("cursor_dy", C.c_int)]


TUNE_SHADOW_AXIS_ON = 1
TUNE_SHADOW_AXIS_OFF = 2
TUNE_NO_LDS_EXACT = 4
TUNE_NO_NEAR_SKIP = 8

MODE_REF = 0
MODE_EXT = 1


assert C.sizeof(RcVec3) == 12 and C.sizeof(RcCamera) == 56
assert C.sizeof(RcScene) == 88 and C.sizeof(RcScreen) == 24 and C.sizeof(RcExt) == 56
assert C.sizeof(RcOpts) == 72 and RcOpts.workspace.offset == 56 and C.sizeof(RcInput) == 40

# every symbol include/rc_abi.h declares: name -> (restype, argtypes)
SYMBOLS = {
    "rc_render": (C.c_int, [C.POINTER(RcScreen), C.POINTER(RcScene), C.POINTER(RcOpts)]),
    "rc_render_static": (C.c_int, [C.POINTER(RcScreen), C.POINTER(RcScene), C.POINTER(RcOpts)]),
    "rc_workspace_bytes": (C.c_size_t, [C.c_int]),
    "rc_workspace_bytes_frame": (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    "rc_prepare": (C.c_int, [C.c_int, C.c_int, C.c_float]),
    "rc_to_rgba8": (C.c_int, [C.POINTER(RcScreen), C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]),
    "rc_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "rc_device_name": (C.c_int, [C.c_int, C.c_char_p, C.c_size_t]),
    "rc_set_device": (C.c_int, [C.c_int]),
    "rc_device_info": (C.c_int, [C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_size_t)]),
    "rc_malloc_pitch": (C.c_int, [C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.c_size_t, C.c_size_t]),
    "rc_malloc": (C.c_int, [C.POINTER(C.c_void_p), C.c_size_t]),
    "rc_free": (C.c_int, [C.c_void_p]),
    "rc_memset": (C.c_int, [C.c_void_p, C.c_int, C.c_size_t, C.c_void_p]),
    "rc_upload": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "rc_download_2d": (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_void_p]),
    "rc_sync": (C.c_int, [C.c_void_p]),
    "rc_download_2d_async": (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_void_p]),
    "rc_stream_create": (C.c_int, [C.POINTER(C.c_void_p)]),
    "rc_stream_destroy": (C.c_int, [C.c_void_p]),
    "rc_stream_wait_event": (C.c_int, [C.c_void_p, C.c_void_p]),
    "rc_host_alloc": (C.c_int, [C.POINTER(C.c_void_p), C.c_size_t]),
    "rc_host_free": (C.c_int, [C.c_void_p]),
    "rc_graph_begin": (C.c_int, [C.c_void_p]),
    "rc_graph_end": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p)]),
    "rc_graph_launch": (C.c_int, [C.c_void_p, C.c_void_p]),
    "rc_graph_destroy": (C.c_int, [C.c_void_p]),
    "rc_event_create": (C.c_int, [C.POINTER(C.c_void_p)]),
    "rc_event_record": (C.c_int, [C.c_void_p, C.c_void_p]),
    "rc_event_elapsed_ms": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(C.c_float)]),
    "rc_event_wait": (C.c_int, [C.c_void_p]),
    "rc_event_destroy": (C.c_int, [C.c_void_p]),
    "rc_camera_data": (None, [C.POINTER(RcVec3), C.POINTER(RcVec3), C.c_int, C.c_int, C.POINTER(RcCamera)]),
    "rc_handle_input": (C.c_int, [C.POINTER(RcVec3), C.POINTER(RcVec3), C.c_float, C.c_float, C.c_float, C.POINTER(RcInput)]),
    "rc_ctx_create": (C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int]),
    "rc_ctx_destroy": (C.c_int, [C.c_void_p]),
    "rc_scene_generate": (C.c_int, [C.c_void_p, C.c_int, C.c_uint32]),
    "rc_scene_upload": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "rc_camera_set": (C.c_int, [C.c_void_p, C.POINTER(RcVec3), C.POINTER(RcVec3)]),
    "rc_camera_input": (C.c_int, [C.c_void_p, C.POINTER(RcInput), C.c_float]),
    "rc_surface_alloc": (C.c_int, [C.c_void_p, C.POINTER(C.c_size_t)]),
    "rc_ctx_render": (C.c_int, [C.c_void_p, C.POINTER(RcOpts)]),
    "rc_readback": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "rc_readback_rgba8": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]),
    "rc_ctx_write_ppm": (C.c_int, [C.c_void_p, C.c_char_p]),
    "rc_ctx_screen": (C.POINTER(RcScreen), [C.c_void_p]),
    "rc_ctx_scene": (C.POINTER(RcScene), [C.c_void_p]),
    "rc_generate_scene": (None, [C.c_uint32, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "rc_write_ppm": (C.c_int, [C.c_char_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int]),
    "rc_selftest_math": (C.c_int, [C.c_int, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "rc_pixels_per_lane": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "rc_last_error_string": (C.c_char_p, []),
    "rc_version": (C.c_char_p, []),
    "rc_abi_version": (C.c_int, []),
    # the source-level drop-in (kernel.cu:114-115); C++ by-value convention, so it is only
    # checked for presence here and exercised by the C++ caller in tests/
    "execute_kernel": (None, None),
}

_lib = None


class RcError(RuntimeError):
    pass


def load():
    """Load librc_hip.so (no GPU needed to load; compute calls need one)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RcError(
            f"{LIB_PATH} is missing: build it with `python -m cudaraytracing_amd.build` "
            "(there is no CPU fallback)")
    lib = C.CDLL(LIB_PATH)
    for name, (restype, argtypes) in SYMBOLS.items():
        fn = getattr(lib, name)   # AttributeError if the symbol is not exported
        if argtypes is not None:
            fn.restype = restype
            fn.argtypes = argtypes
    if lib.rc_abi_version() != ABI_VERSION:
        raise RcError(f"{LIB_PATH} was built for ABI version {lib.rc_abi_version()}, this binding needs {ABI_VERSION}: "
                      "rebuild it with `python -m cudaraytracing_amd.build`")
    _lib = lib
    return lib


def check(rc, what="rc call"):
    if rc != RC_OK:
        msg = load().rc_last_error_string().decode(errors="replace")
        raise RcError(f"{what} failed ({rc}): {msg}")
