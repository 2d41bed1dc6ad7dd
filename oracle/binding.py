"""ctypes binding of oracle/liboracle.so -- the CHECKER, never the product.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "liboracle.so")

# offset basis used by the recorded reference known-answer hashes (tests/golden/ref_kat.json)
KAT_FNV_BASIS = 1469598103934665603
FNV_BASIS = 0xCBF29CE484222325


class Vec3(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float), ("z", C.c_float)]


class Camera(C.Structure):
    _fields_ = [("aspect", C.c_float), ("fov", C.c_float), ("position", Vec3),
                ("forward", Vec3), ("up", Vec3), ("right", Vec3)]


class Scene(C.Structure):
    _fields_ = [("sphere_num", C.c_int), ("position", C.c_void_p), ("radius", C.c_void_p),
                ("color", C.c_void_p), ("cam", Camera)]


class Screen(C.Structure):
    _fields_ = [("surface", C.c_void_p), ("width", C.c_int), ("height", C.c_int), ("pitch", C.c_size_t)]


class Ext(C.Structure):
    """EXT-mode parameters (plane, point light, Phong, shadows); layout of rco_ext / rc_ext."""
    _fields_ = [("plane_normal", Vec3), ("plane_h", C.c_float), ("plane_color", Vec3),
                ("light_pos", Vec3), ("specular", C.c_float), ("shininess_log2", C.c_int),
                ("has_plane", C.c_int), ("shadows", C.c_int)]


_lib = None


def build():
    subprocess.run(["make", "-C", _HERE, "liboracle.so"], check=True, stdout=subprocess.DEVNULL)


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        build()
    lib = C.CDLL(LIB_PATH)
    lib.rco_render_rows.argtypes = [C.POINTER(Screen), C.POINTER(Scene), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
    lib.rco_render_rows.restype = None
    lib.rco_render_rows_ext.argtypes = [C.POINTER(Screen), C.POINTER(Scene), C.POINTER(Ext), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
    lib.rco_render_rows_ext.restype = None
    lib.rco_camera_data.argtypes = [Vec3, Vec3, C.c_int, C.c_int]
    lib.rco_camera_data.restype = Camera
    lib.rco_generate_scene.argtypes = [C.c_uint32, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.rco_generate_scene.restype = None
    lib.rco_to_rgba8.argtypes = [C.POINTER(Screen), C.c_void_p, C.c_size_t, C.c_int]
    lib.rco_to_rgba8.restype = None
    lib.rco_fnv1a_surface.argtypes = [C.POINTER(Screen)]
    lib.rco_fnv1a_surface.restype = C.c_uint64
    lib.rco_fnv1a64_words.argtypes = [C.POINTER(Screen), C.c_uint64]
    lib.rco_fnv1a64_words.restype = C.c_uint64
    lib.rco_hit_mask.argtypes = [C.POINTER(Screen), C.POINTER(Scene), C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    lib.rco_hit_mask.restype = None
    lib.rco_find_intersection.argtypes = [Vec3, C.c_float, Vec3, Vec3]
    lib.rco_find_intersection.restype = C.c_float
    lib.rco_cast_screen_ray.argtypes = [C.POINTER(Camera), C.c_int, C.c_int, C.c_int, C.c_int]
    lib.rco_cast_screen_ray.restype = Vec3
    _lib = lib
    return lib


def camera_data(position, rotation, width, height):
    return load().rco_camera_data(Vec3(*position), Vec3(*rotation), width, height)


def generate_scene(n, seed=0):
    pos = np.zeros((n, 3), np.float32)
    rad = np.zeros((n,), np.float32)
    col = np.zeros((n, 3), np.float32)
    if n:
        load().rco_generate_scene(seed, n, pos.ctypes.data_as(C.c_void_p), rad.ctypes.data_as(C.c_void_p),
                                  col.ctypes.data_as(C.c_void_p))
    return pos, rad, col


def make_scene(pos, rad, col, cam):
    """Scene struct over host numpy arrays (kept alive by the returned tuple)."""
    pos = np.ascontiguousarray(pos, np.float32)
    rad = np.ascontiguousarray(rad, np.float32)
    col = np.ascontiguousarray(col, np.float32)
    s = Scene()
    s.sphere_num = int(rad.shape[0])
    s.position = pos.ctypes.data
    s.radius = rad.ctypes.data
    s.color = col.ctypes.data
    s.cam = cam
    return s, (pos, rad, col)


def screen_over(array, width, height):
    """Screen struct over a host float32 array [height, pitch_floats]."""
    s = Screen()
    s.surface = array.ctypes.data
    s.width, s.height = int(width), int(height)
    s.pitch = array.strides[0]
    return s


def default_ext(has_plane=1, shadows=1):
    """The EXT scene used by the benchmark configurations: floor plane y = -6, one point light."""
    return Ext(Vec3(0.0, 1.0, 0.0), -6.0, Vec3(0.6, 0.6, 0.55), Vec3(8.0, 14.0, 12.0), 0.4, 5, has_plane, shadows)


def render(pos, rad, col, cam, width, height, y_offset=0, full_height=None, threads=None, fill=None, ext=None):
    """Reference-exact frame (or row band) on the CPU -> float32 [height, width, 4].
    With `ext` (an Ext struct) the frame is rendered in EXT mode instead."""
    lib = load()
    full_height = height if full_height is None else full_height
    threads = threads or min(os.cpu_count() or 1, 16)
    out = np.empty((height, width, 4), np.float32)
    if fill is not None:
        out.view(np.uint8)[...] = fill
    scene, keep = make_scene(pos, rad, col, cam)
    screen = screen_over(out.reshape(height, width * 4), width, height)
    if ext is not None:
        lib.rco_render_rows_ext(C.byref(screen), C.byref(scene), C.byref(ext), y_offset, full_height, 0, height, threads)
    else:
        lib.rco_render_rows(C.byref(screen), C.byref(scene), y_offset, full_height, 0, height, threads)
    del keep
    return out


def hit_index(pos, rad, col, cam, width, height, y_offset=0, full_height=None):
    """Winning sphere index per pixel (-1 = miss) -> int32 [height, width]."""
    lib = load()
    full_height = height if full_height is None else full_height
    idx = np.empty((height, width), np.int32)
    dummy = np.empty((1, 4), np.float32)
    scene, keep = make_scene(pos, rad, col, cam)
    screen = Screen(dummy.ctypes.data, width, height, width * 16)
    lib.rco_hit_mask(C.byref(screen), C.byref(scene), y_offset, full_height, None, idx.ctypes.data_as(C.c_void_p))
    del keep
    return idx


def to_rgba8(surface, flip=False):
    lib = load()
    h, w, _ = surface.shape
    surface = np.ascontiguousarray(surface, np.float32)
    out = np.empty((h, w, 4), np.uint8)
    screen = screen_over(surface.reshape(h, w * 4), w, h)
    lib.rco_to_rgba8(C.byref(screen), out.ctypes.data_as(C.c_void_p), w * 4, 1 if flip else 0)
    return out


def fnv_words(surface, basis=FNV_BASIS):
    h, w, _ = surface.shape
    surface = np.ascontiguousarray(surface, np.float32)
    screen = screen_over(surface.reshape(h, w * 4), w, h)
    return int(load().rco_fnv1a64_words(C.byref(screen), basis))
