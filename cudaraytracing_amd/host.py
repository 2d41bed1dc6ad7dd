"""Python mirror of the reference's host classes, on top of the C ABI (test/bench plumbing).

    Camera            Camera.h:21-113
    SceneManager      SceneManager.{h,cpp}
    RayTracingKernel  RayTracingKernel.{h,cpp}  (D3D interop replaced by a plain device
                      framebuffer that is read back to host memory or written as PPM)

The C++ mirror with the same names lives in csrc/host/*.hpp; this module only drives the
shared library from tests and bench.py.  Nothing here computes pixels.
"""
import ctypes as C

import numpy as np

from . import abi
from .abi import RcCamera, RcOpts, RcScene, RcScreen, RcVec3, check


class DeviceBuffer:
    """Device allocation owned through rc_malloc / rc_free (SceneManager.cpp:17-24,49-51)."""

    def __init__(self, nbytes):
        self.lib = abi.load()
        self.ptr = C.c_void_p()
        self.nbytes = int(nbytes)
        check(self.lib.rc_malloc(C.byref(self.ptr), self.nbytes), "rc_malloc")

    def upload(self, array, stream=None):
        a = np.ascontiguousarray(array)
        assert a.nbytes <= self.nbytes
        check(self.lib.rc_upload(self.ptr, a.ctypes.data_as(C.c_void_p), a.nbytes, stream), "rc_upload")

    def free(self):
        if self.ptr:
            check(self.lib.rc_free(self.ptr), "rc_free")
            self.ptr = C.c_void_p()

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Camera:
    """Camera.h:21-113: position + Euler rotation -> CameraData."""

    def __init__(self, width, height):
        self.width, self.height = int(width), int(height)
        self.position = (0.0, 0.0, 0.0)
        self.rotation = (0.0, 0.0, 0.0)
        self.movement_speed = 100.0      # Camera.h:48
        self.rotation_speed = 10.0       # Camera.h:49

    def get_data(self):
        lib = abi.load()
        out = RcCamera()
        p, r = RcVec3(*self.position), RcVec3(*self.rotation)
        lib.rc_camera_data(C.byref(p), C.byref(r), self.width, self.height, C.byref(out))
        return out

    def handle_input(self, dt_ms, keys="", cursor=(0, 0)):
        """One poll of the reference's HandleInput (main.cpp:53-128), scripted.  keys: any of
        'r' (right mouse button), 'shift', 'q', 'z', 'w', 'a', 's', 'd' separated by spaces;
        cursor: (dx, dy) since the previous poll; dt_ms: frame time in milliseconds as at
        main.cpp:137.  Returns True if the camera changed (the caller then updates the scene's
        CameraData, main.cpp:126-127)."""
        lib = abi.load()
        held = set(keys.split())
        unknown = held - {"r", "shift", "q", "z", "w", "a", "s", "d"}
        if unknown:
            raise ValueError(f"unknown keys {sorted(unknown)}")
        inp = abi.RcInput(rbutton="r" in held, lshift="shift" in held, key_q="q" in held, key_z="z" in held,
                          key_w="w" in held, key_a="a" in held, key_s="s" in held, key_d="d" in held,
                          cursor_dx=int(cursor[0]), cursor_dy=int(cursor[1]))
        p, r = RcVec3(*self.position), RcVec3(*self.rotation)
        rc = lib.rc_handle_input(C.byref(p), C.byref(r), self.movement_speed, self.rotation_speed, dt_ms, C.byref(inp))
        if rc < 0:
            check(rc, "rc_handle_input")
        self.position, self.rotation = (p.x, p.y, p.z), (r.x, r.y, r.z)
        return rc == 1


def generate_scene(n, seed=0):
    """SceneManager.cpp:14-15,31-46 through rc_generate_scene -> (position[n,3], radius[n], color[n,3])."""
    lib = abi.load()
    pos = np.zeros((n, 3), np.float32)
    rad = np.zeros((n,), np.float32)
    col = np.zeros((n, 3), np.float32)
    if n:
        lib.rc_generate_scene(seed, n, pos.ctypes.data_as(C.c_void_p), rad.ctypes.data_as(C.c_void_p),
                              col.ctypes.data_as(C.c_void_p))
    return pos, rad, col


class SceneManager:
    """SceneManager.{h,cpp}: owns the camera and the device-resident scene arrays."""

    def __init__(self):
        self.cam = None
        self.scene = RcScene()
        self._bufs = []
        self.host = None

    def init_scene(self, width, height, sphere_num=32 * 16, seed=0, arrays=None):
        self.cam = Camera(width, height)
        self.cam.position = (0.0, 0.0, 20.0)      # SceneManager.cpp:5
        self.update_camera()
        pos, rad, col = arrays if arrays is not None else generate_scene(sphere_num, seed)
        pos = np.ascontiguousarray(pos, np.float32).reshape(-1, 3)
        rad = np.ascontiguousarray(rad, np.float32).reshape(-1)
        col = np.ascontiguousarray(col, np.float32).reshape(-1, 3)
        n = rad.shape[0]
        assert pos.shape[0] == n and col.shape[0] == n
        self.host = (pos, rad, col)
        self.cleanup_arrays()
        self.scene.sphere_num = n
        bufs = []
        for arr in (pos, rad, col):
            b = DeviceBuffer(max(arr.nbytes, 4))
            if arr.nbytes:
                b.upload(arr)
            bufs.append(b)
        self._bufs = bufs
        self.scene.position = bufs[0].ptr.value
        self.scene.radius = bufs[1].ptr.value
        self.scene.color = bufs[2].ptr.value

    def update_camera(self):                     # SceneManager.cpp:55-57
        self.scene.cam = self.cam.get_data()

    def cleanup_arrays(self):
        for b in self._bufs:
            b.free()
        self._bufs = []

    def cleanup(self):                           # SceneManager.cpp:48-53
        self.cleanup_arrays()
        self.cam = None


class RayTracingKernel:
    """RayTracingKernel.{h,cpp} without D3D: Init -> RegisterSurface -> Run -> read back."""

    def __init__(self):
        self.lib = abi.load()
        self.screen = RcScreen()
        self.sm = SceneManager()
        self._surface = None
        self.variant = abi.VARIANT_AUTO

    def find_device(self):                       # RayTracingKernel.cpp:47-82
        n = C.c_int()
        check(self.lib.rc_device_count(C.byref(n)), "rc_device_count")
        names = []
        for d in range(n.value):
            buf = C.create_string_buffer(256)
            check(self.lib.rc_device_name(d, buf, 256), "rc_device_name")
            names.append(buf.value.decode())
        return names

    def init(self, width, height, sphere_num=32 * 16, seed=0, arrays=None):   # :34-45
        self.screen.width, self.screen.height = int(width), int(height)
        if not self.find_device():
            return False
        self.sm.init_scene(width, height, sphere_num, seed, arrays)
        return True

    def register_surface(self, pitch=None):      # RayTracingKernel.cpp:93-94 (pitch alloc + memset 1)
        w, h = self.screen.width, self.screen.height
        if self._surface is not None:
            self._surface.free()
        if pitch is None:
            ptr, p = C.c_void_p(), C.c_size_t()
            check(self.lib.rc_malloc_pitch(C.byref(ptr), C.byref(p), w * 16, h), "rc_malloc_pitch")
            buf = DeviceBuffer.__new__(DeviceBuffer)
            buf.lib, buf.ptr, buf.nbytes = self.lib, ptr, p.value * h
            self._surface, pitch = buf, p.value
        else:
            self._surface = DeviceBuffer(pitch * h)
        self.screen.surface = self._surface.ptr.value
        self.screen.pitch = pitch
        check(self.lib.rc_memset(self._surface.ptr, 1, pitch * h, None), "rc_memset")

    def run(self, opts=None, sync=True):         # RayTracingKernel.cpp:8-32 (launch + sync)
        o = opts if opts is not None else RcOpts(variant=self.variant)
        check(self.lib.rc_render(C.byref(self.screen), C.byref(self.sm.scene), C.byref(o)), "rc_render")
        if sync:
            check(self.lib.rc_sync(None), "rc_sync")

    def read_surface(self):
        """Float RGBA surface -> numpy [height, width, 4] (replaces the interop copy, :22-27)."""
        w, h = self.screen.width, self.screen.height
        out = np.empty((h, w, 4), np.float32)
        check(self.lib.rc_download_2d(out.ctypes.data_as(C.c_void_p), w * 16, self._surface.ptr,
                                      self.screen.pitch, w * 16, h, None), "rc_download_2d")
        return out

    def cleanup(self):                           # RayTracingKernel.cpp:97-102
        if self._surface is not None:
            self._surface.free()
            self._surface = None
        self.sm.cleanup()
