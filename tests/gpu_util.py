"""Helpers for the GPU parity tests: drive librc_hip.so through the C ABI with numpy data."""
import ctypes as C

import numpy as np

from cudaraytracing_amd import abi
from cudaraytracing_amd.host import DeviceBuffer


def to_rc_camera(cam):
    """oracle Camera struct -> ABI RcCamera (identical 56-byte layout)."""
    out = abi.RcCamera()
    C.memmove(C.addressof(out), C.addressof(cam), C.sizeof(out))
    return out


class GpuScene:
    def __init__(self, pos, rad, col):
        self.n = int(np.asarray(rad).shape[0])
        self.bufs = []
        for a in (pos, rad, col):
            a = np.ascontiguousarray(a, np.float32)
            b = DeviceBuffer(max(a.nbytes, 4))
            if a.nbytes:
                b.upload(a)
            self.bufs.append(b)

    def scene(self, cam):
        s = abi.RcScene()
        s.sphere_num = self.n
        s.position, s.radius, s.color = (b.ptr.value for b in self.bufs)
        s.cam = cam if isinstance(cam, abi.RcCamera) else to_rc_camera(cam)
        return s

    def free(self):
        for b in self.bufs:
            b.free()


def to_rc_ext(ext):
    """oracle Ext struct -> ABI RcExt (identical 56-byte layout)."""
    out = abi.RcExt()
    C.memmove(C.addressof(out), C.addressof(ext), C.sizeof(out))
    return out


def gpu_render(pos, rad, col, cam, width, height, variant=abi.VARIANT_AUTO, y_offset=0, full_height=0,
               pitch=None, want_rgba8=False, flip=False, fill=1, scene=None, ext=None, workspace=False, tuning=0):
    """Render through rc_render -> float32 [height, pitch/4] raw rows (and RGBA8 if asked)."""
    lib = abi.load()
    own = scene is None
    gs = scene if scene is not None else GpuScene(pos, rad, col)
    pitch = pitch or width * 16
    surf = DeviceBuffer(pitch * height)
    abi.check(lib.rc_memset(surf.ptr, fill, pitch * height, None), "rc_memset")
    screen = abi.RcScreen(surf.ptr.value, width, height, pitch)
    opts = abi.RcOpts(y_offset=y_offset, full_height=full_height, variant=variant, tuning=tuning)
    if ext is not None:
        rc_ext = ext if isinstance(ext, abi.RcExt) else to_rc_ext(ext)
        opts.mode = abi.MODE_EXT
        opts.ext = C.pointer(rc_ext)
    rgba = None
    if want_rgba8:
        rgba = DeviceBuffer(width * 4 * height)
        opts.rgba8, opts.rgba8_pitch, opts.rgba8_flip = rgba.ptr.value, width * 4, 1 if flip else 0
    sc = gs.scene(cam)
    ws = None
    if workspace:            # culled variant: records once per frame; plus per-bin lists unless "records"
        nbytes = lib.rc_workspace_bytes(gs.n) if workspace == "records" else lib.rc_workspace_bytes_frame(gs.n, width, height)
        ws = DeviceBuffer(max(nbytes, 16))
        opts.workspace, opts.workspace_bytes = ws.ptr.value, ws.nbytes
    abi.check(lib.rc_render(C.byref(screen), C.byref(sc), C.byref(opts)), "rc_render")
    abi.check(lib.rc_sync(None), "rc_sync")
    raw = np.empty((height, pitch // 4), np.float32)
    abi.check(lib.rc_download_2d(raw.ctypes.data_as(C.c_void_p), pitch, surf.ptr, pitch, pitch, height, None), "download")
    out8 = None
    if want_rgba8:
        out8 = np.empty((height, width, 4), np.uint8)
        abi.check(lib.rc_download_2d(out8.ctypes.data_as(C.c_void_p), width * 4, rgba.ptr, width * 4, width * 4, height, None), "download8")
        rgba.free()
    surf.free()
    if ws is not None:
        ws.free()
    if own:
        gs.free()
    return (raw, out8) if want_rgba8 else raw


def pixels(raw, width):
    """raw rows -> [height, width, 4] view of the pixel part."""
    return raw[:, : width * 4].reshape(raw.shape[0], width, 4)


def assert_same_bits(a, b, what=""):
    a = np.ascontiguousarray(a).view(np.uint32)
    b = np.ascontiguousarray(b).view(np.uint32)
    if not np.array_equal(a, b):
        bad = np.argwhere(a != b)
        raise AssertionError(f"{what}: {len(bad)} differing words, first at {bad[0].tolist()}: "
                             f"{a[tuple(bad[0])]:08x} vs {b[tuple(bad[0])]:08x}")
