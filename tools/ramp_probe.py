#!/usr/bin/env python3
"""Per-frame time of the default workload from a cold start: shows how many frames the GPU
needs to reach its steady clock (sizes the untimed warm-up of bench.py)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cudaraytracing_amd import abi, configs  # noqa: E402
from cudaraytracing_amd.host import Camera, DeviceBuffer, generate_scene  # noqa: E402

lib = abi.load()
cfg = configs.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "c3_3840x2160_n1024"]
w, h, n = cfg["width"], cfg["height"], cfg["n"]
pos, rad, col = generate_scene(n, cfg["seed"])
bufs = []
for a in (pos, rad, col):
    b = DeviceBuffer(max(a.nbytes, 4)); b.upload(a); bufs.append(b)
scene = abi.RcScene(); scene.sphere_num = n
scene.position, scene.radius, scene.color = (b.ptr.value for b in bufs)
cam = Camera(w, h); cam.position, cam.rotation = cfg["position"], cfg["rotation"]
scene.cam = cam.get_data()
surf = DeviceBuffer(w * 16 * h)
screen = abi.RcScreen(surf.ptr.value, w, h, w * 16)
abi.check(lib.rc_prepare(w, h, scene.cam.fov))
N = 600
evs = []
for i in range(N + 1):
    e = C.c_void_p(); abi.check(lib.rc_event_create(C.byref(e))); evs.append(e)
abi.check(lib.rc_sync(None))
abi.check(lib.rc_event_record(evs[0], None))
for i in range(N):
    abi.check(lib.rc_render(C.byref(screen), C.byref(scene), None))
    abi.check(lib.rc_event_record(evs[i + 1], None))
abi.check(lib.rc_sync(None))
per = []
for i in range(N):
    ms = C.c_float(); abi.check(lib.rc_event_elapsed_ms(evs[i], evs[i + 1], C.byref(ms))); per.append(ms.value)
for a in range(0, N, 20):
    print(f"frames {a:3d}-{a + 19:3d}: mean {sum(per[a:a + 20]) / 20:.4f} ms  min {min(per[a:a + 20]):.4f}  max {max(per[a:a + 20]):.4f}")
