"""Times one EXT (or REF) frame format: python tools/ext_sweep.py W H N [ref|ext] [tuning bits] -> ms per frame.
Tuning aid for launcher thresholds (rc_opts.tuning forces either side of a heuristic)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cudaraytracing_amd import abi, configs
from cudaraytracing_amd.host import Camera, DeviceBuffer, generate_scene

w, h, n = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
shading = sys.argv[4] if len(sys.argv) > 4 else "ext"
lib = abi.load()
pos, rad, col = generate_scene(n, 0)
bufs = []
for a in (pos, rad, col):
    b = DeviceBuffer(max(a.nbytes, 4)); b.upload(a); bufs.append(b)
scene = abi.RcScene(); scene.sphere_num = n
scene.position, scene.radius, scene.color = (b.ptr.value for b in bufs)
cam = Camera(w, h); cam.position = (0.0, 0.0, 20.0); scene.cam = cam.get_data()
surf = DeviceBuffer(w * 16 * h)
screen = abi.RcScreen(surf.ptr.value, w, h, w * 16)
tuning = int(sys.argv[5]) if len(sys.argv) > 5 else 0
opts = abi.RcOpts(tuning=tuning)
ext = abi.RcExt(**{k: (abi.RcVec3(*v) if isinstance(v, tuple) else v) for k, v in configs.DEFAULT_EXT.items()})
if shading == "ext":
    opts.mode = abi.MODE_EXT; opts.ext = C.pointer(ext)
e0, e1 = C.c_void_p(), C.c_void_p()
lib.rc_event_create(C.byref(e0)); lib.rc_event_create(C.byref(e1))
est = max(1, int(0.15 / (1e-12 * w * h * max(n, 8) / 25)))      # ~0.15 s of warm-up work through the clock ramp
for _ in range(min(est, 4000)): abi.check(lib.rc_render(C.byref(screen), C.byref(scene), C.byref(opts)))
lib.rc_sync(None)
reps = max(20, min(400, est // 2))
lib.rc_event_record(e0, None)
for _ in range(reps): lib.rc_render(C.byref(screen), C.byref(scene), C.byref(opts))
lib.rc_event_record(e1, None)
ms = C.c_float(); lib.rc_event_elapsed_ms(e0, e1, C.byref(ms))
print(f"{w}x{h} n={n} {shading} tuning={tuning}: {ms.value / reps:.4f} ms")
