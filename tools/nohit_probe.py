"""Times the default workload with all radii scaled by 1e-4 (almost no pixel hits anything):
the difference to the normal frame is what flagged blocks, exact tests and shading cost."""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from cudaraytracing_amd import abi, configs
from cudaraytracing_amd.host import Camera, DeviceBuffer, generate_scene

lib = abi.load()
cfg = configs.CONFIGS["c3_3840x2160_n1024"]
w, h, n = cfg["width"], cfg["height"], cfg["n"]
for scale in (1.0, 1e-4):
    pos, rad, col = generate_scene(n, 0)
    rad = (rad * np.float32(scale)).astype(np.float32)
    bufs = []
    for a in (pos, rad, col):
        b = DeviceBuffer(a.nbytes); b.upload(a); bufs.append(b)
    scene = abi.RcScene(); scene.sphere_num = n
    scene.position, scene.radius, scene.color = (b.ptr.value for b in bufs)
    cam = Camera(w, h); cam.position = cfg["position"]; scene.cam = cam.get_data()
    surf = DeviceBuffer(w * 16 * h)
    screen = abi.RcScreen(surf.ptr.value, w, h, w * 16)
    for variant in (abi.VARIANT_TILED8, abi.VARIANT_TILED4):
        opts = abi.RcOpts(variant=variant)
        e0, e1 = C.c_void_p(), C.c_void_p()
        lib.rc_event_create(C.byref(e0)); lib.rc_event_create(C.byref(e1))
        for _ in range(400): lib.rc_render(C.byref(screen), C.byref(scene), C.byref(opts))   # through the clock ramp
        lib.rc_sync(None)
        lib.rc_event_record(e0, None)
        for _ in range(200): lib.rc_render(C.byref(screen), C.byref(scene), C.byref(opts))
        lib.rc_event_record(e1, None)
        ms = C.c_float(); lib.rc_event_elapsed_ms(e0, e1, C.byref(ms))
        print(f"radius x {scale:g} variant {variant}: {ms.value / 200:.4f} ms/frame")
