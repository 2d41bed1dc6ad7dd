"""How much would the EXT shadow pass gain from visiting likely occluders first?  Times the c3 EXT frame with the
scene in its generated order, sorted by apparent size from the light (largest first) and in the reverse of that.
Only the order of the spheres differs (the any-hit exit leaves the loop once every shadow ray of a wave is
occluded), so the spread bounds what an in-kernel ordering could buy.
    python tools/shadow_order_probe.py"""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cudaraytracing_amd import abi, configs
from cudaraytracing_amd.host import Camera, DeviceBuffer, generate_scene

cfg = configs.CONFIGS["c3_3840x2160_n1024"]
w, h, n = cfg["width"], cfg["height"], cfg["n"]
lib = abi.load()
pos, rad, col = generate_scene(n, cfg["seed"])
light = np.array(configs.DEFAULT_EXT["light_pos"], dtype=np.float64)
ang = rad.astype(np.float64) / np.linalg.norm(pos.astype(np.float64) - light, axis=1)
orders = {"generated": np.arange(n), "largest first": np.argsort(-ang), "smallest first": np.argsort(ang)}
surf = DeviceBuffer(w * 16 * h)
e0, e1 = C.c_void_p(), C.c_void_p()
lib.rc_event_create(C.byref(e0)); lib.rc_event_create(C.byref(e1))
ext = abi.RcExt(**{k: (abi.RcVec3(*v) if isinstance(v, tuple) else v) for k, v in configs.DEFAULT_EXT.items()})
for name, order in orders.items():
    bufs = []
    for a in (pos[order], rad[order], col[order]):
        a = np.ascontiguousarray(a); b = DeviceBuffer(a.nbytes); b.upload(a); bufs.append(b)
    scene = abi.RcScene(); scene.sphere_num = n
    scene.position, scene.radius, scene.color = (b.ptr.value for b in bufs)
    cam = Camera(w, h); cam.position, cam.rotation = cfg["position"], cfg["rotation"]; scene.cam = cam.get_data()
    screen = abi.RcScreen(surf.ptr.value, w, h, w * 16)
    for shading in ("ref", "ext"):
        opts = abi.RcOpts()
        if shading == "ext":
            opts.mode = abi.MODE_EXT; opts.ext = C.pointer(ext)
        for _ in range(200): abi.check(lib.rc_render(C.byref(screen), C.byref(scene), C.byref(opts)))
        lib.rc_sync(None)
        lib.rc_event_record(e0, None)
        for _ in range(100): lib.rc_render(C.byref(screen), C.byref(scene), C.byref(opts))
        lib.rc_event_record(e1, None)
        ms = C.c_float(); lib.rc_event_elapsed_ms(e0, e1, C.byref(ms))
        print(f"{name:15s} {shading}: {ms.value / 100:.4f} ms", flush=True)
    for b in bufs: b.free()
