"""Fixed cost of the small-frame configurations: frame time of 640x480 x 8 and 1920x1080 x 64 (REF and EXT) next to
the same frames with NO spheres (ray set-up, record build of nothing, stores: everything but the test loop) and
with the pixels-per-lane choice forced, all in one process so that the numbers share one clock state.
    python tools/small_frame_probe.py [tuning_bits [config,config,... [ref,ext]]]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cudaraytracing_amd import abi, configs  # noqa: E402
from cudaraytracing_amd.host import Camera, DeviceBuffer, generate_scene  # noqa: E402

tuning = int(sys.argv[1]) if len(sys.argv) > 1 else 0
names = sys.argv[2].split(",") if len(sys.argv) > 2 else ["c1_640x480_n8", "c2_1920x1080_n64", "ref_1280x768_n512", "c3_3840x2160_n1024"]
shadings = sys.argv[3].split(",") if len(sys.argv) > 3 else ["ref", "ext"]
lib = abi.load()
e0, e1 = C.c_void_p(), C.c_void_p()
lib.rc_event_create(C.byref(e0)); lib.rc_event_create(C.byref(e1))
surf = DeviceBuffer(3840 * 16 * 2160)
def make_ext(**over):
    return abi.RcExt(**{k: (abi.RcVec3(*v) if isinstance(v, tuple) else v) for k, v in dict(configs.DEFAULT_EXT, **over).items()})


# shading names: ref, ext, and EXT with one feature off (where the EXT time goes)
exts = {"ext": make_ext(), "ext_noshadow": make_ext(shadows=0), "ext_noplane": make_ext(has_plane=0),
        "ext_nospec": make_ext(shininess_log2=0, specular=0.0)}
variants = [("auto", abi.VARIANT_AUTO), ("P2", abi.VARIANT_TILED2), ("P4", abi.VARIANT_TILED4)]


def timed(screen, scene, opts, warm, reps):
    for _ in range(warm):
        abi.check(lib.rc_render(C.byref(screen), C.byref(scene), C.byref(opts)))
    lib.rc_sync(None)
    lib.rc_event_record(e0, None)
    for _ in range(reps):
        lib.rc_render(C.byref(screen), C.byref(scene), C.byref(opts))
    lib.rc_event_record(e1, None)
    ms = C.c_float()
    abi.check(lib.rc_event_elapsed_ms(e0, e1, C.byref(ms)))
    return ms.value / reps * 1e3


for name in names:
    cfg = configs.CONFIGS[name]
    w, h = cfg["width"], cfg["height"]
    for n in (cfg["n"], 0):
        pos, rad, col = generate_scene(max(n, 1), cfg["seed"])
        bufs = []
        for a in (pos, rad, col):
            b = DeviceBuffer(max(a.nbytes, 4)); b.upload(a); bufs.append(b)
        scene = abi.RcScene(); scene.sphere_num = n
        scene.position, scene.radius, scene.color = (b.ptr.value for b in bufs)
        cam = Camera(w, h); cam.position, cam.rotation = cfg["position"], cfg["rotation"]
        scene.cam = cam.get_data()
        screen = abi.RcScreen(surf.ptr.value, w, h, w * 16)
        for shading in shadings:
            cells = []
            for vname, v in variants:
                opts = abi.RcOpts(variant=v, tuning=tuning)
                if shading != "ref":
                    opts.mode = abi.MODE_EXT; opts.ext = C.pointer(exts[shading])
                big = w * h * max(n, 8) > 2e9
                t = timed(screen, scene, opts, 400 if big else 6000, 200 if big else 2000)
                cells.append(f"{vname} {t:8.2f} us")
            print(f"{name:20s} n={n:5d} {shading:12s}  " + "  ".join(cells), flush=True)
        for b in bufs:
            b.free()
