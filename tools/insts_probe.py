"""A fixed number of launches of ONE workload / shading / pixels-per-lane choice, for counter runs:
    rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d OUT -- python3 tools/insts_probe.py c2_1920x1080_n64 ext_noshadow auto
shading: ref | ext | ext_noshadow | ext_noplane | ext_nospec; variant: auto | P2 | P4 | P8."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cudaraytracing_amd import abi, configs  # noqa: E402
from cudaraytracing_amd.host import Camera, DeviceBuffer, generate_scene  # noqa: E402

name, shading, vname = sys.argv[1], sys.argv[2], (sys.argv[3] if len(sys.argv) > 3 else "auto")
launches = int(sys.argv[4]) if len(sys.argv) > 4 else 40
lib = abi.load()
cfg = configs.CONFIGS[name]
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
over = {"ext": {}, "ext_noshadow": dict(shadows=0), "ext_noplane": dict(has_plane=0), "ext_nospec": dict(shininess_log2=0, specular=0.0)}
opts = abi.RcOpts(variant={"auto": abi.VARIANT_AUTO, "P2": abi.VARIANT_TILED2, "P4": abi.VARIANT_TILED4, "P8": abi.VARIANT_TILED8}[vname])
if shading != "ref":
    ext = abi.RcExt(**{k: (abi.RcVec3(*v) if isinstance(v, tuple) else v) for k, v in dict(configs.DEFAULT_EXT, **over[shading]).items()})
    opts.mode = abi.MODE_EXT; opts.ext = C.pointer(ext)
for _ in range(launches):
    abi.check(lib.rc_render(C.byref(screen), C.byref(scene), C.byref(opts)))
abi.check(lib.rc_sync(None))
