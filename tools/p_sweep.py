"""Pixels-per-lane crossover of the brute-force REF kernels: every frame format x sphere count below with 2, 4 and
8 pixels per lane forced (rc_opts.variant), in one process so that all numbers share one clock state.  Prints the
workgroup count of each choice and marks the fastest: the data behind the automatic choice in rc_launch_draw.
    python tools/p_sweep.py [ref|ext] [count,count,...]      (P_SWEEP_TUNING=<rc_opts.tuning bits> for every launch)"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cudaraytracing_amd import abi, configs
from cudaraytracing_amd.host import Camera, DeviceBuffer, generate_scene

shading = sys.argv[1] if len(sys.argv) > 1 else "ref"
tuning = int(os.environ.get("P_SWEEP_TUNING", "0"), 0)
# (width, rows rendered, height of the whole frame, first row): the last two are row bands of a 7680x4320 frame (one of 16
# and one of 8, from the middle, where the spheres are), rendered as bands -- the camera sees the whole frame
formats = [(640, 480, 480, 0), (1280, 768, 768, 0), (1920, 1080, 1080, 0), (2560, 1440, 1440, 0), (3840, 1080, 1080, 0),
           (3840, 2160, 2160, 0), (7680, 270, 4320, 2160), (7680, 540, 4320, 2160)]
if os.environ.get("P_SWEEP_FORMATS"):      # "width,rows,frame height,first row;..." replaces the list above
    formats = [tuple(int(v) for v in f.split(",")) for f in os.environ["P_SWEEP_FORMATS"].split(";")]
counts = [int(c) for c in sys.argv[2].split(",")] if len(sys.argv) > 2 else [16, 64, 256, 1024, 4096]
variants = [("P2", abi.VARIANT_TILED2, 2), ("P4", abi.VARIANT_TILED4, 4)] + ([("P8", abi.VARIANT_TILED8, 8)] if shading == "ref" else [])
lib = abi.load()
e0, e1 = C.c_void_p(), C.c_void_p()
lib.rc_event_create(C.byref(e0)); lib.rc_event_create(C.byref(e1))
surf = DeviceBuffer(7680 * 16 * 2160)
ext = abi.RcExt(**{k: (abi.RcVec3(*v) if isinstance(v, tuple) else v) for k, v in configs.DEFAULT_EXT.items()})

for n in counts:
    pos, rad, col = generate_scene(n, 0)
    bufs = []
    for a in (pos, rad, col):
        b = DeviceBuffer(max(a.nbytes, 4)); b.upload(a); bufs.append(b)
    scene = abi.RcScene(); scene.sphere_num = n
    scene.position, scene.radius, scene.color = (b.ptr.value for b in bufs)
    for (w, h, full_h, y0) in formats:
        cam = Camera(w, full_h); cam.position = (0.0, 0.0, 20.0); scene.cam = cam.get_data()
        screen = abi.RcScreen(surf.ptr.value, w, h, w * 16)
        row = []
        for name, v, p in variants:
            opts = abi.RcOpts(variant=v, y_offset=y0, full_height=full_h, tuning=tuning)
            if shading == "ext":
                opts.mode = abi.MODE_EXT; opts.ext = C.pointer(ext)
            est = max(1, int(0.06 / (1e-12 * w * h * max(n, 8) / 25)))
            for _ in range(min(est, 3000)): abi.check(lib.rc_render(C.byref(screen), C.byref(scene), C.byref(opts)))
            lib.rc_sync(None)
            reps = max(20, min(300, est // 2))
            lib.rc_event_record(e0, None)
            for _ in range(reps): lib.rc_render(C.byref(screen), C.byref(scene), C.byref(opts))
            lib.rc_event_record(e1, None)
            ms = C.c_float(); lib.rc_event_elapsed_ms(e0, e1, C.byref(ms))
            tiles = ((w + 16 * p - 1) // (16 * p)) * ((h + 15) // 16)
            row.append((name, tiles, ms.value / reps))
        opts = abi.RcOpts(variant=abi.VARIANT_AUTO, y_offset=y0, full_height=full_h, tuning=tuning)
        if shading == "ext":
            opts.mode = abi.MODE_EXT; opts.ext = C.pointer(ext)
        for _ in range(reps): lib.rc_render(C.byref(screen), C.byref(scene), C.byref(opts))
        lib.rc_sync(None)
        lib.rc_event_record(e0, None)
        for _ in range(reps): lib.rc_render(C.byref(screen), C.byref(scene), C.byref(opts))
        lib.rc_event_record(e1, None)
        ms = C.c_float(); lib.rc_event_elapsed_ms(e0, e1, C.byref(ms))
        best = min(row, key=lambda r: r[2])
        cells = "  ".join(f"{nm} {t:6d} wg {m:8.4f} ms{'*' if nm == best[0] else ' '}" for nm, t, m in row)
        shape = f"{w}x{h}" + (f" of {full_h}@{y0}" if full_h != h else "")
        print(f"n={n:5d} {shape:>22s} {cells}  auto {ms.value / reps:8.4f} ms ({ms.value / reps / best[2]:.3f} x best)", flush=True)
    for b in bufs:
        b.free()
