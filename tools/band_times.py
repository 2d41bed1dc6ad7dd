"""Per-band render time of one frame cut into N row bands, each band rendered alone on this GPU exactly as
rank r of N would render it (rc_opts.y_offset / full_height), next to the whole-frame time: what the
row-band mode can gain at best on N GPUs before any gather, and how even the bands are.
    python tools/band_times.py [config] [N] [subbands] [contiguous|interleaved] [streams]

streams = 2 puts a rank's pieces on two streams in turn (joined by events at the frame's ends), so that the
second piece's workgroups fill the CUs that the first piece's last workgroups leave idle."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cudaraytracing_amd import abi, configs, sharding
from cudaraytracing_amd.host import Camera, DeviceBuffer, generate_scene

name = sys.argv[1] if len(sys.argv) > 1 else "c4_7680x4320_n4096"
world = int(sys.argv[2]) if len(sys.argv) > 2 else 8
parts = int(sys.argv[3]) if len(sys.argv) > 3 else 1
layout = sys.argv[4] if len(sys.argv) > 4 else "contiguous"
nstreams = int(sys.argv[5]) if len(sys.argv) > 5 else 1
variant = getattr(abi, "VARIANT_" + os.environ.get("BT_VARIANT", "AUTO").upper())   # BT_VARIANT=tiled8 forces a kernel
cfg = configs.CONFIGS[name]
w, h, n = cfg["width"], cfg["height"], cfg["n"]
lib = abi.load()
pos, rad, col = generate_scene(n, cfg["seed"])
bufs = []
for a in (pos, rad, col):
    b = DeviceBuffer(max(a.nbytes, 4)); b.upload(a); bufs.append(b)
scene = abi.RcScene(); scene.sphere_num = n
scene.position, scene.radius, scene.color = (b.ptr.value for b in bufs)
cam = Camera(w, h); cam.position, cam.rotation = cfg["position"], cfg["rotation"]; scene.cam = cam.get_data()
surf = DeviceBuffer(w * 16 * h)
e0, e1 = C.c_void_p(), C.c_void_p()
lib.rc_event_create(C.byref(e0)); lib.rc_event_create(C.byref(e1))


streams = []
for _ in range(nstreams):
    st = C.c_void_p(); abi.check(lib.rc_stream_create(C.byref(st))); streams.append(st)
fork, join = C.c_void_p(), C.c_void_p()
lib.rc_event_create(C.byref(fork)); lib.rc_event_create(C.byref(join))


def timed(pieces, reps):
    def frame():
        if nstreams > 1:                          # the other streams start where stream 0 stands
            lib.rc_event_record(fork, streams[0])
            for st in streams[1:]:
                lib.rc_stream_wait_event(st, fork)
        for k, (b, e) in enumerate(pieces):
            screen = abi.RcScreen(surf.ptr.value + b * w * 16, w, e - b, w * 16)
            opts = abi.RcOpts(y_offset=b, full_height=h, stream=streams[k % nstreams].value, variant=variant)
            abi.check(lib.rc_render(C.byref(screen), C.byref(scene), C.byref(opts)))
        for st in streams[1:]:                    # and stream 0 waits for them
            lib.rc_event_record(join, st)
            lib.rc_stream_wait_event(streams[0], join)
    for _ in range(reps): frame()                 # warm-up through the clock ramp
    lib.rc_sync(None)
    lib.rc_event_record(e0, streams[0])
    for _ in range(reps): frame()
    lib.rc_event_record(e1, streams[0])
    ms = C.c_float(); lib.rc_event_elapsed_ms(e0, e1, C.byref(ms))
    return ms.value / reps


whole = timed([(0, h)], 60)
print(f"{name}: whole frame {whole:.4f} ms; {world} ranks, layout {layout}, {parts} piece(s) where contiguous, {nstreams} stream(s):")
tot, worst = 0.0, 0.0
for r in range(world):
    pieces = sharding.rank_pieces(h, world, r, parts, layout)
    t = timed(pieces, 150)
    tot += t; worst = max(worst, t)
    print(f"  rank {r}: rows {' + '.join(f'[{b}, {e})' for b, e in pieces)}  {t:.4f} ms")
print(f"  sum of bands {tot:.4f} ms ({tot / whole:.3f} x whole); slowest band {worst:.4f} ms -> render-only speed-up "
      f"{whole / worst:.2f} of {world} ({whole / worst / world:.2f} efficiency)")
