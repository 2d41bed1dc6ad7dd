"""Executed-work counts of the draw kernels, from an INSTRUMENTED build of the library (hipcc -DRC_COUNT_TESTS:
wave-level tallies at the points marked RC_COUNT in csrc/rc_kernels.hip; the shipped library carries none).

    python tools/count_tests.py [--build] [--out profiles/r3_executed_tests.json]

One frame of every benchmark workload, REF and EXT: how many (pixel, sphere) tests the loops actually ran, how many
blocks were flagged, how many spheres were visited and given the exact test.  bench.py prices the EXT shadow pass
with `shadow_tests_executed` from the JSON this writes instead of the algorithmic hit pixels x spheres."""
import ctypes as C
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cudaraytracing_amd import abi, configs  # noqa: E402

LIB = os.path.join(ROOT, "build", "variants", "librc_count.so")
if "--build" in sys.argv or not os.path.exists(LIB):
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
                    "-Wno-unused-result", "-DRC_COUNT_TESTS", "-I" + os.path.join(ROOT, "include"),
                    "-I" + os.path.join(ROOT, "cudaraytracing_amd", "csrc"), "-shared", "-o", LIB,
                    os.path.join(ROOT, "cudaraytracing_amd", "csrc", "rc_kernels.hip"),
                    os.path.join(ROOT, "cudaraytracing_amd", "csrc", "rc_host.hip")], check=True)
abi.LIB_PATH = LIB                                   # every binding below talks to the instrumented build
from cudaraytracing_amd.host import Camera, DeviceBuffer, generate_scene  # noqa: E402

lib = abi.load()
lib.rc_counters_fetch.argtypes = [C.POINTER(C.c_ulonglong * 16), C.c_int]
out_path = sys.argv[sys.argv.index("--out") + 1] if "--out" in sys.argv else None
cu = C.c_int()
abi.check(lib.rc_device_info(0, C.byref(cu), None, None))


result = {}
surf = DeviceBuffer(7680 * 16 * 4320)
for name, cfg in configs.CONFIGS.items():
    w, h, n = cfg["width"], cfg["height"], cfg["n"]
    pos, rad, col = generate_scene(n, cfg["seed"])
    bufs = []
    for a in (pos, rad, col):
        b = DeviceBuffer(max(a.nbytes, 4)); b.upload(a); bufs.append(b)
    scene = abi.RcScene(); scene.sphere_num = n
    scene.position, scene.radius, scene.color = (b.ptr.value for b in bufs)
    cam = Camera(w, h); cam.position, cam.rotation = cfg["position"], cfg["rotation"]
    scene.cam = cam.get_data()
    screen = abi.RcScreen(surf.ptr.value, w, h, w * 16)
    ext = abi.RcExt(**{k: (abi.RcVec3(*v) if isinstance(v, tuple) else v) for k, v in configs.DEFAULT_EXT.items()})
    for shading in ("ref", "ext"):
        opts = abi.RcOpts()
        if shading == "ext":
            opts.mode = abi.MODE_EXT; opts.ext = C.pointer(ext)
        c = (C.c_ulonglong * 16)()
        abi.check(lib.rc_sync(None))
        assert lib.rc_counters_fetch(C.byref(c), 1) == 0
        abi.check(lib.rc_render(C.byref(screen), C.byref(scene), C.byref(opts)))
        abi.check(lib.rc_sync(None))
        assert lib.rc_counters_fetch(C.byref(c), 1) == 0
        p = configs.pixels_per_lane(w, h, n, shading == "ext", False, cu.value)
        per_block = 32 * 64 * p                      # (pixel, sphere) pairs one wave evaluates per block
        r = {"pixels_per_lane": p, "waves": c[8], "tests_algorithmic": w * h * n,
             "primary_tests_executed": c[0] * per_block, "primary_blocks": c[0], "primary_blocks_flagged": c[1],
             "primary_spheres_visited": c[2], "primary_exact_lane_tests": c[3] * 64}
        if shading == "ext":
            r.update({"shadow_tests_executed": c[4] * per_block, "shadow_blocks": c[4], "shadow_blocks_flagged": c[5],
                      "shadow_spheres_visited": c[6], "shadow_exact_lane_tests": c[7] * 64 * p,
                      "waves_with_shadow_rays": c[9], "waves_left_shadow_loop_early": c[10]})
        result[f"{name}|{shading}"] = r
        print(name, shading, json.dumps(r), flush=True)
    for b in bufs:
        b.free()
if out_path:
    result["_comment"] = ("Executed-work counts of one frame per workload from the instrumented build (tools/count_tests.py). "
                          "*_tests_executed count every lane of a wave that runs a block, padding lanes and padded spheres "
                          "included; *_exact_lane_tests likewise count whole waves.")
    with open(out_path, "w") as f:
        json.dump(result, f, indent=1, sort_keys=True)
        f.write("\n")
