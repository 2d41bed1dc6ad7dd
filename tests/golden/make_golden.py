"""Generates tests/golden/golden.json from the CPU oracle (run once, in the build container).

The oracle is pinned first (ref_kat.json), then rendered at every benchmark configuration.
Per frame the fixture keeps: 64-bit FNV-1a (word-wise, standard basis) of the float RGBA
surface, of the winner-index map and of the RGBA8 conversion; the hit count; CameraData bits;
and 192 sampled pixels with raw float bits.  Full-size frames take minutes on 8 cores, which is
why GPU tests compare against these hashes instead of re-running the oracle.

    python tests/golden/make_golden.py [--full] [--ext-full]
        --full      adds the 4K and 8K REF frames and the REF frames of the camera path
        --ext-full  adds EXT at 3840x2160 x 1024 and two EXT frames of the sequence (moving camera AND
                    orbiting light, configs.sequence_light); EXT fixtures record their rc_ext parameters
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from cudaraytracing_amd import configs  # noqa: E402
from oracle import binding as ob  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def fnv_words_np(a):
    """Same hash as rco_fnv1a64_words over a uint32 view, computed by the oracle library."""
    a = np.ascontiguousarray(a)
    w = a.view(np.uint32).reshape(1, -1)
    # borrow the oracle's hasher through a fake 1-row screen (width*4 words per row)
    assert w.shape[1] % 4 == 0
    import ctypes as C
    s = ob.Screen(w.ctypes.data, w.shape[1] // 4, 1, w.shape[1] * 4)
    return int(ob.load().rco_fnv1a64_words(C.byref(s), ob.FNV_BASIS))


def cam_bits(cam):
    import ctypes as C
    return np.frombuffer(C.string_at(C.addressof(cam), C.sizeof(cam)), np.uint32).tolist()


def frame_fixture(name, width, height, n, seed, position, rotation, samples=192, ext=None):
    t0 = time.time()
    pos, rad, col = ob.generate_scene(n, seed)
    cam = ob.camera_data(position, rotation, width, height)
    surf = ob.render(pos, rad, col, cam, width, height, threads=os.cpu_count(), ext=ext)
    rgba8 = ob.to_rgba8(surf, flip=True)
    rng = np.random.default_rng(1234)
    hit = np.argwhere(surf[..., :3].any(axis=2))
    picks = []
    if len(hit):
        sel = rng.choice(len(hit), size=min(samples * 2 // 3, len(hit)), replace=False)
        picks += [(int(hit[i][1]), int(hit[i][0])) for i in sel]
    while len(picks) < samples:
        picks.append((int(rng.integers(width)), int(rng.integers(height))))
    bits = surf.view(np.uint32)
    fx = dict(name=name, width=width, height=height, n=n, seed=seed,
              position=list(position), rotation=list(rotation), mode="EXT" if ext is not None else "REF",
              camera_bits=cam_bits(cam),
              ext=(None if ext is None else dict(
                  plane_normal=[ext.plane_normal.x, ext.plane_normal.y, ext.plane_normal.z], plane_h=ext.plane_h,
                  plane_color=[ext.plane_color.x, ext.plane_color.y, ext.plane_color.z],
                  light_pos=[ext.light_pos.x, ext.light_pos.y, ext.light_pos.z], specular=ext.specular,
                  shininess_log2=ext.shininess_log2, has_plane=ext.has_plane, shadows=ext.shadows)),
              fnv=f"{ob.fnv_words(surf):016x}",
              rgba8_flip_fnv=f"{fnv_words_np(rgba8):016x}",
              lit_pixels=int(surf[..., :3].any(axis=2).sum()),
              samples=[[x, y] + bits[y, x].tolist() for x, y in picks])
    print(f"{name}: {time.time() - t0:.1f}s fnv={fx['fnv']} lit={fx['lit_pixels']}", flush=True)
    return fx


def main():
    full = "--full" in sys.argv
    out_path = os.path.join(HERE, "golden.json")
    old = {}
    if os.path.exists(out_path):
        old = {f["name"]: f for f in json.load(open(out_path))["frames"]}
    frames = []
    small = ["c1_640x480_n8", "c2_1920x1080_n64", "ref_1280x768_n512"]
    big = ["c3_3840x2160_n1024", "c4_7680x4320_n4096"]
    for name in small + big:
        c = configs.CONFIGS[name]
        if name in big and not full:
            if name in old:
                frames.append(old[name])
            continue
        frames.append(frame_fixture(name, c["width"], c["height"], c["n"], c["seed"], c["position"], c["rotation"]))
    # EXT mode (plane + point light + Phong + shadows): defined by the oracle, unpinned by the reference
    for name in ("c1_640x480_n8", "c2_1920x1080_n64"):
        c = configs.CONFIGS[name]
        frames.append(frame_fixture(name + "_ext", c["width"], c["height"], c["n"], c["seed"], c["position"],
                                    c["rotation"], ext=ob.Ext(**{k: (ob.Vec3(*v) if isinstance(v, tuple) else v)
                                                                 for k, v in configs.DEFAULT_EXT.items()})))
    def make_ext(**over):
        kw = dict(configs.DEFAULT_EXT, **over)
        return ob.Ext(**{k: (ob.Vec3(*v) if isinstance(v, tuple) else v) for k, v in kw.items()})

    ext_full = "--ext-full" in sys.argv
    s = configs.SEQUENCE
    ext_big = [("c3_3840x2160_n1024_ext", configs.CONFIGS["c3_3840x2160_n1024"]["position"],
                configs.CONFIGS["c3_3840x2160_n1024"]["rotation"], make_ext())]
    for f in (37, 200):
        p, r = configs.sequence_camera(f)
        ext_big.append((f"c5_seq_ext_f{f}", p, r, make_ext(light_pos=configs.sequence_light(f))))
    for name, p, r, e in ext_big:
        if not ext_full:
            if name in old:
                frames.append(old[name])
            continue
        frames.append(frame_fixture(name, s["width"], s["height"], s["n"], s["seed"], p, r, ext=e))
    for f in (0, 150, 299, 37):
        name = f"c5_seq_f{f}"
        if not full:
            if name in old:
                frames.append(old[name])
            continue
        p, r = configs.sequence_camera(f)
        frames.append(frame_fixture(name, s["width"], s["height"], s["n"], s["seed"], p, r))
    json.dump(dict(_comment="generated by tests/golden/make_golden.py from the CPU oracle; see its docstring",
                   fnv_offset_basis=ob.FNV_BASIS, frames=frames), open(out_path, "w"), indent=0)
    print("wrote", out_path)


if __name__ == "__main__":
    main()
