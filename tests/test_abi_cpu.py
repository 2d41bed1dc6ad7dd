"""C-ABI checks that need no GPU: the library loads, exports every symbol the header declares,
its structs have the reference layouts, its host-side helpers agree with the oracle, and
rc_render rejects bad arguments before touching the device."""
import ctypes as C
import os
import re
import subprocess

import numpy as np

from conftest import ROOT
from cudaraytracing_amd import abi, configs, host


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "rc_abi.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = set(re.findall(r"\b(rc_[a-z0-9_]+)\s*\(", text))
    return names


def test_library_exports_every_declared_symbol(rclib):
    declared = _declared_symbols()
    assert len(declared) >= 20
    assert declared <= set(abi.SYMBOLS), declared - set(abi.SYMBOLS)
    out = subprocess.run(["nm", "-D", "--defined-only", abi.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = {line.split()[-1] for line in out.splitlines() if line.strip()}
    for name in declared | {"execute_kernel"}:
        assert name in exported, f"{name} is declared but not exported"


def test_library_does_not_link_the_oracle():
    out = subprocess.run(["ldd", abi.LIB_PATH], capture_output=True, text=True).stdout
    assert "oracle" not in out
    syms = subprocess.run(["nm", "-D", abi.LIB_PATH], capture_output=True, text=True).stdout
    assert "rco_" not in syms


def test_struct_layouts():
    assert C.sizeof(abi.RcVec3) == 12
    assert C.sizeof(abi.RcCamera) == 56
    assert C.sizeof(abi.RcScene) == 88 and abi.RcScene.cam.offset == 32
    assert C.sizeof(abi.RcScreen) == 24 and abi.RcScreen.pitch.offset == 16


def test_version_and_error_string(rclib):
    assert rclib.rc_version().startswith(b"rc_hip")
    assert isinstance(rclib.rc_last_error_string(), bytes)


def test_generate_scene_matches_oracle(oracle):
    for n, seed in [(0, 0), (1, 0), (8, 0), (1024, 0), (4096, 7)]:
        a = host.generate_scene(n, seed)
        b = oracle.generate_scene(n, seed)
        for x, y in zip(a, b):
            assert np.array_equal(x.view(np.uint32), y.view(np.uint32))
    pos, rad, col = host.generate_scene(4096, 0)
    assert rad.min() >= 0.3 and rad.max() <= 1.0
    assert np.all(np.abs(pos[:, 0]) <= 10) and np.all(np.abs(pos[:, 1]) <= 5)
    assert pos[:, 2].min() >= -15 and pos[:, 2].max() <= -5
    assert col.min() >= 0 and col.max() <= 1


def test_camera_data_matches_oracle(oracle):
    rng = np.random.default_rng(0)
    cams = [((0, 0, 20), (0, 0, 0), 640, 480)]
    for _ in range(500):
        cams.append((tuple(rng.uniform(-30, 30, 3)), (rng.uniform(-3, 3), rng.uniform(-3, 3), 0.0),
                     int(rng.integers(1, 8000)), int(rng.integers(1, 5000))))
    for f in (0, 37, 150, 299):
        p, r = configs.sequence_camera(f)
        cams.append((p, r, 3840, 2160))
    for p, r, w, h in cams:
        cam = host.Camera(w, h)
        cam.position, cam.rotation = p, r
        got = cam.get_data()
        want = oracle.camera_data(p, r, w, h)
        assert bytes(got) == bytes(want)


def test_render_rejects_bad_arguments(rclib):
    scene = abi.RcScene()
    fake = 0x1000
    cases = [
        (abi.RcScreen(None, 64, 64, 1024), "null surface"),
        (abi.RcScreen(fake, 0, 64, 1024), "zero width"),
        (abi.RcScreen(fake, 64, -1, 1024), "negative height"),
        (abi.RcScreen(fake, 64, 64, 1000), "pitch too small"),
        (abi.RcScreen(fake, 64, 64, 1032), "pitch not multiple of 16"),
        (abi.RcScreen(fake + 4, 64, 64, 1024), "misaligned surface"),
    ]
    for screen, what in cases:
        assert rclib.rc_render(C.byref(screen), C.byref(scene), None) == abi.RC_EINVAL, what
        assert rclib.rc_last_error_string() != b""
    ok = abi.RcScreen(fake, 64, 64, 1024)
    assert rclib.rc_render(None, C.byref(scene), None) == abi.RC_EINVAL
    assert rclib.rc_render(C.byref(ok), None, None) == abi.RC_EINVAL
    bad = abi.RcScene()
    bad.sphere_num = -1
    assert rclib.rc_render(C.byref(ok), C.byref(bad), None) == abi.RC_EINVAL
    bad.sphere_num = 3                       # arrays are null
    assert rclib.rc_render(C.byref(ok), C.byref(bad), None) == abi.RC_EINVAL
    for o in (abi.RcOpts(y_offset=-1), abi.RcOpts(y_offset=10, full_height=64), abi.RcOpts(variant=99),
              abi.RcOpts(rgba8=fake, rgba8_pitch=8)):
        assert rclib.rc_render(C.byref(ok), C.byref(scene), C.byref(o)) == abi.RC_EINVAL


def test_pixels_per_lane_query(rclib):
    """rc_pixels_per_lane answers on the host with the launcher's own rule: the measured crossovers of
    tools/p_sweep.py (profiles/r3_p_sweep_*.txt), forced variants, and bad arguments."""
    ask = lambda w, rows, n, mode=abi.MODE_REF, variant=abi.VARIANT_AUTO, cu=256, full=0: rclib.rc_pixels_per_lane(w, rows, full, n, mode, variant, cu)  # noqa: E731
    assert [ask(640, 480, 8), ask(1280, 768, 512), ask(1920, 1080, 64), ask(3840, 2160, 1024), ask(7680, 4320, 4096)] == [2, 2, 4, 8, 8]
    # 7-8 workgroups per CU at 8 pixels per lane: 8 only from 1024 spheres (a 7680x540 band is configuration 4's per-rank work)
    assert [ask(7680, 540, 4096), ask(7680, 540, 512), ask(2560, 1440, 1024), ask(2560, 1440, 1023), ask(7680, 270, 4096)] == [8, 4, 8, 4, 4]
    ext = lambda w, rows, n: ask(w, rows, n, mode=abi.MODE_EXT)  # noqa: E731
    assert [ext(1920, 1080, 64), ext(3840, 2160, 1024), ext(2560, 1440, 512), ext(2560, 1440, 768), ext(1920, 1080, 2048),
            ext(1920, 1080, 2049), ext(3840, 2160, 64)] == [2, 4, 2, 4, 2, 4, 4]
    assert ask(3840, 2160, 1024, variant=abi.VARIANT_CULLED) == 4 and ask(640, 480, 8, variant=abi.VARIANT_CULLED) == 2
    assert ask(64, 64, 1, variant=abi.VARIANT_LITERAL) == 1 and ask(64, 64, 1, variant=abi.VARIANT_TILED8) == 8
    assert ask(64, 64, 1, mode=abi.MODE_EXT, variant=abi.VARIANT_TILED8) == 4 and ask(64, 64, 1, variant=abi.VARIANT_TILED2) == 2
    assert ask(3840, 2160, 1024, cu=0) == ask(3840, 2160, 1024, cu=256)          # 0: the 256 CUs of an MI355X
    assert ask(3840, 2160, 1024, cu=1024) == 4 and ask(1280, 768, 1024, cu=64) == 8  # the rule scales with the CU count
    # a band of at most a quarter of the frame's rows takes 8 pixels per lane from 4 workgroups per CU (>= 1024 spheres)
    assert [ask(7680, 270, 4096, full=4320), ask(7680, 270, 1023, full=4320), ask(1920, 1080, 4096, full=4320),
            ask(2560, 800, 4096, full=1440), ask(7680, 270, 4096, full=270), ask(3840, 540, 1024, full=2160)] == [8, 4, 8, 4, 4, 8]
    for bad in [(0, 480, 0, 8), (640, 0, 0, 8), (640, 480, 0, -1), (640, 480, 479, 8)]:
        assert rclib.rc_pixels_per_lane(*bad, abi.MODE_REF, abi.VARIANT_AUTO, 256) == abi.RC_EINVAL
    assert ask(640, 480, 8, mode=7) == abi.RC_EINVAL and ask(640, 480, 8, variant=99) == abi.RC_EINVAL
    from cudaraytracing_amd import configs
    assert configs.pixels_per_lane(3840, 2160, 1024, True, False, 256) == 4


def test_write_ppm(rclib, tmp_path):
    img = np.arange(5 * 3 * 4, dtype=np.uint8).reshape(3, 5, 4)
    path = tmp_path / "t.ppm"
    assert rclib.rc_write_ppm(str(path).encode(), img.ctypes.data_as(C.c_void_p), 20, 5, 3) == 0
    data = path.read_bytes()
    assert data.startswith(b"P6\n5 3\n255\n")
    body = np.frombuffer(data[len(b"P6\n5 3\n255\n"):], np.uint8).reshape(3, 5, 3)
    assert np.array_equal(body, img[..., :3])
    assert rclib.rc_write_ppm(None, None, 0, 0, 0) == abi.RC_EINVAL


def test_band_rows_partition():
    for h in (1, 7, 540, 1080, 2160, 4320, 4321):
        for world in (1, 2, 3, 4, 8):
            spans = [configs.band_rows(h, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == h
            for a, b in zip(spans, spans[1:]):
                assert a[1] == b[0]
            sizes = [e - b for b, e in spans]
            assert max(sizes) - min(sizes) <= 1


def test_sequence_camera_is_float_exact_and_periodic():
    p0, r0 = configs.sequence_camera(0)
    assert p0 == (0.0, 0.0, 20.0) and r0 == (0.0, 0.0, 0.0)
    for f in range(0, 300, 17):
        p, r = configs.sequence_camera(f)
        assert all(float(np.float32(v)) == v for v in p + r)
        assert r[2] == 0.0 and p[2] == 20.0


def test_header_is_plain_c99(tmp_path):
    """include/rc_abi.h and the C example compile as strict C99 (the ABI is C, not C++) and link
    against the library; nothing is run here."""
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "render_frame")
    subprocess.run(["gcc", "-std=c99", "-pedantic", "-Wall", "-Wextra", "-Werror", "-I" + os.path.join(root, "include"),
                    os.path.join(root, "examples", "render_frame.c"), "-L" + os.path.dirname(abi.LIB_PATH), "-lrc_hip",
                    "-Wl,-rpath," + os.path.dirname(abi.LIB_PATH), "-o", exe], check=True, capture_output=True)
    assert os.path.exists(exe)


def test_gather_library_exports_every_declared_symbol():
    """librc_gather.so (row-band assembly over RCCL): loads without a GPU, exports what include/rc_gather.h
    declares, and its partition helpers agree with the Python ones the CPU sharding tests use."""
    from cudaraytracing_amd import gather, sharding
    text = open(os.path.join(ROOT, "include", "rc_gather.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    declared = set(re.findall(r"\b(rc_[a-z0-9_]+)\s*\(", text))
    assert declared == set(gather.SYMBOLS), declared ^ set(gather.SYMBOLS)
    gather.load()
    out = subprocess.run(["nm", "-D", "--defined-only", gather.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = {line.split()[-1] for line in out.splitlines() if line.strip()}
    assert declared <= exported, declared - exported
    assert "rccl" not in subprocess.run(["ldd", abi.LIB_PATH], capture_output=True, text=True).stdout   # only the gather library links RCCL
    for h in (1, 7, 540, 1080, 4320):
        for world in (1, 2, 3, 8):
            for r in range(world):
                b, e = configs.band_rows(h, world, r)
                assert gather.band_rows(h, world, r) == (b, e)
                for parts in (1, 2, 3, 5):
                    py = sharding.sub_bands(b, e, parts)
                    for k in range(parts):
                        sb, se = gather.sub_band(b, e, parts, k)
                        if k < len(py) and py[k][1] > py[k][0]:
                            assert (sb, se) == py[k]
                        else:
                            assert sb == se


def test_dropin_header_compiles_and_links_a_cpp_caller(tmp_path):
    """include/rc_dropin.hpp is what a C++ caller WITHOUT the reference's headers includes to call
    execute_kernel(Screen, Scene) by value (kernel.cu:114-115): it must compile on its own, declare the types
    with the reference's sizes and non-trivial copy, and link against the library's exported symbol."""
    src = tmp_path / "caller.cpp"
    src.write_text(
        '#include "rc_dropin.hpp"\n'
        '#include <type_traits>\n'
        'static_assert(!std::is_trivially_copyable<rc_ref_abi::Scene>::value, "Scene must be passed by invisible reference");\n'
        'static_assert(std::is_trivially_copyable<rc_ref_abi::Screen>::value, "Screen is a plain aggregate");\n'
        'int main(int argc, char **) {\n'
        '    rc_ref_abi::Screen screen = {nullptr, 0, 0, 0};\n'
        '    rc_ref_abi::Scene scene = {};\n'
        '    if (argc > 1000) execute_kernel(screen, scene);      // linked, never run here (no GPU)\n'
        '    return sizeof(scene) == 88 ? 0 : 1;\n'
        '}\n')
    exe = tmp_path / "caller"
    libdir = os.path.dirname(abi.LIB_PATH)
    subprocess.run(["g++", "-std=c++14", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"), str(src), "-o", str(exe),
                    "-L" + libdir, "-lrc_hip", "-Wl,-rpath," + libdir], check=True)
    syms = subprocess.run(["nm", "-D", "--undefined-only", str(exe)], capture_output=True, text=True, check=True).stdout
    assert "execute_kernel" in syms
