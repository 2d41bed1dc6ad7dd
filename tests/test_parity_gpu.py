"""GPU parity tests proper: the HIP path, called through the C ABI, against the CPU oracle.

Bar: bit-exact float surfaces (hence bit-exact hit masks and winner indices); bit-exact RGBA8.
Small frames are compared with the oracle run here on the same inputs; full-size frames with
the committed golden hashes (tests/golden/golden.json) and with size-independent properties
(tiled kernel == literal kernel, row bands == full frame).
"""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT
from cudaraytracing_amd import abi, configs
from cudaraytracing_amd.host import DeviceBuffer
from gpu_util import GpuScene, assert_same_bits, gpu_render, pixels

pytestmark = pytest.mark.gpu

VARIANTS = [abi.VARIANT_TILED, abi.VARIANT_TILED2, abi.VARIANT_TILED4, abi.VARIANT_TILED8, abi.VARIANT_CULLED,
            abi.VARIANT_LITERAL]


def _frame(oracle, name):
    c = configs.CONFIGS[name]
    pos, rad, col = oracle.generate_scene(c["n"], c["seed"])
    cam = oracle.camera_data(c["position"], c["rotation"], c["width"], c["height"])
    return c, pos, rad, col, cam


@pytest.mark.parametrize("variant", VARIANTS)
@pytest.mark.parametrize("name", ["c1_640x480_n8", "c2_1920x1080_n64", "ref_1280x768_n512"])
def test_small_frames_bit_exact(oracle, golden, name, variant):
    c, pos, rad, col, cam = _frame(oracle, name)
    w, h = c["width"], c["height"]
    want = oracle.render(pos, rad, col, cam, w, h)
    got = pixels(gpu_render(pos, rad, col, cam, w, h, variant=variant), w)
    assert_same_bits(got, want, name)
    assert f"{oracle.fnv_words(got):016x}" == golden[name]["fnv"]


@pytest.mark.parametrize("variant", VARIANTS)
def test_reference_known_answers_on_gpu(oracle, ref_kat, kat_scene, variant):
    """The reference's own recorded outputs, reproduced by the HIP path."""
    w, h = ref_kat["width"], ref_kat["height"]
    cam = oracle.camera_data(ref_kat["camera_position"], ref_kat["camera_rotation"], w, h)
    for v in ref_kat["vectors"]:
        pos, rad, col = (a[: v["n"]] for a in kat_scene)
        got = pixels(gpu_render(pos, rad, col, cam, w, h, variant=variant), w)
        assert f"{oracle.fnv_words(got, ref_kat['fnv_offset_basis']):016x}" == v["fnv"]


@pytest.mark.parametrize("variant", VARIANTS)
@pytest.mark.parametrize("w,h,n", [(333, 77, 5), (1, 1, 3), (65, 17, 1), (640, 480, 0), (31, 9, 7),
                                    (200, 120, 2049), (96, 64, 4100), (64, 48, 8200)])
def test_ragged_sizes_and_counts(oracle, w, h, n, variant):
    pos, rad, col = oracle.generate_scene(n, 3)
    cam = oracle.camera_data((0.5, -0.25, 21.0), (0.05, 0.1, 0.0), w, h)
    want = oracle.render(pos, rad, col, cam, w, h)
    got = pixels(gpu_render(pos, rad, col, cam, w, h, variant=variant), w)
    assert_same_bits(got, want, f"{w}x{h} n={n}")


@pytest.mark.parametrize("variant", VARIANTS)
def test_rotated_and_inside_cameras(oracle, variant):
    pos, rad, col = oracle.generate_scene(300, 9)
    w, h = 480, 270
    cams = [((0, 0, 20), (0.3, -0.7, 0)), ((3, 1, -2), (-0.4, 2.9, 0)), ((0, 0, -10), (0, 3.14159, 0)),
            (tuple(pos[17]), (0, 0, 0)),                               # camera at a sphere centre
            (tuple(pos[5] + np.float32([0, 0, 0.9]) * rad[5]), (0.2, 0.1, 0)),   # inside a sphere
            (tuple(pos[8] + np.float32([0, 0, 1.0]) * rad[8]), (0, 0, 0)),       # on a surface
            (tuple(pos[8] + np.float32([0, 0, 1.0 + 2e-6]) * rad[8]), (0, 0, 0))]  # grazing shell
    for p, r in cams:
        cam = oracle.camera_data(p, r, w, h)
        want = oracle.render(pos, rad, col, cam, w, h)
        got = pixels(gpu_render(pos, rad, col, cam, w, h, variant=variant), w)
        assert_same_bits(got, want, f"camera {p} {r}")


@pytest.mark.parametrize("variant", VARIANTS)
def test_ties_and_degenerate_spheres(oracle, variant):
    """Equal distances (duplicate spheres: lowest index wins, kernel.cu:91), zero/negative/
    NaN/Inf radii and positions, hits beyond the 10000 cut-off (kernel.cu:85)."""
    pos, rad, col = oracle.generate_scene(40, 5)
    pos[10] = pos[3]; rad[10] = rad[3]          # duplicate, different colour
    pos[11] = pos[3]; rad[11] = rad[3]
    rad[12] = 0.0
    rad[13] = -0.7                              # r*r is what matters
    rad[14] = np.nan
    rad[15] = np.inf
    pos[16] = [np.nan, 0, -10]
    pos[17] = [np.inf, 0, -10]
    pos[18] = [0, 0, -20000]; rad[18] = 5000    # nearest root > 10000: ignored
    pos[19] = [0, 0, -9990]; rad[19] = 30
    pos[20] = [1e30, 0, 0]; rad[20] = 1e30
    pos[21] = [0, 0, 1e20]; rad[21] = 1e20
    pos[22] = [1e-30, 1e-30, 20]; rad[22] = 1e-20   # tiny sphere at the camera
    w, h = 400, 300
    cam = oracle.camera_data((0, 0, 20), (0, 0, 0), w, h)
    want = oracle.render(pos, rad, col, cam, w, h)
    got = pixels(gpu_render(pos, rad, col, cam, w, h, variant=variant), w)
    assert_same_bits(got, want, "degenerate spheres")


@pytest.mark.parametrize("variant", VARIANTS)
def test_degenerate_camera_vectors(oracle, variant):
    """Non-unit / zero / NaN camera basis (CameraData is caller-supplied, Scene.h:12)."""
    pos, rad, col = oracle.generate_scene(50, 2)
    w, h = 160, 90
    base = oracle.camera_data((0, 0, 20), (0, 0, 0), w, h)
    for tweak in ("zero_forward", "huge", "nan_up", "tiny"):
        cam = oracle.Camera.from_buffer_copy(bytes(base))
        if tweak == "zero_forward":
            cam.forward.x = cam.forward.y = cam.forward.z = 0.0
            cam.right.x = 0.0; cam.up.y = 0.0
        elif tweak == "huge":
            cam.forward.z = -1e30
        elif tweak == "nan_up":
            cam.up.y = float("nan")
        elif tweak == "tiny":
            cam.forward.z = -1e-25; cam.right.x = 1e-25; cam.up.y = 1e-25
        want = oracle.render(pos, rad, col, cam, w, h)
        got = pixels(gpu_render(pos, rad, col, cam, w, h, variant=variant), w)
        assert_same_bits(got, want, tweak)


def test_random_scenes_and_cameras_stress(oracle):
    """Filter soundness, empirically: 40 random scenes with cameras near, inside and far from
    the sphere cloud, arbitrary rotations and radii over four orders of magnitude."""
    rng = np.random.default_rng(2024)
    w, h = 416, 234
    for it in range(40):
        n = int(rng.integers(1, 400))
        pos = rng.uniform(-12, 12, (n, 3)).astype(np.float32)
        rad = (10.0 ** rng.uniform(-3, 1, n)).astype(np.float32)
        col = rng.uniform(0, 1, (n, 3)).astype(np.float32)
        if it % 3 == 0:
            cpos = tuple(rng.uniform(-12, 12, 3))                    # inside the cloud
        elif it % 3 == 1:
            k = int(rng.integers(n))                                 # hugging one sphere
            dirv = rng.normal(size=3); dirv /= np.linalg.norm(dirv)
            cpos = tuple(pos[k] + dirv * rad[k] * rng.choice([0.5, 0.999999, 1.0, 1.000001, 1.00001, 1.01, 2.0]))
        else:
            cpos = tuple(rng.uniform(-60, 60, 3))
        rot = (float(rng.uniform(-3.2, 3.2)), float(rng.uniform(-3.2, 3.2)), 0.0)
        cam = oracle.camera_data(cpos, rot, w, h)
        want = oracle.render(pos, rad, col, cam, w, h)
        variant = [abi.VARIANT_TILED2, abi.VARIANT_TILED4, abi.VARIANT_TILED8][it % 3]
        got = pixels(gpu_render(pos, rad, col, cam, w, h, variant=variant), w)
        assert_same_bits(got, want, f"stress scene {it} (n={n}, cam={cpos}, rot={rot})")
        got = pixels(gpu_render(pos, rad, col, cam, w, h, variant=abi.VARIANT_CULLED), w)
        assert_same_bits(got, want, f"stress scene {it}, culled (n={n}, cam={cpos}, rot={rot})")


def test_filter_soak_tiled_vs_literal_on_device(oracle):
    """600 random scenes: every filtered variant against the unfiltered literal kernel, both on
    the GPU (the literal kernel is itself held to the oracle above).  Cameras are placed at
    random distances from a random sphere's surface, down to 1e-7 relative, to stress the
    hand-over between the filter and the exact path."""
    rng = np.random.default_rng(int(os.environ.get("RC_SOAK_SEED", "99")))
    w, h = 192, 104
    variants = [abi.VARIANT_TILED2, abi.VARIANT_TILED4, abi.VARIANT_TILED8, abi.VARIANT_CULLED]
    with_bins = os.environ.get("RC_SOAK_BINS") == "1"     # culled variant with records + bin lists in a workspace
    for it in range(int(os.environ.get("RC_SOAK_SCENES", "600"))):
        n = int(rng.integers(1, int(os.environ.get("RC_SOAK_NMAX", "120"))))
        spread = 10.0 ** rng.uniform(-1, 2)
        pos = rng.uniform(-spread, spread, (n, 3)).astype(np.float32)
        rad = (spread * 10.0 ** rng.uniform(-3, 0, n)).astype(np.float32)
        col = rng.uniform(0, 1, (n, 3)).astype(np.float32)
        k = int(rng.integers(n))
        dirv = rng.normal(size=3); dirv /= np.linalg.norm(dirv)
        off = 1.0 + rng.choice([-1.0, 1.0]) * 10.0 ** rng.uniform(-7, 1)
        cpos = tuple(float(v) for v in (pos[k] + dirv * rad[k] * off))
        if it % 3 != 2:      # look at the sphere (periods 3 and 5 here against the variants' 4: every combination comes up)
            f = -dirv
            rot = (float(np.arcsin(np.clip(f[1], -1, 1))), float(np.arctan2(-f[0], -f[2])), 0.0)
        else:
            rot = (float(rng.uniform(-3.2, 3.2)), float(rng.uniform(-3.2, 3.2)), 0.0)
        cam = oracle.camera_data(cpos, rot, w, h)
        if it % 5 == 4:
            # cameras the reference's Camera class cannot produce but its CameraData struct can carry:
            # other fields of view and aspects, basis vectors that are neither unit nor orthogonal
            from gpu_util import to_rc_camera
            cam = to_rc_camera(cam)
            cam.fov = float(10.0 ** rng.uniform(-2, 0.49))
            cam.aspect = float(10.0 ** rng.uniform(-0.7, 0.7))
            for v in (cam.forward, cam.up, cam.right):
                sc_, sh_ = float(10.0 ** rng.uniform(-0.3, 0.3)), rng.normal(size=3) * 0.05
                v.x, v.y, v.z = v.x * sc_ + float(sh_[0]), v.y * sc_ + float(sh_[1]), v.z * sc_ + float(sh_[2])
        scene = GpuScene(pos, rad, col)
        want = pixels(gpu_render(pos, rad, col, cam, w, h, variant=abi.VARIANT_LITERAL, scene=scene), w)
        if with_bins:
            got = pixels(gpu_render(pos, rad, col, cam, w, h, variant=abi.VARIANT_CULLED, scene=scene, workspace=True), w)
        else:
            # rounds of the four variants alternate the exact-test staging and the near-bound skip
            got = pixels(gpu_render(pos, rad, col, cam, w, h, variant=variants[it % 4], scene=scene,
                                    tuning=[0, abi.TUNE_NO_LDS_EXACT, abi.TUNE_NO_NEAR_SKIP, abi.TUNE_NO_LDS_EXACT | abi.TUNE_NO_NEAR_SKIP][(it // 4) % 4]), w)
        scene.free()
        assert_same_bits(got, want, f"soak {it}: n={n} cam={cpos} rot={rot} variant={variants[it % 4]}")
        if it % 5000 == 4999:
            print(f"soak: {it + 1} scenes identical", flush=True)      # progress for long runs (pytest -s)


@pytest.mark.parametrize("variant", [abi.VARIANT_TILED2, abi.VARIANT_TILED8, abi.VARIANT_CULLED])
def test_near_bound_skip_depth_ties(oracle, variant):
    """The exact test is skipped for pixels already nearer than a sphere's nearest point.  Scenes built to
    sit on that decision: every sphere's nearest point lies at (almost) the same distance D0 from the
    camera -- offsets from 0 to 1e-3 relative -- with radii of 1e-3 ... 0.3 D0, stacked along a few shared
    directions, so around each shared centre pixel the nearest hit so far and the next sphere's bound differ
    by rounding errors only; duplicated spheres add exact ties.  (A bound raised by 2^-16 |v| fails this test.)"""
    rng = np.random.default_rng(2024)
    w, h = 192, 128
    for it in range(12):
        n = int(rng.integers(40, 400))
        d0 = float(10.0 ** rng.uniform(0.0, 2.5))
        # a handful of directions, several spheres stacked along each: around the shared centre pixel their
        # hit distances differ by the offsets alone
        ang = rng.uniform(-0.3, 0.3, (max(n // 8, 1), 2))[rng.integers(0, max(n // 8, 1), n)]
        dirs = np.stack([np.sin(ang[:, 0]), np.sin(ang[:, 1]), -np.ones(n)], axis=1)
        dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
        rad = (d0 * 10.0 ** rng.uniform(-3, -0.5, n)).astype(np.float32)
        off = rng.choice([0.0, 1e-7, -1e-7, 3e-7, -3e-7, 1e-6, -1e-6, 3e-6, -3e-6, 1e-5, -1e-5, 1e-4, -1e-4, 1e-3], n) * d0
        pos = (dirs * (d0 + rad.astype(np.float64) + off)[:, None]).astype(np.float32)
        pos[1::7] = pos[0::7][: len(pos[1::7])]          # exact duplicates: the lowest index must win
        rad[1::7] = rad[0::7][: len(rad[1::7])]
        col = rng.uniform(0, 1, (n, 3)).astype(np.float32)
        cam = oracle.camera_data((0.0, 0.0, 0.0), (0.0, 0.0, 0.0), w, h)
        want = oracle.render(pos, rad, col, cam, w, h)
        got = pixels(gpu_render(pos, rad, col, cam, w, h, variant=variant, workspace=(variant == abi.VARIANT_CULLED and it % 2 == 0)), w)
        assert_same_bits(got, want, f"depth ties {it}: n={n} d0={d0}")
        assert int(want[..., :3].any(axis=2).sum()) > 1000


@pytest.mark.parametrize("scale", [1e-12, 1e-6, 1e-3, 1e3, 1e6, 1e12, 1e18])
def test_scene_scale_extremes(oracle, scale):
    """Coordinates from 1e-12 to 1e18: the filter's range checks must hand over to the exact
    path wherever its error bounds stop holding (the 1e-6 offset and the 10000 cut-off of
    kernel.cu:61,85 are not scale-free, so the images differ per scale; GPU == oracle always)."""
    pos, rad, col = oracle.generate_scene(120, 11)
    s = np.float32(scale)
    w, h = 320, 180
    cam = oracle.camera_data(tuple(np.float32([0.3, -0.2, 20.0]) * s), (0.02, 0.03, 0), w, h)
    want = oracle.render(pos * s, rad * s, col, cam, w, h)
    for variant in (abi.VARIANT_TILED4, abi.VARIANT_TILED8, abi.VARIANT_CULLED):
        got = pixels(gpu_render(pos * s, rad * s, col, cam, w, h, variant=variant), w)
        assert_same_bits(got, want, f"scale {scale}")


# ---- EXT mode (plane, point light, Phong, hard shadows): parity against this repository's own
# oracle definition; the reference has no such code, so these are "unpinned by the reference".

EXT_VARIANTS = [abi.VARIANT_TILED, abi.VARIANT_TILED2, abi.VARIANT_TILED4, abi.VARIANT_CULLED, abi.VARIANT_LITERAL]


def _default_ext(oracle, **kw):
    d = dict(configs.DEFAULT_EXT)
    d.update(kw)
    return oracle.Ext(**{k: (oracle.Vec3(*v) if isinstance(v, tuple) else v) for k, v in d.items()})


@pytest.mark.parametrize("variant", EXT_VARIANTS)
@pytest.mark.parametrize("name", ["c1_640x480_n8", "c2_1920x1080_n64"])
def test_ext_benchmark_frames_bit_exact(oracle, golden, name, variant):
    c, pos, rad, col, cam = _frame(oracle, name)
    w, h = c["width"], c["height"]
    ext = _default_ext(oracle)
    want = oracle.render(pos, rad, col, cam, w, h, ext=ext)
    got = pixels(gpu_render(pos, rad, col, cam, w, h, variant=variant, ext=ext), w)
    assert_same_bits(got, want, name + " EXT")
    assert f"{oracle.fnv_words(got):016x}" == golden[name + "_ext"]["fnv"]
    # the dominant-axis shadow filter (default only from 1024 spheres up) forced on
    got = pixels(gpu_render(pos, rad, col, cam, w, h, variant=variant, ext=ext, tuning=abi.TUNE_SHADOW_AXIS_ON), w)
    assert_same_bits(got, want, name + " EXT, dominant-axis shadow filter")


@pytest.mark.parametrize("variant", EXT_VARIANTS)
def test_ext_option_matrix_and_odd_lights(oracle, variant):
    pos, rad, col = oracle.generate_scene(150, 21)
    w, h = 400, 226
    cam = oracle.camera_data((1.0, 2.0, 21.0), (-0.1, 0.05, 0), w, h)
    cases = [dict(has_plane=0), dict(shadows=0), dict(has_plane=0, shadows=0), dict(shininess_log2=0),
             dict(shininess_log2=9, specular=1.5),
             dict(light_pos=tuple(float(v) for v in pos[3])),                  # light inside a sphere
             dict(light_pos=(0.0, -20.0, 0.0)),                                 # light below the floor
             dict(plane_normal=(0.0, -2.0, 0.0), plane_h=12.0),                 # non-unit, flipped normal
             dict(plane_normal=(0.3, 1.0, -0.2), plane_h=-4.0, plane_color=(1.0, 0.2, 0.1)),
             dict(plane_normal=(0.0, 0.0, 0.0)),                                # degenerate plane: n.d == 0
             dict(light_pos=(1e6, 1e6, 1e6)), dict(light_pos=(0.0, 0.0, 20.9))]
    for kw in cases:
        ext = _default_ext(oracle, **kw)
        want = oracle.render(pos, rad, col, cam, w, h, ext=ext)
        got = pixels(gpu_render(pos, rad, col, cam, w, h, variant=variant, ext=ext), w)
        assert_same_bits(got, want, f"EXT {kw}")
        got = pixels(gpu_render(pos, rad, col, cam, w, h, variant=variant, ext=ext, tuning=abi.TUNE_SHADOW_AXIS_ON), w)
        assert_same_bits(got, want, f"EXT {kw}, dominant-axis shadow filter")


def test_ext_random_scenes_stress(oracle):
    rng = np.random.default_rng(77)
    w, h = 352, 198
    for it in range(24):
        n = int(rng.integers(1, 300))
        pos = rng.uniform(-12, 12, (n, 3)).astype(np.float32)
        rad = (10.0 ** rng.uniform(-2, 0.7, n)).astype(np.float32)
        col = rng.uniform(0, 1, (n, 3)).astype(np.float32)
        cam = oracle.camera_data(tuple(rng.uniform(-25, 25, 3)), (float(rng.uniform(-3, 3)), float(rng.uniform(-3, 3)), 0.0), w, h)
        ext = _default_ext(oracle, light_pos=tuple(float(v) for v in rng.uniform(-20, 20, 3)),
                           plane_normal=tuple(float(v) for v in rng.normal(size=3)), plane_h=float(rng.uniform(-10, 10)))
        want = oracle.render(pos, rad, col, cam, w, h, ext=ext)
        variant = [abi.VARIANT_TILED2, abi.VARIANT_TILED4, abi.VARIANT_TILED][it % 3]
        tuning = [0, abi.TUNE_SHADOW_AXIS_ON, abi.TUNE_SHADOW_AXIS_ON | abi.TUNE_NO_LDS_EXACT, abi.TUNE_NO_LDS_EXACT][it % 4]
        got = pixels(gpu_render(pos, rad, col, cam, w, h, variant=variant, ext=ext, tuning=tuning), w)
        assert_same_bits(got, want, f"EXT stress {it} tuning {tuning}")


def test_ext_ragged_and_multichunk(oracle):
    for w, h, n in [(333, 77, 5), (65, 17, 0), (96, 64, 4100)]:
        pos, rad, col = oracle.generate_scene(n, 3)
        cam = oracle.camera_data((0.5, -0.25, 21.0), (0.05, 0.1, 0.0), w, h)
        ext = _default_ext(oracle)
        want = oracle.render(pos, rad, col, cam, w, h, ext=ext)
        for variant, tuning in ((abi.VARIANT_AUTO, 0), (abi.VARIANT_CULLED, 0), (abi.VARIANT_AUTO, abi.TUNE_SHADOW_AXIS_ON),
                                (abi.VARIANT_AUTO, abi.TUNE_SHADOW_AXIS_OFF)):
            got = pixels(gpu_render(pos, rad, col, cam, w, h, variant=variant, ext=ext, tuning=tuning), w)
            assert_same_bits(got, want, f"EXT {w}x{h} n={n} variant {variant} tuning {tuning}")


def test_ext_axis_aligned_planes_and_background_waves(oracle):
    """Planes whose normal has ONE non-zero component let the kernels decide per pixel, from the sign of the
    un-normalised ray's component along that axis, that the plane cannot be hit, and skip the ray normalisation and
    the plane test for waves (and the light records for workgroups) that see background only.  Every axis, both
    signs of the normal, zeros of either sign in the other components, planes in front of / behind / through the
    camera, cameras looking every way incl. straight along the plane, with few spheres (large background) and
    with none: bit-identical to the oracle on all EXT variants."""
    rng = np.random.default_rng(77)
    w, h = 208, 120
    normals = [(0.0, 1.0, 0.0), (0.0, -2.5, -0.0), (1.0, 0.0, 0.0), (-0.5, -0.0, 0.0), (0.0, 0.0, 3.0), (-0.0, 0.0, -1.0),
               (0.0, 1.0, 1e-30), (0.0, 0.0, 0.0)]
    it = 0
    for normal in normals:
        for plane_h in (-6.0, 4.0, 0.0):
            for n in (0, 5):
                pos, rad, col = oracle.generate_scene(max(n, 1), 30 + it)
                pos, rad, col = pos[:n], rad[:n], col[:n]
                cpos = (float(rng.uniform(-3, 3)), float(rng.uniform(-3, 3)), 20.0) if it % 3 else (0.0, 0.0, 20.0)
                rot = [(0.0, 0.0, 0.0), (0.4, -0.3, 0.0), (-1.2, 2.0, 0.0), (1.5707964, 0.0, 0.0), (0.0, 1.5707964, 0.0)][it % 5]
                cam = oracle.camera_data(cpos, rot, w, h)
                ext = _default_ext(oracle, plane_normal=normal, plane_h=plane_h, light_pos=(3.0, 9.0, 14.0))
                want = oracle.render(pos, rad, col, cam, w, h, ext=ext)
                for variant in EXT_VARIANTS:
                    got = pixels(gpu_render(pos, rad, col, cam, w, h, variant=variant, ext=ext), w)
                    assert_same_bits(got, want, f"axis plane {normal} h={plane_h} n={n} cam={cpos} rot={rot} variant={variant}")
                it += 1


def test_ext_soak_tiled_vs_literal_on_device(oracle):
    """EXT mode, filter soundness incl. the shadow-ray filter: random scenes, lights and planes with
    the camera and the light near sphere surfaces; tiled / culled kernels against the literal EXT
    kernel on the device (RC_EXT_SOAK_SCENES scenes of up to RC_EXT_SOAK_NMAX spheres, RC_SOAK_SEED)."""
    rng = np.random.default_rng(int(os.environ.get("RC_SOAK_SEED", "99")) + 1)
    w, h = 160, 88
    variants = [abi.VARIANT_TILED2, abi.VARIANT_TILED4, abi.VARIANT_CULLED]
    for it in range(int(os.environ.get("RC_EXT_SOAK_SCENES", "150"))):
        n = int(rng.integers(1, int(os.environ.get("RC_EXT_SOAK_NMAX", "90"))))    # > 1024: several LDS passes, any-hit exit
        spread = 10.0 ** rng.uniform(-0.5, 1.5)
        pos = rng.uniform(-spread, spread, (n, 3)).astype(np.float32)
        rad = (spread * 10.0 ** rng.uniform(-2.5, -0.3, n)).astype(np.float32)
        col = rng.uniform(0, 1, (n, 3)).astype(np.float32)
        k, kl = int(rng.integers(n)), int(rng.integers(n))
        dirv = rng.normal(size=3); dirv /= np.linalg.norm(dirv)
        dirl = rng.normal(size=3); dirl /= np.linalg.norm(dirl)
        cpos = tuple(float(v) for v in (pos[k] + dirv * rad[k] * (1.0 + 10.0 ** rng.uniform(-6, 1))))
        # the light from 1e-6 radii off a sphere's surface (or inside it) to far away
        lpos = tuple(float(v) for v in (pos[kl] + dirl * rad[kl] * (1.0 + rng.choice([-1.0, 1.0]) * 10.0 ** rng.uniform(-6, 1.5))))
        f = -dirv
        rot = (float(np.arcsin(np.clip(f[1], -1, 1))) + float(rng.normal() * 0.2), float(np.arctan2(-f[0], -f[2])) + float(rng.normal() * 0.2), 0.0)
        cam = oracle.camera_data(cpos, rot, w, h)
        if it % 5 == 4:                              # basis vectors neither unit nor orthogonal, other fields of view
            from gpu_util import to_rc_camera
            cam = to_rc_camera(cam)
            cam.fov = float(10.0 ** rng.uniform(-2, 0.49))
            cam.aspect = float(10.0 ** rng.uniform(-0.7, 0.7))
            for v in (cam.forward, cam.up, cam.right):
                sc_, sh_ = float(10.0 ** rng.uniform(-0.3, 0.3)), rng.normal(size=3) * 0.05
                v.x, v.y, v.z = v.x * sc_ + float(sh_[0]), v.y * sc_ + float(sh_[1]), v.z * sc_ + float(sh_[2])
        normal = rng.normal(size=3)
        if it % 4 == 3:                              # axis-aligned planes: the kernels' "can this pixel reach the plane" shortcut
            # (period 4 against the variants' period 3 and the filter forms' period 6: every combination comes up)
            keep = int(rng.integers(3))
            normal = np.array([normal[k] if k == keep else 0.0 * rng.choice([-1.0, 1.0]) for k in range(3)])
        ext = _default_ext(oracle, light_pos=lpos, plane_normal=tuple(float(v) for v in normal),
                           plane_h=float(rng.uniform(-spread, spread)), has_plane=int(rng.integers(2)))
        scene = GpuScene(pos, rad, col)
        want = pixels(gpu_render(pos, rad, col, cam, w, h, variant=abi.VARIANT_LITERAL, scene=scene, ext=ext), w)
        tuning = abi.TUNE_SHADOW_AXIS_ON if (it // 3) % 2 == 0 else 0     # both forms of the shadow filter
        got = pixels(gpu_render(pos, rad, col, cam, w, h, variant=variants[it % 3], scene=scene, ext=ext, tuning=tuning), w)
        scene.free()
        assert_same_bits(got, want, f"EXT soak {it}: n={n} cam={cpos} light={lpos} variant={variants[it % 3]} tuning={tuning}")
        if it % 5000 == 4999:
            print(f"EXT soak: {it + 1} scenes identical", flush=True)


def test_tuning_bits_are_validated(rclib):
    surf = DeviceBuffer(64 * 64 * 16)               # a real surface: a library that let the call through would only draw
    screen = abi.RcScreen(surf.ptr.value, 64, 64, 1024)
    scene = abi.RcScene()
    assert rclib.rc_render(C.byref(screen), C.byref(scene), C.byref(abi.RcOpts(tuning=16))) == abi.RC_EINVAL
    assert b"tuning" in rclib.rc_last_error_string()
    abi.check(rclib.rc_sync(None))
    surf.free()


def test_ext_requires_parameters(rclib):
    surf = DeviceBuffer(64 * 64 * 16)               # rejected calls never touch it; a real surface all the same
    screen = abi.RcScreen(surf.ptr.value, 64, 64, 1024)
    scene = abi.RcScene()
    assert rclib.rc_render(C.byref(screen), C.byref(scene), C.byref(abi.RcOpts(mode=abi.MODE_EXT))) == abi.RC_EINVAL
    assert rclib.rc_render(C.byref(screen), C.byref(scene), C.byref(abi.RcOpts(mode=7))) == abi.RC_EINVAL
    abi.check(rclib.rc_sync(None))
    surf.free()


def test_pitch_padding_untouched(oracle):
    pos, rad, col = oracle.generate_scene(16, 1)
    w, h = 100, 40
    cam = oracle.camera_data((0, 0, 20), (0, 0, 0), w, h)
    pitch = 2048
    raw = gpu_render(pos, rad, col, cam, w, h, pitch=pitch, fill=1)
    want = oracle.render(pos, rad, col, cam, w, h)
    assert_same_bits(pixels(raw, w), want, "pitched surface")
    pad = raw[:, w * 4:].view(np.uint32)
    assert np.all(pad == 0x01010101)            # the memset pattern of RayTracingKernel.cpp:94


@pytest.mark.parametrize("variant", [abi.VARIANT_TILED, abi.VARIANT_LITERAL])
def test_row_bands_equal_full_frame(oracle, variant):
    c, pos, rad, col, cam = _frame(oracle, "c2_1920x1080_n64")
    w, h = c["width"], c["height"]
    scene = GpuScene(pos, rad, col)
    full = pixels(gpu_render(pos, rad, col, cam, w, h, variant=variant, scene=scene), w)
    for world in (2, 8, 7):
        parts = []
        for rank in range(world):
            b, e = configs.band_rows(h, world, rank)
            parts.append(pixels(gpu_render(pos, rad, col, cam, w, e - b, variant=variant, y_offset=b,
                                           full_height=h, scene=scene), w))
        assert_same_bits(np.concatenate(parts, axis=0), full, f"{world} bands")
    scene.free()


def test_rgba8_fused_and_standalone(oracle):
    c, pos, rad, col, cam = _frame(oracle, "c1_640x480_n8")
    w, h = c["width"], c["height"]
    for flip in (False, True):
        raw, fused = gpu_render(pos, rad, col, cam, w, h, want_rgba8=True, flip=flip)
        want8 = oracle.to_rgba8(pixels(raw, w), flip=flip)
        assert np.array_equal(fused, want8)
    # standalone conversion kernel on a surface with NaN / out-of-range values
    lib = abi.load()
    from cudaraytracing_amd.host import DeviceBuffer
    rng = np.random.default_rng(0)
    surf = rng.uniform(-0.5, 1.5, (37, 53, 4)).astype(np.float32)
    surf[3, 4, 0] = np.nan; surf[5, 6, 1] = np.inf; surf[7, 8, 2] = -np.inf
    d_in, d_out = DeviceBuffer(surf.nbytes), DeviceBuffer(37 * 53 * 4)
    d_in.upload(surf)
    screen = abi.RcScreen(d_in.ptr.value, 53, 37, 53 * 16)
    abi.check(lib.rc_to_rgba8(C.byref(screen), d_out.ptr, 53 * 4, 1, None))
    abi.check(lib.rc_sync(None))
    got = np.empty((37, 53, 4), np.uint8)
    abi.check(lib.rc_download_2d(got.ctypes.data_as(C.c_void_p), 53 * 4, d_out.ptr, 53 * 4, 53 * 4, 37, None))
    assert np.array_equal(got, oracle.to_rgba8(surf, flip=True))
    d_in.free(); d_out.free()


@pytest.mark.parametrize("name", ["c3_3840x2160_n1024", "c4_7680x4320_n4096",
                                  "c5_seq_f0", "c5_seq_f37", "c5_seq_f150", "c5_seq_f299"])
def test_full_size_golden_hashes(oracle, golden, name):
    """Benchmark-size frames against the oracle's committed hashes; tiled == literal as well."""
    if name not in golden:
        pytest.fail(f"{name} missing from tests/golden/golden.json (run make_golden.py --full)")
    fx = golden[name]
    w, h = fx["width"], fx["height"]
    pos, rad, col = oracle.generate_scene(fx["n"], fx["seed"])
    cam = oracle.camera_data(fx["position"], fx["rotation"], w, h)
    got_cam = np.frombuffer(C.string_at(C.addressof(cam), C.sizeof(cam)), np.uint32).tolist()
    assert got_cam == fx["camera_bits"]
    scene = GpuScene(pos, rad, col)
    raw, rgba8 = gpu_render(pos, rad, col, cam, w, h, variant=abi.VARIANT_TILED, want_rgba8=True, flip=True, scene=scene)
    got = pixels(raw, w)
    assert f"{oracle.fnv_words(got):016x}" == fx["fnv"]
    assert int(got[..., :3].any(axis=2).sum()) == fx["lit_pixels"]
    bits = got.view(np.uint32)
    for s in fx["samples"]:
        assert bits[s[1], s[0]].tolist() == s[2:6]
    flat = np.ascontiguousarray(rgba8).view(np.uint32).reshape(1, -1)
    scr = oracle.Screen(flat.ctypes.data, flat.shape[1] // 4, 1, flat.shape[1] * 4)
    assert f"{int(oracle.load().rco_fnv1a64_words(C.byref(scr), oracle.FNV_BASIS)):016x}" == fx["rgba8_flip_fnv"]
    culled = pixels(gpu_render(pos, rad, col, cam, w, h, variant=abi.VARIANT_CULLED, scene=scene), w)
    assert_same_bits(culled, got, "culled vs tiled")
    if name.startswith("c3") or name.startswith("c4"):
        lit = pixels(gpu_render(pos, rad, col, cam, w, h, variant=abi.VARIANT_LITERAL, scene=scene), w)
        assert_same_bits(lit, got, "literal vs tiled")
    scene.free()


@pytest.mark.parametrize("name", ["c3_3840x2160_n1024_ext", "c5_seq_ext_f37", "c5_seq_ext_f200"])
def test_full_size_ext_golden_hashes(oracle, golden, name):
    """EXT (plane, point light, Phong, hard shadows) at 3840x2160 x 1024 spheres, and two frames of the
    animated sequence (moving camera AND orbiting light), against the oracle's committed hashes.
    EXT is defined by the oracle; the reference has no such code (unpinned by the reference)."""
    if name not in golden:
        pytest.fail(f"{name} missing from tests/golden/golden.json (run make_golden.py --ext-full)")
    fx = golden[name]
    w, h = fx["width"], fx["height"]
    pos, rad, col = oracle.generate_scene(fx["n"], fx["seed"])
    cam = oracle.camera_data(fx["position"], fx["rotation"], w, h)
    assert np.frombuffer(C.string_at(C.addressof(cam), C.sizeof(cam)), np.uint32).tolist() == fx["camera_bits"]
    e = fx["ext"]
    ext = oracle.Ext(oracle.Vec3(*e["plane_normal"]), e["plane_h"], oracle.Vec3(*e["plane_color"]), oracle.Vec3(*e["light_pos"]),
                     e["specular"], e["shininess_log2"], e["has_plane"], e["shadows"])
    scene = GpuScene(pos, rad, col)
    raw, rgba8 = gpu_render(pos, rad, col, cam, w, h, variant=abi.VARIANT_TILED, want_rgba8=True, flip=True, scene=scene, ext=ext)
    got = pixels(raw, w)
    assert f"{oracle.fnv_words(got):016x}" == fx["fnv"]
    assert int(got[..., :3].any(axis=2).sum()) == fx["lit_pixels"]
    bits = got.view(np.uint32)
    for s in fx["samples"]:
        assert bits[s[1], s[0]].tolist() == s[2:6]
    flat = np.ascontiguousarray(rgba8).view(np.uint32).reshape(1, -1)
    scr = oracle.Screen(flat.ctypes.data, flat.shape[1] // 4, 1, flat.shape[1] * 4)
    assert f"{int(oracle.load().rco_fnv1a64_words(C.byref(scr), oracle.FNV_BASIS)):016x}" == fx["rgba8_flip_fnv"]
    culled = pixels(gpu_render(pos, rad, col, cam, w, h, variant=abi.VARIANT_CULLED, scene=scene, ext=ext), w)
    assert_same_bits(culled, got, "EXT culled vs tiled")
    if name.startswith("c3"):
        lit = pixels(gpu_render(pos, rad, col, cam, w, h, variant=abi.VARIANT_LITERAL, scene=scene, ext=ext), w)
        assert_same_bits(lit, got, "EXT literal vs tiled")
    scene.free()


def test_dropin_execute_kernel_from_reference_call_site(oracle, tmp_path):
    """oracle/_ref/dropin_caller was compiled against the reference's own Screen.h/Scene.h/
    Camera.h and calls execute_kernel(Screen, Scene) by value, as RayTracingKernel.cpp:17."""
    exe = os.path.join(ROOT, "oracle", "_ref", "dropin_caller")
    if not os.path.exists(exe):
        pytest.fail("oracle/_ref/dropin_caller missing: build it in the container with `make -C oracle`")
    w, h, n, seed = 1280, 768, 512, 0      # the reference's own window and sphere count
    out = tmp_path / "dropin.raw"
    r = subprocess.run([exe, str(w), str(h), str(n), str(seed), str(out)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    got = np.fromfile(out, np.float32).reshape(h, w, 4)
    pos, rad, col = oracle.generate_scene(n, seed)
    cam = oracle.camera_data((0, 0, 20), (0, 0, 0), w, h)
    assert_same_bits(got, oracle.render(pos, rad, col, cam, w, h), "drop-in execute_kernel")


def test_cli_writes_reference_exact_ppm(oracle, tmp_path):
    """The C++ host layer (Camera / SceneManager / RayTracingKernel mirrors + rc_render CLI)."""
    exe = os.path.join(ROOT, "cudaraytracing_amd", "rc_render")
    if not os.path.exists(exe):
        pytest.fail("cudaraytracing_amd/rc_render missing: run `python -m cudaraytracing_amd.build`")
    w, h, n = 320, 200, 77
    out = tmp_path / "f.ppm"
    r = subprocess.run([exe, "--width", str(w), "--height", str(h), "--spheres", str(n), "--seed", "4",
                        "--pos", "1,0.5,19", "--rot", "0.1,-0.2", "--out", str(out)],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "frame:" in r.stdout
    data = out.read_bytes()
    head = f"P6\n{w} {h}\n255\n".encode()
    assert data.startswith(head)
    img = np.frombuffer(data[len(head):], np.uint8).reshape(h, w, 3)
    pos, rad, col = oracle.generate_scene(n, 4)
    cam = oracle.camera_data((1, 0.5, 19), (0.1, -0.2, 0), w, h)
    want = oracle.to_rgba8(oracle.render(pos, rad, col, cam, w, h), flip=True)
    assert np.array_equal(img, want[..., :3])


def test_frame_pipeline_matches_frame_by_frame(oracle, tmp_path):
    """Double-buffered render/read-back overlap (FramePipeline, through the CLI): every frame of
    the scripted camera path equals the oracle's frame for that camera."""
    exe = os.path.join(ROOT, "cudaraytracing_amd", "rc_render")
    w, h, n, frames = 320, 180, 60, 7
    pat = str(tmp_path / "f_%03d.ppm")
    r = subprocess.run([exe, "--width", str(w), "--height", str(h), "--spheres", str(n), "--frames", str(frames),
                        "--out-pattern", pat], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    pos, rad, col = oracle.generate_scene(n, 0)
    head = f"P6\n{w} {h}\n255\n".encode()
    for f in range(frames):
        data = open(pat % f, "rb").read()
        img = np.frombuffer(data[len(head):], np.uint8).reshape(h, w, 3)
        p, rot = configs.sequence_camera(f)
        cam = oracle.camera_data(p, rot, w, h)
        want = oracle.to_rgba8(oracle.render(pos, rad, col, cam, w, h), flip=True)
        assert np.array_equal(img, want[..., :3]), f"frame {f}"


def test_concurrent_calls_from_threads_and_streams(oracle):
    """rc_render keeps no per-frame state: four host threads, each with its own stream, scene,
    camera and frame size, render at the same time; every frame must still match the oracle."""
    import threading
    lib = abi.load()
    jobs = []
    for t in range(4):
        w, h, n = 300 + 37 * t, 170 + 11 * t, 60 + 50 * t
        pos, rad, col = oracle.generate_scene(n, 100 + t)
        cam = oracle.camera_data((0.3 * t, -0.2 * t, 20.0 + t), (0.02 * t, -0.03 * t, 0), w, h)
        jobs.append(dict(w=w, h=h, pos=pos, rad=rad, col=col, cam=cam, out=None, err=None))

    def work(job):
        try:
            from cudaraytracing_amd.host import DeviceBuffer
            stream = C.c_void_p()
            abi.check(lib.rc_stream_create(C.byref(stream)))
            scene = GpuScene(job["pos"], job["rad"], job["col"])
            w, h = job["w"], job["h"]
            surf = DeviceBuffer(w * 16 * h)
            screen = abi.RcScreen(surf.ptr.value, w, h, w * 16)
            sc = scene.scene(job["cam"])
            opts = abi.RcOpts(stream=stream)
            for _ in range(25):
                abi.check(lib.rc_render(C.byref(screen), C.byref(sc), C.byref(opts)), "rc_render")
            abi.check(lib.rc_sync(stream))
            out = np.empty((h, w, 4), np.float32)
            abi.check(lib.rc_download_2d(out.ctypes.data_as(C.c_void_p), w * 16, surf.ptr, w * 16, w * 16, h, stream))
            job["out"] = out
            surf.free(); scene.free()
            abi.check(lib.rc_stream_destroy(stream))
        except Exception as e:           # surfaced in the main thread
            job["err"] = e

    threads = [threading.Thread(target=work, args=(j,)) for j in jobs]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    for j in jobs:
        assert j["err"] is None, j["err"]
        want = oracle.render(j["pos"], j["rad"], j["col"], j["cam"], j["w"], j["h"])
        assert_same_bits(j["out"], want, f"threaded frame {j['w']}x{j['h']}")


def test_bench_contract_line():
    """bench.py prints ONE JSON line with the fields the driver reads."""
    import json, sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "5", "--warmup", "2",
                        "--config", "c2_1920x1080_n64"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 5 and d["warmup"] == 2 and d["higher_is_better"] is True
    assert d["unit"] == "Mrays/s" and d["value"] > 0 and d["vs_baseline"] is None
    assert d["config"]["workload"] == "c2_1920x1080_n64"
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in d["roofline"], k
    # a roofline fraction prices work the kernel executes: it cannot exceed 1
    assert 0.0 < d["roofline"]["frac"] <= 1.0
    assert abs(d["roofline"]["frac"] - d["roofline"]["achieved"] / d["roofline"]["peak"]) < 1e-9
    for k in ("value", "unit", "cores", "kind", "sample", "box_threads"):
        assert k in d["cpu_baseline"], k
    assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["cores"] >= 1
    assert d["cpu_baseline"]["box_threads"] >= d["cpu_baseline"]["cores"] or d["cpu_baseline"]["cores"] == d["cpu_baseline"]["job_cpu_share"]


def test_prepare_then_render(oracle, rclib):
    w, h, n = 250, 130, 20
    cam = oracle.camera_data((0, 0, 20), (0, 0, 0), w, h)
    assert rclib.rc_prepare(w, h, cam.fov) == 0
    assert rclib.rc_prepare(0, h, cam.fov) == abi.RC_EINVAL
    pos, rad, col = oracle.generate_scene(n, 8)
    got = pixels(gpu_render(pos, rad, col, cam, w, h), w)
    assert_same_bits(got, oracle.render(pos, rad, col, cam, w, h), "after rc_prepare")


def test_argument_validation_on_device(rclib):
    screen = abi.RcScreen(None, 64, 64, 1024)
    scene = abi.RcScene()
    assert rclib.rc_render(C.byref(screen), C.byref(scene), None) == abi.RC_EINVAL
    assert b"surface" in rclib.rc_last_error_string()


def test_sine_table_cache_eviction(oracle, rclib):
    """More frame formats than the per-device table cache holds (32): old tables are evicted and
    rebuilt on demand; the first and the last formats still render bit-exactly afterwards."""
    n = 12
    pos, rad, col = oracle.generate_scene(n, 21)
    scene = GpuScene(pos, rad, col)
    formats = [(40 + 3 * i, 24 + i) for i in range(40)]
    for rnd in range(2):
        for i, (w, h) in enumerate(formats):
            cam = oracle.camera_data((0, 0, 20), (0, 0, 0), w, h)
            got = pixels(gpu_render(None, None, None, cam, w, h, scene=scene), w)
            if i in (0, 1, 38, 39):
                assert_same_bits(got, oracle.render(pos, rad, col, cam, w, h), f"round {rnd} format {w}x{h}")
    scene.free()


def test_frame_size_limit(rclib):
    scene = abi.RcScene()
    surf = DeviceBuffer(65537 * 16 * 4)
    screen = abi.RcScreen(surf.ptr.value, 65537, 4, 65537 * 16)
    assert rclib.rc_render(C.byref(screen), C.byref(scene), None) == abi.RC_EINVAL
    assert b"65536" in rclib.rc_last_error_string()
    abi.check(rclib.rc_sync(None))
    surf.free()
    assert rclib.rc_prepare(70000, 4, 0.5) == abi.RC_EINVAL


def test_cli_input_script_replay(oracle, tmp_path):
    """rc_render --script: the reference's interactive controls (main.cpp:53-128) replayed from a
    file; the last frame equals the oracle's frame for the camera the same polls produce."""
    from cudaraytracing_amd.host import Camera
    exe = os.path.join(ROOT, "cudaraytracing_amd", "rc_render")
    w, h, n = 320, 192, 48
    polls = [(0.0004, "w", 0, 0), (0.0003, "d+shift", 0, 0), (0.01, "r", 2, -1), (0.0002, "s+q", 0, 0),
             (0.004, "r+a", -3, 2), (0.0005, "-", 5, 5), (0.0002, "z+w", 0, 0)]
    script = tmp_path / "input.txt"
    script.write_text("# dt_ms keys dx dy\n" + "".join(f"{dt!r} {k} {dx} {dy}\n" for dt, k, dx, dy in polls))
    out = tmp_path / "last.ppm"
    r = subprocess.run([exe, "--width", str(w), "--height", str(h), "--spheres", str(n), "--script", str(script),
                        "--out", str(out)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.count("frame: ") == len(polls)
    cam = Camera(w, h)
    cam.position = (0.0, 0.0, 20.0)
    for dt, k, dx, dy in polls:
        cam.handle_input(np.float32(dt).item(), "" if k == "-" else k.replace("+", " "), (dx, dy))
    assert cam.position != (0.0, 0.0, 20.0) and cam.rotation != (0.0, 0.0, 0.0)
    pos, rad, col = oracle.generate_scene(n, 0)
    ocam = oracle.camera_data(cam.position, cam.rotation, w, h)
    want = oracle.to_rgba8(oracle.render(pos, rad, col, ocam, w, h), flip=True)
    head = f"P6\n{w} {h}\n255\n".encode()
    data = out.read_bytes()
    img = np.frombuffer(data[len(head):], np.uint8).reshape(h, w, 3)
    assert np.array_equal(img, want[..., :3])
    assert int((img != 0).any(axis=-1).sum()) > 500      # the scripted camera still sees the scene


def test_graph_capture_and_replay(oracle, rclib):
    """A 10-frame camera sequence captured into one HIP graph (rc_graph_begin/end) and replayed
    twice: every frame of every replay equals the oracle's frame for that camera; a frame format
    that was not prepared is refused inside a capture instead of allocating."""
    from cudaraytracing_amd.host import DeviceBuffer
    w, h, n, frames = 200, 112, 40, 10
    pos, rad, col = oracle.generate_scene(n, 5)
    gs = GpuScene(pos, rad, col)
    stream = C.c_void_p()
    abi.check(rclib.rc_stream_create(C.byref(stream)))
    cams = [oracle.camera_data(*configs.sequence_camera(7 * f), w, h) for f in range(frames)]
    abi.check(rclib.rc_prepare(w, h, cams[0].fov))
    surfs = [DeviceBuffer(w * 16 * h) for _ in range(frames)]
    opts = abi.RcOpts(stream=stream)
    abi.check(rclib.rc_graph_begin(stream), "rc_graph_begin")
    for f in range(frames):
        screen = abi.RcScreen(surfs[f].ptr.value, w, h, w * 16)
        sc = gs.scene(cams[f])
        abi.check(rclib.rc_render(C.byref(screen), C.byref(sc), C.byref(opts)), "rc_render in capture")
    # an unprepared format inside the capture is rejected up front
    odd = abi.RcScreen(surfs[0].ptr.value, w - 8, h - 8, w * 16)
    sc = gs.scene(oracle.camera_data((0, 0, 20), (0, 0, 0), w - 8, h - 8))
    assert rclib.rc_render(C.byref(odd), C.byref(sc), C.byref(opts)) == abi.RC_EINVAL
    assert b"rc_prepare" in rclib.rc_last_error_string()
    exe = C.c_void_p()
    abi.check(rclib.rc_graph_end(stream, C.byref(exe)), "rc_graph_end")
    want = [oracle.render(pos, rad, col, cams[f], w, h) for f in range(frames)]
    for replay in range(2):
        for s in surfs:
            abi.check(rclib.rc_memset(s.ptr, 0x55, w * 16 * h, stream))
        abi.check(rclib.rc_graph_launch(exe, stream), "rc_graph_launch")
        abi.check(rclib.rc_sync(stream))
        for f in range(frames):
            got = np.empty((h, w, 4), np.float32)
            abi.check(rclib.rc_download_2d(got.ctypes.data_as(C.c_void_p), w * 16, surfs[f].ptr, w * 16, w * 16, h, stream))
            assert_same_bits(got, want[f], f"replay {replay} frame {f}")
    abi.check(rclib.rc_graph_destroy(exe))
    assert rclib.rc_graph_begin(None) == abi.RC_EINVAL
    for s in surfs:
        s.free()
    gs.free()
    abi.check(rclib.rc_stream_destroy(stream))


def test_graph_tables_survive_cache_eviction(oracle, rclib):
    """A captured graph keeps the address of its frame format's sine tables.  Rendering more other
    formats than the table cache holds (32) afterwards must not evict (free) those tables: the
    replay still produces the oracle's frame."""
    w, h, n = 168, 96, 30
    pos, rad, col = oracle.generate_scene(n, 9)
    gs = GpuScene(pos, rad, col)
    cam = oracle.camera_data((0, 0, 20), (0.02, 0.04, 0), w, h)
    stream = C.c_void_p()
    abi.check(rclib.rc_stream_create(C.byref(stream)))
    abi.check(rclib.rc_prepare(w, h, cam.fov))
    surf = DeviceBuffer(w * 16 * h)
    screen = abi.RcScreen(surf.ptr.value, w, h, w * 16)
    opts = abi.RcOpts(stream=stream)
    sc = gs.scene(cam)
    abi.check(rclib.rc_graph_begin(stream), "rc_graph_begin")
    abi.check(rclib.rc_render(C.byref(screen), C.byref(sc), C.byref(opts)), "rc_render in capture")
    exe = C.c_void_p()
    abi.check(rclib.rc_graph_end(stream, C.byref(exe)), "rc_graph_end")
    # 40 other formats through the cache; each one allocates (and, from the 33rd on, evicts)
    for i in range(40):
        ow, oh = 48 + 2 * i, 20 + i
        ocam = oracle.camera_data((0, 0, 20), (0, 0, 0), ow, oh)
        got = pixels(gpu_render(None, None, None, ocam, ow, oh, scene=gs), ow)
        if i in (0, 39):
            assert_same_bits(got, oracle.render(pos, rad, col, ocam, ow, oh), f"format {ow}x{oh}")
    abi.check(rclib.rc_memset(surf.ptr, 0x33, w * 16 * h, stream))
    abi.check(rclib.rc_graph_launch(exe, stream), "rc_graph_launch")
    abi.check(rclib.rc_sync(stream))
    got = np.empty((h, w, 4), np.float32)
    abi.check(rclib.rc_download_2d(got.ctypes.data_as(C.c_void_p), w * 16, surf.ptr, w * 16, w * 16, h, stream))
    assert_same_bits(got, oracle.render(pos, rad, col, cam, w, h), "graph replay after 40 other formats")
    abi.check(rclib.rc_graph_destroy(exe))
    surf.free()
    gs.free()
    abi.check(rclib.rc_stream_destroy(stream))


def test_context_api_end_to_end(oracle, rclib, tmp_path):
    """rc_ctx_*: the reference's RayTracingKernel + SceneManager life cycle through plain C calls --
    create, random scene, surface, render, read back; then a caller scene, a camera move, one
    scripted input poll, RGBA8 and PPM output.  Every frame equals the oracle's."""
    w, h, n = 300, 170, 70
    ctx = C.c_void_p()
    abi.check(rclib.rc_ctx_create(C.byref(ctx), 0, w, h), "rc_ctx_create")
    assert rclib.rc_ctx_render(ctx, None) == abi.RC_EINVAL            # no surface yet
    assert b"rc_surface_alloc" in rclib.rc_last_error_string()
    abi.check(rclib.rc_scene_generate(ctx, n, 3), "rc_scene_generate")
    pitch = C.c_size_t()
    abi.check(rclib.rc_surface_alloc(ctx, C.byref(pitch)), "rc_surface_alloc")
    assert pitch.value >= w * 16 and pitch.value % 256 == 0
    assert rclib.rc_ctx_screen(ctx).contents.width == w and rclib.rc_ctx_scene(ctx).contents.sphere_num == n
    abi.check(rclib.rc_ctx_render(ctx, None), "rc_ctx_render")
    got = np.empty((h, w, 4), np.float32)
    abi.check(rclib.rc_readback(ctx, got.ctypes.data_as(C.c_void_p), w * 16), "rc_readback")
    pos, rad, col = oracle.generate_scene(n, 3)
    cam = oracle.camera_data((0, 0, 20), (0, 0, 0), w, h)              # the context's initial camera
    assert_same_bits(got, oracle.render(pos, rad, col, cam, w, h), "context, generated scene")

    # caller scene + camera move + one input poll
    rng = np.random.default_rng(5)
    pos2 = rng.uniform(-6, 6, (25, 3)).astype(np.float32); pos2[:, 2] -= 10
    rad2 = rng.uniform(0.3, 1.5, 25).astype(np.float32)
    col2 = rng.uniform(0, 1, (25, 3)).astype(np.float32)
    abi.check(rclib.rc_scene_upload(ctx, 25, pos2.ctypes.data_as(C.c_void_p), rad2.ctypes.data_as(C.c_void_p),
                                    col2.ctypes.data_as(C.c_void_p)), "rc_scene_upload")
    p, r = abi.RcVec3(1.0, 0.5, 18.0), abi.RcVec3(0.05, -0.1, 0.0)
    abi.check(rclib.rc_camera_set(ctx, C.byref(p), C.byref(r)), "rc_camera_set")
    inp = abi.RcInput(rbutton=1, key_w=1, cursor_dx=2, cursor_dy=-1)
    assert rclib.rc_camera_input(ctx, C.byref(inp), 0.0005) == 1
    assert rclib.rc_camera_input(ctx, C.byref(abi.RcInput()), 0.0005) == 0
    from cudaraytracing_amd.host import Camera
    pc = Camera(w, h); pc.position, pc.rotation = (1.0, 0.5, 18.0), (np.float32(0.05).item(), np.float32(-0.1).item(), 0.0)
    pc.handle_input(np.float32(0.0005).item(), "r w", (2, -1))
    abi.check(rclib.rc_ctx_render(ctx, None), "rc_ctx_render")
    abi.check(rclib.rc_readback(ctx, got.ctypes.data_as(C.c_void_p), w * 16), "rc_readback")
    cam2 = oracle.camera_data(pc.position, pc.rotation, w, h)
    want = oracle.render(pos2, rad2, col2, cam2, w, h)
    assert_same_bits(got, want, "context, caller scene after input")
    # culled variant through the context: it lends its own workspace
    big = oracle.generate_scene(2500, 9)
    abi.check(rclib.rc_scene_upload(ctx, 2500, *(a.ctypes.data_as(C.c_void_p) for a in big)), "rc_scene_upload")
    copts = abi.RcOpts(variant=abi.VARIANT_CULLED)
    for _ in range(2):
        abi.check(rclib.rc_ctx_render(ctx, C.byref(copts)), "rc_ctx_render culled")
    abi.check(rclib.rc_readback(ctx, got.ctypes.data_as(C.c_void_p), w * 16), "rc_readback")
    assert_same_bits(got, oracle.render(*big, cam2, w, h), "context, culled with its own workspace")
    abi.check(rclib.rc_scene_upload(ctx, 25, pos2.ctypes.data_as(C.c_void_p), rad2.ctypes.data_as(C.c_void_p),
                                    col2.ctypes.data_as(C.c_void_p)), "rc_scene_upload")
    abi.check(rclib.rc_ctx_render(ctx, None), "rc_ctx_render")
    img = np.empty((h, w, 4), np.uint8)
    abi.check(rclib.rc_readback_rgba8(ctx, img.ctypes.data_as(C.c_void_p), w * 4, 1), "rc_readback_rgba8")
    want8 = oracle.to_rgba8(want, flip=True)
    assert np.array_equal(img, want8)
    ppm = tmp_path / "ctx.ppm"
    abi.check(rclib.rc_ctx_write_ppm(ctx, str(ppm).encode()), "rc_ctx_write_ppm")
    head = f"P6\n{w} {h}\n255\n".encode()
    data = ppm.read_bytes()
    assert data.startswith(head) and np.array_equal(np.frombuffer(data[len(head):], np.uint8).reshape(h, w, 3), want8[..., :3])
    abi.check(rclib.rc_ctx_destroy(ctx), "rc_ctx_destroy")
    assert rclib.rc_ctx_create(C.byref(ctx), 99, w, h) == abi.RC_EINVAL
    assert rclib.rc_ctx_render(None, None) == abi.RC_EINVAL


@pytest.mark.parametrize("w,h,n", [(333, 77, 5), (640, 480, 1024), (200, 120, 2049), (700, 300, 3000), (64, 48, 8200), (65, 17, 0)])
def test_culled_with_workspace_records(oracle, w, h, n):
    """RC_VARIANT_CULLED with rc_opts.workspace: the records come from the once-per-frame kernel
    instead of every workgroup's own prepass; same bits (REF and EXT)."""
    pos, rad, col = oracle.generate_scene(n, 17)
    cam = oracle.camera_data((0.4, -0.3, 19.0), (0.03, -0.06, 0), w, h)
    want = oracle.render(pos, rad, col, cam, w, h)
    for ws in ("records", True):
        got = pixels(gpu_render(pos, rad, col, cam, w, h, variant=abi.VARIANT_CULLED, workspace=ws), w)
        assert_same_bits(got, want, f"culled+workspace({ws}) {w}x{h} n={n}")
    # cameras inside / hugging spheres and a wide field of view: bins that must list everything
    for k, off in [(0, 0.5), (min(3, max(n - 1, 0)), 1.00001)]:
        if n == 0:
            break
        cpos = tuple(float(v) for v in (pos[k] + np.array([0.6, 0.0, 0.8]) * rad[k] * off))
        from gpu_util import to_rc_camera
        cam2 = to_rc_camera(oracle.camera_data(cpos, (0.2, 0.4, 0), w, h))
        cam2.fov = 2.9
        want2 = pixels(gpu_render(pos, rad, col, cam2, w, h, variant=abi.VARIANT_LITERAL), w)
        got2 = pixels(gpu_render(pos, rad, col, cam2, w, h, variant=abi.VARIANT_CULLED, workspace=True), w)
        assert_same_bits(got2, want2, f"culled+bins camera near sphere {k} n={n}")
    if n <= 3000:
        ext = _default_ext(oracle)
        want = oracle.render(pos, rad, col, cam, w, h, ext=ext)
        got = pixels(gpu_render(pos, rad, col, cam, w, h, variant=abi.VARIANT_CULLED, workspace=True, ext=ext), w)
        assert_same_bits(got, want, f"culled+workspace EXT {w}x{h} n={n}")


def test_workspace_is_validated(rclib):
    from cudaraytracing_amd.host import DeviceBuffer
    assert rclib.rc_workspace_bytes(1000) == 16000 and rclib.rc_workspace_bytes(0) == 0
    surf, arr, ws = DeviceBuffer(64 * 64 * 16), DeviceBuffer(4096), DeviceBuffer(256)
    screen = abi.RcScreen(surf.ptr.value, 64, 64, 1024)
    scene = abi.RcScene(sphere_num=100, position=arr.ptr.value, radius=arr.ptr.value, color=arr.ptr.value)
    scene.cam.fov = 0.5; scene.cam.aspect = 1.0
    opts = abi.RcOpts(variant=abi.VARIANT_CULLED, workspace=ws.ptr.value, workspace_bytes=256)   # needs 1600
    assert rclib.rc_render(C.byref(screen), C.byref(scene), C.byref(opts)) == abi.RC_EINVAL
    assert b"workspace" in rclib.rc_last_error_string()
    for b in (surf, arr, ws):
        b.free()


@pytest.mark.parametrize("mode,extra,launcher", [("bands", ["--gather", "rgba8"], True), ("bands", [], False),
                                                 ("bands", ["--band-layout", "interleaved"], False),
                                                 ("frames", ["--sequence"], True), ("frames", [], False)])
def test_bench_two_ranks_share_one_gpu(mode, extra, launcher):
    """bench.py's N>1 launch contract with two ranks on this one GPU (gloo instead of RCCL, both ranks
    on device 0): one JSON line from rank 0; in bands mode the gathered frame equals the whole frame.
    launcher=False: plain `python bench.py --gpus 2`, which must start its two ranks itself."""
    import json, socket, sys
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(RC_BENCH_BACKEND="gloo", RC_BENCH_SHARE_GPU="1")
    head = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
            "--master-addr", "127.0.0.1", "--master-port", str(port)] if launcher else [sys.executable]
    r = subprocess.run(head + [os.path.join(ROOT, "bench.py"),
                        "--gpus", "2", "--steps", "6", "--warmup", "2", "--mode", mode, "--config", "c2_1920x1080_n64",
                        "--no-clock-ramp"] + extra, capture_output=True, text=True, timeout=400, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 6 and len(d["per_rank_step_ms"]) == 2
    assert d["scaling"] == ("strong" if mode == "bands" else "weak")
    if mode == "bands":
        assert d["gather_check"] is True
    else:
        # frames mode exchanges nothing on the data path, so the line carries the untimed bands_check leg: one
        # 7680x4320 x 4096 frame cut into row bands, gathered on rank 0 and compared with the whole frame
        bc = d["bands_check"]
        assert "error" not in bc, bc
        assert bc["workload"] == "c4_7680x4320_n4096" and len(bc["per_rank_render_ms"]) == 2
        assert bc["rows_of_this_run"] == [[[0, 1080], [1080, 2160]], [[2160, 3240], [3240, 4320]]]
        assert bc["torch"]["gather_check"] is True and bc["torch"]["frame_ms"] > 0
        assert "skipped" in bc["c"]                      # librc_gather.so needs RCCL, this rehearsal runs on gloo


@pytest.mark.parametrize("extra", [[], ["--gather", "rgba8", "--subbands", "3"], ["--gather-impl", "c"],
                                   ["--gather-impl", "c", "--gather", "rgba8", "--subbands", "3"],
                                   ["--band-layout", "interleaved", "--gather", "rgba8"]])
def test_bench_band_assembly_on_device_tensors_one_rank(extra):
    """--mode bands on ONE rank still renders the band piece by piece (y_offset / full_height) and
    assembles the pieces on device tensors under a one-rank RCCL process group: the assembled frame
    must equal the frame rendered whole, bit for bit."""
    import json, sys
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "6", "--warmup", "2",
                        "--mode", "bands", "--config", "c2_1920x1080_n64", "--no-clock-ramp", "--no-cpu"] + extra,
                       capture_output=True, text=True, timeout=400, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["scaling"] == "strong" and d["gather_check"] is True
    assert d["config"]["sharding"] == "bands"


def test_bench_under_the_launcher_with_rccl_one_rank():
    """The driver's multi-GPU launch line with one rank: torch.distributed.run, RCCL process group bound to
    the device, barrier, MAX reduction and per-rank gather of the timed region, one JSON line."""
    import json, socket, sys
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
                        "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"),
                        "--gpus", "1", "--steps", "6", "--warmup", "2", "--no-cpu", "--no-extras", "--no-clock-ramp"],
                       capture_output=True, text=True, timeout=400, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["steps"] == 6 and d["scaling"] == "weak"
    assert d["config"]["backend"] in ("nccl", None) and len(d["per_rank_step_ms"]) == 1
    assert 0.0 < d["roofline"]["frac"] <= 1.0


def test_c_gather_one_rank_assembles_pieces(oracle):
    """librc_gather.so with a one-rank communicator (ncclCommInitRank on this GPU): a band rendered in
    three pieces, each handed to rc_gather_piece, arrives in the full frame bit for bit."""
    from cudaraytracing_amd import gather
    w, h, n = 320, 100, 40
    pos, rad, col = oracle.generate_scene(n, 4)
    cam = oracle.camera_data((0, 0, 20), (0, 0, 0), w, h)
    lib = abi.load()
    gs = GpuScene(pos, rad, col)
    sc = gs.scene(cam)
    band, full = DeviceBuffer(w * 16 * h), DeviceBuffer(w * 16 * h)
    abi.check(lib.rc_memset(full.ptr, 0, w * 16 * h, None))
    g = gather.Gather(1, 0, gather.unique_id())
    parts = 3
    g.begin(parts, full.ptr, w * 16, h, None)            # one rank: nothing to receive, must be accepted all the same
    for k in range(parts):
        sb, se = gather.sub_band(0, h, parts, k)
        screen = abi.RcScreen(band.ptr.value + sb * w * 16, w, se - sb, w * 16)
        opts = abi.RcOpts(y_offset=sb, full_height=h)
        abi.check(lib.rc_render(C.byref(screen), C.byref(sc), C.byref(opts)), "rc_render")
        g.piece(k, parts, band.ptr, full.ptr, w * 16, h, None)
    g.wait(None)
    abi.check(lib.rc_sync(None))
    got = np.empty((h, w, 4), np.float32)
    abi.check(lib.rc_download_2d(got.ctypes.data_as(C.c_void_p), w * 16, full.ptr, w * 16, w * 16, h, None))
    g.destroy()
    for b in (band, full):
        b.free()
    gs.free()
    assert_same_bits(got, oracle.render(pos, rad, col, cam, w, h), "frame assembled by librc_gather.so")


def test_config4_split_eight_bands_two_pieces_each(oracle, golden):
    """The 7680x4320 x 4096 frame exactly as eight GPUs would render it -- eight row bands of 540 rows,
    each in two sub-bands (rc_opts.y_offset / full_height) -- on this one GPU, straight into the rows
    of one full frame: the assembled frame has the committed hash of the frame rendered whole."""
    from cudaraytracing_amd import sharding
    fx = golden["c4_7680x4320_n4096"]
    w, h = fx["width"], fx["height"]
    pos, rad, col = oracle.generate_scene(fx["n"], fx["seed"])
    cam = oracle.camera_data(fx["position"], fx["rotation"], w, h)
    lib = abi.load()
    scene = GpuScene(pos, rad, col)
    sc = scene.scene(cam)
    pitch = w * 16
    surf = DeviceBuffer(pitch * h)
    abi.check(lib.rc_memset(surf.ptr, 1, pitch * h, None))
    pieces = 0
    for r in range(8):
        b, e = configs.band_rows(h, 8, r)
        assert e - b == 540
        for sb, se in sharding.sub_bands(b, e, 2):
            screen = abi.RcScreen(surf.ptr.value + sb * pitch, w, se - sb, pitch)
            opts = abi.RcOpts(y_offset=sb, full_height=h)
            abi.check(lib.rc_render(C.byref(screen), C.byref(sc), C.byref(opts)), "rc_render")
            pieces += 1
    assert pieces == 16
    abi.check(lib.rc_sync(None))
    got = np.empty((h, w, 4), np.float32)
    abi.check(lib.rc_download_2d(got.ctypes.data_as(C.c_void_p), pitch, surf.ptr, pitch, pitch, h, None))
    surf.free()
    scene.free()
    assert f"{oracle.fnv_words(got):016x}" == fx["fnv"]
    assert int(got[..., :3].any(axis=2).sum()) == fx["lit_pixels"]


def test_c_example_renders_the_reference_frame(oracle, tmp_path):
    """examples/render_frame.c (C99, context API): the reference's own window size and sphere count;
    its PPM equals the oracle's frame."""
    exe = str(tmp_path / "render_frame")
    libdir = os.path.join(ROOT, "cudaraytracing_amd")
    subprocess.run(["gcc", "-std=c99", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "render_frame.c"),
                    "-L" + libdir, "-lrc_hip", "-Wl,-rpath," + libdir, "-o", exe], check=True, capture_output=True)
    out = tmp_path / "ref.ppm"
    r = subprocess.run([exe, str(out)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    w, h, n = 1280, 768, 512
    pos, rad, col = oracle.generate_scene(n, 0)
    cam = oracle.camera_data((0, 0, 20), (0, 0, 0), w, h)
    want = oracle.render(pos, rad, col, cam, w, h)
    assert f"{int(want[..., :3].any(axis=2).sum())} lit pixels" in r.stdout
    head = f"P6\n{w} {h}\n255\n".encode()
    data = out.read_bytes()
    img = np.frombuffer(data[len(head):], np.uint8).reshape(h, w, 3)
    assert np.array_equal(img, oracle.to_rgba8(want, flip=True)[..., :3])


@pytest.mark.parametrize("w,h,n", [(65536, 2, 3), (2, 65536, 3), (64, 32, 70000)])
def test_extreme_shapes(oracle, w, h, n):
    """The widest and the tallest frame the library accepts, and more spheres than 16-bit bin lists
    can index (the culled variant then works from the records alone)."""
    pos, rad, col = oracle.generate_scene(n, 2)
    cam = oracle.camera_data((0, 0, 20), (0.01, 0.02, 0), w, h)
    want = oracle.render(pos, rad, col, cam, w, h)
    for variant, ws in [(abi.VARIANT_AUTO, False), (abi.VARIANT_CULLED, True), (abi.VARIANT_LITERAL, False)]:
        got = pixels(gpu_render(pos, rad, col, cam, w, h, variant=variant, workspace=ws), w)
        assert_same_bits(got, want, f"{w}x{h} n={n} variant {variant}")


@pytest.mark.parametrize("w,h,n,variant", [(1280, 768, 673, abi.VARIANT_TILED2), (1280, 768, 1024, abi.VARIANT_AUTO),
                                           (1280, 768, 2100, abi.VARIANT_TILED2), (1296, 771, 4096, abi.VARIANT_AUTO),
                                           (1920, 1080, 2049, abi.VARIANT_TILED4), (1920, 1080, 4000, abi.VARIANT_AUTO)])
def test_lds_pass_sizes_on_frames_with_many_workgroups(w, h, n, variant):
    """The launcher shortens the LDS passes of the 2- and 4-pixel REF kernels (672 / 960 spheres) only on frames
    that cut into more workgroups than the small test frames do: sphere counts just past a pass, several passes
    with a ragged last one, against the literal kernel on the device (itself held to the oracle elsewhere)."""
    rng = np.random.default_rng(n)
    pos = rng.uniform(-12, 12, (n, 3)).astype(np.float32); pos[:, 2] = rng.uniform(-18, -2, n).astype(np.float32)
    rad = rng.uniform(0.05, 0.6, n).astype(np.float32)
    col = rng.uniform(0, 1, (n, 3)).astype(np.float32)
    from cudaraytracing_amd.host import Camera
    cam = Camera(w, h); cam.position, cam.rotation = (0.5, -0.25, 20.0), (0.02, -0.03, 0.0)
    scene = GpuScene(pos, rad, col)
    want = pixels(gpu_render(pos, rad, col, cam.get_data(), w, h, variant=abi.VARIANT_LITERAL, scene=scene), w)
    got = pixels(gpu_render(pos, rad, col, cam.get_data(), w, h, variant=variant, scene=scene), w)
    scene.free()
    assert_same_bits(got, want, f"{w}x{h} n={n} variant {variant}")
