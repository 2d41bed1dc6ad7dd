"""Workload definitions shared by tests, bench.py and the golden-fixture generator.

The reference hard-codes one workload (1280x768, 512 spheres, camera at (0,0,20):
main.cpp:32-33, SceneManager.cpp:5,8); these are the benchmark configurations, all with
the reference's scene distribution (SceneManager.cpp:35-45) from mt19937 seed 0.
"""
import math

import numpy as np

DEFAULT_POSITION = (0.0, 0.0, 20.0)     # SceneManager.cpp:5
DEFAULT_ROTATION = (0.0, 0.0, 0.0)      # Camera.h:46

CONFIGS = {
    "c1_640x480_n8": dict(width=640, height=480, n=8),
    "c2_1920x1080_n64": dict(width=1920, height=1080, n=64),
    "c3_3840x2160_n1024": dict(width=3840, height=2160, n=1024),
    "c4_7680x4320_n4096": dict(width=7680, height=4320, n=4096),
    "ref_1280x768_n512": dict(width=1280, height=768, n=512),   # the reference's own window
}
for _c in CONFIGS.values():
    _c.setdefault("seed", 0)
    _c.setdefault("position", DEFAULT_POSITION)
    _c.setdefault("rotation", DEFAULT_ROTATION)

# EXT-mode additions used by the benchmark configurations that ask for "1 plane, 1 point light,
# Phong + hard shadows": floor plane y = -6, light above and in front of the sphere cloud.
DEFAULT_EXT = dict(plane_normal=(0.0, 1.0, 0.0), plane_h=-6.0, plane_color=(0.6, 0.6, 0.55),
                   light_pos=(8.0, 14.0, 12.0), specular=0.4, shininess_log2=5, has_plane=1, shadows=1)

SEQUENCE_FRAMES = 300
SEQUENCE = dict(width=3840, height=2160, n=1024, seed=0, frames=SEQUENCE_FRAMES)


def sequence_camera(frame, frames=SEQUENCE_FRAMES):
    """Scripted camera path (replaces the interactive input of main.cpp:53-128).

    Evaluated in double and rounded to float once, so CPU and GPU paths share the very same
    position/rotation values: theta = 2*pi*f/frames, rotation = (0.15 sin 2theta,
    0.5 sin theta, 0), position = (6 sin theta, 1.5 sin 2theta, 20).
    """
    th = 2.0 * math.pi * frame / frames
    rot = (0.15 * math.sin(2 * th), 0.5 * math.sin(th), 0.0)
    pos = (6.0 * math.sin(th), 1.5 * math.sin(2 * th), 20.0)
    f32 = lambda t: tuple(float(np.float32(v)) for v in t)
    return f32(pos), f32(rot)


def sequence_light(frame, frames=SEQUENCE_FRAMES):
    """EXT-mode light animation of the sequence: the point light orbits the sphere cloud at height 14,
    radius 14 (double arithmetic, rounded to float once, like the camera path)."""
    th = 2.0 * math.pi * frame / frames
    return tuple(float(np.float32(v)) for v in (14.0 * math.cos(th), 14.0, -10.0 + 22.0 * math.sin(th)))


def band_rows(height, world, rank):
    """Row band [begin, end) of `rank` among `world` ranks; band heights differ by <= 1."""
    base, rem = divmod(height, world)
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)


def pixels_per_lane(width, rows, n, ext, culled, compute_units, full_height=0):
    """Pixels per lane the library's launcher picks for `rows` rows of a frame of `full_height` rows (0: the rows are
    the frame): asked of the library itself (rc_pixels_per_lane, the rule rc_launch_draw applies), for bench.py and
    the tools, which price the executed work per test with it."""
    from . import abi
    p = abi.load().rc_pixels_per_lane(width, rows, full_height, n, abi.MODE_EXT if ext else abi.MODE_REF,
                                      abi.VARIANT_CULLED if culled else abi.VARIANT_AUTO, compute_units)
    if p < 0:
        raise ValueError(f"rc_pixels_per_lane({width}, {rows}, {full_height}, {n}): error {p}")
    return p
