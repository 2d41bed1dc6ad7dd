"""Scripted replacement of the reference's HandleInput (main.cpp:53-128): rc_handle_input.

Runs without a GPU (the function is host arithmetic inside librc_hip.so).  Expected values come
from the reference's own Camera / vec3 types: tests/golden/input_script.json was written by
oracle/_ref/ref_input --emit (built from oracle/ref_input.cpp against the reference's headers)."""
import ctypes as C
import json
import os
import subprocess

import numpy as np
import pytest

from cudaraytracing_amd import abi
from cudaraytracing_amd.host import Camera

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _f32(bits):
    return np.array([bits], np.uint32).view(np.float32)[0].item()


def _bits(x):
    return int(np.array([x], np.float32).view(np.uint32)[0])


def test_script_fixture_replays_bit_exactly():
    lib = abi.load()
    with open(os.path.join(ROOT, "tests", "golden", "input_script.json")) as f:
        fx = json.load(f)
    pos = abi.RcVec3(*[float(v) for v in fx["start_position"]])
    rot = abi.RcVec3(0.0, 0.0, 0.0)
    assert len(fx["polls"]) == 96
    for i, p in enumerate(fx["polls"]):
        k = p["keys"]
        inp = abi.RcInput(k[0], k[1], k[2], k[3], k[4], k[5], k[6], k[7], p["cursor"][0], p["cursor"][1])
        rc = lib.rc_handle_input(C.byref(pos), C.byref(rot), float(fx["movement_speed"]), float(fx["rotation_speed"]),
                                 _f32(p["dt_bits"]), C.byref(inp))
        assert rc == p["update"], f"poll {i}"
        assert [_bits(pos.x), _bits(pos.y), _bits(pos.z)] == p["position_bits"], f"poll {i} position"
        assert [_bits(rot.x), _bits(rot.y), _bits(rot.z)] == p["rotation_bits"], f"poll {i} rotation"


def test_against_reference_camera_types_when_built():
    exe = os.path.join(ROOT, "oracle", "_ref", "ref_input")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/ref_input not built (reference headers absent)")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "100046 polls, 0 mismatches" in r.stdout


def test_camera_mirror_moves_like_the_reference_controls():
    cam = Camera(1280, 768)
    cam.position = (0.0, 0.0, 20.0)
    assert cam.handle_input(2.0, "") is False and cam.position == (0.0, 0.0, 20.0)
    # w: forward = (-0, 0, -1) at zero rotation; speed 100 * 50, dt in milliseconds (main.cpp:73-100,137)
    assert cam.handle_input(2.0, "w") is True
    assert cam.position == (0.0, 0.0, 20.0 - 100.0 * 50.0 * 2.0)
    # shift: x5 instead of x50 (main.cpp:75-80); q: up along world y
    assert cam.handle_input(1.0, "shift q") is True
    assert cam.position[1] == 500.0
    # mouse look: rotation += (-dy, -dx, 0) * dt * rotation_speed (main.cpp:64-70)
    assert cam.handle_input(0.5, "r", cursor=(3, -2)) is True
    assert cam.rotation == (10.0, -15.0, 0.0)
    with pytest.raises(ValueError):
        cam.handle_input(1.0, "x")


def test_null_pointers_are_rejected():
    lib = abi.load()
    assert lib.rc_handle_input(None, None, 1.0, 1.0, 1.0, None) == abi.RC_EINVAL
