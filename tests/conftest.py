import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure); built on demand from oracle/rc_oracle.c."""
    from oracle import binding
    binding.load()
    return binding


@pytest.fixture(scope="session")
def golden():
    with open(os.path.join(GOLDEN_DIR, "golden.json")) as f:
        frames = json.load(f)["frames"]
    return {f["name"]: f for f in frames}


@pytest.fixture(scope="session")
def ref_kat():
    with open(os.path.join(GOLDEN_DIR, "ref_kat.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def kat_scene():
    """(position, radius, color) of the scene the reference known-answer vectors were taken on."""
    rows = np.fromfile(os.path.join(GOLDEN_DIR, "kat_scene_libstdcxx_1024.f32"), np.float32).reshape(-1, 7)
    return rows[:, 4:7].copy(), rows[:, 0].copy(), rows[:, 1:4].copy()


@pytest.fixture(scope="session")
def rclib():
    """librc_hip.so through ctypes; fails (not skips) when the library is missing."""
    from cudaraytracing_amd import abi
    return abi.load()


def pytest_sessionstart(session):
    """A fresh checkout has no librc_hip.so yet: build it once (hipcc cross-compiles without a GPU).
    Only when the file is absent -- a present library is never rebuilt from inside the tests."""
    from cudaraytracing_amd import abi, build
    if not os.path.exists(abi.LIB_PATH):
        try:
            build.build(verbose=False)
        except Exception as e:          # the tests that need the library then fail with abi.load()'s message
            print(f"conftest: could not build librc_hip.so: {e}", file=sys.stderr)
    # the checker and, where the reference's sources are present, the cross-checks against its headers
    ref_dir = os.path.join(ROOT, "oracle", "_ref")
    if os.path.isdir("/root/reference/RayCasting/RayCasting") and os.path.exists(abi.LIB_PATH) and \
            not all(os.path.exists(os.path.join(ref_dir, n)) for n in ("ref_algebra", "ref_pixel", "ref_input", "dropin_caller")):
        import subprocess
        subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "all"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
