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
