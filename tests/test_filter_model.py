"""Soundness of the tiled kernel's conservative filter, on the CPU.

tests/filter_model.c rebuilds the filter record and the per-pixel filter value exactly as the
kernel computes them (binary32, explicit fmaf) and compares "filter flags the pair" with "the
reference's exact test hits" over ~10^8 (pixel, sphere) pairs of random scenes whose cameras sit
anywhere from 1e-7 relative to ten radii away from a sphere's surface.  The hardware's 1-ulp
sqrt / reciprocal are modelled by their worst case (k lowered by several ulps).  Every exact hit
must be flagged, with margin to spare.
"""
import os
import re
import subprocess

import pytest

from conftest import ROOT


@pytest.fixture(scope="module")
def model(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("fm") / "filter_model")
    subprocess.run(["gcc", "-O2", "-ffp-contract=off", "-I" + os.path.join(ROOT, "oracle"),
                    os.path.join(ROOT, "tests", "filter_model.c"), os.path.join(ROOT, "oracle", "rc_oracle.c"),
                    "-o", exe, "-lm", "-lpthread"], check=True)
    return exe


@pytest.mark.parametrize("k_ulps,seed,min_margin", [(0, 1, 5e-6), (4, 2, 5e-6), (64, 3, 1e-6)])
def test_every_exact_hit_is_flagged(model, k_ulps, seed, min_margin):
    r = subprocess.run([model, "40", "192", "108", "200", str(k_ulps), str(seed)], capture_output=True, text=True, timeout=300)
    m = re.search(r"pairs=(\d+) hits=(\d+) missed=(\d+) always_pairs=(\d+) forced_pixels=(\d+) min_margin=(\S+)", r.stdout)
    assert m, r.stdout + r.stderr
    pairs, hits, missed = int(m.group(1)), int(m.group(2)), int(m.group(3))
    assert r.returncode == 0 and missed == 0
    assert pairs > 5e7 and hits > 5e5                   # the sample really contains hits
    assert float(m.group(6)) > min_margin               # and they clear the threshold with room
