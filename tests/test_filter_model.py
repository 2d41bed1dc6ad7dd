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


@pytest.fixture(scope="module")
def bound_model(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("bm") / "bound_model")
    subprocess.run(["gcc", "-O2", "-ffp-contract=off", "-I" + os.path.join(ROOT, "oracle"),
                    os.path.join(ROOT, "tests", "bound_model.c"), os.path.join(ROOT, "oracle", "rc_oracle.c"),
                    "-o", exe, "-lm", "-lpthread"], check=True)
    return exe


def test_near_bound_is_below_every_returned_distance(bound_model):
    """rc_near_bound (the skip of exact tests for pixels already nearer than a sphere's nearest point): over
    ~7 M positive distances the reference's arithmetic returns -- rays through the centre, grazing rays,
    radii from 1e-4 to 1 of the distance -- none is below the bound, evaluated with worst-case 1-ulp
    hardware sqrt / reciprocal; the float shortfall against |v| - r stays inside the derived constant."""
    r = subprocess.run([bound_model, "near", "8000000", "11"], capture_output=True, text=True, timeout=300)
    m = re.search(r"hits=(\d+) usable=(\d+) violations=(\d+) min_slack=(\S+) max_shortfall_in_u_vv_over_r=(\S+)", r.stdout)
    assert m, r.stdout + r.stderr
    assert r.returncode == 0 and int(m.group(3)) == 0
    assert int(m.group(1)) > 5e6 and int(m.group(2)) > 5e6      # hits, and hits whose bound is positive (skips possible)
    assert float(m.group(4)) > 1e-6                             # slack relative to |v|
    assert float(m.group(5)) < 12.25                            # derived worst case of the reference's rounding


@pytest.mark.parametrize("k_ulps,seed,min_margin", [(0, 21, 3e-6), (64, 22, 5e-7)])
def test_dominant_axis_shadow_filter_flags_every_hit(bound_model, k_ulps, seed, min_margin):
    """The two-FMA form of the EXT shadow filter: every shadow ray whose exact test returns a positive
    distance is flagged, for every admissible axis and sign, with the reciprocal one ulp either way."""
    r = subprocess.run([bound_model, "axis", "8000000", str(k_ulps), str(seed)], capture_output=True, text=True, timeout=300)
    m = re.search(r"hits=(\d+) checks=(\d+) missed=(\d+) always=(\d+) forced=(\d+) min_margin=(\S+)", r.stdout)
    assert m, r.stdout + r.stderr
    assert r.returncode == 0 and int(m.group(3)) == 0
    assert int(m.group(1)) > 4e6 and int(m.group(2)) > 1e7
    assert float(m.group(6)) > min_margin
