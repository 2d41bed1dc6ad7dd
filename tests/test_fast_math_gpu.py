"""The kernels' short correctly-rounded square root and reciprocal (rc_sqrt_core / rc_rcp_core in
csrc/rc_kernels.hip) against the compiler's general sequences for EVERY operand they admit -- all 2^23
mantissas of every binade from 2^-96 up to 2^96, both signs for the reciprocal -- evaluated on the device, and the
general sequences against IEEE binary32 arithmetic on the host (numpy) on whole binades.  vec3::norm
(vec3.h:168-176,89-92) is one such square root and one such reciprocal per ray."""
import ctypes as C

import numpy as np
import pytest

from cudaraytracing_amd import abi
from cudaraytracing_amd.host import DeviceBuffer

pytestmark = pytest.mark.gpu

LO_EXP, HI_EXP = 127 - 96, 127 + 96          # biased exponents of 2^-96 (first admitted) and 2^96 (first not admitted)


def _run(which, first_bits, count, want_out=False):
    lib = abi.load()
    counter = DeviceBuffer(8)
    abi.check(lib.rc_memset(counter.ptr, 0, 8, None))
    fast = DeviceBuffer(4 * count) if want_out else None
    gen = DeviceBuffer(4 * count) if want_out else None
    abi.check(lib.rc_selftest_math(which, first_bits, count, counter.ptr, fast.ptr if fast else None,
                                   gen.ptr if gen else None, None), "rc_selftest_math")
    abi.check(lib.rc_sync(None))
    mism = np.zeros(1, np.uint64)
    abi.check(lib.rc_download_2d(mism.ctypes.data_as(C.c_void_p), 8, counter.ptr, 8, 8, 1, None))
    outs = None
    if want_out:
        outs = []
        for b in (fast, gen):
            o = np.empty(count, np.float32)
            abi.check(lib.rc_download_2d(o.ctypes.data_as(C.c_void_p), 4 * count, b.ptr, 4 * count, 4 * count, 1, None))
            outs.append(o)
            b.free()
    counter.free()
    return int(mism[0]), outs


@pytest.mark.parametrize("which,sign", [(0, 0), (1, 0), (1, 1)])
def test_fast_sqrt_rcp_exhaustive(which, sign):
    first = (sign << 31) | (LO_EXP << 23)
    count = (HI_EXP - LO_EXP) << 23                     # 192 binades x 2^23 mantissas = 1.61e9 operands
    mism, _ = _run(which, first, count)
    assert mism == 0, f"{'rcp' if which else 'sqrt'}: {mism} of {count} operands differ from the general sequence"


@pytest.mark.parametrize("which", [0, 1])
def test_general_sequences_are_ieee(which):
    """Two whole binades each near the ends and the middle of the range (even and odd exponents: the square root's
    two cases), device results of both sequences against numpy's correctly rounded float32 sqrt and division."""
    for e in (LO_EXP, LO_EXP + 1, 126, 127, 128, HI_EXP - 2, HI_EXP - 1):
        first = e << 23
        mism, (fast, gen) = _run(which, first, 1 << 23, want_out=True)
        x = (np.arange(1 << 23, dtype=np.uint32) + np.uint32(first)).view(np.float32)
        want = np.sqrt(x) if which == 0 else (np.float32(1.0) / x)
        assert want.dtype == np.float32
        assert np.array_equal(gen.view(np.uint32), want.view(np.uint32)), f"general sequence, exponent {e}"
        assert np.array_equal(fast.view(np.uint32), want.view(np.uint32)), f"short sequence, exponent {e}"
        assert mism == 0


def test_selftest_rejects_bad_arguments():
    lib = abi.load()
    counter = DeviceBuffer(8)
    assert lib.rc_selftest_math(2, 0, 1, counter.ptr, None, None, None) == abi.RC_EINVAL
    assert lib.rc_selftest_math(0, 0xFFFFFFFF, 2, counter.ptr, None, None, None) == abi.RC_EINVAL
    assert lib.rc_selftest_math(0, 0, 1, None, None, None, None) == abi.RC_EINVAL
    counter.free()
