"""MI355X-native ray-casting hot path (gfx950 HIP kernels behind a C ABI).

The product is cudaraytracing_amd/librc_hip.so (sources in csrc/, interface in
include/rc_abi.h).  The Python modules are plumbing for tests and bench.py:
    abi      ctypes binding of the C ABI
    host     mirror of the reference's host classes (Camera, SceneManager, RayTracingKernel)
    configs  benchmark workloads, scripted camera path, row-band partition
    build    hipcc build of the library
"""
__all__ = ["abi", "host", "configs", "build"]
