"""Builds librc_hip.so (HIP kernels + C ABI) for gfx950 with hipcc, in-tree.

hipcc cross-compiles without a GPU.  The library is rebuilt only when a source is newer
than the existing binary, so a tree that already holds the .so (the GPU box) builds nothing.
"""
import os
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_HERE)
CSRC = os.path.join(_HERE, "csrc")
LIB_PATH = os.path.join(_HERE, "librc_hip.so")
CLI_PATH = os.path.join(_HERE, "rc_render")
GATHER_PATH = os.path.join(_HERE, "librc_gather.so")      # row-band assembly over RCCL (links librccl)
GATHER_SRC = os.path.join(CSRC, "rc_gather.cpp")

SOURCES = [os.path.join(CSRC, "rc_kernels.hip"), os.path.join(CSRC, "rc_host.hip")]
HEADERS = [os.path.join(CSRC, "rc_kernels.h"), os.path.join(_ROOT, "include", "rc_abi.h"),
           os.path.join(_ROOT, "include", "rc_dropin.hpp")] + [
    os.path.join(CSRC, "host", h) for h in ("rc_camera.hpp", "rc_scene_manager.hpp", "rc_raycaster.hpp")]

# -ffp-contract=off: every product/sum of the exact path is a separately rounded binary32
# operation; the filter's FMAs are written explicitly with __builtin_fmaf.
HIPCC_FLAGS = [
    "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
    "-fno-fast-math", "-Wall", "-Wno-unused-result",
    "-I" + os.path.join(_ROOT, "include"), "-I" + CSRC,
]


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.exists(d) and os.path.getmtime(d) > t for d in deps)


def hipcc_path():
    p = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(p):
        raise RuntimeError("hipcc not found: cannot build librc_hip.so")
    return p


def build(force=False, verbose=False):
    """Compile librc_hip.so and the rc_render CLI if missing or stale. Returns the .so path."""
    if force or _stale(LIB_PATH, SOURCES + HEADERS):
        # A library older than its sources must never survive a failed build (it would travel to the GPU
        # box and run there against newer bindings and tests): it is removed first and the new one is
        # moved into place only when the compiler succeeded.
        for stale in (LIB_PATH, CLI_PATH):
            if os.path.exists(stale):
                os.remove(stale)
        tmp = LIB_PATH + ".tmp"
        cmd = [hipcc_path()] + HIPCC_FLAGS + ["-shared", "-o", tmp] + SOURCES
        if verbose:
            print(" ".join(cmd))
        try:
            subprocess.run(cmd, check=True, cwd=_ROOT)
            os.replace(tmp, LIB_PATH)
        finally:
            if os.path.exists(tmp):
                os.remove(tmp)
    cli_src = os.path.join(CSRC, "host", "rc_render_main.cpp")
    cli_deps = [cli_src, LIB_PATH] + [
        os.path.join(CSRC, "host", h) for h in ("rc_camera.hpp", "rc_scene_manager.hpp", "rc_raycaster.hpp", "rc_frame_pipeline.hpp")
    ]
    if os.path.exists(cli_src) and (force or _stale(CLI_PATH, cli_deps)):
        cmd = ["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-I" + os.path.join(_ROOT, "include"),
               "-I" + os.path.join(CSRC, "host"), cli_src, "-o", CLI_PATH,
               "-L" + _HERE, "-lrc_hip", "-Wl,-rpath,$ORIGIN", "-lpthread"]
        if verbose:
            print(" ".join(cmd))
        subprocess.run(cmd, check=True, cwd=_ROOT)
    # last: the optional RCCL band-gather library
    gather_deps = [GATHER_SRC, os.path.join(_ROOT, "include", "rc_gather.h")]
    if force or _stale(GATHER_PATH, gather_deps):
        if os.path.exists(GATHER_PATH):
            os.remove(GATHER_PATH)
        rocm = os.path.dirname(os.path.dirname(os.path.realpath(hipcc_path())))
        cmd = [hipcc_path(), "-O2", "-std=c++17", "-fPIC", "-Wall", "-x", "hip", "--offload-arch=gfx950",
               "-I" + os.path.join(_ROOT, "include"), "-shared", "-o", GATHER_PATH + ".tmp", GATHER_SRC,
               "-L" + os.path.join(rocm, "lib"), "-lrccl"]
        if verbose:
            print(" ".join(cmd))
        try:
            subprocess.run(cmd, check=True, cwd=_ROOT)
            os.replace(GATHER_PATH + ".tmp", GATHER_PATH)
        except subprocess.CalledProcessError as e:
            # the band gather is optional (gather.load() reports a missing library): a failed RCCL link must not
            # take the core library and the CLI, already built above, down with it
            print(f"warning: librc_gather.so was not built ({e}); --gather-impl c will be unavailable")
        finally:
            if os.path.exists(GATHER_PATH + ".tmp"):
                os.remove(GATHER_PATH + ".tmp")
    return LIB_PATH


if __name__ == "__main__":
    import sys
    build(force="--force" in sys.argv, verbose=True)
    print(LIB_PATH)
