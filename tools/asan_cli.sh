#!/bin/bash
# tools/asan_cli.sh -- host-side AddressSanitizer build of the whole library + CLI in one binary
# (tools/bin/rc_render_asan; GPU code is not instrumented).  Build here, run on the GPU box:
#   ASAN_OPTIONS=detect_leaks=0 tools/bin/rc_render_asan --width 333 --height 177 --spheres 100
#   ... --frames 6 --out-pattern DIR/f_%03d.ppm        (frame pipeline)
#   ... --script tools/example_input.txt               (scripted controls)
set -e
cd "$(dirname "$0")/.."
mkdir -p tools/bin
hipcc --offload-arch=gfx950 -O1 -g -std=c++17 -ffp-contract=off -fno-fast-math \
    -Xarch_host -fsanitize=address -Xarch_host -fno-omit-frame-pointer \
    -Iinclude -Icudaraytracing_amd/csrc -Icudaraytracing_amd/csrc/host \
    cudaraytracing_amd/csrc/rc_kernels.hip cudaraytracing_amd/csrc/rc_host.hip \
    -x hip cudaraytracing_amd/csrc/host/rc_render_main.cpp -o tools/bin/rc_render_asan
echo tools/bin/rc_render_asan
# the C example (context API) against the same instrumented library code
hipcc --offload-arch=gfx950 -O1 -g -std=c++17 -ffp-contract=off -fno-fast-math \
    -Xarch_host -fsanitize=address -Xarch_host -fno-omit-frame-pointer \
    -Iinclude -Icudaraytracing_amd/csrc -Icudaraytracing_amd/csrc/host \
    cudaraytracing_amd/csrc/rc_kernels.hip cudaraytracing_amd/csrc/rc_host.hip \
    -x hip examples/render_frame.c -o tools/bin/render_frame_asan
echo tools/bin/render_frame_asan
