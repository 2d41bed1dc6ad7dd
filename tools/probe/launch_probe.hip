// launch_probe.hip -- what a launch and a store-only pass cost on this GPU: the floor under the
// small-frame configurations (640x480 x 8, 1920x1080 x 64), whose draw kernels are a few microseconds long.
//
//   hipcc --offload-arch=gfx950 -O3 tools/probe/launch_probe.hip -o tools/probe/launch_probe && tools/probe/launch_probe
//
// Every figure is the average over back-to-back launches on one stream between two HIP events, after a
// warm-up long enough for the clock ramp: the same way bench.py times a frame.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef float rc_f4 __attribute__((ext_vector_type(4)));

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void k_empty() {}

// The draw kernel's store pattern: a 256-thread workgroup owns a (16P x 16) pixel tile, a wave an (8P x 8) tile,
// a lane P pixels of one row, 8 columns apart; float4 per pixel.  NT: nontemporal stores.
template <int P, bool NT>
__global__ __launch_bounds__(256) void k_store(float4 *surface, int width, int rows, int tiles_x)
{
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, lx = lane & 7, ly = lane >> 3;
    const int tyi = blockIdx.x / tiles_x, txi = blockIdx.x - tyi * tiles_x;
    const int x0 = txi * 16 * P + (wave & 1) * 8 * P + lx;
    const int row = tyi * 16 + (wave >> 1) * 8 + ly;
    if (row >= rows) return;
#pragma unroll
    for (int p = 0; p < P; p++) {
        const int x = x0 + 8 * p;
        if (x < width) {
            float4 *dst = surface + (size_t)row * width + x;
            const float4 v = make_float4(0.0f, 0.0f, 0.0f, 1.0f);
            if (NT) __builtin_nontemporal_store(rc_f4{0.0f, 0.0f, 0.0f, 1.0f}, reinterpret_cast<rc_f4 *>(dst));
            else *dst = v;
        }
    }
}

// Row-contiguous variant: a wave writes 64 consecutive pixels (1 KiB) of one row per store instruction.
template <int P, bool NT>
__global__ __launch_bounds__(256) void k_store_rows(float4 *surface, int width, int rows, int tiles_x)
{
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int tyi = blockIdx.x / tiles_x, txi = blockIdx.x - tyi * tiles_x;
    const int x = txi * 64 + lane;
    if (x >= width) return;
#pragma unroll
    for (int p = 0; p < P; p++) {
        const int row = tyi * 4 * P + wave * P + p;
        if (row < rows) {
            float4 *dst = surface + (size_t)row * width + x;
            const float4 v = make_float4(0.0f, 0.0f, 0.0f, 1.0f);
            if (NT) __builtin_nontemporal_store(rc_f4{0.0f, 0.0f, 0.0f, 1.0f}, reinterpret_cast<rc_f4 *>(dst));
            else *dst = v;
        }
    }
}

template <class F>
static double timed_us(F launch, int warm, int reps)
{
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int i = 0; i < warm; i++) launch();
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0, 0));
    for (int i = 0; i < reps; i++) launch();
    CHECK(hipEventRecord(e1, 0));
    CHECK(hipEventSynchronize(e1));
    float ms = 0.0f;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    CHECK(hipGetLastError());
    CHECK(hipEventDestroy(e0));
    CHECK(hipEventDestroy(e1));
    return ms * 1e3 / reps;
}

int main()
{
    float4 *surface = nullptr;
    CHECK(hipMalloc(&surface, (size_t)3840 * 2160 * 16));
    const int warm = 20000, reps = 4000;
    printf("empty kernel, 1 workgroup of 64:        %7.2f us per launch\n", timed_us([&] { hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, 0); }, warm, reps));
    printf("empty kernel, 2040 workgroups of 256:   %7.2f us per launch\n", timed_us([&] { hipLaunchKernelGGL(k_empty, dim3(2040), dim3(256), 0, 0); }, 2000, reps));
    printf("empty kernel, 600 workgroups of 256:    %7.2f us per launch\n", timed_us([&] { hipLaunchKernelGGL(k_empty, dim3(600), dim3(256), 0, 0); }, 2000, reps));
    const int sizes[3][2] = {{640, 480}, {1920, 1080}, {3840, 2160}};
    for (auto &s : sizes) {
        const int w = s[0], h = s[1];
        const double mb = (double)w * h * 16 / 1e6;
        auto tiles = [&](int p) { return ((w + 16 * p - 1) / (16 * p)) * ((h + 15) / 16); };
        auto tx = [&](int p) { return (w + 16 * p - 1) / (16 * p); };
        double t;
#define RUN(NAME, KERNEL, GRID, TX)                                                                         \
        t = timed_us([&] { hipLaunchKernelGGL(KERNEL, dim3(GRID), dim3(256), 0, 0, surface, w, h, TX); }, 3000, reps); \
        printf("%4dx%-4d %-34s %6d wg  %7.2f us  %7.1f TB/s\n", w, h, NAME, GRID, t, mb / t * 1e-3 * 1e6 / 1e3);
        RUN("store only, tile P=2", (k_store<2, false>), tiles(2), tx(2));
        RUN("store only, tile P=4", (k_store<4, false>), tiles(4), tx(4));
        RUN("store only, tile P=8", (k_store<8, false>), tiles(8), tx(8));
        RUN("store only, tile P=2, nontemporal", (k_store<2, true>), tiles(2), tx(2));
        RUN("store only, tile P=4, nontemporal", (k_store<4, true>), tiles(4), tx(4));
        RUN("store only, tile P=8, nontemporal", (k_store<8, true>), tiles(8), tx(8));
        const int rtx = (w + 63) / 64;
        RUN("store only, rows P=4", (k_store_rows<4, false>), rtx * ((h + 15) / 16), rtx);
        RUN("store only, rows P=4, nontemporal", (k_store_rows<4, true>), rtx * ((h + 15) / 16), rtx);
        RUN("store only, rows P=1", (k_store_rows<1, false>), rtx * ((h + 3) / 4), rtx);
    }
    CHECK(hipFree(surface));
    return 0;
}
