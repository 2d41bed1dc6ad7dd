// rc_kernels.hip -- gfx950 (MI355X, CDNA4) kernels of the ray-casting hot path.
//
// What is computed is the reference's draw kernel (RayCasting/RayCasting/kernel.cu:69-112)
// with its device functions castScreenRay (:44-50), findIntersection (:52-67) and
// getSphereColor (:21-42), in IEEE binary32 with every product and sum rounded separately
// (no FMA contraction), correctly rounded sqrt and division.  How it is computed is not:
//
//   rc_draw_tiled<P>  (default)  one wave64 owns an (8P x 8) pixel tile, P pixels per lane.
//     Per workgroup the sphere list is reduced ONCE to a 12-byte record per sphere, staged
//     in LDS and read back as wave-wide broadcasts.  For each (pixel, sphere) pair the
//     inner loop evaluates a conservative FMA filter |v^ . d| >= 1 (3 VALU ops + half a
//     v_max3 per test); only pairs the filter cannot rule out are re-tested with the
//     reference's exact, separately rounded arithmetic.  The filter can only say "maybe",
//     so surfaces are bit-identical to the unfiltered evaluation (soundness argument at
//     rc_prepass_sphere below).
//   rc_draw_literal   one pixel per lane, every pair evaluated exactly as kernel.cu:52-67.
//     Kept as the in-library cross-check and as the "straight translation" speed baseline.
//
// The two sinf() calls per pixel (kernel.cu:46-49) depend only on x and only on y, so the
// host tabulates them with libm (the function the CPU path calls) and the kernels read
// sin_x[x], sin_y[y]: ray directions are then bit-identical between CPU and GPU.
//
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off (the pragma below repeats it).

#include <hip/hip_runtime.h>
#include <stdint.h>
#include "rc_kernels.h"

#pragma clang fp contract(off)

namespace {

constexpr int kWave = 64;
constexpr int kBlock = 256;            // 4 waves, one per SIMD; >= 2 workgroups per CU
constexpr int kGroup = 4;              // spheres per filter group (48 bytes of LDS)
constexpr int kChunk = RC_LDS_CHUNK;   // spheres staged per pass

// ---------------------------------------------------------------------------------------
// Exact arithmetic (reference order of operations)
// ---------------------------------------------------------------------------------------

// kernel.cu:44-50 with sinf(xAngle), sinf(yAngle) supplied: forward + (right*sx)*aspect
// + up*sy, then vec3::norm (vec3.h:168-176,89-92: ONE reciprocal, three multiplies).
__device__ __forceinline__ void rc_ray_dir(const RcKernArgs &a, float sx, float sy,
                                           float &dx, float &dy, float &dz)
{
    float tx = a.rx * sx, ty = a.ry * sx, tz = a.rz * sx;
    tx = tx * a.aspect; ty = ty * a.aspect; tz = tz * a.aspect;
    float qx = a.fx + tx, qy = a.fy + ty, qz = a.fz + tz;
    float wx = a.ux * sy, wy = a.uy * sy, wz = a.uz * sy;
    qx = qx + wx; qy = qy + wy; qz = qz + wz;
    float len = sqrtf((qx * qx + qy * qy) + qz * qz);
    float inv = 1.0f / len;
    dx = qx * inv; dy = qy * inv; dz = qz * inv;
}

// Per-sphere terms of kernel.cu:53-58 that do not depend on the ray (all primary rays
// share the origin): v = o - pos, 2v (exact), 4c (exact scaling of c = v.v - r*r).
struct RcExactSphere { float tvx, tvy, tvz, c4; };

__device__ __forceinline__ RcExactSphere rc_exact_sphere(const RcKernArgs &a, int i)
{
    float px = a.position[3 * i + 0], py = a.position[3 * i + 1], pz = a.position[3 * i + 2];
    float r = a.radius[i];
    float vx = a.ox - px, vy = a.oy - py, vz = a.oz - pz;
    float c = ((vx * vx + vy * vy) + vz * vz) - (r * r);
    RcExactSphere e;
    e.tvx = 2.0f * vx; e.tvy = 2.0f * vy; e.tvz = 2.0f * vz;
    e.c4 = 4.0f * c;
    return e;
}

// kernel.cu:54,58-64 and the update at :91-94.
__device__ __forceinline__ void rc_exact_test(const RcExactSphere &e, float dx, float dy, float dz,
                                              int i, float &minDist, int &idx)
{
    float b = (e.tvx * dx + e.tvy * dy) + e.tvz * dz;
    float disc = b * b - e.c4;
    if (disc >= 0.0f) {
        float tmp = sqrtf(disc);
        float dist = ((-b - tmp) / 2.0f) - 0.000001f;
        if (dist < 0.0f) dist = ((-b + tmp) / 2.0f) - 0.000001f;
        if (dist > 0.0f && dist < minDist) { minDist = dist; idx = i; }
    }
}

// kernel.cu:21-42 (+ the hit point of :99).  The three light directions are
// vec3(1,0,-.5), (-1,0,-.5), (0,1,-.5) normalised in binary32.
__device__ __forceinline__ void rc_shade(const RcKernArgs &a, int idx, float minDist,
                                         float dx, float dy, float dz,
                                         float &r, float &g, float &b)
{
    float px = a.ox + dx * minDist, py = a.oy + dy * minDist, pz = a.oz + dz * minDist;
    float nx = px - a.position[3 * idx + 0];
    float ny = py - a.position[3 * idx + 1];
    float nz = pz - a.position[3 * idx + 2];
    float c = 0.2f;

    float inv = 1.0f / sqrtf((1.0f * 1.0f + 0.0f * 0.0f) + (-0.5f) * (-0.5f));
    float lx = 1.0f * inv, ly = 0.0f * inv, lz = -0.5f * inv;
    float diff = fmaxf((lx * nx + ly * ny) + lz * nz, 0.0f);
    c = c + diff * 0.5f;      // (float)(diff * 0.5 in double) == diff * 0.5f, exactly

    lx = -1.0f * inv; ly = 0.0f * inv; lz = -0.5f * inv;
    diff = fmaxf((lx * nx + ly * ny) + lz * nz, 0.0f);
    c = c + diff * 0.5f;

    lx = 0.0f * inv; ly = 1.0f * inv; lz = -0.5f * inv;
    diff = fmaxf((lx * nx + ly * ny) + lz * nz, 0.0f);
    c = c + diff * 0.5f;

    r = a.color[3 * idx + 0] * c;
    g = a.color[3 * idx + 1] * c;
    b = a.color[3 * idx + 2] * c;
}

// D3D float -> UNORM8: NaN -> 0, clamp [0,1], *255, +0.5, truncate (DXManager.cpp:134,328).
__device__ __forceinline__ unsigned rc_unorm8(float f)
{
    if (!(f == f)) return 0u;
    if (f <= 0.0f) return 0u;
    if (f >= 1.0f) return 255u;
    return (unsigned)(int)(f * 255.0f + 0.5f);
}

__device__ __forceinline__ void rc_store_pixel(const RcKernArgs &a, int x, int row,
                                               float r, float g, float b)
{
    // kernel.cu:78: (float*)((char*)surface + y*pitch) + 4*x ; alpha at :83
    float4 *p = reinterpret_cast<float4 *>(reinterpret_cast<char *>(a.surface) + (size_t)row * a.pitch) + x;
    *p = make_float4(r, g, b, 1.0f);
    if (a.rgba8) {
        int orow = a.rgba8_flip ? (a.rows - 1 - row) : row;
        unsigned v = rc_unorm8(r) | (rc_unorm8(g) << 8) | (rc_unorm8(b) << 16) | (255u << 24);
        *reinterpret_cast<unsigned *>(reinterpret_cast<char *>(a.rgba8) + (size_t)orow * a.rgba8_pitch + 4 * (size_t)x) = v;
    }
}

// ---------------------------------------------------------------------------------------
// Literal kernel: one pixel per lane, 64 x 4 pixel workgroups
// ---------------------------------------------------------------------------------------
} // namespace

__global__ __launch_bounds__(256) void rc_draw_literal(RcKernArgs a)
{
    int x = blockIdx.x * 64 + (threadIdx.x & 63);
    int row = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= a.width || row >= a.rows) return;

    float dx, dy, dz;
    rc_ray_dir(a, a.sin_x[x], a.sin_y[row + a.y_offset], dx, dy, dz);

    float minDist = 10000.0f;
    int idx = -1;
    for (int i = 0; i < a.n; i++) {
        RcExactSphere e = rc_exact_sphere(a, i);
        rc_exact_test(e, dx, dy, dz, i, minDist, idx);
    }
    float r = 0.0f, g = 0.0f, b = 0.0f;
    if (idx >= 0) rc_shade(a, idx, minDist, dx, dy, dz, r, g, b);
    rc_store_pixel(a, x, row, r, g, b);
}

namespace {
// ---------------------------------------------------------------------------------------
// Tiled kernel
// ---------------------------------------------------------------------------------------

// Filter record of one sphere: v^ = v * k with k = 1/sqrt(c - m), m = 2^-17 * (v.v).
//
// Soundness (u = 2^-24; L = |v|; A = sum |v_i d_i| <= L|d|; |d|^2 < 1 + 2^-19 checked per
// lane).  The exact test hits only if fl(beta^2) >= c with beta = b/2 the separately
// rounded dot product, so beta^2 >= c(1-u).  The filter value F = fma(v^z,dz,fma(v^y,dy,
// v^x*dx)) satisfies F = k(t + e) with t the real dot product, |e| <= 4.01uA, and
// |beta - t| <= 3.01uA, hence F^2 >= k^2 (c - 15.3u L^2).  With k^2 = (1+-8.1u)/(c - m)
// and m = 128u (v.v) >= 127u L^2 this gives F^2 >= 1 whenever the exact test hits.
// Spheres for which that argument does not apply (c - m <= 0: camera inside or within
// 2^-17 of the surface; |v|^2 outside [2^-100, 2^100]; NaN/Inf anywhere; v^ not < 2^60)
// are flagged "always": their record is zero and they are always re-tested exactly.
__device__ __forceinline__ void rc_prepass_sphere(const RcKernArgs &a, int i,
                                                  float &hx, float &hy, float &hz, bool &always)
{
    float px = a.position[3 * i + 0], py = a.position[3 * i + 1], pz = a.position[3 * i + 2];
    float r = a.radius[i];
    float vx = a.ox - px, vy = a.oy - py, vz = a.oz - pz;
    float vv = (vx * vx + vy * vy) + vz * vz;
    float c = vv - r * r;
    float cm = c - vv * 0x1p-17f;
    always = !(cm > 0.0f) || !(vv < 0x1p100f) || !(vv > 0x1p-100f);
    float k = 1.0f / sqrtf(cm);
    hx = vx * k; hy = vy * k; hz = vz * k;
    always = always || !(fabsf(hx) < 0x1p60f) || !(fabsf(hy) < 0x1p60f) || !(fabsf(hz) < 0x1p60f);
    if (always) { hx = 0.0f; hy = 0.0f; hz = 0.0f; }
}

template <int P>
__device__ __forceinline__ void rc_tiled_body(const RcKernArgs &a, float4 *s_rec,
                                              unsigned char *s_flag, int *s_has_always)
{
    constexpr int TW = 8 * P;   // wave tile width in pixels
    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const int lx = lane & 7, ly = lane >> 3;
    // workgroup tile: 2 x 2 waves = (16P) x 16 pixels
    const int x0 = blockIdx.x * (2 * TW) + (wave & 1) * TW + lx;
    const int row = blockIdx.y * 16 + (wave >> 1) * 8 + ly;
    const int rowc = min(row, a.rows - 1);

    float dx[P], dy[P], dz[P], minDist[P];
    int idx[P];
    float force = 0.0f;
    const float sy = a.sin_y[rowc + a.y_offset];
#pragma unroll
    for (int p = 0; p < P; p++) {
        int xc = min(x0 + 8 * p, a.width - 1);
        rc_ray_dir(a, a.sin_x[xc], sy, dx[p], dy[p], dz[p]);
        float dd = (dx[p] * dx[p] + dy[p] * dy[p]) + dz[p] * dz[p];
        // directions that are not unit vectors (degenerate camera) bypass the filter
        if (!(dd < 1.0f + 0x1p-19f) || !(dd > 1.0f - 0x1p-19f)) force = __builtin_inff();
        minDist[p] = 10000.0f;
        idx[p] = -1;
    }
    const bool lane_forced = force > 0.0f;

    for (int base = 0; base < a.n; base += kChunk) {
        const int cn = min(a.n - base, kChunk);
        const int cn4 = (cn + kGroup - 1) & ~(kGroup - 1);
        if (tid == 0) *s_has_always = 0;
        __syncthreads();
        {
            float *rec = reinterpret_cast<float *>(s_rec);
            for (int j = tid; j < cn4; j += kBlock) {
                float hx = 0.0f, hy = 0.0f, hz = 0.0f;
                bool always = false;
                if (j < cn) rc_prepass_sphere(a, base + j, hx, hy, hz, always);
                rec[3 * j + 0] = hx; rec[3 * j + 1] = hy; rec[3 * j + 2] = hz;
                s_flag[j] = always ? 1 : 0;
                if (always) *s_has_always = 1;
            }
        }
        __syncthreads();
        const float m0 = (*s_has_always != 0) ? __builtin_inff() : force;
        const int ngroups = cn4 / kGroup;

        for (int g = 0; g < ngroups; g++) {
            const float4 q0 = s_rec[3 * g + 0], q1 = s_rec[3 * g + 1], q2 = s_rec[3 * g + 2];
            const float hx[kGroup] = {q0.x, q0.w, q1.z, q2.y};
            const float hy[kGroup] = {q0.y, q1.x, q1.w, q2.z};
            const float hz[kGroup] = {q0.z, q1.y, q2.x, q2.w};
            float f[kGroup][P];
            float m = m0;
#pragma unroll
            for (int s = 0; s < kGroup; s++) {
#pragma unroll
                for (int p = 0; p < P; p++) {
                    f[s][p] = __builtin_fmaf(hz[s], dz[p], __builtin_fmaf(hy[s], dy[p], hx[s] * dx[p]));
                    m = fmaxf(m, fabsf(f[s][p]));
                }
            }
            if (__builtin_amdgcn_ballot_w64(m >= 1.0f) != 0ull) {
                // Rare path: some lane cannot rule out some sphere of this group.
#pragma unroll
                for (int s = 0; s < kGroup; s++) {
                    const int j = kGroup * g + s;
                    if (j >= cn) break;
                    bool cand = lane_forced || (s_flag[j] != 0);
#pragma unroll
                    for (int p = 0; p < P; p++) cand = cand || (fabsf(f[s][p]) >= 1.0f);
                    if (__builtin_amdgcn_ballot_w64(cand) == 0ull) continue;
                    const RcExactSphere e = rc_exact_sphere(a, base + j);
#pragma unroll
                    for (int p = 0; p < P; p++)
                        rc_exact_test(e, dx[p], dy[p], dz[p], base + j, minDist[p], idx[p]);
                }
            }
        }
        __syncthreads();
    }

    if (row < a.rows) {
#pragma unroll
        for (int p = 0; p < P; p++) {
            const int x = x0 + 8 * p;
            if (x < a.width) {
                float r = 0.0f, g = 0.0f, b = 0.0f;
                if (idx[p] >= 0) rc_shade(a, idx[p], minDist[p], dx[p], dy[p], dz[p], r, g, b);
                rc_store_pixel(a, x, row, r, g, b);
            }
        }
    }
}

} // namespace

template <int P>
__global__ __launch_bounds__(256) void rc_draw_tiled(RcKernArgs a)
{
    __shared__ float4 s_rec[kChunk / kGroup * 3];
    __shared__ unsigned char s_flag[kChunk];
    __shared__ int s_has_always;
    rc_tiled_body<P>(a, s_rec, s_flag, &s_has_always);
}

// float RGBA -> RGBA8 (the display step that follows the path in the reference)
__global__ __launch_bounds__(256) void rc_rgba8_kernel(const char *surface, size_t pitch, int width, int rows,
                                                       char *out, size_t out_pitch, int flip)
{
    int x = blockIdx.x * 64 + (threadIdx.x & 63);
    int row = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= width || row >= rows) return;
    float4 v = *(reinterpret_cast<const float4 *>(surface + (size_t)row * pitch) + x);
    int orow = flip ? (rows - 1 - row) : row;
    unsigned o = rc_unorm8(v.x) | (rc_unorm8(v.y) << 8) | (rc_unorm8(v.z) << 16) | (rc_unorm8(v.w) << 24);
    *reinterpret_cast<unsigned *>(out + (size_t)orow * out_pitch + 4 * (size_t)x) = o;
}

// ---------------------------------------------------------------------------------------
// Launchers (host)
// ---------------------------------------------------------------------------------------

hipError_t rc_launch_draw(const RcKernArgs &a, int variant, hipStream_t stream)
{
    if (variant == RC_KVARIANT_LITERAL) {
        dim3 grid((a.width + 63) / 64, (a.rows + 3) / 4);
        hipLaunchKernelGGL(rc_draw_literal, grid, dim3(256), 0, stream, a);
    } else if (variant == RC_KVARIANT_TILED2) {
        dim3 grid((a.width + 31) / 32, (a.rows + 15) / 16);
        hipLaunchKernelGGL(rc_draw_tiled<2>, grid, dim3(256), 0, stream, a);
    } else {
        dim3 grid((a.width + 63) / 64, (a.rows + 15) / 16);
        hipLaunchKernelGGL(rc_draw_tiled<4>, grid, dim3(256), 0, stream, a);
    }
    return hipGetLastError();
}

hipError_t rc_launch_rgba8(const void *surface, size_t pitch, int width, int rows, void *out,
                           size_t out_pitch, int flip, hipStream_t stream)
{
    dim3 grid((width + 63) / 64, (rows + 3) / 4);
    hipLaunchKernelGGL(rc_rgba8_kernel, grid, dim3(256), 0, stream,
                       reinterpret_cast<const char *>(surface), pitch, width, rows,
                       reinterpret_cast<char *>(out), out_pitch, flip);
    return hipGetLastError();
}
