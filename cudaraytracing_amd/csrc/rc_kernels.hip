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
//     inner loop evaluates a conservative, separable FMA filter (one FMA shared by the
//     lane's P pixels + one FMA + half a v_max3 per test); only pairs the filter cannot
//     rule out are re-tested with the reference's exact, separately rounded arithmetic.
//     The filter can only say "maybe", so surfaces are bit-identical to the unfiltered
//     evaluation (soundness argument at rc_prepass_sphere below).
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

// Work counters of an instrumented build (hipcc -DRC_COUNT_TESTS, tools/count_tests.py; never in the shipped
// library): lane 0 of a wave adds to a device-global tally at the wave-uniform points marked RC_COUNT below.
// Slots: 0 primary blocks of 32 spheres run (x 64 lanes x P pixels x 32 = tests), 1 primary blocks flagged,
// 2 spheres visited (per-pixel filter repeated), 3 pixel columns x spheres given the exact test (x 64 lanes),
// 4 shadow blocks run, 5 shadow blocks flagged, 6 shadow spheres visited, 7 shadow spheres given the exact test
// (x P columns x 64 lanes), 8 waves, 9 waves with a shadow ray, 10 waves that left the shadow loop early.
#ifdef RC_COUNT_TESTS
__device__ unsigned long long rc_counters[16];
#define RC_COUNT(slot, n) do { if ((threadIdx.x & 63) == 0) atomicAdd(&rc_counters[slot], (unsigned long long)(n)); } while (0)
extern "C" int rc_counters_fetch(unsigned long long *host16, int reset)
{
    if (hipMemcpyFromSymbol(host16, HIP_SYMBOL(rc_counters), sizeof(rc_counters)) != hipSuccess) return -1;
    if (reset) {
        const unsigned long long zero[16] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(rc_counters), zero, sizeof(zero)) != hipSuccess) return -1;
    }
    return 0;
}
#else
#define RC_COUNT(slot, n) do { } while (0)
#endif

namespace {

constexpr int kBlock = 256;            // 4 waves, one per SIMD; >= 2 workgroups per CU
constexpr int kGroup = 4;              // spheres per filter group (48 bytes of LDS)

// ---------------------------------------------------------------------------------------
// Exact arithmetic (reference order of operations)
// ---------------------------------------------------------------------------------------

// Correctly rounded square root and reciprocal (what sqrtf(x) and 1.0f / x mean in IEEE binary32), shorter than
// the compiler's general sequences because the operand is first checked to be a normal number well inside the
// exponent range; anything else (zero, denormals, huge, tiny, infinities, NaN) takes the general sequence.
//
// rc_sqrt_core: the hardware's v_sqrt_f32 is within one ulp, so the correctly rounded root is s, its predecessor or
// its successor; the exact residuals x - (s -+ 1ulp) s (one FMA each, no rounding in the product) decide which --
// the core of the compiler's own expansion, minus its rescaling of small operands.
// rc_rcp_core: one Newton step from v_rcp_f32 (within one ulp), r' = r + r (1 - x r) with the residual exact in an FMA,
// then a second one: the first brings r within half an ulp plus 2^-46 relative, the second rounds correctly.
// Both are compared against the general sequences for EVERY binary32 mantissa over the whole admitted exponent
// range on the device itself (rc_selftest_math, tests/test_parity_gpu.py::test_fast_sqrt_rcp_exhaustive).
__device__ __forceinline__ float rc_sqrt_core(float x)
{
    const float s = __builtin_amdgcn_sqrtf(x);
    const float sd = __builtin_bit_cast(float, __builtin_bit_cast(int, s) - 1);
    const float su = __builtin_bit_cast(float, __builtin_bit_cast(int, s) + 1);
    const float rd = __builtin_fmaf(-sd, s, x);
    const float ru = __builtin_fmaf(-su, s, x);
    float r = (rd <= 0.0f) ? sd : s;
    r = (ru > 0.0f) ? su : r;
    return r;
}

__device__ __forceinline__ float rc_rcp_core(float x)
{
    float r = __builtin_amdgcn_rcpf(x);
    r = __builtin_fmaf(__builtin_fmaf(-x, r, 1.0f), r, r);
    r = __builtin_fmaf(__builtin_fmaf(-x, r, 1.0f), r, r);
    return r;
}

__device__ __forceinline__ float rc_rcp(float x)
{
    // wave-uniform choice: one scalar branch, no lane ever takes both paths
    const float ax = fabsf(x);
    if (__builtin_amdgcn_ballot_w64(!(ax >= 0x1p-96f && ax < 0x1p96f)) == 0ull) return rc_rcp_core(x);
    asm volatile("" ::: "memory");
    return 1.0f / x;
}

// root = sqrtf(x) and the returned 1.0f / root, both correctly rounded, behind ONE range check: for x in
// [2^-96, 2^96) the root lies in [2^-48, 2^48], well inside the reciprocal's admitted range.
__device__ __forceinline__ float rc_sqrt_rcp(float x, float &root)
{
    if (__builtin_amdgcn_ballot_w64(!(x >= 0x1p-96f && x < 0x1p96f)) == 0ull) {
        root = rc_sqrt_core(x);
        return rc_rcp_core(root);
    }
    asm volatile("" ::: "memory");
    root = sqrtf(x);
    return 1.0f / root;
}

// vec3::norm (vec3.h:168-176,89-92) of the un-normalised ray q of kernel.cu:49, in place: ONE reciprocal,
// three multiplies.  Returns |q| as the reference rounds it.
__device__ __forceinline__ float rc_normalise(float &x, float &y, float &z)
{
    float len;
    const float inv = rc_sqrt_rcp((x * x + y * y) + z * z, len);
    x = x * inv; y = y * inv; z = z * inv;
    return len;
}

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
    rc_normalise(qx, qy, qz);
    dx = qx; dy = qy; dz = qz;
}

// Per-sphere terms of kernel.cu:53-58 that do not depend on the ray (all primary rays
// share the origin): v = o - pos, 2v (exact), 4c (exact scaling of c = v.v - r*r).
struct RcExactSphere { float tvx, tvy, tvz, c4, near; };

// Lower bound of every POSITIVE distance kernel.cu:52-67 can return for a sphere with v = o - pos,
// vv = v.v and radius r, for any ray direction the reference normalises (|d| within 4.5u of 1,
// u = 2^-24).  A pixel whose minDist is already <= this bound cannot be updated by the sphere
// (kernel.cu:91 needs dist < minDist), so its exact test can be skipped without changing a bit.
// Derivation, camera outside the sphere (c = vv - r^2 > 0).  In real numbers the first root
// t1 = B - sqrt(B^2 - c), B = -(v.d), decreases in B and B <= |v||d|, so t1 >= |v| - r.  In the
// reference's float evaluation (gamma_3 bounds for the three-term dot products):
//   B~ = -b/2 is within 3.01u |v||d| of B, hence B~ <= |v|(1 + 7.6u);
//   c~ is within 5.1u vv of c;  disc~/4 = (B~^2 (1 + d1) - c~)(1 + d2);  s~ = sqrt(disc~)(1 + d3);
//   so (s~/2)^2 <= B~^2 - C' with C' >= c - 9.2u vv, and  B~ - s~/2 >= g(B~) = B~ - sqrt(B~^2 - C') >= g(|v|(1 + 7.6u));
//   there B~^2 - C' <= r^2 + 24.5u vv, whose root is <= r + 12.25u vv/r;
//   the roundings of -b - s, the exact halving and the subtraction of 1e-6f cost another 2.01u |v|:
//      dist >= |v| - r - 12.25u vv/r - 2.01u|v| - 1.0000001e-6 .
// The bound below subtracts 32u vv/r + 64u|v| + 2e-6 instead, which also covers its own float
// evaluation (hardware sqrt and reciprocal: it can overestimate |v| - r by <= 7u|v|).  When the first
// root comes out negative the reference returns the second; then this bound is negative too and never
// skips anything (minDist is always positive).  Spheres behind the camera return no positive distance
// while the bound is positive: their larger root is <= -(|v| - r)/2, more than sixteen times the error
// bound away from zero.  Camera inside or at the sphere, r = 0, overflow, non-finite data: the bound is
// negative, -inf or NaN, and the caller's test  minDist <= bound  is false.
__device__ __forceinline__ float rc_near_bound(float vv, float r)
{
    const float L = __builtin_amdgcn_sqrtf(vv);
    const float ar = fabsf(r);
    return ((L - ar) - 0x1p-19f * (vv * __builtin_amdgcn_rcpf(ar))) - (0x1p-18f * L + 2.0e-6f);
}

__device__ __forceinline__ RcExactSphere rc_exact_sphere_from(const RcKernArgs &a, int i,
                                                             float ox, float oy, float oz)
{
    float px = a.position[3 * i + 0], py = a.position[3 * i + 1], pz = a.position[3 * i + 2];
    float r = a.radius[i];
    float vx = ox - px, vy = oy - py, vz = oz - pz;
    float vv = (vx * vx + vy * vy) + vz * vz;
    float c = vv - (r * r);
    RcExactSphere e;
    e.tvx = 2.0f * vx; e.tvy = 2.0f * vy; e.tvz = 2.0f * vz;
    e.c4 = 4.0f * c;
    e.near = rc_near_bound(vv, r);
    return e;
}

__device__ __forceinline__ RcExactSphere rc_exact_sphere(const RcKernArgs &a, int i)
{
    return rc_exact_sphere_from(a, i, a.ox, a.oy, a.oz);
}

// kernel.cu:54,58-64: the distance findIntersection returns (-1 for a miss).
__device__ __forceinline__ float rc_exact_dist(const RcExactSphere &e, float dx, float dy, float dz)
{
    float b = (e.tvx * dx + e.tvy * dy) + e.tvz * dz;
    float disc = b * b - e.c4;
    float dist = -1.0f;
    if (disc >= 0.0f) {
        float tmp = sqrtf(disc);
        dist = ((-b - tmp) / 2.0f) - 0.000001f;
        if (dist < 0.0f) dist = ((-b + tmp) / 2.0f) - 0.000001f;
    }
    return dist;
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

// ---------------------------------------------------------------------------------------
// EXT mode: plane + one point light + Phong + hard shadows.  The reference has none of it;
// the arithmetic below is this library's own definition, written with the reference's
// building blocks (findIntersection, vec3::norm, vec3::reflect at vec3.h:163-166, the 1e-6
// offset and the strict 0 < t < minDist rule), one rounding per operation.
// ---------------------------------------------------------------------------------------
constexpr int kPlaneIdx = -2;

__device__ __forceinline__ float rc_dot3(float ax, float ay, float az, float bx, float by, float bz)
{
    return (ax * bx + ay * by) + az * bz;                   // vec3.h:153-156
}

// Plane n.x = h tested after the spheres: t = ((h - n.o) / (n.d)) - 1e-6f.
__device__ __forceinline__ void rc_plane_test(const RcKernArgs &a, float dx, float dy, float dz,
                                              float &minDist, int &idx)
{
    if (!a.has_plane) return;
    float denom = rc_dot3(a.pnx, a.pny, a.pnz, dx, dy, dz);
    if (denom != 0.0f) {
        float t = (a.plane_num / denom) - 0.000001f;          // plane_num = plane_h - n.o (host, same statements)
        if (t > 0.0f && t < minDist) { minDist = t; idx = kPlaneIdx; }
    }
}

// Hit point, shadow-ray direction FROM the light, and the distance |light - point|.  The shadow test's distance
// limit is ldist - 0.001f (rc_shadow_limit); the shading divides by the same ldist again.
__device__ __forceinline__ void rc_ext_ray(const RcKernArgs &a, float minDist, float dx, float dy, float dz,
                                           float &ex, float &ey, float &ez, float &ldist)
{
    float px = a.ox + dx * minDist, py = a.oy + dy * minDist, pz = a.oz + dz * minDist;
    float lvx = a.lgx - px, lvy = a.lgy - py, lvz = a.lgz - pz;
    float lightDist;
    float invL = rc_sqrt_rcp((lvx * lvx + lvy * lvy) + lvz * lvz, lightDist);
    ex = (px - a.lgx) * invL; ey = (py - a.lgy) * invL; ez = (pz - a.lgz) * invL;
    ldist = lightDist;
}

// Pixels that cast no shadow ray carry ldist = -1, hence a negative limit.
__device__ __forceinline__ float rc_shadow_limit(float ldist) { return ldist - 0.001f; }

// Plane as an occluder of the shadow ray (not when the plane itself was hit).
__device__ __forceinline__ bool rc_plane_shadow(const RcKernArgs &a, int idx, float ex, float ey, float ez, float limit)
{
    if (!a.has_plane || idx == kPlaneIdx) return false;
    float de = rc_dot3(a.pnx, a.pny, a.pnz, ex, ey, ez);
    if (de == 0.0f) return false;
    float t = (a.plane_num_light / de) - 0.000001f;           // plane_num_light = plane_h - n.light (host)
    return t > 0.0f && t < limit;
}

// ldist: |light - point| as rc_ext_ray returned it for this pixel, or a negative value when no shadow ray was
// cast (shadows off): it is then evaluated here, by the same statements.
__device__ __forceinline__ void rc_ext_shade(const RcKernArgs &a, int idx, float minDist,
                                             float dx, float dy, float dz, bool shadowed, float ldist,
                                             float &r, float &g, float &b)
{
    float px = a.ox + dx * minDist, py = a.oy + dy * minDist, pz = a.oz + dz * minDist;
    float nx, ny, nz, cr, cg, cb, il;
    if (idx >= 0) {
        nx = px - a.position[3 * idx + 0]; ny = py - a.position[3 * idx + 1]; nz = pz - a.position[3 * idx + 2];
        float nlen;
        float inv = rc_sqrt_rcp((nx * nx + ny * ny) + nz * nz, nlen);
        nx = nx * inv; ny = ny * inv; nz = nz * inv;
        cr = a.color[3 * idx + 0]; cg = a.color[3 * idx + 1]; cb = a.color[3 * idx + 2];
        il = 0.0f;                                            // evaluated below, only if the pixel is lit
    } else {
        // the plane's unit normal and the reciprocal of ITS length are per-frame constants (host, same statements)
        nx = a.pux; ny = a.puy; nz = a.puz;
        if (rc_dot3(a.pnx, a.pny, a.pnz, dx, dy, dz) > 0.0f) { nx = nx * -1.0f; ny = ny * -1.0f; nz = nz * -1.0f; }
        cr = a.pcx; cg = a.pcy; cb = a.pcz;
        il = a.pu_il;
    }
    float c = 0.2f, spec = 0.0f;
    if (!shadowed) {
        float lvx = a.lgx - px, lvy = a.lgy - py, lvz = a.lgz - pz;
        float invL;
        if (ldist >= 0.0f) invL = rc_rcp(ldist);
        else invL = rc_sqrt_rcp((lvx * lvx + lvy * lvy) + lvz * lvz, ldist);
        float ldx = lvx * invL, ldy = lvy * invL, ldz = lvz * invL;
        float diff = fmaxf(rc_dot3(nx, ny, nz, ldx, ldy, ldz), 0.0f);
        c = c + diff * 0.5f;
        // Ldir.reflect(N): N * Ldir.dot(N) / N.length() / N.length() * 2 - Ldir  (vec3.h:163-166)
        float dn = rc_dot3(ldx, ldy, ldz, nx, ny, nz);
        float tx = nx * dn, ty = ny * dn, tz = nz * dn;
        if (idx >= 0) { float nl; il = rc_sqrt_rcp((nx * nx + ny * ny) + nz * nz, nl); }
        tx = tx * il; ty = ty * il; tz = tz * il;
        tx = tx * il; ty = ty * il; tz = tz * il;
        tx = tx * 2.0f; ty = ty * 2.0f; tz = tz * 2.0f;
        float rx = tx - ldx, ry = ty - ldy, rz = tz - ldz;
        float s = fmaxf(rc_dot3(rx, ry, rz, dx * -1.0f, dy * -1.0f, dz * -1.0f), 0.0f);
        for (int k = 0; k < a.shininess_log2; k++) s = s * s;
        spec = a.specular * s;
    }
    r = cr * c + spec; g = cg * c + spec; b = cb * c + spec;
}

typedef float rc_f4 __attribute__((ext_vector_type(4)));

// D3D float -> UNORM8: NaN -> 0, clamp [0,1], *255, +0.5, truncate (DXManager.cpp:134,328).
__device__ __forceinline__ unsigned rc_unorm8(float f)
{
    if (!(f == f)) return 0u;
    if (f <= 0.0f) return 0u;
    if (f >= 1.0f) return 255u;
    return (unsigned)(int)(f * 255.0f + 0.5f);
}

// Start of surface row `row` (kernel.cu:78: (char*)surface + y*pitch) and of the matching RGBA8 row (null
// without a second surface): formed once per lane, every pixel of the lane lies in this row.
struct RcRowPtr { char *f32; char *u8; };

__device__ __forceinline__ RcRowPtr rc_row_ptr(const RcKernArgs &a, int row)
{
    RcRowPtr rp;
    rp.f32 = reinterpret_cast<char *>(a.surface) + (size_t)row * a.pitch;
    rp.u8 = nullptr;
    if (a.rgba8) {
        const int orow = a.rgba8_flip ? (a.rows - 1 - row) : row;
        rp.u8 = reinterpret_cast<char *>(a.rgba8) + (size_t)orow * a.rgba8_pitch;
    }
    return rp;
}

__device__ __forceinline__ void rc_store_pixel(const RcRowPtr &rp, int x, float r, float g, float b)
{
    // kernel.cu:78: (float*)(row start) + 4*x ; alpha at :83
    // Nontemporal: the surface is written once and not read back by this kernel, so the lines stream out of
    // the L2 as they fill instead of waiting, dirty, for the write-back at the end of the kernel (measured on
    // one box, same run: 1920x1080 x 64 15.5 -> 13.3 us, 1280x768 x 512 37.4 -> 36.0 us, 3840x2160 x 1024
    // 345.5 -> 344.1 us; EXT 42.2 -> 40.5 us at 1920x1080 x 64).
    __builtin_nontemporal_store(rc_f4{r, g, b, 1.0f}, reinterpret_cast<rc_f4 *>(rp.f32) + x);
    if (rp.u8) {
        const unsigned v = rc_unorm8(r) | (rc_unorm8(g) << 8) | (rc_unorm8(b) << 16) | (255u << 24);
        *(reinterpret_cast<unsigned *>(rp.u8) + x) = v;
    }
}

// ---------------------------------------------------------------------------------------
// Literal kernel: one pixel per lane, 64 x 4 pixel workgroups
// ---------------------------------------------------------------------------------------
} // namespace

template <bool EXT>
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
    if (EXT) {
        rc_plane_test(a, dx, dy, dz, minDist, idx);
        if (idx != -1) {
            bool shadowed = false;
            float ldist = -1.0f;
            if (a.shadows) {
                float ex, ey, ez;
                rc_ext_ray(a, minDist, dx, dy, dz, ex, ey, ez, ldist);
                const float limit = rc_shadow_limit(ldist);
                for (int i = 0; i < a.n; i++) {
                    if (i == idx) continue;
                    RcExactSphere e = rc_exact_sphere_from(a, i, a.lgx, a.lgy, a.lgz);
                    float dist = rc_exact_dist(e, ex, ey, ez);
                    if (dist > 0.0f && dist < limit) shadowed = true;
                }
                if (rc_plane_shadow(a, idx, ex, ey, ez, limit)) shadowed = true;
            }
            rc_ext_shade(a, idx, minDist, dx, dy, dz, shadowed, ldist, r, g, b);
        }
    } else if (idx >= 0) {
        rc_shade(a, idx, minDist, dx, dy, dz, r, g, b);
    }
    rc_store_pixel(rc_row_ptr(a, row), x, r, g, b);
}

namespace {
// ---------------------------------------------------------------------------------------
// Tiled kernel
// ---------------------------------------------------------------------------------------

// Separable conservative filter.
//
// With a = sin(xa)*aspect and b = sin(ya) the un-normalised ray is q = F + a R + b U
// (kernel.cu:49) and d = q/|q|, so for a sphere with v = o - pos (kernel.cu:53)
//
//      -(v.d) * k  =  (alpha + b*gamma + a*beta) / |q|,      alpha = -k (v.F),  beta = -k (v.R),
//                                                             gamma = -k (v.U)
//
// alpha, beta, gamma depend on the sphere only (computed once per workgroup), a on the pixel
// column, b on the pixel row, len = |q| on the pixel.  A pixel can hit a sphere it is outside
// of only if -(v.d) >= sqrt(c) up to rounding (c = v.v - r*r, the discriminant test of
// kernel.cu:58-59 with b < 0; for b >= 0 and c > 0 both roots are negative and kernel.cu:91
// rejects them).  The filter flags the pair when
//
//      T = fma(a, beta, fma(b, gamma, alpha))  >=  len .
//
// Soundness (u = 2^-24, L = |v|, Q = |F| + |a||R| + |b||U| <= 4 len, checked per lane):
//  * the exact test hits only if -(v.d) >= sqrt(c)(1-u) - 3.01uL (c as the reference rounds it);
//  * the reference's float q, len, 1/len, d differ from the real ones by <= 4uQ, 2.5u, u, u
//    (relative), so -(v.q) >= len (sqrt(c)(1-2u) - 4.03uL);
//  * T differs from k(-(v.q)) by <= 12ukLQ <= 48uk L len: 5u for the float records (dot product
//    3u, k and its product 2u), u for a, 2u for the two fma roundings, 4u for q itself;
//  hence T >= k len (sqrt(c)(1-2u) - 52.1uL).  The record uses
//      k = 1 / thr,   thr = sqrt(c)(1 - 2^-19) - 2^-17 sqrt(v.v)
//  evaluated in binary32 with the hardware's 1-ulp sqrt and reciprocal (<= 8u error in all),
//  which is below sqrt(c)(1-5u) - 64uL, so k(...) >= 1 and T >= len whenever the exact test hits.
// Spheres where thr <= 0 or any quantity leaves the range in which these bounds hold (camera
// inside or within ~1e-5 relative of the surface, non-finite data, |v|^2 outside [2^-80, 2^80],
// |record| >= 2^40) are flagged "always" and re-tested exactly every time; lanes whose ray
// set-up is out of range (Q > 4 len, len outside [2^-20, 2^20], degenerate camera basis)
// bypass the filter too.
struct RcRecord { float alpha, gamma, beta; bool always; };

constexpr float kNever = -1.0e30f;   // record value that can never reach len

// centre and radius of one sphere as loaded from the scene arrays
struct RcSphereIn { float px, py, pz, r; };

__device__ __forceinline__ RcSphereIn rc_load_sphere(const RcKernArgs &a, int i)
{
    RcSphereIn s;
    s.px = a.position[3 * i + 0]; s.py = a.position[3 * i + 1]; s.pz = a.position[3 * i + 2];
    s.r = a.radius[i];
    return s;
}

__device__ __forceinline__ RcRecord rc_prepass_values(const RcKernArgs &a, const RcSphereIn &in, RcExactSphere *exact = nullptr)
{
    const float px = in.px, py = in.py, pz = in.pz, r = in.r;
    float vx = a.ox - px, vy = a.oy - py, vz = a.oz - pz;
    float vv = (vx * vx + vy * vy) + vz * vz;
    float c = vv - r * r;                                   // the reference's c, as rounded
    if (exact) {                                            // the same statements as rc_exact_sphere_from
        exact->tvx = 2.0f * vx; exact->tvy = 2.0f * vy; exact->tvz = 2.0f * vz;
        exact->c4 = 4.0f * c;
        exact->near = rc_near_bound(vv, r);
    }
    // hardware sqrt / reciprocal (1 ulp each) are enough here: the margins below absorb them
    float thr = __builtin_amdgcn_sqrtf(c) * (1.0f - 0x1p-19f) - 0x1p-17f * __builtin_amdgcn_sqrtf(vv);
    RcRecord rec;
    rec.always = !(c > 0.0f) || !(thr > 0.0f) || !(vv < 0x1p80f) || !(vv > 0x1p-80f);
    float k = __builtin_amdgcn_rcpf(thr);
    float vF = __builtin_fmaf(vz, a.fz, __builtin_fmaf(vy, a.fy, vx * a.fx));
    float vR = __builtin_fmaf(vz, a.rz, __builtin_fmaf(vy, a.ry, vx * a.rx));
    float vU = __builtin_fmaf(vz, a.uz, __builtin_fmaf(vy, a.uy, vx * a.ux));
    rec.alpha = -k * vF;
    rec.beta = -k * vR;
    rec.gamma = -k * vU;
    rec.always = rec.always || !(fabsf(rec.alpha) < 0x1p40f) || !(fabsf(rec.beta) < 0x1p40f) ||
                 !(fabsf(rec.gamma) < 0x1p40f);
    if (rec.always) { rec.alpha = kNever; rec.beta = 0.0f; rec.gamma = 0.0f; }
    return rec;
}

__device__ __forceinline__ RcRecord rc_prepass_sphere(const RcKernArgs &a, int i, RcExactSphere *exact = nullptr)
{
    return rc_prepass_values(a, rc_load_sphere(a, i), exact);
}

typedef float rc_f2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ rc_f2 rc_fma2(rc_f2 x, rc_f2 y, rc_f2 z) { return __builtin_elementwise_fma(x, y, z); }

constexpr int kBlockSpheres = 32;                       // spheres between two flag checks
constexpr int kGroupsPerBlock = kBlockSpheres / kGroup;

// LDS image of a chunk of `cap` spheres (cap a multiple of 16):
//   float4 rec[cap/4*3]   per group of 4 spheres {alpha0..3}{gamma0..3}{beta0..3}
//   uint   flag[cap/4]    per group, byte s = "always" flag of sphere s
//   int    has_always
__host__ __device__ constexpr size_t rc_lds_bytes(int cap) { return (size_t)cap * 12 + (size_t)cap + 16; }
// The brute-force variant appends, when the chunk is small enough to keep the occupancy (lds_exact):
//   float4 exact[cap]     {2v.x, 2v.y, 2v.z, 4c} of kernel.cu:53-58 per sphere, so that the exact re-test of
//                         a flagged sphere reads one broadcast b128 instead of going back to global memory
//   float  near[cap]      rc_near_bound of the sphere (-inf when the skip is switched off)
__host__ __device__ constexpr size_t rc_lds_bytes_exact(int cap) { return rc_lds_bytes(cap) + (size_t)cap * 20; }
// The culled variant appends: float wg_bounds[4][8] (per-wave tile bounds), ushort list[cap]
// (indices of the spheres that pass the workgroup-tile test, one segment of cap/4 per wave),
// int count[4] (entries per segment) and, with bin lists, int id[cap] (scene index of each slot's sphere).
__host__ __device__ constexpr size_t rc_lds_bytes_cull(int cap, bool with_ids)
{
    return rc_lds_bytes(cap) + 128 + (size_t)cap * 2 + 16 + (with_ids ? (size_t)cap * 4 : 0);
}

// One sphere the tile-level test could not rule out: per-pixel filter, then the exact test if
// some lane still flags it.  j (slot inside the LDS chunk) and sphere (its index in the scene)
// are wave-uniform.
template <int P>
__device__ __forceinline__ void rc_visit_sphere(const RcKernArgs &a, const float *rec, const unsigned char *flagb,
                                                const float4 *s_exact, const float *s_near, int j, int sphere, bool lane_forced,
                                                const float (&fa)[P], float fb, const float (&len)[P],
                                                const float (&dx)[P], const float (&dy)[P], const float (&dz)[P],
                                                float (&minDist)[P], int (&idx)[P])
{
    const float *g = rec + (j >> 2) * 12 + (j & 3);
    const float r = __builtin_fmaf(fb, g[4], g[0]);
    const float be = g[8];
    // pixels whose nearest hit so far is at or below the sphere's near bound cannot be updated by it
    // (rc_near_bound); lanes with an out-of-range ray set-up are exempt, their d need not be a unit vector
    float near_b = s_exact ? s_near[j] : -__builtin_inff();
    const unsigned long long forced_lanes = __builtin_amdgcn_ballot_w64(lane_forced);
    const unsigned long long all_mask = (flagb[j] != 0) ? ~0ull : forced_lanes;
    unsigned long long pm[P], any = 0ull;
#pragma unroll
    for (int p = 0; p < P; p++) {
        // wave masks combined on the scalar unit: no short-circuit evaluation, no divergent branches
        const unsigned long long flag = __builtin_amdgcn_ballot_w64(__builtin_fmaf(fa[p], be, r) >= len[p]);
        const unsigned long long far = __builtin_amdgcn_ballot_w64(minDist[p] <= near_b);
        pm[p] = (flag | all_mask) & (~far | forced_lanes);
        any |= pm[p];
    }
    RC_COUNT(2, 1);
    if (any == 0ull) return;
    RcExactSphere e;
    if (s_exact) {
        const float4 v = s_exact[j];
        e.tvx = v.x; e.tvy = v.y; e.tvz = v.z; e.c4 = v.w;
    } else {
        e = rc_exact_sphere(a, sphere);
        if (a.near_skip) {
            near_b = e.near;
#pragma unroll
            for (int p = 0; p < P; p++) pm[p] &= ~__builtin_amdgcn_ballot_w64(minDist[p] <= near_b) | forced_lanes;
        }
    }
    // the exact test only for pixel columns in which some lane still flags the sphere
#pragma unroll
    for (int p = 0; p < P; p++)
        if (pm[p] != 0ull) { RC_COUNT(3, 1); rc_exact_test(e, dx[p], dy[p], dz[p], sphere, minDist[p], idx[p]); }
}

// Wave-wide minimum / maximum with every lane active, as DPP operands of the min / max instructions
// themselves (two quad permutes, two row mirrors, two row broadcasts: six instructions and one
// v_readlane instead of six LDS-crossbar shuffles with their address arithmetic).  After the mirrors
// every lane holds its row's result; row_bcast15 / row_bcast31 carry lane 15 of rows 0 and 2 into rows 1
// and 3 and lane 31 into rows 2 and 3, so lane 63 ends with the whole wave's.  Lanes a step does not
// write keep `old` = their own value, the identity of min and max.  The result is wave-uniform (SGPR).
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float rc_dpp(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, v), __builtin_bit_cast(int, v),
                                                                 CTRL, ROW_MASK, 0xf, false));
}
__device__ __forceinline__ float rc_wave_min(float v)
{
    v = fminf(v, rc_dpp<0xB1, 0xf>(v));      // quad_perm:[1,0,3,2]
    v = fminf(v, rc_dpp<0x4E, 0xf>(v));      // quad_perm:[2,3,0,1]
    v = fminf(v, rc_dpp<0x141, 0xf>(v));     // row_half_mirror
    v = fminf(v, rc_dpp<0x140, 0xf>(v));     // row_mirror
    v = fminf(v, rc_dpp<0x142, 0xa>(v));     // row_bcast:15 into rows 1 and 3
    v = fminf(v, rc_dpp<0x143, 0xc>(v));     // row_bcast:31 into rows 2 and 3
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
__device__ __forceinline__ float rc_wave_max(float v)
{
    v = fmaxf(v, rc_dpp<0xB1, 0xf>(v));
    v = fmaxf(v, rc_dpp<0x4E, 0xf>(v));
    v = fmaxf(v, rc_dpp<0x141, 0xf>(v));
    v = fmaxf(v, rc_dpp<0x140, 0xf>(v));
    v = fmaxf(v, rc_dpp<0x142, 0xa>(v));
    v = fmaxf(v, rc_dpp<0x143, 0xc>(v));
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

__device__ __forceinline__ float rc_wave_sum(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = v + __shfl_xor(v, o, 64);
    return v;
}

// Shadow-ray filter record of one sphere for rays that all start at the light:
// h = -(light - pos) * k with the same k as above (c taken from the light).  A shadow ray
// with unit direction e can meet the sphere in front of the light only if h.e >= 1 up to
// rounding: the exact test needs -(v.e) >= sqrt(c)(1-u) - 3.01uL, the fma chain differs from
// k(-(v.e)) by <= 7.1ukL, and 1/k = sqrt(c)(1 - 2^-19) - 2^-17 L leaves > 10x that margin.
struct RcLightRecord { float hx, hy, hz; bool always; };

__device__ __forceinline__ RcLightRecord rc_prepass_light(const RcKernArgs &a, int i, RcExactSphere *exact = nullptr)
{
    float px = a.position[3 * i + 0], py = a.position[3 * i + 1], pz = a.position[3 * i + 2];
    float r = a.radius[i];
    float vx = a.lgx - px, vy = a.lgy - py, vz = a.lgz - pz;
    float vv = (vx * vx + vy * vy) + vz * vz;
    float c = vv - r * r;
    if (exact) {                                            // the same statements as rc_exact_sphere_from
        exact->tvx = 2.0f * vx; exact->tvy = 2.0f * vy; exact->tvz = 2.0f * vz;
        exact->c4 = 4.0f * c;
        exact->near = 0.0f;                                 // not used by the shadow pass
    }
    float thr = __builtin_amdgcn_sqrtf(c) * (1.0f - 0x1p-19f) - 0x1p-17f * __builtin_amdgcn_sqrtf(vv);
    RcLightRecord rec;
    rec.always = !(c > 0.0f) || !(thr > 0.0f) || !(vv < 0x1p80f) || !(vv > 0x1p-80f);
    float k = __builtin_amdgcn_rcpf(thr);
    rec.hx = -k * vx; rec.hy = -k * vy; rec.hz = -k * vz;
    rec.always = rec.always || !(fabsf(rec.hx) < 0x1p40f) || !(fabsf(rec.hy) < 0x1p40f) || !(fabsf(rec.hz) < 0x1p40f);
    if (rec.always) { rec.hx = 0.0f; rec.hy = 0.0f; rec.hz = 0.0f; }
    return rec;
}

// Brute-force blocks of the shadow pass over one LDS chunk: every (hit pixel, sphere) pair gets the
// conservative filter, flagged spheres the exact test.
//
// FAST = false: the records hold h = (hx, hy, hz) and the filter is T = h.e >= 1 (mul + two FMAs).
//
// FAST = true: all shadow rays of the WORKGROUP share a dominant axis m and sign s with g = s e_m >= 1/2
// (rays of neighbouring pixels start at one light and end close together, so this is the normal case).
// Dividing h.e >= 1 by g > 0 gives
//      s h_m + u h_j + v h_k  >=  1/g,        u = e_j / g,  v = e_k / g     (j, k the other two axes)
// so with records (s h_m, h_j, h_k) the filter is T = fma(u, R1, fma(v, R2, R0)) >= len, two FMAs per
// pair instead of three.  Soundness on top of the bound at rc_prepass_light (an exact hit implies, in
// real numbers, h.e >= 1 + S with S >= 2^-19 + 0.97 2^-17 |h|; u = 2^-24): |u|, |v| <= sqrt(3) because
// g >= 1/2 and |e| = 1 within 3u; il = rcp(g) is within 2u of 1/g (1-ulp hardware reciprocal), so the
// float u, v are within 3u; the two FMAs then differ from the real T by at most
// u (4|v||h_k| + |h_m| + 3|u||h_j| + |T|) <= 17.7 u |h|; the threshold il (1 - 2^-21) is below
// (1 + S)/g - (S - 2u).  Needed: S - 2u - 17.7 u |h| >= 0, which holds with a factor of 6 to spare.
// Pixels without a shadow ray carry u = v = 0 and len = +inf (never flagged); forced lanes and
// "always" spheres bypass the filter in both forms.
//
// A shadow ray asks for ANY occluder, so a pixel that is in shadow needs no further test: in the FAST form
// its threshold becomes +inf (it stops flagging blocks), and once every shadow ray of the wave is in shadow
// the wave leaves the loop (returns true; measured 3840x2160 x 1024: 0.720 vs 0.734 ms, 7680x4320 x 4096:
// 10.4 vs 10.65 ms; the other form, used by small scenes, got 1 % slower with the check and goes without).  Both only ever skip tests whose outcome could not change a bit.
//
// Which spheres of a flagged block?  In the FAST form, as in the primary search, lane l tests sphere l of the
// block against interval bounds of the wave's filter inputs: T is linear in u and in v, so with the inputs of
// every shadow ray of the wave inside [lo, hi] per component,
//      T <= R0 + max(u_lo R1, u_hi R1) + max(v_lo R2, v_hi R2)
// for every pixel; a sphere whose bound, plus 16u of the summed magnitudes for the roundings of T (three) and of
// the bound (five), stays below the wave's smallest threshold is flagged by no pixel.  Survivors get the
// per-pixel filter and, if a lane still flags them, the exact test (measured: finding the spheres by
// re-evaluating the whole block per pixel was 7 of 38 us at 1920x1080 x 64 and 8 % of 3840x2160 x 1024).
struct RcShadowBounds { float lo1, hi1, lo2, hi2, abs1, abs2, thr_min; };

template <int P, bool FAST>
__device__ __forceinline__ bool rc_shadow_blocks(const RcKernArgs &a, const float4 *s_rec, const unsigned *s_flag,
                                                 const float4 *s_exact, int base, int cn, int nblocks,
                                                 unsigned long long flag0_mask, unsigned long long forced_mask,
                                                 const RcShadowBounds &wb,
                                                 const float (&fu)[P], const float (&fv)[P], float (&flen)[P],
                                                 const float (&ex)[P], const float (&ey)[P], const float (&ez)[P],
                                                 const float (&ldist)[P], const int (&idx)[P], unsigned &shadowed)
{
    for (int blk = 0; blk < nblocks; blk++) {
        RC_COUNT(4, 1);
        float m[P];
#pragma unroll
        for (int p = 0; p < P; p++) m[p] = FAST ? -__builtin_inff() : 0.0f;
#pragma unroll P <= 2 ? 2 : 4   // of 8: deeper unrolling prefetches more records than the register budget holds (spills, measured slower)
        for (int g = 0; g < kGroupsPerBlock; g++) {
            const float4 H0 = s_rec[(blk * kGroupsPerBlock + g) * 3 + 0];
            const float4 H1 = s_rec[(blk * kGroupsPerBlock + g) * 3 + 1];
            const float4 H2 = s_rec[(blk * kGroupsPerBlock + g) * 3 + 2];
#pragma unroll
            for (int p = 0; p < P; p++) {
                rc_f2 f01, f23;
                if (FAST) {
                    const rc_f2 uu = {fu[p], fu[p]}, vv = {fv[p], fv[p]};
                    f01 = rc_fma2(uu, rc_f2{H1.x, H1.y}, rc_fma2(vv, rc_f2{H2.x, H2.y}, rc_f2{H0.x, H0.y}));
                    f23 = rc_fma2(uu, rc_f2{H1.z, H1.w}, rc_fma2(vv, rc_f2{H2.z, H2.w}, rc_f2{H0.z, H0.w}));
                } else {
                    const rc_f2 e0 = {ex[p], ex[p]}, e1 = {ey[p], ey[p]}, e2 = {ez[p], ez[p]};
                    f01 = rc_fma2(e2, rc_f2{H2.x, H2.y}, rc_fma2(e1, rc_f2{H1.x, H1.y}, e0 * rc_f2{H0.x, H0.y}));
                    f23 = rc_fma2(e2, rc_f2{H2.z, H2.w}, rc_fma2(e1, rc_f2{H1.z, H1.w}, e0 * rc_f2{H0.z, H0.w}));
                }
                m[p] = fmaxf(fmaxf(m[p], f01.x), f01.y);
                m[p] = fmaxf(fmaxf(m[p], f23.x), f23.y);
            }
        }
        unsigned long long flagged = flag0_mask;
#pragma unroll
        for (int p = 0; p < P; p++) flagged |= __builtin_amdgcn_ballot_w64(m[p] >= (FAST ? flen[p] : 1.0f));
        if (flagged != 0ull) {
            RC_COUNT(5, 1);
            if constexpr (FAST) {
            const float *rec = reinterpret_cast<const float *>(s_rec);
            const unsigned char *flagb = reinterpret_cast<const unsigned char *>(s_flag);
            const int lane = threadIdx.x & 63;
            const int jl = blk * kBlockSpheres + (lane & (kBlockSpheres - 1));
            unsigned long long cand_mask;
            {
                const float *g = rec + (jl >> 2) * 12 + (jl & 3);
                const float r0 = g[0], r1 = g[4], r2 = g[8];
                const float tmax = (r0 + fmaxf(wb.lo1 * r1, wb.hi1 * r1)) + fmaxf(wb.lo2 * r2, wb.hi2 * r2);
                const float mag = (fabsf(r0) + wb.abs1 * fabsf(r1)) + wb.abs2 * fabsf(r2);
                const bool c = (lane < kBlockSpheres) && (jl < cn) &&
                               (forced_mask != 0ull || flagb[jl] != 0 || !(tmax + 0x1p-20f * mag < wb.thr_min));
                cand_mask = __builtin_amdgcn_ballot_w64(c);
            }
            while (cand_mask != 0ull) {
                const int bit = __builtin_ctzll(cand_mask);
                cand_mask &= cand_mask - 1ull;
                const int j = blk * kBlockSpheres + bit;
                const float *g = rec + (j >> 2) * 12 + (j & 3);
                const float h0 = g[0], h1 = g[4], h2 = g[8];
                const unsigned long long all_mask = (flagb[j] != 0) ? ~0ull : forced_mask;
                unsigned long long any = all_mask;
#pragma unroll
                for (int p = 0; p < P; p++) {
                    const float t = FAST ? __builtin_fmaf(fu[p], h1, __builtin_fmaf(fv[p], h2, h0))
                                         : __builtin_fmaf(ez[p], h2, __builtin_fmaf(ey[p], h1, ex[p] * h0));
                    // (masking out the pixels that lie on sphere j itself -- they always pass their own sphere's filter and
                    // are excluded from its shadow test -- saves exact tests but measured slower: 680 vs 673.5 us)
                    any |= __builtin_amdgcn_ballot_w64(t >= (FAST ? flen[p] : 1.0f));
                }
                RC_COUNT(6, 1);
                if (any == 0ull) continue;
                RC_COUNT(7, 1);
                RcExactSphere e;
                if (s_exact) {
                    const float4 v = s_exact[j];
                    e.tvx = v.x; e.tvy = v.y; e.tvz = v.z; e.c4 = v.w;
                } else {
                    e = rc_exact_sphere_from(a, base + j, a.lgx, a.lgy, a.lgz);
                }
#pragma unroll
                for (int p = 0; p < P; p++) {
                    const float dist = rc_exact_dist(e, ex[p], ey[p], ez[p]);
                    if (base + j != idx[p] && dist > 0.0f && dist < rc_shadow_limit(ldist[p])) shadowed |= 1u << p;
                }
            }
            } else {
                // three-product form (small scenes): the interval bound over (ex, ey, ez) is too loose to pay (measured:
                // 1920x1080 x 64 38.9 vs 37.9 us, 1280x768 x 512 108.9 vs 102.6 us), so every group of the block is
                // evaluated again per pixel and each sphere gets its own wave-wide flag
#pragma nounroll
            for (int g = 0; g < kGroupsPerBlock; g++) {
                const int gi = blk * kGroupsPerBlock + g;
                if (gi * kGroup >= cn) break;
                const float4 H0 = s_rec[gi * 3 + 0];
                const float4 H1 = s_rec[gi * 3 + 1];
                const float4 H2 = s_rec[gi * 3 + 2];
                const unsigned fl = __builtin_amdgcn_readfirstlane(s_flag[gi]);
                unsigned long long cm[kGroup];
#pragma unroll
                for (int s = 0; s < kGroup; s++) cm[s] = ((fl >> (8 * s)) & 0xffu) ? ~0ull : forced_mask;
#pragma unroll
                for (int p = 0; p < P; p++) {
                    rc_f2 f01, f23;
                    if (FAST) {
                        const rc_f2 uu = {fu[p], fu[p]}, vv = {fv[p], fv[p]};
                        f01 = rc_fma2(uu, rc_f2{H1.x, H1.y}, rc_fma2(vv, rc_f2{H2.x, H2.y}, rc_f2{H0.x, H0.y}));
                        f23 = rc_fma2(uu, rc_f2{H1.z, H1.w}, rc_fma2(vv, rc_f2{H2.z, H2.w}, rc_f2{H0.z, H0.w}));
                    } else {
                        const rc_f2 e0 = {ex[p], ex[p]}, e1 = {ey[p], ey[p]}, e2 = {ez[p], ez[p]};
                        f01 = rc_fma2(e2, rc_f2{H2.x, H2.y}, rc_fma2(e1, rc_f2{H1.x, H1.y}, e0 * rc_f2{H0.x, H0.y}));
                        f23 = rc_fma2(e2, rc_f2{H2.z, H2.w}, rc_fma2(e1, rc_f2{H1.z, H1.w}, e0 * rc_f2{H0.z, H0.w}));
                    }
                    const float thr = FAST ? flen[p] : 1.0f;
                    cm[0] |= __builtin_amdgcn_ballot_w64(f01.x >= thr);
                    cm[1] |= __builtin_amdgcn_ballot_w64(f01.y >= thr);
                    cm[2] |= __builtin_amdgcn_ballot_w64(f23.x >= thr);
                    cm[3] |= __builtin_amdgcn_ballot_w64(f23.y >= thr);
                }
#pragma unroll
                for (int s = 0; s < kGroup; s++) {
                    const int j = gi * kGroup + s;
                    RC_COUNT(6, 1);
                    if (cm[s] == 0ull || j >= cn) continue;
                    RC_COUNT(7, 1);
                    RcExactSphere e;
                    if (s_exact) {
                        const float4 v = s_exact[j];
                        e.tvx = v.x; e.tvy = v.y; e.tvz = v.z; e.c4 = v.w;
                    } else {
                        e = rc_exact_sphere_from(a, base + j, a.lgx, a.lgy, a.lgz);
                    }
#pragma unroll
                    for (int p = 0; p < P; p++) {
                        const float dist = rc_exact_dist(e, ex[p], ey[p], ez[p]);
                        if (base + j != idx[p] && dist > 0.0f && dist < rc_shadow_limit(ldist[p])) shadowed |= 1u << p;
                    }
                }
            }
            }
            if (FAST) {                            // large scenes only: at 64 spheres the check costs more than it saves
                bool pending = false;              // a shadow ray of this lane still unoccluded (misses carry ldist = -1)
#pragma unroll
                for (int p = 0; p < P; p++) pending = pending || (ldist[p] >= 0.0f && !((shadowed >> p) & 1u));
                if (__builtin_amdgcn_ballot_w64(pending) == 0ull) { RC_COUNT(10, 1); return true; }
#pragma unroll
                for (int p = 0; p < P; p++)
                    if ((shadowed >> p) & 1u) flen[p] = __builtin_inff();
            }
        }
    }
    return false;
}

// Hard-shadow pass of EXT mode: for every hit pixel, is any sphere other than the one hit
// between the light and the point?  Same structure as the primary search (LDS records,
// conservative FMA filter, exact re-test of what the filter cannot rule out); the rays share
// the light as origin but their directions are arbitrary, so the filter is the 3-FMA dot
// product h.e >= 1 instead of the separable form.
//
// Culled variant (CULL): instead of testing every hit pixel against every sphere, each wave bounds
// its shadow rays by a cone (axis c = the normalised sum of the tile's ray directions, half-angle
// phi = the largest angle between a ray and c) and lane l tests sphere 64 it + l against the cone:
// for a unit e within phi of c, h.e <= |h| if h itself lies within phi of c, and otherwise
// h.e <= (h.c) cos phi + |h - (h.c) c| sin phi.  Spheres whose bound stays below 1 are flagged by no
// pixel of the tile; the others are visited one by one with the per-pixel filter and the exact
// test.  The cone is widened (cos phi lowered by 2^-18, the perpendicular part of h raised by
// 2^-10 |h|, the bound by 2^-16) well beyond the float errors of its own evaluation, and tiles
// whose rays do not fit a cone narrower than 90 degrees test every sphere.
template <int P, bool CULL, bool SHFAST>
__device__ __forceinline__ void rc_shadow_pass(const RcKernArgs &a, char *lds,
                                               const float (&ex)[P], const float (&ey)[P], const float (&ez)[P],
                                               const float (&ldist)[P], const int (&idx)[P], unsigned &shadowed)
{
    const int cap = a.lds_cap;
    float4 *s_rec = reinterpret_cast<float4 *>(lds);
    unsigned *s_flag = reinterpret_cast<unsigned *>(lds + (size_t)cap * 12);
    int *s_has_always = reinterpret_cast<int *>(lds + (size_t)cap * 13);
    float4 *s_exact = (!CULL && a.lds_exact) ? reinterpret_cast<float4 *>(lds + rc_lds_bytes(cap)) : nullptr;
    const int tid = threadIdx.x;

    bool lane_forced = false;
#pragma unroll
    for (int p = 0; p < P; p++) {
        const float ee = (ex[p] * ex[p] + ey[p] * ey[p]) + ez[p] * ez[p];
        // hit pixels whose shadow ray is not a unit vector (point at the light, overflow) bypass the filter
        if (idx[p] != -1 && (!(ee < 1.0f + 0x1p-19f) || !(ee > 1.0f - 0x1p-19f))) lane_forced = true;
    }
    const unsigned long long forced_mask = __builtin_amdgcn_ballot_w64(lane_forced);
    bool any_hit = false;
#pragma unroll
    for (int p = 0; p < P; p++) any_hit = any_hit || (idx[p] != -1);
    bool wave_has_hits = __builtin_amdgcn_ballot_w64(any_hit) != 0ull;   // no hit pixel: no shadow rays
    if (wave_has_hits) RC_COUNT(9, 1);

    float cax = 0.0f, cay = 0.0f, caz = 0.0f, ccos = 1.0f, csin = 0.0f;
    bool cone_forced = (forced_mask != 0ull);
    if (CULL && wave_has_hits) {
        float sx = 0.0f, sy = 0.0f, sz = 0.0f;                 // misses carry e = 0
#pragma unroll
        for (int p = 0; p < P; p++) { sx += ex[p]; sy += ey[p]; sz += ez[p]; }
        sx = rc_wave_sum(sx); sy = rc_wave_sum(sy); sz = rc_wave_sum(sz);
        const float ss = __builtin_fmaf(sz, sz, __builtin_fmaf(sy, sy, sx * sx));
        const float inv = __builtin_amdgcn_rsqf(ss);
        cax = sx * inv; cay = sy * inv; caz = sz * inv;
        float cmin = 2.0f;
#pragma unroll
        for (int p = 0; p < P; p++) {
            const float ce = __builtin_fmaf(ez[p], caz, __builtin_fmaf(ey[p], cay, ex[p] * cax));
            if (idx[p] != -1) cmin = fminf(cmin, ce);
        }
        cmin = rc_wave_min(cmin);
        ccos = cmin - 0x1p-18f;
        if (!(ss > 0x1p-20f) || !(ss < 0x1p40f) || !(ccos > 0.0f) || !(ccos <= 1.0f)) cone_forced = true;
        csin = __builtin_amdgcn_sqrtf(fmaxf(__builtin_fmaf(-ccos, ccos, 1.0f), 0.0f)) * (1.0f + 0x1p-18f) + 0x1p-20f;
    }

    // Brute-force variant: does the whole workgroup share a dominant shadow-ray axis (see rc_shadow_blocks)?
    int fast_axis = -1;
    float fast_sign = 1.0f;
    float fu[P], fv[P], flen[P];
#pragma unroll
    for (int p = 0; p < P; p++) { fu[p] = 0.0f; fv[p] = 0.0f; flen[p] = __builtin_inff(); }
    // One barrier serves three purposes: every wave has finished with the primary records (the light records
    // overwrite them); the waves' votes on a common dominant axis are exchanged; and so is "this wave casts a
    // shadow ray at all" -- a workgroup over background only (no wave with a hit pixel) leaves here, all waves
    // together, without building light records it would never read.
    __shared__ unsigned s_vote[8];
    unsigned ok = 0u;
    if (!CULL && SHFAST) {
        // bit k of `ok`: candidate k = (axis k >> 1, sign k & 1 ? -1 : +1) has s e_m >= 1/2 for every shadow
        // ray of this wave (a wave without shadow rays accepts every candidate)
#pragma unroll
        for (int k = 0; k < 6; k++) {
            bool bad = false;
#pragma unroll
            for (int p = 0; p < P; p++) {
                const float em = (k >> 1) == 0 ? ex[p] : ((k >> 1) == 1 ? ey[p] : ez[p]);
                bad = bad || (idx[p] != -1 && !(((k & 1) ? -em : em) >= 0.5f));
            }
            if (__builtin_amdgcn_ballot_w64(bad) == 0ull) ok |= 1u << k;
        }
    }
    if ((tid & 63) == 0) {
        s_vote[tid >> 6] = ok;
        s_vote[4 + (tid >> 6)] = wave_has_hits ? 1u : 0u;
    }
    __syncthreads();
    if ((s_vote[4] | s_vote[5] | s_vote[6] | s_vote[7]) == 0u) return;
    if (!CULL && SHFAST) {
        ok = s_vote[0] & s_vote[1] & s_vote[2] & s_vote[3];
        if (ok != 0u) {
            const int bk = __builtin_ctz(ok);
            fast_axis = bk >> 1;
            fast_sign = (bk & 1) ? -1.0f : 1.0f;
#pragma unroll
            for (int p = 0; p < P; p++) {
                if (idx[p] != -1) {
                    const float em = fast_axis == 0 ? ex[p] : (fast_axis == 1 ? ey[p] : ez[p]);
                    const float ej = fast_axis == 0 ? ey[p] : (fast_axis == 1 ? ez[p] : ex[p]);
                    const float ek = fast_axis == 0 ? ez[p] : (fast_axis == 1 ? ex[p] : ey[p]);
                    const float il = __builtin_amdgcn_rcpf(fast_sign * em);
                    fu[p] = ej * il;
                    fv[p] = ek * il;
                    flen[p] = il * (1.0f - 0x1p-21f);
                }
            }
        }
    }

    // Dominant-axis form: interval bounds of this wave's filter inputs (u, v, threshold) over its shadow rays
    RcShadowBounds wb = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
    if (!CULL && SHFAST && fast_axis >= 0 && wave_has_hits) {
        const float inf = __builtin_inff();
        float lo1 = inf, hi1 = -inf, lo2 = inf, hi2 = -inf, tm = inf;
#pragma unroll
        for (int p = 0; p < P; p++) {
            if (idx[p] != -1) {
                lo1 = fminf(lo1, fu[p]); hi1 = fmaxf(hi1, fu[p]);
                lo2 = fminf(lo2, fv[p]); hi2 = fmaxf(hi2, fv[p]);
                tm = fminf(tm, flen[p]);
            }
        }
        wb.lo1 = rc_wave_min(lo1); wb.hi1 = rc_wave_max(hi1);
        wb.lo2 = rc_wave_min(lo2); wb.hi2 = rc_wave_max(hi2);
        wb.abs1 = fmaxf(fabsf(wb.lo1), fabsf(wb.hi1));
        wb.abs2 = fmaxf(fabsf(wb.lo2), fabsf(wb.hi2));
        wb.thr_min = rc_wave_min(tm);
    }

    for (int base = 0; base < a.n; base += cap) {
        const int cn = min(a.n - base, cap);
        const int cnb = (cn + kBlockSpheres - 1) & ~(kBlockSpheres - 1);
        if (base > 0) __syncthreads();             // previous chunk consumed (the primary records: barrier above)
        {
            float *rec = reinterpret_cast<float *>(s_rec);
            unsigned char *flagb = reinterpret_cast<unsigned char *>(s_flag);
            bool any_always = false;
            for (int j = tid; j < cnb; j += kBlock) {
                RcLightRecord r;
                RcExactSphere ex = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
                r.hx = 0.0f; r.hy = 0.0f; r.hz = 0.0f; r.always = false;
                if (j < cn) r = rc_prepass_light(a, base + j, &ex);
                if (fast_axis >= 0) {              // (s h_m, h_j, h_k) for the workgroup's dominant axis m
                    const float hm = fast_axis == 0 ? r.hx : (fast_axis == 1 ? r.hy : r.hz);
                    const float hj = fast_axis == 0 ? r.hy : (fast_axis == 1 ? r.hz : r.hx);
                    const float hk = fast_axis == 0 ? r.hz : (fast_axis == 1 ? r.hx : r.hy);
                    r.hx = fast_sign * hm; r.hy = hj; r.hz = hk;
                }
                float *g = rec + (j >> 2) * 12 + (j & 3);
                g[0] = r.hx; g[4] = r.hy; g[8] = r.hz;
                flagb[j] = r.always ? 1 : 0;
                any_always = any_always || r.always;
                if (s_exact) s_exact[j] = make_float4(ex.tvx, ex.tvy, ex.tvz, ex.c4);
            }
            const bool wave_always = __builtin_amdgcn_ballot_w64(any_always) != 0ull;
            if ((tid & 63) == 0) s_has_always[tid >> 6] = wave_always ? 1 : 0;
        }
        __syncthreads();
        const bool has_always = (s_has_always[0] | s_has_always[1] | s_has_always[2] | s_has_always[3]) != 0;
        const unsigned long long flag0_mask = has_always ? ~0ull : forced_mask;
        const int nblocks = (wave_has_hits && !CULL) ? cnb / kBlockSpheres : 0;

        if (CULL && wave_has_hits) {
            const float *rec = reinterpret_cast<const float *>(s_rec);
            const unsigned char *flagb = reinterpret_cast<const unsigned char *>(s_flag);
            const int lane = tid & 63;
            const int iters = (cn + 63) >> 6;
            for (int it = 0; it < iters; it++) {
                const int jl = (it << 6) + lane;
                unsigned long long cand_mask;
                {
                    const int jc = min(jl, cnb - 1);
                    const float *g = rec + (jc >> 2) * 12 + (jc & 3);
                    const float hx = g[0], hy = g[4], hz = g[8];
                    const float hp = __builtin_fmaf(hz, caz, __builtin_fmaf(hy, cay, hx * cax));
                    const float hh = __builtin_fmaf(hz, hz, __builtin_fmaf(hy, hy, hx * hx));
                    const float hl = __builtin_amdgcn_sqrtf(hh) * (1.0f + 0x1p-20f);
                    const float hperp = __builtin_amdgcn_sqrtf(fmaxf(__builtin_fmaf(-hp, hp, hh), 0.0f)) + 0x1p-10f * hl;
                    const bool inside = hp * (1.0f + 0x1p-18f) >= hl * ccos;
                    const float bound = inside ? hl : __builtin_fmaf(hperp, csin, hp * ccos);
                    const bool c = (jl < cn) && (cone_forced || flagb[jc] != 0 || !(bound * (1.0f + 0x1p-16f) < 1.0f));
                    cand_mask = __builtin_amdgcn_ballot_w64(c);
                }
                while (cand_mask != 0ull) {
                    const int bit = __builtin_ctzll(cand_mask);
                    cand_mask &= cand_mask - 1ull;
                    const int j = (it << 6) + bit;
                    const float *g = rec + (j >> 2) * 12 + (j & 3);
                    const float hx = g[0], hy = g[4], hz = g[8];
                    const bool all = lane_forced || (flagb[j] != 0);
                    unsigned long long pm[P], any = 0ull;
#pragma unroll
                    for (int p = 0; p < P; p++) {
                        const float t = __builtin_fmaf(ez[p], hz, __builtin_fmaf(ey[p], hy, ex[p] * hx));
                        pm[p] = __builtin_amdgcn_ballot_w64(all || (t >= 1.0f));
                        any |= pm[p];
                    }
                    if (any == 0ull) continue;
                    const RcExactSphere e = rc_exact_sphere_from(a, base + j, a.lgx, a.lgy, a.lgz);
#pragma unroll
                    for (int p = 0; p < P; p++) {
                        if (pm[p] == 0ull) continue;
                        const float dist = rc_exact_dist(e, ex[p], ey[p], ez[p]);
                        if (base + j != idx[p] && dist > 0.0f && dist < rc_shadow_limit(ldist[p])) shadowed |= 1u << p;
                    }
                }
            }
        }

        if (!CULL && nblocks > 0) {
            bool done;
            if (SHFAST && fast_axis >= 0)
                done = rc_shadow_blocks<P, true>(a, s_rec, s_flag, s_exact, base, cn, nblocks, flag0_mask, forced_mask, wb,
                                                 fu, fv, flen, ex, ey, ez, ldist, idx, shadowed);
            else
                done = rc_shadow_blocks<P, false>(a, s_rec, s_flag, s_exact, base, cn, nblocks, flag0_mask, forced_mask, wb,
                                                  fu, fv, flen, ex, ey, ez, ldist, idx, shadowed);
            if (done) wave_has_hits = false;       // later chunks: this wave only keeps the barriers
        }
    }
}

template <int P, bool EXT, bool CULL, bool BINS, bool SHFAST>
__device__ __forceinline__ void rc_tiled_body(const RcKernArgs &a, char *lds)
{
    constexpr int TW = 8 * P;   // wave tile width in pixels
    const int cap = a.lds_cap;
    float4 *s_rec = reinterpret_cast<float4 *>(lds);
    unsigned *s_flag = reinterpret_cast<unsigned *>(lds + (size_t)cap * 12);
    int *s_has_always = reinterpret_cast<int *>(lds + (size_t)cap * 13);
    float4 *s_exact = (!CULL && a.lds_exact) ? reinterpret_cast<float4 *>(lds + rc_lds_bytes(cap)) : nullptr;
    float *s_near = reinterpret_cast<float *>(lds + rc_lds_bytes(cap) + (size_t)cap * 16);   // valid with s_exact

    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const int lx = lane & 7, ly = lane >> 3;

    // One workgroup per (16P x 16)-pixel tile (grid = tiles_x x tiles_y).  Many short workgroups beat fewer
    // long ones here: hit-heavy tiles cost more than empty ones and the hardware dispatcher evens that
    // out (2-8 tiles per workgroup measured 5-25 % slower at 3840x2160 x 1024 spheres even
    // though the sphere records would be built less often).
    {
        // LAZY (brute-force kernels): the normalised direction d -- an IEEE square root, an IEEE
        // division and three products per pixel -- and the wave tile's interval bounds are needed by the
        // exact re-tests, the plane test and the shading only, so they are evaluated when the wave first
        // meets a flagged block (or, in EXT mode, after the loop if some pixel of the wave can hit the plane);
        // a wave over empty background never pays for them.  Until then (dx, dy, dz) hold the un-normalised
        // q of kernel.cu:49.  The culled variant needs its bounds at once.
        //
        // EXT: can this pixel hit the plane at all?  The plane test accepts t = plane_num / (n.d) - 1e-6 > 0
        // only, so n.d must be non-zero with the sign of plane_num.  When the plane's normal has a single
        // non-zero component n_a (the host checks; plane_side = sign(plane_num) sign(n_a), else 0), the other
        // two products are zeros of either sign and n.d has the sign of n_a d_a = the sign of n_a q_a (d = q
        // times a positive number; if the product underflows to zero, n.d is zero and the test rejects): a
        // pixel with sign(q_a) != plane_side cannot hit.  plane_side = 0 switches the shortcut off (every pixel
        // "can"); NaN components compare false either way and are rejected by the plane test itself.
        constexpr bool LAZY = !CULL;
        RC_COUNT(8, 1);
        const int tyi = blockIdx.y, txi = blockIdx.x;
        const int x0 = txi * (2 * TW) + (wave & 1) * TW + lx;
        const int row = tyi * 16 + (wave >> 1) * 8 + ly;
        const int rowc = min(row, a.rows - 1);

        // every global load of the prologue is issued here, ahead of the arithmetic that hides its latency:
        // the sine-table entries and, in the brute-force variant, this thread's sphere of the first LDS chunk
        const float sy = a.sin_y[rowc + a.y_offset];
        float sxv[P];
#pragma unroll
        for (int p = 0; p < P; p++) sxv[p] = a.sin_x[min(x0 + 8 * p, a.width - 1)];
        // (not in the 4-pixel EXT kernels: they sit at the 128-register limit and the four values would spill)
        constexpr bool PREFETCH = !CULL && !(EXT && P >= 4);
        RcSphereIn first_sphere = {0.0f, 0.0f, 0.0f, 0.0f};
        if (PREFETCH && tid < a.n) first_sphere = rc_load_sphere(a, tid);

        float fa[P], len[P], dx[P], dy[P], dz[P], minDist[P];
        int idx[P];
        bool lane_forced = (a.cam_ok == 0);
        bool plane_maybe = false;                   // EXT: some pixel of this lane can hit the plane
        const float fb = sy;
#pragma unroll
        for (int p = 0; p < P; p++) {
            const float sx = sxv[p];
            // kernel.cu:49, exactly (the statements of rc_ray_dir)
            float tx = a.rx * sx, ty = a.ry * sx, tz = a.rz * sx;
            tx = tx * a.aspect; ty = ty * a.aspect; tz = tz * a.aspect;
            float qx = a.fx + tx, qy = a.fy + ty, qz = a.fz + tz;
            const float wx = a.ux * sy, wy = a.uy * sy, wz = a.uz * sy;
            qx = qx + wx; qy = qy + wy; qz = qz + wz;
            dx[p] = qx; dy[p] = qy; dz[p] = qz;
            if (EXT && a.has_plane) {
                const float qa = a.plane_axis == 0 ? qx : (a.plane_axis == 1 ? qy : qz);
                plane_maybe = plane_maybe || (a.plane_side == 0.0f) || (a.plane_side > 0.0f ? (qa > 0.0f) : (qa < 0.0f));
            }
            if (LAZY) {
                // The filter's threshold: a LOWER bound of the reference's |q| (a lower threshold can only
                // flag more).  The hardware square root is within one ulp, so times (1 - 2^-21) it is
                // below sqrt(s)(1 - 2^-22) <= the correctly rounded value.
                len[p] = __builtin_amdgcn_sqrtf((qx * qx + qy * qy) + qz * qz) * (1.0f - 0x1p-21f);
            } else {
                len[p] = rc_normalise(dx[p], dy[p], dz[p]);
            }
            // filter inputs
            fa[p] = sx * a.aspect;
            if (a.cam_ok != 2) {                    // 2: the host has shown that no ray of this frame can fail the check
                const float Q = a.nF + fabsf(fa[p]) * a.nR + fabsf(fb) * a.nU;
                if (!(Q <= 4.0f * len[p]) || !(len[p] > 0x1p-20f) || !(len[p] < 0x1p20f)) lane_forced = true;
            }
            minDist[p] = 10000.0f;
            idx[p] = -1;
        }
        const rc_f2 bb = {fb, fb};
        const unsigned long long forced_mask = __builtin_amdgcn_ballot_w64(lane_forced);

        // Wave-tile interval bounds of the filter inputs.  Over the tile a lies in [ta_lo, ta_hi],
        // b in [tb_lo, tb_hi] and len >= tlen_min, and T = alpha + b gamma + a beta is linear in a
        // and in b, so T <= alpha + max(tb_lo gamma, tb_hi gamma) + max(ta_lo beta, ta_hi beta) =
        // Tmax for every pixel of the tile.  A sphere whose Tmax, plus 16u of the summed magnitudes
        // for the roundings of T and Tmax, stays below tlen_min is flagged by no pixel of the tile.
        float ta_lo = 0.0f, ta_hi = 0.0f, tb_lo = 0.0f, tb_hi = 0.0f, ta_abs = 0.0f, tb_abs = 0.0f, tlen_min = 0.0f;
        const bool tile_forced = (forced_mask != 0ull);
        auto tile_bounds = [&]() {
            float lo = fa[0], hi = fa[0], lm = len[0];
#pragma unroll
            for (int p = 1; p < P; p++) { lo = fminf(lo, fa[p]); hi = fmaxf(hi, fa[p]); lm = fminf(lm, len[p]); }
            ta_lo = rc_wave_min(lo); ta_hi = rc_wave_max(hi);
            tb_lo = rc_wave_min(fb); tb_hi = rc_wave_max(fb);
            tlen_min = rc_wave_min(lm);
            ta_abs = fmaxf(fabsf(ta_lo), fabsf(ta_hi));
            tb_abs = fmaxf(fabsf(tb_lo), fabsf(tb_hi));
        };
        bool have_d = !LAZY;                        // wave-uniform
        if (!LAZY) tile_bounds();

        // Culled variant, level 1: bounds of the whole workgroup tile (the four wave tiles).
        float *s_wgb = reinterpret_cast<float *>(lds + rc_lds_bytes(cap));
        unsigned short *s_list = reinterpret_cast<unsigned short *>(lds + rc_lds_bytes(cap) + 128);
        int *s_cnt = reinterpret_cast<int *>(lds + rc_lds_bytes(cap) + 128 + (size_t)cap * 2);
        int *s_id = reinterpret_cast<int *>(lds + rc_lds_bytes(cap) + 128 + (size_t)cap * 2 + 16);
        // Level 0 (optional): the spheres that can touch this workgroup tile's bin (4 x 4 workgroup
        // tiles), listed in ascending index by rc_bin_spheres as four segments.
        int total = a.n, bc0 = 0, bc1 = 0, bc2 = 0;
        const unsigned short *bsegs = nullptr;
        if (CULL && BINS) {
            const unsigned char *bin = a.bins + (size_t)((tyi >> 2) * a.bins_x + (txi >> 2)) * a.bin_stride;
            const int *bc = reinterpret_cast<const int *>(bin);
            bc0 = bc[0]; bc1 = bc[1]; bc2 = bc[2];
            total = bc0 + bc1 + bc2 + bc[3];
            bsegs = reinterpret_cast<const unsigned short *>(bin + 16);
        }
        float wa_lo = 0.0f, wa_hi = 0.0f, wb_lo = 0.0f, wb_hi = 0.0f, wa_abs = 0.0f, wb_abs = 0.0f, wlen_min = 0.0f;
        bool wg_forced = false;
        if (CULL) {
            if (lane == 0) {
                float *w = s_wgb + wave * 8;
                w[0] = ta_lo; w[1] = ta_hi; w[2] = tb_lo; w[3] = tb_hi; w[4] = tlen_min;
                w[5] = tile_forced ? 1.0f : 0.0f;
            }
            __syncthreads();
            wa_lo = s_wgb[0]; wa_hi = s_wgb[1]; wb_lo = s_wgb[2]; wb_hi = s_wgb[3]; wlen_min = s_wgb[4];
            wg_forced = s_wgb[5] != 0.0f;
#pragma unroll
            for (int k = 1; k < 4; k++) {
                wa_lo = fminf(wa_lo, s_wgb[8 * k + 0]); wa_hi = fmaxf(wa_hi, s_wgb[8 * k + 1]);
                wb_lo = fminf(wb_lo, s_wgb[8 * k + 2]); wb_hi = fmaxf(wb_hi, s_wgb[8 * k + 3]);
                wlen_min = fminf(wlen_min, s_wgb[8 * k + 4]);   // a NaN length has made its wave tile "forced"
                wg_forced = wg_forced || (s_wgb[8 * k + 5] != 0.0f);
            }
            wa_abs = fmaxf(fabsf(wa_lo), fabsf(wa_hi));
            wb_abs = fmaxf(fabsf(wb_lo), fabsf(wb_hi));
        }

        for (int base = 0; base < total; base += cap) {
            const int cn = min(total - base, cap);
            const int cnb = (cn + kBlockSpheres - 1) & ~(kBlockSpheres - 1);   // whole blocks
            if (CULL) {
                // Records as in the brute-force variant, but wave w builds the contiguous range
                // [w Q, (w+1) Q) of the chunk and tests every sphere it builds against the
                // workgroup tile's interval bounds at once; survivors go, in ascending index, to
                // segment w of the candidate list.
                __syncthreads();                       // previous chunk fully consumed
                float *rec = reinterpret_cast<float *>(s_rec);
                unsigned char *flagb = reinterpret_cast<unsigned char *>(s_flag);
                const int Q = cnb >> 2;
                int count = 0;                         // wave-uniform
                for (int k0 = 0; k0 < Q; k0 += 64) {
                    const int jj = k0 + lane;
                    const int j = wave * Q + jj;
                    bool pass = false;
                    if (jj < Q) {
                        RcRecord r;
                        r.alpha = kNever; r.gamma = 0.0f; r.beta = 0.0f; r.always = false;
                        if (j < cn) {
                            int sphere = base + j;
                            if (BINS) {                // entry base + j of the bin's four segments
                                int e = sphere, seg = 0;
                                if (e >= bc0) { e -= bc0; seg = 1; if (e >= bc1) { e -= bc1; seg = 2; if (e >= bc2) { e -= bc2; seg = 3; } } }
                                sphere = bsegs[seg * a.bin_q + e];
                            }
                            if (BINS) s_id[j] = sphere;
                            if (BINS || a.records) {   // built once per frame by rc_build_records
                                const float4 v = a.records[sphere];
                                r.alpha = v.x; r.gamma = v.y; r.beta = v.z; r.always = (v.w != 0.0f);
                            } else {
                                r = rc_prepass_sphere(a, sphere);
                            }
                        }
                        float *g = rec + (j >> 2) * 12 + (j & 3);
                        g[0] = r.alpha; g[4] = r.gamma; g[8] = r.beta;
                        flagb[j] = r.always ? 1 : 0;
                        const float tmax = (r.alpha + fmaxf(wb_lo * r.gamma, wb_hi * r.gamma)) + fmaxf(wa_lo * r.beta, wa_hi * r.beta);
                        const float mag = (fabsf(r.alpha) + wb_abs * fabsf(r.gamma)) + wa_abs * fabsf(r.beta);
                        pass = (j < cn) && (wg_forced || r.always || (tmax + 0x1p-20f * mag >= wlen_min));
                    }
                    const unsigned long long mask = __builtin_amdgcn_ballot_w64(pass);
                    if (pass) s_list[wave * Q + count + __builtin_popcountll(mask & ((1ull << lane) - 1ull))] = (unsigned short)j;
                    count += __builtin_popcountll(mask);
                }
                if (lane == 0) s_cnt[wave] = count;
                __syncthreads();
            } else {
                if (base > 0) __syncthreads();         // previous chunk fully consumed
                float *rec = reinterpret_cast<float *>(s_rec);
                unsigned char *flagb = reinterpret_cast<unsigned char *>(s_flag);
                bool any_always = false;
                for (int j = tid; j < cnb; j += kBlock) {
                    RcRecord r;
                    RcExactSphere ex = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
                    r.alpha = kNever; r.gamma = 0.0f; r.beta = 0.0f; r.always = false;
                    // the first chunk's sphere of this thread was loaded at the top of the kernel
                    if (j < cn) r = rc_prepass_values(a, (PREFETCH && base == 0 && j == tid) ? first_sphere : rc_load_sphere(a, base + j), &ex);
                    float *g = rec + (j >> 2) * 12 + (j & 3);
                    g[0] = r.alpha; g[4] = r.gamma; g[8] = r.beta;
                    flagb[j] = r.always ? 1 : 0;
                    any_always = any_always || r.always;
                    if (s_exact) {
                        s_exact[j] = make_float4(ex.tvx, ex.tvy, ex.tvz, ex.c4);
                        s_near[j] = (a.near_skip && !r.always && j < cn) ? ex.near : -__builtin_inff();
                    }
                }
                // one flag per wave ("this wave built an always-sphere"), read by everyone after the one barrier
                const bool wave_always = __builtin_amdgcn_ballot_w64(any_always) != 0ull;
                if (lane == 0) s_has_always[wave] = wave_always ? 1 : 0;
                __syncthreads();
            }
            const bool has_always = !CULL && ((s_has_always[0] | s_has_always[1] | s_has_always[2] | s_has_always[3]) != 0);
            const unsigned long long flag0_mask = CULL ? 0ull : (has_always ? ~0ull : forced_mask);
            const int nblocks = cnb / kBlockSpheres;

            if (CULL) {
                // Level 2: lane l tests entry 64 it + l of a segment against its own wave tile's
                // bounds; survivors are visited in ascending index (segments and entries are in
                // ascending order) with the per-pixel filter and the exact test, so the result is
                // that of the brute-force loop.
                const float *rec = reinterpret_cast<const float *>(s_rec);
                const unsigned char *flagb = reinterpret_cast<const unsigned char *>(s_flag);
                const int Q = cnb >> 2;
                for (int seg = 0; seg < 4; seg++) {
                    const int cnt = __builtin_amdgcn_readfirstlane(s_cnt[seg]);
                    for (int it = 0; it < cnt; it += 64) {
                        const int e = it + lane;
                        const int jl = (e < cnt) ? (int)s_list[seg * Q + e] : 0;
                        unsigned long long cand_mask;
                        {
                            const float *g = rec + (jl >> 2) * 12 + (jl & 3);
                            const float al = g[0], ga = g[4], be = g[8];
                            const float tmax = (al + fmaxf(tb_lo * ga, tb_hi * ga)) + fmaxf(ta_lo * be, ta_hi * be);
                            const float mag = (fabsf(al) + tb_abs * fabsf(ga)) + ta_abs * fabsf(be);
                            const bool c = (e < cnt) && (tile_forced || flagb[jl] != 0 || (tmax + 0x1p-20f * mag >= tlen_min));
                            cand_mask = __builtin_amdgcn_ballot_w64(c);
                        }
                        while (cand_mask != 0ull) {
                            const int bit = __builtin_ctzll(cand_mask);
                            cand_mask &= cand_mask - 1ull;
                            const int j = __builtin_amdgcn_readlane(jl, bit);
                            rc_visit_sphere<P>(a, rec, flagb, nullptr, nullptr, j, BINS ? s_id[j] : base + j, lane_forced, fa, fb, len,
                                               dx, dy, dz, minDist, idx);
                        }
                    }
                }
            } else
            for (int blk = 0; blk < nblocks; blk++) {
                RC_COUNT(0, 1);
                float m[P];
#pragma unroll
                for (int p = 0; p < P; p++) m[p] = -__builtin_inff();
#pragma unroll
                for (int g = 0; g < kGroupsPerBlock; g++) {
                    const float4 A = s_rec[(blk * kGroupsPerBlock + g) * 3 + 0];
                    const float4 G = s_rec[(blk * kGroupsPerBlock + g) * 3 + 1];
                    const float4 B = s_rec[(blk * kGroupsPerBlock + g) * 3 + 2];
                    const rc_f2 r01 = rc_fma2(bb, rc_f2{G.x, G.y}, rc_f2{A.x, A.y});
                    const rc_f2 r23 = rc_fma2(bb, rc_f2{G.z, G.w}, rc_f2{A.z, A.w});
#pragma unroll
                    for (int p = 0; p < P; p++) {
                        const rc_f2 aa = {fa[p], fa[p]};
                        const rc_f2 t01 = rc_fma2(aa, rc_f2{B.x, B.y}, r01);
                        const rc_f2 t23 = rc_fma2(aa, rc_f2{B.z, B.w}, r23);
                        m[p] = fmaxf(fmaxf(m[p], t01.x), t01.y);
                        m[p] = fmaxf(fmaxf(m[p], t23.x), t23.y);
                    }
                }
                unsigned long long flagged = flag0_mask;          // wave-wide, lives in SGPRs
#pragma unroll
                for (int p = 0; p < P; p++) flagged |= __builtin_amdgcn_ballot_w64(m[p] >= len[p]);
                if (flagged != 0ull) {
                    RC_COUNT(1, 1);
                    if (LAZY && !have_d) {             // first flagged block of this wave: see LAZY above
#pragma unroll
                        for (int p = 0; p < P; p++) rc_normalise(dx[p], dy[p], dz[p]);
                        tile_bounds();
                        have_d = true;
                    }
                    // Rare path: some lane cannot rule out some sphere of this block.  Find which
                    // spheres: lane l tests sphere l of the block against the wave tile's interval
                    // bounds computed above (a superset of what any pixel flags), then each survivor
                    // gets the per-pixel filter again (same fma, same bits) and, if a lane still
                    // flags it, the reference's exact test.
                    const float *rec = reinterpret_cast<const float *>(s_rec);
                    const unsigned char *flagb = reinterpret_cast<const unsigned char *>(s_flag);
                    const int jl = blk * kBlockSpheres + (lane & (kBlockSpheres - 1));
                    unsigned long long cand_mask;
                    {
                        const float *g = rec + (jl >> 2) * 12 + (jl & 3);
                        const float al = g[0], ga = g[4], be = g[8];
                        const float tmax = (al + fmaxf(tb_lo * ga, tb_hi * ga)) + fmaxf(ta_lo * be, ta_hi * be);
                        const float mag = (fabsf(al) + tb_abs * fabsf(ga)) + ta_abs * fabsf(be);
                        const bool c = (lane < kBlockSpheres) && (jl < cn) &&
                                       (tile_forced || flagb[jl] != 0 || (tmax + 0x1p-20f * mag >= tlen_min));
                        cand_mask = __builtin_amdgcn_ballot_w64(c);
                    }
                    while (cand_mask != 0ull) {
                        const int bit = __builtin_ctzll(cand_mask);
                        cand_mask &= cand_mask - 1ull;
                        rc_visit_sphere<P>(a, rec, flagb, s_exact, s_near, blk * kBlockSpheres + bit, base + blk * kBlockSpheres + bit, lane_forced, fa, fb, len,
                                           dx, dy, dz, minDist, idx);
                    }
                }
            }
        }

        unsigned shadowed = 0u;                     // bit p: pixel p of this lane is in shadow
        float ldist[P];                             // EXT: |light - point| of the pixel's shadow ray, -1 without one
#pragma unroll
        for (int p = 0; p < P; p++) ldist[p] = -1.0f;
        if (EXT) {
            if (LAZY && !have_d && __builtin_amdgcn_ballot_w64(plane_maybe) != 0ull) {
#pragma unroll
                for (int p = 0; p < P; p++) rc_normalise(dx[p], dy[p], dz[p]);
                have_d = true;
            }
            // without d by now: no pixel of this wave hit a sphere and none can hit the plane
            if (have_d) {
#pragma unroll
                for (int p = 0; p < P; p++) rc_plane_test(a, dx[p], dy[p], dz[p], minDist[p], idx[p]);
            }
            if (a.shadows) {
                float ex[P], ey[P], ez[P];
#pragma unroll
                for (int p = 0; p < P; p++) {
                    ex[p] = 0.0f; ey[p] = 0.0f; ez[p] = 0.0f;                      // misses cast no shadow ray
                    if (idx[p] != -1) rc_ext_ray(a, minDist[p], dx[p], dy[p], dz[p], ex[p], ey[p], ez[p], ldist[p]);
                }
#pragma unroll
                for (int p = 0; p < P; p++)
                    if (idx[p] != -1 && rc_plane_shadow(a, idx[p], ex[p], ey[p], ez[p], rc_shadow_limit(ldist[p]))) shadowed |= 1u << p;
                rc_shadow_pass<P, CULL, SHFAST>(a, lds, ex, ey, ez, ldist, idx, shadowed);
            }
        }

        // (EXT: the pixel's row and columns are formed again from an opaque copy of the thread index, so that
        // the compiler does not carry them, sign-extended, across the shadow pass in registers it does not have)
        int xs = x0, rows_ = row;
        if (EXT) {
            int t2 = threadIdx.x;
            asm volatile("" : "+v"(t2));
            xs = txi * (2 * TW) + ((t2 >> 6) & 1) * TW + (t2 & 7);
            rows_ = tyi * 16 + (t2 >> 7) * 8 + ((t2 & 63) >> 3);
        }
        if (rows_ < a.rows) {
            const int row = rows_;
            const RcRowPtr rowp = rc_row_ptr(a, row);
#pragma unroll
            for (int p = 0; p < P; p++) {
                const int x = xs + 8 * p;
                if (x < a.width) {
                    float r = 0.0f, g = 0.0f, b = 0.0f;
                    if (EXT) {
                        if (idx[p] != -1) {
                            // Kernels with the dominant-axis shadow filter evaluate the ray direction again
                            // here (same statements on the same table entries, hence the same bits) instead
                            // of keeping it in registers across the shadow pass; the volatile reads keep the
                            // compiler from merging the two evaluations.
                            float rdx = dx[p], rdy = dy[p], rdz = dz[p];
                            if (SHFAST)
                                rc_ray_dir(a, *(volatile const float *)(a.sin_x + x), *(volatile const float *)(a.sin_y + row + a.y_offset),
                                           rdx, rdy, rdz);
                            rc_ext_shade(a, idx[p], minDist[p], rdx, rdy, rdz, (shadowed >> p) & 1u, (SHFAST && P >= 4) ? -1.0f : ldist[p], r, g, b);   // (128-register kernel: evaluated again rather than kept)
                        }
                    } else if (idx[p] >= 0) {
                        rc_shade(a, idx[p], minDist[p], dx[p], dy[p], dz[p], r, g, b);
                    }
                    rc_store_pixel(rowp, x, r, g, b);
                }
            }
        }
    }
}

} // namespace

// SHFAST (EXT, brute-force shadows only): the shadow pass may use the dominant-axis form of its filter.
// Waves per SIMD the register budget is cut for: 4 for the 8-pixel REF and the 4-pixel EXT kernels (128
// registers); 6 / 5 for the 2-pixel EXT kernels (80 / 96 registers, no spills), whose flagged-block path is a
// chain of dependent LDS reads and scalar masks that more resident waves hide (measured 1920x1080 x 64 38.1 ->
// 36.0 us, 1280x768 x 512 104.0 -> 99.3 us); the compiler's own choice elsewhere.
template <int P, bool EXT, bool CULL, bool BINS = false, bool SHFAST = false>
__global__ __launch_bounds__(256, ((EXT && P == 4) || P == 8) ? 4 : ((EXT && !CULL && P == 2) ? (SHFAST ? 5 : 6) : 1))
void rc_draw_tiled(RcKernArgs a)
{
    extern __shared__ __attribute__((aligned(16))) char rc_lds[];
    rc_tiled_body<P, EXT, CULL, BINS, SHFAST>(a, rc_lds);
}

// Level 0 of the culled variant (caller workspace): one workgroup per bin of 4 x 4 workgroup tiles
// lists, in ascending index, the spheres whose filter value can reach |q| somewhere in the bin.
// Bounds of the bin: a and b ranges by a scan of the sine tables over the bin's columns and rows (the
// values the lanes will use, clamped columns/rows included); a lower bound of len = |F + aR + bU| from
// convexity: |q|^2 >= |q0|^2 + grad.(a - a0, b - b0) >= |q0|^2 - |2 q0.R| ha - |2 q0.U| hb at the
// centre (a0, b0) with half-widths ha, hb, lowered by 2^-18 of the squared magnitudes for the float
// evaluation and by 2^-20 Q for the difference between the reference's float |q| and the real one.
// If that bound cannot exclude that some lane of the bin is "forced" (Q > 4 len, len out of range),
// or the camera is out of range, the bin lists every sphere.
__global__ __launch_bounds__(256) void rc_bin_spheres(RcKernArgs a, int tile_w, unsigned char *bins)
{
    __shared__ float s_red[4][4];
    __shared__ int s_nan[4];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int by = blockIdx.x / a.bins_x, bx = blockIdx.x - by * a.bins_x;
    const int x0 = bx * 4 * tile_w, x1 = min(a.width, x0 + 4 * tile_w);
    const int y0 = by * 64, y1 = min(a.rows, y0 + 64);
    float alo = __builtin_inff(), ahi = -__builtin_inff(), blo = __builtin_inff(), bhi = -__builtin_inff();
    bool nan_seen = false;
    for (int x = x0 + tid; x < x1; x += 256) {
        const float v = a.sin_x[x] * a.aspect;
        nan_seen = nan_seen || !(v == v);
        alo = fminf(alo, v); ahi = fmaxf(ahi, v);
    }
    for (int y = y0 + tid; y < y1; y += 256) {
        const float v = a.sin_y[y + a.y_offset];
        nan_seen = nan_seen || !(v == v);
        blo = fminf(blo, v); bhi = fmaxf(bhi, v);
    }
    alo = rc_wave_min(alo); ahi = rc_wave_max(ahi); blo = rc_wave_min(blo); bhi = rc_wave_max(bhi);
    const bool any_nan = __builtin_amdgcn_ballot_w64(nan_seen) != 0ull;
    if (lane == 0) {
        s_red[wave][0] = alo; s_red[wave][1] = ahi; s_red[wave][2] = blo; s_red[wave][3] = bhi;
        s_nan[wave] = any_nan ? 1 : 0;
    }
    __syncthreads();
    bool forced = (a.cam_ok == 0);
#pragma unroll
    for (int k = 0; k < 4; k++) {
        alo = fminf(alo, s_red[k][0]); ahi = fmaxf(ahi, s_red[k][1]);
        blo = fminf(blo, s_red[k][2]); bhi = fmaxf(bhi, s_red[k][3]);
        forced = forced || (s_nan[k] != 0);
    }
    const float a_abs = fmaxf(fabsf(alo), fabsf(ahi)), b_abs = fmaxf(fabsf(blo), fabsf(bhi));
    const float a0 = 0.5f * (alo + ahi), b0 = 0.5f * (blo + bhi);
    const float ha = fmaxf(ahi - a0, a0 - alo) * (1.0f + 0x1p-20f), hb = fmaxf(bhi - b0, b0 - blo) * (1.0f + 0x1p-20f);
    const float qx = __builtin_fmaf(b0, a.ux, __builtin_fmaf(a0, a.rx, a.fx));
    const float qy = __builtin_fmaf(b0, a.uy, __builtin_fmaf(a0, a.ry, a.fy));
    const float qz = __builtin_fmaf(b0, a.uz, __builtin_fmaf(a0, a.rz, a.fz));
    const float l2 = __builtin_fmaf(qz, qz, __builtin_fmaf(qy, qy, qx * qx));
    const float ga = 2.0f * fabsf(__builtin_fmaf(qz, a.rz, __builtin_fmaf(qy, a.ry, qx * a.rx)));
    const float gb = 2.0f * fabsf(__builtin_fmaf(qz, a.uz, __builtin_fmaf(qy, a.uy, qx * a.ux)));
    const float Q = a.nF + a_abs * a.nR + b_abs * a.nU;              // >= every lane's Q (same expression, monotone)
    const float lb2 = (l2 - (ga * ha + gb * hb)) - 0x1p-18f * (Q * Q + (ga * ha + gb * hb));
    float lb = __builtin_amdgcn_sqrtf(fmaxf(lb2, 0.0f)) * (1.0f - 0x1p-20f) - 0x1p-20f * Q;
    // no lane of the bin may be "forced" unless the bin is: Q_lane <= Q <= 4 lb <= 4 len_lane etc.
    if (!(lb2 > 0.0f) || !(lb > 0x1p-19f) || !(Q < 0x1p19f) || !(Q <= 4.0f * lb)) forced = true;

    unsigned short *segs = reinterpret_cast<unsigned short *>(bins + (size_t)blockIdx.x * a.bin_stride + 16);
    int *counts = reinterpret_cast<int *>(bins + (size_t)blockIdx.x * a.bin_stride);
    const int Qb = a.bin_q;
    int count = 0;
    for (int k0 = 0; k0 < Qb; k0 += 64) {
        const int jj = k0 + lane;
        const int j = wave * Qb + jj;
        bool pass = false;
        if (jj < Qb && j < a.n) {
            const float4 v = a.records[j];
            const float tmax = (v.x + fmaxf(blo * v.y, bhi * v.y)) + fmaxf(alo * v.z, ahi * v.z);
            const float mag = (fabsf(v.x) + b_abs * fabsf(v.y)) + a_abs * fabsf(v.z);
            pass = forced || (v.w != 0.0f) || !(tmax + 0x1p-20f * mag < lb);
        }
        const unsigned long long mask = __builtin_amdgcn_ballot_w64(pass);
        if (pass) segs[wave * Qb + count + __builtin_popcountll(mask & ((1ull << lane) - 1ull))] = (unsigned short)j;
        count += __builtin_popcountll(mask);
    }
    if (lane == 0) counts[wave] = count;
}

// Sphere records of one frame for the culled variant with a caller workspace: the same
// rc_prepass_sphere every workgroup would otherwise repeat, once per sphere.
__global__ __launch_bounds__(256) void rc_build_records(RcKernArgs a, float4 *out)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= a.n) return;
    const RcRecord r = rc_prepass_sphere(a, i);
    out[i] = make_float4(r.alpha, r.gamma, r.beta, r.always ? 1.0f : 0.0f);
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

// Device self-test of rc_sqrt_core / rc_rcp_core: operand i has the bit pattern first_bits + i; counts the operands
// on which the short sequence and the compiler's general one (sqrtf, 1.0f / x) differ in any bit.
__global__ __launch_bounds__(256) void rc_selftest_math_kernel(int which, uint32_t first_bits, uint32_t count,
                                                               unsigned long long *mismatches, float *out_fast, float *out_general)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= count) return;
    const float x = __builtin_bit_cast(float, first_bits + i);
    const float fast = which ? rc_rcp_core(x) : rc_sqrt_core(x);
    const float general = which ? 1.0f / x : sqrtf(x);
    if (__builtin_bit_cast(uint32_t, fast) != __builtin_bit_cast(uint32_t, general)) atomicAdd(mismatches, 1ull);
    if (out_fast) out_fast[i] = fast;
    if (out_general) out_general[i] = general;
}

// ---------------------------------------------------------------------------------------
// Launchers (host)
// ---------------------------------------------------------------------------------------

template <int P>
static long long rc_tile_count(const RcKernArgs &a)
{
    return (long long)((a.width + 16 * P - 1) / (16 * P)) * ((a.rows + 15) / 16);
}

template <int P>
static hipError_t rc_launch_tiled(RcKernArgs a, int compute_units, hipStream_t stream)
{
    // LDS chunk: the whole scene when it fits (rounded up to whole blocks of kBlockSpheres)
    const int n16 = ((a.n > 0 ? a.n : 1) + kBlockSpheres - 1) & ~(kBlockSpheres - 1);
    a.lds_cap = n16 < RC_LDS_CHUNK_MAX ? n16 : RC_LDS_CHUNK_MAX;
    // Above 2048 spheres a whole-scene chunk (13 bytes per sphere, 53 KB at 4096) leaves room for three
    // workgroups per CU only: passes of 1024 spheres keep four resident and let the exact-test terms ride
    // along in LDS.  Measured, REF: 7680x4320 x 4096 4.25 vs 4.43 ms, a 7680x540 band 0.684 vs 0.711 ms,
    // 1280x768 x 4096 0.228 vs 0.257 ms, neutral at 2048 spheres.  EXT gains only on frames with many
    // workgroups (3840x2160 x 4096: 2.67 vs 2.85 ms; 1920x1080 x 4096: 1.03 vs 0.97 ms).
    if (a.n > 2048 && !a.cull && (!a.ext_mode || rc_tile_count<P>(a) >= 4096)) a.lds_cap = RC_LDS_EXACT_MAX;
    // The REF kernels with 4 and 2 pixels per lane fit 5 and 7 waves per SIMD by their registers; passes of 960 / 672
    // spheres (31.7 / 22.2 KB with the exact-test terms) let as many workgroups share the CU's 160 KB of LDS where
    // the frame has them.  1920x1080 x 4096: 0.366 -> 0.359 ms; 1280x768 x 1024 / 2048 / 4096: 0.0615 -> 0.0596,
    // 0.120 -> 0.113, 0.228 -> 0.221 ms (at 640x480, two workgroups per CU, the shorter passes only cost: unchanged).
    if (!a.cull && !a.ext_mode && P == 4 && a.n > 2048) a.lds_cap = 960;
    if (!a.cull && !a.ext_mode && P == 2 && a.n > 672 && rc_tile_count<P>(a) > 4LL * compute_units) a.lds_cap = 672;
    // The culled variant is latency-bound while it builds its records, so it wants many
    // resident workgroups: 1024-sphere passes (17 KB of LDS) measured 1.5x faster at
    // 7680x4320 x 4096 spheres than one 4096-sphere pass (62 KB, two workgroups per CU).
    if (a.cull && a.lds_cap > RC_CULL_CHUNK_MAX) a.lds_cap = RC_CULL_CHUNK_MAX;
    a.tiles_x = (a.width + 16 * P - 1) / (16 * P);
    a.tiles_y = (a.rows + 15) / 16;
    // The dominant-axis shadow filter pays from about a thousand spheres up (its set-up costs one barrier
    // and a second evaluation of the ray directions per pixel; measured on 1280x768 ... 7680x4320 frames:
    // 9 % slower at 256 spheres, even at 512, 4-10 % faster from 1024).
    const bool axis_wanted = (a.tuning & RC_KTUNE_SHADOW_AXIS_ON) ||
                             (a.n >= RC_SHADOW_FAST_MIN && !(a.tuning & RC_KTUNE_SHADOW_AXIS_OFF));
    a.shadow_fast = (a.ext_mode && a.shadows && !a.cull && axis_wanted) ? 1 : 0;
    a.lds_exact = (!a.cull && a.lds_cap <= RC_LDS_EXACT_MAX && !(a.tuning & RC_KTUNE_NO_LDS_EXACT)) ? 1 : 0;
    // The 2-pixel EXT kernels are built for 6 (dominant-axis form: 5) workgroups per CU; the staged exact-test terms
    // must not be what limits them to fewer (160 KB of LDS per CU: at 1024 spheres 33 KB each, four workgroups).
    // 1920x1080 x 1024 EXT 0.274 vs 0.280 ms, a 7680x270 band 0.260 vs 0.272 ms; at 768 spheres staging still fits.
    if (P == 2 && a.ext_mode && !a.cull && a.lds_exact &&
        rc_lds_bytes_exact(a.lds_cap) * (a.shadow_fast ? 5 : 6) > 160u * 1024u) a.lds_exact = 0;
    a.near_skip = (a.tuning & RC_KTUNE_NO_NEAR_SKIP) ? 0 : 1;
    const size_t lds_plain = a.lds_exact ? rc_lds_bytes_exact(a.lds_cap) : rc_lds_bytes(a.lds_cap);
    const dim3 grid((unsigned)a.tiles_x, (unsigned)a.tiles_y);
    if (a.cull && a.bins) {
        // level 0: bins of 4 x 4 workgroup tiles; layout fixed here because it depends on P
        a.bins_x = (a.tiles_x + 3) / 4;
        const int bins_y = (a.tiles_y + 3) / 4;
        a.bin_q = (((a.n + 3) / 4) + 7) & ~7;
        a.bin_stride = 16 + 8 * a.bin_q;
        if ((size_t)a.bins_x * bins_y * a.bin_stride > a.bins_bytes || a.n > 65535 || !a.records) {
            a.bins = nullptr;                      // workspace too small for this tiling: level 0 is skipped
        } else {
            hipLaunchKernelGGL(rc_bin_spheres, dim3((unsigned)(a.bins_x * bins_y)), dim3(256), 0, stream, a, 16 * P,
                               const_cast<unsigned char *>(a.bins));
            hipError_t eb = hipGetLastError();
            if (eb != hipSuccess) return eb;
        }
    }
    if (a.ext_mode) {
        if constexpr (P <= 4) {
            if (a.cull && a.bins) hipLaunchKernelGGL((rc_draw_tiled<P, true, true, true>), grid, dim3(256), rc_lds_bytes_cull(a.lds_cap, true), stream, a);
            else if (a.cull) hipLaunchKernelGGL((rc_draw_tiled<P, true, true>), grid, dim3(256), rc_lds_bytes_cull(a.lds_cap, false), stream, a);
            else if (a.shadow_fast) hipLaunchKernelGGL((rc_draw_tiled<P, true, false, false, true>), grid, dim3(256), lds_plain, stream, a);
            else hipLaunchKernelGGL((rc_draw_tiled<P, true, false>), grid, dim3(256), lds_plain, stream, a);
        } else {
            return hipErrorInvalidValue;           // EXT is built for <= 4 pixels per lane
        }
    } else {
        if (a.cull && a.bins) hipLaunchKernelGGL((rc_draw_tiled<P, false, true, true>), grid, dim3(256), rc_lds_bytes_cull(a.lds_cap, true), stream, a);
        else if (a.cull) hipLaunchKernelGGL((rc_draw_tiled<P, false, true>), grid, dim3(256), rc_lds_bytes_cull(a.lds_cap, false), stream, a);
        else hipLaunchKernelGGL((rc_draw_tiled<P, false, false>), grid, dim3(256), lds_plain, stream, a);
    }
    return hipGetLastError();
}

// Pixels per lane of the tiled kernels when the caller leaves the choice to the library (only a.width, a.rows, a.n,
// a.ext_mode and a.cull are read).
int rc_auto_pixels_per_lane(const RcKernArgs &a, int full_height, int compute_units)
{
    if (compute_units <= 0) compute_units = 256;
    if (full_height < a.rows) full_height = a.rows;
    // Automatic: more pixels per lane share more of the per-sphere work, but the frame must still cut into enough
    // workgroups to fill the chip (tools/p_sweep.py: 6 frame formats and 2 row bands of a 7680x4320 frame x 5 sphere
    // counts, each with every choice forced).  REF: 4 pixels per lane from 5 workgroups per CU; 8 from 9 per CU
    // (3840x2160 and up) -- and, from 1024 spheres, already from 7 per CU: 2560x1440, 3840x1080 and a 7680x540
    // band then gain 1-7 % from 8 pixels per lane and lose 3-5 % with fewer spheres.  With 4 workgroups per CU --
    // a single round of the 8-pixel kernel, no workgroup left to even out tiles of unequal cost -- 8 pixels per lane
    // lose 8-16 % on whole frames (1920x1080, 2560x800 of a 1440-row frame), whose top and bottom tiles are cheap
    // and whose middle ones are not, and win 1-10 % on a band of at most a quarter of the frame's rows, whose tiles
    // cost about the same (7680x270 of 4320 rows at three heights, 3840x540 of 2160 at two, 1920x1080 of 4320; one
    // of six shapes, 5120x400 of 2880, loses 2 % at 4096 spheres).
    const long long want = 5LL * compute_units;
    long long want8 = (a.n >= 1024 ? 7LL : 9LL) * compute_units;
    if (a.n >= 1024 && 4LL * a.rows <= full_height) want8 = 4LL * compute_units - compute_units / 8;
    // EXT keeps shadow-ray state per pixel: at most 4 pixels per lane.  The culled variant
    // prefers small wave tiles (fewer pixels share a visited sphere): 4 at most as well.
    if (!a.ext_mode && !a.cull && rc_tile_count<8>(a) >= want8) return 8;
    // EXT: the 2-pixel kernels run at 5-6 waves per SIMD, the 4-pixel ones at 4, so 4 pixels per lane need more
    // workgroups to pay: above 2048 spheres (several LDS passes) 5 per CU as above; up to 2048 spheres 10 per CU
    // (1920x1080 stays at 2: equal at 1024 spheres, 3 % faster at 2048); below 768 spheres 24 per CU (2560x1440,
    // 3840x1080 and the 7680x540 band are 2-9 % faster with 2 up to 512 spheres, even at 768, 4 % slower at 900;
    // 3840x2160 keeps 4: 4-7 % faster from 256 spheres, 3 % slower at 64).
    const long long want4 = (a.ext_mode && !a.cull && a.n <= 2048) ? (a.n < 768 ? 24LL * compute_units : 2 * want) : want;
    if (rc_tile_count<4>(a) >= want4) return 4;
    return 2;
}

hipError_t rc_launch_draw(const RcKernArgs &a, int variant, int full_height, int compute_units, hipStream_t stream)
{
    if (compute_units <= 0) compute_units = 256;
    if (variant == RC_KVARIANT_LITERAL) {
        dim3 grid((a.width + 63) / 64, (a.rows + 3) / 4);
        if (a.ext_mode) hipLaunchKernelGGL(rc_draw_literal<true>, grid, dim3(256), 0, stream, a);
        else hipLaunchKernelGGL(rc_draw_literal<false>, grid, dim3(256), 0, stream, a);
        return hipGetLastError();
    }
    if (variant == RC_KVARIANT_TILED2) return rc_launch_tiled<2>(a, compute_units, stream);
    if (variant == RC_KVARIANT_TILED4) return rc_launch_tiled<4>(a, compute_units, stream);
    if (variant == RC_KVARIANT_TILED8) return a.ext_mode ? rc_launch_tiled<4>(a, compute_units, stream) : rc_launch_tiled<8>(a, compute_units, stream);
    const int p = rc_auto_pixels_per_lane(a, full_height, compute_units);
    if (p == 8) return rc_launch_tiled<8>(a, compute_units, stream);
    if (p == 4) return rc_launch_tiled<4>(a, compute_units, stream);
    return rc_launch_tiled<2>(a, compute_units, stream);
}

hipError_t rc_launch_records(const RcKernArgs &a, float4 *records, hipStream_t stream)
{
    if (a.n <= 0) return hipSuccess;
    hipLaunchKernelGGL(rc_build_records, dim3((a.n + 255) / 256), dim3(256), 0, stream, a, records);
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

hipError_t rc_launch_selftest_math(int which, uint32_t first_bits, uint32_t count, unsigned long long *mismatches,
                                   float *out_fast, float *out_general, hipStream_t stream)
{
    if (count == 0) return hipSuccess;
    hipLaunchKernelGGL(rc_selftest_math_kernel, dim3((count + 255u) / 256u), dim3(256), 0, stream, which, first_bits, count,
                       mismatches, out_fast, out_general);
    return hipGetLastError();
}
