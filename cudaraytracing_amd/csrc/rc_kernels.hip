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

namespace {

constexpr int kBlock = 256;            // 4 waves, one per SIMD; >= 2 workgroups per CU
constexpr int kGroup = 4;              // spheres per filter group (48 bytes of LDS)

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
//      k = 1 / thr,   thr = sqrt(c)(1 - 2^-19) - 2^-17 sqrt(v.v)   (all binary32, <= 3u error)
//  which is below sqrt(c)(1-5u) - 64uL, so k(...) >= 1 and T >= len whenever the exact test hits.
// Spheres where thr <= 0 or any quantity leaves the range in which these bounds hold (camera
// inside or within ~1e-5 relative of the surface, non-finite data, |v|^2 outside [2^-80, 2^80],
// |record| >= 2^40) are flagged "always" and re-tested exactly every time; lanes whose ray
// set-up is out of range (Q > 4 len, len outside [2^-20, 2^20], degenerate camera basis)
// bypass the filter too.
struct RcRecord { float alpha, gamma, beta; bool always; };

constexpr float kNever = -1.0e30f;   // record value that can never reach len

__device__ __forceinline__ RcRecord rc_prepass_sphere(const RcKernArgs &a, int i)
{
    float px = a.position[3 * i + 0], py = a.position[3 * i + 1], pz = a.position[3 * i + 2];
    float r = a.radius[i];
    float vx = a.ox - px, vy = a.oy - py, vz = a.oz - pz;
    float vv = (vx * vx + vy * vy) + vz * vz;
    float c = vv - r * r;                                   // the reference's c, as rounded
    float thr = sqrtf(c) * (1.0f - 0x1p-19f) - 0x1p-17f * sqrtf(vv);
    RcRecord rec;
    rec.always = !(c > 0.0f) || !(thr > 0.0f) || !(vv < 0x1p80f) || !(vv > 0x1p-80f);
    float k = 1.0f / thr;
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

typedef float rc_f2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ rc_f2 rc_fma2(rc_f2 x, rc_f2 y, rc_f2 z) { return __builtin_elementwise_fma(x, y, z); }

constexpr int kBlockSpheres = 16;                       // spheres between two flag checks
constexpr int kGroupsPerBlock = kBlockSpheres / kGroup;

// LDS image of a chunk of `cap` spheres (cap a multiple of 16):
//   float4 rec[cap/4*3]   per group of 4 spheres {alpha0..3}{gamma0..3}{beta0..3}
//   uint   flag[cap/4]    per group, byte s = "always" flag of sphere s
//   int    has_always
__host__ __device__ constexpr size_t rc_lds_bytes(int cap) { return (size_t)cap * 12 + (size_t)cap + 16; }

template <int P>
__device__ __forceinline__ void rc_tiled_body(const RcKernArgs &a, char *lds)
{
    constexpr int TW = 8 * P;   // wave tile width in pixels
    const int cap = a.lds_cap;
    float4 *s_rec = reinterpret_cast<float4 *>(lds);
    unsigned *s_flag = reinterpret_cast<unsigned *>(lds + (size_t)cap * 12);
    int *s_has_always = reinterpret_cast<int *>(lds + (size_t)cap * 13);

    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const int lx = lane & 7, ly = lane >> 3;

    // One workgroup per (16P x 16)-pixel tile.  Many short workgroups beat fewer long ones
    // here: hit-heavy tiles cost more than empty ones and the hardware dispatcher evens that
    // out (2-8 tiles per workgroup measured 5-25 % slower at 3840x2160 x 1024 spheres even
    // though the sphere records would be built less often).
    {
        const int tile = blockIdx.x;
        const int tyi = tile / a.tiles_x, txi = tile - tyi * a.tiles_x;
        const int x0 = txi * (2 * TW) + (wave & 1) * TW + lx;
        const int row = tyi * 16 + (wave >> 1) * 8 + ly;
        const int rowc = min(row, a.rows - 1);

        float fa[P], len[P], dx[P], dy[P], dz[P], minDist[P];
        int idx[P];
        bool lane_forced = (a.cam_ok == 0);
        const float sy = a.sin_y[rowc + a.y_offset];
        const float fb = sy;
#pragma unroll
        for (int p = 0; p < P; p++) {
            const int xc = min(x0 + 8 * p, a.width - 1);
            const float sx = a.sin_x[xc];
            // kernel.cu:49 + vec3::norm, exactly (same statements as rc_ray_dir, keeping |q|)
            float tx = a.rx * sx, ty = a.ry * sx, tz = a.rz * sx;
            tx = tx * a.aspect; ty = ty * a.aspect; tz = tz * a.aspect;
            float qx = a.fx + tx, qy = a.fy + ty, qz = a.fz + tz;
            const float wx = a.ux * sy, wy = a.uy * sy, wz = a.uz * sy;
            qx = qx + wx; qy = qy + wy; qz = qz + wz;
            len[p] = sqrtf((qx * qx + qy * qy) + qz * qz);
            const float inv = 1.0f / len[p];
            dx[p] = qx * inv; dy[p] = qy * inv; dz[p] = qz * inv;
            // filter inputs
            fa[p] = sx * a.aspect;
            const float Q = a.nF + fabsf(fa[p]) * a.nR + fabsf(fb) * a.nU;
            if (!(Q <= 4.0f * len[p]) || !(len[p] > 0x1p-20f) || !(len[p] < 0x1p20f)) lane_forced = true;
            minDist[p] = 10000.0f;
            idx[p] = -1;
        }
        const rc_f2 bb = {fb, fb};
        const unsigned long long forced_mask = __builtin_amdgcn_ballot_w64(lane_forced);

        for (int base = 0; base < a.n; base += cap) {
            const int cn = min(a.n - base, cap);
            const int cnb = (cn + kBlockSpheres - 1) & ~(kBlockSpheres - 1);   // whole blocks
            {
                __syncthreads();                       // previous chunk fully consumed
                if (tid == 0) *s_has_always = 0;
                __syncthreads();
                float *rec = reinterpret_cast<float *>(s_rec);
                unsigned char *flagb = reinterpret_cast<unsigned char *>(s_flag);
                for (int j = tid; j < cnb; j += kBlock) {
                    RcRecord r;
                    r.alpha = kNever; r.gamma = 0.0f; r.beta = 0.0f; r.always = false;
                    if (j < cn) r = rc_prepass_sphere(a, base + j);
                    float *g = rec + (j >> 2) * 12 + (j & 3);
                    g[0] = r.alpha; g[4] = r.gamma; g[8] = r.beta;
                    flagb[j] = r.always ? 1 : 0;
                    if (r.always) *s_has_always = 1;
                }
                __syncthreads();
            }
            const unsigned long long flag0_mask = (*s_has_always != 0) ? ~0ull : forced_mask;
            const int nblocks = cnb / kBlockSpheres;

            for (int blk = 0; blk < nblocks; blk++) {
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
                    // Rare path: some lane cannot rule out some sphere of this block.  Redo the
                    // filter group by group (same fma, same bits), this time keeping one
                    // wave-wide mask per sphere, and run the reference's exact test for every
                    // sphere some lane still flags.
#pragma nounroll
                    for (int g = 0; g < kGroupsPerBlock; g++) {
                        const int gi = blk * kGroupsPerBlock + g;
                        if (gi * kGroup >= cn) break;
                        const float4 A = s_rec[gi * 3 + 0];
                        const float4 G = s_rec[gi * 3 + 1];
                        const float4 B = s_rec[gi * 3 + 2];
                        const unsigned fl = __builtin_amdgcn_readfirstlane(s_flag[gi]);
                        const rc_f2 r01 = rc_fma2(bb, rc_f2{G.x, G.y}, rc_f2{A.x, A.y});
                        const rc_f2 r23 = rc_fma2(bb, rc_f2{G.z, G.w}, rc_f2{A.z, A.w});
                        unsigned long long cm[kGroup];
#pragma unroll
                        for (int s = 0; s < kGroup; s++) cm[s] = ((fl >> (8 * s)) & 0xffu) ? ~0ull : forced_mask;
#pragma unroll
                        for (int p = 0; p < P; p++) {
                            const rc_f2 aa = {fa[p], fa[p]};
                            const rc_f2 t01 = rc_fma2(aa, rc_f2{B.x, B.y}, r01);
                            const rc_f2 t23 = rc_fma2(aa, rc_f2{B.z, B.w}, r23);
                            cm[0] |= __builtin_amdgcn_ballot_w64(t01.x >= len[p]);
                            cm[1] |= __builtin_amdgcn_ballot_w64(t01.y >= len[p]);
                            cm[2] |= __builtin_amdgcn_ballot_w64(t23.x >= len[p]);
                            cm[3] |= __builtin_amdgcn_ballot_w64(t23.y >= len[p]);
                        }
#pragma unroll
                        for (int s = 0; s < kGroup; s++) {
                            const int j = gi * kGroup + s;
                            if (cm[s] == 0ull || j >= cn) continue;
                            const RcExactSphere e = rc_exact_sphere(a, base + j);
#pragma unroll
                            for (int p = 0; p < P; p++)
                                rc_exact_test(e, dx[p], dy[p], dz[p], base + j, minDist[p], idx[p]);
                        }
                    }
                }
            }
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
}

} // namespace

template <int P>
__global__ __launch_bounds__(256) void rc_draw_tiled(RcKernArgs a)
{
    extern __shared__ __attribute__((aligned(16))) char rc_lds[];
    rc_tiled_body<P>(a, rc_lds);
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

template <int P>
static long long rc_tile_count(const RcKernArgs &a)
{
    return (long long)((a.width + 16 * P - 1) / (16 * P)) * ((a.rows + 15) / 16);
}

template <int P>
static hipError_t rc_launch_tiled(RcKernArgs a, hipStream_t stream)
{
    // LDS chunk: the whole scene when it fits (rounded up to whole 16-sphere blocks)
    const int n16 = ((a.n > 0 ? a.n : 1) + kBlockSpheres - 1) & ~(kBlockSpheres - 1);
    a.lds_cap = n16 < RC_LDS_CHUNK_MAX ? n16 : RC_LDS_CHUNK_MAX;
    a.tiles_x = (a.width + 16 * P - 1) / (16 * P);
    a.tiles_y = (a.rows + 15) / 16;
    const unsigned grid = (unsigned)rc_tile_count<P>(a);
    hipLaunchKernelGGL(rc_draw_tiled<P>, dim3(grid), dim3(256), rc_lds_bytes(a.lds_cap), stream, a);
    return hipGetLastError();
}

hipError_t rc_launch_draw(const RcKernArgs &a, int variant, int compute_units, hipStream_t stream)
{
    if (compute_units <= 0) compute_units = 256;
    if (variant == RC_KVARIANT_LITERAL) {
        dim3 grid((a.width + 63) / 64, (a.rows + 3) / 4);
        hipLaunchKernelGGL(rc_draw_literal, grid, dim3(256), 0, stream, a);
        return hipGetLastError();
    }
    if (variant == RC_KVARIANT_TILED2) return rc_launch_tiled<2>(a, stream);
    if (variant == RC_KVARIANT_TILED4) return rc_launch_tiled<4>(a, stream);
    if (variant == RC_KVARIANT_TILED8) return rc_launch_tiled<8>(a, stream);
    // Automatic: more pixels per lane share more of the per-sphere work, but the frame must
    // still cut into enough workgroups to fill the chip (>= 5 per CU; measured crossovers:
    // 1280x768 -> 2, 1920x1080 -> 4, 3840x2160 and up -> 8 pixels per lane).
    const long long want = 5LL * compute_units;
    if (rc_tile_count<8>(a) >= want) return rc_launch_tiled<8>(a, stream);
    if (rc_tile_count<4>(a) >= want) return rc_launch_tiled<4>(a, stream);
    return rc_launch_tiled<2>(a, stream);
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
