// mfma_filter_probe.hip -- does the matrix pipe carry the sphere filter?  One v_mfma_f32_32x32x16_bf16 evaluates
// 32 spheres x 32 pixels of T = a beta + b gamma + alpha with every fp32 factor split into three bf16 pieces
// (15 of the 16 K slots); the vector unit only folds the 16 outputs per lane into a running maximum and compares
// it with the pixels' |q|.  Measures tests per cycle per SIMD of that loop (mode 0), of the same loop without the
// vector work (mode 1) and without the matrix work (mode 2), and the largest error of the matrix form against
// double precision in units of 2^-24 (|a beta| + |b gamma| + |alpha|).
//   hipcc --offload-arch=gfx950 -O3 -o mfma_filter_probe mfma_filter_probe.hip && ./mfma_filter_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <vector>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int G = 8;            // pixel groups of 32 per wave (4 pixels per lane)

template <int MODE>
__global__ __launch_bounds__(256, 4) void probe(const uint4 *recs, int nblocks, const uint4 *pix, const float *lens, unsigned *out, int reps)
{
    extern __shared__ uint4 s_a[];                       // per block of 32 spheres: 64 x 16 bytes, lane-ordered
    for (int i = threadIdx.x; i < nblocks * 64; i += 256) s_a[i] = recs[i];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    bf16x8 B[G];
    float lenB[G];
#pragma unroll
    for (int g = 0; g < G; g++) {
        uint4 v = pix[(blockIdx.x * 4 + (threadIdx.x >> 6)) % 64 * G * 64 + g * 64 + lane];
        memcpy(&B[g], &v, 16);
        lenB[g] = lens[g * 64 + lane];
    }
    unsigned long long flagged = 0ull;
    unsigned count = 0;
    const f32x16 zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    auto reduce = [&](const f32x16 &c, float len) {
        float m = fmaxf(fmaxf(c[0], c[1]), c[2]);
        m = fmaxf(fmaxf(m, c[3]), c[4]);
        m = fmaxf(fmaxf(m, c[5]), c[6]);
        m = fmaxf(fmaxf(m, c[7]), c[8]);
        m = fmaxf(fmaxf(m, c[9]), c[10]);
        m = fmaxf(fmaxf(m, c[11]), c[12]);
        m = fmaxf(fmaxf(m, c[13]), c[14]);
        m = fmaxf(m, c[15]);
        flagged |= __builtin_amdgcn_ballot_w64(m >= len);
    };
    f32x16 fake;
#pragma unroll
    for (int k = 0; k < 16; k++) fake[k] = lens[k * 3 + lane];
    for (int rep = 0; rep < reps; rep++)
    for (int blk = 0; blk < nblocks; blk++) {
        bf16x8 A;
        uint4 av = s_a[blk * 64 + lane];
        memcpy(&A, &av, 16);
        if (MODE == 0) {
            // software pipeline, two deep: the matrix unit works on groups g + 1 and g + 2 while the vector unit
            // folds group g
            f32x16 c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, B[0], zero, 0, 0, 0);
            f32x16 c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, B[1], zero, 0, 0, 0);
#pragma unroll
            for (int g = 2; g < G; g++) {
                f32x16 c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, B[g], zero, 0, 0, 0);
                reduce(c0, lenB[g - 2]);
                c0 = c1; c1 = c2;
            }
            reduce(c0, lenB[G - 2]);
            reduce(c1, lenB[G - 1]);
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
#pragma unroll
            for (int g = 2; g < G; g++) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 8, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x002, 16, 0);
        } else if (MODE == 1) {
#pragma unroll
            for (int g = 0; g < G; g++) {
                f32x16 c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, B[g], zero, 0, 0, 0);
                flagged |= __builtin_amdgcn_ballot_w64(c[blk & 15] >= lenB[g]);
            }
        } else {
#pragma unroll
            for (int g = 0; g < G; g++) {
                asm volatile("" : "+v"(fake));
                reduce(fake, lenB[g]);
            }
        }
        if (flagged != 0ull) { count += __builtin_popcountll(flagged); flagged = 0ull; }
    }
    if (lane == 0) out[blockIdx.x * 4 + (threadIdx.x >> 6)] = count;
}

// numerical check: one wave, one MFMA per block, all 1024 outputs written
__global__ void numeric(const uint4 *recs, const uint4 *pix, float *outT)
{
    const int lane = threadIdx.x;
    bf16x8 A, B;
    uint4 av = recs[blockIdx.x * 64 + lane], bv = pix[blockIdx.x * 64 + lane];
    memcpy(&A, &av, 16); memcpy(&B, &bv, 16);
    const f32x16 zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    f32x16 c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, B, zero, 0, 0, 0);
    // C[i][j]: j = lane % 32, i = 8 * (v / 4) + 4 * (lane / 32) + v % 4
    for (int v = 0; v < 16; v++) {
        const int i = 8 * (v / 4) + 4 * (lane / 32) + (v % 4), j = lane % 32;
        outT[(size_t)blockIdx.x * 1024 + i * 32 + j] = c[v];
    }
}

static unsigned short bf16_trunc(float x) { unsigned u; memcpy(&u, &x, 4); return (unsigned short)(u >> 16); }
static float bf16_val(unsigned short h) { unsigned u = (unsigned)h << 16; float f; memcpy(&f, &u, 4); return f; }
static void split3(float x, unsigned short s[3])
{
    s[0] = bf16_trunc(x); float r = x - bf16_val(s[0]);
    s[1] = bf16_trunc(r); r = r - bf16_val(s[1]);
    s[2] = bf16_trunc(r);
}
// K layout: sphere row  [bh bm bh bl bh bm | gh gm gh gl gh gm | ah am al 0]
//           pixel col   [ah ah am ah al am | bh bh bm bh bl bm | 1  1  1  0]
static void sphere_k(float alpha, float gamma, float beta, unsigned short k[16])
{
    unsigned short b[3], g[3], a[3];
    split3(beta, b); split3(gamma, g); split3(alpha, a);
    unsigned short v[16] = {b[0], b[1], b[0], b[2], b[0], b[1], g[0], g[1], g[0], g[2], g[0], g[1], a[0], a[1], a[2], 0};
    memcpy(k, v, 32);
}
static void pixel_k(float fa, float fb, unsigned short k[16])
{
    unsigned short a[3], b[3];
    split3(fa, a); split3(fb, b);
    const unsigned short one = bf16_trunc(1.0f);
    unsigned short v[16] = {a[0], a[0], a[1], a[0], a[2], a[1], b[0], b[0], b[1], b[0], b[2], b[1], one, one, one, 0};
    memcpy(k, v, 32);
}
static double urand() { return (double)rand() / RAND_MAX; }

int main()
{
    // ---- numerics
    const int NB = 4096;
    std::vector<unsigned short> A((size_t)NB * 64 * 8), B((size_t)NB * 64 * 8);
    std::vector<float> al(NB * 32), ga(NB * 32), be(NB * 32), fa(NB * 32), fb(NB * 32);
    for (int blk = 0; blk < NB; blk++) {
        for (int i = 0; i < 32; i++) {
            const double scale = pow(10.0, urand() * 4 - 1);
            al[blk * 32 + i] = (float)((urand() * 2 - 1) * scale); ga[blk * 32 + i] = (float)((urand() * 2 - 1) * scale);
            be[blk * 32 + i] = (float)((urand() * 2 - 1) * scale);
            fa[blk * 32 + i] = (float)((urand() * 2 - 1) * 1.2); fb[blk * 32 + i] = (float)((urand() * 2 - 1) * 0.7);
            unsigned short ks[16], kp[16];
            sphere_k(al[blk * 32 + i], ga[blk * 32 + i], be[blk * 32 + i], ks);
            pixel_k(fa[blk * 32 + i], fb[blk * 32 + i], kp);
            // operand layout: lane l holds row/col l % 32, k = 8 * (l / 32) .. + 7
            for (int h = 0; h < 2; h++)
                for (int k = 0; k < 8; k++) {
                    A[((size_t)blk * 64 + h * 32 + i) * 8 + k] = ks[h * 8 + k];
                    B[((size_t)blk * 64 + h * 32 + i) * 8 + k] = kp[h * 8 + k];
                }
        }
    }
    uint4 *dA, *dB; float *dT;
    CK(hipMalloc(&dA, A.size() * 2)); CK(hipMalloc(&dB, B.size() * 2)); CK(hipMalloc(&dT, (size_t)NB * 1024 * 4));
    CK(hipMemcpy(dA, A.data(), A.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dB, B.data(), B.size() * 2, hipMemcpyHostToDevice));
    numeric<<<NB, 64>>>(dA, dB, dT);
    CK(hipDeviceSynchronize());
    std::vector<float> T((size_t)NB * 1024);
    CK(hipMemcpy(T.data(), dT, T.size() * 4, hipMemcpyDeviceToHost));
    double worst = 0, worst_fma = 0;
    for (int blk = 0; blk < NB; blk++)
        for (int i = 0; i < 32; i++)
            for (int j = 0; j < 32; j++) {
                const double a_ = fa[blk * 32 + j], b_ = fb[blk * 32 + j];
                const double exact = a_ * be[blk * 32 + i] + b_ * ga[blk * 32 + i] + al[blk * 32 + i];
                const double mag = fabs(a_ * be[blk * 32 + i]) + fabs(b_ * ga[blk * 32 + i]) + fabs((double)al[blk * 32 + i]);
                const double err = fabs((double)T[(size_t)blk * 1024 + i * 32 + j] - exact) / (mag * ldexp(1.0, -24));
                if (err > worst) worst = err;
                const float tf = fmaf(fa[blk * 32 + j], be[blk * 32 + i], fmaf(fb[blk * 32 + j], ga[blk * 32 + i], al[blk * 32 + i]));
                const double e2 = fabs((double)tf - exact) / (mag * ldexp(1.0, -24));
                if (e2 > worst_fma) worst_fma = e2;
            }
    printf("numerics: %d x 1024 tests, worst |T_mfma - exact| = %.3f x 2^-24 mag (fp32 fma form: %.3f)\n", NB, worst, worst_fma);

    // ---- throughput
    const int nblocks = 32;                       // 1024 spheres, 32 KB of LDS
    unsigned *dOut; float *dLen;
    CK(hipMalloc(&dOut, 1 << 20)); CK(hipMalloc(&dLen, G * 64 * 4));
    std::vector<float> lens(G * 64, 1.0e30f);      // never flagged
    CK(hipMemcpy(dLen, lens.data(), lens.size() * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int grid = 256 * 4 * 8;                 // 8 rounds of 4 workgroups per CU
    for (int mode = 0; mode < 3; mode++) {
        auto launch = [&]() {
            if (mode == 0) probe<0><<<grid, 256, nblocks * 1024>>>(dA, nblocks, dB, dLen, dOut, 8);
            if (mode == 1) probe<1><<<grid, 256, nblocks * 1024>>>(dA, nblocks, dB, dLen, dOut, 8);
            if (mode == 2) probe<2><<<grid, 256, nblocks * 1024>>>(dA, nblocks, dB, dLen, dOut, 8);
        };
        for (int i = 0; i < 60; i++) launch();
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        const int reps = 60;
        for (int i = 0; i < reps; i++) launch();
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        ms /= reps;
        const double tests = (double)grid * 4 * nblocks * G * 1024 * 8;
        printf("mode %d (%s): %.4f ms, %.1f G tests/s, %.2f tests/cycle/SIMD at 2.4 GHz\n", mode,
               mode == 0 ? "mfma + max3" : mode == 1 ? "mfma only" : "max3 only", ms, tests / ms / 1e6,
               tests / (ms * 1e-3) / (1024.0 * 2.4e9));
    }
    return 0;
}
