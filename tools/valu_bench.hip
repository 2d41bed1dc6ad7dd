// valu_bench.hip -- measures VALU issue rates on gfx950 that the tiled kernel's design rests on:
// plain v_fma_f32 vs packed v_pk_fma_f32 / v_pk_mul_f32 vs v_max3_f32, at 1..8 waves per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 tools/valu_bench.hip -o build/valu_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define REP16(x) x x x x x x x x x x x x x x x x


#define KIND8(NUM, INSTR) else if (KIND == NUM) { \
            REP16(asm volatile(INSTR(0) INSTR(1) INSTR(2) INSTR(3) INSTR(4) INSTR(5) INSTR(6) INSTR(7) \
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));) }
#define I_OR3(i) "v_or3_b32 %" #i ", %" #i ", %8, %9\n"
#define I_ANDOR(i) "v_and_or_b32 %" #i ", %" #i ", %8, %9\n"
#define I_MAX2(i) "v_max_f32 %" #i ", %" #i ", %8\n"
#define I_MAX2ABS(i) "v_max_f32_e64 %" #i ", %" #i ", |%8|\n"
#define I_PKMAXI16(i) "v_pk_max_i16 %" #i ", %" #i ", %8\n"
#define I_MAX3I(i) "v_max3_i32 %" #i ", %" #i ", %8, %9\n"
#define I_AND(i) "v_and_b32 %" #i ", %" #i ", %8\n"
#define I_PERM(i) "v_perm_b32 %" #i ", %" #i ", %8, %9\n"
#define I_FMAMIX(i) "v_fma_f32 %" #i ", %8, %9, %" #i "\n"
#define I_MED3(i) "v_med3_f32 %" #i ", %" #i ", %8, %9\n"
#define I_MIN3(i) "v_min3_f32 %" #i ", %" #i ", %8, %9\n"
#define I_ADD3(i) "v_add3_u32 %" #i ", %" #i ", %8, %9\n"
#define I_LSHLOR(i) "v_lshl_or_b32 %" #i ", %" #i ", %8, %9\n"
#define I_CMP(i) "v_cmp_ge_f32 vcc, %" #i ", %8\n"
#define I_PKFMAH(i) "v_pk_fma_f16 %" #i ", %" #i ", %8, %9\n"
#define I_PKMAXH(i) "v_pk_max_f16 %" #i ", %" #i ", %8\n"
#define I_PKMULH(i) "v_pk_mul_f16 %" #i ", %" #i ", %8\n"
#define I_PKADDH(i) "v_pk_add_f16 %" #i ", %" #i ", %8\n"
#define I_PKMAX3H(i) "v_pk_maximum3_f16 %" #i ", %" #i ", %8, %9\n"
#define I_DOT2H(i) "v_dot2_f32_f16 %" #i ", %8, %9, %" #i "\n"
#define I_DOT2CH(i) "v_dot2c_f32_f16 %" #i ", %8, %9\n"
#define I_MAXIMUM3(i) "v_maximum3_f32 %" #i ", %" #i ", %8, %9\n"
#define I_MAX3H(i) "v_max3_f16 %" #i ", %" #i ", %8, %9\n"
#define I_FMAH(i) "v_fma_f16 %" #i ", %" #i ", %8, %9\n"
#define I_PKMINI16(i) "v_pk_min_u16 %" #i ", %" #i ", %8\n"

template <int KIND>
__global__ __launch_bounds__(256) void k(float *out, int iters, float seed)
{
    float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    float b = seed * 0.5f, c = seed * 0.25f;
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, pb = {b, b}, pc = {c, c};
    for (int i = 0; i < iters; i++) {
        if (KIND == 0) {   // 8 independent v_fma_f32 x16
            REP16(asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                         "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));)
        } else if (KIND == 1) {   // 4 independent v_pk_fma_f32 x16 (8 FMAs per line, as KIND 0)
            REP16(asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pb), "v"(pc));)
        } else if (KIND == 2) {   // v_pk_mul_f32
            REP16(asm volatile("v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4\n"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pb));)
        } else if (KIND == 3) {   // 8 x v_max3_f32 with abs modifiers
            REP16(asm volatile("v_max3_f32 %0, %0, |%8|, |%9|\n v_max3_f32 %1, %1, |%8|, |%9|\n v_max3_f32 %2, %2, |%8|, |%9|\n v_max3_f32 %3, %3, |%8|, |%9|\n"
                         "v_max3_f32 %4, %4, |%8|, |%9|\n v_max3_f32 %5, %5, |%8|, |%9|\n v_max3_f32 %6, %6, |%8|, |%9|\n v_max3_f32 %7, %7, |%8|, |%9|\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));)
        } else if (KIND == 4) {   // 8 x v_mul_f32
            REP16(asm volatile("v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n"
                         "v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));)
        } else if (KIND == 5) {   // v_fma_f32 with an SGPR operand
            REP16(asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                         "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "s"(seed), "v"(c));)
        } else if (KIND == 6) {   // v_pk_fma_f32 with op_sel broadcast (as the compiler emits)
            REP16(asm volatile("v_pk_fma_f32 %0, %4, %0, %5 op_sel_hi:[0,1,1]\n v_pk_fma_f32 %1, %4, %1, %5 op_sel_hi:[0,1,1]\n v_pk_fma_f32 %2, %4, %2, %5 op_sel_hi:[0,1,1]\n v_pk_fma_f32 %3, %4, %3, %5 op_sel_hi:[0,1,1]\n"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pb), "v"(pc));)
        }
        KIND8(10, I_OR3) KIND8(11, I_ANDOR) KIND8(12, I_MAX2) KIND8(13, I_MAX2ABS) KIND8(14, I_PKMAXI16)
        KIND8(15, I_MAX3I) KIND8(16, I_AND) KIND8(17, I_PERM) KIND8(18, I_FMAMIX) KIND8(19, I_MED3)
        KIND8(20, I_MIN3) KIND8(21, I_ADD3) KIND8(22, I_LSHLOR)
        KIND8(30, I_PKFMAH) KIND8(31, I_PKMAXH) KIND8(32, I_PKMULH) KIND8(33, I_PKADDH) KIND8(34, I_PKMAX3H)
        KIND8(35, I_DOT2H) KIND8(36, I_DOT2CH) KIND8(37, I_MAXIMUM3) KIND8(38, I_MAX3H) KIND8(39, I_FMAH) KIND8(40, I_PKMINI16)
        else if (KIND == 23) { REP16(asm volatile(I_CMP(0) I_CMP(1) I_CMP(2) I_CMP(3) I_CMP(4) I_CMP(5) I_CMP(6) I_CMP(7)
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "vcc");) }
    }
    float r = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y;
    if (r == 12345.678f) out[threadIdx.x] = r;
}

template <int KIND>
static void run(const char *name, double flop_per_line, int lines_vinst)
{
    float *out;
    hipMalloc(&out, 4096);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    int cus = 256;
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    cus = p.multiProcessorCount;
    for (int wps = 2; wps <= 8; wps *= 4) {
        int blocks = cus * wps;   // 256 threads = 1 wave per SIMD per block
        int iters = 4000;
        hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, 100, 1.0f);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        double vinst_per_wave = (double)iters * 16 * lines_vinst;
        double clk = p.clockRate * 1e3;
        double cyc_per_inst = ms * 1e-3 * clk / (vinst_per_wave * wps);
        double tflops = flop_per_line * 16.0 * iters * 64.0 * 4.0 * wps * cus / (ms * 1e-3) / 1e12;
        printf("%-28s waves/SIMD=%d  %.3f ms  %.2f cyc/inst/SIMD (at %.0f MHz)  %.1f TFLOP/s\n", name, wps, ms,
               cyc_per_inst, clk / 1e6, tflops);
    }
    hipFree(out);
}

int main()
{
    run<0>("v_fma_f32", 16, 8);
    run<1>("v_pk_fma_f32", 16, 4);
    run<6>("v_pk_fma_f32 op_sel", 16, 4);
    run<2>("v_pk_mul_f32", 8, 4);
    run<3>("v_max3_f32 |abs|", 16, 8);
    run<4>("v_mul_f32", 8, 8);
    run<5>("v_fma_f32 sgpr-src", 16, 8);
    run<10>("v_or3_b32", 8, 8);
    run<11>("v_and_or_b32", 8, 8);
    run<12>("v_max_f32 (vop2)", 8, 8);
    run<13>("v_max_f32_e64 |abs|", 8, 8);
    run<14>("v_pk_max_i16", 8, 8);
    run<15>("v_max3_i32", 8, 8);
    run<16>("v_and_b32", 8, 8);
    run<17>("v_perm_b32", 8, 8);
    run<18>("v_fma_f32 acc-last", 16, 8);
    run<19>("v_med3_f32", 8, 8);
    run<20>("v_min3_f32", 8, 8);
    run<21>("v_add3_u32", 8, 8);
    run<22>("v_lshl_or_b32", 8, 8);
    run<23>("v_cmp_ge_f32 (vopc)", 8, 8);
    run<30>("v_pk_fma_f16", 32, 8);
    run<31>("v_pk_max_f16", 16, 8);
    run<32>("v_pk_mul_f16", 16, 8);
    run<33>("v_pk_add_f16", 16, 8);
    run<34>("v_pk_maximum3_f16", 32, 8);
    run<35>("v_dot2_f32_f16", 32, 8);
    run<36>("v_dot2c_f32_f16", 32, 8);
    run<37>("v_maximum3_f32", 16, 8);
    run<38>("v_max3_f16", 16, 8);
    run<39>("v_fma_f16", 16, 8);
    run<40>("v_pk_min_u16", 16, 8);
    return 0;
}
