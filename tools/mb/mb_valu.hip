// VALU issue cost per instruction kind on gfx950, measured at 1, 2, 4 and 8 waves per SIMD:
// is v_pk_fma_f32 (two IEEE fma per lane) as cheap to issue as v_fma_f32?  what do compares,
// selects and the scalar mask arithmetic around them cost?  (decides how k_blend spends its cycles)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

typedef float f32x2 __attribute__((ext_vector_type(2)));
constexpr int UNROLL = 16;   // independent chains per thread

template <int KIND>
__global__ __launch_bounds__(256) void k(float *out, int iters, float seed) {
    float a[UNROLL];
    f32x2 p[UNROLL];
    for (int i = 0; i < UNROLL; i++) { a[i] = seed + i + threadIdx.x; p[i] = f32x2{a[i], a[i] + 1.0f}; }
    const float c0 = seed * 0.5f, c1 = seed * 0.25f;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < UNROLL; i++) {
            if constexpr (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c0), "v"(c1));
            if constexpr (KIND == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(f32x2{c0, c0}), "v"(f32x2{c1, c1}));
            if constexpr (KIND == 2) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c0));
            if constexpr (KIND == 3) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(f32x2{c0, c0}));
            if constexpr (KIND == 4) asm volatile("v_min_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c0));
            if constexpr (KIND == 5) asm volatile("v_cmp_ge_f32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(c0) : "vcc");
            if constexpr (KIND == 6) asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(a[i]) : "v"(c0));
            if constexpr (KIND == 7) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
            if constexpr (KIND == 8) asm volatile("v_cmp_ge_f32 s[20:21], %0, %1\n\ts_and_b64 s[22:23], s[20:21], s[22:23]" : : "v"(a[i]), "v"(c0) : "s20", "s21", "s22", "s23", "scc");   // s_and_b64 writes SCC: undeclared, it broke the loop branch
            if constexpr (KIND == 9) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(f32x2{c0, c0}));
            // round 4: the remaining classes of k_blend_grouped<0,4>'s step (tools/blend_table.py prices the loop with these)
            if constexpr (KIND == 10) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c0));
            if constexpr (KIND == 11) asm volatile("v_cmp_ge_f32 s[20:21], %0, %1" : : "v"(a[i]), "v"(c0) : "s20", "s21");
            if constexpr (KIND == 12) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(c0));
            if constexpr (KIND == 13) asm volatile("v_mov_b64 %0, %1" : "=v"(p[i]) : "v"(p[(i + 1) % UNROLL]));
            if constexpr (KIND == 14) asm volatile("v_and_b32 %0, 0xffff, %0" : "+v"(a[i]));
            if constexpr (KIND == 15) asm volatile("v_cmp_ge_f32 vcc, %0, %1" : : "v"(a[i]), "v"(c0) : "vcc");
            // one DEPENDENT chain per thread, as the exp polynomial of the blend step is: pk_fma -> s_nop 0 -> pk_fma ...
            if constexpr (KIND == 16) asm volatile("v_pk_fma_f32 %0, %0, %1, %2\n\ts_nop 0" : "+v"(p[0]) : "v"(f32x2{c0, c0}), "v"(f32x2{c1, c1}));
            if constexpr (KIND == 17) asm volatile("v_pk_fma_f32 %0, %0, %1, %2 op_sel_hi:[1,0,0]" : "+v"(p[i]) : "s"(f32x2{c0, c0}), "v"(f32x2{c1, c1}));
        }
    }
    float s = 0;
    for (int i = 0; i < UNROLL; i++) s += a[i] + p[i].x + p[i].y;
    if (s == 12345.678f) out[threadIdx.x] = s;
}

template <int KIND>
static void run(const char *name, int instr_per_item) {
    float *out; CK(hipMalloc(&out, 4096));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 4096;
    printf("%-34s", name);
    for (int wps : {1, 2, 4, 7, 8}) {            // waves per SIMD: blocks of 256 threads = 1 wave per SIMD each (the blend runs at 7)
        const int blocks = 256 * wps;
        hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, 16, 1.0f);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        // wave-instructions issued per SIMD = wps * iters * UNROLL * instr_per_item
        const double inst = (double)wps * iters * UNROLL * instr_per_item;
        printf("  %dw: %5.2f ns/inst", wps, ms * 1e6 / inst);
    }
    printf("\n");
}

// --pmc: only what the counter calibration needs (tools/calibrate_valu.sh runs this under
// rocprofv3 --pmc): the saturating v_fma_f32 and v_pk_fma_f32 streams at 8 waves per SIMD, a few
// launches each, wall time per launch printed beside them.  A pure stream of independent VALU
// instructions at 8 waves per SIMD keeps the SIMD's vector issue busy all the time: its counters
// per SIMD-cycle are what "VALU busy = 1.0" looks like on this part.
template <int KIND>
static void run_pmc(const char *name) {
    float *out; CK(hipMalloc(&out, 4096));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 16384, wps = 8, blocks = 256 * wps;
    for (int rep = 0; rep < 4; rep++) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const double inst = (double)wps * iters * UNROLL;     // loop-body VALU wave-instructions per SIMD
        printf("CAL %s kind=%d launch=%d ms=%.4f body_insts_per_simd=%.0f ns_per_inst=%.4f\n", name, KIND, rep, ms, inst,
               ms * 1e6 / inst);
    }
}

int main(int argc, char **argv) {
    setvbuf(stdout, NULL, _IONBF, 0);
    if (argc > 1 && !strcmp(argv[1], "--pmc")) {
        run_pmc<0>("v_fma_f32");
        run_pmc<1>("v_pk_fma_f32");
        return 0;
    }
    run<0>("v_fma_f32", 1);
    run<1>("v_pk_fma_f32", 1);
    run<2>("v_mul_f32", 1);
    run<3>("v_pk_mul_f32", 1);
    run<9>("v_pk_add_f32", 1);
    run<4>("v_min_f32", 1);
    run<5>("v_cmp_ge_f32 + v_cndmask (2)", 2);
    run<6>("v_lshl_add_u32", 1);
    run<7>("v_exp_f32", 1);
    run<8>("v_cmp -> sgpr + s_and_b64 (2)", 2);
    run<10>("v_sub_f32", 1);
    run<11>("v_cmp_ge_f32 -> sgpr pair", 1);
    run<15>("v_cmp_ge_f32 -> vcc", 1);
    run<12>("v_cndmask_b32 (vcc)", 1);
    run<13>("v_mov_b64", 1);
    run<14>("v_and_b32", 1);
    run<16>("v_pk_fma_f32 dependent + s_nop", 1);
    run<17>("v_pk_fma_f32 sgpr operand", 1);
    printf("(ns per wave-instruction per SIMD; at 2.4 GHz one cycle = 0.417 ns)\n");
    return 0;
}
