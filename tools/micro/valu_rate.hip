// Issue cost, in shader-clock cycles per wave instruction, of the VALU / MFMA instructions the attention kernels are
// made of, and whether the matrix pipe and the VALU overlap (within a wave, and between two waves of a SIMD):
//   hipcc --offload-arch=gfx950 -O3 tools/micro/valu_rate.hip -o tools/micro/valu_rate && tools/micro/valu_rate
// One workgroup on one CU (no power effects); W waves per SIMD; every instruction writes its own register (no
// dependent chains); cycles from s_memtime around 64 x 16 x ITERS instructions.
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

enum { M_MUL, M_PKMUL, M_PKFMA, M_EXP, M_CVT, M_MAX3, M_MFMA, M_MFMA_EXP, M_MFMA_MUL, M_SPLIT, M_CHAIN, M_MFMA_MUL_AGPR, M_MFMA_AGPR, N_MODES };

template <int MODE>
__global__ __launch_bounds__(1024) void k(unsigned long long* out, float* sink, int iters) {
    float a[16], r[16];
    f32x2 a2[8], r2[8];
    f32x16 acc0 = {0}, acc1 = {0};
    f16x8 fa, fb;
#pragma unroll
    for (int i = 0; i < 16; ++i) { a[i] = threadIdx.x * 0.001f + i; r[i] = 0.f; a2[i >> 1][i & 1] = a[i]; r2[i >> 1][i & 1] = 0.f; }
#pragma unroll
    for (int i = 0; i < 8; ++i) { fa[i] = (_Float16)(0.01f * i); fb[i] = (_Float16)(0.02f * i); }
    const int wave = threadIdx.x >> 6;
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        if (MODE == M_MUL) {
#define X(i) asm volatile("v_mul_f32 %0, %1, %1" : "=v"(r[i]) : "v"(a[i]));
            REP16(X)
#undef X
        } else if (MODE == M_PKMUL) {
#define X(i) asm volatile("v_pk_mul_f32 %0, %1, %1" : "=v"(r2[(i) & 7]) : "v"(a2[(i) & 7]));
            REP16(X)
#undef X
        } else if (MODE == M_PKFMA) {
#define X(i) asm volatile("v_pk_fma_f32 %0, %1, %1, %1" : "=v"(r2[(i) & 7]) : "v"(a2[(i) & 7]));
            REP16(X)
#undef X
        } else if (MODE == M_EXP) {
#define X(i) asm volatile("v_exp_f32 %0, %1" : "=v"(r[i]) : "v"(a[i]));
            REP16(X)
#undef X
        } else if (MODE == M_CVT) {
#define X(i) asm volatile("v_cvt_pk_f16_f32 %0, %1, %1" : "=v"(r[i]) : "v"(a[i]));
            REP16(X)
#undef X
        } else if (MODE == M_MAX3) {
#define X(i) asm volatile("v_max3_f32 %0, %1, %1, %1" : "=v"(r[i]) : "v"(a[i]));
            REP16(X)
#undef X
        } else if (MODE == M_MFMA) {        // 16 MFMAs, two independent accumulator chains
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, fb, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, fb, acc1, 0, 0, 0);
            }
        } else if (MODE == M_MFMA_EXP || MODE == M_MFMA_MUL) {   // one wave: an MFMA, then 2 exp (or 8 mul) behind it, 8 times
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (u & 1) acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, fb, acc1, 0, 0, 0);
                else acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, fb, acc0, 0, 0, 0);
                if (MODE == M_MFMA_EXP) {
                    asm volatile("v_exp_f32 %0, %1" : "=v"(r[2 * u]) : "v"(a[2 * u]));
                    asm volatile("v_exp_f32 %0, %1" : "=v"(r[2 * u + 1]) : "v"(a[2 * u + 1]));
                } else {
#pragma unroll
                    for (int v = 0; v < 8; ++v) asm volatile("v_mul_f32 %0, %1, %1" : "=v"(r[(2 * u + v) & 15]) : "v"(a[(2 * u + v) & 15]));
                }
            }
        } else if (MODE == M_MFMA_MUL_AGPR || MODE == M_MFMA_AGPR) {   // the same with the accumulators in AGPRs (inline asm)
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (u & 1) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc1) : "v"(fa), "v"(fb));
                else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc0) : "v"(fa), "v"(fb));
                if (MODE == M_MFMA_MUL_AGPR) {
#pragma unroll
                    for (int v = 0; v < 8; ++v) asm volatile("v_mul_f32 %0, %1, %1" : "=v"(r[(2 * u + v) & 15]) : "v"(a[(2 * u + v) & 15]));
                }
            }
        } else if (MODE == M_CHAIN) {       // the backward key step's shape: 4 MFMAs (two chains of two) -> 16 exp + 16 mul + 8 cvt on
                                            // their results -> 2 MFMAs fed by the converted values, accumulating across trips
            f32x16 s0 = {0}, s1 = {0};
            s0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, fb, s0, 0, 0, 0);
            s1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(fb, fa, s1, 0, 0, 0);
            s0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(fb, fa, s0, 0, 0, 0);
            s1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, fb, s1, 0, 0, 0);
            f16x8 p0, p1;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const float pe = __builtin_amdgcn_exp2f(s0[e] * 0.125f) * s1[e];
                if (e < 8) p0[e] = (_Float16)pe; else p1[e - 8] = (_Float16)pe;
            }
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, p0, acc0, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(fb, p1, acc0, 0, 0, 0);
        } else if (MODE == M_SPLIT) {       // even waves: 8 MFMAs; odd waves: 16 exp -- do two waves of a SIMD overlap?
            if (((wave >> 2) & 1) == 0) {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, fb, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, fb, acc1, 0, 0, 0);
                }
            } else {
#define X(i) asm volatile("v_exp_f32 %0, %1" : "=v"(r[i]) : "v"(a[i]));
                REP16(X)
#undef X
            }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += r[i] + acc0[i] + acc1[i] + r2[i >> 1][i & 1];
    sink[threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) out[wave] = t1 - t0;
}

template <int MODE>
void run(const char* name, int waves, unsigned long long* out, float* sink, double per_iter_instr) {
    const int iters = 20000;
    hipLaunchKernelGGL(k<MODE>, dim3(1), dim3(64 * waves), 0, 0, out, sink, 100);
    hipLaunchKernelGGL(k<MODE>, dim3(1), dim3(64 * waves), 0, 0, out, sink, iters);
    hipDeviceSynchronize();
    unsigned long long h[16];
    hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
    unsigned long long mx = 0;
    for (int w = 0; w < waves; ++w) mx = h[w] > mx ? h[w] : mx;
    const double per_simd = waves / 4.0;    // waves per SIMD
    printf("%-34s %2d waves (%.0f / SIMD): %8.1f cycles per loop trip per wave, %6.2f per instruction and SIMD\n", name, waves,
           per_simd, (double)mx / iters, (double)mx / iters / (per_iter_instr * (per_simd < 1 ? 1 : per_simd)));
}

int main() {
    unsigned long long* out;
    float* sink;
    hipMalloc(&out, 16 * sizeof(unsigned long long));
    hipMalloc(&sink, 1024 * sizeof(float));
    for (int waves : {4, 8, 16}) {
        run<M_MUL>("v_mul_f32 x16", waves, out, sink, 16);
        run<M_PKMUL>("v_pk_mul_f32 x16", waves, out, sink, 16);
        run<M_PKFMA>("v_pk_fma_f32 x16", waves, out, sink, 16);
        run<M_EXP>("v_exp_f32 x16", waves, out, sink, 16);
        run<M_CVT>("v_cvt_pk_f16_f32 x16", waves, out, sink, 16);
        run<M_MAX3>("v_max3_f32 x16", waves, out, sink, 16);
        run<M_MFMA>("v_mfma_f32_32x32x16_f16 x16", waves, out, sink, 16);
        run<M_MFMA_EXP>("8 x (mfma + 2 exp), one wave", waves, out, sink, 8);
        run<M_MFMA_MUL>("8 x (mfma + 8 mul), one wave", waves, out, sink, 8);
    }
    for (int waves : {4, 8, 16}) {
        run<M_MFMA_AGPR>("8 x mfma, AGPR accumulators", waves, out, sink, 8);
        run<M_MFMA_MUL_AGPR>("8 x (mfma + 8 mul), AGPR accumulators", waves, out, sink, 8);
    }
    for (int waves : {4, 8, 12, 16})
        run<M_CHAIN>("key step: 4 mfma > 16 exp+16 mul+8 cvt > 2 mfma", waves, out, sink, 1);
    run<M_SPLIT>("8 mfma | 16 exp on paired waves", 8, out, sink, 1);
    run<M_SPLIT>("8 mfma | 16 exp on paired waves", 16, out, sink, 1);
    return 0;
}
