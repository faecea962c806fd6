// Do the matrix pipe and the vector ALU of one gfx950 SIMD overlap (a) between two waves, (b) inside one wave when the
// instruction stream is software-pipelined by hand?  Round 3's `valu_rate` chain (4 MFMA -> 16 exp + 32 mul + 8 cvt on
// their results -> 2 MFMA) ran at the SUM of its matrix and vector time at every occupancy; this program separates the
// two questions, because the answer decides how the attention kernels have to be written.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/pipe_overlap.hip -o tools/micro/pipe_overlap && tools/micro/pipe_overlap
// One workgroup on one CU (no power effects).  Cycles from s_memtime.
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

enum {
    X_MFMA_ONLY,      // waves 0-3: 8 MFMAs per trip; waves 4-7: idle (exit at once)
    X_MUL_ONLY,       // waves 4-7: 64 v_mul per trip; waves 0-3 exit
    X_EXP_ONLY,       // waves 4-7: 32 v_exp per trip
    X_MFMA_MUL,       // both: do they overlap across the two waves of a SIMD?
    X_MFMA_EXP,
    C_PLAIN,          // the chain, as the compiler schedules it
    C_SWP,            // the chain software-pipelined: step i + 1's four MFMAs spread between step i's vector work
    C_SWP_SGB,        // ... with sched_group_barrier forcing 1 MFMA : N VALU interleave
    N_MODES
};

__device__ __forceinline__ void valu_part(const f32x16& s0, const f32x16& s1, f16x8& p0, f16x8& p1) {
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const float pe = __builtin_amdgcn_exp2f(s0[e] * 0.125f) * s1[e];
        if (e < 8) p0[e] = (_Float16)pe; else p1[e - 8] = (_Float16)pe;
    }
}

template <int MODE>
__global__ __launch_bounds__(1024) void k(unsigned long long* out, float* sink, int iters) {
    float a[16], r[16];
    f32x16 acc0 = {0}, acc1 = {0};
    f16x8 fa, fb;
#pragma unroll
    for (int i = 0; i < 16; ++i) { a[i] = threadIdx.x * 0.001f + i * 0.01f; r[i] = 0.f; }
#pragma unroll
    for (int i = 0; i < 8; ++i) { fa[i] = (_Float16)(0.01f * i); fb[i] = (_Float16)(0.02f * i); }
    const int wave = threadIdx.x >> 6;
    const bool grp_b = ((wave >> 2) & 1) != 0;          // waves w and w + 4 share a SIMD
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    if (MODE <= X_MFMA_EXP) {
        const bool do_mfma = MODE == X_MFMA_ONLY || MODE == X_MFMA_MUL || MODE == X_MFMA_EXP;
        const bool do_mul = MODE == X_MUL_ONLY || MODE == X_MFMA_MUL;
        const bool do_exp = MODE == X_EXP_ONLY || MODE == X_MFMA_EXP;
        if (!grp_b) {
            if (do_mfma)
                for (int it = 0; it < iters; ++it) {
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, fb, acc0, 0, 0, 0);
                        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, fb, acc1, 0, 0, 0);
                    }
                }
        } else {
            if (do_mul)
                for (int it = 0; it < iters; ++it) {
#pragma unroll
                    for (int rep = 0; rep < 4; ++rep) {
#define X(i) asm volatile("v_mul_f32 %0, %1, %1" : "=v"(r[i]) : "v"(a[i]));
                        REP16(X)
#undef X
                    }
                }
            if (do_exp)
                for (int it = 0; it < iters; ++it) {
#pragma unroll
                    for (int rep = 0; rep < 2; ++rep) {
#define X(i) asm volatile("v_exp_f32 %0, %1" : "=v"(r[i]) : "v"(a[i]));
                        REP16(X)
#undef X
                    }
                }
        }
    } else if (MODE == C_PLAIN) {
        for (int it = 0; it < iters; ++it) {
            asm volatile("" : "+v"(fa), "+v"(fb));
            f32x16 s0 = {0}, s1 = {0};
            s0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, fb, s0, 0, 0, 0);
            s1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(fb, fa, s1, 0, 0, 0);
            s0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(fb, fa, s0, 0, 0, 0);
            s1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, fb, s1, 0, 0, 0);
            f16x8 p0, p1;
            valu_part(s0, s1, p0, p1);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, p0, acc0, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(fb, p1, acc0, 0, 0, 0);
        }
    } else {
        // software pipeline: s0 / s1 of the NEXT step are produced while the vector work of the CURRENT step runs.
        // C_SWP: source order only (the compiler may re-cluster); C_SWP_SGB: every group fenced by sched_barrier(0), i.e.
        // the stream is M v v M v v M v v M v v ... exactly as written (hazard nops still come from the compiler).
        f32x16 s0 = {0}, s1 = {0};
        s0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, fb, s0, 0, 0, 0);
        s1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(fb, fa, s1, 0, 0, 0);
        s0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(fb, fa, s0, 0, 0, 0);
        s1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, fb, s1, 0, 0, 0);
#define FENCE() do { if (MODE == C_SWP_SGB) __builtin_amdgcn_sched_barrier(0); } while (0)
#define SLICE(q)                                                                                          \
    do {                                                                                                  \
        _Pragma("unroll") for (int e = 4 * (q); e < 4 * (q) + 4; ++e) {                                   \
            const float pe = __builtin_amdgcn_exp2f(s0[e] * 0.125f) * s1[e];                              \
            if (e < 8) p0[e] = (_Float16)pe; else p1[e - 8] = (_Float16)pe;                               \
        }                                                                                                 \
    } while (0)
        for (int it = 0; it < iters; ++it) {
            asm volatile("" : "+v"(fa), "+v"(fb));          // loop-variant operands: nothing can be hoisted
            f32x16 n0 = {0}, n1 = {0};
            f16x8 p0, p1;
            n0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, fb, n0, 0, 0, 0); FENCE();
            SLICE(0); FENCE();
            n1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(fb, fa, n1, 0, 0, 0); FENCE();
            SLICE(1); FENCE();
            n0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(fb, fa, n0, 0, 0, 0); FENCE();
            SLICE(2); FENCE();
            n1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, fb, n1, 0, 0, 0); FENCE();
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, p0, acc0, 0, 0, 0); FENCE();
            SLICE(3); FENCE();
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(fb, p1, acc0, 0, 0, 0); FENCE();
            s0 = n0; s1 = n1;
        }
        acc1 = s0 + s1;
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += r[i] + acc0[i] + acc1[i];
    sink[threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) out[wave] = t1 - t0;
}

template <int MODE>
void run(const char* name, int waves, unsigned long long* out, float* sink, int grid = 1) {
    const int iters = 20000;
    (void)hipGetLastError();
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(64 * waves), 0, 0, out, sink, 100);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(64 * waves), 0, 0, out, sink, iters);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[16];
    hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
    unsigned long long mx = 0;
    for (int w = 0; w < waves; ++w) mx = h[w] > mx ? h[w] : mx;
    const double per_simd = waves / 4.0;
    printf("%-44s grid %4d %2d waves (%.0f / SIMD): %8.1f cycles per trip per wave, %8.1f per trip and SIMD-wave, %7.1f ns wall per trip and SIMD-wave\n",
           name, grid, waves, per_simd, (double)mx / iters, (double)mx / iters / per_simd, ms * 1e6 / iters / per_simd);
}

int main() {
    unsigned long long* out;
    float* sink;
    hipMalloc(&out, 16 * sizeof(unsigned long long));
    hipMalloc(&sink, 1024 * sizeof(float));
    run<X_MFMA_ONLY>("A: 8 mfma (B idle)", 8, out, sink);
    run<X_MUL_ONLY>("B: 64 v_mul (A idle)", 8, out, sink);
    run<X_EXP_ONLY>("B: 32 v_exp (A idle)", 8, out, sink);
    run<X_MFMA_MUL>("A: 8 mfma | B: 64 v_mul, same SIMD", 8, out, sink);
    run<X_MFMA_EXP>("A: 8 mfma | B: 32 v_exp, same SIMD", 8, out, sink);
    for (int waves : {4, 8, 12, 16}) {
        run<C_PLAIN>("chain, compiler order", waves, out, sink);
        run<C_SWP>("chain, software-pipelined (source order)", waves, out, sink);
        run<C_SWP_SGB>("chain, software-pipelined, fenced groups", waves, out, sink);
    }
    // every CU busy (the clock the chip holds under load): one workgroup per CU
    for (int waves : {4, 8, 16}) {
        run<C_PLAIN>("chain, compiler order", waves, out, sink, 256);
        run<C_SWP_SGB>("chain, software-pipelined, fenced groups", waves, out, sink, 256);
    }
    return 0;
}
