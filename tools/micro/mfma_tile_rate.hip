// Sustained rate of a 128 x 64 register tile per wave (the persistent GEMM's) built from 16x16x32 or from 32x32x16 MFMAs,
// operands in registers (no LDS, no memory): cycles per MFMA as the wave sees them, and TFLOP/s with every CU busy.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_tile_rate.hip -o tools/micro/mfma_tile_rate && tools/micro/mfma_tile_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int MODE>
__global__ __launch_bounds__(512, 1) void k(const uint4* __restrict__ in, float* out, unsigned long long* cyc, int iters) {
    // K = 64 per trip: 16x16x32 -> A 8 row blocks x 2 k blocks, B 4 x 2, 64 MFMAs; 32x32x16 -> A 4 x 4, B 2 x 4, 32 MFMAs
    f16x8 a[16], b[8];
#pragma unroll
    for (int i = 0; i < 16; ++i) { uint4 v = in[(threadIdx.x + i * 64) & 1023]; a[i] = *(f16x8*)&v; }
#pragma unroll
    for (int i = 0; i < 8; ++i) { uint4 v = in[(threadIdx.x * 3 + i * 64 + 17) & 1023]; b[i] = *(f16x8*)&v; }
    f32x4 c4[8][4] = {};
    f32x16 c16[4][2] = {};
    const unsigned long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int i = 0; i < 8; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        c4[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[kb * 8 + i], b[kb * 4 + j], c4[i][j], 0, 0, 0);
        } else {
#pragma unroll
            for (int kb = 0; kb < 4; ++kb)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        c16[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[kb * 4 + i], b[kb * 2 + j], c16[i][j], 0, 0, 0);
        }
        asm volatile("" : "+v"(a[0]), "+v"(b[0]));
    }
    const unsigned long long t1 = clock64();
    float s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) s += c4[i][j][0];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) s += c16[i][j][0];
    out[blockIdx.x * 512 + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) cyc[0] = t1 - t0;
}

template <int MODE>
void run(const uint4* in, float* out, unsigned long long* cyc, int iters, const char* name, int waves_per_simd) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int threads = 256 * waves_per_simd;
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, in, out, cyc, 2000);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, in, out, cyc, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c = 0;
    (void)hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    const double flop = 256.0 * 4 * waves_per_simd * (double)iters * 2.0 * 128 * 64 * 64;
    const int per_trip = MODE == 0 ? 64 : 32;
    printf("%-9s %d wave(s)/SIMD: %7.2f ms %7.1f TFLOP/s | %6.1f cycles per MFMA and wave -> %5.1f per MFMA and SIMD (ideal %d) | clock %.2f GHz\n", name,
           waves_per_simd, ms, flop / ms * 1e-9, (double)c / iters / per_trip, (double)c / iters / per_trip / waves_per_simd,
           MODE == 0 ? 16 : 32, (double)c / (ms * 1e6));
}

int main() {
    uint4 h[1024];
    unsigned seed = 12345;
    for (int pass = 0; pass < 2; ++pass) {
        for (int i = 0; i < 1024; ++i) {
            unsigned w[4];
            for (int q = 0; q < 4; ++q) { seed = seed * 1664525u + 1013904223u; w[q] = 0x3c003c00u | (pass ? (seed & 0x03ff03ffu) : 0u); }
            h[i] = make_uint4(w[0], w[1], w[2], w[3]);
        }
        uint4* in; float* out; unsigned long long* cyc;
        (void)hipMalloc(&in, sizeof(h)); (void)hipMalloc(&out, 256 * 512 * 4); (void)hipMalloc(&cyc, 8);
        (void)hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
        printf("%s mantissas\n", pass ? "random" : "zero");
        for (int w = 1; w <= 2; ++w) {
            run<0>(in, out, cyc, 40000, "16x16x32", w);
            run<1>(in, out, cyc, 40000, "32x32x16", w);
        }
        (void)hipFree(in); (void)hipFree(out); (void)hipFree(cyc);
    }
    return 0;
}
