// Issue rate of the fp32 -> 16-bit packed conversions on gfx950 (one wave per SIMD is enough to see the rate):
//   hipcc --offload-arch=gfx950 -O3 tools/micro/cvt_rate.hip -o tools/micro/cvt_rate && tools/micro/cvt_rate
// Round 3 question: the fp16 kernels run 3-12 % behind their bf16 twins with identical instruction mixes -- is
// v_cvt_pk_f16_f32 slower than v_cvt_pk_bf16_f32?
#include <hip/hip_runtime.h>
#include <stdio.h>

template <int MODE>
__global__ void k(float* out, int iters) {
    float a = threadIdx.x * 0.001f + 1.0f, b = a + 0.5f;
    unsigned acc = 0;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            unsigned r;
            if (MODE == 0) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
            else if (MODE == 1) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
            else if (MODE == 2) asm volatile("v_cvt_pkrtz_f16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
            else if (MODE == 3) asm volatile("v_fma_mixlo_f16 %0, %1, 1.0, 0\n\tv_fma_mixhi_f16 %0, %2, 1.0, 0" : "=&v"(r) : "v"(a), "v"(b));
            else if (MODE == 4) { asm volatile("v_mul_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); }
            else { asm volatile("v_exp_f32 %0, %1" : "=v"(r) : "v"(a)); }
            acc ^= r;
            a += 1.0f;
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = (float)acc + a;
}

template <int MODE>
double run(float* out, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(256 * 4), dim3(64), 0, 0, out, 16);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(256 * 4), dim3(64), 0, 0, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main() {
    float* out;
    hipMalloc(&out, 256 * 4 * 64 * sizeof(float));
    const int iters = 200000;
    const char* names[] = {"v_cvt_pk_bf16_f32", "v_cvt_pk_f16_f32", "v_cvt_pkrtz_f16_f32", "v_fma_mixlo+mixhi_f16", "v_mul_f32 (ref)", "v_exp_f32 (ref)"};
    double t[6] = {run<0>(out, iters), run<1>(out, iters), run<2>(out, iters), run<3>(out, iters), run<4>(out, iters), run<5>(out, iters)};
    for (int i = 0; i < 6; ++i)
        printf("%-24s %8.2f ms  -> %.2f ns per wave-instruction group (16 x %d iterations, one wave per SIMD)\n", names[i], t[i],
               t[i] * 1e6 / (16.0 * iters), iters);
    return 0;
}
