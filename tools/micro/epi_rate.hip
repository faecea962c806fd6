// Issue cost of the instructions the 16-bit GEMM epilogue converts with (round 4): is v_fma_mixlo/hi_f16 -- what hipcc
// fuses fmaf + (half) into -- slower than v_fma_f32 + v_cvt_pk_f16_f32?
//   hipcc --offload-arch=gfx950 -O3 tools/micro/epi_rate.hip -o tools/micro/epi_rate && tools/micro/epi_rate
// One workgroup on one CU, 2 waves per SIMD (the persistent GEMM's occupancy), independent destination registers.
#include <hip/hip_runtime.h>
#include <stdio.h>
#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)
enum { M_FMA, M_MIXLO, M_MIXHI, M_CVTPK_F16, M_CVTPK_BF16, M_DSW64, N_MODES };
template <int MODE>
__global__ __launch_bounds__(512) void k(unsigned long long* out, float* sink, int iters) {
    __shared__ unsigned long long lds[2048];
    float a[16], r[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) { a[i] = threadIdx.x * 0.001f + i; r[i] = 0.f; }
    const unsigned addr = (threadIdx.x & 63) * 8 + (threadIdx.x >> 6) * 2048 / 8 * 8;
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        if (MODE == M_FMA) {
#define X(i) asm volatile("v_fma_f32 %0, %1, %1, %1" : "=v"(r[i]) : "v"(a[i]));
            REP16(X)
#undef X
        } else if (MODE == M_MIXLO) {
#define X(i) asm volatile("v_fma_mixlo_f16 %0, %1, %1, %1" : "=v"(r[i]) : "v"(a[i]));
            REP16(X)
#undef X
        } else if (MODE == M_MIXHI) {
#define X(i) asm volatile("v_fma_mixhi_f16 %0, %1, %1, %1" : "+v"(r[i]) : "v"(a[i]));
            REP16(X)
#undef X
        } else if (MODE == M_CVTPK_F16) {
#define X(i) asm volatile("v_cvt_pk_f16_f32 %0, %1, %1" : "=v"(r[i]) : "v"(a[i]));
            REP16(X)
#undef X
        } else if (MODE == M_CVTPK_BF16) {
#define X(i) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %1" : "=v"(r[i]) : "v"(a[i]));
            REP16(X)
#undef X
        } else if (MODE == M_DSW64) {
#define X(i) asm volatile("ds_write_b64 %0, %1 offset:%2" :: "v"(addr), "v"(*(unsigned long long*)&a[(i) & 14]), "n"((i) * 512) : "memory");
            REP16(X)
#undef X
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = (float)lds[threadIdx.x & 7];
#pragma unroll
    for (int i = 0; i < 16; ++i) s += r[i];
    sink[threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) out[threadIdx.x >> 6] = t1 - t0;
}
template <int MODE> void run(const char* name, unsigned long long* out, float* sink) {
    const int iters = 20000;
    hipLaunchKernelGGL(k<MODE>, dim3(1), dim3(512), 0, 0, out, sink, 100);
    hipLaunchKernelGGL(k<MODE>, dim3(1), dim3(512), 0, 0, out, sink, iters);
    hipDeviceSynchronize();
    unsigned long long h[8], mx = 0;
    hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
    for (int w = 0; w < 8; ++w) mx = h[w] > mx ? h[w] : mx;
    printf("%-22s 8 waves (2 / SIMD): %6.2f cycles per instruction and SIMD\n", name, (double)mx / iters / 32.0);
}
int main() {
    unsigned long long* out; float* sink;
    hipMalloc(&out, 64); hipMalloc(&sink, 4096);
    run<M_FMA>("v_fma_f32", out, sink);
    run<M_MIXLO>("v_fma_mixlo_f16", out, sink);
    run<M_MIXHI>("v_fma_mixhi_f16", out, sink);
    run<M_CVTPK_F16>("v_cvt_pk_f16_f32", out, sink);
    run<M_CVTPK_BF16>("v_cvt_pk_bf16_f32", out, sink);
    run<M_DSW64>("ds_write_b64", out, sink);
    return 0;
}
