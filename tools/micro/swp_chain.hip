// The attention key step's instruction mix -- 4 MFMA -> 16 x (mul, exp, mul) + 8 cvt on their results -> 2 MFMA -- as one
// hand-placed inline-assembly loop (tools/micro/gen_swp_chain.py), in several placements of the MFMAs among the vector
// instructions, at 1 / 2 / 4 waves per SIMD, with and without a start stagger of the second half of the waves.
// Question: is the step's time the SUM of its matrix (6 x 32 cycles) and vector (56 instructions) parts whatever one does
// (round 3's finding with compiler-scheduled code), or can a software-pipelined stream overlap them?
//   python3 tools/micro/gen_swp_chain.py && hipcc --offload-arch=gfx950 -O3 tools/micro/swp_chain.hip -o tools/micro/swp_chain
#include <hip/hip_runtime.h>
#include <stdio.h>
#include "swp_chain_gen.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int P>
__global__ __launch_bounds__(1024) void k(unsigned long long* out, float* sink, int iters, int stagger) {
    f32x4 fa, fb;                                   // 8 halves each (bit patterns of small numbers)
    for (int i = 0; i < 4; ++i) { fa[i] = __int_as_float(0x2c002c00 + i); fb[i] = __int_as_float(0x2e002e00 + i); }
    const int wave = threadIdx.x >> 6;
    int n = iters;
    __syncthreads();
    if (stagger && (wave & 4)) for (int i = 0; i < stagger; ++i) __builtin_amdgcn_s_sleep(1);       // ~64 cycles per unit
    const unsigned long long t0 = __builtin_readcyclecounter();
    if (P == 0)
        asm volatile(SWP_BODY_PLAIN : "+s"(n) : "{v[0:3]}"(fa), "{v[4:7]}"(fb) : SWP_CLOBBERS, "scc");
    else if (P == 1)
        asm volatile(SWP_BODY_SWP_TOP : "+s"(n) : "{v[0:3]}"(fa), "{v[4:7]}"(fb) : SWP_CLOBBERS, "scc");
    else if (P == 2)
        asm volatile(SWP_BODY_SWP_SPREAD : "+s"(n) : "{v[0:3]}"(fa), "{v[4:7]}"(fb) : SWP_CLOBBERS, "scc");
    else
        asm volatile(SWP_BODY_SWP_SPREAD2 : "+s"(n) : "{v[0:3]}"(fa), "{v[4:7]}"(fb) : SWP_CLOBBERS, "scc");
    const unsigned long long t1 = __builtin_readcyclecounter();
    sink[threadIdx.x] = fa[0] + n;
    if ((threadIdx.x & 63) == 0) out[wave] = t1 - t0;
}

template <int P>
void run(const char* name, int waves, int stagger, unsigned long long* out, float* sink, int grid) {
    const int iters = 10000;            // 2 steps per iteration
    (void)hipGetLastError();
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<P>, dim3(grid), dim3(64 * waves), 0, 0, out, sink, 100, stagger);
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k<P>, dim3(grid), dim3(64 * waves), 0, 0, out, sink, iters, stagger);
    (void)hipEventRecord(e1, 0);
    (void)hipDeviceSynchronize();
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[16];
    (void)hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
    unsigned long long mx = 0;
    for (int w = 0; w < waves; ++w) mx = h[w] > mx ? h[w] : mx;
    const double per_simd = waves / 4.0;
    printf("%-12s grid %3d, %2d waves (%.0f / SIMD), stagger %2d: %7.1f cycles per step and SIMD-wave, %6.1f ns wall\n", name, grid, waves,
           per_simd, stagger, (double)mx / (2.0 * iters) / per_simd, ms * 1e6 / (2.0 * iters) / per_simd);
}

int main() {
    unsigned long long* out;
    float* sink;
    (void)hipMalloc(&out, 16 * sizeof(unsigned long long));
    (void)hipMalloc(&sink, 1024 * sizeof(float));
    for (int grid : {1, 256})
        for (int waves : {4, 8, 16})
            for (int stagger : {0, 5}) {
                if (waves == 4 && stagger) continue;
                run<0>("plain", waves, stagger, out, sink, grid);
                run<1>("swp_top", waves, stagger, out, sink, grid);
                run<2>("swp_spread", waves, stagger, out, sink, grid);
                run<3>("swp_spread2", waves, stagger, out, sink, grid);
            }
    return 0;
}
