// Microbenchmark: what does a burst of global stores cost the issuing wave on gfx950?
// Each wave issues NST stores of 16 B/lane back to back; patterns: contiguous 1 KB per store,
// or 8 rows x 128 B at a row stride (the GEMM epilogue shape).  Reports cycles per store.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int NST>
__global__ __launch_bounds__(512) void store_kernel(uint4* out, long row_stride_bytes, int mode, int waves_active,
                                                    unsigned long long* cyc, int lds_wait) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (wave >= waves_active) return;
    const long wid = (long)blockIdx.x * 8 + wave;
    char* base = (char*)out;
    uint4 v = make_uint4(lane, wave, blockIdx.x, 7);
    __syncthreads();
    const unsigned long long t0 = clock64();
#pragma unroll
    for (int i = 0; i < NST; ++i) {
        char* p;
        if (mode == 0) p = base + (wid * NST + i) * 1024 + lane * 16;                       // contiguous 1 KB
        else {                                                                             // 8 rows x 128 B
            const long row = wid * (NST * 8) + i * 8 + (lane >> 3);
            p = base + row * row_stride_bytes + (lane & 7) * 16;
        }
        *(uint4*)p = v;
        v.x += 1;
    }
    const unsigned long long t1 = clock64();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t2 = clock64();
    if (lane == 0) { cyc[wid * 2] = t1 - t0; cyc[wid * 2 + 1] = t2 - t0; }
}

int main() {
    const int NST = 32;
    const long rows = 256L * 8 * NST * 8;
    const long stride = 4608;
    uint4* out; unsigned long long* cyc;
    CHECK(hipMalloc(&out, rows * stride + (1 << 20)));
    CHECK(hipMalloc(&cyc, 256 * 8 * 2 * sizeof(unsigned long long)));
    std::vector<unsigned long long> h(256 * 8 * 2);
    for (int mode = 0; mode < 2; ++mode)
        for (int blocks : {256, 8})
            for (int waves : {8, 1}) {
                for (int rep = 0; rep < 3; ++rep) {
                    hipLaunchKernelGGL(store_kernel<NST>, dim3(blocks), dim3(512), 0, 0, out, stride, mode, waves, cyc, 0);
                    CHECK(hipDeviceSynchronize());
                }
                CHECK(hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost));
                double a = 0, b = 0; int n = 0;
                for (int bl = 0; bl < blocks; ++bl) for (int w = 0; w < waves; ++w) { a += h[(bl * 8 + w) * 2]; b += h[(bl * 8 + w) * 2 + 1]; ++n; }
                printf("mode %d (%s) blocks %3d waves/blk %d: issue %7.0f cyc/wave = %6.1f per store; to vmcnt(0) %7.0f cyc = %6.1f per store; %.1f B/clk/CU\n",
                       mode, mode ? "8 rows x128B, stride 4608" : "contiguous 1KB", blocks, waves, a / n, a / n / NST, b / n, b / n / NST,
                       (double)NST * 1024 * waves / (b / n));
            }
    return 0;
}
