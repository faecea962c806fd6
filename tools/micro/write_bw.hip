// Microbenchmark: sustained HBM write / read / copy bandwidth on gfx950 for buffers far beyond the 256 MB Infinity Cache.
// 16 B per lane, grid-stride; stores plain or non-temporal.  Standalone: hipcc --offload-arch=gfx950 -O3 write_bw.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef float v4 __attribute__((ext_vector_type(4)));

template <int NT>
__global__ __launch_bounds__(256) void write_kernel(v4* out, size_t n) {
    v4 v = {1.f, 2.f, 3.f, (float)threadIdx.x};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        if (NT) __builtin_nontemporal_store(v, out + i); else out[i] = v;
    }
}
__global__ __launch_bounds__(256) void read_kernel(const v4* in, size_t n, float* sink) {
    v4 acc = {0, 0, 0, 0};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) acc += in[i];
    if (acc.x + acc.y + acc.z + acc.w == 123.456f) *sink = 1.f;
}
template <int NT>
__global__ __launch_bounds__(256) void copy_kernel(const v4* in, v4* out, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        v4 v = in[i];
        if (NT) __builtin_nontemporal_store(v, out + i); else out[i] = v;
    }
}
int main() {
    const size_t bytes = 1ull << 30, n = bytes / 16;
    v4 *a, *b; float* sink;
    CHECK(hipMalloc(&a, bytes)); CHECK(hipMalloc(&b, bytes)); CHECK(hipMalloc(&sink, 4));
    CHECK(hipMemset(a, 0, bytes)); CHECK(hipMemset(b, 0, bytes));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int grid : {2048, 8192, 65536}) {
        for (int which = 0; which < 5; ++which) {
            float best = 1e9;
            for (int rep = 0; rep < 5; ++rep) {
                CHECK(hipEventRecord(e0));
                switch (which) {
                    case 0: hipLaunchKernelGGL(write_kernel<0>, dim3(grid), dim3(256), 0, 0, a, n); break;
                    case 1: hipLaunchKernelGGL(write_kernel<1>, dim3(grid), dim3(256), 0, 0, a, n); break;
                    case 2: hipLaunchKernelGGL(read_kernel, dim3(grid), dim3(256), 0, 0, a, n, sink); break;
                    case 3: hipLaunchKernelGGL(copy_kernel<0>, dim3(grid), dim3(256), 0, 0, a, b, n); break;
                    case 4: hipLaunchKernelGGL(copy_kernel<1>, dim3(grid), dim3(256), 0, 0, a, b, n); break;
                }
                CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
                float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
                if (ms < best) best = ms;
            }
            const char* nm[] = {"write plain", "write nt", "read", "copy plain (r+w)", "copy nt (r+w)"};
            const double moved = (which >= 3 ? 2.0 : 1.0) * bytes;
            printf("grid %6d  %-18s %7.1f us  %6.2f TB/s\n", grid, nm[which], best * 1e3, moved / (best * 1e-3) / 1e12);
        }
    }
    return 0;
}
