// Microbenchmark: is a FOUR-wave 256 x 256 tile (128 x 128 per wave, one wave per SIMD, 256 accumulators) with register
// staging worth building for the long-K GEMMs of the step?  C[M,N] (fp32) = A[M,K] * B[N,K]^T, bf16 operands.
//   * 3 LDS stages of (A 256 x 32 | B 256 x 32) bf16 = 32 KB each, 64-byte rows, XOR swizzle (kc32) as in gemm.hip
//   * per K = 32 stage and wave: 64 MFMA 16x16x32, 16 ds_read_b128 (fragments of stage s+1 into a second register set),
//     8 global_load_dwordx4 (stage s+3 into staging registers), 8 ds_write_b128 (stage s+2), ONE barrier
//   * everything but the barrier is interleaved with the MFMAs (sched_group_barrier)
// Standalone: hipcc --offload-arch=gfx950 -O3 q4_gemm.hip -o q4_gemm ; prints TF/s per shape and a correctness check.
//
// OUTCOME (round 2): not built further.  hipcc does not keep 256 accumulators resident in the AGPR half of the register
// file: with __builtin_amdgcn_mfma_* it emits ~450 v_accvgpr_read / _write moves and 140 spilled registers per stage; with
// the MFMA as inline asm on "+a" operands the loop body is clean but the accumulators are read back and spilled to scratch
// between iterations (`-Rpass-analysis=kernel-resource-usage`: VGPRs 256, AGPRs 256, scratch 560 B either way).  A
// 128 x 128-per-wave main loop needs hand-allocated registers (an assembly body), which is where hipBLASLt's long-K
// advantage comes from.
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

__device__ __forceinline__ int kc32_swz(int row) { return (-(row >> 2)) & 3; }

#define STAGE 32768

__global__ __launch_bounds__(256) void q4_kernel(const bf16_t* __restrict__ A, const bf16_t* __restrict__ B,
                                                  float* __restrict__ C, int M, int N, int K, int tiles_n) {
    extern __shared__ __attribute__((aligned(16))) char smem[];      // 3 x (A 16 KB | B 16 KB)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int tm = blockIdx.x / tiles_n, tn = blockIdx.x - tm * tiles_n;
    const int m0 = tm * 256, n0 = tn * 256;
    const int nk = K >> 5;

    // global -> register staging: thread t moves 4 x 16 B of A and of B per stage: chunk id c = t + 256 * i covers
    // row c >> 2 (64-byte row = 4 chunks), chunk c & 3
    uint4 sa[4], sb[4];
    auto gload = [&](int s) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = threadIdx.x + 256 * i, row = c >> 2, ch = c & 3;
            const int am = min(m0 + row, M - 1), bn = min(n0 + row, N - 1);
            sa[i] = *(const uint4*)(A + (size_t)am * K + s * 32 + ch * 8);
            sb[i] = *(const uint4*)(B + (size_t)bn * K + s * 32 + ch * 8);
        }
    };
    auto lstore = [&](int slot) {
        char* la = smem + slot * STAGE;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = threadIdx.x + 256 * i, row = c >> 2, ch = c & 3;
            const int off = row * 64 + ((ch ^ kc32_swz(row)) << 4);
            *(uint4*)(la + off) = sa[i];
            *(uint4*)(la + 16384 + off) = sb[i];
        }
    };
    // fragment of 16 rows x 32 k: lane l holds row (l & 15), k-chunk (l >> 4)
    const int frow = lane & 15, fch = lane >> 4;
    const int foff = frow * 64 + ((fch ^ kc32_swz(frow)) << 4);       // rbase is a multiple of 16: same swizzle for every fragment
    auto frag = [&](const char* tile, int rbase) -> bf16x8 { return *(const bf16x8*)(tile + rbase * 64 + foff); };

    f32x4 acc[8][8];
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[j][i] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // prologue: stages 0 and 1 into LDS, stage 2 in registers, fragments of stage 0
    gload(0); lstore(0);
    if (nk > 1) { gload(1); lstore(1); }
    if (nk > 2) gload(2);
    __syncthreads();
    bf16x8 fa[2][8], fb[2][8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        fa[0][i] = frag(smem, wr * 128 + i * 16);
        fb[0][i] = frag(smem + 16384, wc * 128 + i * 16);
    }

    for (int s = 0; s < nk; s += 2) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {                 // unrolled by two so the fragment sets alternate without copies
            const int st = s + u;
            if (st < nk) {
                const char* nxt = smem + ((st + 1) % 3) * STAGE;
                // stage st+2 (held in registers) -> LDS, then request stage st+3 into the same registers
                if (st + 2 < nk) lstore((st + 2) % 3);
                if (st + 3 < nk) gload(st + 3);
                // fragments of stage st+1 (its LDS slot was written during iteration st-1, barrier passed)
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    fa[u ^ 1][i] = frag(nxt, wr * 128 + i * 16);
                    fb[u ^ 1][i] = frag(nxt + 16384, wc * 128 + i * 16);
                }
#pragma unroll
                for (int j = 0; j < 8; ++j)
#pragma unroll
                    for (int i = 0; i < 8; ++i)
                        // accumulators pinned to AGPRs ("+a"): with the builtin hipcc shuffles the 256 accumulators between
                        // the VGPR and AGPR halves (450 v_accvgpr moves + scratch per stage)
                        asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[j][i]) : "v"(fb[u][j]), "v"(fa[u][i]));
                // interleave: 8 groups of {8 MFMA, 1 ds_write pair?, ...}: 64 MFMA, 8 ds_write, 8 vmem, 16 ds_read
#pragma unroll
                for (int g = 0; g < 8; ++g) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);      // 8 MFMA
                    __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);      // 1 DS write
                    __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);      // 1 VMEM read
                    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);      // 2 DS read
                }
                __syncthreads();
            }
        }
    }

    // plain epilogue: D' layout (roles swapped): lane owns row m = lane & 15, 4 consecutive columns n = (lane >> 4) * 4 ..
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int m = m0 + wr * 128 + i * 16 + (lane & 15);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int n = n0 + wc * 128 + j * 16 + (lane >> 4) * 4;
            if (m < M && n < N) *(f32x4*)(C + (size_t)m * N + n) = acc[j][i];
        }
    }
}

__global__ void ref_kernel(const bf16_t* A, const bf16_t* B, float* C, int M, int N, int K) {
    const int n = blockIdx.x * 16 + (threadIdx.x & 15), m = blockIdx.y * 16 + (threadIdx.x >> 4);
    if (m >= M || n >= N) return;
    float s = 0.f;
    for (int k = 0; k < K; ++k) s += (float)A[(size_t)m * K + k] * (float)B[(size_t)n * K + k];
    C[(size_t)m * N + n] = s;
}

int main() {
    struct Shape { int m, n, k; } shapes[] = {{512, 512, 256}, {1000, 776, 160}, {70400, 512, 2048}, {50432, 768, 3072}, {70400, 2048, 512}, {8192, 8192, 8192}};
    CHECK(hipFuncSetAttribute((const void*)q4_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 3 * STAGE));
    for (auto sh : shapes) {
        const size_t na = (size_t)sh.m * sh.k, nb = (size_t)sh.n * sh.k, nc = (size_t)sh.m * sh.n;
        std::vector<bf16_t> ha(na), hb(nb);
        unsigned seed = 12345;
        auto rnd = [&]() { seed = seed * 1664525u + 1013904223u; return ((seed >> 8) & 0xffff) / 65536.0f - 0.5f; };
        for (auto& v : ha) v = (bf16_t)rnd();
        for (auto& v : hb) v = (bf16_t)rnd();
        bf16_t *a, *b; float *c, *cr;
        CHECK(hipMalloc(&a, na * 2)); CHECK(hipMalloc(&b, nb * 2)); CHECK(hipMalloc(&c, nc * 4));
        CHECK(hipMemcpy(a, ha.data(), na * 2, hipMemcpyHostToDevice)); CHECK(hipMemcpy(b, hb.data(), nb * 2, hipMemcpyHostToDevice));
        const int tiles_m = (sh.m + 255) / 256, tiles_n = (sh.n + 255) / 256;
        auto run = [&]() { hipLaunchKernelGGL(q4_kernel, dim3(tiles_m * tiles_n), dim3(256), 3 * STAGE, 0, a, b, c, sh.m, sh.n, sh.k, tiles_n); };
        run(); CHECK(hipDeviceSynchronize());
        if (nc <= (size_t)1 << 21) {
            CHECK(hipMalloc(&cr, nc * 4));
            hipLaunchKernelGGL(ref_kernel, dim3((sh.n + 15) / 16, (sh.m + 15) / 16), dim3(256), 0, 0, a, b, cr, sh.m, sh.n, sh.k);
            std::vector<float> h1(nc), h2(nc);
            CHECK(hipMemcpy(h1.data(), c, nc * 4, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(h2.data(), cr, nc * 4, hipMemcpyDeviceToHost));
            double worst = 0, big = 0;
            for (size_t i = 0; i < nc; ++i) { worst = fmax(worst, fabs((double)h1[i] - h2[i])); big = fmax(big, fabs((double)h2[i])); }
            printf("check %dx%dx%d: max |diff| %.3e (max |ref| %.3e)\n", sh.m, sh.n, sh.k, worst, big);
            CHECK(hipFree(cr));
        }
        hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        for (int i = 0; i < 3; ++i) run();
        CHECK(hipEventRecord(e0));
        for (int i = 0; i < 10; ++i) run();
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); ms /= 10;
        printf("q4 %dx%dx%d: %.1f us  %.0f TF/s  (%d tiles = %.2f per CU)\n", sh.m, sh.n, sh.k, ms * 1e3, 2.0 * sh.m * sh.n * sh.k / ms / 1e9,
               tiles_m * tiles_n, tiles_m * tiles_n / 256.0);
        CHECK(hipFree(a)); CHECK(hipFree(b)); CHECK(hipFree(c));
    }
    return 0;
}
