// Micro-benchmark of the fp32 MFMA GEMM main-loop structure (scratch tool; not part of the product path).
// Build: hipcc --offload-arch=gfx950 -O3 -o mfma_loop mfma_loop.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NT, int WM, int MODE, int PAD>
__global__ __launch_bounds__(256) void k(const float* __restrict__ src, float* __restrict__ dst, int iters) {
    // tile: (4 waves x WM*16 rows) x (NT*16 cols), BK = 16, K-major LDS tiles, double buffered
    constexpr int BM = 4 * WM * 16, BN = NT * 16, LDA = BM + 16, LDB = (BN % 32 == 16) ? BN : BN + 16;
    __shared__ __attribute__((aligned(16))) float lds[2 * 16 * (LDA + LDB) + PAD];
    const int tid = threadIdx.x, lane = tid & 63, wrow = (tid >> 6) * WM * 16, lr = lane & 15, lk = lane >> 4;
    for (int i = tid; i < 2 * 16 * (LDA + LDB); i += 256) lds[i] = (float)(i & 7);
    __syncthreads();
    f32x4 acc[WM][NT];
    for (int i = 0; i < WM; ++i) for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0, 0, 0, 0};
    float af[4][WM], bf[4][NT];
    for (int ks = 0; ks < 4; ++ks) { for (int i = 0; i < WM; ++i) af[ks][i] = lane + i; for (int j = 0; j < NT; ++j) bf[ks][j] = lane - j; }
    f32x4 ra[(BM * 4 + 255) / 256], rb[(BN * 4 + 255) / 256];
    for (auto& r : ra) r = f32x4{1, 2, 3, 4};
    for (auto& r : rb) r = f32x4{1, 2, 3, 4};
    for (int c = 0; c < iters; ++c) {
        const float* As = lds + (c & 1) * 16 * LDA;
        const float* Bs = lds + 2 * 16 * LDA + (c & 1) * 16 * LDB;
        if (MODE >= 4) {
#pragma unroll
            for (int i = 0; i < (BM * 4 + 255) / 256; ++i) { int idx = tid + 256 * i; ra[i] = *(const f32x4*)(src + ((size_t)((blockIdx.x * 131 + c) & 1023) * BM * 16 + idx * 4)); }
#pragma unroll
            for (int i = 0; i < (BN * 4 + 255) / 256; ++i) { int idx = tid + 256 * i; if (idx < BN * 4) rb[i] = *(const f32x4*)(src + ((size_t)((blockIdx.x * 17 + c) & 1023) * BN * 16 + idx * 4)); }
        }
        if (MODE >= 1) {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const int kk = ks * 4 + lk;
#pragma unroll
                for (int i = 0; i < WM; ++i) af[ks][i] = As[kk * LDA + wrow + 16 * i + lr];
#pragma unroll
                for (int j = 0; j < NT; ++j) bf[ks][j] = Bs[kk * LDB + 16 * j + lr];
            }
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int i = 0; i < WM; ++i)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[ks][i], bf[ks][j], acc[i][j], 0, 0, 0);
        if (MODE >= 3) {
            float* Aw = lds + ((c + 1) & 1) * 16 * LDA;
            float* Bw = lds + 2 * 16 * LDA + ((c + 1) & 1) * 16 * LDB;
#pragma unroll
            for (int i = 0; i < (BM * 4 + 255) / 256; ++i) {
                int idx = tid + 256 * i; int row = idx >> 2, kl = (idx & 3) * 4;
                if (idx < BM * 4) { Aw[(kl + 0) * LDA + row] = ra[i].x; Aw[(kl + 1) * LDA + row] = ra[i].y; Aw[(kl + 2) * LDA + row] = ra[i].z; Aw[(kl + 3) * LDA + row] = ra[i].w; }
            }
#pragma unroll
            for (int i = 0; i < (BN * 4 + 255) / 256; ++i) {
                int idx = tid + 256 * i; int n = idx >> 2, kl = (idx & 3) * 4;
                if (idx < BN * 4) { Bw[(kl + 0) * LDB + n] = rb[i].x; Bw[(kl + 1) * LDB + n] = rb[i].y; Bw[(kl + 2) * LDB + n] = rb[i].z; Bw[(kl + 3) * LDB + n] = rb[i].w; }
            }
        }
        if (MODE >= 2) __syncthreads();
    }
    float s = 0;
    for (int i = 0; i < WM; ++i) for (int j = 0; j < NT; ++j) s += acc[i][j].x + acc[i][j].y + acc[i][j].z + acc[i][j].w;
    if (s == 1.2345f) dst[tid] = s;
}

template <int NT, int WM, int MODE, int PAD = 0>
void run(const char* name, const float* src, float* dst, int wgs_per_cu) {
    const int iters = 2000, grid = 256 * wgs_per_cu;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<NT, WM, MODE, PAD>), dim3(grid), dim3(256), 0, 0, src, dst, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<NT, WM, MODE, PAD>), dim3(grid), dim3(256), 0, 0, src, dst, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)grid * 4 /*waves*/ * iters * 4 * NT * WM * 2048.0;
    int occ = 0; hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k<NT, WM, MODE, PAD>, 256, 0);
    printf("%-34s NT=%d WM=%d mode=%d wg/cu=%d (occ %d)  %.3f ms  %.1f TF/s\n", name, NT, WM, MODE, wgs_per_cu, occ, ms, flops / ms * 1e-9);
}

int main() {
    float *src, *dst;
    hipMalloc(&src, 1024 * 128 * 16 * 4 * 2); hipMalloc(&dst, 1 << 20);
    hipMemset(src, 0, 1024 * 128 * 16 * 4 * 2);
#define ALLMODES(NT, WM, W) \
    run<NT, WM, 0>("mfma only", src, dst, W); run<NT, WM, 1>("+lds frag reads", src, dst, W); \
    run<NT, WM, 2>("+barrier", src, dst, W); run<NT, WM, 3>("+lds stores", src, dst, W); run<NT, WM, 4>("+global loads", src, dst, W);
    for (int w = 1; w <= 4; ++w) { ALLMODES(4, 2, w) }
    for (int w = 1; w <= 3; ++w) { ALLMODES(7, 2, w) }
    for (int w = 1; w <= 3; ++w) { ALLMODES(4, 4, w) }
    for (int w = 1; w <= 2; ++w) { ALLMODES(8, 4, w) }
    return 0;
}
