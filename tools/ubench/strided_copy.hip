// Micro-benchmark: HBM throughput of a channel-chunked pass over a [pixels][M] fp32 tensor (what the depthwise kernels do)
// versus the same bytes laid out chunk-major [M/CC][pixels][CC].  Scratch tool.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

// each workgroup: one channel chunk (CC channels), a contiguous range of pixels; 4 loads in flight per thread
template <int CC>
__global__ __launch_bounds__(256) void k_copy(const float* __restrict__ src, float* __restrict__ dst, int P, int M,
                                              int px_per_wg, int blocked) {
    constexpr int CQ = CC / 4;
    const int chunks = M / CC;
    const int L = blockIdx.x;
    // same XCD-aware decode as the dw kernels: consecutive slots of one XCD = consecutive chunks of the same pixel range
    const int x = L & 7, q = L >> 3;
    const int cy = q % chunks, lane = (q / chunks) * 8 + x;
    const int p0 = lane * px_per_wg;
    const int tid = threadIdx.x, cq = tid % CQ, pl = tid / CQ;
    constexpr int PPI = 256 / CQ;   // pixels per iteration
    for (int pb = 0; pb < px_per_wg; pb += PPI * 4) {
        f32x4 v[4];
        size_t a[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int p = p0 + pb + u * PPI + pl;
            a[u] = blocked ? ((size_t)cy * P + p) * CC + 4 * cq : (size_t)p * M + cy * CC + 4 * cq;
            v[u] = (p < P) ? *(const f32x4*)(src + a[u]) : f32x4{0, 0, 0, 0};
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int p = p0 + pb + u * PPI + pl;
            if (p < P) *(f32x4*)(dst + a[u]) = v[u] * 1.5f;
        }
    }
}

template <int CC>
void run(const float* src, float* dst, int P, int M, int px_per_wg, int blocked) {
    const int lanes = (P + px_per_wg - 1) / px_per_wg;   // multiple of 8 by construction below
    const int grid = lanes * (M / CC);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k_copy<CC>, dim3(grid), dim3(256), 0, 0, src, dst, P, M, px_per_wg, blocked);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k_copy<CC>, dim3(grid), dim3(256), 0, 0, src, dst, P, M, px_per_wg, blocked);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    printf("P=%7d M=%5d CC=%2d px/wg=%5d %s  grid=%6d  %.3f ms  %.2f TB/s (R+W)\n", P, M, CC, px_per_wg,
           blocked ? "chunk-major" : "pixel-major", grid, ms, 2.0 * P * M * 4 / ms * 1e-9);
}

int main() {
    const size_t maxel = (size_t)401408 * 864;
    float *src, *dst;
    hipMalloc(&src, maxel * 4); hipMalloc(&dst, maxel * 4);
    hipMemset(src, 0, maxel * 4);
    for (int blocked = 0; blocked < 2; ++blocked) {
        run<16>(src, dst, 401408, 864, 3136, blocked);   // cell 1 soft: one 56x56 image per workgroup lane
        run<32>(src, dst, 401408, 864, 3136, blocked);
        run<16>(src, dst, 401408, 864, 784, blocked);
        run<32>(src, dst, 100352, 1440, 784, blocked);   // cell 3 soft
        run<32>(src, dst, 25088, 4032, 196, blocked);    // cell 11 soft
        run<32>(src, dst, 25088, 4032, 784, blocked);
        run<16>(src, dst, 401408, 144, 3136, blocked);   // cell 1 sampled
        run<32>(src, dst, 25088, 672, 196, blocked);     // cell 11 sampled
    }
    return 0;
}
