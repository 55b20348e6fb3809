#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float sum_bit5(float v) {
    unsigned a = __builtin_bit_cast(unsigned, v), b;
    asm volatile("v_mov_b32 %0, %1" : "=v"(b) : "v"(a));       // (a second REGISTER: the instruction swaps in place)
    asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    return __builtin_bit_cast(float, a) + __builtin_bit_cast(float, b);
}
__device__ __forceinline__ float sum_bit4(float v) {
    unsigned a = __builtin_bit_cast(unsigned, v), b;
    asm volatile("v_mov_b32 %0, %1" : "=v"(b) : "v"(a));
    asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    return __builtin_bit_cast(float, a) + __builtin_bit_cast(float, b);
}
__global__ void k(float* p) {
    float v = p[threadIdx.x];
    float t = sum_bit5(v);
    t = sum_bit4(t);
    t += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, t), 0x128, 0xf, 0xf, true));   // row_ror:8
    p[threadIdx.x] = t;
}
int main() {
    float h[64], *d;
    for (int i = 0; i < 64; ++i) h[i] = (float)(1 << (i >> 3)) + 0.001f * (i & 7);
    hipMalloc(&d, 256); hipMemcpy(d, h, 256, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    float o[64]; hipMemcpy(o, d, 256, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 64; ++i) { float want = 255.f + 8 * 0.001f * (i & 7); if (fabsf(o[i] - want) > 1e-3f) ++bad; }
    printf("bad %d  o[0]=%f o[9]=%f o[63]=%f\n", bad, o[0], o[9], o[63]);
    return 0;
}
