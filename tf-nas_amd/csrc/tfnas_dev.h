// Device-side helpers shared by all gfx950 kernels of the TF-NAS hot path.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "tfnas_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define TFNAS_THREADS 256

#define HIP_TRY(expr)                         \
    do {                                      \
        hipError_t _e = (expr);               \
        if (_e != hipSuccess) return (int)_e; \
    } while (0)

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
static inline uint64_t cdiv64(uint64_t a, uint64_t b) { return (a + b - 1) / b; }

// ----------------------------------------------------------------------------- activations
// reference: Swish = x * sigmoid(x) (models/layers.py:26-35), ReLU (layers.py:470-471)
// 1/(1+e^-x) through v_exp_f32 + v_rcp_f32 (1 ulp each).  An IEEE `/` costs ~10 more VALU instructions per element
// (v_div_scale x2, fma chain, v_div_fmas, v_div_fixup), and the operand loaders that apply the activation are VALU-issue
// bound: 682 VALU instructions per 32 MFMAs in k_project_fwd<4, swish> before this.
#ifdef TFNAS_FAKE_SIGMOID      // timing-only build (tools/r5_sigmoid.sh): what the two transcendentals of every swish / swish' cost
__device__ __forceinline__ float sigmoid_f(float x) { return fmaf(x, 0.25f, 0.5f); }
#else
__device__ __forceinline__ float sigmoid_f(float x) { return __builtin_amdgcn_rcpf(1.f + __expf(-x)); }
#endif
template <int ACT>
__device__ __forceinline__ float act_f(float x) {
    if (ACT == TFNAS_ACT_RELU) return fmaxf(x, 0.f);
    return x * sigmoid_f(x);
}
// derivative w.r.t. the pre-activation x
template <int ACT>
__device__ __forceinline__ float act_d(float x) {
    if (ACT == TFNAS_ACT_RELU) return x > 0.f ? 1.f : 0.f;
    const float s = sigmoid_f(x);
    return s * (1.f + x * (1.f - s));
}
template <int ACT>
__device__ __forceinline__ f32x4 act_f4(f32x4 v) {
    f32x4 r;
    r.x = act_f<ACT>(v.x); r.y = act_f<ACT>(v.y); r.z = act_f<ACT>(v.z); r.w = act_f<ACT>(v.w);
    return r;
}
template <int ACT>
__device__ __forceinline__ f32x4 act_d4(f32x4 v) {
    f32x4 r;
    r.x = act_d<ACT>(v.x); r.y = act_d<ACT>(v.y); r.z = act_d<ACT>(v.z); r.w = act_d<ACT>(v.w);
    return r;
}

// ----------------------------------------------------------------------------- batch-norm constants
// stats layout: [channel][2] doubles = (sum, sum of squares) over `cnt` elements.
// BN of the search net: (x-mean)/sqrt(var_biased+eps), no affine (layers.py:469,498,533).
// eps < 0: the table already holds the EFFECTIVE (mean, rstd) of an affine / eval-mode BatchNorm, written by k_bn_fwd_fix
// (bn_affine.hip): gamma*(x-mu)*rho + beta == (x - mean_eff) * rstd_eff with rstd_eff = gamma*rho, mean_eff = mu - beta/rstd_eff.
__device__ __forceinline__ float2 bn_consts(const double* st, double inv_cnt, float eps) {
    if (eps < 0.f) return make_float2((float)st[0], (float)st[1]);
    double m = st[0] * inv_cnt;
    double v = st[1] * inv_cnt - m * m;
    if (v < 0.0) v = 0.0;
    return make_float2((float)m, (float)(1.0 / sqrt(v + (double)eps)));
}

__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ void st4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
// non-temporal variants for the once-through [pixels][M] streams (see dw_stream.inc)
#ifdef TFNAS_NO_NT
__device__ __forceinline__ f32x4 ld4_nt(const float* p) { return ld4(p); }
__device__ __forceinline__ void st4_nt(float* p, f32x4 v) { st4(p, v); }
#else
__device__ __forceinline__ f32x4 ld4_nt(const float* p) {
    return __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p));
}
__device__ __forceinline__ void st4_nt(float* p, f32x4 v) { __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(p)); }
#endif
// ----------------------------------------------------------------------------- stream-tensor accessors
// E, D, dZ, dEh are fp32.  (Rounds 1-3 carried a second build with bf16 storage of these four tensors, TfnasCellDesc.stor = 1:
// it ended at 0.99-1.04x of the fp32 iteration pair -- the step is bound by launch structure and by LDS / VALU work, not by HBM
// bytes -- and was removed; `stor` must be 0, the parameter below remains in the signatures of the loaders.)  `idx` is an
// ELEMENT index, `base` the tensor's base pointer.
#define TFNAS_STOR(s) 0
#ifdef TFNAS_HALF_BYTES
// TIMING-ONLY build (tools/r5_halfbytes.sh): the four stream tensors are stored as the upper halves of
// their fp32 values (truncation, 8 bytes per quad at byte offset 2 * idx of the SAME buffers) -- every kernel issues the same
// number of memory instructions for half the bytes.  Wrong numerics by construction; never shipped, never tested for parity.
typedef unsigned hb_u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x4 hb_expand(hb_u32x2 r) {
    f32x4 v;
    v.x = __builtin_bit_cast(float, r.x << 16); v.y = __builtin_bit_cast(float, r.x & 0xffff0000u);
    v.z = __builtin_bit_cast(float, r.y << 16); v.w = __builtin_bit_cast(float, r.y & 0xffff0000u);
    return v;
}
__device__ __forceinline__ hb_u32x2 hb_pack(f32x4 v) {
    const unsigned a = __builtin_bit_cast(unsigned, v.x), b = __builtin_bit_cast(unsigned, v.y);
    const unsigned c = __builtin_bit_cast(unsigned, v.z), e = __builtin_bit_cast(unsigned, v.w);
    return hb_u32x2{(a >> 16) | (b & 0xffff0000u), (c >> 16) | (e & 0xffff0000u)};
}
__device__ __forceinline__ const hb_u32x2* hb_ptr(const float* base, size_t idx) {
    return reinterpret_cast<const hb_u32x2*>(reinterpret_cast<const char*>(base) + 2 * idx);
}
__device__ __forceinline__ f32x4 ldS4(const float* base, size_t idx, int) { return hb_expand(*hb_ptr(base, idx)); }
__device__ __forceinline__ f32x4 ldS4_raw(const float* base, size_t idx, int) {
    const hb_u32x2 r = *hb_ptr(base, idx);
    f32x4 v = {__builtin_bit_cast(float, r.x), __builtin_bit_cast(float, r.y), 0.f, 0.f};
    return v;
}
__device__ __forceinline__ f32x4 ldS4_fin(f32x4 raw, int) {
    return hb_expand(hb_u32x2{__builtin_bit_cast(unsigned, raw.x), __builtin_bit_cast(unsigned, raw.y)});
}
__device__ __forceinline__ f32x4 ldS4_nt(const float* base, size_t idx, int) {
    return hb_expand(__builtin_nontemporal_load(hb_ptr(base, idx)));
}
__device__ __forceinline__ void stS4(float* base, size_t idx, f32x4 v, int) { *const_cast<hb_u32x2*>(hb_ptr(base, idx)) = hb_pack(v); }
__device__ __forceinline__ void stS4_nt(float* base, size_t idx, f32x4 v, int) {
    __builtin_nontemporal_store(hb_pack(v), const_cast<hb_u32x2*>(hb_ptr(base, idx)));
}
#else
__device__ __forceinline__ f32x4 ldS4(const float* base, size_t idx, int) { return ld4(base + idx); }
// two-phase form for loaders that must not touch the loaded registers before the MFMAs (gemm_core.h)
__device__ __forceinline__ f32x4 ldS4_raw(const float* base, size_t idx, int) { return ld4(base + idx); }
__device__ __forceinline__ f32x4 ldS4_fin(f32x4 raw, int) { return raw; }
__device__ __forceinline__ f32x4 ldS4_nt(const float* base, size_t idx, int) { return ld4_nt(base + idx); }
__device__ __forceinline__ void stS4(float* base, size_t idx, f32x4 v, int) { st4(base + idx, v); }
__device__ __forceinline__ void stS4_nt(float* base, size_t idx, f32x4 v, int) { st4_nt(base + idx, v); }
#endif

__device__ __forceinline__ f32x4 zero4() { f32x4 z = {0.f, 0.f, 0.f, 0.f}; return z; }
__device__ __forceinline__ f32x4 splat4(float a) { f32x4 z = {a, a, a, a}; return z; }

// load 4 consecutive floats that may be unaligned / partially out of range [0, lim)
__device__ __forceinline__ f32x4 ld4_guard(const float* base, int idx, int lim, bool aligned) {
    f32x4 r = zero4();
    if (aligned && idx + 3 < lim) return ld4(base + idx);
    if (idx < lim) r.x = base[idx];
    if (idx + 1 < lim) r.y = base[idx + 1];
    if (idx + 2 < lim) r.z = base[idx + 2];
    if (idx + 3 < lim) r.w = base[idx + 3];
    return r;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// atomic accumulate into double / float device memory (ordinary coarse-grained allocations)
__device__ __forceinline__ void atomic_add_f64(double* p, double v) { unsafeAtomicAdd(p, v); }
__device__ __forceinline__ void atomic_add_f32(float* p, float v) { unsafeAtomicAdd(p, v); }
