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
__device__ __forceinline__ float sigmoid_f(float x) { return __builtin_amdgcn_rcpf(1.f + __expf(-x)); }
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
// ----------------------------------------------------------------------------- bf16 storage of the stream tensors
// TfnasCellDesc.stor = 1: E, D, dZ, dEh hold bf16 (round-to-nearest-even on store, exact widening on load); `idx` below is an
// ELEMENT index, `base` the tensor's base pointer (typed float* throughout the library).  stor is wave-uniform: the branch is
// a scalar one and the fp32 path executes exactly the instructions it did before.
typedef unsigned tf_u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x4 bf16x4_widen(uint2 u) {
    f32x4 r;
    r.x = __uint_as_float(u.x << 16);
    r.y = __uint_as_float(u.x & 0xffff0000u);
    r.z = __uint_as_float(u.y << 16);
    r.w = __uint_as_float(u.y & 0xffff0000u);
    return r;
}
__device__ __forceinline__ unsigned bf16_rne(float f) {
    const unsigned u = __float_as_uint(f);
    return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;          // (NaN payloads are not preserved; none occur here)
}
// gfx950 has the packed conversion in hardware (v_cvt_pk_bf16_f32, round-to-nearest-even): 2 instructions per quad where the
// integer sequence above took ~20 -- the stores of the bf16 mode sit in VALU-bound kernels
typedef float tf_f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 tf_bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint2 bf16x4_narrow(f32x4 v) {
#ifdef TFNAS_BF16_SW
    uint2 r0;
    r0.x = bf16_rne(v.x) | (bf16_rne(v.y) << 16);
    r0.y = bf16_rne(v.z) | (bf16_rne(v.w) << 16);
    return r0;
#endif
    const tf_f32x2 lo = {v.x, v.y}, hi = {v.z, v.w};
    const tf_bf16x2 a = __builtin_convertvector(lo, tf_bf16x2), b = __builtin_convertvector(hi, tf_bf16x2);
    uint2 r;
    r.x = *reinterpret_cast<const unsigned*>(&a);
    r.y = *reinterpret_cast<const unsigned*>(&b);
    return r;
}
#if defined(TFNAS_NO_BF16)           /* the product library: bf16 branches compiled out */
#define TFNAS_STOR(s) 0
#elif defined(TFNAS_ONLY_BF16)       /* the bf16 library: fp32-storage branches compiled out (its plan accepts stor = 1 only).
                                        As wave-uniform RUNTIME branches around every load / store of the stream tensors they
                                        broke the kernels' load pipelines into dozens of basic blocks: 1.2-1.75x slower per
                                        kernel than the fp32 build although the bytes halve */
#define TFNAS_STOR(s) 1
#else
#define TFNAS_STOR(s) (s)
#endif
__device__ __forceinline__ f32x4 ldS4(const float* base, size_t idx, int stor) {
    if (TFNAS_STOR(stor)) return bf16x4_widen(*reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned short*>(base) + idx));
    return ld4(base + idx);
}
// two-phase form for loaders that must not touch the loaded registers before the MFMAs (gemm_core.h): _raw only issues the
// load (bf16: the 8 raw bytes travel in .x/.y), _fin widens
__device__ __forceinline__ f32x4 ldS4_raw(const float* base, size_t idx, int stor) {
    if (TFNAS_STOR(stor)) {
        const uint2 u = *reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned short*>(base) + idx);
        f32x4 r = {__uint_as_float(u.x), __uint_as_float(u.y), 0.f, 0.f};
        return r;
    }
    return ld4(base + idx);
}
__device__ __forceinline__ f32x4 ldS4_fin(f32x4 raw, int stor) {
    if (TFNAS_STOR(stor)) return bf16x4_widen(make_uint2(__float_as_uint(raw.x), __float_as_uint(raw.y)));
    return raw;
}
__device__ __forceinline__ f32x4 ldS4_nt(const float* base, size_t idx, int stor) {
    if (TFNAS_STOR(stor)) {
        const tf_u32x2 u = __builtin_nontemporal_load(
            reinterpret_cast<const tf_u32x2*>(reinterpret_cast<const unsigned short*>(base) + idx));
        return bf16x4_widen(make_uint2(u.x, u.y));
    }
    return ld4_nt(base + idx);
}
__device__ __forceinline__ void stS4(float* base, size_t idx, f32x4 v, int stor) {
    if (TFNAS_STOR(stor)) *reinterpret_cast<uint2*>(reinterpret_cast<unsigned short*>(base) + idx) = bf16x4_narrow(v);
    else st4(base + idx, v);
}
__device__ __forceinline__ void stS4_nt(float* base, size_t idx, f32x4 v, int stor) {
    if (TFNAS_STOR(stor)) {
        const uint2 n = bf16x4_narrow(v);
        tf_u32x2 u = {n.x, n.y};
        __builtin_nontemporal_store(u, reinterpret_cast<tf_u32x2*>(reinterpret_cast<unsigned short*>(base) + idx));
    } else {
        st4_nt(base + idx, v);
    }
}

// agent-scope (write-through) store of a value another workgroup of the same launch reads back (gemm_core.h: tail_reduce_cols)
__device__ __forceinline__ void st_coherent(float* p, float v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

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
