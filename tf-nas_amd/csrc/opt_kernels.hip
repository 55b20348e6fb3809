// Fused optimizer steps of the search loop: gradient-norm clipping + the parameter update in two
// launches for the weights (multi-range SGD) and ONE launch for the architecture parameters (Adam + log-softmax projection).
//
// Reference: train_search.py:381-385 (clip_grad_norm_(weight_parameters, 5) + SGD(momentum 0.9, wd 1e-5).step()) and
// :414-422 (clip_grad_norm_(arch_parameters, 5) + Adam(lr 0.01, betas (0.5, 0.999), wd 5e-4).step() + the per-parameter
// log_softmax projection).  torch.optim only touches parameters whose .grad is not None; with bi-sampling that is the two
// sampled candidates of every cell plus stems / head -- here: a list of ranges of the flat weight / gradient / momentum
// arenas (tfnas_amd/path.py: WeightArena), one range per sampled candidate.
#include "tfnas_dev.h"
#include "kernels.h"
#include "prof.h"

#define OPT_MAX_RANGES 64
#define OPT_CHUNK 8192                     // floats per workgroup: 256 threads x 8 float4

struct OptRanges {
    int n;
    int nblocks;
    uint64_t off[OPT_MAX_RANGES];          // float offsets into the arenas (multiples of 4)
    uint64_t goff[OPT_MAX_RANGES];         // float offsets of the same ranges in the gradient buffer (== off, or packed)
    uint64_t len[OPT_MAX_RANGES];          // floats (multiples of 4: arena slots are padded to 64)
    int first_block[OPT_MAX_RANGES + 1];   // prefix sum of ceil(len / OPT_CHUNK)
};

// workgroup b -> [beg, end) in the arenas and gbeg = start of the same chunk in the gradient buffer
__device__ __forceinline__ bool opt_locate(const OptRanges& R, int b, uint64_t& beg, uint64_t& end, uint64_t& gbeg) {
    int r = 0;
    while (r + 1 < R.n && b >= R.first_block[r + 1]) ++r;
    const uint64_t lo = (uint64_t)(b - R.first_block[r]) * OPT_CHUNK;
    if (lo >= R.len[r]) return false;
    beg = R.off[r] + lo;
    gbeg = R.goff[r] + lo;
    const uint64_t hi = lo + OPT_CHUNK < R.len[r] ? lo + OPT_CHUNK : R.len[r];
    end = R.off[r] + hi;
    return true;
}

__device__ __forceinline__ double block_sum_d(double v, double* sh) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) sh[wv] = v;
    __syncthreads();
    return (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

// partial[b] = sum of g^2 over workgroup b's chunk (double; fixed order -> deterministic)
__global__ __launch_bounds__(256) void k_opt_sqnorm(const float* __restrict__ g, OptRanges R, double* __restrict__ partial) {
    __shared__ double sh[4];
    uint64_t beg, end, gbeg;
    double acc = 0.0;
    if (opt_locate(R, blockIdx.x, beg, end, gbeg)) {
        for (uint64_t i = 4 * (uint64_t)threadIdx.x; i < end - beg; i += 1024) {
            const f32x4 v = ld4(g + gbeg + i);
            acc += (double)v.x * v.x + (double)v.y * v.y + (double)v.z * v.z + (double)v.w * v.w;
        }
    }
    acc = block_sum_d(acc, sh);
    if (threadIdx.x == 0) partial[blockIdx.x] = acc;
}

// total = sqrt(sum partial) * grad_scale;  coef = min(1, max_norm / (total + 1e-6))   (torch.nn.utils.clip_grad_norm_)
// g' = g * grad_scale * coef;  d = g' + wd * w;  m = momentum * m + d;  w -= lr * m    (torch.optim.SGD, dampening 0)
__global__ __launch_bounds__(256) void k_opt_sgd(float* __restrict__ w, float* __restrict__ g, float* __restrict__ m,
                                                 OptRanges R, const double* __restrict__ partial, float max_norm, float lr,
                                                 float momentum, float wd, float grad_scale, float* __restrict__ norm_out) {
    __shared__ double sh[4];
    double acc = 0.0;
    for (int i = threadIdx.x; i < R.nblocks; i += 256) acc += partial[i];
    acc = block_sum_d(acc, sh);
    const float total = (float)sqrt(acc) * grad_scale;
    float coef = max_norm > 0.f ? max_norm / (total + 1e-6f) : 1.f;
    coef = coef > 1.f ? 1.f : coef;
    if (blockIdx.x == 0 && threadIdx.x == 0 && norm_out) *norm_out = total;
    const float gs = grad_scale * coef;
    uint64_t beg, end, gbeg;
    if (!opt_locate(R, blockIdx.x, beg, end, gbeg)) return;
    for (uint64_t k = 4 * (uint64_t)threadIdx.x; k < end - beg; k += 1024) {
        const uint64_t i = beg + k;
        f32x4 gv = ld4(g + gbeg + k), wv = ld4(w + i), mv = ld4(m + i);
        gv = gv * splat4(gs);
        st4(g + gbeg + k, gv);                           // (the clipped gradient stays readable, as after clip_grad_norm_)
        const f32x4 d = gv + splat4(wd) * wv;
        mv = splat4(momentum) * mv + d;
        wv = wv - splat4(lr) * mv;
        st4(m + i, mv);
        st4(w + i, wv);
    }
}

// dst[packed] = src[ranges]: the sampled candidates' gradient ranges as ONE contiguous message for the all-reduce
__global__ __launch_bounds__(256) void k_opt_pack(const float* __restrict__ src, float* __restrict__ dst, OptRanges R) {
    uint64_t beg, end, gbeg;
    if (!opt_locate(R, blockIdx.x, beg, end, gbeg)) return;
    for (uint64_t k = 4 * (uint64_t)threadIdx.x; k < end - beg; k += 1024) st4(dst + gbeg + k, ld4(src + beg + k));
}

static int fill_ranges(OptRanges& R, int nranges, const uint64_t* off, const uint64_t* goff, const uint64_t* len) {
    if (nranges < 1 || nranges > OPT_MAX_RANGES) return TFNAS_ERANGE;
    R.n = nranges;
    int nb = 0;
    for (int i = 0; i < nranges; ++i) {
        if ((off[i] & 3) || (len[i] & 3) || len[i] == 0 || (goff && (goff[i] & 3))) return TFNAS_EINVAL;
        R.off[i] = off[i];
        R.goff[i] = goff ? goff[i] : off[i];
        R.len[i] = len[i];
        R.first_block[i] = nb;
        nb += (int)((len[i] + OPT_CHUNK - 1) / OPT_CHUNK);
    }
    R.first_block[nranges] = nb;
    R.nblocks = nb;
    return 0;
}

int launch_pack_ranges(const float* src, float* dst, int nranges, const uint64_t* off, const uint64_t* doff,
                       const uint64_t* len, hipStream_t s) {
    OptRanges R;
    int rc = fill_ranges(R, nranges, off, doff, len);
    if (rc) return rc;
    ProfScope _prof(TK_SMALL, s);
    hipLaunchKernelGGL(k_opt_pack, dim3(R.nblocks), dim3(256), 0, s, src, dst, R);
    return (int)hipGetLastError();
}

int launch_sgd_clip_step(float* w, float* g, float* m, int nranges, const uint64_t* off, const uint64_t* goff,
                         const uint64_t* len, float max_norm,
                         float lr, float momentum, float wd, float grad_scale, double* scratch, uint64_t scratch_doubles,
                         float* norm_out, hipStream_t s) {
    OptRanges R;
    int rc = fill_ranges(R, nranges, off, goff, len);
    if (rc) return rc;
    const int nb = R.nblocks;
    if ((uint64_t)nb > scratch_doubles) return TFNAS_ERANGE;
    ProfScope _prof(TK_SMALL, s);
    hipLaunchKernelGGL(k_opt_sqnorm, dim3(nb), dim3(256), 0, s, g, R, scratch);
    hipLaunchKernelGGL(k_opt_sgd, dim3(nb), dim3(256), 0, s, w, g, m, R, scratch, max_norm, lr, momentum, wd, grad_scale,
                       norm_out);
    return (int)hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------
// Architecture parameters: <= 32 tensors of <= 8 floats.  One workgroup: clip over all of them, Adam, projection.
struct ArchOpt {
    int n;
    int len[TFNAS_MAX_CELLS];
    float* p[TFNAS_MAX_CELLS];
    const float* g[TFNAS_MAX_CELLS];
};

__global__ __launch_bounds__(256) void k_arch_adam_project(ArchOpt A, float* __restrict__ m, float* __restrict__ v,
                                                           float max_norm, float lr, float b1, float b2, float eps, float wd,
                                                           float bias1, float bias2_sqrt, float grad_scale,
                                                           float* __restrict__ norm_out) {
    __shared__ float sq[256];
    __shared__ float newp[256];
    const int t = threadIdx.x >> 3, j = threadIdx.x & 7;       // tensor, element (256 threads = 32 tensors x 8)
    const bool live = t < A.n && j < A.len[t];
    float gv = live ? A.g[t][j] * grad_scale : 0.f;
    sq[threadIdx.x] = gv * gv;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) sq[threadIdx.x] += sq[threadIdx.x + o];
        __syncthreads();
    }
    const float total = sqrtf(sq[0]);
    float coef = max_norm > 0.f ? max_norm / (total + 1e-6f) : 1.f;
    coef = coef > 1.f ? 1.f : coef;
    if (threadIdx.x == 0 && norm_out) *norm_out = total;
    float pv = 0.f;
    if (live) {
        pv = A.p[t][j];
        gv = gv * coef + wd * pv;                               // torch Adam: weight decay is added to the gradient
        const int k = t * 8 + j;
        const float mk = b1 * m[k] + (1.f - b1) * gv;
        const float vk = b2 * v[k] + (1.f - b2) * gv * gv;
        m[k] = mk;
        v[k] = vk;
        const float denom = sqrtf(vk) / bias2_sqrt + eps;
        pv = pv - (lr / bias1) * (mk / denom);
    }
    newp[threadIdx.x] = live ? pv : -INFINITY;
    __syncthreads();
    if (live) {                                                 // p <- log_softmax(p) over the tensor (train_search.py:421-422)
        float mx = -INFINITY;
        for (int i = 0; i < A.len[t]; ++i) mx = fmaxf(mx, newp[t * 8 + i]);
        float se = 0.f;
        for (int i = 0; i < A.len[t]; ++i) se += expf(newp[t * 8 + i] - mx);
        A.p[t][j] = (pv - mx) - logf(se);
    }
}

int launch_arch_adam_project(int n, float* const* p, const float* const* g, const int32_t* len, float* m, float* v,
                             float max_norm, float lr, float b1, float b2, float eps, float wd, int step, float grad_scale,
                             float* norm_out, hipStream_t s) {
    if (n < 1 || n > TFNAS_MAX_CELLS) return TFNAS_ERANGE;
    ArchOpt A;
    A.n = n;
    for (int i = 0; i < n; ++i) {
        if (!p[i] || !g[i]) return TFNAS_ENULL;
        if (len[i] < 1 || len[i] > 8) return TFNAS_ERANGE;
        A.len[i] = len[i];
        A.p[i] = p[i];
        A.g[i] = g[i];
    }
    const double bias1 = 1.0 - pow((double)b1, (double)step), bias2 = 1.0 - pow((double)b2, (double)step);
    ProfScope _prof(TK_SMALL, s);
    hipLaunchKernelGGL(k_arch_adam_project, dim3(1), dim3(256), 0, s, A, m, v, max_norm, lr, b1, b2, eps, wd, (float)bias1,
                       (float)sqrt(bias2), grad_scale, norm_out);
    return (int)hipGetLastError();
}
