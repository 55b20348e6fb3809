// Architecture-parameter kernels: Gumbel-softmax over the 8 candidates of every cell (forward, backward,
// bi-sampling index selection) and the sink-connecting softmax(beta) mix of a stage.
//
// Reference: MixedOP.forward (models/model_search.py:58-91), F.gumbel_softmax as called at :62,:66,:87,
//            MixedStage.forward tail (:202-204).
#include "tfnas_dev.h"
#include "kernels.h"
#include "prof.h"

struct PtrPack {
    const float* p[TFNAS_MAX_CELLS];
};
struct MutPtrPack {
    float* p[TFNAS_MAX_CELLS];
};

// ---------------------------------------------------------------------------- gumbel softmax, all cells
__global__ void k_arch_fwd(int ncell, PtrPack la, const float* __restrict__ e, const float* __restrict__ lat,
                           float T, float* __restrict__ w, float* __restrict__ cell_lat) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= ncell) return;
    float y[8], m = -INFINITY;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const float gumbel = -logf(e[c * 8 + i]);            // gumbels = -empty.exponential_().log()
        y[i] = (la.p[c][i] + gumbel) / T;
        m = fmaxf(m, y[i]);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        y[i] = expf(y[i] - m);
        s += y[i];
    }
    float cl = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const float wi = y[i] / s;
        w[c * 8 + i] = wi;
        if (lat) cl += wi * lat[c * 8 + i];                  // sum(w*lat ...) left to right (model_search.py:90)
    }
    if (cell_lat) cell_lat[c] = cl;
}

__global__ void k_arch_bwd(int ncell, const float* __restrict__ w, const float* __restrict__ lat,
                           const float* __restrict__ dw, const float* __restrict__ dcl, float T, MutPtrPack dla) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= ncell) return;
    const float dl = dcl ? dcl[c] : 0.f;
    float gt[8], dot = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        gt[i] = (dw ? dw[c * 8 + i] : 0.f) + ((lat && dcl) ? dl * lat[c * 8 + i] : 0.f);
        dot += gt[i] * w[c * 8 + i];
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) dla.p[c][j] = w[c * 8 + j] * (gt[j] - dot) / T;
}

// mode 0: gumbel-softmax argmax over the switched-on candidates; 1: argmin(log_alpha); 2: argmax(log_alpha)
__global__ void k_arch_sample(int ncell, PtrPack la, const uint8_t* __restrict__ mask, const float* __restrict__ e,
                              float T, int mode, int32_t* __restrict__ pos) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= ncell) return;
    float v[8];
    int n = 0;
    for (int i = 0; i < 8; ++i)
        if (mask[c * 8 + i]) v[n++] = la.p[c][i];
    int best = 0;
    if (mode == 0) {
        float m = -INFINITY;
        for (int i = 0; i < n; ++i) m = fmaxf(m, v[i]);
        float s = 0.f;
        for (int i = 0; i < n; ++i) s += expf(v[i] - m);
        const float lse = m + logf(s);
        float y[8], ym = -INFINITY;
        for (int i = 0; i < n; ++i) {                        // log_softmax, then gumbel softmax at temperature T
            y[i] = ((v[i] - lse) - logf(e[c * 8 + i])) / T;
            ym = fmaxf(ym, y[i]);
        }
        float ys = 0.f;
        for (int i = 0; i < n; ++i) {
            y[i] = expf(y[i] - ym);
            ys += y[i];
        }
        float bv = -INFINITY;
        for (int i = 0; i < n; ++i) {
            const float wi = y[i] / ys;
            if (wi > bv) {
                bv = wi;
                best = i;
            }
        }
    } else {
        float bv = v[0];
        for (int i = 1; i < n; ++i)
            if (mode == 1 ? v[i] < bv : v[i] > bv) {
                bv = v[i];
                best = i;
            }
    }
    pos[c] = best;
}

// ---------------------------------------------------------------------------- sink-connecting stage mix
__device__ __forceinline__ void softmax_k(const float* b, int K, float* bw) {
    float m = -INFINITY, s = 0.f;
    for (int k = 0; k < K; ++k) m = fmaxf(m, b[k]);
    for (int k = 0; k < K; ++k) {
        bw[k] = expf(b[k] - m);
        s += bw[k];
    }
    for (int k = 0; k < K; ++k) bw[k] /= s;
}

struct SinkPtrs {
    const float* res[TFNAS_MAX_SINK];
    float* dres[TFNAS_MAX_SINK];
};

__global__ __launch_bounds__(256) void k_sink_fwd(int K, const float* __restrict__ betas, SinkPtrs ptrs,
                                                  const float* __restrict__ cell_lat, uint64_t count4,
                                                  float* __restrict__ out, float* __restrict__ out_lat,
                                                  float* __restrict__ bw_out) {
    float bw[TFNAS_MAX_SINK];
    softmax_k(betas, K, bw);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        float cum = 0.f, ol = 0.f;
        for (int k = 0; k < K; ++k) {
            bw_out[k] = bw[k];
            if (cell_lat) {
                cum = (k == 0) ? cell_lat[0] : cum + cell_lat[k];   // lat1, lat1+lat2, ...
                ol += bw[k] * cum;
            }
        }
        if (out_lat) *out_lat = ol;
    }
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < count4; i += (uint64_t)gridDim.x * 256) {
        f32x4 v = zero4();
        for (int k = 0; k < K; ++k) v += splat4(bw[k]) * ld4(ptrs.res[k] + 4 * i);
        st4(out + 4 * i, v);
    }
}

// dres[k] = bw[k]*dout ; dots[k] = <dout, res[k]>
__global__ __launch_bounds__(256) void k_sink_bwd(int K, const float* __restrict__ bw, SinkPtrs ptrs,
                                                  const float* __restrict__ dout, uint64_t count4,
                                                  double* __restrict__ dots) {
    __shared__ float red[4 * TFNAS_MAX_SINK];
    float b[TFNAS_MAX_SINK], dot[TFNAS_MAX_SINK];
    for (int k = 0; k < TFNAS_MAX_SINK; ++k) {
        b[k] = k < K ? bw[k] : 0.f;
        dot[k] = 0.f;
    }
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < count4; i += (uint64_t)gridDim.x * 256) {
        const f32x4 dv = ld4(dout + 4 * i);
#pragma unroll
        for (int k = 0; k < TFNAS_MAX_SINK; ++k) {
            if (k < K) {
                const f32x4 r = ld4(ptrs.res[k] + 4 * i);
                dot[k] += (dv.x * r.x + dv.y * r.y) + (dv.z * r.z + dv.w * r.w);
                if (ptrs.dres[k]) st4(ptrs.dres[k] + 4 * i, splat4(b[k]) * dv);
            }
        }
    }
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < TFNAS_MAX_SINK; ++k) {
        dot[k] = wave_sum(dot[k]);
        if (lane == 0) red[wv * TFNAS_MAX_SINK + k] = dot[k];
    }
    __syncthreads();
    if (threadIdx.x < K) {
        float s = 0.f;
        for (int ww = 0; ww < 4; ++ww) s += red[ww * TFNAS_MAX_SINK + threadIdx.x];
        atomic_add_f64(dots + threadIdx.x, (double)s);
    }
}

// dbetas[j] = bw_j (g_j - sum_i g_i bw_i), g_k = dots[k] + dlat*cum_k ;  dcell_lat[j] = dlat * sum_{k>=j} bw_k
__global__ void k_sink_bwd_fin(int K, const float* __restrict__ bw, const float* __restrict__ cell_lat,
                               const float* __restrict__ dlat, const double* __restrict__ dots,
                               float* __restrict__ dbetas, float* __restrict__ dcell_lat) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const float dl = dlat ? *dlat : 0.f;
    float gk[TFNAS_MAX_SINK], cum = 0.f, dot = 0.f;
    for (int k = 0; k < K; ++k) {
        if (cell_lat) cum = (k == 0) ? cell_lat[0] : cum + cell_lat[k];
        gk[k] = (float)dots[k] + dl * cum;
        dot += gk[k] * bw[k];
    }
    for (int j = 0; j < K; ++j) {
        if (dbetas) dbetas[j] = bw[j] * (gk[j] - dot);
        if (dcell_lat) {
            float s = 0.f;
            for (int k = j; k < K; ++k) s += bw[k];
            dcell_lat[j] = dl * s;
        }
    }
}

// ============================================================================ host launchers
int launch_arch_fwd(int ncell, const float* const* la, const float* e, const float* lat, float T, float* w,
                    float* cell_lat, hipStream_t s) {
    ProfScope _prof(TK_SMALL, s);
    PtrPack pk;
    for (int i = 0; i < ncell; ++i) pk.p[i] = la[i];
    hipLaunchKernelGGL(k_arch_fwd, dim3(1), dim3(64), 0, s, ncell, pk, e, lat, T, w, cell_lat);
    return (int)hipGetLastError();
}

int launch_arch_bwd(int ncell, const float* w, const float* lat, const float* dw, const float* dcl, float T,
                    float* const* dla, hipStream_t s) {
    ProfScope _prof(TK_SMALL, s);
    MutPtrPack pk;
    for (int i = 0; i < ncell; ++i) pk.p[i] = dla[i];
    hipLaunchKernelGGL(k_arch_bwd, dim3(1), dim3(64), 0, s, ncell, w, lat, dw, dcl, T, pk);
    return (int)hipGetLastError();
}

// p <- log_softmax(p) for up to TFNAS_MAX_CELLS small vectors: (x - max) - log(sum exp(x - max)), like torch's log_softmax
struct ProjPack {
    float* p[TFNAS_MAX_CELLS];
    int len[TFNAS_MAX_CELLS];
};
__global__ void k_arch_project(int n, ProjPack pk) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    float* __restrict__ x = pk.p[t];
    const int len = pk.len[t];
    float v[8], m = -INFINITY;
    for (int i = 0; i < len; ++i) {
        v[i] = x[i];
        m = fmaxf(m, v[i]);
    }
    float s = 0.f;
    for (int i = 0; i < len; ++i) s += expf(v[i] - m);
    const float ls = logf(s);
    for (int i = 0; i < len; ++i) x[i] = (v[i] - m) - ls;
}

int launch_arch_project(int n, float* const* p, const int32_t* len, hipStream_t s) {
    ProfScope _prof(TK_SMALL, s);
    ProjPack pk;
    for (int i = 0; i < n; ++i) {
        pk.p[i] = p[i];
        pk.len[i] = len[i];
    }
    hipLaunchKernelGGL(k_arch_project, dim3(1), dim3(64), 0, s, n, pk);
    return (int)hipGetLastError();
}

int launch_arch_sample(int ncell, const float* const* la, const uint8_t* mask, const float* e, float T, int mode,
                       int32_t* pos, hipStream_t s) {
    ProfScope _prof(TK_SMALL, s);
    PtrPack pk;
    for (int i = 0; i < ncell; ++i) pk.p[i] = la[i];
    hipLaunchKernelGGL(k_arch_sample, dim3(1), dim3(64), 0, s, ncell, pk, mask, e, T, mode, pos);
    return (int)hipGetLastError();
}

static unsigned stream_blocks(uint64_t count4) {
    uint64_t b = cdiv64(count4, 256 * 2);
    if (b > 2048) b = 2048;
    if (b < 1) b = 1;
    return (unsigned)b;
}

int launch_sink_fwd(int K, const float* betas, const float* const* res, const float* cell_lat, uint64_t count,
                    float* out, float* out_lat, float* bw, hipStream_t s) {
    ProfScope _prof(TK_SMALL, s);
    SinkPtrs pk = {};
    for (int k = 0; k < K; ++k) pk.res[k] = res[k];
    hipLaunchKernelGGL(k_sink_fwd, dim3(stream_blocks(count / 4)), dim3(256), 0, s, K, betas, pk, cell_lat,
                       count / 4, out, out_lat, bw);
    return (int)hipGetLastError();
}

// dst = scale[0] * src  (the last depth output's share of the sink gradient when no d betas is wanted: weight step)
__global__ __launch_bounds__(256) void k_scale_copy(float* __restrict__ dst, const float* __restrict__ src,
                                                    const float* __restrict__ scale, uint64_t count4) {
    const float w = scale[0];
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < count4; i += (uint64_t)gridDim.x * 256)
        st4(dst + 4 * i, splat4(w) * ld4(src + 4 * i));
}

int launch_scale_copy(float* dst, const float* src, const float* scale, uint64_t count, hipStream_t s) {
    ProfScope _prof(TK_SMALL, s);
    hipLaunchKernelGGL(k_scale_copy, dim3(stream_blocks(count / 4)), dim3(256), 0, s, dst, src, scale, count / 4);
    return (int)hipGetLastError();
}

int launch_sink_bwd(int K, const float* bw, const float* const* res, const float* cell_lat, const float* dout,
                    const float* dlat, uint64_t count, float* const* dres, float* dbetas, float* dcell_lat,
                    double* dots, hipStream_t s) {
    ProfScope _prof(TK_SMALL, s);
    SinkPtrs pk = {};
    for (int k = 0; k < K; ++k) {
        pk.res[k] = res[k];
        pk.dres[k] = dres[k];
    }
    hipError_t e = hipMemsetAsync(dots, 0, sizeof(double) * TFNAS_MAX_SINK, s);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(k_sink_bwd, dim3(stream_blocks(count / 4)), dim3(256), 0, s, K, bw, pk, dout, count / 4, dots);
    hipLaunchKernelGGL(k_sink_bwd_fin, dim3(1), dim3(64), 0, s, K, bw, cell_lat, dlat, dots, dbetas, dcell_lat);
    return (int)hipGetLastError();
}
