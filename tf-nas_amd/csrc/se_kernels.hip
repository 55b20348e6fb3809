// Squeeze-excite "excite" stage (two 1x1 convs on the pooled [N, mc] tensor) as small fp32 MFMA GEMMs with the
// batch dimension as GEMM rows, forward / backward / weight gradients.
//
// Reference: squeeze_excite = conv_reduce(+bias) -> act -> conv_expand(+bias), gate = sigmoid(.)
//            (models/layers.py:509-526, forward :548-550) and its autograd backward.
//
//   MODE 0  hpre [N,se] = pooled[N,mc] W_r^T + b_r                                   (NT)
//   MODE 1  gate [N,mc] = sigmoid(act(hpre)[N,se] W_e^T + b_e)                       (NT)
//   MODE 2  dhpre[N,se] = (dgl[N,mc] W_e) * act'(hpre),  dgl = dgate*gate*(1-gate)   (NN)
//   MODE 3  dpool[N,mc] = dhpre[N,se] W_r                                            (NN)
//   MODE 4  g_se_e[mc,se] = dgl^T act(hpre)                                          (TN, K = batch)
//   MODE 5  g_se_r[se,mc] = (pooled^T dhpre)^T                                       (TN, K = batch)
#include <cstring>
#include "gemm_core.h"
#include "kernels.h"
#include "prof.h"

#ifndef TFNAS_SE_PF2
#define TFNAS_SE_PF2 false     /* prefetch distance 2 in the K loop (gemm_core.h); measured: no gain (A/B switch) */
#endif

struct SeArgs {
    const float* pooled;   // [N][M]
    const float* gate;     // [N][M]
    const float* hpre;     // [N][SE]
    const float* dgate;    // [N][M]
    const float* dhpre;    // [N][SE]
    float* out0;           // mode-dependent output
    float* scratch;        // MODE 0 / 2 with ksplit > 1: raw partial products [ksplit][N][SE]
    int ksplit;            // K-splits of the long-K GEMMs (MODE 0 / 2): their output is only N x se, so without
                           // splitting a handful of workgroups would each walk up to 72 K-chunks back to back
};

__device__ __forceinline__ int se_group_idx(const TfnasCellDesc& d, int idx) {
    for (int g = 0; g < d.G; ++g)
        if (d.g[g].se > 0 && idx-- == 0) return g;
    return -1;
}

template <int MODE, int ACT>
__global__ __launch_bounds__(256) void k_se_gemm(TfnasCellDesc d, SeArgs a) {
    constexpr int NT = 4;
    using T = GT<NT>;
    __shared__ __attribute__((aligned(16))) float lds[T::LDS_FLOATS];
    const int g = se_group_idx(d, blockIdx.z);
    if (g < 0) return;
    const int mc = d.g[g].mc, mcp = d.g[g].mcp, off = d.g[g].off, se = d.g[g].se, so = d.g[g].se_off;
    const int N = d.N, M = d.M, SE = d.SE;
    const bool mc_al = (mc & 3) == 0;
    const int tid = threadIdx.x, lane = tid & 63, lr = lane & 15, lq = lane >> 4, wrow = (tid >> 6) * 32;
    const bool split = (MODE == 0 || MODE == 2) && a.ksplit > 1;
    const int ks = split ? blockIdx.x % a.ksplit : 0;
    const int r0 = (split ? blockIdx.x / a.ksplit : blockIdx.x) * 128;   // first GEMM row of this tile
    const int n0 = blockIdx.y * T::BN;       // first GEMM column
    // chunk range of this K-split (MODE 0 / 2: K = mid channels)
    const int cps = split ? (((mcp + 15) >> 4) + a.ksplit - 1) / a.ksplit : 0, cbase = ks * cps;
    f32x4 acc[2][NT];
    acc_zero<NT>(acc);

    // Two-phase loaders (gemm_core.h): the load pieces read unconditionally from clamped addresses, the transforms mask.
    // These launches are 1-20 workgroups walking 8-72 K-chunks back to back -- nothing else on the CU hides a wait that
    // sits between a load and the MFMAs (the one-piece loaders had 2-3 serialized round trips per chunk there).
    struct Pair { f32x4 a, b; };
    const f32x4 one4 = splat4(1.f);
    XfId id;

    if (MODE == 0) {
        if (r0 >= N || n0 >= se) return;
        auto la = [&](int c, int, int row, int kl) -> f32x4 {
            const int n = min(r0 + row, N - 1), k = min((cbase + c) * 16 + kl, mcp - 4);
            return ld4(a.pooled + (size_t)n * M + off + k);
        };
        auto xa = [&](f32x4 v, int c, int, int row, int kl) -> f32x4 {
            return (r0 + row < N && (cbase + c) * 16 + kl < mcp) ? v : zero4();
        };
        auto lb = [&](int c, int, int nn, int kl) -> f32x4 {
            const int j = n0 + nn, k = (cbase + c) * 16 + kl;
            if (!mc_al) return (j < se) ? ld4_guard(d.g[g].w_se_r + (size_t)j * mc, k, mc, false) : zero4();
            return ld4(d.g[g].w_se_r + (size_t)min(j, se - 1) * mc + min(k, mc - 4));
        };
        auto xb = [&](f32x4 v, int c, int, int nn, int kl) -> f32x4 {
            if (!mc_al) return v;
            return (n0 + nn < se && (cbase + c) * 16 + kl < mc) ? v : zero4();
        };
        const int tot = (mcp + 15) >> 4;
        gemm_mainloop2<NT, true, true, true, TFNAS_SE_PF2>(la, xa, lb, xb, split ? max(0, min(cps, tot - cbase)) : tot, acc, lds);
    } else if (MODE == 1) {
        if (r0 >= N || n0 >= mcp) return;
        auto la = [&](int c, int, int row, int kl) -> f32x4 {
            const int n = min(r0 + row, N - 1), k = min(c * 16 + kl, se - 4);
            return ld4(a.hpre + (size_t)n * SE + so + k);
        };
        auto xa = [&](f32x4 v, int c, int, int row, int kl) -> f32x4 {
            return (r0 + row < N && c * 16 + kl < se) ? act_f4<ACT>(v) : zero4();
        };
        auto lb = [&](int c, int, int nn, int kl) -> f32x4 {
            const int col = min(n0 + nn, mc - 1), k = min(c * 16 + kl, se - 4);
            return ld4(d.g[g].w_se_e + (size_t)col * se + k);
        };
        auto xb = [&](f32x4 v, int c, int, int nn, int kl) -> f32x4 {
            return (n0 + nn < mc && c * 16 + kl < se) ? v : zero4();
        };
        gemm_mainloop2<NT, true, true, true, TFNAS_SE_PF2>(la, xa, lb, xb, (se + 15) >> 4, acc, lds);
    } else if (MODE == 2) {
        if (r0 >= N || n0 >= se) return;
        auto la = [&](int c, int, int row, int kl) -> Pair {
            const int n = min(r0 + row, N - 1), k = min((cbase + c) * 16 + kl, mcp - 4);
            Pair r;
            r.a = ld4(a.gate + (size_t)n * M + off + k);
            r.b = ld4(a.dgate + (size_t)n * M + off + k);
            return r;
        };
        auto xa = [&](Pair r, int c, int, int row, int kl) -> f32x4 {
            return (r0 + row < N && (cbase + c) * 16 + kl < mcp) ? r.b * r.a * (one4 - r.a) : zero4();
        };
        auto lb = [&](int c, int, int kl, int nn) -> f32x4 {
            const int k = min((cbase + c) * 16 + kl, mc - 1), j = min(n0 + nn, se - 4);
            return ld4(d.g[g].w_se_e + (size_t)k * se + j);
        };
        auto xb = [&](f32x4 v, int c, int, int kl, int nn) -> f32x4 {
            return ((cbase + c) * 16 + kl < mc && n0 + nn < se) ? v : zero4();
        };
        const int tot = (mcp + 15) >> 4;
        gemm_mainloop2<NT, true, false, true, TFNAS_SE_PF2>(la, xa, lb, xb, split ? max(0, min(cps, tot - cbase)) : tot, acc, lds);
    } else if (MODE == 3) {
        if (r0 >= N || n0 >= mcp) return;
        auto la = [&](int c, int, int row, int kl) -> f32x4 {
            const int n = min(r0 + row, N - 1), k = min(c * 16 + kl, se - 4);
            return ld4(a.dhpre + (size_t)n * SE + so + k);
        };
        auto xa = [&](f32x4 v, int c, int, int row, int kl) -> f32x4 {
            return (r0 + row < N && c * 16 + kl < se) ? v : zero4();
        };
        auto lb = [&](int c, int, int kl, int nn) -> f32x4 {
            const int k = c * 16 + kl;
            if (!mc_al) return (k < se) ? ld4_guard(d.g[g].w_se_r + (size_t)k * mc, n0 + nn, mc, false) : zero4();
            return ld4(d.g[g].w_se_r + (size_t)min(k, se - 1) * mc + min(n0 + nn, mc - 4));
        };
        auto xb = [&](f32x4 v, int c, int, int kl, int nn) -> f32x4 {
            if (!mc_al) return v;
            return (c * 16 + kl < se && n0 + nn < mc) ? v : zero4();
        };
        gemm_mainloop2<NT, true, false, true, TFNAS_SE_PF2>(la, xa, lb, xb, (se + 15) >> 4, acc, lds);
    } else {   // MODE 4 / 5: rows = mid channels, cols = se, K = batch
        if (r0 >= mcp || n0 >= se) return;
        auto la = [&](int c, int, int kl, int m) -> Pair {
            const int n = min(c * 16 + kl, N - 1), ch = min(r0 + m, mcp - 4);
            Pair r;
            if (MODE == 4) {
                r.a = ld4(a.gate + (size_t)n * M + off + ch);
                r.b = ld4(a.dgate + (size_t)n * M + off + ch);
            } else {
                r.a = ld4(a.pooled + (size_t)n * M + off + ch);
                r.b = r.a;
            }
            return r;
        };
        auto xa = [&](Pair r, int c, int, int kl, int m) -> f32x4 {
            const f32x4 v = MODE == 4 ? r.b * r.a * (one4 - r.a) : r.a;
            return (c * 16 + kl < N && r0 + m < mcp) ? v : zero4();
        };
        auto lb = [&](int c, int, int kl, int nn) -> f32x4 {
            const int n = min(c * 16 + kl, N - 1), j = min(n0 + nn, se - 4);
            return ld4((MODE == 4 ? a.hpre : a.dhpre) + (size_t)n * SE + so + j);
        };
        auto xb = [&](f32x4 v, int c, int, int kl, int nn) -> f32x4 {
            const f32x4 w = MODE == 4 ? act_f4<ACT>(v) : v;
            return (c * 16 + kl < N && n0 + nn < se) ? w : zero4();
        };
        gemm_mainloop2<NT, false, false, true, TFNAS_SE_PF2>(la, xa, lb, xb, (N + 15) >> 4, acc, lds);
    }
    (void)id;

#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = r0 + wrow + 16 * i + 4 * lq + r;
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int col = n0 + 16 * j + lr;
                const float v = acc[i][j][r];
                if (split) {
                    if (row < N && col < se) a.scratch[((size_t)ks * N + row) * SE + so + col] = v;
                } else if (MODE == 0) {
                    if (row < N && col < se) a.out0[(size_t)row * SE + so + col] = v + d.g[g].b_se_r[col];
                } else if (MODE == 1) {
                    if (row < N && col < mcp)
                        a.out0[(size_t)row * M + off + col] = col < mc ? sigmoid_f(v + d.g[g].b_se_e[col]) : 0.f;
                } else if (MODE == 2) {
                    if (row < N && col < se)
                        a.out0[(size_t)row * SE + so + col] = v * act_d<ACT>(a.hpre[(size_t)row * SE + so + col]);
                } else if (MODE == 3) {
                    if (row < N && col < mcp) a.out0[(size_t)row * M + off + col] = v;
                } else if (MODE == 4) {
                    if (row < mc && col < se) d.g[g].g_se_e[(size_t)row * se + col] = v;
                } else {
                    if (row < mc && col < se) d.g[g].g_se_r[(size_t)col * mc + row] = v;
                }
            }
        }
}

// second stage of the K-split MODE 0 / 2 GEMMs: sum the splits (fixed order) and apply the epilogue
//   MODE 0: hpre = sum + b_r          MODE 2: dhpre = sum * act'(hpre)
template <int MODE, int ACT>
__global__ __launch_bounds__(256) void k_se_finish(TfnasCellDesc d, SeArgs a) {
    const int N = d.N, SE = d.SE;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= N * SE) return;
    const int n = idx / SE, j = idx - n * SE;
    float s = 0.f;
    for (int k = 0; k < a.ksplit; ++k) s += a.scratch[((size_t)k * N + n) * SE + j];
    if (MODE == 0) {
        int g = 0;
        for (; g < d.G; ++g)
            if (d.g[g].se > 0 && j >= d.g[g].se_off && j < d.g[g].se_off + d.g[g].se) break;
        a.out0[idx] = s + d.g[g].b_se_r[j - d.g[g].se_off];
    } else {
        a.out0[idx] = s * act_d<ACT>(a.hpre[idx]);
    }
}

// bias gradients: gb_se_e[c] = sum_n dgl[n][c] ; gb_se_r[j] = sum_n dhpre[n][j]
// 64 columns x 4 batch-lanes per workgroup, 8 loads in flight per thread
__global__ __launch_bounds__(256) void k_se_bias_grad(TfnasCellDesc d, SeArgs a) {
    __shared__ float buf[4][64];
    const int g = se_group_idx(d, blockIdx.y);
    if (g < 0) return;
    const int mc = d.g[g].mc, off = d.g[g].off, se = d.g[g].se, so = d.g[g].se_off;
    const int N = d.N, M = d.M, SE = d.SE;
    const int cl = threadIdx.x & 63, nl = threadIdx.x >> 6;
    const int idx = blockIdx.x * 64 + cl;
    float s = 0.f;
    if (idx < mc) {
        for (int n0 = nl; n0 < N; n0 += 32) {
            float gt[8], dg[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int n = n0 + 4 * u < N ? n0 + 4 * u : 0;
                gt[u] = a.gate[(size_t)n * M + off + idx];
                dg[u] = a.dgate[(size_t)n * M + off + idx];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (n0 + 4 * u < N) s += dg[u] * gt[u] * (1.f - gt[u]);
        }
    } else if (idx < mc + se) {
        const int j = idx - mc;
        for (int n0 = nl; n0 < N; n0 += 32) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = a.dhpre[(size_t)(n0 + 4 * u < N ? n0 + 4 * u : 0) * SE + so + j];
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (n0 + 4 * u < N) s += v[u];
        }
    }
    buf[nl][cl] = s;
    __syncthreads();
    if (nl == 0) {
        const float t = (buf[0][cl] + buf[1][cl]) + (buf[2][cl] + buf[3][cl]);
        if (idx < mc) d.g[g].gb_se_e[idx] = t;
        else if (idx < mc + se) d.g[g].gb_se_r[idx - mc] = t;
    }
}

// ============================================================================ fused excite stage (forward / backward)
// One workgroup per (image, SE group) runs BOTH 1x1 convolutions of that image: the pooled row lives in LDS, the hidden
// vector never leaves the workgroup.  The excite stage is ~1 MFLOP per image -- as GEMMs it was 2-3 dependent launches
// (K-split MODE 0 + finish + MODE 1) of mostly empty MFMA tiles, 30-50 us on the critical path of every cell; here it is
// one launch whose time is a few L2 round trips.  Plain fp32 FMA chains (fixed order: bit-reproducible).
// Stage `rows` rows of a row-major [.][se] matrix (starting at row c0) into LDS with an odd row stride (se | 1), so that
// thread-per-row and thread-per-column walks are both bank-conflict free.  The global side is one contiguous block.
__device__ __forceinline__ void se_stage_rows(float* wl, const float* __restrict__ w, int c0, int rows, int se, int ld) {
    const int n = rows * se;
    const float* __restrict__ src = w + (size_t)c0 * se;
    for (int i = threadIdx.x; i < n; i += 256) {
        const int r = i / se, j = i - r * se;
        wl[r * ld + j] = src[i];
    }
}

constexpr int SE_CH = 128;   // channels per staged block of W_e

template <int ACT>
__global__ __launch_bounds__(256) void k_se_fused_fwd(TfnasCellDesc d, const float* __restrict__ pooled,
                                                      float* __restrict__ hpre, float* __restrict__ gate) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int g = se_group_idx(d, blockIdx.y);
    if (g < 0) return;
    const int mc = d.g[g].mc, mcp = d.g[g].mcp, off = d.g[g].off, se = d.g[g].se, so = d.g[g].se_off;
    const int n = blockIdx.x, M = d.M, SE = d.SE;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int mcq = (mcp + 3) & ~3, ld = se | 1;
    float* ps = sm;                  // [mcq] pooled row
    float* hs = sm + mcq;            // [se]  act(hpre)
    float* wl = hs + ((se + 3) & ~3);   // [SE_CH][ld] block of W_e
    for (int c = tid; c < mcq; c += 256) ps[c] = c < mc ? pooled[(size_t)n * M + off + c] : 0.f;
    __syncthreads();
    // hidden units: one wave per row of W_r, three rows in flight (rows are contiguous: coalesced 16-byte loads)
    const bool al = (mc & 3) == 0;
    for (int j0 = wave; j0 < se; j0 += 12) {
        float s[3] = {0.f, 0.f, 0.f};
        if (al) {
            for (int c = 4 * lane; c < mc; c += 256) {
                const f32x4 p4 = ld4(ps + c);
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    const int j = j0 + 4 * r;
                    if (j < se) {
                        const f32x4 w4 = ld4(d.g[g].w_se_r + (size_t)j * mc + c);
                        s[r] += (w4.x * p4.x + w4.y * p4.y) + (w4.z * p4.z + w4.w * p4.w);
                    }
                }
            }
        } else {
            for (int c = lane; c < mc; c += 64) {
                const float p = ps[c];
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    const int j = j0 + 4 * r;
                    if (j < se) s[r] += d.g[g].w_se_r[(size_t)j * mc + c] * p;
                }
            }
        }
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const int j = j0 + 4 * r;
            if (j < se) {
                const float v = wave_sum(s[r]) + d.g[g].b_se_r[j];
                if (lane == 0) {
                    hpre[(size_t)n * SE + so + j] = v;
                    hs[j] = act_f<ACT>(v);
                }
            }
        }
    }
    // gate: W_e is [mc][se]; blocks of SE_CH rows go through LDS, two threads per channel (even / odd hidden units)
    for (int c0 = 0; c0 < mcp; c0 += SE_CH) {
        const int rows = min(SE_CH, mc - c0);
        __syncthreads();                                   // hs complete / previous block consumed
        if (rows > 0) se_stage_rows(wl, d.g[g].w_se_e, c0, rows, se, ld);
        __syncthreads();
        const int cl = tid >> 1, half = tid & 1, c = c0 + cl;
        float sacc = 0.f;
        if (cl < rows) {
            const float* row = wl + cl * ld;
            for (int j = half; j < se; j += 2) sacc += row[j] * hs[j];
        }
        sacc += __shfl_xor(sacc, 1, 64);
        if (half == 0 && c < mcp) gate[(size_t)n * M + off + c] = c < mc ? sigmoid_f(sacc + d.g[g].b_se_e[c]) : 0.f;
    }
}

// dgl = dgate*gate*(1-gate);  dhpre = (dgl W_e) * act'(hpre);  dpooled = dhpre W_r
template <int ACT>
__global__ __launch_bounds__(256) void k_se_fused_bwd(TfnasCellDesc d, const float* __restrict__ dgate,
                                                      const float* __restrict__ gate, const float* __restrict__ hpre,
                                                      float* __restrict__ dhpre, float* __restrict__ dpooled) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int g = se_group_idx(d, blockIdx.y);
    if (g < 0) return;
    const int mc = d.g[g].mc, mcp = d.g[g].mcp, off = d.g[g].off, se = d.g[g].se, so = d.g[g].se_off;
    const int n = blockIdx.x, M = d.M, SE = d.SE;
    const int tid = threadIdx.x;
    const int mcq = (mcp + 3) & ~3, ld = se | 1, seq = (se + 3) & ~3;
    float* dgl = sm;                 // [mcq]
    float* dhs = sm + mcq;           // [seq]
    float* red = dhs + seq;          // [256]
    float* wl = red + 256;           // [SE_CH][ld]
    for (int c = tid; c < mcq; c += 256) {
        float v = 0.f;
        if (c < mc) {
            const float gt = gate[(size_t)n * M + off + c];
            v = dgate[(size_t)n * M + off + c] * gt * (1.f - gt);
        }
        dgl[c] = v;
    }
    // dh[j] = sum_c dgl[c] W_e[c][j]: thread = (hidden unit j, channel phase); W_e blocks staged in LDS
    const int sep = se < 256 ? se : 256, nph = 256 / sep;       // phases per hidden unit
    const int jj = tid % sep, ph = tid / sep;
    float dh = 0.f;
    for (int c0 = 0; c0 < mc; c0 += SE_CH) {
        const int rows = min(SE_CH, mc - c0);
        __syncthreads();
        se_stage_rows(wl, d.g[g].w_se_e, c0, rows, se, ld);
        __syncthreads();
        if (ph < nph)
            for (int j = jj; j < se; j += sep)                  // (se > 256: several units per thread, same phase)
                for (int cl = ph; cl < rows; cl += nph) dh += dgl[c0 + cl] * wl[cl * ld + j];
    }
    red[tid] = dh;
    __syncthreads();
    for (int j = tid; j < se && j < sep; j += 256) {
        float t = 0.f;
        for (int p = 0; p < nph; ++p) t += red[p * sep + j];
        const float v = t * act_d<ACT>(hpre[(size_t)n * SE + so + j]);
        dhs[j] = v;
        dhpre[(size_t)n * SE + so + j] = v;
    }
    __syncthreads();
    // dpooled[c] = sum_j dhs[j] W_r[j][c]: coalesced over c, 8 rows in flight
    const bool al = (mc & 3) == 0;
    if (al) {
        for (int c = 4 * tid; c < mcq; c += 1024) {
            f32x4 acc = zero4();
            if (c < mc) {
                const float* __restrict__ wr = d.g[g].w_se_r + c;
                int j = 0;
                for (; j + 7 < se; j += 8) {
                    f32x4 w[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) w[u] = ld4(wr + (size_t)(j + u) * mc);
#pragma unroll
                    for (int u = 0; u < 8; ++u) acc += w[u] * splat4(dhs[j + u]);
                }
                for (; j < se; ++j) acc += ld4(wr + (size_t)j * mc) * splat4(dhs[j]);
            }
            if (c < mcp) st4(dpooled + (size_t)n * M + off + c, acc);
        }
    } else {
        for (int c = tid; c < mcp; c += 256) {
            float s0 = 0.f, s1 = 0.f;
            if (c < mc) {
                const float* __restrict__ wr = d.g[g].w_se_r + c;
                int j = 0;
                for (; j + 1 < se; j += 2) {
                    s0 += dhs[j] * wr[(size_t)j * mc];
                    s1 += dhs[j + 1] * wr[(size_t)(j + 1) * mc];
                }
                if (j < se) s0 += dhs[j] * wr[(size_t)j * mc];
            }
            dpooled[(size_t)n * M + off + c] = s0 + s1;
        }
    }
}


// ============================================================================ wave-level excite GEMMs (no LDS staging)
// The excite stage is N x mc x se MACs per group -- 10^8 flops -- but as LDS-tiled GEMMs every launch walks 6..72 K-chunks of
// (load -> LDS -> barrier -> MFMA) back to back with one workgroup per CU: 12-57 us per launch, 50-80 us per cell and
// direction on the dependency chain of every cell with wide candidates (and 30-40 us for the per-image fused kernels, which
// re-read both weight matrices once per IMAGE).  Here the operands go from global memory straight into MFMA lanes:
//   NT (MODE 0 / 1): both operands K-contiguous.  Lane (i = lane % 16, q = lane / 16) loads A[row i][k0 + 4q .. +3] and
//       B[col i][k0 + 4q .. +3]; MFMA step t takes component t of both, i.e. K slot q of step t carries k = k0 + 4q + t
//       (a sum over k does not care about the order): 2 loads per 4 MFMAs, a 16 x 16 tile per wave.
//   NN (MODE 2 / 3): A K-contiguous as above, B[k][col] column-contiguous: for step t the lane loads the float4
//       B[k0 + 4q + t][col0 + 4i .. +3] whose components are four column tiles with columns 4i + comp: a 16 x 64 tile per wave.
// All loads of a batch of K-steps are issued before its MFMAs; long K (MODE 0 / 2: K = mid channels) is split over the four
// waves of a workgroup and summed through LDS in a fixed order.  Requires mc % 4 == 0 for every SE group (aligned float4 rows
// of W_r; ragged widths keep the GEMM path).  Weights are read once per 16 images instead of once per image.
#define SEW_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

template <int MODE, int ACT>      // MODE 0: hpre = pooled W_r^T + b_r (K = mc, split over the waves);  MODE 1: gate (K = se)
__global__ __launch_bounds__(256) void k_se_nt(TfnasCellDesc d, SeArgs a) {
    __shared__ f32x4 red[4][64];
    const int g = se_group_idx(d, blockIdx.z);
    if (g < 0) return;
    const int mc = d.g[g].mc, mcp = d.g[g].mcp, off = d.g[g].off, se = d.g[g].se, so = d.g[g].se_off;
    const int N = d.N, M = d.M, SE = d.SE;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, lj = lane & 15, lk = lane >> 4;
    const int n0 = blockIdx.x * 16;
    const int K = MODE == 0 ? mc : se, ncol = MODE == 0 ? se : mcp;
    const int j0 = MODE == 0 ? (int)blockIdx.y * 16 : (int)blockIdx.y * 64 + 16 * wave;
    if ((MODE == 0 ? j0 : (int)blockIdx.y * 64) >= ncol) return;          // (workgroup-uniform)
    const bool wave_on = j0 < ncol;
    const int rowA = min(n0 + lj, N - 1);
    const int rowB = min(j0 + lj, (MODE == 0 ? se : mc) - 1);
    const float* __restrict__ pa = MODE == 0 ? a.pooled + (size_t)rowA * M + off : a.hpre + (size_t)rowA * SE + so;
    const float* __restrict__ pb = MODE == 0 ? d.g[g].w_se_r + (size_t)rowB * mc : d.g[g].w_se_e + (size_t)rowB * se;
    const int nsteps = (K + 15) >> 4;
    const int first = MODE == 0 ? wave : 0, stride = MODE == 0 ? 4 : 1;
    f32x4 acc = zero4();
    constexpr int UB = 6;                                                    // K-steps per batch of loads
    for (int s0 = first; wave_on && s0 < nsteps; s0 += UB * stride) {
        f32x4 av[UB], bv[UB];
#pragma unroll
        for (int u = 0; u < UB; ++u) {
            const int k = 16 * (s0 + u * stride) + 4 * lk, kc = min(k, K - 4);
            av[u] = ld4(pa + kc);
            bv[u] = ld4(pb + kc);
        }
#pragma unroll
        for (int u = 0; u < UB; ++u) {
            const int k = 16 * (s0 + u * stride) + 4 * lk;
            f32x4 x = av[u];
            if (MODE == 1) x = act_f4<ACT>(x);
            const f32x4 y = (k < K) ? bv[u] : zero4();                      // (K % 4 == 0: a quad is inside or outside)
#pragma unroll
            for (int t = 0; t < 4; ++t) acc = SEW_MFMA(x[t], y[t], acc);
        }
    }
    if (MODE == 0) {
        red[wave][lane] = acc;
        __syncthreads();
        if (wave != 0) return;
        acc = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
    } else if (!wave_on) {
        return;
    }
    const int j = j0 + lj;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int n = n0 + 4 * lk + r;
        if (n >= N || j >= ncol) continue;
        if (MODE == 0) a.out0[(size_t)n * SE + so + j] = acc[r] + d.g[g].b_se_r[j];
        else a.out0[(size_t)n * M + off + j] = j < mc ? sigmoid_f(acc[r] + d.g[g].b_se_e[j]) : 0.f;
    }
}

template <int MODE, int ACT>      // MODE 2: dhpre = (dgl W_e) * act'(hpre) (K = mc, split over the waves);  MODE 3: dpooled = dhpre W_r
__global__ __launch_bounds__(256) void k_se_nn(TfnasCellDesc d, SeArgs a) {
    __shared__ f32x4 red[MODE == 2 ? 4 : 1][4][64];
    const int g = se_group_idx(d, blockIdx.z);
    if (g < 0) return;
    const int mc = d.g[g].mc, mcp = d.g[g].mcp, off = d.g[g].off, se = d.g[g].se, so = d.g[g].se_off;
    const int N = d.N, M = d.M, SE = d.SE;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, lj = lane & 15, lk = lane >> 4;
    const int n0 = blockIdx.x * 16;
    const int K = MODE == 2 ? mc : se, ncol = MODE == 2 ? se : mcp, ldb = MODE == 2 ? se : mc;
    const int c0 = MODE == 2 ? (int)blockIdx.y * 64 : (int)blockIdx.y * 256 + 64 * wave;
    if ((MODE == 2 ? c0 : (int)blockIdx.y * 256) >= ncol) return;
    const bool wave_on = c0 < ncol;
    const int rowA = min(n0 + lj, N - 1);
    const int colB = min(c0 + 4 * lj, ldb - 4);                             // (ldb % 4 == 0)
    const float* __restrict__ wB = MODE == 2 ? d.g[g].w_se_e : d.g[g].w_se_r;
    const int nsteps = (K + 15) >> 4;
    const int first = MODE == 2 ? wave : 0, stride = MODE == 2 ? 4 : 1;
    f32x4 acc[4] = {zero4(), zero4(), zero4(), zero4()};
    constexpr int UB = 3;
    for (int s0 = first; wave_on && s0 < nsteps; s0 += UB * stride) {
        f32x4 av[UB], gv[UB], bv[UB][4];
#pragma unroll
        for (int u = 0; u < UB; ++u) {
            const int k = 16 * (s0 + u * stride) + 4 * lk, kc = min(k, K - 4);
            if (MODE == 2) {
                av[u] = ld4(a.dgate + (size_t)rowA * M + off + kc);
                gv[u] = ld4(a.gate + (size_t)rowA * M + off + kc);
            } else {
                av[u] = ld4(a.dhpre + (size_t)rowA * SE + so + kc);
            }
#pragma unroll
            for (int t = 0; t < 4; ++t) bv[u][t] = ld4(wB + (size_t)min(k + t, K - 1) * ldb + colB);
        }
#pragma unroll
        for (int u = 0; u < UB; ++u) {
            const int k = 16 * (s0 + u * stride) + 4 * lk;
            f32x4 x = av[u];
            if (MODE == 2) x = x * gv[u] * (splat4(1.f) - gv[u]);
            x = (k < K) ? x : zero4();
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int cp = 0; cp < 4; ++cp) acc[cp] = SEW_MFMA(x[t], bv[u][t][cp], acc[cp]);
        }
    }
    if (MODE == 2) {
#pragma unroll
        for (int cp = 0; cp < 4; ++cp) red[wave][cp][lane] = acc[cp];
        __syncthreads();
        if (wave != 0) return;
#pragma unroll
        for (int cp = 0; cp < 4; ++cp) acc[cp] = (red[0][cp][lane] + red[1][cp][lane]) + (red[2][cp][lane] + red[3][cp][lane]);
    } else if (!wave_on) {
        return;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int n = n0 + 4 * lk + r;
        if (n >= N) continue;
        const int col = c0 + 4 * lj;                                         // four consecutive columns: one 16-byte store
        if (col >= ncol) continue;
        f32x4 q = {acc[0][r], acc[1][r], acc[2][r], acc[3][r]};
        if (MODE == 2) {
            const f32x4 h = ld4(a.hpre + (size_t)n * SE + so + col);
            q = q * act_d4<ACT>(h);
            st4(a.out0 + (size_t)n * SE + so + col, q);
        } else {
#pragma unroll
            for (int cp = 0; cp < 4; ++cp)
                if (col + cp >= mc) q[cp] = 0.f;
            st4(a.out0 + (size_t)n * M + off + col, q);
        }
    }
}

// the wave-level kernels need aligned float4 rows of W_r / the gradients: every SE group's mid width a multiple of 4
// TfnasCellDesc.route, TFNAS_ROUTE_SE_*: 0 wave-level MFMA kernels where the shapes allow | 1 per-image kernels | 2 LDS-tiled GEMMs:
// the three formulations of the excite FCs, every one compared with the oracle (tests/test_gpu_cell.py::test_variant_against_oracle)
static inline int se_variant(const TfnasCellDesc& d) { return route_se(d); }
static bool se_wave_ok(const TfnasCellDesc& d) {
    if (se_variant(d) != 0) return false;
    for (int g = 0; g < d.G; ++g)
        if (d.g[g].se > 0 && ((d.g[g].mc & 3) || (d.g[g].se & 3) || d.g[g].mc < 4)) return false;
    return true;
}
#define SEW_LAUNCH(KERNEL, MODE_, GRID)                                                                   \
    {                                                                                                    \
        if (d.act == TFNAS_ACT_RELU)                                                                     \
            hipLaunchKernelGGL((KERNEL<MODE_, TFNAS_ACT_RELU>), GRID, dim3(256), 0, s, d, a);            \
        else                                                                                             \
            hipLaunchKernelGGL((KERNEL<MODE_, TFNAS_ACT_SWISH>), GRID, dim3(256), 0, s, d, a);           \
    }

// ============================================================================ host launchers
static int se_count(const TfnasCellDesc& d, int& mcp_max, int& se_max) {
    int t = 0;
    mcp_max = se_max = 0;
    for (int g = 0; g < d.G; ++g)
        if (d.g[g].se > 0) {
            ++t;
            if (d.g[g].mcp > mcp_max) mcp_max = d.g[g].mcp;
            if (d.g[g].se > se_max) se_max = d.g[g].se;
        }
    return t;
}

// K-splits of MODE 0 / 2: >= 4 chunks of 16 channels per split, <= 16 splits, partials must fit `cap` floats
static int se_ksplit(const TfnasCellDesc& d, int mcp_max, size_t cap) {
    int ks = ((mcp_max + 15) / 16) / 4;
    if (ks > 16) ks = 16;
    const size_t per = (size_t)d.N * d.SE;
    if (per && (size_t)ks > cap / per) ks = (int)(cap / per);
    return ks < 1 ? 1 : ks;
}

// the fused per-image kernels need the pooled row + hidden vectors in LDS
static size_t se_fused_lds(int mcp_max, int se_max) {
    return (size_t)(((mcp_max + 3) & ~3) + ((se_max + 3) & ~3) + 256 + SE_CH * (se_max | 1)) * sizeof(float);
}
static bool se_fused_ok(const TfnasCellDesc& d, int mcp_max, int se_max) {
    if (se_variant(d) == 2) return false;
    return se_max <= 256 && se_fused_lds(mcp_max, se_max) <= 60 * 1024;
}

#define SE_FINISH(MODE_)                                                                                 \
    if (a.ksplit > 1) {                                                                                  \
        dim3 fgrid(cdiv(d.N * d.SE, 256));                                                               \
        if (d.act == TFNAS_ACT_RELU)                                                                     \
            hipLaunchKernelGGL((k_se_finish<MODE_, TFNAS_ACT_RELU>), fgrid, dim3(256), 0, s, d, a);      \
        else                                                                                             \
            hipLaunchKernelGGL((k_se_finish<MODE_, TFNAS_ACT_SWISH>), fgrid, dim3(256), 0, s, d, a);     \
    }

#define SE_LAUNCH(MODE_, rows, cols)                                                                     \
    {                                                                                                    \
        dim3 grid(cdiv((rows), 128) * (((MODE_) == 0 || (MODE_) == 2) ? a.ksplit : 1), cdiv((cols), 64), ng); \
        if (d.act == TFNAS_ACT_RELU)                                                                     \
            hipLaunchKernelGGL((k_se_gemm<MODE_, TFNAS_ACT_RELU>), grid, dim3(256), 0, s, d, a);         \
        else                                                                                             \
            hipLaunchKernelGGL((k_se_gemm<MODE_, TFNAS_ACT_SWISH>), grid, dim3(256), 0, s, d, a);        \
    }

int launch_se_fc_fwd(const TfnasCellDesc& d, const float* pooled, float* hpre, float* gate, float* scratch,
                     size_t scratch_floats, hipStream_t s) {
    ProfScope _prof(TK_SE_FC_FWD, s);
    int mcp_max, se_max;
    const int ng = se_count(d, mcp_max, se_max);
    if (!ng) return 0;
    if (se_wave_ok(d)) {
        SeArgs a = {pooled, gate, hpre, nullptr, nullptr, hpre, nullptr, 1};
        SEW_LAUNCH(k_se_nt, 0, dim3(cdiv(d.N, 16), cdiv(se_max, 16), ng))
        a.out0 = gate;
        SEW_LAUNCH(k_se_nt, 1, dim3(cdiv(d.N, 16), cdiv(mcp_max, 64), ng))
        return (int)hipGetLastError();
    }
    if (se_fused_ok(d, mcp_max, se_max)) {
        const size_t shm = se_fused_lds(mcp_max, se_max);
        if (d.act == TFNAS_ACT_RELU)
            hipLaunchKernelGGL((k_se_fused_fwd<TFNAS_ACT_RELU>), dim3(d.N, ng), dim3(256), shm, s, d, pooled, hpre, gate);
        else
            hipLaunchKernelGGL((k_se_fused_fwd<TFNAS_ACT_SWISH>), dim3(d.N, ng), dim3(256), shm, s, d, pooled, hpre, gate);
        return (int)hipGetLastError();
    }
    SeArgs a = {pooled, gate, hpre, nullptr, nullptr, hpre, scratch, se_ksplit(d, mcp_max, scratch ? scratch_floats : 0)};
    SE_LAUNCH(0, d.N, se_max)
    SE_FINISH(0)
    a.out0 = gate;
    SE_LAUNCH(1, d.N, mcp_max)
    return (int)hipGetLastError();
}

int launch_se_fc_bwd(const TfnasCellDesc& d, const float* dgate, const float* gate, const float* hpre,
                     float* dgl, float* dhpre, float* dpooled, float* scratch, size_t scratch_floats, hipStream_t s) {
    ProfScope _prof(TK_SE_FC_BWD, s);
    (void)dgl;
    int mcp_max, se_max;
    const int ng = se_count(d, mcp_max, se_max);
    if (!ng) return 0;
    if (se_wave_ok(d)) {
        SeArgs a = {nullptr, gate, hpre, dgate, dhpre, dhpre, nullptr, 1};
        SEW_LAUNCH(k_se_nn, 2, dim3(cdiv(d.N, 16), cdiv(se_max, 64), ng))
        a.out0 = dpooled;
        SEW_LAUNCH(k_se_nn, 3, dim3(cdiv(d.N, 16), cdiv(mcp_max, 256), ng))
        return (int)hipGetLastError();
    }
    if (se_fused_ok(d, mcp_max, se_max)) {
        const size_t shm = se_fused_lds(mcp_max, se_max);
        if (d.act == TFNAS_ACT_RELU)
            hipLaunchKernelGGL((k_se_fused_bwd<TFNAS_ACT_RELU>), dim3(d.N, ng), dim3(256), shm, s, d, dgate, gate, hpre,
                               dhpre, dpooled);
        else
            hipLaunchKernelGGL((k_se_fused_bwd<TFNAS_ACT_SWISH>), dim3(d.N, ng), dim3(256), shm, s, d, dgate, gate, hpre,
                               dhpre, dpooled);
        return (int)hipGetLastError();
    }
    SeArgs a = {nullptr, gate, hpre, dgate, dhpre, dhpre, scratch, se_ksplit(d, mcp_max, scratch ? scratch_floats : 0)};
    SE_LAUNCH(2, d.N, se_max)
    SE_FINISH(2)
    a.out0 = dpooled;
    SE_LAUNCH(3, d.N, mcp_max)
    return (int)hipGetLastError();
}

int launch_se_wgrad(const TfnasCellDesc& d, const float* dgate, const float* gate, const float* dhpre,
                    const float* hpre, const float* pooled, hipStream_t s) {
    ProfScope _prof(TK_SE_WGRAD, s);
    int mcp_max, se_max;
    const int ng = se_count(d, mcp_max, se_max);
    if (!ng) return 0;
    SeArgs a = {pooled, gate, hpre, dgate, dhpre, nullptr, nullptr, 1};
    // (a wave-level TN formulation of these two, K = batch = 128 images, was built and measured: 51 instead of 36 us per cell
    //  for the three launches -- ~100 waves walking 32 dependent K-steps each; the LDS-tiled GEMMs stay)
    SE_LAUNCH(4, mcp_max, se_max)
    SE_LAUNCH(5, mcp_max, se_max)
    hipLaunchKernelGGL(k_se_bias_grad, dim3(cdiv(mcp_max + se_max, 64), ng), dim3(256), 0, s, d, a);
    return (int)hipGetLastError();
}
