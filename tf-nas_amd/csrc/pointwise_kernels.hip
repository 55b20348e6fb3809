// Bandwidth-bound kernels of the MixedOP cell that are not convolutions:
//   squeeze-excite (pool, two tiny FCs, their backward)      models/layers.py:509-526, forward :548-550
//   softmax(alpha)-weighted mixing of the candidates + BN3    models/model_search.py:89, layers.py:528-534
//   BatchNorm backward statistics passes (autograd of BatchNorm2d(affine=False) with batch statistics)
#include "tfnas_dev.h"
#include "kernels.h"
#include "prof.h"

// sum v over the threads of the block that share `col` (rows pr = 0..RP-1); result valid where pr == 0
// (tree reduction over pr with all threads; a serial sum by the pr==0 threads cost ~16 us per workgroup)
__device__ __forceinline__ f32x4 reduce_rows(f32x4 v, f32x4* buf, int pr, int col, int RP, int TQ, bool active) {
    __syncthreads();
    if (active) buf[pr * TQ + col] = v;
    __syncthreads();
    int s = 1;
    while (s * 2 < RP) s *= 2;
    for (; s > 0; s >>= 1) {
        if (active && pr < s && pr + s < RP) buf[pr * TQ + col] += buf[(pr + s) * TQ + col];
        __syncthreads();
    }
    return (active && pr == 0) ? buf[col] : zero4();
}

// locate the `idx`-th chunk of CH channels among SE groups (se_only) or all groups
__device__ __forceinline__ bool chunk_locate(const TfnasCellDesc& d, int cy, int CH, bool se_only, int& g, int& c0) {
    for (g = 0; g < d.G; ++g) {
        if (se_only && d.g[g].se == 0) continue;
        const int t = (d.g[g].mcp + CH - 1) / CH;
        if (cy < t) {
            c0 = cy * CH;
            return true;
        }
        cy -= t;
    }
    return false;
}
static int chunk_count(const TfnasCellDesc& d, int CH, bool se_only) {
    int t = 0;
    for (int g = 0; g < d.G; ++g)
        if (!se_only || d.g[g].se > 0) t += cdiv(d.g[g].mcp, CH);
    return t;
}
// ============================================================================ SE squeeze (global average pool)
#ifndef TFNAS_POOL_ROWS
#define TFNAS_POOL_ROWS 4
#endif
constexpr int PU = TFNAS_POOL_ROWS;      // pixel rows in flight per thread of the pooling kernels
// MODE 0: pooled[n][c] = mean_hw act(BN2(D))            (forward)
// MODE 1: dgate [n][c] = sum_hw  dZ * act(BN2(D))       (backward of the gate multiply)
template <int ACT, int MODE>
__global__ __launch_bounds__(256) void k_se_pool(TfnasCellDesc d, const float* __restrict__ D,
                                                 const double* __restrict__ stats2, const float* __restrict__ dZ,
                                                 float* __restrict__ outp) {
    __shared__ f32x4 buf[256];
    int g, c0;
    if (!chunk_locate(d, blockIdx.y, 64, true, g, c0)) return;
    const int mc = d.g[g].mc, mcp = d.g[g].mcp, off = d.g[g].off;
    const int HW = d.Ho * d.Wo, M = d.M, n = blockIdx.x;
    // threads = (channel quad) x (row lane): a chunk narrower than 64 channels (the 32-channel stem, the last chunk of a
    // 96- / 144- / 240-wide group) gives its spare quad slots to more row lanes instead of leaving half the block idle
    const int nq = (min(64, mcp - c0) + 3) >> 2, cqs = nq <= 4 ? 2 : (nq <= 8 ? 3 : 4), CQN = 1 << cqs, RLN = 256 >> cqs;
    const int tid = threadIdx.x, cq = tid & (CQN - 1), rl = tid >> cqs;
    const int ch = c0 + 4 * cq;
    const bool active = ch < mcp;
    float2 c2[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
        c2[j] = (active && ch + j < mc) ? bn_consts(stats2 + 2 * (size_t)(off + ch + j), 1.0 / ((double)d.N * HW), d.eps)
                                        : make_float2(0.f, 0.f);
    f32x4 acc = zero4();
    if (active) {
        for (int hw = rl; hw < HW; hw += PU * RLN) {        // PU independent rows in flight per thread
            f32x4 v[PU], z[PU];
#pragma unroll
            for (int u = 0; u < PU; ++u) {
                const int h = hw + RLN * u;
                const size_t a = ((size_t)n * HW + (h < HW ? h : 0)) * M + off + ch;
                v[u] = ldS4_nt(D, a, d.stor);
                z[u] = (MODE == 1) ? ldS4_nt(dZ, a, d.stor) : zero4();
            }
#pragma unroll
            for (int u = 0; u < PU; ++u) {
                if (hw + RLN * u < HW) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[u][j] = act_f<ACT>((v[u][j] - c2[j].x) * c2[j].y);
                    acc += (MODE == 1) ? v[u] * z[u] : v[u];
                }
            }
        }
    }
    acc = reduce_rows(acc, buf, rl, cq, RLN, CQN, active);
    if (active && rl == 0) {
        if (MODE == 0) acc *= splat4(1.f / (float)HW);
        st4(outp + (size_t)n * M + off + ch, acc);
    }
}

// ============================================================================ mixing epilogue (forward)
// out[p][o] = sum_g wmix[g] * BN3(Pr[g])[p][o]  (+ (sum_g wmix[g]) * x[p][o] for residual cells)
__global__ __launch_bounds__(256) void k_mix_fwd(TfnasCellDesc d, const float* __restrict__ Pr,
                                                 const double* __restrict__ stats3, const float* __restrict__ wmix,
                                                 const float* __restrict__ x, float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float2* tab = reinterpret_cast<float2*>(lds);   // [G*oc] (mean3, rstd3*w)
    const int oc = d.oc, G = d.G, OQ = oc >> 2;
    const int Po = d.N * d.Ho * d.Wo;
    float sumw = 0.f;
    for (int g = 0; g < G; ++g) sumw += wmix ? wmix[g] : 1.f;
    for (int i = threadIdx.x; i < G * oc; i += 256) {
        const float2 c = bn_consts(stats3 + 2 * (size_t)i, 1.0 / (double)Po, d.eps);
        tab[i] = make_float2(c.x, c.y * (wmix ? wmix[i / oc] : 1.f));
    }
    __syncthreads();
    const size_t total = (size_t)Po * OQ;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        const size_t p = idx / OQ;
        const int o = (int)(idx % OQ) * 4;
        f32x4 v = zero4();
        for (int g = 0; g < G; ++g) {
            const f32x4 pr = ld4(Pr + ((size_t)g * Po + p) * oc + o);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float2 t = tab[g * oc + o + j];
                v[j] += (pr[j] - t.x) * t.y;
            }
        }
        if (d.has_res) v += splat4(sumw) * ld4(x + p * d.ic + o);
        st4(out + p * oc + o, v);
    }
}

// ============================================================================ mixing epilogue (backward sums)
// red3[g][o] = ( S1 = sum_p dout[p][o] ,  S2 = sum_p dout[p][o] * phat_g[p][o] )
__global__ __launch_bounds__(256) void k_mix_bwd_stats(TfnasCellDesc d, const float* __restrict__ dout,
                                                       const float* __restrict__ Pr,
                                                       const double* __restrict__ stats3,
                                                       const float* __restrict__ x, float* __restrict__ part,
                                                       int rows_per_block) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int oc = d.oc, TQ = oc >> 2, RP = 256 / TQ;
    const int Po = d.N * d.Ho * d.Wo;
    const int G = d.G, Gall = d.G;
    float2* tab = reinterpret_cast<float2*>(lds);                    // [G*oc] (mean3, rstd3)
    f32x4* buf = reinterpret_cast<f32x4*>(lds + 2 * Gall * oc);      // [256]
    for (int i = threadIdx.x; i < G * oc; i += 256) tab[i] = bn_consts(stats3 + 2 * (size_t)i, 1.0 / (double)Po, d.eps);
    __syncthreads();
    const int tid = threadIdx.x, oq = tid % TQ, pr = tid / TQ, o = 4 * oq;
    const bool active = pr < RP;
    const int p0 = blockIdx.x * rows_per_block, p1 = min(Po, p0 + rows_per_block);
    f32x4 s1 = zero4(), sx = zero4(), s2[TFNAS_MAX_GROUPS];
#pragma unroll
    for (int g = 0; g < TFNAS_MAX_GROUPS; ++g) s2[g] = zero4();
    if (active) {
        for (int p = p0 + pr; p < p1; p += RP) {
            const f32x4 dv = ld4(dout + (size_t)p * oc + o);
            s1 += dv;
            if (d.has_res) sx += dv * ld4(x + (size_t)p * d.ic + o);   // residual cells: ic == oc
#pragma unroll
            for (int g = 0; g < TFNAS_MAX_GROUPS; ++g) {
                if (g < G) {
                    const f32x4 pv = ld4(Pr + ((size_t)g * Po + p) * oc + o);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float2 t = tab[g * oc + o + j];
                        s2[g][j] += dv[j] * ((pv[j] - t.x) * t.y);
                    }
                }
            }
        }
    }
    // this workgroup's row of the partials matrix: [G*oc][2] (S1,S2) followed by [oc] partial <dout,x> sums
    float* prow = part + (size_t)blockIdx.x * (2 * Gall * oc + oc);
    s1 = reduce_rows(s1, buf, pr, oq, RP, TQ, active);
    sx = reduce_rows(sx, buf, pr, oq, RP, TQ, active);
    if (active && pr == 0) st4(prow + 2 * Gall * oc + o, sx);
#pragma unroll
    for (int g = 0; g < TFNAS_MAX_GROUPS; ++g) {
        if (g < G) {
            const f32x4 r = reduce_rows(s2[g], buf, pr, oq, RP, TQ, active);
            if (active && pr == 0) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    prow[2 * ((size_t)g * oc + o + j) + 0] = s1[j];
                    prow[2 * ((size_t)g * oc + o + j) + 1] = r[j];
                }
            }
        }
    }
}

// dwmix[g] = d loss / d wmix[g] = <dout, phat_g [+ x]> = sum_o S2[g][o] [+ <dout,x>]
// resdot[o] = per-channel <dout,x> sums (residual cells), stored right after red3
__global__ __launch_bounds__(64) void k_mix_dw(TfnasCellDesc d, const double* __restrict__ red3, const double* __restrict__ resdot,
                                               float* __restrict__ dwmix) {
    // one wave per group (blockIdx.x), lanes stride over the output channels, butterfly sum in double (fixed order).  The
    // first version -- ONE thread per group walking up to 640 doubles -- took 15 us on the alpha-step's dependency chain.
    const int g = blockIdx.x, lane = threadIdx.x;
    double s = 0.0;
    for (int o = lane; o < d.oc; o += 64) {
        s += red3[2 * ((size_t)g * d.oc + o) + 1];
        if (d.has_res) s += resdot[o];
    }
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) s += __shfl_xor(s, m, 64);
    if (lane == 0) dwmix[g] = (float)s;
}

// ============================================================================ BN2 backward sums, fused with the SE pool backward
// One pass over (dZ, D) per image and channel chunk produces everything the BN2 backward needs:
//   ddh = (dZ*gate + dpooled/HW) * act'(dhat)   is linear in the per-image scalars gate[n][c], dpooled[n][c], so
//   sum_p ddh        = sum_n ( gate*A1 + dpooled/HW*A2 ),   A1 = sum_hw dZ*act'(dhat),       A2 = sum_hw act'(dhat)
//   sum_p ddh*dhat   = sum_n ( gate*B1 + dpooled/HW*B2 ),   B1 = sum_hw dZ*act'(dhat)*dhat,  B2 = sum_hw act'(dhat)*dhat
// and the same pass yields dgate[n][c] = sum_hw dZ*act(dhat) (what k_se_pool<MODE 1> computed), which the SE FC backward
// needs BEFORE dpooled exists.  k_bn2_finish then folds the [N][M] tables into red2 once dpooled is known.  This replaces
// the separate k_se_pool<bwd> and k_bn2_bwd passes (two reads of dZ and D) and one k_reduce_rows launch.
// pp = [4][N][M] floats (A1 | B1 | A2 | B2; the last two only for SE groups).
template <int ACT>
__global__ __launch_bounds__(256) void k_bn2_pool(TfnasCellDesc d, const float* __restrict__ dZ, const float* __restrict__ D,
                                                  const double* __restrict__ stats2, float* __restrict__ dgate,
                                                  float* __restrict__ pp, int CH) {
    __shared__ f32x4 buf[256];
    int g, c0;
    if (!chunk_locate(d, blockIdx.y, CH, false, g, c0)) return;
    const int mc = d.g[g].mc, mcp = d.g[g].mcp, off = d.g[g].off;
    const bool has_se = d.g[g].se > 0;
    const int HW = d.Ho * d.Wo, M = d.M, n = blockIdx.x;
    const int nq = (min(CH, mcp - c0) + 3) >> 2, cqs = nq <= 4 ? 2 : (nq <= 8 ? 3 : 4), CQN = 1 << cqs, RLN = 256 >> cqs;
    const int tid = threadIdx.x, cq = tid & (CQN - 1), rl = tid >> cqs;       // (see k_se_pool)
    const int ch = c0 + 4 * cq;
    const bool active = ch < mcp;
    float2 c2[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
        c2[j] = (active && ch + j < mc) ? bn_consts(stats2 + 2 * (size_t)(off + ch + j), 1.0 / ((double)d.N * HW), d.eps)
                                        : make_float2(0.f, 0.f);
    f32x4 g0 = zero4(), a1 = zero4(), b1 = zero4(), a2 = zero4(), b2 = zero4();
    if (active) {
        for (int hw = rl; hw < HW; hw += PU * RLN) {        // PU independent rows in flight per thread
            f32x4 v[PU], z[PU];
#pragma unroll
            for (int u = 0; u < PU; ++u) {
                const int h = hw + RLN * u;
                const size_t a = ((size_t)n * HW + (h < HW ? h : 0)) * M + off + ch;
                v[u] = ldS4_nt(D, a, d.stor);
                z[u] = ldS4_nt(dZ, a, d.stor);
            }
#pragma unroll
            for (int u = 0; u < PU; ++u) {
                if (hw + RLN * u < HW) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float dh = (v[u][j] - c2[j].x) * c2[j].y;
                        const float ad = act_d<ACT>(dh), zd = z[u][j] * ad;
                        a1[j] += zd;
                        b1[j] += zd * dh;
                        if (has_se) {
                            g0[j] += z[u][j] * act_f<ACT>(dh);
                            a2[j] += ad;
                            b2[j] += ad * dh;
                        }
                    }
                }
            }
        }
    }
    const size_t NM = (size_t)d.N * M, o = (size_t)n * M + off + ch;
    a1 = reduce_rows(a1, buf, rl, cq, RLN, CQN, active);
    if (active && rl == 0) st4(pp + o, a1);
    b1 = reduce_rows(b1, buf, rl, cq, RLN, CQN, active);
    if (active && rl == 0) st4(pp + NM + o, b1);
    if (has_se) {
        a2 = reduce_rows(a2, buf, rl, cq, RLN, CQN, active);
        if (active && rl == 0) st4(pp + 2 * NM + o, a2);
        b2 = reduce_rows(b2, buf, rl, cq, RLN, CQN, active);
        if (active && rl == 0) st4(pp + 3 * NM + o, b2);
        g0 = reduce_rows(g0, buf, rl, cq, RLN, CQN, active);
        if (active && rl == 0) st4(dgate + o, g0);
    }
}

// Per-image tables from the records the FOLD epilogue of k_project_dgrad wrote (gemm_kernels.hip): image n covers the row tiles
// n*HW/128 .. ((n+1)*HW-1)/128, its slot in tile t is n - (first image of t); summed in tile order, in double.  Writes exactly what
// k_bn2_pool writes: pp = A1 | B1 | A2 | B2 [N][M] and dgate [N][M] (the last three for SE groups only).
__global__ __launch_bounds__(256) void k_bn2_gather(TfnasCellDesc d, const float* __restrict__ rec, float* __restrict__ dgate,
                                                    float* __restrict__ pp) {
    __shared__ double sh[4][FOLD_Q][64];
    int g, c0;
    if (!chunk_locate(d, blockIdx.y, 64, false, g, c0)) return;
    const int mcp = d.g[g].mcp, off = d.g[g].off;
    const bool has_se = d.g[g].se > 0;
    const int HW = d.Ho * d.Wo, M = d.M, n = blockIdx.x, Po = d.N * HW;
    const int tid = threadIdx.x, cl = tid & 63, part = tid >> 6;
    const bool active = c0 + cl < mcp;
    const int t0 = (n * HW) >> 7, t1 = ((n + 1) * HW - 1) >> 7;
    const int nq = has_se ? FOLD_Q : 2;
    double acc[FOLD_Q] = {0, 0, 0, 0, 0};
    if (active) {
        // U tiles' records in flight per round (the 112 x 112 stem has 99 tiles per image: 25 dependent rounds per thread group
        // took 77 us alone on the chip); same summation order as one tile per round (masked tiles add +0.0)
        constexpr int U = 8;
        for (int t = t0 + part; t <= t1; t += 4 * U) {
            float v[U][FOLD_Q];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int tt = min(t + 4 * u, t1);
                const int slot = n - min(tt * 128, Po - 1) / HW;
                const float* r = rec + (((size_t)tt * FOLD_SLOTS + slot) * FOLD_Q) * M + off + c0 + cl;
#pragma unroll
                for (int q = 0; q < FOLD_Q; ++q) v[u][q] = q < nq ? r[(size_t)q * M] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const bool ok = t + 4 * u <= t1;
#pragma unroll
                for (int q = 0; q < FOLD_Q; ++q)
                    if (q < nq) acc[q] += ok ? (double)v[u][q] : 0.0;
            }
        }
    }
#pragma unroll
    for (int q = 0; q < FOLD_Q; ++q) sh[part][q][cl] = acc[q];
    __syncthreads();
    if (part == 0 && active) {
        const size_t NM = (size_t)d.N * M, o = (size_t)n * M + off + c0 + cl;
        double v[FOLD_Q];
#pragma unroll
        for (int q = 0; q < FOLD_Q; ++q) v[q] = ((sh[0][q][cl] + sh[1][q][cl]) + sh[2][q][cl]) + sh[3][q][cl];
        pp[o] = (float)v[0];
        pp[NM + o] = (float)v[1];
        if (has_se) {
            dgate[o] = (float)v[2];
            pp[2 * NM + o] = (float)v[3];
            pp[3 * NM + o] = (float)v[4];
        }
    }
}

// red2[c] = (sum_n gate*A1 + dpooled/HW*A2, sum_n gate*B1 + dpooled/HW*B2) in double (non-SE groups: sum_n A1, sum_n B1)
__global__ __launch_bounds__(256) void k_bn2_finish(TfnasCellDesc d, const float* __restrict__ pp,
                                                    const float* __restrict__ gate, const float* __restrict__ dpooled,
                                                    double* __restrict__ red2) {
    __shared__ double sh[2][16][65];
    int g, c0;
    if (!chunk_locate(d, blockIdx.x, 64, false, g, c0)) return;
    const int mc = d.g[g].mc, mcp = d.g[g].mcp, off = d.g[g].off;
    const bool has_se = d.g[g].se > 0;
    const int M = d.M, N = d.N;
    const int tid = threadIdx.x, cq = tid & 15, rl = tid >> 4;
    const int ch = c0 + 4 * cq;
    const bool active = ch < mcp;
    const size_t NM = (size_t)N * M;
    const float inv_hw = 1.f / (float)(d.Ho * d.Wo);
    double r1[4] = {0, 0, 0, 0}, r2[4] = {0, 0, 0, 0};
    if (active) {
        for (int n = rl; n < N; n += 16) {
            const size_t o = (size_t)n * M + off + ch;
            f32x4 x1 = ld4(pp + o), y1 = ld4(pp + NM + o);
            if (has_se) {
                const f32x4 gt = ld4(gate + o), dp = ld4(dpooled + o) * splat4(inv_hw);
                x1 = gt * x1 + dp * ld4(pp + 2 * NM + o);
                y1 = gt * y1 + dp * ld4(pp + 3 * NM + o);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                r1[j] += (double)x1[j];
                r2[j] += (double)y1[j];
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        sh[0][rl][4 * cq + j] = r1[j];
        sh[1][rl][4 * cq + j] = r2[j];
    }
    __syncthreads();
    if (tid < 128) {
        const int which = tid >> 6, cl = tid & 63;
        if (c0 + cl < mc) {
            double t = 0.0;
#pragma unroll
            for (int r = 0; r < 16; ++r) t += sh[which][r][cl];
            red2[2 * (size_t)(off + c0 + cl) + which] = t;
        } else if (c0 + cl < mcp) {
            red2[2 * (size_t)(off + c0 + cl) + which] = 0.0;
        }
    }
}

// ============================================================================ BN2 backward, pass 1
// ddh = (dZ*gate + dpooled/HW) * act'(dhat) ;  red2[c] = (sum ddh, sum ddh*dhat)   (sums only: the consumers
// k_dw_bwd_data / k_dw_wgrad recompute ddh from dZ in their tile loaders, so nothing is written back here)
template <int ACT>
__global__ __launch_bounds__(256) void k_bn2_bwd(TfnasCellDesc d, const float* __restrict__ dZ, const float* __restrict__ D,
                                                 const double* __restrict__ stats2, const float* __restrict__ gate,
                                                 const float* __restrict__ dpooled, float* __restrict__ part,
                                                 int rows_per_block) {
    __shared__ f32x4 buf[256];
    int g, c0;
    if (!chunk_locate(d, blockIdx.y, 64, false, g, c0)) return;
    const int mc = d.g[g].mc, mcp = d.g[g].mcp, off = d.g[g].off;
    const bool has_se = d.g[g].se > 0;
    const int HW = d.Ho * d.Wo, Po = d.N * HW, M = d.M;
    const int tid = threadIdx.x, cq = tid & 15, rl = tid >> 4;
    const int ch = c0 + 4 * cq;
    const bool active = ch < mcp;
    float2 c2[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
        c2[j] = (active && ch + j < mc) ? bn_consts(stats2 + 2 * (size_t)(off + ch + j), 1.0 / (double)Po, d.eps)
                                        : make_float2(0.f, 0.f);
    const float inv_hw = 1.f / (float)HW;
    const int p0 = blockIdx.x * rows_per_block, p1 = min(Po, p0 + rows_per_block);
    f32x4 r1 = zero4(), r2 = zero4();
    if (active) {
        for (int pb = p0 + rl; pb < p1; pb += 64) {          // 4 independent rows in flight per thread
            f32x4 da[4], dv[4], gt[4], dp[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int p = pb + 16 * u < p1 ? pb + 16 * u : pb;
                const size_t a = (size_t)p * M + off + ch;
                da[u] = ldS4_nt(dZ, a, d.stor);
                dv[u] = ldS4_nt(D, a, d.stor);
                if (has_se) {
                    const size_t b = (size_t)(p / HW) * M + off + ch;
                    gt[u] = ld4(gate + b);
                    dp[u] = ld4(dpooled + b);
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int p = pb + 16 * u;
                if (p < p1) {
                    if (has_se) da[u] = da[u] * gt[u] + dp[u] * splat4(inv_hw);
                    f32x4 ddh;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float dh = (dv[u][j] - c2[j].x) * c2[j].y;
                        ddh[j] = da[u][j] * act_d<ACT>(dh);
                        r1[j] += ddh[j];
                        r2[j] += ddh[j] * dh;
                    }
                }
            }
        }
    }
    r1 = reduce_rows(r1, buf, rl, cq, 16, 16, active);
    r2 = reduce_rows(r2, buf, rl, cq, 16, 16, active);
    if (active && rl == 0) {
        float* prow = part + (size_t)blockIdx.x * 2 * M + 2 * (size_t)(off + ch);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            prow[2 * j + 0] = r1[j];
            prow[2 * j + 1] = r2[j];
        }
    }
}

// cb1[c] = (mean1, rstd1, T1/P, T2/P) for the expand dgrad / wgrad operand loaders
__global__ void k_bn1_consts(TfnasCellDesc d, const double* __restrict__ stats1, const double* __restrict__ red1,
                             float* __restrict__ cb1) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= d.M) return;
    const double inv = 1.0 / ((double)d.N * d.H * d.W);
    const float2 m = bn_consts(stats1 + 2 * (size_t)c, inv, d.eps);
    f32x4 t;
    t.x = m.x;
    t.y = m.y;
    t.z = (float)(red1[2 * (size_t)c + 0] * inv);
    t.w = (float)(red1[2 * (size_t)c + 1] * inv);
    // pad channels (never accumulated) have zero sums -> mean 0; force rstd 0 so they stay exactly 0
    bool is_pad = true;
    for (int g = 0; g < d.G; ++g)
        if (c >= d.g[g].off && c < d.g[g].off + d.g[g].mc) is_pad = false;
    if (is_pad) t = zero4();
    reinterpret_cast<f32x4*>(cb1)[c] = t;
}

// ============================================================================ network head (TFNAS_MODE_HEAD)
// pooled[n][c] = mean_hw act(BN1(E[n][hw][c]))     (feature_mix BN + swish + AdaptiveAvgPool2d(1))
template <int ACT>
__global__ __launch_bounds__(256) void k_head_pool(TfnasCellDesc d, const float* __restrict__ E,
                                                   const double* __restrict__ stats1, float* __restrict__ pooled) {
    __shared__ f32x4 buf[256];
    const int mc = d.g[0].mc, mcp = d.g[0].mcp, M = d.M, HW = d.H * d.W, n = blockIdx.x;
    const int tid = threadIdx.x, cq = tid & 15, rl = tid >> 4;
    const int ch = blockIdx.y * 64 + 4 * cq;
    const bool active = ch < mcp;
    float2 c1[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
        c1[j] = (active && ch + j < mc) ? bn_consts(stats1 + 2 * (size_t)(ch + j), 1.0 / ((double)d.N * HW), d.eps)
                                        : make_float2(0.f, 0.f);
    f32x4 acc = zero4();
    if (active)
        for (int hw = rl; hw < HW; hw += 16) {
            f32x4 v = ld4(E + ((size_t)n * HW + hw) * M + ch);
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = act_f<ACT>((v[j] - c1[j].x) * c1[j].y);
            acc += v;
        }
    acc = reduce_rows(acc, buf, rl, cq, 16, 16, active);
    if (active && rl == 0) st4(pooled + (size_t)n * mc + ch, acc * splat4(1.f / (float)HW));
}

// dEh[p][c] = dpooled[n][c]/HW * act'(ehat) ; per-workgroup partial BN1-backward sums (T1, T2) -> part
template <int ACT>
__global__ __launch_bounds__(256) void k_head_bwd(TfnasCellDesc d, const float* __restrict__ E,
                                                  const double* __restrict__ stats1, const float* __restrict__ dpooled,
                                                  float* __restrict__ dEh, float* __restrict__ part) {
    __shared__ f32x4 buf[256];
    const int mc = d.g[0].mc, mcp = d.g[0].mcp, M = d.M, HW = d.H * d.W, n = blockIdx.x;
    const int tid = threadIdx.x, cq = tid & 15, rl = tid >> 4;
    const int ch = blockIdx.y * 64 + 4 * cq;
    const bool active = ch < mcp;
    float2 c1[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
        c1[j] = (active && ch + j < mc) ? bn_consts(stats1 + 2 * (size_t)(ch + j), 1.0 / ((double)d.N * HW), d.eps)
                                        : make_float2(0.f, 0.f);
    f32x4 t1 = zero4(), t2 = zero4();
    if (active) {
        const f32x4 dp = ld4(dpooled + (size_t)n * mc + ch) * splat4(1.f / (float)HW);
        for (int hw = rl; hw < HW; hw += 16) {
            const size_t a = ((size_t)n * HW + hw) * M + ch;
            const f32x4 e = ld4(E + a);
            f32x4 deh;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float eh = (e[j] - c1[j].x) * c1[j].y;
                deh[j] = dp[j] * act_d<ACT>(eh);
                t1[j] += deh[j];
                t2[j] += deh[j] * eh;
            }
            st4(dEh + a, deh);
        }
    }
    t1 = reduce_rows(t1, buf, rl, cq, 16, 16, active);
    t2 = reduce_rows(t2, buf, rl, cq, 16, 16, active);
    if (active && rl == 0) {
        float* prow = part + (size_t)blockIdx.x * 2 * M + 2 * (size_t)ch;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            prow[2 * j + 0] = t1[j];
            prow[2 * j + 1] = t2[j];
        }
    }
}

// ============================================================================ partials -> totals
// out[c] = sum_{b < nb} part[b*stride + c], summed in double in a fixed order (bit-reproducible).
// k_reduce_rows<CL>: CL columns x (256/CL) row-lanes per workgroup, 16 loads in flight per thread -- for the statistics
// partials (many rows, few columns: CL = 4 folds 1024 rows in one round trip and doubles the workgroup count of the
// narrow sampled-mode launches).  k_reduce_rows_wide: 128 columns (float4 per thread) x 8 row-lanes -- for the split-K
// weight-gradient partials (<= 128 rows of 10^4..10^5 columns), where the scalar version read 32-byte pieces of each row.
template <int CL>
__global__ __launch_bounds__(256) void k_reduce_rows(const float* __restrict__ part, int nb, int ncols, size_t stride,
                                                     double* __restrict__ out_d, float* __restrict__ out_f,
                                                     size_t in_stride, size_t out_stride) {
    constexpr int RL = 256 / CL;
    __shared__ double buf[RL][CL + 1];
    part += blockIdx.y * in_stride;
    if (out_d) out_d += blockIdx.y * out_stride;
    if (out_f) out_f += blockIdx.y * out_stride;
    const int tid = threadIdx.x, cl = tid % CL, rl = tid / CL;
    const int c = blockIdx.x * CL + cl;
    double s = 0.0;
    if (c < ncols) {
        for (int b = rl; b < nb; b += 16 * RL) {
            float v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) v[u] = (b + RL * u < nb) ? part[(size_t)(b + RL * u) * stride + c] : 0.f;
#pragma unroll
            for (int u = 0; u < 16; u += 4) s += ((double)v[u] + (double)v[u + 1]) + ((double)v[u + 2] + (double)v[u + 3]);
        }
    }
    buf[rl][cl] = s;
    __syncthreads();
    if (rl == 0 && c < ncols) {
        double t = 0.0;
#pragma unroll
        for (int r = 0; r < RL; ++r) t += buf[r][cl];
        if (out_d) out_d[c] = t;
        if (out_f) out_f[c] = (float)t;
    }
}

// tall tables (statistics partials: 200..1024 rows of 10^2..10^3 columns) with 16-byte loads: QN column quads x (256/QN) row
// lanes per workgroup, 8 float4 loads in flight per thread -- the scalar k_reduce_rows<4> moved 16-byte pieces per row
template <int QN>
__global__ __launch_bounds__(256) void k_reduce_rows_q(const float* __restrict__ part, int nb, int ncols, size_t stride,
                                                       double* __restrict__ out_d, float* __restrict__ out_f,
                                                       size_t in_stride, size_t out_stride) {
    constexpr int RL = 256 / QN;
    __shared__ double buf[RL][4 * QN + 2];
    part += blockIdx.y * in_stride;
    if (out_d) out_d += blockIdx.y * out_stride;
    if (out_f) out_f += blockIdx.y * out_stride;
    const int tid = threadIdx.x, cq = tid % QN, rl = tid / QN;
    const int c = (blockIdx.x * QN + cq) * 4;
    double s[4] = {0.0, 0.0, 0.0, 0.0};
    if (c < ncols) {
        for (int b = rl; b < nb; b += 8 * RL) {
            f32x4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = (b + RL * u < nb) ? ld4(part + (size_t)(b + RL * u) * stride + c) : zero4();
#pragma unroll
            for (int j = 0; j < 4; ++j)
                s[j] += (((double)v[0][j] + (double)v[1][j]) + ((double)v[2][j] + (double)v[3][j])) +
                        (((double)v[4][j] + (double)v[5][j]) + ((double)v[6][j] + (double)v[7][j]));
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) buf[rl][4 * cq + j] = s[j];
    __syncthreads();
    if (tid < 4 * QN && blockIdx.x * 4 * QN + tid < ncols) {
        double t = 0.0;
#pragma unroll
        for (int r = 0; r < RL; ++r) t += buf[r][tid];
        if (out_d) out_d[blockIdx.x * 4 * QN + tid] = t;
        if (out_f) out_f[blockIdx.x * 4 * QN + tid] = (float)t;
    }
}

__global__ __launch_bounds__(256) void k_reduce_rows_wide(const float* __restrict__ part, int nb, int ncols, size_t stride,
                                                          double* __restrict__ out_d, float* __restrict__ out_f,
                                                          size_t in_stride, size_t out_stride) {
    __shared__ double buf[8][128 + 4];
    part += blockIdx.y * in_stride;
    if (out_d) out_d += blockIdx.y * out_stride;
    if (out_f) out_f += blockIdx.y * out_stride;
    const int tid = threadIdx.x, cq = tid & 31, rl = tid >> 5;
    const int c = blockIdx.x * 128 + 4 * cq;                  // ncols % 4 == 0: a quad is inside or outside as a whole
    double s[4] = {0.0, 0.0, 0.0, 0.0};
    if (c < ncols) {
        for (int b = rl; b < nb; b += 64) {
            f32x4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = (b + 8 * u < nb) ? ld4(part + (size_t)(b + 8 * u) * stride + c) : zero4();
#pragma unroll
            for (int j = 0; j < 4; ++j)
                s[j] += (((double)v[0][j] + (double)v[1][j]) + ((double)v[2][j] + (double)v[3][j])) +
                        (((double)v[4][j] + (double)v[5][j]) + ((double)v[6][j] + (double)v[7][j]));
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) buf[rl][4 * cq + j] = s[j];
    __syncthreads();
    if (tid < 128 && blockIdx.x * 128 + tid < ncols) {
        double t = 0.0;
#pragma unroll
        for (int r = 0; r < 8; ++r) t += buf[r][tid];
        if (out_d) out_d[blockIdx.x * 128 + tid] = t;
        if (out_f) out_f[blockIdx.x * 128 + tid] = (float)t;
    }
}

// BN1-backward partials -> red1 AND the cb1 table in one launch (what k_reduce_rows + k_bn1_consts did in two):
// columns (2c, 2c+1) = (T1, T2) of channel c;  cb1[c] = (mean1, rstd1, T1/P, T2/P), zeros for pad channels.
__global__ __launch_bounds__(256) void k_reduce_bn1(TfnasCellDesc d, const float* __restrict__ part, int nb,
                                                    const double* __restrict__ stats1, double* __restrict__ red1,
                                                    float* __restrict__ cb1) {
    // 4 column quads (16 columns = 8 channels) x 64 row lanes, float4 loads (2 * M is a multiple of 64 floats, rows are 16-byte
    // aligned): the scalar version moved 32-byte pieces per row
    constexpr int QN = 4, CL = 4 * QN, RL = 256 / QN;
    __shared__ double buf[RL][CL + 2];
    __shared__ double tot[CL];
    const int tid = threadIdx.x, cq = tid % QN, rl = tid / QN;
    const int ncols = 2 * d.M, c = (blockIdx.x * QN + cq) * 4;
    const size_t stride = (size_t)ncols;
    double s[4] = {0.0, 0.0, 0.0, 0.0};
    if (c < ncols) {
        for (int b = rl; b < nb; b += 8 * RL) {
            f32x4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = (b + RL * u < nb) ? ld4(part + (size_t)(b + RL * u) * stride + c) : zero4();
#pragma unroll
            for (int j = 0; j < 4; ++j)
                s[j] += (((double)v[0][j] + (double)v[1][j]) + ((double)v[2][j] + (double)v[3][j])) +
                        (((double)v[4][j] + (double)v[5][j]) + ((double)v[6][j] + (double)v[7][j]));
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) buf[rl][4 * cq + j] = s[j];
    __syncthreads();
    if (tid < CL) {
        double t = 0.0;
#pragma unroll
        for (int r = 0; r < RL; ++r) t += buf[r][tid];
        tot[tid] = t;
        if (blockIdx.x * CL + tid < ncols) red1[blockIdx.x * CL + tid] = t;
    }
    __syncthreads();
    if (tid < CL / 2) {
        const int ch = blockIdx.x * (CL / 2) + tid;
        if (ch < d.M) {
            const double inv = 1.0 / ((double)d.N * d.H * d.W);
            const float2 m = bn_consts(stats1 + 2 * (size_t)ch, inv, d.eps);
            f32x4 t;
            t.x = m.x;
            t.y = m.y;
            t.z = (float)(tot[2 * tid] * inv);
            t.w = (float)(tot[2 * tid + 1] * inv);
            bool is_pad = true;
            for (int g = 0; g < d.G; ++g)
                if (ch >= d.g[g].off && ch < d.g[g].off + d.g[g].mc) is_pad = false;
            if (is_pad) t = zero4();
            reinterpret_cast<f32x4*>(cb1)[ch] = t;
        }
    }
}

int launch_reduce_bn1(const TfnasCellDesc& d, const float* part, int nb, const double* stats1, double* red1, float* cb1,
                      hipStream_t s) {
    ProfScope _prof(TK_REDUCE_ROWS, s);
    hipLaunchKernelGGL(k_reduce_bn1, dim3(cdiv(2 * d.M, 16)), dim3(256), 0, s, d, part, nb, stats1, red1, cb1);
    return (int)hipGetLastError();
}

int launch_reduce_rows(const float* part, int nb, int ncols, size_t stride, double* out_d, float* out_f,
                       hipStream_t s, int nbatch, size_t in_stride, size_t out_stride) {
    ProfScope _prof(TK_REDUCE_ROWS, s);
    const bool al4 = (ncols & 3) == 0 && (stride & 3) == 0 && ((uintptr_t)part & 15) == 0 && (in_stride & 3) == 0;
    const unsigned nby = nbatch > 1 ? (unsigned)nbatch : 1u;
    if (al4 && nb <= 128 && ncols >= 1024)
        hipLaunchKernelGGL(k_reduce_rows_wide, dim3(cdiv(ncols, 128), nby), dim3(256), 0, s, part, nb, ncols, stride, out_d,
                           out_f, in_stride, out_stride);
    else if (al4 && nb > 64)
        hipLaunchKernelGGL(k_reduce_rows_q<4>, dim3(cdiv(ncols, 16), nby), dim3(256), 0, s, part, nb, ncols, stride, out_d, out_f,
                           in_stride, out_stride);
    else if (ncols <= 2048 && nb > 256)
        hipLaunchKernelGGL(k_reduce_rows<4>, dim3(cdiv(ncols, 4), nby), dim3(256), 0, s, part, nb, ncols, stride, out_d, out_f,
                           in_stride, out_stride);
    else
        hipLaunchKernelGGL(k_reduce_rows<8>, dim3(cdiv(ncols, 8), nby), dim3(256), 0, s, part, nb, ncols, stride, out_d, out_f,
                           in_stride, out_stride);
    return (int)hipGetLastError();
}

// ============================================================================ host launchers
#define ACT_DISPATCH(act, ...)                                                          \
    if ((act) == TFNAS_ACT_RELU) { constexpr int ACT = TFNAS_ACT_RELU; __VA_ARGS__; }   \
    else { constexpr int ACT = TFNAS_ACT_SWISH; __VA_ARGS__; }

int launch_se_pool(const TfnasCellDesc& d, const float* D, const double* stats2, float* pooled, hipStream_t s) {
    ProfScope _prof(TK_SE_POOL, s);
    const int chunks = chunk_count(d, 64, true);
    if (!chunks) return 0;
    dim3 grid(d.N, chunks);
    ACT_DISPATCH(d.act, {
        hipLaunchKernelGGL((k_se_pool<ACT, 0>), grid, dim3(256), 0, s, d, D, stats2, (const float*)nullptr, pooled);
    })
    return (int)hipGetLastError();
}

int launch_se_bwd_reduce(const TfnasCellDesc& d, const float* dZ, const float* D, const double* stats2,
                         float* dgate, hipStream_t s) {
    ProfScope _prof(TK_SE_BWD_REDUCE, s);
    const int chunks = chunk_count(d, 64, true);
    if (!chunks) return 0;
    dim3 grid(d.N, chunks);
    ACT_DISPATCH(d.act, {
        hipLaunchKernelGGL((k_se_pool<ACT, 1>), grid, dim3(256), 0, s, d, D, stats2, dZ, dgate);
    })
    return (int)hipGetLastError();
}

int launch_mix_fwd(const TfnasCellDesc& d, const float* Pr, const double* stats3, const float* wmix,
                   const float* x, float* out, hipStream_t s) {
    ProfScope _prof(TK_MIX_FWD, s);
    const size_t total = (size_t)d.N * d.Ho * d.Wo * (d.oc / 4);
    size_t blocks = cdiv64(total, 256 * 4);
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    const size_t shm = (size_t)2 * d.G * d.oc * sizeof(float);
    hipLaunchKernelGGL(k_mix_fwd, dim3((unsigned)blocks), dim3(256), shm, s, d, Pr, stats3, wmix, x, out);
    return (int)hipGetLastError();
}

int launch_mix_bwd_stats(const TfnasCellDesc& d, const float* dout, const float* Pr, const double* stats3,
                         const float* x, double* red3, float* part, hipStream_t s) {
    ProfScope _prof(TK_MIX_BWD_STATS, s);
    const int Po = d.N * d.Ho * d.Wo;
    const int ncols = 2 * d.G * d.oc + d.oc;       // (S1,S2) pairs + per-channel <dout,x>; reduced into red3|resdot
    size_t nblk = 1024;
    if (nblk > TFNAS_PART_FLOATS / (size_t)ncols) nblk = TFNAS_PART_FLOATS / (size_t)ncols;
    int rpb = cdiv(Po, (int)nblk);
    const int RP = 256 / (d.oc / 4);
    if (rpb < 8 * RP) rpb = 8 * RP;
    const size_t shm = (size_t)(2 * d.G * d.oc + 4 * 256) * sizeof(float);
    const int gx = cdiv(Po, rpb);
    hipLaunchKernelGGL(k_mix_bwd_stats, dim3(gx), dim3(256), shm, s, d, dout, Pr, stats3, x, part, rpb);
    _prof.stop();
    return launch_reduce_rows(part, gx, ncols, (size_t)ncols, red3, nullptr, s);
}

int launch_mix_dw(const TfnasCellDesc& d, const double* red3, const double* resdot, float* dwmix, hipStream_t s) {
    ProfScope _prof(TK_SMALL, s);
    hipLaunchKernelGGL(k_mix_dw, dim3(d.G), dim3(64), 0, s, d, red3, resdot, dwmix);
    return (int)hipGetLastError();
}

int launch_bn2_bwd(const TfnasCellDesc& d, const float* dZ, const float* D, const double* stats2, const float* gate,
                   const float* dpooled, double* red2, float* part, hipStream_t s) {
    ProfScope _prof(TK_BN2_BWD, s);
    const int Po = d.N * d.Ho * d.Wo;
    const int chunks = chunk_count(d, 64, false);
    int want = cdiv(4096, chunks);
    if (want > 1024) want = 1024;                  // partial rows to reduce afterwards
    const size_t cap = TFNAS_PART_FLOATS / (2 * (size_t)d.M);
    if ((size_t)want > cap) want = (int)cap;
    int rpb = cdiv(Po, want < 1 ? 1 : want);
    if (rpb < 64) rpb = 64;
    dim3 grid(cdiv(Po, rpb), chunks);
    ACT_DISPATCH(d.act, {
        hipLaunchKernelGGL((k_bn2_bwd<ACT>), grid, dim3(256), 0, s, d, dZ, D, stats2, gate, dpooled, part, rpb);
    })
    _prof.stop();
    return launch_reduce_rows(part, grid.x, 2 * d.M, 2 * (size_t)d.M, red2, nullptr, s);
}

// fused replacement of launch_se_bwd_reduce + launch_bn2_bwd (see k_bn2_pool); false if the [4][N][M] table does not fit
bool bn2_fused_fits(const TfnasCellDesc& d) { return 4 * (size_t)d.N * d.M <= TFNAS_PART_FLOATS; }

int launch_bn2_pool(const TfnasCellDesc& d, const float* dZ, const float* D, const double* stats2, float* dgate,
                    float* pp, hipStream_t s) {
    ProfScope _prof(TK_SE_BWD_REDUCE, s);
    // 32-channel chunks (one 128-byte line per pixel) where 64-channel ones leave the chip under-filled: the sampled launches
    // of the 112 x 112 ... 28 x 28 cells are N x (1..4) workgroups of 0.2-1.6 MB each
    const int CH = (d.N * chunk_count(d, 64, false) < 1024 && d.Ho * d.Wo >= 196) ? 32 : 64;
    dim3 grid(d.N, chunk_count(d, CH, false));
    ACT_DISPATCH(d.act, { hipLaunchKernelGGL((k_bn2_pool<ACT>), grid, dim3(256), 0, s, d, dZ, D, stats2, dgate, pp, CH); })
    return (int)hipGetLastError();
}

int launch_bn2_gather(const TfnasCellDesc& d, const float* rec, float* dgate, float* pp, hipStream_t s) {
    ProfScope _prof(TK_SE_BWD_REDUCE, s);
    hipLaunchKernelGGL(k_bn2_gather, dim3(d.N, chunk_count(d, 64, false)), dim3(256), 0, s, d, rec, dgate, pp);
    return (int)hipGetLastError();
}

int launch_bn2_finish(const TfnasCellDesc& d, const float* pp, const float* gate, const float* dpooled, double* red2,
                      hipStream_t s) {
    ProfScope _prof(TK_BN2_BWD, s);
    hipLaunchKernelGGL(k_bn2_finish, dim3(chunk_count(d, 64, false)), dim3(256), 0, s, d, pp, gate, dpooled, red2);
    return (int)hipGetLastError();
}

int launch_head_pool(const TfnasCellDesc& d, const float* E, const double* stats1, float* pooled, hipStream_t s) {
    ProfScope _prof(TK_SMALL, s);
    dim3 grid(d.N, cdiv(d.g[0].mcp, 64));
    ACT_DISPATCH(d.act, { hipLaunchKernelGGL((k_head_pool<ACT>), grid, dim3(256), 0, s, d, E, stats1, pooled); })
    return (int)hipGetLastError();
}

int launch_head_bwd(const TfnasCellDesc& d, const float* E, const double* stats1, const float* dpooled, float* dEh,
                    double* red1, float* part, hipStream_t s) {
    ProfScope _prof(TK_SMALL, s);
    if ((size_t)d.N * 2 * d.M > TFNAS_PART_FLOATS) return TFNAS_ERANGE;
    dim3 grid(d.N, cdiv(d.g[0].mcp, 64));
    ACT_DISPATCH(d.act, { hipLaunchKernelGGL((k_head_bwd<ACT>), grid, dim3(256), 0, s, d, E, stats1, dpooled, dEh, part); })
    _prof.stop();
    return launch_reduce_rows(part, d.N, 2 * d.M, 2 * (size_t)d.M, red1, nullptr, s);
}

int launch_bn1_consts(const TfnasCellDesc& d, const double* stats1, const double* red1, float* cb1,
                      hipStream_t s) {
    ProfScope _prof(TK_SMALL, s);
    hipLaunchKernelGGL(k_bn1_consts, dim3(cdiv(d.M, 256)), dim3(256), 0, s, d, stats1, red1, cb1);
    return (int)hipGetLastError();
}
