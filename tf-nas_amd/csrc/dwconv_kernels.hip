// Depthwise k x k convolution kernels (NHWC fp32) of the MBConv candidates, LDS-tiled.
//
// Reference arithmetic: depth_conv of MBInvertedResBlock (models/layers.py:484-507, forward :547):
//   Conv2d(mc, mc, k, stride, pad=k//2, groups=mc, bias=False) -> BN -> act
// The BN1+act that PRECEDES the depthwise conv is fused into the tile load (applied once per element,
// zero padding applied after it, exactly like conv padding of the activated tensor); the BN2 batch
// statistics of the conv output are accumulated in the epilogue.
//
// Work decomposition: one workgroup = one spatial tile x one chunk of CC channels of one group.
// Threads are (channel-quad, strip) pairs; a strip is 4 consecutive pixels along W, so every LDS access is
// one ds_read_b128 of 4 channels and the k-wide sliding window lives in registers.
#include "tfnas_dev.h"
#include "kernels.h"
#include "prof.h"

struct DwGeom {
    int T0, T1;        // tile height / width (in outputs for fwd & wgrad, in inputs for bwd-data)
    int CC;            // channels per workgroup (16/32/64)
    int tilesH, tilesW;
};

__device__ __forceinline__ int floordiv(int a, int b) { return a >= 0 ? a / b : -((-a + b - 1) / b); }

// locate (group, first channel) of channel-chunk `cy` among the groups whose kernel size is K
template <int K>
__device__ __forceinline__ bool dw_locate(const TfnasCellDesc& d, int cy, int CC, int& g, int& c0) {
    for (g = 0; g < d.G; ++g) {
        if (d.g[g].k != K) continue;
        const int t = (d.g[g].mcp + CC - 1) / CC;
        if (cy < t) {
            c0 = cy * CC;
            return true;
        }
        cy -= t;
    }
    return false;
}

// sum the per-thread float4 partials of all threads that share a channel quad (tid % CQ) and add the
// CC per-channel totals to the double accumulators acc[2*(c)+which]
__device__ __forceinline__ void dw_flush_pair(f32x4 a, f32x4 b, float* red, int CC, int c0, int mc, double* acc) {
    const int tid = threadIdx.x, CQ = CC >> 2;
    __syncthreads();
    st4(red + tid * 8, a);
    st4(red + tid * 8 + 4, b);
    __syncthreads();
    if (tid < CC && c0 + tid < mc) {
        const int cq = tid >> 2, comp = tid & 3;
        float s = 0.f, q = 0.f;
        for (int t = cq; t < 256; t += CQ) {
            s += red[t * 8 + comp];
            q += red[t * 8 + 4 + comp];
        }
        atomic_add_f64(acc + 2 * (size_t)(c0 + tid) + 0, (double)s);
        atomic_add_f64(acc + 2 * (size_t)(c0 + tid) + 1, (double)q);
    }
}

// ============================================================================ forward
template <int K, int S, int ACT>
__global__ __launch_bounds__(256) void k_dw_fwd(TfnasCellDesc d, const float* __restrict__ E,
                                                const double* __restrict__ stats1, float* __restrict__ D,
                                                double* __restrict__ stats2, DwGeom gm) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    int g, c0;
    if (!dw_locate<K>(d, blockIdx.y, gm.CC, g, c0)) return;
    const int mc = d.g[g].mc, mcp = d.g[g].mcp, off = d.g[g].off;
    const float* __restrict__ w = d.g[g].w_dw;
    const int CC = gm.CC, CQ = CC >> 2, TH = gm.T0, TW = gm.T1;
    const int H = d.H, W = d.W, Ho = d.Ho, Wo = d.Wo, M = d.M;
    const int tid = threadIdx.x;
    const int bx = blockIdx.x, tw = bx % gm.tilesW, th = (bx / gm.tilesW) % gm.tilesH, n = bx / (gm.tilesW * gm.tilesH);
    const int ho0 = th * TH, wo0 = tw * TW;
    const int IH = (TH - 1) * S + K, IW = (TW - 1) * S + K;
    const int hi0 = ho0 * S - K / 2, wi0 = wo0 * S - K / 2;

    const int tile_floats = max(IH * IW * CC, 2048);
    float* in_tile = lds;
    float* wts = lds + tile_floats;
    float2* cst = reinterpret_cast<float2*>(wts + K * K * CC);

    if (tid < CC)
        cst[tid] = (c0 + tid < mc) ? bn_consts(stats1 + 2 * (size_t)(off + c0 + tid), 1.0 / ((double)d.N * H * W), d.eps)
                                   : make_float2(0.f, 0.f);
    for (int idx = tid; idx < K * K * CC; idx += 256) {
        const int cl = idx % CC, t = idx / CC;
        wts[t * CC + cl] = (c0 + cl < mc) ? w[(size_t)(c0 + cl) * (K * K) + t] : 0.f;
    }
    __syncthreads();
    for (int idx = tid; idx < IH * IW * CQ; idx += 256) {
        const int cq = idx % CQ, pix = idx / CQ;
        const int hi = hi0 + pix / IW, wi = wi0 + pix % IW;
        f32x4 v = zero4();
        if (hi >= 0 && hi < H && wi >= 0 && wi < W && c0 + 4 * cq < mcp) {
            v = ld4(E + ((size_t)(n * H + hi) * W + wi) * M + off + c0 + 4 * cq);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float2 c = cst[4 * cq + j];
                v[j] = act_f<ACT>((v[j] - c.x) * c.y);
            }
        }
        st4(in_tile + pix * CC + 4 * cq, v);
    }
    __syncthreads();

    constexpr int WIN = 3 * S + K;
    const int nsw = TW >> 2, nstrips = TH * nsw;
    f32x4 ssum = zero4(), ssq = zero4();
    for (int item = tid; item < nstrips * CQ; item += 256) {
        const int cq = item % CQ, st = item / CQ;
        const int oh = st / nsw, ow0 = (st % nsw) * 4;
        f32x4 acc[4] = {zero4(), zero4(), zero4(), zero4()};
#pragma unroll
        for (int ky = 0; ky < K; ++ky) {
            const float* rowp = in_tile + ((oh * S + ky) * IW + ow0 * S) * CC + 4 * cq;
            f32x4 win[WIN];
#pragma unroll
            for (int t = 0; t < WIN; ++t) win[t] = ld4(rowp + t * CC);
#pragma unroll
            for (int kx = 0; kx < K; ++kx) {
                const f32x4 wv = ld4(wts + (ky * K + kx) * CC + 4 * cq);
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[j] += win[j * S + kx] * wv;
            }
        }
        const int ho = ho0 + oh;
        if (ho < Ho && c0 + 4 * cq < mcp) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int wo = wo0 + ow0 + j;
                if (wo < Wo) {
                    st4(D + ((size_t)(n * Ho + ho) * Wo + wo) * M + off + c0 + 4 * cq, acc[j]);
                    ssum += acc[j];
                    ssq += acc[j] * acc[j];
                }
            }
        }
    }
    dw_flush_pair(ssum, ssq, in_tile, CC, c0, mc, stats2 + 2 * (size_t)off);
}

// ---------------------------------------------------------------------------- BN2-backward operand
// ddh = d loss / d dhat (stored in place of dZ by k_bn2_bwd);  dd = rstd2*(ddh - R1/Po - dhat*R2/Po)
// cst2[c] = (mean2, rstd2, R1/Po, R2/Po)
__device__ __forceinline__ f32x4 bn2_dd(const f32x4* cst2, int cl, f32x4 ddh, f32x4 dv) {
    f32x4 r;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const f32x4 t = cst2[cl + j];
        const float dh = (dv[j] - t.x) * t.y;
        r[j] = t.y * (ddh[j] - t.z - dh * t.w);
    }
    return r;
}
__device__ __forceinline__ void fill_cst2(f32x4* cst2, const TfnasCellDesc& d, int CC, int c0, int mc, int off,
                                          const double* stats2, const double* red2) {
    const int tid = threadIdx.x;
    if (tid < CC) {
        f32x4 t = zero4();
        if (c0 + tid < mc) {
            const double inv = 1.0 / ((double)d.N * d.Ho * d.Wo);
            const float2 c = bn_consts(stats2 + 2 * (size_t)(off + c0 + tid), inv, d.eps);
            t.x = c.x;
            t.y = c.y;
            t.z = (float)(red2[2 * (size_t)(off + c0 + tid) + 0] * inv);
            t.w = (float)(red2[2 * (size_t)(off + c0 + tid) + 1] * inv);
        }
        cst2[tid] = t;
    }
}

// ============================================================================ backward w.r.t. input
// dA1[n][hi][wi][c] = sum_{ky,kx} dd[n][(hi+p-ky)/S][(wi+p-kx)/S][c] * w[c][ky][kx]   (only exact divisions)
// epilogue: deh = dA1 * act'(ehat) -> dEh, and the BN1-backward sums (T1 = sum deh, T2 = sum deh*ehat)
template <int K, int S, int ACT>
__global__ __launch_bounds__(256) void k_dw_bwd_data(TfnasCellDesc d, const float* __restrict__ ddh,
                                                     const float* __restrict__ D, const double* __restrict__ stats2,
                                                     const double* __restrict__ red2, const float* __restrict__ E,
                                                     const double* __restrict__ stats1, float* __restrict__ dEh,
                                                     double* __restrict__ red1, DwGeom gm) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    int g, c0;
    if (!dw_locate<K>(d, blockIdx.y, gm.CC, g, c0)) return;
    const int mc = d.g[g].mc, mcp = d.g[g].mcp, off = d.g[g].off;
    const float* __restrict__ w = d.g[g].w_dw;
    const int CC = gm.CC, CQ = CC >> 2, TIH = gm.T0, TIW = gm.T1;
    const int H = d.H, W = d.W, Ho = d.Ho, Wo = d.Wo, M = d.M;
    constexpr int PAD = K / 2;
    const int tid = threadIdx.x;
    const int bx = blockIdx.x, tw = bx % gm.tilesW, th = (bx / gm.tilesW) % gm.tilesH, n = bx / (gm.tilesW * gm.tilesH);
    const int hi0 = th * TIH, wi0 = tw * TIW;
    const int oh0 = floordiv(hi0 + PAD - (K - 1), S), ow0 = floordiv(wi0 + PAD - (K - 1), S);
    const int OH = (hi0 + TIH - 1 + PAD) / S - oh0 + 1, OW = (wi0 + TIW - 1 + PAD) / S - ow0 + 1;

    const int tile_floats = max(OH * OW * CC, 2048);
    float* dd_tile = lds;
    float* wts = lds + tile_floats;
    f32x4* cst2 = reinterpret_cast<f32x4*>(wts + K * K * CC);
    float2* cst1 = reinterpret_cast<float2*>(cst2 + CC);

    fill_cst2(cst2, d, CC, c0, mc, off, stats2, red2);
    if (tid < CC)
        cst1[tid] = (c0 + tid < mc) ? bn_consts(stats1 + 2 * (size_t)(off + c0 + tid), 1.0 / ((double)d.N * H * W), d.eps)
                                    : make_float2(0.f, 0.f);
    for (int idx = tid; idx < K * K * CC; idx += 256) {
        const int cl = idx % CC, t = idx / CC;
        wts[t * CC + cl] = (c0 + cl < mc) ? w[(size_t)(c0 + cl) * (K * K) + t] : 0.f;
    }
    __syncthreads();
    for (int idx = tid; idx < OH * OW * CQ; idx += 256) {
        const int cq = idx % CQ, pix = idx / CQ;
        const int ho = oh0 + pix / OW, wo = ow0 + pix % OW;
        f32x4 v = zero4();
        if (ho >= 0 && ho < Ho && wo >= 0 && wo < Wo && c0 + 4 * cq < mcp) {
            const size_t a = ((size_t)(n * Ho + ho) * Wo + wo) * M + off + c0 + 4 * cq;
            v = bn2_dd(cst2, 4 * cq, ld4(ddh + a), ld4(D + a));
        }
        st4(dd_tile + pix * CC + 4 * cq, v);
    }
    __syncthreads();

    const int nsw = TIW >> 2, nstrips = TIH * nsw;
    f32x4 t1 = zero4(), t2 = zero4();
    for (int item = tid; item < nstrips * CQ; item += 256) {
        const int cq = item % CQ, st = item / CQ;
        const int ih = st / nsw, iw0 = (st % nsw) * 4;
        const int hi = hi0 + ih;
        f32x4 acc[4] = {zero4(), zero4(), zero4(), zero4()};
        if (S == 1) {
            // column index of output (wi + PAD - kx) relative to ow0 = wi0 + PAD - (K-1):  iw0 + j - kx + K - 1
#pragma unroll
            for (int ky = 0; ky < K; ++ky) {
                const int r = hi + PAD - ky - oh0;
                const float* rowp = dd_tile + (r * OW + iw0) * CC + 4 * cq;
                f32x4 win[K + 3];
#pragma unroll
                for (int t = 0; t < K + 3; ++t) win[t] = ld4(rowp + t * CC);
#pragma unroll
                for (int kx = 0; kx < K; ++kx) {
                    const f32x4 wv = ld4(wts + (ky * K + kx) * CC + 4 * cq);
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[j] += win[j - kx + K - 1] * wv;
                }
            }
        } else {
            const int base = wi0 + iw0;   // multiple of 4 -> even
#pragma unroll
            for (int ky = 0; ky < K; ++ky) {
                const int t = hi + PAD - ky;
                if (t & 1) continue;
                const int r = t / 2 - oh0;   // t even: exact also for negatives
#pragma unroll
                for (int kx = 0; kx < K; ++kx) {
                    const f32x4 wv = ld4(wts + (ky * K + kx) * CC + 4 * cq);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        if ((j + PAD - kx) & 1) continue;   // compile-time after unrolling
                        const int c = base / 2 + (j + PAD - kx) / 2 - ow0;
                        acc[j] += ld4(dd_tile + (r * OW + c) * CC + 4 * cq) * wv;
                    }
                }
            }
        }
        if (hi < H && c0 + 4 * cq < mcp) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int wi = wi0 + iw0 + j;
                if (wi < W) {
                    const size_t a = ((size_t)(n * H + hi) * W + wi) * M + off + c0 + 4 * cq;
                    const f32x4 e = ld4(E + a);
                    f32x4 deh;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float2 c = cst1[4 * cq + q];
                        const float eh = (e[q] - c.x) * c.y;
                        deh[q] = acc[j][q] * act_d<ACT>(eh);
                        t1[q] += deh[q];
                        t2[q] += deh[q] * eh;
                    }
                    st4(dEh + a, deh);
                }
            }
        }
    }
    dw_flush_pair(t1, t2, dd_tile, CC, c0, mc, red1 + 2 * (size_t)off);
}

// ============================================================================ weight gradient
// g_dw[c][ky][kx] += sum_{n,ho,wo} dd[n][ho][wo][c] * a1[n][ho*S+ky-p][wo*S+kx-p][c],  a1 = act(BN1(E))
template <int K, int S, int ACT>
__global__ __launch_bounds__(256) void k_dw_wgrad(TfnasCellDesc d, const float* __restrict__ ddh,
                                                  const float* __restrict__ D, const double* __restrict__ stats2,
                                                  const double* __restrict__ red2, const float* __restrict__ E,
                                                  const double* __restrict__ stats1, DwGeom gm) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    int g, c0;
    if (!dw_locate<K>(d, blockIdx.y, gm.CC, g, c0)) return;
    const int mc = d.g[g].mc, mcp = d.g[g].mcp, off = d.g[g].off;
    float* __restrict__ gw = d.g[g].g_dw;
    const int CC = gm.CC, CQ = CC >> 2, TH = gm.T0, TW = gm.T1;
    const int H = d.H, W = d.W, Ho = d.Ho, Wo = d.Wo, M = d.M;
    const int tid = threadIdx.x;
    const int bx = blockIdx.x, tw = bx % gm.tilesW, th = (bx / gm.tilesW) % gm.tilesH, n = bx / (gm.tilesW * gm.tilesH);
    const int ho0 = th * TH, wo0 = tw * TW;
    const int IH = (TH - 1) * S + K, IW = (TW - 1) * S + K;
    const int hi0 = ho0 * S - K / 2, wi0 = wo0 * S - K / 2;

    const int tile_floats = max(IH * IW * CC, 4 * K * K * CC);
    float* in_tile = lds;
    f32x4* cst2 = reinterpret_cast<f32x4*>(lds + tile_floats);
    float2* cst1 = reinterpret_cast<float2*>(cst2 + CC);

    fill_cst2(cst2, d, CC, c0, mc, off, stats2, red2);
    if (tid < CC)
        cst1[tid] = (c0 + tid < mc) ? bn_consts(stats1 + 2 * (size_t)(off + c0 + tid), 1.0 / ((double)d.N * H * W), d.eps)
                                    : make_float2(0.f, 0.f);
    __syncthreads();
    for (int idx = tid; idx < IH * IW * CQ; idx += 256) {
        const int cq = idx % CQ, pix = idx / CQ;
        const int hi = hi0 + pix / IW, wi = wi0 + pix % IW;
        f32x4 v = zero4();
        if (hi >= 0 && hi < H && wi >= 0 && wi < W && c0 + 4 * cq < mcp) {
            v = ld4(E + ((size_t)(n * H + hi) * W + wi) * M + off + c0 + 4 * cq);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float2 c = cst1[4 * cq + j];
                v[j] = act_f<ACT>((v[j] - c.x) * c.y);
            }
        }
        st4(in_tile + pix * CC + 4 * cq, v);
    }
    __syncthreads();

    constexpr int WIN = 3 * S + K;
    const int nsw = TW >> 2, nstrips = TH * nsw;
    f32x4 wacc[K * K];
#pragma unroll
    for (int t = 0; t < K * K; ++t) wacc[t] = zero4();
    for (int item = tid; item < nstrips * CQ; item += 256) {
        const int cq = item % CQ, st = item / CQ;
        const int oh = st / nsw, ow0 = (st % nsw) * 4;
        const int ho = ho0 + oh;
        f32x4 dd[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int wo = wo0 + ow0 + j;
            dd[j] = zero4();
            if (ho < Ho && wo < Wo && c0 + 4 * cq < mcp) {
                const size_t a = ((size_t)(n * Ho + ho) * Wo + wo) * M + off + c0 + 4 * cq;
                dd[j] = bn2_dd(cst2, 4 * cq, ld4(ddh + a), ld4(D + a));
            }
        }
#pragma unroll
        for (int ky = 0; ky < K; ++ky) {
            const float* rowp = in_tile + ((oh * S + ky) * IW + ow0 * S) * CC + 4 * cq;
            f32x4 win[WIN];
#pragma unroll
            for (int t = 0; t < WIN; ++t) win[t] = ld4(rowp + t * CC);
#pragma unroll
            for (int kx = 0; kx < K; ++kx)
#pragma unroll
                for (int j = 0; j < 4; ++j) wacc[ky * K + kx] += dd[j] * win[j * S + kx];
        }
    }
    // reduce over the threads sharing a channel quad: first inside the wave, then across the 4 waves
    for (int o = CQ; o < 64; o <<= 1) {
#pragma unroll
        for (int t = 0; t < K * K; ++t) {
#pragma unroll
            for (int q = 0; q < 4; ++q) wacc[t][q] += __shfl_xor(wacc[t][q], o, 64);
        }
    }
    __syncthreads();
    float* wred = in_tile;   // [4 waves][K*K][CC]
    const int lane = tid & 63, wv = tid >> 6;
    if (lane < CQ) {
#pragma unroll
        for (int t = 0; t < K * K; ++t) st4(wred + (wv * K * K + t) * CC + 4 * lane, wacc[t]);
    }
    __syncthreads();
    for (int idx = tid; idx < K * K * CC; idx += 256) {
        const int cl = idx % CC, t = idx / CC;
        if (c0 + cl < mc) {
            float s = 0.f;
#pragma unroll
            for (int ww = 0; ww < 4; ++ww) s += wred[(ww * K * K + t) * CC + cl];
            atomic_add_f32(gw + (size_t)(c0 + cl) * (K * K) + t, s);
        }
    }
}

// ============================================================================ host side
static void pick_tile(int Th, int Tw, int K, int S, bool fwd_like, DwGeom& gm) {
    // T1 (width) multiple of 4 (strips), up to 16; T0 so that a tile has ~128 pixels
    gm.T1 = Tw >= 16 ? 16 : ((Tw + 3) / 4) * 4;
    gm.T0 = 128 / gm.T1;
    if (gm.T0 > Th) gm.T0 = Th;
    if (gm.T0 < 1) gm.T0 = 1;
    gm.tilesH = cdiv(Th, gm.T0);
    gm.tilesW = cdiv(Tw, gm.T1);
    int lh, lw;   // LDS tile extent
    if (fwd_like) {
        lh = (gm.T0 - 1) * S + K;
        lw = (gm.T1 - 1) * S + K;
    } else {
        lh = (gm.T0 + K - 1 + S - 1) / S + 1;
        lw = (gm.T1 + K - 1 + S - 1) / S + 1;
    }
    const int px = lh * lw;
    const int items32 = gm.T0 * (gm.T1 / 4) * 8;
    gm.CC = 32;
    if (px * 32 * 4 > 56 * 1024) gm.CC = 16;
    else if (items32 < 256 && px * 64 * 4 <= 56 * 1024) gm.CC = 64;
}

static int dw_chunks(const TfnasCellDesc& d, int K, int CC) {
    int t = 0;
    for (int g = 0; g < d.G; ++g)
        if (d.g[g].k == K) t += cdiv(d.g[g].mcp, CC);
    return t;
}

#define DW_DISPATCH(K_, S_, ACT_, ...)                                                         \
    if ((K_) == 3 && (S_) == 1 && (ACT_) == 0) { constexpr int K = 3, S = 1, ACT = 0; __VA_ARGS__; } \
    else if ((K_) == 3 && (S_) == 1) { constexpr int K = 3, S = 1, ACT = 1; __VA_ARGS__; }     \
    else if ((K_) == 3 && (S_) == 2 && (ACT_) == 0) { constexpr int K = 3, S = 2, ACT = 0; __VA_ARGS__; } \
    else if ((K_) == 3 && (S_) == 2) { constexpr int K = 3, S = 2, ACT = 1; __VA_ARGS__; }     \
    else if ((K_) == 5 && (S_) == 1 && (ACT_) == 0) { constexpr int K = 5, S = 1, ACT = 0; __VA_ARGS__; } \
    else if ((K_) == 5 && (S_) == 1) { constexpr int K = 5, S = 1, ACT = 1; __VA_ARGS__; }     \
    else if ((K_) == 5 && (S_) == 2 && (ACT_) == 0) { constexpr int K = 5, S = 2, ACT = 0; __VA_ARGS__; } \
    else { constexpr int K = 5, S = 2, ACT = 1; __VA_ARGS__; }

int launch_dw_fwd(const TfnasCellDesc& d, const float* E, const double* stats1, float* D, double* stats2,
                  hipStream_t s) {
    ProfScope _prof(TK_DW_FWD, s);
    for (int kk = 3; kk <= 5; kk += 2) {
        DwGeom gm;
        pick_tile(d.Ho, d.Wo, kk, d.stride, true, gm);
        const int chunks = dw_chunks(d, kk, gm.CC);
        if (!chunks) continue;
        const int IH = (gm.T0 - 1) * d.stride + kk, IW = (gm.T1 - 1) * d.stride + kk;
        const int tile = IH * IW * gm.CC > 2048 ? IH * IW * gm.CC : 2048;
        const size_t shm = (size_t)(tile + kk * kk * gm.CC + 2 * gm.CC) * sizeof(float);
        dim3 grid(d.N * gm.tilesH * gm.tilesW, chunks);
        DW_DISPATCH(kk, d.stride, d.act, {
            hipLaunchKernelGGL((k_dw_fwd<K, S, ACT>), grid, dim3(256), shm, s, d, E, stats1, D, stats2, gm);
        })
    }
    return (int)hipGetLastError();
}

int launch_dw_bwd_data(const TfnasCellDesc& d, const float* ddh, const float* D, const double* stats2,
                       const double* red2, const float* E, const double* stats1, float* dEh, double* red1,
                       hipStream_t s) {
    ProfScope _prof(TK_DW_BWD_DATA, s);
    for (int kk = 3; kk <= 5; kk += 2) {
        DwGeom gm;
        pick_tile(d.H, d.W, kk, d.stride, false, gm);
        const int chunks = dw_chunks(d, kk, gm.CC);
        if (!chunks) continue;
        const int OH = (gm.T0 + kk - 1 + d.stride - 1) / d.stride + 1, OW = (gm.T1 + kk - 1 + d.stride - 1) / d.stride + 1;
        const int tile = OH * OW * gm.CC > 2048 ? OH * OW * gm.CC : 2048;
        const size_t shm = (size_t)(tile + kk * kk * gm.CC + 4 * gm.CC + 2 * gm.CC) * sizeof(float);
        dim3 grid(d.N * gm.tilesH * gm.tilesW, chunks);
        DW_DISPATCH(kk, d.stride, d.act, {
            hipLaunchKernelGGL((k_dw_bwd_data<K, S, ACT>), grid, dim3(256), shm, s, d, ddh, D, stats2, red2, E,
                               stats1, dEh, red1, gm);
        })
    }
    return (int)hipGetLastError();
}

int launch_dw_wgrad(const TfnasCellDesc& d, const float* ddh, const float* D, const double* stats2,
                    const double* red2, const float* E, const double* stats1, hipStream_t s) {
    ProfScope _prof(TK_DW_WGRAD, s);
    for (int kk = 3; kk <= 5; kk += 2) {
        DwGeom gm;
        pick_tile(d.Ho, d.Wo, kk, d.stride, true, gm);
        const int chunks = dw_chunks(d, kk, gm.CC);
        if (!chunks) continue;
        const int IH = (gm.T0 - 1) * d.stride + kk, IW = (gm.T1 - 1) * d.stride + kk;
        int tile = IH * IW * gm.CC;
        if (tile < 4 * kk * kk * gm.CC) tile = 4 * kk * kk * gm.CC;
        const size_t shm = (size_t)(tile + 4 * gm.CC + 2 * gm.CC) * sizeof(float);
        dim3 grid(d.N * gm.tilesH * gm.tilesW, chunks);
        DW_DISPATCH(kk, d.stride, d.act, {
            hipLaunchKernelGGL((k_dw_wgrad<K, S, ACT>), grid, dim3(256), shm, s, d, ddh, D, stats2, red2, E, stats1,
                               gm);
        })
    }
    return (int)hipGetLastError();
}
