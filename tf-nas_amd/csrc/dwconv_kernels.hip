// Depthwise k x k convolution kernels (NHWC fp32) of the MBConv candidates, LDS-tiled.
//
// Reference arithmetic: depth_conv of MBInvertedResBlock (models/layers.py:484-507, forward :547):
//   Conv2d(mc, mc, k, stride, pad=k//2, groups=mc, bias=False) -> BN -> act
// The BN1+act that PRECEDES the depthwise conv is fused into the tile load (applied once per element,
// zero padding applied after it, exactly like conv padding of the activated tensor); the BN2 batch
// statistics of the conv output are accumulated in the epilogue.
//
// Work decomposition: a workgroup owns one chunk of CC channels of one group and walks over spatial tiles
// (persistent, grid-stride): the depthwise weights and BN constants of the chunk are staged in LDS once, the
// per-channel statistics are accumulated in registers across tiles and flushed with ONE set of double atomics
// per workgroup.  Threads are (channel-quad, strip) pairs; a strip is 4 consecutive pixels along W, so every
// LDS access is one ds_read_b128 of 4 channels and the k-wide sliding window lives in registers.  Tile loads
// are issued 4 at a time per thread before any of them is consumed (memory-level parallelism).
#include <cstring>
#include "tfnas_dev.h"
#include "kernels.h"
#include "prof.h"
#include "efree.h"

struct DwGeom {
    int T0, T1;        // tile height / width (in outputs for fwd & wgrad, in inputs for bwd-data)
    int CC;            // channels per workgroup (16/32/64)
    int cq_shift;      // log2(CC/4)
    int tilesH, tilesW, ntiles;   // ntiles = N * tilesH * tilesW
    int L0, L1;        // LDS tile extent (rows, cols)
};

__device__ __forceinline__ int floordiv(int a, int b) { return a >= 0 ? a / b : -((-a + b - 1) / b); }

// XCD-aware tile order.  Workgroup b is observed to run on XCD b % 8 and every XCD has a private L2; spatially
// adjacent tiles share halo pixels, so they should be in flight on the SAME XCD.  With gridDim.x a multiple of 8 the
// workgroups of XCD x take the contiguous tile range [x*gx/8, (x+1)*gx/8) of every grid-stride round (pure speed
// heuristic: any placement gives the same results).
__device__ __forceinline__ int xcd_first_tile() {
    const int gx = gridDim.x, bx = blockIdx.x;
    return (gx & 7) ? bx : (bx & 7) * (gx >> 3) + (bx >> 3);
}

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

// sum the per-thread float4 partials of all threads that share a channel quad (tid % CQ) and store the CC
// per-channel totals as this workgroup's partial pair: acc[2*c + which] (acc = row blockIdx.x of the partials
// matrix, already offset to the group's first channel); k_reduce_rows sums the rows afterwards
__device__ __forceinline__ void dw_flush_pair(f32x4 a, f32x4 b, float* red, int CC, int c0, int mcp, float* acc,
                                              bool = false) {
    const int tid = threadIdx.x, CQ = CC >> 2;
    __syncthreads();
    st4(red + tid * 8, a);
    st4(red + tid * 8 + 4, b);
    __syncthreads();
    if (tid < CC && c0 + tid < mcp) {
        const int cq = tid >> 2, comp = tid & 3;
        float s = 0.f, q = 0.f;
        for (int t = cq; t < 256; t += CQ) {
            s += red[t * 8 + comp];
            q += red[t * 8 + 4 + comp];
        }
        acc[2 * (size_t)(c0 + tid) + 0] = s;
        acc[2 * (size_t)(c0 + tid) + 1] = q;
    }
}

// Statistics epilogue of the depthwise kernels: this workgroup's partial pair row (row `lane`); a k_reduce_rows / k_reduce_bn1
// launch sums the rows in double.  (Rounds 2-3 also had the producer's last workgroup sum them -- ticket counters + coherent
// read-back; measured no faster in two rounds and removed)
__device__ __forceinline__ void dw_flush_stats(f32x4 a, f32x4 b, float* lds, int CC, int c0, int mcp, int off, int M, float* part,
                                               int lane) {
    dw_flush_pair(a, b, lds, CC, c0, mcp, part + (size_t)lane * 2 * M + 2 * (size_t)off);
}

__device__ __forceinline__ void stage_weights(float* wts, const float* __restrict__ w, int KK, int CC, int c0, int mc) {
    for (int idx = threadIdx.x; idx < KK * CC; idx += 256) {
        const int cl = idx % CC, t = idx / CC;
        wts[t * CC + cl] = (c0 + cl < mc) ? w[(size_t)(c0 + cl) * KK + t] : 0.f;
    }
}

// Cooperative load of an [L0 x L1] pixel tile x CC channels into LDS (layout [pix][CC]).
// fetch(pix_row, pix_col, cq, ok) -> f32x4 is called only to build the value AFTER the raw loads were issued:
//   addr(r, c, cq, valid&) returns the element offset; xf(v, cq) transforms the loaded vector.
// two-source variant: both sources share the addressing; xf(v0, v1, cq) combines them (8 loads in flight)
template <class FAddr, class FXf>
__device__ __forceinline__ void load_tile2(float* tile, int L0, int L1, int CC, int cq_shift,
                                           const float* __restrict__ src0, const float* __restrict__ src1, FAddr addr,
                                           FXf xf, int stor) {
    const int CQ = CC >> 2, total = L0 * L1 * CQ;
    const float inv_l1 = 1.f / (float)L1;
    for (int base = 0; base < total; base += 1024) {
        f32x4 v0[4], v1[4];
        int pix[4], cqv[4];
        bool ok[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int idx = base + u * 256 + threadIdx.x;
            pix[u] = idx >> cq_shift;
            cqv[u] = idx & (CQ - 1);
            const int r = (int)(((float)pix[u] + 0.5f) * inv_l1);
            const int c = pix[u] - r * L1;
            size_t a = 0;
            ok[u] = idx < total && addr(r, c, cqv[u], a);
            v0[u] = ok[u] ? ldS4_nt(src0, a, stor) : zero4();
            v1[u] = ok[u] ? ldS4_nt(src1, a, stor) : zero4();
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int idx = base + u * 256 + threadIdx.x;
            if (idx < total) st4(tile + pix[u] * CC + 4 * cqv[u], ok[u] ? xf(v0[u], v1[u], cqv[u]) : zero4());
        }
    }
}

template <class FAddr, class FXf>
__device__ __forceinline__ void load_tile(float* tile, int L0, int L1, int CC, int cq_shift, const float* __restrict__ src,
                                          FAddr addr, FXf xf, int stor) {
    const int CQ = CC >> 2, total = L0 * L1 * CQ;
    const float inv_l1 = 1.f / (float)L1;
    for (int base = 0; base < total; base += 1024) {
        f32x4 v[4];
        int pix[4], cqv[4];
        bool ok[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int idx = base + u * 256 + threadIdx.x;
            pix[u] = idx >> cq_shift;
            cqv[u] = idx & (CQ - 1);
            const int r = (int)(((float)pix[u] + 0.5f) * inv_l1);
            const int c = pix[u] - r * L1;
            size_t a = 0;
            ok[u] = idx < total && addr(r, c, cqv[u], a);
            v[u] = ok[u] ? ldS4_nt(src, a, stor) : zero4();
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int idx = base + u * 256 + threadIdx.x;
            if (idx < total) st4(tile + pix[u] * CC + 4 * cqv[u], ok[u] ? xf(v[u], cqv[u]) : zero4());
        }
    }
}

// ============================================================================ forward
// KQ = 0: the tile is loaded from E.  KQ = ic/4 > 0 (E-free, efree.h): the tile is recomputed from the cell input x.
template <int K, int S, int ACT, int KQ>
__global__ __launch_bounds__(256, 4) void k_dw_fwd(TfnasCellDesc d, const float* __restrict__ E,
                                                   const float* __restrict__ x,
                                                   const double* __restrict__ stats1, float* __restrict__ D,
                                                   float* __restrict__ part, DwGeom gm) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    int g, c0, lane = blockIdx.x, cy = blockIdx.y;
    if (KQ > 0) efree_lane_chunk(lane, cy);
    if (!dw_locate<K>(d, cy, gm.CC, g, c0)) return;
    const int mc = d.g[g].mc, mcp = d.g[g].mcp, off = d.g[g].off;
    const int CC = gm.CC, CQ = CC >> 2, TH = gm.T0, TW = gm.T1;
    const int H = d.H, W = d.W, Ho = d.Ho, Wo = d.Wo, M = d.M;
    const int tid = threadIdx.x;
    const int IH = gm.L0, IW = gm.L1;

    const int tile_floats = max(IH * IW * CC, 2048);
    float* in_tile = lds;
    float* wts = lds + tile_floats;
    float2* cst = reinterpret_cast<float2*>(wts + K * K * CC);

    if (tid < CC)
        cst[tid] = (c0 + tid < mc) ? bn_consts(stats1 + 2 * (size_t)(off + c0 + tid), 1.0 / ((double)d.N * H * W), d.eps)
                                   : make_float2(0.f, 0.f);
    stage_weights(wts, d.g[g].w_dw, K * K, CC, c0, mc);
    ExpandB<(KQ > 0 ? KQ : 2)> xb;
    if (KQ > 0) {
        __syncthreads();
        expand_b_load(xb, d.g[g].w_expand, d.ic, c0, mc, cst);
    }

    constexpr int WIN = 3 * S + K;
    const int nsw = TW >> 2, nstrips = TH * nsw;
    f32x4 ssum = zero4(), ssq = zero4();
    for (int t = KQ > 0 ? lane : xcd_first_tile(); t < gm.ntiles; t += gridDim.x) {
        const int tw = t % gm.tilesW, th = (t / gm.tilesW) % gm.tilesH, n = t / (gm.tilesW * gm.tilesH);
        const int ho0 = th * TH, wo0 = tw * TW;
        const int hi0 = ho0 * S - K / 2, wi0 = wo0 * S - K / 2;
        __syncthreads();     // previous tile fully consumed (and cst/wts visible on the first pass)
        if (KQ > 0) {
            const float inv_iw = 1.f / (float)IW;
            expand_tile<(KQ > 0 ? KQ : 2), ACT>(in_tile, IH * IW, CC, x, xb, [&](int p, size_t& a) {
                const int r = (int)(((float)p + 0.5f) * inv_iw), c = p - r * IW;
                const int hi = hi0 + r, wi = wi0 + c;
                a = ((size_t)(n * H + hi) * W + wi) * d.ic;
                return hi >= 0 && hi < H && wi >= 0 && wi < W;
            });
        } else
        load_tile(in_tile, IH, IW, CC, gm.cq_shift, E,
                  [&](int r, int c, int cq, size_t& a) {
                      const int hi = hi0 + r, wi = wi0 + c;
                      a = ((size_t)(n * H + hi) * W + wi) * M + off + c0 + 4 * cq;
                      return hi >= 0 && hi < H && wi >= 0 && wi < W && c0 + 4 * cq < mcp;
                  },
                  [&](f32x4 v, int cq) {
#pragma unroll
                      for (int j = 0; j < 4; ++j) {
                          const float2 c = cst[4 * cq + j];
                          v[j] = act_f<ACT>((v[j] - c.x) * c.y);
                      }
                      return v;
                  }, d.stor);
        __syncthreads();
        for (int item = tid; item < nstrips * CQ; item += 256) {
            const int cq = item & (CQ - 1), st = item >> gm.cq_shift;
            const int oh = st / nsw, ow0 = (st - oh * nsw) * 4;
            f32x4 acc[4] = {zero4(), zero4(), zero4(), zero4()};
#pragma unroll 1
            for (int ky = 0; ky < K; ++ky) {
                const float* rowp = in_tile + ((oh * S + ky) * IW + ow0 * S) * CC + 4 * cq;
                const float* wp = wts + ky * K * CC + 4 * cq;
                f32x4 win[WIN];
#pragma unroll
                for (int u = 0; u < WIN; ++u) win[u] = ld4(rowp + u * CC);
#pragma unroll
                for (int kx = 0; kx < K; ++kx) {
                    const f32x4 wv = ld4(wp + kx * CC);
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
                        stS4_nt(D, ((size_t)(n * Ho + ho) * Wo + wo) * M + off + c0 + 4 * cq, acc[j], d.stor);
                        ssum += acc[j];
                        ssq += acc[j] * acc[j];
                    }
                }
            }
        }
    }
    dw_flush_stats(ssum, ssq, in_tile, CC, c0, mcp, off, M, part, lane);
}

// ---------------------------------------------------------------------------- BN2-backward operand
// From dZ (gradient w.r.t. the gated activation that feeds the project conv) to dd (gradient w.r.t. the raw depthwise
// output D), i.e. the backward of  D -> BN2 -> act -> (* gate, SE pool path):
//   dhat = (D - mean2) * rstd2 ; da = dZ*gate + dpooled/HW  (SE groups; else dZ) ; ddh = da * act'(dhat)
//   dd   = rstd2 * (ddh - R1/Po - dhat * R2/Po)            R1 = sum ddh, R2 = sum ddh*dhat  (k_bn2_bwd)
// cst2[c] = (mean2, rstd2, R1/Po, R2/Po).  ddh is recomputed here instead of being written back by k_bn2_bwd
// (saves one write + nothing extra to read: dZ replaces ddh).
template <int ACT>
__device__ __forceinline__ f32x4 bn2_dd(const f32x4* cst2, int cl, f32x4 dz, f32x4 dv, bool has_se, f32x4 gate4,
                                        f32x4 dpool4) {
    f32x4 r;
    if (has_se) dz = dz * gate4 + dpool4;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const f32x4 t = cst2[cl + j];
        const float dh = (dv[j] - t.x) * t.y;
        const float ddh = dz[j] * act_d<ACT>(dh);
        r[j] = t.y * (ddh - t.z - dh * t.w);
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
// KQ > 0 (E-free, efree.h): ehat of the tile's input pixels is recomputed from x into a second LDS tile instead of
// being read from E.
template <int K, int S, int ACT, int KQ>
__global__ __launch_bounds__(256, 4) void k_dw_bwd_data(TfnasCellDesc d, const float* __restrict__ dZ,
                                                        const float* __restrict__ gate, const float* __restrict__ dpooled,
                                                        const float* __restrict__ D, const double* __restrict__ stats2,
                                                        const double* __restrict__ red2, const float* __restrict__ E,
                                                        const float* __restrict__ x,
                                                        const double* __restrict__ stats1, float* __restrict__ dEh,
                                                        float* __restrict__ part, DwGeom gm) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    int g, c0, lane = blockIdx.x, cy = blockIdx.y;
    if (KQ > 0) efree_lane_chunk(lane, cy);
    if (!dw_locate<K>(d, cy, gm.CC, g, c0)) return;
    const int mc = d.g[g].mc, mcp = d.g[g].mcp, off = d.g[g].off;
    const int CC = gm.CC, CQ = CC >> 2, TIH = gm.T0, TIW = gm.T1;
    const int H = d.H, W = d.W, Ho = d.Ho, Wo = d.Wo, M = d.M;
    constexpr int PAD = K / 2;
    const int tid = threadIdx.x;
    const int OH = gm.L0, OW = gm.L1;      // LDS tile extent (host: worst case over tile origins)

    const int tile_floats = max(OH * OW * CC, 2048);
    float* dd_tile = lds;
    float* wts = lds + tile_floats;
    f32x4* cst2 = reinterpret_cast<f32x4*>(wts + K * K * CC);
    float2* cst1 = reinterpret_cast<float2*>(cst2 + CC);

    fill_cst2(cst2, d, CC, c0, mc, off, stats2, red2);
    if (tid < CC)
        cst1[tid] = (c0 + tid < mc) ? bn_consts(stats1 + 2 * (size_t)(off + c0 + tid), 1.0 / ((double)d.N * H * W), d.eps)
                                    : make_float2(0.f, 0.f);
    stage_weights(wts, d.g[g].w_dw, K * K, CC, c0, mc);
    float* eh_tile = reinterpret_cast<float*>(cst1 + CC);     // [TIH*TIW][CC] (E-free only)
    ExpandB<(KQ > 0 ? KQ : 2)> xb;
    if (KQ > 0) {
        __syncthreads();
        expand_b_load(xb, d.g[g].w_expand, d.ic, c0, mc, cst1);
    }

    const int nsw = TIW >> 2, nstrips = TIH * nsw;
    const bool has_se = d.g[g].se > 0;
    const float inv_hw = 1.f / (float)(Ho * Wo);
    f32x4 t1 = zero4(), t2 = zero4();
    for (int t = KQ > 0 ? lane : xcd_first_tile(); t < gm.ntiles; t += gridDim.x) {
        const int tw = t % gm.tilesW, th = (t / gm.tilesW) % gm.tilesH, n = t / (gm.tilesW * gm.tilesH);
        const int hi0 = th * TIH, wi0 = tw * TIW;
        const int oh0 = floordiv(hi0 + PAD - (K - 1), S), ow0 = floordiv(wi0 + PAD - (K - 1), S);
        __syncthreads();
        // this thread always stages the same channel quad (256 % CQ == 0): its SE gate / pool-path terms of image n
        const int mycq = tid & (CQ - 1);
        f32x4 g4 = zero4(), dp4 = zero4();
        if (has_se && c0 + 4 * mycq < mcp) {
            g4 = ld4(gate + (size_t)n * M + off + c0 + 4 * mycq);
            dp4 = ld4(dpooled + (size_t)n * M + off + c0 + 4 * mycq) * splat4(inv_hw);
        }
        load_tile2(dd_tile, OH, OW, CC, gm.cq_shift, dZ, D,
                   [&](int r, int c, int cq, size_t& a) {
                       const int ho = oh0 + r, wo = ow0 + c;
                       a = ((size_t)(n * Ho + ho) * Wo + wo) * M + off + c0 + 4 * cq;
                       return ho >= 0 && ho < Ho && wo >= 0 && wo < Wo && c0 + 4 * cq < mcp;
                   },
                   [&](f32x4 v, f32x4 dv, int cq) { return bn2_dd<ACT>(cst2, 4 * cq, v, dv, has_se, g4, dp4); }, d.stor);
        if (KQ > 0) {
            const float inv_iw = 1.f / (float)TIW;
            expand_tile<(KQ > 0 ? KQ : 2), 2>(eh_tile, TIH * TIW, CC, x, xb, [&](int p, size_t& a) {
                const int r = (int)(((float)p + 0.5f) * inv_iw), c = p - r * TIW;
                const int hi = hi0 + r, wi = wi0 + c;
                a = ((size_t)(n * H + hi) * W + wi) * d.ic;
                return hi < H && wi < W;
            });
        }
        __syncthreads();

        for (int item = tid; item < nstrips * CQ; item += 256) {
            const int cq = item & (CQ - 1), st = item >> gm.cq_shift;
            // stride 2: which taps reach an input row depends on the row's parity, and a wave holds the strips of 2-4
            // consecutive rows -- in natural order half of its lanes sat out every ky iteration of the tap loop.  Rows are
            // therefore dealt even-first (0, 2, 4, .., 1, 3, ..): the rows of a wave share their parity.
            const int ihp = st / nsw, iw0 = (st - ihp * nsw) * 4;
            const int half = (TIH + 1) >> 1;
            const int ih = (S == 2) ? (ihp < half ? 2 * ihp : 2 * (ihp - half) + 1) : ihp;
            const int hi = hi0 + ih;
            f32x4 acc[4] = {zero4(), zero4(), zero4(), zero4()};
            // issue the epilogue's E loads now so that their latency hides under the tap loop
            f32x4 ev[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int wi = wi0 + iw0 + j;
                const bool okj = hi < H && wi < W && c0 + 4 * cq < mcp;
                if (KQ > 0) ev[j] = ld4(eh_tile + (ih * TIW + iw0 + j) * CC + 4 * cq);
                else ev[j] = okj ? ldS4_nt(E, ((size_t)(n * H + hi) * W + wi) * M + off + c0 + 4 * cq, d.stor) : zero4();
            }
            if (S == 1) {
                // column of output (wi + PAD - kx) relative to ow0 = wi0 + PAD - (K-1):  iw0 + j - kx + K - 1
#pragma unroll 1
                for (int ky = 0; ky < K; ++ky) {
                    const int r = hi + PAD - ky - oh0;
                    const float* rowp = dd_tile + (r * OW + iw0) * CC + 4 * cq;
                    const float* wp = wts + ky * K * CC + 4 * cq;
                    f32x4 win[K + 3];
#pragma unroll
                    for (int u = 0; u < K + 3; ++u) win[u] = ld4(rowp + u * CC);
#pragma unroll
                    for (int kx = 0; kx < K; ++kx) {
                        const f32x4 wv = ld4(wp + kx * CC);
#pragma unroll
                        for (int j = 0; j < 4; ++j) acc[j] += win[j - kx + K - 1] * wv;
                    }
                }
            } else {
                const int base = wi0 + iw0;   // multiple of 4 -> even
#pragma unroll 1
                for (int ky = 0; ky < K; ++ky) {
                    const int tt = hi + PAD - ky;
                    if (tt & 1) continue;
                    const int r = tt / 2 - oh0;   // tt even: exact also for negatives
                    const float* wp = wts + ky * K * CC + 4 * cq;
#pragma unroll
                    for (int kx = 0; kx < K; ++kx) {
                        const f32x4 wv = ld4(wp + kx * CC);
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
                        const f32x4 e = ev[j];
                        f32x4 deh;
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const float2 c = cst1[4 * cq + q];
                            const float eh = KQ > 0 ? e[q] : (e[q] - c.x) * c.y;
                            deh[q] = acc[j][q] * act_d<ACT>(eh);
                            t1[q] += deh[q];
                            t2[q] += deh[q] * eh;
                        }
                        stS4_nt(dEh, a, deh, d.stor);
                    }
                }
            }
        }
    }
    dw_flush_stats(t1, t2, dd_tile, CC, c0, mcp, off, M, part, lane);
}

// ============================================================================ weight gradient
// part[bx][poff_g + c*K*K + ky*K + kx] = sum over this workgroup's tiles of dd[n][ho][wo][c] * a1[n][ho*S+ky-p][wo*S+kx-p][c]
// (a1 = act(BN1(E))); k_reduce_rows sums the workgroups into g_dw
template <int K, int S, int ACT>
__global__ __launch_bounds__(256, 2) void k_dw_wgrad(TfnasCellDesc d, const float* __restrict__ dZ,
                                                     const float* __restrict__ gate, const float* __restrict__ dpooled,
                                                     const float* __restrict__ D, const double* __restrict__ stats2,
                                                     const double* __restrict__ red2, const float* __restrict__ E,
                                                     const double* __restrict__ stats1, float* __restrict__ part,
                                                     size_t out_size, DwGeom gm) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    int g, c0;
    if (!dw_locate<K>(d, blockIdx.y, gm.CC, g, c0)) return;
    const int mc = d.g[g].mc, mcp = d.g[g].mcp, off = d.g[g].off;
    size_t poff = 0;
    for (int gg = 0; gg < g; ++gg) poff += (size_t)d.g[gg].mc * d.g[gg].k * d.g[gg].k;
    float* __restrict__ gw = part + (size_t)blockIdx.x * out_size + poff;
    const int CC = gm.CC, CQ = CC >> 2, TH = gm.T0, TW = gm.T1;
    const int H = d.H, W = d.W, Ho = d.Ho, Wo = d.Wo, M = d.M;
    const int tid = threadIdx.x;
    const int IH = gm.L0, IW = gm.L1;

    const int tile_floats = max(IH * IW * CC, 4 * K * K * CC);
    float* in_tile = lds;
    f32x4* cst2 = reinterpret_cast<f32x4*>(lds + tile_floats);
    float2* cst1 = reinterpret_cast<float2*>(cst2 + CC);

    fill_cst2(cst2, d, CC, c0, mc, off, stats2, red2);
    if (tid < CC)
        cst1[tid] = (c0 + tid < mc) ? bn_consts(stats1 + 2 * (size_t)(off + c0 + tid), 1.0 / ((double)d.N * H * W), d.eps)
                                    : make_float2(0.f, 0.f);

    constexpr int WIN = 3 * S + K;
    const int nsw = TW >> 2, nstrips = TH * nsw;
    const bool has_se = d.g[g].se > 0;
    const float inv_hw = 1.f / (float)(Ho * Wo);
    f32x4 wacc[K * K];
#pragma unroll
    for (int u = 0; u < K * K; ++u) wacc[u] = zero4();
    for (int t = xcd_first_tile(); t < gm.ntiles; t += gridDim.x) {
        const int tw = t % gm.tilesW, th = (t / gm.tilesW) % gm.tilesH, n = t / (gm.tilesW * gm.tilesH);
        const int ho0 = th * TH, wo0 = tw * TW;
        const int hi0 = ho0 * S - K / 2, wi0 = wo0 * S - K / 2;
        const int mycq = tid & (CQ - 1);
        f32x4 g4 = zero4(), dp4 = zero4();
        if (has_se && c0 + 4 * mycq < mcp) {
            g4 = ld4(gate + (size_t)n * M + off + c0 + 4 * mycq);
            dp4 = ld4(dpooled + (size_t)n * M + off + c0 + 4 * mycq) * splat4(inv_hw);
        }
        // The (dZ, D) operands of this thread's FIRST item of the tile are requested before the E tile is staged: both are
        // plain global loads with ~2 us of latency at 2 waves per SIMD, and a tile has one item per thread (T0*T1/4 strips x CQ
        // = 256), so without this the two latencies were paid one after the other for every tile.
        const int nitems = nstrips * CQ;
        auto item_ops = [&](int item, f32x4 (&dd)[4], f32x4 (&dv)[4], bool (&ok)[4]) {
            const int cq = item & (CQ - 1), st = item >> gm.cq_shift;
            const int oh = st / nsw, ow0 = (st - oh * nsw) * 4;
            const int ho = ho0 + oh;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int wo = wo0 + ow0 + j;
                ok[j] = ho < Ho && wo < Wo && c0 + 4 * cq < mcp;
                const size_t a = ((size_t)(n * Ho + ho) * Wo + wo) * M + off + c0 + 4 * cq;
                dd[j] = ok[j] ? ldS4_nt(dZ, a, d.stor) : zero4();
                dv[j] = ok[j] ? ldS4_nt(D, a, d.stor) : zero4();
            }
        };
        f32x4 pdd[4], pdv[4];
        bool pok[4] = {false, false, false, false};
        if (tid < nitems) item_ops(tid, pdd, pdv, pok);
        __syncthreads();
        load_tile(in_tile, IH, IW, CC, gm.cq_shift, E,
                  [&](int r, int c, int cq, size_t& a) {
                      const int hi = hi0 + r, wi = wi0 + c;
                      a = ((size_t)(n * H + hi) * W + wi) * M + off + c0 + 4 * cq;
                      return hi >= 0 && hi < H && wi >= 0 && wi < W && c0 + 4 * cq < mcp;
                  },
                  [&](f32x4 v, int cq) {
#pragma unroll
                      for (int j = 0; j < 4; ++j) {
                          const float2 c = cst1[4 * cq + j];
                          v[j] = act_f<ACT>((v[j] - c.x) * c.y);
                      }
                      return v;
                  }, d.stor);
        __syncthreads();
        for (int item = tid; item < nitems; item += 256) {
            const int cq = item & (CQ - 1), st = item >> gm.cq_shift;
            const int oh = st / nsw, ow0 = (st - oh * nsw) * 4;
            f32x4 dd[4], dv[4];
            bool ok[4];
            if (item == tid) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    dd[j] = pdd[j];
                    dv[j] = pdv[j];
                    ok[j] = pok[j];
                }
            } else {
                item_ops(item, dd, dv, ok);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) dd[j] = ok[j] ? bn2_dd<ACT>(cst2, 4 * cq, dd[j], dv[j], has_se, g4, dp4) : zero4();
#pragma unroll
            for (int ky = 0; ky < K; ++ky) {
                const float* rowp = in_tile + ((oh * S + ky) * IW + ow0 * S) * CC + 4 * cq;
                f32x4 win[WIN];
#pragma unroll
                for (int u = 0; u < WIN; ++u) win[u] = ld4(rowp + u * CC);
#pragma unroll
                for (int kx = 0; kx < K; ++kx)
#pragma unroll
                    for (int j = 0; j < 4; ++j) wacc[ky * K + kx] += dd[j] * win[j * S + kx];
            }
        }
    }
    // reduce over the threads sharing a channel quad: first inside the wave, then across the 4 waves
    for (int o = CQ; o < 64; o <<= 1) {
#pragma unroll
        for (int u = 0; u < K * K; ++u) {
#pragma unroll
            for (int q = 0; q < 4; ++q) wacc[u][q] += __shfl_xor(wacc[u][q], o, 64);
        }
    }
    __syncthreads();
    float* wred = in_tile;   // [4 waves][K*K][CC]
    const int lane = tid & 63, wv = tid >> 6;
    if (lane < CQ) {
#pragma unroll
        for (int u = 0; u < K * K; ++u) st4(wred + (wv * K * K + u) * CC + 4 * lane, wacc[u]);
    }
    __syncthreads();
    for (int idx = tid; idx < K * K * CC; idx += 256) {
        const int cl = idx % CC, u = idx / CC;
        if (c0 + cl < mc) {
            float s = 0.f;
#pragma unroll
            for (int ww = 0; ww < 4; ++ww) s += wred[(ww * K * K + u) * CC + cl];
            gw[(size_t)(c0 + cl) * (K * K) + u] = s;
        }
    }
}

// ============================================================================ host side
// force32: the E-free producer (efree.h) works on 32-channel chunks
static void pick_tile(int N, int Th, int Tw, int K, int S, bool fwd_like, DwGeom& gm, bool force32 = false) {
    // T1 (width) multiple of 4 (strips), up to 16; T0 so that a tile has ~128 (64 for stride 2) pixels
    gm.T1 = Tw >= 16 ? 16 : ((Tw + 3) / 4) * 4;
    const int target = (S == 2 && fwd_like) ? 64 : 128;
    gm.T0 = target / gm.T1;
    if (gm.T0 > Th) gm.T0 = Th;
    if (gm.T0 < 1) gm.T0 = 1;
    gm.tilesH = cdiv(Th, gm.T0);
    gm.tilesW = cdiv(Tw, gm.T1);
    gm.ntiles = N * gm.tilesH * gm.tilesW;
    if (fwd_like) {
        gm.L0 = (gm.T0 - 1) * S + K;
        gm.L1 = (gm.T1 - 1) * S + K;
    } else {
        gm.L0 = (gm.T0 + K - 2) / S + 2;      // worst case over tile origins of floor((a+T+K-2)/S)-floor(a/S)+1
        gm.L1 = (gm.T1 + K - 2) / S + 2;
    }
    const int px = gm.L0 * gm.L1;
    const int items32 = gm.T0 * (gm.T1 / 4) * 8;
    gm.CC = 32;
    if (px * 32 * 4 > 56 * 1024) gm.CC = 16;
    else if (items32 < 192 && px * 64 * 4 <= 40 * 1024) gm.CC = 64;
    if (force32) gm.CC = 32;
    gm.cq_shift = gm.CC == 16 ? 2 : gm.CC == 32 ? 3 : 4;
}

static int dw_chunks(const TfnasCellDesc& d, int K, int CC) {
    int t = 0;
    for (int g = 0; g < d.G; ++g)
        if (d.g[g].k == K) t += cdiv(d.g[g].mcp, CC);
    return t;
}

static int dw_grid_x(const DwGeom& gm, int chunks, int target_blocks) {
    int gx = cdiv(target_blocks, chunks);
    if (gx > gm.ntiles) gx = gm.ntiles;
    return gx < 1 ? 1 : gx;
}

#define KQ_DISPATCH(kq, ...)                                      \
    switch (kq) {                                                 \
        case 0: { constexpr int KQ = 0; __VA_ARGS__; } break;     \
        case 4: { constexpr int KQ = 4; __VA_ARGS__; } break;     \
        case 6: { constexpr int KQ = 6; __VA_ARGS__; } break;     \
        case 10: { constexpr int KQ = 10; __VA_ARGS__; } break;   \
        default: return TFNAS_EINVAL;                             \
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

// Both kernel-size passes of one stage share grid.x so that they fill the same rows of the partials matrix.
static int dw_common_gx(const TfnasCellDesc& d, int Th, int Tw, bool fwd_like, int target_blocks, size_t row_floats,
                        bool force32 = false) {
    int gx = 1 << 30;
    for (int kk = 3; kk <= 5; kk += 2) {
        DwGeom gm;
        pick_tile(d.N, Th, Tw, kk, d.stride, fwd_like, gm, force32);
        const int chunks = dw_chunks(d, kk, gm.CC);
        if (!chunks) continue;
        const int g1 = dw_grid_x(gm, chunks, target_blocks);
        if (g1 < gx) gx = g1;
    }
    const size_t cap = TFNAS_PART_FLOATS / (row_floats ? row_floats : 1);
    if ((size_t)gx > cap) gx = (int)cap;
    if (gx > 1024) gx = 1024;                      // partial rows to reduce afterwards
    if (gx >= 8) gx &= ~7;                         // multiple of 8 for the XCD-aware tile order
    return gx < 1 ? 1 : gx;
}

// E == nullptr: E-free mode (the tile is recomputed from x, efree.h)
static int launch_dw_fwd_tiled(const TfnasCellDesc& d, const float* E, const float* x, const double* stats1, float* D,
                               double* stats2, float* part, hipStream_t s) {
    const bool ef = E == nullptr;
    if (ef && !efree_ic_ok(d.ic)) return TFNAS_EINVAL;
    const int kq = ef ? d.ic / 4 : 0;
    const int gx = dw_common_gx(d, d.Ho, d.Wo, true, 4096, 2 * (size_t)d.M, ef);
    for (int kk = 3; kk <= 5; kk += 2) {
        DwGeom gm;
        pick_tile(d.N, d.Ho, d.Wo, kk, d.stride, true, gm, ef);
        const int chunks = dw_chunks(d, kk, gm.CC);
        if (!chunks) continue;
        const int tile = gm.L0 * gm.L1 * gm.CC > 2048 ? gm.L0 * gm.L1 * gm.CC : 2048;
        const size_t shm = (size_t)(tile + kk * kk * gm.CC + 2 * gm.CC) * sizeof(float);
        dim3 grid(gx, chunks);
        ProfScope _prof(TK_DW_FWD, s, d.G > 2);
        if ((size_t)shm > 64 * 1024) return TFNAS_ERANGE;
        DW_DISPATCH(kk, d.stride, d.act, KQ_DISPATCH(kq, {
            hipLaunchKernelGGL((k_dw_fwd<K, S, ACT, KQ>), grid, dim3(256), shm, s, d, E, x, stats1, D, part, gm);
        }))
    }
    return launch_reduce_rows(part, gx, 2 * d.M, 2 * (size_t)d.M, stats2, nullptr, s);
}

static int launch_dw_bwd_data_tiled(const TfnasCellDesc& d, const float* dZ, const float* gate, const float* dpooled,
                       const float* D, const double* stats2,
                       const double* red2, const float* E, const float* x, const double* stats1, float* dEh, double* red1,
                       float* part, hipStream_t s, float* cb1) {
    const bool ef = E == nullptr;
    if (ef && !efree_ic_ok(d.ic)) return TFNAS_EINVAL;
    const int kq = ef ? d.ic / 4 : 0;
    const int gx = dw_common_gx(d, d.H, d.W, false, 4096, 2 * (size_t)d.M, ef);
    for (int kk = 3; kk <= 5; kk += 2) {
        DwGeom gm;
        pick_tile(d.N, d.H, d.W, kk, d.stride, false, gm, ef);
        const int chunks = dw_chunks(d, kk, gm.CC);
        if (!chunks) continue;
        const int tile = gm.L0 * gm.L1 * gm.CC > 2048 ? gm.L0 * gm.L1 * gm.CC : 2048;
        const size_t shm = (size_t)(tile + kk * kk * gm.CC + 4 * gm.CC + 2 * gm.CC + (ef ? gm.T0 * gm.T1 * gm.CC : 0)) *
                           sizeof(float);
        if ((size_t)shm > 64 * 1024) return TFNAS_ERANGE;
        dim3 grid(gx, chunks);
        ProfScope _prof(TK_DW_BWD_DATA, s, d.G > 2);
        DW_DISPATCH(kk, d.stride, d.act, KQ_DISPATCH(kq, {
            hipLaunchKernelGGL((k_dw_bwd_data<K, S, ACT, KQ>), grid, dim3(256), shm, s, d, dZ, gate, dpooled, D, stats2,
                               red2, E, x, stats1, dEh, part, gm);
        }))
    }
    if (cb1) return launch_reduce_bn1(d, part, gx, stats1, red1, cb1, s);
    return launch_reduce_rows(part, gx, 2 * d.M, 2 * (size_t)d.M, red1, nullptr, s);
}

static int launch_dw_wgrad_tiled(const TfnasCellDesc& d, const float* dZ, const float* gate, const float* dpooled, const float* D,
                    const double* stats2,
                    const double* red2, const float* E, const double* stats1, float* part, hipStream_t s) {
    size_t out_size = 0;
    for (int g = 0; g < d.G; ++g) out_size += (size_t)d.g[g].mc * d.g[g].k * d.g[g].k;
    const int gx = dw_common_gx(d, d.Ho, d.Wo, true, 2048, out_size);
    for (int kk = 3; kk <= 5; kk += 2) {
        DwGeom gm;
        pick_tile(d.N, d.Ho, d.Wo, kk, d.stride, true, gm);
        const int chunks = dw_chunks(d, kk, gm.CC);
        if (!chunks) continue;
        int tile = gm.L0 * gm.L1 * gm.CC;
        if (tile < 4 * kk * kk * gm.CC) tile = 4 * kk * kk * gm.CC;
        const size_t shm = (size_t)(tile + 4 * gm.CC + 2 * gm.CC) * sizeof(float);
        dim3 grid(gx, chunks);
        ProfScope _prof(TK_DW_WGRAD, s);
        DW_DISPATCH(kk, d.stride, d.act, {
            hipLaunchKernelGGL((k_dw_wgrad<K, S, ACT>), grid, dim3(256), shm, s, d, dZ, gate, dpooled, D, stats2, red2, E, stats1,
                               part, out_size, gm);
        })
    }
    size_t poff = 0;
    for (int g = 0; g < d.G; ++g) {
        const int n = d.g[g].mc * d.g[g].k * d.g[g].k;
        int rc = launch_reduce_rows(part + poff, gx, n, out_size, nullptr, d.g[g].g_dw, s);
        if (rc) return rc;
        poff += n;
    }
    return (int)hipGetLastError();
}

static int launch_dw_wgrad_direct(const TfnasCellDesc& d, const float* dZ, const float* gate, const float* dpooled,
                                  const float* D, const double* stats2, const double* red2, const float* E,
                                  const double* stats1, float* part, size_t out_size, hipStream_t s, bool& done);
// TfnasCellDesc.route, TFNAS_ROUTE_DW_*: 0 per launch, whichever kernel measured faster | 1 register-window kernels wherever the
// geometry allows | 2 ring / tile kernels only | 3 tile kernels only: every choice is compared with the oracle
// (tests/test_gpu_cell.py::test_variant_against_oracle)
static inline int dw_variant(const TfnasCellDesc& d) { return route_dw(d); }
static bool dwd_enabled(const TfnasCellDesc& d);
static bool dwd_bwd_use(const TfnasCellDesc& d);
static int launch_dw_bwd_data_direct(const TfnasCellDesc& d, const float* dZ, const float* gate, const float* dpooled,
                                     const float* D, const double* stats2, const double* red2, const float* E,
                                     const double* stats1, float* dEh, double* red1, float* part, hipStream_t s, float* cb1,
                                     bool& done, bool fuse_wgrad = false);
static bool dwd_bwd_fuses_wgrad(const TfnasCellDesc& d);
static int launch_dw_fwd_direct(const TfnasCellDesc& d, const float* E, const double* stats1, float* D, double* stats2,
                                float* part, hipStream_t s, bool& done);

#include "dw_stream.inc"
#include "dw_direct.inc"
