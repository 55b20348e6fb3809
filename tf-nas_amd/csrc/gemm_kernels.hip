// 1x1-convolution (pointwise) kernels of the MBConv candidates as fp32 MFMA GEMMs with fused
// BatchNorm / activation / SE prologues and BatchNorm-statistics epilogues.
//
// Reference arithmetic: MBInvertedResBlock.forward, models/layers.py:539-561
//   expand  = inverted_bottleneck.conv (layers.py:463-478)   project = point_linear.conv (layers.py:528-534)
// and the autograd backward of those convolutions + BatchNorm2d(affine=False, batch stats).
#include <stdlib.h>
#include <string.h>
#include "gemm_core.h"
#include "gemm_x3.h"
#include "kernels.h"
#include "prof.h"

// ---------------------------------------------------------------------------- stem im2col operand
// TFNAS_MODE_STEM: the "expand" is first_stem's 3x3 stride-2 pad-1 convolution of the 3-channel NCHW image
// (models/model_search.py:219).  It runs through the same GEMM kernels with a gathering operand loader:
// A(p, t) = img[n][t/9][2*ho + (t%9)/3 - 1][2*wo + t%3 - 1]  (t < 27 = im2col depth, OIHW weight order).
__device__ __forceinline__ f32x4 stem_patch4(const float* __restrict__ img, const TfnasCellDesc& d, int p, int t0) {
    const int HW = d.H * d.W, n = p / HW, r = p - n * HW, ho = r / d.W, wo = r - ho * d.W;
    f32x4 v = zero4();
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int t = t0 + j;
        if (t < 27) {
            const int ci = t / 9, rem = t - 9 * ci, ky = rem / 3, kx = rem - 3 * ky;
            const int hi = 2 * ho + ky - 1, wi = 2 * wo + kx - 1;
            if (hi >= 0 && hi < d.Hi && wi >= 0 && wi < d.Wi) v[j] = img[((size_t)(n * 3 + ci) * d.Hi + hi) * d.Wi + wi];
        }
    }
    return v;
}

// resident workgroups per CU the row-tiled GEMM kernels are compiled for (register cap 512 / n per lane): 5-/7-tile and
// narrower variants
#ifndef TFNAS_WGRAD_PF2
#define TFNAS_WGRAD_PF2 false  /* prefetch distance 2 in the weight-gradient GEMMs (A/B) */
#endif
#ifndef TFNAS_LB_BIG
#define TFNAS_LB_BIG 3
#endif
#ifndef TFNAS_LB_SMALL
#define TFNAS_LB_SMALL 4
#endif
// the split-bf16 loop (MM != 0) keeps two chunks of raw operands and three A planes in registers: one resident workgroup less
#ifndef TFNAS_LB_X3_BIG
#define TFNAS_LB_X3_BIG 2
#endif
#ifndef TFNAS_LB_X3_SMALL
#define TFNAS_LB_X3_SMALL 3
#endif
// (heavy: k_project_dgrad, whose BN3-backward transform keeps 40 table values per chunk pair live: 217-249 registers)
constexpr int gemm_lb(int nt, int mm, bool heavy = false) {
    return mm == 0 ? (nt >= 5 ? TFNAS_LB_BIG : TFNAS_LB_SMALL) : ((nt >= 5 || heavy) ? TFNAS_LB_X3_BIG : TFNAS_LB_X3_SMALL);
}

// -DTFNAS_WG_TIMING (tools/wg_timeline.py): per-workgroup wall-clock stamps (100 MHz s_memrealtime) of the weight-gradient
// GEMMs: [wg][0..3] = start, after prologue, after K loop, end
#ifdef TFNAS_WG_TIMING
__device__ unsigned long long g_wgt[4 * 16384];
#define WGT(slot)                                                                                                      \
    if (threadIdx.x == 0) {                                                                                            \
        const unsigned f_ = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);                            \
        if (f_ < 16384) g_wgt[4 * f_ + (slot)] = wall_clock64();                                                       \
    }
extern "C" int tfnas_dbg_wg_timing(unsigned long long* out, int n) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wgt), sizeof(unsigned long long) * (size_t)n);
}
#else
#define WGT(slot)
#endif

// raw registers of the two-phase loaders (gemm_core.h): what the load phase leaves for the transform phase
struct Raw2 { f32x4 a, b; };             // two stream pieces (D | gate,  dOut | Pr,  dEh | E)
struct RawWS { f32x4 w; float s; };      // a weight quad and its row scale

// ============================================================================ expand forward
// E[p][off_g + m] = sum_c x[p][c] * w_expand_g[m][c]      for all groups in one launch
// epilogue: per-workgroup partial (sum, sumsq) of E per channel -> part (reduced into stats1 = BN1 statistics)
template <int NT, bool STEM, int MM>
__global__ __launch_bounds__(256, gemm_lb(NT, MM)) void k_expand_fwd(TfnasCellDesc d, const float* __restrict__ x,
                                                    float* __restrict__ E, float* __restrict__ part) {
    using T = GT<NT>;
    __shared__ __attribute__((aligned(16))) float lds[T::LDS_FLOATS];
    // all-candidate launches: consecutive workgroups (dispatch order) take consecutive COLUMN tiles of one row block -- the 256-byte
    // row pieces they store are then neighbours in memory instead of 4 * M bytes apart (measured alone, B = 128: cell 10 220 -> 190 us,
    // cell 3 186 -> 167 us, cell 1 443 -> 424 us; one-candidate launches 3-10 % slower that way and keep the row-block-fastest order)
    const int L_ = blockIdx.x + gridDim.x * blockIdx.y;
    const bool ctf_ = d.G > 1;
    const int BX = ctf_ ? L_ / (int)gridDim.y : (int)blockIdx.x;
    int ty = ctf_ ? L_ % (int)gridDim.y : (int)blockIdx.y, g = 0;
    for (; g < d.G - 1; ++g) {
        const int t = (d.g[g].mcp + T::BN - 1) / T::BN;
        if (ty < t) break;
        ty -= t;
    }
    const int mc = d.g[g].mc, mcp = d.g[g].mcp, off = d.g[g].off;
    const float* __restrict__ w = d.g[g].w_expand;
    const int n0 = ty * T::BN;
    const int P = d.N * d.H * d.W, ic = d.ic, M = d.M;
    const int nrt = (P + 127) >> 7, nchunks = (ic + 15) >> 4;
    const int tid = threadIdx.x, lr = tid & 15, wrow = (tid >> 6) * 32;

    float cs[NT], cq[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) cs[j] = cq[j] = 0.f;

    for (int rt = BX; rt < nrt; rt += gridDim.x) {
        f32x4 acc[2][NT];
        acc_zero<NT>(acc);
        // the lane's two A rows stay the same for the whole K loop: row pointers and validity live in registers
        int prow[2];
        const float* arow[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            prow[i] = rt * 128 + wrow + 16 * i + lr;
            arow[i] = x + (size_t)min(prow[i], P - 1) * ic;
        }
        auto la = [&](int c, int i, int kl) -> f32x4 {
            const int k = c * 16 + kl;
            if (STEM) return (prow[i] < P && k < ic) ? stem_patch4(x, d, prow[i], k) : zero4();
            return ld4(arow[i] + min(k, ic - 4));
        };
        auto xa = [&](f32x4 r, int c, int i, int kl) -> f32x4 {
            if (STEM) return r;
            return (prow[i] < P && c * 16 + kl < ic) ? r : zero4();
        };
        auto lb = [&](int c, int n, int kl) -> f32x4 {
            const int col = n0 + n, k = c * 16 + kl;
            if (STEM) return (col < mc) ? ld4_guard(w + (size_t)col * ic, k, ic, false) : zero4();
            return ld4(w + (size_t)min(col, mc - 1) * ic + min(k, ic - 4));        // ic % 4 == 0: k < ic <=> k + 3 < ic
        };
        auto xb = [&](f32x4 r, int c, int n, int kl) -> f32x4 {
            if (STEM) return r;
            return (n0 + n < mc && c * 16 + kl < ic) ? r : zero4();
        };
        PreNone pre;
        gemm_adirect<NT, true, MM>(pre, la, xa, lb, xb, nchunks, acc, lds);
        emit_tile_rows<NT>(acc, lds, [&](int lrow, int lc, f32x4 v) {
            const int p = rt * 128 + lrow;
            if (p < P && n0 + lc < mcp) stS4_nt(E, (size_t)p * M + off + n0 + lc, v, d.stor);
        });
        acc_colstats<NT>(acc, cs, cq);
    }
    flush_colstats<NT>(cs, cq, lds, part + (size_t)BX * 2 * M + 2 * (size_t)off, n0, mcp);
}

// ============================================================================ project forward
// Pr[g][p][o] = sum_c z_g[p][c] * w_proj_g[o][c],   z = act(BN2(D)) * gate   (fused into the A load)
// epilogue: stats3[g][o] (BN3 batch statistics)
// (launch bounds: the 7-tile variant otherwise takes 149 VGPRs + 56 AGPRs = 2 waves/SIMD, measured 1.5 resident; capping
//  it at 168 registers buys the third wave: -17 % on the 112-channel cells)
template <int NT, int ACT, int MM>
__global__ __launch_bounds__(256, gemm_lb(NT, MM)) void k_project_fwd(TfnasCellDesc d, const float* __restrict__ D,
                                                     const float* __restrict__ gate,
                                                     const double* __restrict__ stats2, float* __restrict__ Pr,
                                                     float* __restrict__ part, int nsplit, float* __restrict__ prp) {
    using T = GT<NT>;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    // nsplit > 1 (under-filled launches: few row tiles, long K): K-split ks of group g adds chunks [cb, ce); split 0
    // writes Pr, split z > 0 its partial tile to prp[z-1]; k_pr_reduce sums them and takes the BN3 statistics
    // blockIdx.z -> group in order of DEcreasing K (= mid width): a workgroup's run time is proportional to its group's K
    // and the launch is 1-2 rounds of resident workgroups, so the long ones must not be the last to start (with the
    // natural order the 7x7 cells dispatched the widest group last: 1.5 instead of 1.0 long-workgroup times)
    const int zr = blockIdx.z / nsplit, ks = blockIdx.z - zr * nsplit;
    int g = 0;
    for (int c = 0; c < d.G; ++c) {
        int rank = 0;
        for (int o = 0; o < d.G; ++o) rank += (d.g[o].mcp > d.g[c].mcp) || (d.g[o].mcp == d.g[c].mcp && o < c);
        if (rank == zr) g = c;
    }
    const int mc = d.g[g].mc, mcp = d.g[g].mcp, off = d.g[g].off;
    const bool has_se = d.g[g].se > 0, w_al = (mc & 3) == 0;
    const float* gbase = has_se ? gate : D;        // the gate load is unconditional (see gemm_core.h: no branches
    const size_t se01 = has_se ? 1 : 0;            // around the loads of the prefetch phase)
    const float* __restrict__ w = d.g[g].w_proj;
    const int n0 = blockIdx.y * T::BN;
    const int HW = d.Ho * d.Wo, Po = d.N * HW, oc = d.oc, M = d.M;
    const int nrt = (Po + 127) >> 7, nchunks_all = (mcp + 15) >> 4;
    const int per = (nchunks_all + nsplit - 1) / nsplit, cb = ks * per;
    const int nchunks = max(0, min(nchunks_all, cb + per) - cb);
    float* __restrict__ dst = ks == 0 ? Pr : prp + (size_t)(ks - 1) * d.G * Po * oc;
    const int tid = threadIdx.x, lr = tid & 15, wrow = (tid >> 6) * 32;

    float2* cst = reinterpret_cast<float2*>(lds + T::LDS_FLOATS);
    for (int c = tid; c < ((mcp + 15) & ~15); c += 256)
        cst[c] = (c < mc) ? bn_consts(stats2 + 2 * (size_t)(off + c), 1.0 / (double)Po, d.eps) : make_float2(0.f, 0.f);
    __syncthreads();

    float cs[NT], cq[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) cs[j] = cq[j] = 0.f;

    for (int rt = blockIdx.x; rt < nrt; rt += gridDim.x) {
        f32x4 acc[2][NT];
        acc_zero<NT>(acc);
        // the lane's two A rows stay the same for the whole K loop: element offsets of the rows in D and in the gate table
        // (one division per row tile instead of one per chunk) and validity live in registers
        size_t arow[2], grow[2];
        bool rok[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int p = rt * 128 + wrow + 16 * i + lr, pc = min(p, Po - 1);
            rok[i] = p < Po;
            arow[i] = (size_t)pc * M + off;
            grow[i] = ((size_t)(pc / HW) * M + off) * se01;
        }
        auto la = [&](int c, int i, int kl) -> Raw2 {
            const int k = min((cb + c) * 16 + kl, mcp - 4);
            Raw2 r;
            r.a = ldS4_raw(D, arow[i] + k, d.stor);
            r.b = ld4(gbase + grow[i] + k * se01);                // no SE: one fixed (ignored) quad
            return r;
        };
        auto xa = [&](Raw2 r, int c, int i, int kl) -> f32x4 {
            const int k = (cb + c) * 16 + kl;
            f32x4 v = ldS4_fin(r.a, d.stor);
            const int kc = min(k, mcp - 4);
            const float2 c0 = cst[kc], c1 = cst[kc + 1], c2 = cst[kc + 2], c3 = cst[kc + 3];
            v.x = act_f<ACT>((v.x - c0.x) * c0.y);
            v.y = act_f<ACT>((v.y - c1.x) * c1.y);
            v.z = act_f<ACT>((v.z - c2.x) * c2.y);
            v.w = act_f<ACT>((v.w - c3.x) * c3.y);
            if (has_se) v *= r.b;
            return (rok[i] && k < mcp) ? v : zero4();
        };
        auto lb = [&](int c, int n, int kl) -> f32x4 {
            const int o = n0 + n, k = (cb + c) * 16 + kl;
            if (MM == 0 && !w_al) return (o < oc) ? ld4_guard(w + (size_t)o * mc, k, mc, false) : zero4();
            return ld4(w + (size_t)min(o, oc - 1) * mc + min(k, mc - 4));          // mc % 4 == 0: k < mc <=> k + 3 < mc
        };
        auto xb = [&](f32x4 r, int c, int n, int kl) -> f32x4 {
            if (MM == 0 && !w_al) return r;
            return (n0 + n < oc && (cb + c) * 16 + kl < mc) ? r : zero4();
        };
        PreNone pre;
        gemm_adirect<NT, true, MM>(pre, la, xa, lb, xb, nchunks, acc, lds);
        emit_tile_rows<NT>(acc, lds, [&](int lrow, int lc, f32x4 v) {
            const int p = rt * 128 + lrow;
            if (p < Po && n0 + lc < oc) st4(dst + ((size_t)g * Po + p) * oc + n0 + lc, v);
        });
        acc_colstats<NT>(acc, cs, cq);
    }
    if (nsplit == 1)
        flush_colstats<NT>(cs, cq, lds, part + (size_t)blockIdx.x * 2 * d.G * oc + 2 * (size_t)g * oc, n0, oc);
}

// Pr[g][p][:] += sum_z prp[z][g][p][:] and the per-workgroup partial (sum, sumsq) of the finished Pr per (g, column):
// part[bx][2*(g*oc + o) + {0,1}] (the layout of k_project_fwd's own statistics epilogue).  Flat float4 stream; a
// thread's elements are `stride` float4 apart with stride a multiple of oc/4, so its column quad never changes.
__global__ __launch_bounds__(256) void k_pr_reduce(float* __restrict__ Pr, const float* __restrict__ prp, int nz, int G,
                                                   int Po, int oc, int rps, float* __restrict__ part) {
    __shared__ f32x4 rs[256], rq[256];
    const int tid = threadIdx.x, ocq = oc >> 2, stride = (256 / ocq) * ocq;
    const int g = blockIdx.y;
    const int r0 = blockIdx.x * rps, r1 = min(Po, r0 + rps);
    const size_t base = ((size_t)g * Po + r0) * ocq, zs = (size_t)G * Po * ocq;
    f32x4* __restrict__ p4 = reinterpret_cast<f32x4*>(Pr) + base;
    const f32x4* __restrict__ z4 = reinterpret_cast<const f32x4*>(prp) + base;
    const int n4 = (r1 - r0) * ocq;
    f32x4 s = zero4(), q = zero4();
    if (tid < stride) {
        for (int i = tid; i < n4; i += stride) {
            f32x4 v = p4[i];
            for (int z = 0; z < nz; ++z) v += z4[(size_t)z * zs + i];
            p4[i] = v;
            s += v;
            q += v * v;
        }
    }
    rs[tid] = s;
    rq[tid] = q;
    __syncthreads();
    for (int c = tid; c < oc; c += 256) {
        const int cq = c >> 2, comp = c & 3;
        float ts = 0.f, tq = 0.f;
        for (int k = cq; k < stride; k += ocq) {
            ts += rs[k][comp];
            tq += rq[k][comp];
        }
        float* row = part + (size_t)blockIdx.x * 2 * G * oc + 2 * ((size_t)g * oc + c);
        row[0] = ts;
        row[1] = tq;
    }
}

// ---------------------------------------------------------------------------- BN3-backward operand
// dP_g[p][o] = A3[o] * (dOut[p][o] - B3[o] - phat[p][o]*C3[o]),  phat = (Pr - mean3)*rstd3,
// A3 = wmix_g * rstd3, B3 = S1/Po, C3 = S2/Po  (S1,S2 = red3 sums of dOut and dOut*phat)
struct Bn3Tab {  // 5 floats per output channel, in LDS
    float *mean, *rstd, *a3, *b3, *c3;
};
__device__ __forceinline__ Bn3Tab bn3_tab_fill(float* base, int ocp, const TfnasCellDesc& d, int g,
                                               const double* stats3, const double* red3, const float* wmix) {
    Bn3Tab t{base, base + ocp, base + 2 * ocp, base + 3 * ocp, base + 4 * ocp};
    const int Po = d.N * d.Ho * d.Wo;
    const double inv = 1.0 / (double)Po;
    const float wg = wmix ? wmix[g] : 1.f;
    for (int o = threadIdx.x; o < ocp; o += blockDim.x) {
        if (o < d.oc) {
            const float2 c = bn_consts(stats3 + 2 * ((size_t)g * d.oc + o), inv, d.eps);
            t.mean[o] = c.x;
            t.rstd[o] = c.y;
            t.a3[o] = wg * c.y;
            t.b3[o] = (float)(red3[2 * ((size_t)g * d.oc + o) + 0] * inv);
            t.c3[o] = (float)(red3[2 * ((size_t)g * d.oc + o) + 1] * inv);
        } else {
            t.mean[o] = t.rstd[o] = t.a3[o] = t.b3[o] = t.c3[o] = 0.f;
        }
    }
    return t;
}
__device__ __forceinline__ f32x4 bn3_dp(const Bn3Tab& t, int o, f32x4 dout, f32x4 pr) {
    f32x4 r;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float ph = (pr[j] - t.mean[o + j]) * t.rstd[o + j];
        r[j] = t.a3[o + j] * (dout[j] - t.b3[o + j] - ph * t.c3[o + j]);
    }
    return r;
}

// Folded form for k_project_dgrad: one float4 (mean, a3, a3 b3, rstd c3 a3) per output channel,
//   dP = a3 dOut - a3 b3 - (Pr - mean) (rstd c3 a3)      (the form k_project_wgrad keeps in registers)
// -- one ds_read_b128 and three VALU operations per element instead of five table reads and six operations.
__device__ __forceinline__ void bn3_fold_fill(f32x4* tab, int ocp, const TfnasCellDesc& d, int g, const double* stats3,
                                              const double* red3, const float* wmix) {
    const int Po = d.N * d.Ho * d.Wo;
    const double inv = 1.0 / (double)Po;
    const float wg = wmix ? wmix[g] : 1.f;
    for (int o = threadIdx.x; o < ocp; o += blockDim.x) {
        f32x4 t = zero4();
        if (o < d.oc) {
            const float2 c = bn_consts(stats3 + 2 * ((size_t)g * d.oc + o), inv, d.eps);
            const float a3 = wg * c.y;
            const float b3 = (float)(red3[2 * ((size_t)g * d.oc + o) + 0] * inv);
            const float c3 = (float)(red3[2 * ((size_t)g * d.oc + o) + 1] * inv);
            t = f32x4{c.x, a3, a3 * b3, c.y * c3 * a3};
        }
        tab[o] = t;
    }
}
__device__ __forceinline__ f32x4 bn3_fold_dp(const f32x4* tab, int o, f32x4 dout, f32x4 pr) {
    f32x4 r;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const f32x4 t = tab[o + j];
        r[j] = (t.y * dout[j] - t.z) - (pr[j] - t.x) * t.w;
    }
    return r;
}

// ---------------------------------------------------------------------------- BN2-backward sums in the dgrad epilogue
// FOLD variant of k_project_dgrad (round 4): the per-image tables that k_bn2_pool computed in its own pass
// over (dZ, D) -- A1 = sum dZ act'(dhat), B1 = sum dZ act'(dhat) dhat, dgate = sum dZ act(dhat), A2 = sum act'(dhat),
// B2 = sum act'(dhat) dhat over the pixels of one image (pointwise_kernels.hip) -- are accumulated while the dZ tile is still in
// LDS: dZ is then written once and never read back for them (the D read stays).  A lane owns one or two COLUMNS of the tile and
// walks the 16 rows of its wave's slab: column-contiguous dword loads of D (256 B per row and wave), no cross-lane reduction.
// A wave's 32 rows touch at most two images (HW >= 43), a 128-row tile at most four: one record per (row tile, image slot)
//   rec[((rt * 4 + slot) * 5 + q) * M + channel],  slot = image - image_of(rt * 128),  q = A1 | B1 | dgate | A2 | B2
// each written by exactly one workgroup (no atomics, fixed summation order); k_bn2_gather sums an image's records.
template <int NT, int ACT>
__device__ __forceinline__ void fold_slab(const float* st, int LDC, const float* __restrict__ Dcol, int M, int p0, int Po, int bnd,
                                          const float2 (&c2)[(16 * NT + 63) / 64], const bool (&cok)[(16 * NT + 63) / 64],
                                          bool has_se, float (&sum)[2][FOLD_Q][(16 * NT + 63) / 64]) {
    constexpr int NC = (16 * NT + 63) / 64;
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        if (!cok[c]) continue;
        const int col = lane + 64 * c;
        float dv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) dv[r] = __builtin_nontemporal_load(Dcol + (size_t)min(p0 + r, Po - 1) * M + col);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int p = p0 + r;
            if (p >= Po) break;                                     // (wave-uniform)
            const float z = st[r * LDC + col];
            const float dh = (dv[r] - c2[c].x) * c2[c].y;
            const float ad = act_d<ACT>(dh), zd = z * ad;
            const float v2 = has_se ? z * act_f<ACT>(dh) : 0.f;
            if (p >= bnd) {                                         // (wave-uniform: the row belongs to the wave's second image)
                sum[1][0][c] += zd;
                sum[1][1][c] += zd * dh;
                if (has_se) {
                    sum[1][2][c] += v2;
                    sum[1][3][c] += ad;
                    sum[1][4][c] += ad * dh;
                }
            } else {
                sum[0][0][c] += zd;
                sum[0][1][c] += zd * dh;
                if (has_se) {
                    sum[0][2][c] += v2;
                    sum[0][3][c] += ad;
                    sum[0][4][c] += ad * dh;
                }
            }
        }
    }
}

// ============================================================================ project dgrad
// dZ[p][off_g + c] = sum_o dP_g[p][o] * w_proj_g[o][c]
template <int NT, int MM, bool FOLD = false>
__global__ __launch_bounds__(256, gemm_lb(NT, MM)) void k_project_dgrad(TfnasCellDesc d, const float* __restrict__ dout,
                                                       const float* __restrict__ Pr,
                                                       const double* __restrict__ stats3,
                                                       const double* __restrict__ red3,
                                                       const float* __restrict__ wmix, float* __restrict__ dZ,
                                                       const float* __restrict__ D = nullptr,
                                                       const double* __restrict__ stats2 = nullptr,
                                                       float* __restrict__ rec = nullptr) {
    using T = GT<NT>;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    // (all-candidate launches: column tiles fastest, as in k_expand_fwd -- cell 10 414 -> 376 us, cell 15 232 -> 213 us)
    const int L_ = blockIdx.x + gridDim.x * blockIdx.y;
    const bool ctf_ = d.G > 1;
    const int BX = ctf_ ? L_ / (int)gridDim.y : (int)blockIdx.x;
    int ty = ctf_ ? L_ % (int)gridDim.y : (int)blockIdx.y, g = 0;
    for (; g < d.G - 1; ++g) {
        const int t = (d.g[g].mcp + T::BN - 1) / T::BN;
        if (ty < t) break;
        ty -= t;
    }
    const int mc = d.g[g].mc, mcp = d.g[g].mcp, off = d.g[g].off;
    const bool w_al = (mc & 3) == 0;
    const float* __restrict__ w = d.g[g].w_proj;
    const int n0 = ty * T::BN;
    const int Po = d.N * d.Ho * d.Wo, oc = d.oc, M = d.M;
    const int ocp = (oc + 15) & ~15;
    const int nrt = (Po + 127) >> 7, nchunks = ocp >> 4;
    const int tid = threadIdx.x, lr = tid & 15, wrow = (tid >> 6) * 32;

    f32x4* tab = reinterpret_cast<f32x4*>(lds + T::LDS_FLOATS);
    bn3_fold_fill(tab, ocp, d, g, stats3, red3, wmix);
    // FOLD: BN2 constants of this lane's one or two columns of the (fixed) column tile
    constexpr int NC = (T::BN + 63) / 64;
    float2 c2f[NC];
    bool cokf[NC];
    const int HW = d.Ho * d.Wo;
    const bool has_se = d.g[g].se > 0;
    if (FOLD) {
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int col = (tid & 63) + 64 * c;
            cokf[c] = col < T::BN && n0 + col < mcp;
            c2f[c] = (cokf[c] && n0 + col < mc) ? bn_consts(stats2 + 2 * (size_t)(off + n0 + col), 1.0 / (double)Po, d.eps)
                                               : make_float2(0.f, 0.f);
        }
    }
    __syncthreads();

    for (int rt = BX; rt < nrt; rt += gridDim.x) {
        f32x4 acc[2][NT];
        acc_zero<NT>(acc);
        const float* drow[2];
        const float* qrow[2];
        bool rok[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int p = rt * 128 + wrow + 16 * i + lr, pc = min(p, Po - 1);
            rok[i] = p < Po;
            drow[i] = dout + (size_t)pc * oc;
            qrow[i] = Pr + ((size_t)g * Po + pc) * oc;
        }
        auto la = [&](int c, int i, int kl) -> Raw2 {
            const int o = min(c * 16 + kl, oc - 4);
            Raw2 r;
            r.a = ld4(drow[i] + o);
            r.b = ld4(qrow[i] + o);
            return r;
        };
        auto xa = [&](Raw2 r, int c, int i, int kl) -> f32x4 {
            const int o = c * 16 + kl;
            const f32x4 v = bn3_fold_dp(tab, min(o, oc - 4), r.a, r.b);
            return (rok[i] && o < oc) ? v : zero4();
        };
        auto lb = [&](int c, int kl, int n) -> f32x4 {
            const int o = c * 16 + kl;
            if (MM == 0 && !w_al) return (o < oc) ? ld4_guard(w + (size_t)o * mc, n0 + n, mc, false) : zero4();
            return ld4(w + (size_t)min(o, oc - 1) * mc + min(n0 + n, mc - 4));
        };
        auto xb = [&](f32x4 r, int c, int kl, int n) -> f32x4 {
            if (MM == 0 && !w_al) return r;
            return (c * 16 + kl < oc && n0 + n < mc) ? r : zero4();
        };
        PreNone pre;
        gemm_adirect<NT, false, MM>(pre, la, xa, lb, xb, nchunks, acc, lds);
        if (!FOLD) {
            emit_tile_rows<NT>(acc, lds, [&](int lrow, int lc, f32x4 v) {
                const int p = rt * 128 + lrow;
                if (p < Po && n0 + lc < mcp) stS4_nt(dZ, (size_t)p * M + off + n0 + lc, v, d.stor);
            });
        } else {
            constexpr int LDC = T::BN + 4;
            const int w = tid >> 6, lane = tid & 63;
            const int pw0 = rt * 128 + w * 32;
            const int imgA = min(pw0, Po - 1) / HW, bnd = (imgA + 1) * HW;
            float sum[2][FOLD_Q][NC];
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int q = 0; q < FOLD_Q; ++q)
#pragma unroll
                    for (int c = 0; c < NC; ++c) sum[a][q][c] = 0.f;
            const float* Dcol = D + off + n0;
            emit_tile_rows<NT>(acc, lds,
                               [&](int lrow, int lc, f32x4 v) {
                                   const int p = rt * 128 + lrow;
                                   if (p < Po && n0 + lc < mcp) stS4_nt(dZ, (size_t)p * M + off + n0 + lc, v, d.stor);
                               },
                               [&](int i, const float* st) {        // the wave's slab i (16 rows x BN) is still in LDS
                                   if (d.act == TFNAS_ACT_RELU)
                                       fold_slab<NT, TFNAS_ACT_RELU>(st, LDC, Dcol, M, pw0 + 16 * i, Po, bnd, c2f, cokf, has_se, sum);
                                   else
                                       fold_slab<NT, TFNAS_ACT_SWISH>(st, LDC, Dcol, M, pw0 + 16 * i, Po, bnd, c2f, cokf, has_se, sum);
                               });
            // waves -> image slots of the tile, in wave order (emit_tile_rows ended with a barrier: the GEMM LDS is free)
            static_assert(4 * 2 * FOLD_Q * T::BN + 4 <= T::LDS_FLOATS, "fold staging does not fit the GEMM LDS buffer");
            float* comb = lds;                                      // [4 waves][2 segments][5][BN]
            int* meta = reinterpret_cast<int*>(lds + 4 * 2 * FOLD_Q * T::BN);
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int q = 0; q < FOLD_Q; ++q)
#pragma unroll
                    for (int c = 0; c < NC; ++c)
                        if (lane + 64 * c < T::BN) comb[((w * 2 + a) * FOLD_Q + q) * T::BN + lane + 64 * c] = sum[a][q][c];
            if (lane == 0) meta[w] = imgA;
            __syncthreads();
            const int img0 = min(rt * 128, Po - 1) / HW, imgL = min(rt * 128 + 127, Po - 1) / HW;
            const int nq = has_se ? FOLD_Q : 2;
            for (int o = tid; o < FOLD_SLOTS * FOLD_Q * T::BN; o += 256) {
                const int sl = o / (FOLD_Q * T::BN), q = (o / T::BN) % FOLD_Q, col = o % T::BN;
                if (img0 + sl > imgL || q >= nq || n0 + col >= mcp) continue;
                float t = 0.f;
#pragma unroll
                for (int ww = 0; ww < 4; ++ww)
#pragma unroll
                    for (int a = 0; a < 2; ++a)
                        if (meta[ww] + a - img0 == sl) t += comb[((ww * 2 + a) * FOLD_Q + q) * T::BN + col];
                rec[(((size_t)rt * FOLD_SLOTS + sl) * FOLD_Q + q) * M + off + n0 + col] = t;
            }
            __syncthreads();
        }
    }
}

// Weight-gradient GEMMs: workgroups per CU the register budget must allow (tools/wg_timeline.py, round 4: the 7-tile variants
// took 260 / 248 registers -> ONE workgroup per CU, a 330-workgroup launch ran as two rounds of 65 us with every SIMD waiting on
// one wave's loads).  The split count below is sized to exactly these resident slots.
constexpr int wgrad_lb(int nt) { return nt >= 5 ? 2 : (nt >= 3 ? 3 : 4); }

// ============================================================================ project wgrad (TN, split-K)
// part[split][poff_g + o*mc + c] = sum_{p in split} dP_g[p][o] * z_g[p][c]   (k_reduce_rows sums the splits)
// tile: M side = mid channels c (128), N side = output channels o (16*NT)
template <int NT, int ACT>
__global__ __launch_bounds__(256, wgrad_lb(NT)) void k_project_wgrad(TfnasCellDesc d, const float* __restrict__ dout,
                                                       const float* __restrict__ Pr, const float* __restrict__ D,
                                                       const float* __restrict__ gate,
                                                       const double* __restrict__ stats2,
                                                       const double* __restrict__ stats3,
                                                       const double* __restrict__ red3,
                                                       const float* __restrict__ wmix, int rows_per_split,
                                                       int ntiles_o, float* __restrict__ part, size_t out_size) {
    using T = GT<NT>;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    WGT(0)
    const int g = blockIdx.z / ntiles_o, n0 = (blockIdx.z % ntiles_o) * T::BN;
    const int mc = d.g[g].mc, mcp = d.g[g].mcp, off = d.g[g].off;
    const bool has_se = d.g[g].se > 0;
    const float* gbase = has_se ? gate : D;
    const size_t se01 = has_se ? 1 : 0;
    size_t poff = 0;
    for (int gg = 0; gg < g; ++gg) poff += (size_t)d.g[gg].mc * d.oc;
    float* __restrict__ gw = part + (size_t)blockIdx.x * out_size + poff;
    const int m0 = blockIdx.y * 128;
    if (m0 >= mcp) return;
    const int HW = d.Ho * d.Wo, Po = d.N * HW, oc = d.oc, M = d.M;
    const int ocp = (oc + 15) & ~15;
    const int r0 = blockIdx.x * rows_per_split, r1 = min(Po, r0 + rows_per_split);
    const int nchunks = (r1 - r0 + 15) >> 4;
    const int tid = threadIdx.x, lane = tid & 63, lr = lane & 15, lq = lane >> 4, wrow = (tid >> 6) * 32;

    const Bn3Tab tab = bn3_tab_fill(lds + T::LDS_FLOATS, ocp, d, g, stats3, red3, wmix);
    // A thread stages the same 4 mid channels (A) and the same 4 output channels per B item in every K-chunk; only the
    // pixel advances (by 16 per chunk).  Everything that depends on the channel alone lives in registers: the loaders of
    // this kernel are VALU-issue bound (200 VALU instructions per 16 MFMAs before this), not bandwidth bound.
    const int mch = m0 + (tid & 31) * 4;
    const bool chok = mch < mcp;
    const size_t acol = (size_t)off + min(mch, mcp - 4);
    float2 c2[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
        c2[j] = (mch + j < mc) ? bn_consts(stats2 + 2 * (size_t)(off + mch + j), 1.0 / (double)Po, d.eps)
                               : make_float2(0.f, 0.f);
    __syncthreads();
    // BN3 backward with folded per-column constants:  dP = a3 (dOut - b3) - (Pr - mean) (rstd c3 a3)
    constexpr int BQ = T::BN / 4;
    int bcol[T::B_ITERS];
    bool bok[T::B_ITERS];
    f32x4 k_mean[T::B_ITERS], k_a3[T::B_ITERS], k_ab[T::B_ITERS], k_s[T::B_ITERS];
#pragma unroll
    for (int i = 0; i < T::B_ITERS; ++i) {
        const int idx = tid + 256 * i, o = n0 + (idx % BQ) * 4;
        bok[i] = idx < T::B_ITEMS && o < oc;
        bcol[i] = min(o, oc - 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int oo = bcol[i] + j;
            k_mean[i][j] = tab.mean[oo];
            k_a3[i][j] = tab.a3[oo];
            k_ab[i][j] = tab.a3[oo] * tab.b3[oo];
            k_s[i][j] = tab.rstd[oo] * tab.c3[oo] * tab.a3[oo];
        }
    }
    const float inv_hw = 1.f / (float)HW;
    const bool small_p = Po < (1 << 24);
    auto image_of = [&](int p) -> int {       // p / HW without the 22-instruction integer division (exact for p < 2^24)
        if (!small_p) return p / HW;
        int q = (int)((float)p * inv_hw);
        const int r = p - q * HW;
        q += (r >= HW) ? 1 : 0;
        q -= (r < 0) ? 1 : 0;
        return q;
    };

    f32x4 acc[2][NT];
    acc_zero<NT>(acc);
    auto la = [&](int c, int i, int kl, int m) -> Raw2 {
        const int p = min(r0 + c * 16 + kl, r1 - 1);
        Raw2 r;
        r.a = ldS4_raw(D, (size_t)p * M + acol, d.stor);
        r.b = ld4(gbase + ((size_t)image_of(p) * M + acol) * se01);        // no SE: one fixed (ignored) quad
        return r;
    };
    auto xa = [&](Raw2 r, int c, int i, int kl, int m) -> f32x4 {
        f32x4 v = ldS4_fin(r.a, d.stor);
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = act_f<ACT>((v[j] - c2[j].x) * c2[j].y);
        if (has_se) v *= r.b;
        return (chok && r0 + c * 16 + kl < r1) ? v : zero4();
    };
    auto lb = [&](int c, int i, int kl, int n) -> Raw2 {
        const int p = min(r0 + c * 16 + kl, r1 - 1);
        Raw2 r;
        r.a = ld4(dout + (size_t)p * oc + bcol[i]);
        r.b = ld4(Pr + ((size_t)g * Po + p) * oc + bcol[i]);
        return r;
    };
    auto xb = [&](Raw2 r, int c, int i, int kl, int n) -> f32x4 {
        const f32x4 v = (k_a3[i] * r.a - k_ab[i]) - (r.b - k_mean[i]) * k_s[i];
        return (bok[i] && r0 + c * 16 + kl < r1) ? v : zero4();
    };
    WGT(1)
    gemm_mainloop2<NT, false, false, false, TFNAS_WGRAD_PF2>(la, xa, lb, xb, nchunks, acc, lds);
    WGT(2)
    // The gradient is [oc][mc] (mid channel fastest) and a lane's accumulator quad is 4 consecutive mid channels of one
    // output channel: one 16-byte store per quad (the four lane groups of an output channel then cover 64 contiguous
    // bytes) instead of four 4-byte stores that each scatter a wave over 64 different cache lines.
    const bool vec = (mc & 3) == 0;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int ch0 = m0 + wrow + 16 * i + 4 * lq;
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int o = n0 + 16 * j + lr;
            if (o >= oc) continue;
            float* dst = gw + (size_t)o * mc + ch0;
            if (vec && ch0 + 3 < mc) {
                st4(dst, acc[i][j]);
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (ch0 + r < mc) dst[r] = acc[i][j][r];
            }
        }
    }
    WGT(3)
}

// ---------------------------------------------------------------------------- BN1-backward operand
// de[p][c] = rstd1 * (deh - T1/P - ehat*T2/P), ehat = (E-mean1)*rstd1, deh = dA1*act'(ehat) (stored by the
// depthwise-backward kernel);  cb1[c] = (mean1, rstd1, T1/P, T2/P)
__device__ __forceinline__ f32x4 bn1_de(const f32x4* cb, f32x4 deh, f32x4 e) {
    f32x4 r;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const f32x4 t = cb[j];
        const float eh = (e[j] - t.x) * t.y;
        r[j] = t.y * (deh[j] - t.z - eh * t.w);
    }
    return r;
}

// dx + w*g with the product rounded separately (what the sink's scaled copy followed by autograd's add computed before the
// two were folded into this epilogue): keeps the path level bit-identical to the per-cell route
__device__ __forceinline__ f32x4 sink_add(f32x4 v, float w, f32x4 g) {
    // the product must be rounded on its own: -ffp-contract=fast would fuse it into the add (HIP's __fmul_rn / __fadd_rn
    // are plain * and +, and `#pragma clang fp contract(off)` did not survive inlining here) -- the empty asm makes each
    // product opaque to the contraction pass
    float px = w * g.x, py = w * g.y, pz = w * g.z, pw = w * g.w;
    asm volatile("" : "+v"(px), "+v"(py), "+v"(pz), "+v"(pw));
    f32x4 r;
    r.x = v.x + px; r.y = v.y + py; r.z = v.z + pz; r.w = v.w + pw;
    return r;
}

// ============================================================================ expand dgrad, without reading E
// dx[p][c] = sum_m de[p][m] * W[m][c]  (+ sumw * dout[p][c] for residual cells),  m over all mid channels of all groups,
//   de = rstd * (deh - t1 - ehat*t2),  ehat = (E - mu) * rstd,  (mu, rstd, t1, t2) = cb1[m]   (BN1 backward).
// E is itself linear in x (E[p][m] = sum_c' x[p][c'] W[m][c']), so the ehat term collapses to an ic x ic operator:
//   dx = deh (rstd.W)  -  x G  +  b,      G[c'][c] = sum_m W[m][c'] s_m W[m][c],   s_m = rstd_m^2 t2_m,
//                                          b[c]    = sum_m (s_m mu_m - rstd_m t1_m) W[m][c].
// The kernel therefore streams ONE [P][M] tensor (deh) with plain loads instead of two with a BN transform per element,
// scales the (small) weight operand by rstd, and appends ceil(ic/16) K-chunks whose A operand is x and whose B operand
// is -G; b is added in the epilogue.  G | b come from k_expand_gram ([ic+4][ic] floats, row ic = b).
// K (up to 6912) is split over blockIdx.z when the output grid alone cannot fill the chip (7x7 / 14x14 cells: 49..196 row
// tiles): split z writes its partial tile to dxp[z] and k_dx_reduce adds them (and b, and the residual term).
template <int NT, int MM>
__global__ __launch_bounds__(256, gemm_lb(NT, MM)) void k_expand_dgrad(TfnasCellDesc d, const float* __restrict__ dEh,
                                                      const float* __restrict__ x, const float* __restrict__ cb1,
                                                      const float* __restrict__ gram, const float* __restrict__ dout,
                                                      const float* __restrict__ wmix, float* __restrict__ dx,
                                                      float* __restrict__ dxp, int nsplit,
                                                      const float* __restrict__ add_src,
                                                      const float* __restrict__ add_scale, int xonly_slot = -1) {
    using T = GT<NT>;
    __shared__ __attribute__((aligned(16))) float lds[T::LDS_FLOATS];
    const int n0 = blockIdx.y * T::BN;
    const int P = d.N * d.H * d.W, ic = d.ic, M = d.M;
    const int nrt = (P + 127) >> 7;
    const int tid = threadIdx.x, lr = tid & 15, wrow = (tid >> 6) * 32;
    const int zs = (int)blockIdx.z;
    int mchunks = 0;
    for (int g = 0; g < d.G; ++g) mchunks += (d.g[g].mcp + 15) >> 4;
    const int nchunks_all = mchunks + ((ic + 15) >> 4);      // mid-channel chunks, then the x / -G chunks
    // xonly_slot >= 0 (fused per-image route, fx_kernels.hip: the dE (rstd . W) term already sits in dxp[0 .. xonly_slot) as
    // partial sums): only the trailing x / -G chunks, written as one more partial tile dxp[xonly_slot]; k_dx_reduce adds them up
    const bool xonly = xonly_slot >= 0;
    const int per = xonly ? nchunks_all - mchunks : (nchunks_all + nsplit - 1) / nsplit;
    const int cbeg = xonly ? mchunks : zs * per;
    const int nchunks = max(0, min(nchunks_all, cbeg + per) - cbeg);
    float* __restrict__ dst = xonly ? dxp + (size_t)xonly_slot * P * ic : (nsplit > 1 ? dxp + (size_t)blockIdx.z * P * ic : dx);
    const bool plain = nsplit == 1 && !xonly;
    const bool add_res = d.has_res && plain;
    const bool add_sink = add_src != nullptr && plain;
    const float sink_w = add_src ? add_scale[0] : 0.f;
    float sumw = 1.f;
    if (wmix) {
        sumw = 0.f;
        for (int g = 0; g < d.G; ++g) sumw += wmix[g];
    }

    for (int rt = blockIdx.x; rt < nrt; rt += gridDim.x) {
        f32x4 acc[2][NT];
        acc_zero<NT>(acc);
        // chunk -> operand sources, worked out once per chunk (pre) for the four loader pieces: mid-channel chunks of
        // group g read dEh / rstd-scaled W_g, the trailing chunks read x / -G
        bool is_x;                       // chunk of the x / -G part
        int k0, klim_a, klim_b, ld_a;    // first k of the chunk; valid k of A (mcp | ic) and of B (mc | ic); A row stride
        size_t aoff;                     // element offset of the group's columns in a dEh row
        const float* bsrc;               // W_g or G
        const float* ssrc;               // rstd of the group's channels (cb1[...].y), stride 4
        auto pre = [&](int c) {
            c += cbeg;
            if (c >= mchunks) {
                is_x = true;
                k0 = (c - mchunks) * 16;
                klim_a = klim_b = ld_a = ic;
                aoff = 0;
                bsrc = gram;
                ssrc = cb1 + 1;
                return;
            }
            int g = 0;
            for (; g < d.G - 1; ++g) {
                const int t = (d.g[g].mcp + 15) >> 4;
                if (c < t) break;
                c -= t;
            }
            is_x = false;
            k0 = c * 16;
            klim_a = d.g[g].mcp;
            klim_b = d.g[g].mc;
            ld_a = M;
            aoff = d.g[g].off;
            bsrc = d.g[g].w_expand;
            ssrc = cb1 + 4 * (size_t)d.g[g].off + 1;
        };
        int prow[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) prow[i] = rt * 128 + wrow + 16 * i + lr;
        auto la = [&](int c, int i, int kl) -> f32x4 {
            const int pc = min(prow[i], P - 1), k = min(k0 + kl, klim_a - 4);
#ifdef TFNAS_HALF_BYTES
            if (!is_x) return ldS4(dEh, (size_t)pc * ld_a + aoff + k, 0);      // (timing-only build, tfnas_dev.h)
#endif
            return ld4((is_x ? x : dEh) + (size_t)pc * ld_a + aoff + k);       // wave-uniform select, one unconditional load
        };
        auto xa = [&](f32x4 r, int c, int i, int kl) -> f32x4 {
            return (prow[i] < P && k0 + kl < klim_a) ? r : zero4();
        };
        auto lb = [&](int c, int kl, int n) -> RawWS {
            const int kc = min(k0 + kl, klim_b - 1), col = min(n0 + n, ic - 4);
            RawWS r;
            r.w = ld4(bsrc + (size_t)kc * ic + col);
            r.s = ssrc[is_x ? 0 : 4 * (size_t)kc];      // rstd only: a dword load (a quad whose other lanes die lets their
                                                        // registers be reused before the MFMAs -> a wait); ignored for -G
            return r;
        };
        auto xb = [&](RawWS r, int c, int kl, int n) -> f32x4 {
            const f32x4 v = is_x ? -r.w : splat4(r.s) * r.w;
            return (n0 + n < ic && k0 + kl < klim_b) ? v : zero4();
        };
        gemm_adirect<NT, false, MM>(pre, la, xa, lb, xb, nchunks, acc, lds);
        emit_tile_rows<NT>(acc, lds, [&](int lrow, int lc, f32x4 v) {
            const int p = rt * 128 + lrow, c = n0 + lc;
            if (p < P && c < ic) {
                if (plain) v += ld4(gram + (size_t)ic * ic + c);
                if (add_res) v += splat4(sumw) * ld4(dout + (size_t)p * d.oc + c);
                if (add_sink) v = sink_add(v, sink_w, ld4(add_src + (size_t)p * ic + c));
                st4(dst + (size_t)p * ic + c, v);
            }
        });
    }
}

// dx = sum_z dxp[z] + b (+ sumw * dout for residual cells) (+ add_scale * add_src: sink-connecting gradient)
__global__ __launch_bounds__(256) void k_dx_reduce(TfnasCellDesc d, const float* __restrict__ dxp, int nsplit,
                                                   const float* __restrict__ gram, const float* __restrict__ dout,
                                                   const float* __restrict__ wmix, float* __restrict__ dx,
                                                   const float* __restrict__ add_src,
                                                   const float* __restrict__ add_scale) {
    const float sink_w = add_src ? add_scale[0] : 0.f;
    const size_t n4 = (size_t)d.N * d.H * d.W * d.ic / 4;
    const int iq = d.ic / 4;
    const float* __restrict__ bias = gram + (size_t)d.ic * d.ic;
    float sumw = 1.f;
    if (wmix) {
        sumw = 0.f;
        for (int g = 0; g < d.G; ++g) sumw += wmix[g];
    }
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        f32x4 v = ld4(bias + 4 * (int)(i % iq));
        if (d.has_res) v += splat4(sumw) * ld4(dout + 4 * i);
        for (int z = 0; z < nsplit; ++z) v += ld4(dxp + (size_t)z * n4 * 4 + 4 * i);
        if (add_src) v = sink_add(v, sink_w, ld4(add_src + 4 * i));
        st4(dx + 4 * i, v);
    }
}

// ============================================================================ BN1-backward correction operator
// part[split][(ic+4)*ic]:  rows c' < ic: G[c'][c] = sum_m W[m][c'] s_m W[m][c];  row ic: b[c] = sum_m coef_m W[m][c];
// s_m = rstd_m^2 t2_m, coef_m = s_m mu_m - rstd_m t1_m   (see k_expand_dgrad).  GEMM rows = input channels (+ the b row),
// columns = input channels, K = all mid channels of all groups, split over blockIdx.x (k_reduce_rows sums the splits).
template <int NT>
__global__ __launch_bounds__(256) void k_expand_gram(TfnasCellDesc d, const float* __restrict__ cb1, int chunks_per_split,
                                                     int nsplit, float* __restrict__ part) {
    using T = GT<NT>;
    __shared__ __attribute__((aligned(16))) float lds[T::LDS_FLOATS];
    const int ic = d.ic;
    const int m0 = blockIdx.y * 128, n0 = blockIdx.z * T::BN;
    const int xs = (int)blockIdx.x;             // K-split: partial tile at part[blockIdx.x]
    int mchunks = 0;
    for (int g = 0; g < d.G; ++g) mchunks += (d.g[g].mcp + 15) >> 4;
    const int cbeg = xs * chunks_per_split;
    const int nchunks = max(0, min(mchunks, cbeg + chunks_per_split) - cbeg);
    const f32x4* cb = reinterpret_cast<const f32x4*>(cb1);
    const int tid = threadIdx.x, lane = tid & 63, lr = lane & 15, lq = lane >> 4, wrow = (tid >> 6) * 32;
    float* __restrict__ out = part + (size_t)blockIdx.x * (size_t)(ic + 4) * ic;

    f32x4 acc[2][NT];
    acc_zero<NT>(acc);
    auto locate = [&](int c, int& g, int& k0) {
        c += cbeg;
        g = 0;
        for (; g < d.G - 1; ++g) {
            const int t = (d.g[g].mcp + 15) >> 4;
            if (c < t) break;
            c -= t;
        }
        k0 = c * 16;
    };
    auto fa = [&](int c, int kl, int m) -> f32x4 {       // A(m..m+3, k) = s_k W[k][m..m+3]; row ic = coef_k
        int g, k0;
        locate(c, g, k0);
        const int k = k0 + kl, r = m0 + m;
        if (k >= d.g[g].mc || r > ic) return zero4();
        const f32x4 t = cb[d.g[g].off + k];
        const float sk = t.y * t.y * t.w;
        if (r == ic) {
            f32x4 v = zero4();
            v.x = sk * t.x - t.y * t.z;
            return v;
        }
        return splat4(sk) * ld4(d.g[g].w_expand + (size_t)k * ic + r);
    };
    auto fb = [&](int c, int kl, int n) -> f32x4 {
        int g, k0;
        locate(c, g, k0);
        const int k = k0 + kl;
        return (k < d.g[g].mc && n0 + n < ic) ? ld4(d.g[g].w_expand + (size_t)k * ic + n0 + n) : zero4();
    };
    gemm_mainloop<NT, false, false>(fa, fb, nchunks, acc, lds);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = m0 + wrow + 16 * i + 4 * lq + r;
            if (row <= ic) {
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    const int cc = n0 + 16 * j + lr;
                    if (cc < ic) out[(size_t)row * ic + cc] = acc[i][j][r];
                }
            }
        }
}

// One-launch form of the same operator (round 6; the default -- TFNAS_ROUTE_GRAM2 selects the split-K GEMM + reduction above).
// G | b is tiny (ic <= 320, K = all mid channels: 6..130 MFLOP) and sits on the data-gradient chain of EVERY cell between the
// BN1-backward reduction and k_expand_dgrad; as a 128-row LDS-tiled GEMM split over K plus a k_reduce_rows launch it cost 13..28 us
// + 5 us per cell (w-step trace, round 6: 40 + 40 launches, 0.9 ms of chain time per w-step).  Here a workgroup owns a 16 x 32 tile
// of the output; its waves take batches of 32 mid channels round-robin, a lane loads its own MFMA operands straight from W
// (A[i][k] = s_k W[k][r0 + i], B[k][j] = W[k][c0 + j]: 64-byte row pieces, W stays in L2), a batch of GB steps' loads is in flight
// before their v_mfma_f32_16x16x4_f32; the waves' partial tiles are summed in double through LDS in wave order
// (deterministic).  The kernel is bound by its dependent load rounds (~1.5 us each).
#ifndef GRAM1_MAX_K
#define GRAM1_MAX_K 1536
#endif
constexpr int GRAM1_GB = 8;
// GRAM1_NW waves per workgroup: 4 inside the cells' backward (four queues keep every CU busy there: a 1 024-thread workgroup waits
// for a whole CU to drain -- in situ 44 us on average, 400 us worst case, against 17 us for the 256-thread form, rocprofv3 statistics
// of round 6), 16 for the head (K = 1 280, in the quiet stretch between the paths' forward and backward: 25-70 -> 10 us)
template <int GRAM1_NW>
__global__ __launch_bounds__(64 * GRAM1_NW) void k_gram1(TfnasCellDesc d, const float* __restrict__ cb1, float* __restrict__ gram) {
    __shared__ float part[GRAM1_NW][2][16][17];
    const int ic = d.ic;
    const int r0 = blockIdx.y * 16, c0 = blockIdx.x * 32;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 15, kq = lane >> 4;
    const f32x4* cb = reinterpret_cast<const f32x4*>(cb1);
    const int r = r0 + li, ca = c0 + li, cb_ = c0 + 16 + li;
    const bool row_w = r < ic, row_b = r == ic, oka = ca < ic, okb = cb_ < ic;
    const int ra = row_w ? r : 0, caa = oka ? ca : 0, cbb = okb ? cb_ : 0;      // clamped addresses, values masked below
    f32x4 acc0 = zero4(), acc1 = zero4();
    // the waves take batches of GB steps (4 mid channels each) round-robin over the concatenation of all groups' steps
    int base = 0;                                   // batches before this group
    for (int g = 0; g < d.G; ++g) {
        const float* __restrict__ W = d.g[g].w_expand;
        const int mc = d.g[g].mc, off = d.g[g].off;
        const int nstep = (mc + 3) >> 2, nbatch = (nstep + GRAM1_GB - 1) / GRAM1_GB;
        // first batch of this group that belongs to this wave: (base + b) % NW == wave
        int b = (wave - base % GRAM1_NW + GRAM1_NW) % GRAM1_NW;
        for (; b < nbatch; b += GRAM1_NW) {
            const int s0 = b * GRAM1_GB;
            float a[GRAM1_GB], b0[GRAM1_GB], b1[GRAM1_GB];
            f32x4 t[GRAM1_GB];
#pragma unroll
            for (int u = 0; u < GRAM1_GB; ++u) {
                const int m = 4 * (s0 + u) + kq;
                const int mm = m < mc ? m : mc - 1;
                const float* wr = W + (size_t)mm * ic;
                t[u] = cb[off + mm];
                a[u] = wr[ra];
                b0[u] = wr[caa];
                b1[u] = wr[cbb];
            }
#pragma unroll
            for (int u = 0; u < GRAM1_GB; ++u) {
                const int m = 4 * (s0 + u) + kq;
                const bool okm = m < mc;
                const float sk = t[u].y * t[u].y * t[u].w;
                float av = row_w ? sk * a[u] : (row_b ? sk * t[u].x - t[u].y * t[u].z : 0.f);
                av = okm ? av : 0.f;
                const float bv0 = (okm && oka) ? b0[u] : 0.f, bv1 = (okm && okb) ? b1[u] : 0.f;
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv0, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv1, acc1, 0, 0, 0);
            }
        }
        base += nbatch;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        part[wave][0][4 * kq + q][li] = acc0[q];
        part[wave][1][4 * kq + q][li] = acc1[q];
    }
    __syncthreads();
    for (int e = tid; e < 512; e += 64 * GRAM1_NW) {
        const int h = e >> 8, row = (e >> 4) & 15, col = e & 15;
        double v = 0.0;
#pragma unroll
        for (int w = 0; w < GRAM1_NW; ++w) v += (double)part[w][h][row][col];          // (wave order: deterministic)
        const int rr = r0 + row, cc = c0 + 16 * h + col;
        if (rr <= ic && cc < ic) gram[(size_t)rr * ic + cc] = (float)v;
    }
}

// ============================================================================ expand wgrad (TN, split-K)
// XG = false (stem; TFNAS_XG=0):  part[split][poff_g + m*ic + c] = sum_{p in split} de[p][off_g+m] * x[p][c]  with the
//   BN1-backward operand de = rstd (deh - t1 - ehat t2) formed per element from dEh AND E (k_reduce_rows sums the splits).
// XG = true (default): E is not read.  E is linear in x (E[p][m] = sum_c' W[m][c'] x[p][c']), so
//   sum_p de[p][m] x[p][c] = r_m ( R[m][c] - t1_m sx[c] - t2_m r_m ( (W Gx)[m][c] - mu_m sx[c] ) ),
//   R = sum_p deh x^T,  Gx = sum_p x x^T (ic x ic),  sx = sum_p x,  (mu, r, t1, t2) = cb1[m]
// and the kernel streams ONE [P][M] tensor with plain loads: the rows of R, and -- as ic + 1 more "mid channels" behind the
// last group's rows (operand = x itself / the constant 1) -- the rows of Gx | sx.  k_reduce_rows sums the splits in double,
// k_expand_wgrad_fix applies the formula in double (autograd of inverted_bottleneck.conv + BatchNorm2d, layers.py:463-482).
template <int NT, bool STEM, bool XG>
__global__ __launch_bounds__(256, wgrad_lb(NT)) void k_expand_wgrad(TfnasCellDesc d, const float* __restrict__ dEh,
                                                      const float* __restrict__ E, const float* __restrict__ cb1,
                                                      const float* __restrict__ x, int rows_per_split,
                                                      float* __restrict__ part, size_t out_size, size_t out_main) {
    using T = GT<NT>;
    __shared__ __attribute__((aligned(16))) float lds[T::LDS_FLOATS];
    const int ic = d.ic;
    int ty = blockIdx.y, g = 0;
    for (; g < d.G - 1; ++g) {
        const int t = (d.g[g].mcp + 127) >> 7;
        if (ty < t) break;
        ty -= t;
    }
    const int mc = d.g[g].mc, mcp = d.g[g].mcp, off = d.g[g].off;
    const bool last = XG && g == d.G - 1;           // the group whose row space carries the ic + 1 extension rows
    size_t poff = 0;
    for (int gg = 0; gg < g; ++gg) poff += (size_t)d.g[gg].mc * ic;
    float* __restrict__ gw = part + (size_t)blockIdx.x * out_size + poff;
    float* __restrict__ gx = part + (size_t)blockIdx.x * out_size + out_main;
    const int m0 = ty * 128, n0 = blockIdx.z * T::BN;
    const int P = d.N * d.H * d.W, M = d.M;
    const int r0 = blockIdx.x * rows_per_split, r1 = min(P, r0 + rows_per_split);
    const int nchunks = (r1 - r0 + 15) >> 4;
    const int tid = threadIdx.x, lane = tid & 63, lr = lane & 15, lq = lane >> 4, wrow = (tid >> 6) * 32;

    // per-thread constants of the K loop (see k_project_wgrad): the thread's 4 mid channels and its BN1-backward constants,
    // folded:  de = rstd (deh - t1 - (E - mu) rstd t2)  =  rstd deh - rstd t1 - (E - mu) (rstd^2 t2)
    const int mch = m0 + (tid & 31) * 4;
    const bool chok = mch < mcp;
    const size_t acol = (size_t)off + min(mch, mcp - 4);
    f32x4 k_mu = zero4(), k_r = zero4(), k_rt1 = zero4(), k_s = zero4();
    if (!XG) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const f32x4 t = (mch + j < mcp) ? reinterpret_cast<const f32x4*>(cb1)[off + mch + j] : zero4();
            k_mu[j] = t.x;
            k_r[j] = t.y;
            k_rt1[j] = t.y * t.z;
            k_s[j] = t.y * t.y * t.w;
        }
    }
    // XG: one base pointer and row pitch per thread -- its quad of dEh columns, or (extension rows) its quad of x columns
    const int ext = mch - mcp;                       // >= 0: extension row (x channel `ext`; ext == ic: the constant 1)
    const bool ax = last && ext >= 0 && ext < ic, aone = last && ext == ic;
    const float* __restrict__ abase = ax ? x + ext : dEh + acol;
    const size_t apitch = ax ? (size_t)ic : (size_t)M;
    const bool aok = chok || ax;
    constexpr int BQ = T::BN / 4;
    int bcol[T::B_ITERS];
    bool bok[T::B_ITERS];
#pragma unroll
    for (int i = 0; i < T::B_ITERS; ++i) {
        const int idx = tid + 256 * i, cc = n0 + (idx % BQ) * 4;
        bok[i] = idx < T::B_ITEMS && cc < ic;
        bcol[i] = STEM ? cc : min(cc, ic - 4);
    }

    f32x4 acc[2][NT];
    acc_zero<NT>(acc);
    auto la = [&](int c, int i, int kl, int m) -> Raw2 {
        const int p = min(r0 + c * 16 + kl, r1 - 1);
        Raw2 r;
        if (XG) {
            r.a = ld4(abase + (size_t)p * apitch);
            r.b = r.a;
        } else {
            const size_t at = (size_t)p * M + acol;
            r.a = ldS4_raw(dEh, at, d.stor);
            r.b = ldS4_raw(E, at, d.stor);
        }
        return r;
    };
    auto xa = [&](Raw2 r, int c, int i, int kl, int m) -> f32x4 {
        if (XG) {
            f32x4 v = aok ? r.a : zero4();
            if (aone) v.x = 1.f;
            return (r0 + c * 16 + kl < r1) ? v : zero4();
        }
        const f32x4 v = (k_r * ldS4_fin(r.a, d.stor) - k_rt1) - (ldS4_fin(r.b, d.stor) - k_mu) * k_s;
        return (chok && r0 + c * 16 + kl < r1) ? v : zero4();
    };
    auto lb = [&](int c, int i, int kl, int n) -> f32x4 {
        if (STEM) {
            const int p = r0 + c * 16 + kl;
            return (p < r1 && bok[i]) ? stem_patch4(x, d, p, bcol[i]) : zero4();
        }
        const int p = min(r0 + c * 16 + kl, r1 - 1);
        return ld4(x + (size_t)p * ic + bcol[i]);
    };
    auto xb = [&](f32x4 r, int c, int i, int kl, int n) -> f32x4 {
        if (STEM) return r;
        return (bok[i] && r0 + c * 16 + kl < r1) ? r : zero4();
    };
    gemm_mainloop2<NT, false, false, false, TFNAS_WGRAD_PF2>(la, xa, lb, xb, nchunks, acc, lds);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int ch = m0 + wrow + 16 * i + 4 * lq + r;
            float* __restrict__ row = nullptr;
            if (ch < mc) row = gw + (size_t)ch * ic;
            else if (last && ch >= mcp && ch - mcp <= ic) row = gx + (size_t)(ch - mcp) * ic;
            if (row) {
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    const int cc = n0 + 16 * j + lr;
                    if (cc < ic) row[cc] = acc[i][j][r];
                }
            }
        }
}

// red: the split sums of k_expand_wgrad<XG> in double -- R of every group | Gx [ic][ic] | sx [ic]
//   g_expand_g[m][c] = r ( R - t1 sx[c] - t2 r ( sum_c' W[m][c'] Gx[c'][c] - mu sx[c] ) ),   (mu, r, t1, t2) = cb1[off_g + m]
// A workgroup owns 64 consecutive outputs; the four waves take interleaved quarters of the c' sum (combined through LDS in
// wave order: bit-reproducible).
__global__ __launch_bounds__(256) void k_expand_wgrad_fix(TfnasCellDesc d, const float* __restrict__ cb1,
                                                          const double* __restrict__ red, size_t out_main) {
    __shared__ double ps[4][64];
    const int ic = d.ic, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const size_t e = (size_t)blockIdx.x * 64 + lane;
    const bool ok = e < out_main;
    int g = 0;
    size_t poff = 0;
    for (; g < d.G - 1; ++g) {
        const size_t n = (size_t)d.g[g].mc * ic;
        if (e < poff + n) break;
        poff += n;
    }
    const int m = ok ? (int)((e - poff) / ic) : 0, c = ok ? (int)(e - poff - (size_t)m * ic) : 0;
    const double* __restrict__ Gx = red + out_main;
    const float* __restrict__ w = d.g[g].w_expand + (size_t)m * ic;
    double wg0 = 0.0, wg1 = 0.0;
    for (int k = 4 * wv; k < ic; k += 16) {            // ic % 4 == 0 (cells; the stem keeps the per-element form)
        const f32x4 wq = ld4(w + k);
        wg0 += (double)wq.x * Gx[(size_t)k * ic + c] + (double)wq.z * Gx[(size_t)(k + 2) * ic + c];
        wg1 += (double)wq.y * Gx[(size_t)(k + 1) * ic + c] + (double)wq.w * Gx[(size_t)(k + 3) * ic + c];
    }
    ps[wv][lane] = wg0 + wg1;
    __syncthreads();
    if (wv == 0 && ok) {
        const double wg = (ps[0][lane] + ps[1][lane]) + (ps[2][lane] + ps[3][lane]);
        const f32x4 t = reinterpret_cast<const f32x4*>(cb1)[d.g[g].off + m];
        const double sx = Gx[(size_t)ic * ic + c], mu = t.x, r = t.y, t1 = t.z, t2 = t.w;
        d.g[g].g_expand[e - poff] = (float)(r * (red[e] - t1 * sx - t2 * r * (wg - mu * sx)));
    }
}

// ============================================================================ host launchers
static const int kNtSmall[] = {1, 2, 3, 4, 5, 7};   // N extents that are channel counts (ic / oc)

#define DISPATCH_NT(nt, ...)                                  \
    switch (nt) {                                             \
        case 1: { constexpr int NT = 1; __VA_ARGS__; } break; \
        case 2: { constexpr int NT = 2; __VA_ARGS__; } break; \
        case 3: { constexpr int NT = 3; __VA_ARGS__; } break; \
        case 4: { constexpr int NT = 4; __VA_ARGS__; } break; \
        case 5: { constexpr int NT = 5; __VA_ARGS__; } break; \
        case 7: { constexpr int NT = 7; __VA_ARGS__; } break; \
        default: return TFNAS_EINVAL;                         \
    }
// Arithmetic of the row-tiled 1x1-convolution GEMMs (gemm_x3.h): 6 = split-bf16 with six products per element pair (fp32-level
// accuracy on the bf16 matrix pipe; the default), 0 = v_mfma_f32_16x16x4_f32, 3 = split-bf16 keeping the 2^-16 terms,
// 1 = plain bf16.  tfnas_set_gemm_mode sets the process default (6 until then; the Python mirror seeds it from TFNAS_GEMM).
static int g_gemm_mode = 6;
static int g_gemm_everywhere = 0;      // TFNAS_GEMM_EVERYWHERE: no per-launch shape policy (tests compare every mode with the oracle)
int gemm_mode() { return g_gemm_mode; }
int set_gemm_mode(int m) {
    const int base = m & ~TFNAS_GEMM_EVERYWHERE;
    if (base != 0 && base != 1 && base != 3 && base != 6) return TFNAS_EINVAL;
    g_gemm_mode = base;
    g_gemm_everywhere = (m & TFNAS_GEMM_EVERYWHERE) ? 1 : 0;
    return 0;
}
#define DISPATCH_MM(...) DISPATCH_MM_(gemm_mode(), __VA_ARGS__)
#define DISPATCH_MM_(mode, ...)                               \
    switch (mode) {                                    \
        case 0: { constexpr int MM = 0; __VA_ARGS__; } break; \
        case 1: { constexpr int MM = 1; __VA_ARGS__; } break; \
        case 3: { constexpr int MM = 3; __VA_ARGS__; } break; \
        default: { constexpr int MM = 6; __VA_ARGS__; } break; \
    }
#define DISPATCH_ACT(act, ...)                                                        \
    if ((act) == TFNAS_ACT_RELU) { constexpr int ACT = TFNAS_ACT_RELU; __VA_ARGS__; } \
    else { constexpr int ACT = TFNAS_ACT_SWISH; __VA_ARGS__; }

// kernels with a statistics epilogue: at most 1024 partial rows (k_reduce_rows folds 512 per round trip) and they must fit
static size_t stats_row_cap(size_t row_floats) {
    const size_t fit = TFNAS_PART_FLOATS / (row_floats ? row_floats : 1);
    return fit < 1024 ? fit : 1024;
}

// Number of persistent row blocks (grid.x) of a row-tiled GEMM launch.  The launch runs gx * other workgroups on
// `slots` resident workgroup slots (256 CUs x 3 or 4 per CU, set by the kernel's launch bounds) and each workgroup walks
// ceil(nrt / gx) row tiles, so the launch takes about ceil(gx*other / slots) * ceil(nrt / gx) tile times.  The old rule
// (~4096 workgroups in total) left the 14x14 / 7x7 cells with a nearly empty last round (e.g. 2.04 rounds -> 3: 68 %
// efficient); this picks the gx that minimises the product, the larger gx on ties (finer dynamic balancing), subject to
// `cap` (partial rows of the statistics epilogues must fit the scratch buffer and stay <= 1024).
static int row_blocks(int rows, int other_blocks, size_t cap = 1u << 30, int slots = 1024, int max_gx = 1024) {
    const int nrt = cdiv(rows, 128);
    const int other = other_blocks > 0 ? other_blocks : 1;
    int lim = nrt;
    if ((size_t)lim > cap) lim = (int)cap;
    if (lim > max_gx) lim = max_gx;                    // (bounds the host-side search; measured per kernel family)
    if (lim < 1) lim = 1;
    long best_cost = -1;
    int best = 1;
    for (int gx = 1; gx <= lim; ++gx) {
        const long cost = (long)cdiv(gx * other, slots) * cdiv(nrt, gx);
        if (best_cost < 0 || cost <= best_cost) {
            best_cost = cost;
            best = gx;
        }
    }
    return best;
}
int gemm_mode();
static inline int gemm_slots(int nt, int mode = -1) { return 256 * gemm_lb(nt, mode < 0 ? gemm_mode() : mode); }
// ragged mid widths (mc % 4 != 0 after the elasticity re-masking): the split-bf16 instantiations only have the aligned weight
// loaders (the guarded ones cost ~50 registers, i.e. a resident workgroup), such launches keep the fp32 loop
// the launch's arithmetic: the descriptor's own mode (TFNAS_GEMM_EXPLICIT) or the process default
static inline int gemm_mode_for(const TfnasCellDesc& d) {
    for (int g = 0; g < d.G; ++g)
        if (d.g[g].mc & 3) return 0;
    if (d.gemm_mode & TFNAS_GEMM_EXPLICIT) return d.gemm_mode & 7;
    return gemm_mode();
}
static inline bool gemm_everywhere(const TfnasCellDesc& d) {
    return (d.gemm_mode & TFNAS_GEMM_EXPLICIT) ? (d.gemm_mode & TFNAS_GEMM_EVERYWHERE) != 0 : g_gemm_everywhere != 0;
}
// The data-gradient GEMMs gain little from the bf16 pipe (their K loops are bound by the BN3-backward transform / the chunk ->
// group bookkeeping and, with one candidate or large images, by bytes): measured per cell at B = 128 (tools/r4_cf.sh), split-bf16
// vs fp32 MFMA: all-candidate launches of the 14x14 / 7x7 cells 0.94-0.98x, 28x28 up to 1.19x, one-candidate launches 0.9-1.8x.
// They keep the fp32 loop except where they won.
static inline int gemm_mode_dgrad(const TfnasCellDesc& d) {
    const int m = gemm_mode_for(d);
    if (m == 0 || gemm_everywhere(d)) return m;
    if (m == 1) return m;                                  // plain bf16 is a reduced-precision MODE, not a policy: everywhere
    return (d.G > 1 && d.Ho * d.Wo <= 196) ? m : 0;
}
static inline int gemm_mode_fwd(const TfnasCellDesc& d, int /*hw*/) {
    return gemm_mode_for(d);
}

// Column-tile width of the GEMMs whose N extent is the mid channels of EVERY group (tiles cannot straddle groups): the
// candidate that pads the group widths least (72 | 144 -> 5 x 16: 960 columns for 864, where 4 x 16 needs 1280 and
// 20 tiles instead of 12; 336 | 672 -> 7 x 16 exactly), the wider one on ties.
static int pick_nt_groups(const TfnasCellDesc& d) {
    // only for cells with narrow groups (the HBM-bound 56x56 / 28x28 cells): on the matrix-bound later cells the 64-wide
    // tile at 4 waves/SIMD beats the 80- / 112-wide ones at 3 even when those fit exactly (measured)
    int min_mcp = 1 << 30;
    for (int g = 0; g < d.G; ++g) min_mcp = d.g[g].mcp < min_mcp ? d.g[g].mcp : min_mcp;
    if (min_mcp >= 200) return 4;
    static const int cands[] = {4, 5, 7};
    int best = 4;
    long best_pad = -1;
    for (int i = 0; i < 3; ++i) {
        long pad = 0;
        for (int g = 0; g < d.G; ++g) pad += (long)cdiv(d.g[g].mcp, 16 * cands[i]) * 16 * cands[i];
        if (best_pad < 0 || pad < best_pad || (pad == best_pad && cands[i] > best)) {
            best_pad = pad;
            best = cands[i];
        }
    }
    return best;
}

int launch_expand_fwd(const TfnasCellDesc& d, const float* x, float* E, double* stats1, float* part,
                      hipStream_t s) {
    ProfScope _prof(TK_EXPAND_FWD, s);
    // stem: 32 output channels -> 32-wide tiles (a 64-wide tile would be half empty)
    const int nt = d.mode == TFNAS_MODE_STEM ? (d.g[0].mcp <= 32 ? 2 : 4) : pick_nt_groups(d);
    int tiles = 0;
    for (int g = 0; g < d.G; ++g) tiles += cdiv(d.g[g].mcp, 16 * nt);
    const int mm = gemm_mode_fwd(d, d.H * d.W);
    dim3 grid(row_blocks(d.N * d.H * d.W, tiles, stats_row_cap(2 * (size_t)d.M), gemm_slots(nt, mm)), tiles);
    DISPATCH_MM_(mm, {
        if (d.mode == TFNAS_MODE_STEM) {
            if (nt == 2) hipLaunchKernelGGL((k_expand_fwd<2, true, MM>), grid, dim3(256), 0, s, d, x, E, part);
            else hipLaunchKernelGGL((k_expand_fwd<4, true, MM>), grid, dim3(256), 0, s, d, x, E, part);
        } else {
            switch (nt) {
                case 5: hipLaunchKernelGGL((k_expand_fwd<5, false, MM>), grid, dim3(256), 0, s, d, x, E, part); break;
                case 7: hipLaunchKernelGGL((k_expand_fwd<7, false, MM>), grid, dim3(256), 0, s, d, x, E, part); break;
                default: hipLaunchKernelGGL((k_expand_fwd<4, false, MM>), grid, dim3(256), 0, s, d, x, E, part); break;
            }
        }
    })
    _prof.stop();
    return launch_reduce_rows(part, grid.x, 2 * d.M, 2 * (size_t)d.M, stats1, nullptr, s);
}

#ifndef TFNAS_WSPLIT_DGRAD
#define TFNAS_WSPLIT_DGRAD 2
#endif
#ifndef TFNAS_WSPLIT_PFWD
#define TFNAS_WSPLIT_PFWD 2
#endif
int launch_project_fwd(const TfnasCellDesc& d, const float* D, const float* gate, const double* stats2,
                       float* Pr, double* stats3, float* part, hipStream_t s) {
    ProfScope _prof(TK_PROJECT_FWD, s);
    const int nt = pick_nt(d.oc, kNtSmall, 6), mm = gemm_mode_fwd(d, d.Ho * d.Wo);
    int mcp_max = 0;
    for (int g = 0; g < d.G; ++g) mcp_max = d.g[g].mcp > mcp_max ? d.g[g].mcp : mcp_max;
    const int tiles = cdiv(d.oc, 16 * nt);
    const int ncols2 = 2 * d.G * d.oc;
    const int Po = d.N * d.Ho * d.Wo, nrt = cdiv(Po, 128);
    // K-split for under-filled launches (sampled mode on the 14x14 / 7x7 cells: 150-400 workgroups walking 30-72
    // K-chunks back to back at ~2 us per chunk -- one workgroup per CU cannot hide the load latency)
    int nsplit = 1;
    const int wgs = nrt * tiles * d.G, kch = cdiv(mcp_max, 16);
    if (wgs < 512 && kch >= 16 && (d.oc & 3) == 0 && d.oc <= 1024) {
        nsplit = cdiv(1024, wgs);
        if (nsplit > 4) nsplit = 4;
        if (nsplit > kch / 8) nsplit = kch / 8;
        if (d.need_wgrad && d.G == 1 && nsplit > TFNAS_WSPLIT_PFWD) nsplit = TFNAS_WSPLIT_PFWD;      // (see expand_dgrad_splits)
        const size_t per = (size_t)d.G * Po * d.oc, rows2 = 256 * (size_t)ncols2;
        while (nsplit > 1 && (nsplit - 1) * per + rows2 > TFNAS_PART_FLOATS) --nsplit;
    }
    if (nsplit > 1) {
        float* prp = part;
        float* part2 = part + (size_t)(nsplit - 1) * d.G * Po * d.oc;
        dim3 grid(nrt, tiles, d.G * nsplit);
        DISPATCH_MM_(mm, DISPATCH_NT(nt, DISPATCH_ACT(d.act, {
            const size_t shm = (GT<NT>::LDS_FLOATS + 2 * ((mcp_max + 15) & ~15)) * sizeof(float);
            hipLaunchKernelGGL((k_project_fwd<NT, ACT, MM>), grid, dim3(256), shm, s, d, D, gate, stats2, Pr, part, nsplit, prp);
        })))
        int gx2 = cdiv(Po, 64);
        if (gx2 > 256) gx2 = 256;
        const int rps = cdiv(Po, gx2);
        gx2 = cdiv(Po, rps);
        hipLaunchKernelGGL(k_pr_reduce, dim3(gx2, d.G), dim3(256), 0, s, Pr, prp, nsplit - 1, d.G, Po, d.oc, rps, part2);
        _prof.stop();
        return launch_reduce_rows(part2, gx2, ncols2, (size_t)ncols2, stats3, nullptr, s);
    }
    dim3 grid(row_blocks(Po, tiles * d.G, stats_row_cap((size_t)ncols2), gemm_slots(nt, mm)), tiles, d.G);
    DISPATCH_MM_(mm, DISPATCH_NT(nt, DISPATCH_ACT(d.act, {
        const size_t shm = (GT<NT>::LDS_FLOATS + 2 * ((mcp_max + 15) & ~15)) * sizeof(float);
        hipLaunchKernelGGL((k_project_fwd<NT, ACT, MM>), grid, dim3(256), shm, s, d, D, gate, stats2, Pr, part, 1,
                           (float*)nullptr);
    })))
    _prof.stop();
    return launch_reduce_rows(part, grid.x, ncols2, (size_t)ncols2, stats3, nullptr, s);
}

#define TFNAS_FOLD_LAUNCH(NT_)                                                                                           \
    hipLaunchKernelGGL((k_project_dgrad<NT_, MM, true>), grid, dim3(256),                                               \
                       (GT<NT_>::LDS_FLOATS + 4 * ((d.oc + 15) & ~15)) * sizeof(float), s, d, dout, Pr, stats3, red3, wmix, dZ, \
                       D, stats2, rec)
int launch_project_dgrad(const TfnasCellDesc& d, const float* dout, const float* Pr, const double* stats3,
                         const double* red3, const float* wmix, float* dZ, hipStream_t s, const float* D,
                         const double* stats2, float* rec) {
    ProfScope _prof(TK_PROJECT_DGRAD, s);
    const int nt = pick_nt_groups(d), mm = gemm_mode_dgrad(d);
    int tiles = 0;
    for (int g = 0; g < d.G; ++g) tiles += cdiv(d.g[g].mcp, 16 * nt);
    dim3 grid(row_blocks(d.N * d.Ho * d.Wo, tiles, 1u << 30, gemm_slots(nt, mm)), tiles);
    if (rec) {
        if (!D || !stats2) return TFNAS_ENULL;
        DISPATCH_MM_(mm, {
            switch (nt) {                                   // (pick_nt_groups only returns 4, 5 or 7)
                case 5: TFNAS_FOLD_LAUNCH(5); break;
                case 7: TFNAS_FOLD_LAUNCH(7); break;
                default: TFNAS_FOLD_LAUNCH(4); break;
            }
        })
        return (int)hipGetLastError();
    }
    DISPATCH_MM_(mm, DISPATCH_NT(nt, {
        const size_t shm = (GT<NT>::LDS_FLOATS + 4 * ((d.oc + 15) & ~15)) * sizeof(float);
        hipLaunchKernelGGL((k_project_dgrad<NT, MM>), grid, dim3(256), shm, s, d, dout, Pr, stats3, red3, wmix, dZ);
    }))
    return (int)hipGetLastError();
}

static int pick_rows_per_split(int rows, int out_tiles, size_t out_size, int nt, size_t scratch = TFNAS_PART_FLOATS) {
    // ONE resident round: as many workgroups as the chip holds of this variant (256 CUs x wgrad_lb), never a second, mostly
    // empty round (cell 10: 55 splits x 6 tiles = 330 workgroups on 256 slots ran 2 x 65 us); at least 128 rows (8 K-chunks)
    // per split; partial tiles must fit the scratch
    const int target = 256 * wgrad_lb(nt), min_rows = 128;
    int splits = target / (out_tiles > 0 ? out_tiles : 1);
    const size_t cap = scratch / (out_size > 0 ? out_size : 1);
    if ((size_t)splits > cap) splits = (int)cap;
    if (splits < 1) splits = 1;
    int rps = cdiv(rows, splits);
    if (rps < min_rows) rps = min_rows;
    return ((rps + 15) / 16) * 16;
}

int launch_project_wgrad(const TfnasCellDesc& d, const float* dout, const float* Pr, const float* D,
                         const float* gate, const double* stats2, const double* stats3, const double* red3,
                         const float* wmix, float* part, hipStream_t s) {
    ProfScope _prof(TK_PROJECT_WGRAD, s);
    const int nt = pick_nt(d.oc, kNtSmall, 6);
    const int Po = d.N * d.Ho * d.Wo;
    int mcp_max = 0;
    for (int g = 0; g < d.G; ++g) mcp_max = d.g[g].mcp > mcp_max ? d.g[g].mcp : mcp_max;
    const int mtiles = cdiv(mcp_max, 128), ntiles = cdiv(d.oc, 16 * nt);
    size_t out_size = 0;
    for (int g = 0; g < d.G; ++g) out_size += (size_t)d.g[g].mc * d.oc;
    const int rps = pick_rows_per_split(Po, mtiles * ntiles * d.G, out_size, nt);
    dim3 grid(cdiv(Po, rps), mtiles, ntiles * d.G);
    DISPATCH_NT(nt, DISPATCH_ACT(d.act, {
        size_t shm = (GT<NT>::LDS_FLOATS + 5 * ((d.oc + 15) & ~15)) * sizeof(float);
        hipLaunchKernelGGL((k_project_wgrad<NT, ACT>), grid, dim3(256), shm, s, d, dout, Pr, D, gate, stats2,
                           stats3, red3, wmix, rps, ntiles, part, out_size);
    }))
    _prof.stop();
    size_t poff = 0;
    for (int g = 0; g < d.G; ++g) {
        const int n = d.g[g].mc * d.oc;
        int rc = launch_reduce_rows(part + poff, grid.x, n, out_size, nullptr, d.g[g].g_proj, s);
        if (rc) return rc;
        poff += n;
    }
    return (int)hipGetLastError();
}

// K-splits of expand dgrad: only when the (row tile x column tile) grid cannot fill the chip
int expand_dgrad_splits(const TfnasCellDesc& d) {
    const int nt = pick_nt(d.ic, kNtSmall, 6);
    const int tiles = cdiv(d.N * d.H * d.W, 128) * cdiv(d.ic, 16 * nt);
    if (tiles >= 512) return 1;
    int nchunks = 0;
    for (int g = 0; g < d.G; ++g) nchunks += cdiv(d.g[g].mcp, 16);
    int ns = cdiv(1024, tiles);
    if (ns > 16) ns = 16;
    if (ns > nchunks / 4) ns = nchunks / 4;      // at least 4 K-chunks per split
    // one-candidate launches of the weight step run beside three other queues: the chip is full anyway, and every extra split is
    // one more [P][ic] partial written, read back and summed (k_dx_reduce).  Capped at 2 (round 6, four alternating bench pairs on
    // one box: w-step 16.55 -> 16.37 ms; a cap of 1 loses: the 196-tile launches of the 14 x 14 cells then walk 40+ K-chunks each)
    if (d.need_wgrad && d.G == 1 && ns > TFNAS_WSPLIT_DGRAD) ns = TFNAS_WSPLIT_DGRAD;
    return ns < 1 ? 1 : ns;
}

// floats of the correction operator G | b
size_t expand_gram_floats(const TfnasCellDesc& d) { return (size_t)(d.ic + 4) * d.ic; }

// G | b -> gram ([ic+4][ic] floats); `scratch` holds the K-split partials (scratch_floats available)
int launch_expand_gram(const TfnasCellDesc& d, const float* cb1, float* scratch, size_t scratch_floats, float* gram,
                       hipStream_t s) {
    ProfScope _prof(TK_SMALL, s);
    int mtot = 0;
    for (int g = 0; g < d.G; ++g) mtot += d.g[g].mc;
    // policy (measured, B = 128, alternating bench runs): with the K of ONE candidate (the sampled launches of the w-step, the head)
    // the single launch wins; with all eight candidates' K (4 000 mid channels: 8 dependent load rounds per wave) it is at best equal
    // to the split-K GEMM + reduction (a 4-wave version: alpha-step +0.3 ms), which stays for those launches
    if (!(d.route & TFNAS_ROUTE_GRAM2) && mtot <= GRAM1_MAX_K) {
        // one launch, no K-split partials, no reduction (k_gram1); `scratch` is not used
        for (int g = 0; g < d.G; ++g)
            if (d.g[g].mc < 1) return TFNAS_EINVAL;
        const dim3 grid(cdiv(d.ic, 32), cdiv(d.ic + 1, 16));
        if (d.mode == TFNAS_MODE_HEAD) hipLaunchKernelGGL(k_gram1<16>, grid, dim3(1024), 0, s, d, cb1, gram);
        else hipLaunchKernelGGL(k_gram1<4>, grid, dim3(256), 0, s, d, cb1, gram);
        return (int)hipGetLastError();
    }
    const int nt = pick_nt(d.ic, kNtSmall, 6);
    const int ng = 1;
    const size_t gsz = (size_t)(d.ic + 4) * d.ic;
    int mchunks = 0;                                              // K-chunks of the operator
    for (int g = 0; g < d.G; ++g) mchunks += cdiv(d.g[g].mcp, 16);
    const int mtiles = cdiv(d.ic + 1, 128), ntiles = cdiv(d.ic, 16 * nt);
    int splits = cdiv(512, mtiles * ntiles * ng);
    if (splits > mchunks / 4) splits = mchunks / 4;               // at least 4 K-chunks per split
    if ((size_t)splits * ng > scratch_floats / gsz) splits = (int)(scratch_floats / gsz / ng);
    if (splits < 1) splits = 1;
    if (scratch_floats < gsz * ng) return TFNAS_ERANGE;
    const int cps = cdiv(mchunks, splits);
    splits = cdiv(mchunks, cps);
    dim3 grid(splits * ng, mtiles, ntiles);
    DISPATCH_NT(nt, { hipLaunchKernelGGL(k_expand_gram<NT>, grid, dim3(256), 0, s, d, cb1, cps, splits, scratch); })
    _prof.stop();
    return launch_reduce_rows(scratch, splits, (int)gsz, gsz, nullptr, gram, s, ng, (size_t)splits * gsz, gsz);
}

int launch_expand_dgrad(const TfnasCellDesc& d, const float* dEh, const float* x, const float* cb1, const float* gram,
                        const float* dout, const float* wmix, float* dx, float* dxp, hipStream_t s,
                        const float* add_src, const float* add_scale) {
    ProfScope _prof(TK_EXPAND_DGRAD, s);
    const int nt = pick_nt(d.ic, kNtSmall, 6);
    const int tiles = cdiv(d.ic, 16 * nt);
    const int nsplit = dxp ? expand_dgrad_splits(d) : 1;
    const int ng = 1;
    const int mm = gemm_mode_dgrad(d);
    dim3 grid(row_blocks(d.N * d.H * d.W, tiles * nsplit * ng, 1u << 30, gemm_slots(nt, mm), 4096), tiles, nsplit * ng);
    DISPATCH_MM_(mm, DISPATCH_NT(nt, {
        hipLaunchKernelGGL((k_expand_dgrad<NT, MM>), grid, dim3(256), 0, s, d, dEh, x, cb1, gram, dout, wmix, dx, dxp, nsplit,
                           add_src, add_scale);
    }))
    _prof.stop();
    if (nsplit > 1) {
        ProfScope _p2(TK_SMALL, s);
        const size_t n4 = (size_t)d.N * d.H * d.W * d.ic / 4;
        size_t blocks = cdiv64(n4, 256 * 2);
        if (blocks > 2048) blocks = 2048;
        hipLaunchKernelGGL(k_dx_reduce, dim3((unsigned)blocks, ng), dim3(256), 0, s, d, dxp, nsplit, gram, dout, wmix, dx,
                           add_src, add_scale);
    }
    return (int)hipGetLastError();
}

// fused per-image route (fx_kernels.hip): dxp[0 .. nsl) hold the partial sums of dE (rstd . W1); this adds the BN1-backward
// correction -x G (one more partial tile, MFMA) and sums everything (+ b, + residual, + sink gradient) into dx
int launch_expand_dgrad_x(const TfnasCellDesc& d, const float* x, const float* cb1, const float* gram, const float* dout,
                          const float* wmix, float* dx, float* dxp, int nsl, hipStream_t s, const float* add_src,
                          const float* add_scale) {
    {
        ProfScope _prof(TK_EXPAND_DGRAD, s);
        const int nt = pick_nt(d.ic, kNtSmall, 6);
        const int tiles = cdiv(d.ic, 16 * nt);
        const int mm = gemm_mode_dgrad(d);
        dim3 grid(row_blocks(d.N * d.H * d.W, tiles, 1u << 30, gemm_slots(nt, mm), 4096), tiles, 1);
        DISPATCH_MM_(mm, DISPATCH_NT(nt, {
            hipLaunchKernelGGL((k_expand_dgrad<NT, MM>), grid, dim3(256), 0, s, d, (const float*)nullptr, x, cb1, gram, dout, wmix,
                               dx, dxp, 1, add_src, add_scale, nsl);
        }))
    }
    ProfScope _p2(TK_SMALL, s);
    const size_t n4 = (size_t)d.N * d.H * d.W * d.ic / 4;
    size_t blocks = cdiv64(n4, 256 * 2);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(k_dx_reduce, dim3((unsigned)blocks, 1), dim3(256), 0, s, d, dxp, nsl + 1, gram, dout, wmix, dx, add_src,
                       add_scale);
    return (int)hipGetLastError();
}

// Expand weight gradient without reading E (Gram form, k_expand_wgrad<XG>) where E is at least 100 MB (measured alone at
// B = 128, tools/r5_xg.sh: cell 0 0.33 -> 0.24 ms, cells 1 / 2 equal; from 28 x 28 on E comes from the last-level cache and the
// extension rows + the fix-up launch cost more than the second stream: cell 10 0.08 -> 0.11 ms); TFNAS_ROUTE_XG_OFF: never,
// TFNAS_ROUTE_XG_ALL: wherever the shape allows (tests); every choice is compared with the oracle (tests/test_gpu_cell.py)
static bool expand_wgrad_xg(const TfnasCellDesc& d) {
    if ((d.route & TFNAS_ROUTE_XG_OFF) || d.mode == TFNAS_MODE_STEM || (d.ic & 3) != 0) return false;
    return (d.route & TFNAS_ROUTE_XG_ALL) || (size_t)d.N * d.H * d.W * d.M * sizeof(float) >= ((size_t)100 << 20);
}

static int launch_expand_wgrad_fix(const TfnasCellDesc& d, const float* cb1, const double* red, size_t out_main, hipStream_t s) {
    ProfScope _p2(TK_EXPAND_WGRAD, s);
    hipLaunchKernelGGL(k_expand_wgrad_fix, dim3((unsigned)cdiv64(out_main, 64)), dim3(256), 0, s, d, cb1, red, out_main);
    return (int)hipGetLastError();
}

int launch_expand_wgrad(const TfnasCellDesc& d, const float* dEh, const float* E, const float* cb1,
                        const float* x, float* part, hipStream_t s) {
    ProfScope _prof(TK_EXPAND_WGRAD, s);
    const int nt = pick_nt(d.ic, kNtSmall, 6);
    const int P = d.N * d.H * d.W;
    const bool xg = expand_wgrad_xg(d);
    if (!xg && !E) return TFNAS_ENULL;
    const int next = xg ? d.ic + 1 : 0;                             // extension rows behind the last group's
    int mtiles = 0;
    for (int g = 0; g < d.G; ++g) mtiles += cdiv(d.g[g].mcp + (g == d.G - 1 ? next : 0), 128);
    const int ntiles = cdiv(d.ic, 16 * nt);
    size_t out_main = 0;
    for (int g = 0; g < d.G; ++g) out_main += (size_t)d.g[g].mc * d.ic;
    const size_t out_size = out_main + (size_t)next * d.ic;
    // XG: the double sums of the splits live in the top of `part` (2 floats per element, 16-byte aligned)
    const size_t red_floats = xg ? 2 * out_size + 4 : 0;
    if (red_floats + out_size > TFNAS_PART_FLOATS) return TFNAS_ERANGE;
    const int rps = pick_rows_per_split(P, mtiles * ntiles, out_size, nt, TFNAS_PART_FLOATS - red_floats);
    dim3 grid(cdiv(P, rps), mtiles, ntiles);
    DISPATCH_NT(nt, {
        if (d.mode == TFNAS_MODE_STEM)
            hipLaunchKernelGGL((k_expand_wgrad<NT, true, false>), grid, dim3(256), 0, s, d, dEh, E, cb1, x, rps, part, out_size,
                               out_main);
        else if (xg)
            hipLaunchKernelGGL((k_expand_wgrad<NT, false, true>), grid, dim3(256), 0, s, d, dEh, E, cb1, x, rps, part, out_size,
                               out_main);
        else
            hipLaunchKernelGGL((k_expand_wgrad<NT, false, false>), grid, dim3(256), 0, s, d, dEh, E, cb1, x, rps, part, out_size,
                               out_main);
    })
    _prof.stop();
    if (xg) {
        double* red = reinterpret_cast<double*>((uintptr_t)(part + TFNAS_PART_FLOATS - red_floats + 3) & ~(uintptr_t)15);
        int rc = launch_reduce_rows(part, grid.x, (int)out_size, out_size, red, nullptr, s);
        if (rc) return rc;
        return launch_expand_wgrad_fix(d, cb1, red, out_main, s);
    }
    size_t poff = 0;
    for (int g = 0; g < d.G; ++g) {
        const int n = d.g[g].mc * d.ic;
        int rc = launch_reduce_rows(part + poff, grid.x, n, out_size, nullptr, d.g[g].g_expand, s);
        if (rc) return rc;
        poff += n;
    }
    return (int)hipGetLastError();
}
