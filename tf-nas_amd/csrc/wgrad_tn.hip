// Weight gradients of the two 1x1 convolutions as WAVE-level TN GEMMs without LDS staging:
//   project:  g_proj[o][c]   = sum_p dP[p][o] * z[p][c]        (autograd of point_linear.conv, models/layers.py:528-534)
//   expand:   g_expand[m][c] = sum_p de[p][m] * x[p][c]        (autograd of inverted_bottleneck.conv, layers.py:463-478)
// Both operands are pixel-major with the channel fastest, i.e. K (= pixels) is the SLOW dimension of both.  For
// v_mfma_f32_16x16x4_f32 lane l supplies A[i = l % 16][k = l / 16] and B[k = l / 16][j = l % 16]: with k <-> pixel p0 + l/16
// the float4 a lane loads from its pixel row -- channels 4*(l%16) .. +3 of a 64-channel tile -- already holds the operands of
// four MFMAs whose rows are the channels 4*i + comp.  So one 16-byte load per operand feeds 4 x 4 = 16 MFMAs (a 64 x 64 output
// tile per 4 pixels): no LDS, no barriers, no transposing stores, 2-4 loads per 512 MFMA cycles.  Waves are independent; the
// loads of step s + 2 are in flight under the MFMAs of steps s and s + 1.  (The LDS-tiled kernels this replaces for the
// sampled launches ran 16 K-chunks of one load -> LDS -> barrier -> 56 MFMAs round trip per workgroup at 0.6-0.9 waves per
// SIMD: 90-250 us per launch against 20-60 us of bytes / flops, and they bound the weight step's backward.)
//
// Work split.  UWAVE (early cells: few mid channels, 10^5..10^6 pixels, HBM-bound): a workgroup owns a pixel range and 4
// consecutive 64-channel u-tiles, one per wave -- the narrow operand (24..80 channels) is then read by the 4 waves of one CU
// at the same time (cache hits) instead of once per u-tile from HBM.  KWAVE (late cells: 10^3..10^4 pixels, up to 1536 mid
// channels, MFMA-bound): a workgroup owns one 64 x 64 tile and a pixel range, its 4 waves take every 4th K-step and their
// accumulators are summed through LDS (32 KB), so a launch writes one partial tile per workgroup, not per wave.
// Partial tiles go to part[split][...] in the layout of the gradient; the existing deterministic k_reduce_rows sums the splits.
#include <stdlib.h>
#include "tfnas_dev.h"
#include "kernels.h"
#include "prof.h"

#ifndef TFNAS_TN_LB
#define TFNAS_TN_LB 3          /* resident workgroups per CU the kernels are compiled for (register cap 512 / n per lane) */
#endif

namespace {

struct TnGeom {
    int rows_per_wg;      // pixels per workgroup (multiple of 16)
    int vtiles;           // 64-wide tiles of the narrow operand
    int ublocks;          // blockIdx.y extent: u-tile blocks per group (UWAVE: 4 tiles each, KWAVE: 1)
    size_t out_size;      // floats of one split's partial gradient (all groups)
};

__device__ __forceinline__ int image_of_px(int p, int HW, float inv_hw) {      // p / HW (exact for p < 2^24)
    int q = (int)((float)p * inv_hw);
    const int r = p - q * HW;
    q += (r >= HW) ? 1 : 0;
    q -= (r < 0) ? 1 : 0;
    return q;
}

// Loads of the software pipeline as inline asm with hand-placed s_waitcnt: the compiler's wait-count insertion put
// `s_waitcnt vmcnt(0)` at the loop header (it waits for the loads issued just before the back edge as well, i.e. one full
// memory latency per iteration) whatever the source looked like.  The destination registers are tied to the wait through
// "+v" operands, so no use can be scheduled in front of it; tn_drain() before the epilogue retires the prefetches that ran
// past the range.  (Checked in the ISA: no copies of in-flight registers on the back edge.)
__device__ __forceinline__ void tn_ld(f32x4& dst, const float* p) {
    asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(dst) : "v"(p) : "memory");
}
__device__ __forceinline__ void tn_ld_nt(f32x4& dst, const float* p) {
    asm volatile("global_load_dwordx4 %0, %1, off nt" : "=&v"(dst) : "v"(p) : "memory");
}
#define TN_WAIT4(N, A, B, C, D) asm volatile("s_waitcnt vmcnt(" #N ")" : "+v"(A), "+v"(B), "+v"(C), "+v"(D)::"memory")
#define TN_WAIT3(N, A, B, C) asm volatile("s_waitcnt vmcnt(" #N ")" : "+v"(A), "+v"(B), "+v"(C)::"memory")

#define TN_MFMA16(UQ, VQ)                                                                                 \
    _Pragma("unroll") for (int cu = 0; cu < 4; ++cu) {                                                    \
        _Pragma("unroll") for (int cv = 0; cv < 4; ++cv)                                                  \
            acc[cu][cv] = __builtin_amdgcn_mfma_f32_16x16x4f32((UQ)[cu], (VQ)[cv], acc[cu][cv], 0, 0, 0); \
    }

// sum the 4 waves' accumulators (KWAVE): waves 2,3 -> LDS, waves 0,1 add; wave 1 -> LDS, wave 0 adds.  lds: 2 * 4096 floats.
__device__ __forceinline__ void tn_reduce_waves(f32x4 (&acc)[4][4], float* lds, int wave, int lane) {
    f32x4* l4 = reinterpret_cast<f32x4*>(lds);
    if (wave >= 2) {
#pragma unroll
        for (int cu = 0; cu < 4; ++cu)
#pragma unroll
            for (int cv = 0; cv < 4; ++cv) l4[((wave - 2) * 16 + cu * 4 + cv) * 64 + lane] = acc[cu][cv];
    }
    __syncthreads();
    if (wave < 2) {
#pragma unroll
        for (int cu = 0; cu < 4; ++cu)
#pragma unroll
            for (int cv = 0; cv < 4; ++cv) acc[cu][cv] += l4[(wave * 16 + cu * 4 + cv) * 64 + lane];
    }
    __syncthreads();
    if (wave == 1) {
#pragma unroll
        for (int cu = 0; cu < 4; ++cu)
#pragma unroll
            for (int cv = 0; cv < 4; ++cv) l4[(cu * 4 + cv) * 64 + lane] = acc[cu][cv];
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
        for (int cu = 0; cu < 4; ++cu)
#pragma unroll
            for (int cv = 0; cv < 4; ++cv) acc[cu][cv] += l4[(cu * 4 + cv) * 64 + lane];
    }
}

// ============================================================================ project wgrad
// u = mid channel c (operand z = act(BN2(D)) * gate), v = output channel o (operand dP = BN3 backward of dout);
// part[split][poff_g + o * mc + c]
template <int ACT, bool KWAVE>
__global__ __launch_bounds__(256, TFNAS_TN_LB) void k_project_wgrad_tn(TfnasCellDesc d, const float* __restrict__ dout,
                                                             const float* __restrict__ Pr, const float* __restrict__ D,
                                                             const float* __restrict__ gate,
                                                             const double* __restrict__ stats2,
                                                             const double* __restrict__ stats3,
                                                             const double* __restrict__ red3,
                                                             const float* __restrict__ wmix, TnGeom gm,
                                                             float* __restrict__ part) {
    __shared__ __attribute__((aligned(16))) float lds[KWAVE ? 2 * 4096 : 4];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, lj = lane & 15, lk = lane >> 4;
    const int g = blockIdx.z / gm.vtiles, vt = blockIdx.z - g * gm.vtiles;
    const int mc = d.g[g].mc, mcp = d.g[g].mcp, off = d.g[g].off;
    const int ut = KWAVE ? (int)blockIdx.y : (int)blockIdx.y * 4 + wave;
    const int u0 = ut * 64, v0 = vt * 64;
    if ((KWAVE ? u0 : (int)blockIdx.y * 256) >= mcp) return;              // (whole workgroup: no barrier is skipped by a part)
    const bool wave_on = u0 < mcp;
    const int HW = d.Ho * d.Wo, Po = d.N * HW, oc = d.oc, M = d.M;
    const bool has_se = d.g[g].se > 0;
    size_t poff = 0;
    for (int gg = 0; gg < g; ++gg) poff += (size_t)d.g[gg].mc * oc;
    float* __restrict__ gw = part + (size_t)blockIdx.x * gm.out_size + poff;
    if (d.og) dout += (size_t)g * Po * oc;
    Pr += (size_t)g * Po * oc;
    const int r0 = blockIdx.x * gm.rows_per_wg, r1 = min(Po, r0 + gm.rows_per_wg);

    // per-lane channel constants: 4 mid channels (BN2: mean, rstd), 4 output channels (BN3 backward, folded)
    const int uc = u0 + 4 * lj, vc = v0 + 4 * lj;
    const bool uok = wave_on && uc < mcp, vok = vc < oc;
    const int ucl = min(uc, mcp - 4), vcl = min(vc, oc - 4);
    f32x4 c_mu, c_rs;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float2 c = (uok && uc + j < mc) ? bn_consts(stats2 + 2 * (size_t)(off + uc + j), 1.0 / (double)Po, d.eps)
                                               : make_float2(0.f, 0.f);
        c_mu[j] = c.x;
        c_rs[j] = c.y;
    }
    f32x4 k_a3, k_ab, k_mean, k_s;
    {
        const double inv = 1.0 / (double)Po;
        const float wg = wmix ? wmix[g] : 1.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const size_t o = (size_t)g * oc + vcl + j;
            const float2 c = bn_consts(stats3 + 2 * o, inv, d.eps);
            const float a3 = wg * c.y, b3 = (float)(red3[2 * o] * inv), c3 = (float)(red3[2 * o + 1] * inv);
            k_a3[j] = a3;
            k_ab[j] = a3 * b3;
            k_mean[j] = c.x;
            k_s[j] = c.y * c3 * a3;
        }
    }
    const float inv_hw = 1.f / (float)HW;
    const float* gbase = has_se ? gate : D;
    const size_t se01 = has_se ? 1 : 0;

    f32x4 acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = zero4();

    // this wave's K-steps: pixels r0 + 4 * (first + i * stride) + lk
    const int nsteps_wg = (r1 - r0 + 3) >> 2;
    const int first = KWAVE ? wave : 0, stride = KWAVE ? 4 : 1;
    const int nsteps = wave_on ? (nsteps_wg - first + stride - 1) / stride : 0;

    auto issue = [&](int i, f32x4& dv, f32x4& gt, f32x4& dq, f32x4& pq) {
        const int p = min(r0 + 4 * (first + i * stride) + lk, r1 - 1);           // (clamped: steps past the range re-read the last pixel)
        tn_ld_nt(dv, D + (size_t)p * M + off + ucl);
        tn_ld(gt, gbase + ((size_t)image_of_px(p, HW, inv_hw) * M + off + ucl) * se01);
        tn_ld(dq, dout + (size_t)p * oc + vcl);
        tn_ld(pq, Pr + (size_t)p * oc + vcl);
    };
    auto consume = [&](int i, f32x4 dv, f32x4 gt, f32x4 dq, f32x4 pq) {
        const bool pin = r0 + 4 * (first + i * stride) + lk < r1;
        f32x4 u, v;
#pragma unroll
        for (int j = 0; j < 4; ++j) u[j] = act_f<ACT>((dv[j] - c_mu[j]) * c_rs[j]);
        if (has_se) u *= gt;
        u = (uok && pin) ? u : zero4();
        v = (k_a3 * dq - k_ab) - (pq - k_mean) * k_s;
        v = vok ? v : zero4();
        TN_MFMA16(u, v)
    };
    // Software pipeline, distance 2: set a holds step i, set b step i + 1; every load is issued unconditionally from a
    // clamped address (steps past the range are masked in consume).
    f32x4 a0, a1, a2, a3, b0, b1, b2, b3;
    issue(0, a0, a1, a2, a3);
    issue(1, b0, b1, b2, b3);
    for (int i = 0; i < nsteps; i += 2) {
        TN_WAIT4(4, a0, a1, a2, a3);                 // the 4 younger loads (set b) may still be in flight
        consume(i, a0, a1, a2, a3);
        issue(i + 2, a0, a1, a2, a3);
        TN_WAIT4(4, b0, b1, b2, b3);
        consume(i + 1, b0, b1, b2, b3);
        issue(i + 3, b0, b1, b2, b3);
    }
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3)::"memory");
    if (KWAVE) tn_reduce_waves(acc, lds, wave, lane);
    if (KWAVE ? wave != 0 : !wave_on) return;
    // acc[cu][cv][r] = C[u = u0 + 16 * lk + 4 * r + cu][v = v0 + 4 * lj + cv]; gradient layout [o][mc]: for a fixed v the
    // four cu are four consecutive mid channels -> one 16-byte store
    const bool vec = (mc & 3) == 0;
#pragma unroll
    for (int cv = 0; cv < 4; ++cv) {
        const int o = v0 + 4 * lj + cv;
        if (o >= oc) continue;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int ch = u0 + 16 * lk + 4 * r;
            float* dst = gw + (size_t)o * mc + ch;
            const f32x4 q = {acc[0][cv][r], acc[1][cv][r], acc[2][cv][r], acc[3][cv][r]};
            if (vec && ch + 3 < mc) {
                st4(dst, q);
            } else {
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    if (ch + c < mc) dst[c] = q[c];
            }
        }
    }
}

// ============================================================================ expand wgrad
// u = mid channel m (operand de = BN1 backward of (dEh, E)), v = input channel c (operand x); part[split][poff_g + m * ic + c]
template <bool KWAVE>
__global__ __launch_bounds__(256, TFNAS_TN_LB) void k_expand_wgrad_tn(TfnasCellDesc d, const float* __restrict__ dEh,
                                                            const float* __restrict__ E, const float* __restrict__ cb1,
                                                            const float* __restrict__ x, TnGeom gm,
                                                            float* __restrict__ part) {
    __shared__ __attribute__((aligned(16))) float lds[KWAVE ? 2 * 4096 : 4];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, lj = lane & 15, lk = lane >> 4;
    const int g = blockIdx.z / gm.vtiles, vt = blockIdx.z - g * gm.vtiles;
    const int mc = d.g[g].mc, mcp = d.g[g].mcp, off = d.g[g].off;
    const int ut = KWAVE ? (int)blockIdx.y : (int)blockIdx.y * 4 + wave;
    const int u0 = ut * 64, v0 = vt * 64;
    if ((KWAVE ? u0 : (int)blockIdx.y * 256) >= mcp) return;
    const bool wave_on = u0 < mcp;
    const int P = d.N * d.H * d.W, ic = d.ic, M = d.M;
    size_t poff = 0;
    for (int gg = 0; gg < g; ++gg) poff += (size_t)d.g[gg].mc * ic;
    float* __restrict__ gw = part + (size_t)blockIdx.x * gm.out_size + poff;
    if (d.xg) x += (size_t)g * P * ic;
    const int r0 = blockIdx.x * gm.rows_per_wg, r1 = min(P, r0 + gm.rows_per_wg);

    const int uc = u0 + 4 * lj, vc = v0 + 4 * lj;
    const bool uok = wave_on && uc < mcp, vok = vc < ic;
    const int ucl = min(uc, mcp - 4), vcl = min(vc, ic - 4);
    // de = rstd (deh - t1 - (E - mu) rstd t2)  =  rstd deh - rstd t1 - (E - mu) (rstd^2 t2)
    f32x4 k_mu, k_r, k_rt1, k_s;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const f32x4 t = (uok && uc + j < mcp) ? reinterpret_cast<const f32x4*>(cb1)[off + uc + j] : zero4();
        k_mu[j] = t.x;
        k_r[j] = t.y;
        k_rt1[j] = t.y * t.z;
        k_s[j] = t.y * t.y * t.w;
    }
    f32x4 acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = zero4();

    const int nsteps_wg = (r1 - r0 + 3) >> 2;
    const int first = KWAVE ? wave : 0, stride = KWAVE ? 4 : 1;
    const int nsteps = wave_on ? (nsteps_wg - first + stride - 1) / stride : 0;
    auto issue = [&](int i, f32x4& dq, f32x4& eq, f32x4& xq) {
        const int p = min(r0 + 4 * (first + i * stride) + lk, r1 - 1);
        const size_t at = (size_t)p * M + off + ucl;
        tn_ld_nt(dq, dEh + at);
        tn_ld_nt(eq, E + at);
        tn_ld(xq, x + (size_t)p * ic + vcl);
    };
    auto consume = [&](int i, f32x4 dq, f32x4 eq, f32x4 xq) {
        const bool pin = r0 + 4 * (first + i * stride) + lk < r1;
        f32x4 u = (k_r * dq - k_rt1) - (eq - k_mu) * k_s;
        u = (uok && pin) ? u : zero4();
        const f32x4 v = vok ? xq : zero4();
        TN_MFMA16(u, v)
    };
    f32x4 a0, a1, a2, b0, b1, b2;                // (software pipeline as in k_project_wgrad_tn)
    issue(0, a0, a1, a2);
    issue(1, b0, b1, b2);
    for (int i = 0; i < nsteps; i += 2) {
        TN_WAIT3(3, a0, a1, a2);
        consume(i, a0, a1, a2);
        issue(i + 2, a0, a1, a2);
        TN_WAIT3(3, b0, b1, b2);
        consume(i + 1, b0, b1, b2);
        issue(i + 3, b0, b1, b2);
    }
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(b0), "+v"(b1), "+v"(b2)::"memory");
    if (KWAVE) tn_reduce_waves(acc, lds, wave, lane);
    if (KWAVE ? wave != 0 : !wave_on) return;
    // gradient layout [m][ic]: for a fixed u the four cv are four consecutive input channels -> one 16-byte store
#pragma unroll
    for (int cu = 0; cu < 4; ++cu)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int ch = u0 + 16 * lk + 4 * r + cu;
            const int c = v0 + 4 * lj;
            if (ch < mc && c < ic) {
                const f32x4 q = {acc[cu][0][r], acc[cu][1][r], acc[cu][2][r], acc[cu][3][r]};
                st4(gw + (size_t)ch * ic + c, q);            // (ic % 4 == 0)
            }
        }
}

// pixel ranges: enough workgroups to fill the chip (~8 waves per SIMD slot wanted: the kernels hide latency with waves),
// at least 64 pixels per wave, partial tiles must fit the scratch
TnGeom tn_geometry(int rows, int G, int mcp_max, int vdim, size_t out_size, bool kwave) {
    TnGeom gm;
    gm.vtiles = cdiv(vdim, 64);
    gm.ublocks = kwave ? cdiv(mcp_max, 64) : cdiv(mcp_max, 256);
    gm.out_size = out_size;
    const long tiles = (long)gm.ublocks * gm.vtiles * G;
    static const int target = getenv("TFNAS_TN_WGS") ? atoi(getenv("TFNAS_TN_WGS")) : 2048;
    long splits = (target + tiles - 1) / tiles;
    const size_t cap = TFNAS_PART_FLOATS / (out_size ? out_size : 1);
    if ((size_t)splits > cap) splits = (long)cap;
    if (splits < 1) splits = 1;
    int rps = cdiv(rows, (int)splits);
    const int min_rows = kwave ? 256 : 64;
    if (rps < min_rows) rps = min_rows;
    gm.rows_per_wg = (rps + 15) & ~15;
    return gm;
}

bool tn_kwave(const TfnasCellDesc& d, int rows) {
    // late cells: few pixels, wide groups -> one tile per workgroup, waves split K
    static const int thr = getenv("TFNAS_TN_KWAVE_ROWS") ? atoi(getenv("TFNAS_TN_KWAVE_ROWS")) : 60000;
    return rows <= thr;
}

}   // namespace

// TFNAS_WGRAD_TN: 0 = never (the default), 1 = the KWAVE geometry only (14x14 / 7x7 cells), 2 = every cell.
// Measured at B = 128 on one MI355X, one sampled candidate, kernels alone (tools/cell_family.py, TFNAS_WGRAD_STREAM=0): the late
// cells' weight-gradient GEMMs go 128 -> 84 us (cell 11 project), 95 -> 73 us (cell 17 project), expand about equal, while the
// HBM-bound 112x112 / 56x56 cells are SLOWER than the LDS-tiled kernels (61 -> 124 us / 239 -> 338 us on cell 0: one or two
// 64-channel u-tiles leave most of a workgroup's waves idle, and two loads per wave in flight cannot cover the HBM latency).
// Inside a search iteration pair the late-cell gain disappears in the noise (74.6-75.4 vs 75.6 ms per pair): the weight-gradient
// queues share the chip with the data-gradient chains, so the switch stays off; the kernels are kept as a measured variant
// (parity: tests/test_gpu_variants.py).
static int wgrad_tn_mode() {
    static const int m = [] { const char* e = getenv("TFNAS_WGRAD_TN"); return e ? atoi(e) : 0; }();
    return m;
}
bool wgrad_tn_enabled() { return wgrad_tn_mode() > 0; }

// true when the launch was taken
bool launch_project_wgrad_tn(const TfnasCellDesc& d, const float* dout, const float* Pr, const float* D, const float* gate,
                             const double* stats2, const double* stats3, const double* red3, const float* wmix, float* part,
                             hipStream_t s, int* rc) {
    if (!wgrad_tn_enabled() || (d.oc & 3) || d.oc < 4) return false;
    const int Po = d.N * d.Ho * d.Wo;
    if (Po >= (1 << 24)) return false;
    int mcp_max = 0;
    size_t out_size = 0;
    for (int g = 0; g < d.G; ++g) {
        mcp_max = d.g[g].mcp > mcp_max ? d.g[g].mcp : mcp_max;
        out_size += (size_t)d.g[g].mc * d.oc;
    }
    const bool kw = tn_kwave(d, Po);
    if (!kw && wgrad_tn_mode() < 2) return false;
    const TnGeom gm = tn_geometry(Po, d.G, mcp_max, d.oc, out_size, kw);
    const dim3 grid(cdiv(Po, gm.rows_per_wg), gm.ublocks, gm.vtiles * d.G);
    {
        ProfScope _prof(TK_PROJECT_WGRAD, s);
        if (d.act == TFNAS_ACT_RELU) {
            if (kw) hipLaunchKernelGGL((k_project_wgrad_tn<TFNAS_ACT_RELU, true>), grid, dim3(256), 0, s, d, dout, Pr, D, gate, stats2, stats3, red3, wmix, gm, part);
            else hipLaunchKernelGGL((k_project_wgrad_tn<TFNAS_ACT_RELU, false>), grid, dim3(256), 0, s, d, dout, Pr, D, gate, stats2, stats3, red3, wmix, gm, part);
        } else {
            if (kw) hipLaunchKernelGGL((k_project_wgrad_tn<TFNAS_ACT_SWISH, true>), grid, dim3(256), 0, s, d, dout, Pr, D, gate, stats2, stats3, red3, wmix, gm, part);
            else hipLaunchKernelGGL((k_project_wgrad_tn<TFNAS_ACT_SWISH, false>), grid, dim3(256), 0, s, d, dout, Pr, D, gate, stats2, stats3, red3, wmix, gm, part);
        }
    }
    size_t poff = 0;
    *rc = 0;
    for (int g = 0; g < d.G && *rc == 0; ++g) {
        const int n = d.g[g].mc * d.oc;
        *rc = launch_reduce_rows(part + poff, grid.x, n, out_size, nullptr, d.g[g].g_proj, s);
        poff += n;
    }
    if (*rc == 0) *rc = (int)hipGetLastError();
    return true;
}

bool launch_expand_wgrad_tn(const TfnasCellDesc& d, const float* dEh, const float* E, const float* cb1, const float* x,
                            float* part, hipStream_t s, int* rc) {
    if (!wgrad_tn_enabled() || d.mode != TFNAS_MODE_CELL || (d.ic & 3)) return false;
    const int P = d.N * d.H * d.W;
    int mcp_max = 0;
    size_t out_size = 0;
    for (int g = 0; g < d.G; ++g) {
        mcp_max = d.g[g].mcp > mcp_max ? d.g[g].mcp : mcp_max;
        out_size += (size_t)d.g[g].mc * d.ic;
    }
    const bool kw = tn_kwave(d, P);
    if (!kw && wgrad_tn_mode() < 2) return false;
    const TnGeom gm = tn_geometry(P, d.G, mcp_max, d.ic, out_size, kw);
    const dim3 grid(cdiv(P, gm.rows_per_wg), gm.ublocks, gm.vtiles * d.G);
    {
        ProfScope _prof(TK_EXPAND_WGRAD, s);
        if (kw) hipLaunchKernelGGL((k_expand_wgrad_tn<true>), grid, dim3(256), 0, s, d, dEh, E, cb1, x, gm, part);
        else hipLaunchKernelGGL((k_expand_wgrad_tn<false>), grid, dim3(256), 0, s, d, dEh, E, cb1, x, gm, part);
    }
    size_t poff = 0;
    *rc = 0;
    for (int g = 0; g < d.G && *rc == 0; ++g) {
        const int n = d.g[g].mc * d.ic;
        *rc = launch_reduce_rows(part + poff, grid.x, n, out_size, nullptr, d.g[g].g_expand, s);
        poff += n;
    }
    if (*rc == 0) *rc = (int)hipGetLastError();
    return true;
}
