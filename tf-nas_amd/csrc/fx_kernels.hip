// Fused per-image route of the late cells ("fx"; fx.h): expand 1x1 -> BN1 -> act -> depthwise k x k in ONE kernel per direction,
// the expanded tensor E and its gradient never leave the compute unit.
//
// Reference arithmetic: inverted_bottleneck + depth_conv of MBInvertedResBlock.forward (models/layers.py:542-552; the modules are
// built at layers.py:463-507) and their autograd backward; BatchNorm = batch statistics, no affine (layers.py:469,498).
//
// Why this shape (measured): halving the bytes of every stream tensor leaves the alpha-step unchanged --
// the materialise-once kernels are bound by instructions and latency (ring loaders, LDS-staged GEMM epilogues, K loops of 5-12
// chunks with a DRAM round trip each), not by HBM.  So the fused kernels are built around instruction count:
//   * a workgroup (8 waves) owns NI whole images x a slice of mid channels; the images' x rows are split ONCE into three bf16
//     planes and stay in registers as MFMA B operands for the workgroup's life (x: [pixels][ic], ic = 80..192);
//   * the expand weights arrive pre-split (k_fx_pack: one "blob" per 32-channel chunk = W1 planes in the A-operand layout + BN1
//     constants + depthwise taps), copied to LDS one chunk ahead -- no operand is loaded from memory inside the MFMA loop;
//   * C = W1_chunk x^T puts 4 consecutive CHANNELS of one pixel in a lane: BN1 + activation and one ds_write_b128 drop it
//     into the [pixel][32 channel] LDS tile the stencil reads (zero halo written once; the whole image is resident: no halo
//     recompute, no ring, no per-row bookkeeping);
//   * forward: one barrier per chunk; inside an interval the two waves of a SIMD run the stencil of chunk i and the MFMAs of
//     chunk i + 1 in opposite order (matrix pipe beside LDS / VALU);
//   * backward: dd = BN2-backward(dZ, D) -> LDS, stencil with flipped taps, dE = . * act'(ehat) written back into the LDS tile that
//     held ehat, and that tile is the B operand of the expand-dgrad MFMA (A = rstd (.) W1 planes from the blob): dx partial sums
//     accumulate in registers over the slice; BN1-backward sums t1, t2 come out of the same pass (the correction operator
//     -x G + b of k_expand_dgrad is applied afterwards, it is linear in them).
#include "tfnas_dev.h"
#include "kernels.h"
#include "prof.h"
#include "fx.h"
#include "gemm_x3.h"
#include <type_traits>

// timing-only ablation of the fused kernels (tools/r5_fxabl.sh): bit 0 no stencil taps, 1 no expand MFMAs, 2 no D / E stores,
// 3 no blob copies, 4 no activation in the expand epilogue.  Never set in the product build.
#ifndef FX_ABL
#define FX_ABL 0
#endif

// -DFX_TIMING (tools/fx_timeline.py): shader-clock stamps of one chunk interval (FXT_CHUNK) of the first 64 workgroups, per wave
#ifdef FX_TIMING
#ifndef FXT_CHUNK
#define FXT_CHUNK 3
#endif
__device__ unsigned long long g_fxt[64 * 8 * 16];
#define FXT(chunk_, slot_)                                                                                   \
    if ((threadIdx.x & 63) == 0 && blockIdx.x < 64 && ((chunk_) == FXT_CHUNK || (chunk_) < 0))              \
        g_fxt[(blockIdx.x * 8 + (threadIdx.x >> 6)) * 16 + (slot_)] = __builtin_readcyclecounter();
extern "C" int tfnas_dbg_fx_timing(unsigned long long* out, int n) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_fxt), sizeof(unsigned long long) * (size_t)n);
}
#else
#define FXT(chunk_, slot_)
#endif

// ---------------------------------------------------------------------------------------------------------------- plan
static int fx_ks(int ic) { return (ic + 31) / 32; }

bool fx_plan(const TfnasCellDesc& d, FxPlan& pl, bool bwd) {
#ifdef TFNAS_FXW_TIMING
    // TIMING-ONLY build (tools/r6_fxw.sh): the fused per-image route also for launches that want weight
    // gradients -- the existing weight-gradient kernels then read ehat as if it were E and the dx-partial scratch as if it were dE
    // (wrong numerics by construction; never shipped, never tested for parity): what would the sampled late cells of the w-step
    // gain from a fused route BEFORE anybody builds its weight gradients?
    if (d.mode != TFNAS_MODE_CELL) return false;
#else
    if (d.mode != TFNAS_MODE_CELL || d.need_wgrad) return false;
#endif
    if (stats_sync_on(d)) return false;                     // (BN1 statistics come from the Gram matrix of x: efree_kernels.hip)
    if (d.ic < 64 || d.ic > 192 || (d.ic & 15)) return false;
    if (d.stride != 1) return false;                       // (stride-2 cells keep the materialised route)
    const int HW = d.H * d.W;
    if (HW > 196 || d.W < 4) return false;
    const int KS = fx_ks(d.ic);
    int NI = 128 / HW;                                      // several small images per workgroup: one pixel tile per wave
    if (NI < 1) NI = 1;
    if (NI > 4) NI = 4;
    if (NI > d.N) NI = d.N;
    const int ntiles = (NI * HW + 15) / 16, RT = (ntiles + 7) / 8;
    const int RTF = (ntiles + 6) / 7;                       // forward: 7 worker waves
    if (RTF * KS > 8) return false;                         // register budget of the x planes: 12 * RT * KS VGPRs
    if (NI * ((d.Wo + 3) / 4) * d.Ho > 56) return false;    // one stencil item (4-pixel strip x channel quad) per worker thread
    pl.KS = KS; pl.RT = RT; pl.RTF = RTF; pl.NI = NI;
    pl.nig = (d.N + NI - 1) / NI;
    pl.RS = 64 * KS + FX_RS_PAD;
    int kmax = 3;
    for (int g = 0; g < d.G; ++g) {
        if (d.g[g].k != 3 && d.g[g].k != 5) return false;
        if (d.g[g].k > kmax) kmax = d.g[g].k;
    }
    pl.KMAX = kmax;
    pl.PB = 96 * pl.RS + 256;                               // 3 planes x 32 channels | (mean, rstd) x 32
    pl.WB = 25 * 128;                                       // taps [k*k][32] (room for 5 x 5)
    pl.XB = bwd ? 512 + 3 * d.ic * 64 : 0;                  // cst2 [32] x 4 | rstd (.) W1 planes [3][ic][32 channels]
    pl.BLOB = (pl.PB + pl.WB + pl.XB + 255) & ~255;
    // chunks and slices: a slice never crosses a group (one kernel size per workgroup)
    int nch = 0;
    for (int g = 0; g < d.G; ++g) nch += (d.g[g].mcp + 31) / 32;
    pl.nchunks = nch;
    if (nch > 32000) return false;
    const int target_wgs = bwd ? 640 : 1280;
    int want = (target_wgs + pl.nig - 1) / pl.nig;
    if (bwd) {                                              // every slice leaves a [P][ic] partial of dx in the cell's dEh buffer
        const int room = d.M / d.ic - 1 - d.G;              // (.. [P][M] floats; - G: rounding of the per-group split below)
        if (want > room) want = room;
    }
    if (want < 1) want = 1;
    int cps = (nch + want - 1) / want;                      // chunks per slice
    if (cps < (bwd ? 6 : 3)) cps = bwd ? 6 : 3;
    int ns = 0, chunk0 = 0;
    for (int g = 0; g < d.G; ++g) {
        const int ng = (d.g[g].mcp + 31) / 32;
        const int parts = (ng + cps - 1) / cps;
        for (int p = 0; p < parts; ++p) {
            if (ns >= FX_MAX_SLICES) return false;
            const int a = (int)((long)ng * p / parts), b = (int)((long)ng * (p + 1) / parts);
            pl.sl[ns].g = (int16_t)g;
            pl.sl[ns].c0 = (int16_t)(32 * a);
            pl.sl[ns].nch = (int16_t)(b - a);
            pl.sl[ns].chunk0 = (int16_t)(chunk0 + a);
            ++ns;
        }
        chunk0 += ng;
    }
    pl.nslices = ns;
    return true;
}

// LDS bytes of the forward kernel: two image tiles, two plane buffers, two tap buffers, two statistics buffers
static size_t fx_fwd_lds(const TfnasCellDesc& d, const FxPlan& pl) {
    const int PAD = pl.KMAX / 2, HP = d.H + 2 * PAD, WP = (d.W + 2 * PAD) | 1;
    const size_t tile = (size_t)(pl.NI * HP * WP + 8) * 128;
    return 2 * tile + 2 * (size_t)pl.PB + 2 * (size_t)pl.WB + 2 * 2048;
}

// ---------------------------------------------------------------------------------------------------------------- pack
// blob[chunk] = | W1 planes h, m, l: [32 channels][RS bytes] bf16, k < 32 KS, zeros beyond ic / mc | cst1 [32] (mean, rstd) |
//               | taps [k*k][32] | (backward) cst2 [32] (mean2, rstd2, R1/Po, R2/Po) | Wr planes [3][ic][64 B, k-group swizzled] |
// Wr[c][ch] = rstd1[ch] * W1[ch][c]: the A operand of the expand dgrad (rows = input channels, k = the chunk's 32 mid channels);
// 16-byte group q of row c is stored at group q ^ fx_swz(c) -- conflict-free ds_read_b128 without padding (fx.h).
__device__ __host__ __forceinline__ int fx_swz(int row) { return (0x1230 >> (4 * ((row >> 2) & 3))) & 3; }   // 0, 3, 2, 1

__device__ __forceinline__ void fx_split3(float v, unsigned short& h, unsigned short& m, unsigned short& l) {
    const unsigned hb = x3_bits(v);
    const float r1 = v - x3_float(hb & 0xffff0000u);
    const unsigned mb = x3_bits(r1);
    const float r2 = r1 - x3_float(mb & 0xffff0000u);
    h = (unsigned short)(hb >> 16);
    m = (unsigned short)(mb >> 16);
    l = (unsigned short)(x3_bits(r2) >> 16);
}

__global__ __launch_bounds__(256) void k_fx_pack(TfnasCellDesc d, FxPlan pl, const double* __restrict__ stats1,
                                                 const double* __restrict__ stats2, const double* __restrict__ red2,
                                                 unsigned char* __restrict__ blob) {
    // grid = (chunks, 4): workgroup y packs channels 8 y .. 8 y + 7 of the chunk (126-216 workgroups alone left the chip idle: 21 us)
    const int chunk = blockIdx.x, cy = blockIdx.y * 8, tid = threadIdx.x;
    int si = 0;
    for (; si < pl.nslices - 1; ++si)
        if (chunk < pl.sl[si + 1].chunk0) break;
    const int g = pl.sl[si].g, c0 = pl.sl[si].c0 + 32 * (chunk - pl.sl[si].chunk0);
    const int mc = d.g[g].mc, off = d.g[g].off, ic = d.ic, K = d.g[g].k, KK = K * K;
    const int RS = pl.RS, KP = 32 * pl.KS;
    unsigned char* b = blob + (size_t)chunk * pl.BLOB;
    const float* __restrict__ w1 = d.g[g].w_expand;
    __shared__ float2 cst[8];
    __shared__ float w1s[8][200];                  // the octet's weight rows (ic <= 192; zeros for channels past mc)
    if (tid < 8) {
        float2 c = make_float2(0.f, 0.f);
        if (c0 + cy + tid < mc) c = bn_consts(stats1 + 2 * (size_t)(off + c0 + cy + tid), 1.0 / ((double)d.N * d.H * d.W), d.eps);
        cst[tid] = c;
        reinterpret_cast<float2*>(b + 96 * RS)[cy + tid] = c;
    }
    for (int e = tid; e < 8 * ic; e += 256) {
        const int chl = e / ic, c = e - chl * ic;
        w1s[chl][c] = (c0 + cy + chl < mc) ? w1[(size_t)(c0 + cy + chl) * ic + c] : 0.f;
    }
    __syncthreads();
    // planes: 8 consecutive k of one channel per item -> one 16-byte store per plane
    for (int e = tid; e < 8 * (KP / 8); e += 256) {
        const int chl = e / (KP / 8), k0 = 8 * (e - chl * (KP / 8));
        unsigned short h[8], m[8], l[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) fx_split3(k0 + j < ic ? w1s[chl][k0 + j] : 0.f, h[j], m[j], l[j]);
        unsigned char* row = b + (cy + chl) * RS + 2 * k0;
        *reinterpret_cast<u32x4*>(row) = u32x4{h[0] | (unsigned)h[1] << 16, h[2] | (unsigned)h[3] << 16, h[4] | (unsigned)h[5] << 16, h[6] | (unsigned)h[7] << 16};
        *reinterpret_cast<u32x4*>(row + 32 * RS) = u32x4{m[0] | (unsigned)m[1] << 16, m[2] | (unsigned)m[3] << 16, m[4] | (unsigned)m[5] << 16, m[6] | (unsigned)m[7] << 16};
        *reinterpret_cast<u32x4*>(row + 64 * RS) = u32x4{l[0] | (unsigned)l[1] << 16, l[2] | (unsigned)l[3] << 16, l[4] | (unsigned)l[5] << 16, l[6] | (unsigned)l[7] << 16};
    }
    float* taps = reinterpret_cast<float*>(b + pl.PB);
    for (int e = tid; e < KK * 8; e += 256) {
        const int t = e >> 3, ch = cy + (e & 7);
        taps[t * 32 + ch] = (c0 + ch < mc) ? d.g[g].w_dw[(size_t)(c0 + ch) * KK + t] : 0.f;
    }
    if (pl.XB) {
        f32x4* c2 = reinterpret_cast<f32x4*>(b + pl.PB + pl.WB);
        if (tid < 8) {
            f32x4 t = zero4();
            const int ch = cy + tid;
            if (c0 + ch < mc) {
                const double inv = 1.0 / ((double)d.N * d.Ho * d.Wo);
                const float2 c = bn_consts(stats2 + 2 * (size_t)(off + c0 + ch), inv, d.eps);
                t.x = c.x;
                t.y = c.y;
                t.z = (float)(red2[2 * (size_t)(off + c0 + ch) + 0] * inv);
                t.w = (float)(red2[2 * (size_t)(off + c0 + ch) + 1] * inv);
            }
            c2[ch] = t;
        }
        // Wr: the octet's 8 channels of input channel c are one 16-byte group of row c
        unsigned char* wr = b + pl.PB + pl.WB + 512;
        for (int c = tid; c < ic; c += 256) {
            unsigned short h[8], m[8], l[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) fx_split3(cst[j].y * w1s[j][c], h[j], m[j], l[j]);
            unsigned char* p = wr + c * 64 + (((cy >> 3) ^ fx_swz(c)) * 16);
            *reinterpret_cast<u32x4*>(p) = u32x4{h[0] | (unsigned)h[1] << 16, h[2] | (unsigned)h[3] << 16, h[4] | (unsigned)h[5] << 16, h[6] | (unsigned)h[7] << 16};
            *reinterpret_cast<u32x4*>(p + ic * 64) = u32x4{m[0] | (unsigned)m[1] << 16, m[2] | (unsigned)m[3] << 16, m[4] | (unsigned)m[5] << 16, m[6] | (unsigned)m[7] << 16};
            *reinterpret_cast<u32x4*>(p + 2 * ic * 64) = u32x4{l[0] | (unsigned)l[1] << 16, l[2] | (unsigned)l[3] << 16, l[4] | (unsigned)l[5] << 16, l[6] | (unsigned)l[7] << 16};
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------- shared pieces
typedef unsigned char u8;

template <int KS, int RT>
struct FxX {                       // the workgroup's x rows as MFMA B operands: pixel = column, k = input channel
    bf16x8 h[RT][KS], m[RT][KS], l[RT][KS];
};

__device__ __forceinline__ void fx_split8(f32x4 a, f32x4 b, bf16x8& h, bf16x8& m, bf16x8& l) {
    const X3Planes pa = x3_split4<6>(a), pb = x3_split4<6>(b);
    h = __builtin_bit_cast(bf16x8, u32x4{pa.h.x, pa.h.y, pb.h.x, pb.h.y});
    m = __builtin_bit_cast(bf16x8, u32x4{pa.m.x, pa.m.y, pb.m.x, pb.m.y});
    l = __builtin_bit_cast(bf16x8, u32x4{pa.l.x, pa.l.y, pb.l.x, pb.l.y});
}

// six products of the split operands, smallest terms first (gemm_x3.h)
#define FX_MFMA6(acc, ah, am, al, bh, bm, bl)                                   \
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, acc, 0, 0, 0);        \
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bm, acc, 0, 0, 0);        \
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, acc, 0, 0, 0);        \
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bh, acc, 0, 0, 0);        \
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bm, acc, 0, 0, 0);        \
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, acc, 0, 0, 0);

// copy `bytes` (multiple of 16) from global to LDS with all FX_THREADS threads: loads now, stores later (two-phase)
template <int NV>
struct FxCopy {
    u32x4 v[NV];
    // (every lane always loads -- from a clamped offset: a predicated load leaves a select on the destination registers, and
    //  the compiler then waits for ALL outstanding memory traffic right behind the load instead of at the store below)
    __device__ __forceinline__ void load(const u8* __restrict__ src, int bytes) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int o = (threadIdx.x + i * FX_THREADS) * 16;
            v[i] = *reinterpret_cast<const u32x4*>(src + (o < bytes ? o : bytes - 16));
        }
    }
    __device__ __forceinline__ void store(u8* dst, int bytes) const {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int o = (threadIdx.x + i * FX_THREADS) * 16;
            if (o < bytes) *reinterpret_cast<u32x4*>(dst + o) = v[i];
        }
    }
};

// one of the six split products on every (channel tile, pixel tile) accumulator of the wave: consecutive MFMAs never share an
// accumulator (a dependent v_mfma_f32_16x16x32_bf16 cannot issue before its predecessor has left the pipe)
// (no condition around an MFMA, however uniform: every `if` is a scalar compare + branch and a basic-block boundary the
//  scheduler cannot move MFMAs across -- the first version, with `if (tile exists)` around each one, ran 4x slower than its
//  instruction counts; tiles past the end of the workgroup's pixels multiply zero x planes instead)
#define FX_TERM(A_, B_)                                                                                                  \
    _Pragma("unroll") for (int ct = 0; ct < 2; ++ct) _Pragma("unroll") for (int pt = 0; pt < RT; ++pt)                   \
        if (!(FX_ABL & 2)) acc[ct][pt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A_[ct], X.B_[pt][ks], acc[ct][pt], 0, 0, 0);

// e = BN1(x W1_chunk^T) for the wave's pixel tiles: A = W1 planes (LDS blob), B = x planes (registers).  Lane (n, q) of tile
// (ct, pt) ends up with channels 16 ct + 4 q .. + 3 of pixel 16 (wave + 8 pt) + n; OUT = 0 / 1: act(e) -> tile, 2: e -> tile.
// eg != nullptr: e (normalised, before the activation) also goes to global memory, eg[pixel * M + channel] (what the backward
// of the stored-ehat mode reads back; egoff[pt] = the pixel's row offset, chlim = valid channels of the chunk).
template <int KS, int RT, int OUT>
__device__ __forceinline__ void fx_expand(const u8* P, const FxX<KS, RT>& X, float* tile, const int (&slot)[RT],
                                          const bool (&pv)[RT], int ntiles, float* __restrict__ eg = nullptr,
                                          const size_t* egoff = nullptr, int chlim = 0) {
    constexpr int RS = 64 * KS + FX_RS_PAD;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n = lane & 15, q = lane >> 4;
    f32x4 acc[2][RT];
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int pt = 0; pt < RT; ++pt) acc[ct][pt] = zero4();
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        bf16x8 ah[2], am[2], al[2];
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
            const u8* ap = P + (16 * ct + n) * RS + 64 * ks + 16 * q;
            ah[ct] = *reinterpret_cast<const bf16x8*>(ap);
            am[ct] = *reinterpret_cast<const bf16x8*>(ap + 32 * RS);
            al[ct] = *reinterpret_cast<const bf16x8*>(ap + 64 * RS);
        }
        FX_TERM(al, h) FX_TERM(am, m) FX_TERM(ah, l) FX_TERM(am, h) FX_TERM(ah, m) FX_TERM(ah, h)     // smallest terms first
    }
    const float2* cst = reinterpret_cast<const float2*>(P + 96 * RS);
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
        float2 c[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) c[r] = cst[16 * ct + 4 * q + r];
#pragma unroll
        for (int pt = 0; pt < RT; ++pt) {
            // (pixels past the end carry a spare slot of the tile: the LDS store is unconditional)
            f32x4 e, v;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                e[r] = (acc[ct][pt][r] - c[r].x) * c[r].y;
                v[r] = (OUT == 2 || (FX_ABL & 16)) ? e[r] : act_f<(OUT == 2 ? 0 : OUT)>(e[r]);
            }
            st4(tile + slot[pt] * 32 + 16 * ct + 4 * q, v);
            // (columns between mc and the chunk's end are the group's padding in the [pixels][M] row: zeros may go there)
            if (!(FX_ABL & 4) && eg && pv[pt]) st4_nt(eg + egoff[pt] + 16 * ct + 4 * q, e);
        }
    }
}

// sum a per-thread (sum, sumsq) quad pair over the 8 strips of a wave (lanes sharing lane & 7) and park the wave's totals
__device__ __forceinline__ void fx_stat_park(f32x4 a, f32x4 b, float* stat) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int o = 8; o < 64; o <<= 1) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            a[r] += __shfl_xor(a[r], o, 64);
            b[r] += __shfl_xor(b[r], o, 64);
        }
    }
    if (lane < 8) {
        st4(stat + wave * 64 + 4 * lane, a);
        st4(stat + wave * 64 + 32 + 4 * lane, b);
    }
}
// threads 0..63: total of the 8 waves -> this workgroup's partial row (row = image group), fixed order
__device__ __forceinline__ void fx_stat_emit(const float* stat, float* __restrict__ prow, int c0, int mcp) {
    const int tid = threadIdx.x;
    if (tid < 64) {
        const int ch = tid & 31, which = tid >> 5;
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < 8; ++w) t += stat[w * 64 + which * 32 + ch];
        if (c0 + ch < mcp) prow[2 * (size_t)(c0 + ch) + which] = t;
    }
}

// ---------------------------------------------------------------------------------------------------------------- forward
// Seven WORKER waves (pixel tiles wave + 7 pt; one stencil item each) and one COPIER wave (wave 7).  The workers never wait
// on vector memory inside the chunk loop: they only issue stores (D, ehat).  On gfx950 loads and stores share one counter, so
// a wave that prefetches the next chunk's blob AND stores results ends every interval in `s_waitcnt vmcnt(0)`, i.e. waiting for
// its store acknowledgements (ablation, tools/r5_fxabl.sh: the kernel without taps and without MFMAs still took 60 % of its
// time).  The copier moves the blobs (global -> registers -> LDS, all pieces in flight at once) and writes the statistics rows.
constexpr int FX_WORKERS = 7;

struct FxFItem {
    int toff;        // float offset of the window's top-left pixel in the padded image tile (+ 4 cq)
    unsigned doff;   // element offset of the strip's first output pixel in D, relative to the image group's first pixel row
    int npx;         // valid pixels of the strip; 0: no item
};

template <int K>
__device__ __forceinline__ void fx_fwd_stencil(const float* tile, const float* taps, const FxFItem it, int WP, int M, bool chok,
                                               float* __restrict__ Dc, float* stat) {
    const int cq = threadIdx.x & 7;
    f32x4 ssum = zero4(), ssq = zero4();
    if (it.npx > 0) {
        const float* base = tile + it.toff;
        f32x4 acc[4] = {zero4(), zero4(), zero4(), zero4()};
#pragma unroll 1
        for (int ky = 0; ky < ((FX_ABL & 1) ? 0 : K); ++ky) {
            const float* rowp = base + ky * WP * 32;
            const float* wp = taps + ky * K * 32 + 4 * cq;
            f32x4 win[K + 3];
#pragma unroll
            for (int u = 0; u < K + 3; ++u) win[u] = ld4(rowp + u * 32);
#pragma unroll
            for (int kx = 0; kx < K; ++kx) {
                const f32x4 wv = ld4(wp + kx * 32);
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) acc[jj] += win[jj + kx] * wv;
            }
        }
        if (chok) {
            float* __restrict__ o = Dc + it.doff;
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                if (jj < it.npx) {
                    if (!(FX_ABL & 4)) st4_nt(o + (size_t)jj * M, acc[jj]);
                    ssum += acc[jj];
                    ssq += acc[jj] * acc[jj];
                }
            }
        }
    }
    fx_stat_park(ssum, ssq, stat);
}

// One interval of a worker wave as ONE instruction stream: the MFMAs of chunk i + 1 (k-step by k-step) with the tap rows of chunk i's
// stencil item between them.  A wave is in-order: run one after the other, its interval is t(MFMA) + t(stencil) however busy the
// partner wave keeps the other pipe (measured: the two added up exactly, tools/r5_fxabl.sh); in one basic block the scheduler
// fills the 16-cycle shadow of every v_mfma_f32_16x16x32_bf16 with the stencil's LDS reads and v_pk_fma_f32.
template <int K, int ACT, int KS, int RT>
__device__ __forceinline__ void fx_fwd_interval(const u8* P, const FxX<KS, RT>& X, float* tnext, const int (&slot)[RT],
                                                const bool (&pv)[RT], float* __restrict__ eg, const size_t* egoff, bool do_expand,
                                                const float* tile, const float* taps, const FxFItem it, int WP, int M, bool chok,
                                                float* __restrict__ Dc, float* stat) {
    constexpr int RS = 64 * KS + FX_RS_PAD;
    const int lane = threadIdx.x & 63, n = lane & 15, q = lane >> 4, cq = lane & 7;
    f32x4 acc[2][RT];
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int pt = 0; pt < RT; ++pt) acc[ct][pt] = zero4();
    f32x4 sacc[4] = {zero4(), zero4(), zero4(), zero4()};
    const float* base = tile + it.toff;
    auto tap_row = [&](int ky) __attribute__((always_inline)) {
        const float* rowp = base + ky * WP * 32;
        const float* wp = taps + ky * K * 32 + 4 * cq;
        f32x4 win[K + 3];
#pragma unroll
        for (int u = 0; u < K + 3; ++u) win[u] = ld4(rowp + u * 32);
#pragma unroll
        for (int kx = 0; kx < K; ++kx) {
            const f32x4 wv = ld4(wp + kx * 32);
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) sacc[jj] += win[jj + kx] * wv;
        }
    };
    // 6 KS groups of 2 RT MFMAs (one split product on every accumulator); tap row ky follows group ((ky + 1) * 6 KS) / K - 1, and a
    // scheduling barrier closes every group: the scheduler may mix a group's MFMAs with ONE tap row's LDS reads and FMAs, not more
    // (left alone it hoists every window load to the top: 100-170 spilled registers)
    constexpr int NG = 6 * KS;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        bf16x8 ah[2], am[2], al[2];
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
            const u8* ap = P + (16 * ct + n) * RS + 64 * ks + 16 * q;
            ah[ct] = *reinterpret_cast<const bf16x8*>(ap);
            am[ct] = *reinterpret_cast<const bf16x8*>(ap + 32 * RS);
            al[ct] = *reinterpret_cast<const bf16x8*>(ap + 64 * RS);
        }
#define FX_GROUP(T_, A_, B_)                                                                   \
    {                                                                                          \
        _Pragma("unroll") for (int ct = 0; ct < 2; ++ct) _Pragma("unroll") for (int pt = 0; pt < RT; ++pt)                 \
            acc[ct][pt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A_[ct], X.B_[pt][ks_c], acc[ct][pt], 0, 0, 0);           \
        constexpr int G = 6 * ks_c + T_;                                                       \
        if constexpr (((G + 1) * K) / NG > (G * K) / NG) {                                     \
            tap_row((G * K) / NG);                                                             \
            /* pin the row's FMAs here: pure arithmetic floats freely past a scheduling barrier, an asm with side effects not */ \
            asm volatile("" : "+v"(sacc[0]), "+v"(sacc[1]), "+v"(sacc[2]), "+v"(sacc[3]));     \
        }                                                                                      \
        __builtin_amdgcn_sched_barrier(0);                                                     \
    }
        // (ks is a compile-time constant after unrolling, but not a constant expression: dispatch on it)
        auto groups = [&](auto ksc) __attribute__((always_inline)) {
            constexpr int ks_c = decltype(ksc)::value;
            FX_GROUP(0, al, h) FX_GROUP(1, am, m) FX_GROUP(2, ah, l) FX_GROUP(3, am, h) FX_GROUP(4, ah, m) FX_GROUP(5, ah, h)
        };
        if (ks == 0) groups(std::integral_constant<int, 0>{});
        else if (ks == 1) groups(std::integral_constant<int, 1>{});
        else if (ks == 2) groups(std::integral_constant<int, 2>{});
        else if (ks == 3) groups(std::integral_constant<int, 3>{});
        else if (ks == 4) groups(std::integral_constant<int, 4>{});
        else groups(std::integral_constant<int, 5>{});
#undef FX_GROUP
    }
    // ---- stencil epilogue: D, BN2 partial sums
    // (the sums take every strip pixel through a select, not a branch: results that are only used under a condition get SUNK into
    //  the conditional block by the compiler -- all 200 FMAs of the stencil ended up behind the MFMAs again)
    f32x4 ssum = zero4(), ssq = zero4();
    float* __restrict__ o = Dc + it.doff;
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
        const bool ok = chok && jj < it.npx;
        const f32x4 m = ok ? sacc[jj] : zero4();
        ssum += m;
        ssq += m * m;
        if (ok) st4_nt(o + (size_t)jj * M, sacc[jj]);
    }
    fx_stat_park(ssum, ssq, stat);
    // ---- expand epilogue: BN1 + activation -> next image tile (+ ehat -> E)
    if (do_expand) {
        const float2* cst = reinterpret_cast<const float2*>(P + 96 * RS);
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
            float2 c[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) c[r] = cst[16 * ct + 4 * q + r];
#pragma unroll
            for (int pt = 0; pt < RT; ++pt) {
                f32x4 e, v;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    e[r] = (acc[ct][pt][r] - c[r].x) * c[r].y;
                    v[r] = act_f<ACT>(e[r]);
                }
                st4(tnext + slot[pt] * 32 + 16 * ct + 4 * q, v);
                if (eg && pv[pt]) st4_nt(eg + egoff[pt] + 16 * ct + 4 * q, e);
            }
        }
    }
}

template <int K, int ACT, int KS, int RT>
__device__ __forceinline__ void fx_fwd_body(const TfnasCellDesc& d, const FxPlan& pl, const float* __restrict__ x,
                                            const u8* __restrict__ blob, float* __restrict__ D, float* __restrict__ part,
                                            u8* lds, int ig, const FxSlice sl, float* __restrict__ Eg) {
    constexpr int PAD = K / 2, RS = 64 * KS + FX_RS_PAD, PB = 96 * RS + 256, WBK = K * K * 128;
    constexpr int NVP = (PB + FX_THREADS * 16 - 1) / (FX_THREADS * 16);
    constexpr int NCP = (PB + 1023) / 1024, NCW = (WBK + 1023) / 1024;      // copier: 1 KiB per wave instruction
    constexpr int NCP1 = NCP < 28 ? NCP : 28;                               // pieces held in registers across the barrier
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n = lane & 15, q = lane >> 4;
    const int H = d.H, W = d.W, HW = H * W, ic = d.ic, M = d.M, Ho = d.Ho, Wo = d.Wo;
    const int HP = H + 2 * PAD, WP = (W + 2 * PAD) | 1;
    const int img0 = ig * pl.NI, nimg = min(pl.NI, d.N - img0), NPX = nimg * HW;
    const int g = sl.g, mcp = d.g[g].mcp, goff = d.g[g].off, nch = sl.nch;
    // LDS carve-up (sized on the host for the cell's largest kernel size)
    const int PADM = pl.KMAX / 2;
    const size_t tile_b = (size_t)(pl.NI * (H + 2 * PADM) * ((W + 2 * PADM) | 1) + 8) * 128;
    // (buffers are picked by OFFSET arithmetic on the one LDS base: a pointer taken from an array of pointers is a generic pointer
    //  to the compiler, every access through it becomes a FLAT instruction -- slower than ds_read / ds_write and counted in vmcnt,
    //  i.e. each LDS read then also waits for the global stores in flight: the first version ran 4x slower than its instruction mix)
    const unsigned tile_u = (unsigned)tile_b, pb_u = (unsigned)pl.PB, wb_u = (unsigned)pl.WB;
    auto T = [&](int k) { return reinterpret_cast<float*>(lds + (k & 1) * tile_u); };
    auto Pb = [&](int k) { return lds + 2 * tile_u + (k & 1) * pb_u; };
    auto Wb = [&](int k) { return lds + 2 * tile_u + 2 * pb_u + (k & 1) * wb_u; };
    auto St = [&](int k) { return reinterpret_cast<float*>(lds + 2 * tile_u + 2 * pb_u + 2 * wb_u + (k & 1) * 2048u); };

    // zero both image tiles (the halo stays zero; interiors are rewritten per chunk); first blobs
    for (size_t i = (size_t)tid * 16; i < 2 * tile_b; i += FX_THREADS * 16) *reinterpret_cast<u32x4*>(lds + i) = u32x4{0, 0, 0, 0};
    for (int i = tid; i < 2 * 512; i += FX_THREADS) St(0)[i] = 0.f;           // (the copier wave parks nothing: its rows stay zero)
    const u8* bl0 = blob + (size_t)sl.chunk0 * pl.BLOB;
    {
        FxCopy<NVP> cp;
        FxCopy<1> cw;
        cp.load(bl0, PB);
        cw.load(bl0 + pl.PB, WBK);
        cp.store(Pb(0), PB);
        cw.store(Wb(0), WBK);
        if (nch > 1) {
            cp.load(bl0 + pl.BLOB, PB);
            cp.store(Pb(1), PB);
        }
    }
    float* prow = part + (size_t)ig * 2 * M + 2 * (size_t)goff;

    if (wave == FX_WORKERS) {
        // ------------------------------------------------------------------------------------------------ copier wave
        __syncthreads();
        for (int i = 0; i < nch; ++i) {
            const bool p2 = i + 2 < nch, w1 = i + 1 < nch;
            u32x4 vp[NCP1], vw[NCW];
            const u8* sp = bl0 + (size_t)(p2 ? i + 2 : i) * pl.BLOB;
            const u8* sw = bl0 + (size_t)(w1 ? i + 1 : i) * pl.BLOB + pl.PB;
#pragma unroll
            for (int u = 0; u < NCP1; ++u) {
                const int o = (lane + 64 * u) * 16;
                vp[u] = *reinterpret_cast<const u32x4*>(sp + (o < PB ? o : PB - 16));
            }
#pragma unroll
            for (int u = 0; u < NCW; ++u) {
                const int o = (lane + 64 * u) * 16;
                vw[u] = *reinterpret_cast<const u32x4*>(sw + (o < WBK ? o : WBK - 16));
            }
            FXT(i, 0)
            __syncthreads();
            FXT(i, 1)
            if (i > 0 && lane < 64) {                                      // statistics of chunk i - 1 -> this group's partial row
                const int ch = lane & 31, which = lane >> 5, c0p = sl.c0 + 32 * (i - 1);
                const float* st = St((i - 1) & 1);
                float t = 0.f;
#pragma unroll
                for (int w = 0; w < FX_WORKERS; ++w) t += st[w * 64 + which * 32 + ch];
                if (c0p + ch < mcp) prow[2 * (size_t)(c0p + ch) + which] = t;
            }
            if (p2 && !(FX_ABL & 8)) {
                u8* dp = Pb(i & 1);
#pragma unroll
                for (int u = 0; u < NCP1; ++u) {
                    const int o = (lane + 64 * u) * 16;
                    if (o < PB) *reinterpret_cast<u32x4*>(dp + o) = vp[u];
                }
                if (NCP > NCP1) {                      // (wide blobs: the tail in a second batch -- this wave has nothing else to do)
                    u32x4 vq[NCP > NCP1 ? NCP - NCP1 : 1];
#pragma unroll
                    for (int u = NCP1; u < NCP; ++u) {
                        const int o = (lane + 64 * u) * 16;
                        vq[u - NCP1] = *reinterpret_cast<const u32x4*>(sp + (o < PB ? o : PB - 16));
                    }
#pragma unroll
                    for (int u = NCP1; u < NCP; ++u) {
                        const int o = (lane + 64 * u) * 16;
                        if (o < PB) *reinterpret_cast<u32x4*>(dp + o) = vq[u - NCP1];
                    }
                }
            }
            if (w1 && !(FX_ABL & 8)) {
                u8* dw = Wb((i + 1) & 1);
#pragma unroll
                for (int u = 0; u < NCW; ++u) {
                    const int o = (lane + 64 * u) * 16;
                    if (o < WBK) *reinterpret_cast<u32x4*>(dw + o) = vw[u];
                }
            }
            FXT(i, 2)
        }
        __syncthreads();
        {
            const int ch = lane & 31, which = lane >> 5, c0p = sl.c0 + 32 * (nch - 1);
            const float* st = St((nch - 1) & 1);
            float t = 0.f;
#pragma unroll
            for (int w = 0; w < FX_WORKERS; ++w) t += st[w * 64 + which * 32 + ch];
            if (c0p + ch < mcp) prow[2 * (size_t)(c0p + ch) + which] = t;
        }
        return;
    }
    // ---------------------------------------------------------------------------------------------------- worker waves
    // pixels of the wave's tiles: LDS slot (padded image coordinates) and validity
    int slot[RT];
    bool pv[RT];
    size_t egoff[RT];                                  // stored-ehat mode: the pixel's row in the [pixels][M] tensor
#pragma unroll
    for (int pt = 0; pt < RT; ++pt) {
        const int p = 16 * (wave + FX_WORKERS * pt) + n;
        pv[pt] = p < NPX;
        const int img = p / HW, r = p - img * HW, h = r / W, w = r - h * W;
        slot[pt] = pv[pt] ? (img * HP + h + PAD) * WP + w + PAD : pl.NI * (H + 2 * PADM) * ((W + 2 * PADM) | 1) + 4;
        egoff[pt] = ((size_t)img0 * HW + p) * M + goff;
    }
    // x rows -> split planes (registers, once)
    FxX<KS, RT> X;
#pragma unroll
    for (int pt = 0; pt < RT; ++pt) {
        const int p = 16 * (wave + FX_WORKERS * pt) + n;
        const float* __restrict__ xp = x + ((size_t)img0 * HW + p) * ic + 8 * q;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const bool ok = pv[pt] && 32 * ks + 8 * q < ic;
            const f32x4 a = ok ? ld4(xp + 32 * ks) : zero4(), b = ok ? ld4(xp + 32 * ks + 4) : zero4();
            fx_split8(a, b, X.h[pt][ks], X.m[pt][ks], X.l[pt][ks]);
        }
    }
    // the wave's stencil items: strip (wave * 8 + lane / 8), channel quad lane % 8 -- the same in every chunk
    FxFItem it;
    {
        const int j = wave * 8 + (lane >> 3), cq = lane & 7;
        const int SCN = (Wo + 3) >> 2, per_img = SCN * Ho;
        const int img = j / per_img, r = j - img * per_img, sc = r / Ho, oh = r - sc * Ho, ow0 = 4 * sc;
        it.npx = j < nimg * per_img ? (Wo - ow0 < 4 ? Wo - ow0 : 4) : 0;
        it.toff = ((img * HP + oh) * WP + ow0) * 32 + 4 * cq;
        it.doff = (unsigned)(((img * Ho + oh) * Wo + ow0) * M + 4 * cq);
    }
    float* __restrict__ Dg = D + (size_t)img0 * Ho * Wo * M + goff;
    const int cq = lane & 7;
    __syncthreads();
    fx_expand<KS, RT, ACT>(Pb(0), X, T(0), slot, pv, 0, Eg ? Eg + sl.c0 : nullptr, egoff, 0);
    for (int i = 0; i < nch; ++i) {
        // interval i: stencil of chunk i  ||  MFMAs of chunk i + 1  (the copier brings planes of chunk i + 2, taps of chunk i + 1)
        const bool w1 = i + 1 < nch;
        FXT(i, 0)
        __syncthreads();
        FXT(i, 1)
        const int c0 = sl.c0 + 32 * i;
        const bool chok = c0 + 4 * cq < mcp;
        // (the last interval runs the MFMAs on the stale planes of the buffer and drops the result: no branch in the stream)
        fx_fwd_interval<K, ACT, KS, RT>(Pb((i + 1) & 1), X, T((i + 1) & 1), slot, pv, Eg ? Eg + c0 + 32 : nullptr, egoff, w1,
                                        T(i & 1), reinterpret_cast<const float*>(Wb(i & 1)), it, WP, M, chok, Dg + c0, St(i & 1));
        FXT(i, 2)
    }
    __syncthreads();
}

template <int ACT, int KS, int RT>
__global__ __launch_bounds__(FX_THREADS) void k_fx_fwd(TfnasCellDesc d, FxPlan pl, const float* __restrict__ x,
                                                       const u8* __restrict__ blob, float* __restrict__ D,
                                                       float* __restrict__ part, float* __restrict__ Eg) {
    extern __shared__ __attribute__((aligned(16))) u8 fx_lds[];
    // consecutive workgroups = the image groups of ONE slice: they read the same blobs (L2) at about the same time
    const int si = blockIdx.x / pl.nig, ig = blockIdx.x - si * pl.nig;
    const FxSlice sl = pl.sl[si];
    if (d.g[sl.g].k == 3) fx_fwd_body<3, ACT, KS, RT>(d, pl, x, blob, D, part, fx_lds, ig, sl, Eg);
    else fx_fwd_body<5, ACT, KS, RT>(d, pl, x, blob, D, part, fx_lds, ig, sl, Eg);
}

// ---------------------------------------------------------------------------------------------------------------- backward
// One stencil item per thread (4-pixel strip x channel quad), the same for every chunk: worked out once.
struct FxItem {
    int toff;        // float offset of the window's top-left pixel in the padded image tile (+ 4 cq)
    int eoff;        // float offset of the strip's first pixel in the compact [pixel][32] tile (+ 4 cq)
    int npx;         // valid pixels of the strip; 0: no item
};

// dd = BN2-backward of dZ (dwconv_kernels.hip: bn2_dd): cst2[c] = (mean2, rstd2, R1/Po, R2/Po)
template <int ACT>
__device__ __forceinline__ f32x4 fx_bn2_dd(const f32x4* cst2, f32x4 dz, f32x4 dv, bool has_se, f32x4 gate4, f32x4 dpool4) {
    f32x4 r;
    if (has_se) dz = dz * gate4 + dpool4;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const f32x4 t = cst2[j];
        const float dh = (dv[j] - t.x) * t.y;
        const float ddh = dz[j] * act_d<ACT>(dh);
        r[j] = t.y * (ddh - t.z - dh * t.w);
    }
    return r;
}

template <int K, int ACT>
__device__ __forceinline__ void fx_bwd_stencil(const float* dd, const float* taps, float* eh, const FxItem it, int WP, float* stat) {
    const int lane = threadIdx.x & 63, cq = lane & 7;
    f32x4 t1 = zero4(), t2 = zero4();
    if (it.npx > 0) {
        const float* base = dd + it.toff;
        f32x4 acc[4] = {zero4(), zero4(), zero4(), zero4()};
#pragma unroll 1
        for (int ky = 0; ky < K; ++ky) {
            const float* rowp = base + ky * WP * 32;
            const float* wp = taps + (K * K - 1 - ky * K) * 32 + 4 * cq;          // flipped taps
            f32x4 win[K + 3];
#pragma unroll
            for (int u = 0; u < K + 3; ++u) win[u] = ld4(rowp + u * 32);
#pragma unroll
            for (int kx = 0; kx < K; ++kx) {
                const f32x4 wv = ld4(wp - kx * 32);
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) acc[jj] += win[jj + kx] * wv;
            }
        }
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            if (jj < it.npx) {
                float* ep = eh + it.eoff + jj * 32;
                const f32x4 e = ld4(ep);
                f32x4 de;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    de[r] = acc[jj][r] * act_d<ACT>(e[r]);
                    t1[r] += de[r];
                    t2[r] += de[r] * e[r];
                }
                st4(ep, de);
            }
        }
    }
    fx_stat_park(t1, t2, stat);
}

// LDS bytes of the backward kernel: ehat / dE tile x 2, dd image tile, plane buffer, Wr buffer, cst2, taps x 2, statistics
static size_t fx_bwd_lds(const TfnasCellDesc& d, const FxPlan& pl) {
    const int PAD = pl.KMAX / 2, HP = d.H + 2 * PAD, WP = (d.W + 2 * PAD) | 1;
    const size_t eh = (size_t)(((pl.NI * d.H * d.W + 15) / 16) * 16 + 16) * 128;      // (+ one spare tile: pixels past the end)
    const size_t ddt = (size_t)(pl.NI * HP * WP + 8) * 128;
    return 2 * eh + ddt + (size_t)pl.PB + (size_t)(3 * d.ic * 64) + 512 + 2 * (size_t)pl.WB + 2048;
}

template <int K, int ACT, int CT, int RT>
__device__ __forceinline__ void fx_bwd_body(const TfnasCellDesc& d, const FxPlan& pl, const float* __restrict__ x,
                                            const u8* __restrict__ blob, const float* __restrict__ dZ,
                                            const float* __restrict__ Dt, const float* __restrict__ gate,
                                            const float* __restrict__ dpooled, float* __restrict__ dxp,
                                            float* __restrict__ part, u8* lds, int ig, const FxSlice sl, int si) {
    constexpr int KS = (CT + 1) / 2, PAD = K / 2, RS = 64 * KS + FX_RS_PAD, PB = 96 * RS + 256, WBK = K * K * 128, WRB = 3 * CT * 16 * 64;
    constexpr int NVP = (PB + FX_THREADS * 16 - 1) / (FX_THREADS * 16), NVR = (WRB + FX_THREADS * 16 - 1) / (FX_THREADS * 16);
    constexpr int NR = 2 * RT;                       // rounds of the dd loader: 128 RT pixels x 8 quads / 512 threads
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n = lane & 15, q = lane >> 4, cq = tid & 7;
    const int H = d.H, W = d.W, HW = H * W, ic = d.ic, M = d.M;
    const int HP = H + 2 * PAD, WP = (W + 2 * PAD) | 1;
    const int img0 = ig * pl.NI, nimg = min(pl.NI, d.N - img0), NPX = nimg * HW, ntiles = (NPX + 15) >> 4;
    const int g = sl.g, mcp = d.g[g].mcp, goff = d.g[g].off, nch = sl.nch;
    const bool has_se = d.g[g].se > 0;
    const float inv_hw = 1.f / (float)HW;
    const int PADM = pl.KMAX / 2;
    const int nt16a = ((pl.NI * HW + 15) / 16) * 16;
    const size_t eh_b = (size_t)(nt16a + 16) * 128;
    const size_t dd_b = (size_t)(pl.NI * (H + 2 * PADM) * ((W + 2 * PADM) | 1) + 8) * 128;
    const unsigned eh_u = (unsigned)eh_b;
    auto EH = [&](int k) { return reinterpret_cast<float*>(lds + (k & 1) * eh_u); };       // (offset arithmetic: see fx_fwd_body)
    float* DD = reinterpret_cast<float*>(lds + 2 * eh_b);
    u8* Pb = lds + 2 * eh_b + dd_b;
    u8* Wr = Pb + pl.PB;
    f32x4* C2 = reinterpret_cast<f32x4*>(Wr + WRB);
    const unsigned wb_u = (unsigned)pl.WB;
    auto Wb = [&](int k) { return Wr + WRB + 512 + (k & 1) * wb_u; };
    float* St = reinterpret_cast<float*>(Wr + WRB + 512 + 2 * pl.WB);

    int slot[RT], prw[RT];                           // (compact tile: the pixel's own index; tiles past the end: the spare tile)
    bool pv[RT];
#pragma unroll
    for (int pt = 0; pt < RT; ++pt) {
        prw[pt] = 16 * (wave + 8 * pt) + n;
        pv[pt] = prw[pt] < NPX;
        slot[pt] = prw[pt] < nt16a ? prw[pt] : nt16a + n;
    }
    FxX<KS, RT> X;
#pragma unroll
    for (int pt = 0; pt < RT; ++pt) {
        const float* __restrict__ xp = x + ((size_t)img0 * HW + prw[pt]) * ic + 8 * q;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const bool ok = pv[pt] && 32 * ks + 8 * q < ic;
            const f32x4 a = ok ? ld4(xp + 32 * ks) : zero4(), b = ok ? ld4(xp + 32 * ks + 4) : zero4();
            fx_split8(a, b, X.h[pt][ks], X.m[pt][ks], X.l[pt][ks]);
        }
    }
    // the stencil item of this thread
    FxItem it;
    {
        const int k8 = lane >> 3, j = wave * 8 + k8;
        const int SCN = (W + 3) >> 2, per_img = SCN * H;
        const int img = j / per_img, r = j - img * per_img, sc = r / H, oh = r - sc * H, ow0 = 4 * sc;
        it.npx = (j < nimg * per_img) ? (W - ow0 < 4 ? W - ow0 : 4) : 0;
        it.toff = ((img * HP + oh) * WP + ow0) * 32 + 4 * (lane & 7);
        it.eoff = ((img * H + oh) * W + ow0) * 32 + 4 * (lane & 7);
    }
    // the dd loader's elements of this thread: pixel e >> 3 of the group, channel quad cq
    int lslot[NR];
    unsigned lpix[NR];
    int limg[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        const int pix = (tid + r * FX_THREADS) >> 3;
        const int img = pix / HW, rr = pix - img * HW, h = rr / W, w = rr - h * W;
        lslot[r] = pix < NPX ? ((img * HP + h + PAD) * WP + w + PAD) * 32 + 4 * cq : -1;
        lpix[r] = (unsigned)pix;
        limg[r] = img;
    }
    for (size_t i = (size_t)tid * 16; i < dd_b; i += FX_THREADS * 16) *reinterpret_cast<u32x4*>(lds + 2 * eh_b + i) = u32x4{0, 0, 0, 0};
    const u8* bl0 = blob + (size_t)sl.chunk0 * pl.BLOB;
    {
        FxCopy<NVP> cp;
        FxCopy<1> cw, cc;
        cp.load(bl0, PB);
        cw.load(bl0 + pl.PB, WBK);
        cc.load(bl0 + pl.PB + pl.WB, 512);
        cp.store(Pb, PB);
        cw.store(Wb(0), WBK);
        cc.store(reinterpret_cast<u8*>(C2), 512);
    }
    f32x4 dx[CT][RT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int pt = 0; pt < RT; ++pt) dx[ct][pt] = zero4();
    __syncthreads();

    const size_t pixbase = (size_t)img0 * HW;
    float* prow = part + (size_t)ig * 2 * M + 2 * (size_t)goff;
    for (int i = 0; i <= nch; ++i) {
        const int c0 = sl.c0 + 32 * i;
        // ---- phase A: dd of chunk i -> LDS, ehat of chunk i (MFMA), dx += dE(i - 1) (rstd . W1)(i - 1)  (MFMA)
        f32x4 dz[NR], dv[NR];
        const bool chok = c0 + 4 * cq < mcp;
        if (i < nch) {
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                const bool ok = lslot[r] >= 0 && chok;
                const size_t a = (pixbase + lpix[r]) * M + goff + c0 + 4 * cq;
                dz[r] = ok ? ld4_nt(dZ + a) : zero4();
                dv[r] = ok ? ld4_nt(Dt + a) : zero4();
            }
        }
        if (i > 0) {
            fx_stat_emit(St, prow, c0 - 32, mcp);
            const float* eh = EH((i - 1) & 1);
            bf16x8 bh[RT], bm[RT], bl[RT];
#pragma unroll
            for (int pt = 0; pt < RT; ++pt) {
                const float* bp = eh + slot[pt] * 32 + 8 * q;
                fx_split8(ld4(bp), ld4(bp + 4), bh[pt], bm[pt], bl[pt]);
            }
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
                const int row = 16 * ct + n;
                const u8* ap = Wr + row * 64 + ((q ^ fx_swz(row)) * 16);
                const bf16x8 ah = *reinterpret_cast<const bf16x8*>(ap);
                const bf16x8 am = *reinterpret_cast<const bf16x8*>(ap + CT * 16 * 64);
                const bf16x8 al = *reinterpret_cast<const bf16x8*>(ap + 2 * CT * 16 * 64);
#pragma unroll
                for (int pt = 0; pt < RT; ++pt) {
                    FX_MFMA6(dx[ct][pt], ah, am, al, bh[pt], bm[pt], bl[pt])
                }
            }
        }
        if (i == nch) break;
        fx_expand<KS, RT, 2>(Pb, X, EH(i & 1), slot, pv, ntiles);
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            if (lslot[r] >= 0) {
                f32x4 v = zero4();
                if (chok) {
                    f32x4 g4 = zero4(), dp4 = zero4();
                    if (has_se) {
                        const size_t o = (size_t)(img0 + limg[r]) * M + goff + c0 + 4 * cq;
                        g4 = ld4(gate + o);
                        dp4 = ld4(dpooled + o) * splat4(inv_hw);
                    }
                    v = fx_bn2_dd<ACT>(C2 + 4 * cq, dz[r], dv[r], has_se, g4, dp4);
                }
                st4(DD + lslot[r], v);
            }
        }
        __syncthreads();
        // ---- phase B: stencil of chunk i (dE overwrites ehat in place); the next chunk's blob pieces come in meanwhile
        FxCopy<NVP> cp;
        FxCopy<NVR> cr;
        FxCopy<1> cw, cc;
        const bool nx = i + 1 < nch;
        const u8* bi = bl0 + (size_t)i * pl.BLOB;
        cr.load(bi + pl.PB + pl.WB + 512, WRB);
        if (nx) {
            cp.load(bi + pl.BLOB, PB);
            cw.load(bi + pl.BLOB + pl.PB, WBK);
            cc.load(bi + pl.BLOB + pl.PB + pl.WB, 512);
        }
        fx_bwd_stencil<K, ACT>(DD, reinterpret_cast<const float*>(Wb(i & 1)), EH(i & 1), it, WP, St);
        cr.store(Wr, WRB);
        if (nx) {
            cp.store(Pb, PB);
            cw.store(Wb((i + 1) & 1), WBK);
            cc.store(reinterpret_cast<u8*>(C2), 512);
        }
        __syncthreads();
    }
    // dx partial of this slice
    float* __restrict__ dst = dxp + (size_t)si * d.N * HW * ic;
#pragma unroll
    for (int pt = 0; pt < RT; ++pt) {
        if (pv[pt]) {
            float* __restrict__ o = dst + (pixbase + prw[pt]) * ic + 4 * q;
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) st4(o + 16 * ct, dx[ct][pt]);
        }
    }
}

// ---- backward, stored-ehat mode: the forward left ehat = BN1(x W1^T) in the cell's E buffer (fx_expand, eg); no recompute, no
// x planes: the workgroup's persistent state is the dx accumulators only.
//   phase A(i): dd(i) = BN2-backward(dZ, D) -> LDS image tile;  dx += dE(i-1) (rstd . W1)(i-1)   (MFMA; A = Wr planes, B = dE tile)
//   phase B(i): stencil (flipped taps) on dd(i), * act'(ehat) (ehat: 4 x 16 B per thread from global, requested before the taps),
//               dE(i) -> LDS, t1 / t2 partial sums;  Wr(i), cst2(i+1), taps(i+1) arrive meanwhile
static size_t fx_bwde_lds(const TfnasCellDesc& d, const FxPlan& pl) {
    const int PAD = pl.KMAX / 2, HP = d.H + 2 * PAD, WP = (d.W + 2 * PAD) | 1;
    const size_t de = (size_t)(((pl.NI * d.H * d.W + 15) / 16) * 16 + 16) * 128;
    const size_t ddt = (size_t)(pl.NI * HP * WP + 8) * 128;
    return de + ddt + (size_t)(3 * d.ic * 64) + 512 + 2 * (size_t)pl.WB + 2048;
}

template <int K, int ACT, int CT, int RT>
__device__ __forceinline__ void fx_bwde_body(const TfnasCellDesc& d, const FxPlan& pl, const u8* __restrict__ blob,
                                             const float* __restrict__ Eh, const float* __restrict__ dZ,
                                             const float* __restrict__ Dt, const float* __restrict__ gate,
                                             const float* __restrict__ dpooled, float* __restrict__ dxp,
                                             float* __restrict__ part, u8* lds, int ig, const FxSlice sl, int si) {
    // Seven worker waves + one copier wave, as in the forward (fx_fwd_body).  Per-wave cycle stamps (tools/fx_timeline.py) of the
    // first version -- every wave loading, computing and copying -- showed ~2 200 of the 14 500 cycles of a chunk spent at the top of
    // phase A waiting for the dZ / D loads issued just before the barrier, and ~1 500 in the stencil epilogue waiting for the blob
    // copies issued just before the taps (the vector-memory counter is in-order: a wait for ehat drags every younger load along).
    // Now: the copier moves Wr / taps / cst2 and writes the statistics rows; a worker issues dZ / D of chunk i + 1 at the TOP of
    // phase B(i) and ehat of chunk i at the top of phase A(i): every load has a whole phase to land before anything waits for it.
    constexpr int PAD = K / 2, WBK = K * K * 128, WRB = 3 * CT * 16 * 64, NW = FX_WORKERS, WT = 64 * NW;
    constexpr int NCR = (WRB + 1023) / 1024, NCW = (WBK + 1023) / 1024;
    constexpr int NR = 2 * RT;                       // rounds of the dd loader: 112 RT pixels x 8 quads / 448 worker threads
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n = lane & 15, q = lane >> 4, cq = tid & 7;
    const int H = d.H, W = d.W, HW = H * W, ic = d.ic, M = d.M;
    const int HP = H + 2 * PAD, WP = (W + 2 * PAD) | 1;
    const int img0 = ig * pl.NI, nimg = min(pl.NI, d.N - img0), NPX = nimg * HW;
    const int g = sl.g, mcp = d.g[g].mcp, goff = d.g[g].off, nch = sl.nch;
    const bool has_se = d.g[g].se > 0;
    const float inv_hw = 1.f / (float)HW;
    const int PADM = pl.KMAX / 2;
    const size_t de_b = (size_t)(((pl.NI * HW + 15) / 16) * 16 + 16) * 128;          // (+ a spare tile: pixel rows past the end)
    const size_t dd_b = (size_t)(pl.NI * (H + 2 * PADM) * ((W + 2 * PADM) | 1) + 8) * 128;
    float* DE = reinterpret_cast<float*>(lds);
    float* DD = reinterpret_cast<float*>(lds + de_b);
    u8* Wr = lds + de_b + dd_b;
    f32x4* C2 = reinterpret_cast<f32x4*>(Wr + WRB);
    const unsigned wb_u = (unsigned)pl.WB;
    auto Wb = [&](int k) { return Wr + WRB + 512 + (k & 1) * wb_u; };
    float* St = reinterpret_cast<float*>(Wr + WRB + 512 + 2 * pl.WB);

    for (size_t i = (size_t)tid * 16; i < dd_b; i += FX_THREADS * 16) *reinterpret_cast<u32x4*>(lds + de_b + i) = u32x4{0, 0, 0, 0};
    const u8* bl0 = blob + (size_t)sl.chunk0 * pl.BLOB;
    {
        FxCopy<1> cw, cc;
        cw.load(bl0 + pl.PB, WBK);
        cc.load(bl0 + pl.PB + pl.WB, 512);
        cw.store(Wb(0), WBK);
        cc.store(reinterpret_cast<u8*>(C2), 512);
    }
    float* prow = part + (size_t)ig * 2 * M + 2 * (size_t)goff;

    if (wave == NW) {
        // ------------------------------------------------------------------------------------------------ copier wave
        __syncthreads();
        for (int i = 0; i <= nch; ++i) {
            // phase A(i): the statistics of chunk i - 1; request Wr(i), taps(i + 1), cst2(i + 1)
            if (i > 0) {
                const int ch = lane & 31, which = lane >> 5, c0p = sl.c0 + 32 * (i - 1);
                float t = 0.f;
#pragma unroll
                for (int w = 0; w < NW; ++w) t += St[w * 64 + which * 32 + ch];
                if (c0p + ch < mcp) prow[2 * (size_t)(c0p + ch) + which] = t;
            }
            if (i == nch) break;
            const bool nx = i + 1 < nch;
            const u8* bi = bl0 + (size_t)i * pl.BLOB, *bn = bl0 + (size_t)(nx ? i + 1 : i) * pl.BLOB;
            u32x4 vr[NCR], vw[NCW], vc;
#pragma unroll
            for (int u = 0; u < NCR; ++u) {
                const int o = (lane + 64 * u) * 16;
                vr[u] = *reinterpret_cast<const u32x4*>(bi + pl.PB + pl.WB + 512 + (o < WRB ? o : WRB - 16));
            }
#pragma unroll
            for (int u = 0; u < NCW; ++u) {
                const int o = (lane + 64 * u) * 16;
                vw[u] = *reinterpret_cast<const u32x4*>(bn + pl.PB + (o < WBK ? o : WBK - 16));
            }
            vc = *reinterpret_cast<const u32x4*>(bn + pl.PB + pl.WB + (lane < 32 ? lane * 16 : 0));
            __syncthreads();
            // phase B(i): nobody reads Wr, cst2 or the other tap buffer now
#pragma unroll
            for (int u = 0; u < NCR; ++u) {
                const int o = (lane + 64 * u) * 16;
                if (o < WRB) *reinterpret_cast<u32x4*>(Wr + o) = vr[u];
            }
            if (nx) {
                u8* dw = Wb((i + 1) & 1);
#pragma unroll
                for (int u = 0; u < NCW; ++u) {
                    const int o = (lane + 64 * u) * 16;
                    if (o < WBK) *reinterpret_cast<u32x4*>(dw + o) = vw[u];
                }
                if (lane < 32) *reinterpret_cast<u32x4*>(reinterpret_cast<u8*>(C2) + lane * 16) = vc;
            }
            __syncthreads();
        }
        return;
    }
    // ---------------------------------------------------------------------------------------------------- worker waves
    int prow_[RT];
    bool pv[RT];
#pragma unroll
    for (int pt = 0; pt < RT; ++pt) {
        prow_[pt] = 16 * (wave + NW * pt) + n;
        pv[pt] = prow_[pt] < NPX;
    }
    FxItem it;
    size_t eaddr;                                    // the item's first pixel in the [pixels][M] tensors (+ group, + quad)
    {
        const int k8 = lane >> 3, j = wave * 8 + k8;
        const int SCN = (W + 3) >> 2, per_img = SCN * H;
        const int img = j / per_img, r = j - img * per_img, sc = r / H, oh = r - sc * H, ow0 = 4 * sc;
        const bool live = j < nimg * per_img;
        it.npx = live ? (W - ow0 < 4 ? W - ow0 : 4) : 0;
        it.toff = live ? ((img * HP + oh) * WP + ow0) * 32 + 4 * (lane & 7) : 4 * (lane & 7);
        it.eoff = live ? ((img * H + oh) * W + ow0) * 32 + 4 * (lane & 7) : 4 * (lane & 7);
        eaddr = live ? ((size_t)img0 * HW + (size_t)(img * H + oh) * W + ow0) * M + goff + 4 * (lane & 7) : (size_t)goff;
    }
    int lslot[NR];
    unsigned lpix[NR];
    int limg[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        const int pix = (tid + r * WT) >> 3;
        const int pc = pix < NPX ? pix : NPX - 1;
        const int img = pc / HW, rr = pc - img * HW, h = rr / W, w = rr - h * W;
        lslot[r] = pix < NPX ? ((img * HP + h + PAD) * WP + w + PAD) * 32 + 4 * cq : -1;
        lpix[r] = (unsigned)pc;
        limg[r] = img;
    }
    f32x4 dx[CT][RT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int pt = 0; pt < RT; ++pt) dx[ct][pt] = zero4();
    const size_t pixbase = (size_t)img0 * HW;
    f32x4 dz[NR], dv[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        const size_t a = (pixbase + lpix[r]) * M + goff + sl.c0 + 4 * cq;
        dz[r] = (FX_ABL & 512) ? splat4(0.25f) : ld4_nt(dZ + a);
        dv[r] = (FX_ABL & 512) ? splat4(0.75f) : ld4_nt(Dt + a);
    }
    __syncthreads();
    for (int i = 0; i <= nch; ++i) {
        const int c0 = sl.c0 + 32 * i;
        const bool chok = c0 + 4 * cq < mcp;
        // ---- phase A: dx += dE(i - 1) Wr(i - 1)  (MFMA);  dd(i) = BN2-backward(dZ, D) -> LDS image tile
        FXT(i, 0)
        f32x4 ev[4];
        {
            const int ce = i < nch ? c0 : c0 - 32;
#pragma unroll
            for (int jj = 0; jj < 4; ++jj)
                ev[jj] = (FX_ABL & 256) ? splat4(0.5f) : ld4_nt(Eh + eaddr + ce + (size_t)(jj < it.npx ? jj : 0) * M);
        }
        FXT(i, 1)
        if (i > 0) {
            bf16x8 bh[RT], bm[RT], bl[RT];
#pragma unroll
            for (int pt = 0; pt < RT; ++pt) {
                const float* bp = DE + prow_[pt] * 32 + 8 * q;
                fx_split8(ld4(bp), ld4(bp + 4), bh[pt], bm[pt], bl[pt]);
            }
#pragma unroll
            for (int ct0 = 0; ct0 < CT; ct0 += 2) {
                bf16x8 ah[2], am[2], al[2];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int row = 16 * (ct0 + u < CT ? ct0 + u : ct0) + n;
                    const u8* ap = Wr + row * 64 + ((q ^ fx_swz(row)) * 16);
                    ah[u] = *reinterpret_cast<const bf16x8*>(ap);
                    am[u] = *reinterpret_cast<const bf16x8*>(ap + CT * 16 * 64);
                    al[u] = *reinterpret_cast<const bf16x8*>(ap + 2 * CT * 16 * 64);
                }
#define FX_GTERM(A_, B_)                                                                                             \
    _Pragma("unroll") for (int u = 0; u < 2; ++u) _Pragma("unroll") for (int pt = 0; pt < RT; ++pt) {                \
        if (ct0 + u < CT && !(FX_ABL & 32))                                                                          \
            dx[ct0 + u < CT ? ct0 + u : ct0][pt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(                          \
                A_[u], B_[pt], dx[ct0 + u < CT ? ct0 + u : ct0][pt], 0, 0, 0);                                       \
    }
                FX_GTERM(al, bh) FX_GTERM(am, bm) FX_GTERM(ah, bl) FX_GTERM(am, bh) FX_GTERM(ah, bm) FX_GTERM(ah, bh)
#undef FX_GTERM
            }
        }
        FXT(i, 2)
        if (i == nch) break;
        // (the MFMAs and the BN2-backward transform below were also tried as ONE interleaved stream, a column-tile pair + a loader
        //  round per scheduling region: 5 750 cycles against 2 200 + 3 170 -- no gain, tools/fx_timeline.py)
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            if (lslot[r] >= 0) {
                f32x4 g4 = zero4(), dp4 = zero4();
                if (has_se) {
                    const size_t o = (size_t)(img0 + limg[r]) * M + goff + c0 + 4 * cq;
                    g4 = ld4(gate + o);
                    dp4 = ld4(dpooled + o) * splat4(inv_hw);
                }
                f32x4 v = (FX_ABL & 128) ? dz[r] + dv[r] : fx_bn2_dd<ACT>(C2 + 4 * cq, dz[r], dv[r], has_se, g4, dp4);
                if (!chok) v = zero4();
                st4(DD + lslot[r], v);
            }
        }
        FXT(i, 3)
        __syncthreads();
        FXT(i, 4)
        // ---- phase B: dZ / D of chunk i + 1 requested; stencil of chunk i (flipped taps), * act'(ehat), dE -> LDS, t1 / t2
        // (requested at the END of the phase instead -- behind the wait for ehat -- the next phase A opens with a ~2 000-cycle wait:
        //  the compiler puts a vmcnt(0) in front of the ehat requests; measured, tools/fx_timeline.py)
        {
            const int cn = i + 1 < nch ? c0 + 32 : c0;
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                const size_t a = (pixbase + lpix[r]) * M + goff + cn + 4 * cq;
                dz[r] = (FX_ABL & 512) ? splat4(0.25f) : ld4_nt(dZ + a);
                dv[r] = (FX_ABL & 512) ? splat4(0.75f) : ld4_nt(Dt + a);
            }
        }
        FXT(i, 5)
        {
            const float* taps = reinterpret_cast<const float*>(Wb(i & 1));
            f32x4 t1 = zero4(), t2 = zero4();
            const float* base = DD + it.toff;
            f32x4 acc[4] = {zero4(), zero4(), zero4(), zero4()};
#pragma unroll 1
            for (int ky = 0; ky < ((FX_ABL & 64) ? 0 : K); ++ky) {
                const float* rowp = base + ky * WP * 32;
                const float* wp = taps + (K * K - 1 - ky * K) * 32 + 4 * cq;      // flipped taps
                f32x4 win[K + 3];
#pragma unroll
                for (int u = 0; u < K + 3; ++u) win[u] = ld4(rowp + u * 32);
#pragma unroll
                for (int kx = 0; kx < K; ++kx) {
                    const f32x4 wv = ld4(wp - kx * 32);
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) acc[jj] += win[jj + kx] * wv;
                }
            }
            FXT(i, 6)
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const bool ok = chok && jj < it.npx;
                f32x4 de;
#pragma unroll
                for (int r = 0; r < 4; ++r) de[r] = ok ? acc[jj][r] * act_d<ACT>(ev[jj][r]) : 0.f;
                t1 += de;
#pragma unroll
                for (int r = 0; r < 4; ++r) t2[r] += ok ? de[r] * ev[jj][r] : 0.f;
                if (jj < it.npx) st4(DE + it.eoff + jj * 32, de);
            }
            fx_stat_park(t1, t2, St);
        }
        FXT(i, 7)
        __syncthreads();
        FXT(i, 10)
    }
    float* __restrict__ dst = dxp + (size_t)si * d.N * HW * ic;
#pragma unroll
    for (int pt = 0; pt < RT; ++pt) {
        if (pv[pt]) {
            float* __restrict__ o = dst + (pixbase + prow_[pt]) * ic + 4 * q;
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) st4(o + 16 * ct, dx[ct][pt]);
        }
    }
}

template <int ACT, int CT, int RT>
__global__ __launch_bounds__(FX_THREADS) void k_fx_bwde(TfnasCellDesc d, FxPlan pl, const u8* __restrict__ blob,
                                                        const float* __restrict__ Eh, const float* __restrict__ dZ,
                                                        const float* __restrict__ Dt, const float* __restrict__ gate,
                                                        const float* __restrict__ dpooled, float* __restrict__ dxp,
                                                        float* __restrict__ part) {
    extern __shared__ __attribute__((aligned(16))) u8 fx_lds[];
    const int si = blockIdx.x / pl.nig, ig = blockIdx.x - si * pl.nig;
    const FxSlice sl = pl.sl[si];
    if (d.g[sl.g].k == 3) fx_bwde_body<3, ACT, CT, RT>(d, pl, blob, Eh, dZ, Dt, gate, dpooled, dxp, part, fx_lds, ig, sl, si);
    else fx_bwde_body<5, ACT, CT, RT>(d, pl, blob, Eh, dZ, Dt, gate, dpooled, dxp, part, fx_lds, ig, sl, si);
}

template <int ACT, int CT, int RT>
__global__ __launch_bounds__(FX_THREADS) void k_fx_bwd(TfnasCellDesc d, FxPlan pl, const float* __restrict__ x,
                                                       const u8* __restrict__ blob, const float* __restrict__ dZ,
                                                       const float* __restrict__ Dt, const float* __restrict__ gate,
                                                       const float* __restrict__ dpooled, float* __restrict__ dxp,
                                                       float* __restrict__ part) {
    extern __shared__ __attribute__((aligned(16))) u8 fx_lds[];
    const int si = blockIdx.x / pl.nig, ig = blockIdx.x - si * pl.nig;
    const FxSlice sl = pl.sl[si];
    if (d.g[sl.g].k == 3) fx_bwd_body<3, ACT, CT, RT>(d, pl, x, blob, dZ, Dt, gate, dpooled, dxp, part, fx_lds, ig, sl, si);
    else fx_bwd_body<5, ACT, CT, RT>(d, pl, x, blob, dZ, Dt, gate, dpooled, dxp, part, fx_lds, ig, sl, si);
}

// ---------------------------------------------------------------------------------------------------------------- BN1 statistics
// sum_p E and sum_p E^2 per mid channel from x alone (efree_kernels.hip has the algebra and the ic <= 40 kernels): the centred
// Gram matrix C = sum_p (x - xbar)(x - xbar)^T on the fp32 matrix cores for ic up to 192 -- wave w owns the column tiles
// j = w, w + 4, ..., all row tiles -- and the quadratic forms w_m^T C w_m in double.
template <int CT>
__global__ __launch_bounds__(256) void k_fx_gram(const float* __restrict__ x, int P, int ic, int rps,
                                                 const double* __restrict__ xsum, float* __restrict__ part) {
    // grid = (row blocks, 4): workgroup y owns the column tiles j = y, y + 4, ..; its wave w the row tiles i = w, w + 4, ..
    constexpr int NT = (CT + 3) / 4;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n = lane & 15, q = lane >> 4, jy = blockIdx.y;
    const int r0 = blockIdx.x * rps, r1 = min(P, r0 + rps);
    float mui[NT], muj[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int it = wave + 4 * t, jt = jy + 4 * t;
        mui[t] = it < CT ? (float)(xsum[16 * it + n] / (double)P) : 0.f;
        muj[t] = jt < CT ? (float)(xsum[16 * jt + n] / (double)P) : 0.f;
    }
    f32x4 acc[NT][NT];
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = zero4();
    // 32 rows per iteration: all their loads are issued before the first MFMA (one dependent memory round trip per iteration)
    constexpr int RU = 8;
    for (int p = r0; p < r1; p += 4 * RU) {
        float a[RU][NT], b[RU][NT];
#pragma unroll
        for (int u = 0; u < RU; ++u) {
            const int row = p + 4 * u + q;
            const float* __restrict__ xr = x + (size_t)(row < r1 ? row : r1 - 1) * ic + n;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int it = wave + 4 * t, jt = jy + 4 * t;
                const float va = xr[16 * (it < CT ? it : 0)], vb = xr[16 * (jt < CT ? jt : 0)];
                a[u][t] = (row < r1 && it < CT) ? va - mui[t] : 0.f;
                b[u][t] = (row < r1 && jt < CT) ? vb - muj[t] : 0.f;
            }
        }
#pragma unroll
        for (int u = 0; u < RU; ++u)
#pragma unroll
            for (int i = 0; i < NT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][i], b[u][j], acc[i][j], 0, 0, 0);
    }
    // acc[i][j][r] of lane (n, q) = C[16 (wave + 4 i) + 4 q + r][16 (jy + 4 j) + n]
    float* __restrict__ out = part + (size_t)blockIdx.x * ic * ic;
#pragma unroll
    for (int i = 0; i < NT; ++i) {
        const int it = wave + 4 * i;
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int jt = jy + 4 * j;
            if (it < CT && jt < CT) {
#pragma unroll
                for (int r = 0; r < 4; ++r) out[(size_t)(16 * it + 4 * q + r) * ic + 16 * jt + n] = acc[i][j][r];
            }
        }
    }
}

// stats1[2 col] = sum_p E, [2 col + 1] = sum_p E^2 = w_m^T C w_m + (sum)^2 / P.  T = W C on the fp64 matrix cores
// (v_mfma_f64_16x16x4_f64; gfx950 runs VECTOR fp64 FMAs at a small fraction of that rate: four vector-FMA forms of this kernel
// took 50-290 us): one wave per 16 channels, CT = ic / 16 column tiles of T in registers, A = the 16 weight rows (fp32 -> fp64
// on load), B = rows of C (every wave streams the same C: L2); then q_m = <T_m, w_m> lane-wise + a 16-lane reduction.
typedef double f64x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ const float* fx_wrow(const TfnasCellDesc& d, int col) {
    if (col >= d.M) return nullptr;
    int g = 0;
    for (; g < d.G - 1; ++g)
        if (col < d.g[g + 1].off) break;
    const int m = col - d.g[g].off;
    return (m >= 0 && m < d.g[g].mc) ? d.g[g].w_expand + (size_t)m * d.ic : nullptr;
}
template <int CT>
__global__ __launch_bounds__(256) void k_fx_stats1(TfnasCellDesc d, const double* __restrict__ xsum, const double* __restrict__ C,
                                                   double* __restrict__ stats1) {
    // 4 waves per 16 channels: wave w owns the column tiles t = w, w + 4, .. of T (one wave alone is a chain of CT x ic / 4
    // 64-cycle fp64 MFMAs with a quarter of a wave per SIMD on the chip); the q / s dot products are summed across waves in LDS
    constexpr int NT = (CT + 3) / 4;
    __shared__ double red[2][4][16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n = lane & 15, q = lane >> 4, ic = d.ic, col0 = blockIdx.x * 16;
    const float* __restrict__ wa = fx_wrow(d, col0 + n);                 // A operand: row n, k = q
    f64x4 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = f64x4{0.0, 0.0, 0.0, 0.0};
    const float* __restrict__ wac = wa ? wa : d.g[0].w_expand;           // (always a valid address: the load is unconditional)
    int ctile[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) ctile[t] = wave + 4 * t < CT ? 16 * (wave + 4 * t) : 0;
    constexpr int PD = 4;
    float an[PD];
    double bn[PD][NT];
    const int nks = ic >> 2;
#pragma unroll
    for (int u = 0; u < PD; ++u) {
        const int k = 4 * (u < nks ? u : nks - 1) + q;
        an[u] = wac[k];
#pragma unroll
        for (int t = 0; t < NT; ++t) bn[u][t] = C[(size_t)k * ic + n + ctile[t]];
    }
    for (int ks = 0; ks < nks; ks += PD) {
#pragma unroll
        for (int u = 0; u < PD; ++u) {
            const bool live = ks + u < nks;                                 // (wave-uniform; a dead step multiplies by a = 0)
            const double a = (wa && live) ? (double)an[u] : 0.0;
            double b[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) b[t] = bn[u][t];
            const int kn = ks + u + PD;
            const int k = 4 * (kn < nks ? kn : nks - 1) + q;
            an[u] = wac[k];
#pragma unroll
            for (int t = 0; t < NT; ++t) bn[u][t] = C[(size_t)k * ic + n + ctile[t]];
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b[t], acc[t], 0, 0, 0);
        }
    }
    // acc[t][i] = T[row q + 4 i][column ctile[t] + n]
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int col = col0 + q + 4 * i;
        const float* __restrict__ wr = fx_wrow(d, col);
        double qf = 0.0, sm = 0.0;
        if (wr) {
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                if (wave + 4 * t < CT) {
                    const double wv = (double)wr[ctile[t] + n];
                    qf += acc[t][i] * wv;
                    sm += wv * xsum[ctile[t] + n];
                }
            }
        }
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) {
            qf += __shfl_xor(qf, o, 64);
            sm += __shfl_xor(sm, o, 64);
        }
        if (n == 0) {
            red[0][wave][q + 4 * i] = sm;
            red[1][wave][q + 4 * i] = qf;
        }
    }
    __syncthreads();
    if (tid < 16 && col0 + tid < d.M) {
        const double sm = (red[0][0][tid] + red[0][1][tid]) + (red[0][2][tid] + red[0][3][tid]);
        const double qf = (red[1][0][tid] + red[1][1][tid]) + (red[1][2][tid] + red[1][3][tid]);
        const double P = (double)d.N * d.H * d.W;
        stats1[2 * (size_t)(col0 + tid) + 0] = sm;
        stats1[2 * (size_t)(col0 + tid) + 1] = qf + sm * sm / P;          // sum E^2 = centred sum of squares + P mean^2
    }
}

// scratch (`part`): partial rows from the bottom, the double results xsum[ic] | C[ic * ic] in the top
int launch_fx_stats(const TfnasCellDesc& d, const float* x, double* stats1, float* part, hipStream_t s) {
    const int P = d.N * d.H * d.W, ic = d.ic, ne = ic * ic;
    int nb = 256;        // row blocks = partial Gram matrices, x 4 column-tile workgroups each (64 blocks: slower, the loop is a latency chain)
    while (nb > 1 && (size_t)nb * ne + 2 * (size_t)(ne + ic + 4) + 64 > TFNAS_PART_FLOATS) nb >>= 1;
    int rps = cdiv(P, nb);
    rps = (rps + 3) & ~3;
    if (rps < 16) rps = 16;
    nb = cdiv(P, rps);
    double* xsum = reinterpret_cast<double*>(part + TFNAS_PART_FLOATS) - (ne + ic + 2);
    xsum = reinterpret_cast<double*>((uintptr_t)xsum & ~(uintptr_t)15);
    double* Cm = xsum + ic;
    int rc = launch_x_colsum(x, P, ic, rps, nb, part, s);
    if (rc) return rc;
    rc = launch_reduce_rows(part, nb, ic, (size_t)ic, xsum, nullptr, s);
    if (rc) return rc;
    {
        ProfScope _prof(TK_EXPAND_FWD, s);
        const int CT = ic / 16;
        switch (CT) {
#define FX_GRAM(N_) case N_: hipLaunchKernelGGL(k_fx_gram<N_>, dim3(nb, 4), dim3(256), 0, s, x, P, ic, rps, xsum, part); break;
            FX_GRAM(4) FX_GRAM(5) FX_GRAM(6) FX_GRAM(7) FX_GRAM(8) FX_GRAM(9) FX_GRAM(10) FX_GRAM(11) FX_GRAM(12)
#undef FX_GRAM
            default: return TFNAS_EINVAL;
        }
    }
    rc = launch_reduce_rows(part, nb, ne, (size_t)ne, Cm, nullptr, s);
    if (rc) return rc;
    ProfScope _prof(TK_SMALL, s);
    switch (ic / 16) {
#define FX_ST1(N_) case N_: hipLaunchKernelGGL(k_fx_stats1<N_>, dim3(cdiv(d.M, 16)), dim3(256), 0, s, d, xsum, Cm, stats1); break;
        FX_ST1(4) FX_ST1(5) FX_ST1(6) FX_ST1(7) FX_ST1(8) FX_ST1(9) FX_ST1(10) FX_ST1(11) FX_ST1(12)
#undef FX_ST1
        default: return TFNAS_EINVAL;
    }
    return (int)hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------- launchers
static bool fx_attr_done(const void* fn, size_t shm) {
    return hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm) == hipSuccess;
}

int launch_fx_fwd(const TfnasCellDesc& d, const float* x, const double* stats1, float* E, float* D, double* stats2,
                  float* part, hipStream_t s) {
    FxPlan pl;
    if (!fx_plan(d, pl, false)) return TFNAS_EINVAL;
    const size_t rows = ((size_t)pl.nig * 2 * d.M + 63) & ~(size_t)63;
    u8* blob = reinterpret_cast<u8*>(part + rows);
    {
        ProfScope _prof(TK_SMALL, s);
        hipLaunchKernelGGL(k_fx_pack, dim3(pl.nchunks, 4), dim3(256), 0, s, d, pl, stats1, (const double*)nullptr,
                           (const double*)nullptr, blob);
    }
    const size_t shm = fx_fwd_lds(d, pl);
    const dim3 grid(pl.nig * pl.nslices);
    {
        ProfScope _prof(TK_DW_FWD, s, d.G > 2);
#define FX_FWD(A_, KS_, RT_)                                                                                        \
    {                                                                                                               \
        static bool attr = fx_attr_done((const void*)k_fx_fwd<A_, KS_, RT_>, 160 * 1024);                           \
        if (!attr) return TFNAS_EINVAL;                                                                             \
        hipLaunchKernelGGL((k_fx_fwd<A_, KS_, RT_>), grid, dim3(FX_THREADS), shm, s, d, pl, x, blob, D, part, E);   \
    }
#define FX_FWD_A(KS_, RT_)                                                       \
    {                                                                            \
        if (d.act == TFNAS_ACT_RELU) FX_FWD(0, KS_, RT_) else FX_FWD(1, KS_, RT_) \
    }
        const int key = pl.KS * 10 + pl.RTF;
        switch (key) {
            case 21: FX_FWD_A(2, 1) break;
            case 22: FX_FWD_A(2, 2) break;
            case 31: FX_FWD_A(3, 1) break;
            case 32: FX_FWD_A(3, 2) break;
            case 41: FX_FWD_A(4, 1) break;
            case 42: FX_FWD_A(4, 2) break;
            case 51: FX_FWD_A(5, 1) break;
            case 61: FX_FWD_A(6, 1) break;
            default: return TFNAS_EINVAL;
        }
#undef FX_FWD_A
#undef FX_FWD
    }
    int rc = (int)hipGetLastError();
    if (rc) return rc;
    return launch_reduce_rows(part, pl.nig, 2 * d.M, 2 * (size_t)d.M, stats2, nullptr, s);
}

// TFNAS_ROUTE_FX_OFF in the launch's descriptor: the materialised route instead of the fused per-image kernels (ABI 4: no environment
// variable is read here; a timing build can flip the default with -DTFNAS_FX_DEFAULT=0)
#ifndef TFNAS_FX_DEFAULT
#define TFNAS_FX_DEFAULT 1
#endif
static bool fx_enabled(const TfnasCellDesc& d) { return TFNAS_FX_DEFAULT != 0 && !(d.route & TFNAS_ROUTE_FX_OFF); }

// scratch layout of the backward: dxp [nsl + 1][P][ic] floats | blobs (256-byte aligned) -- in `scratch` (the cell's dEh buffer,
// which the fused route never uses for dE) when it is large enough, else the blobs go behind the statistics rows in `part`
static bool fx_bwd_layout(const TfnasCellDesc& d, const FxPlan& pl, size_t scratch_floats, size_t& blob_off_scratch,
                          size_t& blob_off_part) {
    const size_t P = (size_t)d.N * d.H * d.W;
    const size_t dxp = (((size_t)(pl.nslices + 1) * P * d.ic) + 63) & ~(size_t)63;
    const size_t blobs = ((size_t)pl.nchunks * pl.BLOB + 3) / 4;
    if (dxp > scratch_floats) return false;
    blob_off_scratch = blob_off_part = ~(size_t)0;
    if (dxp + blobs <= scratch_floats) {
        blob_off_scratch = dxp;
        return true;
    }
    const size_t rows = ((size_t)pl.nig * 2 * d.M + 63) & ~(size_t)63;
    if (rows + blobs + 256 > TFNAS_PART_FLOATS) return false;
    blob_off_part = rows;
    return true;
}

bool fx_supported(const TfnasCellDesc& d) {
    FxPlan pl;
    if (!fx_enabled(d) || !fx_plan(d, pl, false)) return false;
    // the fused kernels' expand / dgrad products exist in the split-bf16 x3 arithmetic only: a launch in another mode (the
    // descriptor's own, else the process default) takes the materialised route, so that one model runs ONE arithmetic in both
    // step kinds
    if (((d.gemm_mode & TFNAS_GEMM_EXPLICIT) ? (d.gemm_mode & 7) : gemm_mode()) != TFNAS_GEMM_X3) return false;
    if (fx_fwd_lds(d, pl) > 160 * 1024) return false;
    // forward scratch in `part`: statistics partial rows | blobs | (top) xsum, C
    const size_t rows = (((size_t)pl.nig * 2 * d.M) + 63) & ~(size_t)63, blobs = ((size_t)pl.nchunks * pl.BLOB + 3) / 4;
    const size_t top = 2 * (size_t)(d.ic * d.ic + d.ic + 4) + 64;
    if (rows + blobs + top + 256 > TFNAS_PART_FLOATS) return false;
    FxPlan pb;
    if (!fx_plan(d, pb, true)) return false;
    if (fx_bwd_lds(d, pb) > 160 * 1024 || fx_bwde_lds(d, pb) > 160 * 1024) return false;
    size_t a, b;
    return fx_bwd_layout(d, pb, (size_t)d.N * d.H * d.W * d.M, a, b);
}

int launch_fx_bwd(const TfnasCellDesc& d, const float* x, const float* Eh, const double* stats1, const double* stats2,
                  const double* red2, const float* dZ, const float* D, const float* gate, const float* dpooled, float* scratch,
                  size_t scratch_floats, double* red1, float* cb1, float* part, int* nsl, hipStream_t s) {
    FxPlan pl;
    if (!fx_plan(d, pl, true)) return TFNAS_EINVAL;
    size_t bo_s, bo_p;
    if (!fx_bwd_layout(d, pl, scratch_floats, bo_s, bo_p)) return TFNAS_ERANGE;
    u8* blob = reinterpret_cast<u8*>(bo_s != ~(size_t)0 ? scratch + bo_s : part + bo_p);
    {
        ProfScope _prof(TK_SMALL, s);
        hipLaunchKernelGGL(k_fx_pack, dim3(pl.nchunks, 4), dim3(256), 0, s, d, pl, stats1, stats2, red2, blob);
    }
    const size_t shm = Eh ? fx_bwde_lds(d, pl) : fx_bwd_lds(d, pl);
    const dim3 grid(pl.nig * pl.nslices);
    {
        ProfScope _prof(TK_DW_BWD_DATA, s, d.G > 2);
#define FX_BWD(A_, CT_, RT_)                                                                                         \
    {                                                                                                                \
        static bool attr = fx_attr_done((const void*)k_fx_bwd<A_, CT_, RT_>, 160 * 1024) &&                          \
                           fx_attr_done((const void*)k_fx_bwde<A_, CT_, RT_>, 160 * 1024);                           \
        if (!attr) return TFNAS_EINVAL;                                                                              \
        if (Eh)                                                                                                      \
            hipLaunchKernelGGL((k_fx_bwde<A_, CT_, RT_>), grid, dim3(FX_THREADS), shm, s, d, pl, blob, Eh, dZ, D,    \
                               gate, dpooled, scratch, part);                                                        \
        else                                                                                                         \
            hipLaunchKernelGGL((k_fx_bwd<A_, CT_, RT_>), grid, dim3(FX_THREADS), shm, s, d, pl, x, blob, dZ, D,      \
                               gate, dpooled, scratch, part);                                                        \
    }
#define FX_BWD_A(CT_, RT_)                                                       \
    {                                                                            \
        if (d.act == TFNAS_ACT_RELU) FX_BWD(0, CT_, RT_) else FX_BWD(1, CT_, RT_) \
    }
        const int key = (d.ic / 16) * 10 + (Eh ? pl.RTF : pl.RT);
        switch (key) {
            case 41: FX_BWD_A(4, 1) break;
            case 42: FX_BWD_A(4, 2) break;
            case 51: FX_BWD_A(5, 1) break;
            case 52: FX_BWD_A(5, 2) break;
            case 61: FX_BWD_A(6, 1) break;
            case 62: FX_BWD_A(6, 2) break;
            case 71: FX_BWD_A(7, 1) break;
            case 72: FX_BWD_A(7, 2) break;
            case 81: FX_BWD_A(8, 1) break;
            case 82: FX_BWD_A(8, 2) break;
            case 91: FX_BWD_A(9, 1) break;
            case 101: FX_BWD_A(10, 1) break;
            case 111: FX_BWD_A(11, 1) break;
            case 121: FX_BWD_A(12, 1) break;
            default: return TFNAS_EINVAL;
        }
#undef FX_BWD_A
#undef FX_BWD
    }
    int rc = (int)hipGetLastError();
    if (rc) return rc;
    *nsl = pl.nslices;
    return launch_reduce_bn1(d, part, pl.nig, stats1, red1, cb1, s);
}
