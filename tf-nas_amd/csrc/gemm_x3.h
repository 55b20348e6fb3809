// Split-bf16 MFMA main loop for the row-tiled GEMMs (the A-direct loop of gemm_core.h on the bf16 matrix pipe).
//
// gfx950 runs v_mfma_f32_16x16x4_f32 at the fp32 VECTOR rate (157 TF/s, 1/16 of the bf16 MFMA rate; there is no TF32).  An
// fp32 value splits EXACTLY into three bf16 (truncating split: 8 + 8 + 8 significand bits, v = h + m + l), products of two
// bf16 are exact in fp32, and the MFMA accumulates in fp32, so
//     a*b  =  h h' + (h m' + m h') + (h l' + m m' + l h')  +  O(2^-23 |a b|)
// -- six v_mfma_f32_16x16x32_bf16 (K = 32, ~17 cycles each) replace eight v_mfma_f32_16x16x4_f32 (K = 4, 32 cycles each)
// per 16x16x32 block: 2.5x less matrix-pipe time at fp32-level accuracy (measured, tools/ubench/gemm_x3.hip: max error /
// sum|a b| 1.9e-7 vs 2.8e-7 of the fp32 MFMA chain; 90 -> 132-166 TF/s-equivalent on the late-cell shapes with plain loaders).
// TERMS = 3 keeps the 2^-16 terms only (three MFMAs), TERMS = 1 is plain bf16 (the reduced-precision training mode of the
// derived network, train_eval_amp.py in the reference).
//
// Same functor interface as gemm_mainloop_adirect (gemm_core.h): pre(c), la / xa (the lane's two rows, kl = 4 (lane >> 4)),
// lb / xb, K-chunks of 16 -- the kernels' loaders are unchanged.  One MFMA step consumes TWO consecutive chunks: K slot
// (lk, e) of the 16x16x32 instruction carries k = 16 (e >> 2) + 4 lk + (e & 3) of the chunk pair, for A and B alike (a sum
// over k does not care about the order).  A goes from global memory straight into MFMA lanes (split in registers); B is
// staged in LDS as bf16 planes [plane][n][32 k], 64 bytes per row, the 16-byte slot of k-group lk XOR-permuted by the
// row's quad so that the four 16-lane groups of a ds_read_b128 (MI355X_MICROARCH.md, LDS table: they mix lk with lk ^ 1)
// hit 64 distinct banks without padding.  When the B source is N-contiguous (the data-gradient GEMMs: W[k][n .. n+3]) the
// planes keep that orientation -- [16-column tile][k / 4][4 k][16 n] blocks of 128 bytes, tile stride 1056 bytes (writes of a
// 16-lane group then cover 32 distinct banks) -- and the fragment is fetched with two transposing ds_read_b64_tr_b16 (each
// 16-lane group reads one contiguous block and receives its column): no register transposes, the same quads and the same
// loads as the fp32 loop.  (A first version transposed 4 x 4 blocks in registers: its ds_write_b64 were 16-way bank
// conflicted and the data-gradient kernels ran 15-35 % SLOWER than with fp32 MFMA.)
#pragma once
#include "gemm_core.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned x3_bits(float v) { return __builtin_bit_cast(unsigned, v); }
__device__ __forceinline__ float x3_float(unsigned v) { return __builtin_bit_cast(float, v); }
// upper halves (= truncated bf16) of two floats in one dword: low half <- a, high half <- b
__device__ __forceinline__ unsigned x3_pack(unsigned a, unsigned b) { return __builtin_amdgcn_perm(b, a, 0x07060302u); }

// Non-finite and tiny operands: an Inf operand splits into h = Inf, r1 = Inf - Inf = NaN, so the products carry NaN
// where the fp32 MFMA loop would carry Inf -- non-finite either way, never a finite wrong number (bench.py and the search
// loop's finiteness check treat both the same; a guard would cost two VALU operations per operand element in the hottest loop).
// bf16 has fp32's exponent range, so the residues m / l only vanish where fp32 itself goes denormal (|v| < 2^-126 * 2^16 for l),
// and the matrix core flushes denormal inputs in both forms.
struct X3Planes { u32x2 h, m, l; };   // four values -> three planes of four bf16
template <int TERMS>
__device__ __forceinline__ X3Planes x3_split4(f32x4 v) {
    unsigned hb[4], mb[4], lb[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        hb[e] = x3_bits(v[e]);
        mb[e] = lb[e] = 0;
        if (TERMS > 1) {
            const float r1 = v[e] - x3_float(hb[e] & 0xffff0000u);       // exact
            mb[e] = x3_bits(r1);
            if (TERMS > 3) lb[e] = x3_bits(r1 - x3_float(mb[e] & 0xffff0000u));   // exact, <= 8 significant bits left
        }
    }
    X3Planes p;
    p.h = u32x2{x3_pack(hb[0], hb[1]), x3_pack(hb[2], hb[3])};
    p.m = u32x2{x3_pack(mb[0], mb[1]), x3_pack(mb[2], mb[3])};
    p.l = u32x2{x3_pack(lb[0], lb[1]), x3_pack(lb[2], lb[3])};
    return p;
}
__device__ __forceinline__ int x3_fperm(int q) { return (0x1320 >> (4 * q)) & 3; }

template <int NT, int TERMS>
struct GX3 {
    static constexpr int BN = 16 * NT;
    static constexpr int NPL = TERMS > 3 ? 3 : (TERMS > 1 ? 2 : 1);
    static constexpr int TS = 1056;                  // transposed-read layout: bytes per 16-column tile (8 blocks + 32)
    static constexpr int PLANE = BN * 64;            // bytes, K-contiguous layout
    static constexpr int PLANE_T = NT * TS;
    static constexpr int BUF = NPL * PLANE, BUF_T = NPL * PLANE_T;
    static constexpr int LDS_BYTES = 2 * (BUF_T > BUF ? BUF_T : BUF);
};

template <int NT, bool BKC, int TERMS, class PRE, class LA, class XA, class LB, class XB>
__device__ __forceinline__ void gemm_mainloop_adirect_x3(PRE& pre, LA& la, XA& xa, LB& lb, XB& xb, int nchunks,
                                                         f32x4 (&acc)[2][NT], float* lds_f) {
    using T = GT<NT>;
    using X = GX3<NT, TERMS>;
    static_assert(X::LDS_BYTES <= T::LDS_FLOATS * 4, "x3 B planes do not fit the GEMM LDS buffer");
    using RA = decltype(la(0, 0, 0));
    using RB = decltype(lb(0, 0, 0));
    unsigned char* lds = reinterpret_cast<unsigned char*>(lds_f);
    const int tid = threadIdx.x, lane = tid & 63;
    const int lr = lane & 15, lk = lane >> 4;
    // B staging items of one chunk (one half of an MFMA step): B_ITERS quads per thread, as in the fp32 loop
    //   BKC : (n, kl) = 4 consecutive k of column n        !BKC: (kl, n4) = columns n4 .. n4 + 3 of row kl
    constexpr int NB = T::B_ITERS, Q = T::BN / 4;
    constexpr int BUF = BKC ? X::BUF : X::BUF_T, PLANE = BKC ? X::PLANE : X::PLANE_T;
    RA ra[2][2];
    RB rb[2][NB];
    u32x4 ca[2][3];
    const int nC = (nchunks + 1) >> 1;

    auto gload_half = [&](int c, int h) {
        pre(c);
#pragma unroll
        for (int i = 0; i < 2; ++i) ra[h][i] = la(c, i, 4 * lk);
#pragma unroll
        for (int it = 0; it < NB; ++it) {
            const int idx = tid + 256 * it;
            const int b0 = BKC ? (idx >> 2) : idx / Q, b1 = BKC ? (idx & 3) * 4 : (idx - (idx / Q) * Q) * 4;
            if (idx < T::B_ITEMS) rb[h][it] = lb(c, b0, b1);
        }
    };
    auto gload = [&](int C) {
        gload_half(2 * C, 0);
        if (2 * C + 1 < nchunks) gload_half(2 * C + 1, 1);
    };
    auto sstore_half = [&](int c, int h, bool live, unsigned char* Bs, X3Planes (&pa)[2]) {
        if (live) pre(c);
#pragma unroll
        for (int i = 0; i < 2; ++i) pa[i] = x3_split4<TERMS>(live ? xa(ra[h][i], c, i, 4 * lk) : zero4());
#pragma unroll
        for (int it = 0; it < NB; ++it) {
            const int idx = tid + 256 * it;
            const int b0 = BKC ? (idx >> 2) : idx / Q, b1 = BKC ? (idx & 3) * 4 : (idx - (idx / Q) * Q) * 4;
            if (idx < T::B_ITEMS) {
                const X3Planes p = x3_split4<TERMS>(live ? xb(rb[h][it], c, b0, b1) : zero4());
                // BKC : row n = b0, k-group b1 / 4 -> slot;   !BKC: k = b0 (block 4 h + k / 4, row k & 3), columns b1 .. b1 + 3
                const unsigned off = BKC ? b0 * 64 + (((b1 >> 2) ^ x3_fperm((b0 >> 2) & 3)) * 16) + h * 8
                                         : (b1 >> 4) * X::TS + (4 * h + (b0 >> 2)) * 128 + (b0 & 3) * 32 + ((b1 & 15) >> 2) * 8;
                *reinterpret_cast<u32x2*>(Bs + off) = p.h;
                if (TERMS > 1) *reinterpret_cast<u32x2*>(Bs + PLANE + off) = p.m;
                if (TERMS > 3) *reinterpret_cast<u32x2*>(Bs + 2 * PLANE + off) = p.l;
            }
        }
    };
    auto sstore = [&](int C, unsigned char* Bs) {
        X3Planes p0[2], p1[2];
        sstore_half(2 * C, 0, true, Bs, p0);
        sstore_half(2 * C + 1, 1, 2 * C + 1 < nchunks, Bs, p1);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            ca[i][0] = u32x4{p0[i].h.x, p0[i].h.y, p1[i].h.x, p1[i].h.y};
            ca[i][1] = u32x4{p0[i].m.x, p0[i].m.y, p1[i].m.x, p1[i].m.y};
            ca[i][2] = u32x4{p0[i].l.x, p0[i].l.y, p1[i].l.x, p1[i].l.y};
        }
    };
    // B fragment of column tile j: lane (lr, lk) gets k = 16 h + 4 lk + t, t = 0..3, h = 0, 1 of column 16 j + lr
    const unsigned rd = BKC ? lr * 64 + ((lk ^ x3_fperm(lr >> 2)) * 16) : lk * 128 + lr * 8;
    auto frag = [&](const unsigned char* p, int j) -> bf16x8 {
        if (BKC) return __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(p + rd + j * 1024));
        typedef short s16x4 __attribute__((ext_vector_type(4)));
        typedef short s16x8 __attribute__((ext_vector_type(8)));
        typedef __attribute__((address_space(3))) s16x4* lds_p;
        const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(p + rd + j * X::TS));
        const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(p + rd + j * X::TS + 512));
        const s16x8 v = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
        return __builtin_bit_cast(bf16x8, v);
    };

    if (nC > 0) {
        gload(0);
        sstore(0, lds);
    }
    __syncthreads();
    for (int C = 0; C < nC; ++C) {
        const unsigned char* Bs = lds + (C & 1) * BUF;
        const bool more = C + 1 < nC;
        if (more) gload(C + 1);
        bf16x8 ah[2], am[2], al[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            ah[i] = __builtin_bit_cast(bf16x8, ca[i][0]);
            am[i] = __builtin_bit_cast(bf16x8, ca[i][1]);
            al[i] = __builtin_bit_cast(bf16x8, ca[i][2]);
        }
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const bf16x8 bh = frag(Bs, j);
            bf16x8 bm = bh, bl = bh;
            if (TERMS > 1) bm = frag(Bs + PLANE, j);
            if (TERMS > 3) bl = frag(Bs + 2 * PLANE, j);
#define TFNAS_X3_MF(a, b)                                                                   \
    acc[0][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b, acc[0][j], 0, 0, 0);       \
    acc[1][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], b, acc[1][j], 0, 0, 0);
            if (TERMS > 3) { TFNAS_X3_MF(al, bh) TFNAS_X3_MF(am, bm) TFNAS_X3_MF(ah, bl) }     // smallest terms first
            if (TERMS > 1) { TFNAS_X3_MF(am, bh) TFNAS_X3_MF(ah, bm) }
            TFNAS_X3_MF(ah, bh)
#undef TFNAS_X3_MF
        }
        if (more) sstore(C + 1, lds + ((C + 1) & 1) * BUF);
        __syncthreads();
    }
}

// MM = 0: fp32 MFMA (gemm_mainloop_adirect); MM = 6 / 3 / 1: split-bf16 with that many products per element pair
template <int NT, bool BKC, int MM, class PRE, class LA, class XA, class LB, class XB>
__device__ __forceinline__ void gemm_adirect(PRE& pre, LA& la, XA& xa, LB& lb, XB& xb, int nchunks, f32x4 (&acc)[2][NT],
                                             float* lds) {
    if constexpr (MM == 0) gemm_mainloop_adirect<NT, BKC>(pre, la, xa, lb, xb, nchunks, acc, lds);
    else gemm_mainloop_adirect_x3<NT, BKC, MM>(pre, la, xa, lb, xb, nchunks, acc, lds);
}
