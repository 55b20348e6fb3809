// Block-level fp32 GEMM core for gfx950: 128 x (16*NT) output tile per 256-thread workgroup,
// v_mfma_f32_16x16x4_f32 (exact fp32, == an fmaf chain; 157 TF/s peak = fp32 vector rate), operands
// staged through LDS in K-major layout so that every MFMA operand fetch is a conflict-free ds_read_b32.
//
//   As[k][m]  (leading dim LDA = 144: 144 % 32 == 16 -> lanes 0-15 (k) and 16-31 (k+1) hit disjoint banks)
//   Bs[k][n]  (leading dim LDB chosen the same way)
//
// Operand tiles are produced by *loader functors* so each kernel can fuse its element-wise prologue
// (BatchNorm normalise, activation, SE gate, BatchNorm backward) into the global->LDS staging:
//   AK  = true : A source is K-contiguous  : fa(chunk, row,  kl) -> A(row, kl..kl+3),  row<128, kl in {0,4,8,12}
//   AK  = false: A source is M-contiguous  : fa(chunk, kl,   m ) -> A(m..m+3, kl),     kl<16,  m in {0,4,..,124}
//   BKC = true : B source is K-contiguous  : fb(chunk, n,    kl) -> B(kl..kl+3, n),    n<BN
//   BKC = false: B source is N-contiguous  : fb(chunk, kl,   n ) -> B(kl, n..n+3),     n in {0,4,..,BN-4}
// One K-chunk = 16; global loads of chunk c+1 are issued before the MFMAs of chunk c (register prefetch),
// LDS is double-buffered, one __syncthreads() per chunk.
//
// Accumulator layout (16x16x4 C/D map): acc[i][j][r] = C(wrow + 16*i + 4*(lane>>4) + r, 16*j + (lane&15)),
// wrow = 32 * wave.
#pragma once
#include "tfnas_dev.h"

template <int NT>
struct GT {
    static constexpr int BM = 128, BK = 16, BN = 16 * NT;
    static constexpr int LDA = BM + 16;
    static constexpr int LDB = (BN % 32 == 16) ? BN : BN + 16;
    static constexpr int A_FLOATS = BK * LDA, B_FLOATS = BK * LDB;
    // (the split-bf16 loop of gemm_x3.h stages B as three bf16 planes of 32 k: 2 buffers x 3 planes x NT column tiles x
    //  1056 bytes in the transposed-read layout, x 1024 in the K-contiguous one)
    static constexpr int X3_FLOATS = 1584 * NT;
    static constexpr int LDS_FLOATS = 2 * (A_FLOATS + B_FLOATS) > X3_FLOATS ? 2 * (A_FLOATS + B_FLOATS) : X3_FLOATS;
    static constexpr int B_ITEMS = BN * 4;                 // float4 items of one B tile
    static constexpr int B_ITERS = (B_ITEMS + 255) / 256;
};

template <int NT>
__device__ __forceinline__ void acc_zero(f32x4 (&acc)[2][NT]) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = zero4();
}

struct XfId {   // transform of a one-piece loader: the value is already final
    __device__ __forceinline__ f32x4 operator()(f32x4 v, int, int, int, int) const { return v; }
};

// Two-phase operand loaders.  `la(c, i, i0, i1)` / `lb(c, i, i0, i1)` only ISSUE the global loads of K-chunk c and return the
// raw registers (any struct); i = which of the thread's items (0..1 for A, 0..B_ITERS-1 for B; (i0, i1) of item i are the
// same for every chunk, so kernels keep per-item pointers / constants in registers);
// `xa(raw, c, i, i0, i1)` / `xb(...)` turn them into the f32x4 that goes to LDS (normalise, activate,
// mask rows / columns outside the problem).  The split matters: the loads of chunk c+1 are issued before the MFMAs of chunk
// c and must not be waited for until after them.  With a one-piece loader whose element-wise math sits in the same
// conditional region as its load, the compiler put `s_waitcnt vmcnt(0)` straight after every load -- 2 to 5 serialized DRAM
// round trips per K-chunk in front of the MFMAs (ISA of round 1/2's k_project_* / k_expand_dgrad / k_*_wgrad).  Loaders
// therefore load unconditionally from clamped addresses and mask in the transform.
// BGUARD = false: lb is also called for item slots past the B tile (threads idx >= B_ITEMS; its result is dropped) -- the
// kernel's lb must then be safe for any (i0, i1).  A load inside the divergent `if (idx < B_ITEMS)` region got an
// s_waitcnt vmcnt(0) right behind it (the 5- and 7-tile variants, whose second B item covers only part of the threads).
template <int NT, bool AK, bool BKC, bool BGUARD = true, bool PF2 = false, class LA, class XA, class LB, class XB>
__device__ __forceinline__ void gemm_mainloop2(LA& la, XA& xa, LB& lb, XB& xb, int nchunks, f32x4 (&acc)[2][NT],
                                               float* lds) {
    using T = GT<NT>;
    using RA = decltype(la(0, 0, 0, 0));
    using RB = decltype(lb(0, 0, 0, 0));
    const int tid = threadIdx.x, lane = tid & 63, wrow = (tid >> 6) * 32;
    const int lr = lane & 15, lk = lane >> 4;
    RA ra[2];
    RB rb[T::B_ITERS];

#define TFNAS_A_IDX(i) const int idx = tid + 256 * (i), a0_ = AK ? (idx >> 2) : (idx >> 5), a1_ = AK ? (idx & 3) * 4 : (idx & 31) * 4
#define TFNAS_B_IDX(i)                                                      \
    const int idx = tid + 256 * (i), b0_ = BKC ? (idx >> 2) : idx / (T::BN / 4), \
              b1_ = BKC ? (idx & 3) * 4 : (idx % (T::BN / 4)) * 4
#define TFNAS_GLOAD(c)                                                                      \
    {                                                                                       \
        _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                     \
            TFNAS_A_IDX(i);                                                                 \
            ra[i] = la((c), i, a0_, a1_);                                                      \
        }                                                                                   \
        _Pragma("unroll") for (int i = 0; i < T::B_ITERS; ++i) {                            \
            TFNAS_B_IDX(i);                                                                 \
            if (!BGUARD || idx < T::B_ITEMS) rb[i] = lb((c), i, b0_, b1_);                                \
        }                                                                                   \
    }
#define TFNAS_SSTORE(c, As, Bs)                                                             \
    {                                                                                       \
        _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                     \
            TFNAS_A_IDX(i);                                                                 \
            const f32x4 v = xa(ra[i], (c), i, a0_, a1_);                                       \
            if (AK) {                                                                       \
                (As)[(a1_ + 0) * T::LDA + a0_] = v.x;                                       \
                (As)[(a1_ + 1) * T::LDA + a0_] = v.y;                                       \
                (As)[(a1_ + 2) * T::LDA + a0_] = v.z;                                       \
                (As)[(a1_ + 3) * T::LDA + a0_] = v.w;                                       \
            } else {                                                                        \
                st4(&(As)[a0_ * T::LDA + a1_], v);                                          \
            }                                                                               \
        }                                                                                   \
        _Pragma("unroll") for (int i = 0; i < T::B_ITERS; ++i) {                            \
            TFNAS_B_IDX(i);                                                                 \
            if (idx < T::B_ITEMS) {                                                         \
                const f32x4 v = xb(rb[i], (c), i, b0_, b1_);                                   \
                if (BKC) {                                                                  \
                    (Bs)[(b1_ + 0) * T::LDB + b0_] = v.x;                                   \
                    (Bs)[(b1_ + 1) * T::LDB + b0_] = v.y;                                   \
                    (Bs)[(b1_ + 2) * T::LDB + b0_] = v.z;                                   \
                    (Bs)[(b1_ + 3) * T::LDB + b0_] = v.w;                                   \
                } else {                                                                    \
                    st4(&(Bs)[b0_ * T::LDB + b1_], v);                                      \
                }                                                                           \
            }                                                                               \
        }                                                                                   \
    }

    float* As0 = lds;
    float* As1 = lds + T::A_FLOATS;
    float* Bs0 = lds + 2 * T::A_FLOATS;
    float* Bs1 = Bs0 + T::B_FLOATS;
#define TFNAS_MFMAS(As, Bs)                                                                 \
    _Pragma("unroll") for (int ks = 0; ks < 4; ++ks) {                                      \
        const int k = ks * 4 + lk;                                                          \
        const float a0 = (As)[k * T::LDA + wrow + lr];                                      \
        const float a1 = (As)[k * T::LDA + wrow + 16 + lr];                                 \
        _Pragma("unroll") for (int j = 0; j < NT; ++j) {                                    \
            const float b = (Bs)[k * T::LDB + 16 * j + lr];                                 \
            acc[0][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b, acc[0][j], 0, 0, 0);    \
            acc[1][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b, acc[1][j], 0, 0, 0);    \
        }                                                                                   \
    }

    if (PF2) {
        // prefetch distance 2 for launches that leave a CU with one or two workgroups (SE excite GEMMs: 1-20 workgroups;
        // weight-gradient GEMMs of the 14x14 / 7x7 cells: 1.3 per CU): the loads of chunk c+2 are issued before the MFMAs
        // of chunk c and consumed after those of chunk c+1.  Two raw register sets, alternating statically (loop unrolled
        // by two).  (In the row-tiled GEMMs, which run at 3-4 workgroups per CU under launch bounds, the same scheme
        // was slower: gemm_mainloop_adirect.)
        RA qa[2];
        RB qb[T::B_ITERS];
#define TFNAS_SWAP_SETS()                                                                   \
    {                                                                                       \
        _Pragma("unroll") for (int i = 0; i < 2; ++i) { const RA t_ = ra[i]; ra[i] = qa[i]; qa[i] = t_; }          \
        _Pragma("unroll") for (int i = 0; i < T::B_ITERS; ++i) { const RB t_ = rb[i]; rb[i] = qb[i]; qb[i] = t_; } \
    }
        // ra/rb always name the set the macros work on; the two physical sets trade places by register renaming only
        // (the loop body is unrolled by two, so the swaps are compile-time)
        if (nchunks > 0) {
            TFNAS_GLOAD(0);
            TFNAS_SSTORE(0, As0, Bs0);
        }
        if (nchunks > 1) TFNAS_GLOAD(1);          // set X holds chunk 1
        __syncthreads();
        for (int c = 0; c < nchunks; c += 2) {
            TFNAS_SWAP_SETS();                        // ra = set Y (free), qa = set X (chunk c+1)
            if (c + 2 < nchunks) TFNAS_GLOAD(c + 2);  // set Y <- chunk c+2
            TFNAS_MFMAS(As0, Bs0);
            TFNAS_SWAP_SETS();                        // ra = set X (chunk c+1), qa = set Y (chunk c+2)
            if (c + 1 < nchunks) TFNAS_SSTORE(c + 1, As1, Bs1);
            __syncthreads();
            if (c + 1 >= nchunks) break;
            if (c + 3 < nchunks) TFNAS_GLOAD(c + 3);  // set X <- chunk c+3
            TFNAS_MFMAS(As1, Bs1);
            TFNAS_SWAP_SETS();                        // ra = set Y (chunk c+2), qa = set X (chunk c+3)
            if (c + 2 < nchunks) TFNAS_SSTORE(c + 2, As0, Bs0);
            TFNAS_SWAP_SETS();                        // ra = set X (chunk c+3) -- what the next iteration expects in "X"
            __syncthreads();
        }
#undef TFNAS_SWAP_SETS
    } else {
        if (nchunks > 0) {
            TFNAS_GLOAD(0);
            TFNAS_SSTORE(0, As0, Bs0);
        }
        __syncthreads();
        for (int c = 0; c < nchunks; ++c) {
            const bool more = (c + 1 < nchunks);
            if (more) TFNAS_GLOAD(c + 1);
            if (c & 1) { TFNAS_MFMAS(As1, Bs1); } else { TFNAS_MFMAS(As0, Bs0); }
            if (more) {
                if (c & 1) { TFNAS_SSTORE(c + 1, As0, Bs0); } else { TFNAS_SSTORE(c + 1, As1, Bs1); }
            }
            __syncthreads();
        }
    }
#undef TFNAS_MFMAS
#undef TFNAS_GLOAD
#undef TFNAS_SSTORE
#undef TFNAS_A_IDX
#undef TFNAS_B_IDX
}

// A-direct variant for K-contiguous A operands (AK).  In the LDS-staged loop each wave only ever reads back its OWN 32 rows
// of the A tile, and the float4 a thread loads for the staging store (4 consecutive k of one row) already sits in a lane
// that an MFMA can take it from: lane (lr, lk) asks for A(row = wrow + 16 i + lr, k = 4 lk .. 4 lk + 3) and MFMA step j consumes
// component j, i.e. K slot lk of step j carries k = 4 lk + j.  A sum over k does not care about the order, so the B tile is
// stored with its rows permuted the same way (k -> row (k & 3) * 4 + (k >> 2)) and read exactly as before.  That removes
// the transposing scalar LDS stores of A (the bank-conflicted part of the staging), its fragment reads and half of the LDS
// footprint; global access pattern (64-byte row pieces) and accumulator layout are unchanged.
// (Prefetch distance 2 -- loads of chunk c+2 in flight across two MFMA phases -- was built and measured on top of this:
//  slower everywhere, with and without relaxed launch bounds; the loop is VALU-issue bound in the loaders, not latency bound.)
// Loader signatures here: la(c, i, kl) / xa(raw, c, i, kl) with i in {0, 1} = which of the lane's two rows
// (row = wrow + 16 i + (lane & 15), fixed for the whole K loop -- kernels keep per-row pointers / flags in registers instead
// of recomputing them per chunk) and kl = 4 (lane >> 4);  lb / xb as in gemm_mainloop2;  pre(c) runs once per chunk before
// its loads are issued (chunk -> group bookkeeping shared by the four loader pieces).
struct PreNone {
    __device__ __forceinline__ void operator()(int) const {}
};
template <int NT, bool BKC, class PRE, class LA, class XA, class LB, class XB>
__device__ __forceinline__ void gemm_mainloop_adirect(PRE& pre, LA& la, XA& xa, LB& lb, XB& xb, int nchunks,
                                                      f32x4 (&acc)[2][NT], float* lds) {
    using T = GT<NT>;
    using RA = decltype(la(0, 0, 0));
    using RB = decltype(lb(0, 0, 0));
    const int tid = threadIdx.x, lane = tid & 63;
    const int lr = lane & 15, lk = lane >> 4;
    RA ra[2];
    RB rb[T::B_ITERS];
    f32x4 cur[2];

#define TFNAS_B_IDX(i)                                                      \
    const int idx = tid + 256 * (i), b0_ = BKC ? (idx >> 2) : idx / (T::BN / 4), \
              b1_ = BKC ? (idx & 3) * 4 : (idx % (T::BN / 4)) * 4
#define TFNAS_GLOAD(c)                                                                      \
    {                                                                                       \
        pre(c);                                                                             \
        _Pragma("unroll") for (int i = 0; i < 2; ++i) ra[i] = la((c), i, 4 * lk);           \
        _Pragma("unroll") for (int i = 0; i < T::B_ITERS; ++i) {                            \
            TFNAS_B_IDX(i);                                                                 \
            if (idx < T::B_ITEMS) rb[i] = lb((c), b0_, b1_);                                \
        }                                                                                   \
    }
#define TFNAS_SSTORE(c, Bs)                                                                 \
    {                                                                                       \
        _Pragma("unroll") for (int i = 0; i < 2; ++i) cur[i] = xa(ra[i], (c), i, 4 * lk);   \
        _Pragma("unroll") for (int i = 0; i < T::B_ITERS; ++i) {                            \
            TFNAS_B_IDX(i);                                                                 \
            if (idx < T::B_ITEMS) {                                                         \
                const f32x4 v = xb(rb[i], (c), b0_, b1_);                                   \
                if (BKC) {   /* k = b1_ + t  ->  row 4 t + b1_ / 4 */                       \
                    const int r0_ = b1_ >> 2;                                               \
                    (Bs)[(r0_ + 0) * T::LDB + b0_] = v.x;                                   \
                    (Bs)[(r0_ + 4) * T::LDB + b0_] = v.y;                                   \
                    (Bs)[(r0_ + 8) * T::LDB + b0_] = v.z;                                   \
                    (Bs)[(r0_ + 12) * T::LDB + b0_] = v.w;                                  \
                } else {                                                                    \
                    st4(&(Bs)[((b0_ & 3) * 4 + (b0_ >> 2)) * T::LDB + b1_], v);             \
                }                                                                           \
            }                                                                               \
        }                                                                                   \
    }

    float* Bs0 = lds;
    float* Bs1 = lds + T::B_FLOATS;
    if (nchunks > 0) {
        TFNAS_GLOAD(0);
        TFNAS_SSTORE(0, Bs0);
    }
    __syncthreads();
    for (int c = 0; c < nchunks; ++c) {
        const float* Bs = (c & 1) ? Bs1 : Bs0;
        const bool more = (c + 1 < nchunks);
        if (more) TFNAS_GLOAD(c + 1);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int k = ks * 4 + lk;
            const float a0 = cur[0][ks], a1 = cur[1][ks];
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const float b = Bs[k * T::LDB + 16 * j + lr];
                acc[0][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b, acc[0][j], 0, 0, 0);
                acc[1][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b, acc[1][j], 0, 0, 0);
            }
        }
        if (more) {
            if (c & 1) { TFNAS_SSTORE(c + 1, Bs0); } else { TFNAS_SSTORE(c + 1, Bs1); }
        }
        __syncthreads();
    }
#undef TFNAS_GLOAD
#undef TFNAS_SSTORE
#undef TFNAS_B_IDX
}

// One-piece loaders (the value is ready when the functor returns): for operands whose loads the compiler already keeps in
// flight across the MFMAs (plain unconditional loads, small weight tiles).
template <int NT, bool AK, bool BKC, class FA, class FB>
__device__ __forceinline__ void gemm_mainloop(FA& fa, FB& fb, int nchunks, f32x4 (&acc)[2][NT], float* lds) {
    auto la = [&](int c, int, int i0, int i1) -> f32x4 { return fa(c, i0, i1); };
    auto lb = [&](int c, int, int i0, int i1) -> f32x4 { return fb(c, i0, i1); };
    XfId id;
    gemm_mainloop2<NT, AK, BKC>(la, id, lb, id, nchunks, acc, lds);
}

// Per-column sums of the accumulator tile (rows outside the problem contribute exact zeros because the
// loaders zero-fill them).  Adds this tile's column sum / sum of squares into per-lane running totals.
template <int NT>
__device__ __forceinline__ void acc_colstats(const f32x4 (&acc)[2][NT], float (&s)[NT], float (&q)[NT]) {
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const f32x4 v = acc[i][j];
            s[j] += (v.x + v.y) + (v.z + v.w);
            q[j] += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
        }
}

// Block-reduce the per-lane column totals (over the 4 lane groups and 4 waves) and store them as this
// workgroup's partial (sum, sumsq) pair for columns [col0, col0+BN) that are < col_lim.  `part_row` is the
// workgroup's row of the partials matrix; a follow-up k_reduce_rows launch sums the rows in double precision.
// (Device-scope atomics cost ~1 ns each on MI355X and a stats epilogue issued 10^5..10^6 of them per launch;
//  partials + reduce is also bit-reproducible.)
template <int NT>
__device__ __forceinline__ void flush_colstats(float (&s)[NT], float (&q)[NT], float* lds, float* part_row,
                                               int col0, int col_lim) {
    using T = GT<NT>;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, lr = lane & 15;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        s[j] += __shfl_xor(s[j], 16, 64);
        s[j] += __shfl_xor(s[j], 32, 64);
        q[j] += __shfl_xor(q[j], 16, 64);
        q[j] += __shfl_xor(q[j], 32, 64);
    }
    if (lane < 16) {
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            lds[(w * 2 + 0) * T::BN + 16 * j + lr] = s[j];
            lds[(w * 2 + 1) * T::BN + 16 * j + lr] = q[j];
        }
    }
    __syncthreads();
    if (tid < T::BN && col0 + tid < col_lim) {
        float ss = 0.f, qq = 0.f;
#pragma unroll
        for (int ww = 0; ww < 4; ++ww) {
            ss += lds[(ww * 2 + 0) * T::BN + tid];
            qq += lds[(ww * 2 + 1) * T::BN + tid];
        }
        part_row[2 * (size_t)(col0 + tid) + 0] = ss;
        part_row[2 * (size_t)(col0 + tid) + 1] = qq;
    }
    __syncthreads();
}

// Emit the 128 x BN accumulator tile as 16-byte pieces of rows: emit(local_row, local_col (multiple of 4), f32x4).
// The MFMA C layout gives a lane 4 rows x 1 column per 16x16 tile, so a direct store writes 64-byte row segments;
// staging 16 rows per wave through LDS turns that into whole contiguous row pieces (256 B per row at BN = 64).
struct SlabNone {
    __device__ __forceinline__ void operator()(int, const float*) const {}
};
// `slab(i, st)`: optional hook called by every wave after the stores of its slab i, while the slab (16 rows x BN, leading
// dimension BN + 4) is still in LDS at `st`
template <int NT, class FEmit, class FSlab = SlabNone>
__device__ __forceinline__ void emit_tile_rows(const f32x4 (&acc)[2][NT], float* lds, FEmit emit, FSlab slab = FSlab()) {
    using T = GT<NT>;
    constexpr int LDC = T::BN + 4, Q = T::BN / 4;
    static_assert(4 * 16 * LDC <= T::LDS_FLOATS, "C staging does not fit the GEMM LDS buffer");
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, lr = lane & 15, lq = lane >> 4;
    float* st = lds + w * 16 * LDC;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        __syncthreads();
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) st[(4 * lq + r) * LDC + 16 * j + lr] = acc[i][j][r];
        __syncthreads();
#pragma unroll
        for (int it = 0; it < (16 * Q + 63) / 64; ++it) {
            const int idx = it * 64 + lane;
            if (idx < 16 * Q) {
                const int row = idx / Q, c4 = idx - row * Q;
                emit(w * 32 + 16 * i + row, 4 * c4, ld4(st + row * LDC + 4 * c4));
            }
        }
        slab(i, st);
    }
    __syncthreads();
}

// NT choice for an N extent: minimise padded columns, prefer the wider tile on ties.
static inline int pick_nt(int n, const int* cands, int ncand) {
    int best = cands[0];
    long best_pad = -1;
    for (int i = 0; i < ncand; ++i) {
        const int bn = 16 * cands[i];
        const long pad = (long)((n + bn - 1) / bn) * bn;
        if (best_pad < 0 || pad < best_pad || (pad == best_pad && cands[i] > best)) {
            best_pad = pad;
            best = cands[i];
        }
    }
    return best;
}
