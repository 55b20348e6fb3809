// "E-free" operand producer: the expanded activation a1 = act(BN1(x W_expand^T)) of a channel chunk is recomputed from the
// (narrow) cell input inside the depthwise kernels instead of being written by k_expand_fwd and read back.
//
// For the early cells the expanded tensor E is the dominant HBM stream of the whole supernet (cell 0 at batch 128:
// 3.7 GB, written once and read twice per alpha-step) while x has 16..40 channels.  A 16-pixel x 32-channel block of E
// costs ic/4 v_mfma_f32_16x16x4_f32 per 16 channels -- the same exact-fp32 FMA chain as the expand GEMM, a few percent
// of the matrix pipe -- and x is re-read from L2 / Infinity Cache, so recomputing is far cheaper than the round trip.
//
// Contraction order: MFMA step kk contracts the four k-slots q = lane/16; slot q is given input channel q*KQ + kk
// (KQ = ic/4), so a lane's A operand is KQ *consecutive* floats of its pixel row (vector loads) and its B operand KQ
// consecutive floats of its weight row, held in registers for the whole kernel.  Forward and backward use this same
// function, so the ReLU / swish masks of the backward pass match the forward values bit for bit.
#pragma once
#include "tfnas_dev.h"

// B operand (weights of channels c0 + 16*nt + lane%16, nt = 0, 1) and the BN1 constants of those channels
template <int KQ>
struct ExpandB {
    float b[2][KQ];
    float2 bc[2];
};

// cst: LDS [32] (mean, rstd) of the chunk's channels (zeros beyond mc)
template <int KQ>
__device__ __forceinline__ void expand_b_load(ExpandB<KQ>& B, const float* __restrict__ w_expand, int ic, int c0, int mc,
                                              const float2* cst) {
    const int lane = threadIdx.x & 63, n = lane & 15, q = lane >> 4;
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        const int ch = c0 + 16 * nt + n;
#pragma unroll
        for (int kk = 0; kk < KQ; kk += 2) {
            float2 v = make_float2(0.f, 0.f);
            if (ch < mc) v = *reinterpret_cast<const float2*>(w_expand + (size_t)ch * ic + q * KQ + kk);
            B.b[nt][kk] = v.x;
            B.b[nt][kk + 1] = v.y;
        }
        B.bc[nt] = cst[16 * nt + n];
    }
}

// tile[pix * CS + ch], pix < npix, ch < 32:  OUT = 0/1: act_OUT(BN1(E)),  OUT = 2: BN1(E) (normalised, no activation);
// exact zeros for pixels outside the image (locate(pix, off) == false).  All 256 threads must call it (wave-uniform
// loop; __ballot carries the validity of the 16 pixels of an M-tile to the lanes that hold their results).
// (expand_tile_to: the pixel's 32 floats go to dst(pix) instead of tile + pix * CS -- the row-streaming kernels' LDS ring)
template <int KQ, int OUT, class FLoc, class FDst>
__device__ __forceinline__ void expand_tile_to(int npix, const float* __restrict__ x, const ExpandB<KQ>& B, FLoc locate,
                                               FDst dst);
template <int KQ, int OUT, class FLoc>
__device__ __forceinline__ void expand_tile(float* tile, int npix, int CS, const float* __restrict__ x,
                                            const ExpandB<KQ>& B, FLoc locate) {
    expand_tile_to<KQ, OUT>(npix, x, B, locate, [&](int pix) { return tile + pix * CS; });
}
template <int KQ, int OUT, class FLoc, class FDst>
__device__ __forceinline__ void expand_tile_to(int npix, const float* __restrict__ x, const ExpandB<KQ>& B, FLoc locate,
                                               FDst dst) {
    // NB M-tiles per round: all their x loads are issued before the first MFMA (the loads come from L2 / Infinity
    // Cache, ~1-2 us each; one tile at a time would serialise that latency 5-7 times per spatial tile)
    constexpr int NB = KQ <= 4 ? 4 : (KQ <= 6 ? 3 : 2);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n = lane & 15, q = lane >> 4;
    const int nmt = (npix + 15) >> 4;
    for (int mt0 = wave; mt0 < nmt; mt0 += 4 * NB) {
        float a[NB][KQ];
        unsigned valid[NB];
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const int mt = mt0 + 4 * b;
            const int p = 16 * mt + n;
            size_t off = 0;
            const bool ok = mt < nmt && p < npix && locate(p, off);
#pragma unroll
            for (int kk = 0; kk < KQ; kk += 2) {
                float2 v = make_float2(0.f, 0.f);
                if (ok) v = *reinterpret_cast<const float2*>(x + off + q * KQ + kk);
                a[b][kk] = v.x;
                a[b][kk + 1] = v.y;
            }
            valid[b] = (unsigned)(__ballot(ok) & 0xffffull);
        }
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const int mt = mt0 + 4 * b;
            if (mt >= nmt) break;                                   // wave-uniform
            f32x4 acc0 = zero4(), acc1 = zero4();
#pragma unroll
            for (int kk = 0; kk < KQ; ++kk) {
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[b][kk], B.b[0][kk], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[b][kk], B.b[1][kk], acc1, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int pl = 4 * q + r, pix = 16 * mt + pl;
                if (pix < npix) {
                    float v0 = 0.f, v1 = 0.f;
                    if ((valid[b] >> pl) & 1u) {
                        v0 = (acc0[r] - B.bc[0].x) * B.bc[0].y;
                        v1 = (acc1[r] - B.bc[1].x) * B.bc[1].y;
                        if (OUT != 2) {
                            v0 = act_f<(OUT == 2 ? 0 : OUT)>(v0);
                            v1 = act_f<(OUT == 2 ? 0 : OUT)>(v1);
                        }
                    }
                    float* o = dst(pix);
                    o[n] = v0;
                    o[16 + n] = v1;
                }
            }
        }
    }
}

// Workgroup -> (tile lane, channel chunk) for the E-free depthwise kernels, grid = (gx lanes, chunks).  All channel chunks
// of a spatial tile re-read the same x pixels, so they should run at the same time on the SAME XCD (private L2): the
// hardware deals workgroups round-robin to the 8 XCDs in x-fastest order, so XCD k is given the contiguous range
// [k*total/8, (k+1)*total/8) of a chunk-fastest enumeration.  (With the plain (blockIdx.x, blockIdx.y) assignment every
// chunk slab streamed all of x from HBM again: 18 x 103 MB for cell 0.)
__device__ __forceinline__ void efree_lane_chunk(int& lane, int& cy) {
    const int gx = gridDim.x, chunks = gridDim.y, total = gx * chunks;
    const int L = blockIdx.x + gx * blockIdx.y;
    const int xcd = L & 7, rem = total & 7;
    const int V = xcd * (total >> 3) + (xcd < rem ? xcd : rem) + (L >> 3);
    lane = V / chunks;
    cy = V - lane * chunks;
}

static inline bool efree_ic_ok(int ic) { return ic == 16 || ic == 24 || ic == 40; }
