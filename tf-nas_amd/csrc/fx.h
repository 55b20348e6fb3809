// "fx": the fused per-image route of the late cells (14x14 / 7x7 images, 80..192 input channels) -- host-side plan shared by
// the launchers (fx_kernels.hip) and the launch sequences (capi.hip).
//
// Reference arithmetic replaced: inverted_bottleneck (1x1 conv -> BN -> act) + depth_conv (k x k depthwise conv) of
// MBInvertedResBlock.forward, models/layers.py:542-552, and their autograd backward.  The reference materialises the expanded
// tensor between the two; here a workgroup owns a group of whole images x a slice of mid channels, keeps the images' x rows
// as split-bf16 MFMA operands in registers for its whole life, rebuilds act(BN1(x W1^T)) 32 channels at a time on the matrix
// cores straight into the LDS layout the depthwise stencil reads (no halo recompute, no ring: the whole image is resident),
// and in the backward feeds the stencil's result directly into the expand-dgrad MFMA -- E and dE never exist in memory.
#pragma once
#include <stdint.h>
#include "tfnas_hip.h"

struct FxSlice {
    int16_t g;        // group (candidate) of the slice: one depthwise kernel size per workgroup
    int16_t nch;      // 32-channel chunks in the slice
    int16_t c0;       // first channel of the slice inside its group (multiple of 32)
    int16_t chunk0;   // index of its first chunk in the cell's chunk enumeration (= blob index)
};
constexpr int FX_MAX_SLICES = 96;
constexpr int FX_THREADS = 512;
// padding of a blob's plane rows (bytes): the row pitch 64 KS + FX_RS_PAD decides the bank pattern of the A-operand ds_read_b128.
// 16: the pitch in dwords is 4 (4 KS + 1), an odd multiple of 4 -- the 16 rows a group of 16 lanes reads start in 16 different bank
// quads; with 32, the first choice, rows n and n + 8 share their banks.  Measured (two alternating A/B runs on one box): -2 % on the
// KS = 4 cells (cell 10 / 12: forward 0.372 -> 0.367 / 0.343 -> 0.335 ms, backward 0.511 -> 0.499 / 0.461 -> 0.451), +-0 at KS = 3 and
// KS = 6 -- while the SQ counters of the forward show MORE conflict cycles (0.28 -> 0.37 of the LDS cycles): the clock decides.
#ifndef FX_RS_PAD
#define FX_RS_PAD 16
#endif

struct FxPlan {
    int KS;            // MFMA k-steps of 32 input channels: ceil(ic / 32)
    int RT;            // 16-pixel tiles per wave: backward (8 waves), 1 or 2
    int RTF;           // ... forward (7 worker waves + 1 copier wave)
    int NI;            // images per workgroup
    int nig;           // image groups = ceil(N / NI)
    int RS;            // bytes per channel row of a bf16 plane in a blob (32 * KS * 2 + FX_RS_PAD: conflict-free ds_read_b128)
    int PB, WB, XB;    // blob pieces: W1 planes + BN1 constants | depthwise taps | backward extras; bytes
    int BLOB;          // blob stride (bytes, multiple of 256)
    int KMAX;          // largest depthwise kernel size in the cell (sizes the LDS image tile)
    int nchunks, nslices;
    FxSlice sl[FX_MAX_SLICES];
};

// true: the cell can take the fused route (geometry, widths, scratch); fills `pl`.  bwd: plan for the backward kernel
// (fewer, longer slices: every slice writes a partial dx).
bool fx_plan(const TfnasCellDesc& d, FxPlan& pl, bool bwd);
