// Launch sequences of ONE cell, shared by the per-cell C entry points (capi.hip) and the path level (path.hip).
#pragma once
#include <hip/hip_runtime.h>
#include "tfnas_hip.h"

struct CellFwdBufs {
    const float* x;
    const float* wmix;
    float *E, *D, *Pr, *fsmall;
    double* stats;
    float* part;
    float* out;
    const TfnasBnAffine* bn = nullptr;   // derived-network path: affine BatchNorm / running statistics (bn_affine.hip)
    const float* drop_scale = nullptr;   // per-image drop-connect factor (residual blocks, training)
};

struct CellBwdBufs {
    const float *x, *wmix, *E, *D, *Pr, *fsmall;
    const double* stats;
    const float* dout;
    float *dZ, *dEh, *bsmall;
    double* red;
    float *part, *part_w;        // scratch of the data-gradient chain / of the weight-gradient kernels
    float *dx, *dxp, *dwmix;
    const float* add_src;        // optional extra addend of dx: dx += add_scale[0] * add_src  (sink-connecting gradient,
    const float* add_scale;      //   same shape as dx; models/model_search.py:202-204 backward)
    const TfnasBnAffine* bn = nullptr;
    const float* drop_scale = nullptr;
    float* dout_s = nullptr;     // scratch for drop_scale[n] * dout
    // (appended: the launch sequences aggregate-initialise this struct positionally)
    float *part_w1 = nullptr, *part_w2 = nullptr;   // own scratch of forks 1 / 2 when they run on streams of their own (else part_w)
};

// where the weight-gradient kernels of a cell go: `side` == nullptr -> the caller's stream.  Three forks per cell (project /
// SE + depthwise / expand weight gradients): an event record on the data-gradient chain's stream + a stream wait on the side
// stream each.  (A one-fork-per-cell variant was measured in round 3 -- no gain -- and removed.)
struct CellSide {
    hipStream_t side[3];      // stream of fork k (project / SE + depthwise / expand weight gradients); nullptr: the caller's stream
    hipEvent_t fork[3];
};

int cell_fwd_impl(const TfnasCellDesc& d, const TfnasCellWs& ws, const CellFwdBufs& b, hipStream_t s);
int cell_bwd_impl(const TfnasCellDesc& d, const TfnasCellWs& ws, const CellBwdBufs& b, hipStream_t s, const CellSide* so);
