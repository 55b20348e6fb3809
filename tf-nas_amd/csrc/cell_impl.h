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
};

// An expand weight gradient whose launch was put off to the NEXT cell's fork (path level, merged forks)
struct PendingExpand {
    bool valid = false;
    TfnasCellDesc d;
    const float *dEh, *E, *cb1, *x;
    float* part_w;
};

// where the weight-gradient kernels of a cell go: `side` == nullptr -> the caller's stream.
// merged (path level): ONE fork per cell instead of three.  Every fork is an event record on the data-gradient chain's stream
// followed by a stream wait on the side stream, and the record costs the chain 11-16 us of idle time (rocprofv3 trace of a
// weight step: the gaps in front of k_project_dgrad / the depthwise backward / k_expand_gram) -- 0.7 ms per chain and step
// with three per cell.  Merged: the project / SE / depthwise weight gradients of a cell are launched at the middle fork (after
// the BN2-backward sums), its expand weight gradient -- whose operands only exist at the end of the cell -- at the middle fork of
// the NEXT cell (`pend`), the last one by a final fork of the caller.
struct CellSide {
    hipStream_t side;
    hipEvent_t fork[3];
    bool merged = false;
    PendingExpand* pend = nullptr;
};

int launch_pending_expand(PendingExpand& p, hipStream_t s);

int cell_fwd_impl(const TfnasCellDesc& d, const TfnasCellWs& ws, const CellFwdBufs& b, hipStream_t s);
int cell_bwd_impl(const TfnasCellDesc& d, const TfnasCellWs& ws, const CellBwdBufs& b, hipStream_t s, const CellSide* so);
