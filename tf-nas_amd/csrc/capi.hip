// extern "C" surface of libtfnas_hip.so (declared in include/tfnas_hip.h): descriptor planning, workspace
// sizing and the per-cell forward / backward launch sequences.
#include <string.h>
#include "tfnas_dev.h"
#include "kernels.h"
#include "cell_impl.h"

#include <stdlib.h>
#include <mutex>
#include <vector>

static inline hipStream_t S(void* s) { return reinterpret_cast<hipStream_t>(s); }

// ---------------------------------------------------------------------------------------------------------------
// Weight-gradient side stream.  The weight-gradient kernels of a cell are leaves of the backward dependency chain
// (nothing downstream in the same call consumes them), and like most kernels of the sampled w-step they are
// latency- rather than throughput-bound, so they are enqueued on a second HIP stream owned by the library
// (one per caller stream and device) and overlap with the data-gradient chain of the same cell: fork events
// after the producers of their operands, one join before tfnas_mixedop_bwd returns control to the caller's stream
// (so from the caller's point of view everything is still ordered on `stream`).  They use the second half of `part`.
// TFNAS_ROUTE_WGRAD_INLINE in the descriptor disables the side stream (everything on the caller's stream, identical results).
// TfnasCellDesc.wgrad_stream[k] != NULL: fork k goes to that caller-owned stream instead.
struct SideCtx {
    int device;
    hipStream_t main, side;
    hipEvent_t fork[4], join;
};
static std::mutex g_side_mu;
static std::vector<SideCtx*> g_side;

static SideCtx* side_for(hipStream_t s) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    std::lock_guard<std::mutex> lk(g_side_mu);
    for (SideCtx* c : g_side)
        if (c->main == s && c->device == dev) return c;
    SideCtx* c = new SideCtx();
    c->device = dev;
    c->main = s;
    if (hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking) != hipSuccess) { delete c; return nullptr; }
    bool ok = hipEventCreateWithFlags(&c->join, hipEventDisableTiming) == hipSuccess;
    for (int i = 0; i < 4; ++i) ok = ok && hipEventCreateWithFlags(&c->fork[i], hipEventDisableTiming) == hipSuccess;
    if (!ok) { delete c; return nullptr; }
    g_side.push_back(c);
    return c;
}
// make `side` wait for everything enqueued on `main` so far; returns the stream to launch on
static hipStream_t side_fork(SideCtx* c, int k, hipStream_t main) {
    if (!c) return main;
    if (hipEventRecord(c->fork[k], main) != hipSuccess || hipStreamWaitEvent(c->side, c->fork[k], 0) != hipSuccess)
        return main;
    return c->side;
}
static int side_join(SideCtx* c, hipStream_t main) {
    if (!c) return 0;
    hipError_t e = hipEventRecord(c->join, c->side);
    if (e == hipSuccess) e = hipStreamWaitEvent(main, c->join, 0);
    return (int)e;
}

// Joins the side stream on every exit path of tfnas_mixedop_bwd: an early error return must not leave weight-gradient
// kernels running on buffers the caller is about to free.
struct SideJoinGuard {
    SideCtx* c = nullptr;
    hipStream_t main = nullptr;
    hipStream_t extra[3] = {nullptr, nullptr, nullptr};     // caller-owned weight-gradient streams of this launch (wgrad_stream[])
    bool joined = false;
    int join_extra() {
        int rc = 0;
        if (!c) return 0;
        for (int k = 0; k < 3; ++k) {
            hipStream_t x = extra[k];
            if (!x || x == main) continue;
            bool dup = false;
            for (int q = 0; q < k; ++q) dup = dup || extra[q] == x;
            if (dup) continue;
            // (the context's fork events are free again here: every fork's wait was enqueued long ago)
            hipError_t e = hipEventRecord(c->fork[3], x);
            if (e == hipSuccess) e = hipStreamWaitEvent(main, c->fork[3], 0);
            if (e != hipSuccess) {
                (void)hipStreamSynchronize(x);
                if (!rc) rc = (int)e;
            }
        }
        return rc;
    }
    int join() {
        joined = true;
        const int r0 = side_join(c, main), r1 = join_extra();
        return r0 ? r0 : r1;
    }
    ~SideJoinGuard() {
        if (!joined && c) {
            if (side_join(c, main) != 0) (void)hipStreamSynchronize(c->side);
            (void)join_extra();
        }
    }
};

// Lazy join (derived-network training step): tfnas_mbconv_bwd leaves its weight-gradient kernels running on the side stream
// when this is on; the caller keeps every buffer they read alive (torch: record_stream on the side stream's handle) and joins
// once before it consumes the gradients (tfnas_side_join).
static int g_lazy_join = 0;
extern "C" int tfnas_set_lazy_join(int on) {
    g_lazy_join = on ? 1 : 0;
    return 0;
}
extern "C" int tfnas_side_stream(void* stream, void** side) {
    if (!side) return TFNAS_ENULL;
    *side = nullptr;
    SideCtx* c = side_for(S(stream));
    if (c) *side = (void*)c->side;
    return 0;
}
extern "C" int tfnas_side_join(void* stream) {
    return side_join(side_for(S(stream)), S(stream));
}

extern "C" int tfnas_shutdown(void) {
    std::lock_guard<std::mutex> lk(g_side_mu);
    for (SideCtx* c : g_side) {
        (void)hipStreamSynchronize(c->side);
        for (int i = 0; i < 4; ++i) (void)hipEventDestroy(c->fork[i]);
        (void)hipEventDestroy(c->join);
        (void)hipStreamDestroy(c->side);
        delete c;
    }
    g_side.clear();
    return 0;
}

StatsSync g_stats_sync = {nullptr, nullptr, 1};

extern "C" int tfnas_set_stats_sync(tfnas_stats_sync_fn fn, void* user, int world) {
    if (fn && world < 1) return TFNAS_ERANGE;
    g_stats_sync.fn = fn;
    g_stats_sync.user = user;
    g_stats_sync.world = fn ? world : 1;
    return 0;
}

extern "C" int tfnas_abi_version(void) { return TFNAS_ABI_VERSION; }
extern "C" int tfnas_set_gemm_mode(int mode) { return set_gemm_mode(mode); }
extern "C" int tfnas_gemm_mode(void) { return gemm_mode(); }

extern "C" uint64_t tfnas_sizeof(int which) {
    switch (which) {
        case 0: return sizeof(TfnasGroup);
        case 1: return sizeof(TfnasCellDesc);
        case 2: return sizeof(TfnasCellWs);
        case 3: return sizeof(TfnasStage);
        case 4: return sizeof(TfnasPathDesc);
        case 5: return sizeof(TfnasPathWs);
        case 6: return sizeof(TfnasBnAffine);
        case 7: return TFNAS_PART_ALLOC;      // floats of one `part` scratch region ...
        case 8: return TFNAS_TAIL_SLOTS;      // ... whose last this-many 4-byte words are reserved
        default: return 0;
    }
}

#define TRY(call)               \
    do {                        \
        int _r = (call);        \
        if (_r != 0) return _r; \
    } while (0)

// the per-launch modes of a descriptor (callers may change them between tfnas_cell_plan and a launch: every entry point re-checks)
static int check_modes(const TfnasCellDesc* d) {
    if (d->gemm_mode != 0) {
        const int gm = d->gemm_mode & ~(TFNAS_GEMM_EXPLICIT | TFNAS_GEMM_EVERYWHERE);
        if (!(d->gemm_mode & TFNAS_GEMM_EXPLICIT) || (gm != 0 && gm != 1 && gm != 3 && gm != 6)) return TFNAS_EINVAL;
    }
    if (d->flags & ~TFNAS_CELL_LAZY_JOIN) return TFNAS_EINVAL;
    if (d->route & ~TFNAS_ROUTE_ALL) return TFNAS_EINVAL;
    if ((d->route & TFNAS_ROUTE_XG_OFF) && (d->route & TFNAS_ROUTE_XG_ALL)) return TFNAS_EINVAL;
    if ((d->route & TFNAS_ROUTE_SE_MASK) == TFNAS_ROUTE_SE_MASK) return TFNAS_EINVAL;     // (3 is not an excite-FC variant)
    if (d->fwd_route & ~(TFNAS_ROUTE_TAKEN_VALID | TFNAS_ROUTE_TAKEN_FX)) return TFNAS_EINVAL;
    if (d->sync_fn && d->sync_world < 1) return TFNAS_ERANGE;
    return 0;
}

extern "C" int tfnas_cell_plan(TfnasCellDesc* d) {
    if (!d) return TFNAS_ENULL;
    if (d->G < 1 || d->G > TFNAS_MAX_GROUPS) return TFNAS_ERANGE;
    if (d->mode < TFNAS_MODE_CELL || d->mode > TFNAS_MODE_HEAD) return TFNAS_EINVAL;
    if (d->mode == TFNAS_MODE_STEM) {
        /* x is the NCHW image; the 3x3 stride-2 pad-1 stem conv defines the cell's input extent */
        if (d->ic != 27 || d->G != 1 || d->Hi < 1 || d->Wi < 1 || d->has_res) return TFNAS_EINVAL;
        d->H = (d->Hi - 1) / 2 + 1;
        d->W = (d->Wi - 1) / 2 + 1;
    } else if (d->ic < 4 || (d->ic & 3)) {
        return TFNAS_EINVAL;
    }
    if (d->mode == TFNAS_MODE_HEAD && d->G != 1) return TFNAS_EINVAL;
    if (d->stor != 0) return TFNAS_EINVAL;     // fp32 storage only (the bf16-storage build of rounds 1-3 was removed: tfnas_hip.h)
    if (d->N < 1 || d->H < 1 || d->W < 1 || d->oc < 4 || (d->oc & 3)) return TFNAS_EINVAL;
    if (d->oc > 1024) return TFNAS_EINVAL;
    if (d->stride != 1 && d->stride != 2) return TFNAS_EINVAL;
    if (d->act != TFNAS_ACT_RELU && d->act != TFNAS_ACT_SWISH) return TFNAS_EINVAL;
    if (d->has_res && (d->ic != d->oc || d->stride != 1)) return TFNAS_EINVAL;
    TRY(check_modes(d));
    // conv output size with pad = k/2 (same for k = 3 and 5)
    d->Ho = (d->H - 1) / d->stride + 1;
    d->Wo = (d->W - 1) / d->stride + 1;
    int off = 0, se_off = 0;
    for (int g = 0; g < d->G; ++g) {
        TfnasGroup& gr = d->g[g];
        if (gr.mc < 1 || (gr.k != 3 && gr.k != 5) || gr.se < 0) return TFNAS_EINVAL;
        if (gr.se & 3) return TFNAS_EINVAL;       // SE widths are in_channels x {1, 2} in the search space; the excite
                                                  // GEMMs move hidden units in quads
        gr.mcp = (gr.mc + 3) & ~3;
        // every group starts on a 128-byte line of the [pixels][M] rows (and M is a multiple of 32 floats, so every row
        // does too): the 32-channel segments the depthwise kernels move and the 64-column GEMM tiles are then whole
        // lines.  Packed groups (72 / 144 / 120 / 336 floats wide) left most of them 32..96 bytes off, i.e. partial-line
        // writes and two lines fetched per segment.  The pad columns are never read or written.
        off = (off + 31) & ~31;
        gr.off = off;
        gr.se_off = se_off;
        off += gr.mcp;
        se_off += gr.se;
    }
    off = (off + 31) & ~31;
    if ((off & 511) == 0) off += 32;                  // no power-of-two-ish row stride (HBM channel aliasing)
    d->M = off;
    d->SE = se_off;
    if ((double)d->N * d->H * d->W >= 2147483647.0) return TFNAS_ERANGE;   // row indices are int, offsets size_t
    return 0;
}

extern "C" int tfnas_cell_ws(const TfnasCellDesc* d, TfnasCellWs* ws) {
    if (!d || !ws) return TFNAS_ENULL;
    memset(ws, 0, sizeof(*ws));
    const uint64_t P = (uint64_t)d->N * d->H * d->W, Po = (uint64_t)d->N * d->Ho * d->Wo;
    const uint64_t M = d->M, N = d->N, SE = d->SE, G = d->G, oc = d->oc;
    // the four stream tensors: P*M / Po*M fp32 elements
    ws->E = P * M;
    ws->D = Po * M;
    ws->Pr = G * Po * oc;
    ws->off_pooled = 0;
    ws->off_gate = N * M;
    ws->off_hpre = 2 * N * M;
    ws->fsmall = 2 * N * M + N * SE + 4;
    ws->off_stats1 = 0;
    ws->off_stats2 = 2 * M;
    ws->off_stats3 = 4 * M;
    ws->stats = 4 * M + 2 * G * oc;
    ws->out = Po * oc;
    ws->dZ = Po * M;
    ws->dEh = P * M;
    ws->off_dgate = 0;
    ws->off_dpooled = N * M;
    ws->off_dgl = 2 * N * M;
    ws->off_dhpre = 3 * N * M;
    ws->off_cb1 = 3 * N * M + ((N * SE + 3) & ~(uint64_t)3);
    ws->bsmall = ws->off_cb1 + 4 * M;
    ws->off_red3 = 0;
    ws->off_resdot = 2 * G * oc;                 /* [oc] per-channel <dout,x>, contiguous with red3 */
    ws->off_red2 = 2 * G * oc + oc;
    ws->off_red1 = ws->off_red2 + 2 * M;
    ws->red = ws->off_red1 + 2 * M;
    ws->part = (d->need_wgrad ? 2 : 1) * (uint64_t)TFNAS_PART_ALLOC;   // second half: weight-gradient side stream
    ws->dx = P * d->ic;
    {
        const int ns = d->mode == TFNAS_MODE_STEM ? 1 : expand_dgrad_splits(*d);
        ws->dxp = ns > 1 ? (uint64_t)ns * P * d->ic : 4;   /* split-K partials of the expand dgrad */
    }
    return 0;
}

extern "C" int tfnas_efree_supported(const TfnasCellDesc* dp) { return (dp && efree_supported(*dp)) ? 1 : 0; }
extern "C" int tfnas_fx_supported(const TfnasCellDesc* dp) {
    return (dp && dp->mode == TFNAS_MODE_CELL && !dp->need_wgrad && !efree_ic_small(dp->ic) && fx_supported(*dp)) ? 1 : 0;
}
// what a forward of this descriptor leaves in the saved buffers (tfnas_hip.h): the one decision of cell_fwd_impl that changes
// their MEANING is the fused per-image route (ehat instead of E); affine / eval BatchNorm launches (tfnas_mbconv_*) never take it
static int route_taken(const TfnasCellDesc& d, bool affine) {
    return TFNAS_ROUTE_TAKEN_VALID | ((!affine && fx_supported(d)) ? TFNAS_ROUTE_TAKEN_FX : 0);
}
extern "C" int tfnas_cell_route(const TfnasCellDesc* dp) { return dp ? route_taken(*dp, false) : 0; }

// ---------------------------------------------------------------------------------------------------------------
// One cell, forward / backward: the launch sequences shared by the per-cell entry points below and by the path level
// (path.hip).
// BatchNorm channel counts / element counts of the three sites of a G = 1 block
static void bn_site(const TfnasCellDesc& d, int site, int& nch, uint64_t& cnt) {
    const uint64_t P = (uint64_t)d.N * d.H * d.W, Po = (uint64_t)d.N * d.Ho * d.Wo;
    nch = site == 2 ? d.oc : d.g[0].mc;
    cnt = site == 0 ? P : Po;
}
static int bn_fwd_fix(const TfnasCellDesc& d, const TfnasBnAffine* bn, int site, double* stats, hipStream_t s) {
    int nch;
    uint64_t cnt;
    bn_site(d, site, nch, cnt);
    return launch_bn_fwd_fix(stats, nch, cnt, d.eps, bn->weight[site], bn->bias[site], bn->running_mean[site],
                             bn->running_var[site], bn->momentum, bn->eval, s, stats_world(d));   // (sync-stats: global batch)
}

int cell_fwd_impl(const TfnasCellDesc& d0, const TfnasCellWs& ws, const CellFwdBufs& b, hipStream_t s) {
    double* stats1 = b.stats + ws.off_stats1;
    double* stats2 = b.stats + ws.off_stats2;
    double* stats3 = b.stats + ws.off_stats3;
    float* pooled = b.fsmall + ws.off_pooled;
    float* gate = b.fsmall + ws.off_gate;
    float* hpre = b.fsmall + ws.off_hpre;
    // affine / eval BatchNorm: the producers run as usual; their statistics tables are rewritten as effective (mean, rstd)
    // before the consumers -- which then run with eps = -1 (tfnas_dev.h: bn_consts) -- read them
    const TfnasBnAffine* bn = b.bn;
    TfnasCellDesc dc = d0;
    if (bn) dc.eps = -1.f;
    const TfnasCellDesc& d = dc;
    // E-free late cells (14 x 14 / 7 x 7 images, 64..192 input channels): the fused per-image route (fx_kernels.hip)
    // (frozen weights only: fx_supported refuses need_wgrad; with E the forward leaves ehat there for the backward)
    const bool fx = !bn && fx_supported(d);
    if (fx) TRY(launch_fx_stats(d, b.x, stats1, b.part, s));                  // BN1 statistics from the Gram matrix of x
    else if (b.E) TRY(launch_expand_fwd(d, b.x, b.E, stats1, b.part, s));     // 1x1 expand (all groups) + BN1 statistics
    else TRY(launch_expand_stats_gram(d, b.x, stats1, b.part, s));            // E-free: BN1 statistics from the Gram matrix of x
    const bool sync = !(bn && bn->eval);          // (eval mode normalises with the running statistics: nothing to reduce)
    if (sync) TRY(stats_sync(d, stats1, 2 * (size_t)d.M, s));                    // sync-stats: global-batch sums (no-op without a hook)
    if (bn) TRY(bn_fwd_fix(d0, bn, 0, stats1, s));
    if (fx) TRY(launch_fx_fwd(d, b.x, stats1, b.E, b.D, stats2, b.part, s));  // expand + BN1 + act + depthwise in one kernel
    else TRY(launch_dw_fwd(d, b.E, b.x, stats1, b.D, stats2, b.part, s));     // BN1+act fused load, depthwise, BN2 statistics
    if (sync) TRY(stats_sync(d, stats2, 2 * (size_t)d.M, s));
    if (bn) TRY(bn_fwd_fix(d0, bn, 1, stats2, s));
    TRY(launch_se_pool(d, b.D, stats2, pooled, s));                           // SE squeeze (SE groups only)
    TRY(launch_se_fc_fwd(d, pooled, hpre, gate, b.part, TFNAS_PART_FLOATS, s));    // SE excite (K-split partials in `part`)
    TRY(launch_project_fwd(d, b.D, gate, stats2, b.Pr, stats3, b.part, s));   // BN2+act+gate fused load, 1x1 project, BN3 stats
    if (sync) TRY(stats_sync(d, stats3, 2 * (size_t)d.G * d.oc, s));
    if (bn) TRY(bn_fwd_fix(d0, bn, 2, stats3, s));
    if (b.drop_scale && d.has_res) {
        // drop-connect (tools/utils.py:77-86): out = scale[n] * BN3(.) + x -- the mix kernel without its residual, then one pass
        TfnasCellDesc dn = dc;
        dn.has_res = 0;
        TRY(launch_mix_fwd(dn, b.Pr, stats3, b.wmix, b.x, b.out, s));
        return launch_rowscale(b.out, b.out, b.x, b.drop_scale, d.N, (uint64_t)d.Ho * d.Wo * d.oc, s);
    }
    TRY(launch_mix_fwd(d, b.Pr, stats3, b.wmix, b.x, b.out, s));              // sum_g w_g BN3(.) + residual
    return 0;
}

// make `side` wait for everything enqueued on `main` so far; returns the stream to launch on
static hipStream_t fork_to(const CellSide* so, int k, hipStream_t main) {
    if (!so || !so->side[k] || so->side[k] == main) return main;
    if (hipEventRecord(so->fork[k], main) != hipSuccess || hipStreamWaitEvent(so->side[k], so->fork[k], 0) != hipSuccess)
        return main;
    return so->side[k];
}

static int bn_bwd_fix(const TfnasCellDesc& d, const TfnasBnAffine* bn, int site, double* red, hipStream_t s) {
    int nch;
    uint64_t cnt;
    bn_site(d, site, nch, cnt);
    TRY(launch_bn_bwd_fix(red, nch, cnt, bn->weight[site], bn->bias[site], bn->g_weight[site], bn->g_bias[site], s));
    // eval mode: the statistics are constants, dx = r_eff * d -- the generic form r*(d - t1 - y*t2) with t1 = t2 = 0
    if (bn->eval) return (int)hipMemsetAsync(red, 0, sizeof(double) * 2 * (size_t)nch, s);
    return 0;
}

// TFNAS_ROUTE_FOLD_OFF: BN2-backward tables in their own pass (k_bn2_pool) instead of the epilogue of k_project_dgrad; both
// are compared with the oracle (tests/test_gpu_cell.py::test_variant_against_oracle)

int cell_bwd_impl(const TfnasCellDesc& d0, const TfnasCellWs& ws, const CellBwdBufs& b0, hipStream_t s, const CellSide* so) {
    const TfnasBnAffine* bn = b0.bn;
    TfnasCellDesc dc = d0;
    if (bn) dc.eps = -1.f;
    const TfnasCellDesc& d = dc;
    CellBwdBufs b = b0;
    // the forward's route, when the caller recorded it (tfnas_cell_route -> fwd_route): a backward that would read the E buffer
    // differently from how the forward wrote it (need_wgrad or the sync hook changed in between) refuses
    if ((d0.fwd_route & TFNAS_ROUTE_TAKEN_VALID) && route_taken(d0, bn != nullptr) != d0.fwd_route) return TFNAS_EINVAL;
    const float* dout_res = b0.dout;                       // the residual branch sees the unscaled gradient
    if (b0.drop_scale && d.has_res) {
        if (!b0.dout_s) return TFNAS_ENULL;
        TRY(launch_rowscale(b0.dout_s, b0.dout, nullptr, b0.drop_scale, d.N, (uint64_t)d.Ho * d.Wo * d.oc, s));
        b.dout = b0.dout_s;
    }
    const double* stats1 = b.stats + ws.off_stats1;
    const double* stats2 = b.stats + ws.off_stats2;
    const double* stats3 = b.stats + ws.off_stats3;
    const float* pooled = b.fsmall + ws.off_pooled;
    const float* gate = b.fsmall + ws.off_gate;
    const float* hpre = b.fsmall + ws.off_hpre;
    float* dgate = b.bsmall + ws.off_dgate;
    float* dpooled = b.bsmall + ws.off_dpooled;
    float* dgl = b.bsmall + ws.off_dgl;
    float* dhpre = b.bsmall + ws.off_dhpre;
    float* cb1 = b.bsmall + ws.off_cb1;
    double* red3 = b.red + ws.off_red3;
    double* red2 = b.red + ws.off_red2;
    double* red1 = b.red + ws.off_red1;
    float* part = b.part;
    float* part_w = b.part_w;
    float* part_w1 = b.part_w1 ? b.part_w1 : b.part_w;      // (forks on streams of their own must not share split-K scratch)
    float* part_w2 = b.part_w2 ? b.part_w2 : b.part_w;

    if (d.need_wgrad) {
        for (int g = 0; g < d.G; ++g) {
            const TfnasGroup& gr = d.g[g];
            if (!gr.g_expand || !gr.g_dw || !gr.g_proj) return TFNAS_ENULL;
            if (gr.se > 0 && (!gr.g_se_r || !gr.gb_se_r || !gr.g_se_e || !gr.gb_se_e)) return TFNAS_ENULL;
        }
    }
    TRY(launch_mix_bwd_stats(d, b.dout, b.Pr, stats3, b.x, red3, part, s));           // BN3 backward sums (+ d wmix)
    TRY(stats_sync(d, red3, 2 * (size_t)d.G * d.oc, s));
    if (b.dwmix) TRY(launch_mix_dw(d, red3, b.red + ws.off_resdot, b.dwmix, s));
    if (bn) TRY(bn_bwd_fix(d0, bn, 2, red3, s));
    // Nothing upstream wants a gradient (first cell of the alpha-step: frozen weights, input = stem output):
    // d wmix is the only product, like autograd pruning the same sub-graph in the reference.
    if (!b.dx && !d.need_wgrad) return 0;
    // weight gradients: on the side stream, scratch = part_w
    if (d.need_wgrad)
        TRY(launch_project_wgrad(d, b.dout, b.Pr, b.D, gate, stats2, stats3, red3, b.wmix, part_w, fork_to(so, 0, s)));
    const bool fused2 = bn2_fused_fits(d);
    // dZ = dP W_proj; where the geometry allows, the per-image BN2-backward tables are accumulated in its epilogue (records in
    // dEh, which nothing reads or writes before the SE backward below) and gathered -- no second pass over dZ
    // (policy, measured per cell at B = 128: the epilogue's column-wise D loads cost more than k_bn2_pool's streaming pass on the
    //  write-bound all-candidate launches of the 112 x 112 / 56 x 56 cells -- cell 1: 0.99 -> 1.04 ms -- and less everywhere else:
    //  cell 10 sampled 0.126 -> 0.103 ms, cell 15 all candidates 0.286 -> 0.252 ms)
    const bool fold = fused2 && !(d.route & TFNAS_ROUTE_FOLD_OFF) && project_fold_ok(d, (size_t)ws.dEh) && (d.G == 1 || d.Ho * d.Wo <= 784);
    if (fold) {
        TRY(launch_project_dgrad(d, b.dout, b.Pr, stats3, red3, b.wmix, b.dZ, s, b.D, stats2, b.dEh));
        TRY(launch_bn2_gather(d, b.dEh, dgate, part, s));
    } else {
        TRY(launch_project_dgrad(d, b.dout, b.Pr, stats3, red3, b.wmix, b.dZ, s));
        if (fused2) TRY(launch_bn2_pool(d, b.dZ, b.D, stats2, dgate, part, s));   // d gate + per-image BN2-backward tables
        else TRY(launch_se_bwd_reduce(d, b.dZ, b.D, stats2, dgate, s));         // SE groups: d gate
    }
    // (K-split partials of the SE backward go through dEh, which is only written by the depthwise dgrad further down)
    TRY(launch_se_fc_bwd(d, dgate, gate, hpre, dgl, dhpre, dpooled, b.dEh, (size_t)ws.dEh, s));
    if (fused2) TRY(launch_bn2_finish(d, part, gate, dpooled, red2, s));      // BN2 backward sums
    else TRY(launch_bn2_bwd(d, b.dZ, b.D, stats2, gate, dpooled, red2, part, s));
    TRY(stats_sync(d, red2, 2 * (size_t)d.M, s));
    if (bn) TRY(bn_bwd_fix(d0, bn, 1, red2, s));
    if (!bn && fx_supported(d)) {
        // fused per-image route: depthwise dgrad + act' + the dE (rstd . W1) term of the expand dgrad in one kernel (dE never
        // materialised; partial sums per channel slice in the dEh buffer), then the BN1-backward correction -x G + b
        int nsl = 0;
        TRY(launch_fx_bwd(d, b.x, b.E, stats1, stats2, red2, b.dZ, b.D, gate, dpooled, b.dEh, (size_t)ws.dEh, red1, cb1, part, &nsl, s));
        if (b.dx) {
            float* gram = part + TFNAS_PART_FLOATS - expand_gram_floats(d);
            TRY(launch_expand_gram(d, cb1, part, TFNAS_PART_FLOATS - expand_gram_floats(d), gram, s));
            TRY(launch_expand_dgrad_x(d, b.x, cb1, gram, dout_res, b.wmix, b.dx, b.dEh, nsl, s, b.add_src, b.add_scale));
        }
#ifdef TFNAS_FXW_TIMING
        if (d.need_wgrad) {              // timing only (fx_kernels.hip: fx_plan): the materialised route's weight-gradient launches
            hipStream_t sw = fork_to(so, 1, s);
            if (d.SE > 0) TRY(launch_se_wgrad(d, dgate, gate, dhpre, hpre, pooled, sw));
            TRY(launch_dw_wgrad(d, b.dZ, gate, dpooled, b.D, stats2, red2, b.E, stats1, part_w1, sw));
            TRY(launch_expand_wgrad(d, b.dEh, b.E, cb1, b.x, part_w2, fork_to(so, 2, s)));
        }
#endif
        return 0;
    }
    // stride-1 ring cells: the depthwise weight gradient comes out of the backward-data pass below (same dd window, same E
    // elements: no second read of E, dZ and D -- dw_stream.inc, WGR)
    const bool dw_fused = d.need_wgrad && dw_bwd_fuses_wgrad(d, b.E);
    if (d.need_wgrad) {
        // ONE fork for the SE and the depthwise weight gradients (every fork is an event record + a stream wait on the
        // host-bound w-step; cells without SE launch nothing for it)
        if (d.SE > 0 || !dw_fused) {
            hipStream_t sw = fork_to(so, 1, s);
            if (d.SE > 0) TRY(launch_se_wgrad(d, dgate, gate, dhpre, hpre, pooled, sw));
            if (!dw_fused) TRY(launch_dw_wgrad(d, b.dZ, gate, dpooled, b.D, stats2, red2, b.E, stats1, part_w1, sw));
        }
    }
    // depthwise dgrad + BN1-backward sums; the reduction of its partial rows also fills the cb1 table
    if (bn || stats_sync_on(d)) {
        // (unfused: the reduction that also builds cb1 would use the sums before the affine fix / the cross-rank reduction)
        TRY(launch_dw_bwd_data(d, b.dZ, gate, dpooled, b.D, stats2, red2, b.E, b.x, stats1, b.dEh, red1, part, s, nullptr, dw_fused));
        TRY(stats_sync(d, red1, 2 * (size_t)d.M, s));
        if (bn) TRY(bn_bwd_fix(d0, bn, 0, red1, s));
        TRY(launch_bn1_consts(d, stats1, red1, cb1, s));
    } else {
        TRY(launch_dw_bwd_data(d, b.dZ, gate, dpooled, b.D, stats2, red2, b.E, b.x, stats1, b.dEh, red1, part, s, cb1, dw_fused));
    }
    if (d.need_wgrad) TRY(launch_expand_wgrad(d, b.dEh, b.E, cb1, b.x, part_w2, fork_to(so, 2, s)));
    if (b.dx && d.mode != TFNAS_MODE_STEM) {
        // dx = de W_expand (+ residual) without reading E: BN1-backward correction operator G | b in the top of `part`
        float* gram = part + TFNAS_PART_FLOATS - expand_gram_floats(d);
        TRY(launch_expand_gram(d, cb1, part, TFNAS_PART_FLOATS - expand_gram_floats(d), gram, s));
        TRY(launch_expand_dgrad(d, b.dEh, b.x, cb1, gram, dout_res, b.wmix, b.dx, b.dxp, s, b.add_src, b.add_scale));
    }
    return 0;
}

extern "C" int tfnas_mixedop_fwd(const TfnasCellDesc* dp, const float* x, const float* wmix, float* E, float* D,
                                 float* Pr, float* fsmall, double* stats, float* part, float* out, void* stream) {
    if (!dp || !x || !D || !Pr || !fsmall || !stats || !part || !out) return TFNAS_ENULL;
    const TfnasCellDesc& d = *dp;
    if (d.mode == TFNAS_MODE_HEAD) return TFNAS_EINVAL;
    if (!E && !efree_supported(d)) return TFNAS_ENULL;       // E may be omitted only in E-free mode (tfnas_efree_supported)
    TRY(check_modes(dp));
    TfnasCellWs ws;
    TRY(tfnas_cell_ws(dp, &ws));
    CellFwdBufs b = {x, wmix, E, D, Pr, fsmall, stats, part, out};
    return cell_fwd_impl(d, ws, b, S(stream));
}

extern "C" int tfnas_mixedop_bwd(const TfnasCellDesc* dp, const float* x, const float* wmix, const float* E,
                                 const float* D, const float* Pr, const float* fsmall, const double* stats,
                                 const float* dout, float* dZ, float* dEh, float* bsmall, double* red, float* part,
                                 float* dx, float* dxp, float* dwmix, void* stream) {
    if (!dp || !x || !D || !Pr || !fsmall || !stats || !dout || !dZ || !dEh || !bsmall || !red || !part)
        return TFNAS_ENULL;
    const TfnasCellDesc& d = *dp;
    if (d.mode == TFNAS_MODE_HEAD) return TFNAS_EINVAL;
    if (!E && !efree_supported(d)) return TFNAS_ENULL;
    TRY(check_modes(dp));
    TfnasCellWs ws;
    TRY(tfnas_cell_ws(dp, &ws));
    hipStream_t s = S(stream);
    // weight gradients: on the library's side stream (see SideCtx), scratch = second half of `part`
    SideCtx* sc = (d.need_wgrad && route_side(d)) ? side_for(s) : nullptr;
    SideJoinGuard guard;
    guard.c = sc;
    guard.main = s;
    CellSide so = {};
    if (sc) {
        for (int i = 0; i < 3; ++i) {
            so.side[i] = d.wgrad_stream[i] ? S(d.wgrad_stream[i]) : sc->side;
            guard.extra[i] = d.wgrad_stream[i] ? S(d.wgrad_stream[i]) : nullptr;
            so.fork[i] = sc->fork[i];
        }
    }
    CellBwdBufs b = {x, wmix, E, D, Pr, fsmall, stats, dout, dZ, dEh, bsmall, red, part, part + TFNAS_PART_ALLOC,
                     dx, dxp, dwmix, nullptr, nullptr};
    if (sc && (d.wgrad_stream[0] || d.wgrad_stream[1] || d.wgrad_stream[2])) {
        // forks on different streams run concurrently: one scratch piece each (tfnas_hip.h: `part` then holds FOUR pieces)
        b.part_w1 = part + 2 * TFNAS_PART_ALLOC;
        b.part_w2 = part + 3 * TFNAS_PART_ALLOC;
    }
    TRY(cell_bwd_impl(d, ws, b, s, sc ? &so : nullptr));
    return guard.join();
}

extern "C" int tfnas_mbconv_fwd(const TfnasCellDesc* dp, const TfnasBnAffine* bn, const float* drop_scale, const float* x,
                                float* E, float* D, float* Pr, float* fsmall, double* stats, float* part, float* out,
                                void* stream) {
    if (!dp || !bn || !x || !E || !D || !Pr || !fsmall || !stats || !part || !out) return TFNAS_ENULL;
    const TfnasCellDesc& d = *dp;
    if (d.mode == TFNAS_MODE_HEAD || d.G != 1) return TFNAS_EINVAL;
    TRY(check_modes(dp));
    TfnasCellWs ws;
    TRY(tfnas_cell_ws(dp, &ws));
    CellFwdBufs b = {x, nullptr, E, D, Pr, fsmall, stats, part, out};
    b.bn = bn;
    b.drop_scale = drop_scale;
    return cell_fwd_impl(d, ws, b, S(stream));
}

extern "C" int tfnas_mbconv_bwd(const TfnasCellDesc* dp, const TfnasBnAffine* bn, const float* drop_scale, const float* x,
                                const float* E, const float* D, const float* Pr, const float* fsmall, const double* stats,
                                const float* dout, float* dout_s, float* dZ, float* dEh, float* bsmall, double* red,
                                float* part, float* dx, float* dxp, void* stream) {
    if (!dp || !bn || !x || !E || !D || !Pr || !fsmall || !stats || !dout || !dZ || !dEh || !bsmall || !red || !part)
        return TFNAS_ENULL;
    const TfnasCellDesc& d = *dp;
    if (d.mode == TFNAS_MODE_HEAD || d.G != 1) return TFNAS_EINVAL;
    if (d.wgrad_stream[0] || d.wgrad_stream[1] || d.wgrad_stream[2]) return TFNAS_EINVAL;     // (tfnas_mixedop_bwd only)
    TRY(check_modes(dp));
    TfnasCellWs ws;
    TRY(tfnas_cell_ws(dp, &ws));
    hipStream_t s = S(stream);
    SideCtx* sc = (d.need_wgrad && route_side(d)) ? side_for(s) : nullptr;
    SideJoinGuard guard;
    guard.c = sc;
    guard.main = s;
    CellSide so = {};
    if (sc) {
        for (int i = 0; i < 3; ++i) {
            so.side[i] = d.wgrad_stream[i] ? S(d.wgrad_stream[i]) : sc->side;
            guard.extra[i] = d.wgrad_stream[i] ? S(d.wgrad_stream[i]) : nullptr;
            so.fork[i] = sc->fork[i];
        }
    }
    CellBwdBufs b = {x, nullptr, E, D, Pr, fsmall, stats, dout, dZ, dEh, bsmall, red, part, part + TFNAS_PART_ALLOC,
                     dx, dxp, nullptr, nullptr, nullptr};
    b.bn = bn;
    b.drop_scale = drop_scale;
    b.dout_s = dout_s;
    TRY(cell_bwd_impl(d, ws, b, s, sc ? &so : nullptr));
    if (g_lazy_join || (d.flags & TFNAS_CELL_LAZY_JOIN)) {              // the caller joins later (tfnas_side_join)
        guard.joined = true;
        return guard.join_extra();               // (caller-owned streams of THIS launch are always joined here)
    }
    return guard.join();
}

extern "C" int tfnas_head_affine_fwd(const TfnasCellDesc* dp, const TfnasBnAffine* bn, const float* x, float* E, double* stats,
                                     float* part, float* pooled, void* stream) {
    if (!dp || !bn || !x || !E || !stats || !part || !pooled) return TFNAS_ENULL;
    const TfnasCellDesc& d0 = *dp;
    if (d0.mode != TFNAS_MODE_HEAD) return TFNAS_EINVAL;
    hipStream_t s = S(stream);
    TfnasCellDesc d = d0;
    d.eps = -1.f;
    TRY(launch_expand_fwd(d, x, E, stats, part, s));
    TRY(stats_sync(d, stats, 2 * (size_t)d.M, s));
    TRY(launch_bn_fwd_fix(stats, d0.g[0].mc, (uint64_t)d0.N * d0.H * d0.W, d0.eps, bn->weight[0], bn->bias[0],
                          bn->running_mean[0], bn->running_var[0], bn->momentum, bn->eval, s, stats_world(d0)));
    TRY(launch_head_pool(d, E, stats, pooled, s));
    return 0;
}

extern "C" int tfnas_head_affine_bwd(const TfnasCellDesc* dp, const TfnasBnAffine* bn, const float* x, const float* E,
                                     const double* stats, const float* dpooled, float* dEh, float* cb1, double* red,
                                     float* part, float* dx, float* dxp, void* stream) {
    if (!dp || !bn || !x || !E || !stats || !dpooled || !dEh || !cb1 || !red || !part || !dx) return TFNAS_ENULL;
    const TfnasCellDesc& d0 = *dp;
    if (d0.mode != TFNAS_MODE_HEAD) return TFNAS_EINVAL;
    if (d0.need_wgrad && !d0.g[0].g_expand) return TFNAS_ENULL;
    hipStream_t s = S(stream);
    TfnasCellDesc d = d0;
    d.eps = -1.f;
    const uint64_t cnt = (uint64_t)d0.N * d0.H * d0.W;
    TRY(launch_head_bwd(d, E, stats, dpooled, dEh, red, part, s));
    TRY(stats_sync(d, red, 2 * (size_t)d.M, s));
    TRY(launch_bn_bwd_fix(red, d0.g[0].mc, cnt, bn->weight[0], bn->bias[0], bn->g_weight[0], bn->g_bias[0], s));
    if (bn->eval) HIP_TRY(hipMemsetAsync(red, 0, sizeof(double) * 2 * (size_t)d0.g[0].mc, s));
    TRY(launch_bn1_consts(d, stats, red, cb1, s));
    float* gram = part + TFNAS_PART_FLOATS - expand_gram_floats(d);
    TRY(launch_expand_gram(d, cb1, part, TFNAS_PART_FLOATS - expand_gram_floats(d), gram, s));
    TRY(launch_expand_dgrad(d, dEh, x, cb1, gram, nullptr, nullptr, dx, dxp, s));
    if (d.need_wgrad) TRY(launch_expand_wgrad(d, dEh, E, cb1, x, part, s));
    return 0;
}

extern "C" int tfnas_head_fwd(const TfnasCellDesc* dp, const float* x, float* E, double* stats, float* part,
                              float* pooled, void* stream) {
    if (!dp || !x || !E || !stats || !part || !pooled) return TFNAS_ENULL;
    const TfnasCellDesc& d = *dp;
    if (d.mode != TFNAS_MODE_HEAD) return TFNAS_EINVAL;
    hipStream_t s = S(stream);
    TRY(launch_expand_fwd(d, x, E, stats, part, s));          // 1x1 conv 320->1280 + BN statistics
    TRY(stats_sync(d, stats, 2 * (size_t)d.M, s));
    TRY(launch_head_pool(d, E, stats, pooled, s));            // BN + swish + global average pool
    return 0;
}

extern "C" int tfnas_head_bwd(const TfnasCellDesc* dp, const float* x, const float* E, const double* stats,
                              const float* dpooled, float* dEh, float* cb1, double* red, float* part, float* dx,
                              float* dxp, void* stream) {
    if (!dp || !x || !E || !stats || !dpooled || !dEh || !cb1 || !red || !part || !dx) return TFNAS_ENULL;
    const TfnasCellDesc& d = *dp;
    if (d.mode != TFNAS_MODE_HEAD) return TFNAS_EINVAL;
    if (d.need_wgrad && !d.g[0].g_expand) return TFNAS_ENULL;
    hipStream_t s = S(stream);
    TRY(launch_head_bwd(d, E, stats, dpooled, dEh, red, part, s));       // pool + swish backward, BN-backward sums
    TRY(stats_sync(d, red, 2 * (size_t)d.M, s));
    TRY(launch_bn1_consts(d, stats, red, cb1, s));
    float* gram = part + TFNAS_PART_FLOATS - expand_gram_floats(d);
    TRY(launch_expand_gram(d, cb1, part, TFNAS_PART_FLOATS - expand_gram_floats(d), gram, s));
    TRY(launch_expand_dgrad(d, dEh, x, cb1, gram, nullptr, nullptr, dx, dxp, s));
    if (d.need_wgrad) TRY(launch_expand_wgrad(d, dEh, E, cb1, x, part, s));
    return 0;
}

extern "C" int tfnas_head_wgrad(const TfnasCellDesc* dp, const float* x, const float* E, const float* dEh, const float* cb1,
                                float* part, void* stream) {
    if (!dp || !x || !E || !dEh || !cb1 || !part) return TFNAS_ENULL;
    const TfnasCellDesc& d = *dp;
    if (d.mode != TFNAS_MODE_HEAD) return TFNAS_EINVAL;
    if (!d.g[0].g_expand) return TFNAS_ENULL;
    return launch_expand_wgrad(d, dEh, E, cb1, x, part, S(stream));
}

extern "C" int tfnas_arch_fwd(int ncell, const float* const* log_alpha, const float* e, const float* lat, float T,
                              float* w, float* cell_lat, void* stream) {
    if (ncell < 1 || ncell > TFNAS_MAX_CELLS) return TFNAS_ERANGE;
    if (!log_alpha || !e || !w) return TFNAS_ENULL;
    return launch_arch_fwd(ncell, log_alpha, e, lat, T, w, cell_lat, S(stream));
}

extern "C" int tfnas_arch_bwd(int ncell, const float* w, const float* lat, const float* dw, const float* dcell_lat,
                              float T, float* const* dlog_alpha, void* stream) {
    if (ncell < 1 || ncell > TFNAS_MAX_CELLS) return TFNAS_ERANGE;
    if (!w || !dlog_alpha) return TFNAS_ENULL;
    return launch_arch_bwd(ncell, w, lat, dw, dcell_lat, T, dlog_alpha, S(stream));
}

extern "C" int tfnas_arch_project(int n, float* const* p, const int32_t* len, void* stream) {
    if (n < 1 || n > TFNAS_MAX_CELLS) return TFNAS_ERANGE;
    if (!p || !len) return TFNAS_ENULL;
    for (int i = 0; i < n; ++i) {
        if (!p[i]) return TFNAS_ENULL;
        if (len[i] < 1 || len[i] > 8) return TFNAS_ERANGE;
    }
    return launch_arch_project(n, p, len, S(stream));
}

extern "C" int tfnas_arch_sample(int ncell, const float* const* log_alpha, const uint8_t* mask, const float* e,
                                 float T, int mode, int32_t* pos_out, void* stream) {
    if (ncell < 1 || ncell > TFNAS_MAX_CELLS) return TFNAS_ERANGE;
    if (!log_alpha || !mask || !pos_out || (mode == 0 && !e)) return TFNAS_ENULL;
    if (mode < 0 || mode > 2) return TFNAS_EINVAL;
    return launch_arch_sample(ncell, log_alpha, mask, e, T, mode, pos_out, S(stream));
}

extern "C" int tfnas_sink_fwd(int K, const float* betas, const float* const* res, const float* cell_lat,
                              uint64_t count, float* out, float* out_lat, float* bw_out, void* stream) {
    if (K < 1 || K > TFNAS_MAX_SINK) return TFNAS_ERANGE;
    if (!betas || !res || !out || !bw_out) return TFNAS_ENULL;
    if (count & 3) return TFNAS_EINVAL;
    return launch_sink_fwd(K, betas, res, cell_lat, count, out, out_lat, bw_out, S(stream));
}

extern "C" int tfnas_sink_bwd(int K, const float* bw, const float* const* res, const float* cell_lat,
                              const float* dout, const float* dlat, uint64_t count, float* const* dres,
                              float* dbetas, float* dcell_lat, double* dot_scratch, void* stream) {
    if (K < 1 || K > TFNAS_MAX_SINK) return TFNAS_ERANGE;
    if (!bw || !res || !dout || !dres || !dot_scratch) return TFNAS_ENULL;
    if (count & 3) return TFNAS_EINVAL;
    return launch_sink_bwd(K, bw, res, cell_lat, dout, dlat, count, dres, dbetas, dcell_lat, dot_scratch, S(stream));
}

extern "C" int tfnas_pack_ranges(const float* src, float* dst, int nranges, const uint64_t* off, const uint64_t* doff,
                                 const uint64_t* len, void* stream) {
    if (!src || !dst || !off || !doff || !len) return TFNAS_ENULL;
    return launch_pack_ranges(src, dst, nranges, off, doff, len, S(stream));
}

extern "C" int tfnas_sgd_clip_step(float* w, float* g, float* m, int nranges, const uint64_t* off, const uint64_t* goff,
                                   const uint64_t* len, float max_norm, float lr, float momentum, float wd, float grad_scale, double* scratch,
                                   uint64_t scratch_doubles, float* norm_out, void* stream) {
    if (!w || !g || !m || !off || !len || !scratch) return TFNAS_ENULL;
    return launch_sgd_clip_step(w, g, m, nranges, off, goff, len, max_norm, lr, momentum, wd, grad_scale, scratch, scratch_doubles,
                                norm_out, S(stream));
}

extern "C" int tfnas_arch_adam_project(int n, float* const* p, const float* const* g, const int32_t* len, float* m, float* v,
                                       float max_norm, float lr, float beta1, float beta2, float eps, float wd, int step,
                                       float grad_scale, float* norm_out, void* stream) {
    if (!p || !g || !len || !m || !v) return TFNAS_ENULL;
    if (step < 1) return TFNAS_EINVAL;
    return launch_arch_adam_project(n, p, g, len, m, v, max_norm, lr, beta1, beta2, eps, wd, step, grad_scale, norm_out,
                                    S(stream));
}
