// Path level of the C ABI (include/tfnas_hip.h: tfnas_path_*): a whole chain of MixedOP cells + sink-connecting stage
// mixes per call, all buffers from one caller-allocated arena, several paths enqueued interleaved on their own streams.
// Reference being replaced: Network.forward's stage loop (models/model_search.py:285-297), MixedStage.forward (:157-206)
// and the autograd backward of both.  The kernels are the per-cell ones (cell_impl.h); what changes is who enqueues them.
#include <string.h>
#include <new>
#include "tfnas_dev.h"
#include "kernels.h"
#include "cell_impl.h"

namespace {

constexpr uint64_t ALIGN = 64;                       // floats (256 bytes)
inline uint64_t up(uint64_t v) { return (v + ALIGN - 1) / ALIGN * ALIGN; }

struct CellOff {
    uint64_t E, D, Pr, fsmall, stats, out;           // float offsets into the arena (E == ~0: E-free)
};
struct StageOff {
    uint64_t sink_out, bw, dots, bound;              // bound: gradient of this stage's output (not for the last stage)
    uint64_t count;                                  // elements of the stage output
};
struct ScratchSet {
    uint64_t dZ, dEh, bsmall, red, part, part_w;
};

// The weight-gradient kernels of a cell run on the side stream and may finish up to LAG cells after the data-gradient chain
// has moved on: LAG + 1 scratch sets and LAG + 2 gradient-ring slots keep everything they read alive that long.
#ifndef TFNAS_LAG
#define TFNAS_LAG 2
#endif
constexpr int LAG = TFNAS_LAG;
constexpr int NSET = LAG + 1, NRING = LAG + 2;

struct PathCtx {
    int device = 0;
    bool planned = false;
    TfnasPathDesc pd;
    TfnasCellWs cws[TFNAS_MAX_CELLS];
    CellOff co[TFNAS_MAX_CELLS];
    StageOff so[TFNAS_MAX_STAGES];
    int stage_of[TFNAS_MAX_CELLS];
    ScratchSet set[NSET];
    uint64_t ring[NRING], dxp;
    TfnasPathWs ws;
    hipStream_t side = nullptr;
    bool own_side = true;              // false: the caller supplied the side stream (tfnas_path_set_side_stream)
    hipEvent_t fork[TFNAS_MAX_CELLS][3];
    hipEvent_t wdone[TFNAS_MAX_CELLS];
    hipEvent_t join = nullptr;
    bool events_ok = false;
};

inline hipStream_t S(void* s) { return reinterpret_cast<hipStream_t>(s); }

#define TRY(call)               \
    do {                        \
        int _r = (call);        \
        if (_r != 0) return _r; \
    } while (0)

int plan_path(PathCtx& c, const TfnasPathDesc& in, TfnasPathWs* out) {
    if (in.ncell < 1 || in.ncell > TFNAS_MAX_CELLS || in.nstage < 1 || in.nstage > TFNAS_MAX_STAGES) return TFNAS_ERANGE;
    c.planned = false;
    c.pd = in;
    TfnasPathDesc& pd = c.pd;
    // stages: first_cell / nres, every cell belongs to exactly one stage
    int nc = 0;
    for (int st = 0; st < pd.nstage; ++st) {
        TfnasStage& sg = pd.stage[st];
        if (sg.ncell < 1 || sg.ncell + 1 > TFNAS_MAX_SINK + 1 || (sg.start_res != 0 && sg.start_res != 1)) return TFNAS_EINVAL;
        sg.first_cell = nc;
        sg.nres = sg.ncell + 1 - sg.start_res;
        if (sg.nres < 1 || sg.nres > TFNAS_MAX_SINK) return TFNAS_ERANGE;
        if (!sg.betas) return TFNAS_ENULL;
        for (int j = 0; j < sg.ncell; ++j) c.stage_of[nc + j] = st;
        nc += sg.ncell;
    }
    if (nc != pd.ncell) return TFNAS_EINVAL;
    if (pd.reserved0 != 0) return TFNAS_EINVAL;          // (was `dual`: both bi-sampling paths in one descriptor; removed)
    // cells: chain the geometry, plan, size
    for (int i = 0; i < pd.ncell; ++i) {
        TfnasCellDesc& d = pd.cell[i];
        if (d.mode != TFNAS_MODE_CELL) return TFNAS_EINVAL;
        if (!pd.soft && d.G != 1) return TFNAS_EINVAL;            // a sampled path evaluates one candidate per cell
        if (i > 0) {
            const TfnasCellDesc& p = pd.cell[i - 1];
            if (d.ic != p.oc) return TFNAS_EINVAL;
            d.N = p.N;
            d.H = p.Ho;
            d.W = p.Wo;
        }
        TRY(tfnas_cell_plan(&d));
        TRY(tfnas_cell_ws(&d, &c.cws[i]));
        const bool efree = (pd.efree_mask_lo >> i) & 1;
        if (efree && (d.need_wgrad || !efree_supported(d))) return TFNAS_EINVAL;
    }
    // a stage whose input is a depth choice must keep the extent (ic == oc, stride 1 in its first cell)
    for (int st = 0; st < pd.nstage; ++st) {
        const TfnasStage& sg = pd.stage[st];
        const TfnasCellDesc& f = pd.cell[sg.first_cell];
        if (sg.start_res == 0 && (f.ic != f.oc || f.stride != 1)) return TFNAS_EINVAL;
        for (int j = 1; j < sg.ncell; ++j) {
            const TfnasCellDesc& d = pd.cell[sg.first_cell + j];
            if (d.ic != d.oc || d.stride != 1) return TFNAS_EINVAL;     // depth outputs of a stage share one shape
        }
    }
    // ---- arena layout.  First, at FIXED offsets (the same for every plan of this context, whatever widths it samples): the
    // `part` scratch pieces (kept at fixed offsets since round 2; nothing depends on it any more)
    // stay zero between launches (kernels.h), so they may never land on memory another plan used for something else.
    uint64_t off = 0;
    for (int k = 0; k < NSET; ++k) {
        ScratchSet& s = c.set[k];
        s.part = off; off += up(TFNAS_PART_ALLOC);
        s.part_w = off; off += up(TFNAS_PART_ALLOC);
    }
    // ---- saved region
    for (int i = 0; i < pd.ncell; ++i) {
        const TfnasCellWs& w = c.cws[i];
        const bool efree = (pd.efree_mask_lo >> i) & 1;
        CellOff& o = c.co[i];
        o.E = ~(uint64_t)0;
        if (!efree) { o.E = off; off += up(w.E); }
        o.D = off; off += up(w.D);
        o.Pr = off; off += up(w.Pr);
        o.fsmall = off; off += up(w.fsmall);
        o.stats = off; off += up(2 * w.stats);
        o.out = off; off += up(w.out);
    }
    for (int st = 0; st < pd.nstage; ++st) {
        const TfnasStage& sg = pd.stage[st];
        const TfnasCellDesc& l = pd.cell[sg.first_cell + sg.ncell - 1];
        StageOff& o = c.so[st];
        o.count = (uint64_t)l.N * l.Ho * l.Wo * l.oc;
        o.sink_out = ~(uint64_t)0;
        if (st + 1 < pd.nstage) { o.sink_out = off; off += up(o.count); }
        o.bw = off; off += ALIGN;
        o.dots = off; off += ALIGN;
    }
    const uint64_t saved = off;
    // ---- scratch region
    uint64_t mdZ = 4, mdEh = 4, mbs = 4, mred = 4, mring = 4, mdxp = 4;
    for (int i = 0; i < pd.ncell; ++i) {
        const TfnasCellWs& w = c.cws[i];
        const TfnasCellDesc& d = pd.cell[i];
        const bool light = pd.soft && i == 0 && !pd.need_dx0 && !d.need_wgrad;   // only d wmix: no dZ / dEh
        if (!light) {
            mdZ = w.dZ > mdZ ? w.dZ : mdZ;
            mdEh = w.dEh > mdEh ? w.dEh : mdEh;
        }
        mbs = w.bsmall > mbs ? w.bsmall : mbs;
        mred = w.red > mred ? w.red : mred;
        mring = w.dx > mring ? w.dx : mring;
        mring = w.out > mring ? w.out : mring;
        mdxp = w.dxp > mdxp ? w.dxp : mdxp;
    }
    for (int k = 0; k < NSET; ++k) {
        ScratchSet& s = c.set[k];
        s.dZ = off; off += up(mdZ);
        s.dEh = off; off += up(mdEh);
        s.bsmall = off; off += up(mbs);
        s.red = off; off += up(2 * mred);
    }
    for (int k = 0; k < NRING; ++k) { c.ring[k] = off; off += up(mring); }
    c.dxp = off; off += up(mdxp);
    for (int st = 0; st + 1 < pd.nstage; ++st) { c.so[st].bound = off; off += up(c.so[st].count); }
    c.so[pd.nstage - 1].bound = ~(uint64_t)0;
    const TfnasCellDesc& last = pd.cell[pd.ncell - 1];
    c.ws.saved = saved;
    c.ws.scratch = off - saved;
    c.ws.total = off;
    c.ws.out_count = c.so[pd.nstage - 1].count;
    c.ws.out_h = last.Ho;
    c.ws.out_w = last.Wo;
    c.ws.out_c = last.oc;
    c.ws.pad = 0;
    if (out) *out = c.ws;
    c.planned = true;
    return 0;
}

inline const float* stage_input(const PathCtx& c, int st, const float* x0, const float* arena) {
    return st == 0 ? x0 : arena + c.so[st - 1].sink_out;
}

int fwd_cell(PathCtx& c, int i, const float* x0, const float* wmix, float* arena, hipStream_t s) {
    const TfnasPathDesc& pd = c.pd;
    const int st = c.stage_of[i];
    const TfnasStage& sg = pd.stage[st];
    const CellOff& o = c.co[i];
    CellFwdBufs b;
    b.x = (i == sg.first_cell) ? stage_input(c, st, x0, arena) : arena + c.co[i - 1].out;
    b.wmix = wmix ? wmix + (size_t)i * TFNAS_MAX_GROUPS : nullptr;
    b.E = o.E == ~(uint64_t)0 ? nullptr : arena + o.E;
    b.D = arena + o.D;
    b.Pr = arena + o.Pr;
    b.fsmall = arena + o.fsmall;
    b.stats = reinterpret_cast<double*>(arena + o.stats);
    b.part = arena + c.set[0].part;
    b.out = arena + o.out;
    return cell_fwd_impl(pd.cell[i], c.cws[i], b, s);
}

// res pointers of a stage's sink (model_search.py:200-204: res_list[start_res:])
void sink_res(const PathCtx& c, int st, const float* x0, const float* arena, const float* res[TFNAS_MAX_SINK]) {
    const TfnasStage& sg = c.pd.stage[st];
    for (int k = 0; k < sg.nres; ++k) {
        const int r = sg.start_res + k;                    // index into [stage input, out_1, ..., out_K]
        res[k] = r == 0 ? stage_input(c, st, x0, arena) : arena + c.co[sg.first_cell + r - 1].out;
    }
}

int fwd_sink(PathCtx& c, int st, const float* x0, const float* cell_lat_x, int lat_off, float* arena, float* out,
             float* out_lat, hipStream_t s) {
    const TfnasStage& sg = c.pd.stage[st];
    const float* res[TFNAS_MAX_SINK];
    sink_res(c, st, x0, arena, res);
    float* dst = (st + 1 < c.pd.nstage) ? arena + c.so[st].sink_out : out;
    return launch_sink_fwd(sg.nres, sg.betas, res, cell_lat_x ? cell_lat_x + lat_off : nullptr, c.so[st].count, dst,
                           out_lat ? out_lat + st : nullptr, arena + c.so[st].bw, s);
}

}   // namespace

extern "C" int tfnas_path_create(void** ctx) {
    if (!ctx) return TFNAS_ENULL;
    PathCtx* c = new (std::nothrow) PathCtx();
    if (!c) return TFNAS_ERANGE;
    if (hipGetDevice(&c->device) != hipSuccess) { delete c; return (int)hipGetLastError(); }
    *ctx = c;
    return 0;
}

// The side stream and its events are created by the first backward that has weight gradients to run: HIP multiplexes
// streams onto GPU_MAX_HW_QUEUES hardware queues round-robin in creation order, so every stream that exists but is never
// used (soft-mode contexts, size probes) shifts the mapping and can put a path's data-gradient chain and its own
// weight-gradient stream on the SAME hardware queue (measured: w-step 21 -> 23.5 ms).
static int ensure_side(PathCtx* c) {
    if (c->events_ok) return 0;
    hipError_t e = hipSuccess;
    if (!c->side) {
        e = hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking);
        c->own_side = true;
    }
    for (int i = 0; i < TFNAS_MAX_CELLS && e == hipSuccess; ++i) {
        for (int k = 0; k < 3 && e == hipSuccess; ++k) e = hipEventCreateWithFlags(&c->fork[i][k], hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&c->wdone[i], hipEventDisableTiming);
    }
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->join, hipEventDisableTiming);
    if (e != hipSuccess) return (int)e;       // (a failed create leaks a few events; the process is unusable anyway)
    c->events_ok = true;
    return 0;
}

extern "C" int tfnas_path_destroy(void* ctx) {
    PathCtx* c = static_cast<PathCtx*>(ctx);
    if (!c) return 0;
    if (c->events_ok) {
        (void)hipStreamSynchronize(c->side);
        for (int i = 0; i < TFNAS_MAX_CELLS; ++i) {
            for (int k = 0; k < 3; ++k) (void)hipEventDestroy(c->fork[i][k]);
            (void)hipEventDestroy(c->wdone[i]);
        }
        (void)hipEventDestroy(c->join);
        if (c->own_side) (void)hipStreamDestroy(c->side);
    }
    delete c;
    return 0;
}

extern "C" int tfnas_path_set_side_stream(void* ctx, void* stream) {
    PathCtx* c = static_cast<PathCtx*>(ctx);
    if (!c || !stream) return TFNAS_ENULL;
    if (c->side && c->own_side) {
        (void)hipStreamSynchronize(c->side);
        (void)hipStreamDestroy(c->side);
    }
    c->side = S(stream);
    c->own_side = false;
    return 0;
}

static int join_sides(PathCtx* c, hipStream_t s) {
    hipError_t e = hipEventRecord(c->join, c->side);
    if (e == hipSuccess) e = hipStreamWaitEvent(s, c->join, 0);
    if (e != hipSuccess) (void)hipStreamSynchronize(c->side);
    return (int)e;
}

extern "C" int tfnas_path_plan(void* ctx, const TfnasPathDesc* pd, TfnasPathWs* ws) {
    if (!ctx || !pd) return TFNAS_ENULL;
    return plan_path(*static_cast<PathCtx*>(ctx), *pd, ws);
}

static int check_paths(int npath, void* const* ctx) {
    if (npath < 1 || npath > 4) return TFNAS_ERANGE;
    if (!ctx) return TFNAS_ENULL;
    for (int p = 0; p < npath; ++p) {
        const PathCtx* c = static_cast<const PathCtx*>(ctx[p]);
        if (!c) return TFNAS_ENULL;
        if (!c->planned) return TFNAS_EINVAL;
        const PathCtx* c0 = static_cast<const PathCtx*>(ctx[0]);
        if (c->pd.ncell != c0->pd.ncell || c->pd.nstage != c0->pd.nstage) return TFNAS_EINVAL;
        for (int st = 0; st < c->pd.nstage; ++st)
            if (c->pd.stage[st].ncell != c0->pd.stage[st].ncell) return TFNAS_EINVAL;   // interleaving walks them in lockstep
    }
    return 0;
}

extern "C" int tfnas_paths_fwd(int npath, void* const* ctx, const float* const* x0, const float* const* wmix,
                               const float* const* cell_lat, float* const* arena, float* const* out,
                               float* const* out_lat, void* const* streams) {
    TRY(check_paths(npath, ctx));
    if (!x0 || !arena || !out || !streams) return TFNAS_ENULL;
    for (int p = 0; p < npath; ++p) {
        const PathCtx* c = static_cast<const PathCtx*>(ctx[p]);
        if (!x0[p] || !arena[p] || !out[p]) return TFNAS_ENULL;
        if (c->pd.soft && (!wmix || !wmix[p])) return TFNAS_ENULL;
    }
    PathCtx* c0 = static_cast<PathCtx*>(ctx[0]);
    int lat_off = 0;
    for (int st = 0; st < c0->pd.nstage; ++st) {
        const TfnasStage& sg = c0->pd.stage[st];
        for (int j = 0; j < sg.ncell; ++j)
            for (int p = 0; p < npath; ++p) {
                PathCtx* c = static_cast<PathCtx*>(ctx[p]);
                TRY(fwd_cell(*c, sg.first_cell + j, x0[p], (wmix && c->pd.soft) ? wmix[p] : nullptr, arena[p], S(streams[p])));
            }
        for (int p = 0; p < npath; ++p) {
            PathCtx* c = static_cast<PathCtx*>(ctx[p]);
            TRY(fwd_sink(*c, st, x0[p], (cell_lat && c->pd.soft) ? cell_lat[p] : nullptr, lat_off, arena[p], out[p],
                         (out_lat && c->pd.soft) ? out_lat[p] : nullptr, S(streams[p])));
        }
        lat_off += sg.nres;
    }
    return 0;
}

namespace {

// everything the backward of one path needs while it is being walked
struct BwdRun {
    PathCtx* c;
    const float *x0, *wmix, *cell_lat;
    float* arena;
    const float *dout, *dout_lat;
    float *dx0, *dwmix, *dcell_lat;
    hipStream_t s;
    bool side_on;
};

// before cell i (or the sink step in front of it) reuses scratch set i % NSET and ring slot (i+1) % NRING, the
// weight-gradient kernels of cell i + NSET -- the last ones that read them -- must be done (side stream, <= LAG cells behind)
int wait_wgrads(BwdRun& r, int i) {
    const int j = i + NSET;
    if (!r.side_on || j >= r.c->pd.ncell || !r.c->pd.cell[j].need_wgrad) return 0;
    return (int)hipStreamWaitEvent(r.s, r.c->wdone[j], 0);
}

int bwd_sink(BwdRun& r, int st, int lat_off) {
    PathCtx& c = *r.c;
    const TfnasStage& sg = c.pd.stage[st];
    const int last = sg.first_cell + sg.ncell - 1;
    TRY(wait_wgrads(r, last));
    const float* dsink = (st + 1 < c.pd.nstage) ? r.arena + c.so[st].bound : r.dout;
    float* dlast = r.arena + c.ring[(last + 1) % NRING];
    const float* bw = r.arena + c.so[st].bw;
    const bool want_lat = c.pd.soft && r.dcell_lat && r.cell_lat;
    if (!sg.dbetas && !want_lat) return launch_scale_copy(dlast, dsink, bw + (sg.nres - 1), c.so[st].count, r.s);
    const float* res[TFNAS_MAX_SINK];
    float* dres[TFNAS_MAX_SINK];
    sink_res(c, st, r.x0, r.arena, res);
    for (int k = 0; k < sg.nres; ++k) dres[k] = (k == sg.nres - 1) ? dlast : nullptr;
    return launch_sink_bwd(sg.nres, bw, res, want_lat ? r.cell_lat + lat_off : nullptr, dsink,
                           r.dout_lat ? r.dout_lat + st : nullptr, c.so[st].count, dres, sg.dbetas,
                           want_lat ? r.dcell_lat + lat_off : nullptr, reinterpret_cast<double*>(r.arena + c.so[st].dots),
                           r.s);
}

int bwd_cell(BwdRun& r, int i) {
    PathCtx& c = *r.c;
    const TfnasPathDesc& pd = c.pd;
    const int st = c.stage_of[i];
    const TfnasStage& sg = pd.stage[st];
    const int j = i - sg.first_cell;                     // this cell's input is res_list[j] of its stage
    const CellOff& o = c.co[i];
    const ScratchSet& ss = c.set[i % NSET];
    if (j != sg.ncell - 1) TRY(wait_wgrads(r, i));       // (the stage's last cell waited in bwd_sink)
    CellBwdBufs b;
    b.x = (j == 0) ? stage_input(c, st, r.x0, r.arena) : r.arena + c.co[i - 1].out;
    b.wmix = (r.wmix && pd.soft) ? r.wmix + (size_t)i * TFNAS_MAX_GROUPS : nullptr;
    b.E = o.E == ~(uint64_t)0 ? nullptr : r.arena + o.E;
    b.D = r.arena + o.D;
    b.Pr = r.arena + o.Pr;
    b.fsmall = r.arena + o.fsmall;
    b.stats = reinterpret_cast<const double*>(r.arena + o.stats);
    b.dout = r.arena + c.ring[(i + 1) % NRING];
    b.dZ = r.arena + ss.dZ;
    b.dEh = r.arena + ss.dEh;
    b.bsmall = r.arena + ss.bsmall;
    b.red = reinterpret_cast<double*>(r.arena + ss.red);
    b.part = r.arena + ss.part;
    b.part_w = r.arena + ss.part_w;
    b.dxp = r.arena + c.dxp;
    b.dwmix = (r.dwmix && pd.soft) ? r.dwmix + (size_t)i * TFNAS_MAX_GROUPS : nullptr;
    // where the input gradient goes: the ring (next cell's dout), the previous stage's boundary buffer, or the caller
    if (j > 0) b.dx = r.arena + c.ring[i % NRING];
    else if (st > 0) b.dx = r.arena + c.so[st - 1].bound;
    else b.dx = pd.need_dx0 ? r.dx0 : nullptr;
    // the cell's input also feeds the stage's sink when it is a depth choice: fold bw * dsink into the dx epilogue
    b.add_src = nullptr;
    b.add_scale = nullptr;
    if (j >= sg.start_res && b.dx) {
        b.add_src = (st + 1 < pd.nstage) ? r.arena + c.so[st].bound : r.dout;
        b.add_scale = r.arena + c.so[st].bw + (j - sg.start_res);
    }
    CellSide so = {};
    const bool side = r.side_on && pd.cell[i].need_wgrad;
    if (side) {
        for (int k = 0; k < 3; ++k) {
            so.side[k] = c.side;
            so.fork[k] = c.fork[i][k];
        }
    }
    TRY(cell_bwd_impl(pd.cell[i], c.cws[i], b, r.s, side ? &so : nullptr));
    if (side) HIP_TRY(hipEventRecord(c.wdone[i], c.side));
    return 0;
}

}   // namespace

extern "C" int tfnas_paths_bwd(int npath, void* const* ctx, const float* const* x0, const float* const* wmix,
                               const float* const* cell_lat, float* const* arena, const float* const* dout,
                               const float* const* dout_lat, float* const* dx0, float* const* dwmix,
                               float* const* dcell_lat, void* const* streams, int stage_begin, int stage_end) {
    TRY(check_paths(npath, ctx));
    {
        const int ns = static_cast<const PathCtx*>(ctx[0])->pd.nstage;
        if (stage_end < 0) stage_end = ns;                   // (-1: up to the last stage)
        if (stage_begin < 0 || stage_begin >= stage_end || stage_end > ns) return TFNAS_ERANGE;
    }
    if (!x0 || !arena || !dout || !streams) return TFNAS_ENULL;
    BwdRun run[4];
    for (int p = 0; p < npath; ++p) {
        PathCtx* c = static_cast<PathCtx*>(ctx[p]);
        if (!x0[p] || !arena[p] || !dout[p]) return TFNAS_ENULL;
        if (c->pd.need_dx0 && (!dx0 || !dx0[p])) return TFNAS_ENULL;
        if (c->pd.soft && (!wmix || !wmix[p] || !dwmix || !dwmix[p])) return TFNAS_ENULL;
        BwdRun& r = run[p];
        r.c = c;
        r.x0 = x0[p];
        r.wmix = wmix ? wmix[p] : nullptr;
        r.cell_lat = cell_lat ? cell_lat[p] : nullptr;
        r.arena = arena[p];
        r.dout = dout[p];
        r.dout_lat = dout_lat ? dout_lat[p] : nullptr;
        r.dx0 = dx0 ? dx0[p] : nullptr;
        r.dwmix = dwmix ? dwmix[p] : nullptr;
        r.dcell_lat = dcell_lat ? dcell_lat[p] : nullptr;
        r.s = S(streams[p]);
        bool any_w = false, inl = false;
        for (int i = 0; i < c->pd.ncell; ++i) {
            any_w = any_w || c->pd.cell[i].need_wgrad;
            inl = inl || !route_side(c->pd.cell[i]);          // (TFNAS_ROUTE_WGRAD_INLINE on any cell: the whole path runs without a side stream)
        }
        r.side_on = !inl && any_w;
        if (r.side_on) TRY(ensure_side(c));
    }
    const PathCtx* c0 = run[0].c;
    int lat_off_end = 0;
    for (int st = 0; st < c0->pd.nstage; ++st) lat_off_end += c0->pd.stage[st].nres;
    int rc = 0;
    int lat_off = lat_off_end;
    for (int st = c0->pd.nstage - 1; st >= stage_begin && rc == 0; --st) {
        const TfnasStage& sg = c0->pd.stage[st];
        lat_off -= sg.nres;
        if (st >= stage_end) continue;                       // (a later segment of the same backward: already done)
        for (int p = 0; p < npath && rc == 0; ++p) rc = bwd_sink(run[p], st, lat_off);
        for (int j = sg.ncell - 1; j >= 0 && rc == 0; --j)
            for (int p = 0; p < npath && rc == 0; ++p) rc = bwd_cell(run[p], sg.first_cell + j);
    }
    // join every side stream (also on an error path: the caller frees / reuses the arena next)
    for (int p = 0; p < npath; ++p) {
        BwdRun& r = run[p];
        if (!r.side_on) continue;
        const int e = join_sides(r.c, r.s);
        if (e != 0 && rc == 0) rc = e;
    }
    return rc;
}
