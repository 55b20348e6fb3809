// Internal launcher prototypes (one per kernel family); the extern "C" surface is in capi.hip.
#pragma once
#include <stdlib.h>
#include <hip/hip_runtime.h>
#include "tfnas_hip.h"

// cross-rank BatchNorm statistics hook (tfnas_set_stats_sync, capi.hip)
struct StatsSync {
    tfnas_stats_sync_fn fn;
    void* user;
    int world;
};
extern StatsSync g_stats_sync;              // the process default (tfnas_set_stats_sync); a descriptor's own hook wins
static inline StatsSync sync_of(const TfnasCellDesc& d) {
    if (d.sync_fn) return StatsSync{d.sync_fn, d.sync_user, d.sync_world > 0 ? d.sync_world : 1};
    return g_stats_sync;
}
static inline bool stats_sync_on(const TfnasCellDesc& d) { return sync_of(d).fn != nullptr; }
static inline int stats_sync(const TfnasCellDesc& d, double* table, size_t ndoubles, hipStream_t s) {
    const StatsSync y = sync_of(d);
    return y.fn ? y.fn(y.user, table, (uint64_t)ndoubles, (void*)s) : 0;
}
static inline uint64_t stats_world(const TfnasCellDesc& d) {
    const StatsSync y = sync_of(d);
    return (y.fn && y.world > 1) ? (uint64_t)y.world : 1;
}

// route switches of a launch (TfnasCellDesc.route, tfnas_hip.h: TFNAS_ROUTE_*); 0 = the library's measured per-launch policy
static inline int route_dw(const TfnasCellDesc& d) { return (d.route & TFNAS_ROUTE_DW_MASK) >> TFNAS_ROUTE_DW_SHIFT; }   // 0 auto, 1 direct, 2 lds, 3 tiled
static inline int route_se(const TfnasCellDesc& d) { return (d.route & TFNAS_ROUTE_SE_MASK) >> TFNAS_ROUTE_SE_SHIFT; }   // 0 wave, 1 fused, 2 gemm
static inline bool route_side(const TfnasCellDesc& d) { return !(d.route & TFNAS_ROUTE_WGRAD_INLINE); }

// gemm_kernels.hip
int gemm_mode();            // arithmetic of the row-tiled GEMMs (tfnas_hip.h: TFNAS_GEMM_*)
int set_gemm_mode(int m);
int launch_expand_fwd(const TfnasCellDesc& d, const float* x, float* E, double* stats1, float* part,
                      hipStream_t s);
int launch_project_fwd(const TfnasCellDesc& d, const float* D, const float* gate, const double* stats2,
                       float* Pr, double* stats3, float* part, hipStream_t s);
// D / stats2 / rec non-NULL: FOLD variant -- the per-image BN2-backward tables are accumulated in the epilogue into `rec`
// (project_fold_ok(d, floats of rec) must hold); launch_bn2_gather then replaces launch_bn2_pool
int launch_project_dgrad(const TfnasCellDesc& d, const float* dout, const float* Pr, const double* stats3,
                         const double* red3, const float* wmix, float* dZ, hipStream_t s, const float* D = nullptr,
                         const double* stats2 = nullptr, float* rec = nullptr);
constexpr int FOLD_SLOTS = 4, FOLD_Q = 5;       // records of the FOLD epilogue: rec[((row_tile * 4 + slot) * 5 + q) * M + channel]
static inline bool project_fold_ok(const TfnasCellDesc& d, size_t scratch_floats) {
    const int HW = d.Ho * d.Wo;
    const size_t nrt = ((size_t)d.N * HW + 127) / 128;
    return HW >= 43 && nrt * FOLD_SLOTS * FOLD_Q * (size_t)d.M <= scratch_floats;
}
int launch_project_wgrad(const TfnasCellDesc& d, const float* dout, const float* Pr, const float* D,
                         const float* gate, const double* stats2, const double* stats3, const double* red3,
                         const float* wmix, float* part, hipStream_t s);
size_t expand_gram_floats(const TfnasCellDesc& d);
int launch_expand_gram(const TfnasCellDesc& d, const float* cb1, float* scratch, size_t scratch_floats, float* gram,
                       hipStream_t s);
int launch_expand_dgrad(const TfnasCellDesc& d, const float* dEh, const float* x, const float* cb1, const float* gram,
                        const float* dout, const float* wmix, float* dx, float* dxp, hipStream_t s,
                        const float* add_src = nullptr, const float* add_scale = nullptr);
int expand_dgrad_splits(const TfnasCellDesc& d);

int launch_expand_dgrad_x(const TfnasCellDesc& d, const float* x, const float* cb1, const float* gram, const float* dout,
                          const float* wmix, float* dx, float* dxp, int nsl, hipStream_t s, const float* add_src = nullptr,
                          const float* add_scale = nullptr);
int launch_expand_wgrad(const TfnasCellDesc& d, const float* dEh, const float* E, const float* cb1,
                        const float* x, float* part, hipStream_t s);

// dwconv_kernels.hip
// E == nullptr: E-free mode (efree.h) -- the expanded activation is recomputed from x inside the depthwise kernels
int launch_dw_fwd(const TfnasCellDesc& d, const float* E, const float* x, const double* stats1, float* D,
                  double* stats2, float* part, hipStream_t s);
bool efree_supported(const TfnasCellDesc& d);
static inline bool efree_ic_small(int ic) { return ic == 16 || ic == 24 || ic == 40; }   // (efree.h: the tile / ring E-free kernels)
int launch_expand_stats_gram(const TfnasCellDesc& d, const float* x, double* stats1, float* part, hipStream_t s);
int launch_x_colsum(const float* x, int P, int ic, int rps, int nb, float* part, hipStream_t s);
// fx_kernels.hip: fused per-image route of the late cells (fx.h) -- E-free cells with 64 <= ic <= 192 and images <= 14 x 14
bool fx_supported(const TfnasCellDesc& d);
int launch_fx_stats(const TfnasCellDesc& d, const float* x, double* stats1, float* part, hipStream_t s);
// E != nullptr ("stored-ehat mode"): the forward also leaves ehat = BN1(x W1^T) in E and the backward reads it back instead of
// recomputing it; E == nullptr: nothing is stored, the backward rebuilds ehat from x
int launch_fx_fwd(const TfnasCellDesc& d, const float* x, const double* stats1, float* E, float* D, double* stats2,
                  float* part, hipStream_t s);
// backward: the partial sums of dE (rstd . W1) into scratch[0 .. nsl * P * ic) (nsl returned), the BN1-backward sums into
// red1 and the cb1 table; the caller finishes with launch_expand_gram + launch_expand_dgrad_x(scratch, nsl)
int launch_fx_bwd(const TfnasCellDesc& d, const float* x, const float* Eh, const double* stats1, const double* stats2,
                  const double* red2, const float* dZ, const float* D, const float* gate, const float* dpooled, float* scratch,
                  size_t scratch_floats, double* red1, float* cb1, float* part, int* nsl, hipStream_t s);
int launch_dw_bwd_data(const TfnasCellDesc& d, const float* dZ, const float* gate, const float* dpooled,
                       const float* D, const double* stats2,
                       const double* red2, const float* E, const float* x, const double* stats1, float* dEh,
                       double* red1, float* part, hipStream_t s, float* cb1 = nullptr, bool fuse_wgrad = false);
// true: launch_dw_bwd_data(..., fuse_wgrad = true) also writes the depthwise weight gradients g_dw (stride-1 ring cells: the WGR
// variant of k_dws_bwd) -- launch_dw_wgrad must then not be called for this cell
bool dw_bwd_fuses_wgrad(const TfnasCellDesc& d, const float* E);   // cb1: also fill the BN1 table
int launch_reduce_bn1(const TfnasCellDesc& d, const float* part, int nb, const double* stats1, double* red1, float* cb1,
                      hipStream_t s);
int launch_dw_wgrad(const TfnasCellDesc& d, const float* dZ, const float* gate, const float* dpooled, const float* D,
                    const double* stats2,
                    const double* red2, const float* E, const double* stats1, float* part, hipStream_t s);

// pointwise_kernels.hip (SE squeeze, BN2 backward statistics, mixing epilogue, BN constant tables)
// se_kernels.hip (SE excite FCs as small GEMMs: launch_se_fc_fwd / launch_se_fc_bwd / launch_se_wgrad)
int launch_se_pool(const TfnasCellDesc& d, const float* D, const double* stats2, float* pooled, hipStream_t s);
int launch_se_fc_fwd(const TfnasCellDesc& d, const float* pooled, float* hpre, float* gate, float* scratch,
                     size_t scratch_floats, hipStream_t s);
int launch_mix_fwd(const TfnasCellDesc& d, const float* Pr, const double* stats3, const float* wmix,
                   const float* x, float* out, hipStream_t s);
int launch_mix_bwd_stats(const TfnasCellDesc& d, const float* dout, const float* Pr, const double* stats3,
                         const float* x, double* red3, float* part, hipStream_t s);
int launch_mix_dw(const TfnasCellDesc& d, const double* red3, const double* resdot, float* dwmix, hipStream_t s);
int launch_se_bwd_reduce(const TfnasCellDesc& d, const float* dZ, const float* D, const double* stats2,
                         float* dgate, hipStream_t s);
int launch_se_fc_bwd(const TfnasCellDesc& d, const float* dgate, const float* gate, const float* hpre,
                     float* dgl, float* dhpre, float* dpooled, float* scratch, size_t scratch_floats, hipStream_t s);
int launch_se_wgrad(const TfnasCellDesc& d, const float* dgate, const float* gate, const float* dhpre,
                    const float* hpre, const float* pooled, hipStream_t s);
bool bn2_fused_fits(const TfnasCellDesc& d);
int launch_bn2_pool(const TfnasCellDesc& d, const float* dZ, const float* D, const double* stats2, float* dgate,
                    float* pp, hipStream_t s);
int launch_bn2_gather(const TfnasCellDesc& d, const float* rec, float* dgate, float* pp, hipStream_t s);
int launch_bn2_finish(const TfnasCellDesc& d, const float* pp, const float* gate, const float* dpooled, double* red2,
                      hipStream_t s);
int launch_bn2_bwd(const TfnasCellDesc& d, const float* dZ, const float* D, const double* stats2, const float* gate,
                   const float* dpooled, double* red2, float* part, hipStream_t s);
int launch_head_pool(const TfnasCellDesc& d, const float* E, const double* stats1, float* pooled, hipStream_t s);
int launch_head_bwd(const TfnasCellDesc& d, const float* E, const double* stats1, const float* dpooled, float* dEh,
                    double* red1, float* part, hipStream_t s);
// out[c] = sum_{b<nb} part[b*stride + c]  (double and/or float output); the deterministic replacement of atomics
// nbatch > 1: `nbatch` independent reductions in one launch (blockIdx.y): batch b reads part + b*in_stride and writes
// out + b*out_stride
int launch_reduce_rows(const float* part, int nb, int ncols, size_t stride, double* out_d, float* out_f,
                       hipStream_t s, int nbatch = 1, size_t in_stride = 0, size_t out_stride = 0);
/* One `part` scratch region = TFNAS_PART_ALLOC floats (16 MiB): TFNAS_PART_FLOATS for per-workgroup partial rows / split-K
   tiles; the last TFNAS_TAIL_SLOTS words are reserved (they held the ticket counters of the removed "last workgroup reduces"
   epilogues; the size of the region is part of the workspace ABI and stays). */
#define TFNAS_PART_ALLOC ((size_t)4 << 20)
#define TFNAS_TAIL_SLOTS ((size_t)1024)
#define TFNAS_PART_FLOATS (TFNAS_PART_ALLOC - TFNAS_TAIL_SLOTS)
int launch_bn1_consts(const TfnasCellDesc& d, const double* stats1, const double* red1, float* cb1,
                      hipStream_t s);

// arch_kernels.hip
int launch_arch_fwd(int ncell, const float* const* la, const float* e, const float* lat, float T, float* w,
                    float* cell_lat, hipStream_t s);
int launch_arch_bwd(int ncell, const float* w, const float* lat, const float* dw, const float* dcl, float T,
                    float* const* dla, hipStream_t s);
int launch_arch_project(int n, float* const* p, const int32_t* len, hipStream_t s);
int launch_arch_sample(int ncell, const float* const* la, const uint8_t* mask, const float* e, float T, int mode,
                       int32_t* pos, hipStream_t s);
int launch_sink_fwd(int K, const float* betas, const float* const* res, const float* cell_lat, uint64_t count,
                    float* out, float* out_lat, float* bw, hipStream_t s);
int launch_scale_copy(float* dst, const float* src, const float* scale, uint64_t count, hipStream_t s);
// dres[k] may be NULL (that depth output's share is added elsewhere: path level)
int launch_sink_bwd(int K, const float* bw, const float* const* res, const float* cell_lat, const float* dout,
                    const float* dlat, uint64_t count, float* const* dres, float* dbetas, float* dcell_lat,
                    double* dots, hipStream_t s);

// opt_kernels.hip
int launch_pack_ranges(const float* src, float* dst, int nranges, const uint64_t* off, const uint64_t* doff,
                       const uint64_t* len, hipStream_t s);
int launch_sgd_clip_step(float* w, float* g, float* m, int nranges, const uint64_t* off, const uint64_t* goff,
                         const uint64_t* len, float max_norm,
                         float lr, float momentum, float wd, float grad_scale, double* scratch, uint64_t scratch_doubles,
                         float* norm_out, hipStream_t s);
int launch_arch_adam_project(int n, float* const* p, const float* const* g, const int32_t* len, float* m, float* v,
                             float max_norm, float lr, float b1, float b2, float eps, float wd, int step, float grad_scale,
                             float* norm_out, hipStream_t s);

// bn_affine.hip (derived-network path: affine BatchNorm folded into the statistics tables, drop-connect)
int launch_bn_fwd_fix(double* stats, int nch, uint64_t cnt, float eps, const float* gamma, const float* beta, float* rmean,
                      float* rvar, float momentum, int eval, hipStream_t s, uint64_t world = 1);
int launch_bn_bwd_fix(double* red, int nch, uint64_t cnt, const float* gamma, const float* beta, float* dgamma, float* dbeta,
                      hipStream_t s);
int launch_rowscale(float* out, const float* y, const float* res, const float* scale, int N, uint64_t per_image,
                    hipStream_t s);
