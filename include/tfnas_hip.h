/*
 * tfnas_hip.h -- C ABI of the MI355X (gfx950) hot path of the TF-NAS supernet search step.
 *
 * The reference has NO native/FFI boundary for this path: its boundary is the Python
 * nn.Module API of models/model_search.py.  This header is therefore the boundary *we* define below that
 * API: the Python modules in tf-nas_amd/tfnas_amd/model_search.py (same class names, constructor and
 * forward(x, sampling, mode) signatures as the reference) call these entry points through ctypes with raw
 * device pointers.  Each entry point cites the reference code whose arithmetic it replaces.
 *
 * Conventions
 *  - all tensors are fp32, activations are NHWC
 *    ("channels_last"): x[n][h][w][c], c fastest;
 *  - the caller (PyTorch) owns and allocates every buffer, including saved-for-backward tensors and
 *    scratch; sizes come from tfnas_cell_ws();
 *  - `stream` is a hipStream_t passed as void*; all work is enqueued on it, nothing synchronises;
 *  - return value: 0 ok, <0 invalid argument (TFNAS_E*), >0 a hipError_t;
 *  - re-entrant per stream; every mode and every kernel-variant choice of a launch is given in its descriptor (gemm_mode, flags,
 *    route, sync_fn); the library reads no environment variable.  The only
 *    process-global state is the DEFAULTS of three modes (tfnas_set_gemm_mode / _lazy_join / _stats_sync), a registry of library-owned side streams (one per
 *    caller stream and device, created on first use by tfnas_mixedop_bwd with need_wgrad; released by
 *    tfnas_shutdown()) and the opt-in tfnas_prof_* timers; results never depend on either.
 *
 * A "cell" is one MixedOP (18 per network); its G "groups" are the MBConv candidates evaluated in this
 * call: G = 8 in the soft (alpha-step) mode, G = 1 in the sampled (w-step) mode.  All groups' expanded
 * channels live side by side in one [pixels][M] tensor (M = sum of padded mid widths) so that the 1x1
 * expand of all candidates is ONE GEMM and every per-channel pass (BN statistics, depthwise, SE) runs
 * once over the concatenation.
 */
#ifndef TFNAS_HIP_H
#define TFNAS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 4 (round 6): every route switch of a launch lives in its descriptor (TfnasCellDesc.route: TFNAS_ROUTE_*); the library reads no
 * environment variable at launch time (the TFNAS_* variables only seed the Python-side defaults, functions.HipModes.from_env).
 * tfnas_cell_route() + TfnasCellDesc.fwd_route: a backward refuses a descriptor whose route differs from the forward's.
 * TfnasCellDesc.wgrad_stream[3]: caller-owned streams for the three weight-gradient forks of ONE launch (the stem cell's backward
 * runs alone on the chip: its four weight-gradient kernels spread over the queues the two finished paths left idle).
 * tfnas_cls_ce_fwd_bwd / tfnas_cls_wgrad: classifier + cross-entropy of the weight step in two launches.
 * 3 (round 5): tfnas_fx_supported (fused per-image route of the late cells); per-launch modes in TfnasCellDesc (gemm_mode, flags,
 * sync_fn / sync_user / sync_world: the process-wide setters only provide defaults).
 * 2 (round 4): arithmetic modes of the GEMMs (tfnas_set_gemm_mode); the per-group input / output mode of TfnasCellDesc (xg /
 * og), TfnasPathDesc.dual, tfnas_path_set_side_stream2 and tfnas_path_defer_join / tfnas_path_join were measured and removed */
#define TFNAS_ABI_VERSION 4
#define TFNAS_MAX_GROUPS 8
#define TFNAS_MAX_SINK 4
#define TFNAS_MAX_CELLS 32

/* TfnasCellDesc.mode */
#define TFNAS_MODE_CELL 0   /* a MixedOP cell: 1x1 expand from an NHWC input (default)                          */
#define TFNAS_MODE_STEM 1   /* first_stem + second_stem (model_search.py:219-220) as ONE cell whose "expand" is the
                               3x3 stride-2 pad-1 convolution of the 3-channel NCHW image (x = image, ic = 27 =
                               im2col depth, w_expand = first_stem.conv.weight [32][3*3*3]); G = 1, no dx           */
#define TFNAS_MODE_HEAD 2   /* feature_mix_layer + global average pool (model_search.py:299-300): E = x W^T,
                               pooled = mean_hw act(BN(E)); only tfnas_head_fwd/bwd accept it                      */

#define TFNAS_ACT_RELU 0
#define TFNAS_ACT_SWISH 1

#define TFNAS_EINVAL (-1)   /* bad geometry / alignment (ic, oc must be multiples of 4) */
#define TFNAS_ENULL (-2)    /* required pointer is NULL */
#define TFNAS_ERANGE (-3)   /* count out of range */

/* One MBConv candidate (reference: MBInvertedResBlock, models/layers.py:431-561).  Weight pointers are
 * the reference's OIHW nn.Parameter storages used as-is:
 *   w_expand [mc][ic]      inverted_bottleneck.conv.weight  [mc,ic,1,1]
 *   w_dw     [mc][k*k]     depth_conv.conv.weight           [mc,1,k,k]
 *   w_proj   [oc][mc]      point_linear.conv.weight         [oc,mc,1,1]
 *   w_se_r   [se][mc], b_se_r [se]   squeeze_excite.conv_reduce.{weight,bias}
 *   w_se_e   [mc][se], b_se_e [mc]   squeeze_excite.conv_expand.{weight,bias}
 * g_* are the matching gradient outputs (same shapes), only read/written when need_wgrad != 0. */
typedef struct TfnasGroup {
    int32_t mc;       /* mid channels (any integer > 0, e.g. 53)            [in]  */
    int32_t k;        /* depthwise kernel size: 3 or 5                      [in]  */
    int32_t se;       /* squeeze-excite width (a multiple of 4), 0 = no SE   [in]  */
    int32_t mcp;      /* mc rounded up to a multiple of 4                   [plan] */
    int32_t off;      /* first column of this group in the [.][M] tensors (multiple of 32) [plan] */
    int32_t se_off;   /* first column in the [N][SE] hidden tensors         [plan] */
    int32_t pad0, pad1;
    const float *w_expand, *w_dw, *w_proj, *w_se_r, *b_se_r, *w_se_e, *b_se_e;
    float *g_expand, *g_dw, *g_proj, *g_se_r, *gb_se_r, *g_se_e, *gb_se_e;
} TfnasGroup;

typedef struct TfnasCellDesc {
    int32_t N, H, W;          /* input batch / height / width                       [in] */
    int32_t ic, oc;           /* cell in/out channels (multiples of 4)              [in] */
    int32_t stride;           /* 1 or 2 (depthwise stride; pad = k/2)               [in] */
    int32_t act;              /* TFNAS_ACT_*                                        [in] */
    int32_t has_res;          /* 1 when ic==oc && stride==1 (layers.py:537)         [in] */
    int32_t G;                /* number of groups, 1..8                             [in] */
    int32_t need_wgrad;       /* backward also produces weight gradients            [in] */
    int32_t Ho, Wo;           /* output height / width                              [plan] */
    int32_t M;                /* row length of the [.][M] tensors: groups padded to 32-float (128 B) boundaries [plan] */
    int32_t SE;               /* sum of se over groups                              [plan] */
    float eps;                /* BatchNorm eps (1e-5)                               [in] */
    int32_t mode;             /* TFNAS_MODE_CELL / _STEM / _HEAD                    [in] */
    int32_t Hi, Wi;           /* stem mode: height / width of the NCHW input image  [in] */
    int32_t stor;             /* must be 0 (fp32 storage of the [pixels][M] stream tensors E, D, dZ, dEh).  Rounds 1-3 had a
                                 second build that kept these four tensors in bf16 (stor = 1); it measured 0.99-1.04x of the
                                 fp32 iteration pair (the step is not bound by HBM bytes) and was removed.          [in] */
    int32_t gemm_mode;        /* 0: the process default (tfnas_set_gemm_mode); TFNAS_GEMM_EXPLICIT | TFNAS_GEMM_*: the arithmetic
                                 of THIS launch's 1x1 GEMMs, whatever the default is (two models in one process in different
                                 modes: an fp32-exact search net beside a bf16-GEMM derived net)                      [in] */
    int32_t flags;            /* TFNAS_CELL_LAZY_JOIN: tfnas_mbconv_bwd returns without joining its weight-gradient side
                                 stream (see tfnas_set_lazy_join, which sets the default for descriptors without the bit) [in]
                                 (bit 4 was TFNAS_CELL_FXP, the fused per-image project dgrad of round 5: measured equal to the
                                 default kernels over two rounds and deleted in round 6) */
    TfnasGroup g[TFNAS_MAX_GROUPS];
    /* per-launch cross-rank statistics hook (see tfnas_set_stats_sync: that one is the default for descriptors with
       sync_fn == NULL); sync_world >= 1 */
    int (*sync_fn)(void *user, double *table, uint64_t ndoubles, void *stream);
    void *sync_user;
    int32_t sync_world;
    int32_t route;            /* TFNAS_ROUTE_* bits: deviations of THIS launch from the library's measured per-launch policy
                                 (0 = the policy).  The library reads no environment variable at launch time: every kernel
                                 variant is selected here, per launch, so variants can be compared in one process           [in] */
    int32_t fwd_route;        /* backward only: the value tfnas_cell_route() returned for the descriptor the FORWARD of these
                                 buffers ran with (0 = not given: the backward trusts its own decision).  A backward whose own
                                 route differs in a way that changes what the saved buffers mean (the fused per-image route
                                 leaves ehat = BN1(E), not E, in the E buffer) returns TFNAS_EINVAL instead of silently
                                 normalising twice                                                                          [in] */
    int32_t pad_route;
    void *wgrad_stream[3];    /* optional caller-owned streams for the weight-gradient kernels of this launch: fork 0 = project,
                                 1 = squeeze-excite + depthwise, 2 = expand weight gradient.  NULL entries: the library-owned
                                 side stream of the caller's stream.  Every stream used is joined before tfnas_mixedop_bwd
                                 returns.  tfnas_mixedop_bwd only; with any entry set its `part` buffer must hold FOUR pieces
                                 of tfnas_sizeof(7) floats (chain | fork 0 | fork 1 | fork 2: concurrent forks do not share
                                 split-K scratch) instead of two                                                            [in] */
} TfnasCellDesc;
#define TFNAS_GEMM_EXPLICIT 0x1000
#define TFNAS_CELL_LAZY_JOIN 1
/* TfnasCellDesc.route (ABI 4; rounds 2-5 read these from TFNAS_* environment variables latched once per process) */
#define TFNAS_ROUTE_FX_OFF 0x1        /* frozen-weight launches of the 14 x 14 / 7 x 7 cells through the materialised route instead
                                         of the fused per-image kernels (csrc/fx_kernels.hip)                                   */
#define TFNAS_ROUTE_FOLD_OFF 0x2      /* per-image BN2-backward tables in their own pass over (dZ, D) (k_bn2_pool) instead of the
                                         epilogue of k_project_dgrad + k_bn2_gather                                             */
#define TFNAS_ROUTE_DWWG_OFF 0x4      /* 3x3 depthwise weight gradient of the stride-1 ring cells from its own kernel instead of
                                         the backward-data pass                                                                 */
#define TFNAS_ROUTE_DWWG2_OFF 0x8     /* the same for the register-window pass of the stride-2 cells                           */
#define TFNAS_ROUTE_XG_OFF 0x10       /* expand weight gradient never in Gram form                                              */
#define TFNAS_ROUTE_XG_ALL 0x20       /* ... in Gram form wherever the shape allows (default: where E >= 100 MB)                */
#define TFNAS_ROUTE_DW_SHIFT 6        /* 2 bits: 0 per-launch policy, 1 register-window kernels wherever the geometry allows,   */
#define TFNAS_ROUTE_DW_MASK 0xc0      /*         2 LDS ring / tile kernels only, 3 tile kernels only                            */
#define TFNAS_ROUTE_SE_SHIFT 8        /* 2 bits: 0 wave-level MFMA kernels for the excite FCs, 1 one fused per-image kernel,    */
#define TFNAS_ROUTE_SE_MASK 0x300     /*         2 LDS-tiled GEMMs                                                              */
#define TFNAS_ROUTE_WGRAD_INLINE 0x400 /* weight-gradient kernels on the caller's stream (no side stream)                       */
#define TFNAS_ROUTE_GRAM2 0x800       /* BN1-backward correction operator through the round-2 split-K GEMM + reduction instead
                                         of the one-launch k_gram1 (csrc/gemm_kernels.hip)                                      */
#define TFNAS_ROUTE_ALL 0xfff
/* tfnas_cell_route(): TFNAS_ROUTE_TAKEN_VALID | the routes a forward of the (planned) descriptor takes */
#define TFNAS_ROUTE_TAKEN_VALID 0x1
#define TFNAS_ROUTE_TAKEN_FX 0x2      /* fused per-image route: the E buffer holds ehat = BN1(x W1^T) after the forward         */

/* Element counts / offsets of every caller-allocated buffer of one cell. */
typedef struct TfnasCellWs {
    /* forward (saved for backward) */
    uint64_t E;        /* floats  [N*H*W][M]      raw 1x1-expand output (pre-BN)            */
    uint64_t D;        /* floats  [N*Ho*Wo][M]    raw depthwise output (pre-BN)             */
    uint64_t Pr;       /* floats  [G][N*Ho*Wo][oc] raw 1x1-project outputs (pre-BN)         */
    uint64_t fsmall;   /* floats  pooled[N][M] | gate[N][M] | hpre[N][SE]                   */
    uint64_t off_pooled, off_gate, off_hpre;
    uint64_t stats;    /* doubles stats1[M][2] | stats2[M][2] | stats3[G*oc][2]  (sum,sumsq) */
    uint64_t off_stats1, off_stats2, off_stats3;
    uint64_t out;      /* floats  [N*Ho*Wo][oc]                                             */
    /* backward scratch */
    uint64_t dZ;       /* floats  [N*Ho*Wo][M]                                              */
    uint64_t dEh;      /* floats  [N*H*W][M]                                                */
    uint64_t bsmall;   /* floats  dgate[N][M] | dpooled[N][M] | dgl[N][M] | dhpre[N][SE] | cb1[M][4] */
    uint64_t off_dgate, off_dpooled, off_dgl, off_dhpre, off_cb1;
    uint64_t red;      /* doubles red3[G*oc][2] | resdot[oc] | red2[M][2] | red1[M][2]      */
    uint64_t off_red3, off_red2, off_red1, off_resdot;   /* resdot = per-channel <dout, x> of residual cells */
    uint64_t part;     /* floats  scratch for per-workgroup partial sums (fwd and bwd), summed in a fixed order by a second
                          tiny kernel -- no floating-point atomics, deterministic results.  A multiple of tfnas_sizeof(7)
                          floats; the last tfnas_sizeof(8) 4-byte words of every tfnas_sizeof(7)-float piece are reserved.
                          Doubled when d.need_wgrad is set: tfnas_mixedop_bwd runs the weight-gradient kernels on
                          a library-owned side stream (forked from / joined to `stream` inside the call) and gives
                          them the second half.                                                              */
    uint64_t dx;       /* floats  [N*H*W][ic]                                               */
    uint64_t dxp;      /* floats  split-K partial tiles of the expand dgrad (may be tiny); pass NULL to disable */
} TfnasCellWs;

int tfnas_abi_version(void);

/* ---- cross-rank BatchNorm statistics ("sync-stats") -------------------------------------------------------------------------
 * Data-parallel runs normalise with PER-RANK batch statistics by default (what un-synced DDP / nn.DataParallel do; the
 * reference's search is effectively single-GPU).  With a hook installed every BatchNorm site uses the
 * statistics of the GLOBAL batch -- the analogue of apex's convert_syncbn_model in the reference's retrain script
 * (train_eval_amp.py:155-157) -- forward (sum, sum of squares per channel) and backward (the two BatchNorm-backward sums).
 * fn(user, table, ndoubles, stream) is called by every launch sequence right after a table of per-channel sums has been
 * reduced on `stream` and before any consumer of it is enqueued.  It must enqueue, ordered on `stream`, an all-reduce(SUM) of
 * table[0 .. ndoubles) over the ranks followed by a multiplication by 1 / world: the kernels keep dividing by the LOCAL element
 * count, so the scaled sums give exactly the global mean / variance / gradient means.  Returns 0 or an error code (propagated).
 * world: number of ranks (unbiased running-variance correction of the affine path).  fn = NULL: off (the default).
 * Process-global; E-free mode (statistics from the Gram matrix of x) is refused while a hook is installed. */
typedef int (*tfnas_stats_sync_fn)(void *user, double *table, uint64_t ndoubles, void *stream);
int tfnas_set_stats_sync(tfnas_stats_sync_fn fn, void *user, int world);

/* Destroys the library-owned side streams / events (after synchronising them).  Optional; safe to call more than once. */
int tfnas_shutdown(void);

/* sizeof() of the ABI structs, for binding self-checks: which = 0 TfnasGroup, 1 TfnasCellDesc, 2 TfnasCellWs,
 * 3 TfnasStage, 4 TfnasPathDesc, 5 TfnasPathWs, 6 TfnasBnAffine;  7 = floats of one `part` scratch piece, 8 = reserved
 * words at the end of each piece (TfnasCellWs.part). */
uint64_t tfnas_sizeof(int which);

/* Fill the [plan] fields of a descriptor from its [in] fields.  Returns TFNAS_E* on bad geometry. */
int tfnas_cell_plan(TfnasCellDesc *d);

/* Buffer sizes for a planned descriptor. */
int tfnas_cell_ws(const TfnasCellDesc *d, TfnasCellWs *ws);

/* 1 when the cell can run without the expanded tensor E ("E-free" mode: pass E = NULL to tfnas_mixedop_fwd AND to the
 * matching tfnas_mixedop_bwd; the depthwise kernels then recompute act(BN1(x W_expand^T)) from the narrow cell input,
 * and BN1's batch statistics come from the ic x ic Gram matrix of x).  Currently: TFNAS_MODE_CELL, need_wgrad = 0
 * (the alpha-step: frozen weights), ic in {16, 24, 40}.  Same arithmetic contract as the E path (fp32, <= 1e-3). */
int tfnas_efree_supported(const TfnasCellDesc *d);
/* What a forward of the (planned) descriptor does with the saved buffers -- a pure function of the descriptor (geometry,
 * need_wgrad, route, sync hook): TFNAS_ROUTE_TAKEN_VALID | TFNAS_ROUTE_TAKEN_*.  Callers that plan the backward with a descriptor
 * of their own (other need_wgrad, another sync hook) store it in that descriptor's fwd_route; the backward then refuses
 * (TFNAS_EINVAL) instead of misreading the E buffer.  The Python mirror does this for every launch (functions._cell_forward). */
int tfnas_cell_route(const TfnasCellDesc *d);
/* 1 when an E-free launch of the (planned) cell takes the FUSED PER-IMAGE route (csrc/fx_kernels.hip): stride 1, images of at
 * most 14 x 14 pixels, 64 <= ic <= 192 (a multiple of 16), frozen weights -- the supernet's cells at 14 x 14 and 7 x 7.  One kernel
 * per direction runs expand 1x1 + BN1 + activation + depthwise k x k (models/layers.py:542-552) for a group of whole images x a
 * slice of mid channels; neither E nor its gradient is ever written (the dEh buffer of tfnas_mixedop_bwd is used as scratch for
 * partial sums of dx).  Implies tfnas_efree_supported. */
int tfnas_fx_supported(const TfnasCellDesc *d);

/* MixedOP forward.
 *   soft mode  (G=8, wmix = device float[8] = gumbel-softmax weights):
 *       out = sum_g wmix[g] * (BN3(project_g(SE_g(act(BN2(dw_g(act(BN1(expand_g(x))))))))) [+ x])
 *       replaces MixedOP.forward soft branch, models/model_search.py:86-91 (line 89)
 *   sampled mode (G=1, wmix = NULL meaning weight 1):
 *       out = m_ops[idx](x);  replaces models/model_search.py:84-85 -> MBInvertedResBlock.forward layers.py:539-561
 * BN everywhere = batch statistics, biased variance, no affine (layers.py:469,498,533). */
int tfnas_mixedop_fwd(const TfnasCellDesc *d, const float *x, const float *wmix,
                      float *E, float *D, float *Pr, float *fsmall, double *stats, float *part,
                      float *out, void *stream);

/* MixedOP backward (what autograd does for the graph above).  Produces dx [N*H*W][ic], dwmix[G]
 * (d loss / d wmix[g]; may be NULL in sampled mode) and, when d->need_wgrad, the g_* weight gradients
 * (overwritten, not accumulated).  dZ/dEh/bsmall/red/part are scratch.  dx may be NULL when the input needs
 * no gradient: with need_wgrad == 0 only dwmix is produced (everything else is skipped). */
int tfnas_mixedop_bwd(const TfnasCellDesc *d, const float *x, const float *wmix,
                      const float *E, const float *D, const float *Pr, const float *fsmall,
                      const double *stats, const float *dout,
                      float *dZ, float *dEh, float *bsmall, double *red, float *part, /* then: dx, dxp, dwmix */
                      float *dx, float *dxp, float *dwmix, void *stream);

/* ---- derived-network ("retrain") path: one MBConv block with AFFINE BatchNorm + running statistics + drop-connect ------------
 * Reference: models/model_eval.py:31-244 (Network / NetworkCfg: a chain of MBInvertedResBlock(affine=True), layers.py:431-561),
 * tools/utils.py:77-86 (drop_connect), trained by train_eval.py.  Same kernels as the search cells (G must be 1); the affine
 * transform is folded into the per-channel statistics tables (csrc/bn_affine.hip).
 * BatchNorm site i: 0 = after the 1x1 expand (stem mode: after the 3x3 image conv; head: after the 1x1 feature-mix conv),
 * 1 = after the depthwise conv, 2 = after the 1x1 project.  Channels: mc, mc, oc. */
typedef struct TfnasBnAffine {
    const float *weight[3], *bias[3];          /* gamma / beta (device); NULL = that site has no affine part          */
    float *g_weight[3], *g_bias[3];            /* their gradients, written by the backward when non-NULL               */
    float *running_mean[3], *running_var[3];   /* training: updated with `momentum` (unbiased variance, torch semantics);
                                                  eval: read instead of the batch statistics; NULL: mean 0 / var 1 / no update */
    float momentum;                            /* torch default 0.1                                                    */
    int32_t eval;                              /* 1: normalise with the running statistics (model.eval())              */
} TfnasBnAffine;

/* out = [drop_scale[n] *] BN3(project(SE(act(BN2(dw(act(BN1(expand(x)))))))) [+ x]   with affine BatchNorms.
 * drop_scale: device float[N] = floor(keep + U[0,1)) / keep per image (residual blocks in training) or NULL.
 * Buffers as tfnas_mixedop_fwd (sizes from tfnas_cell_ws); d->G must be 1, wmix is not used. */
int tfnas_mbconv_fwd(const TfnasCellDesc *d, const TfnasBnAffine *bn, const float *drop_scale, const float *x,
                     float *E, float *D, float *Pr, float *fsmall, double *stats, float *part, float *out, void *stream);

/* Backward of tfnas_mbconv_fwd: dx, the block's weight gradients (d->need_wgrad, g_* of the group) and the BatchNorm
 * parameter gradients (bn->g_weight / g_bias).  dout_s: scratch [N*Ho*Wo][oc] floats, only needed with drop_scale.
 * In eval mode (bn->eval) the backward treats the statistics as constants (no batch-statistics terms). */
int tfnas_mbconv_bwd(const TfnasCellDesc *d, const TfnasBnAffine *bn, const float *drop_scale, const float *x,
                     const float *E, const float *D, const float *Pr, const float *fsmall, const double *stats,
                     const float *dout, float *dout_s, float *dZ, float *dEh, float *bsmall, double *red, float *part,
                     float *dx, float *dxp, void *stream);

/* Head with affine BatchNorm (feature_mix_layer of model_eval.py:98 + global pool): site 0 of `bn`. */
int tfnas_head_affine_fwd(const TfnasCellDesc *d, const TfnasBnAffine *bn, const float *x, float *E, double *stats,
                          float *part, float *pooled, void *stream);
int tfnas_head_affine_bwd(const TfnasCellDesc *d, const TfnasBnAffine *bn, const float *x, const float *E,
                          const double *stats, const float *dpooled, float *dEh, float *cb1, double *red, float *part,
                          float *dx, float *dxp, void *stream);

/* Network head (mode TFNAS_MODE_HEAD): pooled[N][mc] = mean over pixels of act(BN(x W_expand^T)).
 * Replaces feature_mix_layer (ConvLayer 1x1 + BN + swish) + AdaptiveAvgPool2d(1), models/model_search.py:299-300.
 * Buffers: E [N*H*W][M], stats doubles [M][2], part = scratch (ws.part floats). */
int tfnas_head_fwd(const TfnasCellDesc *d, const float *x, float *E, double *stats, float *part, float *pooled,
                   void *stream);
/* Backward of tfnas_head_fwd: dx [N*H*W][ic] and (need_wgrad) g_expand.  dEh [N*H*W][M], cb1 [M][4] floats and
 * red doubles [M][2] are scratch. */
int tfnas_head_bwd(const TfnasCellDesc *d, const float *x, const float *E, const double *stats, const float *dpooled,
                   float *dEh, float *cb1, double *red, float *part, float *dx, float *dxp, void *stream);

/* The head's weight gradient alone (what tfnas_head_bwd does last when need_wgrad is set): g_expand of the group from dEh / E / cb1
 * as tfnas_head_bwd left them.  A leaf of the backward: the weight step calls tfnas_head_bwd with need_wgrad = 0 on a path's stream
 * and this on that path's weight-gradient stream, so that the cells' backward does not queue up behind it.  part: scratch of its
 * own (tfnas_cell_ws().part floats). */
int tfnas_head_wgrad(const TfnasCellDesc *d, const float *x, const float *E, const float *dEh, const float *cb1, float *part,
                     void *stream);

/* ---- classifier + cross-entropy tail (models/model_search.py:301-303 `self.classifier(x)`; train_search.py:107 nn.CrossEntropyLoss
 * as called at :333, :376-379, :410, and their autograd) ---------------------------------------------------------------------------
 * tfnas_cls_ce, one launch, everything that depends on ONE image:
 *   logits[n][k] = bias[k] + sum_c pooled[n][c] W[k][c];  loss_n[n] = logsumexp_k logits[n] - logits[n][target[n]]
 *   dlogits[n][k] = scale * (softmax(logits[n])[k] - [k == target[n]])   (scale = 1 / N: the mean reduction of CrossEntropyLoss)
 *   dpooled[n][c] = sum_k dlogits[n][k] W[k][c]
 * pooled [N][C] (C a multiple of 4), W [K][C], bias [K] or NULL, target int64 [N].
 * tfnas_cls_wgrad, one launch, everything that sums over images -- and over the npath <= 2 bi-sampling paths of a weight step:
 *   dW[k][c] = sum_p sum_n dlogits_p[n][k] pooled_p[n][c];  db[k] = sum_p sum_n dlogits_p[n][k];  loss = loss_scale * sum_p sum_n loss_n_p[n]
 * (overwritten, not accumulated; loss may be NULL).  Fixed summation orders, no atomics. */
int tfnas_cls_ce(int N, int C, int K, const float *pooled, const float *W, const float *bias, const int64_t *target, float scale,
                 float *logits, float *loss_n, float *dlogits, float *dpooled, void *stream);
int tfnas_cls_wgrad(int npath, int N, int C, int K, const float *const *pooled, const float *const *dlogits,
                    const float *const *loss_n, float loss_scale, float *dW, float *db, float *loss, void *stream);
/* dst[i] += src[i], count floats (a multiple of 4): the second path's share of a shared parameter's gradient. */
int tfnas_add_into(float *dst, const float *src, uint64_t count, void *stream);

/* Gumbel-softmax over the candidates of `ncell` cells in one launch + expected cell latency.
 *   w[c][i] = softmax_i((log_alpha[c][i] - log(e[c][i])) / T)      (F.gumbel_softmax, model_search.py:87)
 *   cell_lat[c] = sum_i w[c][i] * lat[c][i]                          (model_search.py:90)
 * log_alpha: `ncell` device pointers (each float[8]) -- the reference keeps one nn.Parameter per cell.
 * e: Exp(1) draws, lat: looked-up latencies (model_search.py:93-111), both device float[ncell][8]. */
int tfnas_arch_fwd(int ncell, const float *const *log_alpha, const float *e, const float *lat, float T,
                   float *w, float *cell_lat, void *stream);

/* Backward of tfnas_arch_fwd: dla[c][j] = (1/T) w_j (gt_j - sum_i gt_i w_i), gt_i = dw[c][i] + dlat[c]*lat[c][i].
 * dlog_alpha: `ncell` device pointers receiving float[8] each. */
int tfnas_arch_bwd(int ncell, const float *w, const float *lat, const float *dw, const float *dcell_lat,
                   float T, float *const *dlog_alpha, void *stream);

/* Sampled-mode index selection for `ncell` cells (model_search.py:59-81):
 *   mode 0 'gumbel'/'gumbel_2': pos = argmax gumbel_softmax(log_softmax(log_alpha[mask]), T)
 *   mode 1 'min_alphas', mode 2 'max_alphas': argmin / argmax of log_alpha[mask]
 * mask: device uint8[ncell][8] (the `switches`); pos_out: device int32[ncell] = position among the
 * switched-on candidates (the caller maps it like fink_ori_idx, model_search.py:49-56). */
int tfnas_arch_sample(int ncell, const float *const *log_alpha, const uint8_t *mask, const float *e, float T,
                      int mode, int32_t *pos_out, void *stream);

/* Projection of the architecture parameters after the Adam step (train_search.py:421-422):
 *   p <- log_softmax(p) in place, for n 1-D fp32 device tensors of len[i] <= 8 elements each (log_alphas and betas),
 * one launch for all of them.  p: n device pointers (host array), len: host int32[n]. */
int tfnas_arch_project(int n, float *const *p, const int32_t *len, void *stream);

/* Sink-connecting stage output (MixedStage.forward tail, models/model_search.py:202-204):
 *   bw = softmax(betas[K]);  out = sum_k bw[k]*res[k];  out_lat = sum_k bw[k]*(cell_lat[0]+..+cell_lat[k])
 * res: K device pointers to [count] floats each; cell_lat: device float[K] or NULL (sampled mode: lat 0).
 * bw_out: device float[K] (saved for backward). */
int tfnas_sink_fwd(int K, const float *betas, const float *const *res, const float *cell_lat,
                   uint64_t count, float *out, float *out_lat, float *bw_out, void *stream);

/* Backward of tfnas_sink_fwd: dres[k] = bw[k]*dout; dbetas via the softmax Jacobian of
 * (<dout,res[k]> + dlat*cum_k); dcell_lat[j] = dlat * sum_{k>=j} bw[k].  dot_scratch: device double[K]. */
int tfnas_sink_bwd(int K, const float *bw, const float *const *res, const float *cell_lat, const float *dout,
                   const float *dlat, uint64_t count, float *const *dres, float *dbetas, float *dcell_lat,
                   double *dot_scratch, void *stream);

/* ================================================================================================================
 * Path level: a whole chain of MixedOP cells + the sink-connecting stage mixes of a Network in ONE call per direction.
 *
 * Replaces the body of Network.forward between the stems and the head (models/model_search.py:285-297: six
 * MixedStage.forward calls, :157-206) and its autograd backward, for one "path":
 *   sampled mode  (every cell G = 1)  -- one of the two bi-sampling paths of train_w_arch's weight step
 *                                        (train_search.py:375-379) or train_wo_arch's / validate's single path;
 *   soft mode     (every cell G = 8)  -- the architecture step (train_search.py:409-413).
 * The per-cell entry points above stay (they are what MixedOP.forward calls when a user drives the modules one by one);
 * the path entry points exist because the per-cell route costs one Python/autograd round trip, ~10 tensor allocations
 * and ~25 launches *per cell*, and the weight step then spends a third of its wall time waiting for the host.
 * Differences to 18 x tfnas_mixedop_fwd/bwd + 6 x tfnas_sink_fwd/bwd (same kernels otherwise, bit-identical results):
 *   - all buffers come from ONE caller-allocated arena (sizes from tfnas_path_plan);
 *   - the sink gradient bw[k]*dsink is added in the dx epilogue of the following cell instead of K scaled copies per
 *     stage and one elementwise add per block;
 *   - weight-gradient kernels run on a side stream and may lag ONE cell behind the data-gradient chain (the per-cell
 *     entry joins at the end of every cell);
 *   - several paths (the two bi-sampling paths) are enqueued interleaved, cell by cell, on their own streams.
 * ================================================================================================================ */
#define TFNAS_MAX_STAGES 8

typedef struct TfnasStage {
    int32_t ncell;          /* MixedOP blocks of this stage (model_search.py:126-155: 2,3,4,4,4,1)            [in] */
    int32_t start_res;      /* 0: the stage input is itself a depth choice (ic==oc && stride==1), else 1      [in] */
    int32_t first_cell;     /* index of the stage's first cell in TfnasPathDesc.cell                          [plan] */
    int32_t nres;           /* ncell + 1 - start_res = len(betas)                                            [plan] */
    const float *betas;     /* device float[nres]                                                            [in] */
    float *dbetas;          /* device float[nres] or NULL (architecture parameters frozen: weight step)       [in] */
} TfnasStage;

typedef struct TfnasPathDesc {
    int32_t ncell, nstage;
    int32_t soft;           /* 1: cells carry G = 8 groups, wmix/cell_lat are consumed and d wmix / d cell_lat produced */
    int32_t need_dx0;       /* backward produces the gradient of the path input                               */
    int32_t efree_mask_lo;  /* bit c set: cell c runs E-free (E never materialised; needs tfnas_efree_supported) */
    int32_t reserved0;      /* must be 0 (ABI 1: dual) */
    TfnasStage stage[TFNAS_MAX_STAGES];
    TfnasCellDesc cell[TFNAS_MAX_CELLS];   /* [in] fields + weight / gradient pointers bound; N, H, W chained by plan */
} TfnasPathDesc;

/* Arena requirement of a planned path, in floats (the arena must be 256-byte aligned). */
typedef struct TfnasPathWs {
    uint64_t saved;         /* forward results kept for backward (E, D, Pr, small tensors, statistics, cell / stage outputs) */
    uint64_t scratch;       /* forward + backward scratch (partials, dZ, dEh, gradient ring)                  */
    uint64_t total;         /* saved + scratch                                                                */
    uint64_t out_count;     /* elements of the path output [N][Ho][Wo][oc] of the last stage */
    int32_t out_h, out_w, out_c, pad;
} TfnasPathWs;

/* Opaque path context: the planned descriptor, arena offsets, one library-owned side stream and the events that order
 * it against the caller's stream.  One context per concurrently running path. */
int tfnas_path_create(void **ctx);
int tfnas_path_destroy(void *ctx);
/* Use `stream` (owned by the caller, must outlive the context) for the weight-gradient kernels instead of a stream the library
 * creates.  HIP maps streams onto a few hardware queues in creation order; a caller that has measured which of its streams
 * really run concurrently (tfnas_amd/streams.py) hands the good ones in here. */
int tfnas_path_set_side_stream(void *ctx, void *stream);
/* Validate + plan every cell (tfnas_cell_plan), chain the geometry (cell i+1's input extent = cell i's output), lay out
 * the arena.  May be called again on the same context with different candidates / widths (every weight step does). */
int tfnas_path_plan(void *ctx, const TfnasPathDesc *pd, TfnasPathWs *ws);

/* Forward of `npath` planned paths, enqueued interleaved cell by cell: path p runs on streams[p] with arena[p].
 *   x0[p]     device [N][H][W][ic0] input of the first cell (second_stem output)
 *   wmix[p]   soft mode: device float[ncell][8] gumbel-softmax weights (tfnas_arch_fwd); NULL in sampled mode
 *   cell_lat[p] soft mode: device float[ncell] expected cell latencies; NULL otherwise
 *   out[p]    device [out_count] = last stage's sink output;  out_lat[p]: device float[nstage] per-stage expected latency
 *             (soft mode; NULL otherwise). */
int tfnas_paths_fwd(int npath, void *const *ctx, const float *const *x0, const float *const *wmix,
                    const float *const *cell_lat, float *const *arena, float *const *out, float *const *out_lat,
                    void *const *streams);

/* Backward of the same.  dout[p]: gradient of out[p];  dout_lat[p]: device float[nstage] or NULL;
 * produces dx0[p] (if need_dx0), dwmix[p] float[ncell][8] and dcell_lat[p] float[ncell] (soft mode), stage dbetas, and the
 * cells' weight gradients at the g_* pointers of the planned descriptors (need_wgrad cells).
 * On return every path's side stream has been joined to its stream.
 * stage_begin / stage_end: walk only the stages [stage_begin, stage_end) (in reverse order; stage_end = -1: to the last one).
 * A backward may be issued as consecutive segments, last stages first -- (k, -1) then (0, k) -- so that the caller can start
 * reducing the late stages' weight gradients (89 % of the parameters) across ranks while the early stages are still running. */
int tfnas_paths_bwd(int npath, void *const *ctx, const float *const *x0, const float *const *wmix,
                    const float *const *cell_lat, float *const *arena, const float *const *dout,
                    const float *const *dout_lat, float *const *dx0, float *const *dwmix, float *const *dcell_lat,
                    void *const *streams, int stage_begin, int stage_end);

/* ---- fused optimizer steps ------------------------------------------------------------------
 * Weight step tail, train_search.py:381-385: clip_grad_norm_(weight_parameters, max_norm) then SGD(momentum, weight decay,
 * dampening 0) on the parameters that received a gradient.  w / g / m: flat fp32 arenas with identical layout (weights,
 * gradients, momentum); (off[i], len[i]) i < nranges <= 64: the float ranges to update (multiples of 4; typically one per
 * sampled candidate).  total = ||g over all ranges|| * grad_scale (grad_scale = 1 / world_size after a SUM all-reduce);
 * g <- g * grad_scale * min(1, max_norm / (total + 1e-6));  m <- momentum*m + (g + wd*w);  w <- w - lr*m.
 * max_norm <= 0 disables clipping.  scratch: device doubles, >= sum ceil(len/8192).  norm_out: device float or NULL.
 * goff: NULL (gradients at the same offsets as the weights) or the offsets of the ranges inside `g` when `g` is the packed
 * all-reduce message built by tfnas_pack_ranges.  Two launches, no atomics, deterministic. */
int tfnas_sgd_clip_step(float *w, float *g, float *m, int nranges, const uint64_t *off, const uint64_t *goff, const uint64_t *len,
                        float max_norm, float lr, float momentum, float wd, float grad_scale, double *scratch,
                        uint64_t scratch_doubles, float *norm_out, void *stream);

/* dst[doff[i] .. +len[i]) = src[off[i] .. +len[i]) for nranges <= 64 ranges (one launch): gathers the sampled candidates'
 * gradient ranges into ONE contiguous buffer = one RCCL all-reduce message (no torch.cat, no copy back: the SGD step reads
 * the reduced message through `goff`). */
int tfnas_pack_ranges(const float *src, float *dst, int nranges, const uint64_t *off, const uint64_t *doff,
                      const uint64_t *len, void *stream);

/* Architecture step tail, train_search.py:414-422: clip_grad_norm_(arch_parameters, max_norm), Adam (torch semantics:
 * weight decay added to the gradient, bias correction with `step` >= 1, eps outside the square root) and the projection
 * p <- log_softmax(p), for n <= 32 parameters of len[i] <= 8 floats, in ONE launch.  p / g: host arrays of n device pointers;
 * m / v: device float[n][8] Adam moments (row i, first len[i] entries). */
int tfnas_arch_adam_project(int n, float *const *p, const float *const *g, const int32_t *len, float *m, float *v,
                            float max_norm, float lr, float beta1, float beta2, float eps, float wd, int step,
                            float grad_scale, float *norm_out, void *stream);

/* ---- lazy join of the weight-gradient side stream (derived-network training step) ------------------------------------------
 * tfnas_mixedop_bwd / tfnas_mbconv_bwd run their weight-gradient kernels on a library-owned side stream per caller stream and
 * join it before they return.  With tfnas_set_lazy_join(1), tfnas_mbconv_bwd returns WITHOUT joining: the weight gradients of a
 * block then overlap with the data-gradient chain of the blocks below it.  The caller must (1) keep every buffer passed to the
 * call alive until the side stream is done with it (PyTorch: tensor.record_stream on tfnas_side_stream(stream)), and
 * (2) call tfnas_side_join(stream) before anything on `stream` reads the weight gradients (the optimizer step).
 * (train_eval.py:228-252 has no such notion: one stream; this is scheduling only, the results are bit-identical.) */
int tfnas_set_lazy_join(int on);
int tfnas_side_stream(void *stream, void **side);   /* *side = the library-owned side stream paired with `stream` (created on first use) */
int tfnas_side_join(void *stream);

/* ---- arithmetic of the 1x1-convolution GEMMs --------------------------------------------------------------------------------
 * The reference runs its pointwise convolutions in fp32 (models/layers.py:463-478, 528-534; the search is never AMP).  gfx950
 * executes fp32 MFMA at 1/16 of its bf16 MFMA rate, so the row-tiled GEMMs (expand / project forward and data gradients) split
 * every fp32 operand EXACTLY into three bf16 planes and accumulate the six products that matter in fp32 (csrc/gemm_x3.h):
 *   TFNAS_GEMM_X3   (6, default)  six bf16 MFMAs per element pair, error <= 2^-23 |a||b| per product: fp32-level accuracy
 *   TFNAS_GEMM_F32  (0)           v_mfma_f32_16x16x4_f32, bit-identical to an fmaf chain
 *   TFNAS_GEMM_X2   (3)           three products (error ~2^-16)
 *   TFNAS_GEMM_BF16 (1)           plain bf16 operands, fp32 accumulation (the reduced-precision mode of the derived network's
 *                                 training step; train_eval_amp.py:176-180 in the reference)
 * Process-wide default (TFNAS_GEMM_X3 until set; the library reads no environment variable -- the Python mirror passes TFNAS_GEMM =
 * x3 | f32 | x2 | bf16 on to this call when it loads the library); a descriptor's own gemm_mode wins.  Returns TFNAS_EINVAL for
 * any other mode.  tfnas_gemm_mode() returns the current one. */
#define TFNAS_GEMM_F32 0
#define TFNAS_GEMM_BF16 1
#define TFNAS_GEMM_X2 3
#define TFNAS_GEMM_X3 6
/* OR-ed into the mode: every row-tiled GEMM launch uses it.  Without the flag the library keeps the fp32 loop where the split
 * loop measured slower (the data-gradient GEMMs of one-candidate launches and of the large images). */
#define TFNAS_GEMM_EVERYWHERE 0x100
int tfnas_set_gemm_mode(int mode);
int tfnas_gemm_mode(void);

/* ---- optional diagnostics (used by bench.py for the `roofline` object) --------------------------------------
 * Per-kernel-family timing with HIP events recorded on the launch stream.  tfnas_prof_enable(mask) turns the
 * families whose bit is set on (0 = off, the default); tfnas_prof_collect() waits for the recorded events of
 * one family, returns their count and summed duration in ms, and clears them.  Not re-entrant. */
int tfnas_prof_enable(unsigned mask);
int tfnas_prof_count(void);
const char *tfnas_prof_name(int id);
int tfnas_prof_collect(int id, uint64_t *launches, double *total_ms);
/* of the launches the LAST tfnas_prof_collect(id) returned: those that covered all candidates of a cell (alpha-step launches of
 * the depthwise families), count and summed ms -- the same kernels run at two very different sizes in the two step kinds */
int tfnas_prof_last_split(int id, uint64_t *soft_launches, double *soft_ms);

#ifdef __cplusplus
}
#endif
#endif /* TFNAS_HIP_H */
