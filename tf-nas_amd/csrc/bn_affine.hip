// Affine BatchNorm with running statistics, train / eval mode, and drop-connect for the derived-network ("retrain") path
// (reference: models/layers.py:462-534 with affine=True, models/model_eval.py, tools/utils.py:77-86).
//
// The cell kernels normalise with per-channel (mean, rstd) tables and apply BatchNorm-backward with per-channel
// (mean, rstd, S1/P, S2/P) tables; none of them knows about gamma / beta.  Affine BatchNorm is folded into those TABLES:
//   forward   y = gamma*(x-mu)*rho + beta = (x - m_eff) * r_eff,   r_eff = gamma*rho,  m_eff = mu - beta/r_eff
//             -> k_bn_fwd_fix rewrites the (sum, sumsq) table of a producer as (m_eff, r_eff); consumers run with eps = -1
//                (tfnas_dev.h: bn_consts).  In eval mode mu / rho come from the running statistics instead of the batch.
//   backward  the kernels see y as "the normalised value": they form d = upstream*act'(y) (correct: act is applied to y) and the
//             sums S1 = sum d, S2y = sum d*y, then apply  r * (d - t1 - y*t2).  The true gradient is
//                 gamma*rho * (d - mean(d) - xhat*mean(d*xhat)),   xhat = (y - beta)/gamma
//               = r_eff * (d - (t1 - beta*c) - y*c),   t1 = S1/P,  c = (S2y/P - beta*t1) / gamma^2
//             -> k_bn_bwd_fix rewrites the reduced (S1, S2y) table as P*(t1 - beta*c, c) and emits
//                d gamma = (S2y - beta*S1)/gamma,  d beta = S1.
// gamma == 0 cannot be represented (y would be the constant beta); the fix kernels clamp |gamma| to 1e-12.
#include "tfnas_dev.h"
#include "kernels.h"
#include "prof.h"

__global__ void k_bn_fwd_fix(double* __restrict__ stats, int nch, double inv_cnt, double unbias, float eps,
                             const float* __restrict__ gamma, const float* __restrict__ beta,
                             float* __restrict__ rmean, float* __restrict__ rvar, float momentum, int eval) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= nch) return;
    double mu, var;
    if (eval) {
        mu = rmean ? (double)rmean[c] : 0.0;
        var = rvar ? (double)rvar[c] : 1.0;
    } else {
        mu = stats[2 * c] * inv_cnt;
        var = stats[2 * c + 1] * inv_cnt - mu * mu;
        if (var < 0.0) var = 0.0;
        if (rmean) rmean[c] = (1.f - momentum) * rmean[c] + momentum * (float)mu;
        if (rvar) rvar[c] = (1.f - momentum) * rvar[c] + momentum * (float)(var * unbias);     // unbiased, like torch
    }
    const double rho = 1.0 / sqrt(var + (double)eps);
    double g = gamma ? (double)gamma[c] : 1.0;
    const double b = beta ? (double)beta[c] : 0.0;
    if (fabs(g) < 1e-12) g = g < 0.0 ? -1e-12 : 1e-12;
    const double r = g * rho;
    stats[2 * c] = mu - b / r;
    stats[2 * c + 1] = r;
}

__global__ void k_bn_bwd_fix(double* __restrict__ red, int nch, double inv_cnt, const float* __restrict__ gamma,
                             const float* __restrict__ beta, float* __restrict__ dgamma, float* __restrict__ dbeta) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= nch) return;
    const double S1 = red[2 * c], S2 = red[2 * c + 1];
    double g = gamma ? (double)gamma[c] : 1.0;
    const double b = beta ? (double)beta[c] : 0.0;
    if (fabs(g) < 1e-12) g = g < 0.0 ? -1e-12 : 1e-12;
    if (dbeta) dbeta[c] = (float)S1;
    if (dgamma) dgamma[c] = (float)((S2 - b * S1) / g);
    const double t1 = S1 * inv_cnt;
    const double cc = (S2 * inv_cnt - b * t1) / (g * g);
    red[2 * c] = (t1 - b * cc) / inv_cnt;
    red[2 * c + 1] = cc / inv_cnt;
}

// out[n][..] = scale[n] * y[n][..] (+ res[n][..])        (drop-connect, tools/utils.py:77-86: x.div(keep) * floor(keep + U))
__global__ __launch_bounds__(256) void k_rowscale(float* __restrict__ out, const float* __restrict__ y,
                                                  const float* __restrict__ res, const float* __restrict__ scale,
                                                  uint64_t per_image4, uint64_t total4) {
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < total4; i += (uint64_t)gridDim.x * 256) {
        const float sc = scale[i / per_image4];
        f32x4 v = splat4(sc) * ld4(y + 4 * i);
        if (res) v += ld4(res + 4 * i);
        st4(out + 4 * i, v);
    }
}

int launch_bn_fwd_fix(double* stats, int nch, uint64_t cnt, float eps, const float* gamma, const float* beta, float* rmean,
                      float* rvar, float momentum, int eval, hipStream_t s, uint64_t world) {
    if (nch <= 0) return 0;
    ProfScope _prof(TK_SMALL, s);
    const uint64_t gcnt = cnt * world;                  // sync-stats: the statistics are those of the global batch
    const double unbias = gcnt > 1 ? (double)gcnt / (double)(gcnt - 1) : 1.0;
    hipLaunchKernelGGL(k_bn_fwd_fix, dim3(cdiv(nch, 256)), dim3(256), 0, s, stats, nch, 1.0 / (double)cnt, unbias, eps, gamma,
                       beta, rmean, rvar, momentum, eval);
    return (int)hipGetLastError();
}

int launch_bn_bwd_fix(double* red, int nch, uint64_t cnt, const float* gamma, const float* beta, float* dgamma, float* dbeta,
                      hipStream_t s) {
    if (nch <= 0) return 0;
    ProfScope _prof(TK_SMALL, s);
    hipLaunchKernelGGL(k_bn_bwd_fix, dim3(cdiv(nch, 256)), dim3(256), 0, s, red, nch, 1.0 / (double)cnt, gamma, beta, dgamma,
                       dbeta);
    return (int)hipGetLastError();
}

int launch_rowscale(float* out, const float* y, const float* res, const float* scale, int N, uint64_t per_image,
                    hipStream_t s) {
    if (per_image & 3) return TFNAS_EINVAL;
    ProfScope _prof(TK_SMALL, s);
    const uint64_t total4 = (uint64_t)N * per_image / 4;
    uint64_t blocks = (total4 + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(k_rowscale, dim3((unsigned)blocks), dim3(256), 0, s, out, y, res, scale, per_image / 4, total4);
    return (int)hipGetLastError();
}
