// BN1 batch statistics of the expanded activation WITHOUT forming it (E-free mode, see efree.h):
//   E[p][m] = sum_c x[p][c] w[m][c]   =>   sum_p E[p][m] = w_m . sx ,   sum_p (E[p][m] - mean_m)^2 = w_m^T C w_m
// with sx = sum_p x[p] and C = sum_p (x[p] - xbar)(x[p] - xbar)^T (ic x ic, centred: no mean^2 cancellation at all).
// Two passes over the narrow input (ic = 16..40 channels; ~1/30 of the bytes of E), partial rows folded in double by
// k_reduce_rows; the quadratic form is evaluated per channel in double.
// Reference arithmetic: nn.BatchNorm2d(affine=False) batch statistics of inverted_bottleneck's conv output
// (models/layers.py:470-482 of the reference).
#include "tfnas_dev.h"
#include "kernels.h"
#include "prof.h"
#include "efree.h"

// part[b][c] = sum of x[p][c] over the workgroup's rows.  x is read as a flat float4 stream; a thread's elements are
// `stride` float4 apart with stride a multiple of ic/4, so its channel quad never changes.
__global__ __launch_bounds__(256) void k_x_colsum(const float* __restrict__ x, int P, int ic, int rps,
                                                  float* __restrict__ part) {
    __shared__ f32x4 red[256];
    const int tid = threadIdx.x, icq = ic >> 2, stride = (256 / icq) * icq;
    const int r0 = blockIdx.x * rps, r1 = min(P, r0 + rps);
    const f32x4* __restrict__ x4 = reinterpret_cast<const f32x4*>(x) + (size_t)r0 * icq;
    const int n4 = (r1 - r0) * icq;
    f32x4 s = zero4();
    if (tid < stride) {
        int i = tid;
        for (; i + 7 * stride < n4; i += 8 * stride) {
            f32x4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = x4[i + u * stride];
            s += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
        }
        for (; i < n4; i += stride) s += x4[i];
    }
    red[tid] = s;
    __syncthreads();
    if (tid < ic) {
        const int cq = tid >> 2, comp = tid & 3;
        float t = 0.f;
        for (int k = cq; k < stride; k += icq) t += red[k][comp];
        part[(size_t)blockIdx.x * ic + tid] = t;
    }
}

// part[b][i*ic + j] = sum over the workgroup's rows of (x[p][i] - xbar_i)(x[p][j] - xbar_j): a [ic x rows] x [rows x ic]
// product on the matrix cores.  Both MFMA operands of a 16x16 tile pair are the same centred values (lane = (channel
// l%16 of tile t, row l/16)), four rows per instruction; CT = ceil(ic/16) channel tiles, padded with zeros.
template <int CT>
__global__ __launch_bounds__(256) void k_x_gram(const float* __restrict__ x, int P, int ic, int rps,
                                                const double* __restrict__ xsum, float* __restrict__ part) {
    __shared__ float red[4][CT * CT][4][64 + 1];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n = lane & 15, q = lane >> 4;
    const int r0 = blockIdx.x * rps, r1 = min(P, r0 + rps);
    float mu[CT];
    bool chok[CT];
#pragma unroll
    for (int t = 0; t < CT; ++t) {
        chok[t] = 16 * t + n < ic;
        mu[t] = chok[t] ? (float)(xsum[16 * t + n] / (double)P) : 0.f;
    }
    f32x4 acc[CT][CT];
#pragma unroll
    for (int i = 0; i < CT; ++i)
#pragma unroll
        for (int j = 0; j < CT; ++j) acc[i][j] = zero4();
    for (int p = r0 + 4 * wave; p < r1; p += 64) {                  // 4 rounds of 16 rows (4 per wave) per iteration
        float a[4][CT];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int row = p + 16 * u + q;
#pragma unroll
            for (int t = 0; t < CT; ++t)
                a[u][t] = (row < r1 && chok[t]) ? x[(size_t)row * ic + 16 * t + n] - mu[t] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int i = 0; i < CT; ++i)
#pragma unroll
                for (int j = 0; j < CT; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][i], a[u][j], acc[i][j], 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < CT; ++i)
#pragma unroll
        for (int j = 0; j < CT; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) red[wave][i * CT + j][r][lane] = acc[i][j][r];
    __syncthreads();
    // acc[i][j][r] of lane l = G[16 i + 4 (l/16) + r][16 j + l%16]
    for (int e = tid; e < ic * ic; e += 256) {
        const int gi = e / ic, gj = e - gi * ic;
        const int i = gi >> 4, ri = gi & 15, j = gj >> 4, l = (ri >> 2) * 16 + (gj & 15), r = ri & 3;
        part[(size_t)blockIdx.x * ic * ic + e] =
            (red[0][i * CT + j][r][l] + red[1][i * CT + j][r][l]) + (red[2][i * CT + j][r][l] + red[3][i * CT + j][r][l]);
    }
}

// stats1[2*(off_g + m)] = sum_p E, [.. + 1] = sum_p E^2 (the format of the k_expand_fwd epilogue), zeros for pad channels
__global__ __launch_bounds__(256) void k_stats1_gram(TfnasCellDesc d, const double* __restrict__ xsum,
                                                     const double* __restrict__ C, double* __restrict__ stats1) {
    __shared__ double Cs[40 * 40 + 40];
    const int ic = d.ic;
    for (int i = threadIdx.x; i < ic * ic + ic; i += 256) Cs[i] = i < ic * ic ? C[i] : xsum[i - ic * ic];
    __syncthreads();
    const int col = blockIdx.x * 256 + threadIdx.x;
    if (col >= d.M) return;
    int g = 0;
    for (; g < d.G - 1; ++g)
        if (col < d.g[g + 1].off) break;
    const int m = col - d.g[g].off;
    double s = 0.0, q = 0.0;
    if (m < d.g[g].mc) {
        const float* __restrict__ w = d.g[g].w_expand + (size_t)m * ic;
        const double P = (double)d.N * d.H * d.W;
        float wr[40];
#pragma unroll
        for (int c = 0; c < 40; ++c) wr[c] = c < ic ? w[c] : 0.f;
        for (int c = 0; c < ic; ++c) {
            double t0 = 0.0, t1 = 0.0;
            const double* row = Cs + c * ic;
#pragma unroll
            for (int c2 = 0; c2 < 40; c2 += 2) {
                if (c2 < ic) {
                    t0 += row[c2] * (double)wr[c2];
                    t1 += row[c2 + 1] * (double)wr[c2 + 1];
                }
            }
            q += (double)w[c] * (t0 + t1);
            s += (double)w[c] * Cs[ic * ic + c];
        }
        q += s * s / P;                                              // sum E^2 = centred sum of squares + P mean^2
    }
    stats1[2 * (size_t)col + 0] = s;
    stats1[2 * (size_t)col + 1] = q;
}

bool efree_supported(const TfnasCellDesc& d) {
    if (d.mode != TFNAS_MODE_CELL || d.need_wgrad) return false;
    if (!efree_ic_ok(d.ic)) return fx_supported(d);            // late cells: the fused per-image route (fx_kernels.hip)
    if (stats_sync_on(d)) return false;          // (cross-rank statistics are reduced on the (sum, sumsq) tables of E)
    if ((size_t)d.N * d.H * d.W * d.ic >= ((size_t)1 << 31)) return false;
    for (int g = 0; g < d.G; ++g)
        if (d.g[g].k != 3 && d.g[g].k != 5) return false;
    return true;
}

int launch_x_colsum(const float* x, int P, int ic, int rps, int nb, float* part, hipStream_t s) {
    ProfScope _prof(TK_EXPAND_FWD, s);
    hipLaunchKernelGGL(k_x_colsum, dim3(nb), dim3(256), 0, s, x, P, ic, rps, part);
    return (int)hipGetLastError();
}

// scratch: `part` rows [nb][ic*ic] from the bottom; the double results sx[ic] | C[ic*ic] in the top of the buffer
int launch_expand_stats_gram(const TfnasCellDesc& d, const float* x, double* stats1, float* part, hipStream_t s) {
    const int P = d.N * d.H * d.W, ic = d.ic, ne = ic * ic;
    int rps = cdiv(P, 1024);
    if (rps < 256) rps = 256;
    rps = (rps + 63) & ~63;
    const int nb = cdiv(P, rps);
    double* xsum = reinterpret_cast<double*>(part + TFNAS_PART_FLOATS) - (ne + ic + 2);
    xsum = reinterpret_cast<double*>((uintptr_t)xsum & ~(uintptr_t)15);
    double* C = xsum + ic;
    if ((size_t)nb * ne + 2 * (size_t)(ne + ic + 4) > TFNAS_PART_FLOATS) return TFNAS_ERANGE;
    {
        ProfScope _prof(TK_EXPAND_FWD, s);
        hipLaunchKernelGGL(k_x_colsum, dim3(nb), dim3(256), 0, s, x, P, ic, rps, part);
    }
    int rc = launch_reduce_rows(part, nb, ic, (size_t)ic, xsum, nullptr, s);
    if (rc) return rc;
    {
        ProfScope _prof(TK_EXPAND_FWD, s);
        if (ic <= 16) hipLaunchKernelGGL(k_x_gram<1>, dim3(nb), dim3(256), 0, s, x, P, ic, rps, xsum, part);
        else if (ic <= 32) hipLaunchKernelGGL(k_x_gram<2>, dim3(nb), dim3(256), 0, s, x, P, ic, rps, xsum, part);
        else hipLaunchKernelGGL(k_x_gram<3>, dim3(nb), dim3(256), 0, s, x, P, ic, rps, xsum, part);
    }
    rc = launch_reduce_rows(part, nb, ne, (size_t)ne, C, nullptr, s);
    if (rc) return rc;
    ProfScope _prof(TK_SMALL, s);
    hipLaunchKernelGGL(k_stats1_gram, dim3(cdiv(d.M, 256)), dim3(256), 0, s, d, xsum, C, stats1);
    return (int)hipGetLastError();
}
