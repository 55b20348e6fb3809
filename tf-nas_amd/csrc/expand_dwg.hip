// Expand data gradient AND expand weight gradient of the wide early cells from ONE pass over dEh (round 5).
//
// Reference arithmetic: autograd of inverted_bottleneck.conv + BatchNorm2d (models/layers.py:463-482 of the reference), i.e. what
// k_expand_dgrad + k_expand_wgrad<XG> (gemm_kernels.hip) compute in two launches that each stream all of dEh:
//   dx[p][c]  = sum_m deh[p][m] (r_m W[m][c])  -  sum_c' x[p][c'] G[c'][c]  +  b[c]   (+ sumw dout[p][c])  (+ sink gradient)
//   R[m][c]   = sum_p deh[p][m] x[p][c],    Gx[c'][c] = sum_p x[p][c'] x[p][c],    sx[c] = sum_p x[p][c]
// (G | b: k_expand_gram; the BN1-backward correction of R from Gx | sx: k_expand_wgrad_fix -- both unchanged.)
//
// In the w-step the 112 x 112 / 56 x 56 cells are the byte-bound part of the step: dEh of cell 0 is 300-600 MB
// per sampled candidate and both kernels read all of it.  Here a workgroup walks 128-pixel row tiles; with one candidate per launch
// (G = 1) a tile of dEh is 128 whole rows: it is staged into LDS with coalesced 16-byte loads (LDS row pitch
// mcp + 4 floats: conflict-free fragment reads in both orientations) and consumed twice on the fp32 matrix cores --
//   rows = pixels, K = mid channels (data gradient: each wave its 32 rows; B = r . W and -G resident in LDS)
//   rows = mid channels | x channels | 1, K = pixels (weight gradient: each wave contracts its own 32 pixels; accumulators live
//   in registers across all tiles of the workgroup and are combined through LDS in wave order at the end: bit-reproducible).
// One partial row [R | Gx | sx] per workgroup, summed in double by k_reduce_rows.
#include "tfnas_dev.h"
#include "kernels.h"
#include "prof.h"

namespace {
constexpr int DWG_MT = 11;        // 16-row tiles of the weight-gradient side: mcp + ic + 1 <= 176 rows

__host__ __device__ inline int dwg_lp(int mcp) { return ((mcp >> 2) & 1) ? mcp + 8 : mcp + 4; }   // row pitch = 4 (mod 8) floats
__host__ __device__ inline int dwg_wp(int nt) { return nt == 1 ? 16 : 48; }                         // 16 (mod 32): four k rows, 64 banks

struct DwgLds {
    int lp, xp, wp, o_x, o_w, floats;
};
__host__ __device__ inline DwgLds dwg_lds(int mcp, int ic, int nt) {
    DwgLds L;
    L.lp = dwg_lp(mcp);
    L.xp = ic + 4;
    L.wp = dwg_wp(nt);
    L.o_x = 128 * L.lp;
    L.o_w = L.o_x + 128 * L.xp;
    L.floats = L.o_w + (mcp + ic) * L.wp;
    const int red = DWG_MT * 16 * 16 * nt;          // cross-wave combination of the weight-gradient tiles (reuses the front)
    if (L.floats < red) L.floats = red;
    return L;
}

template <int NT>
__global__ __launch_bounds__(256, 2) void k_expand_dwg(TfnasCellDesc d, const float* __restrict__ dEh, const float* __restrict__ x,
                                                       const float* __restrict__ cb1, const float* __restrict__ gram,
                                                       const float* __restrict__ dout, const float* __restrict__ wmix,
                                                       float* __restrict__ dx, const float* __restrict__ add_src,
                                                       const float* __restrict__ add_scale, float* __restrict__ part,
                                                       size_t out_size, size_t out_main) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int ic = d.ic, mc = d.g[0].mc, mcp = d.g[0].mcp, M = d.M;
    const int P = d.N * d.H * d.W, nrt = (P + 127) >> 7;
    const DwgLds L = dwg_lds(mcp, ic, NT);
    float* __restrict__ Td = lds;                   // [128][lp]   dEh tile
    float* __restrict__ Tx = lds + L.o_x;           // [128][xp]   x tile
    float* __restrict__ Wr = lds + L.o_w;           // [mcp + ic][wp]: rows m: r_m W[m][:], rows mcp + c': -G[c'][:]
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, lr = lane & 15, lq = lane >> 4;
    const int mq = mcp >> 2, iq = ic >> 2;
    const int mt = (mcp + ic + 1 + 15) >> 4;        // weight-gradient row tiles in use

    // ---- resident B operands of the data gradient
    for (int e = tid; e < (mcp + ic) * L.wp; e += 256) {
        const int r = e / L.wp, c = e - r * L.wp;
        float v = 0.f;
        if (c < ic) {
            if (r < mc) v = cb1[4 * (size_t)(d.g[0].off + r) + 1] * d.g[0].w_expand[(size_t)r * ic + c];
            else if (r >= mcp) v = -gram[(size_t)(r - mcp) * ic + c];
        }
        Wr[e] = v;
    }
    float bias[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) bias[j] = (16 * j + lr < ic) ? gram[(size_t)ic * ic + 16 * j + lr] : 0.f;
    const bool add_res = d.has_res != 0, add_sink = add_src != nullptr;
    const float sink_w = add_sink ? add_scale[0] : 0.f;
    float sumw = 1.f;
    if (wmix) {
        sumw = 0.f;
        for (int g = 0; g < d.G; ++g) sumw += wmix[g];
    }
    // weight-gradient A operand of row tile t: this lane's row 16 t + lr comes from a dEh column, an x column, the constant 1 or 0
    int aoff[DWG_MT], apitch[DWG_MT];               // LDS element offset of the row's column / row pitch (0: constant)
    float aconst[DWG_MT];
#pragma unroll
    for (int t = 0; t < DWG_MT; ++t) {
        const int row = 16 * t + lr;
        aoff[t] = 0; apitch[t] = 0; aconst[t] = 0.f;
        if (row < mcp) { aoff[t] = row; apitch[t] = L.lp; }
        else if (row < mcp + ic) { aoff[t] = L.o_x + row - mcp; apitch[t] = L.xp; }
        else if (row == mcp + ic) aconst[t] = 1.f;
    }
    f32x4 wacc[DWG_MT][NT];
#pragma unroll
    for (int t = 0; t < DWG_MT; ++t)
#pragma unroll
        for (int j = 0; j < NT; ++j) wacc[t][j] = zero4();

    for (int rt = blockIdx.x; rt < nrt; rt += gridDim.x) {
        const int p0 = rt * 128, rows = min(128, P - p0);
        __syncthreads();                            // the previous tile is consumed (first pass: Wr is complete)
        // ---- stage the tile: dEh rows are contiguous (G = 1: M == mcp), x rows too
        {
            const float* __restrict__ src = dEh + (size_t)p0 * M;          // (row pitch M >= mcp: the pad columns are never read)
            const int n4 = 128 * mq;
            for (int f = tid; f < n4; f += 256) {
                const int r = f / mq, c4 = f - r * mq;
                const f32x4 v = r < rows ? ld4(src + (size_t)r * M + 4 * c4) : zero4();
                st4(Td + r * L.lp + 4 * c4, v);
            }
            const f32x4* __restrict__ sx4 = reinterpret_cast<const f32x4*>(x + (size_t)p0 * ic);
            const int nx4 = 128 * iq, vx4 = rows * iq;
            for (int f = tid; f < nx4; f += 256) {
                const int r = f / iq, c4 = f - r * iq;
                const f32x4 v = f < vx4 ? sx4[f] : zero4();
                st4(Tx + r * L.xp + 4 * c4, v);
            }
        }
        __syncthreads();
        // ---- data gradient: this wave's 32 rows
        f32x4 dacc[2][NT];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) dacc[i][j] = zero4();
        {
            const float* a0p = Td + (32 * wv + lr) * L.lp + lq;
            const float* a1p = a0p + 16 * L.lp;
            const float* bp = Wr + lq * L.wp + lr;
#pragma unroll 4
            for (int s = 0; s < mq; ++s) {
                const float a0 = a0p[4 * s], a1 = a1p[4 * s];
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    const float b = bp[4 * s * L.wp + 16 * j];
                    dacc[0][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b, dacc[0][j], 0, 0, 0);
                    dacc[1][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b, dacc[1][j], 0, 0, 0);
                }
            }
            const float* x0p = Tx + (32 * wv + lr) * L.xp + lq;
            const float* x1p = x0p + 16 * L.xp;
            const float* gp = Wr + (mcp + lq) * L.wp + lr;
            for (int s = 0; s < iq; ++s) {
                const float a0 = x0p[4 * s], a1 = x1p[4 * s];
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    const float b = gp[4 * s * L.wp + 16 * j];
                    dacc[0][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b, dacc[0][j], 0, 0, 0);
                    dacc[1][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b, dacc[1][j], 0, 0, 0);
                }
            }
        }
        // dacc[i][j][r] = tile row 32 wv + 16 i + 4 lq + r, column 16 j + lr
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int p = p0 + 32 * wv + 16 * i + 4 * lq + r;
                if (p < P) {
#pragma unroll
                    for (int j = 0; j < NT; ++j) {
                        const int c = 16 * j + lr;
                        if (c < ic) {
                            float v = dacc[i][j][r] + bias[j];
                            if (add_res) v += sumw * dout[(size_t)p * d.oc + c];
                            if (add_sink) {
                                float pr = sink_w * add_src[(size_t)p * ic + c];     // product rounded on its own (see sink_add,
                                asm volatile("" : "+v"(pr));                          //  gemm_kernels.hip)
                                v = v + pr;
                            }
                            dx[(size_t)p * ic + c] = v;
                        }
                    }
                }
            }
        // ---- weight gradient: this wave contracts its own 32 pixels (rows past the end of the tensor are zero in both tiles)
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const int pl = 32 * wv + 4 * s + lq;
            float b[NT];
#pragma unroll
            for (int j = 0; j < NT; ++j) b[j] = (16 * j + lr < ic) ? Tx[pl * L.xp + 16 * j + lr] : 0.f;
            const bool prow = p0 + pl < P;
#pragma unroll
            for (int t = 0; t < DWG_MT; ++t) {
                if (t < mt) {
                    float a = apitch[t] ? lds[aoff[t] + pl * apitch[t]] : (prow ? aconst[t] : 0.f);
#pragma unroll
                    for (int j = 0; j < NT; ++j) wacc[t][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b[j], wacc[t][j], 0, 0, 0);
                }
            }
        }
    }
    // ---- combine the four waves' weight-gradient tiles in wave order, then write this workgroup's partial row
    float* __restrict__ red = lds;                  // [mt * 16 rows][16 NT columns]
    constexpr int RC = 16 * NT;
    for (int w = 0; w < 4; ++w) {
        __syncthreads();
        if (wv == w) {
#pragma unroll
            for (int t = 0; t < DWG_MT; ++t) {
                if (t < mt) {
#pragma unroll
                    for (int j = 0; j < NT; ++j)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            float* q = red + (16 * t + 4 * lq + r) * RC + 16 * j + lr;
                            *q = (w == 0 ? 0.f : *q) + wacc[t][j][r];
                        }
                }
            }
        }
    }
    __syncthreads();
    float* __restrict__ gw = part + (size_t)blockIdx.x * out_size;
    for (int e = tid; e < mt * 16 * RC; e += 256) {
        const int row = e / RC, c = e - row * RC;
        if (c >= ic) continue;
        if (row < mc) gw[(size_t)row * ic + c] = red[e];
        else if (row >= mcp && row <= mcp + ic) gw[out_main + (size_t)(row - mcp) * ic + c] = red[e];
    }
}
}  // namespace

// TFNAS_DWG = 1 (default) | 0: expand data + weight gradient of the wide early cells in one pass over dEh / in two kernels
static const bool g_expand_dwg = [] {
    const char* e = getenv("TFNAS_DWG");
    return !(e && e[0] == '0');
}();

bool expand_dwg_supported(const TfnasCellDesc& d) {
    if (!g_expand_dwg || d.mode != TFNAS_MODE_CELL || d.G != 1 || !d.need_wgrad) return false;
    if ((d.ic & 3) || d.ic > 32 || d.g[0].mcp + d.ic + 1 > 16 * DWG_MT || d.g[0].off != 0) return false;
    if (!expand_wgrad_gram_form(d)) return false;                  // the size policy of the Gram form (E >= 100 MB): same cells
    const size_t out_size = (size_t)d.g[0].mc * d.ic + (size_t)(d.ic + 1) * d.ic;
    return expand_gram_floats(d) + 2 * out_size + 8 + 256 * out_size <= TFNAS_PART_FLOATS;
}

// `part`: the data-gradient chain's scratch; its top holds G | b (launch_expand_gram), the partial rows go to the bottom, their
// double sums below G | b.  Everything on stream s.
int launch_expand_dwg(const TfnasCellDesc& d, const float* dEh, const float* x, const float* cb1, const float* gram,
                      const float* dout, const float* wmix, float* dx, const float* add_src, const float* add_scale, float* part,
                      hipStream_t s) {
    const int nt = d.ic <= 16 ? 1 : 2;
    const int P = d.N * d.H * d.W, nrt = (P + 127) >> 7;
    const size_t out_main = (size_t)d.g[0].mc * d.ic, out_size = out_main + (size_t)(d.ic + 1) * d.ic;
    const DwgLds L = dwg_lds(d.g[0].mcp, d.ic, nt);
    const size_t shm = (size_t)L.floats * sizeof(float);
    const int per_cu = shm > 80 * 1024 ? 1 : 2;
    const size_t avail = TFNAS_PART_FLOATS - expand_gram_floats(d) - 2 * out_size - 8;
    int nwg = 256 * per_cu;
    if ((size_t)nwg > avail / out_size) nwg = (int)(avail / out_size);
    if (nwg > nrt) nwg = nrt;
    if (nwg < 1) return TFNAS_ERANGE;
    double* red = reinterpret_cast<double*>((uintptr_t)(part + TFNAS_PART_FLOATS - expand_gram_floats(d) - 2 * out_size - 4) &
                                            ~(uintptr_t)15);
    {
        ProfScope _prof(TK_EXPAND_DGRAD, s);
        if (nt == 1) {
            static const bool ok = hipFuncSetAttribute(reinterpret_cast<const void*>(k_expand_dwg<1>),
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess;
            if (!ok) return TFNAS_EINVAL;
            hipLaunchKernelGGL(k_expand_dwg<1>, dim3(nwg), dim3(256), shm, s, d, dEh, x, cb1, gram, dout, wmix, dx, add_src,
                               add_scale, part, out_size, out_main);
        } else {
            static const bool ok = hipFuncSetAttribute(reinterpret_cast<const void*>(k_expand_dwg<2>),
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess;
            if (!ok) return TFNAS_EINVAL;
            hipLaunchKernelGGL(k_expand_dwg<2>, dim3(nwg), dim3(256), shm, s, d, dEh, x, cb1, gram, dout, wmix, dx, add_src,
                               add_scale, part, out_size, out_main);
        }
        const int rc = (int)hipGetLastError();
        if (rc) return rc;
    }
    int rc = launch_reduce_rows(part, nwg, (int)out_size, out_size, red, nullptr, s);
    if (rc) return rc;
    return launch_expand_wgrad_fix(d, cb1, red, out_main, s);
}
