// Classifier + cross-entropy tail of a search step (reference: `x = self.classifier(x)` models/model_search.py:301-303, a
// LinearLayer(1280, num_classes) with bias, and `criterion = nn.CrossEntropyLoss()` train_search.py:107 as called at :333,376-379,410
// and differentiated by loss.backward()).
//
// Rounds 1-5 left this to stock torch ops: per sampled path a GEMM, log-softmax, nll, their three backward kernels, two more GEMMs and
// a bias reduction -- ten launches of 3-9 us on the w-step's critical chain, twice (bi-sampling), between the end of the cells' forward
// and the start of their backward, when nothing else is on the chip.  Here: ONE launch per path for everything that depends on one
// image (logits, loss, d logits, d pooled) and ONE launch for what sums over images and paths (dW, db, the loss scalar).
// Deterministic: fixed summation orders, no atomics.
#include "tfnas_dev.h"
#include "kernels.h"
#include "prof.h"

// One workgroup of 16 waves per image (the launch sits on the step's critical chain with nothing beside it: latency rounds, not
// occupancy, are what counts -- a first 4-wave version took 45 us, five dependent load rounds per class batch).  pooled[n] is staged
// in LDS; wave w takes classes w, w + 16, ... in batches of CB with every W load of a batch in flight before the first FMA, lanes
// stride the C features in float4 pieces; softmax / loss in one wave; then d pooled = d logits . W with the K classes split over
// KG thread groups (a thread owns four feature columns x one class range, partial sums combined through LDS in group order).
constexpr int CLS_CB = 4, CLS_T = 1024;
__global__ __launch_bounds__(CLS_T) void k_cls_ce(int C, int K, int KG, const float* __restrict__ pooled, const float* __restrict__ W,
                                                  const float* __restrict__ bias, const int64_t* __restrict__ target, float scale,
                                                  float* __restrict__ logits, float* __restrict__ loss_n,
                                                  float* __restrict__ dlogits, float* __restrict__ dpooled) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* xs = sm;                          // [C]
    float* lg = sm + C;                      // [K rounded up to 4] logits, then d logits
    float* pp = lg + ((K + 3) & ~3);         // [KG][C] partial d pooled
    const int n = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    constexpr int NW = CLS_T / 64;
    const float* xp = pooled + (size_t)n * C;
    for (int c = 4 * tid; c < C; c += 4 * CLS_T) st4(xs + c, ld4(xp + c));
    __syncthreads();
    for (int k0 = wave; k0 < K; k0 += NW * CLS_CB) {
        float acc[CLS_CB];
#pragma unroll
        for (int u = 0; u < CLS_CB; ++u) acc[u] = 0.f;
        for (int c0 = 0; c0 < C; c0 += 1024) {
            // four float4 pieces per lane and class (1024 features per round: C = 1280 is two rounds, the second mostly masked)
            f32x4 wv[CLS_CB][4];
#pragma unroll
            for (int u = 0; u < CLS_CB; ++u) {
                const float* wr = W + (size_t)min(k0 + NW * u, K - 1) * C;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int c = c0 + 256 * q + 4 * lane;
                    wv[u][q] = ld4(wr + (c < C ? c : 0));
                }
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int c = c0 + 256 * q + 4 * lane;
                const f32x4 xv = c < C ? ld4(xs + c) : zero4();
#pragma unroll
                for (int u = 0; u < CLS_CB; ++u)
                    acc[u] = fmaf(xv.x, wv[u][q].x, fmaf(xv.y, wv[u][q].y, fmaf(xv.z, wv[u][q].z, fmaf(xv.w, wv[u][q].w, acc[u]))));
            }
        }
#pragma unroll
        for (int u = 0; u < CLS_CB; ++u) {
            const float v = wave_sum(acc[u]);
            const int k = k0 + NW * u;
            if (lane == 0 && k < K) lg[k] = v + (bias ? bias[k] : 0.f);
        }
    }
    __syncthreads();
    if (wave == 0) {
        // log-softmax over K classes in one wave: max, sum of exp, in a fixed (lane-strided, then butterfly) order
        float m = -INFINITY;
        for (int k = lane; k < K; k += 64) m = fmaxf(m, lg[k]);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
        float s = 0.f;
        for (int k = lane; k < K; k += 64) s += expf(lg[k] - m);
        s = wave_sum(s);
        const float lse = m + logf(s);
        const int t = (int)target[n];
        const bool tok = t >= 0 && t < K;                    // (ignore_index-style targets contribute nothing; the reference has none)
        if (lane == 0) loss_n[n] = tok ? lse - lg[t] : 0.f;
        const float inv = 1.f / s;
        for (int k = lane; k < K; k += 64) {
            const float l = lg[k];
            logits[(size_t)n * K + k] = l;
            const float g = tok ? scale * (expf(l - m) * inv - (k == t ? 1.f : 0.f)) : 0.f;
            dlogits[(size_t)n * K + k] = g;
            lg[k] = g;
        }
    }
    __syncthreads();
    const int nq = C >> 2, kper = (K + KG - 1) / KG;
    for (int it = tid; it < nq * KG; it += CLS_T) {
        const int cq = it % nq, kg = it / nq;
        const int kb = kg * kper, ke = min(K, kb + kper);
        f32x4 a = zero4();
        int k = kb;
        for (; k + 8 <= ke; k += 8) {
            f32x4 wv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) wv[u] = ld4(W + (size_t)(k + u) * C + 4 * cq);
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const float g = lg[k + u];
                a.x = fmaf(g, wv[u].x, a.x); a.y = fmaf(g, wv[u].y, a.y); a.z = fmaf(g, wv[u].z, a.z); a.w = fmaf(g, wv[u].w, a.w);
            }
        }
        for (; k < ke; ++k) {
            const f32x4 wv = ld4(W + (size_t)k * C + 4 * cq);
            const float g = lg[k];
            a.x = fmaf(g, wv.x, a.x); a.y = fmaf(g, wv.y, a.y); a.z = fmaf(g, wv.z, a.z); a.w = fmaf(g, wv.w, a.w);
        }
        st4(pp + (size_t)kg * C + 4 * cq, a);
    }
    __syncthreads();
    for (int cq = tid; cq < nq; cq += CLS_T) {
        f32x4 a = ld4(pp + 4 * cq);
        for (int kg = 1; kg < KG; ++kg) {
            const f32x4 b = ld4(pp + (size_t)kg * C + 4 * cq);
            a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
        }
        st4(dpooled + (size_t)n * C + 4 * cq, a);
    }
}

struct ClsPaths {
    const float* pooled[2];
    const float* dlogits[2];
    const float* loss_n[2];
};

// dW[k][c] = sum over paths and images of d logits[n][k] * pooled[n][c]  (blockIdx.y < ceil(K / 4): four classes x 256 features per
// workgroup, image loop unrolled by 8);  the last blockIdx.y row: db[k] = sum d logits[.][k] and loss = scale * sum loss_n.
__global__ __launch_bounds__(256) void k_cls_wgrad(int npath, int N, int C, int K, ClsPaths P, float loss_scale,
                                                   float* __restrict__ dW, float* __restrict__ db, float* __restrict__ loss) {
    __shared__ float dl[256][4];
    const int tid = threadIdx.x;
    const int kgroups = (K + 3) >> 2;
    if ((int)blockIdx.y == kgroups) {
        if (blockIdx.x != 0) return;
        for (int k = tid; k < K; k += 256) {
            double s = 0.0;
            for (int p = 0; p < npath; ++p)
                for (int n = 0; n < N; ++n) s += (double)P.dlogits[p][(size_t)n * K + k];
            db[k] = (float)s;
        }
        if (tid < 64) {
            double s = 0.0;
            for (int p = 0; p < npath; ++p)
                for (int n = tid; n < N; n += 64) s += (double)P.loss_n[p][n];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
            if (tid == 0 && loss) loss[0] = (float)(s * (double)loss_scale);
        }
        return;
    }
    const int k0 = blockIdx.y * 4, c = blockIdx.x * 256 + tid;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    for (int p = 0; p < npath; ++p) {
        const float* __restrict__ xp = P.pooled[p];
        const float* __restrict__ gp = P.dlogits[p];
        for (int nb = 0; nb < N; nb += 256) {
            __syncthreads();
            {
                const int n = nb + tid;
#pragma unroll
                for (int q = 0; q < 4; ++q) dl[tid][q] = (n < N && k0 + q < K) ? gp[(size_t)n * K + k0 + q] : 0.f;
            }
            __syncthreads();
            const int cnt = min(256, N - nb);
            if (c < C) {
                int i = 0;
                for (; i + 8 <= cnt; i += 8) {
                    float xv[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) xv[u] = xp[(size_t)(nb + i + u) * C + c];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        a0 = fmaf(dl[i + u][0], xv[u], a0);
                        a1 = fmaf(dl[i + u][1], xv[u], a1);
                        a2 = fmaf(dl[i + u][2], xv[u], a2);
                        a3 = fmaf(dl[i + u][3], xv[u], a3);
                    }
                }
                for (; i < cnt; ++i) {
                    const float xv = xp[(size_t)(nb + i) * C + c];
                    a0 = fmaf(dl[i][0], xv, a0);
                    a1 = fmaf(dl[i][1], xv, a1);
                    a2 = fmaf(dl[i][2], xv, a2);
                    a3 = fmaf(dl[i][3], xv, a3);
                }
            }
        }
    }
    if (c < C) {
        if (k0 + 0 < K) dW[(size_t)(k0 + 0) * C + c] = a0;
        if (k0 + 1 < K) dW[(size_t)(k0 + 1) * C + c] = a1;
        if (k0 + 2 < K) dW[(size_t)(k0 + 2) * C + c] = a2;
        if (k0 + 3 < K) dW[(size_t)(k0 + 3) * C + c] = a3;
    }
}

// dst += src (n floats, a multiple of 4): the second bi-sampling path's share of a shared parameter's gradient
__global__ __launch_bounds__(256) void k_add_into(float* __restrict__ dst, const float* __restrict__ src, size_t n4) {
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        f32x4 a = ld4(dst + 4 * i);
        const f32x4 b = ld4(src + 4 * i);
        a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
        st4(dst + 4 * i, a);
    }
}

extern "C" int tfnas_cls_ce(int N, int C, int K, const float* pooled, const float* W, const float* bias, const int64_t* target,
                            float scale, float* logits, float* loss_n, float* dlogits, float* dpooled, void* stream) {
    if (!pooled || !W || !target || !logits || !loss_n || !dlogits || !dpooled) return TFNAS_ENULL;
    if (N < 1 || K < 1 || K > 4096 || C < 4 || C > 4096) return TFNAS_ERANGE;
    if (C & 3) return TFNAS_EINVAL;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    ProfScope _prof(TK_SMALL, s);
    // class groups of the d pooled phase: as many as fill the 1024 threads with (four feature columns x class range) items
    int KG = CLS_T / (C >> 2);
    if (KG < 1) KG = 1;
    if (KG > 8) KG = 8;
    if (KG > K) KG = K;
    const size_t shm = ((size_t)C + ((K + 3) & ~3) + (size_t)KG * C) * sizeof(float);
    if (shm > 64 * 1024) return TFNAS_ERANGE;
    hipLaunchKernelGGL(k_cls_ce, dim3(N), dim3(CLS_T), shm, s, C, K, KG, pooled, W, bias, target, scale, logits, loss_n, dlogits,
                       dpooled);
    return (int)hipGetLastError();
}

extern "C" int tfnas_cls_wgrad(int npath, int N, int C, int K, const float* const* pooled, const float* const* dlogits,
                               const float* const* loss_n, float loss_scale, float* dW, float* db, float* loss, void* stream) {
    if (!pooled || !dlogits || !loss_n || !dW || !db) return TFNAS_ENULL;
    if (npath < 1 || npath > 2 || N < 1 || K < 1 || C < 1) return TFNAS_ERANGE;
    ClsPaths P = {};
    for (int p = 0; p < npath; ++p) {
        if (!pooled[p] || !dlogits[p] || !loss_n[p]) return TFNAS_ENULL;
        P.pooled[p] = pooled[p];
        P.dlogits[p] = dlogits[p];
        P.loss_n[p] = loss_n[p];
    }
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    ProfScope _prof(TK_SMALL, s);
    hipLaunchKernelGGL(k_cls_wgrad, dim3(cdiv(C, 256), cdiv(K, 4) + 1), dim3(256), 0, s, npath, N, C, K, P, loss_scale, dW, db, loss);
    return (int)hipGetLastError();
}

extern "C" int tfnas_add_into(float* dst, const float* src, uint64_t count, void* stream) {
    if (!dst || !src) return TFNAS_ENULL;
    if (count & 3) return TFNAS_EINVAL;
    if (!count) return 0;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    ProfScope _prof(TK_SMALL, s);
    size_t blocks = cdiv64(count / 4, 256);
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(k_add_into, dim3((unsigned)blocks), dim3(256), 0, s, dst, src, (size_t)(count / 4));
    return (int)hipGetLastError();
}
