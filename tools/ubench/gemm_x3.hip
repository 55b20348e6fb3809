// Micro-benchmark + layout check of the split-bf16 ("x3") MFMA GEMM main loop against the fp32-MFMA A-direct loop
// (scratch tool; not part of the product path).
//   C[P][N] = A[P][K] * B[N][K]^T, A rows K-contiguous (activations), B rows K-contiguous (weights)
// fp32 value v = h + m + l exactly, each a bf16 (truncating split: 8 + 8 + 8 significand bits); the product is
//   a*b ~= l*h' + m*m' + h*l' + m*h' + h*m' + h*h'      (dropped terms <= 2^-23 |a||b|)
// six v_mfma_f32_16x16x32_bf16 (~17 cycles each, K = 32) replace eight v_mfma_f32_16x16x4_f32 (32 cycles each, K = 4).
// Build: hipcc --offload-arch=gfx950 -O3 -o gemm_x3 gemm_x3.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned fbits(float v) { return __builtin_bit_cast(unsigned, v); }
__device__ __forceinline__ float bitsf(unsigned v) { return __builtin_bit_cast(float, v); }
// upper halves of two floats -> one dword (lo = a, hi = b)
__device__ __forceinline__ unsigned pack_hi(unsigned a, unsigned b) { return __builtin_amdgcn_perm(b, a, 0x07060302u); }

struct Planes4 { u32x2 h, m, l; };   // four values -> three planes of four bf16
template <int TERMS>
__device__ __forceinline__ Planes4 split4(f32x4 v) {
    unsigned hb[4], mb[4], lb[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        hb[e] = fbits(v[e]);
        if (TERMS > 1) {
            const float r1 = v[e] - bitsf(hb[e] & 0xffff0000u);
            mb[e] = fbits(r1);
            if (TERMS > 3) lb[e] = fbits(r1 - bitsf(mb[e] & 0xffff0000u));
        }
    }
    Planes4 p;
    p.h = u32x2{pack_hi(hb[0], hb[1]), pack_hi(hb[2], hb[3])};
    if (TERMS > 1) p.m = u32x2{pack_hi(mb[0], mb[1]), pack_hi(mb[2], mb[3])};
    if (TERMS > 3) p.l = u32x2{pack_hi(lb[0], lb[1]), pack_hi(lb[2], lb[3])};
    return p;
}

// slot permutation that makes the ds_read_b128 lane groups conflict-free on 64-byte rows (groups mix lk with lk^1)
__device__ __forceinline__ int fperm(int q) { return (0x1320 >> (4 * q)) & 3; }

// ---------------------------------------------------------------------------------------------------------------- x3 kernel
// 256 threads = 4 waves x 32 rows; BN = 16 NT columns; one K step = 32 (two 16-chunks: lane (lr, lk) holds k = 16 h + 4 lk + t)
template <int NT, int TERMS, int WPS>
__global__ __launch_bounds__(256, WPS) void k_x3(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C,
                                                  int P, int N, int K) {
    constexpr int BN = 16 * NT, PLANE = BN * 64, NPL = TERMS > 3 ? 3 : (TERMS > 1 ? 2 : 1), BUF = NPL * PLANE;
    constexpr int B_ITEMS = BN * 8, B_ITERS = (B_ITEMS + 255) / 256;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wrow = (tid >> 6) * 32, lr = lane & 15, lk = lane >> 4;
    const int n0 = blockIdx.y * BN, nC = K / 32, nrt = P / 128;
    const unsigned rd_base = lr * 64 + ((lk ^ fperm(lr >> 2)) * 16);
    for (int rt = blockIdx.x; rt < nrt; rt += gridDim.x) {
        f32x4 acc[2][NT];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0, 0, 0, 0};
        const float* arow[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) arow[i] = A + (size_t)(rt * 128 + wrow + 16 * i + lr) * K + 4 * lk;
        f32x4 ra[2][2], rb[B_ITERS];
        u32x4 ca[2][3];
        auto gload = [&](int c) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int h = 0; h < 2; ++h) ra[i][h] = *(const f32x4*)(arow[i] + c * 32 + 16 * h);
#pragma unroll
            for (int it = 0; it < B_ITERS; ++it) {
                const int idx = tid + 256 * it, n = idx >> 3, kq = idx & 7;
                if (idx < B_ITEMS) rb[it] = *(const f32x4*)(B + (size_t)(n0 + n) * K + c * 32 + 4 * kq);
            }
        };
        auto sstore = [&](unsigned char* Bs) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const Planes4 p0 = split4<TERMS>(ra[i][0]), p1 = split4<TERMS>(ra[i][1]);
                ca[i][0] = u32x4{p0.h.x, p0.h.y, p1.h.x, p1.h.y};
                if (TERMS > 1) ca[i][1] = u32x4{p0.m.x, p0.m.y, p1.m.x, p1.m.y};
                if (TERMS > 3) ca[i][2] = u32x4{p0.l.x, p0.l.y, p1.l.x, p1.l.y};
            }
#pragma unroll
            for (int it = 0; it < B_ITERS; ++it) {
                const int idx = tid + 256 * it, n = idx >> 3, kq = idx & 7;
                if (idx < B_ITEMS) {
                    const Planes4 p = split4<TERMS>(rb[it]);
                    // k = 4 kq + t: half = kq >> 2, logical slot = kq & 3
                    const unsigned off = n * 64 + (((kq & 3) ^ fperm((n >> 2) & 3)) * 16) + (kq >> 2) * 8;
                    *(u32x2*)(Bs + off) = p.h;
                    if (TERMS > 1) *(u32x2*)(Bs + PLANE + off) = p.m;
                    if (TERMS > 3) *(u32x2*)(Bs + 2 * PLANE + off) = p.l;
                }
            }
        };
        gload(0);
        sstore(lds);
        __syncthreads();
        for (int c = 0; c < nC; ++c) {
            const unsigned char* Bs = lds + (c & 1) * BUF;
            const bool more = c + 1 < nC;
            if (more) gload(c + 1);
            bf16x8 ah[2], am[2], al[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                ah[i] = __builtin_bit_cast(bf16x8, ca[i][0]);
                if (TERMS > 1) am[i] = __builtin_bit_cast(bf16x8, ca[i][1]);
                if (TERMS > 3) al[i] = __builtin_bit_cast(bf16x8, ca[i][2]);
            }
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const bf16x8 bh = __builtin_bit_cast(bf16x8, *(const u32x4*)(Bs + rd_base + j * 1024));
                bf16x8 bm, bl;
                if (TERMS > 1) bm = __builtin_bit_cast(bf16x8, *(const u32x4*)(Bs + PLANE + rd_base + j * 1024));
                if (TERMS > 3) bl = __builtin_bit_cast(bf16x8, *(const u32x4*)(Bs + 2 * PLANE + rd_base + j * 1024));
#define MF(a, b) \
    acc[0][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b, acc[0][j], 0, 0, 0); \
    acc[1][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], b, acc[1][j], 0, 0, 0);
                if (TERMS > 3) { MF(al, bh) MF(am, bm) MF(ah, bl) }
                if (TERMS > 1) { MF(am, bh) MF(ah, bm) }
                MF(ah, bh)
#undef MF
            }
            if (more) sstore(lds + ((c + 1) & 1) * BUF);
            __syncthreads();
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    C[(size_t)(rt * 128 + wrow + 16 * i + 4 * lk + r) * N + n0 + 16 * j + lr] = acc[i][j][r];
    }
}


// ---------------------------------------------------------------------------------------------------------------- x3 kernel, B given K-major
// B[K][N] (N-contiguous rows, what the data-gradient GEMMs see: W[k][n..n+3] quads): planes staged as [16-col tile][k/4][4][16]
// bf16 blocks of 128 bytes (tile stride 1056 B) and read with the transposing ds_read_b64_tr_b16 -- no register transposes.
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
template <int NT, int TERMS, int WPS>
__global__ __launch_bounds__(256, WPS) void k_x3_tr(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C,
                                                     int P, int N, int K) {
    constexpr int BN = 16 * NT, TS = 1056, PLANE = NT * TS, NPL = TERMS > 3 ? 3 : (TERMS > 1 ? 2 : 1), BUF = NPL * PLANE;
    constexpr int B_ITEMS = BN * 8, B_ITERS = (B_ITEMS + 255) / 256, Q = BN / 4;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wrow = (tid >> 6) * 32, lr = lane & 15, lk = lane >> 4;
    const int n0 = blockIdx.y * BN, nC = K / 32, nrt = P / 128;
    const unsigned rd_base = lk * 128 + lr * 8;
    for (int rt = blockIdx.x; rt < nrt; rt += gridDim.x) {
        f32x4 acc[2][NT];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0, 0, 0, 0};
        const float* arow[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) arow[i] = A + (size_t)(rt * 128 + wrow + 16 * i + lr) * K + 4 * lk;
        f32x4 ra[2][2], rb[B_ITERS];
        u32x4 ca[2][3];
        auto gload = [&](int c) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int h = 0; h < 2; ++h) ra[i][h] = *(const f32x4*)(arow[i] + c * 32 + 16 * h);
#pragma unroll
            for (int it = 0; it < B_ITERS; ++it) {
                const int idx = tid + 256 * it, k = idx / Q, n4 = (idx - k * Q) * 4;
                if (idx < B_ITEMS) rb[it] = *(const f32x4*)(B + (size_t)(c * 32 + k) * N + n0 + n4);
            }
        };
        auto sstore = [&](unsigned char* Bs) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const Planes4 p0 = split4<TERMS>(ra[i][0]), p1 = split4<TERMS>(ra[i][1]);
                ca[i][0] = u32x4{p0.h.x, p0.h.y, p1.h.x, p1.h.y};
                if (TERMS > 1) ca[i][1] = u32x4{p0.m.x, p0.m.y, p1.m.x, p1.m.y};
                if (TERMS > 3) ca[i][2] = u32x4{p0.l.x, p0.l.y, p1.l.x, p1.l.y};
            }
#pragma unroll
            for (int it = 0; it < B_ITERS; ++it) {
                const int idx = tid + 256 * it, k = idx / Q, n4 = (idx - k * Q) * 4;
                if (idx < B_ITEMS) {
                    const Planes4 p = split4<TERMS>(rb[it]);
                    // k = 16 h + 4 lk + row  ->  block kk = 4 h + lk = k >> 2
                    const unsigned off = (n4 >> 4) * TS + (k >> 2) * 128 + (k & 3) * 32 + ((n4 & 15) >> 2) * 8;
                    *(u32x2*)(Bs + off) = p.h;
                    if (TERMS > 1) *(u32x2*)(Bs + PLANE + off) = p.m;
                    if (TERMS > 3) *(u32x2*)(Bs + 2 * PLANE + off) = p.l;
                }
            }
        };
        auto frag = [&](const unsigned char* p) -> bf16x8 {
            typedef __attribute__((address_space(3))) s16x4* lp;
            const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(p));
            const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(p + 512));
            typedef short s16x8 __attribute__((ext_vector_type(8)));
            const s16x8 v = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
            return __builtin_bit_cast(bf16x8, v);
        };
        gload(0);
        sstore(lds);
        __syncthreads();
        for (int c = 0; c < nC; ++c) {
            const unsigned char* Bs = lds + (c & 1) * BUF;
            const bool more = c + 1 < nC;
            if (more) gload(c + 1);
            bf16x8 ah[2], am[2], al[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                ah[i] = __builtin_bit_cast(bf16x8, ca[i][0]);
                if (TERMS > 1) am[i] = __builtin_bit_cast(bf16x8, ca[i][1]);
                if (TERMS > 3) al[i] = __builtin_bit_cast(bf16x8, ca[i][2]);
            }
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const bf16x8 bh = frag(Bs + rd_base + j * TS);
                bf16x8 bm, bl;
                if (TERMS > 1) bm = frag(Bs + PLANE + rd_base + j * TS);
                if (TERMS > 3) bl = frag(Bs + 2 * PLANE + rd_base + j * TS);
#define MF(a, b) \
    acc[0][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b, acc[0][j], 0, 0, 0); \
    acc[1][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], b, acc[1][j], 0, 0, 0);
                if (TERMS > 3) { MF(al, bh) MF(am, bm) MF(ah, bl) }
                if (TERMS > 1) { MF(am, bh) MF(ah, bm) }
                MF(ah, bh)
#undef MF
            }
            if (more) sstore(lds + ((c + 1) & 1) * BUF);
            __syncthreads();
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    C[(size_t)(rt * 128 + wrow + 16 * i + 4 * lk + r) * N + n0 + 16 * j + lr] = acc[i][j][r];
    }
}

// ---------------------------------------------------------------------------------------------------------------- fp32 A-direct loop (as gemm_core.h)
template <int NT, int WPS>
__global__ __launch_bounds__(256, WPS) void k_f32(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C,
                                                   int P, int N, int K) {
    constexpr int BN = 16 * NT, LDB = (BN % 32 == 16) ? BN : BN + 16, BF = 16 * LDB;
    constexpr int B_ITEMS = BN * 4, B_ITERS = (B_ITEMS + 255) / 256;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    float* lds = reinterpret_cast<float*>(lds_raw);
    const int tid = threadIdx.x, lane = tid & 63, wrow = (tid >> 6) * 32, lr = lane & 15, lk = lane >> 4;
    const int n0 = blockIdx.y * BN, nC = K / 16, nrt = P / 128;
    for (int rt = blockIdx.x; rt < nrt; rt += gridDim.x) {
        f32x4 acc[2][NT];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0, 0, 0, 0};
        const float* arow[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) arow[i] = A + (size_t)(rt * 128 + wrow + 16 * i + lr) * K + 4 * lk;
        f32x4 ra[2], rb[B_ITERS], cur[2];
        auto gload = [&](int c) {
#pragma unroll
            for (int i = 0; i < 2; ++i) ra[i] = *(const f32x4*)(arow[i] + c * 16);
#pragma unroll
            for (int it = 0; it < B_ITERS; ++it) {
                const int idx = tid + 256 * it, n = idx >> 2, kl = (idx & 3) * 4;
                if (idx < B_ITEMS) rb[it] = *(const f32x4*)(B + (size_t)(n0 + n) * K + c * 16 + kl);
            }
        };
        auto sstore = [&](float* Bs) {
#pragma unroll
            for (int i = 0; i < 2; ++i) cur[i] = ra[i];
#pragma unroll
            for (int it = 0; it < B_ITERS; ++it) {
                const int idx = tid + 256 * it, n = idx >> 2, kl = (idx & 3) * 4;
                if (idx < B_ITEMS) {
                    const int r0 = kl >> 2;
                    Bs[(r0 + 0) * LDB + n] = rb[it].x;
                    Bs[(r0 + 4) * LDB + n] = rb[it].y;
                    Bs[(r0 + 8) * LDB + n] = rb[it].z;
                    Bs[(r0 + 12) * LDB + n] = rb[it].w;
                }
            }
        };
        gload(0);
        sstore(lds);
        __syncthreads();
        for (int c = 0; c < nC; ++c) {
            const float* Bs = lds + (c & 1) * BF;
            const bool more = c + 1 < nC;
            if (more) gload(c + 1);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const int k = ks * 4 + lk;
                const float a0 = cur[0][ks], a1 = cur[1][ks];
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    const float b = Bs[k * LDB + 16 * j + lr];
                    acc[0][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b, acc[0][j], 0, 0, 0);
                    acc[1][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b, acc[1][j], 0, 0, 0);
                }
            }
            if (more) sstore(lds + ((c + 1) & 1) * BF);
            __syncthreads();
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    C[(size_t)(rt * 128 + wrow + 16 * i + 4 * lk + r) * N + n0 + 16 * j + lr] = acc[i][j][r];
    }
}

static double check(const std::vector<float>& A, const std::vector<float>& B, const float* C, int P, int N, int K, int rows) {
    double worst = 0;
    for (int s = 0; s < rows; ++s) {
        const int p = (int)(((long)s * 7919) % P);
        for (int n = 0; n < N; ++n) {
            double ref = 0, mag = 0;
            for (int k = 0; k < K; ++k) {
                const double t = (double)A[(size_t)p * K + k] * B[(size_t)n * K + k];
                ref += t;
                mag += fabs(t);
            }
            const double e = fabs(ref - C[(size_t)p * N + n]) / (mag + 1e-30);
            if (e > worst) worst = e;
        }
    }
    return worst;
}

template <class F>
static float time_ms(F launch, int reps) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    launch();
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < reps; ++r) launch();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    return ms / reps;
}

template <int NT, int WPS>
static void shape(int P, int N, int K, int gx) {
    std::vector<float> A((size_t)P * K), B((size_t)N * K), C((size_t)P * N);
    srand(1);
    for (auto& v : A) v = (float)rand() / RAND_MAX * 2.f - 1.f + 0.37f;
    for (auto& v : B) v = ((float)rand() / RAND_MAX * 2.f - 1.f) * 0.1f;
    float *dA, *dB, *dC;
    hipMalloc(&dA, A.size() * 4);
    hipMalloc(&dB, B.size() * 4);
    hipMalloc(&dC, C.size() * 4);
    hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
    const dim3 grid(gx, N / (16 * NT));
    const double gf = 2.0 * P * N * K * 1e-9;
    constexpr int BN = 16 * NT;
    auto run_x = [&](auto kern, size_t shm, const char* name) {
        hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
        hipMemset(dC, 0, C.size() * 4);
        const float ms = time_ms([&] { hipLaunchKernelGGL(kern, grid, dim3(256), shm, 0, dA, dB, dC, P, N, K); }, 20);
        hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost);
        int occ = 0;
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kern, 256, shm);
        printf("  %-10s NT=%d wps=%d occ=%d  %8.1f us  %7.1f TF/s-equiv   max err/sum|ab| = %.2e\n", name, NT, WPS, occ, ms * 1e3,
               gf / ms, check(A, B, C.data(), P, N, K, 48));
    };
    printf("P=%d N=%d K=%d grid=(%d,%d)  %.2f GFLOP\n", P, N, K, gx, N / BN, gf);
    run_x(k_f32<NT, WPS>, 2 * 16 * ((BN % 32 == 16) ? BN : BN + 16) * 4, "fp32");
    run_x(k_x3<NT, 6, WPS>, 2 * 3 * BN * 64, "bf16x3");
    run_x(k_x3<NT, 3, WPS>, 2 * 2 * BN * 64, "bf16x2");
    run_x(k_x3<NT, 1, WPS>, 2 * 1 * BN * 64, "bf16");
    {   // the same product with B stored K-major ([K][N]): transposing LDS reads
        std::vector<float> Bt((size_t)N * K);
        for (int n = 0; n < N; ++n)
            for (int k = 0; k < K; ++k) Bt[(size_t)k * N + n] = B[(size_t)n * K + k];
        hipMemcpy(dB, Bt.data(), Bt.size() * 4, hipMemcpyHostToDevice);
        run_x(k_x3_tr<NT, 6, WPS>, 2 * 3 * NT * 1056, "x3 tr");
        run_x(k_x3_tr<NT, 1, WPS>, 2 * 1 * NT * 1056, "bf16 tr");
    }
    hipFree(dA);
    hipFree(dB);
    hipFree(dC);
}

int main() {
    // soft late cell (cell 10: 14x14, ic 112, M 4032): expand forward  P x ic -> P x M   (short K)
    shape<7, 2>(25088, 4032, 128, 196);
    shape<4, 3>(25088, 4032, 128, 196);
    // project-forward-like: long K, one group of 672 -> 112, all 6+ groups side by side = 36 column tiles
    shape<7, 2>(25088, 112 * 8, 672, 196);
    // sampled late cell: one candidate
    shape<7, 3>(25088, 112, 672, 196);
    shape<7, 3>(25088, 672, 128, 196);
    // 28x28 cell (ic 40 -> 64 padded): P = 100352
    shape<4, 3>(100352, 1408, 64, 784);
    return 0;
}
