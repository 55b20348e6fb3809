#pragma once
#include "tfnas_dev.h"

// "Last workgroup reduces": the workgroups that fill the same columns of a partials matrix (one row each) take a ticket when
// their row is written; the one that draws the last ticket sums the column block over all rows (double, fixed order:
// deterministic) and the dependent k_reduce_rows launch disappears.
// Coherence without fences: the 8 XCD L2s are not coherent for ordinary stores, and an agent-scope release fence
// (__threadfence) writes back EVERY dirty line of the XCD's L2 -- the kernel's own output stream: k_expand_fwd 0.25 -> 1.8 ms.
// So the partial rows are written with agent-scope (write-through, sc1) stores, the workgroup waits for their
// acknowledgement (s_waitcnt vmcnt(0)) before it takes its ticket, and the reducer reads them with system-coherent
// buffer loads (sc0 sc1: no stale line of an earlier launch from its own L2).
// `cnt`: this column block's ticket counter in the tail of the `part` scratch (kernels.h), zero between launches.
__device__ __forceinline__ bool tail_ticket(unsigned* cnt, int group_size, float* lds_flag) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    volatile unsigned* flag = reinterpret_cast<volatile unsigned*>(lds_flag);
    if (threadIdx.x == 0) {
        const unsigned t = atomicAdd(cnt, 1u);
        const bool last = t == (unsigned)group_size - 1u;
        if (last) *cnt = 0u;                         // everybody else has drawn: ready for the next launch
        flag[0] = last ? 1u : 0u;
    }
    __syncthreads();
    return flag[0] != 0u;
}
// Sum columns [col0, col0 + ncols) (multiples of 4) of the nb rows of `part` (row stride `stride` floats); out(c, t0, t1) is
// called once per column PAIR (c even: the (sum, sum of squares) / (T1, T2) of one channel).  All 256 threads of the ticket
// winner; scratch = lds_flag[0 .. 4 + 2048) floats.
template <class FOut>
__device__ __forceinline__ void tail_reduce_cols(const float* part, size_t part_floats, size_t stride, int nb, int col0,
                                                 int ncols, float* lds_flag, FOut out) {
    const __amdgpu_buffer_rsrc_t rs =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(part), 0, (int)(unsigned)(part_floats * 4), 0x00020000);
    double* red = reinterpret_cast<double*>(lds_flag + 4);
    const int Q = ncols >> 2;
    int QL = 1;
    while (QL < Q && QL < 64) QL <<= 1;
    const int RL = 256 / QL, ql = threadIdx.x % QL, rl = threadIdx.x / QL;
    for (int q0 = 0; q0 < Q; q0 += QL) {
        const int q = q0 + ql;
        double sx = 0.0, sy = 0.0, sz = 0.0, sw = 0.0;
        if (q < Q) {
            const unsigned cbase = (unsigned)(col0 + 4 * q) * 4u;
            for (int b = rl; b < nb; b += 8 * RL) {
                f32x4 v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int row = b + RL * u;
                    v[u] = row < nb ? __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                          rs, (int)((unsigned)((size_t)row * stride * 4u) + cbase), 0, 17))
                                    : zero4();
                }
#pragma unroll
                for (int u = 0; u < 8; u += 4) {
                    sx += ((double)v[u].x + (double)v[u + 1].x) + ((double)v[u + 2].x + (double)v[u + 3].x);
                    sy += ((double)v[u].y + (double)v[u + 1].y) + ((double)v[u + 2].y + (double)v[u + 3].y);
                    sz += ((double)v[u].z + (double)v[u + 1].z) + ((double)v[u + 2].z + (double)v[u + 3].z);
                    sw += ((double)v[u].w + (double)v[u + 1].w) + ((double)v[u + 2].w + (double)v[u + 3].w);
                }
            }
        }
        __syncthreads();
        red[(rl * QL + ql) * 4 + 0] = sx;
        red[(rl * QL + ql) * 4 + 1] = sy;
        red[(rl * QL + ql) * 4 + 2] = sz;
        red[(rl * QL + ql) * 4 + 3] = sw;
        __syncthreads();
        for (int i = threadIdx.x; i < 2 * QL; i += 256) {
            const int qq = i >> 1, half = (i & 1) * 2;
            if (q0 + qq < Q) {
                double t0 = 0.0, t1 = 0.0;
                for (int r = 0; r < RL; ++r) {
                    t0 += red[(r * QL + qq) * 4 + half];
                    t1 += red[(r * QL + qq) * 4 + half + 1];
                }
                out(col0 + 4 * (q0 + qq) + half, t0, t1);
            }
        }
    }
}

