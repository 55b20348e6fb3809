// Opt-in per-kernel-family timing with HIP events on the launch stream (diagnostics for bench.py's roofline
// line; disabled by default -> zero overhead beyond one load).  See tfnas_prof_* in include/tfnas_hip.h.
#pragma once
#include <hip/hip_runtime.h>

enum TfnasKernelId {
    TK_EXPAND_FWD = 0, TK_DW_FWD, TK_SE_POOL, TK_SE_FC_FWD, TK_PROJECT_FWD, TK_MIX_FWD,
    TK_MIX_BWD_STATS, TK_PROJECT_DGRAD, TK_PROJECT_WGRAD, TK_SE_BWD_REDUCE, TK_SE_FC_BWD, TK_SE_WGRAD,
    TK_BN2_BWD, TK_DW_BWD_DATA, TK_DW_WGRAD, TK_EXPAND_DGRAD, TK_EXPAND_WGRAD, TK_SMALL, TK_REDUCE_ROWS, TK_COUNT
};

// Ablation builds (make EXTRA=-DTFNAS_ABLATE, tools/ablate.sh; NOT the product library): every kernel launch issued inside a
// ProfScope whose family bit is set in the environment variable TFNAS_ABLATE_MASK is dropped -- "what does the step cost without
// this family" (its consumers then read stale data: timing only).
#ifdef TFNAS_ABLATE
extern thread_local int g_tfnas_skip;
#undef hipLaunchKernelGGL
#define hipLaunchKernelGGL(kernelName, ...)                                   \
    do {                                                                      \
        if (!g_tfnas_skip) hipLaunchKernelGGLInternal((kernelName), __VA_ARGS__); \
    } while (0)
#endif

struct ProfScope {
    int id;
    hipStream_t s;
    hipEvent_t e0;
    bool on, soft;
#ifdef TFNAS_ABLATE
    int prev_skip;
#endif
    // soft: the launch covers all candidates of a cell (alpha-step); reported separately by tfnas_prof_last_split
    ProfScope(int id, hipStream_t s, bool soft = false);
    void stop();          // record the end event now (idempotent); the destructor calls it
    ~ProfScope() { stop(); }
};
