#include <mutex>
#include <vector>
#include "prof.h"
#include "tfnas_hip.h"

static unsigned g_mask = 0;
static std::mutex g_mu;
struct ProfEv { hipEvent_t a, b; bool soft; };
static std::vector<ProfEv> g_ev[TK_COUNT];
static uint64_t g_soft_n[TK_COUNT];
static double g_soft_ms[TK_COUNT];

static const char* kNames[TK_COUNT] = {
    "k_expand_fwd", "k_dw_fwd", "k_se_pool<fwd>", "k_se_fc_fwd", "k_project_fwd", "k_mix_fwd",
    "k_mix_bwd_stats", "k_project_dgrad", "k_project_wgrad", "k_se_pool<bwd>", "k_se_fc_bwd", "k_se_wgrad",
    "k_bn2_bwd", "k_dw_bwd_data", "k_dw_wgrad", "k_expand_dgrad", "k_expand_wgrad", "small(arch/sink/consts)",
    "k_reduce_rows"};

#ifdef TFNAS_ABLATE
#include <cstdlib>
thread_local int g_tfnas_skip = 0;
static unsigned ablate_mask() {
    static const unsigned m = getenv("TFNAS_ABLATE_MASK") ? (unsigned)strtoul(getenv("TFNAS_ABLATE_MASK"), nullptr, 0) : 0u;
    return m;
}
#endif

ProfScope::ProfScope(int id_, hipStream_t s_, bool soft_) : id(id_), s(s_), e0(nullptr), on(false), soft(soft_) {
#ifdef TFNAS_ABLATE
    prev_skip = g_tfnas_skip;
    g_tfnas_skip = (ablate_mask() >> id) & 1u;
#endif
    if (g_mask & (1u << id)) {
        if (hipEventCreate(&e0) == hipSuccess && hipEventRecord(e0, s) == hipSuccess) on = true;
    }
}
void ProfScope::stop() {
#ifdef TFNAS_ABLATE
    g_tfnas_skip = prev_skip;
#endif
    if (!on) return;
    on = false;
    hipEvent_t e1;
    if (hipEventCreate(&e1) != hipSuccess) return;
    (void)hipEventRecord(e1, s);
    std::lock_guard<std::mutex> lk(g_mu);
    g_ev[id].push_back({e0, e1, soft});
}

extern "C" int tfnas_prof_enable(unsigned mask) {
    g_mask = mask;
    return 0;
}
extern "C" int tfnas_prof_count(void) { return TK_COUNT; }
extern "C" const char* tfnas_prof_name(int id) { return (id >= 0 && id < TK_COUNT) ? kNames[id] : ""; }

extern "C" int tfnas_prof_collect(int id, uint64_t* launches, double* total_ms) {
    if (id < 0 || id >= TK_COUNT || !launches || !total_ms) return TFNAS_EINVAL;
    std::lock_guard<std::mutex> lk(g_mu);
    *launches = 0;
    *total_ms = 0.0;
    g_soft_n[id] = 0;
    g_soft_ms[id] = 0.0;
    for (auto& pr : g_ev[id]) {
        float ms = 0.f;
        hipError_t e = hipEventSynchronize(pr.b);
        if (e == hipSuccess) e = hipEventElapsedTime(&ms, pr.a, pr.b);
        (void)hipEventDestroy(pr.a);
        (void)hipEventDestroy(pr.b);
        if (e != hipSuccess) return (int)e;
        *launches += 1;
        *total_ms += ms;
        if (pr.soft) {
            g_soft_n[id] += 1;
            g_soft_ms[id] += ms;
        }
    }
    g_ev[id].clear();
    return 0;
}

extern "C" int tfnas_prof_last_split(int id, uint64_t* soft_launches, double* soft_ms) {
    if (id < 0 || id >= TK_COUNT || !soft_launches || !soft_ms) return TFNAS_EINVAL;
    std::lock_guard<std::mutex> lk(g_mu);
    *soft_launches = g_soft_n[id];
    *soft_ms = g_soft_ms[id];
    return 0;
}
