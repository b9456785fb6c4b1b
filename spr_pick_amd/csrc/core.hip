// libsprk.so: error reporting, diagnostics and the event-based kernel timing used by
// bench.py's roofline leg.
#include "common.h"
#include "conv16.h"
#include "wgrad16.h"

#include <mutex>
#include <vector>

namespace sprk {

static thread_local char t_error[512] = "";
std::atomic<long> g_launches{0}, g_wino_launches{0};

void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(t_error, sizeof(t_error), fmt, ap);
    va_end(ap);
}

// ---- profiling: hipEvent pairs around the MFMA convolution launches ----------------------
struct ProfRec {
    hipEvent_t a, b;
    int kclass;
    double flops;
};
static std::mutex g_prof_mu;
static int g_prof_mask = 0;   // bit k: kernel class k is bracketed by events
static std::vector<ProfRec> g_recs;
static std::vector<std::pair<hipEvent_t, hipEvent_t>> g_pool;
static thread_local int t_open = -1;

void prof_begin(int kclass, double flops, hipStream_t s) {
    if (!((g_prof_mask >> kclass) & 1)) return;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    ProfRec r;
    if (!g_pool.empty()) {
        r.a = g_pool.back().first;
        r.b = g_pool.back().second;
        g_pool.pop_back();
    } else {
        if (hipEventCreate(&r.a) != hipSuccess || hipEventCreate(&r.b) != hipSuccess) return;
    }
    r.kclass = kclass;
    r.flops = flops;
    (void)hipEventRecord(r.a, s);
    g_recs.push_back(r);
    t_open = (int)g_recs.size() - 1;
}

void prof_end(int kclass, hipStream_t s) {
    if (!((g_prof_mask >> kclass) & 1) || t_open < 0) return;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    if (t_open < (int)g_recs.size()) (void)hipEventRecord(g_recs[t_open].b, s);
    t_open = -1;
}

}  // namespace sprk

extern "C" {

const char *sprk_last_error(void) { return sprk::t_error; }
int sprk_version(void) { return 100; }
long sprk_launch_count(void) { return sprk::g_launches.load(); }
long sprk_wino_launch_count(void) { return sprk::g_wino_launches.load(); }
long sprk_conv16_launch_count(void) { return sprk::conv16_launches() + sprk::wgrad16_launches(); }
long sprk_wgrad16_launch_count(void) { return sprk::wgrad16_launches(); }

void sprk_prof_enable(int on) {
    std::lock_guard<std::mutex> lk(sprk::g_prof_mu);
    sprk::g_prof_mask = on;
}

int sprk_prof_collect(int kclass, long *launches, double *ms, double *flops) {
    std::lock_guard<std::mutex> lk(sprk::g_prof_mu);
    long n = 0;
    double t = 0.0, f = 0.0;
    std::vector<sprk::ProfRec> keep;
    for (auto &r : sprk::g_recs) {
        if (r.kclass != kclass) {
            keep.push_back(r);
            continue;
        }
        float dt = 0.f;
        if (hipEventSynchronize(r.b) == hipSuccess && hipEventElapsedTime(&dt, r.a, r.b) == hipSuccess) {
            ++n;
            t += dt;
            f += r.flops;
        }
        sprk::g_pool.push_back({r.a, r.b});
    }
    sprk::g_recs.swap(keep);
    if (launches) *launches = n;
    if (ms) *ms = t;
    if (flops) *flops = f;
    return SPRK_OK;
}

}  // extern "C"
