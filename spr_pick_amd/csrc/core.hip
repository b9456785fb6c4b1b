// libsprk.so: error reporting, diagnostics and the event-based kernel timing used by
// bench.py's roofline leg.
#include "common.h"
#include "wprep_dev.h"
#include "conv16.h"
#include "wgrad16.h"

#include <mutex>
#include <tuple>
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

// ---- per-device facts -----------------------------------------------------------------------
constexpr int kMaxDev = 64;
int num_cus() {
    static std::atomic<int> cus[kMaxDev];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDev) return 256;
    int v = cus[dev].load(std::memory_order_relaxed);
    if (v == 0) {
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
        cus[dev].store(v, std::memory_order_relaxed);
    }
    return v;
}

int lds_optin(const void *kernel, size_t bytes, const char *what) {
    static std::mutex mu;
    static std::vector<std::tuple<int, const void *, size_t>> done;
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> lk(mu);
    for (auto &d : done)
        if (std::get<0>(d) == dev && std::get<1>(d) == kernel && std::get<2>(d) >= bytes) return SPRK_OK;
    if (hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess) {
        set_error("%s: cannot reserve %zu bytes of LDS", what, bytes);
        return SPRK_ELAUNCH;
    }
    done.emplace_back(dev, kernel, bytes);
    return SPRK_OK;
}

// ---- profiling: hipEvent pairs around the MFMA convolution launches ----------------------
struct ProfRec {
    hipEvent_t a, b;
    int kclass;
    double flops, bytes;
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
    r.bytes = 0.0;
    (void)hipEventRecord(r.a, s);
    g_recs.push_back(r);
    t_open = (int)g_recs.size() - 1;
}

void prof_bytes(double bytes) {
    if (t_open < 0) return;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    if (t_open < (int)g_recs.size()) g_recs[t_open].bytes = bytes;
}

void prof_end(int kclass, hipStream_t s) {
    if (!((g_prof_mask >> kclass) & 1) || t_open < 0) return;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    if (t_open < (int)g_recs.size()) (void)hipEventRecord(g_recs[t_open].b, s);
    t_open = -1;
}

}  // namespace sprk

namespace {

constexpr int kRedMax = 48;
struct RedTable {
    sprk_reduce_item it[kRedMax];
    int start[kRedMax + 1];   // first workgroup of item i
    int n;
};

// sum over p < parts of p0[p * stride]: four interleaved chains (p mod 4), eight loads in flight
__device__ __forceinline__ float sum4(const float *__restrict__ p0, long stride, int parts) {
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int p = 0;
#pragma unroll 1
    for (; p + 7 < parts; p += 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = p0[(long)(p + u) * stride];
        s0 += v[0]; s1 += v[1]; s2 += v[2]; s3 += v[3];
        s0 += v[4]; s1 += v[5]; s2 += v[6]; s3 += v[7];
    }
    for (; p + 3 < parts; p += 4) {
        s0 += p0[(long)p * stride];
        s1 += p0[(long)(p + 1) * stride];
        s2 += p0[(long)(p + 2) * stride];
        s3 += p0[(long)(p + 3) * stride];
    }
    if (p < parts) s0 += p0[(long)p * stride];
    if (p + 1 < parts) s1 += p0[(long)(p + 1) * stride];
    if (p + 2 < parts) s2 += p0[(long)(p + 2) * stride];
    return (s0 + s1) + (s2 + s3);
}

// one thread per output element; workgroup -> item through the start table (kernel argument, scalar loads)
__global__ __launch_bounds__(256) void reduce_items_kernel(const RedTable t) {
    const int b = blockIdx.x;
    int i = 0;
    while (i + 1 < t.n && b >= t.start[i + 1]) ++i;
    const sprk_reduce_item it = t.it[i];
    const long e = (long)(b - t.start[i]) * 256 + threadIdx.x;
    if (it.kind == SPRK_RED_ROWS) {
        if (e < it.n) it.dst[e] = sum4(it.src + e, it.n, it.parts);
    } else if (it.kind == SPRK_RED_COLS) {
        if (e < it.n) it.dst[e] = sum4(it.src + e * it.parts, 1, it.parts);
    } else {
        const long slab = (long)it.K * it.CoutP;
        if (e < slab) {
            const int co = (int)(e % it.CoutP), k = (int)(e / it.CoutP);
            if (co < it.Cout) it.dst[(long)co * it.K + k] = sum4(it.src + e, slab, it.parts);
        }
    }
}

}  // namespace

namespace sprk {

int reduce_items(const sprk_reduce_item *items, int n, hipStream_t s) {
    int i = 0;
    while (i < n) {
        RedTable t{};
        int blocks = 0;
        while (i < n && t.n < kRedMax) {
            const sprk_reduce_item &it = items[i++];
            if (it.kind == SPRK_RED_NONE) continue;
            if (!it.src || !it.dst || it.parts < 1 || it.n < 0 ||
                (it.kind != SPRK_RED_ROWS && it.kind != SPRK_RED_COLS && it.kind != SPRK_RED_WGRAD) ||
                (it.kind == SPRK_RED_WGRAD && (it.K < 1 || it.Cout < 1 || it.CoutP < it.Cout))) {
                set_error("reduce_items: bad item %d (kind %d)", i - 1, it.kind);
                return SPRK_EINVAL;
            }
            const long work = it.kind == SPRK_RED_WGRAD ? (long)it.K * it.CoutP : (long)it.n;
            if (work == 0) continue;
            t.it[t.n] = it;
            t.start[t.n] = blocks;
            blocks += cdiv(work, 256);
            ++t.n;
        }
        if (t.n == 0) continue;
        t.start[t.n] = blocks;
        hipLaunchKernelGGL(reduce_items_kernel, dim3(blocks), dim3(256), 0, s, t);
        if (int rc = check_launch("reduce_items")) return rc;
    }
    return SPRK_OK;
}

// ---- prepared weights (wprep_dev.h) -------------------------------------------------------------------------------
constexpr int kWprepMax = 40;   // items per launch: the table travels in the kernel argument (72 bytes each)
struct WprepTable {
    sprk_wprep_item it[kWprepMax];
    int start[kWprepMax + 1];
    int n;
};

__global__ __launch_bounds__(256) void wprep_kernel(const WprepTable t) {
    const int b = blockIdx.x;
    int i = 0;
    while (i + 1 < t.n && b >= t.start[i + 1]) ++i;
    const sprk_wprep_item &it = t.it[i];
    const long first = (long)(b - t.start[i]) * 256 + threadIdx.x, stride = (long)(t.start[i + 1] - t.start[i]) * 256;
    switch (it.kind) {
        case WPREP_DIRECT: wprep_direct(it.w, (float *)it.dst, it.p, first, stride); break;
        case WPREP_WINO: wprep_wino(it.w, (float *)it.dst, it.p, first, stride); break;
        case WPREP_BF16: wprep_16<__bf16>(it.w, (__bf16 *)it.dst, it.p, first, stride); break;
        case WPREP_F16: wprep_16<_Float16>(it.w, (_Float16 *)it.dst, it.p, first, stride); break;
        default: break;
    }
}

int wprep_launch(const sprk_wprep_item *items, int n, hipStream_t s) {
    for (int base = 0; base < n; base += kWprepMax) {
        WprepTable t;
        t.n = 0;
        int blocks = 0;
        for (int i = base; i < n && i < base + kWprepMax; ++i) {
            const sprk_wprep_item &it = items[i];
            if (it.kind == WPREP_NONE) continue;
            if (it.kind < 0 || it.kind > WPREP_F16 || !it.w || !it.dst || it.blocks <= 0) {
                set_error("prepare_weights: bad item %d", i);
                return SPRK_EINVAL;
            }
            t.it[t.n] = it;
            t.start[t.n] = blocks;
            blocks += it.blocks;
            ++t.n;
        }
        if (t.n == 0) continue;
        t.start[t.n] = blocks;
        hipLaunchKernelGGL(wprep_kernel, dim3(blocks), dim3(256), 0, s, t);
        if (int rc = check_launch("wprep")) return rc;
    }
    return SPRK_OK;
}

static thread_local sprk_wprep_item *t_wprep_describe = nullptr;
static thread_local bool t_wprep_skip = false;
static thread_local int t_wprep_expect = 0, t_wprep_seen = 0;
WprepScope::WprepScope(sprk_wprep_item *describe, int dtype) {
    t_wprep_describe = describe;
    t_wprep_skip = !describe && (dtype & SPRK_DT_WPREP);
    t_wprep_expect = t_wprep_skip ? SPRK_DT_WPREP_KIND_OF(dtype) : 0;
    t_wprep_seen = 0;
}
WprepScope::~WprepScope() {
    t_wprep_describe = nullptr;
    t_wprep_skip = false;
    t_wprep_expect = t_wprep_seen = 0;
}
// A SPRK_DT_WPREP call that says which transform its workspace holds (SPRK_DT_WPREP_KIND) must reach a site that
// would have produced exactly that kind: a path choice that differs from the describe call would read a slab in
// another layout.
int WprepScope::verify(int rc) const {
    if (rc != SPRK_OK || !t_wprep_skip || !t_wprep_expect || t_wprep_seen == t_wprep_expect) return rc;
    set_error("prepared weights: the workspace holds a transform of another kind than this call's kernel path reads");
    return SPRK_EINVAL;
}
bool wprep_describing() { return t_wprep_describe != nullptr; }
int wprep_site(const sprk_wprep_item &it, hipStream_t s) {
    if (t_wprep_describe) {
        *t_wprep_describe = it;
        return kWprepDescribed;
    }
    if (t_wprep_skip) {
        t_wprep_seen = it.kind;
        if (t_wprep_expect && it.kind != t_wprep_expect) {
            set_error("prepared weights: the workspace holds a transform of another kind than this call's kernel path reads");
            return SPRK_EINVAL;
        }
        return SPRK_OK;
    }
    return wprep_launch(&it, 1, s);
}

}  // namespace sprk

extern "C" {

int sprk_prepare_weights(const sprk_wprep_item *items, int n, void *stream) {
    SPRK_REQUIRE(n >= 0 && (n == 0 || items), "prepare_weights: bad arguments");
    return sprk::wprep_launch(items, n, (hipStream_t)stream);
}

int sprk_reduce_items(const sprk_reduce_item *items, int n, void *stream) {
    SPRK_REQUIRE(n >= 0 && (n == 0 || items), "reduce_items: bad arguments");
    return sprk::reduce_items(items, n, (hipStream_t)stream);
}

const char *sprk_last_error(void) { return sprk::t_error; }
int sprk_version(void) { return SPRK_ABI_VERSION; }
size_t sprk_struct_bytes(int which) {
    switch (which) {
    case 0: return sizeof(sprk_conv_geom);
    case 1: return sizeof(sprk_conv_epilogue);
    case 2: return sizeof(sprk_reduce_item);
    case 3: return sizeof(sprk_adam_item);
    case 4: return sizeof(sprk_wprep_item);
    default: return 0;
    }
}
long sprk_launch_count(void) { return sprk::g_launches.load(); }
long sprk_wino_launch_count(void) { return sprk::g_wino_launches.load(); }
long sprk_conv16_launch_count(void) { return sprk::conv16_launches() + sprk::wgrad16_launches(); }
long sprk_wgrad16_launch_count(void) { return sprk::wgrad16_launches(); }

void sprk_prof_enable(int on) {
    std::lock_guard<std::mutex> lk(sprk::g_prof_mu);
    sprk::g_prof_mask = on;
}

int sprk_prof_collect(int kclass, long *launches, double *ms, double *flops) {
    return sprk_prof_collect_bytes(kclass, launches, ms, flops, nullptr);
}

int sprk_prof_collect_bytes(int kclass, long *launches, double *ms, double *flops, double *bytes) {
    std::lock_guard<std::mutex> lk(sprk::g_prof_mu);
    long n = 0;
    double t = 0.0, f = 0.0, by = 0.0;
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
            by += r.bytes;
        }
        sprk::g_pool.push_back({r.a, r.b});
    }
    sprk::g_recs.swap(keep);
    if (launches) *launches = n;
    if (ms) *ms = t;
    if (flops) *flops = f;
    if (bytes) *bytes = by;
    return SPRK_OK;
}

}  // extern "C"
