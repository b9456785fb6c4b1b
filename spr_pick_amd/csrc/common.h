// Internal helpers shared by the libsprk.so translation units (not part of the ABI).
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>

#include "../../include/sprk.h"

namespace sprk {

void set_error(const char *fmt, ...);
extern std::atomic<long> g_launches, g_wino_launches;

// event bracketing of the MFMA convolution launches (sprk_prof_*)
void prof_begin(int kclass, double flops, hipStream_t s);
void prof_end(int kclass, hipStream_t s);
void prof_bytes(double bytes);   // algorithmic HBM bytes of the launch opened by prof_begin (HBM-bound kernel classes)

// Per-DEVICE facts and one-time set-up (a process may drive any device; nothing here is per process):
// compute units of the calling thread's current device (persistent kernels launch one workgroup per CU)
int num_cus();
// opt a kernel in to `bytes` of dynamic LDS on the current device (hipFuncSetAttribute once per device and function)
int lds_optin(const void *kernel, size_t bytes, const char *what);

inline int check_launch(const char *what) {
    g_launches.fetch_add(1, std::memory_order_relaxed);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: %s", what, hipGetErrorString(e));
        return SPRK_ELAUNCH;
    }
    return SPRK_OK;
}

// second-stage sums (sprk_reduce_items); item == nullptr in the callers below means "finish now"
int reduce_items(const sprk_reduce_item *items, int n, hipStream_t s);
inline int finish_or_defer(const sprk_reduce_item &it, sprk_reduce_item *out, hipStream_t s) {
    if (out) {
        *out = it;
        return SPRK_OK;
    }
    return it.kind == SPRK_RED_NONE ? (int)SPRK_OK : reduce_items(&it, 1, s);
}

inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }
inline int roundup(int a, int b) { return (a + b - 1) / b * b; }
inline int pow2_ceil(int v) {
    int p = 1;
    while (p < v) p <<= 1;
    return p;
}
inline int ilog2(int v) {
    int l = 0;
    while ((1 << l) < v) ++l;
    return l;
}

#define SPRK_REQUIRE(cond, ...)                \
    do {                                       \
        if (!(cond)) {                         \
            sprk::set_error(__VA_ARGS__);      \
            return SPRK_EINVAL;                \
        }                                      \
    } while (0)

// grid-stride launch size for HBM-bound elementwise kernels (guide: cap ~2048 blocks)
inline int ew_blocks(long n, int threads = 256) {
    long b = (n + threads - 1) / threads;
    if (b > 4096) b = 4096;
    if (b < 1) b = 1;
    return (int)b;
}

}  // namespace sprk

// XCD-aware start slot of a persistent workgroup.  Workgroups are dispatched round-robin over the 8 XCDs (id % 8), each
// with its own L2; work items that share input rows (spatially adjacent tiles / regions) should therefore go to
// workgroups of the SAME XCD.  With the workgroups walking items  slot + t * nwg  (t = 0, 1, ...), slot = xcd_slot(id)
// gives XCD k the contiguous items [k * nwg / 8, (k + 1) * nwg / 8) of every round instead of every 8th item.
__device__ __forceinline__ int xcd_slot(int id, int nwg, int on) {
    if (!on || (nwg & 7)) return id;
    return (id & 7) * (nwg >> 3) + (id >> 3);
}
namespace sprk {
// Timing experiments that switch phases of a kernel OFF (the results are then wrong on purpose: SPRK_WG_DIAG,
// SPRK_C16_DIAG, SPRK_NMS_DIAG, SPRK_WINO_DIAG).  They exist only in a library built with -DSPRK_DIAG
// (make DIAG=1); the shipped libsprk.so ignores the variables, so an inherited environment cannot corrupt a run.
inline int diag_env(const char *name) {
#ifdef SPRK_DIAG
    const char *v = getenv(name);
    return v ? atoi(v) : 0;
#else
    (void)name;
    return 0;
#endif
}
inline int xcd_on() {
    static const int on = getenv("SPRK_XCD") ? atoi(getenv("SPRK_XCD")) : 1;   // debug: 0 = plain round-robin order
    return on;
}
}  // namespace sprk

// exact floor(a / d) for 0 <= a < 2^21 via a float reciprocal (inv = 1.0f / d)
__device__ __forceinline__ int fast_div(int a, float inv) { return (int)(((float)a + 0.5f) * inv); }
