// Exact greedy 2-D non-maximum suppression, parallelised.
//
// The reference (utils/algorithms.py:59-103) walks pixels by descending score and keeps a set
// of suppressed flat indices.  Equivalent fixed point, evaluated in parallel:
//   priority(i) > priority(j)  <=>  score_i > score_j  or (score_i == score_j and i > j)
//   a candidate (score > threshold) is PICKED     iff every higher-priority pixel whose
//                                                  footprint covers it is SUPPRESSED;
//                             is SUPPRESSED  iff some higher-priority pixel whose footprint
//                                                  covers it is PICKED.
// Decisions are only ever taken from final (PICKED / SUPPRESSED) neighbour states, so stale
// reads merely postpone a decision: the result is the greedy result exactly, whatever the
// dispatch order.  Each launch relaxes every 32x32 tile to a local fixed point in LDS.
//
// Footprint of a pick (yy,xx): { clip(yy+di,0,H)*W + clip(xx+dj,0,W) : di^2+dj^2 <= r^2 }.
// Clipping to H / W (not H-1 / W-1) means an x overflow lands on column 0 of the NEXT row:
// pixel (y,0), y >= 1, is additionally covered by (yy,xx) when |y-1-yy| <= r and
// xx + dmax(|y-1-yy|) >= W.  y overflow lands past the array; underflow lands inside the disk.
#include "common.h"

namespace {

constexpr int TS = 32;            // tile side
constexpr int kBlk = 256;
constexpr int kMaxLdsR = 36;      // (32+2*36)^2 * 5 B = 54 KB
constexpr int kSortL = 2048;      // keys sorted per workgroup in LDS
enum : unsigned char { NONCAND = 0, UNDECIDED = 1, PICKED = 2, SUPPRESSED = 3 };

__global__ void nms_init_kernel(const float *__restrict__ sc, unsigned char *__restrict__ st, long n, float thr,
                                int *__restrict__ tile_und, int ntiles, int *__restrict__ counters) {
    const long g = (long)blockIdx.x * blockDim.x + threadIdx.x;
    for (long e = g; e < n; e += (long)gridDim.x * blockDim.x) st[e] = sc[e] > thr ? UNDECIDED : NONCAND;
    for (long e = g; e < ntiles; e += (long)gridDim.x * blockDim.x) tile_und[e] = 1;
    if (g < 4) counters[g] = 0;
}

struct NmsArgs {
    const float *sc;
    unsigned char *st;
    int *tile_und;
    int H, W, r, tilesX;
    int first;  // no pick exists yet anywhere: a blocked pixel may stop scanning early
};

__device__ __forceinline__ int disk_dmax(int r, int adi) {
    int d = (int)sqrtf((float)(r * r - adi * adi));
    while ((d + 1) * (d + 1) + adi * adi <= r * r) ++d;
    while (d * d + adi * adi > r * r) --d;
    return d;
}

template <bool USE_LDS>
__global__ __launch_bounds__(kBlk) void nms_round_kernel(const NmsArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    __shared__ int s_flag;
    __shared__ int s_dmax[128];
    const int tile = blockIdx.x;
    if (a.tile_und[tile] == 0) return;  // uniform: every pixel of this tile is decided
    const int tid = threadIdx.x;
    const int r = a.r, H = a.H, W = a.W;
    const int ty0 = (tile / a.tilesX) * TS, tx0 = (tile % a.tilesX) * TS;
    const int side = TS + 2 * r;
    float *F = reinterpret_cast<float *>(smem_raw);
    unsigned char *S = smem_raw + (USE_LDS ? (size_t)side * side * 4 : 0);

    for (int i = tid; i <= r && i < 128; i += kBlk) s_dmax[i] = disk_dmax(r, i);
    if (USE_LDS) {
        for (int e = tid; e < side * side; e += kBlk) {
            const int ly = e / side, lx = e - ly * side;
            const int y = ty0 - r + ly, x = tx0 - r + lx;
            float f = 0.f;
            unsigned char s = NONCAND;
            if ((unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W) {
                f = a.sc[(long)y * W + x];
                s = a.st[(long)y * W + x];
            }
            F[e] = f;
            S[e] = s;
        }
    }
    __syncthreads();

    // each thread owns 4 pixels of the 32x32 tile: rows (tid>>5) + 8k, column tid&31
    const int px = tid & 31;
    for (int iter = 0; iter < 64; ++iter) {
        if (tid == 0) s_flag = 0;
        __syncthreads();
        unsigned char nst[4];
        bool changed = false;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int py = (tid >> 5) + 8 * k;
            const int y = ty0 + py, x = tx0 + px;
            nst[k] = NONCAND;
            if (y >= H || x >= W) continue;
            const unsigned char cur = USE_LDS ? S[(py + r) * side + px + r] : a.st[(long)y * W + x];
            nst[k] = cur;
            if (cur != UNDECIDED) continue;
            const float my = USE_LDS ? F[(py + r) * side + px + r] : a.sc[(long)y * W + x];
            bool picked = false, blocked = false;
            for (int di = -r; di <= r && !picked; ++di) {
                const int yy = y + di;
                if (yy < 0 || yy >= H) continue;
                const int dm = s_dmax[di < 0 ? -di : di];
                for (int dj = -dm; dj <= dm; ++dj) {
                    const int xx = x + dj;
                    if (xx < 0 || xx >= W || (di == 0 && dj == 0)) continue;
                    const unsigned char s = USE_LDS ? S[(py + r + di) * side + px + r + dj] : a.st[(long)yy * W + xx];
                    if (s != UNDECIDED && s != PICKED) continue;
                    const float f = USE_LDS ? F[(py + r + di) * side + px + r + dj] : a.sc[(long)yy * W + xx];
                    const bool higher = f > my || (f == my && (di > 0 || (di == 0 && dj > 0)));
                    if (!higher) continue;
                    if (s == PICKED) {
                        picked = true;
                        break;
                    }
                    blocked = true;
                    if (a.first) break;
                }
                if (a.first && blocked) break;
            }
            // x-overflow wrap of picks near the right border onto column 0 of the next row
            if (!picked && x == 0 && y >= 1) {
                const long me = (long)y * W;
                for (int yy = max(0, y - 1 - r); yy <= min(H - 1, y - 1 + r) && !picked; ++yy) {
                    const int adi = yy > y - 1 ? yy - (y - 1) : (y - 1) - yy;
                    const int dm = s_dmax[adi];
                    for (int xx = max(0, W - dm); xx < W; ++xx) {  // xx + dm >= W
                        const long j = (long)yy * W + xx;
                        if (j == me) continue;
                        const unsigned char s = a.st[j];
                        if (s != UNDECIDED && s != PICKED) continue;
                        const float f = a.sc[j];
                        const bool higher = f > my || (f == my && j > me);
                        if (!higher) continue;
                        if (s == PICKED) {
                            picked = true;
                            break;
                        }
                        blocked = true;
                    }
                }
            }
            if (picked) {
                nst[k] = SUPPRESSED;
                changed = true;
            } else if (!blocked) {
                nst[k] = PICKED;
                changed = true;
            }
        }
        __syncthreads();  // all scans of this sweep are done before any state changes
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int py = (tid >> 5) + 8 * k;
            const int y = ty0 + py, x = tx0 + px;
            if (y >= H || x >= W) continue;
            if (USE_LDS) S[(py + r) * side + px + r] = nst[k];
            if (nst[k] == PICKED || nst[k] == SUPPRESSED) a.st[(long)y * W + x] = nst[k];
        }
        if (changed) s_flag = 1;
        __syncthreads();
        const int again = s_flag;
        __syncthreads();
        if (!again || !USE_LDS || a.first) break;
    }
    // remaining undecided pixels of this tile
    int und = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int py = (tid >> 5) + 8 * k;
        const int y = ty0 + py, x = tx0 + px;
        if (y >= H || x >= W) continue;
        const unsigned char cur = USE_LDS ? S[(py + r) * side + px + r] : a.st[(long)y * W + x];
        und += cur == UNDECIDED;
    }
    if (tid == 0) s_flag = 0;
    __syncthreads();
    if (und) atomicAdd(&s_flag, und);
    __syncthreads();
    if (tid == 0) a.tile_und[tile] = s_flag;
}

__device__ __forceinline__ unsigned ordered_bits(float f) {
    const unsigned b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float from_ordered(unsigned o) {
    const unsigned b = (o & 0x80000000u) ? (o & 0x7FFFFFFFu) : ~o;
    return __uint_as_float(b);
}

__global__ void nms_collect_kernel(const float *__restrict__ sc, const unsigned char *__restrict__ st, long n,
                                   unsigned long long *__restrict__ keys, long cap, int *__restrict__ counters,
                                   const int *__restrict__ tile_und, int ntiles) {
    const long g = (long)blockIdx.x * blockDim.x + threadIdx.x;
    for (long e = g; e < n; e += (long)gridDim.x * blockDim.x) {
        if (st[e] == PICKED) {
            const int pos = atomicAdd(&counters[0], 1);
            if (pos < cap) keys[pos] = ((unsigned long long)ordered_bits(sc[e]) << 32) | (unsigned)e;
        }
    }
    int und = 0;
    for (long e = g; e < ntiles; e += (long)gridDim.x * blockDim.x) und += tile_und[e];
    if (und) atomicAdd(&counters[1], und);
}

// ---- bitonic sort, descending, on `cap` (power of two >= kSortL) 64-bit keys -------------------
__device__ __forceinline__ void cmp_swap(unsigned long long &a, unsigned long long &b, bool desc) {
    if ((a < b) == desc) {
        const unsigned long long t = a;
        a = b;
        b = t;
    }
}

// mode 0: full sort of each kSortL chunk; mode 1: merge strides kSortL/2..1 of stage `size`
__global__ __launch_bounds__(kBlk) void bitonic_local_kernel(unsigned long long *keys, long size_stage, int mode) {
    __shared__ unsigned long long s[kSortL];
    const long base = (long)blockIdx.x * kSortL;
    for (int i = threadIdx.x; i < kSortL; i += kBlk) s[i] = keys[base + i];
    __syncthreads();
    const long first_size = mode == 0 ? 2 : size_stage;
    const long last_size = mode == 0 ? kSortL : size_stage;
    for (long size = first_size; size <= last_size; size <<= 1) {
        for (int stride = (int)((size >> 1) < kSortL ? (size >> 1) : kSortL / 2); stride > 0; stride >>= 1) {
            for (int t = threadIdx.x; t < kSortL / 2; t += kBlk) {
                const int i = ((t / stride) * 2 * stride) + (t % stride);
                const int j = i + stride;
                const bool desc = ((base + i) & size) == 0;
                cmp_swap(s[i], s[j], desc);
            }
            __syncthreads();
        }
    }
    for (int i = threadIdx.x; i < kSortL; i += kBlk) keys[base + i] = s[i];
}

__global__ void bitonic_global_kernel(unsigned long long *keys, long cap, long size, long stride) {
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < cap / 2; t += (long)gridDim.x * blockDim.x) {
        const long i = ((t / stride) * 2 * stride) + (t % stride);
        const long j = i + stride;
        const bool desc = (i & size) == 0;
        unsigned long long a = keys[i], b = keys[j];
        if ((a < b) == desc) {
            keys[i] = b;
            keys[j] = a;
        }
    }
}

__global__ void nms_emit_kernel(const unsigned long long *__restrict__ keys, const int *__restrict__ counters,
                                float *__restrict__ out_scores, int *__restrict__ out_xy, int *__restrict__ out_count,
                                int max_out, long cap, int W) {
    const int n = counters[0];
    const long lim = min((long)min(n, max_out), cap);
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < lim; e += (long)gridDim.x * blockDim.x) {
        const unsigned long long k = keys[e];
        const unsigned idx = (unsigned)(k & 0xFFFFFFFFull);
        out_scores[e] = from_ordered((unsigned)(k >> 32));
        out_xy[2 * e] = (int)(idx % (unsigned)W);
        out_xy[2 * e + 1] = (int)(idx / (unsigned)W);
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        out_count[0] = n;
        out_count[1] = counters[1];
    }
}

struct NmsWs {
    size_t off_state, off_tiles, off_counters, off_keys, total;
    long cap;
    int ntiles, tilesX;
};

NmsWs nms_layout(int H, int W, int max_out) {
    NmsWs w;
    w.tilesX = sprk::cdiv(W, TS);
    w.ntiles = w.tilesX * sprk::cdiv(H, TS);
    long cap = kSortL;
    while (cap < max_out) cap <<= 1;
    w.cap = cap;
    size_t o = 0;
    w.off_state = o;
    o += ((size_t)H * W + 255) / 256 * 256;
    w.off_tiles = o;
    o += ((size_t)w.ntiles * 4 + 255) / 256 * 256;
    w.off_counters = o;
    o += 256;
    w.off_keys = o;
    o += (size_t)cap * 8;
    w.total = o;
    return w;
}

}  // namespace

extern "C" {

size_t sprk_nms2d_ws_bytes(int H, int W, int max_out) {
    if (H <= 0 || W <= 0 || max_out <= 0) return 0;
    return nms_layout(H, W, max_out).total;
}

int sprk_nms2d(const float *scores, int H, int W, int r, float threshold, float *out_scores, int32_t *out_xy,
               int32_t *out_count, int max_out, int rounds, int resume, void *ws, size_t ws_bytes, void *stream) {
    SPRK_REQUIRE(scores && out_scores && out_xy && out_count, "nms2d: null pointer");
    SPRK_REQUIRE(H > 0 && W > 0 && (long)H * W < (1L << 31), "nms2d: bad map size");
    SPRK_REQUIRE(r >= 0 && r < 128 && max_out > 0 && rounds >= 0, "nms2d: bad parameters");
    const NmsWs L = nms_layout(H, W, max_out);
    if (!ws || ws_bytes < L.total) {
        sprk::set_error("nms2d: workspace %zu < %zu", ws_bytes, L.total);
        return SPRK_EWORKSPACE;
    }
    hipStream_t s = (hipStream_t)stream;
    unsigned char *base = (unsigned char *)ws;
    unsigned char *st = base + L.off_state;
    int *tile_und = (int *)(base + L.off_tiles);
    int *counters = (int *)(base + L.off_counters);
    unsigned long long *keys = (unsigned long long *)(base + L.off_keys);
    const long n = (long)H * W;
    if (!resume) {
        hipLaunchKernelGGL(nms_init_kernel, dim3(sprk::ew_blocks(n)), dim3(256), 0, s, scores, st, n, threshold,
                           tile_und, L.ntiles, counters);
        if (int rc = sprk::check_launch("nms_init")) return rc;
    } else {
        if (hipMemsetAsync(counters, 0, 16, s) != hipSuccess) {
            sprk::set_error("nms2d: memset failed");
            return SPRK_ELAUNCH;
        }
    }
    NmsArgs a{scores, st, tile_und, H, W, r, L.tilesX, 0};
    const bool lds = r <= kMaxLdsR;
    const int side = TS + 2 * r;
    const size_t shm = lds ? (size_t)side * side * 5 : 0;
    for (int it = 0; it < rounds; ++it) {
        a.first = (!resume && it == 0) ? 1 : 0;
        if (lds)
            hipLaunchKernelGGL(nms_round_kernel<true>, dim3(L.ntiles), dim3(kBlk), shm, s, a);
        else
            hipLaunchKernelGGL(nms_round_kernel<false>, dim3(L.ntiles), dim3(kBlk), 0, s, a);
        if (int rc = sprk::check_launch("nms_round")) return rc;
    }
    if (hipMemsetAsync(keys, 0, (size_t)L.cap * 8, s) != hipSuccess) {
        sprk::set_error("nms2d: memset failed");
        return SPRK_ELAUNCH;
    }
    hipLaunchKernelGGL(nms_collect_kernel, dim3(sprk::ew_blocks(n)), dim3(256), 0, s, scores, st, n, keys, L.cap,
                       counters, tile_und, L.ntiles);
    if (int rc = sprk::check_launch("nms_collect")) return rc;
    const int nchunks = (int)(L.cap / kSortL);
    hipLaunchKernelGGL(bitonic_local_kernel, dim3(nchunks), dim3(kBlk), 0, s, keys, 0L, 0);
    if (int rc = sprk::check_launch("bitonic_local")) return rc;
    for (long size = 2L * kSortL; size <= L.cap; size <<= 1) {
        for (long stride = size >> 1; stride >= kSortL; stride >>= 1) {
            hipLaunchKernelGGL(bitonic_global_kernel, dim3(sprk::ew_blocks(L.cap / 2)), dim3(256), 0, s, keys, L.cap,
                               size, stride);
            if (int rc = sprk::check_launch("bitonic_global")) return rc;
        }
        hipLaunchKernelGGL(bitonic_local_kernel, dim3(nchunks), dim3(kBlk), 0, s, keys, size, 1);
        if (int rc = sprk::check_launch("bitonic_local_merge")) return rc;
    }
    hipLaunchKernelGGL(nms_emit_kernel, dim3(sprk::ew_blocks(std::min<long>(L.cap, max_out))), dim3(256), 0, s, keys,
                       counters, out_scores, out_xy, out_count, max_out, L.cap, W);
    return sprk::check_launch("nms_emit");
}

}  // extern "C"
