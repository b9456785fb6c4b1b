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
    if (g < 8) counters[g] = 0;
}

struct NmsArgs {
    const float *sc;
    unsigned char *st;
    int *tile_und;
    unsigned long long *keys;   // picks in discovery order: (ordered score bits << 32) | flat index
    int *counters;              // [0] picks so far, [1] undecided (final), [2] start / [3] end of the last round's picks
    long cap;
    int H, W, r, tilesX, diag;
};

__device__ __forceinline__ int disk_dmax(int r, int adi) {
    int d = (int)sqrtf((float)(r * r - adi * adi));
    while ((d + 1) * (d + 1) + adi * adi <= r * r) ++d;
    while (d * d + adi * adi > r * r) --d;
    return d;
}

__device__ __forceinline__ unsigned ordered_bits(float f);

// One relaxation sweep.  An UNDECIDED pixel scans its potential suppressors (the disk, plus for
// column 0 the right-border pixels whose footprint wraps onto it) and stops at the first one of
// higher priority that is UNDECIDED (-> it stays undecided this round) or PICKED (-> it is
// SUPPRESSED, final).  If there is none it is PICKED (final) and appended to the pick list; the
// companion kernel then marks the footprints of the round's new picks, so that pixels blocked only
// by soon-to-be-suppressed neighbours are released in the next sweep.
template <bool USE_LDS>
__global__ __launch_bounds__(kBlk) void nms_round_kernel(const NmsArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    __shared__ int s_und;
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
    if (tid == 0) s_und = 0;
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
    if (a.diag & 1) return;   // timing experiment: the LDS fill alone

    // each thread owns 4 pixels of the 32x32 tile: rows (tid>>5) + 8k, column tid&31
    const int px = tid & 31;
    int und = 0;
    const int lane = tid & 63, wave = tid >> 6;
    const int span = 2 * r + 1, nbox = span * span;
    const float inv_span = 1.0f / (float)span;
#pragma unroll 1
    for (int k = 0; k < 4; ++k) {
        const int py = (tid >> 5) + 8 * k;
        const int y = ty0 + py, x = tx0 + px;
        const bool inside = y < H && x < W;
        const unsigned char cur = !inside ? (unsigned char)NONCAND : USE_LDS ? S[(py + r) * side + px + r] : a.st[(long)y * W + x];
        const bool active = cur == UNDECIDED;
        const float my = !active ? 0.f : USE_LDS ? F[(py + r) * side + px + r] : a.sc[(long)y * W + x];
        int verdict = 0;  // 0 none found, 1 blocked by an undecided, 2 covered by a pick
        // Quick probe: the 8 immediate neighbours (all inside the disk for r >= 2).  On a score map — smooth at the scale
        // of a pixel — every pixel that is not a local maximum has a higher neighbour right next to it, so this settles
        // it in <= 8 probes (which neighbour blocks a pixel is irrelevant: "blocked" only postpones, see the header).
        if (USE_LDS && r >= 2 && active) {
#pragma unroll
            for (int q = 0; q < 8 && !verdict; ++q) {
                const int di = q < 3 ? -1 : q < 5 ? 0 : 1;
                const int dj = q < 3 ? q - 1 : q == 3 ? -1 : q == 4 ? 1 : q - 6;
                const int yy = y + di, xx = x + dj;
                if (yy < 0 || yy >= H || xx < 0 || xx >= W) continue;
                const unsigned char s = S[(py + r + di) * side + px + r + dj];
                if (s != UNDECIDED && s != PICKED) continue;
                const float f = F[(py + r + di) * side + px + r + dj];
                if (f > my || (f == my && (di > 0 || (di == 0 && dj > 0)))) verdict = s == PICKED ? 2 : 1;
            }
        }
        if ((a.diag & 8) && !verdict) verdict = 1;   // timing experiment: quick probe only, survivors stay undecided
        if (USE_LDS && r >= 2) {
            // The survivors (local maxima of their 3x3: ~10 % of the pixels of a noisy map) have to look at the whole
            // disk.  One lane walking 1009 positions while 63 wait was 3/4 of a round's time; instead the WAVE scans
            // each survivor's disk together: the leader's pixel travels by lane read (wavefront shuffle), every lane
            // checks one position of the bounding box per step, rows from the centre outwards, and the 64 findings are
            // combined by ballot.  "Covered by a higher-priority pick" wins over "blocked by an undecided one" — both are
            // statements the fixed point allows (header), the former is final.
            unsigned long long todo = __ballot(active && verdict == 0 && !(a.diag & 2));
            while (todo) {
                const int leader = __ffsll((long long)todo) - 1;
                todo &= todo - 1;
                const int ltid = wave * 64 + leader;
                const int lpy = (ltid >> 5) + 8 * k, lpx = ltid & 31;
                const float lmy = __shfl(my, leader, 64);
                const int ly = ty0 + lpy, lx = tx0 + lpx;
                int v = 0;
                for (int t0 = 0; t0 < nbox && !v; t0 += 64) {
                    const int t = t0 + lane;
                    int found = 0;
                    if (t < nbox) {
                        const int rowj = (int)(((float)t + 0.5f) * inv_span), col = t - rowj * span;
                        const int di = (rowj & 1) ? -((rowj + 1) >> 1) : (rowj >> 1), dj = col - r;
                        const int adj = dj < 0 ? -dj : dj;
                        const int yy = ly + di, xx = lx + dj;
                        if (adj <= s_dmax[di < 0 ? -di : di] && (di | dj) != 0 && yy >= 0 && yy < H && xx >= 0 && xx < W) {
                            const int o = (lpy + r + di) * side + lpx + r + dj;
                            const unsigned char s = S[o];
                            if (s == UNDECIDED || s == PICKED) {
                                const float f = F[o];
                                if (f > lmy || (f == lmy && (di > 0 || (di == 0 && dj > 0)))) found = s == PICKED ? 2 : 1;
                            }
                        }
                    }
                    v = __ballot(found == 2) ? 2 : __ballot(found == 1) ? 1 : 0;
                }
                if (lane == leader) verdict = v;
            }
        } else {
            for (int di = -r; di <= r && !verdict && active && !(a.diag & 2); ++di) {
                const int yy = y + di;
                if (yy < 0 || yy >= H) continue;
                const int dm = s_dmax[di < 0 ? -di : di];
                for (int dj = -dm; dj <= dm; ++dj) {
                    const int xx = x + dj;
                    if (xx < 0 || xx >= W || (di == 0 && dj == 0)) continue;
                    const unsigned char s = USE_LDS ? S[(py + r + di) * side + px + r + dj] : a.st[(long)yy * W + xx];
                    if (s != UNDECIDED && s != PICKED) continue;
                    const float f = USE_LDS ? F[(py + r + di) * side + px + r + dj] : a.sc[(long)yy * W + xx];
                    if (f > my || (f == my && (di > 0 || (di == 0 && dj > 0)))) {
                        verdict = s == PICKED ? 2 : 1;
                        break;
                    }
                }
            }
        }
        if (!active) continue;
        // x-overflow wrap of picks near the right border onto column 0 of the next row
        if (!verdict && x == 0 && y >= 1) {
            const long me = (long)y * W;
            for (int yy = max(0, y - 1 - r); yy <= min(H - 1, y - 1 + r) && !verdict; ++yy) {
                const int adi = yy > y - 1 ? yy - (y - 1) : (y - 1) - yy;
                const int dm = s_dmax[adi];
                for (int xx = max(0, W - dm); xx < W; ++xx) {  // xx + dm >= W
                    const long j = (long)yy * W + xx;
                    if (j == me) continue;
                    const unsigned char s = a.st[j];
                    if (s != UNDECIDED && s != PICKED) continue;
                    const float f = a.sc[j];
                    if (f > my || (f == my && j > me)) {
                        verdict = s == PICKED ? 2 : 1;
                        break;
                    }
                }
            }
        }
        if (verdict == 2) {
            a.st[(long)y * W + x] = SUPPRESSED;
        } else if (verdict == 0) {
            a.st[(long)y * W + x] = PICKED;
            const int pos = atomicAdd(&a.counters[0], 1);
            if (pos < a.cap) a.keys[pos] = ((unsigned long long)ordered_bits(my) << 32) | (unsigned)((long)y * W + x);
        } else {
            ++und;
        }
    }
    if (und) atomicAdd(&s_und, und);
    __syncthreads();
    if (tid == 0) a.tile_und[tile] = s_und;
}

// counters[2..3] = [start, end) of the picks appended by the sweep that just ran
__global__ void nms_snapshot_kernel(int *counters) {
    counters[2] = counters[3];
    counters[3] = counters[0];
}

// footprint of every new pick: UNDECIDED pixels of LOWER priority become SUPPRESSED (a pick only
// suppresses what the greedy walk visits after it); the footprint is the clipped disk plus, when the
// disk overflows the right border, column 0 of the following row (the reference's clip-to-W quirk)
__global__ __launch_bounds__(kBlk) void nms_suppress_kernel(const NmsArgs a) {
    __shared__ int s_dmax[128];
    const int r = a.r, H = a.H, W = a.W;
    for (int i = threadIdx.x; i <= r && i < 128; i += kBlk) s_dmax[i] = disk_dmax(r, i);
    __syncthreads();
    const long start = a.counters[2], end = min((long)a.counters[3], a.cap);
    const int span = 2 * r + 1;
    for (long p = start + blockIdx.x; p < end; p += gridDim.x) {
        const unsigned long long key = a.keys[p];
        const unsigned idx = (unsigned)(key & 0xFFFFFFFFull);
        const int yy = (int)(idx / (unsigned)W), xx = (int)(idx % (unsigned)W);
        for (int e = threadIdx.x; e < span * (span + 1); e += kBlk) {
            const int di = e / (span + 1) - r, c = e % (span + 1);
            const int dm = s_dmax[di < 0 ? -di : di];
            int ty, tx;
            if (c < span) {  // disk element
                const int dj = c - r;
                if (dj < -dm || dj > dm) continue;
                ty = yy + di;
                tx = xx + dj;
                if (ty < 0 || ty >= H || tx < 0 || tx >= W) continue;
            } else {         // wrap element of this row
                if (xx + dm < W) continue;
                ty = min(max(yy + di, 0), H) + 1;
                tx = 0;
                if (ty >= H) continue;
            }
            const long t = (long)ty * W + tx;
            if (t == (long)idx || a.st[t] != UNDECIDED) continue;
            const unsigned long long tk = ((unsigned long long)ordered_bits(a.sc[t]) << 32) | (unsigned)t;
            if (tk < key) a.st[t] = SUPPRESSED;
        }
    }
}

__device__ __forceinline__ unsigned ordered_bits(float f) {
    const unsigned b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float from_ordered(unsigned o) {
    const unsigned b = (o & 0x80000000u) ? (o & 0x7FFFFFFFu) : ~o;
    return __uint_as_float(b);
}

__global__ void nms_count_kernel(int *__restrict__ counters, const unsigned char *__restrict__ st, long n) {
    int und = 0;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long)gridDim.x * blockDim.x)
        und += st[e] == UNDECIDED;
    if (und) atomicAdd(&counters[1], und);
}

// copy the discovery-order pick list into the sort buffer, zero padded to `cap`
__global__ void nms_fill_sort_kernel(const unsigned long long *__restrict__ picks, unsigned long long *__restrict__ keys,
                                     const int *__restrict__ counters, long cap) {
    const long n = min((long)counters[0], cap);
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < cap; e += (long)gridDim.x * blockDim.x)
        keys[e] = e < n ? picks[e] : 0ull;
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
    size_t off_state, off_tiles, off_counters, off_keys, off_picks, total;
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
    w.off_picks = o;
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
    unsigned long long *picks = (unsigned long long *)(base + L.off_picks);
    const long n = (long)H * W;
    if (!resume) {
        hipLaunchKernelGGL(nms_init_kernel, dim3(sprk::ew_blocks(n)), dim3(256), 0, s, scores, st, n, threshold,
                           tile_und, L.ntiles, counters);
        if (int rc = sprk::check_launch("nms_init")) return rc;
    } else {
        if (hipMemsetAsync(counters + 1, 0, 4, s) != hipSuccess) {
            sprk::set_error("nms2d: memset failed");
            return SPRK_ELAUNCH;
        }
    }
    static const int nms_diag = sprk::diag_env("SPRK_NMS_DIAG");   // timing experiments only
    NmsArgs a{scores, st, tile_und, picks, counters, L.cap, H, W, r, L.tilesX, nms_diag};
    const bool lds = r <= kMaxLdsR;
    const int side = TS + 2 * r;
    const size_t shm = lds ? (size_t)side * side * 5 : 0;
    for (int it = 0; it < rounds; ++it) {
        if (lds)
            hipLaunchKernelGGL(nms_round_kernel<true>, dim3(L.ntiles), dim3(kBlk), shm, s, a);
        else
            hipLaunchKernelGGL(nms_round_kernel<false>, dim3(L.ntiles), dim3(kBlk), 0, s, a);
        if (int rc = sprk::check_launch("nms_round")) return rc;
        hipLaunchKernelGGL(nms_snapshot_kernel, dim3(1), dim3(1), 0, s, counters);
        hipLaunchKernelGGL(nms_suppress_kernel, dim3(512), dim3(kBlk), 0, s, a);
        if (int rc = sprk::check_launch("nms_suppress")) return rc;
    }
    hipLaunchKernelGGL(nms_count_kernel, dim3(sprk::ew_blocks(n)), dim3(256), 0, s, counters, st, n);
    hipLaunchKernelGGL(nms_fill_sort_kernel, dim3(sprk::ew_blocks(L.cap)), dim3(256), 0, s, picks, keys, counters, L.cap);
    if (int rc = sprk::check_launch("nms_fill_sort")) return rc;
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
