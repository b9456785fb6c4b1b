// Weight transforms of the convolution kernels as "prepared-weight items" (sprk_wprep_item, include/sprk.h): every
// convolution path re-lays its weights out before its main kernel (k-chunked slabs for the implicit GEMM, G g G^T for
// Winograd, 16-bit slabs for the bf16 / fp16 kernels).  Each of those is a 5-microsecond launch in front of a kernel
// that depends on it, ~100 of them per training step (measured: 0.39 ms of an 18.8 ms fp32 step, 0.32 of 11.1 ms at
// batch 16).  Here the transforms are data: a call site fills an item, and ONE launch (wprep_kernel) runs any number of
// items — the per-call path launches it with one item, sprk_prepare_weights with all layers of a step.
#pragma once
#include <hip/hip_runtime.h>

#include "common.h"

namespace sprk {

enum { WPREP_NONE = 0, WPREP_DIRECT = 1, WPREP_WINO = 2, WPREP_BF16 = 3, WPREP_F16 = 4 };
constexpr int kWprepZeroFloats = 256;   // zero block in front of the direct kernels' slabs (DMA source of outside lanes)
constexpr int kWinoCK = 4;              // channels per Winograd K chunk (wino.hip CK)

// ---- bodies: grid-stride loops over the destination elements, `first` = this thread's first element, `stride` = threads
// direct implicit GEMM:  W[Cout][Cin][KHW] -> zeros[256] ++ Wt[nblk][rows][ldw]   (see conv.hip)
// p: Cout, Cin, KHW, mode, CK, R4, rows, NT16, ldw, nblk
__device__ __forceinline__ void wprep_direct(const float *__restrict__ w, float *__restrict__ ws, const int *p, long first,
                                             long stride) {
    const int Cout = p[0], Cin = p[1], KHW = p[2], mode = p[3], CK = p[4], R4 = p[5], rows = p[6], NT16 = p[7], ldw = p[8],
              nblk = p[9];
    const int Ck = mode == 0 ? Cin : Cout;   // channels along GEMM-k
    const int Nn = mode == 0 ? Cout : Cin;   // GEMM-n extent
    const long total = (long)nblk * rows * ldw + kWprepZeroFloats;
    for (long e = first; e < total; e += stride) {
        float v = 0.f;
        if (e >= kWprepZeroFloats) {
            long t = e - kWprepZeroFloats;
            const int col = (int)(t % ldw);
            t /= ldw;
            const int row = (int)(t % rows);
            const int nb = (int)(t / rows);
            const int n = nb * NT16 + col;
            const int q = row / R4, rr = row - q * R4;
            const int cke = min(CK, Ck - q * CK);
            if (col < NT16 && n < Nn && rr < cke * KHW) {
                const int tap = rr / cke, cl = rr - tap * cke;
                const int ck = q * CK + cl;
                if (mode == 0)
                    v = w[((long)n * Cin + ck) * KHW + tap];
                else
                    v = w[((long)ck * Cin + n) * KHW + (KHW - 1 - tap)];
            }
        }
        ws[e] = v;
    }
}

// Winograd: U[group][chunk][pos][nt][k][16] = (G g G^T)[pos]   (see wino.hip; even NT: [pos][nt / 2][k][16][2]); one
// thread per (group, chunk, nt, k, j)
// p: Cout, C1, C2, nc1, nch, NT, groups, mode
__device__ __forceinline__ void wprep_wino(const float *__restrict__ w, float *__restrict__ U, const int *p, long first,
                                           long stride) {
    const int Cout = p[0], C1 = p[1], C2 = p[2], nc1 = p[3], nch = p[4], NT = p[5], groups = p[6], mode = p[7];
    constexpr int CK = kWinoCK;
    const long total = (long)groups * nch * NT * CK * 16;
    for (long e = first; e < total; e += stride) {
        const int j = e % 16;
        long r = e / 16;
        const int k = r % CK;
        r /= CK;
        const int nt = r % NT;
        r /= NT;
        const int c = r % nch, grp = r / nch;
        const int co = (grp * NT + nt) * 16 + j;
        int ci = -1;
        if (c < nc1) {
            if (c * CK + k < C1) ci = c * CK + k;
        } else if ((c - nc1) * CK + k < C2) {
            ci = C1 + (c - nc1) * CK + k;
        }
        float g[3][3] = {};
        if (ci >= 0 && co < Cout) {
            const int Cin = C1 + C2;
            for (int t = 0; t < 9; ++t) {
                const int u = t / 3, v = t % 3;
                g[u][v] = mode == 0 ? w[((long)co * Cin + ci) * 9 + t] : w[((long)ci * Cout + co) * 9 + (2 - u) * 3 + (2 - v)];
            }
        }
        float t4[4][3];
        for (int v = 0; v < 3; ++v) {
            t4[0][v] = g[0][v];
            t4[1][v] = 0.5f * (g[0][v] + g[1][v] + g[2][v]);
            t4[2][v] = 0.5f * (g[0][v] - g[1][v] + g[2][v]);
            t4[3][v] = g[2][v];
        }
        // inside a position: [nt][k][16], or for an even NT [nt / 2][k][16][2] — a lane's two channel tiles side by side,
        // so that one 8-byte LDS read (two of them per ds_read2st64_b64) fetches both B operands
        const int in_pos = (NT & 1) ? (nt * CK + k) * 16 + j : (nt >> 1) * (2 * CK * 16) + (k * 16 + j) * 2 + (nt & 1);
        float *dst = U + ((long)grp * nch + c) * (16 * NT * CK * 16) + in_pos;
        for (int i = 0; i < 4; ++i) {
            const float u4[4] = {t4[i][0], 0.5f * (t4[i][0] + t4[i][1] + t4[i][2]), 0.5f * (t4[i][0] - t4[i][1] + t4[i][2]),
                                 t4[i][2]};
            for (int q = 0; q < 4; ++q) dst[(long)(i * 4 + q) * NT * CK * 16] = u4[q];
        }
    }
}

// 16-bit slabs: fp32 [Cout][Cin][KHW] -> [N-block][chunk][group][n][8] of T   (see conv16.hip)
// p: Cout, Cin, KHW, mode, CK, G4, NT16, nblk, nchunks
template <typename T>
__device__ __forceinline__ void wprep_16(const float *__restrict__ w, T *__restrict__ ws, const int *p, long first,
                                         long stride) {
    const int Cout = p[0], Cin = p[1], KHW = p[2], mode = p[3], CK = p[4], G4 = p[5], NT16 = p[6], nblk = p[7], nchunks = p[8];
    const int Ck = mode == 0 ? Cin : Cout;
    const int Nn = mode == 0 ? Cout : Cin;
    const long total = (long)nblk * nchunks * G4 * NT16 * 8;
    for (long e = first; e < total; e += stride) {
        long t = e;
        const int j = (int)(t & 7);
        t >>= 3;
        const int nl = (int)(t % NT16);
        t /= NT16;
        const int G = (int)(t % G4);
        t /= G4;
        const int q = (int)(t % nchunks);
        const int nb = (int)(t / nchunks);
        const int cke = min(CK, Ck - q * CK), c8 = (cke + 7) >> 3;
        const int n = nb * NT16 + nl;
        float v = 0.f;
        if (G < KHW * c8 && n < Nn) {
            const int tap = G / c8, cg = G - tap * c8;
            const int cl = cg * 8 + j;
            if (cl < cke) {
                const int ck = q * CK + cl;
                v = mode == 0 ? w[((long)n * Cin + ck) * KHW + tap] : w[((long)ck * Cin + n) * KHW + (KHW - 1 - tap)];
            }
        }
        ws[e] = (T)v;
    }
}

// ---- host side ---------------------------------------------------------------------------------------------------
// element counts -> grid size of an item (what the stand-alone kernels used: at most 4096 blocks of 256 threads)
inline sprk_wprep_item wprep_item(int kind, const float *w, void *dst, long elements, std::initializer_list<int> params) {
    sprk_wprep_item it{};
    it.w = w;
    it.dst = dst;
    it.kind = kind;
    int i = 0;
    for (int v : params) it.p[i++] = v;
    it.blocks = ew_blocks(elements);
    return it;
}

// A call site's weight transform.  Three behaviours, chosen by the entry point that is running on this thread:
//   describe (sprk_conv2d_*_wprep): the item is handed back and the call ends BEFORE any launch (returns kWprepDescribed);
//   skip (SPRK_DT_WPREP in the call's dtype): the workspace already holds the result — nothing to do;
//   otherwise: launched now, as a table of one item.
constexpr int kWprepDescribed = 1;
int wprep_site(const sprk_wprep_item &it, hipStream_t s);
int wprep_launch(const sprk_wprep_item *items, int n, hipStream_t s);
// RAII: the mode of the entry point running on this thread
struct WprepScope {
    WprepScope(sprk_wprep_item *describe, int dtype);      // dtype: the call's sprk_conv_geom.dtype (SPRK_DT_WPREP bits)
    ~WprepScope();
    int verify(int rc) const;                               // rc of the call -> rc, or SPRK_EINVAL on a kind mismatch
};
bool wprep_describing();

}  // namespace sprk
