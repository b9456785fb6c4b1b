// Convolution kernels of libsprk.so (gfx950 / MI355X).
//
// All convolutions of the hot path are implicit GEMMs on the fp32 matrix cores
// (v_mfma_f32_16x16x4_f32: exact fp32, k-ordered fma chain) over NCHW tensors:
//
//   forward / backward-data :  Out[pixel][cout] += A[pixel][k] * Wt[k][cout],  k = (tap, cin)
//        A-fragment lane l -> pixel l&15, k l>>4 : 16 consecutive pixels of a row, straight
//        from an LDS copy of the input tile (with halo, zero-filled outside the image), so
//        the im2col matrix is never formed;  B-fragment from an LDS copy of the
//        pre-transposed weight rows.
//   backward-weight         :  dW[k][cout] += X[k][pixel] * G[pixel][cout]  (pixels are the
//        reduced dimension; per-block partials + a deterministic reduction pass).
//
// LDS plane strides are chosen per kernel so that the two k-values (fwd) / two pixels (bwd-w)
// held by one 32-lane ds_read group fall on disjoint banks (MI355X_MICROARCH.md §LDS).
// The "direct" kernels at the end are the plain per-output-element statement of the same
// maths: they serve strided backward-data (tiny detector layers) and as an on-device
// cross-check (SPRK_DT_NAIVE in sprk_conv_geom.dtype).
#include "common.h"
#include "conv16.h"
#include "wgrad16.h"
#include "wino.h"
#include "wprep_dev.h"

#include <cstdlib>

#include "conv_dev.h"
#include "io_dev.h"


namespace {

// per-call switch: SPRK_DT_NAIVE in sprk_conv_geom.dtype routes the call to the plain (non-MFMA) kernels
inline bool naive_of(const sprk_conv_geom *g) { return (g->dtype & SPRK_DT_NAIVE) != 0; }

constexpr int kClass16 = 5;   // profiling class of the 16-bit-operand forward / backward-data kernel (conv16.hip)
constexpr int kClass16W = 6;  // ... of the 16-bit-operand backward-weight kernel (wgrad16.hip)

// ------------------------------------------------------------------------------------------
// weight transform:  W[Cout][Cin][KHW]  ->  Wt[nblk][rows][ldw]   (rows = k in chunked order)
//   mode 0 (forward):        GEMM-k channel = cin,  n = cout, tap as is
//   mode 1 (backward-data):  GEMM-k channel = cout, n = cin,  tap flipped
// Row order: chunks of CK k-channels; inside a chunk row = tap * cke + cl (cke = channels
// actually present in the chunk); every chunk occupies R4 = roundup(CK*KHW, 4) rows (16 for MT = 1).
// Each N-block (NT*16 output channels) is stored as its own [rows][ldw] slab with the LDS row
// stride ldw (zero padded), so a chunk of a slab is one contiguous run that LDS-DMA copies 1:1.
// The first 256 floats of the workspace are zeroed: the DMA source of out-of-image lanes.
// ------------------------------------------------------------------------------------------
constexpr int kZeroFloats = 256;
__device__ __attribute__((aligned(64))) float g_zero_block[kZeroFloats];  // zero-initialised; source of out-of-image DMA lanes

// nch channel planes from one source tensor: `base` points at channel 0 of image 0 of the staged
// channel range; imgStride / cs are the image / channel strides in floats; (Hs, Ws) the source plane
// size; up: source is half resolution, nearest-upsampled on load.
__device__ __forceinline__ void stage_planes(float *dst, const float *base, long imgStride, long cs, int nch, int N,
                                             int Hin, int Win, int Ws, int up, int vec, const float *zeros,
                                             const PlaneGeom &g, int n0, int iy0, int ixa, int lw, int lane) {
    const int imgElems = g.inRows * g.pitch;
    const int planeElems = g.NI * imgElems;
    const int per = vec ? 256 : 64;
    const int nGroups = (planeElems + per - 1) / per;
    // work items (group, channel) are dealt round-robin to the issuing waves: every wave walks all
    // groups but only every kIssuers-th channel, so the issue load is even whatever nGroups is
    const int nw = g.nw;
    const int gstart = g.deal ? 0 : lw, gstep = g.deal ? 1 : nw, cstep = g.deal ? nw : 1;
    for (int gi = gstart; gi < nGroups; gi += gstep) {
        const int e = vec ? gi * 256 + lane * 4 : gi * 64 + lane;
        const int il = fast_div(e, g.invImg);
        const int rem = e - il * imgElems;
        const int r = fast_div(rem, g.invPitch);
        const int j = rem - r * g.pitch;
        const int n = n0 + il, iy = iy0 + r, ix = ixa + j;
        const bool inb = e < planeElems;
        const bool ok = inb && n < N && (unsigned)iy < (unsigned)Hin && (unsigned)ix < (unsigned)Win;
        const int cl0 = g.deal ? (lw + nw - (gi % nw)) % nw : 0;
        const float *p = base + (long)n * imgStride + (up ? (long)(iy >> 1) * Ws + (ix >> 1) : (long)iy * Ws + ix) +
                         (long)cl0 * cs;
        long step = (long)cstep * cs;
        if (!ok) {  // lanes outside the image read the zero block
            p = zeros + lane * 4;
            step = 0;
        }
        float *d = dst + gi * per + cl0 * g.cplane;
        for (int cl = cl0; cl < nch; cl += cstep) {
            if (inb) {
                if (vec)
                    dma16(p, d);
                else
                    dma4(p, d);
            }
            p += step;
            d += cstep * g.cplane;
        }
    }
}

// Source of the (virtual) conv input: channel concat of x (optionally 2x nearest-upsampled on the
// fly) and x2, zero outside the image.
struct TileSrc {
    const float *x, *x2, *zeros;
    int N, C1, C2, Hin, Win, up1, H1, W1;
    int vec1, vec2;  // 16-byte DMA allowed for source 1 / source 2
};

// channels [c0, c0+cke) of the concatenated input
__device__ __forceinline__ void stage_input_dma(float *in_lds, const TileSrc &s, const PlaneGeom &g, int n0, int iy0,
                                                int ix0, int c0, int cke, int lw, int lane) {
    const int ixa = ix0 - g.colOff;
    const int n1 = max(0, min(c0 + cke, s.C1) - c0);  // channels taken from x
    if (n1 > 0) {
        const long cs1 = (long)s.H1 * s.W1;
        stage_planes(in_lds, s.x + (long)c0 * cs1, (long)s.C1 * cs1, cs1, n1, s.N, s.Hin, s.Win, s.W1, s.up1, s.vec1,
                     s.zeros, g, n0, iy0, ixa, lw, lane);
    }
    if (n1 < cke) {
        const long cs2 = (long)s.Hin * s.Win;
        const int cb = max(c0, s.C1) - s.C1;
        stage_planes(in_lds + n1 * g.cplane, s.x2 + (long)cb * cs2, (long)s.C2 * cs2, cs2, cke - n1, s.N, s.Hin, s.Win,
                     s.Win, 0, s.vec2, s.zeros, g, n0, iy0, ixa, lw, lane);
    }
}

// The K loop of one staged chunk.  Non-MFMA VALU instructions take issue time from the MFMA stream (about a
// quarter of an fp32 16x16x4 MFMA each), so a k-step is written to need few of them: all operand addresses
// are absolute LDS byte addresses; the RB row bases of this wave's MT pixel tiles take one add each (tiles
// of one row are 16 floats apart: immediates), the weight row one add (cout tiles: immediates), the k-row
// offset one add for its table address.  The operands of step kq+1 are fetched under the MFMAs of step kq.
//   abase[r]: address of this lane's pixel of row base r at channel 0, tap 0 of the stage
//   kaddr:    address of koff[lq] (byte offsets of k rows 4kq+lq);   baddr: address of w[lq][l15]
template <int MT, int NT, int RB>
__device__ __forceinline__ void chunk_mma(f32x4 (&acc)[MT][NT], const int (&abase)[RB], int kaddr, int baddr,
                                          int bstep, int nkq) {
    constexpr int MPR = MT / RB;
    float av[MT], bv[NT];
    {
        const int ko = *lds_i(kaddr);
#pragma unroll
        for (int r = 0; r < RB; ++r) {
            lds_cfp ap = lds_f(abase[r] + ko);
#pragma unroll
            for (int j = 0; j < MPR; ++j) av[r * MPR + j] = ap[16 * j];
        }
        lds_cfp bp = lds_f(baddr);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) bv[nt] = bp[16 * nt];
    }
    const int last = nkq - 1;
    for (int kq = 0; kq < nkq; ++kq) {
        const int kn = min(kq + 1, last);   // the last step re-reads itself (unused)
        const int ko = lds_i(kaddr)[kn * 4];
        float an[MT], bn[NT];
#pragma unroll
        for (int r = 0; r < RB; ++r) {
            lds_cfp ap = lds_f(abase[r] + ko);
#pragma unroll
            for (int j = 0; j < MPR; ++j) an[r * MPR + j] = ap[16 * j];
        }
        lds_cfp bp = lds_f(baddr + kn * bstep);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) bn[nt] = bp[16 * nt];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[mt], bv[nt], acc[mt][nt], 0, 0, 0);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) av[mt] = an[mt];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) bv[nt] = bn[nt];
    }
}

// MT = 1, NT <= 3 (planes of a few pixels: the U-Net levels <= 16x16 and the detector's patch layers): one pixel tile per wave
// means 1..6 MFMAs per k-step, far less than the two LDS round trips (k-row offset -> operand) a step of chunk_mma
// exposes, and these launches are latency-bound (8..512 workgroups, each alone on its CU).  Here a wave walks the
// chunk in groups of U k-steps: the offsets of group g+2 and the operands of group g+1 are in flight under the MFMAs
// of group g (U = 4, or 2 with three cout tiles), and with one or two cout tiles the steps of a group alternate
// between two accumulator sets (half the dependent-MFMA chain).  nkq is a multiple of U (the plan rounds a chunk's
// rows to 16: zero weight rows).
template <int NT>
__device__ __forceinline__ void chunk_mma_small(f32x4 (&acc)[1][NT], int abase, int kaddr, int baddr, int bstep, int nkq) {
    static_assert(NT <= 3, "four or more cout tiles: chunk_mma hides the latencies under its MFMAs (measured: this form is slower)");
    constexpr int U = NT <= 2 ? 4 : 2;
    constexpr bool TWO = NT <= 2;
    const int ng = nkq / U, lastg = ng - 1;
    int ko[U];
    float av[U], bv[U][NT];
    f32x4 acc2[TWO ? NT : 1];
#pragma unroll
    for (int nt = 0; nt < (TWO ? NT : 1); ++nt) acc2[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < U; ++u) ko[u] = lds_i(kaddr)[u * 4];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        av[u] = *lds_f(abase + ko[u]);
        lds_cfp bp = lds_f(baddr + u * bstep);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) bv[u][nt] = bp[16 * nt];
    }
    {
        const int g1 = min(1, lastg);
#pragma unroll
        for (int u = 0; u < U; ++u) ko[u] = lds_i(kaddr)[(g1 * U + u) * 4];
    }
    for (int g = 0; g < ng; ++g) {
        const int gn = min(g + 1, lastg), gnn = min(g + 2, lastg);   // the last groups re-read themselves (unused)
        float an[U], bn[U][NT];
        int kn[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            an[u] = *lds_f(abase + ko[u]);
            lds_cfp bp = lds_f(baddr + (gn * U + u) * bstep);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) bn[u][nt] = bp[16 * nt];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) kn[u] = lds_i(kaddr)[(gnn * U + u) * 4];
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                if (TWO && (u & 1))
                    acc2[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u], bv[u][nt], acc2[nt], 0, 0, 0);
                else
                    acc[0][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u], bv[u][nt], acc[0][nt], 0, 0, 0);
            }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            av[u] = an[u];
            ko[u] = kn[u];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) bv[u][nt] = bn[u][nt];
        }
    }
    if (TWO) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[0][nt] += acc2[nt];
    }
}

// RB: row bases per wave (1: the wave's MT pixel tiles lie in one tile row, ... MT: one base per tile).
// XTAB: inputs staged by buffer loads through the per-tile offset tables (the normal case); the other
// instantiation keeps the pointer-arithmetic staging for geometries the tables do not cover (rows that are
// not 16-byte aligned).  Two kernels instead of a run-time branch: the K-chunk loop is short on scalar
// registers, and every spilled scalar comes back through a v_readlane, i.e. a VALU instruction.
#ifdef SPRK_DIAG
__device__ long long sprk_diag_conv_clk[40];   // phase time stamps of workgroup 0 (wall_clock64: 10 ns units)
#define SPRK_DIAG_STAMP(i) do { if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) sprk_diag_conv_clk[i] = wall_clock64(); } while (0)
#else
#define SPRK_DIAG_STAMP(i)
#endif

template <int MT, int NT, int RB, bool XTAB>
__global__ __launch_bounds__(kBlock, 2) void conv_mfma_kernel(const ConvArgs a) {
    SPRK_DIAG_STAMP(0);
    // LDS: koff[2][R4] (full chunk | last chunk) | xtab1[nG1*64] xtab2[nG2*64] | stage 0: input[CK*cplane]
    //      weights[R4*ldw] | stage 1: ...
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int stageFloats = a.CK * a.cplane + a.R4 * a.ldw;
    int *koff_base = reinterpret_cast<int *>(smem);
    int *xtab1 = koff_base + 2 * a.R4;
    int *xtab2 = xtab1 + a.nG1 * 64;
    float *stage_base = smem + 2 * a.R4 + (a.nG1 + a.nG2) * 64;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, lq = lane >> 4;
    constexpr int TM = 64 * MT;
    constexpr int kRowUnit = (MT == 1 && NT <= 3) ? 16 : 4;   // a chunk's k rows are rounded to this (plan_fwd: R4)
    const int lgT = a.lgTC + a.lgTR;
    const int NI = TM >> lgT;
    // XCD-aware tile order: workgroups are dealt round-robin to the 8 XCDs (each with its own L2), so
    // give every XCD a contiguous run of tiles — vertically adjacent tiles share their halo rows in L2.
    // (bijective remap; only a speed matter, see cdna_hip_programming.md T1)
    int bid = blockIdx.x;
    if (a.xcdRemap) {
        const int nwg = gridDim.x, q = nwg >> 3, rr = nwg & 7, xcd = bid & 7;
        bid = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + (bid >> 3);
    }
    const int tx = bid % a.tilesX;
    bid /= a.tilesX;
    const int ty = bid % a.tilesY;
    const int ig = bid / a.tilesY;
    const int nb = blockIdx.y;
    const int oy0 = ty << a.lgTR, ox0 = tx << a.lgTC, n0 = ig * NI;
    const int iy0 = oy0 * a.stride - a.padT, ix0 = ox0 * a.stride - a.padL;
    const int KHW = a.KH * a.KW, Cin = a.C1 + a.C2;
    const int TCm = (1 << a.lgTC) - 1, TRm = (1 << a.lgTR) - 1;
    const int lw = wave;

    int abase[RB];   // byte offset of this lane's pixel (row base r) inside a stage's input image
#pragma unroll
    for (int r = 0; r < RB; ++r) {
        const int p = (wave * MT + r * (MT / RB)) * 16 + l15;
        const int il = p >> lgT, rr = (p >> a.lgTC) & TRm, c = p & TCm;
        abase[r] = ((il * a.inRows + rr * a.stride) * a.pitch + c * a.stride + a.colOff) * 4;
    }
    f32x4 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};

    const PlaneGeom pg{NI, a.inRows, a.pitch, a.colOff, a.cplane, a.invImg, a.invPitch, a.deal, 4};
    if (XTAB) {
        const int ixa = ix0 - a.colOff;
        if (a.nG1)
            build_xtab(xtab1, a.nG1, a.vec1, a.up1, a.W1, (long)a.C1 * a.H1 * a.W1, a.N, a.Hin, a.Win, pg, n0, iy0, ixa,
                       tid, kBlock);
        if (a.nG2)
            build_xtab(xtab2, a.nG2, a.vec2, 0, a.Win, (long)a.C2 * a.Hin * a.Win, a.N, a.Hin, a.Win, pg, n0, iy0, ixa,
                       tid, kBlock);
    }
    const float *wslab = a.wT + (long)nb * a.rows * a.ldw;

    // issue the DMA of the chunk starting at channel c0 into stage `b`
    auto issue = [&](int c0, int b) {
        const int cke = min(a.CK, Cin - c0);
        const int kchunk = (cke * KHW + kRowUnit - 1) & ~(kRowUnit - 1);
        float *in_lds = stage_base + b * stageFloats;
        float *w_lds = in_lds + a.CK * a.cplane;
        const float *wsrc = wslab + (long)(c0 / a.CK) * a.R4 * a.ldw;
        const int total4 = (kchunk * a.ldw) >> 2;
        if (XTAB) {
            const int n1 = max(0, min(c0 + cke, a.C1) - c0);   // channels of this chunk taken from x
            if (n1 > 0) {
                const int cs1 = a.H1 * a.W1;
                stage_planes_buf<4>(in_lds, make_rsrc(a.x + ((long)n0 * a.C1 + c0) * cs1), cs1 * 4, n1, xtab1, a.nG1,
                                    a.vec1, a.cplane, lw, lane);
            }
            if (n1 < cke) {
                const int cs2 = a.Hin * a.Win;
                const int cb = max(c0, a.C1) - a.C1;
                stage_planes_buf<4>(in_lds + n1 * a.cplane, make_rsrc(a.x2 + ((long)n0 * a.C2 + cb) * cs2), cs2 * 4,
                                    cke - n1, xtab2, a.nG2, a.vec2, a.cplane, lw, lane);
            }
            const rsrc_t wr = make_rsrc(wsrc);
            const int wv = lane * 16, room = total4 - lane;   // lanes past the slab must not write (next stage)
            for (int gi = lw; gi * 64 < total4; gi += 4)
                if (gi * 64 < room) bdma16(wr, wv, gi * 1024, w_lds + gi * 256);
        } else {
            const TileSrc src{a.x, a.x2, a.zeros, a.N, a.C1, a.C2, a.Hin, a.Win, a.up1, a.H1, a.W1, a.vec1, a.vec2};
            stage_input_dma(in_lds, src, pg, n0, iy0, ix0, c0, cke, lw, lane);
            for (int gi = lw; gi * 64 < total4; gi += 4) {
                const int idx = gi * 64 + lane;
                if (idx < total4) dma16(wsrc + idx * 4, w_lds + gi * 256);
            }
        }
    };

    SPRK_DIAG_STAMP(1);
    __syncthreads();   // tables visible
    SPRK_DIAG_STAMP(2);
    issue(0, 0);
    SPRK_DIAG_STAMP(3);
    // k-row byte offsets (tap-major inside a chunk) for a full chunk and for the last, possibly shorter one; built
    // under the first chunk's DMA (the chunk loop's first barrier publishes them)
    const int ckeLast = Cin - ((Cin - 1) / a.CK) * a.CK;
    for (int idx = tid; idx < 2 * a.R4; idx += kBlock) {
        const int which = idx >= a.R4, k = idx - which * a.R4;
        const int cke = which ? ckeLast : a.CK;
        int v = 0;
        if (k < cke * KHW) {
            const int tap = k / cke, cl = k - tap * cke;
            const int ky = tap / a.KW, kx = tap - ky * a.KW;
            v = (cl * a.cplane + (ky * a.pitch + kx) * a.dil) * 4;
        }
        koff_base[idx] = v;
    }

    const int baddr0 = lds_addr(stage_base + a.CK * a.cplane + lq * a.ldw + l15);
    const int kaddr0 = lds_addr(koff_base + lq);
    const int abyte0 = lds_addr(stage_base);
    const int bstep = a.ldw * 16;
    int ci = 0;
    for (int c0 = 0; c0 < Cin; c0 += a.CK, ++ci) {
        // every wave's DMA of this chunk has landed (vmcnt(0)) and every wave is done reading the
        // other stage (previous chunk) once all have passed the barrier
        __syncthreads();
        if (c0 == 0) SPRK_DIAG_STAMP(4);
        if (ci < 8) SPRK_DIAG_STAMP(8 + 3 * ci);
        if (c0 + a.CK < Cin) issue(c0 + a.CK, (ci + 1) & 1);
        if (c0 == 0) SPRK_DIAG_STAMP(5);
        if (ci < 8) SPRK_DIAG_STAMP(9 + 3 * ci);
        const int cke = min(a.CK, Cin - c0);
        const int nkq = ((cke * KHW + kRowUnit - 1) & ~(kRowUnit - 1)) >> 2;
        const int soff = (ci & 1) * stageFloats * 4;
        int ab[RB];
#pragma unroll
        for (int r = 0; r < RB; ++r) ab[r] = abyte0 + soff + abase[r];
        if constexpr (MT == 1 && NT <= 3)
            chunk_mma_small<NT>(acc, ab[0], kaddr0 + (cke == a.CK ? 0 : a.R4 * 4), baddr0 + soff, bstep, nkq);
        else
            chunk_mma<MT, NT, RB>(acc, ab, kaddr0 + (cke == a.CK ? 0 : a.R4 * 4), baddr0 + soff, bstep, nkq);
        if (c0 == 0) SPRK_DIAG_STAMP(6);
        if (ci < 8) SPRK_DIAG_STAMP(10 + 3 * ci);
    }
    SPRK_DIAG_STAMP(7);

#include "conv_epilogue.inc"
}

// ------------------------------------------------------------------------------------------
// backward-weight MFMA kernel:  partial[group][k = c*KHW + tap][CoutP]
//   D[k-row][cout] += X[k-row][pixel] * G[pixel][cout] over the block's 64-pixel tiles; both
//   operands arrive by LDS-DMA, double buffered, so the next tile streams in under the MFMAs.
//   The G tile is stored [cout][64 pixels] with its 16 16-byte chunks XOR-swizzled by (cout & 15)
//   (applied on the DMA source address; the image itself is lane-linear): the 16 couts x 4 pixels one
//   B-operand read touches fall into 64 distinct words.
// ------------------------------------------------------------------------------------------
struct WgArgs {
    const float *x, *x2, *gy, *zeros;
    float *partial;
    int N, C1, C2, Hin, Win, up1, H1, W1;
    int Cout, CoutP, Hout, Wout;
    int KH, KW, stride, dil, padT, padL;
    int lgTC, lgTR;          // 64-pixel tile
    int tilesX, tilesY, nTiles, tilesPerGroup;
    int CKW;                 // input channels per blockIdx.y
    int ioffN;               // ints reserved for the k-row offset table
    int inRows, inCols, pitch, cplane, colOff;
    int xrow, g4, vec1, vec2, deal;
    int xtab;                // buffer-load staging of the row tiles (G, and x for 1x1)
    int nG1, nG2;            // MODE 2: 256-float groups per staged x plane (source 1 / 2): table staging
    float invImg, invPitch;
    int diag;                // timing experiments only (SPRK_WG_DIAG): 1 = stage the first tile of a group only
};

// one 64-pixel tile: 16 k-steps of MFMAs on NIT k-tiles x NT cout-tiles, software pipelined by hand: the LDS
// operands of step ks+1 are fetched into a second register set before the MFMAs of step ks are issued.
// The reduced dimension of this GEMM is the pixel, so which pixel feeds which (k-step, k-lane) slot is free as
// long as both operands agree:  k-step 4j+s, k-lane lq  <-  pixel 16j + 4lq + s.  Then the four G values a lane
// needs for steps 4j..4j+3 are one 16-byte chunk of its cout row: one ds_read_b128 per cout tile and 4 steps
// instead of four ds_read_b32, and the 8 lanes a b128 read serves per cycle hit 8 different swizzled chunks (no
// bank conflicts; the per-step b32 form was 2-way conflicted).
// Operand addresses are absolute LDS byte addresses and cost as few VALU instructions as possible (each one
// takes issue time from the MFMA stream):
//   MODE 2 (stride-1 tiles, rows of >= 4 pixels): A operand of k-tile t at  ab[t] + 4*(16j+s)  — an immediate —
//          where ab[t] is re-based per 16-pixel group from rowd (byte delta of a tile row against a straight run);
//   MODE 1 (1x1: the x tile is staged as swizzled 64-pixel rows like the G tile);
//   MODE 0 (any stride): pixel offsets from the pixoff table, one add per k-tile and step.
//   B operand: chunk ((4j+lq) ^ gsw) of the swizzled G row of this lane's cout, + nt * 4096 bytes.
template <int NIT, int IT, int NT, int MODE, bool DB>
__device__ __forceinline__ void wgrad_tile(f32x4 (&acc)[IT][NT], int xaddr, const int (&ioffb)[IT], int gaddr,
                                           lds_cip pixoffp, lds_cip rowdp, int lgTC, int gsw, int lq) {
    typedef const __attribute__((address_space(3))) f32x4 *lds_c4p;
    float a0[NIT], a1[NIT];
    f32x4 b0[NT], b1[DB ? NT : 1];   // DB: a second G register set (workgroups with one wave per SIMD)
    int ab[NIT];
    const int lsw = lq ^ gsw;
    auto loadB = [&](int j, f32x4 (&bv)[NT]) {
        const int q = ((4 * j) ^ lsw) << 4;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) bv[nt] = *(lds_c4p)(__SIZE_TYPE__)(unsigned)(gaddr + q + nt * 4096);
    };
    auto baseA = [&](int j) {   // MODE 2: row base of this lane's 4-pixel chunk of group j
        if (MODE == 2 && (j == 0 || lgTC < 6)) {
            const int d = rowdp[(16 * j + 4 * lq) >> lgTC];
#pragma unroll
            for (int t = 0; t < NIT; ++t) ab[t] = xaddr + ioffb[t] + d;
        }
    };
    auto loadA = [&](int j, int sx, float (&av)[NIT]) {
        if (MODE == 2) {
#pragma unroll
            for (int t = 0; t < NIT; ++t) av[t] = lds_f(ab[t])[16 * j + sx];
        } else {
            const int xo = MODE == 1 ? ((((4 * j) ^ lsw) << 4) + 4 * sx) : pixoffp[16 * j + 4 * lq + sx];
#pragma unroll
            for (int t = 0; t < NIT; ++t) av[t] = *lds_f(xaddr + ioffb[t] + xo);
        }
    };
    auto mma = [&](const float (&av)[NIT], const f32x4 (&bv)[NT], int sx) {
#pragma unroll
        for (int t = 0; t < NIT; ++t)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                acc[t][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[t], bv[nt][sx], acc[t][nt], 0, 0, 0);
    };
    loadB(0, b0);
    baseA(0);
    loadA(0, 0, a0);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if constexpr (DB) {
            if (j + 1 < 4) {
                if (j & 1) loadB(j + 1, b0); else loadB(j + 1, b1);
            }
        }
#pragma unroll
        for (int sx = 0; sx < 4; ++sx) {
            const int ks = 4 * j + sx;
            if (ks + 1 < 16) {            // A operands of the next step
                const int jn = (ks + 1) >> 2, sn = (ks + 1) & 3;
                if (sn == 0) baseA(jn);
                if (ks & 1) loadA(jn, sn, a0); else loadA(jn, sn, a1);
            }
            if constexpr (DB) {
                if (ks & 1) {
                    if (j & 1) mma(a1, b1, sx); else mma(a1, b0, sx);
                } else {
                    if (j & 1) mma(a0, b1, sx); else mma(a0, b0, sx);
                }
            } else {
                if (ks & 1) mma(a1, b0, sx); else mma(a0, b0, sx);
            }
        }
        // eight-wave workgroups: the G chunk of the next four steps goes into the same registers (a second set
        // does not fit next to 84 accumulators at two waves per SIMD; the SIMD partner covers the wait)
        if constexpr (!DB) {
            if (j + 1 < 4) loadB(j + 1, b0);
        }
    }
}

// (A first version of MODE 2's table staging of the x planes kept the per-group offsets in a small register
// array and was slower than pointer arithmetic: the array cost more VALU — initialisation, guarded reads,
// indexed access — than it saved.  Written as plain loops that read each table entry where it is used, it wins.)

// WJ = 2: eight waves (two per SIMD); wave (wi, wj) owns k-tiles wi, wi+4, ... x the wj-th half of the
// NT cout tiles, so that one wave's LDS latency and barrier skew are covered by its SIMD partner.
// MODE: operand addressing of wgrad_tile.  1 (1x1 convolutions: x staged as rows) and 2 also mean that the row
// tiles are staged by buffer loads (16-byte aligned rows, 32-bit offsets: the host checks); 0 is the general
// form with pointer-arithmetic staging.  Separate instantiations rather than run-time branches: spilled scalar
// registers come back as v_readlane, i.e. VALU instructions in the tile loop.
template <int IT, int NT, int WJ, int MODE>
__global__ __launch_bounds__(256 * WJ) void conv_wgrad_mfma_kernel(const WgArgs a) {
    constexpr bool XROW = MODE == 1;
    constexpr bool FAST = MODE != 0;
    constexpr int NTW = NT / WJ;          // cout tiles per wave
    constexpr int kWgThreads = 256 * WJ;
    constexpr int kWgWaves = 4 * WJ;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    // LDS: ioff[ioffN] | pixoff[64] | rowd[16] | xp[(nG1+nG2)*64] xv[(nG1+nG2)*64] (MODE 2) | 2 x { x, g }
    int *ioff = reinterpret_cast<int *>(smem);
    int *pixoff = ioff + a.ioffN;
    int *rowd = pixoff + 64;
    int *xp = rowd + 16, *xv = xp + (a.nG1 + a.nG2) * 64;
    float *stage_base = smem + a.ioffN + 64 + 16 + 2 * (a.nG1 + a.nG2) * 64;
    const int stageFloats = a.CKW * a.cplane + NT * 16 * 64;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, lq = lane >> 4;
    const int lgT = a.lgTC + a.lgTR;
    const int NI = 64 >> lgT;
    const int TCm = (1 << a.lgTC) - 1, TRm = (1 << a.lgTR) - 1;
    const int KHW = a.KH * a.KW, Cin = a.C1 + a.C2;
    const int c0 = blockIdx.y * a.CKW;
    const int cke = min(a.CKW, Cin - c0);
    const int kvalid = cke * KHW;
    const int nIT = (kvalid + 15) >> 4;
    const int co0 = blockIdx.z * (NT * 16);
    const long planeO = (long)a.Hout * a.Wout;
    const long planeI = (long)a.Hin * a.Win;
    const int wi = wave & 3, wj = wave >> 2;
    const int lw = wave;

    for (int k = tid; k < nIT * 16; k += kWgThreads) {
        int v = 0;
        if (k < kvalid) {
            const int tap = k / cke, cl = k - tap * cke;
            const int ky = tap / a.KW, kx = tap - ky * a.KW;
            v = (cl * a.cplane + (ky * a.pitch + kx) * a.dil) * 4;   // bytes
        }
        ioff[k] = v;
    }
    if (tid < 64) {
        const int il = tid >> lgT, r = (tid >> a.lgTC) & TRm, c = tid & TCm;
        const int off = (il * a.inRows + r * a.stride) * a.pitch + c * a.stride + a.colOff;
        pixoff[tid] = off * 4;   // bytes
        // byte delta of the tile row holding pixel `tid` against a straight 4-bytes-per-pixel run from pixel 0
        if (c == 0) rowd[tid >> a.lgTC] = (off - tid) * 4;
    }
    const PlaneGeom pg{NI, a.inRows, a.pitch, a.colOff, a.cplane, a.invImg, a.invPitch, a.deal, kWgWaves};
    if (MODE == 2) {
        // x planes by buffer loads: per lane and 256-float group of a plane, the packed in-tile coordinates
        // (-1: past the plane) and the byte offset against the tile origin; sources 1 and 2 back to back
        const int imgElems = a.inRows * a.pitch, planeElems = NI * imgElems;
        for (int idx = tid; idx < (a.nG1 + a.nG2) * 64; idx += kWgThreads) {
            const int second = idx >= a.nG1 * 64;
            const int gi = (idx - (second ? a.nG1 * 64 : 0)) >> 6, ln = idx & 63;
            const int e = gi * 256 + ln * 4;
            const int il = fast_div(e, a.invImg);
            const int rem = e - il * imgElems;
            const int r = fast_div(rem, a.invPitch);
            const int j = rem - r * a.pitch;
            const long imgStride = second ? (long)a.C2 * planeI : (long)a.C1 * planeI;
            xp[idx] = e < planeElems ? (il << 20) | (r << 10) | j : -1;
            xv[idx] = (int)((il * imgStride + (long)r * a.Win + j) * 4);
        }
    }
    // G rows (and, for 1x1, x rows): lane -> (row within the item's 4 rows, swizzled 16-byte chunk) -> pixel
    const int rowl = lane >> 4;
    int gpk, gvo, xvo;
    {
        const int rsw = ((lw & 3) << 2) | rowl;             // (row & 15) of every item this wave stages
        const int p = ((lane & 15) ^ rsw) << 2;
        const int il = p >> lgT, r = (p >> a.lgTC) & TRm, c = p & TCm;
        gpk = (il << 20) | (r << 10) | c;
        gvo = (int)((((long)il * a.Cout + rowl) * planeO + (long)r * a.Wout + c) * 4);
        xvo = (int)((((long)il * a.C1 + rowl) * planeI + (long)r * a.Win + c) * 4);
    }
    const TileSrc src{a.x, a.x2, a.zeros, a.N, a.C1, a.C2, a.Hin, a.Win, a.up1, a.H1, a.W1, a.vec1, a.vec2};

    // rows of 64 pixels, 16-byte chunks XOR-swizzled by (row & 15): LDS chunk (row, qs) <- pixels
    // 4q..4q+3, q = qs ^ (row & 15).  base: tensor [N][C][H*W]; row r is channel ch0 + r.
    auto stage_rows16 = [&](float *dst, const float *base, int C, long plane, int W, int H, int ch0, int nrows,
                            int chlim, int n0, int oy0, int ox0) {
        for (int gi = lw; gi * 4 < nrows; gi += kWgWaves) {
            const int idx = gi * 64 + lane;
            const int row = idx >> 4, qs = idx & 15;
            const int p = (qs ^ (row & 15)) << 2;
            const int il = p >> lgT, r = (p >> a.lgTC) & TRm, c = p & TCm;
            const int n = n0 + il, oy = oy0 + r, ox = ox0 + c, ch = ch0 + row;
            const float *s = a.zeros + lane * 4;
            if (ch < chlim && n < a.N && oy < H && ox < W) s = base + ((long)n * C + ch) * plane + (long)oy * W + ox;
            if (row < nrows) dma16(s, dst + gi * 256);
        }
    };
    // the same through buffer loads: the pixel part of the address is this lane's fixed offset against the
    // tile origin (vo), made invalid (zero fill) per tile; the channel part is wave-uniform
    auto stage_rows_buf = [&](float *dst, const float *base, int C, long plane, int W, int ch0, int nrows, int chlim,
                              int n0, int oy0, int ox0, int vo, bool inside) {
        const rsrc_t r = make_rsrc(base + ((long)n0 * C + ch0) * plane + (long)oy0 * W + ox0);
        const int v = inside ? vo : kXZero;
        const int rowBytes4 = (int)(plane * 16);
        for (int gi = lw; gi * 4 < nrows; gi += kWgWaves) {
            int vv = v;
            if (ch0 + gi * 4 + 3 >= chlim) vv = (ch0 + gi * 4 + rowl < chlim) ? v : kXZero;   // ragged last cout block
            if (gi * 4 + rowl < nrows) bdma16(r, vv, gi * rowBytes4, dst + gi * 256);
        }
    };
    auto issue = [&](int tile, int b) {
        if (a.diag && tile != (int)blockIdx.x * a.tilesPerGroup) return;
        int t = tile;
        const int tx = t % a.tilesX;
        t /= a.tilesX;
        const int ty = t % a.tilesY;
        const int ig = t / a.tilesY;
        const int oy0 = ty << a.lgTR, ox0 = tx << a.lgTC, n0 = ig * NI;
        float *x_lds = stage_base + b * stageFloats;
        float *g_lds = x_lds + a.CKW * a.cplane;
        bool inside = false;
        if (FAST) {
            const int r = (gpk >> 10) & 1023, c = gpk & 1023, il = gpk >> 20;
            inside = n0 + il < a.N && oy0 + r < a.Hout && ox0 + c < a.Wout;
        }
        if (XROW) {
            stage_rows_buf(x_lds, a.x, a.C1, planeI, a.Win, c0, cke, Cin, n0, oy0, ox0, xvo, inside);
        } else {
            if (MODE == 2) {
                // table + buffer-load staging: one validity pass per 256-float group and tile, then a
                // wave-uniform channel offset per DMA; lanes outside the image read zeros by range check
                const int iy0 = oy0 - a.padT, ixa = ox0 - a.padL - a.colOff;
                const int n1 = max(0, min(c0 + cke, a.C1) - c0);   // channels of this chunk taken from x
                auto planes = [&](const float *base, int nch, int nG, int tb, float *dstp) {
                    const rsrc_t r = make_rsrc(base + (long)iy0 * a.Win + ixa);
                    for (int gi = 0; gi < nG; ++gi) {
                        const int pk = xp[tb + gi * 64 + lane], vo = xv[tb + gi * 64 + lane];
                        const int rr = (pk >> 10) & 1023, jj = pk & 1023, il = pk >> 20;
                        const bool ok = n0 + il < a.N && (unsigned)(iy0 + rr) < (unsigned)a.Hin &&
                                        (unsigned)(ixa + jj) < (unsigned)a.Win;
                        const int o = ok ? vo : kXZero;
                        const int cl0 = (lw + kWgWaves - (gi % kWgWaves)) % kWgWaves;
                        float *d = dstp + gi * 256 + cl0 * a.cplane;
                        int soff = cl0 * (int)planeI * 4;
                        if (pk >= 0) {
                            for (int cl = cl0; cl < nch; cl += kWgWaves) {
                                bdma16(r, o, soff, d);
                                soff += kWgWaves * (int)planeI * 4;
                                d += kWgWaves * a.cplane;
                            }
                        }
                    }
                };
                if (n1 > 0) planes(a.x + ((long)n0 * a.C1 + c0) * planeI, n1, a.nG1, 0, x_lds);
                if (n1 < cke)
                    planes(a.x2 + ((long)n0 * a.C2 + (max(c0, a.C1) - a.C1)) * planeI, cke - n1, a.nG2, a.nG1 * 64,
                           x_lds + n1 * a.cplane);
            } else {
                stage_input_dma(x_lds, src, pg, n0, oy0 * a.stride - a.padT, ox0 * a.stride - a.padL, c0, cke, lw, lane);
            }
        }
        if (FAST) {
            stage_rows_buf(g_lds, a.gy, a.Cout, planeO, a.Wout, co0, NT * 16, a.Cout, n0, oy0, ox0, gvo, inside);
        } else if (a.g4) {
            stage_rows16(g_lds, a.gy, a.Cout, planeO, a.Wout, a.Hout, co0, NT * 16, a.Cout, n0, oy0, ox0);
        } else {
            for (int col = lw; col < NT * 16; col += kWgWaves) {
                const int p = (((lane >> 2) ^ (col & 15)) << 2) | (lane & 3);
                const int il = p >> lgT, r = (p >> a.lgTC) & TRm, c = p & TCm;
                const int n = n0 + il, oy = oy0 + r, ox = ox0 + c, co = co0 + col;
                const float *s = a.zeros + lane;
                if (co < a.Cout && n < a.N && oy < a.Hout && ox < a.Wout)
                    s = a.gy + ((long)n * a.Cout + co) * planeO + (long)oy * a.Wout + ox;
                dma4(s, g_lds + col * 64);
            }
        }
    };

    const int t_begin = blockIdx.x * a.tilesPerGroup;
    const int t_end = min(a.nTiles, t_begin + a.tilesPerGroup);
    __syncthreads();  // tables visible
    if (t_begin < t_end) issue(t_begin, 0);
    int ioffb[IT];   // byte offset of this lane's k row (k-tile t) inside a stage's x image (+ its pixel quad)
#pragma unroll
    for (int t = 0; t < IT; ++t) {
        const int it = wi + 4 * t;
        ioffb[t] = (it < nIT ? ioff[it * 16 + l15] : 0) + (MODE == 2 ? lq * 16 : 0);
    }
    f32x4 acc[IT][NTW];
#pragma unroll
    for (int t = 0; t < IT; ++t)
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) acc[t][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int gsw = l15;   // rows are multiples of 16 apart per cout tile: row & 15 == l15
    const int myIT = nIT > wi ? (nIT - wi + 3) >> 2 : 0;  // k-tiles owned by this wave (uniform)
    const int xaddr0 = lds_addr(stage_base);
    const int gaddr0 = lds_addr(stage_base + a.CKW * a.cplane + wj * (NTW * 16 * 64) + l15 * 64);
    const lds_cip pixoffp = lds_i(lds_addr(pixoff)), rowdp = lds_i(lds_addr(rowd));

    // The tile loop is written once per k-tile count of this wave (a wave-uniform choice made once): with the
    // choice inside the loop the three variants get different accumulator registers and the compiler copies
    // all 84 of them at every join — 36-72 v_mov per tile, each costing MFMA issue time.
    auto tiles = [&](auto body) {
        int bsel = 0;
        for (int tile = t_begin; tile < t_end; ++tile, bsel ^= 1) {
            __syncthreads();  // this tile landed everywhere; previous tile fully consumed
            if (tile + 1 < t_end) issue(tile + 1, bsel ^ 1);
            const int soff = bsel * stageFloats * 4;
            body(xaddr0 + soff, gaddr0 + soff);
        }
    };
    if (myIT >= IT) {
        tiles([&](int xaddr, int gaddr) {
            wgrad_tile<IT, IT, NTW, MODE, WJ == 1>(acc, xaddr, ioffb, gaddr, pixoffp, rowdp, a.lgTC, gsw, lq);
        });
    } else if (IT > 1 && myIT == IT - 1) {
        tiles([&](int xaddr, int gaddr) {
            wgrad_tile<(IT > 1 ? IT - 1 : 1), IT, NTW, MODE, WJ == 1>(acc, xaddr, ioffb, gaddr, pixoffp, rowdp, a.lgTC, gsw,
                                                                      lq);
        });
    } else {
        // short tail chunk: predicate per k-tile (rare: last channel chunk of a layer); same pixel order.
        // (waves without k-tiles still take part in the staging and the barriers)
        tiles([&](int xaddr, int gaddr) {
            if (myIT <= 0) return;
#pragma unroll 1
            for (int ks = 0; ks < 16; ++ks) {
                const int j = ks >> 2, sx = ks & 3;
                const int q = ((4 * j) ^ lq ^ gsw) << 4;
                const int p = 16 * j + 4 * lq + sx;
                const int xo = MODE == 1 ? q + 4 * sx : MODE == 2 ? (16 * j + sx) * 4 + rowdp[p >> a.lgTC] : pixoffp[p];
                float bv[NTW];
#pragma unroll
                for (int nt = 0; nt < NTW; ++nt) bv[nt] = lds_f(gaddr + q + 4 * sx)[nt * 1024];
#pragma unroll
                for (int t = 0; t < IT; ++t) {
                    if (t < myIT) {
                        const float av = *lds_f(xaddr + ioffb[t] + xo);
#pragma unroll
                        for (int nt = 0; nt < NTW; ++nt)
                            acc[t][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv[nt], acc[t][nt], 0, 0, 0);
                    }
                }
            }
        });
    }
    // partial slab of this group: rows = global k (c*KHW + tap), cols = cout (64-byte runs per store)
    float *dst = a.partial + (long)blockIdx.x * Cin * KHW * a.CoutP;
#pragma unroll
    for (int t = 0; t < IT; ++t) {
        const int it = wi + 4 * t;
        if (it >= nIT) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int k = it * 16 + lq * 4 + j;
            if (k >= kvalid) continue;
            const int tap = k / cke, cl = k - tap * cke;
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt) {
                const int co = co0 + (wj * NTW + nt) * 16 + l15;
                if (co < a.CoutP) dst[((long)(c0 + cl) * KHW + tap) * a.CoutP + co] = acc[t][nt][j];
            }
        }
    }
}


// ------------------------------------------------------------------------------------------
// direct kernels (plain statement of the maths)
// ------------------------------------------------------------------------------------------
struct DirectArgs {
    const float *x, *x2, *w, *gy;
    float *y;
    sprk_conv_geom g;
    sprk_conv_epilogue ep;
};

__device__ __forceinline__ float fetch_in(const DirectArgs &a, int n, int c, int iy, int ix) {
    const sprk_conv_geom &g = a.g;
    if ((unsigned)iy >= (unsigned)g.Hin || (unsigned)ix >= (unsigned)g.Win) return 0.f;
    if (c < g.C1) {
        if (g.up1) return a.x[(((long)n * g.C1 + c) * (g.Hin >> 1) + (iy >> 1)) * (g.Win >> 1) + (ix >> 1)];
        return a.x[(((long)n * g.C1 + c) * g.Hin + iy) * g.Win + ix];
    }
    return a.x2[(((long)n * g.C2 + (c - g.C1)) * g.Hin + iy) * g.Win + ix];
}

__global__ void conv_fwd_direct_kernel(const DirectArgs a) {
    const sprk_conv_geom &g = a.g;
    const int Cin = g.C1 + g.C2;
    const long total = (long)g.N * g.Cout * g.Hout * g.Wout;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        long t = e;
        const int ox = (int)(t % g.Wout); t /= g.Wout;
        const int oy = (int)(t % g.Hout); t /= g.Hout;
        const int co = (int)(t % g.Cout);
        const int n = (int)(t / g.Cout);
        float s = 0.f;
        for (int c = 0; c < Cin; ++c)
            for (int ky = 0; ky < g.KH; ++ky)
                for (int kx = 0; kx < g.KW; ++kx)
                    s += fetch_in(a, n, c, oy * g.stride - g.pad_top + ky * g.dil, ox * g.stride - g.pad_left + kx * g.dil) *
                         a.w[(((long)co * Cin + c) * g.KH + ky) * g.KW + kx];
        if (a.ep.res) s += a.ep.res[(((long)n * g.Cout + co) * a.ep.res_h + oy + a.ep.res_off) * a.ep.res_w + ox + a.ep.res_off];
        if (a.ep.scale)
            s = s * a.ep.scale[co] + a.ep.shift[co];
        else if (a.ep.bias)
            s += a.ep.bias[co];
        const float v = apply_act(s, a.ep.act);
        if (a.ep.up2) {
            float *q = a.y + (((long)n * g.Cout + co) * (2 * g.Hout) + 2 * oy) * (2L * g.Wout) + 2 * ox;
            q[0] = v;
            q[1] = v;
            q[2L * g.Wout] = v;
            q[2L * g.Wout + 1] = v;
        } else {
            a.y[e] = v;
        }
    }
}

// y here is gin [N,Cin,Hin,Win]
__global__ void conv_bwd_data_direct_kernel(const DirectArgs a) {
    const sprk_conv_geom &g = a.g;
    const int Cin = g.C1 + g.C2;
    const long total = (long)g.N * Cin * g.Hin * g.Win;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        long t = e;
        const int ix = (int)(t % g.Win); t /= g.Win;
        const int iy = (int)(t % g.Hin); t /= g.Hin;
        const int c = (int)(t % Cin);
        const int n = (int)(t / Cin);
        float s = 0.f;
        for (int ky = 0; ky < g.KH; ++ky) {
            const int ny = iy + g.pad_top - ky * g.dil;
            if (ny < 0 || ny % g.stride) continue;
            const int oy = ny / g.stride;
            if (oy >= g.Hout) continue;
            for (int kx = 0; kx < g.KW; ++kx) {
                const int nx = ix + g.pad_left - kx * g.dil;
                if (nx < 0 || nx % g.stride) continue;
                const int ox = nx / g.stride;
                if (ox >= g.Wout) continue;
                for (int co = 0; co < g.Cout; ++co)
                    s += a.gy[(((long)n * g.Cout + co) * g.Hout + oy) * g.Wout + ox] *
                         a.w[(((long)co * Cin + c) * g.KH + ky) * g.KW + kx];
            }
        }
        a.y[e] = s;
    }
}

// one block per (co, c); y here is gw
__global__ __launch_bounds__(256) void conv_bwd_weight_direct_kernel(const DirectArgs a) {
    const sprk_conv_geom &g = a.g;
    const int Cin = g.C1 + g.C2;
    const int co = blockIdx.x / Cin, c = blockIdx.x % Cin;
    __shared__ float red[256];
    const long npix = (long)g.N * g.Hout * g.Wout;
    for (int ky = 0; ky < g.KH; ++ky)
        for (int kx = 0; kx < g.KW; ++kx) {
            float s = 0.f;
            for (long p = threadIdx.x; p < npix; p += 256) {
                long t = p;
                const int ox = (int)(t % g.Wout); t /= g.Wout;
                const int oy = (int)(t % g.Hout);
                const int n = (int)(t / g.Hout);
                s += fetch_in(a, n, c, oy * g.stride - g.pad_top + ky * g.dil, ox * g.stride - g.pad_left + kx * g.dil) *
                     a.gy[(((long)n * g.Cout + co) * g.Hout + oy) * g.Wout + ox];
            }
            red[threadIdx.x] = s;
            __syncthreads();
            for (int w = 128; w > 0; w >>= 1) {
                if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
                __syncthreads();
            }
            if (threadIdx.x == 0) a.y[(((long)co * Cin + c) * g.KH + ky) * g.KW + kx] = red[0];
            __syncthreads();
        }
}

// ------------------------------------------------------------------------------------------
// activation backward (+ bias gradient) and concat/upsample gradient split
// ------------------------------------------------------------------------------------------
// grid (C, nsplit): block (c, s) walks images n = s, s+nsplit, ... of channel c.
// up2: g and y are the 2x upsampled tensors [N,C,2H,2W] written by a conv with the fused upsample
// epilogue; the incoming gradient is first summed over each 2x2 block (backward of nn.Upsample).
// gstride: elements between consecutive images of g (a channel slice of a wider tensor: the backward-data output of
// a concat layer is handed on as views, not copied); y and gpre are dense.
// TG / TY / TO: element types of g, y and gpre (fp32 or the 16-bit storage type, io_dev.h); arithmetic and the bias
// sums are fp32.
template <typename TG, typename TY, typename TO>
__global__ __launch_bounds__(256) void act_bwd_kernel(const TG *g, const TY *__restrict__ y, TO *gpre,
                                                      float *__restrict__ partial, int act, int N, int C, int HW,
                                                      int W, int up2, int nsplit, long gstride) {
    const int c = blockIdx.x, s = blockIdx.y;
    float sum = 0.f;
    const bool al16 = (((uintptr_t)g | (uintptr_t)y | (uintptr_t)gpre) & 15) == 0 && (gstride & 3) == 0;
    const bool v4 = !up2 && (HW & 3) == 0 && al16;
    const bool v2u = up2 && (W & 1) == 0 && al16;
    for (int n = s; n < N; n += nsplit) {
        const long base = ((long)n * C + c) * HW;
        const long gbase = (long)n * gstride + (long)c * (up2 ? 4L * HW : HW);     // first element of g's plane
        if (v4) {  // four elements per lane
            for (int i = threadIdx.x * 4; i < HW; i += 1024) {
                float4 v = IO<TG>::ld4(g + gbase + i);
                if (act != SPRK_ACT_NONE) {
                    const float4 yv = IO<TY>::ld4(y + base + i);
                    const bool lk = act == SPRK_ACT_LEAKY;
                    v.x = yv.x > 0.f ? v.x : (lk ? v.x * kLeak : 0.f);
                    v.y = yv.y > 0.f ? v.y : (lk ? v.y * kLeak : 0.f);
                    v.z = yv.z > 0.f ? v.z : (lk ? v.z * kLeak : 0.f);
                    v.w = yv.w > 0.f ? v.w : (lk ? v.w * kLeak : 0.f);
                    IO<TO>::st4(gpre + base + i, v);
                } else if (IO<TG>::code != IO<TO>::code && gpre) {
                    IO<TO>::st4(gpre + base + i, v);       // a pure change of storage type on the way (+ the bias sum)
                }
                sum += (v.x + v.y) + (v.z + v.w);
            }
            continue;
        }
        if (v2u) {  // two outputs per lane: four elements from each of the two source rows
            for (int i = threadIdx.x * 2; i < HW; i += 512) {
                const int oy = i / W, ox = i - oy * W;
                const long q = base * 4 + (long)(2 * oy) * (2 * W) + 2 * ox;
                const long gq = gbase + (long)(2 * oy) * (2 * W) + 2 * ox;
                const float4 a = IO<TG>::ld4(g + gq);
                const float4 b = IO<TG>::ld4(g + gq + 2 * W);
                float2 v = make_float2((a.x + a.y) + (b.x + b.y), (a.z + a.w) + (b.z + b.w));
                if (act != SPRK_ACT_NONE) {
                    const float4 yv = IO<TY>::ld4(y + q);
                    const bool lk = act == SPRK_ACT_LEAKY;
                    v.x = yv.x > 0.f ? v.x : (lk ? v.x * kLeak : 0.f);
                    v.y = yv.z > 0.f ? v.y : (lk ? v.y * kLeak : 0.f);
                }
                IO<TO>::st2(gpre + base + i, v);
                sum += v.x + v.y;
            }
            continue;
        }
        for (int i = threadIdx.x; i < HW; i += 256) {
            float v, yv = 0.f;
            if (up2) {
                const int oy = i / W, ox = i - oy * W;
                const long q = base * 4 + (long)(2 * oy) * (2 * W) + 2 * ox;
                const long gq = gbase + (long)(2 * oy) * (2 * W) + 2 * ox;
                v = (IO<TG>::ld(g + gq) + IO<TG>::ld(g + gq + 1)) + (IO<TG>::ld(g + gq + 2 * W) + IO<TG>::ld(g + gq + 2 * W + 1));
                if (act != SPRK_ACT_NONE) yv = IO<TY>::ld(y + q);
            } else {
                v = IO<TG>::ld(g + gbase + i);
                if (act != SPRK_ACT_NONE) yv = IO<TY>::ld(y + base + i);
            }
            if (act == SPRK_ACT_LEAKY)
                v = yv > 0.f ? v : v * kLeak;
            else if (act == SPRK_ACT_RELU)
                v = yv > 0.f ? v : 0.f;
            if (act != SPRK_ACT_NONE || up2 || (IO<TG>::code != IO<TO>::code && gpre)) IO<TO>::st(gpre + base + i, v);
            sum += v;
        }
    }
    if (partial) {
        __shared__ float red[256];
        red[threadIdx.x] = sum;
        __syncthreads();
        for (int w = 128; w > 0; w >>= 1) {
            if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
            __syncthreads();
        }
        if (threadIdx.x == 0) partial[c * nsplit + s] = red[0];
    }
}

__global__ void concat_up_bwd_kernel(const float *__restrict__ gin, float *__restrict__ ga, float *__restrict__ gb,
                                     int N, int C1, int C2, int H, int W, int up1) {
    const int Cin = C1 + C2;
    const int H1 = up1 ? H >> 1 : H, W1 = up1 ? W >> 1 : W;
    const long nA = (long)N * C1 * H1 * W1, nB = (long)N * C2 * H * W;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < nA + nB; e += (long)gridDim.x * blockDim.x) {
        if (e < nA) {
            long t = e;
            const int x = (int)(t % W1); t /= W1;
            const int y = (int)(t % H1); t /= H1;
            const int c = (int)(t % C1);
            const int n = (int)(t / C1);
            const float *p = gin + (((long)n * Cin + c) * H) * W;
            float v;
            if (up1) {
                const float *q = p + (long)(2 * y) * W + 2 * x;
                v = (q[0] + q[1]) + (q[W] + q[W + 1]);
            } else {
                v = p[(long)y * W + x];
            }
            ga[e] = v;
        } else {
            long t = e - nA;
            const int x = (int)(t % W); t /= W;
            const int y = (int)(t % H); t /= H;
            const int c = (int)(t % C2);
            const int n = (int)(t / C2);
            gb[e - nA] = gin[(((long)n * Cin + C1 + c) * H + y) * W + x];
        }
    }
}

// ------------------------------------------------------------------------------------------
// host-side planning
// ------------------------------------------------------------------------------------------
constexpr size_t kLdsLimit = 160 * 1024;

int dbg_int(const char *name, int dflt) {
    const char *v = getenv(name);
    return v ? atoi(v) : dflt;
}

struct FwdPlan {
    int MT, NT, lgTC, lgTR, tilesX, tilesY, imgGroups, nblkN;
    int CK, R4, rows;
    int inRows, inCols, pitch, cplane, colOff, ldw, NI;
    int xtab, nG1, nG2;   // table-driven input staging (conv_mfma_kernel): groups per plane, source 1 / 2
    size_t ldsBytes, wsBytes;
};

int pad_to_residue(int raw, int residue) {  // smallest v >= raw with v % 32 == residue
    return raw + ((residue - raw % 32) + 32) % 32;
}

int pick_nt(int ntile) {
    // N tiles per block: every block re-stages the input tile, so fewer, wider blocks win; the
    // last block may be ragged (its surplus columns are zero weights, never stored)
    int NT = 1, best = 1 << 30;
    for (int cand : {6, 4, 3, 2, 1}) {
        const int cost = sprk::cdiv(ntile, cand) * (cand + 2);
        if (cost < best) {
            best = cost;
            NT = cand;
        }
    }
    return NT;
}

// Ck: channels along GEMM-k; Nn: GEMM-n extent; output spatial dims Ho x Wo over Nimg images
bool plan_fwd(int Nimg, int Ck, int Nn, int Ho, int Wo, int KH, int KW, int stride, int dil, int padL, int up1,
              int hasC2, int WinSrc, FwdPlan *p) {
    const int KHW = KH * KW;
    const int ntile = sprk::cdiv(Nn, 16);
    int NT = pick_nt(ntile);
    int MT = 4;
    auto geometry = [&](int mt, int nt) {
        const int TM = 64 * mt;
        int TC = std::min(sprk::pow2_ceil(Wo), 64);
        if (dil * (KW - 1) >= 8 && TM >= 256) TC = std::min(TC, 16);
        TC = std::min(TC, TM);
        int TR = std::min(TM / TC, sprk::pow2_ceil(Ho));
        int NI = TM / (TC * TR);
        p->MT = mt;
        p->NT = nt;
        p->NI = NI;
        p->lgTC = sprk::ilog2(TC);
        p->lgTR = sprk::ilog2(TR);
        p->tilesX = sprk::cdiv(Wo, TC);
        p->tilesY = sprk::cdiv(Ho, TR);
        p->imgGroups = sprk::cdiv(Nimg, NI);
        p->nblkN = sprk::cdiv(ntile, nt);
        p->inRows = (TR - 1) * stride + (KH - 1) * dil + 1;
        p->inCols = (TC - 1) * stride + (KW - 1) * dil + 1;
        // image column 0 sits on a 16-byte boundary of the global rows (tile origins are multiples of 4)
        p->colOff = ((TC * stride) % 4 == 0) ? (((-padL) % 4) + 4) % 4 : 0;
        p->pitch = sprk::roundup(p->colOff + p->inCols, 4);
        p->cplane = pad_to_residue(NI * p->inRows * p->pitch, 16);
        return (long)p->imgGroups * p->tilesX * p->tilesY * p->nblkN;
    };
    long blocks = geometry(MT, NT);
    while (blocks < 512 && MT > 1) {
        MT >>= 1;
        blocks = geometry(MT, NT);
    }
    static const int kNtBlocks = dbg_int("SPRK_FWD_NTBLK", 512);   // debug knob (sweeps)
    while (blocks < kNtBlocks && NT > 1) {
        NT = (NT == 6) ? 3 : (NT == 4) ? 2 : 1;
        blocks = geometry(MT, NT);
    }
    p->ldw = (NT % 2) ? NT * 16 : NT * 16 + 16;
    // per-tile source-offset tables (one entry per DMA lane and 256- or 64-float group of a channel plane):
    // source 1 by 16-byte DMA, or dword DMA when it is upsampled on load; source 2 by 16-byte DMA
    {
        const int planeElems = p->NI * p->inRows * p->pitch;
        const bool vecGeo = (((1 << p->lgTC) * stride) % 4 == 0) && (WinSrc % 4 == 0);
        p->nG1 = up1 ? sprk::cdiv(planeElems, 64) : sprk::cdiv(planeElems, 256);
        p->nG2 = hasC2 ? sprk::cdiv(planeElems, 256) : 0;
        p->xtab = (vecGeo && p->nG1 <= kMaxXG && p->nG2 <= kMaxXG) ? 1 : 0;
        if (!p->xtab) p->nG1 = p->nG2 = 0;
    }
    // channels per K-chunk: two stages (double buffer) should leave room for 3 workgroups per CU
    const int rowUnit = (MT == 1 && NT <= 3) ? 16 : 4;   // chunk_mma_small walks a chunk in groups of 4 k-steps
    auto lds = [&](int ck) {
        const int r4 = sprk::roundup(ck * KHW, rowUnit);
        const int stages = ck >= Ck ? 1 : 2;   // the whole of K in one chunk: nothing to double-buffer
        return (size_t)(2 * r4 + (p->nG1 + p->nG2) * 64 + stages * (ck * p->cplane + r4 * p->ldw)) * 4;
    };
    int CK = std::max(1, std::min(Ck, KHW == 1 ? 16 : std::max(1, 36 / KHW)));
    while (CK > 1 && lds(CK) > 52 * 1024) CK >>= 1;
    // small tiles: longer K-chunks (fewer barriers) while the stages stay small
    while (CK * 2 <= Ck && CK * 2 * KHW <= 288 && lds(CK * 2) <= (size_t)(MT == 1 ? 64 : 24) * 1024) CK *= 2;
    if (MT == 1 && blocks <= 2 * sprk::num_cus()) {
        // planes of a few pixels on a grid the CUs hold at once: these launches are latency-bound (a workgroup's serial
        // preamble -> DMA -> MFMA -> store chain, not the machine's throughput), and every chunk adds a DMA issue, a
        // landing wait and a barrier to that chain: as few chunks as the LDS allows — all of K in one stage when it
        // fits.  One workgroup per CU is all a grid of <= 256 workgroups needs, two up to 512; larger grids keep the
        // smaller stages (more resident workgroups hide more latency than fewer chunks save: measured).
        const size_t room = blocks <= sprk::num_cus() ? 156 * 1024 : 78 * 1024;
        for (int nCh = 1; nCh <= Ck; ++nCh) {
            const int ck = sprk::cdiv(Ck, nCh);
            if (ck <= CK) break;
            if (lds(ck) <= room) {
                CK = ck;
                break;
            }
        }
    }
    if (lds(CK) > kLdsLimit) return false;
    p->CK = CK;
    p->R4 = sprk::roundup(CK * KHW, rowUnit);
    p->rows = sprk::cdiv(Ck, CK) * p->R4;
    p->ldsBytes = lds(CK);
    p->wsBytes = ((size_t)kZeroFloats + (size_t)p->nblkN * p->rows * p->ldw) * sizeof(float);
    return true;
}

template <typename K>
int set_lds(K kernel, size_t bytes) {
    if (bytes > 64 * 1024) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)bytes) != hipSuccess) {
            sprk::set_error("cannot reserve %zu bytes of LDS", bytes);
            return SPRK_ELAUNCH;
        }
    }
    return SPRK_OK;
}

template <int MT, int NT, int RB>
int launch_fwd_one(const ConvArgs &a, const FwdPlan &p, dim3 grid, hipStream_t s) {
    if (a.xtab) {
        if (int rc = set_lds(conv_mfma_kernel<MT, NT, RB, true>, p.ldsBytes)) return rc;
        hipLaunchKernelGGL((conv_mfma_kernel<MT, NT, RB, true>), grid, dim3(kBlock), p.ldsBytes, s, a);
    } else {
        if (int rc = set_lds(conv_mfma_kernel<MT, NT, RB, false>, p.ldsBytes)) return rc;
        hipLaunchKernelGGL((conv_mfma_kernel<MT, NT, RB, false>), grid, dim3(kBlock), p.ldsBytes, s, a);
    }
    return SPRK_OK;
}

template <int MT, int RB>
int launch_fwd_nt(const ConvArgs &a, const FwdPlan &p, dim3 grid, hipStream_t s) {
    switch (p.NT) {
        case 1: return launch_fwd_one<MT, 1, RB>(a, p, grid, s);
        case 2: return launch_fwd_one<MT, 2, RB>(a, p, grid, s);
        case 3: return launch_fwd_one<MT, 3, RB>(a, p, grid, s);
        case 4: return launch_fwd_one<MT, 4, RB>(a, p, grid, s);
        default: return launch_fwd_one<MT, 6, RB>(a, p, grid, s);
    }
}

int launch_fwd(const ConvArgs &a, const FwdPlan &p, hipStream_t s) {
    dim3 grid(p.imgGroups * p.tilesX * p.tilesY, p.nblkN);
    // pixel tiles of a wave that lie in one tile row are 16 floats apart in LDS (stride 1): one address
    // base per row instead of one per tile
    const int rows = (a.stride == 1) ? std::max(1, (p.MT * 16) >> p.lgTC) : p.MT;
    switch (p.MT) {
        case 1: return launch_fwd_nt<1, 1>(a, p, grid, s);
        case 2: return rows == 1 ? launch_fwd_nt<2, 1>(a, p, grid, s) : launch_fwd_nt<2, 2>(a, p, grid, s);
        default:
            return rows == 1 ? launch_fwd_nt<4, 1>(a, p, grid, s)
                             : rows == 2 ? launch_fwd_nt<4, 2>(a, p, grid, s) : launch_fwd_nt<4, 4>(a, p, grid, s);
    }
}

bool aligned16(const void *p) {
    static const bool novec = getenv("SPRK_NOVEC") != nullptr;  // debug: force the 4-byte DMA path
    return !novec && (((uintptr_t)p) & 15) == 0;
}

void fill_args(ConvArgs &a, const FwdPlan &p) {
    a.lgTC = p.lgTC;
    a.lgTR = p.lgTR;
    a.tilesX = p.tilesX;
    a.tilesY = p.tilesY;
    a.CK = p.CK;
    a.R4 = p.R4;
    a.rows = p.rows;
    a.inRows = p.inRows;
    a.inCols = p.inCols;
    a.pitch = p.pitch;
    a.cplane = p.cplane;
    a.colOff = p.colOff;
    a.ldw = p.ldw;
    a.invImg = 1.0f / (float)(p.inRows * p.pitch);
    a.invPitch = 1.0f / (float)p.pitch;
    // 16-byte DMA: tile origins on 4-column boundaries, rows 16-byte aligned, source at full resolution
    const bool geo = ((1 << p.lgTC) * a.stride) % 4 == 0 && (a.Win % 4) == 0;
    a.vec1 = (geo && !a.up1 && aligned16(a.x)) ? 1 : 0;
    a.vec2 = (geo && a.x2 && aligned16(a.x2)) ? 1 : 0;
    a.deal = 1;
    a.xcdRemap = 1;
    // the tables hold byte offsets (image within the tile's group + in-plane part) below 2^31 and were sized
    // for the expected DMA widths; a chunk's channel offset (soffset) stays below 2^32 bytes
    const long NIm1 = p.NI - 1;
    const bool small1 = (NIm1 * a.C1 + 1) * a.H1 * a.W1 < (1L << 29) && (long)a.CK * a.H1 * a.W1 < (1L << 29);
    const bool small2 = (NIm1 * a.C2 + 1) * a.Hin * a.Win < (1L << 29) && (long)a.CK * a.Hin * a.Win < (1L << 29);
    static const int tabmode = dbg_int("SPRK_XTAB", 2);   // debug: 0 = pointer-arithmetic staging, 1 = not for 1x1
    const bool want = tabmode == 2 || (tabmode == 1 && a.KH * a.KW > 1);
    a.xtab = (p.xtab && want && small1 && small2 && (a.up1 || a.vec1) && (a.C2 == 0 || a.vec2)) ? 1 : 0;
    a.nG1 = p.nG1;
    a.nG2 = p.nG2;
}

int transform_weights(const float *w, float *ws, int Cout, int Cin, int KHW, int mode, const FwdPlan &p,
                      hipStream_t s) {
    const long total = (long)p.nblkN * p.rows * p.ldw + kZeroFloats;
    return sprk::wprep_site(sprk::wprep_item(sprk::WPREP_DIRECT, w, ws, total,
                                             {Cout, Cin, KHW, mode, p.CK, p.R4, p.rows, p.NT * 16, p.ldw, p.nblkN}), s);
}

int check_geom(const sprk_conv_geom *g) {
    SPRK_REQUIRE(g, "conv: null geometry");
    SPRK_REQUIRE(g->N > 0 && g->C1 >= 0 && g->C2 >= 0 && g->C1 + g->C2 > 0 && g->Cout > 0, "conv: bad channel/batch counts");
    SPRK_REQUIRE(g->Hin > 0 && g->Win > 0 && g->Hout > 0 && g->Wout > 0, "conv: bad spatial dims");
    SPRK_REQUIRE(g->KH > 0 && g->KW > 0 && g->stride > 0 && g->dil > 0, "conv: bad kernel params");
    SPRK_REQUIRE(!g->up1 || (g->Hin % 2 == 0 && g->Win % 2 == 0), "conv: up1 needs even Hin/Win");
    SPRK_REQUIRE((long)g->N * (g->C1 + g->C2) * g->Hin * g->Win < (1L << 40), "conv: tensor too large");
    return SPRK_OK;
}

struct WgPlan {
    int IT, NT, lgTC, lgTR, tilesX, tilesY, nTiles, tilesPerGroup, groups;
    int CKW, nChunks, nblkN, ioffN, CoutP;
    int inRows, inCols, pitch, cplane, colOff, xrow;
    int mode, nG1, nG2;
    size_t ldsBytes, wsBytes;
};

bool plan_wgrad(const sprk_conv_geom *g, WgPlan *p) {
    const int KHW = g->KH * g->KW, Cin = g->C1 + g->C2;
    const int ntile = sprk::cdiv(g->Cout, 16);
    const int NT = pick_nt(ntile);
    p->NT = NT;
    p->nblkN = sprk::cdiv(ntile, NT);
    p->CoutP = p->nblkN * NT * 16;
    const int TM = 64;
    int TC = std::min(sprk::pow2_ceil(g->Wout), 64);
    if (g->dil * (g->KW - 1) >= 8) TC = std::min(TC, 8);
    int TR = std::min(TM / TC, sprk::pow2_ceil(g->Hout));
    int NI = TM / (TC * TR);
    p->lgTC = sprk::ilog2(TC);
    p->lgTR = sprk::ilog2(TR);
    p->tilesX = sprk::cdiv(g->Wout, TC);
    p->tilesY = sprk::cdiv(g->Hout, TR);
    p->nTiles = sprk::cdiv(g->N, NI) * p->tilesX * p->tilesY;
    p->inRows = (TR - 1) * g->stride + (g->KH - 1) * g->dil + 1;
    p->inCols = (TC - 1) * g->stride + (g->KW - 1) * g->dil + 1;
    p->colOff = ((TC * g->stride) % 4 == 0) ? (((-g->pad_left) % 4) + 4) % 4 : 0;
    p->pitch = sprk::roundup(p->colOff + p->inCols, 4);
    // channel stride: 16-byte DMA needs a multiple of 4 floats (then 16 channels x 2 pixels of one A read
    // share banks pairwise: 2-way); the 4-byte DMA fallback can use the conflict-free stride = 2 (mod 32)
    const bool vecGeo = (TC * g->stride) % 4 == 0 && (g->Win % 4) == 0;
    p->cplane = pad_to_residue(NI * p->inRows * p->pitch, vecGeo ? 4 : 2);
    // 1x1, stride 1, unpadded, single full-resolution source: the x tile is 64 contiguous pixels per
    // channel, staged like the G tile (swizzled 64-float rows, 16-byte DMA)
    p->xrow = (KHW == 1 && g->stride == 1 && g->pad_top == 0 && g->pad_left == 0 && !g->up1 && g->C2 == 0 &&
               TC >= 4 && (g->Win % 4) == 0 && g->Hin == g->Hout && g->Win == g->Wout)
                  ? 1
                  : 0;
    if (p->xrow) {
        p->cplane = 64;
        p->colOff = 0;
    }
    // channel chunks: each wave (wi = 0..3) owns k-tiles wi, wi+4, ... of a chunk's ceil(CKW*KHW/16) k-tiles, at
    // most kMaxIT of them.  Cost of a split = chunks x busiest wave's k-tiles (+ a little per chunk for staging
    // the G tile again); equal-sized chunks so every workgroup carries the same load.
    constexpr int kMaxIT = 7;
    // operand addressing of the tile loop: 1 = 1x1 rows, 2 = stride-1 tiles with rows of >= 4 pixels, 0 = any
    // (MODE 2 also stages the x planes through offset tables and 16-byte buffer loads: aligned rows, no upsampling)
    p->mode = p->xrow ? 1 : (g->stride == 1 && TC >= 4 && vecGeo && !g->up1) ? 2 : 0;
    {
        const int planeElems = NI * p->inRows * p->pitch;
        p->nG1 = p->mode == 2 ? sprk::cdiv(planeElems, 256) : 0;
        p->nG2 = (p->mode == 2 && g->C2 > 0) ? sprk::cdiv(planeElems, 256) : 0;
        if (p->inRows >= 1024 || p->pitch >= 1024) p->mode = p->xrow ? 1 : 0, p->nG1 = p->nG2 = 0;
    }
    auto lds = [&](int ck) {
        const int ioffN = sprk::roundup(sprk::roundup(ck * KHW, 16), 4);
        return (size_t)(ioffN + 64 + 16 + 2 * (p->nG1 + p->nG2) * 64 + 2 * (ck * p->cplane + NT * 16 * 64)) * 4;
    };
    // The cost of a split is the time of ONE workgroup (they all run at once: never more workgroups than CUs, see
    // below): tiles per workgroup x (the busiest wave's k-tiles + a little for staging the G tile again).  With many
    // tiles that is proportional to chunks x k-tiles, the total work; with few tiles (the U-Net levels <= 16x16: every
    // workgroup has one tile whatever the split) it is the k-tiles alone, so small planes are cut into more, shorter
    // chunks on more CUs (96->96 at 64x8x8: 2 chunks of 7 k-tiles per wave -> 4 of 4).
    static const int kWgKnob = dbg_int("SPRK_WG_BLOCKS", 0);   // debug knob (sweeps); default: one workgroup per CU
    const int kWgBlocks = kWgKnob > 0 ? kWgKnob : sprk::num_cus();
    int CKW = 0;
    long bestCost = 1L << 60;
    for (int nCh = 1; nCh <= Cin; ++nCh) {
        const int ck = sprk::cdiv(Cin, nCh);
        if (sprk::cdiv(Cin, ck) != nCh) continue;
        const int itw = sprk::cdiv(sprk::cdiv(ck * KHW, 16), 4);
        if (itw > kMaxIT || lds(ck) > 150 * 1024 || (p->xrow && ck > 192)) continue;
        const int g = std::max(1, std::min(p->nTiles, kWgBlocks / (nCh * p->nblkN)));
        const long cost = (long)sprk::cdiv(p->nTiles, g) * (4 * itw + 1);
        if (cost < bestCost) {
            bestCost = cost;
            CKW = ck;
        }
        if (itw == 1) break;
    }
    if (CKW == 0 || lds(CKW) > kLdsLimit) return false;
    p->CKW = CKW;
    p->nChunks = sprk::cdiv(Cin, CKW);
    p->ioffN = sprk::roundup(sprk::roundup(CKW * KHW, 16), 4);
    p->IT = sprk::cdiv(sprk::cdiv(CKW * KHW, 16), 4);
    p->ldsBytes = lds(CKW);
    // one workgroup per CU (the stages fill most of the LDS): never more workgroups than CUs, or the
    // surplus runs as a second, nearly empty round
    const int per = p->nChunks * p->nblkN;
    int groups = std::max(1, std::min(p->nTiles, kWgBlocks / per));
    p->tilesPerGroup = sprk::cdiv(p->nTiles, groups);
    p->groups = sprk::cdiv(p->nTiles, p->tilesPerGroup);
    p->wsBytes = ((size_t)kZeroFloats + (size_t)p->groups * Cin * KHW * p->CoutP) * sizeof(float);
    return true;
}

template <int IT, int NT, int MODE>
int launch_wg_one(const WgArgs &a, const WgPlan &p, dim3 grid, hipStream_t s) {
    constexpr int WJ = (NT % 2 == 0) ? 2 : 1;   // two waves per SIMD whenever the cout tiles split evenly
    if (int rc = set_lds(conv_wgrad_mfma_kernel<IT, NT, WJ, MODE>, p.ldsBytes)) return rc;
    hipLaunchKernelGGL((conv_wgrad_mfma_kernel<IT, NT, WJ, MODE>), grid, dim3(256 * WJ), p.ldsBytes, s, a);
    return SPRK_OK;
}

template <int IT, int MODE>
int launch_wg_nt(const WgArgs &a, const WgPlan &p, dim3 grid, hipStream_t s) {
    switch (p.NT) {
        case 1: return launch_wg_one<IT, 1, MODE>(a, p, grid, s);
        case 2: return launch_wg_one<IT, 2, MODE>(a, p, grid, s);
        case 3: return launch_wg_one<IT, 3, MODE>(a, p, grid, s);
        case 4: return launch_wg_one<IT, 4, MODE>(a, p, grid, s);
        default: return launch_wg_one<IT, 6, MODE>(a, p, grid, s);
    }
}

template <int MODE>
int launch_wg(const WgArgs &a, const WgPlan &p, dim3 grid, hipStream_t s) {
    switch (p.IT) {
        case 1: return launch_wg_nt<1, MODE>(a, p, grid, s);
        case 2: return launch_wg_nt<2, MODE>(a, p, grid, s);
        case 3: return launch_wg_nt<3, MODE>(a, p, grid, s);
    }
    if constexpr (MODE != 1) {   // the row-staged 1x1 form never needs more than 192 channels = 3 k-tiles per wave
        switch (p.IT) {
            case 4: return launch_wg_nt<4, MODE>(a, p, grid, s);
            case 5: return launch_wg_nt<5, MODE>(a, p, grid, s);
            case 6: return launch_wg_nt<6, MODE>(a, p, grid, s);
            case 7: return launch_wg_nt<7, MODE>(a, p, grid, s);
        }
    }
    sprk::set_error("conv2d_bwd_weight: no kernel for %d k-tiles per wave", p.IT);
    return SPRK_EINVAL;
}

// Winograd path (wino.hip): geometry of the forward layer / of its backward-data correlation
sprk::WinoGeom wino_geom_fwd(const sprk_conv_geom *g, const sprk_conv_epilogue *ep) {
    return sprk::WinoGeom{g->N, g->C1, g->C2, g->Cout, g->Hin, g->Win, g->Hout, g->Wout, g->KH, g->KW, g->stride, g->dil,
                          g->pad_top, g->pad_left, g->up1, ep ? ep->up2 : 0, (ep && ep->res) ? 1 : 0, (g->dtype & SPRK_DT_PIN) ? 1 : 0};
}
sprk::WinoGeom wino_geom_bwd(const sprk_conv_geom *g) {
    return sprk::WinoGeom{g->N, g->Cout, 0, g->C1 + g->C2, g->Hout, g->Wout, g->Hin, g->Win, g->KH, g->KW, g->stride, g->dil,
                          (g->KH - 1) * g->dil - g->pad_top, (g->KW - 1) * g->dil - g->pad_left, g->up1, 0, 0, (g->dtype & SPRK_DT_PIN) ? 1 : 0};
}
// the 16-bit-operand kernels' view of a forward call / of a backward-data call (a forward-shaped convolution of gy
// with the flipped, channel-transposed taps and mirrored padding)
sprk::Conv16Call call16_fwd(const sprk_conv_geom *g, const sprk_conv_epilogue *ep) {
    sprk::Conv16Call c{};
    c.dtype = g->dtype; c.mode = 0;
    c.x16 = (g->dtype & SPRK_DT_X16) ? 1 : 0; c.y16 = (g->dtype & SPRK_DT_Y16) ? 1 : 0;
    c.N = g->N; c.C1 = g->C1; c.C2 = g->C2; c.Hin = g->Hin; c.Win = g->Win; c.Cout = g->Cout; c.Hout = g->Hout;
    c.Wout = g->Wout; c.KH = g->KH; c.KW = g->KW; c.stride = g->stride; c.dil = g->dil; c.padT = g->pad_top;
    c.padL = g->pad_left; c.up1 = g->up1;
    if (ep) {
        c.up2 = ep->up2; c.res = ep->res != nullptr; c.act = ep->act; c.bias = ep->bias; c.scale = ep->scale;
        c.shift = ep->shift;
    }
    c.kclass = kClass16;
    c.flops = 2.0 * g->N * g->Hout * g->Wout * (double)g->Cout * (g->C1 + g->C2) * g->KH * g->KW;
    return c;
}
sprk::Conv16Call call16_bwd(const sprk_conv_geom *g) {
    sprk::Conv16Call c{};
    c.dtype = g->dtype; c.mode = 1;
    c.x16 = (g->dtype & SPRK_DT_X16) ? 1 : 0; c.y16 = (g->dtype & SPRK_DT_Y16) ? 1 : 0;    // gy / gin
    c.N = g->N; c.C1 = g->Cout; c.C2 = 0; c.Hin = g->Hout; c.Win = g->Wout; c.Cout = g->C1 + g->C2; c.Hout = g->Hin;
    c.Wout = g->Win; c.KH = g->KH; c.KW = g->KW; c.stride = g->stride; c.dil = g->dil;
    c.padT = (g->KH - 1) * g->dil - g->pad_top; c.padL = (g->KW - 1) * g->dil - g->pad_left; c.up1 = g->up1;
    c.act = SPRK_ACT_NONE;
    c.kclass = kClass16;
    c.flops = 2.0 * g->N * g->Hout * g->Wout * (double)g->Cout * (g->C1 + g->C2) * g->KH * g->KW;
    return c;
}

sprk::Wgrad16Call call16_wgrad(const sprk_conv_geom *g) {
    sprk::Wgrad16Call c{};
    c.dtype = g->dtype;
    c.x16 = (g->dtype & SPRK_DT_X16) ? 1 : 0;
    c.N = g->N; c.C1 = g->C1; c.C2 = g->C2; c.H = g->Hin; c.W = g->Win; c.Cout = g->Cout; c.Hout = g->Hout;
    c.Wout = g->Wout; c.KH = g->KH; c.KW = g->KW; c.stride = g->stride; c.dil = g->dil; c.padT = g->pad_top;
    c.padL = g->pad_left; c.up1 = g->up1;
    c.kclass = kClass16W;
    c.flops = 2.0 * g->N * g->Hout * g->Wout * (double)g->Cout * (g->C1 + g->C2) * g->KH * g->KW;
    return c;
}

size_t wino_ws_fwd(const sprk_conv_geom *g) {
    return sprk::wino_eligible(wino_geom_fwd(g, nullptr)) ? sprk::wino_ws_bytes(g->C1, g->C2, g->Cout) : 0;
}
size_t wino_ws_bwd(const sprk_conv_geom *g) {
    return sprk::wino_eligible(wino_geom_bwd(g)) ? sprk::wino_ws_bytes(g->Cout, 0, g->C1 + g->C2) : 0;
}
constexpr int kClassWino = 3;   // profiling class of the 96-channel Winograd kernel

}  // namespace

// ==========================================================================================
// C ABI
// ==========================================================================================
extern "C" {

#ifdef SPRK_DIAG
// diagnostic builds only: the phase time stamps workgroup 0 of the last conv_mfma_kernel launch left (10 ns units)
int sprk_diag_conv_clock(long long *out8) {
    return hipMemcpyFromSymbol(out8, HIP_SYMBOL(sprk_diag_conv_clk), 40 * sizeof(long long)) == hipSuccess ? SPRK_OK : SPRK_ELAUNCH;
}
#endif

// Which calls of a layer exist for 16-bit ACTIVATION tensors (SPRK_DT_X16 / SPRK_DT_Y16)?  Bit 0: forward, bit 1:
// backward-data, bit 2: backward-weight.  g->dtype carries the operand type; the storage bits are ignored (every
// input / output storage combination of an eligible call exists).  ep: the forward call's epilogue (may be null).
int sprk_conv2d_storage16(const sprk_conv_geom *g, const sprk_conv_epilogue *ep) {
    if (!g || check_geom(g) != SPRK_OK) return 0;
    const int dt = g->dtype & SPRK_DT_MASK;
    if ((dt != SPRK_DT_BF16 && dt != SPRK_DT_F16) || naive_of(g) || g->stride != 1) return 0;
    sprk_conv_geom q = *g;
    q.dtype = (g->dtype & ~(SPRK_DT_WPREP | SPRK_DT_WPREP_KIND(7))) | SPRK_DT_X16 | SPRK_DT_Y16;
    int r = 0;
    if (sprk::conv16_kind(call16_fwd(&q, ep)) >= 2) r |= 1;
    if (!g->up1 && sprk::conv16_kind(call16_bwd(&q)) >= 2) r |= 2;
    if (sprk::wgrad16_eligible(call16_wgrad(&q))) r |= 4;
    return r;
}

size_t sprk_conv2d_fwd_ws_bytes(const sprk_conv_geom *g) {
    if (!g) return 0;
    FwdPlan p;
    if (!plan_fwd(g->N, g->C1 + g->C2, g->Cout, g->Hout, g->Wout, g->KH, g->KW, g->stride, g->dil, g->pad_left, g->up1,
                  g->C2 > 0, g->Win, &p)) return 0;
    size_t need = std::max(p.wsBytes, wino_ws_fwd(g));
    if ((g->dtype & SPRK_DT_MASK) != SPRK_DT_F32) {
        // the query does not know the epilogue the call will carry: the maximum over the 16-bit plans it can select
        // (fused up-sampling and a residual exclude some kernels, and the remaining one may need more)
        for (int v = 0; v < 4; ++v) {
            sprk_conv_epilogue e = {nullptr, nullptr, nullptr, (v & 2) ? (const float *)g : nullptr, 0, 0, 0, SPRK_ACT_NONE, v & 1};
            need = std::max(need, sprk::conv16_ws_bytes(call16_fwd(g, &e)));
        }
    }
    return need;
}

static int conv2d_fwd_impl(const float *x, const float *x2, const float *w, float *y, const sprk_conv_geom *g,
                           const sprk_conv_epilogue *ep, void *ws, size_t ws_bytes, void *stream);

int sprk_conv2d_fwd(const float *x, const float *x2, const float *w, float *y, const sprk_conv_geom *g,
                    const sprk_conv_epilogue *ep, void *ws, size_t ws_bytes, void *stream) {
    sprk::WprepScope scope(nullptr, g ? g->dtype : 0);
    return scope.verify(conv2d_fwd_impl(x, x2, w, y, g, ep, ws, ws_bytes, stream));
}

int sprk_conv2d_fwd_wprep(const float *w, const sprk_conv_geom *g, const sprk_conv_epilogue *ep, void *ws,
                          size_t ws_bytes, sprk_wprep_item *item) {
    SPRK_REQUIRE(item && w && ws, "conv2d_fwd_wprep: null argument");
    *item = sprk_wprep_item{};
    sprk::WprepScope scope(item, 0);
    // the tensors are never touched in describe mode: the call ends at its weight-transform site
    const float *dummy = (const float *)ws;
    const int rc = conv2d_fwd_impl(dummy, g && g->C2 ? dummy : nullptr, w, (float *)ws, g, ep, ws, ws_bytes, nullptr);
    return rc == sprk::kWprepDescribed ? (int)SPRK_OK : rc;
}

static int conv2d_fwd_impl(const float *x, const float *x2, const float *w, float *y, const sprk_conv_geom *g,
                           const sprk_conv_epilogue *ep, void *ws, size_t ws_bytes, void *stream) {
    if (int rc = check_geom(g)) return rc;
    SPRK_REQUIRE(x && w && y, "conv2d_fwd: null tensor");
    SPRK_REQUIRE(g->C2 == 0 || x2, "conv2d_fwd: C2 > 0 but x2 is null");
    hipStream_t s = (hipStream_t)stream;
    sprk_conv_epilogue e0 = {nullptr, nullptr, nullptr, nullptr, 0, 0, 0, SPRK_ACT_NONE, 0};
    if (!ep) ep = &e0;
    SPRK_REQUIRE(!ep->scale || ep->shift, "conv2d_fwd: scale without shift");
    if (naive_of(g)) {
        if (sprk::wprep_describing()) return SPRK_OK;   // no weight transform on this path
        DirectArgs a{x, x2, w, nullptr, y, *g, *ep};
        const long total = (long)g->N * g->Cout * g->Hout * g->Wout;
        hipLaunchKernelGGL(conv_fwd_direct_kernel, dim3(sprk::ew_blocks(total)), dim3(256), 0, s, a);
        return sprk::check_launch("conv_fwd_direct");
    }
    if ((g->dtype & SPRK_DT_MASK) != SPRK_DT_F32 && !naive_of(g)) {
        const sprk::Conv16Call c16 = call16_fwd(g, ep);
        if (sprk::conv16_eligible(c16)) return sprk::conv16_run(c16, x, x2, w, y, ws, ws_bytes, s);
    }
    SPRK_REQUIRE(!(g->dtype & (SPRK_DT_X16 | SPRK_DT_Y16)),
                 "conv2d_fwd: no kernel takes this geometry with 16-bit activation tensors (ask sprk_conv2d_storage16 first)");
    if (sprk::wino_eligible(wino_geom_fwd(g, ep)) && !ep->res) {
        const size_t need = sprk::wino_ws_bytes(g->C1, g->C2, g->Cout);
        if (ws_bytes < need || !ws) {
            sprk::set_error("conv2d_fwd: workspace %zu < %zu", ws_bytes, need);
            return SPRK_EWORKSPACE;
        }
        sprk::WinoArgs wa{x, x2, w, ep->bias, ep->scale, ep->shift, y, (float *)ws, g->N, g->C1, g->C2, g->Hin, g->Win,
                          g->Cout, g->pad_top, g->pad_left, ep->act, 0, g->Cout > 48 ? kClassWino : 2,
                          2.0 * g->N * g->Hout * g->Wout * (double)g->Cout * (g->C1 + g->C2) * 9, ep->up2 ? 1 : 0};
        if (int rc = sprk::wino_conv(wa, s)) return rc;
        return sprk::check_launch("wino_conv");
    }
    FwdPlan p;
    SPRK_REQUIRE(plan_fwd(g->N, g->C1 + g->C2, g->Cout, g->Hout, g->Wout, g->KH, g->KW, g->stride, g->dil, g->pad_left, g->up1,
                  g->C2 > 0, g->Win, &p),
                 "conv2d_fwd: geometry does not fit LDS");
    if (ws_bytes < p.wsBytes || !ws) {
        sprk::set_error("conv2d_fwd: workspace %zu < %zu", ws_bytes, p.wsBytes);
        return SPRK_EWORKSPACE;
    }
    float *wsf = (float *)ws;
    if (int rc = transform_weights(w, wsf, g->Cout, g->C1 + g->C2, g->KH * g->KW, 0, p, s)) return rc;
    ConvArgs a{};
    a.x = x; a.x2 = x2; a.zeros = wsf; a.wT = wsf + kZeroFloats;
    a.bias = ep->bias; a.scale = ep->scale; a.shift = ep->shift; a.res = ep->res;
    a.y = y;
    a.N = g->N; a.C1 = g->C1; a.C2 = g->C2; a.Hin = g->Hin; a.Win = g->Win; a.up1 = g->up1;
    a.H1 = g->up1 ? g->Hin / 2 : g->Hin;
    a.W1 = g->up1 ? g->Win / 2 : g->Win;
    a.Cout = g->Cout; a.Hout = g->Hout; a.Wout = g->Wout;
    a.KH = g->KH; a.KW = g->KW; a.stride = g->stride; a.dil = g->dil; a.padT = g->pad_top; a.padL = g->pad_left;
    a.act = ep->act;
    a.resH = ep->res_h; a.resW = ep->res_w; a.resOff = ep->res_off;
    a.up2 = ep->up2;
    fill_args(a, p);
    a.vec4 = (g->Wout % 4 == 0) && (p.lgTC >= 2) && (((uintptr_t)y & 15) == 0);
    const double flops = 2.0 * g->N * g->Hout * g->Wout * (double)g->Cout * (g->C1 + g->C2) * g->KH * g->KW;
    const int kclass = (p.MT == 4 && p.NT == 6) ? 0 : 2;   // 0: the dominant instantiation conv_mfma_kernel<4, 6>
    sprk::prof_begin(kclass, flops, s);
    if (int rc = launch_fwd(a, p, s)) return rc;
    sprk::prof_end(kclass, s);
    return sprk::check_launch("conv_mfma");
}

size_t sprk_conv2d_bwd_data_ws_bytes(const sprk_conv_geom *g) {
    if (!g || g->stride != 1) return 0;
    FwdPlan p;
    if (!plan_fwd(g->N, g->Cout, g->C1 + g->C2, g->Hin, g->Win, g->KH, g->KW, 1, g->dil, (g->KW - 1) * g->dil - g->pad_left, 0, 0,
                  g->Wout, &p)) return 0;
    size_t need = std::max(p.wsBytes, wino_ws_bwd(g));
    if ((g->dtype & SPRK_DT_MASK) != SPRK_DT_F32) need = std::max(need, sprk::conv16_ws_bytes(call16_bwd(g)));
    return need;
}

int sprk_conv2d_bwd_data(const float *gy, const float *w, float *gin, const sprk_conv_geom *g, void *ws,
                         size_t ws_bytes, void *stream) {
    return sprk_conv2d_bwd_data_masked(gy, w, gin, g, nullptr, SPRK_ACT_NONE, ws, ws_bytes, stream);
}

// in-place mask of a finished gradient (the layers whose backward-data kernel has no masked epilogue)
// (with SPRK_DT_Y16 the gradient and the mask — the consumer's saved 16-bit input — are tensors of the operand type)
static int mask_in_place(float *gin, const float *mask_y, int mask_act, const sprk_conv_geom *g, hipStream_t s) {
    const int t = (g->dtype & SPRK_DT_Y16) ? (g->dtype & SPRK_DT_MASK) : 0;
    return sprk_act_bwd(gin, mask_y, gin, nullptr, mask_act, g->N, g->C1 + g->C2, g->Hin, g->Win, 0, 0, SPRK_IO3(t, t, t),
                        nullptr, 0, s);
}

static int conv2d_bwd_data_impl(const float *gy, const float *w, float *gin, const sprk_conv_geom *g,
                                const float *mask_y, int mask_act, void *ws, size_t ws_bytes, void *stream);

int sprk_conv2d_bwd_data_masked(const float *gy, const float *w, float *gin, const sprk_conv_geom *g,
                                const float *mask_y, int mask_act, void *ws, size_t ws_bytes, void *stream) {
    sprk::WprepScope scope(nullptr, g ? g->dtype : 0);
    return scope.verify(conv2d_bwd_data_impl(gy, w, gin, g, mask_y, mask_act, ws, ws_bytes, stream));
}

int sprk_conv2d_bwd_data_wprep(const float *w, const sprk_conv_geom *g, void *ws, size_t ws_bytes, sprk_wprep_item *item) {
    SPRK_REQUIRE(item && w && ws, "conv2d_bwd_data_wprep: null argument");
    *item = sprk_wprep_item{};
    sprk::WprepScope scope(item, 0);
    const int rc = conv2d_bwd_data_impl((const float *)ws, w, (float *)ws, g, nullptr, SPRK_ACT_NONE, ws, ws_bytes, nullptr);
    return rc == sprk::kWprepDescribed ? (int)SPRK_OK : rc;
}

static int conv2d_bwd_data_impl(const float *gy, const float *w, float *gin, const sprk_conv_geom *g,
                                const float *mask_y, int mask_act, void *ws, size_t ws_bytes, void *stream) {
    if (int rc = check_geom(g)) return rc;
    SPRK_REQUIRE(gy && w && gin, "conv2d_bwd_data: null tensor");
    SPRK_REQUIRE(mask_act == SPRK_ACT_NONE || (mask_y && !g->up1), "conv2d_bwd_data: mask needs the saved input, no upsampling");
    if (mask_act == SPRK_ACT_NONE) mask_y = nullptr;
    hipStream_t s = (hipStream_t)stream;
    const int Cin = g->C1 + g->C2;
    if (naive_of(g) || g->stride != 1) {
        if (sprk::wprep_describing()) return SPRK_OK;   // no weight transform on this path
        sprk_conv_epilogue e0 = {nullptr, nullptr, nullptr, nullptr, 0, 0, 0, SPRK_ACT_NONE, 0};
        DirectArgs a{nullptr, nullptr, w, gy, gin, *g, e0};
        const long total = (long)g->N * Cin * g->Hin * g->Win;
        hipLaunchKernelGGL(conv_bwd_data_direct_kernel, dim3(sprk::ew_blocks(total)), dim3(256), 0, s, a);
        if (int rc = sprk::check_launch("conv_bwd_data_direct")) return rc;
        return mask_y ? mask_in_place(gin, mask_y, mask_act, g, s) : (int)SPRK_OK;
    }
    // gin = correlation of gy with the flipped, channel-transposed kernel
    if ((g->dtype & SPRK_DT_MASK) != SPRK_DT_F32 && !naive_of(g)) {
        const sprk::Conv16Call c16 = call16_bwd(g);
        if (sprk::conv16_eligible(c16)) {
            // (a masked epilogue in conv16_tile_kernel<6> takes it from 231 to 256 VGPRs + 92 bytes of scratch — the next
            // tile's fetch is in flight during the stores — and the bf16 step from 11.22 to 11.27 ms: in-place pass)
            if (int rc = sprk::conv16_run(c16, gy, nullptr, w, gin, ws, ws_bytes, s)) return rc;
            return mask_y ? mask_in_place(gin, mask_y, mask_act, g, s) : (int)SPRK_OK;
        }
    }
    SPRK_REQUIRE(!(g->dtype & (SPRK_DT_X16 | SPRK_DT_Y16)),
                 "conv2d_bwd_data: no kernel takes this geometry with 16-bit activation tensors (ask sprk_conv2d_storage16 first)");
    if (sprk::wino_eligible(wino_geom_bwd(g))) {
        const size_t need = sprk::wino_ws_bytes(g->Cout, 0, Cin);
        if (ws_bytes < need || !ws) {
            sprk::set_error("conv2d_bwd_data: workspace %zu < %zu", ws_bytes, need);
            return SPRK_EWORKSPACE;
        }
        sprk::WinoArgs wa{gy, nullptr, w, nullptr, nullptr, nullptr, gin, (float *)ws, g->N, g->Cout, 0, g->Hout, g->Wout,
                          Cin, (g->KH - 1) * g->dil - g->pad_top, (g->KW - 1) * g->dil - g->pad_left, SPRK_ACT_NONE, 1,
                          Cin > 48 ? kClassWino : 2, 2.0 * g->N * g->Hout * g->Wout * (double)g->Cout * Cin * 9};
        wa.mask = mask_y;          // d act / d y of the layer that produced this conv's input, applied in the output
        wa.mask_act = mask_act;    // transform: gin leaves the kernel as that layer's pre-activation gradient
        if (int rc = sprk::wino_conv(wa, s)) return rc;
        return sprk::check_launch("wino_conv(bwd_data)");
    }
    FwdPlan p;
    SPRK_REQUIRE(plan_fwd(g->N, g->Cout, Cin, g->Hin, g->Win, g->KH, g->KW, 1, g->dil, (g->KW - 1) * g->dil - g->pad_left, 0, 0,
                          g->Wout, &p),
                 "conv2d_bwd_data: geometry does not fit LDS");
    if (ws_bytes < p.wsBytes || !ws) {
        sprk::set_error("conv2d_bwd_data: workspace %zu < %zu", ws_bytes, p.wsBytes);
        return SPRK_EWORKSPACE;
    }
    float *wsf = (float *)ws;
    if (int rc = transform_weights(w, wsf, g->Cout, Cin, g->KH * g->KW, 1, p, s)) return rc;
    ConvArgs a{};
    a.x = gy; a.x2 = nullptr; a.zeros = wsf; a.wT = wsf + kZeroFloats; a.y = gin;
    a.N = g->N; a.C1 = g->Cout; a.C2 = 0; a.Hin = g->Hout; a.Win = g->Wout; a.up1 = 0;
    a.H1 = g->Hout; a.W1 = g->Wout;
    a.Cout = Cin; a.Hout = g->Hin; a.Wout = g->Win;
    a.KH = g->KH; a.KW = g->KW; a.stride = 1; a.dil = g->dil;
    a.padT = (g->KH - 1) * g->dil - g->pad_top;
    a.padL = (g->KW - 1) * g->dil - g->pad_left;
    a.act = SPRK_ACT_NONE;
    fill_args(a, p);
    a.vec4 = (g->Win % 4 == 0) && (p.lgTC >= 2) && (((uintptr_t)gin & 15) == 0);
    // (a masked epilogue in this kernel costs 26 VGPRs — 160 -> 186, two workgroups per CU instead of three — for
    // every call, masked or not: measured in round 1 and again in round 3; the mask is applied by an in-place pass)
    const double flops = 2.0 * g->N * g->Hout * g->Wout * (double)g->Cout * Cin * g->KH * g->KW;
    const int kclass = (p.MT == 4 && p.NT == 6) ? 0 : 2;
    sprk::prof_begin(kclass, flops, s);
    if (int rc = launch_fwd(a, p, s)) return rc;
    sprk::prof_end(kclass, s);
    if (int rc = sprk::check_launch("conv_mfma(bwd_data)")) return rc;
    return mask_y ? mask_in_place(gin, mask_y, mask_act, g, s) : (int)SPRK_OK;
}

size_t sprk_conv2d_bwd_weight_ws_bytes(const sprk_conv_geom *g) {
    if (!g) return 0;
    const size_t wino = sprk::wino_wgrad_eligible(wino_geom_fwd(g, nullptr))
                            ? sprk::wino_wgrad_ws_bytes(g->C1, g->C2, g->Cout) : 0;
    const size_t w16 = (g->dtype & SPRK_DT_MASK) != SPRK_DT_F32 ? sprk::wgrad16_ws_bytes(call16_wgrad(g)) : 0;
    WgPlan p;
    if (!plan_wgrad(g, &p)) return std::max(wino, w16);
    return std::max(std::max(p.wsBytes, wino), w16);
}

int sprk_conv2d_bwd_weight(const float *x, const float *x2, const float *gy, float *gw, const sprk_conv_geom *g,
                           void *ws, size_t ws_bytes, void *stream) {
    return sprk_conv2d_bwd_weight_partial(x, x2, gy, gw, g, ws, ws_bytes, nullptr, stream);
}

// item == nullptr: finish the sum over the partial buffers now; otherwise describe it in *item (sprk.h)
int sprk_conv2d_bwd_weight_partial(const float *x, const float *x2, const float *gy, float *gw, const sprk_conv_geom *g,
                                   void *ws, size_t ws_bytes, sprk_reduce_item *item, void *stream) {
    if (item) *item = sprk_reduce_item{nullptr, nullptr, SPRK_RED_NONE, 0, 0, 0, 0, 0};
    if (int rc = check_geom(g)) return rc;
    SPRK_REQUIRE(x && gy && gw, "conv2d_bwd_weight: null tensor");
    SPRK_REQUIRE(g->C2 == 0 || x2, "conv2d_bwd_weight: C2 > 0 but x2 is null");
    hipStream_t s = (hipStream_t)stream;
    const int Cin = g->C1 + g->C2;
    if ((g->dtype & SPRK_DT_MASK) != SPRK_DT_F32 && !naive_of(g)) {
        const sprk::Wgrad16Call c16 = call16_wgrad(g);
        if (sprk::wgrad16_eligible(c16)) return sprk::wgrad16_run(c16, x, x2, gy, gw, ws, ws_bytes, item, s);
    }
    SPRK_REQUIRE(!(g->dtype & (SPRK_DT_X16 | SPRK_DT_Y16)),
                 "conv2d_bwd_weight: no kernel takes this geometry with 16-bit activation tensors (ask sprk_conv2d_storage16 first)");
    if (!naive_of(g) && sprk::wino_wgrad_eligible(wino_geom_fwd(g, nullptr))) {
        const size_t need = sprk::wino_wgrad_ws_bytes(g->C1, g->C2, g->Cout);
        if (ws_bytes < need || !ws) {
            sprk::set_error("conv2d_bwd_weight: workspace %zu < %zu", ws_bytes, need);
            return SPRK_EWORKSPACE;
        }
        sprk::WinoWgArgs wa{x, x2, gy, gw, (float *)ws, g->N, g->C1, g->C2, g->Hin, g->Win, g->Cout, g->pad_top, g->pad_left, 4,
                            2.0 * g->N * g->Hout * g->Wout * (double)g->Cout * Cin * 9};   // profiling class 4
        return sprk::wino_wgrad(wa, s);
    }
    WgPlan p;
    const bool ok = plan_wgrad(g, &p);
    if (naive_of(g) || !ok) {
        sprk_conv_epilogue e0 = {nullptr, nullptr, nullptr, nullptr, 0, 0, 0, SPRK_ACT_NONE, 0};
        DirectArgs a{x, x2, nullptr, gy, gw, *g, e0};
        hipLaunchKernelGGL(conv_bwd_weight_direct_kernel, dim3(g->Cout * Cin), dim3(256), 0, s, a);
        return sprk::check_launch("conv_bwd_weight_direct");
    }
    if (ws_bytes < p.wsBytes || !ws) {
        sprk::set_error("conv2d_bwd_weight: workspace %zu < %zu", ws_bytes, p.wsBytes);
        return SPRK_EWORKSPACE;
    }
    float *wsf = (float *)ws;
    const float *zeros = nullptr;  // device-resident, never written: no per-call memset
    if (hipGetSymbolAddress((void **)&zeros, HIP_SYMBOL(g_zero_block)) != hipSuccess || !zeros) {
        sprk::set_error("conv2d_bwd_weight: zero block unavailable");
        return SPRK_ELAUNCH;
    }
    WgArgs a{};
    a.x = x; a.x2 = x2; a.gy = gy; a.zeros = zeros; a.partial = wsf + kZeroFloats;
    a.N = g->N; a.C1 = g->C1; a.C2 = g->C2; a.Hin = g->Hin; a.Win = g->Win; a.up1 = g->up1;
    a.H1 = g->up1 ? g->Hin / 2 : g->Hin;
    a.W1 = g->up1 ? g->Win / 2 : g->Win;
    a.Cout = g->Cout; a.CoutP = p.CoutP; a.Hout = g->Hout; a.Wout = g->Wout;
    a.KH = g->KH; a.KW = g->KW; a.stride = g->stride; a.dil = g->dil; a.padT = g->pad_top; a.padL = g->pad_left;
    a.lgTC = p.lgTC; a.lgTR = p.lgTR; a.tilesX = p.tilesX; a.tilesY = p.tilesY; a.nTiles = p.nTiles;
    a.tilesPerGroup = p.tilesPerGroup;
    a.CKW = p.CKW; a.ioffN = p.ioffN;
    a.inRows = p.inRows; a.inCols = p.inCols; a.pitch = p.pitch; a.cplane = p.cplane; a.colOff = p.colOff;
    a.invImg = 1.0f / (float)(p.inRows * p.pitch);
    a.invPitch = 1.0f / (float)p.pitch;
    {
        const bool geo = ((1 << p.lgTC) * g->stride) % 4 == 0 && (g->Win % 4) == 0;
        a.vec1 = (geo && !g->up1 && aligned16(x)) ? 1 : 0;
        a.vec2 = (geo && x2 && aligned16(x2)) ? 1 : 0;
        a.deal = 1;
    }
    a.xrow = p.xrow;
    a.g4 = (p.lgTC >= 2 && (g->Wout % 4) == 0 && ((uintptr_t)gy & 15) == 0) ? 1 : 0;
    // buffer-load staging of the row tiles needs 16-byte aligned rows and 32-bit byte offsets inside one tile
    // group and one channel block; otherwise the general kernel (MODE 0) runs
    const long planeI = (long)g->Hin * g->Win, planeO = (long)g->Hout * g->Wout;
    const int NIw = 64 >> (p.lgTC + p.lgTR);
    const bool fits = (long)NIw * Cin * planeI < (1L << 28) && (long)NIw * g->Cout * planeO < (1L << 28) &&
                      (long)p.CKW * planeI < (1L << 28) && (long)p.NT * 16 * planeO < (1L << 28);
    static const int rowbuf = dbg_int("SPRK_WG_ROWBUF", 1);
    const bool fast = rowbuf && fits && a.g4;
    a.xtab = fast ? 1 : 0;
    if (a.xrow && (!fast || (((uintptr_t)x & 15) != 0))) {
        sprk::set_error("conv2d_bwd_weight: 1x1 row staging needs 16-byte aligned x / gy and < 2^28-element tiles");
        return SPRK_EINVAL;
    }
    // MODE 2 additionally needs 16-byte aligned x / x2 (the general kernel ignores the table area in LDS)
    const bool planesOk = a.vec1 && (g->C2 == 0 || a.vec2);
    const int mode = !fast ? 0 : (p.mode == 2 && !planesOk) ? 0 : p.mode;
    static const int wg_diag = sprk::diag_env("SPRK_WG_DIAG");
    a.diag = wg_diag;
    a.nG1 = p.nG1;
    a.nG2 = p.nG2;
    dim3 grid(p.groups, p.nChunks, p.nblkN);
    const double flops = 2.0 * g->N * g->Hout * g->Wout * (double)g->Cout * Cin * g->KH * g->KW;
    sprk::prof_begin(1, flops, s);
    const int rc = mode == 1 ? launch_wg<1>(a, p, grid, s) : mode == 2 ? launch_wg<2>(a, p, grid, s) : launch_wg<0>(a, p, grid, s);
    if (rc) return rc;
    sprk::prof_end(1, s);
    if (int rc2 = sprk::check_launch("conv_wgrad_mfma")) return rc2;
    const int K = Cin * g->KH * g->KW;
    const sprk_reduce_item it{wsf + kZeroFloats, gw, SPRK_RED_WGRAD, p.groups, 0, K, g->Cout, p.CoutP};
    return sprk::finish_or_defer(it, item, s);
}

static int act_nsplit(int N, int C) {
    int ns = 1;
    while (ns < N && (long)C * ns < 1024) ns <<= 1;
    return std::min(ns, N);
}

size_t sprk_act_bwd_ws_bytes(int N, int C, int HW) {
    (void)HW;
    return (size_t)C * act_nsplit(N, C) * sizeof(float);
}

int sprk_act_bwd(const void *g, const void *y, void *gpre, float *gbias, int act, int N, int C, int H, int W,
                 int up2, long g_image_stride, int io, void *ws, size_t ws_bytes, void *stream) {
    return sprk_act_bwd_partial(g, y, gpre, gbias, act, N, C, H, W, up2, g_image_stride, io, ws, ws_bytes, nullptr, stream);
}

int sprk_act_bwd_partial(const void *g, const void *y, void *gpre, float *gbias, int act, int N, int C, int H, int W,
                         int up2, long g_image_stride, int io, void *ws, size_t ws_bytes, sprk_reduce_item *item,
                         void *stream) {
    if (item) *item = sprk_reduce_item{nullptr, nullptr, SPRK_RED_NONE, 0, 0, 0, 0, 0};
    SPRK_REQUIRE(g && N > 0 && C > 0 && H > 0 && W > 0, "act_bwd: bad arguments");
    SPRK_REQUIRE(act == SPRK_ACT_NONE || (y && gpre), "act_bwd: activation needs the saved output and gpre");
    SPRK_REQUIRE(!up2 || (gpre && gpre != g), "act_bwd: up2 needs a separate low-resolution gpre");
    hipStream_t s = (hipStream_t)stream;
    const bool retype = gpre && (io & 15) != ((io >> 8) & 15);    // gpre has another storage type than gy
    SPRK_REQUIRE(!retype || gpre != g, "act_bwd: a change of storage type needs a separate gpre");
    if (act == SPRK_ACT_NONE && !gbias && !up2 && !retype) return SPRK_OK;
    const int ns = act_nsplit(N, C);
    if (gbias && (ws_bytes < (size_t)C * ns * sizeof(float) || !ws)) {
        sprk::set_error("act_bwd: workspace too small");
        return SPRK_EWORKSPACE;
    }
    const long dense = (long)C * H * W * (up2 ? 4 : 1);
    const long gs = g_image_stride ? g_image_stride : dense;
    SPRK_REQUIRE(gs >= dense, "act_bwd: image stride of gy smaller than one image");
    SPRK_REQUIRE(gs == dense || (gpre && gpre != g), "act_bwd: a strided gy needs a separate gpre");
    const int drc = dispatch_io3(io, [&](auto tg, auto ty, auto to) {
        using TG = decltype(tg);
        using TY = decltype(ty);
        using TO = decltype(to);
        hipLaunchKernelGGL((act_bwd_kernel<TG, TY, TO>), dim3(C, ns), dim3(256), 0, s, (const TG *)g, (const TY *)y, (TO *)gpre,
                           gbias ? (float *)ws : nullptr, act, N, C, H * W, W, up2, ns, gs);
        return 0;
    });
    SPRK_REQUIRE(drc == 0, "act_bwd: bad storage types (io)");
    if (int rc = sprk::check_launch("act_bwd")) return rc;
    if (gbias) {
        const sprk_reduce_item it{(const float *)ws, gbias, SPRK_RED_COLS, ns, C, 0, 0, 0};
        return sprk::finish_or_defer(it, item, s);
    }
    return SPRK_OK;
}

int sprk_concat_up_bwd(const float *gin, float *ga, float *gb, int N, int C1, int C2, int H, int W, int up1,
                       void *stream) {
    SPRK_REQUIRE(gin && N > 0 && C1 >= 0 && C2 >= 0 && H > 0 && W > 0, "concat_up_bwd: bad arguments");
    SPRK_REQUIRE((C1 == 0 || ga) && (C2 == 0 || gb), "concat_up_bwd: null output");
    SPRK_REQUIRE(!up1 || (H % 2 == 0 && W % 2 == 0), "concat_up_bwd: odd size with upsampling");
    const long total = (long)N * C1 * (up1 ? (H / 2) * (W / 2) : H * W) + (long)N * C2 * H * W;
    hipLaunchKernelGGL(concat_up_bwd_kernel, dim3(sprk::ew_blocks(total)), dim3(256), 0, (hipStream_t)stream, gin, ga,
                       gb, N, C1, C2, H, W, up1);
    return sprk::check_launch("concat_up_bwd");
}

}  // extern "C"
