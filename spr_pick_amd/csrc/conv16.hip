// 16-bit-operand convolutions of libsprk.so (gfx950 / MI355X): forward and backward-data of the U-Net layers on
// v_mfma_f32_16x16x32_{bf16,f16} (BASELINE configs[4]: "fp16 MFMA conv").
//
// What is 16-bit and what is not.  Master weights, accumulation and the epilogue arithmetic are fp32; the two MFMA
// operands are bf16 / fp16 (round-to-nearest-even, v_cvt_pk_*), products are exact in fp32 and accumulated in fp32 by
// the matrix core.  Per output the error is that of rounding each input once: |err| <= 2 u * sum|a_k w_k| with
// u = 2^-8 (bf16) or 2^-11 (fp16), the unit roundoffs of 8 / 11 significant bits — the bound the parity tests use.
// The ACTIVATION tensors in HBM are fp32 (SPRK_DT_BF16 / SPRK_DT_F16 alone: operands rounded on their way into the
// matrix core) or, with SPRK_DT_X16 / SPRK_DT_Y16, tensors of the operand type themselves (round 4): the input is
// then fetched two pixels per dword and goes to LDS without a conversion, the output is rounded once at the store —
// half the bytes on both sides of kernels whose bound is HBM.  conv16_tile_kernel (3x3) and conv16_head_kernel (1x1)
// exist in all four input / output storage combinations; conv16_mfma_kernel (small grids) is fp32-storage only.
//
// Same implicit GEMM as conv.hip:  Out[pixel][n] += A[pixel][k] * Wt[k][n],  k = (tap, channel).
//   * the input tile (with halo, zeros outside the image) is staged exactly as in conv_mfma_kernel: fp32, by
//     buffer_load ... lds through the per-workgroup offset tables (conv_dev.h), double buffered;
//   * a k-step is 32 deep: lane (pixel l&15, quarter lq = l>>4) supplies 8 consecutive k.  k is ordered in
//     GROUPS of 8 channels at one tap, so a lane's 8 values are 8 channel planes at one (pixel + tap) position:
//     8 ds_read_b32 (plane stride apart) + 4 v_cvt_pk — the conversion happens here, on the way into the matrix
//     core;  the four quarters of a k-step take four consecutive groups (group -> (tap, channel group) through a
//     small LDS table, so chunks of 8, 16, 32 channels and 1x1 / 3x3 taps all use the same loop);
//   * weights are converted once per call by weight_transform16_kernel into [N-block][chunk][group][n][8] 16-bit
//     slabs: the B operand of (group, n) is one 16-byte ds_read_b128, rows of 16 n are 256 B apart (conflict free);
//   * a channel count of 8m + 1 (the raw image concatenated to 96 features; the 1-channel first layer) puts the
//     odd channel in a group of its own whose other 7 slots carry zero weights and re-read the same plane.
// Epilogue (bias / BN affine / activation / fused 2x upsampling store) is conv_dev.h's, shared with the fp32 kernel.
#include "conv16.h"
#include "wprep_dev.h"

#include "conv_dev.h"

#include <algorithm>
#include <cstdlib>

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <typename T>
struct Op16;
template <>
struct Op16<__bf16> {
    using v8 = bf16x8;
    static __device__ __forceinline__ f32x4 mma(v8 a, v8 b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
    }
};
template <>
struct Op16<_Float16> {
    using v8 = f16x8;
    static __device__ __forceinline__ f32x4 mma(v8 a, v8 b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    }
};

struct Conv16Args {
    ConvArgs c;          // geometry / staging / epilogue block shared with the fp32 kernel (wT unused here)
    const void *w16;     // [nblk][nchunks][G4][NT16][8] 16-bit
    int G4;              // group slots per chunk (a multiple of 4, zero-weight padded)
    int ngFull, ngLast;  // groups actually present in a full chunk / in the last chunk
    int c8Full, c8Last;  // channel groups per tap: ceil(cke / 8)
    int ckeLast;         // channels in the last chunk
    int nchunks;
    int diag;            // debug bits (SPRK_C16_DIAG): 1 no input DMA, 2 no weight DMA, 4 no MFMA loop, 8 no store
};

// ---- weights: fp32 [Cout][Cin][KHW] -> 16-bit slabs: a prepared-weight item (wprep_dev.h: wprep_16) ----------------
static int transform16(const sprk::Conv16Call &c, const float *w, void *ws, long total, int wCout, int wCin, int KHW,
                       int CK, int G4, int NT16, int nblk, int nchunks, hipStream_t s) {
    const int kind = (c.dtype & SPRK_DT_MASK) == SPRK_DT_BF16 ? sprk::WPREP_BF16 : sprk::WPREP_F16;
    return sprk::wprep_site(sprk::wprep_item(kind, w, ws, total, {wCout, wCin, KHW, c.mode, CK, G4, NT16, nblk, nchunks}), s);
}

// ---- one staged chunk: nks k-steps of 32 ---------------------------------------------------------------------
//   abase[r]: LDS byte address of this lane's pixel (row base r) at plane 0, tap 0 of the stage
//   gaddr:    LDS byte address of gtab[lq] = {byte offset of the group's first plane + tap, plane stride}
//   baddr:    LDS byte address of w16[group lq][n = l15]
template <typename T, int MT, int NT, int RB>
__device__ __forceinline__ void chunk_mma16(f32x4 (&acc)[MT][NT], const int (&abase)[RB], int gaddr, int baddr,
                                            int bstep, int nks) {
    using V8 = typename Op16<T>::v8;
    constexpr int MPR = MT / RB;
    typedef const __attribute__((address_space(3))) V8 *lds_v8p;
    for (int ks = 0; ks < nks; ++ks) {
        const int goff = lds_i(gaddr)[ks * 8], gstr = lds_i(gaddr)[ks * 8 + 1];
        V8 av[MT];
#pragma unroll
        for (int r = 0; r < RB; ++r) {
            int ad = abase[r] + goff;
            float f[MPR][8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                lds_cfp ap = lds_f(ad);
#pragma unroll
                for (int t = 0; t < MPR; ++t) f[t][j] = ap[16 * t];
                ad += gstr;
            }
#pragma unroll
            for (int t = 0; t < MPR; ++t)
#pragma unroll
                for (int j = 0; j < 8; ++j) av[r * MPR + t][j] = (T)f[t][j];
        }
        V8 bv[NT];
        const int bb = baddr + ks * bstep;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) bv[nt] = *(lds_v8p)(__SIZE_TYPE__)(unsigned)(bb + nt * 256);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = Op16<T>::mma(av[mt], bv[nt], acc[mt][nt]);
    }
}

template <typename T, int MT, int NT, int RB>
__global__ __launch_bounds__(kBlock, 2) void conv16_mfma_kernel(const Conv16Args k) {
    const ConvArgs &a = k.c;
    // LDS: gtab[2][G4] {off, stride} (full chunk | last chunk) | xtab1 | xtab2 | stage 0: input fp32 [CK*cplane],
    //      weights [G4][NT16][8] 16-bit | stage 1: ...
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int NT16 = NT * 16;
    const int wFloats = k.G4 * NT16 * 4;                  // 16 bytes per (group, n) = 4 floats
    const int stageFloats = a.CK * a.cplane + wFloats;
    int *gtab = reinterpret_cast<int *>(smem);
    int *xtab1 = gtab + 4 * k.G4;
    int *xtab2 = xtab1 + a.nG1 * 64;
    float *stage_base = smem + 4 * k.G4 + (a.nG1 + a.nG2) * 64;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, lq = lane >> 4;
    constexpr int TM = 64 * MT;
    const int lgT = a.lgTC + a.lgTR;
    const int NI = TM >> lgT;
    int bid = blockIdx.x;
    if (a.xcdRemap) {   // contiguous runs of tiles per XCD (vertically adjacent tiles share halo rows in its L2)
        const int nwg = gridDim.x, q = nwg >> 3, rr = nwg & 7, xcd = bid & 7;
        bid = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + (bid >> 3);
    }
    const int tx = bid % a.tilesX;
    bid /= a.tilesX;
    const int ty = bid % a.tilesY;
    const int ig = bid / a.tilesY;
    const int nb = blockIdx.y;
    const int oy0 = ty << a.lgTR, ox0 = tx << a.lgTC, n0 = ig * NI;
    const int iy0 = oy0 - a.padT, ix0 = ox0 - a.padL;      // stride 1
    const int Cin = a.C1 + a.C2;
    const int TCm = (1 << a.lgTC) - 1, TRm = (1 << a.lgTR) - 1;
    const int lw = wave;

    int abase[RB];
#pragma unroll
    for (int r = 0; r < RB; ++r) {
        const int p = (wave * MT + r * (MT / RB)) * 16 + l15;
        const int il = p >> lgT, rr = (p >> a.lgTC) & TRm, c = p & TCm;
        abase[r] = ((il * a.inRows + rr) * a.pitch + c + a.colOff) * 4;
    }
    f32x4 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};

    // group table: slot G of a chunk -> {byte offset of (first plane of the channel group, tap), plane stride}.
    // A one-channel group (cke = 8m + 1) re-reads its plane 8 times (stride 0; slots 1..7 carry zero weights), and so
    // do the padding slots behind the last group (all weights zero): every address read is inside the staged image.
    for (int idx = tid; idx < 2 * k.G4; idx += kBlock) {
        const int which = idx >= k.G4, G = idx - which * k.G4;
        const int ng = which ? k.ngLast : k.ngFull, c8 = which ? k.c8Last : k.c8Full;
        const int cke = which ? k.ckeLast : a.CK;
        int off = 0, str = 0;
        if (G < ng) {
            const int tap = G / c8, cg = G - tap * c8;
            const int ky = tap / a.KW, kx = tap - ky * a.KW;
            off = (cg * 8 * a.cplane + ky * a.pitch + kx) * 4;
            str = (cke - cg * 8 >= 8) ? a.cplane * 4 : 0;
        }
        gtab[2 * idx] = off;
        gtab[2 * idx + 1] = str;
    }
    const PlaneGeom pg{NI, a.inRows, a.pitch, a.colOff, a.cplane, a.invImg, a.invPitch, a.deal, 4};
    {
        const int ixa = ix0 - a.colOff;
        if (a.nG1)
            build_xtab(xtab1, a.nG1, a.vec1, 0, a.W1, (long)a.C1 * a.H1 * a.W1, a.N, a.Hin, a.Win, pg, n0, iy0, ixa, tid,
                       kBlock);
        if (a.nG2)
            build_xtab(xtab2, a.nG2, a.vec2, 0, a.Win, (long)a.C2 * a.Hin * a.Win, a.N, a.Hin, a.Win, pg, n0, iy0, ixa,
                       tid, kBlock);
    }
    const float *wslab = reinterpret_cast<const float *>(k.w16) + (long)nb * k.nchunks * wFloats;

    auto issue = [&](int c0, int b) {
        const int cke = min(a.CK, Cin - c0);
        float *in_lds = stage_base + b * stageFloats;
        float *w_lds = in_lds + a.CK * a.cplane;
        const int n1 = max(0, min(c0 + cke, a.C1) - c0);   // channels of this chunk taken from x
        if (n1 > 0 && !(k.diag & 1)) {
            const int cs1 = a.H1 * a.W1;
            stage_planes_buf<4>(in_lds, make_rsrc(a.x + ((long)n0 * a.C1 + c0) * cs1), cs1 * 4, n1, xtab1, a.nG1, a.vec1,
                                a.cplane, lw, lane);
        }
        if (n1 < cke && !(k.diag & 1)) {
            const int cs2 = a.Hin * a.Win;
            const int cb = max(c0, a.C1) - a.C1;
            stage_planes_buf<4>(in_lds + n1 * a.cplane, make_rsrc(a.x2 + ((long)n0 * a.C2 + cb) * cs2), cs2 * 4, cke - n1,
                                xtab2, a.nG2, a.vec2, a.cplane, lw, lane);
        }
        const rsrc_t wr = make_rsrc(wslab + (long)(c0 / a.CK) * wFloats);
        const int total4 = wFloats >> 2;                    // 16-byte pieces of the chunk's weight slab
        const int wv = lane * 16, room = total4 - lane;
        if (!(k.diag & 2))
            for (int gi = lw; gi * 64 < total4; gi += 4)
                if (gi * 64 < room) bdma16(wr, wv, gi * 1024, w_lds + gi * 256);
    };

    __syncthreads();   // tables visible
    issue(0, 0);
    const int baddr0 = lds_addr(stage_base + a.CK * a.cplane) + (lq * NT16 + l15) * 16;
    const int gaddr0 = lds_addr(gtab + 2 * lq);
    const int abyte0 = lds_addr(stage_base);
    constexpr int bstep = 4 * NT16 * 16;
    int ci = 0;
    for (int c0 = 0; c0 < Cin; c0 += a.CK, ++ci) {
        __syncthreads();   // this chunk's DMA has landed for every wave; nobody reads the other stage any more
        if (c0 + a.CK < Cin) issue(c0 + a.CK, (ci + 1) & 1);
        const bool lastc = c0 + a.CK >= Cin;
        const int ng = lastc ? k.ngLast : k.ngFull;
        const int soff = (ci & 1) * stageFloats * 4;
        int ab[RB];
#pragma unroll
        for (int r = 0; r < RB; ++r) ab[r] = abyte0 + soff + abase[r];
        if (!(k.diag & 4))
            chunk_mma16<T, MT, NT, RB>(acc, ab, gaddr0 + (lastc ? k.G4 * 8 : 0), baddr0 + soff, bstep, (ng + 3) >> 2);
    }
    if (k.diag & 8) return;
#include "conv_epilogue.inc"
}

// =====================================================================================================================
// conv16_tile_kernel: the 3x3 layers.  The kernel above keeps conv_mfma_kernel's staging (fp32 tiles by LDS-DMA, 8
// channels per barrier) and is bound by it: with 16x less matrix time per byte the DMA issue, the per-chunk barrier and
// the 8 ds_read_b32 + 4 conversions per operand dominate (measured on 96->96 at 128x64^2: 420 us with the DMA switched
// off, 150 us for the data movement alone, 570 us together — no overlap).  This kernel is built for the 16-bit
// compute/byte ratio instead:
//   * 512-thread workgroups (8 waves, 2 per SIMD), one per CU, 512 output pixels x 16*NT output channels: half the
//     weight re-staging per pixel and a smaller halo share than 256-pixel tiles;
//   * the input tile of the NEXT 16-channel chunk is fetched HBM -> registers (buffer_load_dword, 24 in flight per
//     lane, zeros outside the image by the buffer range check) while the matrix cores work on the current chunk, then
//     converted ONCE (v_cvt_pk) and written to LDS channel-innermost: [8-channel group][halo pixel][8 x 16 bit], so an
//     A operand (pixel, 8 channels at one tap) is ONE ds_read_b128 and every tap re-uses the converted tile;
//   * weights as above (16-bit slabs [group][n][8], one ds_read_b128 per B operand) by LDS-DMA, double buffered;
//   * a k-step's four lane quarters take groups (tap t, channels 0-7), (t, 8-15), (t+1, 0-7), (t+1, 8-15): the two
//     quarters that share an LDS service group differ by a whole channel-group plane (a multiple of 256 B), so the
//     reads are conflict free;  one barrier per 16 channels x 9 taps = 5 k-steps x 24 MFMAs per wave.
// Any channel count works (missing channels of the last chunk are zero registers and zero weights).
// =====================================================================================================================
struct Tile16Args {
    ConvArgs c;          // N, C1, C2, Hin, Win, Cout, Hout, Wout, pads, epilogue, lgTC, lgTR, tilesX, tilesY (store_acc)
    const void *w16;     // [nblk][nchunks][G4][NT16][8]
    int G4, nchunks, ngFull, ngLast, c8Last;
    int PX, PXP, inRows, inCols;   // halo pixels of a tile (all its images), padded to 16; halo rows / cols per image
                                   // (16-bit input: inCols is the even-aligned column grid, see cshift)
    int cshift;                    // 16-bit input: LDS column 0 is image column ix0 - cshift (an even column: pairs of
                                   // pixels are fetched as aligned dwords); fp32 input: 0
    int ntiles;                    // imgGroups * tilesX * tilesY (workgroups are persistent over them)
    int diag;
};

constexpr int kTileThreads = 512;
constexpr int kTileCK = 16;

// X16 / Y16: the input tensors (both sources) / the output tensor are 16-bit tensors of type T instead of fp32.
template <typename T, int NT, bool X16, bool Y16>
__global__ __launch_bounds__(kTileThreads) void conv16_tile_kernel(const Tile16Args k) {
    using V8 = typename Op16<T>::v8;
    typedef const __attribute__((address_space(3))) V8 *lds_v8p;
    typedef __attribute__((address_space(3))) V8 *lds_v8w;
    const ConvArgs &a = k.c;
    constexpr int NT16 = NT * 16, MT = 4;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    int *gtab = reinterpret_cast<int *>(smem);                 // [2][G4] byte offsets (full chunk | last chunk)
    const int inBytes = 2 * k.PXP * 16, wBytes = k.G4 * NT16 * 16, stageBytes = inBytes + wBytes;
    const int stage0 = lds_addr(smem) + 256;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, lq = lane >> 4;
    const int lgT = a.lgTC + a.lgTR;
    const int NI = 512 >> lgT;
    const int nb = blockIdx.y;
    const int Cin = a.C1 + a.C2, HW = a.Hin * a.Win;
    const int TCm = (1 << a.lgTC) - 1, TRm = (1 << a.lgTR) - 1;
    const int ntiles = k.ntiles;

    // ---- this thread's share of a tile fetch: up to R (channel group, halo item) items --------------------------------
    // fp32 input: an item is one halo pixel (8 channel loads of 4 bytes), R = 3.  16-bit input: an item is an aligned PAIR
    // of halo pixels (8 channel loads of 4 bytes = 2 pixels each), R = 2.
    // item i = tid + 512 r: channel group cg = i / P64 (wave-uniform: P64 is a multiple of 64), halo item i % P64.
    // Fixed per thread: the LDS slot and the halo position (image, row, column inside the tile); per tile: the offsets.
    constexpr int R = X16 ? 2 : 3, ES = X16 ? 2 : 4;
    const int nItems = X16 ? k.PX >> 1 : k.PX;
    const int P64 = (nItems + 63) & ~63;
    const int imgPix = k.inRows * k.inCols;
    int wofs[R], hil[R], hrr[R], hcc[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int i = tid + r * kTileThreads;
        const int cg = i / P64, it = i - cg * P64;
        wofs[r] = -1;
        hil[r] = hrr[r] = hcc[r] = 0;
        if (cg < 2 && it < nItems) {
            const int px = X16 ? 2 * it : it;          // (inCols is even for 16-bit input: a pair never straddles rows)
            hil[r] = px / imgPix;
            const int rem = px - hil[r] * imgPix;
            hrr[r] = rem / k.inCols;
            hcc[r] = rem - hrr[r] * k.inCols - k.cshift;
            wofs[r] = (cg * k.PXP + px) * 16;
        }
    }
    int voff1[R], voff2[R];
    int n0 = 0, oy0 = 0, ox0 = 0;     // tile whose input is being fetched
    auto setup = [&](int t) {
        // contiguous runs of tiles per XCD (blocks b and b + 8 share an XCD and its L2; vertically adjacent tiles
        // share halo rows), then tile -> (image group, tile row, tile column)
        int bid = t;
        {
            const int q = ntiles >> 3, rr = ntiles & 7, xcd = bid & 7;
            bid = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + (bid >> 3);
        }
        bid = __builtin_amdgcn_readfirstlane(bid);
        const int tx = bid % a.tilesX;
        bid /= a.tilesX;
        const int ty = bid % a.tilesY;
        const int ig = bid / a.tilesY;
        oy0 = ty << a.lgTR; ox0 = tx << a.lgTC; n0 = ig * NI;
        const int iy0 = oy0 - a.padT, ix0 = ox0 - a.padL;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            // (16-bit input: ix is even and Win is even, so the pair is inside or outside the image as a whole)
            const int n = n0 + hil[r], iy = iy0 + hrr[r], ix = ix0 + hcc[r];
            const bool ok = wofs[r] >= 0 && n < a.N && (unsigned)iy < (unsigned)a.Hin && (unsigned)ix < (unsigned)a.Win;
            const int pix = iy * a.Win + ix;
            voff1[r] = ok ? (hil[r] * a.C1 * HW + pix) * ES : (int)0x80000000;   // beyond the buffer range: the load returns 0
            voff2[r] = ok ? (hil[r] * a.C2 * HW + pix) * ES : (int)0x80000000;
        }
    };
    // Register sets of fetched chunks: DEPTH chunks ahead of the one being multiplied.  DEPTH = 2 (16-bit input only: 16
    // registers per set) was built to keep ~44 KB per CU in flight instead of 22 KB — together with the LDS-only barrier
    // below, which does not drain the prefetch — and MEASURED: 120.3 -> 125.3 us on the 96 -> 96 layer of the bf16 step
    // (gpurun_out/p16_a vs p16_d).  The phase experiment (scratch/r4/c16bench.py, SPRK_C16_DIAG) says why: the input
    // loads are 21 us of 139; the MFMA loop with its LDS operand reads is 62 (38 at the matrix peak), the stores 37, and
    // the phases of a tile add up instead of overlapping (one workgroup per CU).  So: one set.
    constexpr int DEPTH = 1;
    unsigned fA[R][8], fB[DEPTH == 2 ? R : 1][8];
    auto fetch = [&](int c0, unsigned (&f)[R][8]) {      // channels c0 .. c0+15 of the concatenated input -> registers
        // buffer bases at the chunk's first channel of each source: the per-channel scalar offset stays small
        // whatever the plane size (4096^2 planes: 64 MB per channel).
        // EVERY path issues exactly R * 8 (or twice that) loads, unconditionally and without per-load vector selects: the
        // compiler's wait-count pass then knows how many operations are younger than a held register set or the weight
        // DMA, and waits for those with a non-zero count (a conditional load makes it drain the queue: vmcnt(0)).
        // A thread without an item carries the out-of-range offset (the load returns 0); channels past the end of the
        // last chunk re-read the last plane (their weights are zero).
#ifdef SPRK_DIAG
        if (k.diag & 1) return;
#endif
        const int b1 = min(c0, a.C1), b2 = max(c0 - a.C1, 0);
        const char *x1 = reinterpret_cast<const char *>(a.x), *x2 = reinterpret_cast<const char *>(a.x2);
        const rsrc_t r1 = make_rsrc(x1 + ((long)n0 * a.C1 + b1) * HW * ES);
        const rsrc_t r2 = make_rsrc(a.C2 ? x2 + ((long)n0 * a.C2 + b2) * HW * ES : x1);
        if (a.C2 == 0 || c0 + kTileCK <= a.C1) {          // the whole chunk comes from the first source
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int cg = min(__builtin_amdgcn_readfirstlane((tid + r * kTileThreads) / P64), 1);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int ch = min(c0 + cg * 8 + j, a.C1 - 1);     // wave-uniform
                    f[r][j] = __builtin_amdgcn_raw_buffer_load_b32(r1, voff1[r], (ch - b1) * HW * ES, 0);
                }
            }
        } else if (c0 >= a.C1) {                          // ... from the second
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int cg = min(__builtin_amdgcn_readfirstlane((tid + r * kTileThreads) / P64), 1);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int ch = min(c0 + cg * 8 + j, Cin - 1);
                    f[r][j] = __builtin_amdgcn_raw_buffer_load_b32(r2, voff2[r], (ch - a.C1 - b2) * HW * ES, 0);
                }
            }
        } else {                                          // the chunk straddles the sources: one load from each, one of
            const rsrc_t rz = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(x1), 0, 0, 0x00020000);   // them from an
#pragma unroll                                                                                              // empty buffer
            for (int r = 0; r < R; ++r) {
                const int cg = min(__builtin_amdgcn_readfirstlane((tid + r * kTileThreads) / P64), 1);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int ch = min(c0 + cg * 8 + j, Cin - 1);
                    const bool s1 = ch < a.C1;
                    const unsigned va = __builtin_amdgcn_raw_buffer_load_b32(s1 ? r1 : rz, voff1[r], s1 ? (ch - b1) * HW * ES : 0, 0);
                    const unsigned vb = __builtin_amdgcn_raw_buffer_load_b32(s1 ? rz : r2, voff2[r], s1 ? 0 : (ch - a.C1 - b2) * HW * ES, 0);
                    f[r][j] = va | vb;
                }
            }
        }
    };
    auto convert_store = [&](int b, const unsigned (&f)[R][8]) {   // registers -> 16 bit -> LDS stage b, [group][pixel][8]
        const int base = stage0 + b * stageBytes;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            if constexpr (X16) {
                // dword j = channel j of two neighbouring pixels (low half: the even column).  Transpose to
                // pixel-major: v_perm_b32 picks the low (high) halves of a channel pair
                typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
                typedef __attribute__((address_space(3))) u32x4 *lds_u4w;
                u32x4 lo, hi;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    lo[j] = __builtin_amdgcn_perm(f[r][2 * j + 1], f[r][2 * j], 0x05040100u);
                    hi[j] = __builtin_amdgcn_perm(f[r][2 * j + 1], f[r][2 * j], 0x07060302u);
                }
                if (wofs[r] >= 0) {
                    *(lds_u4w)(__SIZE_TYPE__)(unsigned)(base + wofs[r]) = lo;
                    *(lds_u4w)(__SIZE_TYPE__)(unsigned)(base + wofs[r] + 16) = hi;
                }
            } else {
                V8 v;
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = (T)__uint_as_float(f[r][j]);
                if (wofs[r] >= 0) *(lds_v8w)(__SIZE_TYPE__)(unsigned)(base + wofs[r]) = v;
            }
        }
    };
    const float *wslab = reinterpret_cast<const float *>(k.w16) + (long)nb * k.nchunks * (wBytes >> 2);
    auto weights = [&](int chunk, int b) {   // the chunk's weight slab -> LDS stage b (LDS-DMA, 1 KB per wave instruction)
        if (k.diag & 2) return;
        const rsrc_t wr = make_rsrc(wslab + (long)chunk * (wBytes >> 2));
        float *w_lds = smem + 64 + ((b * stageBytes + inBytes) >> 2);
        const int total16 = wBytes >> 4;
        const int wv = lane * 16, room = total16 - lane;
        for (int gi = wave; gi * 64 < total16; gi += 8)
            if (gi * 64 < room) bdma16(wr, wv, gi * 1024, w_lds + gi * 256);
    };

    // group table: slot G -> byte offset of (channel group, tap) inside a stage's input image; padding slots point at
    // offset 0 (their weights are zero)
    for (int idx = tid; idx < 2 * k.G4; idx += kTileThreads) {
        const int which = idx >= k.G4, G = idx - which * k.G4;
        const int ng = which ? k.ngLast : k.ngFull, c8 = which ? k.c8Last : 2;
        int off = 0;
        if (G < ng) {
            const int tap = G / c8, cg = G - tap * c8;
            const int ky = tap / 3, kx = tap - ky * 3;
            off = (cg * k.PXP + ky * k.inCols + kx) * 16;
        }
        gtab[idx] = off;
    }
    // this lane's four A-operand bases: output pixel (wave, mt, l15) -> its halo pixel at tap (0,0)
    int abase[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int p = (wave * MT + mt) * 16 + l15;
        const int il = p >> lgT, rr = (p >> a.lgTC) & TRm, c = p & TCm;
        abase[mt] = ((il * k.inRows + rr) * k.inCols + c + k.cshift) * 16;
    }
    const int baddr0 = stage0 + inBytes + (lq * NT16 + l15) * 16;
    constexpr int bstep = 4 * NT16 * 16;
    const long planeO = (long)a.Hout * a.Wout, planeY = a.up2 ? planeO * 4 : planeO;
    const long W2 = 2L * a.Wout;
    // per-channel epilogue constants, loaded ONCE up front: a load inside the store loop would make every channel
    // tile wait (vmcnt counts in order) until all stores issued before it have completed
    float esc[NT], esh[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int co = nb * NT16 + nt * 16 + l15;
        esc[nt] = 1.f;
        esh[nt] = 0.f;
        if (co < a.Cout) {
            if (a.scale) {
                esc[nt] = a.scale[co];
                esh[nt] = a.shift[co];
            } else if (a.bias) {
                esh[nt] = a.bias[co];
            }
        }
    }

    // The workgroup is persistent: tiles blockIdx.x, + gridDim.x, ...  After a tile's last chunk the NEXT tile's first
    // fetch and weight chunk are issued before this tile's stores, so the stores of one tile and the loads of the next
    // share the memory system instead of taking turns.
    int tile = xcd_slot(blockIdx.x, gridDim.x, a.xcdRemap);   // gridDim.x <= ntiles: the slot is a valid tile
    if (tile >= ntiles) return;
    f32x4 acc[MT][NT];
    auto mma_chunk = [&](int ci) {
        if (k.diag & 4) return;
        const bool lastc = ci + 1 >= k.nchunks;
        const int nks = ((lastc ? k.ngLast : k.ngFull) + 3) >> 2;
        const int sb = (ci & 1) * stageBytes;
        const int gaddr = lds_addr(gtab + (lastc ? k.G4 : 0) + lq);
        const int ain = stage0 + sb, bb0 = baddr0 + sb;
        for (int ks = 0; ks < nks; ++ks) {
            const int goff = lds_i(gaddr)[ks * 4] + ain;
            V8 av[MT], bv[NT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) av[mt] = *(lds_v8p)(__SIZE_TYPE__)(unsigned)(abase[mt] + goff);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) bv[nt] = *(lds_v8p)(__SIZE_TYPE__)(unsigned)(bb0 + ks * bstep + nt * 256);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = Op16<T>::mma(av[mt], bv[nt], acc[mt][nt]);
        }
    };
    // one chunk: `held` carries chunk ci + 1 (fetched one or two steps ago) and is written to the other LDS stage at the
    // end; `into` receives the chunk DEPTH steps ahead.  The weight DMA is issued BEFORE the register loads: vmcnt counts
    // in order, so waiting for the DMA (and for `held`) at the end of the step does not wait for the loads issued after it.
    auto step = [&](int ci, unsigned (&held)[R][8], unsigned (&into)[R][8]) {
        // stage ci&1 complete — every wave's ds_writes (lgkmcnt, drained by the barrier) and its share of the weight DMA:
        // issued a step ago in front of R * 8 register loads, so "at most R * 8 operations outstanding" means it has
        // landed while those loads (the chunk DEPTH - 1 steps ahead) stay in flight across the barrier
        if (DEPTH == 2 && ci + DEPTH - 1 < k.nchunks)     // (every wave issues all R * 8 loads of a fetch)
            asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        else
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        lds_only_barrier();     // (the other stage is free again)
        if (ci + 1 < k.nchunks) weights(ci + 1, (ci + 1) & 1);
        if (ci + DEPTH < k.nchunks) fetch((ci + DEPTH) * kTileCK, into);
        mma_chunk(ci);
        if (ci + 1 < k.nchunks) convert_store((ci + 1) & 1, held);   // waits for `held` (vmcnt), converts, writes the stage
    };
    setup(tile);
    fetch(0, fA);
    weights(0, 0);
    if constexpr (DEPTH == 2) {
        if (k.nchunks > 1) fetch(kTileCK, fB);
    }
    convert_store(0, fA);
    for (;;) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
        if constexpr (DEPTH == 2) {
            for (int ci = 0; ci < k.nchunks; ci += 2) {      // even chunks live in fA, odd ones in fB
                step(ci, fB, fA);
                if (ci + 1 < k.nchunks) step(ci + 1, fA, fB);
            }
        } else {
            for (int ci = 0; ci < k.nchunks; ++ci) step(ci, fA, fA);
        }
        // ---- this tile's output coordinates, then the next tile's first loads, then the stores --------------------------
        const int e_n0 = n0, e_oy0 = oy0, e_ox0 = ox0;
        const int next = tile + (int)gridDim.x;
        const bool more_tiles = next < ntiles;
        lds_only_barrier();                              // every wave has finished reading the stages
        if (more_tiles) {
            setup(next);
            fetch(0, fA);
            weights(0, 0);
            if constexpr (DEPTH == 2) {
                if (k.nchunks > 1) fetch(kTileCK, fB);
            }
        }
        // 16-bit output of a tile that lies wholly inside the tensor: through LDS, so that a wave's store instruction
        // writes whole 128-byte rows of one channel (64 lanes x 16 bytes = the tile's 512 pixels of that channel) instead
        // of 16 channels x 32 bytes.  The stores were 37 of the kernel's 139 us on 96 -> 96 at 128 x 64^2, un-overlapped
        // (scratch/r4/c16bench.py).  The transposed tile lives in stage 1 (free until the next tile's first barrier):
        // [channel][512 pixels] 16 bit, rows of 1040 bytes, PN channel tiles per pass.
        bool ldsEp = false;
        if constexpr (Y16) {
            ldsEp = (a.Wout & 7) == 0 && e_n0 + NI <= a.N && e_oy0 + (TRm + 1) <= a.Hout && e_ox0 + (TCm + 1) <= a.Wout &&
                    !(k.diag & 8);
        }
        if constexpr (Y16) {
            if (ldsEp) {
                constexpr int PN = NT == 6 ? 3 : 1, ROWB = 1040, PASSES = NT / PN, UNITS = PN * 16 * 64 / kTileThreads;
                typedef unsigned u32x2t __attribute__((ext_vector_type(2)));
                typedef unsigned u32x4t __attribute__((ext_vector_type(4)));
                typedef __attribute__((address_space(3))) u32x2t *lds_u2w;
                typedef const __attribute__((address_space(3))) u32x4t *lds_u4r;
                typedef T t4v __attribute__((ext_vector_type(4)));
                const int tbase = stage0 + stageBytes;
                // reader geometry: unit i of this thread = (channel wave + 8 i of the pass, pixels 8 lane .. 8 lane + 7)
                const int rp = lane * 8;
                const int ril = rp >> lgT, rr = (rp >> a.lgTC) & TRm, rc = rp & TCm;
                const long rpix = a.up2 ? (long)(2 * (e_oy0 + rr)) * W2 + 2 * (e_ox0 + rc) : (long)(e_oy0 + rr) * a.Wout + e_ox0 + rc;
                T *const ybase = reinterpret_cast<T *>(a.y) + (long)(e_n0 + ril) * a.Cout * planeY + rpix;
#pragma unroll
                for (int ps = 0; ps < PASSES; ++ps) {
#pragma unroll
                    for (int pn = 0; pn < PN; ++pn) {
                        const int nt = ps * PN + pn;
                        const float sc = esc[nt], sh = esh[nt];
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt) {
                            const f32x4 c = acc[mt][nt];
                            t4v o;
#pragma unroll
                            for (int j = 0; j < 4; ++j) o[j] = (T)apply_act(c[j] * sc + sh, a.act);
                            *(lds_u2w)(__SIZE_TYPE__)(unsigned)(tbase + (pn * 16 + l15) * ROWB + ((wave * MT + mt) * 16 + lq * 4) * 2) =
                                __builtin_bit_cast(u32x2t, o);
                        }
                    }
                    lds_only_barrier();
#pragma unroll
                    for (int i = 0; i < UNITS; ++i) {
                        const int chl = wave + 8 * i;                       // wave-uniform
                        const int co = nb * NT16 + ps * PN * 16 + chl;
                        if (co >= a.Cout) continue;
                        const u32x4t v = *(lds_u4r)(__SIZE_TYPE__)(unsigned)(tbase + chl * ROWB + lane * 16);
                        T *q = ybase + co * planeY;
                        if (a.up2) {
                            u32x4t lo, hi;
#pragma unroll
                            for (int j = 0; j < 2; ++j) {
                                lo[2 * j] = __builtin_amdgcn_perm(v[j], v[j], 0x01000100u);
                                lo[2 * j + 1] = __builtin_amdgcn_perm(v[j], v[j], 0x03020302u);
                                hi[2 * j] = __builtin_amdgcn_perm(v[2 + j], v[2 + j], 0x01000100u);
                                hi[2 * j + 1] = __builtin_amdgcn_perm(v[2 + j], v[2 + j], 0x03020302u);
                            }
                            *reinterpret_cast<u32x4t *>(q) = lo;
                            *reinterpret_cast<u32x4t *>(q + 8) = hi;
                            *reinterpret_cast<u32x4t *>(q + W2) = lo;
                            *reinterpret_cast<u32x4t *>(q + W2 + 8) = hi;
                        } else {
                            *reinterpret_cast<u32x4t *>(q) = v;
                        }
                    }
                    if (ps + 1 < PASSES) lds_only_barrier();                // the next pass overwrites the transposed tile
                }
            }
        }
        if (!(k.diag & 8) && !ldsEp) {
            // D layout: col(n) = lane & 15 -> output channel, row(m) = (lane >> 4) * 4 + reg -> 4 consecutive pixels
            long poff[MT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const int p = (wave * MT + mt) * 16 + lq * 4;
                const int il = p >> lgT, r = (p >> a.lgTC) & TRm, cc = p & TCm;
                const int n = e_n0 + il, oy = e_oy0 + r, ox = e_ox0 + cc;
                const bool ok = n < a.N && oy < a.Hout && ox < a.Wout;
                const long pix = a.up2 ? (long)(2 * oy) * W2 + 2 * ox : (long)oy * a.Wout + ox;
                poff[mt] = ok ? (long)n * a.Cout * planeY + pix : -1;
            }
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int co = nb * NT16 + nt * 16 + l15;
                if (co >= a.Cout) continue;
                const float sc = esc[nt], sh = esh[nt];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    if (poff[mt] < 0) continue;
                    const f32x4 c = acc[mt][nt];
                    float v[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = apply_act(c[j] * sc + sh, a.act);
                    if constexpr (Y16) {
                        // the one rounding of a 16-bit activation tensor: here, at the store
                        typedef T t4 __attribute__((ext_vector_type(4)));
                        typedef T t8 __attribute__((ext_vector_type(8)));
                        T *q = reinterpret_cast<T *>(a.y) + co * planeY + poff[mt];
                        if (a.up2) {
                            t8 row;
#pragma unroll
                            for (int j = 0; j < 4; ++j) row[2 * j] = row[2 * j + 1] = (T)v[j];
                            *reinterpret_cast<t8 *>(q) = row;
                            *reinterpret_cast<t8 *>(q + W2) = row;
                        } else {
                            t4 o;
#pragma unroll
                            for (int j = 0; j < 4; ++j) o[j] = (T)v[j];
                            *reinterpret_cast<t4 *>(q) = o;
                        }
                    } else {
                        float *q = a.y + co * planeY + poff[mt];
                        if (a.up2) {
                            const float4 lo = make_float4(v[0], v[0], v[1], v[1]), hi = make_float4(v[2], v[2], v[3], v[3]);
                            *reinterpret_cast<float4 *>(q) = lo;
                            *reinterpret_cast<float4 *>(q + 4) = hi;
                            *reinterpret_cast<float4 *>(q + W2) = lo;
                            *reinterpret_cast<float4 *>(q + W2 + 4) = hi;
                        } else {
                            *reinterpret_cast<float4 *>(q) = make_float4(v[0], v[1], v[2], v[3]);
                        }
                    }
                }
            }
        }
        if (!more_tiles) break;
        convert_store(0, fA);
        tile = next;
    }
}

// =====================================================================================================================
// conv16_head_kernel: the 1x1 layers (the per-pixel heads: 384 -> 384 -> 96 of the blind-spot U-Net, 96 -> 96 -> 96 of
// the sigma net, and their backward-data 384 <- 384 <- 96, 96 <- 96).
// conv16_mfma_kernel gives every 96-channel output block its own workgroup, so a 384-channel input tile is staged four
// times (and in fp32).  Here ONE workgroup computes ALL output channels of a run of consecutive pixels: 8 waves =
// (8 / CQ) pixel groups of 64 x CQ output-channel quarters of 96 (CQ = 4: up to 384 outputs, 128 pixels per tile;
// CQ = 1: up to 96 outputs, 512 pixels per tile), 64 pixels x 96 channels = 96 accumulator registers per wave.  The
// input is fetched once HBM -> registers (8 channels of one pixel — or, from a 16-bit tensor, of two pixels — per lane,
// consecutive lanes on consecutive pixels), converted if it is fp32 and written channel-innermost
// [8-channel group][pixel][8 x 16 bit]; the weight chunk (CK input channels x all outputs) comes from L2 by LDS-DMA.
// CK / 32 k-steps per barrier, double buffered.
// =====================================================================================================================
struct Head16Args {
    const void *x;       // [N][Cin][HW], fp32 or T
    const void *w16;     // [chunks][CK / 8 groups][96 CQ][8]  (wprep_16 with G4 = CK / 8, NT16 = 96 CQ)
    const float *bias, *scale, *shift;
    void *y;             // [N][Cout][HW], fp32 or T
    int N, Cin, Cout, HW, act, nchunks, tilesPerImage;
};

template <typename T, int CQ, int CK, bool X16, bool Y16>
__global__ __launch_bounds__(512) void conv16_head_kernel(const Head16Args a) {
    using V8 = typename Op16<T>::v8;
    typedef const __attribute__((address_space(3))) V8 *lds_v8p;
    typedef __attribute__((address_space(3))) V8 *lds_v8w;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int GC = CK / 8, PXT = 64 * (8 / CQ), COUTP = 96 * CQ, NKS = CK / 32;
    constexpr int inBytes = GC * PXT * 16, wBytes = GC * COUTP * 16, stageBytes = inBytes + wBytes;
    constexpr int ES = X16 ? 2 : 4;
    constexpr int IPG = X16 ? PXT / 2 : PXT;                    // fetch items per channel group (pixel pairs / pixels)
    constexpr int ITEMS = GC * IPG, R = (ITEMS + 511) / 512;
    const int lds0 = lds_addr(smem);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, lq = lane >> 4;
    const int pg = wave / CQ, cq = wave - pg * CQ;              // pixel group of 64, output-channel quarter
    const int tile = blockIdx.x;
    const int n = tile / a.tilesPerImage, p0 = (tile - n * a.tilesPerImage) * PXT;
    // fetch item i = tid + 512 r -> (channel group i / IPG (wave-uniform: IPG is a multiple of 64), pixel or pixel pair)
    int fgrp[R], fpx[R], voff[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int i = tid + 512 * r;
        fgrp[r] = __builtin_amdgcn_readfirstlane(i / IPG);
        const int ip = i - fgrp[r] * IPG;
        fpx[r] = X16 ? 2 * ip : ip;
        const bool ok = i < ITEMS && p0 + fpx[r] < a.HW;         // HW % 4 == 0: a pair is inside or outside as a whole
        voff[r] = ok ? (p0 + fpx[r]) * ES : (int)0x80000000;
        if (i >= ITEMS) fgrp[r] = -1;
    }
    // register sets of fetched chunks: two for 16-bit input (the chunk after next is in flight too: 168 -> 162 us on
    // 384 -> 384 at 32 x 64^2 in the bf16 step), see conv16_tile_kernel
    constexpr int DEPTH = X16 ? 2 : 1;
    unsigned fA[R][8], fB[DEPTH == 2 ? R : 1][8];
    auto fetch = [&](int chunk, unsigned (&f)[R][8]) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            // buffer based at the 8-channel group (wave-uniform): the scalar channel offset stays below 2^31 on 4096^2 planes
            const int ch0 = chunk * CK + max(fgrp[r], 0) * 8;
            const rsrc_t rx = make_rsrc(reinterpret_cast<const char *>(a.x) + ((long)n * a.Cin + ch0) * a.HW * ES);
            // unconditional, straight-line loads (see conv16_tile_kernel's fetch): a thread without an item carries the
            // out-of-range offset; Cin is a multiple of CK, so every channel exists
#pragma unroll
            for (int j = 0; j < 8; ++j) f[r][j] = __builtin_amdgcn_raw_buffer_load_b32(rx, voff[r], j * a.HW * ES, 0);
        }
    };
    auto convert_store = [&](int b, const unsigned (&f)[R][8]) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            if (fgrp[r] < 0) continue;
            const int dst = lds0 + b * stageBytes + (fgrp[r] * PXT + fpx[r]) * 16;
            if constexpr (X16) {
                typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
                typedef __attribute__((address_space(3))) u32x4 *lds_u4w;
                u32x4 lo, hi;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    lo[j] = __builtin_amdgcn_perm(f[r][2 * j + 1], f[r][2 * j], 0x05040100u);
                    hi[j] = __builtin_amdgcn_perm(f[r][2 * j + 1], f[r][2 * j], 0x07060302u);
                }
                *(lds_u4w)(__SIZE_TYPE__)(unsigned)dst = lo;
                *(lds_u4w)(__SIZE_TYPE__)(unsigned)(dst + 16) = hi;
            } else {
                V8 v;
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = (T)__uint_as_float(f[r][j]);
                *(lds_v8w)(__SIZE_TYPE__)(unsigned)dst = v;
            }
        }
    };
    auto weights = [&](int chunk, int b) {
        const rsrc_t wr = make_rsrc(reinterpret_cast<const float *>(a.w16) + (long)chunk * (wBytes >> 2));
        float *w_lds = smem + ((b * stageBytes + inBytes) >> 2);
        constexpr int total16 = wBytes >> 4;                    // pieces of 16 bytes: a multiple of 64
        for (int gi = wave; gi * 64 < total16; gi += 8) bdma16(wr, lane * 16, gi * 1024, w_lds + gi * 256);
    };
    f32x4 acc[4][6];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nt = 0; nt < 6; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    // A: [group][pixel][8]: lane (pixel l15 of tile mt, quarter lq -> group 4 ks + lq);  B: [group][cout][8]
    const int abase = lds0 + (lq * PXT + pg * 64 + l15) * 16;
    const int bbase = lds0 + inBytes + (lq * COUTP + cq * 96 + l15) * 16;
    // one chunk (conv16_tile_kernel's step): weight DMA of the next chunk first, then the register loads DEPTH chunks
    // ahead; the barrier drains LDS traffic and waits for the DMA only (vmcnt counts in order: R * 8 loads behind it)
    auto step = [&](int c, unsigned (&held)[R][8], unsigned (&into)[R][8]) {
        if (DEPTH == 2 && c + DEPTH - 1 < a.nchunks)     // (every wave issues all R * 8 loads of a fetch)
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(R * 8) : "memory");
        else
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        lds_only_barrier();
        if (c + 1 < a.nchunks) weights(c + 1, (c + 1) & 1);
        if (c + DEPTH < a.nchunks) fetch(c + DEPTH, into);
        const int sb = (c & 1) * stageBytes;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            V8 av[4], bv[6];
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
                av[mt] = *(lds_v8p)(__SIZE_TYPE__)(unsigned)(abase + sb + ks * 4 * PXT * 16 + mt * 256);
#pragma unroll
            for (int nt = 0; nt < 6; ++nt)
                bv[nt] = *(lds_v8p)(__SIZE_TYPE__)(unsigned)(bbase + sb + ks * 4 * COUTP * 16 + nt * 256);
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int nt = 0; nt < 6; ++nt) acc[mt][nt] = Op16<T>::mma(av[mt], bv[nt], acc[mt][nt]);
        }
        if (c + 1 < a.nchunks) convert_store((c + 1) & 1, held);
    };
    fetch(0, fA);
    weights(0, 0);
    if constexpr (DEPTH == 2) {
        if (a.nchunks > 1) fetch(1, fB);
    }
    convert_store(0, fA);
    if constexpr (DEPTH == 2) {
        for (int c = 0; c < a.nchunks; c += 2) {
            step(c, fB, fA);
            if (c + 1 < a.nchunks) step(c + 1, fA, fB);
        }
    } else {
        for (int c = 0; c < a.nchunks; ++c) step(c, fA, fA);
    }
    // epilogue: D layout col(n) = lane & 15 -> output channel, row(m) = (lane >> 4) * 4 + reg -> 4 consecutive pixels
    // (per-channel constants first, all of them: a load between stores would wait for every store before it)
    float esc[6], esh[6];
#pragma unroll
    for (int nt = 0; nt < 6; ++nt) {
        const int co = cq * 96 + nt * 16 + l15;
        esc[nt] = 1.f;
        esh[nt] = 0.f;
        if (co < a.Cout) {
            if (a.scale) {
                esc[nt] = a.scale[co];
                esh[nt] = a.shift[co];
            } else if (a.bias) {
                esh[nt] = a.bias[co];
            }
        }
    }
#pragma unroll
    for (int nt = 0; nt < 6; ++nt) {
        const int co = cq * 96 + nt * 16 + l15;
        if (co >= a.Cout) continue;
        const float sc = esc[nt], sh = esh[nt];
        const long e0 = ((long)n * a.Cout + co) * a.HW + p0 + pg * 64 + lq * 4;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            if (p0 + pg * 64 + mt * 16 + lq * 4 >= a.HW) continue;      // HW % 4 == 0: a quad is all in or all out
            const f32x4 c4 = acc[mt][nt];
            float v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = apply_act(c4[j] * sc + sh, a.act);
            if constexpr (Y16) {
                typedef T t4 __attribute__((ext_vector_type(4)));
                t4 o;
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = (T)v[j];
                *reinterpret_cast<t4 *>(reinterpret_cast<T *>(a.y) + e0 + mt * 16) = o;
            } else {
                *reinterpret_cast<float4 *>(reinterpret_cast<float *>(a.y) + e0 + mt * 16) = make_float4(v[0], v[1], v[2], v[3]);
            }
        }
    }
}

// ---- host side ---------------------------------------------------------------------------------------------
constexpr size_t kLdsBudget16 = 78 * 1024;   // two workgroups per CU

struct Plan16 {
    int MT, NT, lgTC, lgTR, tilesX, tilesY, imgGroups, nblkN, NI;
    int CK, G4, nchunks;
    int inRows, inCols, pitch, cplane, colOff;
    int nG1, nG2;
    size_t ldsBytes, wsBytes;
};

int pad_to_residue16(int raw, int residue) { return raw + ((residue - raw % 32) + 32) % 32; }

bool plan16(int Nimg, int Ck, int Nn, int Ho, int Wo, int KH, int KW, int padL, int hasC2, int WinSrc, Plan16 *p) {
    using sprk::cdiv;
    const int KHW = KH * KW;
    if (!((KH == 3 && KW == 3) || (KH == 1 && KW == 1))) return false;
    if (Ck % 8 > 1 || Nn < 33 || WinSrc % 4 != 0) return false;
    if ((long)Ho * Wo < 256 || Wo % 16 != 0) return false;
    const int ntile = cdiv(Nn, 16);
    int NT = 6, best = 1 << 30;
    for (int cand : {6, 4, 3}) {                      // instantiated channel-tile counts
        const int cost = cdiv(ntile, cand) * (cand + 2);
        if (cost < best) best = cost, NT = cand;
    }
    for (int MT : {4, 2}) {
        const int TM = 64 * MT;
        const int TC = std::min(std::min(sprk::pow2_ceil(Wo), 64), TM);
        const int TR = std::min(TM / TC, sprk::pow2_ceil(Ho));
        const int NI = TM / (TC * TR);
        p->MT = MT; p->NT = NT; p->NI = NI;
        p->lgTC = sprk::ilog2(TC); p->lgTR = sprk::ilog2(TR);
        p->tilesX = cdiv(Wo, TC); p->tilesY = cdiv(Ho, TR);
        p->imgGroups = cdiv(Nimg, NI);
        p->nblkN = cdiv(ntile, NT);
        p->inRows = TR + KH - 1;
        p->inCols = TC + KW - 1;
        p->colOff = (((-padL) % 4) + 4) % 4;
        p->pitch = sprk::roundup(p->colOff + p->inCols, 4);
        p->cplane = pad_to_residue16(NI * p->inRows * p->pitch, 16);
        const long blocks = (long)p->imgGroups * p->tilesX * p->tilesY * p->nblkN;
        if (MT == 4 && blocks < 512) continue;        // too few workgroups for 256 CUs x 2: smaller tiles
        const int planeElems = NI * p->inRows * p->pitch;
        p->nG1 = cdiv(planeElems, 256);
        p->nG2 = hasC2 ? p->nG1 : 0;
        if (p->nG1 > kMaxXG) continue;
        // channels per chunk: a multiple of 8 (or all of them when fewer); 1x1 layers take 32 (one full k-step)
        for (int CK : (KHW == 1 ? std::initializer_list<int>{32, 16, 8} : std::initializer_list<int>{16, 8})) {
            const int ck = std::min(CK, sprk::roundup(Ck, 8));
            const int G4 = sprk::roundup(KHW * cdiv(ck, 8), 4);
            const size_t lds = (size_t)(4 * G4 + (p->nG1 + p->nG2) * 64) * 4 +
                               2 * ((size_t)ck * p->cplane * 4 + (size_t)G4 * NT * 16 * 16);
            if (lds > kLdsBudget16) continue;
            p->CK = ck; p->G4 = G4; p->nchunks = cdiv(Ck, ck);
            p->ldsBytes = lds;
            p->wsBytes = (size_t)p->nblkN * p->nchunks * G4 * NT * 16 * 16;
            return true;
        }
    }
    return false;
}

template <typename T, int MT, int RB>
int launch16_nt(const Conv16Args &k, const Plan16 &p, dim3 grid, hipStream_t s) {
    auto go = [&](auto kernel) {
        if (p.ldsBytes > 64 * 1024 &&
            hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)p.ldsBytes) != hipSuccess) {
            sprk::set_error("conv16: cannot reserve %zu bytes of LDS", p.ldsBytes);
            return (int)SPRK_ELAUNCH;
        }
        hipLaunchKernelGGL(kernel, grid, dim3(kBlock), p.ldsBytes, s, k);
        return (int)SPRK_OK;
    };
    switch (p.NT) {
        case 3: return go(conv16_mfma_kernel<T, MT, 3, RB>);
        case 4: return go(conv16_mfma_kernel<T, MT, 4, RB>);
        default: return go(conv16_mfma_kernel<T, MT, 6, RB>);
    }
}

template <typename T>
int launch16(const Conv16Args &k, const Plan16 &p, hipStream_t s) {
    dim3 grid(p.imgGroups * p.tilesX * p.tilesY, p.nblkN);
    const int rows = std::max(1, (p.MT * 16) >> p.lgTC);   // tile rows a wave's pixel tiles span
    if (p.MT == 4)
        return rows == 1 ? launch16_nt<T, 4, 1>(k, p, grid, s)
                         : rows == 2 ? launch16_nt<T, 4, 2>(k, p, grid, s) : launch16_nt<T, 4, 4>(k, p, grid, s);
    return rows == 1 ? launch16_nt<T, 2, 1>(k, p, grid, s) : launch16_nt<T, 2, 2>(k, p, grid, s);
}

std::atomic<long> g_conv16_launches{0};

// ---- 3x3 layers: conv16_tile_kernel -----------------------------------------------------------------------------
struct PlanT {
    int NT, lgTC, lgTR, tilesX, tilesY, imgGroups, nblkN, NI;
    int G4, nchunks, PX, PXP, inRows, inCols, cshift;
    size_t ldsBytes, wsBytes;
};

bool plan_tile(int Nimg, int Ck, int Nn, int Ho, int Wo, int x16, int padL, PlanT *p) {
    using sprk::cdiv;
    if (Nn < 33 || (long)Ho * Wo < 256 || Wo % 4 != 0) return false;
    const int ntile = cdiv(Nn, 16);
    p->NT = (ntile <= 3) ? 3 : 6;
    const int TC = std::min(sprk::pow2_ceil(Wo), 64);
    const int TR = std::min(512 / TC, sprk::pow2_ceil(Ho));
    const int NI = 512 / (TC * TR);
    p->NI = NI;
    p->lgTC = sprk::ilog2(TC); p->lgTR = sprk::ilog2(TR);
    p->tilesX = cdiv(Wo, TC); p->tilesY = cdiv(Ho, TR);
    p->imgGroups = cdiv(Nimg, NI);
    p->nblkN = cdiv(ntile, p->NT);
    if ((long)p->imgGroups * p->tilesX * p->tilesY * p->nblkN < 128) return false;   // too few workgroups: fp32 path
    // 16-bit input: the halo columns ix0 .. ix0 + TC + 1 (ix0 = ox0 - padL) sit on a grid of aligned pixel pairs that
    // starts at the even column ix0 - cshift
    p->cshift = x16 ? (padL & 1) : 0;
    p->inRows = TR + 2; p->inCols = x16 ? sprk::roundup(TC + 2 + p->cshift, 2) : TC + 2;
    p->PX = NI * p->inRows * p->inCols;
    if (p->PX > 768) return false;
    p->PXP = sprk::roundup(p->PX, 16);
    p->G4 = 20;                                   // 9 taps x 2 channel groups, padded to whole k-steps
    p->nchunks = cdiv(Ck, kTileCK);
    p->ldsBytes = 256 + 2 * ((size_t)2 * p->PXP * 16 + (size_t)p->G4 * p->NT * 16 * 16);
    p->wsBytes = (size_t)p->nblkN * p->nchunks * p->G4 * p->NT * 16 * 16;
    return true;
}

template <typename T, bool X16, bool Y16>
int launch_tile(const Tile16Args &k, const PlanT &p, hipStream_t s) {
    // persistent workgroups: one per CU (8 waves at 159-225 VGPRs fill it), shared among the output-channel blocks
    const int ntiles = p.imgGroups * p.tilesX * p.tilesY;
    static const int persist = getenv("SPRK_C16_PERSIST") ? atoi(getenv("SPRK_C16_PERSIST")) : 1;   // debug: 0 = one tile each
    const int slots = std::max(1, sprk::num_cus() / p.nblkN);
    dim3 grid(persist ? std::min(ntiles, slots) : ntiles, p.nblkN);
    auto go = [&](auto kernel) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)p.ldsBytes) != hipSuccess) {
            sprk::set_error("conv16: cannot reserve %zu bytes of LDS", p.ldsBytes);
            return (int)SPRK_ELAUNCH;
        }
        hipLaunchKernelGGL(kernel, grid, dim3(kTileThreads), p.ldsBytes, s, k);
        return (int)SPRK_OK;
    };
    return p.NT == 3 ? go(conv16_tile_kernel<T, 3, X16, Y16>) : go(conv16_tile_kernel<T, 6, X16, Y16>);
}
template <typename T>
int launch_tile_io(const Tile16Args &k, const PlanT &p, int x16, int y16, hipStream_t s) {
    if (x16) return y16 ? launch_tile<T, true, true>(k, p, s) : launch_tile<T, true, false>(k, p, s);
    return y16 ? launch_tile<T, false, true>(k, p, s) : launch_tile<T, false, false>(k, p, s);
}

}  // namespace

namespace sprk {

// head kernel shape of a call: CQ = 4 (97..384 outputs) or 1 (<= 96); CK = 64 where the input channels allow, else 32
static void head_shape(const Conv16Call &c, int *cq, int *ck) {
    *cq = c.Cout > 96 ? 4 : 1;
    *ck = (*cq == 4 && c.C1 % 64 == 0) ? 64 : 32;
}
static size_t head_ws_bytes(const Conv16Call &c) {
    int cq, ck;
    head_shape(c, &cq, &ck);
    return (size_t)(c.C1 / ck) * (ck / 8) * 96 * cq * 16;
}

// which 16-bit kernel (if any) takes this call: 2 = conv16_tile_kernel (3x3), 1 = conv16_mfma_kernel (1x1), 0 = none
static int plan_of(const Conv16Call &c, Plan16 *p, PlanT *pt) {
    const int dt = c.dtype & SPRK_DT_MASK;
    if (dt != SPRK_DT_BF16 && dt != SPRK_DT_F16) return 0;
    static const int on = getenv("SPRK_CONV16") ? atoi(getenv("SPRK_CONV16")) : 1;   // debug: 0 = always fp32
    if (!on) return 0;
    if (c.stride != 1 || c.dil != 1 || c.up1 || c.res) return 0;
    if (c.Hout != c.Hin || c.Wout != c.Win) return 0;                 // same-size layers (U-Net body)
    if (c.padL < 0 || c.padL > 4 || c.padT < 0) return 0;
    if (c.KH == 1 && c.KW == 1 && !c.C2 && !c.up2 && c.padT == 0 && c.padL == 0) {
        // conv16_head_kernel: 33..384 output channels, input channels a multiple of 32, planes of 4k pixels
        static const int head_on = getenv("SPRK_CONV16_HEAD") ? atoi(getenv("SPRK_CONV16_HEAD")) : 1;   // debug
        const long HW = (long)c.Hin * c.Win;
        const int pxt = c.Cout > 96 ? 128 : 512;
        if (head_on && c.Cout >= 33 && c.Cout <= 384 && c.C1 % 32 == 0 && HW % 4 == 0 && 8 * HW * 4 < (1L << 31) &&
            (long)c.N * cdiv(HW, pxt) >= 128)
            return 3;
    }
    if (c.KH == 3 && c.KW == 3) {
        static const int tile_on = getenv("SPRK_CONV16_TILE") ? atoi(getenv("SPRK_CONV16_TILE")) : 1;   // debug
        if (tile_on && plan_tile(c.N, c.C1 + c.C2, c.Cout, c.Hout, c.Wout, c.x16, c.padL, pt)) {
            // 32-bit byte offsets: an image group's span and 16 channel planes
            const long HW = (long)c.Hin * c.Win, cmax = std::max(c.C1, c.C2);
            if (((pt->NI - 1) * cmax + 1) * HW * 4 < (1L << 31) && 16 * HW * 4 < (1L << 31)) return 2;
        }
    }
    if (c.x16 || c.y16) return 0;       // 16-bit activation storage: conv16_tile_kernel / conv16_head_kernel only
    // Where the fp32 Winograd kernel is the faster one (measured on MI355X, scratch/convbench.py): on the wide 3x3
    // layers at 64x64 and up the 256-pixel-tile kernel is bound by moving fp32 tiles through LDS-DMA (96->96 at
    // 128x64^2: 455 us fp32 Winograd, 570 us here).  Without SPRK_DT_FORCE those layers stay fp32.
    if (!(c.dtype & SPRK_DT_FORCE) && c.KH == 3 && c.Cout > 48 && (long)c.Hout * c.Wout >= 4096) return 0;
    // table offsets are 32-bit byte offsets inside one image group; a chunk's channel offset below 2^32
    if (!plan16(c.N, c.C1 + c.C2, c.Cout, c.Hout, c.Wout, c.KH, c.KW, c.padL, c.C2 > 0, c.Win, p)) return 0;
    const long NIm1 = p->NI - 1;
    if ((NIm1 * c.C1 + 1) * c.Hin * c.Win >= (1L << 29) || (long)p->CK * c.Hin * c.Win >= (1L << 29)) return 0;
    if (c.C2 && ((NIm1 * c.C2 + 1) * c.Hin * c.Win >= (1L << 29))) return 0;
    return 1;
}

bool conv16_eligible(const Conv16Call &c) { return conv16_kind(c) != 0; }
int conv16_kind(const Conv16Call &c) {
    Plan16 p;
    PlanT pt;
    return plan_of(c, &p, &pt);
}

size_t conv16_ws_bytes(const Conv16Call &c) {
    Plan16 p;
    PlanT pt;
    const int which = plan_of(c, &p, &pt);
    if (which == 3) return head_ws_bytes(c) + 512;
    return which == 2 ? pt.wsBytes + 512 : which == 1 ? p.wsBytes + 512 : 0;
}

long conv16_launches() { return g_conv16_launches.load(); }

static int run_tile(const Conv16Call &c, const PlanT &p, const void *x, const void *x2, const float *w, void *y,
                    void *ws, size_t ws_bytes, hipStream_t s) {
    if (ws_bytes < p.wsBytes || !ws) {
        set_error("conv16: workspace too small (%zu < %zu)", ws_bytes, p.wsBytes);
        return SPRK_EWORKSPACE;
    }
    if ((((uintptr_t)x | (uintptr_t)x2 | (uintptr_t)y | (uintptr_t)ws) & 15) != 0) {
        set_error("conv16: tensors must be 16-byte aligned");
        return SPRK_EINVAL;
    }
    const int Cin = c.C1 + c.C2;
    const int dt = c.dtype & SPRK_DT_MASK;
    const long total = (long)p.nblkN * p.nchunks * p.G4 * p.NT * 16 * 8;
    const int wCout = c.mode == 0 ? c.Cout : Cin, wCin = c.mode == 0 ? Cin : c.Cout;
    if (int rc = transform16(c, w, ws, total, wCout, wCin, 9, kTileCK, p.G4, p.NT * 16, p.nblkN, p.nchunks, s)) return rc;
    Tile16Args k{};
    ConvArgs &a = k.c;
    a.x = (const float *)x; a.x2 = (const float *)x2; a.bias = c.bias; a.scale = c.scale; a.shift = c.shift; a.res = nullptr;
    a.y = (float *)y;
    a.N = c.N; a.C1 = c.C1; a.C2 = c.C2; a.Hin = c.Hin; a.Win = c.Win; a.H1 = c.Hin; a.W1 = c.Win;
    a.Cout = c.Cout; a.Hout = c.Hout; a.Wout = c.Wout; a.KH = 3; a.KW = 3; a.stride = 1; a.dil = 1;
    a.padT = c.padT; a.padL = c.padL; a.act = c.act;
    a.lgTC = p.lgTC; a.lgTR = p.lgTR; a.tilesX = p.tilesX; a.tilesY = p.tilesY;
    a.vec4 = 1; a.up2 = c.up2; a.xcdRemap = xcd_on();
    k.w16 = ws;
    k.G4 = p.G4; k.nchunks = p.nchunks;
    k.ngFull = 18;
    const int ckeLast = Cin - (p.nchunks - 1) * kTileCK;
    k.c8Last = cdiv(ckeLast, 8);
    k.ngLast = 9 * k.c8Last;
    k.PX = p.PX; k.PXP = p.PXP; k.inRows = p.inRows; k.inCols = p.inCols; k.cshift = p.cshift;
    k.ntiles = p.imgGroups * p.tilesX * p.tilesY;
    static const int diag = sprk::diag_env("SPRK_C16_DIAG");
    k.diag = diag;
    prof_begin(c.kclass, c.flops, s);
    prof_bytes(c.N * ((c.x16 ? 2.0 : 4.0) * (c.C1 + c.C2) * c.Hin * c.Win +
                      (c.y16 ? 2.0 : 4.0) * c.Cout * c.Hout * c.Wout * (c.up2 ? 4 : 1)));
    const int rc = dt == SPRK_DT_BF16 ? launch_tile_io<__bf16>(k, p, c.x16, c.y16, s)
                                      : launch_tile_io<_Float16>(k, p, c.x16, c.y16, s);
    if (rc) return rc;
    prof_end(c.kclass, s);
    g_conv16_launches.fetch_add(1, std::memory_order_relaxed);
    return check_launch("conv16_tile");
}

template <typename T, int CQ, int CK>
static int launch_head(const Head16Args &a, int x16, int y16, hipStream_t s) {
    constexpr size_t lds = 2 * (size_t)((CK / 8) * 64 * (8 / CQ) * 16 + (CK / 8) * 96 * CQ * 16);
    auto go = [&](auto kernel) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds) != hipSuccess) {
            set_error("conv16: cannot reserve %zu bytes of LDS", lds);
            return (int)SPRK_ELAUNCH;
        }
        hipLaunchKernelGGL(kernel, dim3(a.N * a.tilesPerImage), dim3(512), lds, s, a);
        return (int)SPRK_OK;
    };
    if (x16) return y16 ? go(conv16_head_kernel<T, CQ, CK, true, true>) : go(conv16_head_kernel<T, CQ, CK, true, false>);
    return y16 ? go(conv16_head_kernel<T, CQ, CK, false, true>) : go(conv16_head_kernel<T, CQ, CK, false, false>);
}

static int run_head(const Conv16Call &c, const void *x, const float *w, void *y, void *ws, size_t ws_bytes,
                    hipStream_t s) {
    int cq, ck;
    head_shape(c, &cq, &ck);
    const int nchunks = c.C1 / ck, gc = ck / 8, coutp = 96 * cq, pxt = 64 * (8 / cq);
    const size_t need = head_ws_bytes(c);
    if (ws_bytes < need || !ws) {
        set_error("conv16: workspace too small (%zu < %zu)", ws_bytes, need);
        return SPRK_EWORKSPACE;
    }
    if ((((uintptr_t)x | (uintptr_t)y | (uintptr_t)ws) & 15) != 0) {
        set_error("conv16: tensors must be 16-byte aligned");
        return SPRK_EINVAL;
    }
    const int dt = c.dtype & SPRK_DT_MASK;
    const long total = (long)nchunks * gc * coutp * 8;
    const int wCout = c.mode == 0 ? c.Cout : c.C1, wCin = c.mode == 0 ? c.C1 : c.Cout;
    if (int rc = transform16(c, w, ws, total, wCout, wCin, 1, ck, gc, coutp, 1, nchunks, s)) return rc;
    Head16Args a{};
    a.x = x; a.w16 = ws; a.bias = c.bias; a.scale = c.scale; a.shift = c.shift; a.y = y;
    a.N = c.N; a.Cin = c.C1; a.Cout = c.Cout; a.HW = c.Hin * c.Win; a.act = c.act; a.nchunks = nchunks;
    a.tilesPerImage = cdiv(a.HW, pxt);
    prof_begin(c.kclass, c.flops, s);
    prof_bytes(c.N * ((c.x16 ? 2.0 : 4.0) * c.C1 * c.Hin * c.Win + (c.y16 ? 2.0 : 4.0) * c.Cout * c.Hout * c.Wout));
    auto pick = [&](auto tag) {
        using T = decltype(tag);
        if (cq == 4) return ck == 64 ? launch_head<T, 4, 64>(a, c.x16, c.y16, s) : launch_head<T, 4, 32>(a, c.x16, c.y16, s);
        return launch_head<T, 1, 32>(a, c.x16, c.y16, s);
    };
    const int rc = dt == SPRK_DT_BF16 ? pick(__bf16{}) : pick(_Float16{});
    if (rc) return rc;
    prof_end(c.kclass, s);
    g_conv16_launches.fetch_add(1, std::memory_order_relaxed);
    return check_launch("conv16_head");
}

int conv16_run(const Conv16Call &c, const void *xv, const void *x2v, const float *w, void *yv, void *ws, size_t ws_bytes,
               hipStream_t s) {
    Plan16 p;
    PlanT pt;
    const int which = plan_of(c, &p, &pt);
    if (which == 3) return run_head(c, xv, w, yv, ws, ws_bytes, s);
    if (which == 2) return run_tile(c, pt, xv, x2v, w, yv, ws, ws_bytes, s);
    if (which == 0) {
        set_error("conv16: geometry not eligible%s", (c.x16 || c.y16) ? " for 16-bit activation storage" : "");
        return SPRK_EINVAL;
    }
    const float *x = (const float *)xv, *x2 = (const float *)x2v;
    float *y = (float *)yv;
    if (ws_bytes < p.wsBytes || !ws) {
        set_error("conv16: workspace too small (%zu < %zu)", ws_bytes, p.wsBytes);
        return SPRK_EWORKSPACE;
    }
    if ((((uintptr_t)x | (uintptr_t)x2 | (uintptr_t)y | (uintptr_t)ws) & 15) != 0) {
        set_error("conv16: tensors must be 16-byte aligned");
        return SPRK_EINVAL;
    }
    const int Cin = c.C1 + c.C2, KHW = c.KH * c.KW;
    // GEMM view: forward k = input channels (w[cout][cin]); backward-data k = the forward layer's output channels
    // (x = gy), n = its input channels, taps flipped.  c.* describe the GEMM (C1 + C2 = k channels, Cout = n).
    const long total = (long)p.nblkN * p.nchunks * p.G4 * p.NT * 16 * 8;
    const int wCout = c.mode == 0 ? c.Cout : Cin, wCin = c.mode == 0 ? Cin : c.Cout;
    const int dt = c.dtype & SPRK_DT_MASK;
    if (int rc = transform16(c, w, ws, total, wCout, wCin, KHW, p.CK, p.G4, p.NT * 16, p.nblkN, p.nchunks, s)) return rc;

    Conv16Args k{};
    ConvArgs &a = k.c;
    a.x = x; a.x2 = x2; a.wT = nullptr; a.zeros = nullptr; a.bias = c.bias; a.scale = c.scale; a.shift = c.shift;
    a.res = nullptr; a.y = y;
    a.N = c.N; a.C1 = c.C1; a.C2 = c.C2; a.Hin = c.Hin; a.Win = c.Win; a.up1 = 0; a.H1 = c.Hin; a.W1 = c.Win;
    a.Cout = c.Cout; a.Hout = c.Hout; a.Wout = c.Wout; a.KH = c.KH; a.KW = c.KW; a.stride = 1; a.dil = 1;
    a.padT = c.padT; a.padL = c.padL; a.act = c.act;
    a.lgTC = p.lgTC; a.lgTR = p.lgTR; a.tilesX = p.tilesX; a.tilesY = p.tilesY;
    a.CK = p.CK; a.R4 = 0; a.rows = 0;
    a.inRows = p.inRows; a.inCols = p.inCols; a.pitch = p.pitch; a.cplane = p.cplane; a.colOff = p.colOff; a.ldw = 0;
    a.resH = a.resW = a.resOff = 0;
    a.vec1 = 1; a.vec2 = c.C2 ? 1 : 0; a.vec4 = (c.Wout % 4 == 0) ? 1 : 0; a.up2 = c.up2; a.deal = 1; a.xcdRemap = 1;
    a.xtab = 1; a.nG1 = p.nG1; a.nG2 = p.nG2;
    a.invImg = 1.0f / (float)(p.inRows * p.pitch);
    a.invPitch = 1.0f / (float)p.pitch;
    k.w16 = ws;
    k.G4 = p.G4;
    k.nchunks = p.nchunks;
    k.c8Full = cdiv(p.CK, 8);
    k.ngFull = KHW * k.c8Full;
    k.ckeLast = Cin - (p.nchunks - 1) * p.CK;
    k.c8Last = cdiv(k.ckeLast, 8);
    k.ngLast = KHW * k.c8Last;
    static const int diag = sprk::diag_env("SPRK_C16_DIAG");
    k.diag = diag;
    prof_begin(c.kclass, c.flops, s);
    prof_bytes(4.0 * c.N * ((double)(c.C1 + c.C2) * c.Hin * c.Win + (double)c.Cout * c.Hout * c.Wout * (c.up2 ? 4 : 1)));
    const int rc = dt == SPRK_DT_BF16 ? launch16<__bf16>(k, p, s) : launch16<_Float16>(k, p, s);
    if (rc) return rc;
    prof_end(c.kclass, s);
    g_conv16_launches.fetch_add(1, std::memory_order_relaxed);
    return check_launch("conv16_mfma");
}

}  // namespace sprk
