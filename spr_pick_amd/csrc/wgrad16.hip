// 16-bit-operand backward-weight of the 3x3 stride-1 layers (gfx950 / MI355X):
//   dW[cout][cin][ky][kx] = sum over images and pixels of  gy[n][cout][r][c] * x[n][cin][r + ky - padT][c + kx - padL]
// on v_mfma_f32_16x16x32_{bf16,f16}.  As in conv16.hip the activation tensors (x, gy) are fp32 in HBM with the two operands
// of each product rounded on the way in (round to nearest even), or — X16 — 16-bit tensors of the operand type already
// (half the bytes, no conversion); products exact, sums fp32, dW fp32.
//
// GEMM view: D[m = cout][n = cin] per tap, the reduced dimension k is the PIXEL.  A lane's 8 consecutive k are 8
// consecutive pixels of one row of one channel plane, for both operands:
//   A (lane: cout l&15, pixels 8(l>>4)..+7)  = one ds_read_b128 of the gy tile  [cout][128 pixels]            (16 bit)
//   B (lane: cin  l&15, pixels 8(l>>4)..+7)  = one ds_read_b128 of the x tile   [cin][rows + 2][cols + 16]    (16 bit)
//     at (row + ky, col + kx): the kx shift makes this read 2 or 4 bytes off 16-byte alignment, which gfx950's LDS
//     serves (unaligned DS access); the tile is stored once, not once per shift.
// Workgroup = 512 threads (8 waves, one per CU), persistent over REGIONS of 128 output pixels (2 x 64, 4 x 32 or
// 8 x 16) of all images: it owns all (<= 96) output channels x a block of 48 input channels x 9 taps, i.e. 6 x 27
// accumulator tiles, dealt to the waves as 6 cout tiles x {every 8th (cin tile, tap) pair}: 96 accumulator registers,
// per k-step 6 + 4 LDS reads for 24 MFMAs.  The next region is fetched HBM -> registers (32-byte runs per lane, zeros
// outside the image through the buffer range check) under the current region's MFMAs, converted once and written
// to the other LDS stage; one barrier per region.  Each workgroup leaves one partial dW; sprk_reduce_items adds
// them in a fixed order (deterministic).  gy is read once per 48-channel input block, x once.
#include "wgrad16.h"

#include "conv_dev.h"

#include <algorithm>
#include <cstdlib>
#include <type_traits>

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4v __attribute__((ext_vector_type(4)));

template <typename T>
struct OpW;
template <>
struct OpW<__bf16> {
    using v8 = bf16x8;
    static __device__ __forceinline__ f32x4 mma(v8 a, v8 b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
    }
};
template <>
struct OpW<_Float16> {
    using v8 = f16x8;
    static __device__ __forceinline__ f32x4 mma(v8 a, v8 b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    }
};

constexpr int kWgThreads = 512;
constexpr int kCB = 48;          // input channels per block (3 tiles of 16)
constexpr int kRegionPx = 128;   // output pixels per region = 4 k-steps of 32
constexpr int kGyStride = 272;   // bytes per cout row of the gy tile: 128 pixels x 2 B + 16 (bank spread)

struct Wg16Args {
    const void *x, *x2, *gy; // fp32, or (X16) 16-bit tensors of the operand type
    float *partial;          // [parts][Cout][CinTot][9]
    int N, C1, C2, H, W, Cout, padT, padL;
    int regX, regY;          // regions per row of regions / per column (region = RH rows x RW columns = 128 pixels)
    int seg, segLen;         // a strip (image, column block) is cut into `seg` vertical segments of segLen regions
    int nUnits;              // N * regX * seg
    int xcs;                 // bytes per input channel of the rolling x buffer (SLOTS rows x pitch x 2, bank spread)
    int tail;                // 1: the last input channel (Cin = 48 k + 1) rides in block 0 as plane 48 (see below)
    int diag, xcd;
};

// LGRW: log2 of the region width (6: 2 rows x 64 columns, 5: 4 rows x 32 columns, 4: 8 rows x 16 columns).
// The x tile is a ROLLING window of SLOTS = 2 RH + 2 tile rows per channel: consecutive regions of a strip share RH + 2
// - RH = 2 halo rows... precisely: region j reads tile rows j RH .. j RH + RH + 1 and only the RH rows below are new, so
// x is fetched once (the first version re-fetched the halo rows of every region: 2x the bytes at 2-row regions).
// PL1: padL == 1 (every layer of the networks): the B operands of the kx = 0 and kx = 2 taps start one pixel (2 bytes)
// left / right of a 16-byte boundary.  gfx950's LDS serves such a ds_read_b128, but slowly: with every B read forced onto
// the boundary the kernel ran in 148 instead of 202 us (96 -> 96 at 128 x 64^2, scratch/r4/c16bench.py).  So the
// operand is read ALIGNED (the kx = 1 window) plus the one dword next to it, and shifted by a pixel in registers
// (4 v_alignbit).  The (cin tile, tap) pairs are dealt to the waves so that slot q of EVERY wave is a kx = q tap
// (compile-time shift); slot 3 takes the three left-over pairs (waves 0-2) and the tail tile (wave 3).
template <typename T, int MC, int LGRW, bool X16, bool PL1>   // MC: output-channel tiles (6: 49..96 channels, 3: 33..48)
__global__ __launch_bounds__(kWgThreads) void wgrad16_kernel(const Wg16Args a) {
    constexpr int ES = X16 ? 2 : 4;                            // bytes per element of x / gy in HBM
    using V8 = typename OpW<T>::v8;
    typedef const __attribute__((address_space(3))) V8 *lds_v8p;
    constexpr int RW = 1 << LGRW, RH = kRegionPx / RW, SLOTS = 2 * RH + 2;
    constexpr int PITCH = RW + 16, XG4 = PITCH / 4;            // tile columns; column 8 = the region's first output column
    constexpr int GYN = MC * 16 * 32 / kWgThreads;             // gy items (cout, 4-pixel group) per thread: 6 or 3
    constexpr int XROWIT = (kCB + 1) * XG4;                    // x items (cin plane, 4-column group) per tile row
    constexpr int XN = (RH * XROWIT + kWgThreads - 1) / kWgThreads;   // per thread for a region's RH new rows: 4 or 5
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, kq = lane >> 4;
    const int CinTot = a.C1 + a.C2, HW = a.H * a.W;
    const int cb0 = blockIdx.y * kCB;                          // first input channel of this block
    const int cbn = min(kCB, CinTot - a.tail - cb0);           // channels present
    // Cin = 48 k + 1 (a 96-channel skip concat + the 1-channel image): the odd channel does not get a block of its
    // own.  Its plane is staged as plane 48 of block 0 and ONE more accumulator tile per cout tile covers all 9 taps
    // at once: the B operand's n index is the TAP (lane l&15 reads the plane at its own (ky, kx) shift), in the
    // 28th (cin tile, tap) slot, which wave 3 had free.
    const bool tailHere = a.tail && blockIdx.y == 0;
    const bool tailWave = tailHere && wave == 3;
    constexpr int gyBytes = MC * 16 * kGyStride;
    const int ldsGy = lds_addr(smem), ldsX = ldsGy + 2 * gyBytes;

    typedef unsigned u32x4 __attribute__((__vector_size__(16)));
    typedef unsigned u32x2 __attribute__((__vector_size__(8)));
    typedef unsigned short u16x4 __attribute__((ext_vector_type(4)));
    typedef __attribute__((address_space(3))) u16x4 *lds_u16x4w;
    // one fetched item = 4 consecutive pixels: 16 bytes of fp32 or 8 bytes of 16-bit values
    using fetch_t = typename std::conditional<X16, u32x2, u32x4>::type;
    auto load_item = [&](const rsrc_t r, int voff, int soff) -> fetch_t {
        if constexpr (X16)
            return __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0);
        else
            return __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
    };
    auto cvt4 = [&](const fetch_t &v) {   // -> 4 x 16 bit (8 bytes): fp32 values are rounded here, 16-bit ones pass through
        if constexpr (X16) {
            return __builtin_bit_cast(u16x4, v);
        } else {
            typedef T t4 __attribute__((ext_vector_type(4)));
            // (element first into a scalar: __builtin_bit_cast applied to a vector element expression reads element 0)
            const unsigned e0 = v[0], e1 = v[1], e2 = v[2], e3 = v[3];
            t4 r;
            r[0] = (T)__uint_as_float(e0);
            r[1] = (T)__uint_as_float(e1);
            r[2] = (T)__uint_as_float(e2);
            r[3] = (T)__uint_as_float(e3);
            return __builtin_bit_cast(u16x4, r);
        }
    };

    // ---- per-thread item geometry (16-byte items: consecutive lanes on consecutive 16 bytes) -------------------------
    // gy item i of this thread = (cout (tid >> 5) + 16 i, 4-pixel group tid & 31): byte offset inside an image relative
    // to the region origin and LDS offset are those of item 0 plus a constant per i (not held in registers)
    const int gyco = tid >> 5, gyLim = (a.diag & 1) ? 0 : a.Cout;
    int gyc0, gyl0;
    {
        const int p = (tid & 31) * 4;
        const int rr = p >> LGRW, cc = p & (RW - 1);
        gyc0 = ((gyco * a.H + rr) * a.W + cc) * ES;
        gyl0 = gyco * kGyStride + (tid & 31) * 8;
    }
    const int gycStep = 16 * HW * ES;
    // x items: byte offset inside the item's source relative to (tile row 0, column block), or the out-of-range marker |
    // packed: LDS offset (bits 0-19), tile row within the fetch (bits 20-23, 15 = no item), source 2 flag (bit 24)
    int xc[XN], xp[XN];
    auto x_items = [&](int c0) {            // (re)computed per strip: column validity depends on the column block
#pragma unroll
        for (int i = 0; i < XN; ++i) {
            const int it = tid + i * kWgThreads;
            const int rloc = it / XROWIT, rem = it - rloc * XROWIT;
            const int cl = rem / XG4, g4 = rem - cl * XG4;
            const int ch = cl < kCB ? cb0 + cl : CinTot - 1, ix = c0 - 8 + g4 * 4;
            const bool ok = rloc < RH && (cl < cbn || (cl == kCB && tailHere)) && (unsigned)ix < (unsigned)a.W &&
                            !(a.diag & 1);
            const bool s1 = ch < a.C1;
            const int cc = s1 ? ch : ch - a.C1;
            xc[i] = ok ? (cc * a.H * a.W + ix) * ES : (int)0x80000000;
            xp[i] = (cl * a.xcs + g4 * 8) | ((rloc < RH ? rloc : 15) << 20) | (s1 ? 0 : 1 << 24);
        }
    };
    // Register sets of fetched regions, DEPTH regions ahead.  DEPTH = 2 (16-bit tensors: half the registers per set) was
    // built and measured: 249 us either way on 96 -> 96 at 128 x 64^2 (and 8-200 bytes of scratch at 256 VGPRs).  The phase
    // experiment (scratch/r4/c16bench.py) shows the loads are not the bound: 209 us all told, 195 without them, 87 without
    // the MFMA loop — its LDS operand reads (the kx-shifted, 16-byte-misaligned B reads among them) are.  So: one set.
    constexpr int DEPTH = 1;
    fetch_t fgA[GYN], fxA[XN], fgB[DEPTH == 2 ? GYN : 1], fxB[DEPTH == 2 ? XN : 1];
    // fetch `rows` tile rows starting at tile row t0 (image row y0 + t0) of the strip: rows outside the image are zeros
    auto fetch_x = [&](fetch_t (&fx)[XN], const rsrc_t r1, const rsrc_t r2, int y0, int t0, int rows) {
#pragma unroll
        for (int i = 0; i < XN; ++i) {
            const int rloc = (xp[i] >> 20) & 15;
            const int iy = y0 + t0 + rloc;
            const bool ok = rloc < rows && (unsigned)iy < (unsigned)a.H && xc[i] >= 0;
            const int off = ok ? xc[i] + iy * a.W * ES : (int)0x80000000;
            if (a.C2) {
                const bool s2 = (xp[i] >> 24) & 1;
                const fetch_t v1 = load_item(r1, s2 ? (int)0x80000000 : off, 0);
                const fetch_t v2 = load_item(r2, s2 ? off : (int)0x80000000, 0);
                fx[i] = v1 | v2;
            } else {
                fx[i] = load_item(r1, off, 0);
            }
        }
    };
    auto store_x = [&](const fetch_t (&fx)[XN], int t0, int rows) {   // tile row t -> slot t mod SLOTS
#pragma unroll
        for (int i = 0; i < XN; ++i) {
            const int rloc = (xp[i] >> 20) & 15;
            const int slot = (t0 + rloc) % SLOTS;
            if (rloc < rows)
                *(lds_u16x4w)(__SIZE_TYPE__)(unsigned)(ldsX + (xp[i] & 0xFFFFF) + slot * (PITCH * 2)) = cvt4(fx[i]);
        }
    };
    auto fetch_gy = [&](fetch_t (&fg)[GYN], const rsrc_t rg, int org) {
#pragma unroll
        for (int i = 0; i < GYN; ++i)
            fg[i] = load_item(rg, gyco + 16 * i < gyLim ? gyc0 + i * gycStep : (int)0x80000000, org);
    };
    auto store_gy = [&](const fetch_t (&fg)[GYN], int b) {
#pragma unroll
        for (int i = 0; i < GYN; ++i)
            *(lds_u16x4w)(__SIZE_TYPE__)(unsigned)(ldsGy + b * gyBytes + gyl0 + i * 16 * kGyStride) = cvt4(fg[i]);
    };

    // ---- this wave's accumulator tiles: all MC cout tiles x pairs {wave, wave + 8, wave + 16, wave + 24} of the 27
    // (cin tile, tap) pairs of the block; pair q -> cin tile q / 9, tap q % 9
    constexpr int NP = 4;
    f32x4 acc[MC][NP];
#pragma unroll
    for (int m = 0; m < MC; ++m)
#pragma unroll
        for (int q = 0; q < NP; ++q) acc[m][q] = f32x4{0.f, 0.f, 0.f, 0.f};
    // a k-step is 32 consecutive region pixels: one row piece (regions >= 32 wide) or two 16-pixel rows (LGRW = 4),
    // where the lane's pixel group kq sits in row kq >> 1 at column 8 (kq & 1)
    const int kcol = LGRW == 4 ? (kq & 1) * 8 : kq * 8;
    const bool krow1 = LGRW == 4 && (kq >> 1);
    // pair slots of this wave: q = 0..2 -> (cin tile wave / 3, tap row wave % 3, tap column q); q = 3 -> (cin tile 2, tap
    // row 2, tap column wave) for waves 0..2 (the 27th.. pairs), the tail tile for wave 3, nothing for waves 4..7
    int blane[NP], bky[NP];   // lane part of the ALIGNED B address (channel, pixel group; tap column kx = padL) | tap row
    int bshift[NP];           // bytes from there to the tap's own window: (kx - padL) * 2
#pragma unroll
    for (int q = 0; q < NP; ++q) {
        const int it = q < 3 ? wave / 3 : 2, ky = q < 3 ? wave % 3 : 2, kx = q < 3 ? q : min(wave, 2);
        bky[q] = __builtin_amdgcn_readfirstlane(ky);
        blane[q] = ldsX + (it * 16 + l15) * a.xcs + (8 + kcol) * 2;
        // (diag bit 16, diagnostic builds only: every B read 16-byte aligned — what do the kx-shifted reads cost?)
        bshift[q] = __builtin_amdgcn_readfirstlane((a.diag & 16) ? 0 : (kx - a.padL) * 2);
    }
    typedef unsigned u32x4b __attribute__((ext_vector_type(4)));
    typedef const __attribute__((address_space(3))) u32x4b *lds_u4p;
    typedef const __attribute__((address_space(3))) unsigned *lds_u1p;
    // B operand of a tap whose window starts SH pixels from the aligned one at byte address ba
    auto read_b = [&](int ba, auto shc) -> V8 {
        constexpr int SH = decltype(shc)::value;
        const u32x4b d = *(lds_u4p)(__SIZE_TYPE__)(unsigned)ba;
        if constexpr (SH == 0) {
            return __builtin_bit_cast(V8, d);
        } else if constexpr (SH < 0) {
            const unsigned e = *(lds_u1p)(__SIZE_TYPE__)(unsigned)(ba - 4);
            u32x4b r;
            r[0] = __builtin_amdgcn_alignbit(d[0], e, 16);
            r[1] = __builtin_amdgcn_alignbit(d[1], d[0], 16);
            r[2] = __builtin_amdgcn_alignbit(d[2], d[1], 16);
            r[3] = __builtin_amdgcn_alignbit(d[3], d[2], 16);
            return __builtin_bit_cast(V8, r);
        } else {
            const unsigned e = *(lds_u1p)(__SIZE_TYPE__)(unsigned)(ba + 16);
            u32x4b r;
            r[0] = __builtin_amdgcn_alignbit(d[1], d[0], 16);
            r[1] = __builtin_amdgcn_alignbit(d[2], d[1], 16);
            r[2] = __builtin_amdgcn_alignbit(d[3], d[2], 16);
            r[3] = __builtin_amdgcn_alignbit(e, d[3], 16);
            return __builtin_bit_cast(V8, r);
        }
    };
    using ICm1 = std::integral_constant<int, -1>;
    using IC0 = std::integral_constant<int, 0>;
    using ICp1 = std::integral_constant<int, 1>;
    const int alane = ldsGy + l15 * kGyStride + kq * 16;
    // tail tile: lane's tap = min(l & 15, 8) (columns 9..15 of D are not stored)
    const int ttap = min(l15, 8), tky = ttap / 3, tkx = ttap - tky * 3;
    const int tlane = ldsX + kCB * a.xcs + (8 + tkx - a.padL + kcol) * 2;

    for (int u = xcd_slot(blockIdx.x, gridDim.x, a.xcd); u < a.nUnits; u += gridDim.x) {
        const int sgi = u % a.seg;
        const int t = u / a.seg;
        const int cx = t % a.regX, n = t / a.regX;
        const int c0 = cx << LGRW;
        const int ry0 = sgi * a.segLen, nreg = min(a.segLen, a.regY - ry0);
        const int y0 = ry0 * RH - a.padT;                  // image row of tile row 0 of this unit
        x_items(c0);
        const char *xb = (const char *)a.x, *x2b = (const char *)a.x2, *gyb = (const char *)a.gy;
        const rsrc_t rg = make_rsrc(gyb + (long)n * a.Cout * HW * ES);
        const rsrc_t r1 = make_rsrc(xb + (long)n * a.C1 * HW * ES);
        const rsrc_t r2 = make_rsrc(a.C2 ? x2b + (long)n * a.C2 * HW * ES : xb);
        lds_only_barrier();                                // the previous unit's last region has been consumed
        auto gy_org = [&](int j) { return (((ry0 + j) * RH) * a.W + c0) * ES; };
        // prologue: the two halo rows on top, then region 0's RH rows and its gy tile (and region 1 on its way)
        fetch_x(fxA, r1, r2, y0, 0, 2);
        store_x(fxA, 0, 2);
        fetch_x(fxA, r1, r2, y0, 2, RH);
        fetch_gy(fgA, rg, gy_org(0));
        if constexpr (DEPTH == 2) {
            if (nreg > 1) {
                fetch_x(fxB, r1, r2, y0, 2 + RH, RH);
                fetch_gy(fgB, rg, gy_org(1));
            }
        }
        store_x(fxA, 2, RH);
        store_gy(fgA, 0);
        // region j: `hx / hg` hold region j + 1 (written to LDS at the end), `ix / ig` receive region j + DEPTH
        auto region = [&](int j, fetch_t (&hx)[XN], fetch_t (&hg)[GYN], fetch_t (&ix)[XN], fetch_t (&ig)[GYN]) {
            lds_only_barrier();                            // region j's tiles are complete; the older ones are free
                                                           // (LDS only: the register prefetch stays in flight)
            if (j + DEPTH < nreg) {                        // in flight under this region's (and the next one's) MFMAs
                fetch_x(ix, r1, r2, y0, 2 + (j + DEPTH) * RH, RH);
                fetch_gy(ig, rg, gy_org(j + DEPTH));
            }
            if (!(a.diag & 4)) {
                const int ga = alane + (j & 1) * gyBytes;
                const int jrow = (j * RH) % SLOTS;
                // byte offset of tile row (j RH + d) in the rolling buffer, d = 0 .. RH + 1 (one modulo per region)
                int srow[RH + 2];
#pragma unroll
                for (int dd = 0; dd < RH + 2; ++dd) {
                    const int t = jrow + dd;
                    srow[dd] = (t >= SLOTS ? t - SLOTS : t) * (PITCH * 2);
                }
#pragma unroll
                for (int ks = 0; ks < kRegionPx / 32; ++ks) {
                    const int rr = (ks * 32) >> LGRW;                     // region row of this k-step (compile time)
                    const int cc2 = ((ks * 32) & (RW - 1)) * 2;           // byte offset of its first column
                    V8 av[MC], bv[NP];
#pragma unroll
                    for (int m = 0; m < MC; ++m)
                        av[m] = *(lds_v8p)(__SIZE_TYPE__)(unsigned)(ga + m * 16 * kGyStride + ks * 64);
#pragma unroll
                    for (int q = 0; q < NP; ++q) {
                        // (the 2-row regions allocate better with the modulo written out, the 4-row ones with the table)
                        int so = LGRW == 6 ? ((jrow + rr + bky[q]) % SLOTS) * (PITCH * 2)
                                           : (bky[q] == 0 ? srow[rr] : bky[q] == 1 ? srow[rr + 1] : srow[rr + 2]);   // scalar
                        if constexpr (LGRW == 4) {                        // second row of the k-step: per-lane select
                            const int so1 = bky[q] == 0 ? srow[rr + 1] : bky[q] == 1 ? srow[rr + 2] : srow[rr + 3];
                            so = krow1 ? so1 : so;
                        }
                        const int ba = blane[q] + so + cc2;               // the aligned (kx = padL) window
                        if (q == NP - 1 && tailWave) {                    // wave-uniform: per-lane tap row and column
                            int t0 = srow[rr], t1 = srow[rr + 1], t2 = srow[rr + 2];
                            if constexpr (LGRW == 4) {
                                if (krow1) t0 = srow[rr + 1], t1 = srow[rr + 2], t2 = srow[rr + 3];
                            }
                            bv[q] = *(lds_v8p)(__SIZE_TYPE__)(unsigned)(tlane + cc2 + (tky == 0 ? t0 : tky == 1 ? t1 : t2));
                        } else if constexpr (!PL1) {
                            bv[q] = *(lds_v8p)(__SIZE_TYPE__)(unsigned)(ba + bshift[q]);       // any padL: the direct read
                        } else if (q == 0) {
                            bv[q] = read_b(ba, ICm1{});
                        } else if (q == 1) {
                            bv[q] = read_b(ba, IC0{});
                        } else if (q == 2) {
                            bv[q] = read_b(ba, ICp1{});
                        } else {                                          // slot 3: the wave's own tap column
                            bv[q] = bshift[q] < 0 ? read_b(ba, ICm1{}) : bshift[q] == 0 ? read_b(ba, IC0{}) : read_b(ba, ICp1{});
                        }
                    }
#pragma unroll
                    for (int m = 0; m < MC; ++m)
#pragma unroll
                        for (int q = 0; q < NP; ++q) acc[m][q] = OpW<T>::mma(av[m], bv[q], acc[m][q]);
                }
            }
            if (j + 1 < nreg) {
                store_x(hx, 2 + (j + 1) * RH, RH);
                store_gy(hg, (j + 1) & 1);
            }
        };
        if constexpr (DEPTH == 2) {
            for (int j = 0; j < nreg; j += 2) {            // even regions live in set A, odd ones in set B
                region(j, fxB, fgB, fxA, fgA);
                if (j + 1 < nreg) region(j + 1, fxA, fgA, fxB, fgB);
            }
        } else {
            for (int j = 0; j < nreg; ++j) region(j, fxA, fgA, fxA, fgA);
        }
    }

    // ---- partial dW of this workgroup: D layout col(n = cin) = lane & 15, row(m = cout) = (lane >> 4) * 4 + reg -------
    float *part = a.partial + (long)blockIdx.x * a.Cout * CinTot * 9;
#pragma unroll
    for (int q = 0; q < NP; ++q) {
        if (q == 3 && wave == 3 && tailHere) {       // the tail channel's tile: column = tap
            if (l15 < 9) {
#pragma unroll
                for (int m = 0; m < MC; ++m)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int co = m * 16 + kq * 4 + j;
                        if (co < a.Cout) part[((long)co * CinTot + CinTot - 1) * 9 + l15] = acc[m][q][j];
                    }
            }
            continue;
        }
        if (q == 3 && wave >= 3) continue;
        const int it = q < 3 ? wave / 3 : 2, tap = q < 3 ? (wave % 3) * 3 + q : 6 + wave;
        const int ci = cb0 + it * 16 + l15;
        if (it * 16 + l15 >= cbn) continue;
#pragma unroll
        for (int m = 0; m < MC; ++m)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int co = m * 16 + kq * 4 + j;
                if (co < a.Cout) part[((long)co * CinTot + ci) * 9 + tap] = acc[m][q][j];
            }
    }
}


// ---- 1x1 layers (the 384 -> 384 -> 96 head of the blind-spot network) ----------------------------------------------
//   dW[cout][cin] = sum over images and pixels of gy[n][cout][px] * x[n][cin][px]
// No taps and no halo: both operands are [channel][64 pixels] tiles (16-bit, rows of 144 bytes: conflict-free
// ds_read_b128 for the 16 channels x 4 pixel groups of a fragment).  The 8 waves form a WM x WN grid over a
// (WM MC 16) x (WN NC 16) block of dW — 192 x 192 (Cout > 96: x and gy are each read Cout/192 resp. Cin/192 times) or
// 96 x 384 (gy and x read once) — 18 accumulator tiles per wave, 9 LDS reads per 18 MFMAs.  The kernel is HBM-bound
// (two MFMA k-steps per 96 KB of fp32 input per workgroup), so the tile shape is chosen for the fewest re-reads.
// Regions of 64 pixels are dealt to the workgroups in contiguous runs; the next region's 12 x 16 bytes per lane are
// in flight under the current region's MFMAs, converted and written to the other LDS stage.
struct Wg1Args {
    const void *x, *gy;      // fp32, or (X16) 16-bit tensors of the operand type
    float *partial;          // [parts][Cout][Cin]
    int N, Cin, Cout, HW;
    int regPerImg, nRegions, perPart;
    int diag;
};

template <typename T, int WM, int MC, int NC, bool X16>
__global__ __launch_bounds__(kWgThreads) void wgrad16_1x1_kernel(const Wg1Args a) {
    constexpr int ES = X16 ? 2 : 4;
    using V8 = typename OpW<T>::v8;
    typedef const __attribute__((address_space(3))) V8 *lds_v8p;
    constexpr int WN = 8 / WM;
    constexpr int CO = WM * MC * 16, CI = WN * NC * 16;
    constexpr int kPx = 64, kRow = kPx * 2 + 16;               // bytes per channel row of a tile
    constexpr int GN = CO * 16 / kWgThreads, XN = CI * 16 / kWgThreads;   // 16-byte items per thread
    constexpr int gyBytes = CO * kRow, xBytes = CI * kRow;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave - wm * WN;
    const int l15 = lane & 15, kq = lane >> 4;
    const int co0 = blockIdx.y * CO, ci0 = blockIdx.z * CI;
    const int ldsGy = lds_addr(smem), ldsX = ldsGy + 2 * gyBytes;

    typedef unsigned u32x4 __attribute__((__vector_size__(16)));
    typedef unsigned u32x2 __attribute__((__vector_size__(8)));
    typedef unsigned short u16x4 __attribute__((ext_vector_type(4)));
    typedef __attribute__((address_space(3))) u16x4 *lds_u16x4w;
    using fetch_t = typename std::conditional<X16, u32x2, u32x4>::type;
    auto load_item = [&](const rsrc_t r, int voff, int soff) -> fetch_t {
        if constexpr (X16)
            return __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0);
        else
            return __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
    };
    auto cvt4 = [&](const fetch_t &v) {
        if constexpr (X16) {
            return __builtin_bit_cast(u16x4, v);
        } else {
            typedef T t4 __attribute__((ext_vector_type(4)));
            const unsigned e0 = v[0], e1 = v[1], e2 = v[2], e3 = v[3];
            t4 r;
            r[0] = (T)__uint_as_float(e0);
            r[1] = (T)__uint_as_float(e1);
            r[2] = (T)__uint_as_float(e2);
            r[3] = (T)__uint_as_float(e3);
            return __builtin_bit_cast(u16x4, r);
        }
    };
    // item i of this thread: channel row (tid >> 4) + 32 i, pixel group tid & 15
    const int row0 = tid >> 4, g4 = tid & 15;
    const int goff0 = ((co0 + row0) * a.HW + g4 * 4) * ES, xoff0 = ((ci0 + row0) * a.HW + g4 * 4) * ES;
    const int rstep = 32 * a.HW * ES;
    const int lrow = row0 * kRow + g4 * 8;
    const int gyLim = (a.diag & 1) ? 0 : a.Cout - co0, xLim = (a.diag & 1) ? 0 : a.Cin - ci0;
    fetch_t fg[GN], fx[XN];
    auto fetch = [&](const rsrc_t rg, const rsrc_t rx, int org) {
#pragma unroll
        for (int i = 0; i < GN; ++i)
            fg[i] = load_item(rg, row0 + 32 * i < gyLim ? goff0 + i * rstep : (int)0x80000000, org);
#pragma unroll
        for (int i = 0; i < XN; ++i)
            fx[i] = load_item(rx, row0 + 32 * i < xLim ? xoff0 + i * rstep : (int)0x80000000, org);
    };
    auto store = [&](int b) {
#pragma unroll
        for (int i = 0; i < GN; ++i)
            *(lds_u16x4w)(__SIZE_TYPE__)(unsigned)(ldsGy + b * gyBytes + lrow + i * 32 * kRow) = cvt4(fg[i]);
#pragma unroll
        for (int i = 0; i < XN; ++i)
            *(lds_u16x4w)(__SIZE_TYPE__)(unsigned)(ldsX + b * xBytes + lrow + i * 32 * kRow) = cvt4(fx[i]);
    };
    f32x4 acc[MC][NC];
#pragma unroll
    for (int m = 0; m < MC; ++m)
#pragma unroll
        for (int n = 0; n < NC; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int alane = ldsGy + (wm * MC * 16 + l15) * kRow + kq * 16;
    const int blane = ldsX + (wn * NC * 16 + l15) * kRow + kq * 16;

    const int r0 = blockIdx.x * a.perPart, r1 = min(a.nRegions, r0 + a.perPart);
    auto rsrcs = [&](int r, rsrc_t &rg, rsrc_t &rx, int &org) {
        const int n = r / a.regPerImg, pb = r - n * a.regPerImg;
        rg = make_rsrc((const char *)a.gy + (long)n * a.Cout * a.HW * ES);
        rx = make_rsrc((const char *)a.x + (long)n * a.Cin * a.HW * ES);
        org = pb * kPx * ES;
    };
    if (r0 < r1) {
        rsrc_t rg, rx;
        int org;
        rsrcs(r0, rg, rx, org);
        fetch(rg, rx, org);
        store(0);
    }
    for (int r = r0; r < r1; ++r) {
        lds_only_barrier();                                // region r's tiles are complete; the other stage is free
        const int b = (r - r0) & 1;
        const bool more = r + 1 < r1;
        if (more) {
            rsrc_t rg, rx;
            int org;
            rsrcs(r + 1, rg, rx, org);
            fetch(rg, rx, org);
        }
        if (!(a.diag & 4)) {
#pragma unroll
            for (int ks = 0; ks < kPx / 32; ++ks) {
                V8 av[MC], bv[NC];
#pragma unroll
                for (int m = 0; m < MC; ++m)
                    av[m] = *(lds_v8p)(__SIZE_TYPE__)(unsigned)(alane + b * gyBytes + m * 16 * kRow + ks * 64);
#pragma unroll
                for (int n = 0; n < NC; ++n)
                    bv[n] = *(lds_v8p)(__SIZE_TYPE__)(unsigned)(blane + b * xBytes + n * 16 * kRow + ks * 64);
#pragma unroll
                for (int m = 0; m < MC; ++m)
#pragma unroll
                    for (int n = 0; n < NC; ++n) acc[m][n] = OpW<T>::mma(av[m], bv[n], acc[m][n]);
            }
        }
        if (more) store(b ^ 1);
    }
    // partial dW of this workgroup: D layout col(n = cin) = lane & 15, row(m = cout) = (lane >> 4) * 4 + reg
    float *part = a.partial + (long)blockIdx.x * a.Cout * a.Cin;
#pragma unroll
    for (int m = 0; m < MC; ++m)
#pragma unroll
        for (int n = 0; n < NC; ++n) {
            const int ci = ci0 + (wn * NC + n) * 16 + l15;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int co = co0 + (wm * MC + m) * 16 + kq * 4 + j;
                if (co < a.Cout && ci < a.Cin) part[(long)co * a.Cin + ci] = acc[m][n][j];
            }
        }
}

struct Plan1 {
    int wide;                // 1: 192 x 192 blocks, 0: 96 x 384
    int coBlocks, ciBlocks, parts, perPart, regPerImg, nRegions;
    size_t ldsBytes, wsBytes;
};

struct PlanW {
    int MC, lgRW, regX, regY, seg, segLen, nUnits, xcs, parts, nBlocks, tail;
    size_t ldsBytes, wsBytes;
};

int spread_stride(int bytes) {   // smallest stride >= bytes, multiple of 16, whose dword stride mod 64 is 4 * odd
    int s = (bytes + 15) & ~15;
    while (((s >> 2) & 7) != 4) s += 16;
    return s;
}

std::atomic<long> g_wgrad16_launches{0};

}  // namespace

namespace sprk {

static bool plan_wg16(const Wgrad16Call &c, PlanW *p) {
    const int dt = c.dtype & SPRK_DT_MASK;
    if (dt != SPRK_DT_BF16 && dt != SPRK_DT_F16) return false;
    static const int on = getenv("SPRK_WGRAD16") ? atoi(getenv("SPRK_WGRAD16")) : 1;   // debug: 0 = fp32 kernels
    if (!on) return false;
    if (c.KH != 3 || c.KW != 3 || c.stride != 1 || c.dil != 1 || c.up1) return false;
    if (c.Hout != c.H || c.Wout != c.W) return false;
    if (c.padL < 0 || c.padL > 4 || c.padT < 0 || c.padT > 4) return false;
    if (c.Cout < 33 || c.Cout > 96) return false;
    const int W = c.W, H = c.H;
    int RW;
    if (W % 64 == 0) RW = 64;
    else if (W == 32 || W == 16) RW = W;
    else return false;
    const int RH = kRegionPx / RW;
    if (H % RH) return false;
    p->MC = c.Cout > 48 ? 6 : 3;
    p->lgRW = ilog2(RW);
    p->regX = W / RW; p->regY = H / RH;
    const long regions = (long)c.N * p->regX * p->regY;
    if (regions < 512 || regions > (1L << 30)) return false;
    if ((long)std::max(std::max(c.C1, c.C2), c.Cout) * H * W * 4 >= (1L << 31)) return false;   // 32-bit byte offsets per image
    const int CinTot = c.C1 + c.C2;
    p->tail = (CinTot > kCB && CinTot % kCB == 1) ? 1 : 0;
    p->nBlocks = cdiv(CinTot - p->tail, kCB);
    const int wgs = std::max(1, num_cus() / p->nBlocks);   // workgroups per input-channel block: one per CU in all
    // vertical segments per strip: enough units to give every workgroup >= 2, segments of >= 4 regions
    int seg = 1;
    while ((long)c.N * p->regX * seg < 2L * wgs && p->regY % (seg * 2) == 0 && p->regY / (seg * 2) >= 4) seg *= 2;
    p->seg = seg; p->segLen = p->regY / seg;
    p->nUnits = c.N * p->regX * seg;
    p->parts = std::min(wgs, p->nUnits);
    const int SLOTS = 2 * RH + 2, PITCH = RW + 16;
    p->xcs = spread_stride(SLOTS * PITCH * 2);
    p->ldsBytes = 2 * (size_t)p->MC * 16 * kGyStride + (size_t)(kCB + 1) * p->xcs;
    p->wsBytes = (size_t)p->parts * c.Cout * (c.C1 + c.C2) * 9 * sizeof(float);
    return true;
}

static bool plan_wg1x1(const Wgrad16Call &c, Plan1 *p) {
    const int dt = c.dtype & SPRK_DT_MASK;
    if (dt != SPRK_DT_BF16 && dt != SPRK_DT_F16) return false;
    static const int on = getenv("SPRK_WGRAD16_1X1") ? atoi(getenv("SPRK_WGRAD16_1X1")) : 1;   // debug: 0 = fp32 kernel
    if (!on) return false;
    if (c.KH != 1 || c.KW != 1 || c.stride != 1 || c.dil != 1 || c.up1 || c.C2 != 0) return false;
    if (c.padL != 0 || c.padT != 0 || c.Hout != c.H || c.Wout != c.W) return false;
    const int HW = c.H * c.W, Cin = c.C1;
    if (HW % 64 || c.Cout < 33 || Cin < 97) return false;
    if ((long)std::max(Cin, c.Cout) * HW * 4 >= (1L << 31)) return false;       // 32-bit byte offsets inside an image
    p->regPerImg = HW / 64;
    const long regions = (long)c.N * p->regPerImg;
    if (regions < 1024 || regions > (1L << 30)) return false;
    p->nRegions = (int)regions;
    p->wide = c.Cout > 96 ? 1 : 0;
    const int CO = p->wide ? 192 : 96, CI = p->wide ? 192 : 384;
    p->coBlocks = cdiv(c.Cout, CO);
    p->ciBlocks = cdiv(Cin, CI);
    const int blocks = p->coBlocks * p->ciBlocks;
    if (blocks > 64) return false;
    const int parts = std::max(1, num_cus() / blocks);                          // one workgroup per CU in all
    p->perPart = cdiv(regions, parts);
    p->parts = cdiv(regions, p->perPart);
    p->ldsBytes = 2 * (size_t)(CO + CI) * (64 * 2 + 16);
    p->wsBytes = (size_t)p->parts * c.Cout * Cin * sizeof(float);
    return true;
}

bool wgrad16_eligible(const Wgrad16Call &c) {
    PlanW p;
    Plan1 p1;
    return plan_wg16(c, &p) || plan_wg1x1(c, &p1);
}

size_t wgrad16_ws_bytes(const Wgrad16Call &c) {
    PlanW p;
    Plan1 p1;
    return plan_wg16(c, &p) ? p.wsBytes : plan_wg1x1(c, &p1) ? p1.wsBytes : 0;
}

static int wgrad16_run_1x1(const Wgrad16Call &c, const Plan1 &p, const void *x, const void *gy, float *gw, void *ws,
                           size_t ws_bytes, sprk_reduce_item *item, hipStream_t s) {
    if (ws_bytes < p.wsBytes || !ws) {
        set_error("wgrad16 (1x1): workspace too small (%zu < %zu)", ws_bytes, p.wsBytes);
        return SPRK_EWORKSPACE;
    }
    if ((((uintptr_t)x | (uintptr_t)gy) & 15) != 0) {
        set_error("wgrad16 (1x1): tensors must be 16-byte aligned");
        return SPRK_EINVAL;
    }
    Wg1Args a{};
    a.x = x; a.gy = gy; a.partial = (float *)ws;
    a.N = c.N; a.Cin = c.C1; a.Cout = c.Cout; a.HW = c.H * c.W;
    a.regPerImg = p.regPerImg; a.nRegions = p.nRegions; a.perPart = p.perPart;
    static const int diag = sprk::diag_env("SPRK_C16_DIAG");
    a.diag = diag;
    dim3 grid(p.parts, p.coBlocks, p.ciBlocks);
    auto go = [&](auto kernel) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)p.ldsBytes) != hipSuccess) {
            set_error("wgrad16 (1x1): cannot reserve %zu bytes of LDS", p.ldsBytes);
            return (int)SPRK_ELAUNCH;
        }
        hipLaunchKernelGGL(kernel, grid, dim3(kWgThreads), p.ldsBytes, s, a);
        return (int)SPRK_OK;
    };
    auto pick = [&](auto tag) {
        using T = decltype(tag);
        if (c.x16) return p.wide ? go(wgrad16_1x1_kernel<T, 4, 3, 6, true>) : go(wgrad16_1x1_kernel<T, 2, 3, 6, true>);
        return p.wide ? go(wgrad16_1x1_kernel<T, 4, 3, 6, false>) : go(wgrad16_1x1_kernel<T, 2, 3, 6, false>);
    };
    prof_begin(c.kclass, c.flops, s);
    prof_bytes((c.x16 ? 2.0 : 4.0) * c.N * ((double)(c.C1 + c.C2) * c.H * c.W + (double)c.Cout * c.Hout * c.Wout));
    const int rc = (c.dtype & SPRK_DT_MASK) == SPRK_DT_BF16 ? pick(__bf16{}) : pick(_Float16{});
    if (rc) return rc;
    prof_end(c.kclass, s);
    if (int rc2 = check_launch("wgrad16_1x1")) return rc2;
    g_wgrad16_launches.fetch_add(1, std::memory_order_relaxed);
    const sprk_reduce_item it{(const float *)ws, gw, SPRK_RED_ROWS, p.parts, c.Cout * c.C1, 0, 0, 0};
    return finish_or_defer(it, item, s);
}

long wgrad16_launches() { return g_wgrad16_launches.load(); }

int wgrad16_run(const Wgrad16Call &c, const void *x, const void *x2, const void *gy, float *gw, void *ws,
                size_t ws_bytes, sprk_reduce_item *item, hipStream_t s) {
    PlanW p;
    if (!plan_wg16(c, &p)) {
        Plan1 p1;
        if (plan_wg1x1(c, &p1)) return wgrad16_run_1x1(c, p1, x, gy, gw, ws, ws_bytes, item, s);
        set_error("wgrad16: geometry not eligible");
        return SPRK_EINVAL;
    }
    if (ws_bytes < p.wsBytes || !ws) {
        set_error("wgrad16: workspace too small (%zu < %zu)", ws_bytes, p.wsBytes);
        return SPRK_EWORKSPACE;
    }
    Wg16Args a{};
    a.x = x; a.x2 = x2; a.gy = gy; a.partial = (float *)ws;
    a.N = c.N; a.C1 = c.C1; a.C2 = c.C2; a.H = c.H; a.W = c.W; a.Cout = c.Cout; a.padT = c.padT; a.padL = c.padL;
    a.regX = p.regX; a.regY = p.regY; a.seg = p.seg; a.segLen = p.segLen; a.nUnits = p.nUnits; a.xcs = p.xcs; a.tail = p.tail;
    static const int diag = sprk::diag_env("SPRK_C16_DIAG");
    a.diag = diag;
    a.xcd = xcd_on();
    const int dt = c.dtype & SPRK_DT_MASK;
    dim3 grid(p.parts, p.nBlocks);
    auto go = [&](auto kernel) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)p.ldsBytes) != hipSuccess) {
            set_error("wgrad16: cannot reserve %zu bytes of LDS", p.ldsBytes);
            return (int)SPRK_ELAUNCH;
        }
        hipLaunchKernelGGL(kernel, grid, dim3(kWgThreads), p.ldsBytes, s, a);
        return (int)SPRK_OK;
    };
    auto pick = [&](auto tag) {
        using T = decltype(tag);
        auto shape = [&](auto x16c, auto pl1c) {
            constexpr bool X = decltype(x16c)::value, P1 = decltype(pl1c)::value;
            if (p.lgRW == 6) return p.MC == 6 ? go(wgrad16_kernel<T, 6, 6, X, P1>) : go(wgrad16_kernel<T, 3, 6, X, P1>);
            if (p.lgRW == 5) return p.MC == 6 ? go(wgrad16_kernel<T, 6, 5, X, P1>) : go(wgrad16_kernel<T, 3, 5, X, P1>);
            return p.MC == 6 ? go(wgrad16_kernel<T, 6, 4, X, P1>) : go(wgrad16_kernel<T, 3, 4, X, P1>);
        };
        const bool pl1 = c.padL == 1;
        if (c.x16) return pl1 ? shape(std::true_type{}, std::true_type{}) : shape(std::true_type{}, std::false_type{});
        return pl1 ? shape(std::false_type{}, std::true_type{}) : shape(std::false_type{}, std::false_type{});
    };
    prof_begin(c.kclass, c.flops, s);
    prof_bytes((c.x16 ? 2.0 : 4.0) * c.N * ((double)(c.C1 + c.C2) * c.H * c.W + (double)c.Cout * c.Hout * c.Wout));
    const int rc = dt == SPRK_DT_BF16 ? pick(__bf16{}) : pick(_Float16{});
    if (rc) return rc;
    prof_end(c.kclass, s);
    if (int rc2 = check_launch("wgrad16")) return rc2;
    const long n = (long)c.Cout * (c.C1 + c.C2) * 9;
    g_wgrad16_launches.fetch_add(1, std::memory_order_relaxed);
    const sprk_reduce_item it{(const float *)ws, gw, SPRK_RED_ROWS, p.parts, (int)n, 0, 0, 0};
    return finish_or_defer(it, item, s);
}

}  // namespace sprk
