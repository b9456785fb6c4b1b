// Winograd F(2x2, 3x3) convolution on the fp32 MFMA pipes for the wide 3x3 layers of the U-Net (stride 1,
// dilation 1, output size = input size): forward (ShiftConv2d body, models/joint_network_v2.py:565-584 of the
// reference) and backward-data (the same correlation with flipped, channel-transposed taps).  2.25x fewer
// multiplies than the direct implicit GEMM of conv.hip, which stays the path for every other geometry.
//
//   Y = A^T [ sum_c (G g G^T) .* (B^T d B) ] A      per 2x2 output tile, 4x4 input tile d, 3x3 taps g
//
// Workgroup = 8 waves, 8 x 32 output pixels (4 x 16 tiles) x one group of NT*16 output channels.  Wave
// (pg, th) owns row pg of the 4x4 transform domain for the 32 tiles of half th: 4 positions x 2 pixel tiles x
// NT channel tiles of 16x16 accumulators.  Per K-chunk of 4 input channels:
//   * the raw input tile (10 x 40 floats per channel) and the pre-transformed weights U[pos][k][cout] arrive
//     by buffer_load ... lds, double-buffered, one barrier per chunk; U is laid out [pos][nt][k][16] so
//     that a B read is 64 consecutive floats and every operand offset is an immediate;
//   * lane (tile l15, channel lq) reads its two raw rows and builds its own A operands in registers
//     (row pg of B^T d, then the column pass): 8 FMAs + 8 adds per 48 MFMAs, no LDS round trip;
//   * B operands are ds_read from U.
// After the K loop the column half of A^T . A happens in registers, the row half across the four pg waves
// through LDS, laid out so that both the writes and the transposing reads are bank-conflict free.
#include <algorithm>
#include <cstdio>
#include <cstdlib>

#include "common.h"
#include "wino.h"
#include "wprep_dev.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ unsigned fbits(float v) { return __builtin_bit_cast(unsigned, v); }
typedef __attribute__((address_space(3))) void lds_void;
typedef __amdgpu_buffer_rsrc_t rsrc_t;
constexpr int kXZero = (int)0x80000000;   // beyond num_records: the DMA writes zeros
constexpr float kLeak = 0.1f;

__device__ __forceinline__ rsrc_t make_rsrc(const void *p) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, 0x7FFFFFFF, 0x00020000);
}
// a wave-uniform pointer that the compiler computed on the vector ALU (64-bit multiplies): back into SGPRs
__device__ __forceinline__ const float *uniform_ptr(const float *p) {
    const unsigned long v = (unsigned long)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return (const float *)(((unsigned long)hi << 32) | lo);
}
__device__ __forceinline__ void bdma16(rsrc_t r, int voff, int soff, float *lds_wave_base) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void *)lds_wave_base, 16, voff, soff, 0, 0);
}
// LDS addresses as plain integers: one VGPR base per operand stream, everything else immediates
typedef const __attribute__((address_space(3))) float *lds_cfp;
__device__ __forceinline__ int lds_addr(const void *p) {
    return (int)(unsigned)(__SIZE_TYPE__)(const __attribute__((address_space(3))) char *)p;
}
__device__ __forceinline__ lds_cfp lds_f(int byte_addr) { return (lds_cfp)(__SIZE_TYPE__)(unsigned)byte_addr; }

// workgroup barrier that orders LDS traffic only: no wait for outstanding global loads, stores or LDS-DMA (vmcnt), which
// __syncthreads() would add — their issuers wait where the data is needed
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

__device__ __forceinline__ float act_fn(float v, int act) {
    if (act == SPRK_ACT_LEAKY) return v > 0.f ? v : v * kLeak;
    if (act == SPRK_ACT_RELU) return v > 0.f ? v : 0.f;
    return v;
}

constexpr int CK = sprk::kWinoCK;                      // channels per chunk = one MFMA k-step per position
// Workgroup geometry: 64 tiles as 4 x 16 (8 x 32 output pixels; images whose width is a multiple of 32) or as
// 8 x 8 (16 x 16 output pixels; widths that are only a multiple of 16, e.g. the 16 x 16 U-Net level)
template <int SQ>
struct Geo {
    static constexpr int TR = SQ ? 16 : 8, TC = SQ ? 16 : 32;   // output pixels per workgroup
    static constexpr int RP = TC + 8, RPLANE = (TR + 2) * RP;   // raw tile rows x columns (16-byte column groups)
    static constexpr int LGTX = SQ ? 3 : 4;                     // log2 of the tiles per tile row
};
constexpr int RAWF = 1792;                             // 4 planes = 400 / 432 DMA lanes -> 7 waves x 256 floats
static_assert(CK * Geo<0>::RPLANE <= RAWF && CK * Geo<1>::RPLANE <= RAWF, "raw tile fits its stage");
constexpr int ETS = 68;                                // exchange stride per channel: 64 tiles + 4
constexpr int kThreads = 512;
constexpr int kEFloats = 4 * 2 * 32 * ETS;
constexpr int kECFloats = 2 * 96;                      // per-channel (scale, shift) of the workgroup's channel group

// U of one chunk: [pos][nt][k][16]: the 64 lanes of a B read (k = lane / 16, channel = lane % 16) hit 64
// consecutive floats (no padding, no bank conflicts), and every (pos, nt) operand sits a multiple of 256 bytes
// from the lane's base address, which is what ds_read2st64_b32 encodes as an immediate.  Even NT: [pos][nt / 2][k][16][2]
// — the lane's operands of two channel tiles in one 8-byte read, pairs 512 bytes apart (ds_read2st64_b64): half the LDS
// instructions and waits per position (every instruction between the MFMAs costs the SIMD ~6.5 cycles)
__host__ __device__ constexpr int ufloats_of(int NT) { return 16 * NT * CK * 16; }
__host__ __device__ constexpr size_t lds_bytes_of(int NT) {
    return (size_t)(2 * RAWF + 2 * ufloats_of(NT) + kEFloats + kECFloats) * 4;   // two stages + the output-transform
                                                                                // exchange + epilogue constants
}

// The pre-transformed weights U[group][chunk][pos][nt][k][16] = (G g G^T)[pos] are a prepared-weight item
// (wprep_dev.h: wprep_wino): zero rows for the channels a source's last chunk does not have, zero columns past Cout;
// mode 0: g = w[cout][cin][u][v] (forward), mode 1: g = w[k][n][2-u][2-v] (backward-data).
template <int V>
struct IC {
    static constexpr int value = V;
};

// diagnostic build (-DWINO_STAMP; scratch/r3/stamp_run.sh): s_memtime stamps around the phases of iteration 10 of a
// workgroup's first tile, per wave, and of the kernel (clock = s_memtime / s_memrealtime), dumped by the host after the
// fifth launch of the 96-channel kernel
#ifdef WINO_STAMP
#define STAMP(k) do { __builtin_amdgcn_sched_barrier(0); if (P == 0) { unsigned long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); st[(k)] = (unsigned)t_; } __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define STAMP(k)
#endif
struct KArgs {
    const float *x, *x2, *U, *bias, *scale, *shift, *mask;
    float *y;
    int N, C1, C2, H, W, Cout, padT, padL, act, tilesX, tilesY, ntiles, nc1, nch, up2, xcd, diag, mact;
#ifdef WINO_STAMP
    unsigned *dbg;
#endif
};

template <int NT, int SQ>
__global__ __launch_bounds__(kThreads, 1) void wino_conv_kernel(const KArgs a) {
    constexpr int UFLOATS = ufloats_of(NT);
    constexpr int TR = Geo<SQ>::TR, TC = Geo<SQ>::TC, RP = Geo<SQ>::RP, RPLANE = Geo<SQ>::RPLANE, LGTX = Geo<SQ>::LGTX;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *Ub = smem;                  // 2 x UFLOATS
    float *Rb = smem + 2 * UFLOATS;    // 2 x RAWF
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), l15 = lane & 15,
              lq = lane >> 4;
    const int pg = wave & 3, th = wave >> 2;
    const int grp = blockIdx.y;
    const long HW = (long)a.H * a.W;
    // per-channel epilogue constants of this channel group, once per workgroup: v * EC[c] + EC[96 + c] (a load inside
    // the output transform would put a memory round trip on the critical path of every tile)
    float *EC = smem + 2 * UFLOATS + 2 * RAWF + kEFloats;
    if (tid < NT * 16) {
        const int co = grp * (NT * 16) + tid;
        float sc = 1.f, sh = 0.f;
        if (co < a.Cout) {
            if (a.scale) {
                sc = a.scale[co];
                sh = a.shift[co];
            } else if (a.bias) {
                sh = a.bias[co];
            }
        }
        EC[tid] = sc;
        EC[96 + tid] = sh;
    }

    // raw-tile DMA: lane q moves 16 bytes of plane q / (RPLANE/4), row and column group from the remainder.
    // The workgroup is persistent (tiles blockIdx.x, + gridDim.x, ...): the lane keeps its position inside the raw
    // tile, the tile origin goes into the scalar base pointer, validity is re-derived per tile.
    constexpr int RAWQ = CK * RPLANE / 4;
    int dch = 4, lrow = 0x7000, lcol = 0, loff = 0;
    if (tid < RAWQ) {
        dch = tid / (RPLANE / 4);
        const int rem = tid % (RPLANE / 4);
        lrow = rem / (RP / 4);
        lcol = 4 * (rem % (RP / 4)) - 4;
        loff = (int)(((long)dch * a.H + lrow) * a.W + lcol + 4) * 4;
    }
    int n = 0, by = 0, bx = 0, voff = kXZero;   // tile whose input is being staged
    auto set_tile = [&](int tile) {
        // wave-uniform, but integer division runs on the vector ALU: back into SGPRs
        const int t0 = __builtin_amdgcn_readfirstlane(tile);
        const int q0 = __builtin_amdgcn_readfirstlane(t0 / a.tilesX);
        bx = t0 - q0 * a.tilesX;
        n = __builtin_amdgcn_readfirstlane(q0 / a.tilesY);
        by = q0 - n * a.tilesY;
        const unsigned gy = (unsigned)(by * TR - a.padT + lrow), gx = (unsigned)(bx * TC + lcol);
        voff = (gy < (unsigned)a.H && gx < (unsigned)a.W) ? loff : kXZero;   // W % 4 == 0: a group is all in or all out
    };
    // The raw chunks of a tile are issued in order (0, 1, 2, ...): a scalar cursor walks the planes of source 1, then of
    // source 2, so that an issue is a handful of scalar instructions (every instruction a wave issues between its MFMAs
    // costs the SIMD 5 - 8 cycles: see the stamps in DESIGN.md section 4.1b).
    const float *rawcur = nullptr, *rawsrc2 = nullptr;
    auto raw_begin = [&]() {   // after set_tile
        const long org = (long)(by * TR - a.padT) * a.W + bx * TC - 4;   // first raw pixel of the tile (may lie outside)
        rawcur = uniform_ptr(a.x + (long)n * a.C1 * HW + org);
        rawsrc2 = a.C2 > 0 ? uniform_ptr(a.x2 + (long)n * a.C2 * HW + org) : nullptr;
    };
    // chunks with fewer than CK channels: the last one of a source whose channel count is not a multiple of CK
    const int rag1 = (a.C1 % CK) ? a.nc1 - 1 : -1, rag2 = (a.C2 % CK) ? a.nch - 1 : -1;
    auto issue_raw = [&](int c, float *dst, auto ragged) {
        if (c == a.nc1) rawcur = rawsrc2;
        const rsrc_t rs = make_rsrc(rawcur);
        rawcur += (long)CK * HW;
        int vo = voff;
        if constexpr (decltype(ragged)::value != 0)
            if (c == rag1 || c == rag2) vo = dch < (c == rag1 ? a.C1 % CK : a.C2 % CK) ? voff : kXZero;
        // lanes past the tile's RAWQ groups (wave 6) carry kXZero: zeros into the stage's padding, no lane mask
        if (wave < 7) bdma16(rs, vo, 0, dst + wave * 256);
    };
    const float *Ug = a.U + (long)grp * a.nch * UFLOATS;
    auto issue_u = [&](int c, float *dst) {
        const rsrc_t ur = make_rsrc(Ug + (long)c * UFLOATS);
        constexpr int full = UFLOATS / 2048, rest = UFLOATS % 2048;   // 2048 floats per sweep of the workgroup
#pragma unroll
        for (int gi = 0; gi < full; ++gi) bdma16(ur, tid * 16, gi * 8192, dst + gi * 2048 + wave * 256);
        if (rest && wave < rest / 256) bdma16(ur, tid * 16, full * 8192, dst + full * 2048 + wave * 256);
    };

    f32x4 acc[4][2][NT];

    // row pg of B^T d = d[ra] + sgn * d[rb]:  (0,2,-), (1,2,+), (2,1,-), (1,3,-)
    const int ra = pg == 0 ? 0 : pg == 2 ? 2 : 1, rb = pg == 3 ? 3 : pg == 2 ? 1 : 2;
    const float sgn = pg == 1 ? 1.f : -1.f;
    // one VGPR base per operand stream and stage, kept opaque so that every other offset is an immediate of
    // ds_read2_b32 / ds_read2st64_b32 (no address arithmetic in the K loop)
    int rawA[2], rawB[2];
    // tile 32 th + 16 mt + l15 -> tile row (tile >> LGTX), tile column (tile & (2^LGTX - 1))
    const int trow0 = (32 * th + l15) >> LGTX, tcol0 = l15 & ((1 << LGTX) - 1);
    constexpr int MTROWS = 16 >> LGTX;   // tile rows between the wave's two tile groups
    rawA[0] = lds_addr(Rb + lq * RPLANE + (2 * trow0 + ra) * RP + (4 - a.padL) + 2 * tcol0);
    rawB[0] = lds_addr(Rb + lq * RPLANE + (2 * trow0 + rb) * RP + (4 - a.padL) + 2 * tcol0);
    rawA[1] = rawA[0] + RAWF * 4;
    rawB[1] = rawB[0] + RAWF * 4;
    int bbase = lds_addr(Ub + (4 * pg) * NT * 64 + (NT % 2 == 0 ? 2 * lane : lane));
    asm volatile("" : "+v"(rawA[0]), "+v"(rawA[1]), "+v"(rawB[0]), "+v"(rawB[1]), "+v"(bbase));

    // lane (l15, lq): tiles 32 th + 16 mt + l15 of channel lq -> its A operands for the wave's 4 positions
    auto transform = [&](auto stage, float (&av)[2][4]) {
        constexpr int S = decltype(stage)::value;
        const lds_cfp pa = lds_f(rawA[S]), pb = lds_f(rawB[S]);
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            float xv[4];
#pragma unroll
            for (int j = 0; j < 4; ++j)
                xv[j] = __builtin_fmaf(pb[mt * 2 * MTROWS * RP + j], sgn, pa[mt * 2 * MTROWS * RP + j]);
            av[mt][0] = xv[0] - xv[2];
            av[mt][1] = xv[1] + xv[2];
            av[mt][2] = xv[2] - xv[1];
            av[mt][3] = xv[1] - xv[3];
        }
    };
    // first: the tile's first chunk — the C operand is the constant 0 instead of 192 accumulator registers cleared
    // by 192 VALU moves per tile (each of which costs MFMA issue time)
    auto mfma12 = [&](const float (&av)[2][4], const float (&bv)[NT], auto pos, auto first) {
        constexpr int p = decltype(pos)::value;
        constexpr bool FIRST = decltype(first)::value != 0;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                if constexpr (FIRST)
                    acc[p][mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[mt][p], bv[nt], (f32x4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                else
                    acc[p][mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[mt][p], bv[nt], acc[p][mt][nt], 0, 0, 0);
            }
    };
    auto loadb = [&](float (&bv)[NT], auto stage, auto pos) {
        constexpr int S = decltype(stage)::value, p = decltype(pos)::value;
        if constexpr (NT % 2 == 0) {   // pairs of channel tiles: 8-byte reads, 512 bytes apart (ds_read2st64_b64)
            typedef const __attribute__((address_space(3))) f32x2 *lds_cf2p;
#pragma unroll
            for (int q = 0; q < NT / 2; ++q) {
                const f32x2 v = *(lds_cf2p)(__SIZE_TYPE__)(unsigned)(bbase + (S * UFLOATS + (p * (NT / 2) + q) * 128) * 4);
                bv[2 * q] = v[0];
                bv[2 * q + 1] = v[1];
            }
        } else {
            const lds_cfp bp = lds_f(bbase);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) bv[nt] = bp[S * UFLOATS + (p * NT + nt) * 64];
        }
    };

#ifdef WINO_STAMP
    unsigned st[8] = {}, sv[8] = {};
    if (NT == 6 && a.dbg && blockIdx.x < 4 && lane == 0) {
        a.dbg[(blockIdx.x * 8 + wave) * 16 + 10] = (unsigned)__builtin_amdgcn_s_memtime();   // kernel start
        a.dbg[(blockIdx.x * 8 + wave) * 16 + 11] = (unsigned)__builtin_amdgcn_s_memrealtime();
    }
#endif
    const int nch = a.nch;
    int tile = xcd_slot(blockIdx.x, gridDim.x, a.xcd);
    set_tile(tile);
    auto first_loads = [&]() {
        raw_begin();
        issue_raw(0, Rb, IC<1>{});
        issue_u(0, Ub);
        if (nch > 1) issue_raw(1, Rb + RAWF, IC<1>{});
    };
    first_loads();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    float aop[2][2][4];
    transform(IC<0>{}, aop[0]);
    __syncthreads();
    // iteration c: aop[c&1] = operands of chunk c, U[c&1] holds chunk c, raw[(c+1)&1] the raw tile of chunk c+1
    // checked = 0: the steady-state copies (a full chunk c + 2 exists), no compares and branches around the DMA
    auto iter = [&](auto par, int c, auto first, auto checked) {
        constexpr int P = decltype(par)::value;
        constexpr bool CHK = decltype(checked)::value != 0;
        STAMP(0);
        if (!CHK || c + 1 < nch) issue_u(c + 1, Ub + (1 - P) * UFLOATS);
        if (!CHK || c + 2 < nch) issue_raw(c + 2, Rb + P * RAWF, checked);
        STAMP(1);
        // the first position's 12 MFMAs go out before the next chunk's input transform: its LDS reads and ~40 VALU
        // instructions then run while the matrix pipe works, instead of in front of an idle pipe (both waves of a SIMD
        // leave the barrier together, so neither covers the other's stall)
        float b0[NT], b1[NT];
        if constexpr (NT == 6) {
            // next chunk's input transform, then the four positions
            transform(IC<1 - P>{}, aop[1 - P]);   // past the last chunk this reads a stale tile: never used
            STAMP(2);
            loadb(b0, IC<P>{}, IC<0>{}); mfma12(aop[P], b0, IC<0>{}, first);
            STAMP(3);
            loadb(b0, IC<P>{}, IC<1>{}); mfma12(aop[P], b0, IC<1>{}, first);
            STAMP(4);
            loadb(b0, IC<P>{}, IC<2>{}); mfma12(aop[P], b0, IC<2>{}, first);
            STAMP(5);
            loadb(b0, IC<P>{}, IC<3>{}); mfma12(aop[P], b0, IC<3>{}, first);
            STAMP(6);
            (void)b1;
        } else {
            // 48-channel layers (24 MFMAs per chunk and wave: LDS latency weighs twice as much): the B operands of
            // position p + 1 are in flight while position p is multiplied, and the input transform of the next chunk
            // sits between positions 0 and 1.  Same-box A/B (scratch/r3/wino_ab.sh): 48->48 at 128x64^2 132.9 ->
            // 128.3 us; the same order on the 96-channel kernel (256 VGPRs, no room for the second operand set) is
            // neutral to 1 % slower, as are "position 0 first" and an MFMA / LDS / VALU interleave by
            // sched_group_barrier (445 -> 467 us) — its K loop is not bound by instruction order.
            loadb(b0, IC<P>{}, IC<0>{});
            loadb(b1, IC<P>{}, IC<1>{});
            __builtin_amdgcn_sched_barrier(0);
            mfma12(aop[P], b0, IC<0>{}, first);
            __builtin_amdgcn_sched_barrier(0);
            loadb(b0, IC<P>{}, IC<2>{});
            transform(IC<1 - P>{}, aop[1 - P]);
            __builtin_amdgcn_sched_barrier(0);
            mfma12(aop[P], b1, IC<1>{}, first);
            __builtin_amdgcn_sched_barrier(0);
            loadb(b1, IC<P>{}, IC<3>{});
            __builtin_amdgcn_sched_barrier(0);
            mfma12(aop[P], b0, IC<2>{}, first);
            __builtin_amdgcn_sched_barrier(0);
            mfma12(aop[P], b1, IC<3>{}, first);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        STAMP(7);
        __syncthreads();
#ifdef WINO_STAMP
        if (c == 10) for (int k = 0; k < 8; ++k) sv[k] = st[k];
#endif
    };
    for (;;) {
        iter(IC<0>{}, 0, IC<1>{}, IC<1>{});
        if (1 < nch) iter(IC<1>{}, 1, IC<0>{}, IC<1>{});
        int c = 2;
        if (rag1 < 0)
            for (; c + 4 < nch; c += 2) {   // both iterations of the pair fetch a full chunk (not the last one: rag2)
                iter(IC<0>{}, c, IC<0>{}, IC<0>{});
                iter(IC<1>{}, c + 1, IC<0>{}, IC<0>{});
            }
        for (; c < nch; c += 2) {
            iter(IC<0>{}, c, IC<0>{}, IC<1>{});
            if (c + 1 < nch) iter(IC<1>{}, c + 1, IC<0>{}, IC<1>{});
        }
        // The stages are free again: start the next tile's first loads now, under this tile's output transform.
        const int e_n = n, e_by = by, e_bx = bx;
        const int next = tile + (int)gridDim.x;
        const bool more = next < a.ntiles;
        if (more) {
            set_tile(next);
            first_loads();
        }

    // output transform.  E[pg][b][channel'][tile]: an accumulator quad holds tiles 32 th + 16 mt + 4 lq + r, r = 0..3 —
    // four consecutive floats of a channel row, ONE ds_write_b128 (the 8 lanes a 16-byte write serves per cycle are 8
    // channels, 68 floats apart: 32 different banks; the first form wrote 4 x ds_write_b32 to a permuted tile index);
    // 32 channels a pass.
    float *E = smem + 2 * UFLOATS + 2 * RAWF;
    // reader lane -> tile (row rl_ty, column rl_tx) = tile index T = (rl_ty << LGTX) + rl_tx = lane
    const int rl_tx = lane & ((1 << LGTX) - 1), rl_ty = lane >> LGTX;
    const int tprime = lane;
    // stores: a wave-uniform plane pointer (scalar arithmetic) + the lane's fixed byte offset inside the tile
    const int lane_off = a.up2 ? ((4 * rl_ty) * (2 * a.W) + 4 * rl_tx) * 4 : ((2 * rl_ty) * a.W + 2 * rl_tx) * 4;
    const long tile_org = a.up2 ? ((long)(2 * e_by * TR) * (2 * a.W) + 2 * e_bx * TC) : ((long)(e_by * TR) * a.W + e_bx * TC);
    const long plane = a.up2 ? 4 * HW : HW;
    constexpr int PASSES = (NT + 1) / 2;
    if (!(a.diag & 1))
#pragma unroll
    for (int pass = 0; pass < PASSES; ++pass) {
#pragma unroll
        for (int n2 = 0; n2 < 2; ++n2) {
            const int nt = pass * 2 + n2;
            if (nt < NT) {
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
                    // packed adds on the accumulator quads; same association as the scalar form
                    const f32x4 s0 = (acc[0][mt][nt] + acc[1][mt][nt]) + acc[2][mt][nt];
                    const f32x4 s1 = (acc[1][mt][nt] - acc[2][mt][nt]) - acc[3][mt][nt];
                    float *e = E + ((pg * 2) * 32 + n2 * 16 + l15) * ETS + 4 * lq + 16 * mt + 32 * th;
                    *reinterpret_cast<f32x4 *>(e) = s0;
                    *reinterpret_cast<f32x4 *>(e + 32 * ETS) = s1;
                }
            }
        }
        lds_barrier();   // LDS only: the next tile's first loads and the previous pass's stores stay in flight
        // the reader in three forms picked once per pass (plain / masked by a fused activation backward / fused x2
        // up-sampling stores) instead of run-time branches per channel; channel planes by pointer increments
        auto reader = [&](auto modec) {
            constexpr int MODE = decltype(modec)::value;   // 0 plain, 1 mask, 2 up2
            const long ch0 = (long)e_n * a.Cout + grp * (NT * 16) + pass * 32 + wave;
            const float *yc = uniform_ptr(a.y + ch0 * plane + tile_org);
            const float *mc = MODE == 1 ? uniform_ptr(a.mask + ch0 * plane + tile_org) : nullptr;
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int cp = wave + 8 * it;   // channel within the pass
                const int co = grp * (NT * 16) + pass * 32 + cp;
                if (pass * 32 + cp < NT * 16 && co < a.Cout) {
                    // mask of a fused activation backward: the saved activations at the output positions, fetched first
                    f32x2 mk0 = {1.f, 1.f}, mk1 = {1.f, 1.f};
                    if constexpr (MODE == 1) {
                        const rsrc_t mr = make_rsrc(mc);
                        mk0 = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(mr, lane_off, 0, 0));
                        mk1 = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(mr, lane_off, 4 * a.W, 0));
                    }
                    f32x2 s[4];
#pragma unroll
                    for (int w = 0; w < 4; ++w) {
                        s[w][0] = E[((w * 2 + 0) * 32 + cp) * ETS + tprime];
                        s[w][1] = E[((w * 2 + 1) * 32 + cp) * ETS + tprime];
                    }
                    const float sc = EC[pass * 32 + cp], sh = EC[96 + pass * 32 + cp];
                    f32x2 o0 = ((s[0] + s[1]) + s[2]) * sc + sh;
                    f32x2 o1 = ((s[1] - s[2]) - s[3]) * sc + sh;
                    if (a.act == SPRK_ACT_LEAKY) {   // max(v, 0.1 v) = v > 0 ? v : 0.1 v
                        o0 = __builtin_elementwise_max(o0, o0 * kLeak);
                        o1 = __builtin_elementwise_max(o1, o1 * kLeak);
                    } else if (a.act == SPRK_ACT_RELU) {
                        o0 = __builtin_elementwise_max(o0, (f32x2){0.f, 0.f});
                        o1 = __builtin_elementwise_max(o1, (f32x2){0.f, 0.f});
                    }
                    if constexpr (MODE == 1) {
                        const float neg = a.mact == SPRK_ACT_LEAKY ? kLeak : 0.f;
                        o0[0] *= mk0[0] > 0.f ? 1.f : neg;
                        o0[1] *= mk0[1] > 0.f ? 1.f : neg;
                        o1[0] *= mk1[0] > 0.f ? 1.f : neg;
                        o1[1] *= mk1[1] > 0.f ? 1.f : neg;
                    }
                    // buffer stores: wave-uniform descriptor (plane + tile origin, scalar arithmetic), the lane's fixed
                    // byte offset, the row stride in the scalar offset — no vector address arithmetic in the epilogue
                    const rsrc_t yr = make_rsrc(yc);
                    if constexpr (MODE == 2) {   // nearest x2 upsampling fused into the stores: each value to its 2x2 block
                        const int W2b = 8 * a.W;
                        const u32x4 r0 = {fbits(o0[0]), fbits(o0[0]), fbits(o0[1]), fbits(o0[1])},
                                    r1 = {fbits(o1[0]), fbits(o1[0]), fbits(o1[1]), fbits(o1[1])};
                        __builtin_amdgcn_raw_buffer_store_b128(r0, yr, lane_off, 0, 0);
                        __builtin_amdgcn_raw_buffer_store_b128(r0, yr, lane_off, W2b, 0);
                        __builtin_amdgcn_raw_buffer_store_b128(r1, yr, lane_off, 2 * W2b, 0);
                        __builtin_amdgcn_raw_buffer_store_b128(r1, yr, lane_off, 3 * W2b, 0);
                    } else {
                        __builtin_amdgcn_raw_buffer_store_b64((u32x2){fbits(o0[0]), fbits(o0[1])}, yr, lane_off, 0, 0);
                        __builtin_amdgcn_raw_buffer_store_b64((u32x2){fbits(o1[0]), fbits(o1[1])}, yr, lane_off, 4 * a.W, 0);
                    }
                }
                yc += 8 * plane;
                if constexpr (MODE == 1) mc += 8 * plane;
            }
        };
        if (a.up2)
            reader(IC<2>{});
        else if (a.mask)
            reader(IC<1>{});
        else
            reader(IC<0>{});
        lds_barrier();
    }
#ifdef WINO_STAMP
        if (NT == 6 && a.dbg && blockIdx.x < 4 && lane == 0) {
            if (tile < (int)gridDim.x) a.dbg[(blockIdx.x * 8 + wave) * 16 + 12] = (unsigned)__builtin_amdgcn_s_memtime();   // first tile done
            a.dbg[(blockIdx.x * 8 + wave) * 16 + 13] = (unsigned)__builtin_amdgcn_s_memtime();   // (last) tile done
            a.dbg[(blockIdx.x * 8 + wave) * 16 + 14] = (unsigned)__builtin_amdgcn_s_memrealtime();
        }
#endif
        if (!more) break;
        // next tile: its first raw tile and U chunk were issued before the output transform
        tile = next;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        transform(IC<0>{}, aop[0]);
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------
// Winograd backward-weight:  dW = G^T [ sum over images and 2x2 tiles of (A dY A^T) .* (B^T d B) ] G
// 16 independent [Cin x tiles].[tiles x Cout] GEMMs (one per position of the 4x4 transform domain), the
// reduced dimension is the tile index (k = 4 tiles per MFMA step).  Both operands are transformed in registers
// from raw LDS tiles: lane (l15 = input channel, lq = tile) builds its A operands exactly as in the forward
// kernel, lane (lq = tile, l15 = output channel) its B operands (row pg of A dY, then the column pass; the two
// minus signs of A are folded into the final transform).  Workgroup = 8 waves (pg = row of the transform
// domain, h = half of the 96 output channels), 16*MT input channels (grid.y groups), accumulators
// 4 positions x MT x 3 tiles; it walks regions of 4 x 16 output pixels (16 tiles = 4 k-steps) of all images,
// grid.x-strided, double-buffered by buffer_load ... lds, and leaves a partial dW per grid.x for a fixed-order
// reduction.  LDS layouts keep every operand of a k-step within ds_read2_b32's 1020-byte immediate range of one
// base register per (operand row, stage):
//   raw d : [l15 (16)][row (6)][mt (MT)][24 floats] + 4 pad  -> channel 16 mt + l15
//   dY    : [h (2)][l15 (16)][row (4)][nt (3)][16 floats] + 4 pad -> channel 48 h + 16 nt + l15
// ------------------------------------------------------------------------------------------
constexpr int WRH = 4, WRW = 16, WRP = 24;
constexpr int WPB = 4 * 3 * 16 + 4;
constexpr int WAF = 4 * 2048, WSTAGE = 2 * WAF;
constexpr size_t kWgLdsBytes = (size_t)(2 * WSTAGE) * 4;   // two stages

struct WgKArgs {
    const float *x, *gy;
    float *partial;         // [parts][Cout][CinTot][9]
    int N, Csrc, cbase, ci0, CinTot, H, W, Cout, padT, padL, regionsX, regionsY, xcd;   // cbase: first channel of group 0
};

template <int MT>
__global__ __launch_bounds__(kThreads, 1) void wino_wgrad_kernel(const WgKArgs a) {
    constexpr int PA = 6 * MT * WRP + 4, NA4 = 16 * (PA / 4), NB4 = 32 * (WPB / 4);
    static_assert(16 * PA <= WAF && 32 * WPB <= WAF, "stage holds both operands");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), l15 = lane & 15,
              lq = lane >> 4;
    const int pg = wave & 3, h = wave >> 2;
    const int grp = blockIdx.y;
    const int H = a.H, W = a.W;
    const long HW = (long)H * W;
    const int nregions = a.N * a.regionsY * a.regionsX;
    const int c0 = a.cbase + grp * 16 * MT;   // first channel of this group inside the source
    const int have = a.Csrc - c0;             // valid channels of this group

    // DMA lane table, fixed per workgroup, in REGISTERS (12 of the ~70 the packed transforms freed; it lived in LDS and was
    // re-read, unpacked and range-checked for every region: 195 instructions per region and wave, a quarter of the
    // region's matrix time — every instruction a wave issues beside its MFMAs costs the SIMD ~6.5 cycles).
    // Per slot: byte offset of the lane's 16 bytes inside the region's raw tile / gradient tile (kXZero: pad lane or
    // missing channel), and for the raw tile a 16-bit mask: bit 4 yt + xt says whether the lane's row and column group
    // lie inside the image for a region of row type yt / column type xt (0 first, 1 inner, 2 last, 3 first and last).
    int offA[4], offB[4], mskA[4];
    {
        const int RY = a.regionsY, RX = a.regionsX;
#pragma unroll
        for (int rd = 0; rd < 4; ++rd) {
            const int q = tid + rd * kThreads;
            const int sl = q / (PA / 4), rem = q % (PA / 4);
            const int row = rem / (6 * MT), mt = (rem % (6 * MT)) / 6, c4 = rem % 6;
            const bool va = q < NA4 && rem < 36 * MT && 16 * mt + sl < have;
            offA[rd] = va ? (int)(((16 * mt + sl) * HW + (long)row * W + 4 * c4) * 4) : kXZero;
            int m = 0;
#pragma unroll
            for (int yt = 0; yt < 4; ++yt) {
                const int ry = yt == 0 ? 0 : yt == 1 ? 1 : yt == 2 ? RY - 1 : 0;
                const bool rok = (unsigned)(ry * WRH - a.padT + row) < (unsigned)H;
#pragma unroll
                for (int xt = 0; xt < 4; ++xt) {
                    const int rx = xt == 0 ? 0 : xt == 1 ? 1 : xt == 2 ? RX - 1 : 0;
                    const bool cok = (unsigned)(rx * WRW - 4 + 4 * c4) < (unsigned)W;
                    m |= (rok && cok) ? 1 << (4 * yt + xt) : 0;
                }
            }
            mskA[rd] = va ? m : 0;
            const int sb = q / (WPB / 4), remb = q % (WPB / 4);
            const int rowb = remb / 12, ntb = (remb % 12) / 4, c4b = remb % 4;
            const int chb = (sb >> 4) * 48 + 16 * ntb + (sb & 15);
            offB[rd] = (q < NB4 && remb < 48 && chb < a.Cout) ? (int)((chb * HW + (long)rowb * W + 4 * c4b) * 4) : kXZero;
        }
    }
    // region cursor (scalar): the workgroup walks regions slot, slot + gridDim.x, ...; (rx, ry, n) advance by the
    // decomposed stride with carries instead of two divisions per region
    int cur_rx, cur_ry, cur_n, d_rx, d_ry, d_n;
    {
        const int r0 = xcd_slot(blockIdx.x, gridDim.x, a.xcd);
        const int q0 = r0 / a.regionsX;
        cur_rx = __builtin_amdgcn_readfirstlane(r0 - q0 * a.regionsX);
        cur_n = __builtin_amdgcn_readfirstlane(q0 / a.regionsY);
        cur_ry = __builtin_amdgcn_readfirstlane(q0 - (q0 / a.regionsY) * a.regionsY);
        const int st = gridDim.x, t0 = st / a.regionsX;
        d_rx = __builtin_amdgcn_readfirstlane(st - t0 * a.regionsX);
        d_n = __builtin_amdgcn_readfirstlane(t0 / a.regionsY);
        d_ry = __builtin_amdgcn_readfirstlane(t0 - (t0 / a.regionsY) * a.regionsY);
    }
    const float *xbase = a.x + (long)c0 * HW - (long)a.padT * W - 4;
    const long strideA = (long)a.Csrc * HW, strideB = (long)a.Cout * HW;
    // issues the region under the cursor, then advances the cursor
    auto issue = [&](float *st) {
        const int RY = a.regionsY, RX = a.regionsX;
        const int yt = RY == 1 ? 3 : cur_ry == 0 ? 0 : cur_ry == RY - 1 ? 2 : 1;
        const int xt = RX == 1 ? 3 : cur_rx == 0 ? 0 : cur_rx == RX - 1 ? 2 : 1;
        const int bit = 4 * yt + xt;
        const int inimg = cur_ry * (WRH * W) + cur_rx * WRW;
        const rsrc_t ra = make_rsrc(xbase + cur_n * strideA + inimg), rb = make_rsrc(a.gy + cur_n * strideB + inimg);
#pragma unroll
        for (int rd = 0; rd < 4; ++rd) {
            const int m = __builtin_amdgcn_sbfe(mskA[rd], bit, 1);   // -1: inside the image
            bdma16(ra, (offA[rd] & m) | (kXZero & ~m), 0, st + rd * 2048 + wave * 256);
            bdma16(rb, offB[rd], 0, st + WAF + rd * 2048 + wave * 256);
        }
        cur_rx += d_rx;
        const int cx = cur_rx >= RX;
        cur_rx -= cx ? RX : 0;
        cur_ry += d_ry + cx;
        const int cy = cur_ry >= RY;
        cur_ry -= cy ? RY : 0;
        cur_n += d_n + cy;
    };

    f32x4 acc[4][MT][3];
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < 3; ++nt) acc[p][mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int ra_ = pg == 0 ? 0 : pg == 2 ? 2 : 1, rb_ = pg == 3 ? 3 : pg == 2 ? 1 : 2;
    const float sgn = pg == 1 ? 1.f : -1.f;
    const float zs = pg == 1 ? 1.f : -1.f;
    const int zrow = pg == 3 ? 1 : 0;
    const bool zmix = pg == 1 || pg == 2;
    int baseA[2], baseB2[2], baseB[2];
#pragma unroll
    for (int sgi = 0; sgi < 2; ++sgi) {
        baseA[sgi] = lds_addr(smem + sgi * WSTAGE + l15 * PA + ra_ * MT * WRP + (4 - a.padL) + 2 * lq);
        baseB2[sgi] = lds_addr(smem + sgi * WSTAGE + l15 * PA + rb_ * MT * WRP + (4 - a.padL) + 2 * lq);
        baseB[sgi] = lds_addr(smem + sgi * WSTAGE + WAF + (h * 16 + l15) * WPB + zrow * 48 + 2 * lq);
    }
    asm volatile("" : "+v"(baseA[0]), "+v"(baseA[1]), "+v"(baseB2[0]), "+v"(baseB2[1]), "+v"(baseB[0]), "+v"(baseB[1]));

    // zm: the wave mixes two gradient rows (pg 1, 2) or takes one (pg 0, 3) — a compile-time property of each copy of
    // the region loop (a run-time branch per channel tile and k-step cost 12 taken branches per region)
    auto kstep = [&](auto stage, auto kstp, auto zm) {
        constexpr int S = decltype(stage)::value, ks = decltype(kstp)::value;
        constexpr bool ZMIX = decltype(zm)::value != 0;
        constexpr int trow = ks >> 1, tcol = 8 * (ks & 1);   // tiles 4 ks .. 4 ks + 3: tile row, first column
        const lds_cfp pa = lds_f(baseA[S]), pb = lds_f(baseB2[S]), pz = lds_f(baseB[S]);
        // Both transforms in PACKED fp32 (v_pk_fma_f32 / v_pk_add_f32 on the register pairs ds_read2_b32 returns, with
        // op_sel / neg modifiers doing the shuffles and signs): 30-36 VALU instructions per 36 MFMAs were a quarter of
        // this loop's issue time; now 15-18.  Position 2 is carried NEGATED on both operands (x1 - x2 and z1 - z0), so the
        // products are unchanged.
        float av[MT][4], bv[3][4];
        const f32x2 sgn2 = {sgn, sgn}, zs2 = {zs, zs};
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int o = mt * WRP + 2 * trow * MT * WRP + tcol;
            const f32x2 pa01 = {pa[o], pa[o + 1]}, pa23 = {pa[o + 2], pa[o + 3]};
            const f32x2 pb01 = {pb[o], pb[o + 1]}, pb23 = {pb[o + 2], pb[o + 3]};
            const f32x2 x01 = __builtin_elementwise_fma(pb01, sgn2, pa01), x23 = __builtin_elementwise_fma(pb23, sgn2, pa23);
            f32x2 a01, a23;
            // (x0 - x2, x1 + x2)
            asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,0]" : "=v"(a01) : "v"(x01), "v"(x23));
            // (x1 - x2, x1 - x3): the first is -(x2 - x1)
            asm("v_pk_add_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,1] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(a23) : "v"(x01), "v"(x23));
            av[mt][0] = a01[0];
            av[mt][1] = a01[1];
            av[mt][2] = a23[0];   // negated
            av[mt][3] = a23[1];
        }
#pragma unroll
        for (int nt = 0; nt < 3; ++nt) {
            const int o = nt * 16 + 2 * trow * 48 + tcol;
            f32x2 z = {pz[o], pz[o + 1]};
            if constexpr (ZMIX) z = __builtin_elementwise_fma((f32x2){pz[o + 48], pz[o + 48 + 1]}, zs2, z);
            f32x2 b12;
            // (z0 + z1, z1 - z0): the second is -(z0 - z1)
            asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,0] neg_hi:[0,1]" : "=v"(b12) : "v"(z), "v"(z));
            bv[nt][0] = z[0];
            bv[nt][1] = b12[0];
            bv[nt][2] = b12[1];   // negated
            bv[nt][3] = z[1];     // true value -z1: folded into the final transform
        }
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < 3; ++nt)
                    acc[p][mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[mt][p], bv[nt][p], acc[p][mt][nt], 0, 0, 0);
    };
    auto stage_body = [&](auto stage, int region, auto zm) {
        constexpr int S = decltype(stage)::value;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (region + (int)gridDim.x < nregions) issue(smem + (1 - S) * WSTAGE);
        // (no scheduling fences between the k-steps any more: with the packed transforms the kernel has 186 VGPRs and no
        // spills, and the compiler may run a k-step's LDS reads under the previous one's MFMAs: 446.5 -> 442 us)
        // fences between the k-steps: without the old per-tile branches a region is one basic block of 144 MFMAs, and the
        // scheduler hoists so many LDS reads that the kernel needs 256 registers and scratch
        kstep(stage, IC<0>{}, zm);
        __builtin_amdgcn_sched_barrier(0);
        kstep(stage, IC<1>{}, zm);
        __builtin_amdgcn_sched_barrier(0);
        kstep(stage, IC<2>{}, zm);
        __builtin_amdgcn_sched_barrier(0);
        kstep(stage, IC<3>{}, zm);
    };

    int region = xcd_slot(blockIdx.x, gridDim.x, a.xcd);
    if (region < nregions) issue(smem);
    auto regions = [&](auto zm) {
        for (; region < nregions; region += 2 * gridDim.x) {
            stage_body(IC<0>{}, region, zm);
            if (region + (int)gridDim.x < nregions) stage_body(IC<1>{}, region + gridDim.x, zm);
        }
    };
    // both copies run to the end of the kernel (a join after the loop made the compiler shuffle the 144 accumulator
    // registers through scratch)
    auto finish = [&]() {
    __syncthreads();

    // final transform dW = G^T M G: column pass (over p) in registers, row pass (over pg) through LDS.
    // M[.][3] and M[3][.] carry a folded minus sign.
    float *X = smem;   // X[pg][h][v][mt][r][lane], one nt at a time
    const float s3 = pg == 3 ? -1.f : 1.f;
#pragma unroll
    for (int nt = 0; nt < 3; ++nt) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float m0 = acc[0][mt][nt][r] * s3, m1 = acc[1][mt][nt][r] * s3, m2 = acc[2][mt][nt][r] * s3,
                            m3 = -acc[3][mt][nt][r] * s3;
                float *dst = X + (((pg * 2 + h) * 3 + 0) * (4 * MT) + mt * 4 + r) * 64 + lane;
                dst[0 * 4 * MT * 64] = m0 + 0.5f * (m1 + m2);
                dst[1 * 4 * MT * 64] = 0.5f * (m1 - m2);
                dst[2 * 4 * MT * 64] = 0.5f * (m1 + m2) + m3;
            }
        __syncthreads();
        // reader: wave (pg, h) handles items pg*3*MT .. of half h; item = (v, mt, r)
        for (int it = 0; it < 3 * MT; ++it) {
            const int item = pg * 3 * MT + it;
            const int v = item / (4 * MT), mt = (item % (4 * MT)) / 4, r = item % 4;
            float xg[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) xg[i] = X[(((i * 2 + h) * 3 + v) * (4 * MT) + mt * 4 + r) * 64 + lane];
            const int co = h * 48 + nt * 16 + l15, cl = mt * 16 + 4 * lq + r;
            if (co < a.Cout && cl < have) {
                float *o = a.partial + (((long)blockIdx.x * a.Cout + co) * a.CinTot + a.ci0 + c0 + cl) * 9 + v;
                o[0] = xg[0] + 0.5f * (xg[1] + xg[2]);
                o[3] = 0.5f * (xg[1] - xg[2]);
                o[6] = 0.5f * (xg[1] + xg[2]) + xg[3];
            }
        }
        __syncthreads();
    }
    };
    if (zmix) {
        regions(IC<1>{});
        finish();
    } else {
        regions(IC<0>{});
        finish();
    }
}

// gw[co][ci][tap] = sum over the parts that wrote column ci (launches differ in their K split): up to 4 column
// ranges [.., end[i]) with parts[i] partial results each, summed in a fixed order
struct WgRanges {
    int end[4], parts[4];
};
// 32 elements x 8 slices of the partial index per workgroup: slice j sums parts j, j + 8, ... (one chain per thread,
// 8x the loads in flight of a thread-per-element walk, which at ~1.5 workgroups per CU was latency-bound: 56 us for
// 43-64 MB), the slices are added in slice order through LDS — a fixed order, deterministic.
__global__ __launch_bounds__(256) void wino_wgrad_reduce_kernel(const float *__restrict__ partial, float *__restrict__ gw,
                                                                long n, int CinTot, WgRanges rg) {
    __shared__ float red[8][33];
    const int el = threadIdx.x & 31, sl = threadIdx.x >> 5;
    const long e = (long)blockIdx.x * 32 + el;
    float s = 0.f;
    if (e < n) {
        const int ci = (int)((e / 9) % CinTot);
        int parts = rg.parts[3];
#pragma unroll
        for (int i = 2; i >= 0; --i)
            if (ci < rg.end[i]) parts = rg.parts[i];
        float s0 = 0.f, s1 = 0.f;
        int p = sl;
        for (; p + 8 < parts; p += 16) {
            s0 += partial[(long)p * n + e];
            s1 += partial[(long)(p + 8) * n + e];
        }
        if (p < parts) s0 += partial[(long)p * n + e];
        s = s0 + s1;
    }
    red[sl][el] = s;
    __syncthreads();
    if (sl == 0 && e < n) {
        float t = red[0][el];
#pragma unroll
        for (int j = 1; j < 8; ++j) t += red[j][el];
        gw[e] = t;
    }
}

// channels of one source -> (groups of 48, one tail group of <= 16) or not representable
bool wg_split(int C, int *g48, int *tail) {
    *g48 = C / 48;
    *tail = C % 48;
    return *tail <= 16;
}

int nt_of(int Cout) { return Cout <= 48 ? 3 : 6; }

// workgroup geometry for an H x W image: 0 = 8 x 32 pixels, 1 = 16 x 16 pixels, -1 = neither divides it
int wino_square(int H, int W) {
    if (H % Geo<0>::TR == 0 && W % Geo<0>::TC == 0) return 0;
    if (H % Geo<1>::TR == 0 && W % Geo<1>::TC == 0) return 1;
    return -1;
}

}  // namespace

namespace sprk {

bool wino_eligible(const WinoGeom &g) {
    static const int on = getenv("SPRK_WINO") ? atoi(getenv("SPRK_WINO")) : 1;   // debug: 0 = direct kernels only
    if (!on) return false;
    if (g.KH != 3 || g.KW != 3 || g.stride != 1 || g.dil != 1 || g.up1 || g.res) return false;
    if (g.Hout != g.H || g.Wout != g.W) return false;
    const int sq = wino_square(g.H, g.W);
    if (sq < 0) return false;
    const int TR = sq ? Geo<1>::TR : Geo<0>::TR, TC = sq ? Geo<1>::TC : Geo<0>::TC;
    if (g.padL < 0 || g.padL > 4 || g.padT < 0) return false;
    if (g.C1 < 1 || g.C2 < 0 || g.Cout < 33) return false;
    const int NT = nt_of(g.Cout), groups = cdiv(g.Cout, NT * 16);
    if ((double)g.Cout / (groups * NT * 16) < 0.7) return false;   // padded output channels are wasted MFMAs
    // too few workgroups for 256 CUs: the direct kernel spreads the same layer over all of them, this one does 4/9 of
    // the MFMA work on as many CUs as it has 256-pixel tiles — break-even near 256 x 4/9 = 114 (measured: 96->96 at
    // 128x16x16, 128 tiles, 68 us direct)
    if (!g.pin && (long)g.N * (g.H / TR) * (g.W / TC) * groups < 128) return false;
    if ((long)4 * g.H * g.W * 4 >= 0x7FFFFFFFL) return false;               // 4 planes inside one buffer range
    return true;
}

size_t wino_ws_bytes(int C1, int C2, int Cout) {
    const int NT = nt_of(Cout), groups = cdiv(Cout, NT * 16), nch = cdiv(C1, CK) + cdiv(C2, CK);
    return (size_t)groups * nch * ufloats_of(NT) * sizeof(float);
}

int wino_conv(const WinoArgs &w, hipStream_t s) {
    const int NT = nt_of(w.Cout), groups = cdiv(w.Cout, NT * 16);
    KArgs a{};
    a.x = w.x; a.x2 = w.x2; a.U = w.U; a.bias = w.bias; a.scale = w.scale; a.shift = w.shift; a.y = w.y;
    a.N = w.N; a.C1 = w.C1; a.C2 = w.C2; a.H = w.H; a.W = w.W; a.Cout = w.Cout; a.padT = w.padT; a.padL = w.padL;
    a.act = w.act;
    a.up2 = w.up2;
    a.mask = w.up2 ? nullptr : w.mask;
    a.mact = w.mask_act;
    a.xcd = xcd_on();
    static const int diag = sprk::diag_env("SPRK_WINO_DIAG");   // timing experiments only
    a.diag = diag;
    const int sq = wino_square(w.H, w.W);
    a.tilesX = w.W / (sq ? Geo<1>::TC : Geo<0>::TC); a.tilesY = w.H / (sq ? Geo<1>::TR : Geo<0>::TR);
    a.ntiles = a.tilesX * a.tilesY * w.N;
    a.nc1 = cdiv(w.C1, CK);
    a.nch = a.nc1 + cdiv(w.C2, CK);
    if ((((uintptr_t)w.x | (uintptr_t)w.x2 | (uintptr_t)w.y | (uintptr_t)w.U) & 15) != 0) {
        set_error("wino_conv: tensors must be 16-byte aligned");
        return SPRK_EINVAL;
    }
    const long total = (long)groups * a.nch * NT * CK * 16;
    if (int rc = wprep_site(wprep_item(WPREP_WINO, w.w, w.U, total, {w.Cout, w.C1, w.C2, a.nc1, a.nch, NT, groups, w.mode}), s))
        return rc;
    // persistent workgroups, one per CU: each walks tiles blockIdx.x, + gridDim.x, ... and starts the next tile's
    // loads under its output transform (same-box A/B: +5..7 % on the 48-channel layers with their short K loops,
    // +1.5 % on the 96-channel ones).  SPRK_WINO_PERSIST=0: one workgroup per tile.
    static const int persist = getenv("SPRK_WINO_PERSIST") ? atoi(getenv("SPRK_WINO_PERSIST")) : 1;
    const dim3 grid(persist ? std::min(a.ntiles, std::max(1, num_cus() / groups)) : a.ntiles, groups);
    const size_t lds = lds_bytes_of(NT);
    auto launch = [&](auto kernel) {
        if (int rc = lds_optin(reinterpret_cast<const void *>(kernel), lds, "wino_conv")) return rc;
        hipLaunchKernelGGL(kernel, grid, dim3(kThreads), lds, s, a);
        return (int)SPRK_OK;
    };
#ifdef WINO_STAMP
    static unsigned *dbg = nullptr;
    static int dumped = 0;
    if (!dbg) { hipMalloc(&dbg, 4 * 8 * 16 * 4); hipMemset(dbg, 0, 4 * 8 * 16 * 4); }
    a.dbg = dbg;
#endif
    prof_begin(w.kclass, w.flops, s);
    if (int rc = NT == 6 ? (sq ? launch(wino_conv_kernel<6, 1>) : launch(wino_conv_kernel<6, 0>))
                         : (sq ? launch(wino_conv_kernel<3, 1>) : launch(wino_conv_kernel<3, 0>)))
        return rc;
    prof_end(w.kclass, s);
#ifdef WINO_STAMP
    if (NT == 6 && !sq && ++dumped == 5 && getenv("SPRK_WINO_STAMP")) {
        hipDeviceSynchronize();
        unsigned h[4 * 8 * 16];
        hipMemcpy(h, dbg, sizeof h, hipMemcpyDeviceToHost);
        FILE *f = fopen(getenv("SPRK_WINO_STAMP"), "a");
        for (int wg = 0; wg < 4; ++wg)
            for (int wv = 0; wv < 8; ++wv) {
                fprintf(f, "wg%d w%d", wg, wv);
                for (int k = 0; k < 16; ++k) fprintf(f, " %u", h[(wg * 8 + wv) * 16 + k]);
                fprintf(f, "\n");
            }
        fclose(f);
    }
#endif
    g_wino_launches.fetch_add(1, std::memory_order_relaxed);
    return SPRK_OK;
}

bool wino_wgrad_eligible(const WinoGeom &g) {
    static const int on = getenv("SPRK_WINO_WGRAD") ? atoi(getenv("SPRK_WINO_WGRAD")) : 1;   // debug: 0 = direct kernel only
    if (!on) return false;
    if (g.KH != 3 || g.KW != 3 || g.stride != 1 || g.dil != 1 || g.up1) return false;
    if (g.Hout != g.H || g.Wout != g.W || g.H % WRH || g.W % WRW) return false;
    if (g.padL < 0 || g.padL > 4 || g.padT < 0 || g.padT > 4) return false;   // padT <= 4: inner regions see no border
    if (g.Cout < 81 || g.Cout > 96) return false;   // two halves of 3 channel tiles
    int g48, tail;
    if (g.C1 < 1 || !wg_split(g.C1, &g48, &tail)) return false;
    if (g.C2 > 0 && !wg_split(g.C2, &g48, &tail)) return false;
    // The partial results (256 x Cout x Cin x 9 floats written, then reduced) and the per-workgroup final transform are
    // fixed costs.  Measured against the direct kernel (round 3, after the packed transforms; scratch/convbench.py):
    // 4096 regions (144->96 / 96->96 at 256 x 32^2): 589 -> 472 us, 365 -> 269 us; 2048 regions (96->96 at 32 x 64^2):
    // 191 -> 178 us; 1024 regions (16^2 planes at 256 images, 32^2 at 64): 114 -> 132, 178 -> 247, 164 -> 242 us: slower.
    // A one-channel tail group (the raw image concatenated into decode_block_1) is a launch of its own that costs a
    // third of a full group whatever the size: 96+1->96 at 32 x 64^2 196 -> 236 us, so layers with a tail need 8192.
    static const long min_regions = getenv("SPRK_WINO_WGRAD_MIN") ? atol(getenv("SPRK_WINO_WGRAD_MIN")) : 0;   // sweeps
    const bool has_tail = (g.C1 % 48) != 0 || (g.C2 % 48) != 0;
    const long need = min_regions > 0 ? min_regions : (has_tail ? 8192 : 2048);
    if ((long)g.N * (g.H / WRH) * (g.W / WRW) < need) return false;
    if ((long)48 * g.H * g.W * 4 >= 0x7FFFFFFFL || (long)g.Cout * g.H * g.W * 4 >= 0x7FFFFFFFL) return false;
    return true;
}

constexpr int kWgParts = 256;   // most partial results any launch leaves (one workgroup per CU)

size_t wino_wgrad_ws_bytes(int C1, int C2, int Cout) {
    return (size_t)kWgParts * Cout * (C1 + C2) * 9 * sizeof(float);
}

int wino_wgrad(const WinoWgArgs &w, hipStream_t s) {
    if ((((uintptr_t)w.x | (uintptr_t)w.x2 | (uintptr_t)w.gy | (uintptr_t)w.partial) & 15) != 0) {
        set_error("wino_wgrad: tensors must be 16-byte aligned");
        return SPRK_EINVAL;
    }
    const int CinTot = w.C1 + w.C2;
    WgRanges rg{};
    int nr = 0;
    auto launch = [&](const float *src, int Csrc, int ci0) -> int {
        int g48, tail;
        wg_split(Csrc, &g48, &tail);
        WgKArgs a{src, w.gy, w.partial, w.N, Csrc, 0, ci0, CinTot, w.H, w.W, w.Cout, w.padT, w.padL, w.W / WRW, w.H / WRH, xcd_on()};
        if (g48 > 0) {
            if (int rc = lds_optin(reinterpret_cast<const void *>(wino_wgrad_kernel<3>), kWgLdsBytes, "wino_wgrad")) return rc;
            const int parts = std::min(kWgParts, num_cus()) / g48;   // K split: one workgroup per CU over all channel groups
            prof_begin(w.kclass, w.flops * (48.0 * g48) / CinTot, s);
            hipLaunchKernelGGL(wino_wgrad_kernel<3>, dim3(parts, g48), dim3(kThreads), kWgLdsBytes, s, a);
            prof_end(w.kclass, s);
            if (int rc = check_launch("wino_wgrad<3>")) return rc;
            rg.end[nr] = ci0 + 48 * g48;
            rg.parts[nr++] = parts;
        }
        if (tail > 0) {
            if (int rc = lds_optin(reinterpret_cast<const void *>(wino_wgrad_kernel<1>), kWgLdsBytes, "wino_wgrad")) return rc;
            WgKArgs t = a;
            t.cbase = 48 * g48;   // the tail group sits after the 48-channel groups
            prof_begin(w.kclass, w.flops * (double)tail / CinTot, s);
            hipLaunchKernelGGL(wino_wgrad_kernel<1>, dim3(std::min(kWgParts, num_cus()), 1), dim3(kThreads), kWgLdsBytes, s, t);
            prof_end(w.kclass, s);
            if (int rc = check_launch("wino_wgrad<1>")) return rc;
            rg.end[nr] = ci0 + Csrc;
            rg.parts[nr++] = std::min(kWgParts, num_cus());
        }
        return SPRK_OK;
    };
    if (int rc = launch(w.x, w.C1, 0)) return rc;
    if (w.C2 > 0)
        if (int rc = launch(w.x2, w.C2, w.C1)) return rc;
    for (int i = nr; i < 4; ++i) rg.end[i] = CinTot, rg.parts[i] = rg.parts[nr - 1];
    const long n = (long)w.Cout * CinTot * 9;
    hipLaunchKernelGGL(wino_wgrad_reduce_kernel, dim3(cdiv(n, 32)), dim3(256), 0, s, w.partial, w.gw, n, CinTot, rg);
    if (int rc = check_launch("wino_wgrad_reduce")) return rc;
    g_wino_launches.fetch_add(1, std::memory_order_relaxed);
    return SPRK_OK;
}

}  // namespace sprk
