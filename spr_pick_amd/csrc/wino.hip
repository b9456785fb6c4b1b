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
#include <cstdlib>

#include "common.h"
#include "wino.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) void lds_void;
typedef __amdgpu_buffer_rsrc_t rsrc_t;
constexpr int kXZero = (int)0x80000000;   // beyond num_records: the DMA writes zeros
constexpr float kLeak = 0.1f;

__device__ __forceinline__ rsrc_t make_rsrc(const void *p) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, 0x7FFFFFFF, 0x00020000);
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

__device__ __forceinline__ float act_fn(float v, int act) {
    if (act == SPRK_ACT_LEAKY) return v > 0.f ? v : v * kLeak;
    if (act == SPRK_ACT_RELU) return v > 0.f ? v : 0.f;
    return v;
}

constexpr int CK = 4;                                  // channels per chunk = one MFMA k-step per position
constexpr int TR = 8, TC = 32;                         // output pixels per workgroup
constexpr int RP = 40, RPLANE = (TR + 2) * RP;         // raw tile: 10 rows x 40 columns, 16-byte column groups
constexpr int RAWF = 1792;                             // 4 planes = 400 DMA lanes -> 7 waves x 256 floats
constexpr int ETS = 68;                                // exchange stride per channel: 64 tiles + 4
constexpr int kThreads = 512;
constexpr int kEFloats = 4 * 2 * 32 * ETS;

// U of one chunk: [pos][nt][k][16]: the 64 lanes of a B read (k = lane / 16, channel = lane % 16) hit 64
// consecutive floats (no padding, no bank conflicts), and every (pos, nt) operand sits a multiple of 256 bytes
// from the lane's base address, which is what ds_read2st64_b32 encodes as an immediate
__host__ __device__ constexpr int ufloats_of(int NT) { return 16 * NT * CK * 16; }
__host__ __device__ constexpr size_t lds_bytes_of(int NT) {
    return (size_t)((2 * RAWF + 2 * ufloats_of(NT)) > kEFloats ? (2 * RAWF + 2 * ufloats_of(NT)) : kEFloats) * 4;
}

// U[group][chunk][pos][nt][k][16] = (G g G^T)[pos] of the tap matrix of (output channel (group*NT + nt)*16 + j,
// GEMM-k channel of chunk/k); zero rows for the channels a source's last chunk does not have, zero columns
// past Cout.
//   mode 0: g = w[cout][cin][u][v]           (forward)
//   mode 1: g = w[k][n][2-u][2-v]            (backward-data: k = forward output channel, n = forward input)
__global__ void wino_weights_kernel(const float *__restrict__ w, float *__restrict__ U, int Cout, int C1, int C2,
                                    int nc1, int nch, int NT, int groups, int mode) {
    const long total = (long)groups * nch * NT * CK * 16;
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= total) return;
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
    float *dst = U + ((long)grp * nch + c) * (16 * NT * CK * 16) + (nt * CK + k) * 16 + j;
    for (int i = 0; i < 4; ++i) {
        const float u4[4] = {t4[i][0], 0.5f * (t4[i][0] + t4[i][1] + t4[i][2]), 0.5f * (t4[i][0] - t4[i][1] + t4[i][2]),
                             t4[i][2]};
        for (int q = 0; q < 4; ++q) dst[(long)(i * 4 + q) * NT * CK * 16] = u4[q];
    }
}

template <int V>
struct IC {
    static constexpr int value = V;
};

struct KArgs {
    const float *x, *x2, *U, *bias, *scale, *shift;
    float *y;
    int N, C1, C2, H, W, Cout, padT, padL, act, tilesX, tilesY, nc1, nch;
};

template <int NT>
__global__ __launch_bounds__(kThreads, 1) void wino_conv_kernel(const KArgs a) {
    constexpr int UFLOATS = ufloats_of(NT);
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *Ub = smem;                  // 2 x UFLOATS
    float *Rb = smem + 2 * UFLOATS;    // 2 x RAWF
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), l15 = lane & 15,
              lq = lane >> 4;
    const int pg = wave & 3, th = wave >> 2;
    int b = blockIdx.x;
    const int bx = b % a.tilesX;
    b /= a.tilesX;
    const int by = b % a.tilesY, n = b / a.tilesY;
    const int grp = blockIdx.y;
    const long HW = (long)a.H * a.W;

    // raw-tile DMA: lane q moves 16 bytes of plane q / 100, row (q % 100) / 10, column group q % 10
    int voff = kXZero, dch = 4;
    if (tid < CK * RPLANE / 4) {
        dch = tid / 100;
        const int rem = tid % 100, r = rem / 10, c4 = rem % 10;
        const int gy = by * TR - a.padT + r, gx = bx * TC - 4 + 4 * c4;
        if (gy >= 0 && gy < a.H && gx >= 0 && gx + 3 < a.W) voff = (int)(((long)dch * a.H + gy) * a.W + gx) * 4;
    }
    auto issue_raw = [&](int c, float *dst) {
        const bool s1 = c < a.nc1;
        const int c0 = (s1 ? c : c - a.nc1) * CK, have = (s1 ? a.C1 : a.C2) - c0;
        const float *src = s1 ? a.x + ((long)n * a.C1 + c0) * HW : a.x2 + ((long)n * a.C2 + c0) * HW;
        if (tid < CK * RPLANE / 4) bdma16(make_rsrc(src), dch < have ? voff : kXZero, 0, dst + wave * 256);
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
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[p][mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // row pg of B^T d = d[ra] + sgn * d[rb]:  (0,2,-), (1,2,+), (2,1,-), (1,3,-)
    const int ra = pg == 0 ? 0 : pg == 2 ? 2 : 1, rb = pg == 3 ? 3 : pg == 2 ? 1 : 2;
    const float sgn = pg == 1 ? 1.f : -1.f;
    // one VGPR base per operand stream and stage, kept opaque so that every other offset is an immediate of
    // ds_read2_b32 / ds_read2st64_b32 (no address arithmetic in the K loop)
    int rawA[2], rawB[2];
    rawA[0] = lds_addr(Rb + lq * RPLANE + (4 * th + ra) * RP + (4 - a.padL) + 2 * l15);
    rawB[0] = lds_addr(Rb + lq * RPLANE + (4 * th + rb) * RP + (4 - a.padL) + 2 * l15);
    rawA[1] = rawA[0] + RAWF * 4;
    rawB[1] = rawB[0] + RAWF * 4;
    int bbase = lds_addr(Ub + (4 * pg) * NT * 64 + lane);
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
                xv[j] = __builtin_fmaf(pb[mt * 2 * RP + j], sgn, pa[mt * 2 * RP + j]);
            av[mt][0] = xv[0] - xv[2];
            av[mt][1] = xv[1] + xv[2];
            av[mt][2] = xv[2] - xv[1];
            av[mt][3] = xv[1] - xv[3];
        }
    };
    auto mma = [&](const float (&av)[2][4], auto stage) {
        constexpr int S = decltype(stage)::value;
        const lds_cfp bp = lds_f(bbase);
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            float bv[NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) bv[nt] = bp[S * UFLOATS + (p * NT + nt) * 64];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    acc[p][mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[mt][p], bv[nt], acc[p][mt][nt], 0, 0, 0);
        }
    };

    const int nch = a.nch;
    issue_raw(0, Rb);
    issue_u(0, Ub);
    if (nch > 1) issue_raw(1, Rb + RAWF);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    float aop[2][2][4];
    transform(IC<0>{}, aop[0]);
    __syncthreads();
    // iteration c: aop[c&1] = operands of chunk c, U[c&1] holds chunk c, raw[(c+1)&1] the raw tile of chunk c+1
    auto iter = [&](auto par, int c) {
        constexpr int P = decltype(par)::value;
        if (c + 1 < nch) issue_u(c + 1, Ub + (1 - P) * UFLOATS);
        if (c + 2 < nch) issue_raw(c + 2, Rb + P * RAWF);
        transform(IC<1 - P>{}, aop[1 - P]);   // past the last chunk this reads a stale tile: never used
        mma(aop[P], IC<P>{});
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    };
    for (int c = 0; c < nch; c += 2) {
        iter(IC<0>{}, c);
        if (c + 1 < nch) iter(IC<1>{}, c + 1);
    }

    // output transform.  E[pg][b][channel'][tile'] with tile' = lq + 4 r + 16 mt + 32 th; 32 channels a pass.
    float *E = smem;
    const int rl_tx = lane & 15, rl_ty = lane >> 4;
    const int tprime = (rl_tx >> 2) + 4 * (rl_tx & 3) + 16 * rl_ty;
    const int oy = by * TR + 2 * rl_ty, ox = bx * TC + 2 * rl_tx;
    constexpr int PASSES = (NT + 1) / 2;
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
                    float *e = E + ((pg * 2) * 32 + n2 * 16 + l15) * ETS + lq + 16 * mt + 32 * th;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        e[4 * r] = s0[r];
                        e[32 * ETS + 4 * r] = s1[r];
                    }
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int cp = wave + 8 * it;   // channel within the pass
            const int co = grp * (NT * 16) + pass * 32 + cp;
            if (pass * 32 + cp < NT * 16 && co < a.Cout) {
                f32x2 s[4];
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    s[w][0] = E[((w * 2 + 0) * 32 + cp) * ETS + tprime];
                    s[w][1] = E[((w * 2 + 1) * 32 + cp) * ETS + tprime];
                }
                float sc = 1.f, sh = 0.f;
                if (a.scale) {
                    sc = a.scale[co];
                    sh = a.shift[co];
                } else if (a.bias) {
                    sh = a.bias[co];
                }
                f32x2 o0 = ((s[0] + s[1]) + s[2]) * sc + sh;
                f32x2 o1 = ((s[1] - s[2]) - s[3]) * sc + sh;
                if (a.act == SPRK_ACT_LEAKY) {   // max(v, 0.1 v) = v > 0 ? v : 0.1 v
                    o0 = __builtin_elementwise_max(o0, o0 * kLeak);
                    o1 = __builtin_elementwise_max(o1, o1 * kLeak);
                } else if (a.act == SPRK_ACT_RELU) {
                    o0 = __builtin_elementwise_max(o0, (f32x2){0.f, 0.f});
                    o1 = __builtin_elementwise_max(o1, (f32x2){0.f, 0.f});
                }
                float *yp = a.y + (((long)n * a.Cout + co) * a.H + oy) * a.W + ox;
                *reinterpret_cast<f32x2 *>(yp) = o0;
                *reinterpret_cast<f32x2 *>(yp + a.W) = o1;
            }
        }
        __syncthreads();
    }
}

int nt_of(int Cout) { return Cout <= 48 ? 3 : 6; }

}  // namespace

namespace sprk {

bool wino_eligible(const WinoGeom &g) {
    static const int on = getenv("SPRK_WINO") ? atoi(getenv("SPRK_WINO")) : 1;   // debug: 0 = direct kernels only
    if (!on) return false;
    if (g.KH != 3 || g.KW != 3 || g.stride != 1 || g.dil != 1 || g.up1 || g.up2 || g.res) return false;
    if (g.Hout != g.H || g.Wout != g.W || g.H % TR || g.W % TC) return false;
    if (g.padL < 0 || g.padL > 4 || g.padT < 0) return false;
    if (g.C1 < 1 || g.C2 < 0 || g.Cout < 33) return false;
    const int NT = nt_of(g.Cout), groups = cdiv(g.Cout, NT * 16);
    if ((double)g.Cout / (groups * NT * 16) < 0.7) return false;   // padded output channels are wasted MFMAs
    if ((long)g.N * (g.H / TR) * (g.W / TC) * groups < 192) return false;   // too few workgroups for 256 CUs
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
    a.tilesX = w.W / TC; a.tilesY = w.H / TR;
    a.nc1 = cdiv(w.C1, CK);
    a.nch = a.nc1 + cdiv(w.C2, CK);
    if ((((uintptr_t)w.x | (uintptr_t)w.x2 | (uintptr_t)w.y | (uintptr_t)w.U) & 15) != 0) {
        set_error("wino_conv: tensors must be 16-byte aligned");
        return SPRK_EINVAL;
    }
    const long total = (long)groups * a.nch * NT * CK * 16;
    hipLaunchKernelGGL(wino_weights_kernel, dim3(cdiv(total, 256)), dim3(256), 0, s, w.w, w.U, w.Cout, w.C1, w.C2, a.nc1,
                       a.nch, NT, groups, w.mode);
    if (int rc = check_launch("wino_weights")) return rc;
    const dim3 grid(a.tilesX * a.tilesY * w.N, groups);
    const size_t lds = lds_bytes_of(NT);
    auto launch = [&](auto kernel) {
        static bool attr_done = false;   // per instantiation
        if (!attr_done) {
            if (hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)lds) != hipSuccess) {
                set_error("wino_conv: cannot reserve %zu bytes of LDS", lds);
                return (int)SPRK_ELAUNCH;
            }
            attr_done = true;
        }
        hipLaunchKernelGGL(kernel, grid, dim3(kThreads), lds, s, a);
        return (int)SPRK_OK;
    };
    prof_begin(w.kclass, w.flops, s);
    if (int rc = NT == 6 ? launch(wino_conv_kernel<6>) : launch(wino_conv_kernel<3>)) return rc;
    prof_end(w.kclass, s);
    g_wino_launches.fetch_add(1, std::memory_order_relaxed);
    return SPRK_OK;
}

}  // namespace sprk
