// Prototype: Winograd F(2x2, 3x3) forward convolution on fp32 MFMA for the 96-output-channel 3x3 layers
// (stride 1, zero padding given as top/left pad; output size = input size).  Stand-alone: builds its own
// reference with a naive direct kernel, checks, and times.  Not part of libsprk.so.
//   hipcc -O3 --offload-arch=gfx950 -o scratch/bin/wino_proto scratch/wino_proto.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void;
typedef __amdgpu_buffer_rsrc_t rsrc_t;
constexpr int kXZero = (int)0x80000000;

__device__ __forceinline__ rsrc_t make_rsrc(const void *p) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, 0x7FFFFFFF, 0x00020000);
}
__device__ __forceinline__ void bdma16(rsrc_t r, int voff, int soff, float *lds_wave_base) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void *)lds_wave_base, 16, voff, soff, 0, 0);
}

constexpr int CK = 4;            // channels per K-chunk = one MFMA k-step per transform position
constexpr int NT = 6;            // 16-wide output-channel tiles: 96 channels
constexpr int LDW = 112;         // U row stride (floats)
constexpr int TR = 8, TC = 32;   // output pixels per workgroup: 4 x 16 Winograd tiles
constexpr int RR = TR + 2, RP = 40, RPLANE = RR * RP;   // raw input tile: 10 rows x 40 columns (16-byte columns)
constexpr int RAWF = 1792;                              // 4 planes = 1600 floats = 400 DMA lanes -> 7 waves
constexpr int UFLOATS = 16 * CK * LDW;                  // 7168
constexpr int ETS = 68;                                 // epilogue exchange: floats per (wave row, b, channel)
constexpr int kThreads = 512;
constexpr int kEFloats = 4 * 2 * 32 * ETS;
constexpr int kStageFloats = 2 * RAWF + 2 * UFLOATS;
constexpr size_t kLdsBytes = (size_t)(kStageFloats > kEFloats ? kStageFloats : kEFloats) * 4;

// U[chunk][pos][k][LDW] = (G g G^T)[pos] of weight (cout, cin = 4 chunk + k); zero for cout >= Cout
__global__ void wino_weights(const float *__restrict__ w, float *__restrict__ U, int Cout, int Cin) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= Cin * LDW) return;
    const int cin = e / LDW, co = e % LDW;
    float g[3][3] = {};
    if (co < Cout)
        for (int i = 0; i < 9; ++i) g[i / 3][i % 3] = w[((long)co * Cin + cin) * 9 + i];
    float t[4][3];
    for (int j = 0; j < 3; ++j) {
        t[0][j] = g[0][j];
        t[1][j] = 0.5f * (g[0][j] + g[1][j] + g[2][j]);
        t[2][j] = 0.5f * (g[0][j] - g[1][j] + g[2][j]);
        t[3][j] = g[2][j];
    }
    for (int i = 0; i < 4; ++i) {
        const float u[4] = {t[i][0], 0.5f * (t[i][0] + t[i][1] + t[i][2]), 0.5f * (t[i][0] - t[i][1] + t[i][2]), t[i][2]};
        for (int j = 0; j < 4; ++j)
            U[(((long)(cin / CK) * 16 + (i * 4 + j)) * CK + (cin % CK)) * LDW + co] = u[j];
    }
}

template <int V>
struct IC {
    static constexpr int value = V;
};

__global__ __launch_bounds__(kThreads, 1) void wino_fwd(const float *__restrict__ x, const float *__restrict__ Ug,
                                                        const float *__restrict__ bias, float *__restrict__ y, int N,
                                                        int C, int H, int W, int Cout, int pt, int pl, int tilesX,
                                                        int tilesY, float slope) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *Ub = smem;                       // 2 x UFLOATS
    float *Rb = smem + 2 * UFLOATS;         // 2 x RAWF
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), l15 = lane & 15,
              lq = lane >> 4;
    const int pg = wave & 3, th = wave >> 2;   // transform-position row of this wave, tile half
    int b = blockIdx.x;
    const int bx = b % tilesX;
    b /= tilesX;
    const int by = b % tilesY, n = b / tilesY;
    const long HW = (long)H * W;

    // raw-tile DMA: lane q moves 16 bytes of plane q / 100, row (q % 100) / 10, column group q % 10
    int voff = kXZero;
    if (tid < CK * RPLANE / 4) {
        const int ch = tid / 100, rem = tid % 100, r = rem / 10, c4 = rem % 10;
        const int gy = by * TR - pt + r, gx = bx * TC - 4 + 4 * c4;
        if (gy >= 0 && gy < H && gx >= 0 && gx + 3 < W) voff = (int)(((long)ch * H + gy) * W + gx) * 4;
    }
    auto issue_raw = [&](int c, float *dst) {
        const rsrc_t xr = make_rsrc(x + ((long)n * C + c * CK) * HW);
        if (tid < CK * RPLANE / 4) bdma16(xr, voff, 0, dst + wave * 256);
    };
    auto issue_u = [&](int c, float *dst) {
        const rsrc_t ur = make_rsrc(Ug + (long)c * UFLOATS);
#pragma unroll
        for (int gi = 0; gi < 3; ++gi) bdma16(ur, tid * 16, gi * 8192, dst + gi * 2048 + wave * 256);
        if (wave < 4) bdma16(ur, tid * 16, 3 * 8192, dst + 3 * 2048 + wave * 256);
    };

    f32x4 acc[4][2][NT];
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[p][mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // This wave's A operands are row pg of B^T d B of its own 32 tiles: lane (l15, lq) transforms tiles
    // 32 th + 16 mt + l15 of channel lq straight from the raw tile, no LDS round trip.
    // row pg of B^T d = d[ra] + sgn * d[rb]:  (0,2,-), (1,2,+), (2,1,-), (1,3,-)
    const int ra = pg == 0 ? 0 : pg == 2 ? 2 : 1, rb = pg == 3 ? 3 : pg == 2 ? 1 : 2;
    const float sgn = pg == 1 ? 1.f : -1.f;
    const int rawA = lq * RPLANE + (4 * th + ra) * RP + (4 - pl) + 2 * l15;
    const int rawB = lq * RPLANE + (4 * th + rb) * RP + (4 - pl) + 2 * l15;
    const int boff = ((4 * pg) * CK + lq) * LDW + l15;

    auto load_raw = [&](const float *raw, float (&da)[2][4], float (&db)[2][4]) {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                da[mt][j] = raw[rawA + mt * 2 * RP + j];
                db[mt][j] = raw[rawB + mt * 2 * RP + j];
            }
    };
    auto transform = [&](const float (&da)[2][4], const float (&db)[2][4], float (&av)[2][4]) {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            float xv[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) xv[j] = __builtin_fmaf(db[mt][j], sgn, da[mt][j]);
            av[mt][0] = xv[0] - xv[2];
            av[mt][1] = xv[1] + xv[2];
            av[mt][2] = xv[2] - xv[1];
            av[mt][3] = xv[1] - xv[3];
        }
    };
    auto mma = [&](const float (&av)[2][4], const float *U, int p0, int p1) {
        const float *bp = U + boff;
#pragma unroll
        for (int p = p0; p < p1; ++p) {
            float bv[NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) bv[nt] = bp[p * CK * LDW + 16 * nt];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    acc[p][mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[mt][p], bv[nt], acc[p][mt][nt], 0, 0, 0);
        }
    };

    const int nch = C / CK;
    issue_raw(0, Rb);
    issue_u(0, Ub);
    if (nch > 1) issue_raw(1, Rb + RAWF);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    float aop[2][2][4];
    {
        float da[2][4], db[2][4];
        load_raw(Rb, da, db);
        transform(da, db, aop[0]);
    }
    __syncthreads();
    // iteration c: aop[c&1] = operands of chunk c, U[c&1] holds chunk c, raw[(c+1)&1] the raw tile of chunk c+1
    auto iter = [&](auto par, int c) {
        constexpr int P = decltype(par)::value;
        if (c + 1 < nch) issue_u(c + 1, Ub + (1 - P) * UFLOATS);
        if (c + 2 < nch) issue_raw(c + 2, Rb + P * RAWF);
        float da[2][4], db[2][4];
        load_raw(Rb + (1 - P) * RAWF, da, db);   // past the last chunk this reads a stale tile: never used
        __builtin_amdgcn_sched_barrier(0);
        mma(aop[P], Ub + P * UFLOATS, 0, 2);
        __builtin_amdgcn_sched_barrier(0);
        transform(da, db, aop[1 - P]);
        __builtin_amdgcn_sched_barrier(0);
        mma(aop[P], Ub + P * UFLOATS, 2, 4);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    };
    for (int c = 0; c < nch; c += 2) {
        iter(IC<0>{}, c);
        if (c + 1 < nch) iter(IC<1>{}, c + 1);
    }

    // output transform: columns in registers (this wave holds row pg of the 4x4 product), rows across the
    // waves through LDS.  E[pg][b][channel'][tile'] with tile' = lq + 4 r + 16 mt + 32 th.
    float *E = smem;
    const int rl_tx = lane & 15, rl_ty = lane >> 4;
    const int tprime = (rl_tx >> 2) + 4 * (rl_tx & 3) + 16 * rl_ty;
#pragma unroll
    for (int pass = 0; pass < 3; ++pass) {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int n2 = 0; n2 < 2; ++n2) {
                const int nt = pass * 2 + n2;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float m0 = acc[0][mt][nt][r], m1 = acc[1][mt][nt][r], m2 = acc[2][mt][nt][r],
                                m3 = acc[3][mt][nt][r];
                    float *e = E + ((pg * 2) * 32 + n2 * 16 + l15) * ETS + lq + 4 * r + 16 * mt + 32 * th;
                    e[0] = m0 + m1 + m2;
                    e[32 * ETS] = m1 - m2 - m3;
                }
            }
        __syncthreads();
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int cp = wave + 8 * it;   // channel within the pass
            const int co = pass * 32 + cp;
            float s[4][2];
#pragma unroll
            for (int w = 0; w < 4; ++w)
#pragma unroll
                for (int bb = 0; bb < 2; ++bb) s[w][bb] = E[((w * 2 + bb) * 32 + cp) * ETS + tprime];
            if (co < Cout) {
                const float bs = bias ? bias[co] : 0.f;
                float2 o0 = make_float2(s[0][0] + s[1][0] + s[2][0] + bs, s[0][1] + s[1][1] + s[2][1] + bs);
                float2 o1 = make_float2(s[1][0] - s[2][0] - s[3][0] + bs, s[1][1] - s[2][1] - s[3][1] + bs);
                o0.x = o0.x > 0.f ? o0.x : o0.x * slope;
                o0.y = o0.y > 0.f ? o0.y : o0.y * slope;
                o1.x = o1.x > 0.f ? o1.x : o1.x * slope;
                o1.y = o1.y > 0.f ? o1.y : o1.y * slope;
                const int oy = by * TR + 2 * rl_ty, ox = bx * TC + 2 * rl_tx;
                float *yp = y + (((long)n * Cout + co) * H + oy) * W + ox;
                *reinterpret_cast<float2 *>(yp) = o0;
                *reinterpret_cast<float2 *>(yp + W) = o1;
            }
        }
        __syncthreads();
    }
}

__global__ void naive_conv(const float *__restrict__ x, const float *__restrict__ w, const float *__restrict__ bias,
                           float *__restrict__ y, int N, int C, int H, int W, int Cout, int pt, int pl, float slope) {
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (long)N * Cout * H * W) return;
    const int ox = e % W, oy = (e / W) % H, co = (e / ((long)W * H)) % Cout, n = e / ((long)W * H * Cout);
    double s = bias ? bias[co] : 0.f;
    for (int c = 0; c < C; ++c)
        for (int u = 0; u < 3; ++u)
            for (int v = 0; v < 3; ++v) {
                const int iy = oy - pt + u, ix = ox - pl + v;
                if (iy < 0 || iy >= H || ix < 0 || ix >= W) continue;
                s += (double)x[(((long)n * C + c) * H + iy) * W + ix] * w[((long)co * C + c) * 9 + u * 3 + v];
            }
    const float f = (float)s;
    y[e] = f > 0.f ? f : f * slope;
}

#define CHECK(e)                                                                     \
    do {                                                                             \
        hipError_t _e = (e);                                                         \
        if (_e != hipSuccess) {                                                      \
            printf("HIP error %s at line %d\n", hipGetErrorString(_e), __LINE__);    \
            return 1;                                                                \
        }                                                                            \
    } while (0)

int main(int argc, char **argv) {
    const int N = argc > 1 ? atoi(argv[1]) : 2, C = argc > 2 ? atoi(argv[2]) : 96, H = argc > 3 ? atoi(argv[3]) : 128,
              W = argc > 4 ? atoi(argv[4]) : 128, check = argc > 5 ? atoi(argv[5]) : 1;
    const int Cout = 96, pt = 2, pl = 1;
    if (C % CK || H % TR || W % TC) {
        printf("bad shape\n");
        return 1;
    }
    const size_t nx = (size_t)N * C * H * W, ny = (size_t)N * Cout * H * W, nw = (size_t)Cout * C * 9;
    std::vector<float> hx(nx), hw(nw), hb(Cout);
    srand(1);
    for (auto &v : hx) v = (rand() / (float)RAND_MAX) * 2.f - 1.f;
    for (auto &v : hw) v = ((rand() / (float)RAND_MAX) * 2.f - 1.f) * 0.05f;
    for (auto &v : hb) v = (rand() / (float)RAND_MAX) - 0.5f;
    float *dx, *dw, *db, *dy, *dr, *dU;
    CHECK(hipMalloc(&dx, nx * 4));
    CHECK(hipMalloc(&dw, nw * 4));
    CHECK(hipMalloc(&db, Cout * 4));
    CHECK(hipMalloc(&dy, ny * 4));
    CHECK(hipMalloc(&dr, ny * 4));
    CHECK(hipMalloc(&dU, (size_t)(C / CK) * UFLOATS * 4));
    CHECK(hipMemcpy(dx, hx.data(), nx * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(dw, hw.data(), nw * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(db, hb.data(), Cout * 4, hipMemcpyHostToDevice));
    CHECK(hipMemset(dy, 0, ny * 4));
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(wino_fwd), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)kLdsBytes));
    const int tilesX = W / TC, tilesY = H / TR;
    const dim3 grid(tilesX * tilesY * N);
    auto run = [&]() {
        hipLaunchKernelGGL(wino_weights, dim3((C * LDW + 255) / 256), dim3(256), 0, 0, dw, dU, Cout, C);
        hipLaunchKernelGGL(wino_fwd, grid, dim3(kThreads), kLdsBytes, 0, dx, dU, db, dy, N, C, H, W, Cout, pt, pl, tilesX,
                           tilesY, 0.1f);
    };
    run();
    CHECK(hipDeviceSynchronize());
    CHECK(hipGetLastError());
    if (check) {
        hipLaunchKernelGGL(naive_conv, dim3((ny + 255) / 256), dim3(256), 0, 0, dx, dw, db, dr, N, C, H, W, Cout, pt, pl,
                           0.1f);
        CHECK(hipDeviceSynchronize());
        std::vector<float> hy(ny), hr(ny);
        CHECK(hipMemcpy(hy.data(), dy, ny * 4, hipMemcpyDeviceToHost));
        CHECK(hipMemcpy(hr.data(), dr, ny * 4, hipMemcpyDeviceToHost));
        double maxerr = 0, maxref = 0;
        size_t worst = 0;
        for (size_t i = 0; i < ny; ++i) {
            const double e = fabs((double)hy[i] - hr[i]);
            if (e > maxerr) maxerr = e, worst = i;
            maxref = fmax(maxref, fabs((double)hr[i]));
        }
        printf("check N=%d C=%d %dx%d: max |err| %.3e (max |ref| %.3e) at %zu: got %f want %f\n", N, C, H, W, maxerr,
               maxref, worst, hy[worst], hr[worst]);
    }
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    const int reps = 10;
    hipLaunchKernelGGL(wino_fwd, grid, dim3(kThreads), kLdsBytes, 0, dx, dU, db, dy, N, C, H, W, Cout, pt, pl, tilesX, tilesY,
                       0.1f);
    CHECK(hipEventRecord(e0, 0));
    for (int i = 0; i < reps; ++i)
        hipLaunchKernelGGL(wino_fwd, grid, dim3(kThreads), kLdsBytes, 0, dx, dU, db, dy, N, C, H, W, Cout, pt, pl, tilesX,
                           tilesY, 0.1f);
    CHECK(hipEventRecord(e1, 0));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    ms /= reps;
    const double flops = 2.0 * N * H * W * (double)Cout * C * 9;
    printf("wino_fwd N=%d C=%d %dx%d: %.1f us, %.1f TFLOP/s direct-equivalent (%.1f TFLOP/s on the MFMA pipes)\n", N, C, H,
           W, ms * 1e3, flops / ms / 1e9, flops / 2.25 / ms / 1e9);
    return 0;
}
