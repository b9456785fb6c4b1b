// Prototype: Winograd F(2x2, 3x3) backward-weight on fp32 MFMA.
//   dW = G^T [ sum over images and 2x2 tiles of (A dY A^T) .* (B^T d B) ] G
// 16 independent [Cin x tiles].[tiles x Cout] GEMMs (one per position of the 4x4 transform domain); both
// operands are transformed in registers from raw LDS tiles, k = 4 tiles per MFMA step.
// Stand-alone: own naive reference, check and timing.  Not part of libsprk.so.
//   hipcc -O3 --offload-arch=gfx950 -o scratch/bin/wino_wgrad_proto scratch/wino_wgrad_proto.hip
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

typedef const __attribute__((address_space(3))) float *lds_cfp;
__device__ __forceinline__ int lds_addr(const void *p) {
    return (int)(unsigned)(__SIZE_TYPE__)(const __attribute__((address_space(3))) char *)p;
}
__device__ __forceinline__ lds_cfp lds_f(int byte_addr) { return (lds_cfp)(__SIZE_TYPE__)(unsigned)byte_addr; }

constexpr int CIG = 48, COG = 96;        // input channels per workgroup (grid.y groups), output channels
constexpr int RH = 4, RW = 16;           // region: output pixels per stage (2 x 8 tiles of one image)
constexpr int RP = 24;                   // raw row pitch (floats): columns x0-4 .. x0+19
// LDS layouts, chosen so that every operand of a k-step is within ds_read2_b32's 1020-byte immediate range of
// one base register per (operand row, stage):
//   raw d : [l15 (16)][row (6)][mt (3)][24 floats]  + 4 pad  -> channel 16 mt + l15; slot stride 436 = 4 * odd
//   dY    : [h (2)][l15 (16)][row (4)][nt (3)][16 floats] + 4 pad -> channel 48 h + 16 nt + l15; stride 196
constexpr int PA = 6 * 3 * RP + 4, PB = 4 * 3 * 16 + 4;
constexpr int NA4 = 16 * (PA / 4), NB4 = 32 * (PB / 4);   // DMA lanes per stage: 1744, 1568
constexpr int AF = 4 * 2048, BF = 4 * 2048;                 // floats reserved per stage and operand (4 sweeps)
constexpr int STAGE = AF + BF;
constexpr int kThreads = 512;
constexpr size_t kLdsBytes = (size_t)(2 * STAGE + 8 * kThreads) * 4;   // two stages + the DMA lane table

template <int V>
struct IC {
    static constexpr int value = V;
};

// partial[part][cout][cin][9]
__global__ __launch_bounds__(kThreads, 1) void wino_wgrad(const float *__restrict__ x, const float *__restrict__ gy,
                                                          float *__restrict__ partial, int N, int C, int H, int W,
                                                          int Cout, int pt, int pl, int regionsX, int regionsY) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), l15 = lane & 15,
              lq = lane >> 4;
    const int pg = wave & 3, h = wave >> 2;   // row of the 4x4 transform domain, half of the output channels
    const int grp = blockIdx.y;               // group of 48 input channels
    const long HW = (long)H * W;
    const int nregions = N * regionsY * regionsX;

    // DMA lane geometry (fixed): raw plane lanes and dY plane lanes of the 4 sweeps; rc packs the raw row
    // (pad lanes: a row that is never inside the image) and the column offset of the lane's 4 floats
    // byte offsets are multiples of 16: the raw row (7 = pad lane, never inside the image) rides in the low bits
    // of aoffl, the raw column group in the low bits of boffl
    // They live in LDS (8 words per thread): registers are the scarce resource of this kernel.
    int *geo = reinterpret_cast<int *>(smem + 2 * STAGE) + tid;
#pragma unroll
    for (int rd = 0; rd < 4; ++rd) {
        const int q = tid + rd * kThreads;
        const int sl = q / (PA / 4), rem = q % (PA / 4);          // slot l15, position inside
        const int row = rem / 18, mt = (rem % 18) / 6, c4 = rem % 6;
        const bool va = q < NA4 && rem < 108;
        geo[(2 * rd) * kThreads] = (int)(((16 * mt + sl) * HW + (long)row * W + 4 * c4) * 4) | (va ? row : 15);
        const int sb = q / (PB / 4), remb = q % (PB / 4);         // slot (h, l15)
        const int rowb = remb / 12, ntb = (remb % 12) / 4, c4b = remb % 4;
        const int chb = (sb >> 4) * 48 + 16 * ntb + (sb & 15);
        geo[(2 * rd + 1) * kThreads] = ((q < NB4 && remb < 48) ? (int)((chb * HW + (long)rowb * W + 4 * c4b) * 4) : kXZero) | c4;
    }
    auto issue = [&](int region, float *st) {
        int r = region;
        const int rx = r % regionsX;
        r /= regionsX;
        const int ry = r % regionsY, n = r / regionsY;
        const int y0 = ry * RH, x0 = rx * RW;
        const float *xa = x + ((long)n * C + grp * CIG) * HW + (long)(y0 - pt) * W + (x0 - 4);
        const float *gb = gy + (long)n * Cout * HW + (long)y0 * W + x0;
        const rsrc_t ra = make_rsrc(xa), rb = make_rsrc(gb);
#pragma unroll
        for (int rd = 0; rd < 4; ++rd) {
            const int pa_ = geo[(2 * rd) * kThreads], pb_ = geo[(2 * rd + 1) * kThreads];
            // row 15 (pad lane) lands far outside the image for any y0
            const unsigned gyy = (unsigned)(y0 - pt + ((pa_ & 15) == 15 ? 0x100000 : (pa_ & 15)));
            const unsigned gxx = (unsigned)(x0 - 4 + 4 * (pb_ & 15));
            const bool ok = gyy < (unsigned)H && gxx < (unsigned)W;
            bdma16(ra, ok ? (pa_ & ~15) : kXZero, 0, st + rd * 2048 + wave * 256);
            bdma16(rb, pb_ & ~15, 0, st + AF + rd * 2048 + wave * 256);
        }
    };

    f32x4 acc[4][3][3];
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int mt = 0; mt < 3; ++mt)
#pragma unroll
            for (int nt = 0; nt < 3; ++nt) acc[p][mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // A operand: lane (l15 = input channel in tile mt, lq = tile of the k-step): row pg of B^T d = d[ra] + sgn d[rb]
    const int ra_ = pg == 0 ? 0 : pg == 2 ? 2 : 1, rb_ = pg == 3 ? 3 : pg == 2 ? 1 : 2;
    const float sgn = pg == 1 ? 1.f : -1.f;
    // B operand: lane (lq = tile, l15 = output channel in tile nt): row pg of A dY = dY[0], dY[0]+dY[1],
    // dY[0]-dY[1], (-)dY[1]; the minus signs of row 3 and of column 3 are folded into the final transform
    const float zs = pg == 1 ? 1.f : -1.f;
    const int zrow = pg == 3 ? 1 : 0;
    const bool zmix = pg == 1 || pg == 2;
    // one VGPR base per operand stream and stage (kept opaque), every other offset an immediate
    int baseA[2], baseB2[2], baseB[2];
#pragma unroll
    for (int sgi = 0; sgi < 2; ++sgi) {
        baseA[sgi] = lds_addr(smem + sgi * STAGE + l15 * PA + ra_ * 3 * RP + (4 - pl) + 2 * lq);
        baseB2[sgi] = lds_addr(smem + sgi * STAGE + l15 * PA + rb_ * 3 * RP + (4 - pl) + 2 * lq);
        baseB[sgi] = lds_addr(smem + sgi * STAGE + AF + (h * 16 + l15) * PB + zrow * 48 + 2 * lq);
    }
    asm volatile("" : "+v"(baseA[0]), "+v"(baseA[1]), "+v"(baseB2[0]), "+v"(baseB2[1]), "+v"(baseB[0]), "+v"(baseB[1]));

    auto kstep = [&](auto stage, auto kstp) {
        constexpr int S = decltype(stage)::value, ks = decltype(kstp)::value;
        constexpr int trow = ks >> 1, tcol = 8 * (ks & 1);   // tiles 4 ks .. 4 ks + 3: tile row, first column
        const lds_cfp pa = lds_f(baseA[S]), pb = lds_f(baseB2[S]), pz = lds_f(baseB[S]);
        float av[3][4], bv[3][4];
#pragma unroll
        for (int mt = 0; mt < 3; ++mt) {
            const int o = mt * RP + 2 * trow * 3 * RP + tcol;
            float xv[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) xv[j] = __builtin_fmaf(pb[o + j], sgn, pa[o + j]);
            av[mt][0] = xv[0] - xv[2];
            av[mt][1] = xv[1] + xv[2];
            av[mt][2] = xv[2] - xv[1];
            av[mt][3] = xv[1] - xv[3];
        }
#pragma unroll
        for (int nt = 0; nt < 3; ++nt) {
            const int o = nt * 16 + 2 * trow * 48 + tcol;
            float z0 = pz[o], z1 = pz[o + 1];
            if (zmix) {
                z0 = __builtin_fmaf(pz[o + 48], zs, z0);
                z1 = __builtin_fmaf(pz[o + 48 + 1], zs, z1);
            }
            bv[nt][0] = z0;
            bv[nt][1] = z0 + z1;
            bv[nt][2] = z0 - z1;
            bv[nt][3] = z1;   // true value -z1
        }
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
            for (int mt = 0; mt < 3; ++mt)
#pragma unroll
                for (int nt = 0; nt < 3; ++nt)
                    acc[p][mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[mt][p], bv[nt][p], acc[p][mt][nt], 0, 0, 0);
    };
    auto stage_body = [&](auto stage, int region) {
        constexpr int S = decltype(stage)::value;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (region + (int)gridDim.x < nregions) issue(region + gridDim.x, smem + (1 - S) * STAGE);
        kstep(stage, IC<0>{});
        __builtin_amdgcn_sched_barrier(0);
        kstep(stage, IC<1>{});
        __builtin_amdgcn_sched_barrier(0);
        kstep(stage, IC<2>{});
        __builtin_amdgcn_sched_barrier(0);
        kstep(stage, IC<3>{});
    };

    int region = blockIdx.x;
    if (region < nregions) issue(region, smem);
    for (; region < nregions; region += 2 * gridDim.x) {
        stage_body(IC<0>{}, region);
        if (region + (int)gridDim.x < nregions) stage_body(IC<1>{}, region + gridDim.x);
    }
    __syncthreads();

    // final transform dW = G^T M G.  Column pass (over p = j) in registers, row pass (over pg) through LDS.
    // M[.][3] and M[3][.] carry a folded minus sign.
    float *X = smem;   // X[pg][h][item = (v, mt, r)][lane], one nt at a time
    const float s3 = pg == 3 ? -1.f : 1.f;   // this wave's row sign
    for (int nt = 0; nt < 3; ++nt) {
#pragma unroll
        for (int mt = 0; mt < 3; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float m0 = 0.f, m1 = 0.f, m2 = 0.f, m3 = 0.f;
#pragma unroll
                for (int n2 = 0; n2 < 3; ++n2)
                    if (n2 == nt) {
                        m0 = acc[0][mt][n2][r] * s3;
                        m1 = acc[1][mt][n2][r] * s3;
                        m2 = acc[2][mt][n2][r] * s3;
                        m3 = -acc[3][mt][n2][r] * s3;
                    }
                float *dst = X + (((pg * 2 + h) * 3 + 0) * 12 + mt * 4 + r) * 64 + lane;
                dst[0 * 12 * 64] = m0 + 0.5f * (m1 + m2);
                dst[1 * 12 * 64] = 0.5f * (m1 - m2);
                dst[2 * 12 * 64] = 0.5f * (m1 + m2) + m3;
            }
        __syncthreads();
        // reader: wave (pg, h) handles items pg*9 .. pg*9+8 of half h
        for (int it = 0; it < 9; ++it) {
            const int item = pg * 9 + it;   // (v, mt, r) = item / 12, (item % 12) / 4, item % 4
            const int v = item / 12, mt = (item % 12) / 4, r = item % 4;
            float xg[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) xg[i] = X[(((i * 2 + h) * 3 + v) * 12 + mt * 4 + r) * 64 + lane];
            const float u0 = xg[0] + 0.5f * (xg[1] + xg[2]);
            const float u1 = 0.5f * (xg[1] - xg[2]);
            const float u2 = 0.5f * (xg[1] + xg[2]) + xg[3];
            const int co = h * 48 + nt * 16 + l15, ci = grp * CIG + mt * 16 + 4 * lq + r;
            float *o = partial + (((long)blockIdx.x * Cout + co) * C + ci) * 9 + v;
            o[0] = u0;
            o[3] = u1;
            o[6] = u2;
        }
        __syncthreads();
    }
}

__global__ void reduce_parts(const float *__restrict__ partial, float *__restrict__ gw, long n, int parts) {
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    float s = 0.f;
    for (int p = 0; p < parts; ++p) s += partial[(long)p * n + e];
    gw[e] = s;
}

__global__ void naive_wgrad(const float *__restrict__ x, const float *__restrict__ gy, float *__restrict__ gw, int N,
                            int C, int H, int W, int Cout, int pt, int pl) {
    const int e = blockIdx.x;   // (co, ci, tap)
    const int tap = e % 9, ci = (e / 9) % C, co = e / (9 * C);
    const int u = tap / 3, v = tap % 3;
    double s = 0;
    for (long i = threadIdx.x; i < (long)N * H * W; i += blockDim.x) {
        const int xx = i % W, yy = (i / W) % H, n = i / ((long)W * H);
        const int iy = yy - pt + u, ix = xx - pl + v;
        if (iy < 0 || iy >= H || ix < 0 || ix >= W) continue;
        s += (double)gy[(((long)n * Cout + co) * H + yy) * W + xx] * x[(((long)n * C + ci) * H + iy) * W + ix];
    }
    __shared__ double red[256];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) gw[e] = (float)red[0];
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
    const int N = argc > 1 ? atoi(argv[1]) : 2, C = argc > 2 ? atoi(argv[2]) : 48, H = argc > 3 ? atoi(argv[3]) : 32,
              W = argc > 4 ? atoi(argv[4]) : 32, check = argc > 5 ? atoi(argv[5]) : 1;
    const int Cout = COG, pt = 2, pl = 1;
    if (C % CIG || H % RH || W % RW) {
        printf("bad shape\n");
        return 1;
    }
    const size_t nx = (size_t)N * C * H * W, ny = (size_t)N * Cout * H * W, nw = (size_t)Cout * C * 9;
    std::vector<float> hx(nx), hg(ny);
    srand(1);
    for (auto &v : hx) v = (rand() / (float)RAND_MAX) * 2.f - 1.f;
    for (auto &v : hg) v = (rand() / (float)RAND_MAX) * 2.f - 1.f;
    const int groups = C / CIG;
    const int regionsX = W / RW, regionsY = H / RH, nregions = N * regionsX * regionsY;
    int parts = 256 / groups;
    if (parts > nregions) parts = nregions;
    float *dx, *dg, *dw, *dr, *dp;
    CHECK(hipMalloc(&dx, nx * 4));
    CHECK(hipMalloc(&dg, ny * 4));
    CHECK(hipMalloc(&dw, nw * 4));
    CHECK(hipMalloc(&dr, nw * 4));
    CHECK(hipMalloc(&dp, (size_t)parts * nw * 4));
    CHECK(hipMemcpy(dx, hx.data(), nx * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(dg, hg.data(), ny * 4, hipMemcpyHostToDevice));
    CHECK(hipMemset(dp, 0, (size_t)parts * nw * 4));
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(wino_wgrad), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)kLdsBytes));
    const dim3 grid(parts, groups);
    auto run = [&]() {
        hipLaunchKernelGGL(wino_wgrad, grid, dim3(kThreads), kLdsBytes, 0, dx, dg, dp, N, C, H, W, Cout, pt, pl, regionsX,
                           regionsY);
        hipLaunchKernelGGL(reduce_parts, dim3((nw + 255) / 256), dim3(256), 0, 0, dp, dw, (long)nw, parts);
    };
    run();
    CHECK(hipDeviceSynchronize());
    CHECK(hipGetLastError());
    if (check) {
        hipLaunchKernelGGL(naive_wgrad, dim3(nw), dim3(256), 0, 0, dx, dg, dr, N, C, H, W, Cout, pt, pl);
        CHECK(hipDeviceSynchronize());
        std::vector<float> hw(nw), hr(nw);
        CHECK(hipMemcpy(hw.data(), dw, nw * 4, hipMemcpyDeviceToHost));
        CHECK(hipMemcpy(hr.data(), dr, nw * 4, hipMemcpyDeviceToHost));
        double maxerr = 0, maxref = 0;
        size_t worst = 0;
        for (size_t i = 0; i < nw; ++i) {
            const double e = fabs((double)hw[i] - hr[i]);
            if (e > maxerr) maxerr = e, worst = i;
            maxref = fmax(maxref, fabs((double)hr[i]));
        }
        printf("check N=%d C=%d %dx%d: max |err| %.3e (max |ref| %.3e) at %zu (co %zu ci %zu tap %zu): got %f want %f\n", N, C,
               H, W, maxerr, maxref, worst, worst / (9 * C), (worst / 9) % C, worst % 9, hw[worst], hr[worst]);
    }
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    const int reps = 10;
    run();
    CHECK(hipEventRecord(e0, 0));
    for (int i = 0; i < reps; ++i) run();
    CHECK(hipEventRecord(e1, 0));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    ms /= reps;
    const double flops = 2.0 * N * H * W * (double)Cout * C * 9;
    printf("wino_wgrad N=%d C=%d %dx%d (%d parts x %d groups): %.1f us, %.1f TFLOP/s direct-equivalent\n", N, C, H, W, parts,
           groups, ms * 1e3, flops / ms / 1e9);
    return 0;
}
