// Fused per-pixel head of the U-Nets (inference):  out = W3 . lrelu(W2 . lrelu(W1 . f + b1) + b2) + b3
//   blind-spot U-Net   384 -> 384 -> 96 -> 2   (models/joint_network_v2.py:123-153, 241-244 of the reference)
//   sigma net           96 ->  96 -> 96 -> 1   (models/joint_network_v2_shallow.py: output_block + output_conv)
// One launch instead of three 1x1 convolutions: the 128-pixel tile stays in the workgroup from the input features to
// the two output values; the two hidden activations (the largest tensors of whole-micrograph inference: 2 x 25.8 GB at
// 4096^2) never exist in HBM.  fp32 MFMA (v_mfma_f32_16x16x4_f32), fp32 accumulation; the k order of every sum is the
// channel order, as in the three-launch path.
//
// Workgroup = 8 waves, persistent over tiles of 128 consecutive pixels of one image's plane (HW % 128 == 0).
//   layer 1  [128 px x K0] . [K0 x N1]: wave (mh, nq) owns 128 / MW pixels x 96 output channels (MT x 6 accumulator tiles;
//            N1 = 384: 2 pixel halves x 4 channel quarters, MT = 4; N1 = 96: 8 pixel tiles x 1, MT = 1).  K in chunks of 16
//            channels, double-buffered by LDS-DMA: X chunk [16 ch][128 px] and the chunk's rows of W1^T [16][N1 + 16].
//   layer 2  [128 px x N1] . [N1 x 96]: the hidden activations go through LDS in rounds of 96 channels (the waves of
//            channel quarter r write theirs, everybody multiplies): wave w owns pixel tile w x 6 channel tiles.
//   layer 3  96 -> N3 <= 2: per lane over its 6 channel tiles, then a 16-lane shuffle reduction; lane 0 of each
//            quarter stores 4 consecutive pixels.
// LDS layouts are chosen for conflict-free 16x16x4 operand reads (lane = 16 rows x 4 k): every k row starts 16 banks
// after the previous one.  The X chunk arrives lane-linear (two channels of 128 floats per wave DMA), so the odd
// channel of a pair is rotated by 16 pixels ON THE SOURCE side and pairs are 288 floats apart.
#include <algorithm>
#include <cstdlib>

#include "common.h"
#include "wprep_dev.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void;
typedef __amdgpu_buffer_rsrc_t rsrc_t;
typedef const __attribute__((address_space(3))) float *lds_cfp;
constexpr float kLeak = 0.1f;
constexpr int kHThreads = 512, kTM = 128, kKC = 16, kN2 = 96;
constexpr int kPairStride = 288;          // floats between channel pairs of an X chunk (288 = 256 + 32: 32 banks on)
constexpr int kXFloats = 8 * kPairStride; // 16 channels
constexpr int kLDH = 144;                 // hidden-activation row stride (128 px + 16)
constexpr int kLDW2 = kN2 + 16;           // 112

__device__ __forceinline__ rsrc_t make_rsrc(const void *p) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, 0x7FFFFFFF, 0x00020000);
}
__device__ __forceinline__ const float *uniform_ptr(const float *p) {
    const unsigned long v = (unsigned long)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return (const float *)(((unsigned long)hi << 32) | lo);
}
__device__ __forceinline__ void bdma16(rsrc_t r, int voff, int soff, float *lds_wave_base) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void *)lds_wave_base, 16, voff, soff, 0, 0);
}
__device__ __forceinline__ void bdma4(rsrc_t r, int voff, int soff, float *lds_wave_base) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void *)lds_wave_base, 4, voff, soff, 0, 0);
}
constexpr int kXZero = (int)0x80000000;   // beyond num_records: the DMA writes zeros
__device__ __forceinline__ int lds_addr(const void *p) {
    return (int)(unsigned)(__SIZE_TYPE__)(const __attribute__((address_space(3))) char *)p;
}
__device__ __forceinline__ lds_cfp lds_f(int byte_addr) { return (lds_cfp)(__SIZE_TYPE__)(unsigned)byte_addr; }
__device__ __forceinline__ float lrelu(float v) { return v > 0.f ? v : v * kLeak; }

struct HeadArgs {
    const float *f;              // [B][K0][HW]
    const float *w1t, *w2t;      // prepared: W1^T [K0][N1 + 16], W2^T [N1][112]  (wprep_direct slabs, zero padded)
    const float *b1, *b2, *w3, *b3;
    float *out;                  // [B][N3][HW]
    int B, N3, tilesPerImage, ntiles;
    long HW;
    int P;                       // ROT: side of the square plane, f = d
};

template <int K0, int N1>
struct HeadGeo {
    static constexpr int LDW1 = N1 + 16;
    static constexpr int NWN = N1 / 96;             // waves along the output channels of layer 1
    static constexpr int MW = 8 / NWN;              // waves along the pixels
    static constexpr int MT = 8 / MW;               // pixel tiles (16 px) per wave in layer 1
    static constexpr int STAGE = kXFloats + kKC * LDW1;
    static constexpr int ROUNDS = NWN;              // layer-2 rounds of 96 hidden channels
    static constexpr int W2BUF = 96 * kLDW2;        // one round's rows of W2^T
    static constexpr int NW2 = ROUNDS > 1 ? 2 : 1;  // double-buffered when there is more than one round
    // region A: the two layer-1 stages, re-used for the hidden-activation tile of layer 2 and the h2 hand-over
    static constexpr int REGA = (2 * STAGE > 96 * kLDH ? 2 * STAGE : 96 * kLDH);
    static constexpr int CONST = N1 + kN2 + 2 * kN2 + 4;   // b1, b2, W3 (2 rows), b3
    static constexpr size_t LDS_BYTES = (size_t)(REGA + NW2 * W2BUF + CONST) * 4;
};

// ROT (blind-spot U-Net): the input features are read straight out of the rotated stack d [4B][96][P][P] — the
// Shift2d + chunk + rotate + concat of joint_network_v2.py:230-239 (sprk_unrot4_shift_concat_fwd) becomes the address
// computation of the X gather: f[b, 96 k + c, i, j] = d[k B + b, c][u - 1][v], (u, v) = R_k(i, j), zero for u = 0.  Tiles
// are then 8 rows x 16 columns (a row = one 16-pixel MFMA tile), so that every rotation reads 32- or 64-byte runs, and
// the 25.8 GB tensor f (4096^2) is neither written nor read.
template <int K0, int N1, bool ROT>
__global__ __launch_bounds__(kHThreads, 1) void head_fwd_kernel(const HeadArgs a) {
    using G = HeadGeo<K0, N1>;
    constexpr int LDW1 = G::LDW1, MT = G::MT, STAGE = G::STAGE, ROUNDS = G::ROUNDS, NCH = K0 / kKC;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *regA = smem;
    float *w2buf = smem + G::REGA;
    float *cst = w2buf + G::NW2 * G::W2BUF;      // b1[N1] b2[96] w3[2][96] b3[2..4]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), l15 = lane & 15,
              lq = lane >> 4;
    const int nq = wave % G::NWN, mh = wave / G::NWN;     // layer-1 ownership

    for (int i = tid; i < N1; i += kHThreads) cst[i] = a.b1[i];
    if (tid < kN2) {
        cst[N1 + tid] = a.b2[tid];
        cst[N1 + kN2 + tid] = a.w3[tid];
        cst[N1 + 2 * kN2 + tid] = a.N3 > 1 ? a.w3[kN2 + tid] : 0.f;
    }
    if (tid < 2) cst[N1 + 3 * kN2 + tid] = tid < a.N3 ? a.b3[tid] : 0.f;

    // X DMA: thread -> (channel of the chunk, 16-byte segment of its 128 pixels); the odd channel of a pair reads
    // its pixels rotated by 16 (4 segments), so that the two rows of a pair start 16 banks apart in LDS
    const int xch = tid >> 5, xseg = tid & 31;
    const int xvoff = (((xseg + 4 * (xch & 1)) & 31) * 4) * 4;          // byte offset inside the channel's 128 pixels
    float *const xdst_w = regA + (wave * kPairStride);                   // + stage offset; wave w holds pair w
    // W1 chunk: kKC * LDW1 floats, lane-linear
    constexpr int W1F = kKC * LDW1, W1FULL = W1F / 2048, W1REST = W1F % 2048;

    // operand addresses (bytes): A of layer 1
    int aoff[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int m = mh * MT + mt;
        aoff[mt] = lds_addr(regA + (lq >> 1) * kPairStride + (lq & 1) * 128 + (((m - (lq & 1)) & 7) * 16) + l15);
    }
    const int boff = lds_addr(regA + kXFloats + lq * LDW1 + nq * 96 + l15);

    for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
        const int b = tile / a.tilesPerImage, trem = tile - b * a.tilesPerImage;
        const int p0 = trem * kTM;
        const float *fimg = a.f + (long)b * K0 * a.HW + p0;
        // ROT: tile = 8 rows x 16 columns at (ty0, tx0); per thread the byte offsets of its 4 gather lanes (q: channel
        // of the wave's pair, half of the 128 pixel positions) under the 4 rotations; outside the shifted plane: zeros
        int ty0 = 0, tx0 = 0;
        int xo[4][4];
        if constexpr (ROT) {
            const int tilesX = a.P >> 4;
            const int tyi = trem / tilesX;
            ty0 = tyi * 8;
            tx0 = (trem - tyi * tilesX) * 16;
            const int P = a.P;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int ch = 2 * wave + (q >> 1), pos = (q & 1) * 64 + lane;
                const int px = (pos + 16 * (ch & 1)) & 127;
                const int i = ty0 + (px >> 4), j = tx0 + (px & 15);
                const int cho = ch * (int)a.HW;
                xo[0][q] = i >= 1 ? (cho + (i - 1) * P + j) * 4 : kXZero;
                xo[1][q] = j <= P - 2 ? (cho + (P - 2 - j) * P + i) * 4 : kXZero;
                xo[2][q] = i <= P - 2 ? (cho + (P - 2 - i) * P + (P - 1 - j)) * 4 : kXZero;
                xo[3][q] = j >= 1 ? (cho + (j - 1) * P + (P - 1 - i)) * 4 : kXZero;
            }
        }
        auto issue = [&](int c, int st) {
            if constexpr (ROT) {
                // chunk c = channels 16 cc .. of rotation group k; 4-byte gather lanes, 64 per instruction
                const int k = __builtin_amdgcn_readfirstlane(c / 6), cc = c - 6 * k;
                const rsrc_t xr = make_rsrc(uniform_ptr(a.f + (((long)k * a.B + b) * 96 + cc * kKC) * a.HW));
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int off = k == 0 ? xo[0][q] : k == 1 ? xo[1][q] : k == 2 ? xo[2][q] : xo[3][q];
                    bdma4(xr, off, 0, xdst_w + st * STAGE + q * 64);
                }
            } else {
                // X: channels c*16 .. +15; one 16-byte DMA per thread
                const rsrc_t xr = make_rsrc(uniform_ptr(fimg + (long)(c * kKC) * a.HW));
                // per-channel offset: channel stride HW*4 bytes may exceed what voffset + soffset should carry for
                // huge planes (HW * 16 channels * 4 < 2^31 for HW < 2^25 = 5792^2): checked on the host
                bdma16(xr, xvoff + xch * (int)(a.HW * 4), 0, xdst_w + st * STAGE);
            }
            const rsrc_t wr = make_rsrc(a.w1t + (long)c * W1F);
            float *wd = regA + st * STAGE + kXFloats;
#pragma unroll
            for (int gi = 0; gi < W1FULL; ++gi) bdma16(wr, tid * 16, gi * 8192, wd + gi * 2048 + wave * 256);
            if (W1REST && wave < W1REST / 256) bdma16(wr, tid * 16, W1FULL * 8192, wd + W1FULL * 2048 + wave * 256);
        };
        auto issue_w2 = [&](int r, int buf) {
            const rsrc_t wr = make_rsrc(a.w2t + (long)r * G::W2BUF);
            float *wd = w2buf + buf * G::W2BUF;
            constexpr int full = G::W2BUF / 2048, rest = G::W2BUF % 2048;
#pragma unroll
            for (int gi = 0; gi < full; ++gi) bdma16(wr, tid * 16, gi * 8192, wd + gi * 2048 + wave * 256);
            if (rest && wave < rest / 256) bdma16(wr, tid * 16, full * 8192, wd + full * 2048 + wave * 256);
        };

        // ---- layer 1 -------------------------------------------------------------------------------------------
        f32x4 acc[MT][6];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < 6; ++nt) acc[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        __syncthreads();                 // the previous tile's readers of region A / the constants are done
        issue(0, 0);
        issue_w2(0, 0);
        for (int c = 0; c < NCH; ++c) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (c + 1 < NCH) issue(c + 1, (c + 1) & 1);
            const int sb = (c & 1) * STAGE * 4;
#pragma unroll
            for (int ks = 0; ks < kKC / 4; ++ks) {
                float av[MT], bv[6];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) av[mt] = lds_f(aoff[mt] + sb)[ks * 2 * kPairStride];
#pragma unroll
                for (int nt = 0; nt < 6; ++nt) bv[nt] = lds_f(boff + sb)[ks * 4 * LDW1 + nt * 16];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int nt = 0; nt < 6; ++nt)
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[mt], bv[nt], acc[mt][nt], 0, 0, 0);
            }
        }
        // ---- layer 2 in rounds of 96 hidden channels ---------------------------------------------------------------
        f32x4 acc2[6];
#pragma unroll
        for (int nt = 0; nt < 6; ++nt) acc2[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        float *H1 = regA;                 // [96][kLDH]
        const int a2 = lds_addr(H1 + lq * kLDH + wave * 16 + l15);
#pragma unroll
        for (int r = 0; r < ROUNDS; ++r) {
            __syncthreads();              // layer-1 stages (r = 0) / the previous round's H1 tile are no longer read
            if (nq == r) {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int nt = 0; nt < 6; ++nt) {
                        const float bias = cst[r * 96 + nt * 16 + l15];
                        f32x4 v = acc[mt][nt];
#pragma unroll
                        for (int j = 0; j < 4; ++j) v[j] = lrelu(v[j] + bias);
                        *reinterpret_cast<f32x4 *>(H1 + (nt * 16 + l15) * kLDH + (mh * MT + mt) * 16 + 4 * lq) = v;
                    }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this round's rows of W2^T (issued a round ago)
            __syncthreads();
            // the next round's rows load while this round is multiplied; their buffer was last read in round r - 1
            if (r + 1 < ROUNDS) issue_w2(r + 1, (r + 1) & 1);
            const int b2a = lds_addr(w2buf + (r & (G::NW2 - 1)) * G::W2BUF + lq * kLDW2 + l15);
#pragma unroll 4
            for (int ks = 0; ks < 24; ++ks) {
                const float av = lds_f(a2)[ks * 4 * kLDH];
                float bv[6];
#pragma unroll
                for (int nt = 0; nt < 6; ++nt) bv[nt] = lds_f(b2a)[ks * 4 * kLDW2 + nt * 16];
#pragma unroll
                for (int nt = 0; nt < 6; ++nt) acc2[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv[nt], acc2[nt], 0, 0, 0);
            }
        }
        // ---- layer 3: lane (l15 = channel in tile, lq) holds h2 of pixels wave*16 + 4 lq + j ----------------------------
        float s0[4] = {0.f, 0.f, 0.f, 0.f}, s1[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int nt = 0; nt < 6; ++nt) {
            const float bias = cst[N1 + nt * 16 + l15], wa = cst[N1 + kN2 + nt * 16 + l15], wb = cst[N1 + 2 * kN2 + nt * 16 + l15];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float h = lrelu(acc2[nt][j] + bias);
                s0[j] = __builtin_fmaf(h, wa, s0[j]);
                s1[j] = __builtin_fmaf(h, wb, s1[j]);
            }
        }
        // sum over the 16 lanes of a quarter (wavefront shuffles; fixed order: xor 8, 4, 2, 1)
#pragma unroll
        for (int m = 8; m >= 1; m >>= 1)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                s0[j] += __shfl_xor(s0[j], m, 64);
                s1[j] += __shfl_xor(s1[j], m, 64);
            }
        if (l15 == 0) {
            const float c0 = cst[N1 + 3 * kN2], c1 = cst[N1 + 3 * kN2 + 1];
            float *o = ROT ? a.out + (long)b * a.N3 * a.HW + (long)(ty0 + wave) * a.P + tx0 + 4 * lq
                           : a.out + (long)b * a.N3 * a.HW + p0 + wave * 16 + 4 * lq;
            *reinterpret_cast<f32x4 *>(o) = (f32x4){s0[0] + c0, s0[1] + c0, s0[2] + c0, s0[3] + c0};
            if (a.N3 > 1) *reinterpret_cast<f32x4 *>(o + a.HW) = (f32x4){s1[0] + c1, s1[1] + c1, s1[2] + c1, s1[3] + c1};
        }
    }
}

template <int K0, int N1, bool ROT>
int launch_head(const HeadArgs &a, hipStream_t s) {
    using G = HeadGeo<K0, N1>;
    if (int rc = sprk::lds_optin(reinterpret_cast<const void *>(head_fwd_kernel<K0, N1, ROT>), G::LDS_BYTES, "head1x1_fwd"))
        return rc;
    hipLaunchKernelGGL((head_fwd_kernel<K0, N1, ROT>), dim3(std::min(a.ntiles, sprk::num_cus())), dim3(kHThreads), G::LDS_BYTES, s, a);
    return sprk::check_launch("head_fwd");
}

size_t w1_floats(int K0, int N1) { return (size_t)sprk::kWprepZeroFloats + (size_t)K0 * (N1 + 16); }
size_t w2_floats(int N1) { return (size_t)sprk::kWprepZeroFloats + (size_t)N1 * kLDW2; }

// P > 0: the input is the rotated stack d [4B][96][P][P] (ROT)
int run_head(const float *f, const float *w1, const float *b1, const float *w2, const float *b2, const float *w3,
             const float *b3, float *out, int B, int K0, int N1, int N3, long HW, int P, void *ws, void *stream) {
    hipStream_t s = (hipStream_t)stream;
    float *wsf = (float *)ws;
    float *w1t = wsf, *w2t = wsf + w1_floats(K0, N1);
    // W^T slabs in the k-chunked layout of the implicit-GEMM kernels (rows = input channels, row stride = LDS stride)
    const sprk_wprep_item items[2] = {
        sprk::wprep_item(sprk::WPREP_DIRECT, w1, w1t, (long)w1_floats(K0, N1), {N1, K0, 1, 0, kKC, kKC, K0, N1, N1 + 16, 1}),
        sprk::wprep_item(sprk::WPREP_DIRECT, w2, w2t, (long)w2_floats(N1), {kN2, N1, 1, 0, kKC, kKC, N1, kN2, kLDW2, 1})};
    if (int rc = sprk::wprep_launch(items, 2, s)) return rc;
    HeadArgs a{f, w1t + sprk::kWprepZeroFloats, w2t + sprk::kWprepZeroFloats, b1, b2, w3, b3, out, B, N3,
               (int)(HW / kTM), (int)(B * (HW / kTM)), HW, P};
    if (P > 0) return launch_head<384, 384, true>(a, s);
    return K0 == 384 ? launch_head<384, 384, false>(a, s) : launch_head<96, 96, false>(a, s);
}

}  // namespace

extern "C" {

size_t sprk_head1x1_fwd_ws_bytes(int K0, int N1) {
    if (!((K0 == 384 && N1 == 384) || (K0 == 96 && N1 == 96))) return 0;
    return (w1_floats(K0, N1) + w2_floats(N1)) * sizeof(float);
}

int sprk_head1x1_fwd(const float *f, const float *w1, const float *b1, const float *w2, const float *b2,
                     const float *w3, const float *b3, float *out, int B, int K0, int N1, int N3, long HW, void *ws,
                     size_t ws_bytes, void *stream) {
    SPRK_REQUIRE(f && w1 && b1 && w2 && b2 && w3 && b3 && out, "head1x1_fwd: null tensor");
    SPRK_REQUIRE((K0 == 384 && N1 == 384) || (K0 == 96 && N1 == 96), "head1x1_fwd: supported heads are 384->384->96 and 96->96->96");
    SPRK_REQUIRE(B > 0 && (N3 == 1 || N3 == 2) && HW > 0 && HW % kTM == 0 && HW < (1L << 25),
                 "head1x1_fwd: needs HW %% 128 == 0, HW < 2^25, 1 or 2 outputs");
    SPRK_REQUIRE(((((uintptr_t)f | (uintptr_t)out | (uintptr_t)ws) & 15) == 0), "head1x1_fwd: tensors must be 16-byte aligned");
    const size_t need = sprk_head1x1_fwd_ws_bytes(K0, N1);
    if (!ws || ws_bytes < need) {
        sprk::set_error("head1x1_fwd: workspace %zu < %zu", ws_bytes, need);
        return SPRK_EWORKSPACE;
    }
    return run_head(f, w1, b1, w2, b2, w3, b3, out, B, K0, N1, N3, HW, 0, ws, stream);
}

int sprk_head1x1_unrot_fwd(const float *d, const float *w1, const float *b1, const float *w2, const float *b2,
                           const float *w3, const float *b3, float *out, int B, int C, int P, int N3, void *ws,
                           size_t ws_bytes, void *stream) {
    SPRK_REQUIRE(d && w1 && b1 && w2 && b2 && w3 && b3 && out, "head1x1_unrot_fwd: null tensor");
    SPRK_REQUIRE(C == 96, "head1x1_unrot_fwd: the blind-spot head has 4 x 96 input channels");
    SPRK_REQUIRE(B > 0 && (N3 == 1 || N3 == 2) && P > 0 && P % 16 == 0 && (long)P * P < (1L << 25),
                 "head1x1_unrot_fwd: needs P %% 16 == 0, P^2 < 2^25, 1 or 2 outputs");
    SPRK_REQUIRE(((((uintptr_t)d | (uintptr_t)out | (uintptr_t)ws) & 15) == 0), "head1x1_unrot_fwd: tensors must be 16-byte aligned");
    const size_t need = sprk_head1x1_fwd_ws_bytes(384, 384);
    if (!ws || ws_bytes < need) {
        sprk::set_error("head1x1_unrot_fwd: workspace %zu < %zu", ws_bytes, need);
        return SPRK_EWORKSPACE;
    }
    return run_head(d, w1, b1, w2, b2, w3, b3, out, B, 384, 384, N3, (long)P * P, P, ws, stream);
}

}  // extern "C"
