// HBM-bound plumbing kernels of the blind-spot U-Net and the per-pixel pipeline maths.
// Every kernel reads each input element once and writes each output element once
// (coalesced along x); none of them is reshaped into a GEMM.
#include "common.h"
#include "io_dev.h"

#include <algorithm>

namespace {

// ---- Shift2d((shift,0)) + MaxPool2d(2) -------------------------------------------------------
// xs = x shifted down by `shift` rows (zeros on top, last rows dropped); y = 2x2/2 max of xs.
// (TX / TY / TG ...: element types of the tensors, fp32 or the 16-bit storage type: io_dev.h.  max commutes with the
// monotone rounding of a 16-bit tensor, so pooling 16-bit values equals rounding the pooled fp32 values.)
template <typename TX>
__device__ __forceinline__ float shifted(const TX *p, int u, int v, int W, int shift) {
    const int r = u - shift;
    return r >= 0 ? IO<TX>::ld(p + (long)r * W + v) : 0.f;
}

template <typename TX, typename TY>
__global__ void shift_maxpool2_fwd_kernel(const TX *__restrict__ x, TY *__restrict__ y, int NC, int H, int W,
                                          int shift) {
    const int Ho = H >> 1, Wo = W >> 1;
    const long total = (long)NC * Ho * Wo;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        long t = e;
        const int j = (int)(t % Wo); t /= Wo;
        const int i = (int)(t % Ho);
        const long nc = t / Ho;
        const TX *p = x + nc * H * W;
        float m = shifted(p, 2 * i, 2 * j, W, shift);
        float v = shifted(p, 2 * i, 2 * j + 1, W, shift);
        if (v > m || v != v) m = v;
        v = shifted(p, 2 * i + 1, 2 * j, W, shift);
        if (v > m || v != v) m = v;
        v = shifted(p, 2 * i + 1, 2 * j + 1, W, shift);
        if (v > m || v != v) m = v;
        IO<TY>::st(y + e, m);
    }
}

// the same with four outputs per thread (W % 8 == 0, 16-byte aligned planes): two float4 pairs in, one float4 out, no
// 64-bit index division.  (The scalar kernel ran at 0.9 TB/s on the 3.2 GB encoder planes of a 4096^2 micrograph.)
__device__ __forceinline__ float max4_first(float a, float b, float c, float d) {
    float m = a;
    if (b > m || b != b) m = b;
    if (c > m || c != c) m = c;
    if (d > m || d != d) m = d;
    return m;
}

template <typename TX, typename TY>
__global__ __launch_bounds__(256) void shift_maxpool2_fwd_v4_kernel(const TX *__restrict__ x, TY *__restrict__ y,
                                                                    int NC, int H, int W, int shift) {
    const int Ho = H >> 1, Wo4 = W >> 3;
    const int per = Ho * Wo4;
    for (int nc = blockIdx.y; nc < NC; nc += gridDim.y) {
        const TX *p = x + (long)nc * H * W;
        TY *q = y + (long)nc * Ho * (W >> 1);
        for (int t = blockIdx.x * 256 + threadIdx.x; t < per; t += gridDim.x * 256) {
            const int i = t / Wo4, j4 = t - i * Wo4;
            const int ra = 2 * i - shift, rb = ra + 1;
            const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
            const float4 a0 = ra >= 0 ? IO<TX>::ld4(p + (long)ra * W + 8 * j4) : z;
            const float4 a1 = ra >= 0 ? IO<TX>::ld4(p + (long)ra * W + 8 * j4 + 4) : z;
            const float4 b0 = rb >= 0 ? IO<TX>::ld4(p + (long)rb * W + 8 * j4) : z;
            const float4 b1 = rb >= 0 ? IO<TX>::ld4(p + (long)rb * W + 8 * j4 + 4) : z;
            float4 o;
            o.x = max4_first(a0.x, a0.y, b0.x, b0.y);
            o.y = max4_first(a0.z, a0.w, b0.z, b0.w);
            o.z = max4_first(a1.x, a1.y, b1.x, b1.y);
            o.w = max4_first(a1.z, a1.w, b1.z, b1.w);
            IO<TY>::st4(q + (long)i * (W >> 1) + 4 * j4, o);
        }
    }
}

// one thread per x element: it receives the window's gradient iff it is the FIRST maximum of
// its window in row-major order (torch's max_pool2d backward rule).
template <typename TG, typename TX, typename TO>
__global__ void shift_maxpool2_bwd_kernel(const TG *__restrict__ gy, const TX *__restrict__ x,
                                          TO *__restrict__ gx, int NC, int H, int W, int shift, int act) {
    const int Ho = H >> 1, Wo = W >> 1;
    const long total = (long)NC * H * W;
    const float neg = act == SPRK_ACT_LEAKY ? 0.1f : 0.f;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        long t = e;
        const int v0 = (int)(t % W); t /= W;
        const int r0 = (int)(t % H);
        const long nc = t / H;
        const int u = r0 + shift;  // row in the shifted image
        float g = 0.f;
        if (u < 2 * Ho && v0 < 2 * Wo) {
            const int i = u >> 1, j = v0 >> 1;
            const TX *p = x + nc * H * W;
            int best = 0;
            float m = shifted(p, 2 * i, 2 * j, W, shift);
#pragma unroll
            for (int k = 1; k < 4; ++k) {
                const float v = shifted(p, 2 * i + (k >> 1), 2 * j + (k & 1), W, shift);
                if (v > m || v != v) {
                    m = v;
                    best = k;
                }
            }
            const int mine = ((u & 1) << 1) | (v0 & 1);
            if (best == mine) g = IO<TG>::ld(gy + (nc * Ho + i) * Wo + j);
        }
        if (act != SPRK_ACT_NONE && !(IO<TX>::ld(x + e) > 0.f)) g *= neg;
        IO<TO>::st(gx + e, g);
    }
}

// the same rule with four x elements of one row per thread (W % 4 == 0, H even, 16-byte aligned planes): two
// windows; the thread loads its own row and the partner row of the window pair (float4 each, the partner's from
// cache), the two gradients (float2) and writes one float4 — no 64-bit index division, 16-byte accesses
__device__ __forceinline__ int first_max4(float a, float b, float c, float d) {
    int best = 0;
    float m = a;
    if (b > m || b != b) m = b, best = 1;
    if (c > m || c != c) m = c, best = 2;
    if (d > m || d != d) m = d, best = 3;
    return best;
}

template <typename TG, typename TX, typename TO>
__global__ __launch_bounds__(256) void shift_maxpool2_bwd_v4_kernel(const TG *__restrict__ gy,
                                                                    const TX *__restrict__ x, TO *__restrict__ gx,
                                                                    int NC, int H, int W, int shift, int act) {
    const float neg = act == SPRK_ACT_LEAKY ? 0.1f : 0.f;
    const int Ho = H >> 1, Wo = W >> 1, W4 = W >> 2;
    const int per = H * W4;
    for (int nc = blockIdx.y; nc < NC; nc += gridDim.y) {
        const TX *p = x + (long)nc * H * W;
        for (int t = blockIdx.x * 256 + threadIdx.x; t < per; t += gridDim.x * 256) {
            const int r0 = t / W4, q = t - r0 * W4;
            const int u = r0 + shift;               // row in the shifted image
            float4 out = make_float4(0.f, 0.f, 0.f, 0.f);
            if (u < 2 * Ho) {
                const int i = u >> 1, ra = (u & ~1) - shift, rb = ra + 1;   // x rows of the window (ra < 0: zeros)
                const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
                const float4 A = ra >= 0 ? IO<TX>::ld4(p + (long)ra * W + 4 * q) : z;
                const float4 B = rb >= 0 ? IO<TX>::ld4(p + (long)rb * W + 4 * q) : z;
                const float2 g = IO<TG>::ld2(gy + ((long)nc * Ho + i) * Wo + 2 * q);
                const int b0 = first_max4(A.x, A.y, B.x, B.y), b1 = first_max4(A.z, A.w, B.z, B.w);
                const int k0 = (u & 1) << 1;        // window positions of this thread's row: k0, k0 + 1
                out.x = b0 == k0 ? g.x : 0.f;
                out.y = b0 == k0 + 1 ? g.x : 0.f;
                out.z = b1 == k0 ? g.y : 0.f;
                out.w = b1 == k0 + 1 ? g.y : 0.f;
                if (act != SPRK_ACT_NONE) {   // this thread's own x row: A for the window's first row, B for the second
                    const float4 X = (u & 1) ? B : A;
                    out.x *= X.x > 0.f ? 1.f : neg;
                    out.y *= X.y > 0.f ? 1.f : neg;
                    out.z *= X.z > 0.f ? 1.f : neg;
                    out.w *= X.w > 0.f ? 1.f : neg;
                }
            }
            IO<TO>::st4(gx + (long)nc * H * W + (long)r0 * W + 4 * q, out);
        }
    }
}

// ---- rotations -------------------------------------------------------------------------------
// clockwise rotation R_k (k*90 degrees) of a PxP plane: out[i][j] = in[src(i,j)]
__device__ __forceinline__ void rot_src(int k, int i, int j, int P, int &u, int &v) {
    switch (k) {
        case 0: u = i; v = j; break;
        case 1: u = j; v = P - 1 - i; break;           // 90
        case 2: u = P - 1 - i; v = P - 1 - j; break;   // 180
        default: u = P - 1 - j; v = i; break;          // 270
    }
}
// inverse: given a source position (u,v) return the output position (i,j) it lands on
__device__ __forceinline__ void rot_dst(int k, int u, int v, int P, int &i, int &j) {
    switch (k) {
        case 0: i = u; j = v; break;
        case 1: i = P - 1 - v; j = u; break;
        case 2: i = P - 1 - u; j = P - 1 - v; break;
        default: i = v; j = P - 1 - u; break;
    }
}

__global__ void rot4_stack_fwd_kernel(const float *__restrict__ x, float *__restrict__ y, int B, int C, int P) {
    const long plane = (long)P * P, per = (long)B * C * plane, total = 4 * per;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int k = (int)(e / per);
        long t = e - k * per;
        const int j = (int)(t % P); t /= P;
        const int i = (int)(t % P);
        const long bc = t / P;
        int u, v;
        rot_src(k, i, j, P, u, v);
        y[e] = x[bc * plane + (long)u * P + v];
    }
}

__global__ void rot4_stack_bwd_kernel(const float *__restrict__ gy, float *__restrict__ gx, int B, int C, int P) {
    const long plane = (long)P * P, per = (long)B * C * plane;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < per; e += (long)gridDim.x * blockDim.x) {
        long t = e;
        const int v = (int)(t % P); t /= P;
        const int u = (int)(t % P);
        const long bc = t / P;
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            int i, j;
            rot_dst(k, u, v, P, i, j);
            s += gy[k * per + bc * plane + (long)i * P + j];
        }
        gx[e] = s;
    }
}

// f[b, k*C + c, i, j] = R_{a_k}( shift_down_1( d[k*B + b, c] ) )[i, j],  a = (0, 270, 180, 90)
// 32x32 tiles through LDS: the source tile is read along its rows and the output tile written along
// its rows (both coalesced) whatever the rotation; P is a multiple of 32 (blind-spot networks need
// P % 32 == 0 anyway).  grid: (tiles per plane, planes), block 32x8.
//   FWD:  out (i,j) of plane (b, k*C+c)  <- shifted source (u,v) = rot_src(a_k; i,j), value d[u-1][v] (0 if u == 0)
//   BWD:  out (s,v) of plane (k*B+b, c)  <- gf (i,j) = rot_dst(a_k; s+1, v)                 (0 if s == P-1)
template <bool FWD, typename TI, typename TO>
__global__ __launch_bounds__(256) void unrot4_tiled_kernel(const TI *__restrict__ in, TO *__restrict__ out,
                                                           int B, int C, int P) {
    __shared__ float tile[32][33];
    const int tilesPer = P >> 5;
    const int ti = blockIdx.x / tilesPer, tj = blockIdx.x % tilesPer;
    const int pl = blockIdx.y;  // output plane index
    int k, b, c;
    if (FWD) {
        b = pl / (4 * C);
        const int kc = pl - b * 4 * C;
        k = kc / C;
        c = kc - k * C;
    } else {
        k = pl / (B * C);
        const int bc = pl - k * B * C;
        b = bc / C;
        c = bc - b * C;
    }
    const int rot = (4 - k) & 3;
    const long plane = (long)P * P;
    const TI *src = in + (FWD ? (((long)k * B + b) * C + c) : (((long)b * 4 + k) * C + c)) * plane;
    TO *dst = out + (long)pl * plane;
    const int o0 = ti << 5, o1 = tj << 5;  // output tile origin (row, col)
    // source tile origin: image of the output tile's corners under the (affine) index map
    int a0, a1, b0, b1;
    if (FWD) {
        rot_src(rot, o0, o1, P, a0, a1);
        rot_src(rot, o0 + 31, o1 + 31, P, b0, b1);
    } else {
        rot_dst(rot, o0 + 1, o1, P, a0, a1);
        rot_dst(rot, o0 + 32, o1 + 31, P, b0, b1);
    }
    const int s0 = min(a0, b0), s1 = min(a1, b1);
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
    for (int r = ty; r < 32; r += 8) {
        // source coordinates in the shifted (FWD) / gf (BWD) plane
        const int u = s0 + r, v = s1 + tx;
        float val = 0.f;
        if (FWD) {
            if (u >= 1 && u < P && v >= 0 && v < P) val = IO<TI>::ld(src + (long)(u - 1) * P + v);
        } else {
            if (u >= 0 && u < P && v >= 0 && v < P) val = IO<TI>::ld(src + (long)u * P + v);
        }
        tile[r][tx] = val;
    }
    __syncthreads();
#pragma unroll
    for (int r = ty; r < 32; r += 8) {
        const int i = o0 + r, j = o1 + tx;
        int u, v;
        float val;
        if (FWD) {
            rot_src(rot, i, j, P, u, v);
            val = tile[u - s0][v - s1];
        } else {
            if (i + 1 < P) {
                rot_dst(rot, i + 1, j, P, u, v);
                val = tile[u - s0][v - s1];
            } else {
                val = 0.f;
            }
        }
        IO<TO>::st(dst + (long)i * P + j, val);
    }
}

// The same for planes whose side is a multiple of 64 (every training and inference size of the blind-spot networks):
// 64x64 tiles, one per workgroup — at P = 64 a whole plane.  Each wave reads whole 64-element source rows (256 / 128
// contiguous bytes) into the LDS tile; every thread then writes four consecutive output elements per pass as one 16- /
// 8-byte store.  (The 32x32 kernel above moved 1 K elements per 256-thread workgroup with 4- / 2-byte accesses:
// 1.9 TB/s on the [32,384,64,64] tensor of a training step; this one moves 4 K per workgroup.)
template <bool FWD, typename TI, typename TO>
__global__ __launch_bounds__(256) void unrot4_tile64_kernel(const TI *__restrict__ in, TO *__restrict__ out,
                                                            int B, int C, int P) {
    __shared__ float tile[64][65];
    const int tilesPer = P >> 6;
    const int ti = blockIdx.x / tilesPer, tj = blockIdx.x % tilesPer;
    const int pl = blockIdx.y;  // output plane index
    int k, b, c;
    if (FWD) {
        b = pl / (4 * C);
        const int kc = pl - b * 4 * C;
        k = kc / C;
        c = kc - k * C;
    } else {
        k = pl / (B * C);
        const int bc = pl - k * B * C;
        b = bc / C;
        c = bc - b * C;
    }
    const int rot = (4 - k) & 3;
    const long plane = (long)P * P;
    const TI *src = in + (FWD ? (((long)k * B + b) * C + c) : (((long)b * 4 + k) * C + c)) * plane;
    TO *dst = out + (long)pl * plane;
    const int o0 = ti << 6, o1 = tj << 6;  // output tile origin (row, col)
    int a0, a1, b0, b1;
    if (FWD) {
        rot_src(rot, o0, o1, P, a0, a1);
        rot_src(rot, o0 + 63, o1 + 63, P, b0, b1);
    } else {
        rot_dst(rot, o0 + 1, o1, P, a0, a1);
        rot_dst(rot, o0 + 64, o1 + 63, P, b0, b1);
    }
    const int s0 = min(a0, b0), s1 = min(a1, b1);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll 4
    for (int r = wave; r < 64; r += 4) {
        // source coordinates in the shifted (FWD) / gf (BWD) plane
        const int u = s0 + r, v = s1 + lane;
        float val = 0.f;
        if (FWD) {
            if (u >= 1 && u < P && v >= 0 && v < P) val = IO<TI>::ld(src + (long)(u - 1) * P + v);
        } else {
            if (u >= 0 && u < P && v >= 0 && v < P) val = IO<TI>::ld(src + (long)u * P + v);
        }
        tile[r][lane] = val;
    }
    __syncthreads();
    const int j4 = (threadIdx.x & 15) << 2;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int r = (threadIdx.x >> 4) + 16 * p;
        const int i = o0 + r;
        float o[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int j = o1 + j4 + q;
            int u, v;
            if (FWD) {
                rot_src(rot, i, j, P, u, v);
                o[q] = tile[u - s0][v - s1];
            } else if (i + 1 < P) {
                rot_dst(rot, i + 1, j, P, u, v);
                o[q] = tile[u - s0][v - s1];
            } else {
                o[q] = 0.f;
            }
        }
        IO<TO>::st4(dst + (long)i * P + o1 + j4, make_float4(o[0], o[1], o[2], o[3]));
    }
}

template <typename TI, typename TO>
__global__ void unrot4_fwd_kernel(const TI *__restrict__ d, TO *__restrict__ f, int B, int C, int P) {
    const long plane = (long)P * P, total = (long)B * 4 * C * plane;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        long t = e;
        const int j = (int)(t % P); t /= P;
        const int i = (int)(t % P); t /= P;
        const int kc = (int)(t % (4 * C));
        const int b = (int)(t / (4 * C));
        const int k = kc / C, c = kc - k * C;
        int u, v;
        rot_src((4 - k) & 3, i, j, P, u, v);
        float val = 0.f;
        if (u >= 1) val = IO<TI>::ld(d + (((long)k * B + b) * C + c) * plane + (long)(u - 1) * P + v);
        IO<TO>::st(f + e, val);
    }
}

template <typename TI, typename TO>
__global__ void unrot4_bwd_kernel(const TI *__restrict__ gf, TO *__restrict__ gd, int B, int C, int P) {
    const long plane = (long)P * P, total = (long)4 * B * C * plane;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        long t = e;
        const int v = (int)(t % P); t /= P;
        const int s = (int)(t % P); t /= P;
        const int c = (int)(t % C); t /= C;
        const int b = (int)(t % B);
        const int k = (int)(t / B);
        float g = 0.f;
        if (s + 1 < P) {
            int i, j;
            rot_dst((4 - k) & 3, s + 1, v, P, i, j);
            g = IO<TI>::ld(gf + (((long)b * 4 + k) * C + c) * plane + (long)i * P + j);
        }
        IO<TO>::st(gd + e, g);
    }
}

// ---- reparameterisation, sigmoid --------------------------------------------------------------
__global__ void reparam_fwd_kernel(const float *__restrict__ o, const float *__restrict__ eps, float *__restrict__ z,
                                   int B, int HW) {
    const long total = (long)B * HW;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const long b = e / HW, i = e - b * HW;
        const float mu = o[(2 * b) * HW + i], a = o[(2 * b + 1) * HW + i];
        z[e] = mu + eps[e] * (a * a);
    }
}

__global__ void reparam_bwd_kernel(const float *__restrict__ gz, const float *__restrict__ o,
                                   const float *__restrict__ eps, float *__restrict__ go, int B, int HW) {
    const long total = (long)B * HW;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const long b = e / HW, i = e - b * HW;
        const float g = gz[e], a = o[(2 * b + 1) * HW + i];
        go[(2 * b) * HW + i] = g;
        go[(2 * b + 1) * HW + i] = g * eps[e] * 2.f * a;
    }
}

__device__ __forceinline__ float sigmoidf(float x) { return 1.f / (1.f + expf(-x)); }

__global__ void sigmoid_clamp_fwd_kernel(const float *__restrict__ x, float *__restrict__ p, long n) {
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long)gridDim.x * blockDim.x) {
        const float s = sigmoidf(x[e]);
        p[e] = fminf(fmaxf(s, 1e-4f), 1.f - 1e-4f);
    }
}

__global__ void sigmoid_clamp_bwd_kernel(const float *__restrict__ gp, const float *__restrict__ x,
                                         float *__restrict__ gx, long n) {
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long)gridDim.x * blockDim.x) {
        const float s = sigmoidf(x[e]);
        const bool pass = s >= 1e-4f && s <= 1.f - 1e-4f;  // torch.clamp passes the gradient on [min, max]
        gx[e] = pass ? gp[e] * s * (1.f - s) : 0.f;
    }
}

// ---- SSDN gaussian likelihood -------------------------------------------------------------------
constexpr int kSsdnBlk = 256;

__device__ __forceinline__ float block_sum(float v, float *red) {
    red[threadIdx.x] = v;
    __syncthreads();
    for (int w = kSsdnBlk / 2; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
        __syncthreads();
    }
    const float r = red[0];
    __syncthreads();
    return r;
}

// ---- PU detection loss (utils/losses.py:303-349), value and gradient in one workgroup ------------------------
__device__ __forceinline__ float block_max(float v, float *red) {
    red[threadIdx.x] = v;
    __syncthreads();
    for (int w = kSsdnBlk / 2; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + w]);
        __syncthreads();
    }
    const float r = red[0];
    __syncthreads();
    return r;
}

__global__ __launch_bounds__(kSsdnBlk) void pu_loss_kernel(const float *__restrict__ p, const float *__restrict__ y,
                                                           const float *__restrict__ log_binom, int B, float slack,
                                                           float *__restrict__ loss, float *__restrict__ gp) {
    __shared__ float red[kSsdnBlk];
    const int tid = threadIdx.x;
    float nl = 0.f, nu = 0.f, bce = 0.f, mu = 0.f, var = 0.f;
    for (int i = tid; i < B; i += kSsdnBlk) {
        const float pi = p[i], yi = y[i];
        if (yi >= 0.f) {
            nl += 1.f;
            bce -= yi * logf(pi) + (1.f - yi) * logf(1.f - pi);
        } else if (yi == -1.f) {
            nu += 1.f;
            mu += pi;
            var += pi * (1.f - pi);
        }
    }
    nl = block_sum(nl, red);
    nu = block_sum(nu, red);
    bce = block_sum(bce, red);
    mu = block_sum(mu, red);
    var = block_sum(var, red);
    const int N = (int)nu;
    const float inv_lab = 1.f / fmaxf(nl, 1.f);
    const float iv = 1.f / (var + 1e-7f);
    const float *lb = log_binom + (long)N * (B + 1);
    // softmax over k = 0..N of -(mu - k)^2 / (2 (var + 1e-7)), and the sums its gradient needs
    float mx = -INFINITY;
    for (int k = tid; k <= N; k += kSsdnBlk) {
        const float dk = mu - (float)k;
        mx = fmaxf(mx, -0.5f * dk * dk * iv);
    }
    mx = block_max(mx, red);
    float Z = 0.f, A = 0.f, B1 = 0.f, B2 = 0.f, C1 = 0.f, C2 = 0.f;
    for (int k = tid; k <= N; k += kSsdnBlk) {
        const float dk = mu - (float)k;
        const float e = expf(-0.5f * dk * dk * iv - mx);
        const float l = lb[k];
        const float d1 = -dk * iv;                 // d logit_k / d mu
        const float d2 = 0.5f * dk * dk * iv * iv; // d logit_k / d var
        Z += e;
        A += e * l;
        B1 += e * l * d1;
        B2 += e * d1;
        C1 += e * l * d2;
        C2 += e * d2;
    }
    Z = block_sum(Z, red);
    A = block_sum(A, red);
    B1 = block_sum(B1, red);
    B2 = block_sum(B2, red);
    C1 = block_sum(C1, red);
    C2 = block_sum(C2, red);
    const float S = A / Z;                         // sum_k log_binom_k q_k
    const float g_mu = -slack * (B1 - S * B2) / Z; // d loss / d mu
    const float g_var = -slack * (C1 - S * C2) / Z;
    if (tid == 0) loss[0] = bce * inv_lab - slack * S;
    for (int i = tid; i < B; i += kSsdnBlk) {
        const float pi = p[i], yi = y[i];
        float g = 0.f;
        if (yi >= 0.f)
            g = -(yi / pi - (1.f - yi) / (1.f - pi)) * inv_lab;
        else if (yi == -1.f)
            g = g_mu + g_var * (1.f - 2.f * pi);
        gp[i] = g;
    }
}

// grid (nblk, B); partial[b*nblk + blk] = sum of nll over the block's pixels.
// POISSON (denoiser_v2.py:412-424): ns[b] is the remapped estimate e, the noise variance is signal dependent,
// var_n = max(mu, 1e-3) * e per pixel, and the regulariser -0.05 * sqrt(var_n) is per pixel too.
template <bool POISSON>
__global__ __launch_bounds__(kSsdnBlk) void ssdn_fwd_kernel(const float *__restrict__ x, const float *__restrict__ o,
                                                            const float *__restrict__ ns, float *__restrict__ partial,
                                                            float *__restrict__ pme, float *__restrict__ mstd,
                                                            float *__restrict__ nsmap, int HW, float *__restrict__ direct) {
    __shared__ float red[kSsdnBlk];
    const int b = blockIdx.y;
    const float e = ns[b];
    float acc = 0.f;
    for (int i = blockIdx.x * kSsdnBlk + threadIdx.x; i < HW; i += gridDim.x * kSsdnBlk) {
        const float xv = x[(long)b * HW + i];
        const float mu = o[(long)(2 * b) * HW + i], a = o[(long)(2 * b + 1) * HW + i];
        float s, vn;
        if (POISSON) {
            s = sqrtf(fmaxf(mu, 1e-3f) * e);
            vn = s * s;                                 // the reference squares the root again (noise_std ** 2)
        } else {
            s = e;
            vn = s * s;
        }
        const float vx = a * a, vy = vx + vn, dd = xv - mu;
        acc += dd * dd / vy + logf(vy) - 0.05f * s;
        if (pme) pme[(long)b * HW + i] = (xv * vx + mu * vn) / (vx + vn);
        if (mstd) mstd[(long)b * HW + i] = sqrtf(vx);
        if (POISSON && nsmap) nsmap[(long)b * HW + i] = s;
    }
    const float tot = block_sum(acc, red);
    if (threadIdx.x == 0) {
        if (direct)   // one workgroup per image (patches): the image's mean, no second launch
            direct[b] = tot / (float)HW;
        else
            partial[b * gridDim.x + blockIdx.x] = tot;
    }
}

__global__ void ssdn_finish_kernel(const float *__restrict__ partial, float *__restrict__ out, int B, int nblk,
                                   float scale) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    float s = 0.f;
    for (int i = 0; i < nblk; ++i) s += partial[b * nblk + i];
    out[b] = s * scale;
}

template <bool POISSON>
__global__ __launch_bounds__(kSsdnBlk) void ssdn_bwd_kernel(const float *__restrict__ gl, const float *__restrict__ x,
                                                            const float *__restrict__ o, const float *__restrict__ ns,
                                                            float *__restrict__ go, float *__restrict__ partial, int HW,
                                                            float *__restrict__ direct) {
    __shared__ float red[kSsdnBlk];
    const int b = blockIdx.y;
    const float e = ns[b];
    const float g = gl[b] / (float)HW;
    float acc = 0.f;
    for (int i = blockIdx.x * kSsdnBlk + threadIdx.x; i < HW; i += gridDim.x * kSsdnBlk) {
        const float xv = x[(long)b * HW + i];
        const float mu = o[(long)(2 * b) * HW + i], a = o[(long)(2 * b + 1) * HW + i];
        if (POISSON) {
            const float m = fmaxf(mu, 1e-3f);
            const float s = sqrtf(m * e), vn = s * s;
            const float vy = a * a + vn, dd = xv - mu;
            const float dvy = 1.f / vy - dd * dd / (vy * vy);  // d nll / d var_y
            const float gvn = dvy * 2.f * s * (0.5f / s) - 0.05f * (0.5f / s);   // through var_n = s^2 and through -0.05 s
            // torch.maximum hands the gradient to mu above the floor, half of it at a tie, none below
            const float pass = mu > 1e-3f ? 1.f : (mu == 1e-3f ? 0.5f : 0.f);
            go[(long)(2 * b) * HW + i] = g * (-2.f * dd / vy + gvn * e * pass);
            go[(long)(2 * b + 1) * HW + i] = g * dvy * 2.f * a;
            acc += gvn * m;
        } else {
            const float s = e, vn = s * s;
            const float vy = a * a + vn, dd = xv - mu;
            const float dvy = 1.f / vy - dd * dd / (vy * vy);
            go[(long)(2 * b) * HW + i] = g * (-2.f * dd / vy);
            go[(long)(2 * b + 1) * HW + i] = g * dvy * 2.f * a;
            acc += dvy * 2.f * s - 0.05f;
        }
    }
    const float tot = block_sum(acc, red);
    if (threadIdx.x == 0) {
        if (direct)
            direct[b] = tot * g;
        else
            partial[b * gridDim.x + blockIdx.x] = tot * g;
    }
}

// ---- small fused pieces of the training step's tail: each replaces a chain of framework launches by one (a kernel of
// a replayed step costs ~5 us whatever it does: profiles/r04_timeline_*) -------------------------------------------
// ResidA (models/feature_extractor.py:384-416): out = y + x[:, :, off + s i, off + s j] (y == nullptr: the crop alone)
__global__ void crop_add_fwd_kernel(const float *__restrict__ y, const float *__restrict__ x, float *__restrict__ out,
                                    int Ho, int Wo, int Hx, int Wx, int off, int stride, long total) {
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int j = (int)(idx % Wo);
        const long t = idx / Wo;
        const int i = (int)(t % Ho);
        const long nc = t / Ho;
        const float v = x[(nc * Hx + off + i * stride) * Wx + off + j * stride];
        out[idx] = y ? y[idx] + v : v;
    }
}

// its gradient with respect to x: g scattered into the cropped (strided) positions, zero elsewhere
__global__ void crop_embed_bwd_kernel(const float *__restrict__ g, float *__restrict__ gx, int Ho, int Wo, int Hx, int Wx,
                                      int off, int stride, long total) {
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int jx = (int)(idx % Wx);
        const long t = idx / Wx;
        const int ix = (int)(t % Hx);
        const long nc = t / Hx;
        const int a = ix - off, b = jx - off;
        float v = 0.f;
        if (a >= 0 && b >= 0 && a % stride == 0 && b % stride == 0) {
            const int i = a / stride, j = b / stride;
            if (i < Ho && j < Wo) v = g[(nc * Ho + i) * Wo + j];
        }
        gx[idx] = v;
    }
}

// noise level of an image from the estimator's map (denoiser_v2.py:392-402):  z = mean(est) - 4,
// out = softplus(z) + 1e-3 (softplus with torch's threshold 20).  One workgroup per image.
__global__ __launch_bounds__(kSsdnBlk) void noise_std_fwd_kernel(const float *__restrict__ est, float *__restrict__ out,
                                                                 float *__restrict__ z_out, int HW) {
    __shared__ float red[kSsdnBlk];
    const int b = blockIdx.x;
    float acc = 0.f;
    for (int i = threadIdx.x; i < HW; i += kSsdnBlk) acc += est[(long)b * HW + i];
    const float z = block_sum(acc, red) / (float)HW - 4.0f;
    if (threadIdx.x == 0) {
        out[b] = (z > 20.f ? z : log1pf(expf(z))) + 1e-3f;
        z_out[b] = z;
    }
}

__global__ void noise_std_bwd_kernel(const float *__restrict__ g, const float *__restrict__ z, float *__restrict__ gest,
                                     int HW, long total) {
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int b = (int)(idx / HW);
        const float zz = z[b];
        const float sg = zz > 20.f ? 1.f : 1.f / (1.f + expf(-zz));
        gest[idx] = g[b] * sg / (float)HW;
    }
}

// the training loss of the joint pipeline (denoiser_v2.py:516-519):
//   consis = mean((p - flip(pf))^2);   final[b] = alpha * loss_out[b] + (1 - alpha) * pred + wc * consis
// p, pf: [B,1,H,W] scores of the batch and of the flipped batch (pf NOT yet flipped back: axis 0 = along W, 1 = along H).
// One workgroup, sums in a fixed order.
__device__ __forceinline__ int flip_index(int i, int H, int W, int axis) {
    const int w = i % W, h = (i / W) % H, b = i / (W * H);
    return (b * H + (axis ? H - 1 - h : h)) * W + (axis ? w : W - 1 - w);
}

__global__ __launch_bounds__(kSsdnBlk) void joint_loss_fwd_kernel(const float *__restrict__ loss_out,
                                                                  const float *__restrict__ pred,
                                                                  const float *__restrict__ p, const float *__restrict__ pf,
                                                                  float *__restrict__ final_loss, float *__restrict__ consis,
                                                                  int B, int n, int H, int W, int axis, float alpha, float wc) {
    __shared__ float red[kSsdnBlk];
    float acc = 0.f;
    for (int i = threadIdx.x; i < n; i += kSsdnBlk) {
        const float d = p[i] - pf[flip_index(i, H, W, axis)];
        acc += d * d;
    }
    const float mse = block_sum(acc, red) / (float)n;
    if (threadIdx.x == 0) consis[0] = mse;
    const float rest = (1.f - alpha) * pred[0] + wc * mse;
    for (int b = threadIdx.x; b < B; b += kSsdnBlk) final_loss[b] = alpha * loss_out[b] + rest;
}

__global__ __launch_bounds__(kSsdnBlk) void joint_loss_bwd_kernel(const float *__restrict__ go, const float *__restrict__ p,
                                                                  const float *__restrict__ pf, float *__restrict__ g_loss_out,
                                                                  float *__restrict__ g_pred, float *__restrict__ gp,
                                                                  float *__restrict__ gpf, int B, int n, int H, int W,
                                                                  int axis, float alpha, float wc) {
    __shared__ float red[kSsdnBlk];
    float acc = 0.f;
    for (int b = threadIdx.x; b < B; b += kSsdnBlk) {
        acc += go[b];
        g_loss_out[b] = alpha * go[b];
    }
    const float S = block_sum(acc, red);
    if (threadIdx.x == 0) g_pred[0] = (1.f - alpha) * S;
    const float k = wc * S * 2.f / (float)n;
    for (int i = threadIdx.x; i < n; i += kSsdnBlk) {
        const int fi = flip_index(i, H, W, axis);
        const float v = k * (p[i] - pf[fi]);
        gp[i] = v;
        gpf[fi] = -v;
    }
}

// patches (<= 128x128): one workgroup per image writes the image's sum itself; larger images: partial sums + a finishing launch
int ssdn_nblk(int HW) { return HW <= 16384 ? 1 : std::min(64, sprk::cdiv(HW, kSsdnBlk * 4)); }


// ---- Adam over many parameter tensors in one launch --------------------------------------------------------------
__global__ __launch_bounds__(256) void adam_multi_kernel(const sprk_adam_item *__restrict__ items,
                                                         const int *__restrict__ start, int n_items,
                                                         const float *__restrict__ lr, const float *__restrict__ step_in,
                                                         float *__restrict__ step_out, float b1, float b2, float eps) {
    // workgroup -> item: binary search in the start table (wave-uniform)
    const int b = blockIdx.x;
    int lo = 0, hi = n_items - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (start[mid] <= b) lo = mid; else hi = mid - 1;
    }
    const sprk_adam_item it = items[lo];
    const float t = step_in[0] + 1.f;
    const float bc1 = 1.f - powf(b1, t), bc2 = 1.f - powf(b2, t);
    const float step_size = lr[0] / bc1, rs2 = 1.f / sqrtf(bc2);
    const long e0 = (long)(b - start[lo]) * 1024 + threadIdx.x * 4;
    if (e0 + 3 < it.n && ((((uintptr_t)it.p | (uintptr_t)it.g | (uintptr_t)it.m | (uintptr_t)it.v) & 15) == 0)) {
        const float4 g = *reinterpret_cast<const float4 *>(it.g + e0);
        float4 m = *reinterpret_cast<const float4 *>(it.m + e0), v = *reinterpret_cast<const float4 *>(it.v + e0);
        float4 p = *reinterpret_cast<const float4 *>(it.p + e0);
        m.x = b1 * m.x + (1.f - b1) * g.x; m.y = b1 * m.y + (1.f - b1) * g.y;
        m.z = b1 * m.z + (1.f - b1) * g.z; m.w = b1 * m.w + (1.f - b1) * g.w;
        v.x = b2 * v.x + (1.f - b2) * g.x * g.x; v.y = b2 * v.y + (1.f - b2) * g.y * g.y;
        v.z = b2 * v.z + (1.f - b2) * g.z * g.z; v.w = b2 * v.w + (1.f - b2) * g.w * g.w;
        p.x -= step_size * m.x / (sqrtf(v.x) * rs2 + eps); p.y -= step_size * m.y / (sqrtf(v.y) * rs2 + eps);
        p.z -= step_size * m.z / (sqrtf(v.z) * rs2 + eps); p.w -= step_size * m.w / (sqrtf(v.w) * rs2 + eps);
        *reinterpret_cast<float4 *>(it.m + e0) = m;
        *reinterpret_cast<float4 *>(it.v + e0) = v;
        *reinterpret_cast<float4 *>(it.p + e0) = p;
    } else {
        for (long e = e0; e < it.n && e < e0 + 4; ++e) {
            const float g = it.g[e];
            const float m = b1 * it.m[e] + (1.f - b1) * g, v = b2 * it.v[e] + (1.f - b2) * g * g;
            it.m[e] = m;
            it.v[e] = v;
            it.p[e] -= step_size * m / (sqrtf(v) * rs2 + eps);
        }
    }
    if (b == 0 && threadIdx.x == 0) step_out[0] = t;
}

}  // namespace

extern "C" {

int sprk_shift_maxpool2_fwd(const void *x, void *y, int NC, int H, int W, int shift, int io, void *stream) {
    SPRK_REQUIRE(x && y && NC > 0 && H >= 2 && W >= 2 && shift >= 0, "shift_maxpool2_fwd: bad arguments");
    const long total = (long)NC * (H / 2) * (W / 2);
    const bool v4 = W % 8 == 0 && H % 2 == 0 && ((((uintptr_t)x | (uintptr_t)y) & 15) == 0);
    const int drc = dispatch_io2(io, [&](auto tx, auto ty) {
        using TX = decltype(tx);
        using TY = decltype(ty);
        if (v4) {
            const int per = (H / 2) * (W / 8);
            dim3 grid(std::min(sprk::cdiv(per, 256), 64), std::min(NC, 32768));
            hipLaunchKernelGGL((shift_maxpool2_fwd_v4_kernel<TX, TY>), grid, dim3(256), 0, (hipStream_t)stream, (const TX *)x,
                               (TY *)y, NC, H, W, shift);
        } else {
            hipLaunchKernelGGL((shift_maxpool2_fwd_kernel<TX, TY>), dim3(sprk::ew_blocks(total)), dim3(256), 0,
                               (hipStream_t)stream, (const TX *)x, (TY *)y, NC, H, W, shift);
        }
        return 0;
    });
    SPRK_REQUIRE(drc == 0, "shift_maxpool2_fwd: bad storage types (io)");
    return sprk::check_launch("shift_maxpool2_fwd");
}

int sprk_shift_maxpool2_bwd(const void *gy, const void *x, void *gx, int NC, int H, int W, int shift, int act, int io,
                            void *stream) {
    SPRK_REQUIRE(gy && x && gx && NC > 0 && H >= 2 && W >= 2 && shift >= 0, "shift_maxpool2_bwd: bad arguments");
    const long total = (long)NC * H * W;
    const bool v4 = W % 4 == 0 && H % 2 == 0 && ((((uintptr_t)x | (uintptr_t)gx) & 15) == 0) && (((uintptr_t)gy & 7) == 0);
    const int drc = dispatch_io3(io, [&](auto tg, auto tx, auto to) {
        using TG = decltype(tg);
        using TX = decltype(tx);
        using TO = decltype(to);
        if (v4) {
            const int per = H * (W / 4);
            dim3 grid(std::min(sprk::cdiv(per, 256), 64), std::min(NC, 32768));
            hipLaunchKernelGGL((shift_maxpool2_bwd_v4_kernel<TG, TX, TO>), grid, dim3(256), 0, (hipStream_t)stream,
                               (const TG *)gy, (const TX *)x, (TO *)gx, NC, H, W, shift, act);
        } else {
            hipLaunchKernelGGL((shift_maxpool2_bwd_kernel<TG, TX, TO>), dim3(sprk::ew_blocks(total)), dim3(256), 0,
                               (hipStream_t)stream, (const TG *)gy, (const TX *)x, (TO *)gx, NC, H, W, shift, act);
        }
        return 0;
    });
    SPRK_REQUIRE(drc == 0, "shift_maxpool2_bwd: bad storage types (io)");
    return sprk::check_launch("shift_maxpool2_bwd");
}

int sprk_rot4_stack_fwd(const float *x, float *y, int B, int C, int P, void *stream) {
    SPRK_REQUIRE(x && y && B > 0 && C > 0 && P > 0, "rot4_stack_fwd: bad arguments");
    hipLaunchKernelGGL(rot4_stack_fwd_kernel, dim3(sprk::ew_blocks(4L * B * C * P * P)), dim3(256), 0,
                       (hipStream_t)stream, x, y, B, C, P);
    return sprk::check_launch("rot4_stack_fwd");
}

int sprk_rot4_stack_bwd(const float *gy, float *gx, int B, int C, int P, void *stream) {
    SPRK_REQUIRE(gy && gx && B > 0 && C > 0 && P > 0, "rot4_stack_bwd: bad arguments");
    hipLaunchKernelGGL(rot4_stack_bwd_kernel, dim3(sprk::ew_blocks((long)B * C * P * P)), dim3(256), 0,
                       (hipStream_t)stream, gy, gx, B, C, P);
    return sprk::check_launch("rot4_stack_bwd");
}

int sprk_unrot4_shift_concat_fwd(const void *d, void *f, int B, int C, int P, int io, void *stream) {
    SPRK_REQUIRE(d && f && B > 0 && C > 0 && P > 0, "unrot4_shift_concat_fwd: bad arguments");
    const bool tiled = P % 32 == 0 && (long)4 * B * C < 65536;
    const bool t64 = tiled && P % 64 == 0 && (((uintptr_t)f) & 15) == 0;
    const int drc = dispatch_io2(io, [&](auto ti, auto to) {
        using TI = decltype(ti);
        using TO = decltype(to);
        if (t64)
            hipLaunchKernelGGL((unrot4_tile64_kernel<true, TI, TO>), dim3((P / 64) * (P / 64), 4 * B * C), dim3(256), 0,
                               (hipStream_t)stream, (const TI *)d, (TO *)f, B, C, P);
        else if (tiled)
            hipLaunchKernelGGL((unrot4_tiled_kernel<true, TI, TO>), dim3((P / 32) * (P / 32), 4 * B * C), dim3(256), 0,
                               (hipStream_t)stream, (const TI *)d, (TO *)f, B, C, P);
        else
            hipLaunchKernelGGL((unrot4_fwd_kernel<TI, TO>), dim3(sprk::ew_blocks(4L * B * C * P * P)), dim3(256), 0,
                               (hipStream_t)stream, (const TI *)d, (TO *)f, B, C, P);
        return 0;
    });
    SPRK_REQUIRE(drc == 0, "unrot4_shift_concat_fwd: bad storage types (io)");
    return sprk::check_launch("unrot4_fwd");
}

int sprk_unrot4_shift_concat_bwd(const void *gf, void *gd, int B, int C, int P, int io, void *stream) {
    SPRK_REQUIRE(gf && gd && B > 0 && C > 0 && P > 0, "unrot4_shift_concat_bwd: bad arguments");
    const bool tiled = P % 32 == 0 && (long)4 * B * C < 65536;
    const bool t64 = tiled && P % 64 == 0 && (((uintptr_t)gd) & 15) == 0;
    const int drc = dispatch_io2(io, [&](auto ti, auto to) {
        using TI = decltype(ti);
        using TO = decltype(to);
        if (t64)
            hipLaunchKernelGGL((unrot4_tile64_kernel<false, TI, TO>), dim3((P / 64) * (P / 64), 4 * B * C), dim3(256), 0,
                               (hipStream_t)stream, (const TI *)gf, (TO *)gd, B, C, P);
        else if (tiled)
            hipLaunchKernelGGL((unrot4_tiled_kernel<false, TI, TO>), dim3((P / 32) * (P / 32), 4 * B * C), dim3(256), 0,
                               (hipStream_t)stream, (const TI *)gf, (TO *)gd, B, C, P);
        else
            hipLaunchKernelGGL((unrot4_bwd_kernel<TI, TO>), dim3(sprk::ew_blocks(4L * B * C * P * P)), dim3(256), 0,
                               (hipStream_t)stream, (const TI *)gf, (TO *)gd, B, C, P);
        return 0;
    });
    SPRK_REQUIRE(drc == 0, "unrot4_shift_concat_bwd: bad storage types (io)");
    return sprk::check_launch("unrot4_bwd");
}

int sprk_reparam_fwd(const float *out_stats, const float *eps, float *z, int B, int HW, void *stream) {
    SPRK_REQUIRE(out_stats && eps && z && B > 0 && HW > 0, "reparam_fwd: bad arguments");
    hipLaunchKernelGGL(reparam_fwd_kernel, dim3(sprk::ew_blocks((long)B * HW)), dim3(256), 0, (hipStream_t)stream,
                       out_stats, eps, z, B, HW);
    return sprk::check_launch("reparam_fwd");
}

int sprk_reparam_bwd(const float *gz, const float *out_stats, const float *eps, float *g_out_stats, int B, int HW,
                     void *stream) {
    SPRK_REQUIRE(gz && out_stats && eps && g_out_stats && B > 0 && HW > 0, "reparam_bwd: bad arguments");
    hipLaunchKernelGGL(reparam_bwd_kernel, dim3(sprk::ew_blocks((long)B * HW)), dim3(256), 0, (hipStream_t)stream, gz,
                       out_stats, eps, g_out_stats, B, HW);
    return sprk::check_launch("reparam_bwd");
}

int sprk_sigmoid_clamp_fwd(const float *x, float *p, long n, void *stream) {
    SPRK_REQUIRE(x && p && n > 0, "sigmoid_clamp_fwd: bad arguments");
    hipLaunchKernelGGL(sigmoid_clamp_fwd_kernel, dim3(sprk::ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, x, p, n);
    return sprk::check_launch("sigmoid_clamp_fwd");
}

int sprk_sigmoid_clamp_bwd(const float *gp, const float *x, float *gx, long n, void *stream) {
    SPRK_REQUIRE(gp && x && gx && n > 0, "sigmoid_clamp_bwd: bad arguments");
    hipLaunchKernelGGL(sigmoid_clamp_bwd_kernel, dim3(sprk::ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, gp, x, gx,
                       n);
    return sprk::check_launch("sigmoid_clamp_bwd");
}

int sprk_adam_multi(const sprk_adam_item *items, const int *start, int n_items, int n_blocks, const float *lr,
                    const float *step_in, float *step_out, float beta1, float beta2, float eps, void *stream) {
    SPRK_REQUIRE(items && start && n_items > 0 && n_blocks > 0 && lr && step_in && step_out && step_in != step_out,
                 "adam_multi: bad arguments");
    hipLaunchKernelGGL(adam_multi_kernel, dim3(n_blocks), dim3(256), 0, (hipStream_t)stream, items, start, n_items, lr,
                       step_in, step_out, beta1, beta2, eps);
    return sprk::check_launch("adam_multi");
}

int sprk_pu_loss(const float *p, const float *y, const float *log_binom, int B, float slack, float *loss, float *gp,
                 void *stream) {
    SPRK_REQUIRE(p && y && log_binom && loss && gp && B > 0 && B < (1 << 20), "pu_loss: bad arguments");
    hipLaunchKernelGGL(pu_loss_kernel, dim3(1), dim3(kSsdnBlk), 0, (hipStream_t)stream, p, y, log_binom, B, slack, loss, gp);
    return sprk::check_launch("pu_loss");
}

int sprk_crop_add_fwd(const float *y, const float *x, float *out, long NC, int Ho, int Wo, int Hx, int Wx, int off, int stride,
                      void *stream) {
    SPRK_REQUIRE(x && out && NC > 0 && Ho > 0 && Wo > 0 && off >= 0 && stride > 0, "crop_add_fwd: bad arguments");
    SPRK_REQUIRE(off + (Ho - 1) * stride < Hx && off + (Wo - 1) * stride < Wx, "crop_add_fwd: the crop leaves the source plane");
    const long total = NC * Ho * Wo;
    hipLaunchKernelGGL(crop_add_fwd_kernel, dim3(sprk::ew_blocks(total)), dim3(256), 0, (hipStream_t)stream, y, x, out, Ho, Wo,
                       Hx, Wx, off, stride, total);
    return sprk::check_launch("crop_add_fwd");
}

int sprk_crop_embed_bwd(const float *g, float *gx, long NC, int Ho, int Wo, int Hx, int Wx, int off, int stride, void *stream) {
    SPRK_REQUIRE(g && gx && NC > 0 && Ho > 0 && Wo > 0 && off >= 0 && stride > 0, "crop_embed_bwd: bad arguments");
    SPRK_REQUIRE(off + (Ho - 1) * stride < Hx && off + (Wo - 1) * stride < Wx, "crop_embed_bwd: the crop leaves the source plane");
    const long total = NC * Hx * Wx;
    hipLaunchKernelGGL(crop_embed_bwd_kernel, dim3(sprk::ew_blocks(total)), dim3(256), 0, (hipStream_t)stream, g, gx, Ho, Wo,
                       Hx, Wx, off, stride, total);
    return sprk::check_launch("crop_embed_bwd");
}

int sprk_noise_std_fwd(const float *est, float *noise_std, float *z, int B, int HW, void *stream) {
    SPRK_REQUIRE(est && noise_std && z && B > 0 && HW > 0, "noise_std_fwd: bad arguments");
    hipLaunchKernelGGL(noise_std_fwd_kernel, dim3(B), dim3(kSsdnBlk), 0, (hipStream_t)stream, est, noise_std, z, HW);
    return sprk::check_launch("noise_std_fwd");
}

int sprk_noise_std_bwd(const float *g, const float *z, float *g_est, int B, int HW, void *stream) {
    SPRK_REQUIRE(g && z && g_est && B > 0 && HW > 0, "noise_std_bwd: bad arguments");
    const long total = (long)B * HW;
    hipLaunchKernelGGL(noise_std_bwd_kernel, dim3(sprk::ew_blocks(total)), dim3(256), 0, (hipStream_t)stream, g, z, g_est, HW,
                       total);
    return sprk::check_launch("noise_std_bwd");
}

int sprk_joint_loss_fwd(const float *loss_out, const float *pred, const float *p, const float *pf, float *final_loss,
                        float *consis, int B, int H, int W, int axis, float alpha, float w_consis, void *stream) {
    SPRK_REQUIRE(loss_out && pred && p && pf && final_loss && consis && B > 0 && H > 0 && W > 0, "joint_loss_fwd: bad arguments");
    SPRK_REQUIRE(axis == 0 || axis == 1, "joint_loss_fwd: axis is 0 (flip along W) or 1 (along H)");
    SPRK_REQUIRE((long)B * H * W < (1L << 24), "joint_loss_fwd: score maps too large for the one-workgroup form");
    hipLaunchKernelGGL(joint_loss_fwd_kernel, dim3(1), dim3(kSsdnBlk), 0, (hipStream_t)stream, loss_out, pred, p, pf, final_loss,
                       consis, B, B * H * W, H, W, axis, alpha, w_consis);
    return sprk::check_launch("joint_loss_fwd");
}

int sprk_joint_loss_bwd(const float *g_final, const float *p, const float *pf, float *g_loss_out, float *g_pred, float *gp,
                        float *gpf, int B, int H, int W, int axis, float alpha, float w_consis, void *stream) {
    SPRK_REQUIRE(g_final && p && pf && g_loss_out && g_pred && gp && gpf && B > 0 && H > 0 && W > 0,
                 "joint_loss_bwd: bad arguments");
    SPRK_REQUIRE(axis == 0 || axis == 1, "joint_loss_bwd: axis is 0 (flip along W) or 1 (along H)");
    SPRK_REQUIRE((long)B * H * W < (1L << 24), "joint_loss_bwd: score maps too large for the one-workgroup form");
    hipLaunchKernelGGL(joint_loss_bwd_kernel, dim3(1), dim3(kSsdnBlk), 0, (hipStream_t)stream, g_final, p, pf, g_loss_out,
                       g_pred, gp, gpf, B, B * H * W, H, W, axis, alpha, w_consis);
    return sprk::check_launch("joint_loss_bwd");
}

size_t sprk_ssdn_ws_bytes(int B, int HW) { return (size_t)B * ssdn_nblk(HW) * sizeof(float); }

int sprk_ssdn_fwd(const float *x, const float *out_stats, const float *noise_std, float *loss, float *pme,
                  float *model_std, float *noise_std_map, int style, int B, int HW, void *ws, size_t ws_bytes,
                  void *stream) {
    SPRK_REQUIRE(x && out_stats && noise_std && loss && B > 0 && HW > 0, "ssdn_fwd: bad arguments");
    SPRK_REQUIRE(style == SPRK_NOISE_GAUSSIAN || style == SPRK_NOISE_POISSON, "ssdn_fwd: unknown noise style");
    const int nblk = ssdn_nblk(HW);
    if (!ws || ws_bytes < (size_t)B * nblk * sizeof(float)) {
        sprk::set_error("ssdn_fwd: workspace too small");
        return SPRK_EWORKSPACE;
    }
    hipStream_t s = (hipStream_t)stream;
    if (style == SPRK_NOISE_POISSON)
        hipLaunchKernelGGL(ssdn_fwd_kernel<true>, dim3(nblk, B), dim3(kSsdnBlk), 0, s, x, out_stats, noise_std, (float *)ws,
                           pme, model_std, noise_std_map, HW, nblk == 1 ? loss : nullptr);
    else
        hipLaunchKernelGGL(ssdn_fwd_kernel<false>, dim3(nblk, B), dim3(kSsdnBlk), 0, s, x, out_stats, noise_std, (float *)ws,
                           pme, model_std, noise_std_map, HW, nblk == 1 ? loss : nullptr);
    if (int rc = sprk::check_launch("ssdn_fwd")) return rc;
    if (nblk == 1) return SPRK_OK;
    hipLaunchKernelGGL(ssdn_finish_kernel, dim3(sprk::cdiv(B, 64)), dim3(64), 0, s, (const float *)ws, loss, B, nblk,
                       1.0f / (float)HW);
    return sprk::check_launch("ssdn_finish");
}

int sprk_ssdn_bwd(const float *gloss, const float *x, const float *out_stats, const float *noise_std,
                  float *g_out_stats, float *g_noise_std, int style, int B, int HW, void *ws, size_t ws_bytes,
                  void *stream) {
    SPRK_REQUIRE(gloss && x && out_stats && noise_std && g_out_stats && g_noise_std && B > 0 && HW > 0,
                 "ssdn_bwd: bad arguments");
    SPRK_REQUIRE(style == SPRK_NOISE_GAUSSIAN || style == SPRK_NOISE_POISSON, "ssdn_bwd: unknown noise style");
    const int nblk = ssdn_nblk(HW);
    if (!ws || ws_bytes < (size_t)B * nblk * sizeof(float)) {
        sprk::set_error("ssdn_bwd: workspace too small");
        return SPRK_EWORKSPACE;
    }
    hipStream_t s = (hipStream_t)stream;
    if (style == SPRK_NOISE_POISSON)
        hipLaunchKernelGGL(ssdn_bwd_kernel<true>, dim3(nblk, B), dim3(kSsdnBlk), 0, s, gloss, x, out_stats, noise_std,
                           g_out_stats, (float *)ws, HW, nblk == 1 ? g_noise_std : nullptr);
    else
        hipLaunchKernelGGL(ssdn_bwd_kernel<false>, dim3(nblk, B), dim3(kSsdnBlk), 0, s, gloss, x, out_stats, noise_std,
                           g_out_stats, (float *)ws, HW, nblk == 1 ? g_noise_std : nullptr);
    if (int rc = sprk::check_launch("ssdn_bwd")) return rc;
    if (nblk == 1) return SPRK_OK;
    hipLaunchKernelGGL(ssdn_finish_kernel, dim3(sprk::cdiv(B, 64)), dim3(64), 0, s, (const float *)ws, g_noise_std, B,
                       nblk, 1.0f);
    return sprk::check_launch("ssdn_finish");
}

}  // extern "C"
