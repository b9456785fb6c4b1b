// Device-side building blocks shared by the convolution translation units (conv.hip: fp32 MFMA kernels,
// conv16.hip: bf16 / fp16-operand MFMA kernels): LDS address helpers, LDS-DMA staging of input tiles through
// per-workgroup offset tables, the kernel argument block and the fused epilogue.  Not part of the ABI.
#pragma once
#include "common.h"

using f32x4 = __attribute__((ext_vector_type(4))) float;

namespace {

constexpr int kThreads = 256;       // 4 compute waves (one per SIMD)
// Optional dedicated DMA loader waves (measured slower than letting the 4 MFMA waves issue their own
// share of the DMA: LDS-DMA issue is paced by the CU's address path, not by the issuing wave): 0 = off.
// Also measured slower (fwd 100 -> 90 TF, wgrad 63 -> 44 TF): issuing the next stage's DMA a few
// instructions at a time between the MFMAs of the running stage instead of in one burst after the barrier.
constexpr int kLoaders = 0;
constexpr int kBlock = kThreads + 64 * kLoaders;
constexpr float kLeak = 0.1f;

__device__ __forceinline__ float apply_act(float v, int act) {
    if (act == SPRK_ACT_LEAKY) return v > 0.f ? v : v * kLeak;
    if (act == SPRK_ACT_RELU) return v > 0.f ? v : 0.f;
    return v;
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() is a fence + barrier and drains EVERY outstanding
// vector memory operation first (s_waitcnt vmcnt(0)): register prefetches that are meant to stay in flight across the
// barrier would be waited for at each one.  Here only lgkmcnt (ds_read / ds_write) is drained; a kernel that stages
// through LDS-DMA (counted by vmcnt) adds the s_waitcnt vmcnt(n) it needs itself, n = the vector memory operations it
// issued AFTER that DMA (vmcnt counts in order).
__device__ __forceinline__ void lds_only_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// ------------------------------------------------------------------------------------------
// LDS-DMA staging (global_load_lds: HBM/L2 -> LDS without passing through VGPRs)
// ------------------------------------------------------------------------------------------
typedef __attribute__((address_space(3))) void lds_void;
// LDS addresses as plain integers: one VGPR add per operand address, immediates for the rest
typedef const __attribute__((address_space(3))) float *lds_cfp;
typedef const __attribute__((address_space(3))) int *lds_cip;
__device__ __forceinline__ int lds_addr(const void *p) {
    return (int)(unsigned)(__SIZE_TYPE__)(const __attribute__((address_space(3))) char *)p;
}
__device__ __forceinline__ lds_cfp lds_f(int byte_addr) { return (lds_cfp)(__SIZE_TYPE__)(unsigned)byte_addr; }
__device__ __forceinline__ lds_cip lds_i(int byte_addr) { return (lds_cip)(__SIZE_TYPE__)(unsigned)byte_addr; }

__device__ __forceinline__ void dma4(const float *src, float *lds_wave_base) {
    __builtin_amdgcn_global_load_lds(src, (lds_void *)lds_wave_base, 4, 0, 0);
}
__device__ __forceinline__ void dma16(const float *src, float *lds_wave_base) {
    __builtin_amdgcn_global_load_lds(src, (lds_void *)lds_wave_base, 16, 0, 0);
}

// LDS image of an input tile: [channel][image][row][col], row pitch `pitch` (a multiple of 4), channel
// stride `cplane`; column j of the image is input column ixa + j where ixa = ix0 - colOff is a multiple
// of 4, so every 4-column chunk of the image is a 16-byte aligned run of one global row that lies
// entirely inside or entirely outside the image (Win % 4 == 0).  The [image][row][col] plane is
// contiguous, i.e. lane-linear for LDS-DMA: 64 chunks (16 B per lane) or 64 elements (4 B per lane,
// the fallback for upsampled-on-load or unaligned sources) per wave instruction; lanes outside the
// image read the zero block.
struct PlaneGeom {
    int NI, inRows, pitch, colOff, cplane;
    float invImg, invPitch;
    int deal;  // 1: deal (group, channel) items round-robin to waves; 0: whole groups per wave
    int nw;    // number of waves sharing the DMA issue
};

// ---- buffer-addressed, table-driven staging ---------------------------------------------------------
// Non-MFMA VALU instructions take issue time from the MFMA stream of the SIMD (measured: ~8 cycles each,
// a quarter of a 16x16x4 fp32 MFMA), and the pointer arithmetic of stage_planes (divisions, bounds checks,
// 64-bit multiplies, zero-block selects) was ~1.3 VALU per MFMA over a 96-channel convolution.  Here the
// per-lane part of a DMA address is computed once per workgroup (build_xtab, a byte offset per lane and
// 256/64-float group of a channel plane, kept in LDS) and the per-channel part is wave-uniform:
//   buffer_load_dword[x4] voffset(table), rsrc(source tensor at the chunk's first channel), soffset(channel) lds
// Lanes outside the image carry voffset 0x80000000: beyond num_records, the load returns 0 into LDS (checked
// on gfx950 by scratch/buflds_test.hip; soffset is not part of the range check) — the zero fill costs nothing.
// 0xFFFFFFFF marks lanes past the staged plane: no load at all.
typedef __amdgpu_buffer_rsrc_t rsrc_t;
constexpr int kMaxXG = 12;
constexpr int kXZero = (int)0x80000000, kXSkip = -1;

__device__ __forceinline__ rsrc_t make_rsrc(const void *p) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, 0x7FFFFFFF, 0x00020000);
}
__device__ __forceinline__ void bdma16(rsrc_t r, int voff, int soff, float *lds_wave_base) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void *)lds_wave_base, 16, voff, soff, 0, 0);
}
__device__ __forceinline__ void bdma4(rsrc_t r, int voff, int soff, float *lds_wave_base) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void *)lds_wave_base, 4, voff, soff, 0, 0);
}

__device__ __forceinline__ void build_xtab(int *tab, int nG, int vec, int up, int Ws, long imgStride, int N, int Hin,
                                           int Win, const PlaneGeom &g, int n0, int iy0, int ixa, int tid,
                                           int nthreads) {
    const int imgElems = g.inRows * g.pitch;
    const int planeElems = g.NI * imgElems;
    for (int idx = tid; idx < nG * 64; idx += nthreads) {
        const int gi = idx >> 6, ln = idx & 63;
        const int e = vec ? gi * 256 + ln * 4 : gi * 64 + ln;
        const int il = fast_div(e, g.invImg);
        const int rem = e - il * imgElems;
        const int r = fast_div(rem, g.invPitch);
        const int j = rem - r * g.pitch;
        const int n = n0 + il, iy = iy0 + r, ix = ixa + j;
        int v = kXSkip;
        if (e < planeElems) {
            const bool ok = n < N && (unsigned)iy < (unsigned)Hin && (unsigned)ix < (unsigned)Win;
            v = ok ? (int)(((long)il * imgStride + (up ? (long)(iy >> 1) * Ws + (ix >> 1) : (long)iy * Ws + ix)) * 4)
                   : kXZero;
        }
        tab[idx] = v;
    }
}

// nch channel planes, channel stride csBytes, of the tensor behind `r` (based at the first staged channel).
// A plain loop over the groups with the table entry read where it is used: a register array of entries costs
// more (initialisation, guarded reads, indexed register access) than the LDS latency it hides.
template <int NW>
__device__ __forceinline__ void stage_planes_buf(float *dst, rsrc_t r, int csBytes, int nch, const int *tab, int nG,
                                                 int vec, int cplane, int lw, int lane) {
    const int per = vec ? 256 : 64;
    for (int gi = 0; gi < nG; ++gi) {
        const int o = tab[gi * 64 + lane];
        const int cl0 = (lw + NW - (gi % NW)) % NW;
        float *d = dst + gi * per + cl0 * cplane;
        int soff = cl0 * csBytes;
        if (o != kXSkip) {
            for (int cl = cl0; cl < nch; cl += NW) {
                if (vec)
                    bdma16(r, o, soff, d);
                else
                    bdma4(r, o, soff, d);
                soff += NW * csBytes;
                d += NW * cplane;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// forward / backward-data MFMA kernel
// ------------------------------------------------------------------------------------------
struct ConvArgs {
    const float *x, *x2, *wT, *zeros, *bias, *scale, *shift, *res;
    float *y;
    int N, C1, C2, Hin, Win, up1, H1, W1;
    int Cout, Hout, Wout;
    int KH, KW, stride, dil, padT, padL;
    int act;
    int lgTC, lgTR;
    int tilesX, tilesY;
    int CK, R4, rows;
    int inRows, inCols, pitch, cplane, colOff;
    int ldw;
    int resH, resW, resOff;
    int vec4, vec1, vec2, up2, deal, xcdRemap;
    int xtab, nG1, nG2;      // table-driven input staging: groups per plane for source 1 / source 2
    float invImg, invPitch;
};

// Epilogue of one 16x16 accumulator tile: this lane holds output channel `co` for the 4 consecutive
// tile pixels pb..pb+3.  (+ residual) -> affine / bias -> activation -> store (optionally 2x upsampled).
__device__ __forceinline__ void store_tile(const ConvArgs &a, const f32x4 c, int pb, int co, int n0, int oy0, int ox0,
                                           int lgT, int TRm, int TCm) {
    if (co >= a.Cout) return;
    const long planeO = (long)a.Hout * a.Wout;
    float sc = 1.f, sh = 0.f;
    if (a.scale) {
        sc = a.scale[co];
        sh = a.shift[co];
    } else if (a.bias) {
        sh = a.bias[co];
    }
    const float cv[4] = {c[0], c[1], c[2], c[3]};
    float v[4];
    int n_[4], oy_[4], ox_[4];
    bool ok_[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int p = pb + j;
        const int il = p >> lgT, r = (p >> a.lgTC) & TRm, cc = p & TCm;
        n_[j] = n0 + il;
        oy_[j] = oy0 + r;
        ox_[j] = ox0 + cc;
        ok_[j] = n_[j] < a.N && oy_[j] < a.Hout && ox_[j] < a.Wout;
        float t = cv[j];
        if (a.res && ok_[j])
            t += a.res[(((long)n_[j] * a.Cout + co) * a.resH + oy_[j] + a.resOff) * a.resW + ox_[j] + a.resOff];
        v[j] = apply_act(t * sc + sh, a.act);
    }
    if (a.up2) {
        // fused nn.Upsample(2, nearest): every value is written to its 2x2 block of y[N,Cout,2H,2W]
        const long W2 = 2L * a.Wout;
        if (a.vec4) {
            if (ok_[0]) {
                float *q = a.y + ((long)n_[0] * a.Cout + co) * planeO * 4 + (long)(2 * oy_[0]) * W2 + 2 * ox_[0];
                const float4 lo = make_float4(v[0], v[0], v[1], v[1]), hi = make_float4(v[2], v[2], v[3], v[3]);
                *reinterpret_cast<float4 *>(q) = lo;
                *reinterpret_cast<float4 *>(q + 4) = hi;
                *reinterpret_cast<float4 *>(q + W2) = lo;
                *reinterpret_cast<float4 *>(q + W2 + 4) = hi;
            }
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (ok_[j]) {
                    float *q = a.y + ((long)n_[j] * a.Cout + co) * planeO * 4 + (long)(2 * oy_[j]) * W2 + 2 * ox_[j];
                    q[0] = v[j];
                    q[1] = v[j];
                    q[W2] = v[j];
                    q[W2 + 1] = v[j];
                }
        }
    } else if (a.vec4) {
        if (ok_[0])
            *reinterpret_cast<float4 *>(a.y + ((long)n_[0] * a.Cout + co) * planeO + (long)oy_[0] * a.Wout + ox_[0]) =
                make_float4(v[0], v[1], v[2], v[3]);
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (ok_[j]) a.y[((long)n_[j] * a.Cout + co) * planeO + (long)oy_[j] * a.Wout + ox_[j]] = v[j];
    }
}

}  // namespace
