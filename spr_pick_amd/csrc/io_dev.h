// Storage types of activation tensors (round 4): fp32, or — in the 16-bit modes — bf16 / fp16 tensors in HBM.
// The HBM-bound plumbing kernels (activation backward, pooling, un-rotation) are templates over the element type of
// each tensor they touch; the arithmetic is fp32 in registers either way (a 16-bit tensor is rounded once, at the store).
// `io` arguments of the C ABI pack one SPRK_DT_* code (0 fp32, 1 bf16, 2 fp16) per tensor, 4 bits each (SPRK_IO*).
#pragma once
#include "common.h"

namespace {

template <typename T>
struct IO;

template <>
struct IO<float> {
    static constexpr int code = 0;
    static __device__ __forceinline__ float ld(const float *p) { return *p; }
    static __device__ __forceinline__ void st(float *p, float v) { *p = v; }
    static __device__ __forceinline__ float2 ld2(const float *p) { return *reinterpret_cast<const float2 *>(p); }
    static __device__ __forceinline__ void st2(float *p, float2 v) { *reinterpret_cast<float2 *>(p) = v; }
    static __device__ __forceinline__ float4 ld4(const float *p) { return *reinterpret_cast<const float4 *>(p); }
    static __device__ __forceinline__ void st4(float *p, float4 v) { *reinterpret_cast<float4 *>(p) = v; }
};

template <typename T>
struct IO16 {
    typedef T t2 __attribute__((ext_vector_type(2)));
    typedef T t4 __attribute__((ext_vector_type(4)));
    static __device__ __forceinline__ float ld(const T *p) { return (float)*p; }
    static __device__ __forceinline__ void st(T *p, float v) { *p = (T)v; }
    static __device__ __forceinline__ float2 ld2(const T *p) {
        const t2 v = *reinterpret_cast<const t2 *>(p);
        return make_float2((float)v[0], (float)v[1]);
    }
    static __device__ __forceinline__ void st2(T *p, float2 v) {
        t2 o;
        o[0] = (T)v.x; o[1] = (T)v.y;
        *reinterpret_cast<t2 *>(p) = o;
    }
    static __device__ __forceinline__ float4 ld4(const T *p) {
        const t4 v = *reinterpret_cast<const t4 *>(p);
        return make_float4((float)v[0], (float)v[1], (float)v[2], (float)v[3]);
    }
    static __device__ __forceinline__ void st4(T *p, float4 v) {
        t4 o;
        o[0] = (T)v.x; o[1] = (T)v.y; o[2] = (T)v.z; o[3] = (T)v.w;
        *reinterpret_cast<t4 *>(p) = o;
    }
};
template <>
struct IO<__bf16> : IO16<__bf16> {
    static constexpr int code = 1;
};
template <>
struct IO<_Float16> : IO16<_Float16> {
    static constexpr int code = 2;
};

// io codes -> element types: every 16-bit tensor of one call has the same type (all bf16 or all fp16), each tensor is
// that type or fp32.  f(tag0, tag1, tag2) is called with value-initialised objects of the three element types.
template <typename F>
int dispatch_io3(int io, F &&f) {
    const int c0 = io & 15, c1 = (io >> 4) & 15, c2 = (io >> 8) & 15;
    int t16 = 0;
    for (int c : {c0, c1, c2}) {
        if (c < 0 || c > 2) return -1;
        if (c) {
            if (t16 && t16 != c) return -1;
            t16 = c;
        }
    }
    auto with = [&](auto h) {
        using H = decltype(h);
        const int m = (c0 ? 1 : 0) | (c1 ? 2 : 0) | (c2 ? 4 : 0);
        switch (m) {
            case 0: return f(float{}, float{}, float{});
            case 1: return f(H{}, float{}, float{});
            case 2: return f(float{}, H{}, float{});
            case 3: return f(H{}, H{}, float{});
            case 4: return f(float{}, float{}, H{});
            case 5: return f(H{}, float{}, H{});
            case 6: return f(float{}, H{}, H{});
            default: return f(H{}, H{}, H{});
        }
    };
    return t16 == 2 ? with(_Float16{}) : with(__bf16{});
}
template <typename F>
int dispatch_io2(int io, F &&f) {
    return dispatch_io3(io & 0xff, [&](auto a, auto b, auto) { return f(a, b); });
}

}  // namespace
