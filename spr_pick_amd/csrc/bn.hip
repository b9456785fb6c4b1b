// BatchNorm2d (+ optional ReLU) for the detector head.
// Training mode: every channel's N*HW elements are split over `S` workgroups (C*S ~ 1024 so the
// whole chip works even for C = 1); pass A accumulates per-slice sums in fp64 (sum, sum of squares:
// one pass, no cancellation problem at fp64), pass B combines the S slices in a fixed order,
// normalises its slice and (slice 0) updates the running statistics.  Backward has the same shape.
// GROUPS: the batch may be `groups` independent passes stacked along N (the detector is called once on the patches
// and once on their flipped copies; the convolutions run once on the stacked batch).  Every group gets its own
// batch statistics (grid z = group) and the running averages are updated group after group, exactly as if the module
// had been called once per pass; the parameter gradients are the sums over the groups in group order.
#include "common.h"

namespace {

constexpr int kBlk = 256;

__device__ __forceinline__ double block_sum(double v, double *red) {
    red[threadIdx.x] = v;
    __syncthreads();
    for (int w = kBlk / 2; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
        __syncthreads();
    }
    const double r = red[0];
    __syncthreads();
    return r;
}

// slice s of channel c covers flattened (n, i) elements [s*per, min(M, (s+1)*per))
__device__ __forceinline__ long elem_index(long e, int c, int C, int HW) {
    const long n = e / HW, i = e - n * HW;
    return (n * C + c) * HW + i;
}

// grid (C, S): partial[(c*S + s)*2 + {0,1}] = sum a, sum b over the slice, where
// (a, b) = (x, x^2) for the forward and (g, g*(x-mean)*invstd) for the backward
template <bool BWD>
__global__ __launch_bounds__(kBlk) void bn_partial_kernel(const float *__restrict__ x, const float *__restrict__ gy,
                                                          const float *__restrict__ y,
                                                          const float *__restrict__ save_mean,
                                                          const float *__restrict__ save_invstd,
                                                          double *__restrict__ partial, int N, int C, int HW, long per,
                                                          int relu) {
    __shared__ double red[kBlk];
    const int c = blockIdx.x, s = blockIdx.y, S = gridDim.y, grp = blockIdx.z;
    const long M = (long)N * HW;                      // N: images per group
    const long gofs = (long)grp * N * C * HW;         // first element of this group
    x += gofs;
    if (BWD) gy += gofs, y = y ? y + gofs : y;
    partial += (long)grp * C * S * 2;
    const long lo = (long)s * per, hi = min(M, lo + per);
    const float mean = BWD ? save_mean[grp * C + c] : 0.f, invstd = BWD ? save_invstd[grp * C + c] : 0.f;
    double a = 0.0, b = 0.0;
    for (long e = lo + threadIdx.x; e < hi; e += kBlk) {
        const long idx = elem_index(e, c, C, HW);
        if (BWD) {
            float g = gy[idx];
            if (relu && !(y[idx] > 0.f)) g = 0.f;
            a += g;
            b += (double)g * (double)((x[idx] - mean) * invstd);
        } else {
            const float v = x[idx];
            a += v;
            b += (double)v * (double)v;
        }
    }
    const double ta = block_sum(a, red), tb = block_sum(b, red);
    if (threadIdx.x == 0) {
        partial[((long)c * S + s) * 2] = ta;
        partial[((long)c * S + s) * 2 + 1] = tb;
    }
}

__global__ __launch_bounds__(kBlk) void bn_train_apply_kernel(const float *__restrict__ x, float *__restrict__ y,
                                                              const float *__restrict__ gamma,
                                                              const float *__restrict__ beta, float *running_mean,
                                                              float *running_var, float *save_mean, float *save_invstd,
                                                              const double *__restrict__ partial, int N, int C, int HW,
                                                              long per, float momentum, float eps, int relu) {
    const int c = blockIdx.x, s = blockIdx.y, S = gridDim.y, grp = blockIdx.z, G = gridDim.z;
    const long M = (long)N * HW;
    auto stats = [&](int gi, double &mean_d, double &var_d) {
        const double *pp = partial + (long)gi * C * S * 2;
        double sa = 0.0, sb = 0.0;
        for (int i = 0; i < S; ++i) {
            sa += pp[((long)c * S + i) * 2];
            sb += pp[((long)c * S + i) * 2 + 1];
        }
        mean_d = sa / (double)M;
        var_d = sb / (double)M - mean_d * mean_d;
        if (var_d < 0.0) var_d = 0.0;
    };
    double mean_d, var_d;
    stats(grp, mean_d, var_d);
    const float mean = (float)mean_d, invstd = (float)(1.0 / sqrt(var_d + (double)eps));
    if (s == 0 && threadIdx.x == 0) {
        save_mean[grp * C + c] = mean;
        save_invstd[grp * C + c] = invstd;
        if (running_mean && grp == 0) {        // one thread per channel applies the groups' updates in group order
            float rm = running_mean[c], rv = running_var[c];
            for (int gi = 0; gi < G; ++gi) {
                double m2, v2;
                stats(gi, m2, v2);
                const double unbiased = M > 1 ? v2 * (double)M / (double)(M - 1) : v2;
                rm = (1.f - momentum) * rm + momentum * (float)m2;
                rv = (1.f - momentum) * rv + momentum * (float)unbiased;
            }
            running_mean[c] = rm;
            running_var[c] = rv;
        }
    }
    const long gofs = (long)grp * N * C * HW;
    x += gofs;
    y += gofs;
    const float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
    const long lo = (long)s * per, hi = min(M, lo + per);
    for (long e = lo + threadIdx.x; e < hi; e += kBlk) {
        const long idx = elem_index(e, c, C, HW);
        float v = (x[idx] - mean) * invstd * g + b;
        if (relu) v = v > 0.f ? v : 0.f;
        y[idx] = v;
    }
}

__global__ __launch_bounds__(kBlk) void bn_train_bwd_apply_kernel(const float *__restrict__ gy,
                                                                  const float *__restrict__ x,
                                                                  const float *__restrict__ y,
                                                                  const float *__restrict__ gamma,
                                                                  const float *__restrict__ save_mean,
                                                                  const float *__restrict__ save_invstd,
                                                                  float *__restrict__ gx, float *__restrict__ ggamma,
                                                                  float *__restrict__ gbeta,
                                                                  const double *__restrict__ partial, int N, int C,
                                                                  int HW, long per, int relu) {
    const int c = blockIdx.x, s = blockIdx.y, S = gridDim.y, grp = blockIdx.z, G = gridDim.z;
    const long M = (long)N * HW;
    auto sums = [&](int gi, float &sg, float &sgx) {
        const double *pp = partial + (long)gi * C * S * 2;
        double sa = 0.0, sb = 0.0;
        for (int i = 0; i < S; ++i) {
            sa += pp[((long)c * S + i) * 2];
            sb += pp[((long)c * S + i) * 2 + 1];
        }
        sg = (float)sa;
        sgx = (float)sb;
    };
    float sum_g, sum_gx;
    sums(grp, sum_g, sum_gx);
    if (s == 0 && grp == 0 && threadIdx.x == 0) {     // parameter gradients: sum over the groups in group order
        float tg = sum_g, tgx = sum_gx;
        for (int gi = 1; gi < G; ++gi) {
            float a2, b2;
            sums(gi, a2, b2);
            tg += a2;
            tgx += b2;
        }
        if (ggamma) ggamma[c] = tgx;
        if (gbeta) gbeta[c] = tg;
    }
    const long gofs = (long)grp * N * C * HW;
    gy += gofs;
    x += gofs;
    if (y) y += gofs;
    gx += gofs;
    const float mean = save_mean[grp * C + c], invstd = save_invstd[grp * C + c];
    const float k = (gamma ? gamma[c] : 1.f) * invstd, invM = 1.f / (float)M;
    const long lo = (long)s * per, hi = min(M, lo + per);
    for (long e = lo + threadIdx.x; e < hi; e += kBlk) {
        const long idx = elem_index(e, c, C, HW);
        float g = gy[idx];
        if (relu && !(y[idx] > 0.f)) g = 0.f;
        const float xhat = (x[idx] - mean) * invstd;
        gx[idx] = k * (g - sum_g * invM - xhat * sum_gx * invM);
    }
}

__global__ void bn_eval_fwd_kernel(const float *__restrict__ x, float *__restrict__ y, const float *__restrict__ gamma,
                                   const float *__restrict__ beta, const float *__restrict__ rm,
                                   const float *__restrict__ rv, long total, int C, int HW, float eps, int relu) {
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int c = (int)((e / HW) % C);
        const float invstd = rsqrtf(rv[c] + eps);
        float v = (x[e] - rm[c]) * invstd * (gamma ? gamma[c] : 1.f) + (beta ? beta[c] : 0.f);
        if (relu) v = v > 0.f ? v : 0.f;
        y[e] = v;
    }
}

int slices(int N, int C, int HW) {
    const long M = (long)N * HW;
    long S = 1;
    while ((long)C * S < 1024 && M / (S * 2) >= 2048) S *= 2;
    return (int)S;
}

}  // namespace

extern "C" {

size_t sprk_bn_ws_bytes(int N, int C, int HW, int groups) {
    if (groups < 1 || N % groups) return 0;
    return (size_t)groups * C * slices(N / groups, C, HW) * 2 * sizeof(double);
}

int sprk_bn_train_fwd(const float *x, float *y, const float *gamma, const float *beta, float *running_mean,
                      float *running_var, float *save_mean, float *save_invstd, int N, int C, int HW, int groups,
                      float momentum, float eps, int relu, void *ws, size_t ws_bytes, void *stream) {
    SPRK_REQUIRE(x && y && save_mean && save_invstd && N > 0 && C > 0 && HW > 0, "bn_train_fwd: bad arguments");
    SPRK_REQUIRE(groups >= 1 && groups <= 64 && N % groups == 0, "bn_train_fwd: the batch does not divide into the groups");
    SPRK_REQUIRE((running_mean == nullptr) == (running_var == nullptr), "bn_train_fwd: running stats mismatch");
    const int Ng = N / groups;
    const int S = slices(Ng, C, HW);
    if (!ws || ws_bytes < (size_t)groups * C * S * 2 * sizeof(double)) {
        sprk::set_error("bn_train_fwd: workspace too small");
        return SPRK_EWORKSPACE;
    }
    const long per = ((long)Ng * HW + S - 1) / S;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(bn_partial_kernel<false>, dim3(C, S, groups), dim3(kBlk), 0, s, x, (const float *)nullptr,
                       (const float *)nullptr, (const float *)nullptr, (const float *)nullptr, (double *)ws, Ng, C, HW,
                       per, 0);
    if (int rc = sprk::check_launch("bn_partial")) return rc;
    hipLaunchKernelGGL(bn_train_apply_kernel, dim3(C, S, groups), dim3(kBlk), 0, s, x, y, gamma, beta, running_mean,
                       running_var, save_mean, save_invstd, (const double *)ws, Ng, C, HW, per, momentum, eps, relu);
    return sprk::check_launch("bn_train_apply");
}

int sprk_bn_eval_fwd(const float *x, float *y, const float *gamma, const float *beta, const float *running_mean,
                     const float *running_var, int N, int C, int HW, float eps, int relu, void *stream) {
    SPRK_REQUIRE(x && y && running_mean && running_var && N > 0 && C > 0 && HW > 0, "bn_eval_fwd: bad arguments");
    const long total = (long)N * C * HW;
    hipLaunchKernelGGL(bn_eval_fwd_kernel, dim3(sprk::ew_blocks(total)), dim3(256), 0, (hipStream_t)stream, x, y, gamma,
                       beta, running_mean, running_var, total, C, HW, eps, relu);
    return sprk::check_launch("bn_eval_fwd");
}

int sprk_bn_train_bwd(const float *gy, const float *x, const float *y, const float *gamma, const float *save_mean,
                      const float *save_invstd, float *gx, float *ggamma, float *gbeta, int N, int C, int HW, int groups,
                      int relu, void *ws, size_t ws_bytes, void *stream) {
    SPRK_REQUIRE(gy && x && save_mean && save_invstd && gx && N > 0 && C > 0 && HW > 0, "bn_train_bwd: bad arguments");
    SPRK_REQUIRE(groups >= 1 && groups <= 64 && N % groups == 0, "bn_train_bwd: the batch does not divide into the groups");
    SPRK_REQUIRE(!relu || y, "bn_train_bwd: relu needs the saved output");
    const int Ng = N / groups;
    const int S = slices(Ng, C, HW);
    if (!ws || ws_bytes < (size_t)groups * C * S * 2 * sizeof(double)) {
        sprk::set_error("bn_train_bwd: workspace too small");
        return SPRK_EWORKSPACE;
    }
    const long per = ((long)Ng * HW + S - 1) / S;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(bn_partial_kernel<true>, dim3(C, S, groups), dim3(kBlk), 0, s, x, gy, y, save_mean, save_invstd,
                       (double *)ws, Ng, C, HW, per, relu);
    if (int rc = sprk::check_launch("bn_partial_bwd")) return rc;
    hipLaunchKernelGGL(bn_train_bwd_apply_kernel, dim3(C, S, groups), dim3(kBlk), 0, s, gy, x, y, gamma, save_mean,
                       save_invstd, gx, ggamma, gbeta, (const double *)ws, Ng, C, HW, per, relu);
    return sprk::check_launch("bn_train_bwd_apply");
}

}  // extern "C"
