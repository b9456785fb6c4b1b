// BatchNorm2d (+ optional ReLU) for the detector head.  Training-mode tensors on this path are
// small ([B,C<=128,<=29,<=29]); one workgroup per channel walks its N*HW elements (L2 resident)
// three times: mean, centred variance, normalise — the same two-pass statistics torch uses.
#include "common.h"

namespace {

constexpr int kBlk = 256;

__device__ __forceinline__ float block_sum(float v, float *red) {
    red[threadIdx.x] = v;
    __syncthreads();
    for (int w = kBlk / 2; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
        __syncthreads();
    }
    const float r = red[0];
    __syncthreads();
    return r;
}

__global__ __launch_bounds__(kBlk) void bn_train_fwd_kernel(const float *__restrict__ x, float *__restrict__ y,
                                                            const float *__restrict__ gamma,
                                                            const float *__restrict__ beta, float *running_mean,
                                                            float *running_var, float *save_mean, float *save_invstd,
                                                            int N, int C, int HW, float momentum, float eps, int relu) {
    __shared__ float red[kBlk];
    const int c = blockIdx.x;
    const long M = (long)N * HW;
    float s = 0.f;
    for (long e = threadIdx.x; e < M; e += kBlk) {
        const long n = e / HW, i = e - n * HW;
        s += x[(n * C + c) * HW + i];
    }
    const float mean = block_sum(s, red) / (float)M;
    s = 0.f;
    for (long e = threadIdx.x; e < M; e += kBlk) {
        const long n = e / HW, i = e - n * HW;
        const float d = x[(n * C + c) * HW + i] - mean;
        s += d * d;
    }
    const float ss = block_sum(s, red);
    const float var = ss / (float)M;
    const float invstd = rsqrtf(var + eps);
    if (threadIdx.x == 0) {
        save_mean[c] = mean;
        save_invstd[c] = invstd;
        if (running_mean) {
            const float unbiased = M > 1 ? ss / (float)(M - 1) : var;
            running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mean;
            running_var[c] = (1.f - momentum) * running_var[c] + momentum * unbiased;
        }
    }
    const float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
    for (long e = threadIdx.x; e < M; e += kBlk) {
        const long n = e / HW, i = e - n * HW;
        const long idx = (n * C + c) * HW + i;
        float v = (x[idx] - mean) * invstd * g + b;
        if (relu) v = v > 0.f ? v : 0.f;
        y[idx] = v;
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

__global__ __launch_bounds__(kBlk) void bn_train_bwd_kernel(const float *__restrict__ gy, const float *__restrict__ x,
                                                            const float *__restrict__ y,
                                                            const float *__restrict__ gamma,
                                                            const float *__restrict__ save_mean,
                                                            const float *__restrict__ save_invstd,
                                                            float *__restrict__ gx, float *__restrict__ ggamma,
                                                            float *__restrict__ gbeta, int N, int C, int HW, int relu) {
    __shared__ float red[kBlk];
    const int c = blockIdx.x;
    const long M = (long)N * HW;
    const float mean = save_mean[c], invstd = save_invstd[c];
    float s1 = 0.f, s2 = 0.f;
    for (long e = threadIdx.x; e < M; e += kBlk) {
        const long n = e / HW, i = e - n * HW;
        const long idx = (n * C + c) * HW + i;
        float g = gy[idx];
        if (relu && !(y[idx] > 0.f)) g = 0.f;
        s1 += g;
        s2 += g * (x[idx] - mean) * invstd;
    }
    const float sum_g = block_sum(s1, red);
    const float sum_gx = block_sum(s2, red);
    if (threadIdx.x == 0) {
        if (ggamma) ggamma[c] = sum_gx;
        if (gbeta) gbeta[c] = sum_g;
    }
    const float gm = gamma ? gamma[c] : 1.f;
    const float k = gm * invstd, invM = 1.f / (float)M;
    for (long e = threadIdx.x; e < M; e += kBlk) {
        const long n = e / HW, i = e - n * HW;
        const long idx = (n * C + c) * HW + i;
        float g = gy[idx];
        if (relu && !(y[idx] > 0.f)) g = 0.f;
        const float xhat = (x[idx] - mean) * invstd;
        gx[idx] = k * (g - sum_g * invM - xhat * sum_gx * invM);
    }
}

}  // namespace

extern "C" {

int sprk_bn_train_fwd(const float *x, float *y, const float *gamma, const float *beta, float *running_mean,
                      float *running_var, float *save_mean, float *save_invstd, int N, int C, int HW, float momentum,
                      float eps, int relu, void *stream) {
    SPRK_REQUIRE(x && y && save_mean && save_invstd && N > 0 && C > 0 && HW > 0, "bn_train_fwd: bad arguments");
    SPRK_REQUIRE((running_mean == nullptr) == (running_var == nullptr), "bn_train_fwd: running stats mismatch");
    hipLaunchKernelGGL(bn_train_fwd_kernel, dim3(C), dim3(kBlk), 0, (hipStream_t)stream, x, y, gamma, beta,
                       running_mean, running_var, save_mean, save_invstd, N, C, HW, momentum, eps, relu);
    return sprk::check_launch("bn_train_fwd");
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
                      const float *save_invstd, float *gx, float *ggamma, float *gbeta, int N, int C, int HW, int relu,
                      void *stream) {
    SPRK_REQUIRE(gy && x && save_mean && save_invstd && gx && N > 0 && C > 0 && HW > 0, "bn_train_bwd: bad arguments");
    SPRK_REQUIRE(!relu || y, "bn_train_bwd: relu needs the saved output");
    hipLaunchKernelGGL(bn_train_bwd_kernel, dim3(C), dim3(kBlk), 0, (hipStream_t)stream, gy, x, y, gamma, save_mean,
                       save_invstd, gx, ggamma, gbeta, N, C, HW, relu);
    return sprk::check_launch("bn_train_bwd");
}

}  // extern "C"
