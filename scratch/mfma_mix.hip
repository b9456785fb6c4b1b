// How much do LDS reads / VALU ops between MFMAs cost?  24 MFMAs per iteration plus a configurable mix.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NDS, int NVALU, bool DEP>
__global__ __launch_bounds__(256) void mix_loop(float *out, int iters, int stride) {
    __shared__ float lds[8192];
    for (int i = threadIdx.x; i < 8192; i += 256) lds[i] = (float)i;
    __syncthreads();
    f32x4 acc[24];
    for (int i = 0; i < 24; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float a[4] = {1.f, 2.f, 3.f, 4.f}, b[6] = {1.f, 2.f, 3.f, 4.f, 5.f, 6.f};
    int addr = threadIdx.x & 63;
    int vv = threadIdx.x;
    for (int it = 0; it < iters; ++it) {
        float na[4], nb[6];
#pragma unroll
        for (int i = 0; i < 4; ++i) na[i] = a[i];
#pragma unroll
        for (int i = 0; i < 6; ++i) nb[i] = b[i];
#pragma unroll
        for (int i = 0; i < NDS; ++i) {
            const float v = lds[(addr + i * 64 + it * stride) & 8191];
            if (i < 4) na[i] = v; else nb[(i - 4) % 6] = v;
        }
#pragma unroll
        for (int i = 0; i < NVALU; ++i) vv = vv * 3 + i;
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int n = 0; n < 6; ++n)
                acc[m * 6 + n] = __builtin_amdgcn_mfma_f32_16x16x4f32(DEP ? na[m] : a[m], DEP ? nb[n] : b[n], acc[m * 6 + n], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (!DEP) {
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i] += na[i] * 1e-30f;
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i] = na[i];
#pragma unroll
            for (int i = 0; i < 6; ++i) b[i] = nb[i];
        }
    }
    float s = (float)vv;
    for (int i = 0; i < 24; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 12345.678f) out[threadIdx.x] = s;
}

template <int NDS, int NVALU, bool DEP>
void run(const char *tag, int bpc) {
    float *out; (void)hipMalloc(&out, 4096);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 4000, grid = 256 * bpc;
    hipLaunchKernelGGL((mix_loop<NDS, NVALU, DEP>), dim3(grid), dim3(256), 0, 0, out, iters, 1);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL((mix_loop<NDS, NVALU, DEP>), dim3(grid), dim3(256), 0, 0, out, iters, 1);
    (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double flops = 2.0 * 16 * 16 * 4 * 24.0 * iters * 4 * grid;
    printf("%-34s blocks/CU %d : %7.1f TFLOP/s\n", tag, bpc, flops / ms / 1e9);
    (void)hipFree(out);
}

int main() {
    for (int bpc = 1; bpc <= 3; ++bpc) {
        run<0, 0, false>("24 mfma only", bpc);
        run<10, 0, false>("+10 ds_read (results unused by mfma)", bpc);
        run<10, 0, true>("+10 ds_read feeding next mfma", bpc);
        run<0, 10, false>("+10 valu", bpc);
        run<10, 6, true>("+10 ds_read(dep) +6 valu", bpc);
        run<20, 0, true>("+20 ds_read(dep)", bpc);
    }
    return 0;
}
