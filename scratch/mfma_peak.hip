// Achievable fp32 MFMA rate on this GPU: independent v_mfma_f32_16x16x4f32 streams, no memory traffic.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(256) void mfma_loop(float *out, int iters, float seed) {
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float a = seed + threadIdx.x, b = seed * 0.5f + threadIdx.x;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 12345.678f) out[threadIdx.x] = s;
}

template <int NACC>
void run(int blocks_per_cu, int waves, int iters) {
    float *out;
    hipMalloc(&out, 4096);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int grid = 256 * blocks_per_cu;
    hipLaunchKernelGGL(mfma_loop<NACC>, dim3(grid), dim3(64 * waves), 0, 0, out, iters, 1.0f);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(mfma_loop<NACC>, dim3(grid), dim3(64 * waves), 0, 0, out, iters, 1.0f);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flops = 2.0 * 16 * 16 * 4 * (double)NACC * iters * waves * grid;
    printf("NACC %2d  waves/block %d  blocks/CU %d : %8.3f ms  %7.1f TFLOP/s\n", NACC, waves, blocks_per_cu, ms, flops / ms / 1e9);
    hipFree(out);
}

int main() {
    const int iters = 20000;
    run<4>(1, 4, iters); run<8>(1, 4, iters); run<16>(1, 4, iters); run<24>(1, 4, iters);
    run<4>(2, 4, iters); run<8>(2, 4, iters); run<16>(2, 4, iters);
    run<8>(3, 4, iters); run<24>(3, 4, iters);
    run<8>(1, 8, iters); run<16>(1, 8, iters);
    run<16>(4, 4, iters);
    // long run: sustained clocks
    run<16>(2, 4, iters * 20);
    return 0;
}
