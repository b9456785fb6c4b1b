// Does the sustained fp32 MFMA rate depend on the data (power / clock management)?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float rnd(unsigned x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return (float)(x & 0xFFFFFF) / 8388608.0f - 1.0f;   // [-1, 1)
}

// MODE 0: all MFMAs share one (a, b); 1: 4 x 6 distinct random operands, loop invariant;
// 2: distinct random operands refreshed every iteration (cheap VALU update)
template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, int iters) {
    f32x4 acc[24];
    for (int i = 0; i < 24; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float a[4], b[6];
    const unsigned t = threadIdx.x + blockIdx.x * 256;
    for (int i = 0; i < 4; ++i) a[i] = MODE == 0 ? 1.0f : rnd(t * 16 + i);
    for (int i = 0; i < 6; ++i) b[i] = MODE == 0 ? 0.5f : rnd(t * 16 + 8 + i);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int n = 0; n < 6; ++n)
                acc[m * 6 + n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m], b[n], acc[m * 6 + n], 0, 0, 0);
        if (MODE == 2) {
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i] = -a[i] * 0.999f + 0.001f * b[i];
#pragma unroll
            for (int i = 0; i < 6; ++i) b[i] = -b[i];
        }
    }
    float s = 0.f;
    for (int i = 0; i < 24; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 12345.678f) out[threadIdx.x] = s;
}

template <int MODE>
void run(const char *tag, int bpc, int iters) {
    float *out; (void)hipMalloc(&out, 4096);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int grid = 256 * bpc;
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, out, iters / 10);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, out, iters);
    (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double flops = 2.0 * 16 * 16 * 4 * 24.0 * iters * 4 * grid;
    printf("%-44s blocks/CU %d  %8.2f ms : %7.1f TFLOP/s\n", tag, bpc, ms, flops / ms / 1e9);
    (void)hipFree(out);
}

int main() {
    for (int rep = 0; rep < 2; ++rep) {
        run<0>("one shared (a,b), constants", 2, 20000);
        run<1>("4x6 random operands, loop invariant", 2, 20000);
        run<2>("4x6 random operands, sign-flipping each iter", 2, 20000);
        run<2>("same, 3 blocks/CU", 3, 20000);
        run<2>("same, long (sustained)", 2, 200000);
    }
    return 0;
}
