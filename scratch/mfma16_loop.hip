// Micro-benchmark of conv16.hip's k-loop: LDS reads (8 x ds_read_b32 per A fragment + ds_read_b128 per B fragment),
// v_cvt_pk_bf16_f32, 24 x v_mfma_f32_16x16x32_bf16 per k-step.  Variants by argv[1] bitmask:
//   1: no conversion (A read as one b128)   2: no B reads (registers)   4: no A reads   8: one wave per SIMD (4 waves/CU)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(3))) float *lds_cfp;
typedef const __attribute__((address_space(3))) bf16x8 *lds_v8p;
__device__ __forceinline__ lds_cfp lds_f(int a) { return (lds_cfp)(__SIZE_TYPE__)(unsigned)a; }
__device__ __forceinline__ int lds_addr(const void *p) { return (int)(unsigned)(__SIZE_TYPE__)(const __attribute__((address_space(3))) char *)p; }

template <int VAR>
__global__ __launch_bounds__(256, 2) void k(float *out, int iters, int cplane, int pitch) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, lq = lane >> 4;
    for (int i = tid; i < 14000; i += 256) smem[i] = (float)((i * 7 + 3) % 13) * 0.125f;
    __syncthreads();
    f32x4 acc[4][6];
    for (int m = 0; m < 4; ++m) for (int n = 0; n < 6; ++n) acc[m][n] = f32x4{0, 0, 0, 0};
    const int abase = lds_addr(smem) + ((wave * pitch) + l15) * 4;
    const int baddr = lds_addr(smem + 8 * cplane) + (lq * 96 + l15) * 16;
    bf16x8 breg[6];
    for (int n = 0; n < 6; ++n) breg[n] = *(lds_v8p)(__SIZE_TYPE__)(unsigned)(baddr + n * 256);
    for (int it = 0; it < iters; ++it) {
        for (int ks = 0; ks < 3; ++ks) {
            const int goff = ((ks * 4 + lq) % 9 / 3 * pitch + (ks * 4 + lq) % 3) * 4, gstr = cplane * 4;
            bf16x8 av[4];
            if (VAR & 4) {
                for (int t = 0; t < 4; ++t) av[t] = breg[t];
            } else if (VAR & 1) {
                for (int t = 0; t < 4; ++t) av[t] = *(lds_v8p)(__SIZE_TYPE__)(unsigned)(baddr + (t + ks) * 256);
            } else {
                int ad = abase + goff;
                float f[4][8];
                for (int j = 0; j < 8; ++j) {
                    lds_cfp ap = lds_f(ad);
                    for (int t = 0; t < 4; ++t) f[t][j] = ap[16 * t];
                    ad += gstr;
                }
                for (int t = 0; t < 4; ++t) for (int j = 0; j < 8; ++j) av[t][j] = (__bf16)f[t][j];
            }
            bf16x8 bv[6];
            for (int n = 0; n < 6; ++n) bv[n] = (VAR & 2) ? breg[n] : *(lds_v8p)(__SIZE_TYPE__)(unsigned)(baddr + ks * 6144 + n * 256);
            for (int m = 0; m < 4; ++m) for (int n = 0; n < 6; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[m], bv[n], acc[m][n], 0, 0, 0);
        }
        if (!(VAR & 16)) __syncthreads();
    }
    float s = 0;
    for (int m = 0; m < 4; ++m) for (int n = 0; n < 6; ++n) s += acc[m][n][0] + acc[m][n][3];
    out[blockIdx.x * 256 + tid] = s;
}

int main(int argc, char **argv) {
    const int var = argc > 1 ? atoi(argv[1]) : 0;
    float *out; hipMalloc(&out, 4096 * 256 * 4);
    const int iters = 12 * 8, blocks = (var & 8) ? 256 : 512;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    auto run = [&](auto kern) {
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(a);
            hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 64000, 0, out, iters, 432, 72);
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            double mf = (double)blocks * 4 * iters * 3 * 24;
            printf("var %d: %.1f us, %.1f cycles@2.4GHz per k-step per wave, %.0f TF\n", var, ms * 1e3, ms * 1e-3 * 2.4e9 / (iters * 3), mf * 16384 / ms / 1e9);
        }
    };
    switch (var & 23) {
        case 0: run(k<0>); break; case 1: run(k<1>); break; case 2: run(k<2>); break; case 3: run(k<3>); break;
        case 4: run(k<4>); break; case 6: run(k<6>); break; case 16: run(k<16>); break; case 22: run(k<22>); break;
        default: run(k<0>);
    }
    return 0;
}
