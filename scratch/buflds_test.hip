// Semantics check of buffer_load ... lds on gfx950: out-of-range lanes must write zeros to LDS; soffset is
// added to the address but not range-checked.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((address_space(3))) void lds_void;

__global__ void k(const float *x, float *y, int soff_bytes, int nrec) {
    __shared__ float lds[512];
    for (int i = threadIdx.x; i < 512; i += 64) lds[i] = -7.f;
    __syncthreads();
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void *)x, 0, nrec, 0x00020000);
    const int lane = threadIdx.x;
    int voff = lane * 16;
    if (lane >= 40 && lane < 50) voff = 0x80000000;     // marked invalid
    if (lane >= 50 && lane < 56) voff = nrec + lane * 16;  // beyond num_records
    if (lane >= 56) voff = -16 * (lane - 55);            // negative
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void *)lds, 16, voff, soff_bytes, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void *)(lds + 256), 4, voff, soff_bytes, 0, 0);
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    for (int i = threadIdx.x; i < 512; i += 64) y[i] = lds[i];
}

int main() {
    const int n = 1 << 16;
    std::vector<float> h(n);
    for (int i = 0; i < n; ++i) h[i] = (float)i;
    float *x, *y;
    (void)hipMalloc(&x, n * 4); (void)hipMalloc(&y, 512 * 4);
    (void)hipMemcpy(x, h.data(), n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, x, y, 4096 * 4, 0x7fffffff);
    std::vector<float> o(512);
    (void)hipMemcpy(o.data(), y, 512 * 4, hipMemcpyDeviceToHost);
    printf("dwordx4: lane0 %g %g %g %g | lane39 %g | lane40 %g %g | lane49 %g | lane50 %g | lane55 %g | lane56 %g | lane63 %g\n", o[0], o[1], o[2], o[3],
           o[39 * 4], o[40 * 4], o[40 * 4 + 3], o[49 * 4], o[50 * 4], o[55 * 4], o[56 * 4], o[63 * 4]);
    printf("dword  : lane0 %g lane1 %g | lane39 %g | lane40 %g | lane50 %g | lane56 %g\n", o[256], o[257], o[256 + 39], o[256 + 40], o[256 + 50], o[256 + 56]);
    return 0;
}
