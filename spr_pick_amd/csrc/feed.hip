// Training-patch feed: micrographs stay resident in HBM (uint8 or float32, packed back to back) and
// each step's batch of crops is cut out on the device.  Replaces the reference's per-item PIL path
// (datasets/micrograph.py:60-122: img.crop((x-P/2, y-P/2, x+P/2, y+P/2)) with zero fill outside the
// image, RandomHorizontalFlip, to_tensor (uint8 -> float32 / 255), then the CWH -> CHW permute that
// transposes the patch) + 4-worker DataLoader collation.
#include "common.h"

namespace {

// grid (tiles, B); one 32x32 tile per block.  Source rows (along image x) are read coalesced,
// transposed through LDS, and written coalesced along the output's inner axis (image y).
//   out[b, 0, u, v] = src_b[y - P/2 + v][x - P/2 + (flip ? P-1-u : u)]   (0 outside the image)
template <typename T>
__global__ __launch_bounds__(256) void gather_patches_kernel(const T *__restrict__ mics,
                                                             const long *__restrict__ offsets,
                                                             const int *__restrict__ dims,
                                                             const int *__restrict__ items,
                                                             float *__restrict__ out, int P, int n_mics) {
    __shared__ float tile[32][33];
    const int b = blockIdx.y;
    const int tpr = P / 32;
    const int tu = (blockIdx.x % tpr) * 32, tv = (blockIdx.x / tpr) * 32;
    const int img = items[b * 4], x = items[b * 4 + 1], y = items[b * 4 + 2], flip = items[b * 4 + 3];
    const bool known = img >= 0 && img < n_mics;  // an unknown image index yields a zero patch
    const int rows = known ? dims[img * 2] : 0, cols = known ? dims[img * 2 + 1] : 0;
    const T *src = mics + (known ? offsets[img] : 0);
    const int lx = threadIdx.x & 31, ly = threadIdx.x >> 5;
    for (int j = ly; j < 32; j += 8) {
        const int v = tv + j, u = tu + lx;
        const int uu = flip ? P - 1 - u : u;
        const int sy = y - P / 2 + v, sx = x - P / 2 + uu;
        float val = 0.f;
        if (sy >= 0 && sy < rows && sx >= 0 && sx < cols) {
            if constexpr (sizeof(T) == 1)
                val = (float)src[(long)sy * cols + sx] / 255.0f;
            else
                val = (float)src[(long)sy * cols + sx];
        }
        tile[j][lx] = val;  // tile[v][u]
    }
    __syncthreads();
    float *o = out + (long)b * P * P;
    for (int j = ly; j < 32; j += 8) {
        const int u = tu + j, v = tv + lx;
        o[(long)u * P + v] = tile[lx][j];
    }
}

}  // namespace

extern "C" {

int sprk_gather_patches(const void *mics, int dtype, const long *offsets, const int *dims, const int *items,
                        float *out, int n_mics, int B, int P, void *stream) {
    SPRK_REQUIRE(mics && offsets && dims && items && out, "gather_patches: null pointer");
    SPRK_REQUIRE(n_mics > 0 && B > 0 && B < 65536 && P > 0 && P % 32 == 0, "gather_patches: need 0 < B < 65536 and P %% 32 == 0");
    SPRK_REQUIRE(dtype == SPRK_MIC_U8 || dtype == SPRK_MIC_F32, "gather_patches: dtype must be SPRK_MIC_U8/F32");
    const dim3 grid((P / 32) * (P / 32), B);
    if (dtype == SPRK_MIC_U8)
        hipLaunchKernelGGL(gather_patches_kernel<unsigned char>, grid, dim3(256), 0, (hipStream_t)stream,
                           (const unsigned char *)mics, offsets, dims, items, out, P, n_mics);
    else
        hipLaunchKernelGGL(gather_patches_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream,
                           (const float *)mics, offsets, dims, items, out, P, n_mics);
    return sprk::check_launch("gather_patches");
}

}  // extern "C"
