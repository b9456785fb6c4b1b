// 16-bit-operand backward-weight of the 3x3 stride-1 layers (wgrad16.hip); called from conv.hip when the geometry
// asks for 16-bit operands (sprk_conv_geom.dtype) and the layer is eligible.  Not part of the ABI.
#pragma once
#include <hip/hip_runtime.h>

#include "common.h"

namespace sprk {

struct Wgrad16Call {
    int dtype;     // SPRK_DT_BF16 | SPRK_DT_F16 (| SPRK_DT_FORCE)
    int x16;       // x, x2 and gy are 16-bit tensors of that type (SPRK_DT_X16)
    int N, C1, C2, H, W, Cout, Hout, Wout, KH, KW, stride, dil, padT, padL, up1;
    int kclass;
    double flops;
};

bool wgrad16_eligible(const Wgrad16Call &c);
size_t wgrad16_ws_bytes(const Wgrad16Call &c);
// item: see sprk_conv2d_bwd_weight_partial (nullptr = add the workgroups' partial dW now)
int wgrad16_run(const Wgrad16Call &c, const void *x, const void *x2, const void *gy, float *gw, void *ws,
                size_t ws_bytes, sprk_reduce_item *item, hipStream_t s);
long wgrad16_launches();

}  // namespace sprk
