// 16-bit-operand (bf16 / fp16) MFMA convolution path (conv16.hip); called from conv.hip when the geometry asks
// for it (sprk_conv_geom.dtype) and the layer is eligible.  Not part of the ABI.
#pragma once
#include <hip/hip_runtime.h>

#include "common.h"

namespace sprk {

// One forward-shaped call: y[N,Cout,Hout,Wout] = act(conv(cat(x[N,C1], x2[N,C2]), taps) + bias / affine).
// mode 0: w is the layer's [Cout][C1+C2][KH][KW]; mode 1 (backward-data): w is the FORWARD layer's
// [C1][Cout][KH][KW] (this call's k channels are its output channels), taps flipped.
struct Conv16Call {
    int dtype;     // SPRK_DT_BF16 | SPRK_DT_F16
    int x16, y16;  // the input tensors (x, x2) / the output tensor are 16-bit tensors of that type (SPRK_DT_X16 / _Y16)
    int mode;
    int N, C1, C2, Hin, Win, Cout, Hout, Wout, KH, KW, stride, dil, padT, padL, up1, up2, res, act;
    const float *bias, *scale, *shift;
    int kclass;    // sprk_prof_* class
    double flops;
};

bool conv16_eligible(const Conv16Call &c);
// 0 = no 16-bit kernel, 1 = conv16_mfma_kernel (fp32 storage only), 2 = conv16_tile_kernel, 3 = conv16_head_kernel
int conv16_kind(const Conv16Call &c);
size_t conv16_ws_bytes(const Conv16Call &c);
// x, x2, y: fp32 tensors, or 16-bit tensors of the operand type where c.x16 / c.y16 say so
int conv16_run(const Conv16Call &c, const void *x, const void *x2, const float *w, void *y, void *ws,
               size_t ws_bytes, hipStream_t s);
long conv16_launches();

}  // namespace sprk
