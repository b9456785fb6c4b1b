// Winograd F(2x2, 3x3) path of the 3x3 stride-1 convolutions (wino.hip); called from conv.hip.
#pragma once
#include <hip/hip_runtime.h>

namespace sprk {

struct WinoGeom {
    int N, C1, C2, Cout, H, W, Hout, Wout, KH, KW, stride, dil, padT, padL, up1, up2, res;
    int pin;   // SPRK_DT_PIN: the choice must not depend on the image count or the plane size
};
struct WinoArgs {
    const float *x, *x2;   // sources [N,C1,H,W], [N,C2,H,W] (x2 may be null)
    const float *w;        // taps in the layout of the forward layer, [CoutF][CinF][3][3]
    const float *bias, *scale, *shift;
    float *y;              // [N,Cout,H,W]
    float *U;              // workspace of wino_ws_bytes()
    int N, C1, C2, H, W, Cout, padT, padL, act;
    int mode;              // 0: forward taps w[cout][cin]; 1: backward-data taps w[k][cout] flipped
    int kclass;            // profiling class of the main kernel launch (sprk_prof_*)
    double flops;          // algorithmic (direct-convolution) FLOPs of this call, for the same
    int up2;               // 1: y is [N,Cout,2H,2W], every output written to its 2x2 block (fused nn.Upsample)
    const float *mask = nullptr;   // [N,Cout,H,W] or null: y *= d act / d (mask) (backward-data: the saved conv input)
    int mask_act = 0;              // SPRK_ACT_* of that mask
};

struct WinoWgArgs {
    const float *x, *x2, *gy;   // sources [N,C1,H,W], [N,C2,H,W] (x2 may be null), output gradient [N,Cout,H,W]
    float *gw;                  // [Cout][C1+C2][3][3]
    float *partial;             // workspace of wino_wgrad_ws_bytes()
    int N, C1, C2, H, W, Cout, padT, padL;
    int kclass;
    double flops;
};

bool wino_eligible(const WinoGeom &g);
bool wino_wgrad_eligible(const WinoGeom &g);
size_t wino_wgrad_ws_bytes(int C1, int C2, int Cout);
int wino_wgrad(const WinoWgArgs &a, hipStream_t s);
size_t wino_ws_bytes(int C1, int C2, int Cout);
int wino_conv(const WinoArgs &a, hipStream_t s);

}  // namespace sprk
