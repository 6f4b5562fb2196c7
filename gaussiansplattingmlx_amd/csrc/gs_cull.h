// gs_cull.h -- the staging cull of the blend kernels: which list entries cannot move any pixel of a rectangle
#pragma once
#include <hip/hip_runtime.h>

namespace gs {

// Cull bound on the quadratic form q: an entry is dropped for a wave when exp(-q/2) < 2^(-0.7213 * 40) = 2^-28.9 on
// every one of its pixels.  Image distance to the float32 oracle on the raw bench scene (L-inf / rms), forward ms:
//   q > 58: 4.34e-5 / 4.48e-7, 0.197   q > 48: same, 0.190   q > 40: same, 0.184   q > 34: 4.34e-5 / 4.49e-7, 0.180
//   q > 28: 6.95e-5 / 8.34e-7, 0.175   (tools/full_size_parity.py with -DGS_CULL_QMIN=...)
#ifndef GS_CULL_QMIN
#define GS_CULL_QMIN 40.0f
#endif
constexpr float CULL_QMIN = GS_CULL_QMIN;


// Minimum over the rectangle [X0,X1] x [Y0,Y1] (coordinates relative to the mean) of the splat's quadratic form
//   q(dx, dy) = c00 dx^2 + (c01 + c10) dx dy + c11 dy^2.
// For a positive definite form the minimum is 0 if the mean is inside, else it sits on an edge, where q is a 1-D
// parabola with a clamped closed-form minimiser.  Anything else (not positive definite) returns 0: never culled.
__device__ __forceinline__ float rect_min_q(float c00, float c01, float c10, float c11, float X0, float X1, float Y0,
                                            float Y1)
{
    const float b = 0.5f * (c01 + c10);
    if (!(c00 > 0.0f && c11 > 0.0f && c00 * c11 > b * b)) return 0.0f;
    if (X0 <= 0.0f && X1 >= 0.0f && Y0 <= 0.0f && Y1 >= 0.0f) return 0.0f;
    const float ib = -b / c11, ia = -b / c00;
    auto edge_x = [&](float X) {
        const float dy = fminf(fmaxf(ib * X, Y0), Y1);
        return c00 * X * X + 2.0f * b * X * dy + c11 * dy * dy;
    };
    auto edge_y = [&](float Y) {
        const float dx = fminf(fmaxf(ia * Y, X0), X1);
        return c00 * dx * dx + 2.0f * b * dx * Y + c11 * Y * Y;
    };
    return fminf(fminf(edge_x(X0), edge_x(X1)), fminf(edge_y(Y0), edge_y(Y1)));
}


}  // namespace gs
