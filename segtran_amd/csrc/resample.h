// resample.h -- the source-coordinate rule of F.interpolate(mode='bilinear'|'trilinear', align_corners=False), shared by the FPN
// resampling kernels (fpn.hip) and the sliding-window inference kernels (infer.hip).
#pragma once
#include "common.h"

namespace segx {

// Source coordinate of destination index d along one axis, exactly as ATen:
//   src = max(scale * (d + 0.5) - 0.5, 0),  scale = n_in / n_out (float);  i0 = floor(src), i1 = min(i0+1, n_in-1), l = src - i0
// scale < 0 selects align_corners=True with |scale| = (n_in - 1) / (n_out - 1):  src = |scale| * d
struct Axis { int i0, i1; float l; };
__device__ __forceinline__ Axis axis_src(int d, int n_in, float scale) {
    float src = scale < 0.f ? -scale * (float)d : scale * ((float)d + 0.5f) - 0.5f;
    src = src < 0.f ? 0.f : src;
    Axis a; a.i0 = (int)src; if (a.i0 > n_in - 1) a.i0 = n_in - 1;
    a.i1 = a.i0 + (a.i0 < n_in - 1 ? 1 : 0); a.l = src - (float)a.i0;
    return a;
}
struct InterpDims { int d, h, w, D, H, W; float sd, sh, sw; };
static inline InterpDims make_dims(int d, int h, int w, int D, int H, int W) {
    InterpDims q; q.d = d; q.h = h; q.w = w; q.D = D; q.H = H; q.W = W;
    q.sd = (float)d / (float)D; q.sh = (float)h / (float)H; q.sw = (float)w / (float)W;
    return q;
}
// value of the resampled plane s[d][h][w] at destination (z, y, x), blended in ATen's order (x, then y, then z)
__device__ __forceinline__ float interp_at(const float* __restrict__ s, const InterpDims& q, int z, int y, int x) {
    const Axis az = axis_src(z, q.d, q.sd), ay = axis_src(y, q.h, q.sh), ax = axis_src(x, q.w, q.sw);
    auto at = [&](int zz, int yy, int xx) { return s[((int64_t)zz * q.h + yy) * q.w + xx]; };
    const float c00 = at(az.i0, ay.i0, ax.i0) * (1.f - ax.l) + at(az.i0, ay.i0, ax.i1) * ax.l;
    const float c01 = at(az.i0, ay.i1, ax.i0) * (1.f - ax.l) + at(az.i0, ay.i1, ax.i1) * ax.l;
    const float c10 = at(az.i1, ay.i0, ax.i0) * (1.f - ax.l) + at(az.i1, ay.i0, ax.i1) * ax.l;
    const float c11 = at(az.i1, ay.i1, ax.i0) * (1.f - ax.l) + at(az.i1, ay.i1, ax.i1) * ax.l;
    return (c00 * (1.f - ay.l) + c01 * ay.l) * (1.f - az.l) + (c10 * (1.f - ay.l) + c11 * ay.l) * az.l;
}

}  // namespace segx
