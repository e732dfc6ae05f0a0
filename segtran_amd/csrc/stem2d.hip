// stem2d.hip -- the dense 3 x 3 stem of EfficientNet (efficientnet/model.py:128, 163: Conv2dStaticSamePadding(3, 48, 3, stride 2) on the 1024 x 1024 up-sized image) as
// a DIRECT convolution (round 6).  With three input channels the contraction is 27 deep: as an implicit GEMM (conv3d.hip, depth 1) it gathered 27 stride-2 dwords per
// position and k-tile and ran at 20 TFLOP/s forward (205 us) and 11 TFLOP/s for the weight gradient (366 us) against the 75 us its 0.38 GB take at the HBM roof.
//   forward : thread = one output position; its 27 inputs are loaded once into registers, the 48 x 27 filter taps are wave-uniform (scalar loads, one SGPR operand
//             per FMA), 48 coalesced stores -- 4.1 GFLOP of plain fp32 FMAs in the shadow of 0.38 GB of traffic;
//   dW      : the input windows are laid out as an im2col matrix Xcol [B][rows][OH OW] (rows = 27 padded to 28: 176 MB, written once by im2col_kernel) and
//             dW = sum_b dY_b Xcol_b^T is the batch-reduced skinny product gemm_skinny.hip streams at ~4.3 TB/s (48 + 28 rows).
// The input needs no gradient (network input); a stem whose input does takes the implicit-GEMM path as before.
#include "common.h"

namespace segx {

template <int CIN, int K>
__global__ __launch_bounds__(256) void conv2d_stem_fwd_kernel(const float* __restrict__ X, const float* __restrict__ Wt, float* __restrict__ Y, int Cout, int H, int W,
                                                              int OH, int OW, int st, int pt, int pl) {
    constexpr int T = CIN * K * K;
    const int64_t P = (int64_t)OH * OW;
    const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int b = blockIdx.y;
    const bool live = p < P;
    const int64_t pc = live ? p : P - 1;
    const int oy = (int)(pc / OW), ox = (int)(pc - (int64_t)oy * OW);
    float v[T];
#pragma unroll
    for (int ci = 0; ci < CIN; ++ci)
#pragma unroll
        for (int ky = 0; ky < K; ++ky)
#pragma unroll
            for (int kx = 0; kx < K; ++kx) {
                const int iy = oy * st + ky - pt, ix = ox * st + kx - pl;
                const bool ok = iy >= 0 && iy < H && ix >= 0 && ix < W;
                const float q = X[((int64_t)(b * CIN + ci) * H + min(max(iy, 0), H - 1)) * W + min(max(ix, 0), W - 1)];
                v[(ci * K + ky) * K + kx] = ok ? q : 0.f;
            }
    float* y = Y + (int64_t)b * Cout * P + pc;
    for (int co = 0; co < Cout; ++co) {
        const float* w = Wt + (int64_t)co * T;              // wave-uniform: scalar loads
        float a = 0.f;
#pragma unroll
        for (int t = 0; t < T; ++t) a += w[t] * v[t];
        if (live) y[(int64_t)co * P] = a;
    }
}

// Xcol[b][t][p] = x[b][ci][oy st + ky - pt][ox st + kx - pl] (0 outside the image), t = (ci K + ky) K + kx; rows t >= CIN K K are zero
template <int CIN, int K>
__global__ __launch_bounds__(256) void conv2d_stem_im2col_kernel(const float* __restrict__ X, float* __restrict__ Xcol, int H, int W, int OH, int OW, int st, int pt, int pl,
                                                                 int rows) {
    constexpr int T = CIN * K * K;
    const int64_t P = (int64_t)OH * OW;
    const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int b = blockIdx.y;
    if (p >= P) return;
    const int oy = (int)(p / OW), ox = (int)(p - (int64_t)oy * OW);
    float* o = Xcol + (int64_t)b * rows * P + p;
#pragma unroll
    for (int ci = 0; ci < CIN; ++ci)
#pragma unroll
        for (int ky = 0; ky < K; ++ky)
#pragma unroll
            for (int kx = 0; kx < K; ++kx) {
                const int iy = oy * st + ky - pt, ix = ox * st + kx - pl;
                const bool ok = iy >= 0 && iy < H && ix >= 0 && ix < W;
                const float q = X[((int64_t)(b * CIN + ci) * H + min(max(iy, 0), H - 1)) * W + min(max(ix, 0), W - 1)];
                o[(int64_t)((ci * K + ky) * K + kx) * P] = ok ? q : 0.f;
            }
    for (int t = T; t < rows; ++t) o[(int64_t)t * P] = 0.f;
}

}  // namespace segx

using namespace segx;

static bool stem2d_ok(int B, int Cin, int Cout, int H, int W, int OH, int OW, int K, int stride, int pt, int pl) {
    return B > 0 && B <= 65535 && Cin == 3 && K == 3 && Cout > 0 && H > 0 && W > 0 && OH > 0 && OW > 0 && stride >= 1 && stride <= 2 && pt >= 0 && pl >= 0 &&
           (int64_t)(OH - 1) * stride + K - pt <= H + K && (int64_t)(OW - 1) * stride + K - pl <= W + K;
}
extern "C" int segx_conv2d_stem_fwd(const float* X, const float* Wt, float* Y, int B, int Cin, int Cout, int H, int W, int OH, int OW, int K, int stride, int pt, int pl,
                                    void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    SEGX_REQUIRE(X && Wt && Y && stem2d_ok(B, Cin, Cout, H, W, OH, OW, K, stride, pt, pl), "segx_conv2d_stem_fwd: 3 input channels, 3 x 3 window, stride 1 or 2 (B=%d Cin=%d K=%d stride=%d)", B, Cin, K, stride);
    const int64_t P = (int64_t)OH * OW;
    hipLaunchKernelGGL((conv2d_stem_fwd_kernel<3, 3>), dim3((unsigned)((P + 255) / 256), (unsigned)B), dim3(256), 0, stream, X, Wt, Y, Cout, H, W, OH, OW, stride, pt, pl);
    return check_launch("segx_conv2d_stem_fwd");
}
extern "C" int segx_conv2d_stem_im2col(const float* X, float* Xcol, int B, int Cin, int H, int W, int OH, int OW, int K, int stride, int pt, int pl, int rows, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    SEGX_REQUIRE(X && Xcol && stem2d_ok(B, Cin, 1, H, W, OH, OW, K, stride, pt, pl) && rows >= Cin * K * K && rows <= 64, "segx_conv2d_stem_im2col: bad args (rows=%d)", rows);
    const int64_t P = (int64_t)OH * OW;
    hipLaunchKernelGGL((conv2d_stem_im2col_kernel<3, 3>), dim3((unsigned)((P + 255) / 256), (unsigned)B), dim3(256), 0, stream, X, Xcol, H, W, OH, OW, stride, pt, pl, rows);
    return check_launch("segx_conv2d_stem_im2col");
}
