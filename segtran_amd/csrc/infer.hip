// infer.hip -- sliding-window evaluation path: test_single_batch (test_util2d.py:153-227),
// test_single_case (test_util3d.py:93-184), harden_segmap2d/3d (datasets2d.py:178-196, datasets3d.py:92-111),
// make_brats_pred_consistent (datasets3d.py:43-63), calc_dice (test_util2d.py:233-240).
//
// All HBM-bound, one pass each: the per-window tail `F.interpolate(scores -> window) ; sigmoid ; preds_soft[window] += ; cnt += 1`
// is ONE kernel (the reference makes four full-size temporaries per window), and the per-image tail
// `preds_soft / cnt ; consistency ; >= 0.5 ; background = no other class` is another.
#include "resample.h"

namespace segx {

struct Canvas { int CD, CH, CW, oz, oy, ox; };          // canvas spatial dims and the window's origin inside it

// acc[b][c][oz+z][oy+y][ox+x] += sigmoid(resample(scores[b][c]))(z, y, x);  cnt[b][oz+z][oy+y][ox+x] += 1.   One thread per
// window voxel (all classes), so a launch never touches a canvas cell twice; overlapping windows are separate launches.
__global__ __launch_bounds__(256) void window_accum_kernel(const float* __restrict__ scores, float* __restrict__ acc, float* __restrict__ cnt,
                                                           int B, int C, InterpDims q, Canvas cv) {
    const int64_t wsz = (int64_t)q.D * q.H * q.W, ssz = (int64_t)q.d * q.h * q.w, csz = (int64_t)cv.CD * cv.CH * cv.CW, total = (int64_t)B * wsz;
    const bool same = q.d == q.D && q.h == q.H && q.w == q.W;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        const int b = (int)(idx / wsz); int64_t r = idx - (int64_t)b * wsz;
        const int z = (int)(r / ((int64_t)q.H * q.W)); r -= (int64_t)z * q.H * q.W;
        const int y = (int)(r / q.W), x = (int)(r - (int64_t)y * q.W);
        const int64_t cell = ((int64_t)(cv.oz + z) * cv.CH + (cv.oy + y)) * cv.CW + (cv.ox + x);
        for (int c = 0; c < C; ++c) {
            const float* s = scores + ((int64_t)b * C + c) * ssz;
            const float v = same ? s[((int64_t)z * q.h + y) * q.w + x] : interp_at(s, q, z, y, x);
            acc[((int64_t)b * C + c) * csz + cell] += 1.0f / (1.0f + expf(-v));
        }
        cnt[(int64_t)b * csz + cell] += 1.0f;
    }
}

// soft = acc / cnt (cnt == NULL: acc already is the soft map); mode 1 first makes a BraTS prediction consistent, the permissive way
// (is_conservative=False): P(WT) = max(P(ET), P(WT), P(TC)), P(TC) = max(P(ET), P(TC));  then hard[c >= 1] = soft[c] >= T and
// hard[0] = (no other class is on).  One thread per voxel.
__global__ __launch_bounds__(256) void harden_kernel(const float* __restrict__ acc, const float* __restrict__ cnt, float* __restrict__ soft,
                                                     float* __restrict__ hard, int B, int C, int64_t S, int mode, float T) {
    const int64_t total = (int64_t)B * S;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        const int64_t b = idx / S, s = idx - b * S;
        const float n = cnt ? cnt[idx] : 1.0f;
        float p[8];
        for (int c = 0; c < C; ++c) p[c] = cnt ? acc[(b * C + c) * S + s] / n : acc[(b * C + c) * S + s];
        if (mode == 1) {                                   // C == 4: [bg, ET, WT, TC]
            const float et = p[1], wt = p[2], tc = p[3];
            p[2] = fmaxf(fmaxf(et, wt), tc);
            p[3] = fmaxf(et, tc);
        }
        bool any = false;
        for (int c = 1; c < C; ++c) {
            const bool on = p[c] >= T;
            any = any || on;
            hard[(b * C + c) * S + s] = on ? 1.0f : 0.0f;
            if (soft) soft[(b * C + c) * S + s] = p[c];
        }
        hard[(b * C) * S + s] = any ? 0.0f : 1.0f;
        if (soft) soft[(b * C) * S + s] = p[0];
    }
}

// per (plane, chunk): sum pred*gt, sum pred^2, sum gt^2  ->  part[chunk][plane][3]  (summed over chunks by segx_colsum)
__global__ __launch_bounds__(256) void dice_sums_kernel(const float* __restrict__ pred, const float* __restrict__ gt, float* __restrict__ part,
                                                        int64_t S) {
    __shared__ float red[4];
    const int plane = blockIdx.y, chunk = blockIdx.x, planes = gridDim.y;
    const float* p = pred + (int64_t)plane * S; const float* g = gt + (int64_t)plane * S;
    float a = 0.f, b = 0.f, c = 0.f;
    for (int64_t s = (int64_t)chunk * 256 + threadIdx.x; s < S; s += (int64_t)gridDim.x * 256) {
        const float pv = p[s], gv = g[s];
        a += pv * gv; b += pv * pv; c += gv * gv;
    }
    a = block_sum<4>(a, red); b = block_sum<4>(b, red); c = block_sum<4>(c, red);
    if (threadIdx.x == 0) { float* o = part + ((int64_t)chunk * planes + plane) * 3; o[0] = a; o[1] = b; o[2] = c; }
}

}  // namespace segx

using namespace segx;
#define SEGX_STREAM hipStream_t stream = (hipStream_t)stream_

/* scores [B, C, d, h, w] -> window [D, H, W] at origin (oz, oy, ox) of the canvas acc [B, C, CD, CH, CW], cnt [B, CD, CH, CW];
 * geom (int32[12]) = {d, h, w, D, H, W, CD, CH, CW, oz, oy, ox};  2-D: d = D = CD = 1, oz = 0 */
extern "C" int segx_window_accum(const float* scores, float* acc, float* cnt, int B, int C, const int* geom, void* stream_) {
    SEGX_STREAM; SEGX_REQUIRE(scores && acc && cnt && geom && B > 0 && C > 0, "segx_window_accum: bad args");
    const InterpDims q = make_dims(geom[0], geom[1], geom[2], geom[3], geom[4], geom[5]);
    const Canvas cv{geom[6], geom[7], geom[8], geom[9], geom[10], geom[11]};
    SEGX_REQUIRE(q.d > 0 && q.h > 0 && q.w > 0 && q.D > 0 && q.H > 0 && q.W > 0 && cv.oz >= 0 && cv.oy >= 0 && cv.ox >= 0 &&
                 cv.oz + q.D <= cv.CD && cv.oy + q.H <= cv.CH && cv.ox + q.W <= cv.CW, "segx_window_accum: window outside the canvas");
    const int64_t total = (int64_t)B * q.D * q.H * q.W;
    hipLaunchKernelGGL(window_accum_kernel, dim3((unsigned)i64min(65536, (total + 255) / 256)), dim3(256), 0, stream, scores, acc, cnt, B, C, q, cv);
    return check_launch("segx_window_accum");
}
/* mode 0: n-hot harden (harden_segmap2d/3d); mode 1: BraTS (make_brats_pred_consistent(is_conservative=False) then harden, C == 4).
 * cnt may be NULL (acc is already a probability map); soft may be NULL. */
extern "C" int segx_harden_segmap(const float* acc, const float* cnt, float* soft, float* hard, int B, int C, int64_t S, int mode, float T,
                                  void* stream_) {
    SEGX_STREAM; SEGX_REQUIRE(acc && hard && B > 0 && C >= 2 && C <= 8 && S > 0 && (mode == 0 || (mode == 1 && C == 4)), "segx_harden_segmap: bad args");
    const int64_t total = (int64_t)B * S;
    hipLaunchKernelGGL(harden_kernel, dim3((unsigned)i64min(65536, (total + 255) / 256)), dim3(256), 0, stream, acc, cnt, soft, hard, B, C, S, mode, T);
    return check_launch("segx_harden_segmap");
}
extern "C" int64_t segx_dice_ws_floats(int64_t planes, int64_t S) { return 3 * planes * i64max(1, i64min(64, (S + 4095) / 4096)); }
/* part: segx_dice_ws_floats(planes, S) floats laid out [chunks][planes][3]; sums[planes][3] = segx_colsum over the chunks */
extern "C" int segx_dice_sums(const float* pred, const float* gt, float* part, int64_t planes, int64_t S, void* stream_) {
    SEGX_STREAM; SEGX_REQUIRE(pred && gt && part && planes > 0 && planes <= 65535 && S > 0, "segx_dice_sums: bad args");
    const int chunks = (int)i64max(1, i64min(64, (S + 4095) / 4096));
    hipLaunchKernelGGL(dice_sums_kernel, dim3(chunks, (unsigned)planes), dim3(256), 0, stream, pred, gt, part, S);
    return check_launch("segx_dice_sums");
}
