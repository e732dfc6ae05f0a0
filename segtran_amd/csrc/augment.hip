// augment.hip -- the trainers' data augmentation on the device ("data pipeline on device"): the reference runs these per
// sample on the CPU inside DataLoader workers (numpy / imgaug / torchvision); here a batch that is already in HBM is transformed by a handful of
// one-pass kernels whose random parameters the host draws (same distributions, and for the numpy-only 3-D transforms the same generator and
// order as the reference).  All kernels are HBM-bound gathers / maps: coalesced over the contiguous axis of the OUTPUT.
//   3-D (dataloaders/datasets3d.py):  RandomRotFlip :547-579 + RandomCrop :491-545 (+ its zero padding) = ONE axis-permuting gather;
//                                     RandomNoise :581-597 = one map (Philox + Box-Muller, or an injected normal field for parity tests)
//   2-D (train_util.py:15-128):       iaa.Resize / CropAndPad(keep_size) = resize2d (cubic images, nearest segmentation maps);
//                                     Fliplr / Flipud / Rot90 / PadToFixedSize / CropToFixedSize = the same axis gather;
//                                     iaa.Grayscale(alpha) + transforms.ColorJitter = color_blend (+ gray_mean for the contrast pivot);
//                                     ToTensor + Normalize = normalize
#include "common.h"

namespace segx {

// out[p][o0][o1][o2] = in[p][i] with i[src[a]] = sgn[a] > 0 ? o_a + off[a] : off[a] - o_a; 0 outside the input (padding)
struct AxisMap { int I[3], O[3], src[3], sgn[3], off[3]; };
__global__ __launch_bounds__(256) void axis_gather_kernel(const float* __restrict__ X, float* __restrict__ Y, AxisMap m, int64_t planes) {
    const int64_t osz = (int64_t)m.O[0] * m.O[1] * m.O[2], isz = (int64_t)m.I[0] * m.I[1] * m.I[2], total = planes * osz;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        const int64_t pl = idx / osz; const int64_t r = idx - pl * osz;
        const int o[3] = {(int)(r / ((int64_t)m.O[1] * m.O[2])), (int)((r / m.O[2]) % m.O[1]), (int)(r % m.O[2])};
        int i[3] = {0, 0, 0};
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const int v = m.sgn[a] > 0 ? o[a] + m.off[a] : m.off[a] - o[a];
            if (m.src[a] == 0) i[0] = v; else if (m.src[a] == 1) i[1] = v; else i[2] = v;
        }
        const bool in = (unsigned)i[0] < (unsigned)m.I[0] && (unsigned)i[1] < (unsigned)m.I[1] && (unsigned)i[2] < (unsigned)m.I[2];
        Y[idx] = in ? X[pl * isz + ((int64_t)i[0] * m.I[1] + i[1]) * m.I[2] + i[2]] : 0.f;
    }
}

// y = x + (clip(sigma * z, -2 sigma, 2 sigma) + mu) * (nonzero_only ? x != 0 : 1), z ~ N(0, 1): noise[i] when given, else Box-Muller on Philox
__global__ __launch_bounds__(256) void add_noise_kernel(const float* __restrict__ X, const float* __restrict__ noise, float* __restrict__ Y, int64_t n,
                                                        float mu, float sigma, int nonzero_only, uint64_t seed, uint64_t offset) {
    for (int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x; 4 * q < n; q += (int64_t)gridDim.x * 256) {
        float z[4];
        if (noise) {
#pragma unroll
            for (int j = 0; j < 4; ++j) z[j] = 4 * q + j < n ? noise[4 * q + j] : 0.f;
        } else {
            const u32x4 r = philox4(seed, (offset >> 2) + (uint64_t)q);
            const float u0 = ((float)(r.x >> 8) + 0.5f) * (1.0f / 16777216.0f), u1 = (float)(r.y >> 8) * (1.0f / 16777216.0f);
            const float u2 = ((float)(r.z >> 8) + 0.5f) * (1.0f / 16777216.0f), u3 = (float)(r.w >> 8) * (1.0f / 16777216.0f);
            const float ra = sqrtf(-2.0f * __logf(u0)), rb = sqrtf(-2.0f * __logf(u2));
            z[0] = ra * cosf(6.28318530718f * u1); z[1] = ra * sinf(6.28318530718f * u1);
            z[2] = rb * cosf(6.28318530718f * u3); z[3] = rb * sinf(6.28318530718f * u3);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int64_t i = 4 * q + j;
            if (i < n) {
                const float x = X[i];
                const float e = fminf(fmaxf(sigma * z[j], -2.0f * sigma), 2.0f * sigma) + mu;
                Y[i] = x + ((nonzero_only && x == 0.f) ? 0.f : e);
            }
        }
    }
}

// 2-D resampling with the conventions of cv2.resize (what imgaug calls): half-pixel centres, source index = (dst + 0.5) * (in / out) - 0.5;
// mode 0 nearest (cv2.INTER_NEAREST: floor(dst * in / out)), 1 bilinear, 2 bicubic (A = -0.75, replicated border); quantize: round + clamp to
// [0, 255] (the uint8 image the reference's pipeline carries between augmenters)
__device__ __forceinline__ void cubic_w(float t, float (&w)[4]) {
    const float A = -0.75f;
    w[0] = ((A * (t + 1.f) - 5.f * A) * (t + 1.f) + 8.f * A) * (t + 1.f) - 4.f * A;
    w[1] = ((A + 2.f) * t - (A + 3.f)) * t * t + 1.f;
    w[2] = ((A + 2.f) * (1.f - t) - (A + 3.f)) * (1.f - t) * (1.f - t) + 1.f;
    w[3] = 1.f - w[0] - w[1] - w[2];
}
__global__ __launch_bounds__(256) void resize2d_kernel(const float* __restrict__ X, float* __restrict__ Y, int64_t planes, int h, int w, int H, int W,
                                                       int mode, int quantize) {
    const int64_t osz = (int64_t)H * W, isz = (int64_t)h * w, total = planes * osz;
    const float sy = (float)h / (float)H, sx = (float)w / (float)W;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        const int64_t pl = idx / osz; const int r = (int)(idx - pl * osz);
        const int oy = r / W, ox = r - oy * W;
        const float* p = X + pl * isz;
        float v;
        if (mode == 0) {
            const int iy = min((int)floorf(oy * sy), h - 1), ix = min((int)floorf(ox * sx), w - 1);
            v = p[(int64_t)iy * w + ix];
        } else {
            const float fy = (oy + 0.5f) * sy - 0.5f, fx = (ox + 0.5f) * sx - 0.5f;
            const int y0 = (int)floorf(fy), x0 = (int)floorf(fx);
            const float ty = fy - y0, tx = fx - x0;
            if (mode == 1) {
                const int ya = max(y0, 0), yb = min(y0 + 1, h - 1), xa = max(x0, 0), xb = min(x0 + 1, w - 1);
                const float top = p[(int64_t)ya * w + xa] * (1.f - tx) + p[(int64_t)ya * w + xb] * tx;
                const float bot = p[(int64_t)yb * w + xa] * (1.f - tx) + p[(int64_t)yb * w + xb] * tx;
                v = top * (1.f - ty) + bot * ty;
            } else {
                float wy[4], wx[4];
                cubic_w(ty, wy); cubic_w(tx, wx);
                v = 0.f;
#pragma unroll
                for (int a = 0; a < 4; ++a) {
                    const int yy = min(max(y0 - 1 + a, 0), h - 1);
                    float rowv = 0.f;
#pragma unroll
                    for (int b = 0; b < 4; ++b) rowv += wx[b] * p[(int64_t)yy * w + min(max(x0 - 1 + b, 0), w - 1)];
                    v += wy[a] * rowv;
                }
            }
        }
        if (quantize) v = fminf(fmaxf(rintf(v), 0.f), 255.f);
        Y[idx] = v;
    }
}

// ITU-R 601-2 luma (PIL convert('L') / cv2 RGB2GRAY): 0.299 R + 0.587 G + 0.114 B
__device__ __forceinline__ float luma(float r, float g, float b) { return 0.299f * r + 0.587f * g + 0.114f * b; }

// per-sample mean of the luma image (the pivot of a contrast change); stage 1: partial sums per (sample, slab), stage 2: one thread per sample
__global__ __launch_bounds__(256) void gray_mean_stage1(const float* __restrict__ X, float* __restrict__ ws, int64_t HW, int quantize) {
    __shared__ float red[4];
    const int b = blockIdx.y;
    const float* p = X + (int64_t)b * 3 * HW;
    float s = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < HW; i += (int64_t)gridDim.x * 256) {
        float l = luma(p[i], p[HW + i], p[2 * HW + i]);
        if (quantize) l = floorf(l + 0.5f);                    // PIL's 'L' image is uint8
        s += l;
    }
    s = block_sum<4>(s, red);
    if (threadIdx.x == 0) ws[(int64_t)b * gridDim.x + blockIdx.x] = s;
}
__global__ void gray_mean_stage2(const float* __restrict__ ws, float* __restrict__ mean, int B, int nsl, int64_t HW, int quantize) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    float s = 0.f;
    for (int i = 0; i < nsl; ++i) s += ws[(int64_t)b * nsl + i];
    s /= (float)HW;
    mean[b] = quantize ? floorf(s + 0.5f) : s;                  // ImageEnhance.Contrast: int(mean + 0.5)
}

// y = f * x + (1 - f) * degenerate(x), per sample factor f[b]; mode 0 brightness (degenerate = 0), 1 contrast (= pivot[b], the mean luma),
// 2 saturation (= luma of the pixel), 3 grayscale-alpha (iaa.Grayscale(alpha): y = (1 - alpha) x + alpha luma, f = 1 - alpha); channels-first RGB
__global__ __launch_bounds__(256) void color_blend_kernel(const float* __restrict__ X, float* __restrict__ Y, int64_t HW, int mode,
                                                          const float* __restrict__ factor, const float* __restrict__ pivot, int quantize) {
    const int b = blockIdx.y;
    const float f = factor[b], pv = (mode == 1) ? pivot[b] : 0.f;
    const float* p = X + (int64_t)b * 3 * HW; float* q = Y + (int64_t)b * 3 * HW;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < HW; i += (int64_t)gridDim.x * 256) {
        const float r = p[i], g = p[HW + i], bl = p[2 * HW + i];
        float d0 = 0.f, d1 = 0.f, d2 = 0.f;
        if (mode == 1) d0 = d1 = d2 = pv;
        else if (mode >= 2) { float l = luma(r, g, bl); if (quantize) l = floorf(l + 0.5f); d0 = d1 = d2 = l; }
        float o0 = f * r + (1.f - f) * d0, o1 = f * g + (1.f - f) * d1, o2 = f * bl + (1.f - f) * d2;
        if (quantize) { o0 = fminf(fmaxf(floorf(o0 + 0.5f), 0.f), 255.f); o1 = fminf(fmaxf(floorf(o1 + 0.5f), 0.f), 255.f); o2 = fminf(fmaxf(floorf(o2 + 0.5f), 0.f), 255.f); }
        q[i] = o0; q[HW + i] = o1; q[2 * HW + i] = o2;
    }
}

// transforms.ToTensor + Normalize: y[b][c] = (x[b][c] * scale - mean[c]) / std[c]
__global__ __launch_bounds__(256) void normalize_kernel(const float* __restrict__ X, float* __restrict__ Y, int C, int64_t HW, float scale,
                                                        const float* __restrict__ mean, const float* __restrict__ std, int64_t total) {
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        const int c = (int)((idx / HW) % C);
        Y[idx] = (X[idx] * scale - mean[c]) / std[c];
    }
}

}  // namespace segx

using namespace segx;
#define SEGX_STREAM hipStream_t stream = (hipStream_t)stream_

extern "C" int segx_axis_gather(const float* X, float* Y, int64_t planes, const int* geom, void* stream_) {
    SEGX_STREAM; SEGX_REQUIRE(X && Y && geom && planes > 0, "segx_axis_gather: bad args");
    AxisMap m;
    int seen = 0;
    for (int a = 0; a < 3; ++a) {
        m.I[a] = geom[a]; m.O[a] = geom[3 + a]; m.src[a] = geom[6 + a]; m.sgn[a] = geom[9 + a]; m.off[a] = geom[12 + a];
        SEGX_REQUIRE(m.I[a] > 0 && m.O[a] > 0 && m.src[a] >= 0 && m.src[a] < 3 && (m.sgn[a] == 1 || m.sgn[a] == -1), "segx_axis_gather: bad geom[%d]", a);
        seen |= 1 << m.src[a];
    }
    SEGX_REQUIRE(seen == 7, "segx_axis_gather: src axes are not a permutation");
    SEGX_REQUIRE((int64_t)m.I[0] * m.I[1] * m.I[2] < 2147483647LL && (int64_t)m.O[0] * m.O[1] * m.O[2] < 2147483647LL, "segx_axis_gather: plane too large");
    const int64_t total = planes * m.O[0] * m.O[1] * m.O[2];
    hipLaunchKernelGGL(axis_gather_kernel, dim3((unsigned)i64min(1 << 20, (total + 255) / 256)), dim3(256), 0, stream, X, Y, m, planes);
    return check_launch("segx_axis_gather");
}
extern "C" int segx_add_noise(const float* X, const float* noise, float* Y, int64_t n, float mu, float sigma, int nonzero_only, uint64_t seed,
                              uint64_t offset, void* stream_) {
    SEGX_STREAM; SEGX_REQUIRE(X && Y && n > 0 && sigma >= 0.f && offset % 4 == 0, "segx_add_noise: bad args");
    hipLaunchKernelGGL(add_noise_kernel, dim3((unsigned)i64min(65536, (n / 4 + 256) / 256)), dim3(256), 0, stream, X, noise, Y, n, mu, sigma, nonzero_only, seed, offset);
    return check_launch("segx_add_noise");
}
extern "C" int segx_resize2d(const float* X, float* Y, int64_t planes, int h, int w, int H, int W, int mode, int quantize, void* stream_) {
    SEGX_STREAM; SEGX_REQUIRE(X && Y && planes > 0 && h > 0 && w > 0 && H > 0 && W > 0 && mode >= 0 && mode <= 2, "segx_resize2d: bad args");
    const int64_t total = planes * H * W;
    hipLaunchKernelGGL(resize2d_kernel, dim3((unsigned)i64min(1 << 20, (total + 255) / 256)), dim3(256), 0, stream, X, Y, planes, h, w, H, W, mode, quantize);
    return check_launch("segx_resize2d");
}
extern "C" int64_t segx_gray_mean_ws_floats(int B, int64_t HW) { (void)HW; return (int64_t)B * 64; }
extern "C" int segx_gray_mean(const float* X, float* mean, float* ws, int B, int64_t HW, int quantize, void* stream_) {
    SEGX_STREAM; SEGX_REQUIRE(X && mean && ws && B > 0 && HW > 0, "segx_gray_mean: bad args");
    const int nsl = (int)i64min(64, (HW + 4095) / 4096);
    hipLaunchKernelGGL(gray_mean_stage1, dim3(nsl, B), dim3(256), 0, stream, X, ws, HW, quantize);
    hipLaunchKernelGGL(gray_mean_stage2, dim3((B + 63) / 64), dim3(64), 0, stream, (const float*)ws, mean, B, nsl, HW, quantize);
    return check_launch("segx_gray_mean");
}
extern "C" int segx_color_blend(const float* X, float* Y, int B, int64_t HW, int mode, const float* factor, const float* pivot, int quantize,
                                void* stream_) {
    SEGX_STREAM; SEGX_REQUIRE(X && Y && factor && B > 0 && HW > 0 && mode >= 0 && mode <= 3 && (mode != 1 || pivot), "segx_color_blend: bad args");
    hipLaunchKernelGGL(color_blend_kernel, dim3((unsigned)i64min(1024, (HW + 255) / 256), B), dim3(256), 0, stream, X, Y, HW, mode, factor, pivot, quantize);
    return check_launch("segx_color_blend");
}
extern "C" int segx_normalize(const float* X, float* Y, int B, int C, int64_t HW, float scale, const float* mean, const float* std, void* stream_) {
    SEGX_STREAM; SEGX_REQUIRE(X && Y && mean && std && B > 0 && C > 0 && HW > 0, "segx_normalize: bad args");
    const int64_t total = (int64_t)B * C * HW;
    hipLaunchKernelGGL(normalize_kernel, dim3((unsigned)i64min(65536, (total + 255) / 256)), dim3(256), 0, stream, X, Y, C, HW, scale, mean, std, total);
    return check_launch("segx_normalize");
}
