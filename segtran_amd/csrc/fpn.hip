// fpn.hip -- HBM-bound kernels of the input / output feature pyramids (K16-K18): GroupNorm(8) forward/backward
// and (bi/tri)linear resampling with align_corners=False, with the FPN's "lateral + upsampled" addition fused
// into the resampling pass.  Tensors are NC[D]HW fp32; a (sample, channel) plane is contiguous.
#include "resample.h"

namespace segx {

constexpr int GN_SLABS = 64;            // 32 (sample, group) pairs x 64 slabs = 2048 workgroups on the 616 MB 3-D FPN tensors

// =================================================================================================
// GroupNorm (segtran2d.py:148-149,190-192 nn.GroupNorm(G=8, C), eps 1e-5).  Channels of a group are adjacent, so a
// (sample, group) is one contiguous run of (C/G)*S floats.
// =================================================================================================
__global__ __launch_bounds__(256) void gn_stats_stage1(const float* __restrict__ X, float* __restrict__ ws, int64_t L) {
    __shared__ float red[4];
    const int bg = blockIdx.x, slab = blockIdx.y;
    const float* x = X + (int64_t)bg * L;
    const float pivot = x[0];
    const int64_t per = ((L + GN_SLABS - 1) / GN_SLABS + 3) / 4 * 4, s0 = slab * per, s1 = i64min(L, s0 + per);
    float a = 0.f, q = 0.f;
    if ((L & 3) == 0 && (reinterpret_cast<uintptr_t>(X) & 15) == 0) {
        for (int64_t s = s0 + 4 * threadIdx.x; s < s1; s += 1024) {
            const float4 v = *reinterpret_cast<const float4*>(x + s);
            const float d0 = v.x - pivot, d1 = v.y - pivot, d2 = v.z - pivot, d3 = v.w - pivot;
            a += (d0 + d1) + (d2 + d3); q += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
        }
    } else {
        for (int64_t s = s0 + threadIdx.x; s < s1; s += 256) { const float d = x[s] - pivot; a += d; q += d * d; }
    }
    a = block_sum<4>(a, red); q = block_sum<4>(q, red);
    if (threadIdx.x == 0) { ws[((int64_t)bg * GN_SLABS + slab) * 2] = a; ws[((int64_t)bg * GN_SLABS + slab) * 2 + 1] = q; }
}
__global__ __launch_bounds__(256) void gn_stats_stage2(const float* __restrict__ X, const float* __restrict__ ws, float* __restrict__ mean,
                                                       float* __restrict__ rstd, int BG, int64_t L, float eps) {
    const int bg = blockIdx.x * 256 + threadIdx.x;
    if (bg >= BG) return;
    float a = 0.f, q = 0.f;
    for (int i = 0; i < GN_SLABS; ++i) { a += ws[((int64_t)bg * GN_SLABS + i) * 2]; q += ws[((int64_t)bg * GN_SLABS + i) * 2 + 1]; }
    const float n = (float)L, md = a / n;
    mean[bg] = X[(int64_t)bg * L] + md;
    rstd[bg] = rsqrtf(fmaxf(q / n - md * md, 0.f) + eps);
}
// grid (chunks, B*C)
__global__ __launch_bounds__(256) void gn_apply_kernel(const float* __restrict__ X, const float* __restrict__ mean, const float* __restrict__ rstd,
                                                       const float* __restrict__ w, const float* __restrict__ b, float* __restrict__ Y,
                                                       int C, int G, int64_t S) {
    const int bc = blockIdx.y, c = bc % C, bg = (bc / C) * G + c / (C / G);
    const float sc = rstd[bg] * w[c], sh = b[c] - mean[bg] * sc;
    const float* x = X + (int64_t)bc * S; float* y = Y + (int64_t)bc * S;
    for (int64_t s = (int64_t)blockIdx.x * 256 + threadIdx.x; s < S; s += (int64_t)gridDim.x * 256) y[s] = x[s] * sc + sh;
}
// per plane (b,c): psum[bc] = (sum dy, sum dy * xhat)
__global__ __launch_bounds__(256) void gn_bwd_plane_sums(const float* __restrict__ dY, const float* __restrict__ X, const float* __restrict__ mean,
                                                         const float* __restrict__ rstd, float* __restrict__ psum, int C, int G, int64_t S) {
    __shared__ float red[4];
    const int bc = blockIdx.x, c = bc % C, bg = (bc / C) * G + c / (C / G);
    const float m = mean[bg], r = rstd[bg];
    const float* x = X + (int64_t)bc * S; const float* g = dY + (int64_t)bc * S;
    float a = 0.f, q = 0.f;
    for (int64_t s = threadIdx.x; s < S; s += 256) { const float gv = g[s]; a += gv; q += gv * ((x[s] - m) * r); }
    a = block_sum<4>(a, red); q = block_sum<4>(q, red);
    if (threadIdx.x == 0) { psum[2 * bc] = a; psum[2 * bc + 1] = q; }
}
// tiny: group sums gsum[bg] = (sum_c w_c * psum0, sum_c w_c * psum1) ; dw[c] = sum_b psum1 ; db[c] = sum_b psum0
__global__ __launch_bounds__(256) void gn_bwd_finalize(const float* __restrict__ psum, const float* __restrict__ w, float* __restrict__ gsum,
                                                       float* __restrict__ dw, float* __restrict__ db, int B, int C, int G) {
    const int t = blockIdx.x * 256 + threadIdx.x, cpg = C / G;
    if (t < B * G) {
        const int b = t / G, g = t % G;
        float a = 0.f, q = 0.f;
        for (int c = g * cpg; c < (g + 1) * cpg; ++c) { a += w[c] * psum[2 * (b * C + c)]; q += w[c] * psum[2 * (b * C + c) + 1]; }
        gsum[2 * t] = a; gsum[2 * t + 1] = q;
    }
    if (t < C) {
        float a = 0.f, q = 0.f;
        for (int b = 0; b < B; ++b) { a += psum[2 * (b * C + t)]; q += psum[2 * (b * C + t) + 1]; }
        db[t] = a; dw[t] = q;
    }
}
// dx = rstd * (dy*w - s1/n - xhat * s2/n)
__global__ __launch_bounds__(256) void gn_bwd_apply(const float* __restrict__ dY, const float* __restrict__ X, const float* __restrict__ mean,
                                                    const float* __restrict__ rstd, const float* __restrict__ w, const float* __restrict__ gsum,
                                                    float* __restrict__ dX, int C, int G, int64_t S) {
    const int bc = blockIdx.y, c = bc % C, bg = (bc / C) * G + c / (C / G);
    const float m = mean[bg], r = rstd[bg], wc = w[c];
    const float inv_n = 1.0f / ((float)(C / G) * (float)S), k1 = gsum[2 * bg] * inv_n, k2 = gsum[2 * bg + 1] * inv_n;
    const float* x = X + (int64_t)bc * S; const float* g = dY + (int64_t)bc * S; float* d = dX + (int64_t)bc * S;
    for (int64_t s = (int64_t)blockIdx.x * 256 + threadIdx.x; s < S; s += (int64_t)gridDim.x * 256)
        d[s] = r * (g[s] * wc - k1 - (x[s] - m) * r * k2);
}

// =================================================================================================
// Linear resampling, align_corners=False (F.interpolate 'bilinear'/'trilinear': segtran2d.py:249,291,305,435;
// segtran3d.py:304,319,351,364,384,495).  Source coordinate of destination index d along one axis, exactly as ATen:
//   src = max(scale * (d + 0.5) - 0.5, 0),  scale = n_in / n_out (float);  i0 = floor(src), i1 = min(i0+1, n_in-1), l = src - i0
// 2-D tensors use d = D = 1.  Forward optionally adds a base tensor (the FPN lateral) in the same pass.
// =================================================================================================
__global__ __launch_bounds__(256) void interp_fwd_kernel(const float* __restrict__ in, const float* __restrict__ base, float* __restrict__ out,
                                                         InterpDims q, int64_t planes) {
    const int64_t osz = (int64_t)q.D * q.H * q.W, isz = (int64_t)q.d * q.h * q.w, total = planes * osz;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        const int64_t p = idx / osz; int64_t r = idx - p * osz;
        const int z = (int)(r / ((int64_t)q.H * q.W)); r -= (int64_t)z * q.H * q.W;
        const int y = (int)(r / q.W), x = (int)(r - (int64_t)y * q.W);
        float v = interp_at(in + p * isz, q, z, y, x);
        if (base) v += base[idx];
        out[idx] = v;
    }
}
// float4 variant (W % 4 == 0, W <= 1024): a thread owns four adjacent outputs of one output row, so the (plane, z, y) decode and the
// two row-axis sources are computed once per four outputs, the lateral is read and the result written as float4.  Same blend order
// as interp_at (x, then y, then z), hence the same bits.
__global__ __launch_bounds__(256) void interp_fwd_rows_kernel(const float* __restrict__ in, const float* __restrict__ base, float* __restrict__ out,
                                                              InterpDims q, int w4, int rpb, FastDiv divH, FastDiv divW4) {
    // grid (row blocks of one plane, plane): no 64-bit division anywhere; one output row per thread group of w4 lanes
    const int tr = fdiv(threadIdx.x, divW4), tx = threadIdx.x - tr * w4;
    const int row = blockIdx.x * rpb + tr;                                   // (z, y) row of this plane
    if (tr >= rpb || row >= q.D * q.H) return;
    const int z = fdiv(row, divH), y = row - z * q.H;
    const int64_t p = blockIdx.y;
    const Axis az = axis_src(z, q.d, q.sd), ay = axis_src(y, q.h, q.sh);
    const float* s = in + p * ((int64_t)q.d * q.h * q.w);
    const float* r00 = s + (az.i0 * q.h + ay.i0) * q.w; const float* r01 = s + (az.i0 * q.h + ay.i1) * q.w;
    const float* r10 = s + (az.i1 * q.h + ay.i0) * q.w; const float* r11 = s + (az.i1 * q.h + ay.i1) * q.w;
    float v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const Axis ax = axis_src(tx * 4 + j, q.w, q.sw);
        const float l = ax.l; const int i0 = ax.i0, i1 = ax.i1;
        const float c00 = r00[i0] * (1.f - l) + r00[i1] * l, c01 = r01[i0] * (1.f - l) + r01[i1] * l;
        const float c10 = r10[i0] * (1.f - l) + r10[i1] * l, c11 = r11[i0] * (1.f - l) + r11[i1] * l;
        v[j] = (c00 * (1.f - ay.l) + c01 * ay.l) * (1.f - az.l) + (c10 * (1.f - ay.l) + c11 * ay.l) * az.l;
    }
    const int64_t o = (p * q.D * q.H + row) * q.W + tx * 4;
    if (base) { const float4 bv = *reinterpret_cast<const float4*>(base + o); v[0] += bv.x; v[1] += bv.y; v[2] += bv.z; v[3] += bv.w; }
    *reinterpret_cast<float4*>(out + o) = make_float4(v[0], v[1], v[2], v[3]);
}
// adjoint as a GATHER (deterministic, no atomics): an input cell collects from every output cell it was blended into
__device__ __forceinline__ void cand_range(int i, int n_out, float scale, int& lo, int& hi) {
    if (scale < 0.f) {                                   // align_corners: src = |scale| * d lies in (i - 1, i + 1)
        const float inv = -1.0f / scale;
        lo = (int)floorf(((float)i - 1.f) * inv) - 1; hi = (int)ceilf(((float)i + 1.f) * inv) + 1;
        lo = lo < 0 ? 0 : lo; hi = hi > n_out - 1 ? n_out - 1 : hi;
        return;
    }
    const float inv = 1.0f / scale;
    lo = (int)floorf(((float)i - 0.5f) * inv - 0.5f) - 1; hi = (int)ceilf(((float)i + 1.5f) * inv - 0.5f) + 1;
    lo = lo < 0 ? 0 : lo; hi = hi > n_out - 1 ? n_out - 1 : hi;
}
__device__ __forceinline__ float axis_weight(int i, int d, int n_in, float scale) {
    const Axis a = axis_src(d, n_in, scale);
    return (a.i0 == i ? 1.f - a.l : 0.f) + (a.i1 == i ? a.l : 0.f);
}
__global__ __launch_bounds__(256) void interp_bwd_kernel(const float* __restrict__ dout, float* __restrict__ din, InterpDims q, int64_t planes) {
    const int64_t osz = (int64_t)q.D * q.H * q.W, isz = (int64_t)q.d * q.h * q.w, total = planes * isz;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        const int64_t p = idx / isz; int64_t r = idx - p * isz;
        const int iz = (int)(r / ((int64_t)q.h * q.w)); r -= (int64_t)iz * q.h * q.w;
        const int iy = (int)(r / q.w), ix = (int)(r - (int64_t)iy * q.w);
        int z0, z1, y0, y1, x0, x1;
        cand_range(iz, q.D, q.sd, z0, z1); cand_range(iy, q.H, q.sh, y0, y1); cand_range(ix, q.W, q.sw, x0, x1);
        const float* g = dout + p * osz;
        float acc = 0.f;
        for (int z = z0; z <= z1; ++z) {
            const float wz = axis_weight(iz, z, q.d, q.sd);
            if (wz == 0.f) continue;
            for (int y = y0; y <= y1; ++y) {
                const float wy = axis_weight(iy, y, q.h, q.sh) * wz;
                if (wy == 0.f) continue;
                float rowacc = 0.f;
                for (int x = x0; x <= x1; ++x) rowacc += axis_weight(ix, x, q.w, q.sw) * g[((int64_t)z * q.H + y) * q.W + x];
                acc += wy * rowacc;
            }
        }
        din[idx] = acc;
    }
}

// one-axis adjoint on a tensor viewed as [outer, n, inner]: the separable form of interp_bwd_kernel (three cheap passes
// instead of one pass with prod(2*scale+1) candidates per cell -- 10x faster for the x4 trilinear up-sampling of the 3-D FPN)
__global__ __launch_bounds__(256) void interp_bwd_axis_kernel(const float* __restrict__ dout, float* __restrict__ din, int64_t outer,
                                                              int n_out, int n_in, int inner, FastDiv divInner, FastDiv divPer, float scale) {
    // flat grid over outer * n_in * inner cells; as in the forward kernel the (slice, cell) split of a workgroup's first element is one
    // uniform 64-bit division, the per-thread remainder 32-bit multiply-high divisions (per = n_in * inner < 2^31: host check)
    const int per = n_in * inner;
    const int64_t total = outer * per;
    for (int64_t b0 = (int64_t)blockIdx.x * 256; b0 < total; b0 += (int64_t)gridDim.x * 256) {
        const int64_t ob0 = b0 / per;
        const int e0 = (int)(b0 - ob0 * per) + threadIdx.x, qo = fdiv(e0, divPer);
        const int64_t o = ob0 + qo; const int e = e0 - qo * per;
        if (o >= outer) continue;
        const int i = fdiv(e, divInner), in_ = e - i * inner;
        int lo, hi; cand_range(i, n_out, scale, lo, hi);
        const float* g = dout + (o * n_out) * inner + in_;
        float acc = 0.f;
        for (int d = lo; d <= hi; ++d) acc += axis_weight(i, d, n_in, scale) * g[(int64_t)d * inner];
        din[o * per + e] = acc;
    }
}

// float4 over the inner (contiguous) extent: the candidate range and the blend weights depend on the axis index only
__global__ __launch_bounds__(256) void interp_bwd_axis4_kernel(const float* __restrict__ dout, float* __restrict__ din, int64_t outer,
                                                               int n_out, int n_in, int inner4, FastDiv divInner, FastDiv divPer, float scale) {
    const int per = n_in * inner4;
    const int64_t total = outer * per;
    for (int64_t b0 = (int64_t)blockIdx.x * 256; b0 < total; b0 += (int64_t)gridDim.x * 256) {
        const int64_t ob0 = b0 / per;
        const int e0 = (int)(b0 - ob0 * per) + threadIdx.x, qo = fdiv(e0, divPer);
        const int64_t o = ob0 + qo; const int e = e0 - qo * per;
        if (o >= outer) continue;
        const int i = fdiv(e, divInner), in_ = e - i * inner4;
        int lo, hi; cand_range(i, n_out, scale, lo, hi);
        const float4* g = reinterpret_cast<const float4*>(dout) + (o * n_out) * inner4 + in_;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int d = lo; d <= hi; ++d) {
            const float w = axis_weight(i, d, n_in, scale);
            const float4 v = g[(int64_t)d * inner4];
            acc.x += w * v.x; acc.y += w * v.y; acc.z += w * v.z; acc.w += w * v.w;
        }
        reinterpret_cast<float4*>(din)[o * per + e] = acc;
    }
}

// Forward resampling along ONE axis of a tensor viewed as [outer, n_in, inner] -> [outer, n_out, inner] (+ base).  Chaining the axes
// x -> y -> z performs exactly the operations of the fused formula in the same order (x blend, then y blend, then z blend), so the
// result is bit-identical; each pass is a plain stream.  VEC: float4 over the contiguous inner extent.
template <bool VEC>
__global__ __launch_bounds__(256) void interp_fwd_axis_kernel(const float* __restrict__ in, const float* __restrict__ base, float* __restrict__ out,
                                                              int n_in, int n_out, int inner, FastDiv divInner, FastDiv divNout, float scale,
                                                              int64_t outer) {
    // flat grid over outer * n_out * inner elements: the (slice, element) split of a workgroup's first element is uniform (scalar
    // 64-bit division once), the per-thread remainder is a 32-bit multiply-high
    const int per = n_out * inner;
    const int64_t total = outer * per;
    for (int64_t b0 = (int64_t)blockIdx.x * 256; b0 < total; b0 += (int64_t)gridDim.x * 256) {
        const int64_t ob0 = b0 / per;
        const int e0 = (int)(b0 - ob0 * per) + threadIdx.x, qo = fdiv(e0, divNout);      // divNout = divider by `per`
        const int64_t o = ob0 + qo; const int e = e0 - qo * per;
        if (o >= outer) continue;
        const int i = fdiv(e, divInner), c = e - i * inner;
        const Axis a = axis_src(i, n_in, scale);
        const int64_t ob = o * per + e, i0 = (o * n_in + a.i0) * (int64_t)inner + c, i1 = (o * n_in + a.i1) * (int64_t)inner + c;
        if (VEC) {
            const float4 v0 = reinterpret_cast<const float4*>(in)[i0], v1 = reinterpret_cast<const float4*>(in)[i1];
            float4 r = make_float4(v0.x * (1.f - a.l) + v1.x * a.l, v0.y * (1.f - a.l) + v1.y * a.l, v0.z * (1.f - a.l) + v1.z * a.l,
                                   v0.w * (1.f - a.l) + v1.w * a.l);
            if (base) { const float4 b = reinterpret_cast<const float4*>(base)[ob]; r.x += b.x; r.y += b.y; r.z += b.z; r.w += b.w; }
            reinterpret_cast<float4*>(out)[ob] = r;
        } else {
            float r = in[i0] * (1.f - a.l) + in[i1] * a.l;
            if (base) r += base[ob];
            out[ob] = r;
        }
    }
}

// =================================================================================================
// r04: trilinear up-sampling (+ lateral) of the 3-D FPN as ONE pass each way.  The separable form above moves 29 coarse-tensor sizes per 2 x 2 x 2
// up-sampling forward (x: 1 + 2, y: 2 + 4, z: 4 + 8 + the 8 of the lateral) and 21 backward; with the source tile of an output tile staged in LDS the
// forward reads the coarse tensor once, the lateral once and writes the fine tensor once (17), the adjoint reads the fine gradient once and writes
// the coarse one (9).  The blends happen in the same order as in the separable passes (x, then y, then z; adjoint: z, then y, then x), so the
// results are the same numbers.  Conditions (host): every axis up-samples by at most 2 (0.5 <= n_in / n_out <= 1), W % 4 == 0, align_corners = False.
// =================================================================================================
constexpr int IT_TZ = 8, IT_TY = 16, IT_TX = 32;                       // output tile of the forward kernel / fine-gradient region of the adjoint
constexpr int IT_SZ = IT_TZ + 2, IT_SY = IT_TY + 2, IT_SX = IT_TX + 2;  // its source tile (scale <= 1: at most tile + 2 per axis)
__global__ __launch_bounds__(256) void interp3d_fwd_tile_kernel(const float* __restrict__ in, const float* __restrict__ base, float* __restrict__ out,
                                                                InterpDims q, int tiles_y, int tiles_x) {
    __shared__ float src[IT_SZ * IT_SY * IT_SX];
    const int64_t p = blockIdx.y;
    int t = blockIdx.x;
    const int txi = t % tiles_x; t /= tiles_x;
    const int tyi = t % tiles_y, tzi = t / tiles_y;
    const int z0 = tzi * IT_TZ, y0 = tyi * IT_TY, x0 = txi * IT_TX;
    const int zl = min(z0 + IT_TZ, q.D) - 1, yl = min(y0 + IT_TY, q.H) - 1, xl = min(x0 + IT_TX, q.W) - 1;
    const int sz0 = axis_src(z0, q.d, q.sd).i0, sy0 = axis_src(y0, q.h, q.sh).i0, sx0 = axis_src(x0, q.w, q.sw).i0;
    const int nz = axis_src(zl, q.d, q.sd).i1 - sz0 + 1, ny = axis_src(yl, q.h, q.sh).i1 - sy0 + 1, nx = axis_src(xl, q.w, q.sw).i1 - sx0 + 1;
    const float* s = in + p * ((int64_t)q.d * q.h * q.w);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int row = wave; row < nz * ny; row += 4) {                      // a wave per source row: (zz, yy) is wave-uniform, the lanes run along x
        const int zz = row / ny, yy = row - zz * ny;
        if (lane < nx) src[(zz * IT_SY + yy) * IT_SX + lane] = s[((int64_t)(sz0 + zz) * q.h + (sy0 + yy)) * q.w + sx0 + lane];
    }
    __syncthreads();
    const int tx4 = threadIdx.x & 7, ty = (threadIdx.x >> 3) & 15, tzp = threadIdx.x >> 7;
    const int y = y0 + ty, x = x0 + 4 * tx4;
    if (y >= q.H || x >= q.W) return;
    const Axis ay = axis_src(y, q.h, q.sh);
    Axis ax[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) { ax[j] = axis_src(x + j, q.w, q.sw); ax[j].i0 -= sx0; ax[j].i1 -= sx0; }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int z = z0 + tzp * 4 + k;
        if (z >= q.D) break;
        const Axis az = axis_src(z, q.d, q.sd);
        const float* r00 = src + ((az.i0 - sz0) * IT_SY + (ay.i0 - sy0)) * IT_SX; const float* r01 = src + ((az.i0 - sz0) * IT_SY + (ay.i1 - sy0)) * IT_SX;
        const float* r10 = src + ((az.i1 - sz0) * IT_SY + (ay.i0 - sy0)) * IT_SX; const float* r11 = src + ((az.i1 - sz0) * IT_SY + (ay.i1 - sy0)) * IT_SX;
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float l = ax[j].l; const int i0 = ax[j].i0, i1 = ax[j].i1;
            const float c00 = r00[i0] * (1.f - l) + r00[i1] * l, c01 = r01[i0] * (1.f - l) + r01[i1] * l;
            const float c10 = r10[i0] * (1.f - l) + r10[i1] * l, c11 = r11[i0] * (1.f - l) + r11[i1] * l;
            v[j] = (c00 * (1.f - ay.l) + c01 * ay.l) * (1.f - az.l) + (c10 * (1.f - ay.l) + c11 * ay.l) * az.l;
        }
        const int64_t o = ((p * q.D + z) * q.H + y) * q.W + x;
        if (base) { const float4 bv = *reinterpret_cast<const float4*>(base + o); v[0] += bv.x; v[1] += bv.y; v[2] += bv.z; v[3] += bv.w; }
        *reinterpret_cast<float4*>(out + o) = make_float4(v[0], v[1], v[2], v[3]);
    }
}
// exact first / last output index that input cells [i_first, i_last] of an axis were blended into (cand_range is conservative: trim its zero weights)
__device__ __forceinline__ void axis_cover(int i_first, int i_last, int n_out, int n_in, float scale, int& lo, int& hi) {
    int l2, h2;
    cand_range(i_first, n_out, scale, lo, h2);
    while (lo < h2 && axis_weight(i_first, lo, n_in, scale) == 0.f) ++lo;
    cand_range(i_last, n_out, scale, l2, hi);
    while (hi > l2 && axis_weight(i_last, hi, n_in, scale) == 0.f) --hi;
}
constexpr int IB_TZ = 4, IB_TY = 8, IB_TX = 16;                          // coarse cells per workgroup of the adjoint (two per thread, along z)
__global__ __launch_bounds__(256) void interp3d_bwd_tile_kernel(const float* __restrict__ dout, float* __restrict__ din, InterpDims q, int tiles_y, int tiles_x) {
    __shared__ float reg[IT_SZ * IT_SY * IT_SX];                          // (2 * 4 + 2) x (2 * 8 + 2) x (2 * 16 + 2) at most (scale >= 0.5)
    const int64_t p = blockIdx.y;
    int t = blockIdx.x;
    const int txi = t % tiles_x; t /= tiles_x;
    const int tyi = t % tiles_y, tzi = t / tiles_y;
    const int z0 = tzi * IB_TZ, y0 = tyi * IB_TY, x0 = txi * IB_TX;
    int rz0, rz1, ry0, ry1, rx0, rx1;
    axis_cover(z0, min(z0 + IB_TZ, q.d) - 1, q.D, q.d, q.sd, rz0, rz1);
    axis_cover(y0, min(y0 + IB_TY, q.h) - 1, q.H, q.h, q.sh, ry0, ry1);
    axis_cover(x0, min(x0 + IB_TX, q.w) - 1, q.W, q.w, q.sw, rx0, rx1);
    const int nz = rz1 - rz0 + 1, ny = ry1 - ry0 + 1, nx = rx1 - rx0 + 1;
    const float* g = dout + p * ((int64_t)q.D * q.H * q.W);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int row = wave; row < nz * ny; row += 4) {
        const int zz = row / ny, yy = row - zz * ny;
        if (lane < nx) reg[(zz * IT_SY + yy) * IT_SX + lane] = g[((int64_t)(rz0 + zz) * q.H + (ry0 + yy)) * q.W + rx0 + lane];
    }
    __syncthreads();
    const int ix = x0 + (threadIdx.x & 15), iy = y0 + ((threadIdx.x >> 4) & 7);
    if (ix >= q.w || iy >= q.h) return;
    int xa, xb, ya, yb;
    cand_range(ix, q.W, q.sw, xa, xb); cand_range(iy, q.H, q.sh, ya, yb);
    xa = max(xa, rx0); xb = min(xb, rx1); ya = max(ya, ry0); yb = min(yb, ry1);
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int iz = z0 + (threadIdx.x >> 7) * 2 + k;
        if (iz >= q.d) break;
        int za, zb;
        cand_range(iz, q.D, q.sd, za, zb);
        za = max(za, rz0); zb = min(zb, rz1);
        float acc = 0.f;
        for (int X = xa; X <= xb; ++X) {                                 // the separable adjoint's order: z sums innermost, then y, then x
            const float wx = axis_weight(ix, X, q.w, q.sw);
            if (wx == 0.f) continue;
            float colacc = 0.f;
            for (int Y = ya; Y <= yb; ++Y) {
                const float wy = axis_weight(iy, Y, q.h, q.sh);
                if (wy == 0.f) continue;
                const float* col = reg + (Y - ry0) * IT_SX + (X - rx0);
                float rowacc = 0.f;
                for (int Z = za; Z <= zb; ++Z) rowacc += axis_weight(iz, Z, q.d, q.sd) * col[(Z - rz0) * (IT_SY * IT_SX)];
                colacc += wy * rowacc;
            }
            acc += wx * colacc;
        }
        din[((p * q.d + iz) * q.h + iy) * q.w + ix] = acc;
    }
}

// nn.AvgPool2d(2) (PolyformerLayer.pool2x, polyformer.py:28,40): [planes, H, W] -> [planes, H/2, W/2] (floor), and its adjoint
__global__ __launch_bounds__(256) void avgpool2_fwd_kernel(const float* __restrict__ X, float* __restrict__ Y, int64_t planes, int H, int W) {
    const int OH = H / 2, OW = W / 2; const int64_t total = planes * OH * OW;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int ox = (int)(i % OW); const int64_t r = i / OW; const int oy = (int)(r % OH); const int64_t p = r / OH;
        const float* x = X + (p * H + 2 * oy) * W + 2 * ox;
        Y[i] = ((x[0] + x[1]) + (x[W] + x[W + 1])) * 0.25f;
    }
}
__global__ __launch_bounds__(256) void avgpool2_bwd_kernel(const float* __restrict__ dY, float* __restrict__ dX, int64_t planes, int H, int W) {
    const int OH = H / 2, OW = W / 2; const int64_t total = planes * H * W;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int x = (int)(i % W); const int64_t r = i / W; const int y = (int)(r % H); const int64_t p = r / H;
        dX[i] = (y / 2 < OH && x / 2 < OW) ? 0.25f * dY[(p * OH + y / 2) * OW + x / 2] : 0.f;
    }
}
// batched 2-D transpose [batch, R, C] -> [batch, C, R] through a 32 x 33 LDS tile (channel-major feature maps <-> token-major rows)
__global__ __launch_bounds__(256) void transpose_kernel(const float* __restrict__ X, float* __restrict__ Y, int R, int C) {
    __shared__ float tile[32][33];
    const int64_t b = blockIdx.z;
    const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;      // 32 x 8 threads
    for (int j = ty; j < 32; j += 8) if (r0 + j < R && c0 + tx < C) tile[j][tx] = X[(b * R + r0 + j) * C + c0 + tx];
    __syncthreads();
    for (int j = ty; j < 32; j += 8) if (c0 + j < C && r0 + tx < R) Y[(b * C + c0 + j) * R + r0 + tx] = tile[tx][j];
}

static inline int fpn_chunks(int64_t S, int per_thread) { return (int)i64max(1, i64min(64, (S + 256 * per_thread - 1) / (256 * per_thread))); }

}  // namespace segx

using namespace segx;
#define SEGX_STREAM hipStream_t stream = (hipStream_t)stream_

extern "C" int64_t segx_gn_ws_floats(int B, int C, int G) { return (int64_t)B * G * GN_SLABS * 2 + (int64_t)2 * B * C + (int64_t)2 * B * G; }
extern "C" int segx_groupnorm_fwd(const float* X, const float* w, const float* b, float* Y, float* mean, float* rstd, float* ws,
                                  int B, int C, int G, int64_t S, float eps, void* stream_) {
    SEGX_STREAM; SEGX_REQUIRE(X && w && b && Y && mean && rstd && ws && B > 0 && C > 0 && G > 0 && C % G == 0 && S > 0, "segx_groupnorm_fwd: bad args");
    SEGX_REQUIRE((int64_t)B * C <= 65535, "segx_groupnorm_fwd: more than 65535 planes");
    const int64_t L = (int64_t)(C / G) * S;
    hipLaunchKernelGGL(gn_stats_stage1, dim3(B * G, GN_SLABS), dim3(256), 0, stream, X, ws, L);
    hipLaunchKernelGGL(gn_stats_stage2, dim3((B * G + 255) / 256), dim3(256), 0, stream, X, (const float*)ws, mean, rstd, B * G, L, eps);
    hipLaunchKernelGGL(gn_apply_kernel, dim3(fpn_chunks(S, 8), B * C), dim3(256), 0, stream, X, (const float*)mean, (const float*)rstd, w, b, Y, C, G, S);
    return check_launch("segx_groupnorm_fwd");
}
extern "C" int segx_groupnorm_bwd(const float* dY, const float* X, const float* w, const float* mean, const float* rstd, float* dX, float* dw,
                                  float* db, float* ws, int B, int C, int G, int64_t S, void* stream_) {
    SEGX_STREAM; SEGX_REQUIRE(dY && X && w && mean && rstd && dX && dw && db && ws && B > 0 && C > 0 && G > 0 && C % G == 0 && S > 0, "segx_groupnorm_bwd: bad args");
    SEGX_REQUIRE((int64_t)B * C <= 65535, "segx_groupnorm_bwd: more than 65535 planes");
    float* psum = ws + (int64_t)B * G * GN_SLABS * 2; float* gsum = psum + (int64_t)2 * B * C;
    hipLaunchKernelGGL(gn_bwd_plane_sums, dim3(B * C), dim3(256), 0, stream, dY, X, mean, rstd, psum, C, G, S);
    const int n = B * G > C ? B * G : C;
    hipLaunchKernelGGL(gn_bwd_finalize, dim3((n + 255) / 256), dim3(256), 0, stream, (const float*)psum, w, gsum, dw, db, B, C, G);
    hipLaunchKernelGGL(gn_bwd_apply, dim3(fpn_chunks(S, 8), B * C), dim3(256), 0, stream, dY, X, mean, rstd, w, (const float*)gsum, dX, C, G, S);
    return check_launch("segx_groupnorm_bwd");
}
extern "C" int segx_avgpool2_fwd(const float* X, float* Y, int64_t planes, int H, int W, void* stream_) {
    SEGX_STREAM; SEGX_REQUIRE(X && Y && planes > 0 && H >= 2 && W >= 2, "segx_avgpool2_fwd: bad args");
    const int64_t total = planes * (H / 2) * (W / 2);
    hipLaunchKernelGGL(avgpool2_fwd_kernel, dim3((unsigned)i64min(65536, (total + 255) / 256)), dim3(256), 0, stream, X, Y, planes, H, W);
    return check_launch("segx_avgpool2_fwd");
}
extern "C" int segx_avgpool2_bwd(const float* dY, float* dX, int64_t planes, int H, int W, void* stream_) {
    SEGX_STREAM; SEGX_REQUIRE(dY && dX && planes > 0 && H >= 2 && W >= 2, "segx_avgpool2_bwd: bad args");
    const int64_t total = planes * H * W;
    hipLaunchKernelGGL(avgpool2_bwd_kernel, dim3((unsigned)i64min(65536, (total + 255) / 256)), dim3(256), 0, stream, dY, dX, planes, H, W);
    return check_launch("segx_avgpool2_bwd");
}
extern "C" int segx_transpose(const float* X, float* Y, int64_t batch, int R, int C, void* stream_) {
    SEGX_STREAM; SEGX_REQUIRE(X && Y && batch > 0 && R > 0 && C > 0 && (R + 31) / 32 <= 65535, "segx_transpose: bad args");
    for (int64_t b0 = 0; b0 < batch; b0 += 65535) {                        // gridDim.z <= 65535
        const int64_t n = i64min(65535, batch - b0);
        hipLaunchKernelGGL(transpose_kernel, dim3((C + 31) / 32, (R + 31) / 32, (unsigned)n), dim3(256), 0, stream, X + b0 * R * C, Y + b0 * R * C, R, C);
    }
    return check_launch("segx_transpose");
}
// Standalone inverted dropout (nn.Dropout on the out-FPN output, --outdrop, segtran2d.py:308-310): y = x * keep(seed, offset + i) / (1 - p).
// The mask is regenerated from the Philox stream, so the backward pass is the same kernel applied to dy.
__global__ __launch_bounds__(256) void dropout_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t n, float p, float inv_keep,
                                                      uint64_t seed, uint64_t offset, const uint64_t* __restrict__ rbase) {
    offset += rbase ? *rbase : 0;
    const int64_t n4 = n >> 2;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        const float4 v = reinterpret_cast<const float4*>(x)[i];
        const float4 k = dropout_scale4(seed, offset, (uint64_t)i * 4, p, inv_keep);
        reinterpret_cast<float4*>(y)[i] = make_float4(v.x * k.x, v.y * k.y, v.z * k.z, v.w * k.w);
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const int64_t i = (n4 << 2) + threadIdx.x;
        y[i] = x[i] * dropout_scale(seed, offset, (uint64_t)i, p, inv_keep);
    }
}
extern "C" int segx_dropout(const float* x, float* y, int64_t n, float p, uint64_t seed, uint64_t offset, void* stream_) {
    SEGX_STREAM; SEGX_REQUIRE(x && y && n > 0 && p >= 0.f && p < 1.f && (offset & 3) == 0 && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 15) == 0,
                              "segx_dropout: bad args");
    hipLaunchKernelGGL(dropout_kernel, dim3((unsigned)i64min(1 << 20, (n / 4 + 256) / 256)), dim3(256), 0, stream, x, y, n, p, 1.0f / (1.0f - p), seed, offset, segx::rng_base());
    return check_launch("segx_dropout");
}
__global__ void rng_advance_kernel(uint64_t* base, uint64_t span) { if (threadIdx.x == 0 && blockIdx.x == 0) *base += span; }
/* Device-side base of every dropout Philox stream: each kernel adds *base to the `offset` it was launched with.  A train step captured into a
 * hipGraph replays the SAME offsets; advancing *base by the step's span (segx_rng_advance, itself a captured launch) gives every replay fresh
 * masks, and forward / backward of one replay still regenerate identical ones.  NULL (default): offsets are used as passed. */
extern "C" int segx_set_rng_base(const uint64_t* base) { segx::rng_base() = base; return 0; }
extern "C" int segx_rng_advance(uint64_t* base, uint64_t span, void* stream_) {
    SEGX_STREAM; SEGX_REQUIRE(base && span % 4 == 0, "segx_rng_advance: bad args");
    hipLaunchKernelGGL(rng_advance_kernel, dim3(1), dim3(64), 0, stream, base, span);
    return check_launch("segx_rng_advance");
}
extern "C" int segx_tune(int knob, int value) {
    segx::Knobs& k = segx::knobs();
    // every knob accepts only the settings the product suite exercises (tests/): an unknown value is an error, never a silent new code path
    if (knob == 1) { if (value < 0 || value > 2) return -1; k.interp_variant = value; return 0; }
    if (knob == 2) { if (value != 0 && value != 1) return -1; k.conv_small_policy = value; return 0; }
    if (knob == 4) { if (value != SEGX_ENGINE_F32 && value != SEGX_ENGINE_BF16X6) return -1; return k.engine.exchange(value); }
    if (knob == 8) { if (value < 256) return -1; k.dw_strip_outputs = value; return 0; }
    if (knob == 7) { if (value < 0 || value > 2) return -1; k.conv_x6_wgrad_all = value; return 0; }
#ifdef SEGX_BENCH
    if (knob == 6) { if (value < 0 || value > 7) return -1; k.x6_variant = value; return 0; }      // 2..5: ablations whose results are NOT the GEMM
#else
    if (knob == 6) { if (value != 0 && value != 1 && value != 6 && value != 7) return -1; k.x6_variant = value; return 0; }
#endif
    if (knob == 9) { if (value < 8 || value > 4096 || value % 8) return -1; k.ws_grid = value; return 0; }
    if (knob == 5) { return k.x6_launches.exchange(0); }
    return -1;
}
// RandomResizedCrop (datasets3d.py:611-665) as ONE gather pass: the volume is (virtually) resampled to (D, H, W) with the trilinear
// align_corners=False rule, zero-padded, and a window of (od, oh, ow) voxels is cut out at offset (oz, oy, ox) measured in the resampled,
// UNPADDED grid (i.e. crop start - front pad; negative / beyond-the-end coordinates fall into the padding and read 0).  Only the voxels
// of the window are ever computed; the resampled volume and its padded copy (the reference materialises both) never exist.
__global__ __launch_bounds__(256) void resized_crop3d_kernel(const float* __restrict__ X, float* __restrict__ Y, InterpDims q, int od, int oh, int ow,
                                                             int oz, int oy, int ox, int64_t planes) {
    const int64_t osz = (int64_t)od * oh * ow, isz = (int64_t)q.d * q.h * q.w, total = planes * osz;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        const int64_t pl = idx / osz; const int64_t r = idx - pl * osz;
        const int x = (int)(r % ow) + ox, y = (int)((r / ow) % oh) + oy, z = (int)(r / ((int64_t)ow * oh)) + oz;
        float v = 0.f;
        if ((unsigned)z < (unsigned)q.D && (unsigned)y < (unsigned)q.H && (unsigned)x < (unsigned)q.W) v = interp_at(X + pl * isz, q, z, y, x);
        Y[idx] = v;
    }
}

// nn.ConvTranspose2d(kernel 2, stride 2) = pointwise convolution onto 4 Cout channels + this re-arrangement (unet_parts.py:53, the
// bilinear=False decoder): Y[p][2 i + a][2 j + c] = X[4 p + 2 a + c][i][j] (F.pixel_shuffle with r = 2); inverse = 1 runs it backwards
// (Y[4 p + 2 a + c][i][j] = X[p][2 i + a][2 j + c]: the gradient).  A thread handles the 2 x 2 output cell of one input position pair.
__global__ __launch_bounds__(256) void pixel_shuffle2_kernel(const float* __restrict__ X, float* __restrict__ Y, int64_t planes, int h, int w, int inverse) {
    const int64_t hw = (int64_t)h * w, total = planes * hw;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        const int64_t pl = idx / hw; const int r = (int)(idx - pl * hw), i = r / w, j = r - i * w;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const int64_t small = (4 * pl + 2 * a + c) * hw + r, big = pl * 4 * hw + (int64_t)(2 * i + a) * (2 * w) + 2 * j + c;
                if (inverse) Y[small] = X[big]; else Y[big] = X[small];
            }
    }
}
extern "C" int segx_pixel_shuffle2(const float* X, float* Y, int64_t planes, int h, int w, int inverse, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    SEGX_REQUIRE(X && Y && planes > 0 && h > 0 && w > 0, "segx_pixel_shuffle2: bad args");
    const int64_t total = planes * h * w;
    hipLaunchKernelGGL(pixel_shuffle2_kernel, dim3((unsigned)i64min(1 << 20, (total + 255) / 256)), dim3(256), 0, stream, X, Y, planes, h, w, inverse);
    return check_launch("segx_pixel_shuffle2");
}

extern "C" int segx_interp_linear_fwd(const float* in, const float* base, float* out, int64_t planes, int d, int h, int w, int D, int H, int W,
                                      void* stream_) {
    SEGX_STREAM; SEGX_REQUIRE(in && out && planes > 0 && d > 0 && h > 0 && w > 0 && D > 0 && H > 0 && W > 0, "segx_interp_linear_fwd: bad args");
    const int64_t total = planes * D * H * W;
    const bool al = ((reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(base)) & 15) == 0;
    if (segx::kget(segx::knobs().interp_variant) != 1 && W % 4 == 0 && W <= 1024 && al && planes <= 65535 && (int64_t)d * h * w < 2147483647LL && (int64_t)D * H < 2147483647LL) {
        const int w4 = W / 4, rpb = 256 / w4;
        hipLaunchKernelGGL(interp_fwd_rows_kernel, dim3((unsigned)((D * H + rpb - 1) / rpb), (unsigned)planes), dim3(256), 0, stream, in, base, out,
                           make_dims(d, h, w, D, H, W), w4, rpb, make_fastdiv(H), make_fastdiv(w4));
    } else {
        hipLaunchKernelGGL(interp_fwd_kernel, dim3((unsigned)i64min(65536, (total + 255) / 256)), dim3(256), 0, stream, in, base, out, make_dims(d, h, w, D, H, W), planes);
    }
    return check_launch("segx_interp_linear_fwd");
}
/* forward along one axis: in [outer, n_in, inner] -> out [outer, n_out, inner] (+ base, same shape as out) */
// fused 3-D forms (interp3d_*_tile_kernel): 1 when launched, 0 when the shape needs the separable passes (nothing launched), < 0 on error
static bool interp3d_tile_ok(int d, int h, int w, int D, int H, int W, int64_t planes) {
    return planes > 0 && planes <= 65535 && d <= D && h <= H && w <= W && 2 * d >= D && 2 * h >= H && 2 * w >= W && (d < D || h < H || w < W) && D > 1;
}
extern "C" int segx_interp3d_fwd_fused(const float* in, const float* base, float* out, int64_t planes, int d, int h, int w, int D, int H, int W, void* stream_) {
    SEGX_STREAM; SEGX_REQUIRE(in && out && d > 0 && h > 0 && w > 0 && D > 0 && H > 0 && W > 0, "segx_interp3d_fwd_fused: bad args");
    if (!interp3d_tile_ok(d, h, w, D, H, W, planes) || W % 4 != 0 || ((reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(base)) & 15) != 0) return 0;
    const int tz = (D + IT_TZ - 1) / IT_TZ, ty = (H + IT_TY - 1) / IT_TY, tx = (W + IT_TX - 1) / IT_TX;
    if ((int64_t)tz * ty * tx > 2147483647LL) return 0;
    hipLaunchKernelGGL(interp3d_fwd_tile_kernel, dim3((unsigned)(tz * ty * tx), (unsigned)planes), dim3(256), 0, stream, in, base, out, make_dims(d, h, w, D, H, W), ty, tx);
    const int rc = check_launch("segx_interp3d_fwd_fused");
    return rc ? (rc > 0 ? -rc - 1000 : rc) : 1;
}
extern "C" int segx_interp3d_bwd_fused(const float* dout, float* din, int64_t planes, int d, int h, int w, int D, int H, int W, void* stream_) {
    SEGX_STREAM; SEGX_REQUIRE(dout && din && d > 0 && h > 0 && w > 0 && D > 0 && H > 0 && W > 0, "segx_interp3d_bwd_fused: bad args");
    if (!interp3d_tile_ok(d, h, w, D, H, W, planes)) return 0;
    const int tz = (d + IB_TZ - 1) / IB_TZ, ty = (h + IB_TY - 1) / IB_TY, tx = (w + IB_TX - 1) / IB_TX;
    if ((int64_t)tz * ty * tx > 2147483647LL) return 0;
    hipLaunchKernelGGL(interp3d_bwd_tile_kernel, dim3((unsigned)(tz * ty * tx), (unsigned)planes), dim3(256), 0, stream, dout, din, make_dims(d, h, w, D, H, W), ty, tx);
    const int rc = check_launch("segx_interp3d_bwd_fused");
    return rc ? (rc > 0 ? -rc - 1000 : rc) : 1;
}
extern "C" int segx_interp_linear_fwd_axis(const float* in, const float* base, float* out, int64_t outer, int n_in, int n_out, int64_t inner,
                                           float src_scale, void* stream_) {
    SEGX_STREAM; SEGX_REQUIRE(in && out && outer > 0 && outer <= 2147483647LL && n_in > 0 && n_out > 0 && inner > 0 && (int64_t)n_out * inner < 2147483647LL,
                              "segx_interp_linear_fwd_axis: bad args");
    const bool vec = inner % 4 == 0 && ((reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(base)) & 15) == 0;
    const int in_ = (int)(vec ? inner / 4 : inner);
    const int64_t per = (int64_t)n_out * in_;
    const int64_t total = outer * per;
    const dim3 grid((unsigned)i64min(1 << 20, (total + 255) / 256));
    const float scale = src_scale != 0.f ? src_scale : (float)n_in / (float)n_out;     // > 0: F.interpolate(scale_factor=1/src_scale); < 0: align_corners=True, |.| = (n_in-1)/(n_out-1)
    if (vec) hipLaunchKernelGGL((interp_fwd_axis_kernel<true>), grid, dim3(256), 0, stream, in, base, out, n_in, n_out, in_, make_fastdiv(in_),
                                make_fastdiv((int)per), scale, outer);
    else hipLaunchKernelGGL((interp_fwd_axis_kernel<false>), grid, dim3(256), 0, stream, in, base, out, n_in, n_out, in_, make_fastdiv(in_),
                            make_fastdiv((int)per), scale, outer);
    return check_launch("segx_interp_linear_fwd_axis");
}
/* RandomResizedCrop (reference dataloaders/datasets3d.py:611-665): resample [planes, d, h, w] to (D, H, W) (trilinear, align_corners=False), zero-pad,
 * crop (od, oh, ow) at offset (oz, oy, ox) of the resampled grid (crop start minus front pad), in one pass. geom (int32[12]) =
 * {d, h, w, D, H, W, od, oh, ow, oz, oy, ox} */
extern "C" int segx_resized_crop3d(const float* X, float* Y, int64_t planes, const int* geom, void* stream_) {
    SEGX_STREAM; SEGX_REQUIRE(X && Y && geom && planes > 0, "segx_resized_crop3d: bad args");
    for (int i = 0; i < 9; ++i) SEGX_REQUIRE(geom[i] > 0, "segx_resized_crop3d: geom[%d] = %d", i, geom[i]);
    SEGX_REQUIRE((int64_t)geom[0] * geom[1] * geom[2] < 2147483647LL, "segx_resized_crop3d: plane too large");
    const int64_t total = planes * geom[6] * geom[7] * geom[8];
    hipLaunchKernelGGL(resized_crop3d_kernel, dim3((unsigned)i64min(1 << 20, (total + 255) / 256)), dim3(256), 0, stream, X, Y,
                       make_dims(geom[0], geom[1], geom[2], geom[3], geom[4], geom[5]), geom[6], geom[7], geom[8], geom[9], geom[10], geom[11], planes);
    return check_launch("segx_resized_crop3d");
}
extern "C" int segx_interp_linear_bwd(const float* dout, float* din, int64_t planes, int d, int h, int w, int D, int H, int W, void* stream_) {
    SEGX_STREAM; SEGX_REQUIRE(dout && din && planes > 0 && d > 0 && h > 0 && w > 0 && D > 0 && H > 0 && W > 0, "segx_interp_linear_bwd: bad args");
    const int64_t total = planes * d * h * w;
    hipLaunchKernelGGL(interp_bwd_kernel, dim3((unsigned)i64min(65536, (total + 255) / 256)), dim3(256), 0, stream, dout, din, make_dims(d, h, w, D, H, W), planes);
    return check_launch("segx_interp_linear_bwd");
}
extern "C" int segx_interp_linear_bwd_axis(const float* dout, float* din, int64_t outer, int n_out, int n_in, int64_t inner, float src_scale,
                                           void* stream_) {
    SEGX_STREAM; SEGX_REQUIRE(dout && din && outer > 0 && n_out > 0 && n_in > 0 && inner > 0, "segx_interp_linear_bwd_axis: bad args");
    const int64_t total = outer * n_in * inner;
    const float scale = src_scale != 0.f ? src_scale : (float)n_in / (float)n_out;
    SEGX_REQUIRE((int64_t)n_in * inner < 2147483647LL - 256, "segx_interp_linear_bwd_axis: slice too large");
    if (inner % 4 == 0 && ((reinterpret_cast<uintptr_t>(dout) | reinterpret_cast<uintptr_t>(din)) & 15) == 0) {
        const int in4 = (int)(inner / 4);
        hipLaunchKernelGGL(interp_bwd_axis4_kernel, dim3((unsigned)i64min(1 << 20, (total / 4 + 255) / 256)), dim3(256), 0, stream, dout, din, outer,
                           n_out, n_in, in4, make_fastdiv(in4), make_fastdiv(n_in * in4), scale);
    } else
        hipLaunchKernelGGL(interp_bwd_axis_kernel, dim3((unsigned)i64min(1 << 20, (total + 255) / 256)), dim3(256), 0, stream, dout, din, outer, n_out,
                           n_in, (int)inner, make_fastdiv((int)inner), make_fastdiv((int)(n_in * inner)), scale);
    return check_launch("segx_interp_linear_bwd_axis");
}
