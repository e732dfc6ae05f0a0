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
// per plane (b,c): psum[3 bc ..] = (sum dy, sum dy * xhat, sum xhat)
__global__ __launch_bounds__(256) void gn_bwd_plane_sums(const float* __restrict__ dY, const float* __restrict__ X, const float* __restrict__ mean,
                                                         const float* __restrict__ rstd, float* __restrict__ psum, int C, int G, int64_t S) {
    __shared__ float red[4];
    const int bc = blockIdx.x, c = bc % C, bg = (bc / C) * G + c / (C / G);
    const float m = mean[bg], r = rstd[bg];
    const float* x = X + (int64_t)bc * S; const float* g = dY + (int64_t)bc * S;
    float a = 0.f, q = 0.f, h = 0.f;
    for (int64_t s = threadIdx.x; s < S; s += 256) { const float gv = g[s], xh = (x[s] - m) * r; a += gv; q += gv * xh; h += xh; }
    a = block_sum<4>(a, red); q = block_sum<4>(q, red); h = block_sum<4>(h, red);
    if (threadIdx.x == 0) { psum[3 * bc] = a; psum[3 * bc + 1] = q; psum[3 * bc + 2] = h; }
}
// tiny: group sums gsum[bg] = (sum_c w_c * psum0, sum_c w_c * psum1) ; dw[c] = sum_b psum1 ; db[c] = sum_b psum0 ; and (r05) the per-plane sums of the
// input gradient the apply pass is about to write, in closed form:  sum_s dx[b,c,s] = rstd * (w_c * psum0 - S * k1 - k2 * psum2)  (k1, k2 as in gn_bwd_apply).
// They are the bias gradient of the pointwise convolution that produced the normalised tensor (segtran3d.py:336-360: conv -> up + add -> GroupNorm): a
// row-sum pass over a 2 - 3.5 GB tensor that need not run (rowsum_kernel: 0.63 / 1.1 ms of the cfg4 / cfg5 step).
__global__ __launch_bounds__(256) void gn_bwd_finalize(const float* __restrict__ psum, const float* __restrict__ w, const float* __restrict__ rstd,
                                                       float* __restrict__ gsum, float* __restrict__ dw, float* __restrict__ db, float* __restrict__ rsum,
                                                       int B, int C, int G, float S) {
    const int t = blockIdx.x * 256 + threadIdx.x, cpg = C / G;
    if (t < B * C) {                                          // every plane's thread forms its group's two sums (cpg terms: redundant, but there is no exchange)
        const int b = t / C, c = t - b * C, g = c / cpg, bg = b * G + g;
        float a = 0.f, q = 0.f;
        for (int cc = g * cpg; cc < (g + 1) * cpg; ++cc) { a += w[cc] * psum[3 * (b * C + cc)]; q += w[cc] * psum[3 * (b * C + cc) + 1]; }
        if (c == g * cpg) { gsum[2 * bg] = a; gsum[2 * bg + 1] = q; }
        if (rsum) {
            const float inv_n = 1.0f / ((float)cpg * S), k1 = a * inv_n, k2 = q * inv_n;
            rsum[t] = rstd[bg] * (w[c] * psum[3 * t] - S * k1 - k2 * psum[3 * t + 2]);
        }
    }
    if (t < C) {
        float a = 0.f, q = 0.f;
        for (int b = 0; b < B; ++b) { a += psum[3 * (b * C + t)]; q += psum[3 * (b * C + t) + 1]; }
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

// r05: GroupNorm statistics from PARTIALS written by the pass that produced the tensor (interp_fwd_axis2_stats_kernel below): parts[bg][P] float4 =
// (count, mean, M2, -) of consecutive runs of the group's elements, merged with Chan's formula -- one wave per (sample, group): lane l folds
// partials l, l + 64, ... in order, then a symmetric butterfly (the same bits in every lane and every run).
struct GnPart { float n, mean, m2; };
__device__ __forceinline__ GnPart gn_merge(const GnPart& a, const GnPart& b) {
    const float n = a.n + b.n;
    if (!(n > 0.f)) return GnPart{0.f, 0.f, 0.f};
    const float d = b.mean - a.mean, f = b.n / n;
    return GnPart{n, a.mean + d * f, a.m2 + b.m2 + d * d * a.n * f};
}
__global__ __launch_bounds__(256) void gn_stats_from_parts(const float* __restrict__ parts, int P, float* __restrict__ mean, float* __restrict__ rstd, int BG, float eps) {
    const int bg = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (bg >= BG) return;
    const float4* p = reinterpret_cast<const float4*>(parts) + (int64_t)bg * P;
    GnPart s{0.f, 0.f, 0.f};
    for (int i = lane; i < P; i += 64) { const float4 v = p[i]; s = gn_merge(s, GnPart{v.x, v.y, v.z}); }
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        GnPart t{__shfl_xor(s.n, o), __shfl_xor(s.mean, o), __shfl_xor(s.m2, o)};
        s = (lane & o) ? gn_merge(t, s) : gn_merge(s, t);          // the lower lane's partial first on both sides: identical operands, identical result
    }
    if (lane == 0) { mean[bg] = s.mean; rstd[bg] = rsqrtf(fmaxf(s.m2 / fmaxf(s.n, 1.f), 0.f) + eps); }
}

// r05: GroupNorm FOLDED into its consumer (segtran_amd/functional.py: _UpGNFold).  Where the normalised tensor's only consumer is a pointwise convolution
// (segtran3d.py:336-367: out_fpn23_conv3d(out_gn2b(.)), the class projection of out_gn3b(.)), y = x * sc[b, c] + sh[b, c] never has to exist:
// conv(y) = (W * sc_b) x + (W sh_b + bias) -- per-sample weights and biases (tiny tensors, host-side autograd).  Forward: no apply pass.  Backward: the
// gradients with respect to sc and sh come out of the per-sample WEIGHT gradient GEMM (which reads x anyway), so the plane-sums pass is gone too; what is left
// is ONE pass: dx = g + A[b, g] + Bc[b, g] * xhat, g = the convolution's data gradient (already carrying sc), A / Bc = the statistics' share of the chain rule.
// Each workgroup also leaves the sum of what it wrote (psum[plane][chunk]): the plane sums of dx are the lateral convolution's bias gradient.
__global__ __launch_bounds__(256) void gn_fold_bwd_apply(const float* __restrict__ Gd, const float* __restrict__ X, const float* __restrict__ mean,
                                                         const float* __restrict__ rstd, const float* __restrict__ A, const float* __restrict__ Bc,
                                                         float* __restrict__ dX, float* __restrict__ psum, int C, int G, int64_t S) {
    __shared__ float red[4];
    const int bc = blockIdx.y, c = bc % C, bg = (bc / C) * G + c / (C / G);
    const float m = mean[bg], r = rstd[bg], a = A[bg], bq = Bc[bg] * r;
    const float* x = X + (int64_t)bc * S; const float* g = Gd + (int64_t)bc * S; float* d = dX + (int64_t)bc * S;
    float acc = 0.f;
    if ((S & 3) == 0 && ((reinterpret_cast<uintptr_t>(X) | reinterpret_cast<uintptr_t>(Gd) | reinterpret_cast<uintptr_t>(dX)) & 15) == 0) {
        const int64_t S4 = S >> 2;
        for (int64_t s = (int64_t)blockIdx.x * 256 + threadIdx.x; s < S4; s += (int64_t)gridDim.x * 256) {
            const float4 gv = reinterpret_cast<const float4*>(g)[s], xv = reinterpret_cast<const float4*>(x)[s];
            float4 o;
            o.x = gv.x + a + bq * (xv.x - m); o.y = gv.y + a + bq * (xv.y - m); o.z = gv.z + a + bq * (xv.z - m); o.w = gv.w + a + bq * (xv.w - m);
            reinterpret_cast<float4*>(d)[s] = o;
            acc += (o.x + o.y) + (o.z + o.w);
        }
    } else {
        for (int64_t s = (int64_t)blockIdx.x * 256 + threadIdx.x; s < S; s += (int64_t)gridDim.x * 256) { const float o = g[s] + a + bq * (x[s] - m); d[s] = o; acc += o; }
    }
    acc = block_sum<4>(acc, red);
    if (threadIdx.x == 0) psum[(int64_t)bc * gridDim.x + blockIdx.x] = acc;
}
// The same pass where the consumer is a projection onto NC <= 8 channels (the class projection behind out_gn3b): its data gradient
// g[b, c, s] = sum_o Wb[b, o, c] dOut[b, o, s] is formed HERE from the NC-channel gradient (a few MB per sample: L2-resident across the planes of the sample)
// instead of being written as a full-size tensor by a K = NC GEMM (3.5 GB at cfg5: 0.86 ms to write, 0.6 ms to read back).
// A workgroup serves CB = 8 consecutive channels of one sample over the same spatial chunk: the NC-channel gradient is loaded ONCE per position and applied to the
// eight planes (one plane per workgroup re-read it from L2 for each of the 832 channels: 13 GB of L2 traffic at cfg5, 3.9 TB/s of useful bytes -- r05_k).
template <int NC>
__global__ __launch_bounds__(256) void gn_fold_bwd_apply_proj(const float* __restrict__ dOut, const float* __restrict__ Wb, const float* __restrict__ X,
                                                              const float* __restrict__ mean, const float* __restrict__ rstd, const float* __restrict__ A,
                                                              const float* __restrict__ Bc, float* __restrict__ dX, float* __restrict__ psum, int C, int G, int64_t S) {
    constexpr int CB = 8;
    __shared__ float red[4];
    const int cblocks = (C + CB - 1) / CB;
    const int b = blockIdx.y / cblocks, c0 = (blockIdx.y - b * cblocks) * CB;
    const int cpg = C / G;
    float w[CB][NC], m[CB], a[CB], bq[CB];
#pragma unroll
    for (int j = 0; j < CB; ++j) {
        const int c = c0 + j < C ? c0 + j : C - 1, bg = b * G + c / cpg;
        m[j] = mean[bg]; a[j] = A[bg]; bq[j] = Bc[bg] * rstd[bg];
#pragma unroll
        for (int o = 0; o < NC; ++o) w[j][o] = Wb[((int64_t)b * NC + o) * C + c];
    }
    const float* go = dOut + (int64_t)b * NC * S;
    float acc[CB];
#pragma unroll
    for (int j = 0; j < CB; ++j) acc[j] = 0.f;
    const bool vec = (S & 3) == 0 && ((reinterpret_cast<uintptr_t>(X) | reinterpret_cast<uintptr_t>(dOut) | reinterpret_cast<uintptr_t>(dX)) & 15) == 0;
    if (vec) {
        const int64_t S4 = S >> 2;
        for (int64_t s = (int64_t)blockIdx.x * 256 + threadIdx.x; s < S4; s += (int64_t)gridDim.x * 256) {
            float4 gv[NC];
#pragma unroll
            for (int o = 0; o < NC; ++o) gv[o] = reinterpret_cast<const float4*>(go + (int64_t)o * S)[s];
#pragma unroll
            for (int j = 0; j < CB; ++j) {
                if (c0 + j >= C) break;
                const int64_t pl = ((int64_t)b * C + c0 + j) * S;
                const float4 xv = reinterpret_cast<const float4*>(X + pl)[s];
                float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int o = 0; o < NC; ++o) { g.x += w[j][o] * gv[o].x; g.y += w[j][o] * gv[o].y; g.z += w[j][o] * gv[o].z; g.w += w[j][o] * gv[o].w; }
                float4 ov;
                ov.x = g.x + a[j] + bq[j] * (xv.x - m[j]); ov.y = g.y + a[j] + bq[j] * (xv.y - m[j]);
                ov.z = g.z + a[j] + bq[j] * (xv.z - m[j]); ov.w = g.w + a[j] + bq[j] * (xv.w - m[j]);
                reinterpret_cast<float4*>(dX + pl)[s] = ov;
                acc[j] += (ov.x + ov.y) + (ov.z + ov.w);
            }
        }
    } else {
        for (int64_t s = (int64_t)blockIdx.x * 256 + threadIdx.x; s < S; s += (int64_t)gridDim.x * 256) {
            float gv[NC];
#pragma unroll
            for (int o = 0; o < NC; ++o) gv[o] = go[(int64_t)o * S + s];
#pragma unroll
            for (int j = 0; j < CB; ++j) {
                if (c0 + j >= C) break;
                const int64_t pl = ((int64_t)b * C + c0 + j) * S;
                float g = 0.f;
#pragma unroll
                for (int o = 0; o < NC; ++o) g += w[j][o] * gv[o];
                const float ov = g + a[j] + bq[j] * (X[pl + s] - m[j]);
                dX[pl + s] = ov; acc[j] += ov;
            }
        }
    }
#pragma unroll
    for (int j = 0; j < CB; ++j) {
        const float t = block_sum<4>(acc[j], red);
        if (threadIdx.x == 0 && c0 + j < C) psum[((int64_t)b * C + c0 + j) * gridDim.x + blockIdx.x] = t;
    }
}
__global__ __launch_bounds__(256) void gn_fold_plane_sums(const float* __restrict__ psum, float* __restrict__ out, int planes, int chunks) {
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= planes) return;
    float s = 0.f;
    for (int k = 0; k < chunks; ++k) s += psum[(int64_t)p * chunks + k];
    out[p] = s;
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

// r05: the CONTIGUOUS axis itself (inner == 1: the x pass of the pyramid's trilinear up-sampling, 109 -> 436 MB at cfg5), n_out % 4 == 0.
// Forward: a thread writes FOUR adjacent outputs as one float4 (the writes are 4/5 of the pass's bytes; the scalar form ran at 1.9 TB/s).  Backward: a
// thread reads the candidates of its cell as aligned float4s of the output row (the reads are 4/5 of the bytes; scalar: 1.7 TB/s).  Same blend
// expressions / the same ascending candidate order with the same weights (zero-weight candidates included) as the scalar kernels: identical bits.
__global__ __launch_bounds__(256) void interp_fwd_x4_kernel(const float* __restrict__ in, const float* __restrict__ base, float* __restrict__ out,
                                                            int n_in, int n_out4, FastDiv divQ, float scale, int64_t rows) {
    const int64_t total = rows * n_out4;
    for (int64_t b0 = (int64_t)blockIdx.x * 256; b0 < total; b0 += (int64_t)gridDim.x * 256) {
        const int64_t r0 = b0 / n_out4;
        const int e0 = (int)(b0 - r0 * n_out4) + threadIdx.x, qr = fdiv(e0, divQ);
        const int64_t row = r0 + qr; const int j = e0 - qr * n_out4;
        if (row >= rows) continue;
        const float* s = in + row * n_in;
        float v[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) { const Axis a = axis_src(4 * j + t, n_in, scale); v[t] = s[a.i0] * (1.f - a.l) + s[a.i1] * a.l; }
        const int64_t o = row * n_out4 + j;
        if (base) { const float4 bv = reinterpret_cast<const float4*>(base)[o]; v[0] += bv.x; v[1] += bv.y; v[2] += bv.z; v[3] += bv.w; }
        reinterpret_cast<float4*>(out)[o] = make_float4(v[0], v[1], v[2], v[3]);
    }
}
__global__ __launch_bounds__(256) void interp_bwd_x4_kernel(const float* __restrict__ dout, float* __restrict__ din, int n_out, int n_in, FastDiv divN,
                                                            float scale, int64_t rows) {
    const int64_t total = rows * n_in;
    for (int64_t b0 = (int64_t)blockIdx.x * 256; b0 < total; b0 += (int64_t)gridDim.x * 256) {
        const int64_t r0 = b0 / n_in;
        const int e0 = (int)(b0 - r0 * n_in) + threadIdx.x, qr = fdiv(e0, divN);
        const int64_t row = r0 + qr; const int i = e0 - qr * n_in;
        if (row >= rows) continue;
        int lo, hi; cand_range(i, n_out, scale, lo, hi);
        const float4* g4 = reinterpret_cast<const float4*>(dout + row * n_out);
        float acc = 0.f;
        for (int d4 = lo & ~3; d4 <= hi; d4 += 4) {
            const float4 v = g4[d4 >> 2];
            const float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int t = 0; t < 4; ++t) { const int d = d4 + t; if (d >= lo && d <= hi) acc += axis_weight(i, d, n_in, scale) * e[t]; }
        }
        din[row * n_in + i] = acc;
    }
}

// r05 (second form of the backward x pass): the candidate range and the weights of a cell depend on its column i only, so a thread keeps ONE column for
// ROWS consecutive rows: the <= 4 x NQ weights (zero where an element of the covered float4s is not a contributor) are evaluated once, and a row costs NQ
// float4 loads, 4 NQ multiply-adds and a store.  (The per-cell form above evaluates 12 blend weights -- ~200 vector instructions -- per output float:
// 1.8 TB/s, ALU-bound, r05_f.)  Same terms in the same ascending order with the same weights as interp_bwd_x4_kernel (identical on the emulator; on the device hipcc may contract the weight expression differently per kernel: an ulp).
template <int NQ, int ROWS>
__global__ __launch_bounds__(256) void interp_bwd_xcol_kernel(const float* __restrict__ dout, float* __restrict__ din, int n_out, int n_in, FastDiv divN,
                                                              float scale, int64_t rows) {
    const int64_t groups = (rows + ROWS - 1) / ROWS, total = groups * n_in;
    for (int64_t b0 = (int64_t)blockIdx.x * 256; b0 < total; b0 += (int64_t)gridDim.x * 256) {
        const int64_t g0 = b0 / n_in;
        const int e0 = (int)(b0 - g0 * n_in) + threadIdx.x, qg = fdiv(e0, divN);
        const int64_t grp = g0 + qg; const int i = e0 - qg * n_in;
        if (grp >= groups) continue;
        int lo, hi; cand_range(i, n_out, scale, lo, hi);
        const int lo4 = lo & ~3;
        float w[4 * NQ];
#pragma unroll
        for (int t = 0; t < 4 * NQ; ++t) { const int d = lo4 + t; w[t] = (d >= lo && d <= hi) ? axis_weight(i, d, n_in, scale) : 0.f; }
        const int q0 = lo4 >> 2, qmax = (n_out >> 2) - 1;
        int qi[NQ];
#pragma unroll
        for (int k = 0; k < NQ; ++k) qi[k] = q0 + k <= qmax ? q0 + k : qmax;          // float4s past the row: clamped, their weights are 0
        const int64_t r0 = grp * ROWS;
#pragma unroll
        for (int rr = 0; rr < ROWS; ++rr) {
            const int64_t row = r0 + rr < rows ? r0 + rr : rows - 1;
            const float4* g4 = reinterpret_cast<const float4*>(dout + row * n_out);
            float4 v[NQ];
#pragma unroll
            for (int k = 0; k < NQ; ++k) v[k] = g4[qi[k]];
            float acc = 0.f;
#pragma unroll
            for (int k = 0; k < NQ; ++k) { acc += w[4 * k] * v[k].x; acc += w[4 * k + 1] * v[k].y; acc += w[4 * k + 2] * v[k].z; acc += w[4 * k + 3] * v[k].w; }
            if (r0 + rr < rows) din[row * n_in + i] = acc;
        }
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
// r04: the two OUTER axes of a trilinear resampling in one streaming pass.  The separable form moves 29 coarse-tensor sizes per 2 x 2 x 2 up-sampling
// forward (x: 1 + 2, y: 2 + 4, z: 4 + 8 + the lateral's 8) and 21 backward; with y and z in one pass over [outer, n1, n2, inner] (inner = the contiguous
// x extent, float4) it is 21 forward (x: 3, yz: 2 + 8 + 8) and 13 backward (zy: 8 + 2, x: 3): the 2 x 2 source rows of an output row are re-read from
// L2, not HBM.  Same blends in the same order (y before z; adjoint: z before y) -> the same numbers as the one-axis passes.  (A fully fused kernel with
// the source tile in LDS was built and measured first, r04_d: 1.35 ms against 0.93 for the separable passes forward and 8 ms backward -- bound by its LDS
// gathers and per-candidate weight arithmetic, not by bytes.  Removed.)
// =================================================================================================
__global__ __launch_bounds__(256) void interp_fwd_axis2_kernel(const float* __restrict__ in, const float* __restrict__ base, float* __restrict__ out,
                                                               int n1_in, int n1_out, int n2_in, int n2_out, int inner4, FastDiv divInner, FastDiv divRow,
                                                               FastDiv divPer, float s1, float s2, int64_t outer) {
    const int row = n2_out * inner4, per = n1_out * row;                    // float4 elements per output slice of axis 1 / per outer index
    const int64_t total = outer * per, in_per = (int64_t)n1_in * n2_in * inner4;
    for (int64_t b0 = (int64_t)blockIdx.x * 256; b0 < total; b0 += (int64_t)gridDim.x * 256) {
        const int64_t ob0 = b0 / per;
        const int e0 = (int)(b0 - ob0 * per) + threadIdx.x, qo = fdiv(e0, divPer);
        const int64_t o = ob0 + qo; const int e = e0 - qo * per;
        if (o >= outer) continue;
        const int i1 = fdiv(e, divRow), r = e - i1 * row, i2 = fdiv(r, divInner), c = r - i2 * inner4;
        const Axis a1 = axis_src(i1, n1_in, s1), a2 = axis_src(i2, n2_in, s2);
        const float4* s = reinterpret_cast<const float4*>(in) + o * in_per + c;
        const float4 v00 = s[((int64_t)a1.i0 * n2_in + a2.i0) * inner4], v01 = s[((int64_t)a1.i0 * n2_in + a2.i1) * inner4];
        const float4 v10 = s[((int64_t)a1.i1 * n2_in + a2.i0) * inner4], v11 = s[((int64_t)a1.i1 * n2_in + a2.i1) * inner4];
        const float l1 = a1.l, l2 = a2.l;
        float4 rr;
        rr.x = (v00.x * (1.f - l2) + v01.x * l2) * (1.f - l1) + (v10.x * (1.f - l2) + v11.x * l2) * l1;
        rr.y = (v00.y * (1.f - l2) + v01.y * l2) * (1.f - l1) + (v10.y * (1.f - l2) + v11.y * l2) * l1;
        rr.z = (v00.z * (1.f - l2) + v01.z * l2) * (1.f - l1) + (v10.z * (1.f - l2) + v11.z * l2) * l1;
        rr.w = (v00.w * (1.f - l2) + v01.w * l2) * (1.f - l1) + (v10.w * (1.f - l2) + v11.w * l2) * l1;
        const int64_t ob = o * per + e;
        if (base) { const float4 bv = reinterpret_cast<const float4*>(base)[ob]; rr.x += bv.x; rr.y += bv.y; rr.z += bv.z; rr.w += bv.w; }
        reinterpret_cast<float4*>(out)[ob] = rr;
    }
}
// r05: the same pass, also leaving GroupNorm partials of what it writes.  A (sample, group) is a run of cpg planes = cpg * per float4; a workgroup owns CH
// consecutive 256-float4 chunks of ONE run (grid (P, B * G), P * CH >= chunks of a run) and reduces them around a pivot (its first value) to one
// (count, mean, M2) partial: the separate statistics pass over the 2 - 3.5 GB pyramid tensors (gn_stats_stage1: 0.6 / 1.0 ms of the cfg4 / cfg5 step) is gone.
// Needs per % 256 == 0 (host check): a chunk never straddles two planes' worth of index arithmetic beyond what the plain kernel does.
template <bool TWO>                                                   // TWO = false: axis 1 is not resized (n1_in == n1_out): the one-axis blend, half the source reads
__global__ __launch_bounds__(256) void interp_fwd_axis2_stats_kernel(const float* __restrict__ in, const float* __restrict__ base, float* __restrict__ out,
                                                                     int n1_in, int n1_out, int n2_in, int n2_out, int inner4, FastDiv divInner, FastDiv divRow,
                                                                     FastDiv divPer, float s1, float s2, int cpg, int CH, float* __restrict__ parts) {
    __shared__ float red[4];
    __shared__ float piv;
    const int row = n2_out * inner4, per = n1_out * row;
    const int64_t in_per = (int64_t)n1_in * n2_in * inner4;
    const int bg = blockIdx.y, P = gridDim.x;
    const int run_chunks = cpg * (per >> 8);                          // 256-float4 chunks of this (sample, group)
    const int ch0 = blockIdx.x * CH, ch1 = min(ch0 + CH, run_chunks);
    float a = 0.f, q = 0.f, pivot = 0.f;
    for (int ch = ch0; ch < ch1; ++ch) {
        const int e_run = (ch << 8) + threadIdx.x;                    // float4 index inside the run
        const int pl = fdiv(e_run, divPer), e = e_run - pl * per;
        const int64_t o = (int64_t)bg * cpg + pl;
        const int i1 = fdiv(e, divRow), r = e - i1 * row, i2 = fdiv(r, divInner), c = r - i2 * inner4;
        const Axis a2 = axis_src(i2, n2_in, s2);
        const float4* s = reinterpret_cast<const float4*>(in) + o * in_per + c;
        const float l2 = a2.l;
        float4 rr;
        if (TWO) {
            const Axis a1 = axis_src(i1, n1_in, s1);
            const float4 v00 = s[((int64_t)a1.i0 * n2_in + a2.i0) * inner4], v01 = s[((int64_t)a1.i0 * n2_in + a2.i1) * inner4];
            const float4 v10 = s[((int64_t)a1.i1 * n2_in + a2.i0) * inner4], v11 = s[((int64_t)a1.i1 * n2_in + a2.i1) * inner4];
            const float l1 = a1.l;
            rr.x = (v00.x * (1.f - l2) + v01.x * l2) * (1.f - l1) + (v10.x * (1.f - l2) + v11.x * l2) * l1;
            rr.y = (v00.y * (1.f - l2) + v01.y * l2) * (1.f - l1) + (v10.y * (1.f - l2) + v11.y * l2) * l1;
            rr.z = (v00.z * (1.f - l2) + v01.z * l2) * (1.f - l1) + (v10.z * (1.f - l2) + v11.z * l2) * l1;
            rr.w = (v00.w * (1.f - l2) + v01.w * l2) * (1.f - l1) + (v10.w * (1.f - l2) + v11.w * l2) * l1;
        } else {                                                      // the blend of interp_fwd_axis_kernel<true> along axis 2 (slice i1 of the source = slice i1 of the output)
            const float4 v0 = s[((int64_t)i1 * n2_in + a2.i0) * inner4], v1 = s[((int64_t)i1 * n2_in + a2.i1) * inner4];
            rr = make_float4(v0.x * (1.f - l2) + v1.x * l2, v0.y * (1.f - l2) + v1.y * l2, v0.z * (1.f - l2) + v1.z * l2, v0.w * (1.f - l2) + v1.w * l2);
        }
        const int64_t ob = o * per + e;
        if (base) { const float4 bv = reinterpret_cast<const float4*>(base)[ob]; rr.x += bv.x; rr.y += bv.y; rr.z += bv.z; rr.w += bv.w; }
        reinterpret_cast<float4*>(out)[ob] = rr;
        if (ch == ch0) {                                              // the pivot: this workgroup's first value (close to everything it will see)
            if (threadIdx.x == 0) piv = rr.x;
            __syncthreads();
            pivot = piv;
        }
        const float d0 = rr.x - pivot, d1 = rr.y - pivot, d2 = rr.z - pivot, d3 = rr.w - pivot;
        a += (d0 + d1) + (d2 + d3); q += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
    }
    a = block_sum<4>(a, red); q = block_sum<4>(q, red);
    if (threadIdx.x == 0) {
        const float n = 1024.0f * (float)(ch1 > ch0 ? ch1 - ch0 : 0);
        reinterpret_cast<float4*>(parts)[(int64_t)bg * P + blockIdx.x] = n > 0.f ? make_float4(n, pivot + a / n, fmaxf(q - a * a / n, 0.f), 0.f) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
}
// adjoint of the pass above: din[o][i1][i2] = sum_{d2} w2(d2) * (sum_{d1} w1(d1) * dout[o][d1][d2]) -- axis 1 (the outer one) innermost, as the one-axis
// passes run (outermost axis first).  Only candidates with a non-zero weight are loaded (cand_range is conservative: 8 per axis for 4 contributors).
__global__ __launch_bounds__(256) void interp_bwd_axis2_kernel(const float* __restrict__ dout, float* __restrict__ din, int n1_out, int n1_in, int n2_out,
                                                               int n2_in, int inner4, FastDiv divInner, FastDiv divRow, FastDiv divPer, float s1, float s2,
                                                               int64_t outer) {
    const int row = n2_in * inner4, per = n1_in * row;
    const int64_t total = outer * per, out_per = (int64_t)n1_out * n2_out * inner4;
    for (int64_t b0 = (int64_t)blockIdx.x * 256; b0 < total; b0 += (int64_t)gridDim.x * 256) {
        const int64_t ob0 = b0 / per;
        const int e0 = (int)(b0 - ob0 * per) + threadIdx.x, qo = fdiv(e0, divPer);
        const int64_t o = ob0 + qo; const int e = e0 - qo * per;
        if (o >= outer) continue;
        const int i1 = fdiv(e, divRow), r = e - i1 * row, i2 = fdiv(r, divInner), c = r - i2 * inner4;
        int lo1, hi1, lo2, hi2;
        cand_range(i1, n1_out, s1, lo1, hi1); cand_range(i2, n2_out, s2, lo2, hi2);
        const float4* g = reinterpret_cast<const float4*>(dout) + o * out_per + c;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int d2 = lo2; d2 <= hi2; ++d2) {
            const float w2 = axis_weight(i2, d2, n2_in, s2);
            if (w2 == 0.f) continue;
            float4 col = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int d1 = lo1; d1 <= hi1; ++d1) {
                const float w1 = axis_weight(i1, d1, n1_in, s1);
                if (w1 == 0.f) continue;
                const float4 v = g[((int64_t)d1 * n2_out + d2) * inner4];
                col.x += w1 * v.x; col.y += w1 * v.y; col.z += w1 * v.z; col.w += w1 * v.w;
            }
            acc.x += w2 * col.x; acc.y += w2 * col.y; acc.z += w2 * col.z; acc.w += w2 * col.w;
        }
        reinterpret_cast<float4*>(din)[o * per + e] = acc;
    }
}

// r05: the same adjoint with the contributing outputs of a cell as a FIXED list.  cand_range is conservative (2 f + 4 candidates per axis for an f-fold
// up-sampling, 2 f of them with a non-zero weight) and the kernel above walks it with `if (w == 0) continue` around every load: 64 weight evaluations and 16
// loads, each under its own branch, per cell at f = 2 -- the loads of one cell waited for each other (3.3 TB/s over the cfg5 pyramid).  Here a lane first
// collects the <= MAXC contributors (index, weight) of each axis in ascending order -- the order the loop above adds them in -- then requests the MAXC float4
// of one d2 column back to back (indices past the list repeat its first entry with weight 0: x + 0 * v == x) and blends.  Same sums in the same order.
// A list of 4 serves ratios up to 2, of 8 up to 4 (chosen per axis); anything else keeps the kernel above (host check: interp_contributors).
template <int MAXC1, int MAXC2>
__global__ __launch_bounds__(256) void interp_bwd_axis2_fixed_kernel(const float* __restrict__ dout, float* __restrict__ din, int n1_out, int n1_in, int n2_out,
                                                                     int n2_in, int inner4, FastDiv divInner, FastDiv divRow, FastDiv divPer, float s1, float s2,
                                                                     int64_t outer) {
    const int row = n2_in * inner4, per = n1_in * row;
    const int64_t total = outer * per, out_per = (int64_t)n1_out * n2_out * inner4;
    for (int64_t b0 = (int64_t)xcd_block(blockIdx.x, gridDim.x) * 256; b0 < total; b0 += (int64_t)gridDim.x * 256) {      // XCD-contiguous runs of blocks (common.h)
        const int64_t ob0 = b0 / per;
        const int e0 = (int)(b0 - ob0 * per) + threadIdx.x, qo = fdiv(e0, divPer);
        const int64_t o = ob0 + qo; const int e = e0 - qo * per;
        if (o >= outer) continue;
        const int i1 = fdiv(e, divRow), r = e - i1 * row, i2 = fdiv(r, divInner), c = r - i2 * inner4;
        int lo1, hi1, lo2, hi2;
        cand_range(i1, n1_out, s1, lo1, hi1); cand_range(i2, n2_out, s2, lo2, hi2);
        int d1s[MAXC1], d2s[MAXC2]; float w1s[MAXC1], w2s[MAXC2];
        int c1 = 0, c2 = 0;
#pragma unroll
        for (int k = 0; k < MAXC1; ++k) { d1s[k] = lo1; w1s[k] = 0.f; }
#pragma unroll
        for (int k = 0; k < MAXC2; ++k) { d2s[k] = lo2; w2s[k] = 0.f; }
        for (int d = lo1; d <= hi1; ++d) {                   // <= 2 f + 4 weight evaluations per axis, no loads
            const float w = axis_weight(i1, d, n1_in, s1);
            if (w != 0.f && c1 < MAXC1) {
#pragma unroll
                for (int k = 0; k < MAXC1; ++k) if (k == c1) { d1s[k] = d; w1s[k] = w; }
                ++c1;
            }
        }
        for (int d = lo2; d <= hi2; ++d) {
            const float w = axis_weight(i2, d, n2_in, s2);
            if (w != 0.f && c2 < MAXC2) {
#pragma unroll
                for (int k = 0; k < MAXC2; ++k) if (k == c2) { d2s[k] = d; w2s[k] = w; }
                ++c2;
            }
        }
#pragma unroll
        for (int k = 0; k < MAXC1; ++k) if (k >= c1) d1s[k] = d1s[0];
#pragma unroll
        for (int k = 0; k < MAXC2; ++k) if (k >= c2) d2s[k] = d2s[0];
        const float4* g = reinterpret_cast<const float4*>(dout) + o * out_per + c;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int k2 = 0; k2 < MAXC2; ++k2) {
            float4 v[MAXC1];
#pragma unroll
            for (int k1 = 0; k1 < MAXC1; ++k1) v[k1] = g[((int64_t)d1s[k1] * n2_out + d2s[k2]) * inner4];
            float4 col = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int k1 = 0; k1 < MAXC1; ++k1) { const float w1 = w1s[k1]; col.x += w1 * v[k1].x; col.y += w1 * v[k1].y; col.z += w1 * v[k1].z; col.w += w1 * v[k1].w; }
            const float w2 = w2s[k2];
            acc.x += w2 * col.x; acc.y += w2 * col.y; acc.z += w2 * col.z; acc.w += w2 * col.w;
        }
        reinterpret_cast<float4*>(din)[o * per + e] = acc;
    }
}
// upper bound of the outputs one input cell contributes to along an axis resampled n_in -> n_out (align_corners = False): src(d) within (i - 1, i + 1)
static inline int interp_contributors(int n_in, int n_out) { return n_out <= n_in ? 3 : (int)((2LL * n_out + n_in - 1) / n_in); }      // ceil(2 f), f = n_out / n_in
// one-axis form of the fixed-list adjoint (float4 over the contiguous inner extent)
template <int MAXC>
__global__ __launch_bounds__(256) void interp_bwd_axis4_fixed_kernel(const float* __restrict__ dout, float* __restrict__ din, int64_t outer,
                                                                     int n_out, int n_in, int inner4, FastDiv divInner, FastDiv divPer, float scale) {
    const int per = n_in * inner4;
    const int64_t total = outer * per;
    for (int64_t b0 = (int64_t)xcd_block(blockIdx.x, gridDim.x) * 256; b0 < total; b0 += (int64_t)gridDim.x * 256) {      // XCD-contiguous runs of blocks (common.h)
        const int64_t ob0 = b0 / per;
        const int e0 = (int)(b0 - ob0 * per) + threadIdx.x, qo = fdiv(e0, divPer);
        const int64_t o = ob0 + qo; const int e = e0 - qo * per;
        if (o >= outer) continue;
        const int i = fdiv(e, divInner), in_ = e - i * inner4;
        int lo, hi; cand_range(i, n_out, scale, lo, hi);
        int ds[MAXC]; float ws[MAXC];
        int cnt = 0;
#pragma unroll
        for (int k = 0; k < MAXC; ++k) { ds[k] = lo; ws[k] = 0.f; }
        for (int d = lo; d <= hi; ++d) {
            const float w = axis_weight(i, d, n_in, scale);
            if (w != 0.f && cnt < MAXC) {
#pragma unroll
                for (int k = 0; k < MAXC; ++k) if (k == cnt) { ds[k] = d; ws[k] = w; }
                ++cnt;
            }
        }
#pragma unroll
        for (int k = 0; k < MAXC; ++k) if (k >= cnt) ds[k] = ds[0];
        const float4* g = reinterpret_cast<const float4*>(dout) + (o * n_out) * inner4 + in_;
        float4 v[MAXC];
#pragma unroll
        for (int k = 0; k < MAXC; ++k) v[k] = g[(int64_t)ds[k] * inner4];
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int k = 0; k < MAXC; ++k) { const float w = ws[k]; acc.x += w * v[k].x; acc.y += w * v[k].y; acc.z += w * v[k].z; acc.w += w * v[k].w; }
        reinterpret_cast<float4*>(din)[o * per + e] = acc;
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

extern "C" int64_t segx_gn_ws_floats(int B, int C, int G) { return (int64_t)B * G * GN_SLABS * 2 + (int64_t)3 * B * C + (int64_t)2 * B * G; }
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
/* GroupNorm whose statistics come from partials left by the pass that wrote X (segx_interp_linear_fwd_axis2 with parts): parts [B * G][nparts] float4 */
extern "C" int segx_groupnorm_fwd_parts(const float* X, const float* parts, int nparts, const float* w, const float* b, float* Y, float* mean, float* rstd,
                                        int B, int C, int G, int64_t S, float eps, void* stream_) {
    SEGX_STREAM; SEGX_REQUIRE(X && parts && nparts > 0 && w && b && Y && mean && rstd && B > 0 && C > 0 && G > 0 && C % G == 0 && S > 0, "segx_groupnorm_fwd_parts: bad args");
    SEGX_REQUIRE((int64_t)B * C <= 65535 && (reinterpret_cast<uintptr_t>(parts) & 15) == 0, "segx_groupnorm_fwd_parts: more than 65535 planes / unaligned partials");
    hipLaunchKernelGGL(gn_stats_from_parts, dim3((B * G + 3) / 4), dim3(256), 0, stream, parts, nparts, mean, rstd, B * G, eps);
    hipLaunchKernelGGL(gn_apply_kernel, dim3(fpn_chunks(S, 8), B * C), dim3(256), 0, stream, X, (const float*)mean, (const float*)rstd, w, b, Y, C, G, S);
    return check_launch("segx_groupnorm_fwd_parts");
}
/* r05: the statistics half of segx_groupnorm_fwd_parts alone (mean / rstd [B * G] from the partials), and the one backward pass of a GroupNorm folded into its
 * consumer: dX = Gd + A[b, g] + Bc[b, g] * xhat; plane_sums [B * C] = the sum of dX over every plane; ws: B * C * 64 floats */
extern "C" int segx_groupnorm_stats_parts(const float* parts, int nparts, float* mean, float* rstd, int BG, float eps, void* stream_) {
    SEGX_STREAM; SEGX_REQUIRE(parts && nparts > 0 && mean && rstd && BG > 0 && (reinterpret_cast<uintptr_t>(parts) & 15) == 0, "segx_groupnorm_stats_parts: bad args");
    hipLaunchKernelGGL(gn_stats_from_parts, dim3((BG + 3) / 4), dim3(256), 0, stream, parts, nparts, mean, rstd, BG, eps);
    return check_launch("segx_groupnorm_stats_parts");
}
/* ... with the consumer's data gradient formed on the fly: dOut [B][NC][S] (NC <= 8), Wb [B][NC][C] per-sample weights */
extern "C" int segx_gn_fold_bwd_proj(const float* dOut, const float* Wb, int NC, const float* X, const float* mean, const float* rstd, const float* A, const float* Bc,
                                     float* dX, float* plane_sums, float* ws, int B, int C, int G, int64_t S, void* stream_) {
    SEGX_STREAM; SEGX_REQUIRE(dOut && Wb && X && mean && rstd && A && Bc && dX && plane_sums && ws && B > 0 && C > 0 && G > 0 && C % G == 0 && S > 0 && NC >= 1 && NC <= 8,
                              "segx_gn_fold_bwd_proj: bad args");
    SEGX_REQUIRE((int64_t)B * C <= 65535, "segx_gn_fold_bwd_proj: more than 65535 planes");
    const int chunks = fpn_chunks((S & 3) ? S : S / 4, 8);
    const dim3 grid(chunks, B * ((C + 7) / 8));                  // a workgroup = one spatial chunk of EIGHT consecutive channels
#define SEGX_GNP(N) case N: hipLaunchKernelGGL((gn_fold_bwd_apply_proj<N>), grid, dim3(256), 0, stream, dOut, Wb, X, mean, rstd, A, Bc, dX, ws, C, G, S); break;
    switch (NC) { SEGX_GNP(1) SEGX_GNP(2) SEGX_GNP(3) SEGX_GNP(4) SEGX_GNP(5) SEGX_GNP(6) SEGX_GNP(7) SEGX_GNP(8) }
#undef SEGX_GNP
    hipLaunchKernelGGL(gn_fold_plane_sums, dim3((B * C + 255) / 256), dim3(256), 0, stream, (const float*)ws, plane_sums, B * C, chunks);
    return check_launch("segx_gn_fold_bwd_proj");
}
extern "C" int segx_gn_fold_bwd(const float* Gd, const float* X, const float* mean, const float* rstd, const float* A, const float* Bc, float* dX, float* plane_sums,
                                float* ws, int B, int C, int G, int64_t S, void* stream_) {
    SEGX_STREAM; SEGX_REQUIRE(Gd && X && mean && rstd && A && Bc && dX && plane_sums && ws && B > 0 && C > 0 && G > 0 && C % G == 0 && S > 0, "segx_gn_fold_bwd: bad args");
    SEGX_REQUIRE((int64_t)B * C <= 65535, "segx_gn_fold_bwd: more than 65535 planes");
    const int chunks = fpn_chunks((S & 3) ? S : S / 4, 8);
    hipLaunchKernelGGL(gn_fold_bwd_apply, dim3(chunks, B * C), dim3(256), 0, stream, Gd, X, mean, rstd, A, Bc, dX, ws, C, G, S);
    hipLaunchKernelGGL(gn_fold_plane_sums, dim3((B * C + 255) / 256), dim3(256), 0, stream, (const float*)ws, plane_sums, B * C, chunks);
    return check_launch("segx_gn_fold_bwd");
}
/* plane_dx_sums (may be NULL): [B * C] floats = sum over the plane of the dX this call writes (closed form from the plane sums: no extra pass) */
extern "C" int segx_groupnorm_bwd(const float* dY, const float* X, const float* w, const float* mean, const float* rstd, float* dX, float* dw,
                                  float* db, float* ws, int B, int C, int G, int64_t S, float* plane_dx_sums, void* stream_) {
    SEGX_STREAM; SEGX_REQUIRE(dY && X && w && mean && rstd && dX && dw && db && ws && B > 0 && C > 0 && G > 0 && C % G == 0 && S > 0, "segx_groupnorm_bwd: bad args");
    SEGX_REQUIRE((int64_t)B * C <= 65535, "segx_groupnorm_bwd: more than 65535 planes");
    float* psum = ws + (int64_t)B * G * GN_SLABS * 2; float* gsum = psum + (int64_t)3 * B * C;
    hipLaunchKernelGGL(gn_bwd_plane_sums, dim3(B * C), dim3(256), 0, stream, dY, X, mean, rstd, psum, C, G, S);
    hipLaunchKernelGGL(gn_bwd_finalize, dim3((B * C + 255) / 256), dim3(256), 0, stream, (const float*)psum, w, rstd, gsum, dw, db, plane_dx_sums, B, C, G, (float)S);
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
    if (knob == 3) { if (value < 0 || value > 2) return -1; k.bn_path = value; return 0; }     // training BatchNorm: 0 resident -> teams -> two launches, 1 no teams, 2 teams everywhere
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
    if (knob == 14) { if (value < 0 || value > 2) return -1; k.pool_slab = value; return 0; }
    if (knob == 15) { if (value != 0 && value != 1) return -1; k.pool_dslide = value; return 0; }
    if (knob == 16) { if (value != 0 && value != 1) return -1; k.conv_halo = value; return 0; }
    if (knob == 17) { if (value < 1 || value > (1 << 24)) return -1; k.conv_halo_min_tiles = value; return 0; }
    if (knob == 18) { if (value != 0 && value != 1) return -1; k.skinny_nt = value; return 0; }
    if (knob == 19) { if (value != 0 && value != 1) return -1; k.tile_walk = value; return 0; }
    if (knob == 12) { if (value < 32 || value > (1 << 24)) return -1; k.team_spin = value; return 0; }     // poll bound of a team exchange
    if (knob == 13) { if (value < 0 || value > 4096) return -1; k.team_drop = value; return 0; }          // fault injection (tests): unlaunched tail of a team grid
    return -1;
}
extern "C" int segx_tune_get(int knob) {
    const segx::Knobs& k = segx::knobs();
    switch (knob) {
        case 1: return segx::kget(k.interp_variant);
        case 2: return segx::kget(k.conv_small_policy);
        case 3: return segx::kget(k.bn_path);
        case 4: return segx::kget(k.engine);
        case 6: return segx::kget(k.x6_variant);
        case 7: return segx::kget(k.conv_x6_wgrad_all);
        case 8: return segx::kget(k.dw_strip_outputs);
        case 9: return segx::kget(k.ws_grid);
        case 12: return segx::kget(k.team_spin);
        case 13: return segx::kget(k.team_drop);
        case 14: return segx::kget(k.pool_slab);
        case 15: return segx::kget(k.pool_dslide);
        case 16: return segx::kget(k.conv_halo);
        case 17: return segx::kget(k.conv_halo_min_tiles);
        case 18: return segx::kget(k.skinny_nt);
        case 19: return segx::kget(k.tile_walk);
        default: return -1;
    }
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
// two outer axes in one pass (interp_{fwd,bwd}_axis2_kernel): tensors [outer, n1, n2, inner] with inner % 4 == 0 and 16-byte aligned pointers
extern "C" int segx_interp_linear_fwd_axis2(const float* in, const float* base, float* out, int64_t outer, int n1_in, int n1_out, int n2_in, int n2_out,
                                            int64_t inner, void* stream_) {
    SEGX_STREAM; SEGX_REQUIRE(in && out && outer > 0 && n1_in > 0 && n1_out > 0 && n2_in > 0 && n2_out > 0 && inner > 0 && inner % 4 == 0, "segx_interp_linear_fwd_axis2: bad args");
    SEGX_REQUIRE(((reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(base)) & 15) == 0, "segx_interp_linear_fwd_axis2: alignment");
    const int64_t in4 = inner / 4, per = (int64_t)n1_out * n2_out * in4, total = outer * per;
    SEGX_REQUIRE(per < 2147483647LL - 256 && (int64_t)n1_in * n2_in * in4 < 2147483647LL, "segx_interp_linear_fwd_axis2: slice too large");
    hipLaunchKernelGGL(interp_fwd_axis2_kernel, dim3((unsigned)i64min(1 << 20, (total + 255) / 256)), dim3(256), 0, stream, in, base, out, n1_in, n1_out, n2_in,
                       n2_out, (int)in4, make_fastdiv((int)in4), make_fastdiv((int)(n2_out * in4)), make_fastdiv((int)per), (float)n1_in / (float)n1_out,
                       (float)n2_in / (float)n2_out, outer);
    return check_launch("segx_interp_linear_fwd_axis2");
}
/* r05: the same pass leaving GroupNorm partials of its output (cpg = channels per group; outer = B * C planes, planes of a group adjacent):
 * parts [outer / cpg][nparts] float4 with nparts = segx_interp_gn_nparts(n1_out * n2_out * inner / 4, cpg) (0: this shape keeps the separate statistics pass) */
static inline int interp_gn_ch(int64_t per4, int cpg) { const int64_t rc = (int64_t)cpg * (per4 >> 8); return (int)((rc + 255) / 256); }       // chunks per workgroup: <= 256 partials per group
extern "C" int64_t segx_interp_gn_nparts(int64_t per4, int cpg) {
    if (per4 <= 0 || cpg <= 0 || (per4 & 255) != 0 || (int64_t)cpg * per4 >= 2147483647LL - 256) return 0;
    const int64_t rc = (int64_t)cpg * (per4 >> 8); const int ch = interp_gn_ch(per4, cpg);
    return (rc + ch - 1) / ch;
}
extern "C" int segx_interp_linear_fwd_axis2_gn(const float* in, const float* base, float* out, int64_t outer, int n1_in, int n1_out, int n2_in, int n2_out,
                                               int64_t inner, int cpg, float* parts, int nparts, void* stream_) {
    SEGX_STREAM; SEGX_REQUIRE(in && out && parts && outer > 0 && n1_in > 0 && n1_out > 0 && n2_in > 0 && n2_out > 0 && inner > 0 && inner % 4 == 0 && cpg > 0 && outer % cpg == 0,
                              "segx_interp_linear_fwd_axis2_gn: bad args");
    SEGX_REQUIRE(((reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(base) | reinterpret_cast<uintptr_t>(parts)) & 15) == 0, "segx_interp_linear_fwd_axis2_gn: alignment");
    const int64_t in4 = inner / 4, per = (int64_t)n1_out * n2_out * in4;
    SEGX_REQUIRE((int64_t)n1_in * n2_in * in4 < 2147483647LL && outer / cpg <= 65535, "segx_interp_linear_fwd_axis2_gn: slice too large");
    SEGX_REQUIRE(nparts > 0 && nparts == segx_interp_gn_nparts(per, cpg), "segx_interp_linear_fwd_axis2_gn: nparts %d is not segx_interp_gn_nparts(%lld, %d)", nparts, (long long)per, cpg);
#define SEGX_AX2S_ARGS in, base, out, n1_in, n1_out, n2_in, n2_out, (int)in4, make_fastdiv((int)in4), make_fastdiv((int)(n2_out * in4)), make_fastdiv((int)per), \
                       (float)n1_in / (float)n1_out, (float)n2_in / (float)n2_out, cpg, interp_gn_ch(per, cpg), parts
    if (n1_in != n1_out) hipLaunchKernelGGL((interp_fwd_axis2_stats_kernel<true>), dim3((unsigned)nparts, (unsigned)(outer / cpg)), dim3(256), 0, stream, SEGX_AX2S_ARGS);
    else hipLaunchKernelGGL((interp_fwd_axis2_stats_kernel<false>), dim3((unsigned)nparts, (unsigned)(outer / cpg)), dim3(256), 0, stream, SEGX_AX2S_ARGS);
#undef SEGX_AX2S_ARGS
    return check_launch("segx_interp_linear_fwd_axis2_gn");
}
extern "C" int segx_interp_linear_bwd_axis2(const float* dout, float* din, int64_t outer, int n1_out, int n1_in, int n2_out, int n2_in, int64_t inner, void* stream_) {
    SEGX_STREAM; SEGX_REQUIRE(dout && din && outer > 0 && n1_in > 0 && n1_out > 0 && n2_in > 0 && n2_out > 0 && inner > 0 && inner % 4 == 0, "segx_interp_linear_bwd_axis2: bad args");
    SEGX_REQUIRE(((reinterpret_cast<uintptr_t>(dout) | reinterpret_cast<uintptr_t>(din)) & 15) == 0, "segx_interp_linear_bwd_axis2: alignment");
    const int64_t in4 = inner / 4, per = (int64_t)n1_in * n2_in * in4, total = outer * per;
    SEGX_REQUIRE(per < 2147483647LL - 256 && (int64_t)n1_out * n2_out * in4 < 2147483647LL, "segx_interp_linear_bwd_axis2: slice too large");
    const int nc1 = interp_contributors(n1_in, n1_out), nc2 = interp_contributors(n2_in, n2_out);
    const bool fixed = kget(knobs().interp_variant) != 1 && nc1 <= 8 && nc2 <= 8;
    const dim3 grid((unsigned)i64min(1 << 20, (total + 255) / 256));
#define SEGX_AXIS2_ARGS dout, din, n1_out, n1_in, n2_out, n2_in, (int)in4, make_fastdiv((int)in4), make_fastdiv((int)(n2_in * in4)), make_fastdiv((int)per), \
                        (float)n1_in / (float)n1_out, (float)n2_in / (float)n2_out, outer
    if (fixed && nc1 <= 4 && nc2 <= 4) hipLaunchKernelGGL((interp_bwd_axis2_fixed_kernel<4, 4>), grid, dim3(256), 0, stream, SEGX_AXIS2_ARGS);
    else if (fixed && nc1 <= 4) hipLaunchKernelGGL((interp_bwd_axis2_fixed_kernel<4, 8>), grid, dim3(256), 0, stream, SEGX_AXIS2_ARGS);
    else if (fixed && nc2 <= 4) hipLaunchKernelGGL((interp_bwd_axis2_fixed_kernel<8, 4>), grid, dim3(256), 0, stream, SEGX_AXIS2_ARGS);
    else if (fixed) hipLaunchKernelGGL((interp_bwd_axis2_fixed_kernel<8, 8>), grid, dim3(256), 0, stream, SEGX_AXIS2_ARGS);
    else hipLaunchKernelGGL(interp_bwd_axis2_kernel, grid, dim3(256), 0, stream, SEGX_AXIS2_ARGS);
#undef SEGX_AXIS2_ARGS
    return check_launch("segx_interp_linear_bwd_axis2");
}
extern "C" int segx_interp_linear_fwd_axis(const float* in, const float* base, float* out, int64_t outer, int n_in, int n_out, int64_t inner,
                                           float src_scale, void* stream_) {
    SEGX_STREAM; SEGX_REQUIRE(in && out && outer > 0 && outer <= 2147483647LL && n_in > 0 && n_out > 0 && inner > 0 && (int64_t)n_out * inner < 2147483647LL,
                              "segx_interp_linear_fwd_axis: bad args");
    const bool vec = inner % 4 == 0 && ((reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(base)) & 15) == 0;
    if (inner == 1 && n_out % 4 == 0 && ((reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(base)) & 15) == 0) {      // the contiguous axis itself: four outputs per thread
        const float sc = src_scale != 0.f ? src_scale : (float)n_in / (float)n_out;
        const int n4 = n_out / 4;
        hipLaunchKernelGGL(interp_fwd_x4_kernel, dim3((unsigned)i64min(1 << 20, (outer * n4 + 255) / 256)), dim3(256), 0, stream, in, base, out, n_in, n4, make_fastdiv(n4), sc, outer);
        return check_launch("segx_interp_linear_fwd_axis/x4");
    }
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
    if (inner == 1 && n_out % 4 == 0 && (reinterpret_cast<uintptr_t>(dout) & 15) == 0) {        // the contiguous axis itself: candidates read as aligned float4s of the output row
        // float4s one cell's candidate range can touch: cand_range spans 2 f + 4 outputs (f = n_out / n_in), plus up to 3 for the alignment of its first one
        // (hi - lo + 1 <= 2 f + 5, + 3 for the alignment of lo: <= ceil((ceil(2 f) + 8) / 4) float4s)
        const int span4 = src_scale == 0.f && n_out >= n_in ? (int)(((2LL * n_out + n_in - 1) / n_in + 8 + 3) / 4) : 99;
        if (span4 <= 4 && outer >= 8 && kget(knobs().interp_variant) != 1) {
            const int64_t tot = ((outer + 7) / 8) * n_in;
            hipLaunchKernelGGL((interp_bwd_xcol_kernel<4, 8>), dim3((unsigned)i64min(1 << 20, (tot + 255) / 256)), dim3(256), 0, stream, dout, din, n_out, n_in, make_fastdiv(n_in), scale, outer);
            return check_launch("segx_interp_linear_bwd_axis/xcol");
        }
        hipLaunchKernelGGL(interp_bwd_x4_kernel, dim3((unsigned)i64min(1 << 20, (total + 255) / 256)), dim3(256), 0, stream, dout, din, n_out, n_in, make_fastdiv(n_in), scale, outer);
        return check_launch("segx_interp_linear_bwd_axis/x4");
    }
    if (inner % 4 == 0 && ((reinterpret_cast<uintptr_t>(dout) | reinterpret_cast<uintptr_t>(din)) & 15) == 0) {
        const int in4 = (int)(inner / 4);
        const dim3 g4((unsigned)i64min(1 << 20, (total / 4 + 255) / 256));
        const int nc = src_scale == 0.f && kget(knobs().interp_variant) != 1 ? interp_contributors(n_in, n_out) : 99;      // the fixed-list form: plain size-ratio resampling only
        if (nc <= 4) hipLaunchKernelGGL((interp_bwd_axis4_fixed_kernel<4>), g4, dim3(256), 0, stream, dout, din, outer, n_out, n_in, in4, make_fastdiv(in4), make_fastdiv(n_in * in4), scale);
        else if (nc <= 8) hipLaunchKernelGGL((interp_bwd_axis4_fixed_kernel<8>), g4, dim3(256), 0, stream, dout, din, outer, n_out, n_in, in4, make_fastdiv(in4), make_fastdiv(n_in * in4), scale);
        else hipLaunchKernelGGL(interp_bwd_axis4_kernel, g4, dim3(256), 0, stream, dout, din, outer,
                           n_out, n_in, in4, make_fastdiv(in4), make_fastdiv(n_in * in4), scale);
    } else
        hipLaunchKernelGGL(interp_bwd_axis_kernel, dim3((unsigned)i64min(1 << 20, (total + 255) / 256)), dim3(256), 0, stream, dout, din, outer, n_out,
                           n_in, (int)inner, make_fastdiv((int)inner), make_fastdiv((int)(n_in * inner)), scale);
    return check_launch("segx_interp_linear_bwd_axis");
}
