// backbone.hip -- HBM-bound kernels of the backbone feature extractors (EfficientNet MBConv blocks K19,
// I3D units K20): BatchNorm(+activation) forward/backward over NC[D]HW tensors, depthwise k3/k5 convolution
// forward / backward-data / backward-weight, squeeze-excite pooling / gating.
//
// None of these has a contraction worth the matrix cores: they are priced against the HBM
// roofline.  Layout rule: a (sample, channel) plane is contiguous (S = D*H*W floats), so a workgroup always
// streams contiguous float4 runs of one plane and carries the per-channel constants in scalars.
#include "common.h"
#include <mutex>
#include <unordered_map>

namespace segx {

enum { ACT_NONE = 0, ACT_SWISH = 1, ACT_RELU = 2, ACT_LEAKY = 3 /* nn.LeakyReLU(0.2): the domain discriminator, networks/discriminator.py:14-21 */ };

__device__ __forceinline__ float sigm(float x) { return 1.0f / (1.0f + __expf(-x)); }
__device__ __forceinline__ float act_fwd(float u, int act) {
    return act == ACT_SWISH ? u * sigm(u) : act == ACT_RELU ? fmaxf(u, 0.f) : act == ACT_LEAKY ? (u > 0.f ? u : 0.2f * u) : u;
}
// d act(u) / du   (swish: efficientnet/utils.py:64-79)
__device__ __forceinline__ float act_grad(float u, int act) {
    if (act == ACT_SWISH) { const float s = sigm(u); return s * (1.0f + u * (1.0f - s)); }
    if (act == ACT_RELU) return u > 0.f ? 1.0f : 0.f;
    if (act == ACT_LEAKY) return u > 0.f ? 1.0f : 0.2f;
    return 1.0f;
}

// per-sample drop_connect scale (efficientnet/utils.py:129-154: floor(keep + U[0,1)) / keep): element `sample` of the Philox stream (seed, offset)
__device__ __forceinline__ float drop_connect_scale(float p, uint64_t seed, uint64_t offset, const uint64_t* __restrict__ rbase, int sample) {
    if (!(p > 0.f)) return 1.0f;
    return dropout_scale(seed, 0, offset + (rbase ? *rbase : 0) + (uint64_t)sample, p, 1.0f / (1.0f - p));
}
// the same for plain sums: ws[c][nparts] pairs (a, q)
__device__ __forceinline__ void bn_fold_sums(const float* __restrict__ ws, int c, int nparts, float& a, float& q) {
    const int lane = threadIdx.x & 63;
    const float2* p = reinterpret_cast<const float2*>(ws) + (int64_t)c * nparts;
    a = 0.f; q = 0.f;
    for (int i = lane; i < nparts; i += 64) { const float2 v = p[i]; a += v.x; q += v.y; }
    a = wave_sum(a); q = wave_sum(q);
}

// =================================================================================================
// BatchNorm (+ activation).  Forward: bn_stats_partial_kernel / bn_act_fwd2_kernel / the channel-resident forms further down (r04).  Backward:
// per-channel reductions in slabs (stage 1: grid (C, B, slabs)), summed by the apply pass.
// =================================================================================================
constexpr int BN_SLABS = 8;               // upper bound (workspace size); the launches use bn_slabs(S) <= BN_SLABS = gridDim.z
// slabs per plane: ~8K floats each, so a 32 x 32 plane is ONE fully populated workgroup instead of eight with 32 live threads
static inline int bn_slabs(int64_t S) { const int64_t n = S / 8192; return (int)(n < 1 ? 1 : n > BN_SLABS ? BN_SLABS : n); }

// backward reductions per channel: sums[c] = (sum du, sum du * xhat), du = dy * act'(u).  grid (C, B, slabs)
__global__ __launch_bounds__(256) void bn_act_bwd_stage1(const float* __restrict__ dY, const float* __restrict__ X, const float* __restrict__ mean,
                                                         const float* __restrict__ var, const float* __restrict__ w, const float* __restrict__ b,
                                                         float* __restrict__ ws, int C, int64_t S, float eps, int act,
                                                         const float* __restrict__ gate, const float* __restrict__ dpool, float inv_S,
                                                         float dc_p, uint64_t seed, uint64_t offset, const uint64_t* __restrict__ rbase, int64_t dy_bs) {
    __shared__ float red[4];
    const int c = blockIdx.x, bb = blockIdx.y, slab = blockIdx.z;
    const float rstd = rsqrtf(var[c] + eps), m = mean[c], wc = w[c], bc_ = b[c];
    // squeeze-excite behind this BatchNorm (bn_act_se): the incoming gradient is dz (w.r.t. y * gate); dy = dz * gate[plane] + dpool[plane] / S is
    // formed on the fly instead of being written by a pass of its own (plane_scale_bwd).  dc_p > 0: the output was scaled by the sample's
    // drop_connect factor before the skip add (bn_act_fwd2_kernel): the same factor multiplies the incoming gradient
    const float gt = (gate ? gate[(int64_t)bb * C + c] : 1.0f) * drop_connect_scale(dc_p, seed, offset, rbase, bb), dp = dpool ? dpool[(int64_t)bb * C + c] * inv_S : 0.f;
    const float* x = X + ((int64_t)bb * C + c) * S; const float* g = dY + (int64_t)bb * dy_bs + (int64_t)c * S;      // dY: a channel slice of a wider tensor (dy_bs = its batch stride)
    const int nsl = gridDim.z;
    const int64_t per = ((S + nsl - 1) / nsl + 3) / 4 * 4, s0 = slab * per, s1 = i64min(S, s0 + per);
    float a = 0.f, q = 0.f;
    if ((S & 3) == 0) {
#pragma unroll 4
        for (int64_t s = s0 + 4 * threadIdx.x; s < s1; s += 1024) {
            const float4 xv = *reinterpret_cast<const float4*>(x + s), gv = *reinterpret_cast<const float4*>(g + s);
            const float h0 = (xv.x - m) * rstd, h1 = (xv.y - m) * rstd, h2 = (xv.z - m) * rstd, h3 = (xv.w - m) * rstd;
            const float d0 = (gv.x * gt + dp) * act_grad(h0 * wc + bc_, act), d1 = (gv.y * gt + dp) * act_grad(h1 * wc + bc_, act);
            const float d2 = (gv.z * gt + dp) * act_grad(h2 * wc + bc_, act), d3 = (gv.w * gt + dp) * act_grad(h3 * wc + bc_, act);
            a += (d0 + d1) + (d2 + d3); q += (d0 * h0 + d1 * h1) + (d2 * h2 + d3 * h3);
        }
    } else {
        for (int64_t s = s0 + threadIdx.x; s < s1; s += 256) {
            const float xh = (x[s] - m) * rstd, du = (g[s] * gt + dp) * act_grad(xh * wc + bc_, act);
            a += du; q += du * xh;
        }
    }
    a = block_sum<4>(a, red); q = block_sum<4>(q, red);
    if (threadIdx.x == 0) { float* o = ws + (((int64_t)c * gridDim.y + bb) * nsl + slab) * 2; o[0] = a; o[1] = q; }
}
__global__ __launch_bounds__(256) void bn_act_bwd_stage2(const float* __restrict__ ws, float* __restrict__ dw, float* __restrict__ db, int B, int C, int nsl) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    float a = 0.f, q = 0.f;
    for (int i = 0; i < B * nsl; ++i) { a += ws[((int64_t)c * B * nsl + i) * 2]; q += ws[((int64_t)c * B * nsl + i) * 2 + 1]; }
    db[c] = a; dw[c] = q;
}
// dx = w * rstd * (du - [training] (db + xhat * dw) / n).  FOLD: the sums come as stage-1 partials ws[c][nparts] (a = sum du, q = sum du * xhat) and
// are added up here (bn_fold_sums); the workgroup of (chunk 0, sample 0) writes them out as the parameter gradients db_out / dw_out.
template <bool FOLD>
__global__ __launch_bounds__(256) void bn_act_bwd_apply(const float* __restrict__ dY, const float* __restrict__ X, const float* __restrict__ mean,
                                                        const float* __restrict__ var, const float* __restrict__ w, const float* __restrict__ b,
                                                        const float* __restrict__ dw, const float* __restrict__ db, float* __restrict__ dX,
                                                        int C, int64_t S, float eps, int act, float inv_n /* 0 in eval mode */,
                                                        const float* __restrict__ gate, const float* __restrict__ dpool, float inv_S,
                                                        float dc_p, uint64_t seed, uint64_t offset, const uint64_t* __restrict__ rbase,
                                                        const float* __restrict__ ws, int nparts, float* __restrict__ dw_out, float* __restrict__ db_out, int64_t dy_bs) {
    const int bc = blockIdx.y, c = bc % C;
    const float rstd = rsqrtf(var[c] + eps), m = mean[c], wc = w[c], bc_ = b[c];
    const float gt = (gate ? gate[bc] : 1.0f) * drop_connect_scale(dc_p, seed, offset, rbase, bc / C), dp = dpool ? dpool[bc] * inv_S : 0.f;       // see bn_act_bwd_stage1
    float sdb, sdw;
    if (FOLD) {
        bn_fold_sums(ws, c, nparts, sdb, sdw);
        if (blockIdx.x == 0 && bc < C && threadIdx.x == 0) { db_out[c] = sdb; dw_out[c] = sdw; }
    } else { sdb = db[c]; sdw = dw[c]; }
    const float k1 = sdb * inv_n, k2 = sdw * inv_n, sc = wc * rstd;
    const float* x = X + (int64_t)bc * S; const float* g = dY + (int64_t)(bc / C) * dy_bs + (int64_t)c * S; float* d = dX + (int64_t)bc * S;
    if ((S & 3) == 0) {
        for (int64_t s = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4; s < S; s += (int64_t)gridDim.x * 1024) {
            const float4 xv = *reinterpret_cast<const float4*>(x + s), gv = *reinterpret_cast<const float4*>(g + s);
            const float h0 = (xv.x - m) * rstd, h1 = (xv.y - m) * rstd, h2 = (xv.z - m) * rstd, h3 = (xv.w - m) * rstd;
            float4 o;
            o.x = sc * ((gv.x * gt + dp) * act_grad(h0 * wc + bc_, act) - k1 - h0 * k2); o.y = sc * ((gv.y * gt + dp) * act_grad(h1 * wc + bc_, act) - k1 - h1 * k2);
            o.z = sc * ((gv.z * gt + dp) * act_grad(h2 * wc + bc_, act) - k1 - h2 * k2); o.w = sc * ((gv.w * gt + dp) * act_grad(h3 * wc + bc_, act) - k1 - h3 * k2);
            *reinterpret_cast<float4*>(d + s) = o;
        }
    } else {
        for (int64_t s = (int64_t)blockIdx.x * 256 + threadIdx.x; s < S; s += (int64_t)gridDim.x * 256) {
            const float xh = (x[s] - m) * rstd, du = (g[s] * gt + dp) * act_grad(xh * wc + bc_, act);
            d[s] = sc * (du - k1 - xh * k2);
        }
    }
}

// =================================================================================================
// r04: training-mode BatchNorm (+ activation, + squeeze-excite pooling, + drop_connect scale and skip add) in TWO launches instead of four / six.
//   launch 1 writes statistics PARTIALS -- (n, mean, M2) of a run of elements, around the run's own first element, so no cancellation --
//   launch 2 (the apply pass) merges the partials of its channel itself (Chan et al.: exact for any split of the data) before it streams the
//   plane: the one-thread-per-channel finalisation kernels (96 + 96 launches of ~5 us per cfg2 step) are gone, and because a partial carries its
//   own pivot, ANY producer of the tensor can emit them (bn_stats_partial_kernel here; the depthwise convolution's epilogue: dw4 STATS).
// Backward: the (sum du, sum du * xhat) partials of stage 1 are summed by the apply pass the same way.
// =================================================================================================
struct BnPart { float n, mean, m2; };
__device__ __forceinline__ BnPart bn_merge(const BnPart& lo, const BnPart& hi) {
    BnPart r;
    r.n = lo.n + hi.n;
    const float inv = r.n > 0.f ? 1.0f / r.n : 0.f, d = hi.mean - lo.mean;
    r.mean = lo.mean + d * (hi.n * inv);
    r.m2 = lo.m2 + hi.m2 + d * d * (lo.n * hi.n * inv);
    return r;
}
// a run's shifted sums (a = sum (x - pivot), q = sum (x - pivot)^2 over n elements) as a partial
__device__ __forceinline__ BnPart bn_part_of(float n, float a, float q, float pivot) {
    BnPart r; r.n = n;
    const float md = n > 0.f ? a / n : 0.f;
    r.mean = pivot + md; r.m2 = fmaxf(q - a * md, 0.f);
    return r;
}
// All partials of channel c (parts[c][nparts] as float4 (n, mean, M2, -)) merged by ONE wave; every wave of a workgroup does it redundantly (no LDS,
// no barrier).  Lane l folds partials l, l + 64, ... in index order, then a butterfly in which both partners merge (lower lane, higher lane): the
// result is bit-identical in every lane, every wave and every workgroup.
__device__ __forceinline__ BnPart bn_fold(const float* __restrict__ parts, int c, int nparts, int64_t cstride = -1, int64_t istride = 1) {
    // default layout parts[c][nparts]; (cstride, istride) = (1, C): parts[nparts][C] -- the all-gathered per-rank partials of synchronised BatchNorm
    const int lane = threadIdx.x & 63;
    const float4* p = reinterpret_cast<const float4*>(parts) + (int64_t)c * (cstride < 0 ? nparts : cstride);
    BnPart s; s.n = 0.f; s.mean = 0.f; s.m2 = 0.f;
    for (int i = lane; i < nparts; i += 64) { const float4 v = p[(int64_t)i * istride]; BnPart t; t.n = v.x; t.mean = v.y; t.m2 = v.z; s = bn_merge(s, t); }
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        BnPart t; t.n = __shfl_xor(s.n, o); t.mean = __shfl_xor(s.mean, o); t.m2 = __shfl_xor(s.m2, o);
        s = (lane & o) ? bn_merge(t, s) : bn_merge(s, t);
    }
    return s;
}
// the same fold over partials written by OTHER workgroups of this launch (team BatchNorm): coherent loads (common.h: team_load)
__device__ __forceinline__ BnPart bn_fold_team(const float* slots, int nparts) {
    const int lane = threadIdx.x & 63;
    const float* p = slots;
    BnPart s; s.n = 0.f; s.mean = 0.f; s.m2 = 0.f;
    for (int i = lane; i < nparts; i += 64) { BnPart t; t.n = team_load(p + TEAM_SLOT * i); t.mean = team_load(p + TEAM_SLOT * i + 1); t.m2 = team_load(p + TEAM_SLOT * i + 2); s = bn_merge(s, t); }
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        BnPart t; t.n = __shfl_xor(s.n, o); t.mean = __shfl_xor(s.mean, o); t.m2 = __shfl_xor(s.m2, o);
        s = (lane & o) ? bn_merge(t, s) : bn_merge(s, t);
    }
    return s;
}
// launch 1: grid (C, B, slabs) as bn_stats_stage1; parts[c][b * nsl + slab]
__global__ __launch_bounds__(256) void bn_stats_partial_kernel(const float* __restrict__ X, float* __restrict__ parts, int C, int64_t S) {
    __shared__ float red[4];
    const int c = blockIdx.x, b = blockIdx.y, slab = blockIdx.z, nsl = gridDim.z;
    const float* x = X + ((int64_t)b * C + c) * S;
    const int64_t per = ((S + nsl - 1) / nsl + 3) / 4 * 4, s0 = slab * per, s1 = i64min(S, s0 + per);
    const float pivot = s0 < S ? x[s0] : 0.f;
    float a = 0.f, q = 0.f;
    if ((S & 3) == 0 && ((reinterpret_cast<uintptr_t>(X) & 15) == 0)) {
#pragma unroll 4
        for (int64_t s = s0 + 4 * threadIdx.x; s < s1; s += 1024) {
            const float4 v = *reinterpret_cast<const float4*>(x + s);
            const float d0 = v.x - pivot, d1 = v.y - pivot, d2 = v.z - pivot, d3 = v.w - pivot;
            a += (d0 + d1) + (d2 + d3); q += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
        }
    } else {
        for (int64_t s = s0 + threadIdx.x; s < s1; s += 256) { const float d = x[s] - pivot; a += d; q += d * d; }
    }
    a = block_sum<4>(a, red); q = block_sum<4>(q, red);
    if (threadIdx.x == 0) {
        const BnPart r = bn_part_of((float)(s1 > s0 ? s1 - s0 : 0), a, q, pivot);
        reinterpret_cast<float4*>(parts)[((int64_t)c * gridDim.y + b) * nsl + slab] = make_float4(r.n, r.mean, r.m2, 0.f);
    }
}
// many partials per channel (a producer with small tiles on a large map): merge them once, one wave per channel, into parts_out[c][1]
__global__ __launch_bounds__(256) void bn_parts_merge_kernel(const float* __restrict__ parts, float* __restrict__ parts_out, int C, int nparts) {
    const int c = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (c >= C) return;
    const BnPart s = bn_fold(parts, c, nparts);
    if ((threadIdx.x & 63) == 0) reinterpret_cast<float4*>(parts_out)[c] = make_float4(s.n, s.mean, s.m2, 0.f);
}
struct BnFwdArgs {
    const float* X; const float* w; const float* b; float* Y;
    const float* parts; int nparts;              // training: statistics partials (bn_fold); NULL: use mean / var as given (running statistics)
    int64_t part_cstride, part_istride;          // layout of parts: (-1, 1) = [C][nparts]; (1, C) = [nparts][C] (gathered per-rank partials)
    float* mean; float* var;                     // training: OUT (saved for backward); else IN
    float* run_mean; float* run_var; float momentum;
    float* psum;                                 // POOL: psum[plane][chunk] = sum of this chunk's outputs (the squeeze-excite pooling, model.py:106)
    const float* resid;                          // y = act(bn(x)) * dcs[sample] + resid   (MBConv skip connection, model.py:118-122); NULL: none
    float dc_p; uint64_t seed, offset; const uint64_t* rbase;
    int C; int64_t S; float eps; int act;
    int64_t y_bs;                                // floats between consecutive samples of Y: C * S (dense) or the batch stride of the wider tensor Y is a channel slice of
};
// launch 2: grid (chunks, B * C).  The workgroup of (chunk 0, sample 0) also writes mean / var and updates the running statistics.
template <bool POOL>
__global__ __launch_bounds__(256) void bn_act_fwd2_kernel(BnFwdArgs g) {
    __shared__ float red[4];
    const int bc = blockIdx.y, C = g.C, c = bc % C, bb = bc / C;
    const int64_t S = g.S;
    float m, v;
    if (g.parts) {
        const BnPart st = bn_fold(g.parts, c, g.nparts, g.part_cstride, g.part_istride);
        m = st.mean; v = st.n > 0.f ? fmaxf(st.m2 / st.n, 0.f) : 0.f;
        if (blockIdx.x == 0 && bb == 0 && threadIdx.x == 0) {
            g.mean[c] = m; g.var[c] = v;
            if (g.run_mean) {
                g.run_mean[c] = (1.0f - g.momentum) * g.run_mean[c] + g.momentum * m;
                g.run_var[c] = (1.0f - g.momentum) * g.run_var[c] + g.momentum * (st.m2 / fmaxf(st.n - 1.0f, 1.0f));
            }
        }
    } else { m = g.mean[c]; v = g.var[c]; }
    const float sc = rsqrtf(v + g.eps) * g.w[c], sh = g.b[c] - m * sc;
    const float dcs = drop_connect_scale(g.dc_p, g.seed, g.offset, g.rbase, bb);
    const float* x = g.X + (int64_t)bc * S; float* y = g.Y + (int64_t)bb * g.y_bs + (int64_t)c * S;
    const float* r = g.resid ? g.resid + (int64_t)bc * S : nullptr;
    const int act = g.act;
    float acc = 0.f;
    if ((S & 3) == 0) {
        for (int64_t s = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4; s < S; s += (int64_t)gridDim.x * 1024) {
            const float4 xv = *reinterpret_cast<const float4*>(x + s);
            float4 o;
            o.x = act_fwd(xv.x * sc + sh, act); o.y = act_fwd(xv.y * sc + sh, act); o.z = act_fwd(xv.z * sc + sh, act); o.w = act_fwd(xv.w * sc + sh, act);
            if (r) { const float4 rv = *reinterpret_cast<const float4*>(r + s); o.x = o.x * dcs + rv.x; o.y = o.y * dcs + rv.y; o.z = o.z * dcs + rv.z; o.w = o.w * dcs + rv.w; }
            *reinterpret_cast<float4*>(y + s) = o;
            if (POOL) acc += (o.x + o.y) + (o.z + o.w);
        }
    } else {
        for (int64_t s = (int64_t)blockIdx.x * 256 + threadIdx.x; s < S; s += (int64_t)gridDim.x * 256) {
            float o = act_fwd(x[s] * sc + sh, act);
            if (r) o = o * dcs + r[s];
            y[s] = o; if (POOL) acc += o;
        }
    }
    if (POOL) {
        acc = block_sum<4>(acc, red);
        if (threadIdx.x == 0) g.psum[(int64_t)bc * gridDim.x + blockIdx.x] = acc;
    }
}

// =================================================================================================
// r04: CHANNEL-RESIDENT BatchNorm for small planes.  79 of EfficientNet-B4's 96 BatchNorm layers at 512 x 512 / batch 6 work on planes of <= 4096
// floats (down to 16 x 16); their nine tensor passes move 1.5 ms worth of bytes per step but took ~11 ms: one workgroup per (sample, channel)
// plane means up to 16 128 workgroups of 256 threads for 256 floats each, in four launches forward and backward.  Here a TEAM of threads (one
// wave: four channels per workgroup; or the whole workgroup) owns ONE channel with all its B planes in registers:
//   forward : load once -> mean -> variance around the mean (two passes over registers: exact, no pivot needed) -> normalise / activate /
//             drop_connect scale + skip add / squeeze-excite pooling -> store.      ONE launch, 1 read + 1 write  (was 2 launches, 2 reads + 1 write)
//   backward: load x and dy once -> du, xhat in registers -> the two channel sums -> dx.   ONE launch, 2 reads + 1 write  (was 2 launches, 4 + 1)
// A lane owns float4 j = lane + TEAM * k (k < KP) of every plane b < BMAX (predicated on b < B and j < S / 4), so the plane of a register is a
// compile-time index and S needs no special form beyond S % 4 == 0.
// =================================================================================================
template <int TEAM> __device__ __forceinline__ float team_sum(float v, float* red) { return TEAM == 64 ? wave_sum(v) : block_sum<4>(v, red); }
// ACT >= 0: the activation is a compile-time constant of the body (straight-line code); ACT < 0: the runtime value
template <int ACT> __device__ __forceinline__ float act_fwd_c(float u, int act) { return act_fwd(u, ACT >= 0 ? ACT : act); }
template <int ACT> __device__ __forceinline__ float act_grad_c(float u, int act) { return act_grad(u, ACT >= 0 ? ACT : act); }

// r04-g: every load of a phase is issued before the first value is used.  The first version predicated each load (`ok ? *p : 0`) and switched on the
// activation per element: hipcc turned that into branch + load + s_waitcnt vmcnt(0) per float4 -- 32 dependent HBM round trips per workgroup,
// 27 us for a 98 KB channel however few channels the layer has (r04_e trace: 112 channels 27.4 us, 672 channels 28.7 us).  Now the loads go to
// CLAMPED addresses unconditionally (the select happens after a scheduling barrier) and the normalise / activate / store phase is instantiated
// per activation, so it is straight-line code.

// normalise / activate / (drop_connect scale + skip) / store / pool for the planes held in v.  Addresses: a wave-uniform base per plane (SGPRs) + one
// 32-bit byte offset per float4 of a lane (off[k], shared by all planes) -- no 64-bit address arithmetic or address registers on the vector pipe.
template <int TEAM, int KP, int BMAX, bool POOL, int ACT, int RESID>      // RESID: 0 none, 1 skip connection, -1 decided at run time (g.resid)
__device__ __forceinline__ void bn_res_apply(const BnFwdArgs& g, int B, const f32x4 (&v)[BMAX][KP], const unsigned (&off)[KP], float sc, float sh, int tl, int c, float* red) {
    const int C = g.C, S4 = (int)(g.S >> 2), act = g.act;
    const int64_t S = g.S;
    f32x4 rv[2][KP];                                             // the skip connection's plane b + 1 is in flight while plane b is computed
    const bool resid = RESID < 0 ? g.resid != nullptr : RESID != 0;
    if (resid) {
        const ws_gptr rb = ws_uniform_base(g.resid + (int64_t)c * S);
#pragma unroll
        for (int k = 0; k < KP; ++k) rv[0][k] = ws_load<f32x4>(rb, off[k]);
    }
#pragma unroll
    for (int b = 0; b < BMAX; ++b) {
        if (b >= B) break;
        const float dcs = drop_connect_scale(g.dc_p, g.seed, g.offset, g.rbase, b);
        const ws_gptr_w yb = ws_uniform_base_w(g.Y + (int64_t)b * g.y_bs + (int64_t)c * S);
        if (resid && b + 1 < BMAX) {
            const ws_gptr rb = ws_uniform_base(g.resid + ((int64_t)(b + 1 < B ? b + 1 : b) * C + c) * S);
#pragma unroll
            for (int k = 0; k < KP; ++k) rv[(b + 1) & 1][k] = ws_load<f32x4>(rb, off[k]);
        }
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < KP; ++k) {
            f32x4 o;
            o.x = act_fwd_c<ACT>(v[b][k].x * sc + sh, act); o.y = act_fwd_c<ACT>(v[b][k].y * sc + sh, act);
            o.z = act_fwd_c<ACT>(v[b][k].z * sc + sh, act); o.w = act_fwd_c<ACT>(v[b][k].w * sc + sh, act);
            if (resid) {
                const f32x4 r = rv[b & 1][k];
                o.x = o.x * dcs + r.x; o.y = o.y * dcs + r.y; o.z = o.z * dcs + r.z; o.w = o.w * dcs + r.w;
            }
            if (tl + TEAM * k < S4) {
                ws_store<f32x4>(yb, off[k], o);
                if (POOL) acc += (o.x + o.y) + (o.z + o.w);
            }
        }
        if (POOL) {
            acc = team_sum<TEAM>(acc, red);
            if (tl == 0) g.psum[(int64_t)b * C + c] = acc;          // ONE chunk per plane (segx_bn_pool_chunks)
        }
        __builtin_amdgcn_sched_barrier(0);                       // one plane at a time: interleaving all planes' arithmetic costs registers (occupancy), buys nothing
    }
}

// STATS: stop after the statistics and leave ONE partial (n, mean, M2) per channel in g.psum[c] -- the local half of synchronised BatchNorm
// ACT / RESID are KERNEL parameters: one straight-line body per kernel.  (Branching inside one kernel into per-activation bodies made the register
// allocator spill the resident planes at the loads: 832 bytes per lane in the largest backward form.)  ACT = -1 / RESID = -1: the run-time values.
template <int TEAM, int KP, int BMAX, bool POOL, int ACT, int RESID, bool STATS = false>
// waves per SIMD: the state is KP * BMAX float4 (+ two planes of the skip connection in flight where there is one: the 24-float4 form with a skip spilled
// 76 bytes per lane under the 168-register cap of three waves)
__global__ __launch_bounds__(256) SEGX_MIN_WAVES_PER_SIMD(KP * BMAX <= 8 ? 4 : (KP * BMAX <= 16 || (KP * BMAX <= 24 && RESID == 0)) ? 3 : 2) void bn_act_fwd_res_kernel(BnFwdArgs g, int B) {
    __shared__ float red[4];
    const int tl = TEAM == 64 ? (threadIdx.x & 63) : threadIdx.x;
    const int c = TEAM == 64 ? blockIdx.x * 4 + (threadIdx.x >> 6) : blockIdx.x;
    if (TEAM == 64 && c >= g.C) return;                        // whole waves leave together; no workgroup barrier in the wave-team form
    const int C = g.C, S4 = (int)(g.S >> 2);
    const int64_t S = g.S;
    unsigned off[KP];
#pragma unroll
    for (int k = 0; k < KP; ++k) { const int j = tl + TEAM * k; off[k] = 16u * (unsigned)(j < S4 ? j : 0); }
    f32x4 v[BMAX][KP];
#pragma unroll
    for (int b = 0; b < BMAX; ++b) {
        const ws_gptr xb = ws_uniform_base(g.X + ((int64_t)(b < B ? b : 0) * C + c) * S);
#pragma unroll
        for (int k = 0; k < KP; ++k) v[b][k] = ws_load<f32x4>(xb, off[k]);
    }
    __builtin_amdgcn_sched_barrier(0);
    float s = 0.f;
#pragma unroll
    for (int b = 0; b < BMAX; ++b)
#pragma unroll
        for (int k = 0; k < KP; ++k) {
            const bool ok = b < B && tl + TEAM * k < S4;
            if (!ok) v[b][k] = f32x4{0.f, 0.f, 0.f, 0.f};
            s += (v[b][k].x + v[b][k].y) + (v[b][k].z + v[b][k].w);
        }
    const float n = (float)B * (float)S;
    const float m = team_sum<TEAM>(s, red) / n;
    float q = 0.f;
#pragma unroll
    for (int b = 0; b < BMAX; ++b)
#pragma unroll
        for (int k = 0; k < KP; ++k) {
            const bool ok = b < B && tl + TEAM * k < S4;
            const float d0 = v[b][k].x - m, d1 = v[b][k].y - m, d2 = v[b][k].z - m, d3 = v[b][k].w - m;
            q += ok ? (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3) : 0.f;
        }
    q = team_sum<TEAM>(q, red);
    if (STATS) {
        if (tl == 0) reinterpret_cast<float4*>(g.psum)[c] = make_float4(n, m, q, 0.f);
        return;
    }
    const float var = q / n;
    if (tl == 0) {
        g.mean[c] = m; g.var[c] = var;
        if (g.run_mean) {
            g.run_mean[c] = (1.0f - g.momentum) * g.run_mean[c] + g.momentum * m;
            g.run_var[c] = (1.0f - g.momentum) * g.run_var[c] + g.momentum * (q / fmaxf(n - 1.0f, 1.0f));
        }
    }
    const float sc = rsqrtf(var + g.eps) * g.w[c], sh = g.b[c] - m * sc;
    bn_res_apply<TEAM, KP, BMAX, POOL, ACT, RESID>(g, B, v, off, sc, sh, tl, c, red);
}

struct BnBwdArgs {
    const float* dY; const float* X; const float* mean; const float* var; const float* w; const float* b;
    float* dX; float* dw; float* db;
    const float* gate; const float* dpool; float inv_S;
    float dc_p; uint64_t seed, offset; const uint64_t* rbase;
    int C; int64_t S; float eps; int act;
    int64_t dy_bs;                               // batch stride of dY (a channel slice of a wider tensor reads in place); C * S when dense
};
// (x, dy) in h / d  ->  (xhat, du) in place -> the two channel sums -> dx
template <int TEAM, int KP, int BMAX, int ACT>
__device__ __forceinline__ void bn_res_bwd_tail(const BnBwdArgs& g, int B, f32x4 (&h)[BMAX][KP], f32x4 (&d)[BMAX][KP], const unsigned (&off)[KP], int tl, int c, float* red) {
    const int C = g.C, S4 = (int)(g.S >> 2), act = g.act;
    const int64_t S = g.S;
    const float rstd = rsqrtf(g.var[c] + g.eps), m = g.mean[c], wc = g.w[c], bc_ = g.b[c];
    float a = 0.f, q = 0.f;
#pragma unroll
    for (int b = 0; b < BMAX; ++b) {
        const bool bok = b < B;
        const int pl = (bok ? b : 0) * C + c;
        const float gt = (g.gate ? g.gate[pl] : 1.0f) * drop_connect_scale(g.dc_p, g.seed, g.offset, g.rbase, b), dp = g.dpool ? g.dpool[pl] * g.inv_S : 0.f;
#pragma unroll
        for (int k = 0; k < KP; ++k) {
            const bool ok = bok && tl + TEAM * k < S4;
            const f32x4 xv = h[b][k], gv = d[b][k];
            f32x4 hh, dd;
            hh.x = (xv.x - m) * rstd; hh.y = (xv.y - m) * rstd; hh.z = (xv.z - m) * rstd; hh.w = (xv.w - m) * rstd;
            dd.x = (gv.x * gt + dp) * act_grad_c<ACT>(hh.x * wc + bc_, act); dd.y = (gv.y * gt + dp) * act_grad_c<ACT>(hh.y * wc + bc_, act);
            dd.z = (gv.z * gt + dp) * act_grad_c<ACT>(hh.z * wc + bc_, act); dd.w = (gv.w * gt + dp) * act_grad_c<ACT>(hh.w * wc + bc_, act);
            if (!ok) { dd = f32x4{0.f, 0.f, 0.f, 0.f}; hh = dd; }
            h[b][k] = hh; d[b][k] = dd;
            a += (dd.x + dd.y) + (dd.z + dd.w); q += (dd.x * hh.x + dd.y * hh.y) + (dd.z * hh.z + dd.w * hh.w);
        }
        __builtin_amdgcn_sched_barrier(0);                       // one plane at a time (registers)
    }
    a = team_sum<TEAM>(a, red); q = team_sum<TEAM>(q, red);
    if (tl == 0) { g.db[c] = a; g.dw[c] = q; }
    const float inv_n = 1.0f / ((float)B * (float)S), k1 = a * inv_n, k2 = q * inv_n, sc = wc * rstd;
#pragma unroll
    for (int b = 0; b < BMAX; ++b) {
        if (b >= B) break;
        const ws_gptr_w ob = ws_uniform_base_w(g.dX + ((int64_t)b * C + c) * S);
#pragma unroll
        for (int k = 0; k < KP; ++k) {
            if (tl + TEAM * k < S4) {
                f32x4 o;
                o.x = sc * (d[b][k].x - k1 - h[b][k].x * k2); o.y = sc * (d[b][k].y - k1 - h[b][k].y * k2);
                o.z = sc * (d[b][k].z - k1 - h[b][k].z * k2); o.w = sc * (d[b][k].w - k1 - h[b][k].w * k2);
                ws_store<f32x4>(ob, off[k], o);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}
template <int TEAM, int KP, int BMAX, int ACT>
__global__ __launch_bounds__(256) SEGX_MIN_WAVES_PER_SIMD(KP * BMAX <= 8 ? (ACT >= 0 ? 4 : 3) : 2) void bn_act_bwd_res_kernel(BnBwdArgs g, int B) {     // run-time activation: 40 bytes spilled at four waves
    __shared__ float red[4];
    const int tl = TEAM == 64 ? (threadIdx.x & 63) : threadIdx.x;
    const int c = TEAM == 64 ? blockIdx.x * 4 + (threadIdx.x >> 6) : blockIdx.x;
    if (TEAM == 64 && c >= g.C) return;
    const int C = g.C, S4 = (int)(g.S >> 2);
    const int64_t S = g.S;
    unsigned off[KP];
#pragma unroll
    for (int k = 0; k < KP; ++k) { const int j = tl + TEAM * k; off[k] = 16u * (unsigned)(j < S4 ? j : 0); }
    f32x4 h[BMAX][KP], d[BMAX][KP];           // x, dy as loaded; then xhat, du
#pragma unroll
    for (int b = 0; b < BMAX; ++b) {
        const int bb = b < B ? b : 0;
        const ws_gptr xb = ws_uniform_base(g.X + ((int64_t)bb * C + c) * S), gb = ws_uniform_base(g.dY + (int64_t)bb * g.dy_bs + (int64_t)c * S);
#pragma unroll
        for (int k = 0; k < KP; ++k) { h[b][k] = ws_load<f32x4>(xb, off[k]); d[b][k] = ws_load<f32x4>(gb, off[k]); }
    }
    __builtin_amdgcn_sched_barrier(0);
    bn_res_bwd_tail<TEAM, KP, BMAX, ACT>(g, B, h, d, off, tl, c, red);
}
// =================================================================================================
// r04-h: TEAM BatchNorm for the planes that do not fit one workgroup (S >= 16384 floats at batch 6: 30 of EfficientNet-B4's 96 layers, but 5.6 of
// the 8.5 GB one pass over all BatchNorm inputs moves).  A team of B * cpp workgroups owns a channel; a workgroup keeps ITS chunk of one plane
// (256 x KP float4) in registers across a team exchange (common.h: team_exchange):
//   forward : load -> (n, mean, M2) of the chunk -> global partial -> barrier -> every member folds the team's partials (bn_fold: bit-identical in
//             every workgroup) -> normalise / activate / skip / pool -> store.       1 read + 1 write   (two launches: 2 reads + 1 write)
//   backward: load x, dy -> xhat, du in registers -> the chunk's two sums -> barrier -> the team's sums in member order -> dx.
//                                                                                     2 reads + 1 write  (two launches: 4 reads + 1 write)
// One launch each, nothing to initialise (the exchange marks its words with a per-launch tag).  Not used by the synchronised form (the exchange between ranks
// sits where the team exchange is).
// =================================================================================================
// the tag of a team launch (common.h: team_exchange): unique per launch of this process, never (0, 0)
static inline void team_next_tag(unsigned& lo, unsigned& hi) {
    // splitmix64 of a launch counter: distinct per launch AND unlike what recycled memory holds.  (A plain counter is not: r04_v -- small integers left
    // behind by int64 tensors matched the tags 1, 2, 3 ... of a fresh process and a member read its mailbox before it was posted.)
    static std::atomic<uint64_t> n{1};
    for (;;) {
        uint64_t z = n.fetch_add(1, std::memory_order_relaxed) * 0x9E3779B97F4A7C15ull;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; z ^= z >> 31;
        lo = (unsigned)z; hi = (unsigned)(z >> 32);
        if (lo && hi) return;
    }
}
// slots: [C][B * cpp][TEAM_SLOT] floats, mbox: [C][B * cpp][TEAM_MBOX] (common.h: team_exchange); err / spin: the process's error word and the poll bound
struct BnTeam { float* slots; float* mbox; unsigned tag_lo, tag_hi; int B, cpp; unsigned* err; unsigned spin; };
// Host state of the team launches.  The ERROR WORD is ONE word of pinned host memory per process, PORTABLE and mapped (every device of the process can add to
// it): a kernel whose poll expires adds one (system-scope atomic), segx_team_status() reads it on the host without synchronising anything.  cap = the largest
// team a device is given: half the compute units the runtime reports FOR THE DEVICE THAT IS CURRENT AT THE CALL (cached per device id), at most 128 (forward
// progress argument in common.h; a CU-masked / partitioned device gets smaller teams, below 8 none).  r06: the cap used to be taken once from
// whichever device was current at the first call -- possibly a sizing call before set_device -- and the word was not portable.
#ifndef hipHostMallocPortable
#define hipHostMallocPortable 0
#endif
struct TeamHost { unsigned* host; unsigned* dev; };
static TeamHost& team_host() {
    static TeamHost t = [] {
        TeamHost h{nullptr, nullptr};
        void* p = nullptr;
        if (hipHostMalloc(&p, 64, hipHostMallocPortable | hipHostMallocMapped | hipHostMallocCoherent) == hipSuccess && p) {
            memset(p, 0, 64);
            void* d = nullptr;
            if (hipHostGetDevicePointer(&d, p, 0) == hipSuccess && d) { h.host = (unsigned*)p; h.dev = (unsigned*)d; }
        }
        (void)hipGetLastError();                               // a process without a device (sizing calls on the build host) is not an error here
        if (!h.host) { static unsigned fallback[16]; h.host = h.dev = fallback; }
        return h;
    }();
    return t;
}
static int team_cap() {
    constexpr int MAXDEV = 64;
    static std::atomic<int> caps[MAXDEV];                      // 0 = not asked yet
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return 128; }
    if (dev < 0 || dev >= MAXDEV) dev = 0;
    int c = caps[dev].load(std::memory_order_relaxed);
    if (c == 0) {
        c = 128;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus > 0) c = cus / 2 < 128 ? cus / 2 : 128;
        else (void)hipGetLastError();
        caps[dev].store(c > 0 ? c : 1, std::memory_order_relaxed);
    }
    return c;
}
// the occupancy the forward-progress argument counts on (>= 2 workgroups per CU), checked once per kernel instantiation against the runtime's own figure
static int team_occupancy_ok(const void* kernel, const char* what) {
    static std::mutex mu; static std::unordered_map<const void*, int> seen;
    std::lock_guard<std::mutex> lk(mu);
    auto it = seen.find(kernel);
    int n = it == seen.end() ? -1 : it->second;
    if (n < 0) {
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, kernel, 256, 0) != hipSuccess) { (void)hipGetLastError(); n = 0; }
        seen[kernel] = n;
    }
    return n >= 2 ? 0 : fail(-1, "%s: the team kernel fits %d workgroup(s) per compute unit, the team exchange needs 2 (segx_tune(3, 1) selects the two-launch form)", what, n);
}
static inline void team_fill(BnTeam& t) {
    const TeamHost& h = team_host();
    t.err = h.dev; t.spin = (unsigned)kget(knobs().team_spin);
    team_next_tag(t.tag_lo, t.tag_hi);
}
// fault injection (knob 13, tests): the grid without its last n workgroups -- the mates of the missing members time out
static inline dim3 team_grid(int64_t wgs) { const int drop = kget(knobs().team_drop); return dim3((unsigned)(wgs > drop ? wgs - drop : 1)); }
template <int KP, bool POOL, int ACT, int RESID>
__global__ __launch_bounds__(256) SEGX_MIN_WAVES_PER_SIMD(KP <= 8 ? 4 : KP <= 16 ? 3 : 2) void bn_act_fwd_team_kernel(BnFwdArgs g, BnTeam t) {
    __shared__ float red[4];
    const int TS = t.B * t.cpp, tl = threadIdx.x;
    const int c = blockIdx.x / TS, r = blockIdx.x - c * TS, b = r / t.cpp, ch = r - b * t.cpp;
    const int C = g.C, S4 = (int)(g.S >> 2), j0 = ch * 256 * KP, act = g.act;
    const int64_t S = g.S, plane = ((int64_t)b * C + c) * S;
    // byte offset of float4 k of this lane inside the plane (clamped: loads past the plane re-read its last float4); recomputed where used -- KP registers of
    // offsets would be a quarter of the 32-float4 form's budget
    auto offk = [&](int k) { const int j = j0 + tl + 256 * k; return 16u * (unsigned)(j < S4 ? j : S4 - 1); };
    f32x4 v[KP];
    const ws_gptr xb = ws_uniform_base(g.X + plane);
#pragma unroll
    for (int k = 0; k < KP; ++k) v[k] = ws_load<f32x4>(xb, offk(k));
    __builtin_amdgcn_sched_barrier(0);
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < KP; ++k) {
        if (!(j0 + tl + 256 * k < S4)) v[k] = f32x4{0.f, 0.f, 0.f, 0.f};
        s += (v[k].x + v[k].y) + (v[k].z + v[k].w);
    }
    const int left4 = S4 - j0;
    const float n = 4.0f * (float)(left4 < 256 * KP ? left4 : 256 * KP);            // floats of this chunk
    const float m = block_sum<4>(s, red) / n;
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < KP; ++k) {
        const float d0 = v[k].x - m, d1 = v[k].y - m, d2 = v[k].z - m, d3 = v[k].w - m;
        q += (j0 + tl + 256 * k < S4) ? (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3) : 0.f;
    }
    q = block_sum<4>(q, red);
    const TeamBufs tb{t.slots + (int64_t)c * TS * TEAM_SLOT, t.mbox + (int64_t)c * TS * TEAM_MBOX, t.tag_lo, t.tag_hi, t.err, t.spin};
    if (tl == 0) { float* pp = tb.slots + r * TEAM_SLOT; team_store(pp, n); team_store(pp + 1, m); team_store(pp + 2, q); }
    float res[3];
    team_exchange(tb, r, TS, res, [](const float* slots, int members, float (&o)[3]) { const BnPart f = bn_fold_team(slots, members); o[0] = f.n; o[1] = f.mean; o[2] = f.m2; });
    BnPart st; st.n = res[0]; st.mean = res[1]; st.m2 = res[2];
    // (a timed-out exchange hands NaN to its members: the variance must carry it, fmaxf would turn it into 0)
    const float mean = st.mean, var = st.n != st.n ? st.n : st.n > 0.f ? fmaxf(st.m2 / st.n, 0.f) : 0.f;
    if (r == 0 && tl == 0) {
        g.mean[c] = mean; g.var[c] = var;
        if (g.run_mean && st.n == st.n) {                         // a timed-out exchange (n = NaN) poisons this step's output, NOT the running statistics a later checkpoint would keep
            g.run_mean[c] = (1.0f - g.momentum) * g.run_mean[c] + g.momentum * mean;
            g.run_var[c] = (1.0f - g.momentum) * g.run_var[c] + g.momentum * (st.m2 / fmaxf(st.n - 1.0f, 1.0f));
        }
    }
    const float sc = rsqrtf(var + g.eps) * g.w[c], sh = g.b[c] - mean * sc;
    const bool resid = RESID < 0 ? g.resid != nullptr : RESID != 0;
    const float dcs = drop_connect_scale(g.dc_p, g.seed, g.offset, g.rbase, b);
    const ws_gptr_w yb = ws_uniform_base_w(g.Y + (int64_t)b * g.y_bs + (int64_t)c * S);
    const ws_gptr rb = ws_uniform_base((resid ? g.resid : g.X) + plane);
    float acc = 0.f;
    constexpr int G = 4;                                         // the skip connection's float4 are requested G at a time
    // the store phase recomputes its offsets from an opaque copy of the lane index: shared with the load phase's (common sub-expressions), KP offset
    // registers stay alive across the exchange next to the KP float4 of state -- the 32-float4 forms then spilled 52..164 bytes per lane (r04_w trace)
    int tl2 = tl;
    SEGX_PIN(tl2);
    auto offs = [&](int k) { const int j = j0 + tl2 + 256 * k; return 16u * (unsigned)(j < S4 ? j : S4 - 1); };
#pragma unroll
    for (int k0 = 0; k0 < KP; k0 += G) {
        f32x4 rv[G];
        if (resid) {
#pragma unroll
            for (int e = 0; e < G; ++e) rv[e] = ws_load<f32x4>(rb, offs(k0 + e));
        }
#pragma unroll
        for (int e = 0; e < G; ++e) {
            const int k = k0 + e;
            f32x4 o;
            o.x = act_fwd_c<ACT>(v[k].x * sc + sh, act); o.y = act_fwd_c<ACT>(v[k].y * sc + sh, act);
            o.z = act_fwd_c<ACT>(v[k].z * sc + sh, act); o.w = act_fwd_c<ACT>(v[k].w * sc + sh, act);
            if (resid) { o.x = o.x * dcs + rv[e].x; o.y = o.y * dcs + rv[e].y; o.z = o.z * dcs + rv[e].z; o.w = o.w * dcs + rv[e].w; }
            if (j0 + tl2 + 256 * k < S4) {
                ws_store<f32x4>(yb, offs(k), o);
                if (POOL) acc += (o.x + o.y) + (o.z + o.w);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    if (POOL) {
        acc = block_sum<4>(acc, red);
        if (tl == 0) g.psum[((int64_t)b * C + c) * t.cpp + ch] = acc;
    }
}
template <int KP, int ACT>
__global__ __launch_bounds__(256) SEGX_MIN_WAVES_PER_SIMD(KP <= 4 ? 4 : KP <= 8 ? 3 : 2) void bn_act_bwd_team_kernel(BnBwdArgs g, BnTeam t) {
    __shared__ float red[4];
    const int TS = t.B * t.cpp, tl = threadIdx.x;
    const int c = blockIdx.x / TS, r = blockIdx.x - c * TS, b = r / t.cpp, ch = r - b * t.cpp;
    const int C = g.C, S4 = (int)(g.S >> 2), j0 = ch * 256 * KP, act = g.act;
    const int64_t S = g.S;
    unsigned off[KP];
#pragma unroll
    for (int k = 0; k < KP; ++k) { const int j = j0 + tl + 256 * k; off[k] = 16u * (unsigned)(j < S4 ? j : S4 - 1); }
    f32x4 h[KP], d[KP];
    const ws_gptr xb = ws_uniform_base(g.X + ((int64_t)b * C + c) * S), gb = ws_uniform_base(g.dY + (int64_t)b * g.dy_bs + (int64_t)c * S);
#pragma unroll
    for (int k = 0; k < KP; ++k) { h[k] = ws_load<f32x4>(xb, off[k]); d[k] = ws_load<f32x4>(gb, off[k]); }
    __builtin_amdgcn_sched_barrier(0);
    const float rstd = rsqrtf(g.var[c] + g.eps), m = g.mean[c], wc = g.w[c], bc_ = g.b[c];
    const int pl = b * C + c;
    const float gt = (g.gate ? g.gate[pl] : 1.0f) * drop_connect_scale(g.dc_p, g.seed, g.offset, g.rbase, b), dp = g.dpool ? g.dpool[pl] * g.inv_S : 0.f;
    float a = 0.f, q = 0.f;
#pragma unroll
    for (int k = 0; k < KP; ++k) {
        const bool ok = j0 + tl + 256 * k < S4;
        const f32x4 xv = h[k], gv = d[k];
        f32x4 hh, dd;
        hh.x = (xv.x - m) * rstd; hh.y = (xv.y - m) * rstd; hh.z = (xv.z - m) * rstd; hh.w = (xv.w - m) * rstd;
        dd.x = (gv.x * gt + dp) * act_grad_c<ACT>(hh.x * wc + bc_, act); dd.y = (gv.y * gt + dp) * act_grad_c<ACT>(hh.y * wc + bc_, act);
        dd.z = (gv.z * gt + dp) * act_grad_c<ACT>(hh.z * wc + bc_, act); dd.w = (gv.w * gt + dp) * act_grad_c<ACT>(hh.w * wc + bc_, act);
        if (!ok) { dd = f32x4{0.f, 0.f, 0.f, 0.f}; hh = dd; }
        h[k] = hh; d[k] = dd;
        a += (dd.x + dd.y) + (dd.z + dd.w); q += (dd.x * hh.x + dd.y * hh.y) + (dd.z * hh.z + dd.w * hh.w);
    }
    a = block_sum<4>(a, red); q = block_sum<4>(q, red);
    const TeamBufs tb{t.slots + (int64_t)c * TS * TEAM_SLOT, t.mbox + (int64_t)c * TS * TEAM_MBOX, t.tag_lo, t.tag_hi, t.err, t.spin};
    if (tl == 0) { float* pp = tb.slots + r * TEAM_SLOT; team_store(pp, a); team_store(pp + 1, q); }
    float res[3];
    // the team's sums: lane l takes members l, l + 64 (team <= 128) in that order, then a symmetric butterfly -- the same bits in every lane and every run
    team_exchange(tb, r, TS, res, [](const float* parts, int members, float (&o)[3]) {
        const int lane = threadIdx.x & 63;
        const bool h0 = lane < members, h1 = lane + 64 < members;
        const float a0 = team_load(parts + TEAM_SLOT * (h0 ? lane : 0)), q0 = team_load(parts + TEAM_SLOT * (h0 ? lane : 0) + 1);
        const float a1 = team_load(parts + TEAM_SLOT * (h1 ? lane + 64 : 0)), q1 = team_load(parts + TEAM_SLOT * (h1 ? lane + 64 : 0) + 1);
        float A = (h0 ? a0 : 0.f) + (h1 ? a1 : 0.f), Q = (h0 ? q0 : 0.f) + (h1 ? q1 : 0.f);
#pragma unroll
        for (int o2 = 1; o2 < 64; o2 <<= 1) { A += __shfl_xor(A, o2); Q += __shfl_xor(Q, o2); }
        o[0] = A; o[1] = Q; o[2] = 0.f;
    });
    const float A = res[0], Q = res[1];
    if (r == 0 && tl == 0) { g.db[c] = A; g.dw[c] = Q; }
    const float inv_n = 1.0f / ((float)t.B * (float)S), k1 = A * inv_n, k2 = Q * inv_n, sc = wc * rstd;
    const ws_gptr_w ob = ws_uniform_base_w(g.dX + ((int64_t)b * C + c) * S);
#pragma unroll
    for (int k = 0; k < KP; ++k) {
        if (j0 + tl + 256 * k < S4) {
            f32x4 o;
            o.x = sc * (d[k].x - k1 - h[k].x * k2); o.y = sc * (d[k].y - k1 - h[k].y * k2);
            o.z = sc * (d[k].z - k1 - h[k].z * k2); o.w = sc * (d[k].w - k1 - h[k].w * k2);
            ws_store<f32x4>(ob, off[k], o);
        }
    }
}
// which channel-resident form serves (B planes of S floats) -- 0 none, else TEAM * 16 + KP packed: registers per lane <= cap float4 (forward: x; backward: xhat + du)
static inline int bn_res_form(int B, int64_t S, bool backward) {
    if ((S & 3) || B < 1 || B > 8 || S > 8192) return 0;
    const int S4 = (int)(S >> 2);
    const int kw = (S4 + 63) / 64, kb = (S4 + 255) / 256;
    const int capw = backward ? 1 : 2, capb = backward ? 2 : 4;          // per-plane float4 per lane; x BMAX = 8 planes (6 for the largest backward form)
    if (kw <= capw) return 64 * 16 + (kw <= 1 ? 1 : 2);
    if (kb <= capb) return 256 * 16 + (kb <= 1 ? 1 : kb <= 2 ? 2 : 4);
    return 0;
}
// backward of planes of 2049..4096 floats at batch <= 6 (672 / 960 channels of 64 x 64 at cfg2): a team of one workgroup per plane.  (r04 also had a
// six-plane resident form here -- 192 registers of state, 180..312 bytes spilled per lane, slower than the team even so: r04_s 60.8 -> 49.9 us -- removed.)
static inline bool bn_bwd_plane_team(int B, int64_t S) { return (S & 3) == 0 && S > 2048 && S <= 4096 && B >= 1 && B <= 6; }

// float4 per lane of a team workgroup (chunk = 1024 x KP floats of ONE plane), 0 = the team form does not serve (B, S).  Backward: 16 (x and dy: 128
// registers of state), teams up to 128 workgroups.  Forward: 16 while the team stays within 32 workgroups, else 32 (teams up to 64): a forward team
// waits longer than the second read costs once it gets large (r04_l: 144 channels of 512 x 512 at batch 6 with 16 float4 per lane, team 96: 484 us
// either way; backward 521 against 779 us).
static inline int bn_team_chunks(int64_t S, int kp) { return (int)((S / 4 + 256 * kp - 1) / (256 * kp)); }
static inline int bn_team_kp_small(int64_t S) { return S <= 4096 ? 4 : S <= 8192 ? 8 : 16; }     // one chunk per plane: the smallest form that holds it
static inline int bn_team_kp(int B, int64_t S, bool backward) {
    if ((S & 3) || B < 1 || S < 16384) return 0;
    const int cap = team_cap();
    if (backward) return (int64_t)bn_team_chunks(S, 16) * B <= cap ? 16 : 0;
    if ((int64_t)bn_team_chunks(S, 16) * B <= (cap < 32 ? cap : 32)) return 16;
    return (int64_t)bn_team_chunks(S, 32) * B <= (cap < 64 ? cap : 64) ? 32 : 0;
}
// policy (knob 3): which form serves training BatchNorm on (B, S) when the library computes the statistics itself: 2 resident, 1 team, 0 two launches;
// *kp = float4 per lane of the team form
static inline int bn_auto_form(int B, int64_t S, bool backward, int* kp = nullptr) {
    const int path = kget(knobs().bn_path);
    int k = 16;
    int form = 0;
    if (path == 2 && (S & 3) == 0 && B <= 128 && S >= 4 && (int64_t)bn_team_chunks(S, 16) * B <= team_cap()) { form = 1; k = bn_team_kp_small(S); }      // tests: the team form on small planes too
    else if (bn_res_form(B, S, backward)) form = 2;
    else if (backward && path != 1 && bn_bwd_plane_team(B, S) && B <= team_cap()) { form = 1; k = bn_team_kp_small(S); }
    else if (path != 1 && (k = bn_team_kp(B, S, backward)) != 0) form = 1;
    if (kp) *kp = form == 1 ? k : 0;
    return form;
}
template <int KP>
static int bn_team_launch_fwd(const BnFwdArgs& g, const BnTeam& t, dim3 grid, hipStream_t stream, bool pool) {
    const bool resid = g.resid != nullptr;
#define SEGX_BN_TF(P, A, R) do { if (int rc = team_occupancy_ok((const void*)bn_act_fwd_team_kernel<KP, P, A, R>, "segx_bn_act_fwd2/team")) return rc; \
                                 hipLaunchKernelGGL((bn_act_fwd_team_kernel<KP, P, A, R>), grid, dim3(256), 0, stream, g, t); } while (0)
    if (pool) { if (!resid && g.act == ACT_SWISH) SEGX_BN_TF(true, ACT_SWISH, 0); else SEGX_BN_TF(true, -1, -1); }
    else if (!resid && g.act == ACT_SWISH) SEGX_BN_TF(false, ACT_SWISH, 0);
    else if (g.act == ACT_NONE) { if (resid) SEGX_BN_TF(false, ACT_NONE, 1); else SEGX_BN_TF(false, ACT_NONE, 0); }
    else if (!resid && g.act == ACT_RELU) SEGX_BN_TF(false, ACT_RELU, 0);
    else SEGX_BN_TF(false, -1, -1);
#undef SEGX_BN_TF
    return check_launch("segx_bn_act_fwd2/team");
}
template <int KP>
static int bn_team_launch_bwd(const BnBwdArgs& g, const BnTeam& t, dim3 grid, hipStream_t stream) {
#define SEGX_BN_TB(A) do { if (int rc = team_occupancy_ok((const void*)bn_act_bwd_team_kernel<KP, A>, "segx_bn_act_bwd2/team")) return rc; \
                           hipLaunchKernelGGL((bn_act_bwd_team_kernel<KP, A>), grid, dim3(256), 0, stream, g, t); } while (0)
    if (g.act == ACT_SWISH) SEGX_BN_TB(ACT_SWISH); else if (g.act == ACT_NONE) SEGX_BN_TB(ACT_NONE); else if (g.act == ACT_RELU) SEGX_BN_TB(ACT_RELU); else SEGX_BN_TB(-1);
#undef SEGX_BN_TB
    return check_launch("segx_bn_act_bwd2/team");
}
// The resident kernels exist per (activation, pooling, skip) combination the backbones use -- swish (+ pooling), none (+ skip), relu -- and once with
// the run-time values for everything else.
template <int T, int K, int BM>
static void bn_res_launch_fwd(const BnFwdArgs& g, int B, dim3 grid, hipStream_t stream, bool pool) {
    const bool resid = g.resid != nullptr;
#define SEGX_BN_F(P, A, R) hipLaunchKernelGGL((bn_act_fwd_res_kernel<T, K, BM, P, A, R>), grid, dim3(256), 0, stream, g, B)
    if (pool) { if (!resid && g.act == ACT_SWISH) SEGX_BN_F(true, ACT_SWISH, 0); else SEGX_BN_F(true, -1, -1); }
    else if (!resid && g.act == ACT_SWISH) SEGX_BN_F(false, ACT_SWISH, 0);
    else if (g.act == ACT_NONE) { if (resid) SEGX_BN_F(false, ACT_NONE, 1); else SEGX_BN_F(false, ACT_NONE, 0); }
    else if (!resid && g.act == ACT_RELU) SEGX_BN_F(false, ACT_RELU, 0);
    else SEGX_BN_F(false, -1, -1);
#undef SEGX_BN_F
}
template <int T, int K, int BM>
static void bn_res_launch_bwd(const BnBwdArgs& g, int B, dim3 grid, hipStream_t stream) {
#define SEGX_BN_B(A) hipLaunchKernelGGL((bn_act_bwd_res_kernel<T, K, BM, A>), grid, dim3(256), 0, stream, g, B)
    if (g.act == ACT_SWISH) SEGX_BN_B(ACT_SWISH); else if (g.act == ACT_NONE) SEGX_BN_B(ACT_NONE); else if (g.act == ACT_RELU) SEGX_BN_B(ACT_RELU); else SEGX_BN_B(-1);
#undef SEGX_BN_B
}

// =================================================================================================
// Depthwise convolution (efficientnet/model.py:100, groups == channels), static TF-'same' padding (N6):
//   y[b,c,oy,ox] = sum_{ky,kx} w[c,ky,kx] * x[b,c,oy*S+ky-pt, ox*S+kx-pl]
// HBM-bound (one read of x, one write of y).  A thread owns ONE output column and DW_TY consecutive output rows and
// slides down the input rows keeping nothing but the K values of the current row: a row of the plane is read as whole
// 128-byte lines by neighbouring lanes (the x-halo comes out of L1), the y-halo costs (TY+K-1)/TY.  The earlier
// 16x16-tile version fetched 3.6x the plane (r01-e PMC): 64-byte tile rows used half of every line they touched and
// the neighbouring tile sat on another XCD's L2.
//   workgroup = (256 >> txw_log2) row groups x (1 << txw_log2) columns;  FLIP = correlate with the 180-degree rotated
//   filter (the stride-1 data gradient is exactly that, with pads K-1-pt / K-1-pl).
// =================================================================================================
constexpr int DW_TY = 8;
struct DwTile { int ox, oy0; bool live; };
__device__ __forceinline__ DwTile dw_tile(int tile, int tiles_x, int txw_log2, int OH, int OW) {
    const int txw = 1 << txw_log2, rg = threadIdx.x >> txw_log2, nrg = 256 >> txw_log2;
    const int tyi = tile / tiles_x, txi = tile - tyi * tiles_x;
    DwTile t;
    t.ox = txi * txw + (threadIdx.x & (txw - 1));
    t.oy0 = (tyi * nrg + rg) * DW_TY;
    t.live = t.ox < OW && t.oy0 < OH;
    return t;
}
template <int K, int ST, bool FLIP>
__global__ __launch_bounds__(256) void dwconv_rows_kernel(const float* __restrict__ X, const float* __restrict__ Wt, float* __restrict__ Y,
                                                          int C, int H, int W, int OH, int OW, int pt, int pl, int tiles_x, int txw_log2) {
    const int bc = blockIdx.y, c = bc % C;
    const DwTile t = dw_tile(blockIdx.x, tiles_x, txw_log2, OH, OW);
    if (!t.live) return;
    const float* x = X + (int64_t)bc * H * W;
    float w[K * K];
#pragma unroll
    for (int i = 0; i < K * K; ++i) w[i] = Wt[(int64_t)c * K * K + (FLIP ? K * K - 1 - i : i)];
    float acc[DW_TY];
#pragma unroll
    for (int i = 0; i < DW_TY; ++i) acc[i] = 0.f;
    const int ix0 = t.ox * ST - pl, iy0 = t.oy0 * ST - pt;
    constexpr int NR = (DW_TY - 1) * ST + K;            // input rows under DW_TY output rows
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        const int iy = iy0 + r;
        const bool rowok = iy >= 0 && iy < H;
        const float* xr = x + (int64_t)min(max(iy, 0), H - 1) * W;
        float v[K];
#pragma unroll
        for (int kx = 0; kx < K; ++kx) {
            const int ix = ix0 + kx;
            const float q = xr[min(max(ix, 0), W - 1)];
            v[kx] = (rowok && ix >= 0 && ix < W) ? q : 0.f;
        }
#pragma unroll
        for (int ky = 0; ky < K; ++ky) {
            if ((r - ky) < 0 || (r - ky) % ST != 0 || (r - ky) / ST >= DW_TY) continue;      // compile-time
#pragma unroll
            for (int kx = 0; kx < K; ++kx) acc[(r - ky) / ST] += w[ky * K + kx] * v[kx];
            SEGX_PIN(acc[(r - ky) / ST]);             // keep the FMAs next to their loads (see dwconv_rows4_kernel)
        }
    }
    float* y = Y + (int64_t)bc * OH * OW + t.ox;
#pragma unroll
    for (int i = 0; i < DW_TY; ++i) if (t.oy0 + i < OH) y[(int64_t)(t.oy0 + i) * OW] = acc[i];
}
// ---- float4 variant (W % 4 == 0, OW % 4 == 0, left pad PL known at compile time): a thread owns FOUR adjacent output columns.
// Every input row is read as NV aligned float4 per thread (the one under the outputs plus its neighbours, which are L1
// hits), so a wave issues 1 KB loads/stores like a streaming kernel.  Row groups are flattened over (plane, tile) so small
// planes (32 x 32) still fill the workgroup.
constexpr int DW4_TY = 8, DW4_AHEAD = 2;
template <int K, int ST> struct Dw4 { static constexpr int NV = ST == 1 ? 3 : 4; };   // aligned float4 loaded per input row
template <int NV>
__device__ __forceinline__ void dw4_load_row(float (&e)[4 * NV], const float* __restrict__ xr, const int (&xo)[NV], const bool (&okv)[NV], bool rowok) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const float4 q = *reinterpret_cast<const float4*>(xr + xo[i]);
        const bool ok = rowok && okv[i];
        e[4 * i] = ok ? q.x : 0.f; e[4 * i + 1] = ok ? q.y : 0.f; e[4 * i + 2] = ok ? q.z : 0.f; e[4 * i + 3] = ok ? q.w : 0.f;
    }
}
template <int K, int ST, int PL, bool FLIP>
__global__ __launch_bounds__(256) SEGX_MIN_WAVES_PER_SIMD(4) void dwconv_rows4_kernel(const float* __restrict__ X, const float* __restrict__ Wt, float* __restrict__ Y,
                                                           int C, int H, int W, int OH, int OW, int pt, int tiles_x, int tiles_y, int tpr_log2,
                                                           int64_t ngroups) {
    constexpr int NV = Dw4<K, ST>::NV;
    static_assert(PL <= 4 && 4 + 3 * ST + K - 1 - PL < 4 * NV, "window does not fit the loaded float4s");
    const int64_t gid = ((int64_t)xcd_block(blockIdx.x, gridDim.x) * 256 + threadIdx.x) >> tpr_log2;      // XCD-contiguous runs of workgroups (common.h): halo rows shared with the next row group stay in ONE XCD's L2
    if (gid >= ngroups) return;
    const int gpp = tiles_x * tiles_y;
    const int64_t plane = gid / gpp;
    const int rem = (int)(gid - plane * gpp), tyi = rem / tiles_x, txi = rem - tyi * tiles_x;
    const int ox = ((txi << tpr_log2) + (threadIdx.x & ((1 << tpr_log2) - 1))) * 4, oy0 = tyi * DW4_TY;
    if (ox >= OW) return;
    const int c = (int)(plane % C);
    const float* x = X + plane * H * W;
    float w[K * K];
#pragma unroll
    for (int i = 0; i < K * K; ++i) w[i] = Wt[(int64_t)c * K * K + (FLIP ? K * K - 1 - i : i)];
    float acc[DW4_TY][4];
#pragma unroll
    for (int i = 0; i < DW4_TY; ++i) { acc[i][0] = 0.f; acc[i][1] = 0.f; acc[i][2] = 0.f; acc[i][3] = 0.f; }
    int xo[NV]; bool okv[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) { const int col = ox * ST - 4 + 4 * i; okv[i] = col >= 0 && col < W; xo[i] = min(max(col, 0), W - 4); }
    const int iy0 = oy0 * ST - pt;
    constexpr int NR = (DW4_TY - 1) * ST + K;
    // rows are loaded DW4_AHEAD ahead of their use into a statically rotated ring; the scheduling barriers keep the compiler
    // from hoisting every load of the unrolled loop to the top (206-256 VGPRs, one wave per SIMD, when left alone)
    float e[DW4_AHEAD + 1][4 * NV];
#pragma unroll
    for (int r = 0; r < DW4_AHEAD && r < NR; ++r)
        dw4_load_row<NV>(e[r], x + (int64_t)min(max(iy0 + r, 0), H - 1) * W, xo, okv, iy0 + r >= 0 && iy0 + r < H);
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        if (r + DW4_AHEAD < NR) {
            const int iy = iy0 + r + DW4_AHEAD;
            dw4_load_row<NV>(e[(r + DW4_AHEAD) % (DW4_AHEAD + 1)], x + (int64_t)min(max(iy, 0), H - 1) * W, xo, okv, iy >= 0 && iy < H);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ky = 0; ky < K; ++ky) {
            if ((r - ky) < 0 || (r - ky) % ST != 0 || (r - ky) / ST >= DW4_TY) continue;     // compile-time
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int kx = 0; kx < K; ++kx)
                    acc[(r - ky) / ST][j] += w[ky * K + kx] * e[r % (DW4_AHEAD + 1)][4 + j * ST + kx - PL];
#pragma unroll
            for (int j = 0; j < 4; ++j) SEGX_PIN(acc[(r - ky) / ST][j]);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    float* y = Y + plane * OH * OW + ox;
#pragma unroll
    for (int i = 0; i < DW4_TY; ++i)
        if (oy0 + i < OH) *reinterpret_cast<float4*>(y + (int64_t)(oy0 + i) * OW) = make_float4(acc[i][0], acc[i][1], acc[i][2], acc[i][3]);
}
// weight gradient, float4 variant: ONE WAVE per (plane, strip); it walks every strips-th tile of the plane ((64 >> tpr_log2)
// row groups x 4 rows x (4 << tpr_log2) columns), then reduces its K*K sums with wave shuffles -- no LDS, no barrier.
constexpr int DWG_TY = 4;
// one wave's walk over every strips-th tile of a plane: acc[ky][kx] += sum over the wave's outputs of dy * x(window)
template <int K, int ST, int PL>
__device__ __forceinline__ void dw4_wgrad_walk(float (&acc)[K * K], const float* __restrict__ x, const float* __restrict__ g, int H, int W, int OH, int OW, int pt,
                                               int tiles_x, int ntiles, int tpr_log2, int strip, int strips, int lane) {
    constexpr int NV = Dw4<K, ST>::NV;
    static_assert(PL <= 4 && 4 + 3 * ST + K - 1 - PL < 4 * NV, "window does not fit the loaded float4s");
    const int lr = lane & ((1 << tpr_log2) - 1), rg = lane >> tpr_log2, nrg = 64 >> tpr_log2;
    for (int tile = strip; tile < ntiles; tile += strips) {
        const int tyi = tile / tiles_x, txi = tile - tyi * tiles_x;
        const int ox = ((txi << tpr_log2) + lr) * 4, oy0 = (tyi * nrg + rg) * DWG_TY;
        if (ox >= OW || oy0 >= OH) continue;
        float gv[DWG_TY][4];
#pragma unroll
        for (int i = 0; i < DWG_TY; ++i) {
            const float4 q = *reinterpret_cast<const float4*>(g + (int64_t)min(oy0 + i, OH - 1) * OW + ox);
            const bool ok = oy0 + i < OH;
            gv[i][0] = ok ? q.x : 0.f; gv[i][1] = ok ? q.y : 0.f; gv[i][2] = ok ? q.z : 0.f; gv[i][3] = ok ? q.w : 0.f;
        }
        int xo[NV]; bool okv[NV];
#pragma unroll
        for (int i = 0; i < NV; ++i) { const int col = ox * ST - 4 + 4 * i; okv[i] = col >= 0 && col < W; xo[i] = min(max(col, 0), W - 4); }
        const int iy0 = oy0 * ST - pt;
        constexpr int NR = (DWG_TY - 1) * ST + K;
        float e[DW4_AHEAD + 1][4 * NV];
#pragma unroll
        for (int r = 0; r < DW4_AHEAD && r < NR; ++r)
            dw4_load_row<NV>(e[r], x + (int64_t)min(max(iy0 + r, 0), H - 1) * W, xo, okv, iy0 + r >= 0 && iy0 + r < H);
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            if (r + DW4_AHEAD < NR) {
                const int iy = iy0 + r + DW4_AHEAD;
                dw4_load_row<NV>(e[(r + DW4_AHEAD) % (DW4_AHEAD + 1)], x + (int64_t)min(max(iy, 0), H - 1) * W, xo, okv, iy >= 0 && iy < H);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ky = 0; ky < K; ++ky) {
                if ((r - ky) < 0 || (r - ky) % ST != 0 || (r - ky) / ST >= DWG_TY) continue;  // compile-time
#pragma unroll
                for (int kx = 0; kx < K; ++kx)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[ky * K + kx] += gv[(r - ky) / ST][j] * e[r % (DW4_AHEAD + 1)][4 + j * ST + kx - PL];
#pragma unroll
                for (int kx = 0; kx < K; ++kx) SEGX_PIN(acc[ky * K + kx]);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}
template <int K, int ST, int PL>
__global__ __launch_bounds__(256) SEGX_MIN_WAVES_PER_SIMD(4) void dwconv_wgrad4_kernel(const float* __restrict__ dY, const float* __restrict__ X, float* __restrict__ part,
                                                            int C, int H, int W, int OH, int OW, int pt, int tiles_x, int ntiles, int tpr_log2,
                                                            int strips, int64_t nwork) {
    const int64_t wid = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 6;
    if (wid >= nwork) return;                                   // whole waves leave together
    const int64_t plane = wid / strips;
    const int strip = (int)(wid - plane * strips), b = (int)(plane / C), c = (int)(plane - (int64_t)b * C);
    const int lane = threadIdx.x & 63;
    float acc[K * K];
#pragma unroll
    for (int i = 0; i < K * K; ++i) acc[i] = 0.f;
    dw4_wgrad_walk<K, ST, PL>(acc, X + plane * H * W, dY + plane * OH * OW, H, W, OH, OW, pt, tiles_x, ntiles, tpr_log2, strip, strips, lane);
    float* o = part + (((int64_t)b * strips + strip) * C + c) * (K * K);
#pragma unroll
    for (int i = 0; i < K * K; ++i) {
        const float sw = wave_sum(acc[i]);
        if (lane == 0) o[i] = sw;
    }
}
// r04: planes that one wave walks alone (strips == 1: <= 8192 outputs, 22 of EfficientNet-B4's 32 depthwise layers at 512 x 512) and B <= 8: ONE workgroup of
// eight waves per CHANNEL, wave b on sample b; the B partial filters are added in sample order through LDS and the workgroup writes dw[c] itself --
// no [B][C][K*K] partial tensor, no column-sum launches behind it (two per layer).
template <int K, int ST, int PL>
__global__ __launch_bounds__(512) void dwconv_wgrad4c_kernel(const float* __restrict__ dY, const float* __restrict__ X, float* __restrict__ dW,
                                                             int B, int C, int H, int W, int OH, int OW, int pt, int tiles_x, int ntiles, int tpr_log2) {
    __shared__ float red[8][K * K];
    const int c = blockIdx.x, b = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float acc[K * K];
#pragma unroll
    for (int i = 0; i < K * K; ++i) acc[i] = 0.f;
    if (b < B) {
        const int64_t plane = (int64_t)b * C + c;
        dw4_wgrad_walk<K, ST, PL>(acc, X + plane * H * W, dY + plane * OH * OW, H, W, OH, OW, pt, tiles_x, ntiles, tpr_log2, 0, 1, lane);
    }
#pragma unroll
    for (int i = 0; i < K * K; ++i) {
        const float sw = wave_sum(acc[i]);
        if (lane == 0) red[b][i] = sw;
    }
    __syncthreads();
    if (threadIdx.x < K * K) {
        float t = 0.f;
        for (int q = 0; q < B; ++q) t += red[q][threadIdx.x];
        dW[(int64_t)c * (K * K) + threadIdx.x] = t;
    }
}

// ---- r06: stride-1 'same' depthwise backward, DATA AND WEIGHT gradient from one pass over dy ------------------------------------------------------
//   dx[iy][ix]  = sum_{ky,kx} w[ky][kx]        dy[iy - ky + p][ix - kx + p]
//   dw[ky][kx]  = sum_{iy,ix} x[iy][ix]        dy[iy - ky + p][ix - kx + p]          (p = (K - 1) / 2: both walk the SAME dy window of an input position)
// The data-gradient kernel above (dwconv_rows4_kernel<K, 1, P, true>) already holds, per thread, the K + 7 dy rows x 12 columns under its 8 x 4 dx positions; the weight
// gradient was a kernel of its own that read dy AGAIN (plus x) -- one full pass over the expanded tensor per MBConv block more than needed, and the one-wave-per-strip
// walk of dwconv_wgrad4_kernel ran at 3.4 TB/s.  Here the thread also loads x at its 8 x 4 positions (a ring of K rows) and adds x * dy into K * K sums next to the
// w * dy of the data gradient; a wave reduces its sums once (every lane of a wave lies in ONE plane: the launcher checks it) and lane 0 writes them as one row of the
// partial tensor [B * strips][C][K * K] that segx_colsum adds in a fixed order, as for the separate kernel.  No early exits: lanes beyond the plane keep all-zero windows.
template <int K>
__global__ __launch_bounds__(256) void dwconv_bwd_s1_fused_kernel(const float* __restrict__ dY, const float* __restrict__ X, const float* __restrict__ Wt,
                                                                  float* __restrict__ dX, float* __restrict__ part, int C, int H, int W, int tiles_x, int tiles_y,
                                                                  int tpr_log2, int64_t ngroups, int strips) {
    constexpr int P = (K - 1) / 2, NV = 3, NR = DW4_TY - 1 + K;
    static_assert(P <= 4 && 4 + 3 + K - 1 - P < 4 * NV, "window does not fit the loaded float4s");
    const int64_t gid0 = ((int64_t)xcd_block(blockIdx.x, gridDim.x) * 256 + threadIdx.x) >> tpr_log2;
    const bool glive = gid0 < ngroups;
    const int64_t gid = glive ? gid0 : ngroups - 1;
    const int gpp = tiles_x * tiles_y;
    const int64_t plane = gid / gpp;
    const int rem = (int)(gid - plane * gpp), tyi = rem / tiles_x, txi = rem - tyi * tiles_x;
    const int ox0 = ((txi << tpr_log2) + (threadIdx.x & ((1 << tpr_log2) - 1))) * 4, oy0 = tyi * DW4_TY;
    const bool live = glive && ox0 < W;
    const int ox = live ? ox0 : 0;
    const int c = (int)(plane % C);
    const float* g = dY + plane * H * W;
    const float* x = X + plane * H * W;
    float w[K * K], gw[K * K];
#pragma unroll
    for (int i = 0; i < K * K; ++i) { w[i] = Wt[(int64_t)c * K * K + (K * K - 1 - i)]; gw[i] = 0.f; }
    float acc[DW4_TY][4];
#pragma unroll
    for (int i = 0; i < DW4_TY; ++i) { acc[i][0] = 0.f; acc[i][1] = 0.f; acc[i][2] = 0.f; acc[i][3] = 0.f; }
    int xo[NV]; bool okv[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) { const int col = ox - 4 + 4 * i; okv[i] = live && col >= 0 && col < W; xo[i] = min(max(col, 0), W - 4); }
    const int iy0 = oy0 - (K - 1 - P);                       // first dy row under dx row oy0 (the rotated filter's top pad)
    float e[DW4_AHEAD + 1][4 * NV];
    float xr[K][4];                                          // x rows oy0 + i, ring over i % K (row i is used while r = i .. i + K - 1)
#pragma unroll
    for (int r = 0; r < DW4_AHEAD && r < NR; ++r)
        dw4_load_row<NV>(e[r], g + (int64_t)min(max(iy0 + r, 0), H - 1) * W, xo, okv, iy0 + r >= 0 && iy0 + r < H);
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        if (r + DW4_AHEAD < NR) {
            const int iy = iy0 + r + DW4_AHEAD;
            dw4_load_row<NV>(e[(r + DW4_AHEAD) % (DW4_AHEAD + 1)], g + (int64_t)min(max(iy, 0), H - 1) * W, xo, okv, iy >= 0 && iy < H);
        }
        if (r < DW4_TY) {                                    // x row of dx row r (first needed now, with ky = 0)
            const float4 q = *reinterpret_cast<const float4*>(x + (int64_t)min(oy0 + r, H - 1) * W + ox);
            const bool ok = live && oy0 + r < H;
            xr[r % K][0] = ok ? q.x : 0.f; xr[r % K][1] = ok ? q.y : 0.f; xr[r % K][2] = ok ? q.z : 0.f; xr[r % K][3] = ok ? q.w : 0.f;
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ky = 0; ky < K; ++ky) {
            if ((r - ky) < 0 || (r - ky) >= DW4_TY) continue;     // compile-time
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int kx = 0; kx < K; ++kx) {
                    const float dv = e[r % (DW4_AHEAD + 1)][4 + j + kx - (K - 1 - P)];
                    acc[r - ky][j] += w[ky * K + kx] * dv;
                    gw[ky * K + kx] += xr[(r - ky) % K][j] * dv;
                }
#pragma unroll
            for (int j = 0; j < 4; ++j) SEGX_PIN(acc[r - ky][j]);
#pragma unroll
            for (int kx = 0; kx < K; ++kx) SEGX_PIN(gw[ky * K + kx]);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    float* d = dX + plane * H * W + ox;
#pragma unroll
    for (int i = 0; i < DW4_TY; ++i)
        if (live && oy0 + i < H) *reinterpret_cast<float4*>(d + (int64_t)(oy0 + i) * W) = make_float4(acc[i][0], acc[i][1], acc[i][2], acc[i][3]);
    // gw[t] holds the sum for the ROTATED tap t (w was read rotated): dw index K K - 1 - t.  One row of `part` per wave.
    const int lane = threadIdx.x & 63, gpw = 64 >> tpr_log2;
    const int strip = rem / gpw, b = (int)(plane / C);
    float* o = part + (((int64_t)b * strips + strip) * C + c) * (K * K);
#pragma unroll
    for (int t = 0; t < K * K; ++t) {
        const float sw = wave_sum(gw[t]);
        if (lane == 0 && glive) o[K * K - 1 - t] = sw;
    }
}

// Stride-2 data gradient, float4 variant (W % 8 == 0, OW % 4 == 0, pads known at compile time):
//   dx[iy,ix] = sum over (ky,kx) with iy+PT-ky and ix+PL-kx even of  w[ky,kx] * dy[(iy+PT-ky)/2, (ix+PL-kx)/2]
// A thread owns 8 adjacent dx columns x 4 dx rows (tile origin multiple of (4, 8), so every parity test is a compile-time
// constant) and walks the <= 4 dy rows underneath, each read as three aligned float4.
constexpr int DWB_TY = 4;
constexpr int dw_floor_half(int v) { return v >= 0 ? v / 2 : -((-v + 1) / 2); }
template <int K, int PL, int PT>
__global__ __launch_bounds__(256) SEGX_MIN_WAVES_PER_SIMD(4) void dwconv_bwd4s2_kernel(const float* __restrict__ dY, const float* __restrict__ Wt,
                                                                                   float* __restrict__ dX, int C, int H, int W, int OH, int OW,
                                                                                   int tiles_x, int tiles_y, int tpr_log2, int64_t ngroups) {
    constexpr int CY = dw_floor_half(PT - K + 1), NQ = (DWB_TY - 1 + PT) / 2 - CY + 1;       // dy rows under 4 dx rows
    static_assert(4 + dw_floor_half(PL - K + 1) >= 0 && 4 + (7 + PL) / 2 < 12, "window does not fit the loaded float4s");
    const int64_t gid = ((int64_t)xcd_block(blockIdx.x, gridDim.x) * 256 + threadIdx.x) >> tpr_log2;      // XCD-contiguous runs of workgroups (common.h): halo rows shared with the next row group stay in ONE XCD's L2
    if (gid >= ngroups) return;
    const int gpp = tiles_x * tiles_y;
    const int64_t plane = gid / gpp;
    const int rem = (int)(gid - plane * gpp), tyi = rem / tiles_x, txi = rem - tyi * tiles_x;
    const int ix0 = ((txi << tpr_log2) + (threadIdx.x & ((1 << tpr_log2) - 1))) * 8, iy0 = tyi * DWB_TY;
    if (ix0 >= W) return;
    const int c = (int)(plane % C);
    const float* g = dY + plane * OH * OW;
    float w[K * K];
#pragma unroll
    for (int i = 0; i < K * K; ++i) w[i] = Wt[(int64_t)c * K * K + i];
    float acc[DWB_TY][8];
#pragma unroll
    for (int i = 0; i < DWB_TY; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = 0.f;
    int xo[3]; bool okv[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) { const int col = ix0 / 2 - 4 + 4 * i; okv[i] = col >= 0 && col < OW; xo[i] = min(max(col, 0), OW - 4); }
    const int oyb = iy0 / 2 + CY;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int oy = oyb + q;
        float e[12];
        dw4_load_row<3>(e, g + (int64_t)min(max(oy, 0), OH - 1) * OW, xo, okv, oy >= 0 && oy < OH);
#pragma unroll
        for (int i = 0; i < DWB_TY; ++i) {
            const int ky = i + PT - 2 * (q + CY);                      // compile-time after unrolling
            if (ky < 0 || ky >= K) continue;
#pragma unroll
            for (int j = 0; j < 8; ++j)
#pragma unroll
                for (int kx = 0; kx < K; ++kx) {
                    if ((j + PL - kx) % 2 != 0) continue;              // compile-time
                    acc[i][j] += w[ky * K + kx] * e[4 + dw_floor_half(j + PL - kx)];
                }
#pragma unroll
            for (int j = 0; j < 8; ++j) SEGX_PIN(acc[i][j]);
        }
    }
    float* d = dX + plane * H * W + ix0;
#pragma unroll
    for (int i = 0; i < DWB_TY; ++i)
        if (iy0 + i < H) {
            *reinterpret_cast<float4*>(d + (int64_t)(iy0 + i) * W) = make_float4(acc[i][0], acc[i][1], acc[i][2], acc[i][3]);
            *reinterpret_cast<float4*>(d + (int64_t)(iy0 + i) * W + 4) = make_float4(acc[i][4], acc[i][5], acc[i][6], acc[i][7]);
        }
}
// dx[iy,ix] = sum_{ky,kx} w[ky,kx] * dy[(iy+pt-ky)/S, (ix+pl-kx)/S]   (terms with a non-integer quotient vanish)
template <int K, int ST>
__global__ __launch_bounds__(256) void dwconv_bwd_data_kernel(const float* __restrict__ dY, const float* __restrict__ Wt, float* __restrict__ dX,
                                                              int C, int H, int W, int OH, int OW, int pt, int pl, int tiles_x) {
    constexpr int TO = (15 + K - 1) / ST + 2;       // dy rows/cols that can touch a 16-wide dx tile
    __shared__ float tile[TO][TO + 1];
    const int bc = blockIdx.y, c = bc % C;
    const int iy0 = (blockIdx.x / tiles_x) * 16, ix0 = (blockIdx.x % tiles_x) * 16;
    const float* g = dY + (int64_t)bc * OH * OW;
    // first dy row that can contribute: oy >= (iy0 + pt - (K-1)) / S  (ceil), clamp below by 0 handled by the guard
    const int oy0 = (iy0 + pt - (K - 1) >= 0) ? (iy0 + pt - (K - 1) + ST - 1) / ST : -((-(iy0 + pt - (K - 1))) / ST);
    const int ox0 = (ix0 + pl - (K - 1) >= 0) ? (ix0 + pl - (K - 1) + ST - 1) / ST : -((-(ix0 + pl - (K - 1))) / ST);
    for (int i = threadIdx.x; i < TO * TO; i += 256) {
        const int r = i / TO, q = i - r * TO, oy = oy0 + r, ox = ox0 + q;
        tile[r][q] = (oy >= 0 && oy < OH && ox >= 0 && ox < OW) ? g[(int64_t)oy * OW + ox] : 0.f;
    }
    __syncthreads();
    const int ly = threadIdx.x >> 4, lx = threadIdx.x & 15, iy = iy0 + ly, ix = ix0 + lx;
    float acc = 0.f;
    const float* w = Wt + (int64_t)c * K * K;
#pragma unroll
    for (int ky = 0; ky < K; ++ky) {
        const int ny = iy + pt - ky;
        if (ny < 0 || (ny % ST) != 0) continue;
        const int r = ny / ST - oy0;
#pragma unroll
        for (int kx = 0; kx < K; ++kx) {
            const int nx = ix + pl - kx;
            if (nx < 0 || (nx % ST) != 0) continue;
            const int q = nx / ST - ox0;
            if (r >= 0 && r < TO && q >= 0 && q < TO) acc += w[ky * K + kx] * tile[r][q];
        }
    }
    if (iy < H && ix < W) dX[(int64_t)bc * H * W + (int64_t)iy * W + ix] = acc;
}
// dw[c,ky,kx] = sum_{b,oy,ox} dy[b,c,oy,ox] x[b,c,oy*S+ky-pt, ox*S+kx-pl]: same row-sliding walk (K loads of x and one of dy per
// output instead of K*K+1).  grid (strips, B*C): a workgroup walks every strips-th tile of its plane, then reduces its K*K
// sums once (wave shuffles + one LDS pass).  part[(b*strips + strip)][c][K*K] is summed over its rows by segx_colsum:
// deterministic, no float atomics.
template <int K, int ST>
__global__ __launch_bounds__(256) void dwconv_wgrad_rows_kernel(const float* __restrict__ dY, const float* __restrict__ X, float* __restrict__ part,
                                                                int C, int H, int W, int OH, int OW, int pt, int pl, int tiles_x, int ntiles,
                                                                int txw_log2) {
    __shared__ float red[4][K * K];
    const int bc = blockIdx.y, b = bc / C, c = bc - b * C, strips = gridDim.x;
    const float* x = X + (int64_t)bc * H * W; const float* g = dY + (int64_t)bc * OH * OW;
    float acc[K * K];
#pragma unroll
    for (int i = 0; i < K * K; ++i) acc[i] = 0.f;
    for (int tile = blockIdx.x; tile < ntiles; tile += strips) {
        const DwTile t = dw_tile(tile, tiles_x, txw_log2, OH, OW);
        if (!t.live) continue;
        float gv[DW_TY];
#pragma unroll
        for (int i = 0; i < DW_TY; ++i) gv[i] = (t.oy0 + i < OH) ? g[(int64_t)(t.oy0 + i) * OW + t.ox] : 0.f;
        const int ix0 = t.ox * ST - pl, iy0 = t.oy0 * ST - pt;
        constexpr int NR = (DW_TY - 1) * ST + K;
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            const int iy = iy0 + r;
            const bool rowok = iy >= 0 && iy < H;
            const float* xr = x + (int64_t)min(max(iy, 0), H - 1) * W;
            float v[K];
#pragma unroll
            for (int kx = 0; kx < K; ++kx) {
                const int ix = ix0 + kx;
                const float q = xr[min(max(ix, 0), W - 1)];
                v[kx] = (rowok && ix >= 0 && ix < W) ? q : 0.f;
            }
#pragma unroll
            for (int ky = 0; ky < K; ++ky) {
                if ((r - ky) < 0 || (r - ky) % ST != 0 || (r - ky) / ST >= DW_TY) continue;  // compile-time
#pragma unroll
                for (int kx = 0; kx < K; ++kx) acc[ky * K + kx] += gv[(r - ky) / ST] * v[kx];
            }
        }
    }
#pragma unroll
    for (int i = 0; i < K * K; ++i) {
        const float sw = wave_sum(acc[i]);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][i] = sw;
    }
    __syncthreads();
    if (threadIdx.x < K * K)
        part[(((int64_t)b * strips + blockIdx.x) * C + c) * (K * K) + threadIdx.x] =
            (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

// =================================================================================================
// Squeeze-excite gate folded into the projection weights (exact re-association of efficientnet/model.py:110-113):
//   project_conv(y * gate[b, :, None, None]) == conv1x1 of y with the per-sample weights Wb[b][m][k] = W[m][k] * gate[b][k]
// so the gated activation is never written or read.  Backward, from the per-sample weight gradient dWb the GEMM returns:
//   dW[m][k] = sum_b dWb[b][m][k] * gate[b][k] ;   dgate[b][k] = sum_m dWb[b][m][k] * W[m][k]
// (the second is sum_s dz * y of the unfused form, with the sum over the plane done by the weight-gradient GEMM)
// =================================================================================================
// workgroup = 64 columns of sample b x 4 row lanes (rows m = lane, lane + 4, ...: consecutive threads read consecutive floats of a row);
// the four partial sums are added in lane order through LDS

// =================================================================================================
// Squeeze-excite plane ops (efficientnet/model.py:105-110): y = x * gate[b,c] ; dgate[b,c] = sum_s dy * x ;
// dx = dy * gate + dpool[b,c]   (dpool = gradient of the mean pooled value, already divided by S)
// =================================================================================================
__global__ __launch_bounds__(256) void plane_scale_kernel(const float* __restrict__ X, const float* __restrict__ gate, float* __restrict__ Y, int64_t S) {
    const float gt = gate[blockIdx.y];
    const float* x = X + (int64_t)blockIdx.y * S; float* y = Y + (int64_t)blockIdx.y * S;
    for (int64_t s = (int64_t)blockIdx.x * 256 + threadIdx.x; s < S; s += (int64_t)gridDim.x * 256) y[s] = x[s] * gt;
}
// y = x + bias[plane % C]   (bias of a dense convolution whose GEMM runs without one)
__global__ __launch_bounds__(256) void plane_bias_add_kernel(const float* __restrict__ X, const float* __restrict__ bias, float* __restrict__ Y, int C, int64_t S) {
    const float bv = bias[blockIdx.y % C];
    const float* x = X + (int64_t)blockIdx.y * S; float* y = Y + (int64_t)blockIdx.y * S;
    for (int64_t s = (int64_t)blockIdx.x * 256 + threadIdx.x; s < S; s += (int64_t)gridDim.x * 256) y[s] = x[s] + bv;
}
__global__ __launch_bounds__(256) void plane_dot_kernel(const float* __restrict__ A, const float* __restrict__ Bm, float* __restrict__ out, int64_t S) {
    __shared__ float red[4];
    const float* a = A + (int64_t)blockIdx.x * S; const float* b = Bm + (int64_t)blockIdx.x * S;
    float s = 0.f;
    for (int64_t i = threadIdx.x; i < S; i += 256) s += a[i] * b[i];
    s = block_sum<4>(s, red);
    if (threadIdx.x == 0) out[blockIdx.x] = s;
}

// ---- squeeze-excite excitation MLP on the pooled [B, C] vector (efficientnet/model.py:106-110) -------------------
//   p = pooled_sum / S ; hpre = W1 p + b1 ; h = swish(hpre) ; gate = sigmoid(W2 h + b2)
// Every dot product is one WAVE reading a contiguous weight row (W1 [Cs][C], W2 [C][Cs]); the kernels are further down (r04: 2 + 3 launches).
constexpr int SE_MAX_C = 4096, SE_MAX_CS = 256;
// dhpre[b][j] = swish'(hpre) * sum_chunks part ; then dpool[b][c] = sum_j dhpre[b][j] * W1[j][c] / S.  grid (ceil(C/256), B); every
// workgroup recomputes the (tiny) dhpre vector of its sample into LDS, workgroup 0 also writes it out
__global__ __launch_bounds__(256) void se_bwd_pool_kernel(const float* __restrict__ part, const float* __restrict__ hpre, const float* __restrict__ W1,
                                                          float inv_S, float* __restrict__ dhpre_out, float* __restrict__ dpool, int C, int Cs,
                                                          int nchunks) {
    __shared__ float dh[SE_MAX_CS];
    const int b = blockIdx.y;
    for (int j = threadIdx.x; j < Cs; j += 256) {
        float s = 0.f;
#pragma unroll 8
        for (int k = 0; k < nchunks; ++k) s += part[((int64_t)b * nchunks + k) * Cs + j];
        const float v = s * act_grad(hpre[(int64_t)b * Cs + j], ACT_SWISH);
        dh[j] = v;
        if (blockIdx.x == 0) dhpre_out[(int64_t)b * Cs + j] = v;
    }
    __syncthreads();
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    float s = 0.f;
#pragma unroll 16
    for (int j = 0; j < Cs; ++j) s += dh[j] * W1[(int64_t)j * C + c];
    dpool[(int64_t)b * C + c] = s * inv_S;
}

// ---- r04: the same excitation MLP in fewer launches (32 MBConv blocks x 10 micro-kernels of 4..24 us were 2.9 ms of a 76-ms cfg2 step) ---------
// forward  : se_hidden2 (reads the pooling CHUNKS of bn_act_fwd2_kernel<POOL> directly: no chunk-sum launch) -> se_gate_weights (gate + the
//            gate folded into the per-sample projection weights, one launch): 2 launches instead of 5
// backward : se_bwd_gate (dgate from the per-sample weight gradient, dz2, hidden partials over 64-channel chunks) -> se_bwd_pool_kernel ->
//            se_wgrad_all (dW1, db1, dW2, db2 AND the projection weight gradient dW): 3 launches instead of 5
constexpr int SE2_CHUNK = 64, SE2_ROWS = 64;
// hpre[b][j] = b1[j] + sum_c W1[j][c] * p[b][c], p = (sum of the plane's nch pooling chunks) / S; also p (kept for the weight gradients).  grid (ceil(Cs/4), B)
__global__ __launch_bounds__(256) void se_hidden2_kernel(const float* __restrict__ psum, int nch, float inv_S, const float* __restrict__ W1,
                                                         const float* __restrict__ b1, float* __restrict__ p_out, float* __restrict__ hpre_out, int C, int Cs) {
    // r04-p: the pooled means of the sample go through LDS once per workgroup (every wave used to re-add the pooling chunks of every channel it touched)
    __shared__ float psh[SE_MAX_C];
    const int b = blockIdx.y, lane = threadIdx.x & 63, j = blockIdx.x * 4 + (threadIdx.x >> 6);
    const float* ps = psum + (int64_t)b * C * nch;
    for (int c = threadIdx.x; c < C; c += 256) {
        float a = 0.f;
        for (int k = 0; k < nch; ++k) a += ps[(int64_t)c * nch + k];
        a *= inv_S;
        psh[c] = a;
        if (blockIdx.x == 0) p_out[(int64_t)b * C + c] = a;
    }
    __syncthreads();
    if (j >= Cs) return;
    float s = 0.f;
    const float* w1 = W1 + (int64_t)j * C;
#pragma unroll 8
    for (int c = lane; c < C; c += 64) s += w1[c] * psh[c];
    s = wave_sum(s);
    if (lane == 0) hpre_out[(int64_t)b * Cs + j] = s + b1[j];
}
// gate[b][k] = sigmoid(b2[k] + sum_j W2[k][j] swish(hpre[b][j])) for the 64 channels k of this workgroup, then Wb[b][m][k] = W[m][k] * gate[b][k] for every
// output row m of the projection (efficientnet/model.py:110-113 re-associated).  grid (ceil(K / 64), B); W may be NULL (gate only)
__global__ __launch_bounds__(256) void se_gate_weights_kernel(const float* __restrict__ hpre, const float* __restrict__ W2, const float* __restrict__ b2,
                                                              const float* __restrict__ W, float* __restrict__ gate, float* __restrict__ Wb,
                                                              int K, int Cs, int M) {
    // r04-p: lane = gate column, wave w takes the hidden units j = w, w + 4, ... (independent loads, a few round trips), the four partial sums are added in
    // wave order through LDS.  Before, a wave worked through 16 columns one after the other, each a load -> wave_sum chain: 19 us per layer for kilobytes.
    __shared__ float hsh[SE_MAX_CS];
    __shared__ float psh[4][SE2_CHUNK];
    __shared__ float gsh[SE2_CHUNK];
    const int b = blockIdx.y, lane = threadIdx.x & 63, wv = threadIdx.x >> 6, k0 = blockIdx.x * SE2_CHUNK;
    for (int j = threadIdx.x; j < Cs; j += 256) { const float hp = hpre[(int64_t)b * Cs + j]; hsh[j] = hp * sigm(hp); }
    __syncthreads();
    {
        const int k = k0 + lane;
        float s = 0.f;
        if (k < K) {
            const float* w2 = W2 + (int64_t)k * Cs;
#pragma unroll 8
            for (int j = wv; j < Cs; j += 4) s += w2[j] * hsh[j];
        }
        psh[wv][lane] = s;
    }
    __syncthreads();
    if (wv == 0) {
        const int k = k0 + lane;
        const float gt = k < K ? sigm((((psh[0][lane] + psh[1][lane]) + psh[2][lane]) + psh[3][lane]) + b2[k]) : 0.f;
        gsh[lane] = gt;
        if (k < K && blockIdx.z == 0) gate[(int64_t)b * K + k] = gt;
    }
    __syncthreads();
    if (!W) return;
    const int k = k0 + lane;
    if (k >= K) return;
    const float gt = gsh[lane];
    float* o = Wb + (int64_t)b * M * K;
    // gridDim.z slabs of SE2_ROWS output rows: every slab recomputes the 64 gates (a few hundred cycles) so that the M x K scaling is spread over
    // M / 64 times as many workgroups (r04_d: 24 us per layer with one workgroup per (64 columns, sample) -- latency-bound -- against 9.5 us for the two
    // kernels this one replaces); slab 0 alone writes the gate
    const int m1 = min(M, ((int)blockIdx.z + 1) * SE2_ROWS);
#pragma unroll 4
    for (int m = (int)blockIdx.z * SE2_ROWS + wv; m < m1; m += 4) o[(int64_t)m * K + k] = W[(int64_t)m * K + k] * gt;
}
// dgate[b][k] = sum_m dWb[b][m][k] W[m][k] (W == NULL: dgate is given), dz2 = dgate * gate * (1 - gate), part[b][chunk][j] = sum_{k in chunk} dz2[k] W2[k][j].
// grid (ceil(K / 64), B): 64 columns x 4 row lanes, the four partial sums added in lane order through LDS (as gate_weights_bwd_g_kernel)
__global__ __launch_bounds__(256) void se_bwd_gate_kernel(const float* __restrict__ dWb, const float* __restrict__ W, const float* __restrict__ dgate_in,
                                                          const float* __restrict__ gate, const float* __restrict__ W2, float* __restrict__ dz2_out,
                                                          float* __restrict__ part, int M, int K, int Cs) {
    __shared__ float sh[256];
    __shared__ float dz[SE2_CHUNK];
    const int b = blockIdx.y, col = threadIdx.x & 63, rl = threadIdx.x >> 6, k0 = blockIdx.x * SE2_CHUNK, k = k0 + col;
    float s = 0.f;
    if (W && k < K) {
        const float* d = dWb + (int64_t)b * M * K;
#pragma unroll 16
        for (int m = rl; m < M; m += 4) s += d[(int64_t)m * K + k] * W[(int64_t)m * K + k];
    }
    sh[threadIdx.x] = s;
    __syncthreads();
    if (rl == 0) {
        float v = 0.f;
        if (k < K) {
            const float dg = W ? (sh[col] + sh[64 + col]) + (sh[128 + col] + sh[192 + col]) : dgate_in[(int64_t)b * K + k];
            const float gt = gate[(int64_t)b * K + k];
            v = dg * gt * (1.0f - gt);
            dz2_out[(int64_t)b * K + k] = v;
        }
        dz[col] = v;
    }
    __syncthreads();
    for (int j = threadIdx.x; j < Cs; j += 256) {
        const int n = min(SE2_CHUNK, K - k0);
        float a = 0.f;
#pragma unroll 16
        for (int i = 0; i < n; ++i) a += dz[i] * W2[(int64_t)(k0 + i) * Cs + j];
        part[((int64_t)b * gridDim.x + blockIdx.x) * Cs + j] = a;
    }
}
// every weight gradient of the block's squeeze-excite part in one launch: idx < C * Cs: (dW1, dW2, db1, db2) as se_gate_wgrad_kernel; then M * K
// elements of dW[m][k] = sum_b dWb[b][m][k] gate[b][k] (as gate_weights_bwd_w_kernel; skipped when dWb == NULL)
__global__ __launch_bounds__(256) void se_wgrad_all_kernel(const float* __restrict__ dz2, const float* __restrict__ dhpre, const float* __restrict__ p,
                                                           const float* __restrict__ hpre, float* __restrict__ dW1, float* __restrict__ db1,
                                                           float* __restrict__ dW2, float* __restrict__ db2, const float* __restrict__ dWb,
                                                           const float* __restrict__ gate, float* __restrict__ dW, int B, int C, int Cs, int64_t MK) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x, n1 = (int64_t)C * Cs;
    if (idx < n1) {
        const int c = (int)(idx / Cs), j = (int)(idx % Cs);
        float a1 = 0.f, a2 = 0.f, s1 = 0.f, s2 = 0.f;
        for (int b = 0; b < B; ++b) {
            const float hp = hpre[(int64_t)b * Cs + j], dz = dz2[(int64_t)b * C + c], dh = dhpre[(int64_t)b * Cs + j];
            a2 += dz * (hp * sigm(hp)); a1 += dh * p[(int64_t)b * C + c]; s2 += dz; s1 += dh;
        }
        dW2[(int64_t)c * Cs + j] = a2; dW1[(int64_t)j * C + c] = a1;
        if (j == 0) db2[c] = s2;
        if (c == 0) db1[j] = s1;
        return;
    }
    const int64_t i = idx - n1;
    if (!dWb || i >= MK) return;
    const int k = (int)(i % C);
    float s = 0.f;
    for (int b = 0; b < B; ++b) s += dWb[(int64_t)b * MK + i] * gate[(int64_t)b * C + k];
    dW[i] = s;
}

static inline int plane_chunks(int64_t S, int per_thread) { return (int)i64max(1, i64min(64, (S + 256 * per_thread - 1) / (256 * per_thread))); }

}  // namespace segx

using namespace segx;
#define SEGX_STREAM hipStream_t stream = (hipStream_t)stream_

static inline int64_t bn_team_floats(int B, int C, int64_t S, bool backward) {       // partial slots + mailboxes + counters of the team form, 0 where it does not serve
    int kp = 0;
    if (S <= 0 || bn_auto_form(B, S, backward, &kp) != 1) return 0;
    return (int64_t)C * B * bn_team_chunks(S, kp) * (TEAM_SLOT + TEAM_MBOX);
}
// diagnostics: hold compute units for `ms` milliseconds -- `wgs` workgroups of 256 threads, each with 80 KB of LDS (heavy != 0: two fill a CU's LDS, so no
// kernel that needs LDS becomes resident beside them) or 64 bytes.  The team-exchange tests run the team BatchNorm next to it on a second stream.
template <int LDS_FLOATS>
__global__ __launch_bounds__(256) void occupy_kernel(unsigned long long ticks, float* sink) {
    __shared__ float hold[LDS_FLOATS];
    const unsigned long long t0 = wall_clock64();
    float acc = 0.f;
    while (wall_clock64() - t0 < ticks) { acc += hold[threadIdx.x & 15]; __builtin_amdgcn_s_sleep(32); }
    if (sink && acc == 12345.678f) sink[0] = acc;                      // keeps the LDS reads (and with them the allocation) alive
}
extern "C" int segx_occupy(int wgs, int heavy, float ms, float* sink, void* stream_) {
    SEGX_STREAM; SEGX_REQUIRE(wgs > 0 && wgs <= 65535 && ms >= 0.f && ms <= 2000.f, "segx_occupy: bad args");
    const unsigned long long ticks = (unsigned long long)(ms * 1e5f);                   // wall_clock64 counts at 100 MHz
    if (heavy) hipLaunchKernelGGL((occupy_kernel<20 * 1024>), dim3(wgs), dim3(256), 0, stream, ticks, sink);
    else hipLaunchKernelGGL((occupy_kernel<16>), dim3(wgs), dim3(256), 0, stream, ticks, sink);
    return check_launch("segx_occupy");
}
/* team exchanges whose poll expired since the last clear (common.h: team_exchange): read from pinned host memory, no synchronisation */
extern "C" int segx_team_status(int clear) {
    const TeamHost& h = team_host();
    const unsigned n = __atomic_load_n(h.host, __ATOMIC_RELAXED);
    if (clear && n) __atomic_fetch_sub(h.host, n, __ATOMIC_RELAXED);
    return (int)(n > 0x7fffffffu ? 0x7fffffffu : n);
}
extern "C" int segx_team_cap(void) { return team_cap(); }
extern "C" int64_t segx_bn_ws_floats(int B, int C, int64_t S) { return i64max((int64_t)B * C * BN_SLABS * 2, bn_team_floats(B, C, S, true)); }
/* the same pass that also leaves pooled[b][c] = sum over the plane of y (the squeeze-excite pooling of efficientnet/model.py:106); ws: B*C*64 floats */
extern "C" int segx_bn_act_bwd_reduce(const float* dY, const float* X, const float* mean, const float* var, const float* w, const float* b,
                                      float* dw, float* db, float* ws, int B, int C, int64_t S, float eps, int act,
                                      const float* gate, const float* dpool, float inv_S, float dc_p, uint64_t seed, uint64_t offset, void* stream_) {
    SEGX_STREAM; SEGX_REQUIRE(dY && X && mean && var && w && b && dw && db && ws && B > 0 && C > 0 && S > 0 && (dpool || !gate), "segx_bn_act_bwd_reduce: bad args");
    hipLaunchKernelGGL(bn_act_bwd_stage1, dim3(C, B, bn_slabs(S)), dim3(256), 0, stream, dY, X, mean, var, w, b, ws, C, S, eps, act, gate, dpool, inv_S, dc_p, seed, offset,
                       rng_base(), (int64_t)C * S);
    hipLaunchKernelGGL(bn_act_bwd_stage2, dim3((C + 255) / 256), dim3(256), 0, stream, (const float*)ws, dw, db, B, C, bn_slabs(S));
    return check_launch("segx_bn_act_bwd_reduce");
}
extern "C" int segx_bn_act_bwd_apply(const float* dY, const float* X, const float* mean, const float* var, const float* w, const float* b,
                                     const float* sum_dw, const float* sum_db, float* dX, int B, int C, int64_t S, float eps, int act,
                                     float inv_n, const float* gate, const float* dpool, float inv_S, float dc_p, uint64_t seed, uint64_t offset, void* stream_) {
    SEGX_STREAM; SEGX_REQUIRE(dY && X && mean && var && w && b && sum_dw && sum_db && dX && B > 0 && C > 0 && S > 0 && (dpool || !gate), "segx_bn_act_bwd_apply: bad args");
    SEGX_REQUIRE((int64_t)B * C <= 65535, "segx_bn_act_bwd_apply: more than 65535 (sample, channel) planes");
    hipLaunchKernelGGL((bn_act_bwd_apply<false>), dim3(plane_chunks(S, 8), B * C), dim3(256), 0, stream, dY, X, mean, var, w, b, sum_dw, sum_db, dX, C, S, eps, act, inv_n,
                       gate, dpool, inv_S, dc_p, seed, offset, rng_base(), (const float*)nullptr, 0, (float*)nullptr, (float*)nullptr, (int64_t)C * S);
    return check_launch("segx_bn_act_bwd_apply");
}

// ---- r04: two-launch training BatchNorm (bn_stats_partial_kernel / a producer's partials -> bn_act_fwd2_kernel) ------------------------------
extern "C" int64_t segx_plane_chunks(int64_t S) { return plane_chunks(S, 8); }
/* pooling chunks per plane that segx_bn_act_fwd2 writes into psum: auto_stats != 0 = the call computes the statistics itself (parts given, nparts = 0) */
extern "C" int64_t segx_bn_pool_chunks(int B, int64_t S, int auto_stats) {
    int kp = 0;
    const int af = auto_stats ? bn_auto_form(B, S, false, &kp) : 0;
    return af == 2 ? 1 : af == 1 ? bn_team_chunks(S, kp) : plane_chunks(S, 8);
}
extern "C" int64_t segx_bn_parts_floats(int B, int C, int64_t S) { return i64max(((int64_t)B * BN_SLABS + 1) * C * 4, bn_team_floats(B, C, S, false)); }
extern "C" int segx_bn_act_fwd2(const float* X, const float* parts, int nparts, float* mean, float* var, float* run_mean, float* run_var, float momentum,
                                const float* w, const float* b, float* Y, float* psum, const float* resid, float dc_p, uint64_t seed, uint64_t offset,
                                int B, int C, int64_t S, float eps, int act, int64_t parts_floats, int64_t y_bs, void* stream_) {
    SEGX_STREAM; SEGX_REQUIRE(X && mean && var && w && b && Y && B > 0 && C > 0 && S > 0 && act >= 0 && act <= 3 && (!run_mean == !run_var), "segx_bn_act_fwd2: bad args");
    // y_bs != 0: Y is a channel slice of a wider [B][Ctot][S] tensor (a branch of a channel concatenation written in place): planes contiguous, 16-byte aligned
    SEGX_REQUIRE(y_bs == 0 || (y_bs >= (int64_t)C * S && (y_bs & 3) == 0 && (S & 3) == 0 && (reinterpret_cast<uintptr_t>(Y) & 15) == 0),
                 "segx_bn_act_fwd2: an output slice needs S %% 4 == 0, a batch stride that is a multiple of 4 floats and >= C * S, and a 16-byte aligned base");
    // the buffer behind `parts` is written by this call in the AUTO and merge cases: its size is part of the contract, re-derived here under the knob
    // settings of THIS call (a knob change between sizing and launch used to write past it)
    if (parts) {
        const int64_t need = nparts == 0 ? segx_bn_parts_floats(B, C, S) : nparts < 0 ? (int64_t)C * (-(int64_t)nparts) * 4 : (int64_t)C * ((int64_t)nparts + (nparts > 256 ? 1 : 0)) * 4;
        SEGX_REQUIRE(parts_floats >= need, "segx_bn_act_fwd2: the partials buffer holds %lld floats, this call needs %lld (segx_bn_parts_floats under the current knob 3)",
                     (long long)parts_floats, (long long)need);
    }
    SEGX_REQUIRE((int64_t)B * C <= 65535, "segx_bn_act_fwd2: more than 65535 (sample, channel) planes");
    SEGX_REQUIRE(!parts || (reinterpret_cast<uintptr_t>(parts) & 15) == 0, "segx_bn_act_fwd2: bad partials");
    SEGX_REQUIRE(dc_p >= 0.f && dc_p < 1.f && (dc_p == 0.f || resid), "segx_bn_act_fwd2: drop_connect needs the skip input and 0 <= p < 1");
    BnFwdArgs g;
    g.X = X; g.w = w; g.b = b; g.Y = Y; g.parts = parts; g.nparts = nparts; g.mean = mean; g.var = var; g.run_mean = run_mean; g.run_var = run_var;
    g.momentum = momentum; g.psum = psum; g.resid = resid; g.dc_p = dc_p; g.seed = seed; g.offset = offset; g.rbase = rng_base();
    g.C = C; g.S = S; g.eps = eps; g.act = act; g.part_cstride = -1; g.part_istride = 1; g.y_bs = y_bs ? y_bs : (int64_t)C * S;
    if (parts && nparts < 0) {                               // -nparts per-rank partials laid out [ranks][C] (segx_bn_stats_local on every rank, all-gathered)
        g.nparts = nparts = -nparts; g.part_cstride = 1; g.part_istride = C;
    }
    if (parts && nparts == 0) {
        // AUTO: the library computes the batch statistics itself -- channel-resident (one launch) where the channel's B planes fit a team's registers,
        // otherwise partials into `parts` (segx_bn_parts_floats) + the folding apply pass below
        int tkp = 0;
        const int af = bn_auto_form(B, S, false, &tkp);
        if (af == 1) {
            BnTeam t; t.B = B; t.cpp = bn_team_chunks(S, tkp); t.slots = const_cast<float*>(parts); t.mbox = t.slots + (int64_t)C * B * t.cpp * TEAM_SLOT;
            team_fill(t);
            SEGX_REQUIRE(B * t.cpp <= team_cap() && (int64_t)C * B * t.cpp < 2147483647LL, "segx_bn_act_fwd2: team of %d workgroups", B * t.cpp);
            g.parts = nullptr;
            const dim3 tg = team_grid((int64_t)C * B * t.cpp);
            if (tkp == 32) return bn_team_launch_fwd<32>(g, t, tg, stream, psum != nullptr);
            if (tkp == 4) return bn_team_launch_fwd<4>(g, t, tg, stream, psum != nullptr);
            if (tkp == 8) return bn_team_launch_fwd<8>(g, t, tg, stream, psum != nullptr);
            return bn_team_launch_fwd<16>(g, t, tg, stream, psum != nullptr);
        }
        const int form = af == 2 ? bn_res_form(B, S, false) : 0;
        if (form) {
            g.parts = nullptr;
            const int team = form >> 4, kp = form & 15;
            const dim3 rgrid(team == 64 ? (C + 3) / 4 : C);
#define SEGX_BN_RES(T, K, BM) if (team == T && kp == K) { bn_res_launch_fwd<T, K, BM>(g, B, rgrid, stream, psum != nullptr); return check_launch("segx_bn_act_fwd2/resident"); }
            if (B <= 6) { SEGX_BN_RES(256, 4, 6) SEGX_BN_RES(256, 1, 6) }     // six planes: no registers (and no clamped re-reads) for planes 7 and 8
            SEGX_BN_RES(64, 1, 8) SEGX_BN_RES(64, 2, 8) SEGX_BN_RES(256, 1, 8) SEGX_BN_RES(256, 2, 8) SEGX_BN_RES(256, 4, 8)
#undef SEGX_BN_RES
            return fail(-1, "segx_bn_act_fwd2: no resident form %d", form);
        }
        hipLaunchKernelGGL(bn_stats_partial_kernel, dim3(C, B, bn_slabs(S)), dim3(256), 0, stream, X, const_cast<float*>(parts), C, S);
        g.nparts = nparts = B * bn_slabs(S);
    }
    if (parts && nparts > 256 && g.part_istride == 1) {
        // a producer with many small tiles: one merge launch, the apply pass then folds ONE partial per channel.  The merged partials live behind the
        // producer's (the caller sized the buffer for nparts + 1 per channel).
        float* merged = const_cast<float*>(parts) + (int64_t)C * nparts * 4;
        hipLaunchKernelGGL(bn_parts_merge_kernel, dim3((C + 3) / 4), dim3(256), 0, stream, parts, merged, C, nparts);
        g.parts = merged; g.nparts = 1;
    }
    const dim3 grid(plane_chunks(S, 8), B * C);
    if (psum) hipLaunchKernelGGL((bn_act_fwd2_kernel<true>), grid, dim3(256), 0, stream, g);
    else hipLaunchKernelGGL((bn_act_fwd2_kernel<false>), grid, dim3(256), 0, stream, g);
    return check_launch("segx_bn_act_fwd2");
}
/* ONE partial (n, mean, M2) per channel of this process's batch: part [C] float4.  One launch where the channel fits a team's registers, else the
 * slab partials (into ws: segx_bn_parts_floats) + one merge launch */
extern "C" int segx_bn_stats_local(const float* X, float* part, float* ws, int B, int C, int64_t S, void* stream_) {
    SEGX_STREAM; SEGX_REQUIRE(X && part && ws && B > 0 && B <= 65535 && C > 0 && S > 0 && ((reinterpret_cast<uintptr_t>(part) | reinterpret_cast<uintptr_t>(ws)) & 15) == 0, "segx_bn_stats_local: bad args");
    const int form = bn_res_form(B, S, false);
    if (form) {
        BnFwdArgs g; memset(&g, 0, sizeof(g));
        g.X = X; g.psum = part; g.C = C; g.S = S;
        const int team = form >> 4, kp = form & 15;
        const dim3 rgrid(team == 64 ? (C + 3) / 4 : C);
#define SEGX_BN_ST(T, K) if (team == T && kp == K) { hipLaunchKernelGGL((bn_act_fwd_res_kernel<T, K, 8, false, -1, 0, true>), rgrid, dim3(256), 0, stream, g, B); return check_launch("segx_bn_stats_local"); }
        SEGX_BN_ST(64, 1) SEGX_BN_ST(64, 2) SEGX_BN_ST(256, 1) SEGX_BN_ST(256, 2) SEGX_BN_ST(256, 4)
#undef SEGX_BN_ST
        return fail(-1, "segx_bn_stats_local: no resident form %d", form);
    }
    const int nsl = bn_slabs(S);
    hipLaunchKernelGGL(bn_stats_partial_kernel, dim3(C, B, nsl), dim3(256), 0, stream, X, ws, C, S);
    hipLaunchKernelGGL(bn_parts_merge_kernel, dim3((C + 3) / 4), dim3(256), 0, stream, (const float*)ws, part, C, B * nsl);
    return check_launch("segx_bn_stats_local");
}
extern "C" int segx_bn_act_bwd2(const float* dY, const float* X, const float* mean, const float* var, const float* w, const float* b,
                                float* dX, float* dw, float* db, float* ws, int B, int C, int64_t S, float eps, int act, int training,
                                const float* gate, const float* dpool, float inv_S, float dc_p, uint64_t seed, uint64_t offset, int64_t dy_bs, int64_t ws_floats, void* stream_) {
    SEGX_STREAM; SEGX_REQUIRE(dY && X && mean && var && w && b && dX && dw && db && ws && B > 0 && C > 0 && S > 0 && (dpool || !gate), "segx_bn_act_bwd2: bad args");
    {
        const int64_t need = segx_bn_ws_floats(B, C, training ? S : 0);
        SEGX_REQUIRE(ws_floats >= need, "segx_bn_act_bwd2: the scratch holds %lld floats, this call needs %lld (segx_bn_ws_floats under the current knob 3)", (long long)ws_floats, (long long)need);
    }
    if (dy_bs == 0) dy_bs = (int64_t)C * S;
    SEGX_REQUIRE(dy_bs >= (int64_t)C * S && ((S & 3) != 0 || (dy_bs & 3) == 0), "segx_bn_act_bwd2: bad dY batch stride %lld", (long long)dy_bs);
    SEGX_REQUIRE((int64_t)B * C <= 65535 && dc_p >= 0.f && dc_p < 1.f, "segx_bn_act_bwd2: more than 65535 (sample, channel) planes / bad drop_connect rate");
    int tkp = 0;
    const int af = training ? bn_auto_form(B, S, true, &tkp) : 0;
    BnBwdArgs g;
    g.dY = dY; g.X = X; g.mean = mean; g.var = var; g.w = w; g.b = b; g.dX = dX; g.dw = dw; g.db = db; g.gate = gate; g.dpool = dpool; g.inv_S = inv_S;
    g.dc_p = dc_p; g.seed = seed; g.offset = offset; g.rbase = rng_base(); g.C = C; g.S = S; g.eps = eps; g.act = act; g.dy_bs = dy_bs;
    if (af == 1) {
        BnTeam t; t.B = B; t.cpp = bn_team_chunks(S, tkp); t.slots = ws; t.mbox = ws + (int64_t)C * B * t.cpp * TEAM_SLOT;
        team_fill(t);
        SEGX_REQUIRE(B * t.cpp <= team_cap() && (int64_t)C * B * t.cpp < 2147483647LL, "segx_bn_act_bwd2: team of %d workgroups", B * t.cpp);
        const dim3 tg = team_grid((int64_t)C * B * t.cpp);
        if (tkp == 4) return bn_team_launch_bwd<4>(g, t, tg, stream);
        if (tkp == 8) return bn_team_launch_bwd<8>(g, t, tg, stream);
        return bn_team_launch_bwd<16>(g, t, tg, stream);
    }
    const int form = af == 2 ? bn_res_form(B, S, true) : 0;
    if (form) {
        const int team = form >> 4, kp = form & 15;
        const dim3 rgrid(team == 64 ? (C + 3) / 4 : C);
        if (team == 64) bn_res_launch_bwd<64, 1, 8>(g, B, rgrid, stream);
        else if (kp == 1 && B <= 6) bn_res_launch_bwd<256, 1, 6>(g, B, rgrid, stream);
        else if (kp == 1) bn_res_launch_bwd<256, 1, 8>(g, B, rgrid, stream);
        else if (kp == 2) bn_res_launch_bwd<256, 2, 8>(g, B, rgrid, stream);
        else return fail(-1, "segx_bn_act_bwd2: no resident form %d", form);
        return check_launch("segx_bn_act_bwd2/resident");
    }
    const int nsl = bn_slabs(S);
    hipLaunchKernelGGL(bn_act_bwd_stage1, dim3(C, B, nsl), dim3(256), 0, stream, dY, X, mean, var, w, b, ws, C, S, eps, act, gate, dpool, inv_S, dc_p, seed, offset, rng_base(), dy_bs);
    const float inv_n = training ? 1.0f / ((float)B * (float)S) : 0.f;
    hipLaunchKernelGGL((bn_act_bwd_apply<true>), dim3(plane_chunks(S, 8), B * C), dim3(256), 0, stream, dY, X, mean, var, w, b, (const float*)nullptr,
                       (const float*)nullptr, dX, C, S, eps, act, inv_n, gate, dpool, inv_S, dc_p, seed, offset, rng_base(), (const float*)ws, B * nsl, dw, db, dy_bs);
    return check_launch("segx_bn_act_bwd2");
}
// ---- r04: squeeze-excite in 2 + 3 launches ------------------------------------------------------------------------------------------------------
extern "C" int segx_se_fwd2(const float* psum, int nch, float inv_S, const float* W1, const float* b1, const float* W2, const float* b2, const float* Wproj,
                            float* p, float* hpre, float* gate, float* Wb, int B, int C, int Cs, int M, void* stream_) {
    SEGX_STREAM; SEGX_REQUIRE(psum && nch > 0 && W1 && b1 && W2 && b2 && p && hpre && gate && B > 0 && C > 0 && Cs > 0 && (!Wproj || (Wb && M > 0)), "segx_se_fwd2: bad args");
    SEGX_REQUIRE(C <= SE_MAX_C && Cs <= SE_MAX_CS && B <= 65535, "segx_se_fwd2: C=%d / Cs=%d exceed %d / %d", C, Cs, SE_MAX_C, SE_MAX_CS);
    hipLaunchKernelGGL(se_hidden2_kernel, dim3((Cs + 3) / 4, B), dim3(256), 0, stream, psum, nch, inv_S, W1, b1, p, hpre, C, Cs);
    hipLaunchKernelGGL(se_gate_weights_kernel, dim3((C + SE2_CHUNK - 1) / SE2_CHUNK, B, Wproj ? (M + SE2_ROWS - 1) / SE2_ROWS : 1), dim3(256), 0, stream, (const float*)hpre, W2, b2, Wproj, gate, Wb, C, Cs, M);
    return check_launch("segx_se_fwd2");
}
extern "C" int64_t segx_se_ws2_floats(int B, int C, int Cs) { return (int64_t)B * (C + Cs) + (int64_t)B * ((C + SE2_CHUNK - 1) / SE2_CHUNK) * Cs; }
extern "C" int segx_se_bwd2(const float* dWb, const float* Wproj, const float* dgate, const float* gate, const float* hpre, const float* p, const float* W1,
                            const float* W2, float inv_S, float* dpool, float* dW1, float* db1, float* dW2, float* db2, float* dWproj, float* ws,
                            int B, int C, int Cs, int M, void* stream_) {
    SEGX_STREAM; SEGX_REQUIRE(gate && hpre && p && W1 && W2 && dpool && dW1 && db1 && dW2 && db2 && ws && B > 0 && C > 0 && Cs > 0, "segx_se_bwd2: bad args");
    SEGX_REQUIRE((dWb && Wproj && dWproj && M > 0) || (!dWb && !Wproj && dgate), "segx_se_bwd2: either (dWb, Wproj, dWproj, M) or dgate");
    SEGX_REQUIRE(C <= SE_MAX_C && Cs <= SE_MAX_CS && B <= 65535, "segx_se_bwd2: C=%d / Cs=%d exceed %d / %d", C, Cs, SE_MAX_C, SE_MAX_CS);
    const int nchunks = (C + SE2_CHUNK - 1) / SE2_CHUNK;
    float* dz2 = ws; float* dhpre = ws + (int64_t)B * C; float* part = dhpre + (int64_t)B * Cs;
    hipLaunchKernelGGL(se_bwd_gate_kernel, dim3(nchunks, B), dim3(256), 0, stream, dWb, Wproj, dgate, gate, W2, dz2, part, M, C, Cs);
    hipLaunchKernelGGL(se_bwd_pool_kernel, dim3((C + 255) / 256, B), dim3(256), 0, stream, (const float*)part, hpre, W1, inv_S, dhpre, dpool, C, Cs, nchunks);
    const int64_t MK = dWb ? (int64_t)M * C : 0, total = (int64_t)C * Cs + MK;
    hipLaunchKernelGGL(se_wgrad_all_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, (const float*)dz2, (const float*)dhpre, p, hpre,
                       dW1, db1, dW2, db2, dWb, gate, dWproj, B, C, Cs, MK);
    return check_launch("segx_se_bwd2");
}

#define SEGX_DW_DISPATCH(KERNEL, ...)                                                                          \
    if (k == 3 && stride == 1) hipLaunchKernelGGL((KERNEL<3, 1>), grid, dim3(256), 0, stream, __VA_ARGS__);     \
    else if (k == 3 && stride == 2) hipLaunchKernelGGL((KERNEL<3, 2>), grid, dim3(256), 0, stream, __VA_ARGS__); \
    else if (k == 5 && stride == 1) hipLaunchKernelGGL((KERNEL<5, 1>), grid, dim3(256), 0, stream, __VA_ARGS__); \
    else if (k == 5 && stride == 2) hipLaunchKernelGGL((KERNEL<5, 2>), grid, dim3(256), 0, stream, __VA_ARGS__); \
    else return segx::fail(-1, "depthwise conv: kernel %d stride %d unsupported (k in {3,5}, stride in {1,2})", k, stride);
#define SEGX_DW_ROWS_DISPATCH(FLIP, ...)                                                                                          \
    if (k == 3 && stride == 1) hipLaunchKernelGGL((dwconv_rows_kernel<3, 1, FLIP>), grid, dim3(256), 0, stream, __VA_ARGS__);      \
    else if (k == 3 && stride == 2) hipLaunchKernelGGL((dwconv_rows_kernel<3, 2, FLIP>), grid, dim3(256), 0, stream, __VA_ARGS__); \
    else if (k == 5 && stride == 1) hipLaunchKernelGGL((dwconv_rows_kernel<5, 1, FLIP>), grid, dim3(256), 0, stream, __VA_ARGS__); \
    else if (k == 5 && stride == 2) hipLaunchKernelGGL((dwconv_rows_kernel<5, 2, FLIP>), grid, dim3(256), 0, stream, __VA_ARGS__); \
    else return segx::fail(-1, "depthwise conv: kernel %d stride %d unsupported (k in {3,5}, stride in {1,2})", k, stride);

namespace {
struct DwGrid { int txw_log2, tiles_x, tiles_y; };
// scalar kernels: column tile = smallest power of two >= OW in [32, 256]; (256 / tile) groups of DW_TY rows per workgroup
DwGrid dw_grid(int OH, int OW) {
    int l = 5;
    while (l < 8 && (1 << l) < OW) ++l;
    const int rows = (256 >> l) * segx::DW_TY;
    return {l, (OW + (1 << l) - 1) >> l, (OH + rows - 1) / rows};
}
// float4 kernels: threads per row = smallest power of two >= OW/4 in [8, 64]
struct Dw4Grid { int tpr_log2, tiles_x; };
Dw4Grid dw4_grid(int OW) {
    int l = 3;
    while (l < 6 && (4 << l) < OW) ++l;
    return {l, (OW + (4 << l) - 1) / (4 << l)};
}
// strips per plane for the weight gradient (depends on the output size only): one wave per (plane, strip), ~8K outputs per strip, at most 32.
// More, smaller strips (1K outputs: 8x the waves on the early 128 x 128 stages) were measured 0.3 ms/step SLOWER on cfg2 (r02_r, same box,
// segx_tune knob 8): the partial-sum rows and their column sums grow with the strip count, the kernel itself does not speed up.
int dw_wgrad_strips(int OH, int OW) {
    const int g_dw_strip_outputs = segx::kget(segx::knobs().dw_strip_outputs);
    const int64_t n = ((int64_t)OH * OW + g_dw_strip_outputs - 1) / g_dw_strip_outputs;
    return (int)(n < 1 ? 1 : n > 32 ? 32 : n);
}
bool dw4_ok(const void* a, const void* b, int W, int OW) {
    return W % 4 == 0 && OW % 4 == 0 && W >= 4 && ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b)) & 15) == 0;
}
// float4 forward-type launch for the (k, stride, pad_l) combinations the model uses; false = not instantiated
template <bool FLIP>
bool dw4_rows_launch(hipStream_t stream, const float* X, const float* W, float* Y, int planes, int C, int H, int Wd, int OH, int OW, int k,
                     int stride, int pt, int pl) {
    const Dw4Grid g = dw4_grid(OW);
    const int tiles_y = (OH + segx::DW4_TY - 1) / segx::DW4_TY;
    const int64_t ngroups = (int64_t)planes * g.tiles_x * tiles_y;
    const int64_t nblocks = ((ngroups << g.tpr_log2) + 255) / 256;
    if (nblocks > 2147483647LL) return false;
#define SEGX_DW4(KK, SS, PP)                                                                                                        \
    if (k == KK && stride == SS && pl == PP) {                                                                                       \
        hipLaunchKernelGGL((segx::dwconv_rows4_kernel<KK, SS, PP, FLIP>), dim3((unsigned)nblocks), dim3(256), 0, stream, X, W, Y, C, \
                           H, Wd, OH, OW, pt, g.tiles_x, tiles_y, g.tpr_log2, ngroups);                                              \
        return true;                                                                                                                 \
    }
    SEGX_DW4(3, 1, 1) SEGX_DW4(5, 1, 2) SEGX_DW4(3, 2, 0) SEGX_DW4(3, 2, 1) SEGX_DW4(5, 2, 1) SEGX_DW4(5, 2, 2)
#undef SEGX_DW4
    return false;
}
bool dw4_wgrad_launch(hipStream_t stream, const float* dY, const float* X, float* part, int B, int C, int H, int Wd, int OH, int OW, int k,
                      int stride, int pt, int pl, int strips) {
    const Dw4Grid g = dw4_grid(OW);
    const int rows = (64 >> g.tpr_log2) * segx::DWG_TY, ntiles = g.tiles_x * ((OH + rows - 1) / rows);
    const int64_t nwork = (int64_t)B * C * strips;
    const int64_t nblocks = (nwork + 3) / 4;
#define SEGX_DW4(KK, SS, PP)                                                                                                        \
    if (k == KK && stride == SS && pl == PP) {                                                                                       \
        hipLaunchKernelGGL((segx::dwconv_wgrad4_kernel<KK, SS, PP>), dim3((unsigned)nblocks), dim3(256), 0, stream, dY, X, part, C,  \
                           H, Wd, OH, OW, pt, g.tiles_x, ntiles, g.tpr_log2, strips, nwork);                                         \
        return true;                                                                                                                 \
    }
    SEGX_DW4(3, 1, 1) SEGX_DW4(5, 1, 2) SEGX_DW4(3, 2, 0) SEGX_DW4(3, 2, 1) SEGX_DW4(5, 2, 1) SEGX_DW4(5, 2, 2)
#undef SEGX_DW4
    return false;
}
}  // namespace

extern "C" int segx_dwconv2d_fwd(const float* X, const float* W, float* Y, int B, int C, int H, int Wd, int OH, int OW, int k, int stride,
                                 int pad_t, int pad_l, void* stream_) {
    SEGX_STREAM; SEGX_REQUIRE(X && W && Y && B > 0 && C > 0 && H > 0 && Wd > 0 && OH > 0 && OW > 0 && (int64_t)B * C <= 65535, "segx_dwconv2d_fwd: bad args");
    if (dw4_ok(X, Y, Wd, OW) && dw4_rows_launch<false>(stream, X, W, Y, B * C, C, H, Wd, OH, OW, k, stride, pad_t, pad_l))
        return check_launch("segx_dwconv2d_fwd");
    const DwGrid g = dw_grid(OH, OW);
    dim3 grid(g.tiles_x * g.tiles_y, B * C);
    SEGX_DW_ROWS_DISPATCH(false, X, W, Y, C, H, Wd, OH, OW, pad_t, pad_l, g.tiles_x, g.txw_log2);
    return check_launch("segx_dwconv2d_fwd");
}
extern "C" int segx_dwconv2d_bwd_data(const float* dY, const float* W, float* dX, int B, int C, int H, int Wd, int OH, int OW, int k, int stride,
                                      int pad_t, int pad_l, void* stream_) {
    SEGX_STREAM; SEGX_REQUIRE(dY && W && dX && B > 0 && C > 0 && H > 0 && Wd > 0 && OH > 0 && OW > 0 && (int64_t)B * C <= 65535, "segx_dwconv2d_bwd_data: bad args");
    if (stride == 1) {      // correlation of dy with the rotated filter: "input" dy [OH,OW], "output" dx [H,W], pads K-1-p
        if (dw4_ok(dY, dX, OW, Wd) && dw4_rows_launch<true>(stream, dY, W, dX, B * C, C, OH, OW, H, Wd, k, 1, k - 1 - pad_t, k - 1 - pad_l))
            return check_launch("segx_dwconv2d_bwd_data");
        const DwGrid g = dw_grid(H, Wd);
        dim3 grid(g.tiles_x * g.tiles_y, B * C);
        SEGX_DW_ROWS_DISPATCH(true, dY, W, dX, C, OH, OW, H, Wd, k - 1 - pad_t, k - 1 - pad_l, g.tiles_x, g.txw_log2);
        return check_launch("segx_dwconv2d_bwd_data");
    }
    if (stride == 2 && Wd % 8 == 0 && OW % 4 == 0 && pad_t == pad_l && ((reinterpret_cast<uintptr_t>(dY) | reinterpret_cast<uintptr_t>(dX)) & 15) == 0) {
        int l = 3;
        while (l < 6 && (8 << l) < Wd) ++l;                                  // threads per dx row: power of two >= W/8 in [8, 64]
        const int tiles_x = (Wd + (8 << l) - 1) / (8 << l), tiles_y = (H + segx::DWB_TY - 1) / segx::DWB_TY;
        const int64_t ngroups = (int64_t)B * C * tiles_x * tiles_y, nblocks = ((ngroups << l) + 255) / 256;
#define SEGX_DWB(KK, PP)                                                                                                             \
        if (k == KK && pad_l == PP && nblocks <= 2147483647LL) {                                                                      \
            hipLaunchKernelGGL((segx::dwconv_bwd4s2_kernel<KK, PP, PP>), dim3((unsigned)nblocks), dim3(256), 0, stream, dY, W, dX, C, \
                               H, Wd, OH, OW, tiles_x, tiles_y, l, ngroups);                                                          \
            return check_launch("segx_dwconv2d_bwd_data");                                                                            \
        }
        SEGX_DWB(3, 0) SEGX_DWB(3, 1) SEGX_DWB(5, 1) SEGX_DWB(5, 2)
#undef SEGX_DWB
    }
    const int tx = (Wd + 15) / 16, ty = (H + 15) / 16;
    dim3 grid(tx * ty, B * C);
    SEGX_DW_DISPATCH(dwconv_bwd_data_kernel, dY, W, dX, C, H, Wd, OH, OW, pad_t, pad_l, tx);
    return check_launch("segx_dwconv2d_bwd_data");
}
/* weight gradient WITHOUT the partial tensor where one workgroup per channel can do it (returns 1 and writes dW [C][k*k]); returns 0 when the shape
 * needs the two-stage form (segx_dwconv2d_bwd_weight + segx_colsum), < 0 on error */
extern "C" int segx_dwconv2d_bwd_weight_direct(const float* dY, const float* X, float* dW, int B, int C, int H, int Wd, int OH, int OW,
                                               int k, int stride, int pad_t, int pad_l, void* stream_) {
    SEGX_STREAM; SEGX_REQUIRE(dY && X && dW && B > 0 && C > 0 && OH > 0 && OW > 0, "segx_dwconv2d_bwd_weight_direct: bad args");
    if (B > 8 || dw_wgrad_strips(OH, OW) != 1 || !dw4_ok(dY, X, Wd, OW)) return 0;
    const Dw4Grid g = dw4_grid(OW);
    const int rows = (64 >> g.tpr_log2) * segx::DWG_TY, ntiles = g.tiles_x * ((OH + rows - 1) / rows);
#define SEGX_DW4C(KK, SS, PP)                                                                                                       \
    if (k == KK && stride == SS && pad_l == PP) {                                                                                    \
        hipLaunchKernelGGL((segx::dwconv_wgrad4c_kernel<KK, SS, PP>), dim3(C), dim3(512), 0, stream, dY, X, dW, B, C, H, Wd, OH, OW,  \
                           pad_t, g.tiles_x, ntiles, g.tpr_log2);                                                                     \
        const int rc = check_launch("segx_dwconv2d_bwd_weight_direct");                                                               \
        return rc ? (rc > 0 ? -rc - 1000 : rc) : 1;                                                                                  \
    }
    SEGX_DW4C(3, 1, 1) SEGX_DW4C(5, 1, 2) SEGX_DW4C(3, 2, 0) SEGX_DW4C(3, 2, 1) SEGX_DW4C(5, 2, 1) SEGX_DW4C(5, 2, 2)
#undef SEGX_DW4C
    return 0;
}
/* r06: data AND weight gradient of a stride-1 'same' depthwise convolution in one pass over dY (dwconv_bwd_s1_fused_kernel).  _rows: rows of `part` per sample
 * (one per wave of a plane), 0 where the shape is not served (stride 2, rows that are not float4 multiples, planes smaller than a wave's share, k other than 3 / 5) */
static int dw_fused_strips(int H, int Wd, int OH, int OW, int k, int stride, int pad_t, int pad_l) {
    if (stride != 1 || (k != 3 && k != 5) || pad_t != (k - 1) / 2 || pad_l != (k - 1) / 2 || OH != H || OW != Wd || Wd % 4 != 0 || Wd < 4) return 0;
    const Dw4Grid g = dw4_grid(Wd);
    const int tiles_y = (H + segx::DW4_TY - 1) / segx::DW4_TY, gpp = g.tiles_x * tiles_y, gpw = 64 >> g.tpr_log2;
    return gpp % gpw == 0 ? gpp / gpw : 0;
}
extern "C" int64_t segx_dwconv2d_bwd_fused_rows(int H, int Wd, int OH, int OW, int k, int stride, int pad_t, int pad_l) {
    return dw_fused_strips(H, Wd, OH, OW, k, stride, pad_t, pad_l);
}
extern "C" int segx_dwconv2d_bwd_fused(const float* dY, const float* X, const float* W, float* dX, float* part /* [B * rows][C][k * k] */, int B, int C, int H, int Wd,
                                       int OH, int OW, int k, int stride, int pad_t, int pad_l, void* stream_) {
    SEGX_STREAM; SEGX_REQUIRE(dY && X && W && dX && part && B > 0 && C > 0 && (int64_t)B * C <= 65535, "segx_dwconv2d_bwd_fused: bad args");
    const int strips = dw_fused_strips(H, Wd, OH, OW, k, stride, pad_t, pad_l);
    SEGX_REQUIRE(strips > 0 && ((reinterpret_cast<uintptr_t>(dY) | reinterpret_cast<uintptr_t>(X) | reinterpret_cast<uintptr_t>(dX)) & 15) == 0,
                 "segx_dwconv2d_bwd_fused: shape not served (segx_dwconv2d_bwd_fused_rows == 0) or unaligned tensors");
    const Dw4Grid g = dw4_grid(Wd);
    const int tiles_y = (H + segx::DW4_TY - 1) / segx::DW4_TY;
    const int64_t ngroups = (int64_t)B * C * g.tiles_x * tiles_y, nblocks = ((ngroups << g.tpr_log2) + 255) / 256;
    SEGX_REQUIRE(nblocks <= 2147483647LL, "segx_dwconv2d_bwd_fused: too many workgroups");
    if (k == 3) hipLaunchKernelGGL((segx::dwconv_bwd_s1_fused_kernel<3>), dim3((unsigned)nblocks), dim3(256), 0, stream, dY, X, W, dX, part, C, H, Wd, g.tiles_x, tiles_y, g.tpr_log2, ngroups, strips);
    else hipLaunchKernelGGL((segx::dwconv_bwd_s1_fused_kernel<5>), dim3((unsigned)nblocks), dim3(256), 0, stream, dY, X, W, dX, part, C, H, Wd, g.tiles_x, tiles_y, g.tpr_log2, ngroups, strips);
    return check_launch("segx_dwconv2d_bwd_fused");
}
/* rows of `part` per sample (see segx_dwconv2d_bwd_weight) */
extern "C" int64_t segx_dwconv2d_wgrad_rows(int OH, int OW) { return OH > 0 && OW > 0 ? dw_wgrad_strips(OH, OW) : 0; }
extern "C" int segx_dwconv2d_bwd_weight(const float* dY, const float* X, float* part /* [B*rows][C][k*k] */, int B, int C, int H, int Wd, int OH, int OW,
                                        int k, int stride, int pad_t, int pad_l, void* stream_) {
    SEGX_STREAM; SEGX_REQUIRE(dY && X && part && B > 0 && C > 0 && OH > 0 && OW > 0 && (int64_t)B * C <= 65535, "segx_dwconv2d_bwd_weight: bad args");
    const int strips = dw_wgrad_strips(OH, OW);
    if (dw4_ok(dY, X, Wd, OW) && dw4_wgrad_launch(stream, dY, X, part, B, C, H, Wd, OH, OW, k, stride, pad_t, pad_l, strips))
        return check_launch("segx_dwconv2d_bwd_weight");
    const DwGrid g = dw_grid(OH, OW);
    dim3 grid(strips, B * C);
    SEGX_DW_DISPATCH(dwconv_wgrad_rows_kernel, dY, X, part, C, H, Wd, OH, OW, pad_t, pad_l, g.tiles_x, g.tiles_x * g.tiles_y, g.txw_log2);
    return check_launch("segx_dwconv2d_bwd_weight");
}
extern "C" int segx_plane_scale(const float* X, const float* gate, float* Y, int64_t planes, int64_t S, void* stream_) {
    SEGX_STREAM; SEGX_REQUIRE(X && gate && Y && planes > 0 && S > 0 && planes <= 65535, "segx_plane_scale: bad args");
    hipLaunchKernelGGL(plane_scale_kernel, dim3(plane_chunks(S, 8), (unsigned)planes), dim3(256), 0, stream, X, gate, Y, S);
    return check_launch("segx_plane_scale");
}
extern "C" int segx_plane_bias_add(const float* X, const float* bias, float* Y, int64_t planes, int C, int64_t S, void* stream_) {
    SEGX_STREAM; SEGX_REQUIRE(X && bias && Y && planes > 0 && C > 0 && S > 0 && planes <= 65535, "segx_plane_bias_add: bad args");
    hipLaunchKernelGGL(plane_bias_add_kernel, dim3(plane_chunks(S, 8), (unsigned)planes), dim3(256), 0, stream, X, bias, Y, C, S);
    return check_launch("segx_plane_bias_add");
}
extern "C" int segx_plane_dot(const float* A, const float* Bm, float* out, int64_t planes, int64_t S, void* stream_) {
    SEGX_STREAM; SEGX_REQUIRE(A && Bm && out && planes > 0 && S > 0 && planes < 2147483647LL, "segx_plane_dot: bad args");
    hipLaunchKernelGGL(plane_dot_kernel, dim3((unsigned)planes), dim3(256), 0, stream, A, Bm, out, S);
    return check_launch("segx_plane_dot");
}
