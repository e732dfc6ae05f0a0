// train.hip -- train-step glue kernels: fused BCE(pos_weight)+Dice loss (fwd+bwd, K21) and the
// multi-tensor two-level-clip BertAdam (K22).  All HBM-bound: one read of every input element,
// one write of every output element, deterministic two-stage reductions (no float atomics).
#include "common.h"

namespace segx {

// =================================================================================================
// Loss: 0.5 * BCEWithLogits(pos_weight) + 0.5 * sum_c w_c * mean_b Dice_bc     (train2d.py:1219-1242,1314-1318)
// logits / mask: [B, C, S] (S = flattened spatial, contiguous).  Stage 1 partial sums per (b, c, slab):
//   [0] sum_s bce   [1] sum_s sig*y   [2] sum_s sig^2   [3] sum_s y^2
// =================================================================================================
constexpr int LOSS_SLABS = 64;

__device__ __forceinline__ float softplus(float x) { return fmaxf(x, 0.f) + log1pf(__expf(-fabsf(x))); }
__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + __expf(-x)); }

__global__ __launch_bounds__(256) void loss_stage1(const float* __restrict__ logits, const float* __restrict__ mask,
                                                   const float* __restrict__ pos_weight, float* __restrict__ ws, int C, int64_t S) {
    __shared__ float red[4];
    const int bc = blockIdx.y, slab = blockIdx.x, c = bc % C;
    const float pw = pos_weight[c];
    const int64_t per = (S + LOSS_SLABS - 1) / LOSS_SLABS, s0 = slab * per, s1 = i64min(S, s0 + per);
    const float* x = logits + (int64_t)bc * S; const float* y = mask + (int64_t)bc * S;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    for (int64_t s = s0 + threadIdx.x; s < s1; s += 256) {
        const float xv = x[s], yv = y[s], sg = sigmoidf_(xv);
        a0 += pw * yv * softplus(-xv) + (1.0f - yv) * softplus(xv);
        a1 += sg * yv; a2 += sg * sg; a3 += yv * yv;
    }
    a0 = block_sum<4>(a0, red); a1 = block_sum<4>(a1, red); a2 = block_sum<4>(a2, red); a3 = block_sum<4>(a3, red);
    if (threadIdx.x == 0) {
        float* o = ws + ((int64_t)bc * LOSS_SLABS + slab) * 4;
        o[0] = a0; o[1] = a1; o[2] = a2; o[3] = a3;
    }
}
// one block: reduce slabs; out[0]=loss, out[1]=ce, out[2]=dice_total, out[3+c]=dice_c ; coef[bc] = {I, Z+Y+smooth}
__global__ __launch_bounds__(256) void loss_stage2(const float* __restrict__ ws, const float* __restrict__ class_w, float* __restrict__ out,
                                                   float* __restrict__ coef, int B, int C, int64_t S, float dice_w) {
    __shared__ float s_ce[256];
    const int t = threadIdx.x;
    float ce = 0.f;
    for (int bc = t; bc < B * C; bc += 256) {
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        for (int k = 0; k < LOSS_SLABS; ++k) { const float* o = ws + ((int64_t)bc * LOSS_SLABS + k) * 4; a0 += o[0]; a1 += o[1]; a2 += o[2]; a3 += o[3]; }
        ce += a0;
        coef[2 * bc] = a1; coef[2 * bc + 1] = a2 + a3 + 1e-5f;
    }
    s_ce[t] = ce;
    __syncthreads();
    if (t == 0) {
        float tot = 0.f;
        for (int i = 0; i < 256; ++i) tot += s_ce[i];
        const float cev = tot / ((float)B * (float)C * (float)S);
        float dice_tot = 0.f;
        for (int c = 0; c < C; ++c) {
            float d = 0.f;
            for (int b = 0; b < B; ++b) { const int bc = b * C + c; d += 1.0f - (2.0f * coef[2 * bc] + 1e-5f) / coef[2 * bc + 1]; }
            d /= (float)B;
            out[3 + c] = d;
            dice_tot += d * class_w[c];
        }
        out[0] = (1.0f - dice_w) * cev + dice_w * dice_tot; out[1] = cev; out[2] = dice_tot;
    }
}
__global__ __launch_bounds__(256) void loss_bwd(const float* __restrict__ logits, const float* __restrict__ mask, const float* __restrict__ pos_weight,
                                                const float* __restrict__ class_w, const float* __restrict__ coef, const float* __restrict__ gout,
                                                float* __restrict__ dlogits, int B, int C, int64_t S, float dice_w) {
    const int bc = blockIdx.y, c = bc % C;
    const float pw = pos_weight[c], cw = class_w[c], g = gout[0];
    const float I = coef[2 * bc], den = coef[2 * bc + 1];
    const float k_ce = g * (1.0f - dice_w) / ((float)B * (float)C * (float)S);
    const float k_d = g * dice_w * cw / (float)B;
    const float num = 2.0f * I + 1e-5f, inv_den2 = 1.0f / (den * den);
    const float* x = logits + (int64_t)bc * S; const float* y = mask + (int64_t)bc * S; float* d = dlogits + (int64_t)bc * S;
    for (int64_t s = (int64_t)blockIdx.x * 256 + threadIdx.x; s < S; s += (int64_t)gridDim.x * 256) {
        const float xv = x[s], yv = y[s], sg = sigmoidf_(xv), dsg = sg * (1.0f - sg);
        const float dce = (1.0f - yv) * sg - pw * yv * (1.0f - sg);
        const float ddice = -(2.0f * yv * dsg * den - num * 2.0f * sg * dsg) * inv_den2;
        d[s] = k_ce * dce + k_d * ddice;
    }
}

// =================================================================================================
// Multi-tensor BertAdam (optimization.py:90-164) preceded by the trainer's global clip (train2d.py:1324-1325).
// Tensors are cut into fixed chunks; tables live on the device (built once): chunk -> (tensor, offset).
//   stage A: per-chunk sum of squares          stage B: per-tensor norms, global norm, clip coefficients
//   stage C: m,v,p update with  g' = g * coef[tensor]
// =================================================================================================
struct MtTables {
    float* const* params; const float* const* grads; float* const* m; float* const* v;
    const int64_t* sizes; const int* chunk_tensor; const int64_t* chunk_off;
};

__global__ __launch_bounds__(256) void mt_sumsq_kernel(MtTables T, int chunk, float* __restrict__ chunk_ws) {
    __shared__ float red[4];
    const int ch = blockIdx.x, t = T.chunk_tensor[ch];
    const int64_t off = T.chunk_off[ch], n = i64min(chunk, T.sizes[t] - off);
    const float* gbase = T.grads[t];                       // NULL: the parameter has no gradient (inactive, N3)
    float s = 0.f;
    if (gbase) {
        const float* g = gbase + off;
        int64_t i0 = 0;
        if ((reinterpret_cast<uintptr_t>(g) & 15) == 0) {      // 16-byte loads (a chunk starts on a multiple of `chunk` floats of an allocation; gradient VIEWS may not)
            const int64_t n4 = n >> 2;
            for (int64_t i = threadIdx.x; i < n4; i += 256) { const float4 x = reinterpret_cast<const float4*>(g)[i]; s += (x.x * x.x + x.y * x.y) + (x.z * x.z + x.w * x.w); }
            i0 = n4 << 2;
        }
        for (int64_t i = i0 + threadIdx.x; i < n; i += 256) { const float x = g[i]; s += x * x; }
    }
    s = block_sum<4>(s, red);
    if (threadIdx.x == 0) chunk_ws[ch] = s;
}
// single block; chunks of one tensor are consecutive.  coef[t] = global_coef * per_tensor_coef ; stats = {global norm, global coef}
__global__ __launch_bounds__(256) void mt_clipcoef_kernel(const float* __restrict__ chunk_ws, const int* __restrict__ chunk_first /* [nt+1] */,
                                                          const int* __restrict__ active, int nt, float max_global, float max_tensor,
                                                          float* __restrict__ tnorm, float* __restrict__ coef, float* __restrict__ stats) {
    __shared__ float red[4];
    __shared__ float s_g;
    float part = 0.f;
    for (int t = threadIdx.x; t < nt; t += 256) {
        float s = 0.f;
        for (int c = chunk_first[t]; c < chunk_first[t + 1]; ++c) s += chunk_ws[c];
        const float nrm = sqrtf(s);
        tnorm[t] = nrm;
        if (active[t]) part += s;                       // clip_grad_norm_ sees only parameters that HAVE a gradient (N3)
    }
    // norm of per-tensor norms == sqrt of the total sum of squares
    const float tot = block_sum<4>(part, red);
    if (threadIdx.x == 0) {
        const float gn = sqrtf(tot);
        float gc = max_global > 0.f ? max_global / (gn + 1e-6f) : 1.0f;
        gc = fminf(gc, 1.0f);
        s_g = gc; stats[0] = gn; stats[1] = gc;
    }
    __syncthreads();
    const float gc = s_g;
    for (int t = threadIdx.x; t < nt; t += 256) {
        float c = gc;
        if (max_tensor > 0.f) c *= fminf(max_tensor / (gc * tnorm[t] + 1e-6f), 1.0f);
        coef[t] = c;
    }
}
__global__ __launch_bounds__(256) void mt_bertadam_kernel(MtTables T, int chunk, const float* __restrict__ coef, const float* __restrict__ lr,
                                                          const float* __restrict__ wd, const int* __restrict__ active,
                                                          float sched, float b1, float b2, float eps) {
    const int ch = blockIdx.x, t = T.chunk_tensor[ch];
    if (!active[t]) return;                               // `if p.grad is None: continue` (optimization.py:100-101)
    const int64_t off = T.chunk_off[ch], n = i64min(chunk, T.sizes[t] - off);
    float* p = T.params[t] + off; const float* g = T.grads[t] + off; float* m = T.m[t] + off; float* v = T.v[t] + off;
    const float c = coef[t], step = lr[t] * sched, decay = wd[t];
    // one element of the update (optimization.py:104-130); the same arithmetic in the 16-byte and the scalar loop
    auto upd1 = [&](float gi, float& mi, float& vi, float& pi) {
        gi *= c;
        mi = mi * b1 + (1.0f - b1) * gi;
        vi = vi * b2 + (1.0f - b2) * gi * gi;
        float upd = mi / (sqrtf(vi) + eps);
        if (decay > 0.f) upd += decay * pi;
        pi = pi - step * upd;
    };
    int64_t i0 = 0;
    if (((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(m) | reinterpret_cast<uintptr_t>(v)) & 15) == 0) {
        // r04-g: 16-byte accesses (the scalar loop moved 2.33 GB in 571 us = 4.1 TB/s, four 4-byte loads in flight per lane)
        const int64_t n4 = n >> 2;
        float4* p4 = reinterpret_cast<float4*>(p); const float4* g4 = reinterpret_cast<const float4*>(g);
        float4* m4 = reinterpret_cast<float4*>(m); float4* v4 = reinterpret_cast<float4*>(v);
        for (int64_t i = threadIdx.x; i < n4; i += 256) {
            const float4 gv = g4[i]; float4 mv = m4[i], vv = v4[i], pv = p4[i];
            upd1(gv.x, mv.x, vv.x, pv.x); upd1(gv.y, mv.y, vv.y, pv.y); upd1(gv.z, mv.z, vv.z, pv.z); upd1(gv.w, mv.w, vv.w, pv.w);
            p4[i] = pv; m4[i] = mv; v4[i] = vv;
        }
        i0 = n4 << 2;
    }
    for (int64_t i = i0 + threadIdx.x; i < n; i += 256) {
        float mi = m[i], vi = v[i], pi = p[i];
        upd1(g[i], mi, vi, pi);
        p[i] = pi; m[i] = mi; v[i] = vi;
    }
}

}  // namespace segx

using namespace segx;

extern "C" int64_t segx_loss_ws_floats(int B, int C) { return (int64_t)B * C * LOSS_SLABS * 4 + (int64_t)2 * B * C; }
extern "C" int segx_seg_loss_fwd(const float* logits, const float* mask, const float* pos_weight, const float* class_w, float* out /* 3 + C */,
                                 float* ws, int B, int C, int64_t S, float dice_w, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    SEGX_REQUIRE(logits && mask && pos_weight && class_w && out && ws && B > 0 && C > 0 && C <= 64 && S > 0, "segx_seg_loss_fwd: bad args");
    hipLaunchKernelGGL(loss_stage1, dim3(LOSS_SLABS, B * C), dim3(256), 0, stream, logits, mask, pos_weight, ws, C, S);
    float* coef = ws + (int64_t)B * C * LOSS_SLABS * 4;
    hipLaunchKernelGGL(loss_stage2, dim3(1), dim3(256), 0, stream, (const float*)ws, class_w, out, coef, B, C, S, dice_w);
    return check_launch("segx_seg_loss_fwd");
}
extern "C" int segx_seg_loss_bwd(const float* logits, const float* mask, const float* pos_weight, const float* class_w, const float* ws,
                                 const float* grad_out, float* dlogits, int B, int C, int64_t S, float dice_w, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    SEGX_REQUIRE(logits && mask && pos_weight && class_w && ws && grad_out && dlogits && B > 0 && C > 0 && S > 0, "segx_seg_loss_bwd: bad args");
    const float* coef = ws + (int64_t)B * C * LOSS_SLABS * 4;
    const int gx = (int)i64min(1024, (S + 255) / 256);
    hipLaunchKernelGGL(loss_bwd, dim3(gx, B * C), dim3(256), 0, stream, logits, mask, pos_weight, class_w, coef, grad_out, dlogits, B, C, S, dice_w);
    return check_launch("segx_seg_loss_bwd");
}

// Multi-tensor gather: chunks [chunk_begin, chunk_begin + gridDim.x) of the tensors listed in src -> the flat slices listed in dst
// (data parallel: this step's gradient tensors into their all-reduce bucket, one launch per bucket).  src[t] == NULL: the slice is
// left untouched (parameters without a gradient stay zero in the bucket).
__global__ __launch_bounds__(256) void mt_gather_kernel(const float* const* __restrict__ src, float* const* __restrict__ dst,
                                                        const int64_t* __restrict__ sizes, const int* __restrict__ chunk_tensor,
                                                        const int64_t* __restrict__ chunk_off, int chunk_begin, int chunk) {
    const int ch = chunk_begin + blockIdx.x, t = chunk_tensor[ch];
    const int64_t off = chunk_off[ch], n = i64min(chunk, sizes[t] - off);
    const float* s = src[t];
    if (!s) return;
    s += off; float* d = dst[t] + off;
    if (((reinterpret_cast<uintptr_t>(s) | reinterpret_cast<uintptr_t>(d)) & 15) == 0) {
        const int64_t n4 = n >> 2;
        for (int64_t i = threadIdx.x; i < n4; i += 256) reinterpret_cast<float4*>(d)[i] = reinterpret_cast<const float4*>(s)[i];
        for (int64_t i = (n4 << 2) + threadIdx.x; i < n; i += 256) d[i] = s[i];
    } else {
        for (int64_t i = threadIdx.x; i < n; i += 256) d[i] = s[i];
    }
}
extern "C" int segx_mt_gather(const void* const* src, void* const* dst, const int64_t* sizes, const int* chunk_tensor, const int64_t* chunk_off,
                              int chunk_begin, int nchunks, int chunk, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    SEGX_REQUIRE(src && dst && sizes && chunk_tensor && chunk_off && chunk_begin >= 0 && nchunks > 0 && chunk > 0, "segx_mt_gather: bad args");
    hipLaunchKernelGGL(mt_gather_kernel, dim3(nchunks), dim3(256), 0, stream, (const float* const*)src, (float* const*)dst, sizes, chunk_tensor,
                       chunk_off, chunk_begin, chunk);
    return check_launch("segx_mt_gather");
}

extern "C" int segx_mt_bertadam_step(void* const* params, const void* const* grads, void* const* m, void* const* v, const int64_t* sizes,
                                     const int* chunk_tensor, const int64_t* chunk_off, const int* chunk_first, const int* active,
                                     const float* lr, const float* wd, int ntensors, int nchunks, int chunk,
                                     float max_global_norm, float max_tensor_norm, float sched, float b1, float b2, float eps,
                                     float* ws /* nchunks + 2*ntensors + 2 floats */, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    SEGX_REQUIRE(params && grads && m && v && sizes && chunk_tensor && chunk_off && chunk_first && active && lr && wd && ws,
                 "segx_mt_bertadam_step: null table");
    SEGX_REQUIRE(ntensors > 0 && nchunks > 0 && chunk > 0, "segx_mt_bertadam_step: bad sizes");
    MtTables T{(float* const*)params, (const float* const*)grads, (float* const*)m, (float* const*)v, sizes, chunk_tensor, chunk_off};
    float* chunk_ws = ws; float* tnorm = ws + nchunks; float* coef = tnorm + ntensors; float* stats = coef + ntensors;
    hipLaunchKernelGGL(mt_sumsq_kernel, dim3(nchunks), dim3(256), 0, stream, T, chunk, chunk_ws);
    hipLaunchKernelGGL(mt_clipcoef_kernel, dim3(1), dim3(256), 0, stream, (const float*)chunk_ws, chunk_first, active, ntensors,
                       max_global_norm, max_tensor_norm, tnorm, coef, stats);
    hipLaunchKernelGGL(mt_bertadam_kernel, dim3(nchunks), dim3(256), 0, stream, T, chunk, (const float*)coef, lr, wd, active, sched, b1, b2, eps);
    return check_launch("segx_mt_bertadam_step");
}
