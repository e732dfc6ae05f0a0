// conv3d.hip -- Inception-I3D spatial convolutions (K20, aj_i3d.py:57-62,92) as IMPLICIT GEMM on the fp32 MFMA tile
// engine of gemm_core.h, plus the TF-'same' zero-padded max-pool (aj_i3d.py:6-30).
//
//   forward      Y[b][co][p]      = sum_{k=(ci,kd,kh,kw)} W[co][k] * X[b][ci][pos(p) + tap(k)]         M=Cout, N=P,       K=Cin*KV
//   backward-data (stride 1)      = the same kernel on (dY, W flipped & transposed), pad' = K-1-pad
//   backward-weight dW[co][n=(ci,tap)] = sum_{b,p} dY[b][co][p] * X[b][ci][pos(p) + tap]                M=Cout, N=Cin*KV, K=P (split-K, per-sample slabs)
//
// The im2col matrix is never materialised: the B-operand loader gathers straight from the NCDHW activation with
// zero padding (dynamic 'same' padding N7: front = pad // 2) — unconditional loads from clamped addresses plus a
// validity mask applied at LDS-store time, exactly like the dense loader.  The A operand (weights, or dY) is dense.
#include "gemm_core.h"

namespace segx {

struct ConvGeom {
    int Cin, ID, IH, IW, OD, OH, OW, KD, KH, KW, sd, sh, sw, pd, ph, pw;
};

// ---- forward / backward-data: B(n = output position, k = (ci, tap)), positions contiguous ---------------------
// thread map (row-contiguous operand): piece i -> k-row (tid>>5) + 8*i, positions n0 + 4*(tid&31) + j
struct ConvFwdLoaderB {
    static constexpr bool kc = false;
    const float* X; ConvGeom q; int K;
    int bd[4], bh[4], bw[4]; unsigned nvalid;
    __device__ __forceinline__ ConvFwdLoaderB(const float* X_, const ConvGeom& q_, int n0, int P, int K_) : X(X_), q(q_), K(K_) {
        nvalid = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = n0 + ((threadIdx.x & 31) << 2) + j;
            const int nn = n < P ? n : 0;
            const int od = nn / (q.OH * q.OW), r = nn - od * q.OH * q.OW, oh = r / q.OW, ow = r - oh * q.OW;
            bd[j] = od * q.sd - q.pd; bh[j] = oh * q.sh - q.ph; bw[j] = ow * q.sw - q.pw;
            if (n < P) nvalid |= 1u << j;
        }
    }
    __device__ __forceinline__ unsigned load(float4 (&r)[4], int k0, int kend, int tid) const {
        unsigned okmask = 0;
        const int KV = q.KD * q.KH * q.KW, KHW = q.KH * q.KW;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int k = k0 + (tid >> 5) + 8 * i;
            const bool kok = k < kend;
            const int kk = kok ? k : 0;
            const int ci = kk / KV, t = kk - ci * KV, kd = t / KHW, t2 = t - kd * KHW, kh = t2 / q.KW, kw = t2 - kh * q.KW;
            const int64_t cbase = (int64_t)ci * q.ID * q.IH * q.IW;
            float v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int id = bd[j] + kd, ih = bh[j] + kh, iw = bw[j] + kw;
                const bool ok = kok && ((nvalid >> j) & 1u) && id >= 0 && id < q.ID && ih >= 0 && ih < q.IH && iw >= 0 && iw < q.IW;
                const int64_t off = ok ? cbase + ((int64_t)id * q.IH + ih) * q.IW + iw : 0;
                v[j] = X[off];
                if (ok) okmask |= 1u << (4 * i + j);
            }
            r[i] = make_float4(v[0], v[1], v[2], v[3]);
        }
        return okmask;
    }
};

// ---- backward-weight: B(n = (ci, tap), k = output position) ------------------------------------------------------
// thread map (k-contiguous operand): piece i -> row n0 + (tid>>3) + 32*i, k-chunk 4*(tid&7) + j
struct ConvWgradLoaderB {
    static constexpr bool kc = true;
    const float* X; ConvGeom q; int P;
    int64_t cbase[4]; int kd[4], kh[4], kw[4]; unsigned rvalid;
    __device__ __forceinline__ ConvWgradLoaderB(const float* X_, const ConvGeom& q_, int n0, int N, int P_) : X(X_), q(q_), P(P_) {
        const int KV = q.KD * q.KH * q.KW, KHW = q.KH * q.KW;
        rvalid = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int n = n0 + (threadIdx.x >> 3) + 32 * i;
            const int nn = n < N ? n : 0;
            const int ci = nn / KV, t = nn - ci * KV;
            kd[i] = t / KHW; const int t2 = t - kd[i] * KHW; kh[i] = t2 / q.KW; kw[i] = t2 - kh[i] * q.KW;
            cbase[i] = (int64_t)ci * q.ID * q.IH * q.IW;
            if (n < N) rvalid |= 1u << i;
        }
    }
    __device__ __forceinline__ unsigned load(float4 (&r)[4], int k0, int kend, int tid) const {
        unsigned okmask = 0;
        int od[4], oh[4], ow[4]; bool pok[4];
        const int p0 = k0 + ((tid & 7) << 2);
        {   // decode the first position once, then carry
            const int pp = p0 < P ? p0 : 0;
            int d = pp / (q.OH * q.OW), rr = pp - d * q.OH * q.OW, h = rr / q.OW, w = rr - h * q.OW;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                od[j] = d; oh[j] = h; ow[j] = w; pok[j] = (p0 + j) < kend;
                if (++w == q.OW) { w = 0; if (++h == q.OH) { h = 0; ++d; } }
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int id = od[j] * q.sd - q.pd + kd[i], ih = oh[j] * q.sh - q.ph + kh[i], iw = ow[j] * q.sw - q.pw + kw[i];
                const bool ok = pok[j] && ((rvalid >> i) & 1u) && id >= 0 && id < q.ID && ih >= 0 && ih < q.IH && iw >= 0 && iw < q.IW;
                const int64_t off = ok ? cbase[i] + ((int64_t)id * q.IH + ih) * q.IW + iw : 0;
                v[j] = X[off];
                if (ok) okmask |= 1u << (4 * i + j);
            }
            r[i] = make_float4(v[0], v[1], v[2], v[3]);
        }
        return okmask;
    }
};

template <bool VEC>
__global__ __launch_bounds__(256) void conv3d_fwd_kernel(GemmArgs g, ConvGeom q) {
    __shared__ __attribute__((aligned(16))) float As[BKT][LDT];
    __shared__ __attribute__((aligned(16))) float Bs[BKT][LDT];
    const TileCoord t = tile_coord(g);
    const DenseLoader<true, VEC> la{g.A, g.a_m, 1, t.m0, g.M};                       // weights [Cout][Cin*KV]
    const ConvFwdLoaderB lb(g.B + (int64_t)t.zb * g.b_b0, q, t.n0, g.N, g.K);        // X[b]
    f32x16 acc[2][2];
    gemm_mainloop(acc, la, lb, t.kbeg, t.kend, As, Bs);
    gemm_epilogue<SEGX_EPI_NONE>(acc, g, t);
}
template <bool VEC>
__global__ __launch_bounds__(256) void conv3d_wgrad_kernel(GemmArgs g, ConvGeom q) {
    __shared__ __attribute__((aligned(16))) float As[BKT][LDT];
    __shared__ __attribute__((aligned(16))) float Bs[BKT][LDT];
    const TileCoord t = tile_coord(g);
    const DenseLoader<true, VEC> la{g.A + (int64_t)t.zb * g.a_b0, g.a_m, 1, t.m0, g.M};   // dY[b] [Cout][P]
    const ConvWgradLoaderB lb(g.B + (int64_t)t.zb * g.b_b0, q, t.n0, g.N, g.K);           // X[b]
    f32x16 acc[2][2];
    gemm_mainloop(acc, la, lb, t.kbeg, t.kend, As, Bs);
    gemm_epilogue<SEGX_EPI_NONE>(acc, g, t);
}

// Wt[ci][co][t] = W[co][ci][KV-1-t]: the transposed, spatially flipped filter bank of the backward-data convolution
__global__ __launch_bounds__(256) void flip_weights_kernel(const float* __restrict__ W, float* __restrict__ Wt, int Cout, int Cin, int KV) {
    const int64_t total = (int64_t)Cout * Cin * KV;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int t = (int)(i % KV); const int64_t r = i / KV; const int co = (int)(r % Cout), ci = (int)(r / Cout);
        Wt[i] = W[((int64_t)co * Cin + ci) * KV + (KV - 1 - t)];
    }
}

// =================================================================================================
// Max-pool with TF-'same' ZERO padding (aj_i3d.py:28-30 pads with F.pad, i.e. zeros, then pools; inputs are
// post-ReLU so this equals -inf padding, N7).  The scan order and "first maximum wins" rule of ATen are kept so the
// gradient routing is identical; a padded zero that wins receives (and drops) the gradient, as in the reference.
// =================================================================================================
struct PoolGeom { int ID, IH, IW, OD, OH, OW, KD, KH, KW, sd, sh, sw, pd, ph, pw; };

__global__ __launch_bounds__(256) void maxpool3d_fwd_kernel(const float* __restrict__ X, float* __restrict__ Y, int* __restrict__ arg,
                                                            PoolGeom q, int64_t planes) {
    const int64_t osz = (int64_t)q.OD * q.OH * q.OW, isz = (int64_t)q.ID * q.IH * q.IW, total = planes * osz;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        const int64_t p = idx / osz; int64_t r = idx - p * osz;
        const int od = (int)(r / ((int64_t)q.OH * q.OW)); r -= (int64_t)od * q.OH * q.OW;
        const int oh = (int)(r / q.OW), ow = (int)(r - (int64_t)oh * q.OW);
        const float* x = X + p * isz;
        float best = -INFINITY; int bi = -1;
        for (int kd = 0; kd < q.KD; ++kd) for (int kh = 0; kh < q.KH; ++kh) for (int kw = 0; kw < q.KW; ++kw) {
            const int id = od * q.sd - q.pd + kd, ih = oh * q.sh - q.ph + kh, iw = ow * q.sw - q.pw + kw;
            const bool in = id >= 0 && id < q.ID && ih >= 0 && ih < q.IH && iw >= 0 && iw < q.IW;
            const int li = (id * q.IH + ih) * q.IW + iw;
            const float v = in ? x[li] : 0.f;                      // zero padding
            if (v > best || v != v) { best = v; bi = in ? li : -1; }
        }
        Y[idx] = best; arg[idx] = bi;
    }
}
// gather form: an input cell sums the gradients of the windows whose arg-max it is
__global__ __launch_bounds__(256) void maxpool3d_bwd_kernel(const float* __restrict__ dY, const int* __restrict__ arg, float* __restrict__ dX,
                                                            PoolGeom q, int64_t planes) {
    const int64_t osz = (int64_t)q.OD * q.OH * q.OW, isz = (int64_t)q.ID * q.IH * q.IW, total = planes * isz;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        const int64_t p = idx / isz; const int li = (int)(idx - p * isz);
        const int id = li / (q.IH * q.IW), r = li - id * q.IH * q.IW, ih = r / q.IW, iw = r - ih * q.IW;
        const float* g = dY + p * osz; const int* a = arg + p * osz;
        float acc = 0.f;
        // windows covering (id,ih,iw): od in [ceil((id+pd-KD+1)/sd), floor((id+pd)/sd)]
        const int d1 = (id + q.pd) / q.sd, h1 = (ih + q.ph) / q.sh, w1 = (iw + q.pw) / q.sw;
        const int dn = id + q.pd - q.KD + 1, hn = ih + q.ph - q.KH + 1, wn = iw + q.pw - q.KW + 1;
        const int d0 = dn > 0 ? (dn + q.sd - 1) / q.sd : 0, h0 = hn > 0 ? (hn + q.sh - 1) / q.sh : 0, w0 = wn > 0 ? (wn + q.sw - 1) / q.sw : 0;
        for (int od = d0; od <= d1 && od < q.OD; ++od) for (int oh = h0; oh <= h1 && oh < q.OH; ++oh) for (int ow = w0; ow <= w1 && ow < q.OW; ++ow) {
            const int64_t o = ((int64_t)od * q.OH + oh) * q.OW + ow;
            if (a[o] == li) acc += g[o];
        }
        dX[idx] = acc;
    }
}

static bool aligned16c(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace segx

using namespace segx;
#define SEGX_STREAM hipStream_t stream = (hipStream_t)stream_

static ConvGeom make_geom(const int* g) {
    ConvGeom q; q.Cin = g[0]; q.ID = g[1]; q.IH = g[2]; q.IW = g[3]; q.OD = g[4]; q.OH = g[5]; q.OW = g[6];
    q.KD = g[7]; q.KH = g[8]; q.KW = g[9]; q.sd = g[10]; q.sh = g[11]; q.sw = g[12]; q.pd = g[13]; q.ph = g[14]; q.pw = g[15];
    return q;
}
static void fill_common(GemmArgs& g, int M, int N, int K, int nbatch, int splitk, float* workspace) {
    g.bias = nullptr; g.aux = nullptr; g.gmax = nullptr; g.nb1 = 1; g.bias_b1 = 0; g.alpha = 1.0f; g.epilogue = SEGX_EPI_NONE; g.bias_mode = SEGX_BIAS_NONE;
    g.a_b1 = g.b_b1 = g.c_b1 = 0; g.b_n = g.b_k = 0; g.a_k = 1; g.vecA = g.vecB = 0;
    g.M = M; g.N = N; g.K = K; g.tiles_m = ceil_div(M, BM); g.tiles_n = ceil_div(N, BN);
    g.dropout_p = 0.f; g.seed = g.offset = 0; g.splitk = splitk;
    g.k_chunk = splitk == 1 ? K : ceil_div(ceil_div(K, splitk), BKT) * BKT;
    g.c_split = (int64_t)nbatch * M * N;
    if (splitk > 1) g.C = workspace;
}

/* geom = {Cin, ID, IH, IW, OD, OH, OW, KD, KH, KW, sd, sh, sw, pd, ph, pw} (front pads) */
extern "C" int segx_conv3d_fwd(const float* X, const float* W, float* Y, int B, int Cout, const int* geom, void* stream_) {
    SEGX_STREAM; SEGX_REQUIRE(X && W && Y && geom && B > 0 && Cout > 0 && B <= 65535, "segx_conv3d_fwd: bad args");
    const ConvGeom q = make_geom(geom);
    const int64_t P = (int64_t)q.OD * q.OH * q.OW; const int K = q.Cin * q.KD * q.KH * q.KW;
    SEGX_REQUIRE(P > 0 && P < 2147483647LL && K > 0, "segx_conv3d_fwd: bad geometry");
    GemmArgs g; g.A = W; g.B = X; g.C = Y;
    g.a_b0 = 0; g.a_m = K; g.b_b0 = (int64_t)q.Cin * q.ID * q.IH * q.IW; g.c_b0 = (int64_t)Cout * P; g.c_m = P;
    fill_common(g, Cout, (int)P, K, B, 1, nullptr);
    dim3 grid(g.tiles_m * g.tiles_n, B, 1);
    if (aligned16c(W) && K % 4 == 0) hipLaunchKernelGGL((conv3d_fwd_kernel<true>), grid, dim3(256), 0, stream, g, q);
    else hipLaunchKernelGGL((conv3d_fwd_kernel<false>), grid, dim3(256), 0, stream, g, q);
    return check_launch("segx_conv3d_fwd");
}
extern "C" int segx_conv3d_flip_weights(const float* W, float* Wt, int Cout, int Cin, int KV, void* stream_) {
    SEGX_STREAM; SEGX_REQUIRE(W && Wt && Cout > 0 && Cin > 0 && KV > 0, "segx_conv3d_flip_weights: bad args");
    const int64_t total = (int64_t)Cout * Cin * KV;
    hipLaunchKernelGGL(flip_weights_kernel, dim3((unsigned)i64min(4096, (total + 255) / 256)), dim3(256), 0, stream, W, Wt, Cout, Cin, KV);
    return check_launch("segx_conv3d_flip_weights");
}
/* dWb[b][Cout][Cin*KV] per-sample weight gradients (sum over b with segx_colsum); workspace: splitk*B*Cout*Cin*KV floats when splitk > 1 */
extern "C" int segx_conv3d_bwd_weight(const float* dY, const float* X, float* dWb, int B, int Cout, const int* geom, int splitk,
                                      float* workspace, void* stream_) {
    SEGX_STREAM; SEGX_REQUIRE(dY && X && dWb && geom && B > 0 && Cout > 0 && B <= 65535, "segx_conv3d_bwd_weight: bad args");
    const ConvGeom q = make_geom(geom);
    const int64_t P = (int64_t)q.OD * q.OH * q.OW; const int N = q.Cin * q.KD * q.KH * q.KW;
    SEGX_REQUIRE(P > 0 && P < 2147483647LL && N > 0, "segx_conv3d_bwd_weight: bad geometry");
    if (splitk < 1) splitk = 1;
    SEGX_REQUIRE(splitk == 1 || workspace, "segx_conv3d_bwd_weight: split-K needs a workspace");
    GemmArgs g; g.A = dY; g.B = X; g.C = dWb;
    g.a_b0 = (int64_t)Cout * P; g.a_m = P; g.b_b0 = (int64_t)q.Cin * q.ID * q.IH * q.IW; g.c_b0 = (int64_t)Cout * N; g.c_m = N;
    fill_common(g, Cout, N, (int)P, B, splitk, workspace);
    dim3 grid(g.tiles_m * g.tiles_n, B, splitk);
    if (aligned16c(dY) && P % 4 == 0) hipLaunchKernelGGL((conv3d_wgrad_kernel<true>), grid, dim3(256), 0, stream, g, q);
    else hipLaunchKernelGGL((conv3d_wgrad_kernel<false>), grid, dim3(256), 0, stream, g, q);
    int rc = check_launch("segx_conv3d_bwd_weight");
    if (rc || splitk == 1) return rc;
    const int64_t total = g.c_split;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)i64min(2048, (total + 255) / 256)), dim3(256), 0, stream, (const float*)workspace, dWb,
                       (const float*)nullptr, Cout, N, 1, splitk, g.c_split, (int64_t)Cout * N, (int64_t)0, (int64_t)N, 1.0f, (int)SEGX_BIAS_NONE,
                       (int64_t)0, total);
    return check_launch("segx_conv3d_bwd_weight/reduce");
}

static PoolGeom make_pool(const int* g) {
    PoolGeom q; q.ID = g[0]; q.IH = g[1]; q.IW = g[2]; q.OD = g[3]; q.OH = g[4]; q.OW = g[5]; q.KD = g[6]; q.KH = g[7]; q.KW = g[8];
    q.sd = g[9]; q.sh = g[10]; q.sw = g[11]; q.pd = g[12]; q.ph = g[13]; q.pw = g[14];
    return q;
}
/* geom = {ID, IH, IW, OD, OH, OW, KD, KH, KW, sd, sh, sw, pd, ph, pw}; arg = int32 arg-max index per output (or -1 = padding) */
extern "C" int segx_maxpool3d_fwd(const float* X, float* Y, int* arg, int64_t planes, const int* geom, void* stream_) {
    SEGX_STREAM; SEGX_REQUIRE(X && Y && arg && geom && planes > 0, "segx_maxpool3d_fwd: bad args");
    const PoolGeom q = make_pool(geom);
    const int64_t total = planes * q.OD * q.OH * q.OW;
    hipLaunchKernelGGL(maxpool3d_fwd_kernel, dim3((unsigned)i64min(65536, (total + 255) / 256)), dim3(256), 0, stream, X, Y, arg, q, planes);
    return check_launch("segx_maxpool3d_fwd");
}
extern "C" int segx_maxpool3d_bwd(const float* dY, const int* arg, float* dX, int64_t planes, const int* geom, void* stream_) {
    SEGX_STREAM; SEGX_REQUIRE(dY && arg && dX && geom && planes > 0, "segx_maxpool3d_bwd: bad args");
    const PoolGeom q = make_pool(geom);
    const int64_t total = planes * q.ID * q.IH * q.IW;
    hipLaunchKernelGGL(maxpool3d_bwd_kernel, dim3((unsigned)i64min(65536, (total + 255) / 256)), dim3(256), 0, stream, dY, arg, dX, q, planes);
    return check_launch("segx_maxpool3d_bwd");
}
