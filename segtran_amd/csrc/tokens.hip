// tokens.hip -- HBM-bound row kernels of the Squeeze-and-Expansion transformer (fwd + bwd).
//
// Token tensors are [rows, C] fp32 with C in {1792, 896, 448, 1024, ...}.  Every kernel here is
// bandwidth-bound, so the design rule is "one HBM read per input element, one write per output
// element": a 64-lane wave owns one row and keeps it in registers as NV4 float4 per lane
// (lane l holds columns 4*(l + 64*i) .. +3, i.e. every wave load/store is a contiguous 1-KiB
// segment), statistics are wave shuffles, and dropout masks are regenerated from a Philox
// counter instead of being stored.  4 rows per 256-thread workgroup; grids are >> 256 CUs.
// Parameter gradients (column sums over all rows) use a deterministic two-stage column reduction.
#include "common.h"

namespace segx {

constexpr int ROWS_PER_BLOCK = 4;
constexpr int RED_CHUNKS = 256;          // stage-1 row chunks of the column reductions

template <int NV4> struct Row { float4 v[NV4]; };

#define SEGX_FOR_ROW(i, c4, F) _Pragma("unroll") for (int i = 0, c4 = (threadIdx.x & 63); i < NV4; ++i, c4 += 64) if (c4 * 4 < (F))

template <int NV4> __device__ __forceinline__ void row_load(Row<NV4>& r, const float* __restrict__ p, int F) {
#pragma unroll
    for (int i = 0; i < NV4; ++i) r.v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    SEGX_FOR_ROW(i, c4, F) r.v[i] = reinterpret_cast<const float4*>(p)[c4];
}
template <int NV4> __device__ __forceinline__ void row_store(const Row<NV4>& r, float* __restrict__ p, int F) {
    SEGX_FOR_ROW(i, c4, F) reinterpret_cast<float4*>(p)[c4] = r.v[i];
}
template <int NV4> __device__ __forceinline__ float row_sum(const Row<NV4>& r) {     // padding lanes hold 0
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV4; ++i) s += (r.v[i].x + r.v[i].y) + (r.v[i].z + r.v[i].w);
    return wave_sum(s);
}
template <int NV4> __device__ __forceinline__ float row_dot(const Row<NV4>& a, const Row<NV4>& b) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV4; ++i) s += (a.v[i].x * b.v[i].x + a.v[i].y * b.v[i].y) + (a.v[i].z * b.v[i].z + a.v[i].w * b.v[i].w);
    return wave_sum(s);
}
// two-pass mean / rstd (matches torch's row-wise moments to fp32 round-off; eps = 1e-12 gives no slack)
template <int NV4> __device__ __forceinline__ void row_stats(const Row<NV4>& r, int F, float eps, float& mean, float& rstd) {
    mean = row_sum(r) / (float)F;
    float s = 0.f;
    SEGX_FOR_ROW(i, c4, F) {
        const float a = r.v[i].x - mean, b = r.v[i].y - mean, c = r.v[i].z - mean, d = r.v[i].w - mean;
        s += (a * a + b * b) + (c * c + d * d);
    }
    rstd = rsqrtf(wave_sum(s) / (float)F + eps);
}
#define SEGX_F4_OP(dst, expr_x, expr_y, expr_z, expr_w) do { dst.x = (expr_x); dst.y = (expr_y); dst.z = (expr_z); dst.w = (expr_w); } while (0)

// all dropout streams use offsets that are multiples of 4, so idx0 % 4 == 0 <=> (offset + idx0) % 4 == 0
__device__ __forceinline__ float4 f4_keep(uint64_t seed, uint64_t off, uint64_t idx0, float p, float inv_keep) {
    return dropout_scale4(seed, off, idx0, p, inv_keep);
}

static inline int nv4_for(int F) { const int n = (F + 255) / 256; return n <= 1 ? 1 : n <= 2 ? 2 : n <= 4 ? 4 : n <= 7 ? 7 : n <= 8 ? 8 : n <= 16 ? 16 : 0; }
#define SEGX_DISPATCH_NV4(F, KERNEL_CALL)                                                    \
    switch (segx::nv4_for(F)) {                                                              \
        case 1: { constexpr int NV4 = 1; KERNEL_CALL; } break;                               \
        case 2: { constexpr int NV4 = 2; KERNEL_CALL; } break;                               \
        case 4: { constexpr int NV4 = 4; KERNEL_CALL; } break;                               \
        case 7: { constexpr int NV4 = 7; KERNEL_CALL; } break;                               \
        case 8: { constexpr int NV4 = 8; KERNEL_CALL; } break;                               \
        case 16: { constexpr int NV4 = 16; KERNEL_CALL; } break;                             \
        default: return segx::fail(-1, "row width %d unsupported (max 4096, multiple of 4)", (int)(F)); \
    }
static inline dim3 row_grid(int64_t rows) { return dim3((unsigned)((rows + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK)); }
#define SEGX_ROW_ID() ((int64_t)blockIdx.x * ROWS_PER_BLOCK + (threadIdx.x >> 6))

// =================================================================================================
// Row softmax over the key axis (networks/segtran_shared.py:578-580 clip, :601 softmax, :605 dropout)
// =================================================================================================
template <int NV4>
__global__ __launch_bounds__(256) void softmax_fwd_kernel(const float* __restrict__ S, float* __restrict__ P, float* __restrict__ Pd,
                                                          int64_t rows, int L, float clip, const float* __restrict__ gmax,
                                                          float p, uint64_t seed, uint64_t off, const uint64_t* __restrict__ rbase) {
    off += rbase ? *rbase : 0;              // device-side step base of the Philox stream (segx_set_rng_base): replayed graphs draw fresh masks
    const int64_t row = SEGX_ROW_ID();
    if (row >= rows) return;
    const bool clamp = gmax && (*gmax > clip);                 // N5: clamp only when the GLOBAL max exceeds the clip
    Row<NV4> r; row_load(r, S + row * L, L);
    float m = -INFINITY;
    SEGX_FOR_ROW(i, c4, L) {
        if (clamp) SEGX_F4_OP(r.v[i], fminf(fmaxf(r.v[i].x, -clip), clip), fminf(fmaxf(r.v[i].y, -clip), clip),
                              fminf(fmaxf(r.v[i].z, -clip), clip), fminf(fmaxf(r.v[i].w, -clip), clip));
        m = fmaxf(m, fmaxf(fmaxf(r.v[i].x, r.v[i].y), fmaxf(r.v[i].z, r.v[i].w)));
    }
    m = wave_max(m);
    float s = 0.f;
    SEGX_FOR_ROW(i, c4, L) {
        SEGX_F4_OP(r.v[i], expf(r.v[i].x - m), expf(r.v[i].y - m), expf(r.v[i].z - m), expf(r.v[i].w - m));
        s += (r.v[i].x + r.v[i].y) + (r.v[i].z + r.v[i].w);
    }
    s = wave_sum(s);
    SEGX_FOR_ROW(i, c4, L) SEGX_F4_OP(r.v[i], r.v[i].x / s, r.v[i].y / s, r.v[i].z / s, r.v[i].w / s);
    row_store(r, P + row * L, L);
    if (Pd) {
        const float ik = 1.0f / (1.0f - p);
        SEGX_FOR_ROW(i, c4, L) {
            const float4 k = f4_keep(seed, off, (uint64_t)row * L + c4 * 4, p, ik);
            SEGX_F4_OP(r.v[i], r.v[i].x * k.x, r.v[i].y * k.y, r.v[i].z * k.z, r.v[i].w * k.w);
        }
        row_store(r, Pd + row * L, L);
    }
}

template <int NV4>
__global__ __launch_bounds__(256) void softmax_bwd_kernel(const float* __restrict__ P, const float* __restrict__ dPd, const float* __restrict__ S,
                                                          float* __restrict__ dS, int64_t rows, int L, float clip,
                                                          const float* __restrict__ gmax, float p, uint64_t seed, uint64_t off, const uint64_t* __restrict__ rbase) {
    off += rbase ? *rbase : 0;              // device-side step base of the Philox stream (segx_set_rng_base): replayed graphs draw fresh masks
    const int64_t row = SEGX_ROW_ID();
    if (row >= rows) return;
    const bool clamp = gmax && S && (*gmax > clip);
    Row<NV4> pr, g; row_load(pr, P + row * L, L); row_load(g, dPd + row * L, L);
    if (p > 0.f) {
        const float ik = 1.0f / (1.0f - p);
        SEGX_FOR_ROW(i, c4, L) {
            const float4 k = f4_keep(seed, off, (uint64_t)row * L + c4 * 4, p, ik);
            SEGX_F4_OP(g.v[i], g.v[i].x * k.x, g.v[i].y * k.y, g.v[i].z * k.z, g.v[i].w * k.w);
        }
    }
    const float dot = row_dot(pr, g);
    SEGX_FOR_ROW(i, c4, L) SEGX_F4_OP(g.v[i], pr.v[i].x * (g.v[i].x - dot), pr.v[i].y * (g.v[i].y - dot),
                                       pr.v[i].z * (g.v[i].z - dot), pr.v[i].w * (g.v[i].w - dot));
    if (clamp) {                                               // torch.clamp passes gradient where -clip <= s <= clip
        Row<NV4> s; row_load(s, S + row * L, L);
        SEGX_FOR_ROW(i, c4, L) SEGX_F4_OP(g.v[i], fabsf(s.v[i].x) <= clip ? g.v[i].x : 0.f, fabsf(s.v[i].y) <= clip ? g.v[i].y : 0.f,
                                           fabsf(s.v[i].z) <= clip ? g.v[i].z : 0.f, fabsf(s.v[i].w) <= clip ? g.v[i].w : 0.f);
    }
    row_store(g, dS + row * L, L);
}

// =================================================================================================
// Sliding positional biases (SlidingPosBiases2D/3D, networks/segtran_shared.py:1002-1175; K14): the [N, N] bias is a
// relative-offset lookup  bias[i][j] = table[pos(j) - pos(i) + R]  (0 outside the radius) and is NEVER materialised
// (the reference scatters it into a zero [H,W,H+2R,W+2R] tensor: 64 MB at N = 4096).
//   out = clamp_if_global_max_exceeds_clip(S) + w * bias        (:578-580 then :590-592)
// =================================================================================================
struct PosGeom { int D, H, W, R, nd; };          // token grid (D = 1 for 2-D), radius, number of position dims (2 or 3)
__device__ __forceinline__ int posbias_index(int i, int j, const PosGeom& q) {
    const int HW = q.H * q.W, T = 2 * q.R + 1;
    const int di = i / HW, ri = i - di * HW, hi = ri / q.W, wi = ri - hi * q.W;
    const int dj = j / HW, rj = j - dj * HW, hj = rj / q.W, wj = rj - hj * q.W;
    const int dd = dj - di, dh = hj - hi, dw = wj - wi;
    if (dd < -q.R || dd > q.R || dh < -q.R || dh > q.R || dw < -q.R || dw > q.R) return -1;
    return q.nd == 3 ? ((dd + q.R) * T + (dh + q.R)) * T + (dw + q.R) : (dh + q.R) * T + (dw + q.R);
}
__global__ __launch_bounds__(256) void posbias_fwd_kernel(const float* __restrict__ S, float* __restrict__ out, const float* __restrict__ table,
                                                          int64_t rows, int N, PosGeom q, float w, float clip, const float* __restrict__ gmax) {
    const bool clamp = gmax && (*gmax > clip);
    const int64_t total = rows * N;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        const int64_t row = idx / N; const int j = (int)(idx - row * N), i = (int)(row % N);
        float v = S[idx];
        if (clamp) v = fminf(fmaxf(v, -clip), clip);
        const int t = posbias_index(i, j, q);
        out[idx] = t >= 0 ? v + w * table[t] : v;
    }
}
// dS = dOut where the clamp passed the gradient (only launched when the clamp was active)
__global__ __launch_bounds__(256) void posbias_bwd_clamp_kernel(const float* __restrict__ dOut, const float* __restrict__ S, float* __restrict__ dS,
                                                                int64_t total, float clip) {
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256)
        dS[idx] = fabsf(S[idx]) <= clip ? dOut[idx] : 0.f;
}
// dtable[t] = w * sum over all (score matrix z, query i) of dOut[z][i][i + offset(t)]: one workgroup per table entry
__global__ __launch_bounds__(256) void posbias_bwd_table_kernel(const float* __restrict__ dOut, float* __restrict__ dtable, int64_t nmat, int N,
                                                                PosGeom q, float w) {
    __shared__ float red[4];
    const int T = 2 * q.R + 1, t = blockIdx.x;
    int dd = 0, dh, dw;
    if (q.nd == 3) { dd = t / (T * T) - q.R; dh = (t / T) % T - q.R; dw = t % T - q.R; } else { dh = t / T - q.R; dw = t % T - q.R; }
    const int HW = q.H * q.W;
    float s = 0.f;
    for (int64_t e = threadIdx.x; e < nmat * N; e += 256) {
        const int64_t z = e / N; const int i = (int)(e - z * N);
        const int di = i / HW, ri = i - di * HW, hi = ri / q.W, wi = ri - hi * q.W;
        const int dj = di + dd, hj = hi + dh, wj = wi + dw;
        if (dj < 0 || dj >= q.D || hj < 0 || hj >= q.H || wj < 0 || wj >= q.W) continue;
        s += dOut[(z * N + i) * N + (dj * q.H + hj) * q.W + wj];
    }
    s = block_sum<4>(s, red);
    if (threadIdx.x == 0) dtable[t] = w * s;
}

// =================================================================================================
// LayerNorm (eps 1e-12, N4): fwd, dX.  (:262,361 affine; :889 non-affine -> w == nullptr)
// =================================================================================================
// Softmax over rows of ANY width (token counts that are not a multiple of 4 -- an 88 x 88 image gives 11 x 11 = 121 tokens -- or exceed
// the 4096 a register-resident row holds): one workgroup per row, three passes over the row (max, sum, write), which stay in L1/L2.
// Same clamp rule and the same Philox stream indexing (element row * L + c) in forward and backward.
__global__ __launch_bounds__(256) void softmax_fwd_generic_kernel(const float* __restrict__ S, float* __restrict__ P, float* __restrict__ Pd,
                                                                  int64_t rows, int L, float clip, const float* __restrict__ gmax,
                                                                  float p, uint64_t seed, uint64_t off, const uint64_t* __restrict__ rbase) {
    off += rbase ? *rbase : 0;              // device-side step base of the Philox stream (segx_set_rng_base): replayed graphs draw fresh masks
    __shared__ float red[4];
    const bool clamp = gmax && (*gmax > clip);
    const float ik = p > 0.f ? 1.0f / (1.0f - p) : 1.0f;
    for (int64_t row = blockIdx.x; row < rows; row += gridDim.x) {
        const float* s = S + row * L;
        float m = -INFINITY;
        for (int c = threadIdx.x; c < L; c += 256) { float v = s[c]; if (clamp) v = fminf(fmaxf(v, -clip), clip); m = fmaxf(m, v); }
        m = block_max<4>(m, red);
        float sum = 0.f;
        for (int c = threadIdx.x; c < L; c += 256) { float v = s[c]; if (clamp) v = fminf(fmaxf(v, -clip), clip); sum += expf(v - m); }
        sum = block_sum<4>(sum, red);
        for (int c = threadIdx.x; c < L; c += 256) {
            float v = s[c]; if (clamp) v = fminf(fmaxf(v, -clip), clip);
            const float pr = expf(v - m) / sum;
            P[row * L + c] = pr;
            if (Pd) Pd[row * L + c] = pr * dropout_scale(seed, off, (uint64_t)row * L + c, p, ik);
        }
    }
}
__global__ __launch_bounds__(256) void softmax_bwd_generic_kernel(const float* __restrict__ P, const float* __restrict__ dPd, const float* __restrict__ S,
                                                                  float* __restrict__ dS, int64_t rows, int L, float clip,
                                                                  const float* __restrict__ gmax, float p, uint64_t seed, uint64_t off, const uint64_t* __restrict__ rbase) {
    off += rbase ? *rbase : 0;              // device-side step base of the Philox stream (segx_set_rng_base): replayed graphs draw fresh masks
    __shared__ float red[4];
    const bool clamp = gmax && S && (*gmax > clip);
    const float ik = p > 0.f ? 1.0f / (1.0f - p) : 1.0f;
    for (int64_t row = blockIdx.x; row < rows; row += gridDim.x) {
        float dot = 0.f;
        for (int c = threadIdx.x; c < L; c += 256) {
            float g = dPd[row * L + c];
            if (p > 0.f) g *= dropout_scale(seed, off, (uint64_t)row * L + c, p, ik);
            dot += P[row * L + c] * g;
        }
        dot = block_sum<4>(dot, red);
        for (int c = threadIdx.x; c < L; c += 256) {
            float g = dPd[row * L + c];
            if (p > 0.f) g *= dropout_scale(seed, off, (uint64_t)row * L + c, p, ik);
            float d = P[row * L + c] * (g - dot);
            if (clamp && !(fabsf(S[row * L + c]) <= clip)) d = 0.f;
            dS[row * L + c] = d;
        }
    }
}
template <int NV4>
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const float* __restrict__ X, const float* __restrict__ w, const float* __restrict__ b,
                                                            float* __restrict__ Y, float* __restrict__ mean_o, float* __restrict__ rstd_o,
                                                            int64_t rows, int C, float eps) {
    const int64_t row = SEGX_ROW_ID();
    if (row >= rows) return;
    Row<NV4> r; row_load(r, X + row * C, C);
    float mean, rstd; row_stats(r, C, eps, mean, rstd);
    SEGX_FOR_ROW(i, c4, C) {
        float4 ww = make_float4(1.f, 1.f, 1.f, 1.f), bb = make_float4(0.f, 0.f, 0.f, 0.f);
        if (w) { ww = reinterpret_cast<const float4*>(w)[c4]; bb = reinterpret_cast<const float4*>(b)[c4]; }
        SEGX_F4_OP(r.v[i], (r.v[i].x - mean) * rstd * ww.x + bb.x, (r.v[i].y - mean) * rstd * ww.y + bb.y,
                   (r.v[i].z - mean) * rstd * ww.z + bb.z, (r.v[i].w - mean) * rstd * ww.w + bb.w);
    }
    row_store(r, Y + row * C, C);
    if ((threadIdx.x & 63) == 0) { mean_o[row] = mean; rstd_o[row] = rstd; }
}

// dx = rstd * (g - mean(g) - xhat * mean(g * xhat)),  g = dy * w
template <int NV4>
__device__ __forceinline__ void ln_bwd_row(Row<NV4>& g, const Row<NV4>& xhat, int C, float rstd) {
    const float m1 = row_sum(g) / (float)C, m2 = row_dot(g, xhat) / (float)C;
    SEGX_FOR_ROW(i, c4, C) SEGX_F4_OP(g.v[i], rstd * (g.v[i].x - m1 - xhat.v[i].x * m2), rstd * (g.v[i].y - m1 - xhat.v[i].y * m2),
                                       rstd * (g.v[i].z - m1 - xhat.v[i].z * m2), rstd * (g.v[i].w - m1 - xhat.v[i].w * m2));
}

template <int NV4>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const float* __restrict__ dY, const float* __restrict__ X, const float* __restrict__ w,
                                                            const float* __restrict__ mean_i, const float* __restrict__ rstd_i,
                                                            float* __restrict__ dX, int64_t rows, int C) {
    const int64_t row = SEGX_ROW_ID();
    if (row >= rows) return;
    Row<NV4> x, g; row_load(x, X + row * C, C); row_load(g, dY + row * C, C);
    const float mean = mean_i[row], rstd = rstd_i[row];
    SEGX_FOR_ROW(i, c4, C) {
        SEGX_F4_OP(x.v[i], (x.v[i].x - mean) * rstd, (x.v[i].y - mean) * rstd, (x.v[i].z - mean) * rstd, (x.v[i].w - mean) * rstd);
        if (w) { const float4 ww = reinterpret_cast<const float4*>(w)[c4];
                 SEGX_F4_OP(g.v[i], g.v[i].x * ww.x, g.v[i].y * ww.y, g.v[i].z * ww.z, g.v[i].w * ww.w); }
    }
    ln_bwd_row(g, x, C, rstd);
    row_store(g, dX + row * C, C);
}

// =================================================================================================
// Column reductions (parameter gradients).  Stage 1: thread = column, block = (256 columns, row chunk);
// consecutive threads read consecutive columns (coalesced).  Stage 2 sums the chunk partials in order.
// =================================================================================================
// mode 0: out0 = sum_r X              (bias grads, pos-code batch sum)
// mode 1: out0 = sum_r dY * xhat, out1 = sum_r dY      (LayerNorm weight / bias grads), xhat = (X - mean) * rstd
__global__ __launch_bounds__(256) void colreduce_stage1(const float* __restrict__ A, const float* __restrict__ X, const float* __restrict__ mean,
                                                        const float* __restrict__ rstd, float* __restrict__ ws, int64_t rows, int64_t C,
                                                        int mode, int nchunks) {
    const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int chunk = blockIdx.y;
    const int64_t per = (rows + nchunks - 1) / nchunks, r0 = chunk * per, r1 = i64min(rows, r0 + per);
    if (c >= C) return;
    float s0 = 0.f, s1 = 0.f;
    if (mode == 0) {
#pragma unroll 8
        for (int64_t r = r0; r < r1; ++r) s0 += A[r * C + c];          // unrolled: eight loads in flight instead of one
    } else {
#pragma unroll 4
        for (int64_t r = r0; r < r1; ++r) {
            const float a = A[r * C + c];
            s0 += a * ((X[r * C + c] - mean[r]) * rstd[r]); s1 += a;
        }
    }
    ws[(int64_t)chunk * C + c] = s0;
    if (mode == 1) ws[((int64_t)nchunks + chunk) * C + c] = s1;
}
// four adjacent columns per thread (16-byte loads): wide tensors only (C >= 1024: enough threads per row chunk), same sums in the same order per column
__global__ __launch_bounds__(256) void colreduce_stage1_v4(const float* __restrict__ A, const float* __restrict__ X, const float* __restrict__ mean,
                                                           const float* __restrict__ rstd, float* __restrict__ ws, int64_t rows, int64_t C,
                                                           int mode, int nchunks) {
    const int64_t c = 4 * ((int64_t)blockIdx.x * 256 + threadIdx.x);
    const int chunk = blockIdx.y;
    const int64_t per = (rows + nchunks - 1) / nchunks, r0 = chunk * per, r1 = i64min(rows, r0 + per);
    if (c >= C) return;
    float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0;
    if (mode == 0) {
#pragma unroll 8
        for (int64_t r = r0; r < r1; ++r) { const float4 a = *reinterpret_cast<const float4*>(A + r * C + c); s0.x += a.x; s0.y += a.y; s0.z += a.z; s0.w += a.w; }
    } else {
#pragma unroll 4
        for (int64_t r = r0; r < r1; ++r) {
            const float4 a = *reinterpret_cast<const float4*>(A + r * C + c), x = *reinterpret_cast<const float4*>(X + r * C + c);
            const float m = mean[r], rs = rstd[r];
            s0.x += a.x * ((x.x - m) * rs); s0.y += a.y * ((x.y - m) * rs); s0.z += a.z * ((x.z - m) * rs); s0.w += a.w * ((x.w - m) * rs);
            s1.x += a.x; s1.y += a.y; s1.z += a.z; s1.w += a.w;
        }
    }
    *reinterpret_cast<float4*>(ws + (int64_t)chunk * C + c) = s0;
    if (mode == 1) *reinterpret_cast<float4*>(ws + ((int64_t)nchunks + chunk) * C + c) = s1;
}
// r04-p: 64 columns x 4 parts per workgroup (part p adds chunks p, p + 4, ... in that order; the four partial sums are added in part order through LDS: a fixed
// tree).  One thread per column walked all chunks alone -- seven workgroups on the chip for 1792 columns, 15.8 us of dependent loads for a few hundred KB.
__global__ __launch_bounds__(256) void colreduce_stage2(const float* __restrict__ ws, float* __restrict__ out0, float* __restrict__ out1,
                                                        float* __restrict__ out2, int64_t C, int nchunks, int nout) {
    __shared__ float part[3][4][64];
    const int col = threadIdx.x & 63, p = threadIdx.x >> 6;
    const int64_t c = (int64_t)blockIdx.x * 64 + col;
    for (int o = 0; o < nout; ++o) {
        float s = 0.f;
        if (c < C) {
#pragma unroll 8
            for (int k = p; k < nchunks; k += 4) s += ws[((int64_t)o * nchunks + k) * C + c];
        }
        part[o][p][col] = s;
    }
    __syncthreads();
    if (p == 0 && c < C)
        for (int o = 0; o < nout; ++o) (o == 0 ? out0 : o == 1 ? out1 : out2)[c] = ((part[o][0][col] + part[o][1][col]) + part[o][2][col]) + part[o][3][col];
}
// r06: the one-pass backward kernels (modes_aggr_bwd_all, prenorm_bwd_all, gelu_bwd_colsum) leave 1 000 - 2 000 chunks; with four parts per column a thread walked 250 - 500 dependent
// loads on 28 workgroups (25 - 80 us per call, r06_z).  16 columns x 16 parts per workgroup: four times the workgroups, chains a quarter as long; part p adds chunks p, p + 16, ... in that
// order and the sixteen partial sums are added in part order through LDS -- a fixed tree.
__global__ __launch_bounds__(256) void colreduce_stage2_p16(const float* __restrict__ ws, float* __restrict__ out0, float* __restrict__ out1,
                                                            float* __restrict__ out2, int64_t C, int nchunks, int nout) {
    __shared__ float part[3][16][16];
    const int col = threadIdx.x & 15, p = threadIdx.x >> 4;
    const int64_t c = (int64_t)blockIdx.x * 16 + col;
    for (int o = 0; o < nout; ++o) {
        float s = 0.f;
        if (c < C) {
#pragma unroll 8
            for (int k = p; k < nchunks; k += 16) s += ws[((int64_t)o * nchunks + k) * C + c];
        }
        part[o][p][col] = s;
    }
    __syncthreads();
    if (p == 0 && c < C)
        for (int o = 0; o < nout; ++o) {
            float t = part[o][0][col];
#pragma unroll
            for (int q = 1; q < 16; ++q) t += part[o][q][col];
            (o == 0 ? out0 : o == 1 ? out1 : out2)[c] = t;
        }
}
static inline void launch_colreduce2(hipStream_t stream, const float* ws, float* o0, float* o1, float* o2, int64_t C, int nch, int nout) {
    if (nch >= 256) hipLaunchKernelGGL(colreduce_stage2_p16, dim3((unsigned)((C + 15) / 16)), dim3(256), 0, stream, ws, o0, o1, o2, C, nch, nout);
    else hipLaunchKernelGGL(colreduce_stage2, dim3((unsigned)((C + 63) / 64)), dim3(256), 0, stream, ws, o0, o1, o2, C, nch, nout);
}
static inline int chunks_for(int64_t rows) { return (int)i64max(1, i64min(RED_CHUNKS, rows / 8)); }

// row sums of X [R, S] (conv-style bias gradients: one value per (sample, channel)): one workgroup per row
__global__ __launch_bounds__(256) void rowsum_kernel(const float* __restrict__ X, float* __restrict__ out, int64_t S, int vec) {
    __shared__ float red[4];
    const float* x = X + (int64_t)blockIdx.x * S;
    float s = 0.f;
    if (vec) {
        const float4* x4 = reinterpret_cast<const float4*>(x);
        for (int64_t i = threadIdx.x; i < S / 4; i += 256) { const float4 v = x4[i]; s += (v.x + v.y) + (v.z + v.w); }
    } else {
        for (int64_t i = threadIdx.x; i < S; i += 256) s += x[i];
    }
    s = block_sum<4>(s, red);
    if (threadIdx.x == 0) out[blockIdx.x] = s;
}

// full sum of a flat array (loss pieces, scalar bias grads): two-stage, deterministic
__global__ __launch_bounds__(256) void sum_stage1(const float* __restrict__ x, int64_t n, float* __restrict__ ws) {
    __shared__ float red[4];
    float s = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) s += x[i];
    s = block_sum<4>(s, red);
    if (threadIdx.x == 0) ws[blockIdx.x] = s;
}
__global__ __launch_bounds__(256) void sum_stage2(const float* __restrict__ ws, int nb, float* __restrict__ out, float scale) {
    __shared__ float red[4];
    float s = 0.f;
    for (int i = threadIdx.x; i < nb; i += 256) s += ws[i];
    s = block_sum<4>(s, red);
    if (threadIdx.x == 0) out[0] = s * scale;
}

// =================================================================================================
// Token pre-norm (SegtranFusionEncoder.forward :916-946, K12):
//   y = mask * dropout( LN_noaffine( LN_affine(x) + pos_weight * pos[n, :C] ) )
// stats[0..3][row] = mean1, rstd1, mean2, rstd2
// =================================================================================================
template <int NV4>
__global__ __launch_bounds__(256) void prenorm_fwd_kernel(const float* __restrict__ X, const float* __restrict__ w1, const float* __restrict__ b1,
                                                          const float* __restrict__ pos, int64_t pos_ld, float pos_w, const float* __restrict__ mask,
                                                          float* __restrict__ Y, float* __restrict__ stats, int64_t rows, int N, int C, float eps,
                                                          float p, uint64_t seed, uint64_t off, const uint64_t* __restrict__ rbase) {
    off += rbase ? *rbase : 0;              // device-side step base of the Philox stream (segx_set_rng_base): replayed graphs draw fresh masks
    const int64_t row = SEGX_ROW_ID();
    if (row >= rows) return;
    const int n = (int)(row % N);
    Row<NV4> r; row_load(r, X + row * C, C);
    float m1, r1; row_stats(r, C, eps, m1, r1);
    SEGX_FOR_ROW(i, c4, C) {
        const float4 ww = reinterpret_cast<const float4*>(w1)[c4], bb = reinterpret_cast<const float4*>(b1)[c4];
        const float4 pp = pos ? reinterpret_cast<const float4*>(pos + (int64_t)n * pos_ld)[c4] : make_float4(0.f, 0.f, 0.f, 0.f);
        SEGX_F4_OP(r.v[i], (r.v[i].x - m1) * r1 * ww.x + bb.x + pos_w * pp.x, (r.v[i].y - m1) * r1 * ww.y + bb.y + pos_w * pp.y,
                   (r.v[i].z - m1) * r1 * ww.z + bb.z + pos_w * pp.z, (r.v[i].w - m1) * r1 * ww.w + bb.w + pos_w * pp.w);
    }
    float m2 = 0.f, r2 = 1.f;                     // pos == NULL ('bias' codes, :937-940): no second norm
    if (pos) row_stats(r, C, eps, m2, r2);
    const float mk = mask[row];
    const float ik = p > 0.f ? 1.0f / (1.0f - p) : 1.0f;
    SEGX_FOR_ROW(i, c4, C) {
        float4 k = make_float4(1.f, 1.f, 1.f, 1.f);
        if (p > 0.f) k = f4_keep(seed, off, (uint64_t)row * C + c4 * 4, p, ik);
        SEGX_F4_OP(r.v[i], (r.v[i].x - m2) * r2 * k.x * mk, (r.v[i].y - m2) * r2 * k.y * mk,
                   (r.v[i].z - m2) * r2 * k.z * mk, (r.v[i].w - m2) * r2 * k.w * mk);
    }
    row_store(r, Y + row * C, C);
    if ((threadIdx.x & 63) == 0) { stats[row] = m1; stats[rows + row] = r1; stats[2 * rows + row] = m2; stats[3 * rows + row] = r2; }
}

// outputs dX and dU (= grad wrt the sum LN_affine(x) + pos_w*pos, i.e. wrt LN_affine's output)
template <int NV4>
__global__ __launch_bounds__(256) void prenorm_bwd_kernel(const float* __restrict__ dY, const float* __restrict__ X, const float* __restrict__ w1,
                                                          const float* __restrict__ b1, const float* __restrict__ pos, int64_t pos_ld, float pos_w,
                                                          const float* __restrict__ mask, const float* __restrict__ stats,
                                                          float* __restrict__ dX, float* __restrict__ dU, int64_t rows, int N, int C,
                                                          float p, uint64_t seed, uint64_t off, const uint64_t* __restrict__ rbase) {
    off += rbase ? *rbase : 0;              // device-side step base of the Philox stream (segx_set_rng_base): replayed graphs draw fresh masks
    const int64_t row = SEGX_ROW_ID();
    if (row >= rows) return;
    const int n = (int)(row % N);
    const float m1 = stats[row], r1 = stats[rows + row], m2 = stats[2 * rows + row], r2 = stats[3 * rows + row];
    const float mk = mask[row];
    const float ik = p > 0.f ? 1.0f / (1.0f - p) : 1.0f;
    Row<NV4> xh, u, g; row_load(xh, X + row * C, C); row_load(g, dY + row * C, C);
    SEGX_FOR_ROW(i, c4, C) {
        const float4 ww = reinterpret_cast<const float4*>(w1)[c4], bb = reinterpret_cast<const float4*>(b1)[c4];
        const float4 pp = pos ? reinterpret_cast<const float4*>(pos + (int64_t)n * pos_ld)[c4] : make_float4(0.f, 0.f, 0.f, 0.f);
        SEGX_F4_OP(xh.v[i], (xh.v[i].x - m1) * r1, (xh.v[i].y - m1) * r1, (xh.v[i].z - m1) * r1, (xh.v[i].w - m1) * r1);
        SEGX_F4_OP(u.v[i], ((xh.v[i].x * ww.x + bb.x + pos_w * pp.x) - m2) * r2, ((xh.v[i].y * ww.y + bb.y + pos_w * pp.y) - m2) * r2,
                   ((xh.v[i].z * ww.z + bb.z + pos_w * pp.z) - m2) * r2, ((xh.v[i].w * ww.w + bb.w + pos_w * pp.w) - m2) * r2);
        float4 k = make_float4(1.f, 1.f, 1.f, 1.f);
        if (p > 0.f) k = f4_keep(seed, off, (uint64_t)row * C + c4 * 4, p, ik);
        SEGX_F4_OP(g.v[i], g.v[i].x * k.x * mk, g.v[i].y * k.y * mk, g.v[i].z * k.z * mk, g.v[i].w * k.w * mk);
    }
#pragma unroll
    for (int i = 0; i < NV4; ++i) if (!(((threadIdx.x & 63) + 64 * i) * 4 < C)) u.v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (pos) ln_bwd_row(g, u, C, r2);             // g := dU
    row_store(g, dU + row * C, C);
    SEGX_FOR_ROW(i, c4, C) { const float4 ww = reinterpret_cast<const float4*>(w1)[c4];
                             SEGX_F4_OP(g.v[i], g.v[i].x * ww.x, g.v[i].y * ww.y, g.v[i].z * ww.z, g.v[i].w * ww.w); }
    ln_bwd_row(g, xh, C, r1);
    row_store(g, dX + row * C, C);
}

// r06: the same backward with everything that was computed FROM dU formed in the kernel, so that dU is never written: the LayerNorm-1 parameter gradients
// (dw1 = sum dU * xhat1, db1 = sum dU over all rows) and the positional code's gradient (dsum[n] = sum_b dU[b][n], the caller scales it by pos_weight).  A wave owns
// TOKEN n and walks the B samples (row b N + n): dsum[n] is a register row summed in sample order, dw1 / db1 are added into LDS vectors of the wave's own (as in
// modes_aggr_bwd_kernel<.., true>), one chunk of the column-reduction workspace per workgroup.  Before: dU written (0.18 GB), read twice by segx_ln_param_grad (with X)
// and once more by the batch sum -- 0.7 GB per call for two vectors and one [N, C] matrix.
template <int NV4>
__global__ __launch_bounds__(256) void prenorm_bwd_all_kernel(const float* __restrict__ dY, const float* __restrict__ X, const float* __restrict__ w1,
                                                              const float* __restrict__ b1, const float* __restrict__ pos, int64_t pos_ld, float pos_w,
                                                              const float* __restrict__ mask, const float* __restrict__ stats,
                                                              float* __restrict__ dX, float* __restrict__ dsum, float* __restrict__ pws, int B, int N, int C,
                                                              float p, uint64_t seed, uint64_t off, const uint64_t* __restrict__ rbase) {
    __shared__ float4 pacc[4 * 2 * NV4 * 64];
    off += rbase ? *rbase : 0;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, C4 = C >> 2;
    const int n0 = blockIdx.x * ROWS_PER_BLOCK + wv;
    const bool live = n0 < N;
    const int n = live ? n0 : N - 1;
    const int64_t rows = (int64_t)B * N;
    const float ik = p > 0.f ? 1.0f / (1.0f - p) : 1.0f;
    float4* const myacc = pacc + wv * 2 * NV4 * 64;
#pragma unroll
    for (int i = 0; i < 2 * NV4; ++i) myacc[lane + 64 * i] = make_float4(0.f, 0.f, 0.f, 0.f);
    Row<NV4> su;
#pragma unroll
    for (int i = 0; i < NV4; ++i) su.v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int b = 0; b < B; ++b) {
        const int64_t row = (int64_t)b * N + n;
        const float m1 = stats[row], r1 = stats[rows + row], m2 = stats[2 * rows + row], r2 = stats[3 * rows + row];
        const float mk = mask[row];
        Row<NV4> xh, u, g; row_load(xh, X + row * C, C); row_load(g, dY + row * C, C);
        SEGX_FOR_ROW(i, c4, C) {
            const float4 ww = reinterpret_cast<const float4*>(w1)[c4], bb = reinterpret_cast<const float4*>(b1)[c4];
            const float4 pp = pos ? reinterpret_cast<const float4*>(pos + (int64_t)n * pos_ld)[c4] : make_float4(0.f, 0.f, 0.f, 0.f);
            SEGX_F4_OP(xh.v[i], (xh.v[i].x - m1) * r1, (xh.v[i].y - m1) * r1, (xh.v[i].z - m1) * r1, (xh.v[i].w - m1) * r1);
            SEGX_F4_OP(u.v[i], ((xh.v[i].x * ww.x + bb.x + pos_w * pp.x) - m2) * r2, ((xh.v[i].y * ww.y + bb.y + pos_w * pp.y) - m2) * r2,
                       ((xh.v[i].z * ww.z + bb.z + pos_w * pp.z) - m2) * r2, ((xh.v[i].w * ww.w + bb.w + pos_w * pp.w) - m2) * r2);
            float4 k = make_float4(1.f, 1.f, 1.f, 1.f);
            if (p > 0.f) k = f4_keep(seed, off, (uint64_t)row * C + c4 * 4, p, ik);
            SEGX_F4_OP(g.v[i], g.v[i].x * k.x * mk, g.v[i].y * k.y * mk, g.v[i].z * k.z * mk, g.v[i].w * k.w * mk);
        }
#pragma unroll
        for (int i = 0; i < NV4; ++i) if (!((lane + 64 * i) * 4 < C)) { u.v[i] = make_float4(0.f, 0.f, 0.f, 0.f); xh.v[i] = make_float4(0.f, 0.f, 0.f, 0.f); g.v[i] = make_float4(0.f, 0.f, 0.f, 0.f); }
        if (pos) ln_bwd_row(g, u, C, r2);             // g := dU
        if (live) {
#pragma unroll
            for (int i = 0; i < NV4; ++i) {
                const int c4 = lane + 64 * i;
                float4 a0 = myacc[c4], a1 = myacc[NV4 * 64 + c4];
                SEGX_F4_OP(a0, a0.x + g.v[i].x * xh.v[i].x, a0.y + g.v[i].y * xh.v[i].y, a0.z + g.v[i].z * xh.v[i].z, a0.w + g.v[i].w * xh.v[i].w);
                SEGX_F4_OP(a1, a1.x + g.v[i].x, a1.y + g.v[i].y, a1.z + g.v[i].z, a1.w + g.v[i].w);
                myacc[c4] = a0; myacc[NV4 * 64 + c4] = a1;
                SEGX_F4_OP(su.v[i], su.v[i].x + g.v[i].x, su.v[i].y + g.v[i].y, su.v[i].z + g.v[i].z, su.v[i].w + g.v[i].w);
            }
        }
        SEGX_FOR_ROW(i, c4, C) { const float4 ww = reinterpret_cast<const float4*>(w1)[c4];
                                 SEGX_F4_OP(g.v[i], g.v[i].x * ww.x, g.v[i].y * ww.y, g.v[i].z * ww.z, g.v[i].w * ww.w); }
        ln_bwd_row(g, xh, C, r1);
        if (live) row_store(g, dX + row * C, C);
    }
    if (dsum && live) row_store(su, dsum + (int64_t)n * C, C);
    __syncthreads();
    const int64_t nchunks = gridDim.x;
    for (int idx = threadIdx.x; idx < C4; idx += 256) {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const float4 v0 = pacc[(0 * 2 + k) * NV4 * 64 + idx], v1 = pacc[(1 * 2 + k) * NV4 * 64 + idx], v2 = pacc[(2 * 2 + k) * NV4 * 64 + idx],
                         v3 = pacc[(3 * 2 + k) * NV4 * 64 + idx];
            float4 o;
            SEGX_F4_OP(o, ((v0.x + v1.x) + v2.x) + v3.x, ((v0.y + v1.y) + v2.y) + v3.y, ((v0.z + v1.z) + v2.z) + v3.z, ((v0.w + v1.w) + v2.w) + v3.w);
            reinterpret_cast<float4*>(pws + ((int64_t)k * nchunks + blockIdx.x) * C)[idx] = o;
        }
    }
}

// =================================================================================================
// Learned sinusoidal positional code (LearnedSinuPosEmbedder.forward :989-998, K13), batch-invariant:
//   z = posn @ Wp^T + bp ; mix[c] = c even ? sin z : cos z ; out = LN_noaffine(mix)
// =================================================================================================
template <int NV4>
__global__ __launch_bounds__(256) void posembed_fwd_kernel(const float* __restrict__ posn, const float* __restrict__ Wp, const float* __restrict__ bp,
                                                           float* __restrict__ out, float* __restrict__ stats, int64_t N, int C, int pd, float eps) {
    const int64_t row = SEGX_ROW_ID();
    if (row >= N) return;
    float pc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int d = 0; d < pd; ++d) pc[d] = posn[row * pd + d];
    Row<NV4> r;
#pragma unroll
    for (int i = 0; i < NV4; ++i) r.v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    SEGX_FOR_ROW(i, c4, C) {
        float z[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = c4 * 4 + j;
            float a = 0.f;
            for (int d = 0; d < pd; ++d) a += pc[d] * Wp[(int64_t)c * pd + d];
            z[j] = a + bp[c];
        }
        SEGX_F4_OP(r.v[i], sinf(z[0]), cosf(z[1]), sinf(z[2]), cosf(z[3]));      // c4*4 is even
    }
    float m, rs; row_stats(r, C, eps, m, rs);
    SEGX_FOR_ROW(i, c4, C) SEGX_F4_OP(r.v[i], (r.v[i].x - m) * rs, (r.v[i].y - m) * rs, (r.v[i].z - m) * rs, (r.v[i].w - m) * rs);
    row_store(r, out + row * C, C);
    if ((threadIdx.x & 63) == 0) { stats[row] = m; stats[N + row] = rs; }
}
// dZ[n,c] = d(loss)/d z[n,c];  then dWp = dZ^T posn (GEMM), dbp = colsum(dZ)
template <int NV4>
__global__ __launch_bounds__(256) void posembed_bwd_kernel(const float* __restrict__ dOut, const float* __restrict__ posn, const float* __restrict__ Wp,
                                                           const float* __restrict__ bp, const float* __restrict__ stats, float* __restrict__ dZ,
                                                           int64_t N, int C, int pd) {
    const int64_t row = SEGX_ROW_ID();
    if (row >= N) return;
    float pc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int d = 0; d < pd; ++d) pc[d] = posn[row * pd + d];
    const float m = stats[row], rs = stats[N + row];
    Row<NV4> mh, dv, g; row_load(g, dOut + row * C, C);
#pragma unroll
    for (int i = 0; i < NV4; ++i) { mh.v[i] = make_float4(0.f, 0.f, 0.f, 0.f); dv.v[i] = make_float4(0.f, 0.f, 0.f, 0.f); }
    SEGX_FOR_ROW(i, c4, C) {
        float z[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = c4 * 4 + j;
            float a = 0.f;
            for (int d = 0; d < pd; ++d) a += pc[d] * Wp[(int64_t)c * pd + d];
            z[j] = a + bp[c];
        }
        const float s0 = sinf(z[0]), c0 = cosf(z[0]), s1 = sinf(z[1]), c1 = cosf(z[1]);
        const float s2 = sinf(z[2]), c2 = cosf(z[2]), s3 = sinf(z[3]), c3 = cosf(z[3]);
        SEGX_F4_OP(mh.v[i], (s0 - m) * rs, (c1 - m) * rs, (s2 - m) * rs, (c3 - m) * rs);
        SEGX_F4_OP(dv.v[i], c0, -s1, c2, -s3);                                   // d mix / d z
    }
    ln_bwd_row(g, mh, C, rs);
    SEGX_FOR_ROW(i, c4, C) SEGX_F4_OP(g.v[i], g.v[i].x * dv.v[i].x, g.v[i].y * dv.v[i].y, g.v[i].z * dv.v[i].z, g.v[i].w * dv.v[i].w);
    row_store(g, dZ + row * C, C);
}

// =================================================================================================
// Expansion tail (MMPrivateOutput dropout + LayerNorm :273-274, LearnedSoftAggregate :318-325; K10b+K11)
//   Z [Mo, R, F] mode-major -> zn_m = LN(dropout(z_m)) ; s_m = zn_m . wa + ba ; pr = softmax_m(s) ; y = sum_m pr_m zn_m
// stats: mean[Mo*R], rstd[Mo*R], prob[Mo*R]
// =================================================================================================
// r04-g: all MO rows of a token are requested before the first is used, and the LayerNorm / aggregation vectors come from LDS (staged once per
// workgroup of four tokens) -- see modes_aggr_bwd_kernel.
template <int NV4, int MO>
__global__ __launch_bounds__(256) void modes_aggr_fwd_kernel(const float* __restrict__ Z, const float* __restrict__ lnw, const float* __restrict__ lnb,
                                                             const float* __restrict__ wa, const float* __restrict__ ba, float* __restrict__ Y,
                                                             float* __restrict__ stats, int64_t R, int F, float eps,
                                                             float p, uint64_t seed, uint64_t off, const uint64_t* __restrict__ rbase) {
    __shared__ float4 sw[NV4 * 64], sb[NV4 * 64], sa[NV4 * 64];
    off += rbase ? *rbase : 0;              // device-side step base of the Philox stream (segx_set_rng_base): replayed graphs draw fresh masks
    const int64_t row_id = SEGX_ROW_ID();
    const bool live = row_id < R;                                    // wave-uniform; a dead wave of the last workgroup works on row R - 1 and stores nothing
    const int64_t row = live ? row_id : R - 1;
    const int lane = threadIdx.x & 63, F4 = F >> 2;
    const float ik = p > 0.f ? 1.0f / (1.0f - p) : 1.0f;
    for (int idx = threadIdx.x; idx < NV4 * 64; idx += 256) {
        const int c4 = idx < F4 ? idx : F4 - 1;
        sw[idx] = lnw ? reinterpret_cast<const float4*>(lnw)[c4] : make_float4(1.f, 1.f, 1.f, 1.f);
        sb[idx] = lnw ? reinterpret_cast<const float4*>(lnb)[c4] : make_float4(0.f, 0.f, 0.f, 0.f);
        sa[idx] = reinterpret_cast<const float4*>(wa)[c4];
    }
    Row<NV4> z[MO];
    float sc[MO];
#pragma unroll
    for (int m = 0; m < MO; ++m) {
        const int64_t mr = (int64_t)m * R + row;
#pragma unroll
        for (int i = 0; i < NV4; ++i) { const int c4 = lane + 64 * i; z[m].v[i] = reinterpret_cast<const float4*>(Z + mr * F)[c4 < F4 ? c4 : F4 - 1]; }
    }
    const float ba0 = ba[0];
    __syncthreads();
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int m = 0; m < MO; ++m) {
        const int64_t mr = (int64_t)m * R + row;
#pragma unroll
        for (int i = 0; i < NV4; ++i) {
            const int c4 = lane + 64 * i;
            if (!(c4 < F4)) z[m].v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            else if (p > 0.f) {
                const float4 k = f4_keep(seed, off, (uint64_t)mr * F + c4 * 4, p, ik);
                SEGX_F4_OP(z[m].v[i], z[m].v[i].x * k.x, z[m].v[i].y * k.y, z[m].v[i].z * k.z, z[m].v[i].w * k.w);
            }
        }
        float mean = 0.f, rstd = 1.f;                     // lnw == NULL: aggregate the raw mode features (no-FFN branch, :452-457)
        if (lnw) row_stats(z[m], F, eps, mean, rstd);
        float s = 0.f;
        SEGX_FOR_ROW(i, c4, F) {
            const float4 ww = sw[c4], bb = sb[c4], aa = sa[c4];
            SEGX_F4_OP(z[m].v[i], (z[m].v[i].x - mean) * rstd * ww.x + bb.x, (z[m].v[i].y - mean) * rstd * ww.y + bb.y,
                       (z[m].v[i].z - mean) * rstd * ww.z + bb.z, (z[m].v[i].w - mean) * rstd * ww.w + bb.w);
            s += (z[m].v[i].x * aa.x + z[m].v[i].y * aa.y) + (z[m].v[i].z * aa.z + z[m].v[i].w * aa.w);
        }
        sc[m] = wave_sum(s) + ba0;
        if (lane == 0 && live) { stats[mr] = mean; stats[(int64_t)MO * R + mr] = rstd; }
        __builtin_amdgcn_sched_barrier(0);
    }
    float mx = sc[0];
#pragma unroll
    for (int m = 1; m < MO; ++m) mx = fmaxf(mx, sc[m]);
    float den = 0.f;
#pragma unroll
    for (int m = 0; m < MO; ++m) { sc[m] = expf(sc[m] - mx); den += sc[m]; }
    Row<NV4> y;
#pragma unroll
    for (int i = 0; i < NV4; ++i) y.v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int m = 0; m < MO; ++m) {
        const float pr = sc[m] / den;
        if (lane == 0 && live) stats[(int64_t)2 * MO * R + (int64_t)m * R + row] = pr;
        SEGX_FOR_ROW(i, c4, F) SEGX_F4_OP(y.v[i], y.v[i].x + z[m].v[i].x * pr, y.v[i].y + z[m].v[i].y * pr,
                                           y.v[i].z + z[m].v[i].z * pr, y.v[i].w + z[m].v[i].w * pr);
    }
    if (live) row_store(y, Y + row * F, F);
}

// dZ and the per-(mode,row) score gradients ds (needed again by the parameter-gradient pass)
// r04-g: the MO rows of a token stay in registers for both passes and ALL their loads are issued before the first use.  The first version re-read Z
// in the second pass (2.3 GB at cfg2 where 1.6 are needed) and, worse, its predicated loads were fused with the loops that consume them: a wave
// waited for every float4 separately -- 56 dependent round trips per token, 724 us for the 24576 x 4 x 1792 launch = what two waves per SIMD of
// that chain take.  The LayerNorm / aggregation vectors (shared by the four rows of a workgroup) are staged in LDS once; the dropout keep mask
// is applied to z in place and kept as one bit per element for the final multiply.
// r06 (PG = true): the PARAMETER gradients of the tail are formed here too.  They are column sums over all (mode, token) rows of quantities this kernel already holds
// in registers -- dlnw = sum dzn * zhat, dlnb = sum dzn, dwa = sum ds * zn with dzn = pr dY + ds wa -- and used to be a pass of their own
// (modes_aggr_pgrad_stage1_v4) that read Z and dY AGAIN and regenerated the dropout mask: 0.79 ms of the cfg2 step for 0.9 GB it had just seen.  Now a wave walks
// `tpw` tokens (the four waves of a workgroup on four consecutive rows at a time, as before), adds each token's three column terms -- summed over the modes in
// registers, one float4 column group at a time -- into its own [3][F] LDS vectors (a lane only ever touches its own slots: no atomics, a fixed order), and at the
// end the workgroup adds its four waves' vectors in wave order and writes ONE chunk of the column-reduction workspace; colreduce_stage2 finishes as before.
template <int NV4, int MO, bool PG>
__global__ __launch_bounds__(256) void modes_aggr_bwd_kernel(const float* __restrict__ dY, const float* __restrict__ Z, const float* __restrict__ lnw,
                                                             const float* __restrict__ lnb, const float* __restrict__ wa, const float* __restrict__ stats,
                                                             float* __restrict__ dZ, float* __restrict__ dscore, int64_t R, int F,
                                                             float p, uint64_t seed, uint64_t off, const uint64_t* __restrict__ rbase,
                                                             float* __restrict__ pws, int tpw) {
    __shared__ float4 sw[NV4 * 64], sb[NV4 * 64], sa[NV4 * 64];
    __shared__ float4 pacc[PG ? 4 * 3 * NV4 * 64 : 1];
    off += rbase ? *rbase : 0;              // device-side step base of the Philox stream (segx_set_rng_base): replayed graphs draw fresh masks
    const int lane = threadIdx.x & 63, F4 = F >> 2;
    const float ik = p > 0.f ? 1.0f / (1.0f - p) : 1.0f;
    for (int idx = threadIdx.x; idx < NV4 * 64; idx += 256) {
        const int c4 = idx < F4 ? idx : F4 - 1;
        sw[idx] = lnw ? reinterpret_cast<const float4*>(lnw)[c4] : make_float4(1.f, 1.f, 1.f, 1.f);
        sb[idx] = lnw ? reinterpret_cast<const float4*>(lnb)[c4] : make_float4(0.f, 0.f, 0.f, 0.f);
        sa[idx] = reinterpret_cast<const float4*>(wa)[c4];
    }
    float4* const myacc = pacc + (PG ? (threadIdx.x >> 6) * 3 * NV4 * 64 : 0);
    if (PG) {
#pragma unroll
        for (int i = 0; i < 3 * NV4; ++i) myacc[lane + 64 * i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    __syncthreads();
  for (int it = 0; it < tpw; ++it) {
    const int64_t row_id = ((int64_t)blockIdx.x * tpw + it) * ROWS_PER_BLOCK + (threadIdx.x >> 6);
    const bool live = row_id < R;                                    // wave-uniform; a dead wave of the last workgroup works on row R - 1 and stores nothing
    const int64_t row = live ? row_id : R - 1;
    Row<NV4> g, z[MO];
    float mean[MO], rstd[MO], pr[MO], dp[MO];
#pragma unroll
    for (int i = 0; i < NV4; ++i) { const int c4 = lane + 64 * i; g.v[i] = reinterpret_cast<const float4*>(dY + row * F)[c4 < F4 ? c4 : F4 - 1]; }
#pragma unroll
    for (int m = 0; m < MO; ++m) {
        const int64_t mr = (int64_t)m * R + row;
        mean[m] = stats[mr]; rstd[m] = stats[(int64_t)MO * R + mr]; pr[m] = stats[(int64_t)2 * MO * R + mr];
#pragma unroll
        for (int i = 0; i < NV4; ++i) { const int c4 = lane + 64 * i; z[m].v[i] = reinterpret_cast<const float4*>(Z + mr * F)[c4 < F4 ? c4 : F4 - 1]; }
    }
    __builtin_amdgcn_sched_barrier(0);
    unsigned kbits[MO];
    // pass 1: z -> zhat = (z * keep - mean) * rstd in place (0 outside the row);  dp_m = dY . (zhat * w + b)
#pragma unroll
    for (int m = 0; m < MO; ++m) {
        const int64_t mr = (int64_t)m * R + row;
        float s = 0.f;
        unsigned kb = 0xFFFFFFFFu;
#pragma unroll
        for (int i = 0; i < NV4; ++i) {
            const int c4 = lane + 64 * i;
            const bool ok = c4 < F4;
            float4 k = make_float4(1.f, 1.f, 1.f, 1.f);
            if (p > 0.f) {
                k = f4_keep(seed, off, (uint64_t)mr * F + (ok ? c4 : 0) * 4, p, ik);
                if (NV4 <= 8) kb = (kb & ~(0xFu << (4 * i))) | ((k.x != 0.f ? 1u : 0u) | (k.y != 0.f ? 2u : 0u) | (k.z != 0.f ? 4u : 0u) | (k.w != 0.f ? 8u : 0u)) << (4 * i);
            }
            if (!ok) g.v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            const float4 ww = sw[c4], bb = sb[c4];
            float4 zh;
            zh.x = (z[m].v[i].x * k.x - mean[m]) * rstd[m]; zh.y = (z[m].v[i].y * k.y - mean[m]) * rstd[m];
            zh.z = (z[m].v[i].z * k.z - mean[m]) * rstd[m]; zh.w = (z[m].v[i].w * k.w - mean[m]) * rstd[m];
            if (!ok) zh = make_float4(0.f, 0.f, 0.f, 0.f);
            z[m].v[i] = zh;
            s += ok ? ((zh.x * ww.x + bb.x) * g.v[i].x + (zh.y * ww.y + bb.y) * g.v[i].y) + ((zh.z * ww.z + bb.z) * g.v[i].z + (zh.w * ww.w + bb.w) * g.v[i].w) : 0.f;
        }
        kbits[m] = kb;
        dp[m] = wave_sum(s);
        __builtin_amdgcn_sched_barrier(0);
    }
    float dot = 0.f;
#pragma unroll
    for (int m = 0; m < MO; ++m) dot += pr[m] * dp[m];
    if (PG && live) {
        // this token's terms of the three parameter gradients (z holds zhat, 0 outside the row; columns beyond F collect garbage that is never written out)
#pragma unroll
        for (int i = 0; i < NV4; ++i) {
            const int c4 = lane + 64 * i;
            const float4 ww = sw[c4], bb = sb[c4], aa = sa[c4];
            float4 t0 = make_float4(0.f, 0.f, 0.f, 0.f), t1 = t0, t2 = t0;
#pragma unroll
            for (int m = 0; m < MO; ++m) {
                const float dsm = pr[m] * (dp[m] - dot);
                const float4 zh = z[m].v[i];
#define SEGX_PGT(E) { const float dzn = pr[m] * g.v[i].E + dsm * aa.E; t0.E += dzn * zh.E; t1.E += dzn; t2.E += dsm * (zh.E * ww.E + bb.E); }
                SEGX_PGT(x) SEGX_PGT(y) SEGX_PGT(z) SEGX_PGT(w)
#undef SEGX_PGT
            }
            float4 a0 = myacc[c4], a1 = myacc[NV4 * 64 + c4], a2 = myacc[2 * NV4 * 64 + c4];
            SEGX_F4_OP(a0, a0.x + t0.x, a0.y + t0.y, a0.z + t0.z, a0.w + t0.w);
            SEGX_F4_OP(a1, a1.x + t1.x, a1.y + t1.y, a1.z + t1.z, a1.w + t1.w);
            SEGX_F4_OP(a2, a2.x + t2.x, a2.y + t2.y, a2.z + t2.z, a2.w + t2.w);
            myacc[c4] = a0; myacc[NV4 * 64 + c4] = a1; myacc[2 * NV4 * 64 + c4] = a2;
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    // pass 2: per mode, dzn = pr*dY + ds*wa -> LN backward -> dropout backward
#pragma unroll
    for (int m = 0; m < MO; ++m) {
        const int64_t mr = (int64_t)m * R + row;
        const float ds = pr[m] * (dp[m] - dot);
        if (lane == 0 && live) dscore[mr] = ds;
        Row<NV4> d;
#pragma unroll
        for (int i = 0; i < NV4; ++i) {
            const int c4 = lane + 64 * i;
            const float4 ww = sw[c4], aa = sa[c4];
            SEGX_F4_OP(d.v[i], (pr[m] * g.v[i].x + ds * aa.x) * ww.x, (pr[m] * g.v[i].y + ds * aa.y) * ww.y,
                       (pr[m] * g.v[i].z + ds * aa.z) * ww.z, (pr[m] * g.v[i].w + ds * aa.w) * ww.w);
            if (!(c4 < F4)) d.v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        if (lnw) ln_bwd_row(d, z[m], F, rstd[m]);
        if (p > 0.f) {
#pragma unroll
            for (int i = 0; i < NV4; ++i) {
                float4 k;
                if (NV4 <= 8) {
                    const unsigned q = kbits[m] >> (4 * i);
                    k = make_float4((q & 1u) ? ik : 0.f, (q & 2u) ? ik : 0.f, (q & 4u) ? ik : 0.f, (q & 8u) ? ik : 0.f);
                } else k = f4_keep(seed, off, (uint64_t)mr * F + (lane + 64 * i < F4 ? lane + 64 * i : 0) * 4, p, ik);
                SEGX_F4_OP(d.v[i], d.v[i].x * k.x, d.v[i].y * k.y, d.v[i].z * k.z, d.v[i].w * k.w);
            }
        }
        if (live) row_store(d, dZ + mr * F, F);
        __builtin_amdgcn_sched_barrier(0);
    }
  }
    if (PG) {
        __syncthreads();
        const int64_t nchunks = gridDim.x;
        for (int idx = threadIdx.x; idx < F4; idx += 256) {
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const float4 v0 = pacc[(0 * 3 + k) * NV4 * 64 + idx], v1 = pacc[(1 * 3 + k) * NV4 * 64 + idx], v2 = pacc[(2 * 3 + k) * NV4 * 64 + idx],
                             v3 = pacc[(3 * 3 + k) * NV4 * 64 + idx];
                float4 o;
                SEGX_F4_OP(o, ((v0.x + v1.x) + v2.x) + v3.x, ((v0.y + v1.y) + v2.y) + v3.y, ((v0.z + v1.z) + v2.z) + v3.z, ((v0.w + v1.w) + v2.w) + v3.w);
                reinterpret_cast<float4*>(pws + ((int64_t)k * nchunks + blockIdx.x) * F)[idx] = o;
            }
        }
    }
}

// parameter gradients of the expansion tail: thread = column f, block.y = row chunk.
//   dlnw = sum dzn * zhat ; dlnb = sum dzn ; dwa = sum ds * zn       with dzn = pr*dY + ds*wa
__global__ __launch_bounds__(256) void modes_aggr_pgrad_stage1(const float* __restrict__ dY, const float* __restrict__ Z, const float* __restrict__ lnw,
                                                               const float* __restrict__ lnb, const float* __restrict__ wa, const float* __restrict__ stats,
                                                               const float* __restrict__ dscore, float* __restrict__ ws, int Mo, int64_t R, int F,
                                                               int nchunks, float p, uint64_t seed, uint64_t off, const uint64_t* __restrict__ rbase) {
    off += rbase ? *rbase : 0;              // device-side step base of the Philox stream (segx_set_rng_base): replayed graphs draw fresh masks
    const int c = blockIdx.x * 256 + threadIdx.x;
    const int chunk = blockIdx.y;
    const int64_t per = (R + nchunks - 1) / nchunks, r0 = chunk * per, r1 = i64min(R, r0 + per);
    if (c >= F) return;
    const float ik = p > 0.f ? 1.0f / (1.0f - p) : 1.0f;
    const float w = lnw ? lnw[c] : 1.f, b = lnw ? lnb[c] : 0.f, a = wa[c];      // lnw == NULL: no LayerNorm (stats hold mean 0, rstd 1)
    float s0 = 0.f, s1 = 0.f, s2 = 0.f;
    for (int m = 0; m < Mo; ++m)
        for (int64_t r = r0; r < r1; ++r) {
            const int64_t mr = (int64_t)m * R + r;
            float z = Z[mr * F + c];
            if (p > 0.f) z *= dropout_scale(seed, off, (uint64_t)mr * F + c, p, ik);
            const float zh = (z - stats[mr]) * stats[(int64_t)Mo * R + mr];
            const float ds = dscore[mr];
            const float dzn = stats[(int64_t)2 * Mo * R + mr] * dY[r * F + c] + ds * a;
            s0 += dzn * zh; s1 += dzn; s2 += ds * (zh * w + b);
        }
    ws[(int64_t)chunk * F + c] = s0;
    ws[((int64_t)nchunks + chunk) * F + c] = s1;
    ws[((int64_t)2 * nchunks + chunk) * F + c] = s2;
}

// float4 form (F % 4 == 0): a thread owns FOUR adjacent columns -- 16-byte loads, and ONE Philox counter per four elements instead of four (the scalar form
// spent most of its time regenerating the dropout mask: 182 us per call at 2.3 TB/s on the cfg2 shapes); same sums in the same order per column
__global__ __launch_bounds__(256) void modes_aggr_pgrad_stage1_v4(const float* __restrict__ dY, const float* __restrict__ Z, const float* __restrict__ lnw,
                                                                  const float* __restrict__ lnb, const float* __restrict__ wa, const float* __restrict__ stats,
                                                                  const float* __restrict__ dscore, float* __restrict__ ws, int Mo, int64_t R, int F,
                                                                  int nchunks, float p, uint64_t seed, uint64_t off, const uint64_t* __restrict__ rbase) {
    off += rbase ? *rbase : 0;
    const int c = 4 * (blockIdx.x * 256 + threadIdx.x);
    const int chunk = blockIdx.y;
    const int64_t per = (R + nchunks - 1) / nchunks, r0 = chunk * per, r1 = i64min(R, r0 + per);
    if (c >= F) return;
    const float ik = p > 0.f ? 1.0f / (1.0f - p) : 1.0f;
    const float4 one = make_float4(1.f, 1.f, 1.f, 1.f), zero = make_float4(0.f, 0.f, 0.f, 0.f);
    const float4 w = lnw ? *reinterpret_cast<const float4*>(lnw + c) : one, b = lnw ? *reinterpret_cast<const float4*>(lnb + c) : zero;
    const float4 a = *reinterpret_cast<const float4*>(wa + c);
    float4 s0 = zero, s1 = zero, s2 = zero;
    for (int m = 0; m < Mo; ++m)
#pragma unroll 4
        for (int64_t r = r0; r < r1; ++r) {
            const int64_t mr = (int64_t)m * R + r;
            float4 z = *reinterpret_cast<const float4*>(Z + mr * F + c);
            if (p > 0.f) { const float4 k = f4_keep(seed, off, (uint64_t)mr * F + c, p, ik); z.x *= k.x; z.y *= k.y; z.z *= k.z; z.w *= k.w; }
            const float mu = stats[mr], rs = stats[(int64_t)Mo * R + mr], pr = stats[(int64_t)2 * Mo * R + mr], ds = dscore[mr];
            const float4 g = *reinterpret_cast<const float4*>(dY + r * F + c);
#define SEGX_PG(E)                                                                   \
            { const float zh = (z.E - mu) * rs, dzn = pr * g.E + ds * a.E;             \
              s0.E += dzn * zh; s1.E += dzn; s2.E += ds * (zh * w.E + b.E); }
            SEGX_PG(x) SEGX_PG(y) SEGX_PG(z) SEGX_PG(w)
#undef SEGX_PG
        }
    *reinterpret_cast<float4*>(ws + (int64_t)chunk * F + c) = s0;
    *reinterpret_cast<float4*>(ws + ((int64_t)nchunks + chunk) * F + c) = s1;
    *reinterpret_cast<float4*>(ws + ((int64_t)2 * nchunks + chunk) * F + c) = s2;
}

// =================================================================================================
// GELU backward (+ dropout of MMSharedMid :244-245): dT = dH * keep * gelu'(T)
// =================================================================================================
__global__ __launch_bounds__(256) void gelu_bwd_kernel(const float* __restrict__ dH, const float* __restrict__ T, float* __restrict__ dT, int64_t n4,
                                                       float p, uint64_t seed, uint64_t off, const uint64_t* __restrict__ rbase) {
    off += rbase ? *rbase : 0;              // device-side step base of the Philox stream (segx_set_rng_base): replayed graphs draw fresh masks
    const float ik = p > 0.f ? 1.0f / (1.0f - p) : 1.0f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        const float4 g = reinterpret_cast<const float4*>(dH)[i], t = reinterpret_cast<const float4*>(T)[i];
        float4 k = make_float4(1.f, 1.f, 1.f, 1.f);
        if (p > 0.f) k = f4_keep(seed, off, (uint64_t)i * 4, p, ik);
        float4 o;
        SEGX_F4_OP(o, g.x * k.x * gelu_erf_grad(t.x), g.y * k.y * gelu_erf_grad(t.y), g.z * k.z * gelu_erf_grad(t.z), g.w * k.w * gelu_erf_grad(t.w));
        reinterpret_cast<float4*>(dT)[i] = o;
    }
}

// The same with the COLUMN SUMS of dT collected on the way (r06): dT [rows, N] is the output gradient of the nn.Linear in front of the GELU (MMSharedMid :244), its
// column sums are that layer's bias gradient -- which used to be a pass of its own over the 0.7 GB this kernel has just written (colreduce_stage1_v4, 0.41 ms of the
// cfg2 step).  thread = four adjacent columns, block.y = a chunk of rows (the layout of colreduce_stage1_v4: same chunk partials, same second stage).
__global__ __launch_bounds__(512) void gelu_bwd_colsum_kernel(const float* __restrict__ dH, const float* __restrict__ T, float* __restrict__ dT, float* __restrict__ ws,
                                                              int64_t rows, int N, int nchunks, float p, uint64_t seed, uint64_t off, const uint64_t* __restrict__ rbase) {
    // a workgroup covers WHOLE rows (N / 4 <= 512 threads, one float4 column group each) of a contiguous chunk of rows: it streams one contiguous piece of
    // dH / T / dT like the flat kernel does (the first form -- 256 threads x 4 columns, rows 7 KB apart -- moved the same bytes at 4.7 TB/s against the flat
    // kernel's 7.9: r06_v)
    off += rbase ? *rbase : 0;
    const int c = 4 * threadIdx.x;
    const int chunk = blockIdx.x;
    const int64_t per = (rows + nchunks - 1) / nchunks, r0 = chunk * per, r1 = i64min(rows, r0 + per);
    if (c >= N) return;
    const float ik = p > 0.f ? 1.0f / (1.0f - p) : 1.0f;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 4
    for (int64_t r = r0; r < r1; ++r) {
        const int64_t e = r * N + c;
        const float4 g = *reinterpret_cast<const float4*>(dH + e), t = *reinterpret_cast<const float4*>(T + e);
        float4 k = make_float4(1.f, 1.f, 1.f, 1.f);
        if (p > 0.f) k = f4_keep(seed, off, (uint64_t)e, p, ik);
        float4 o;
        SEGX_F4_OP(o, g.x * k.x * gelu_erf_grad(t.x), g.y * k.y * gelu_erf_grad(t.y), g.z * k.z * gelu_erf_grad(t.z), g.w * k.w * gelu_erf_grad(t.w));
        *reinterpret_cast<float4*>(dT + e) = o;
        s.x += o.x; s.y += o.y; s.z += o.z; s.w += o.w;
    }
    *reinterpret_cast<float4*>(ws + (int64_t)chunk * N + c) = s;
}

}  // namespace segx

using namespace segx;
#define SEGX_STREAM hipStream_t stream = (hipStream_t)stream_
#define SEGX_ROWCHK(F) SEGX_REQUIRE((F) > 0 && (F) % 4 == 0, "row width %d must be a positive multiple of 4", (int)(F))

extern "C" int segx_softmax_fwd(const float* S, float* P, float* Pdrop, int64_t rows, int L, float clip, const float* gmax,
                                float p, uint64_t seed, uint64_t offset, void* stream_) {
    SEGX_STREAM; SEGX_REQUIRE(S && P && rows > 0 && L > 0, "segx_softmax_fwd: bad args");
    SEGX_REQUIRE(p >= 0.f && p < 1.f && (p == 0.f || Pdrop), "segx_softmax_fwd: dropout needs Pdrop");
    if (L % 4 != 0 || L > 4096) {
        hipLaunchKernelGGL(softmax_fwd_generic_kernel, dim3((unsigned)i64min(rows, 65536)), dim3(256), 0, stream, S, P, p > 0.f ? Pdrop : nullptr, rows, L, clip, gmax, p, seed, offset, rng_base());
        return check_launch("segx_softmax_fwd");
    }
    SEGX_DISPATCH_NV4(L, hipLaunchKernelGGL((softmax_fwd_kernel<NV4>), row_grid(rows), dim3(256), 0, stream, S, P, p > 0.f ? Pdrop : nullptr, rows, L, clip, gmax, p, seed, offset, rng_base()));
    return check_launch("segx_softmax_fwd");
}
extern "C" int segx_softmax_bwd(const float* P, const float* dPdrop, const float* S, float* dS, int64_t rows, int L, float clip,
                                const float* gmax, float p, uint64_t seed, uint64_t offset, void* stream_) {
    SEGX_STREAM; SEGX_REQUIRE(P && dPdrop && dS && rows > 0 && L > 0, "segx_softmax_bwd: bad args");
    if (L % 4 != 0 || L > 4096) {
        hipLaunchKernelGGL(softmax_bwd_generic_kernel, dim3((unsigned)i64min(rows, 65536)), dim3(256), 0, stream, P, dPdrop, S, dS, rows, L, clip, gmax, p, seed, offset, rng_base());
        return check_launch("segx_softmax_bwd");
    }
    SEGX_DISPATCH_NV4(L, hipLaunchKernelGGL((softmax_bwd_kernel<NV4>), row_grid(rows), dim3(256), 0, stream, P, dPdrop, S, dS, rows, L, clip, gmax, p, seed, offset, rng_base()));
    return check_launch("segx_softmax_bwd");
}
extern "C" int segx_layernorm_fwd(const float* X, const float* w, const float* b, float* Y, float* mean, float* rstd,
                                  int64_t rows, int C, float eps, void* stream_) {
    SEGX_STREAM; SEGX_REQUIRE(X && Y && mean && rstd && rows > 0 && (!w == !b), "segx_layernorm_fwd: bad args"); SEGX_ROWCHK(C);
    SEGX_DISPATCH_NV4(C, hipLaunchKernelGGL((layernorm_fwd_kernel<NV4>), row_grid(rows), dim3(256), 0, stream, X, w, b, Y, mean, rstd, rows, C, eps));
    return check_launch("segx_layernorm_fwd");
}
extern "C" int segx_layernorm_bwd(const float* dY, const float* X, const float* w, const float* mean, const float* rstd, float* dX,
                                  int64_t rows, int C, void* stream_) {
    SEGX_STREAM; SEGX_REQUIRE(dY && X && mean && rstd && dX && rows > 0, "segx_layernorm_bwd: bad args"); SEGX_ROWCHK(C);
    SEGX_DISPATCH_NV4(C, hipLaunchKernelGGL((layernorm_bwd_kernel<NV4>), row_grid(rows), dim3(256), 0, stream, dY, X, w, mean, rstd, dX, rows, C));
    return check_launch("segx_layernorm_bwd");
}
extern "C" int64_t segx_colreduce_ws_floats(int64_t rows, int64_t C, int nout) { return (int64_t)nout * chunks_for(rows) * C; }
// stage 1 of a column reduction: the four-column form for wide, 16-byte aligned tensors, the column-per-thread form otherwise
static void launch_colreduce1(hipStream_t stream, const float* A, const float* X, const float* mean, const float* rstd, float* ws, int64_t rows, int64_t C, int mode, int nch) {
    const bool v4 = C >= 1024 && C % 4 == 0 && ((reinterpret_cast<uintptr_t>(A) | reinterpret_cast<uintptr_t>(X) | reinterpret_cast<uintptr_t>(ws)) & 15) == 0;
    if (v4) hipLaunchKernelGGL(colreduce_stage1_v4, dim3((unsigned)((C / 4 + 255) / 256), nch), dim3(256), 0, stream, A, X, mean, rstd, ws, rows, C, mode, nch);
    else hipLaunchKernelGGL(colreduce_stage1, dim3((unsigned)((C + 255) / 256), nch), dim3(256), 0, stream, A, X, mean, rstd, ws, rows, C, mode, nch);
}
extern "C" int segx_colsum(const float* X, float* out, float* ws, int64_t rows, int64_t C, void* stream_) {
    SEGX_STREAM; SEGX_REQUIRE(X && out && ws && rows > 0 && C > 0, "segx_colsum: bad args");
    if (rows <= 64) {          // few rows (per-sample partials, depthwise weight-gradient chunks): one pass straight into `out`, no second launch
        launch_colreduce1(stream, X, nullptr, nullptr, nullptr, out, rows, C, 0, 1);
        return check_launch("segx_colsum");
    }
    const int nch = chunks_for(rows);
    launch_colreduce1(stream, X, nullptr, nullptr, nullptr, ws, rows, C, 0, nch);
    hipLaunchKernelGGL(colreduce_stage2, dim3((unsigned)((C + 63) / 64)), dim3(256), 0, stream, (const float*)ws, out, (float*)nullptr, (float*)nullptr, C, nch, 1);
    return check_launch("segx_colsum");
}
extern "C" int segx_ln_param_grad(const float* dY, const float* X, const float* mean, const float* rstd, float* dw, float* db, float* ws,
                                  int64_t rows, int C, void* stream_) {
    SEGX_STREAM; SEGX_REQUIRE(dY && X && mean && rstd && dw && db && ws && rows > 0 && C > 0, "segx_ln_param_grad: bad args");
    const int nch = chunks_for(rows);
    launch_colreduce1(stream, dY, X, mean, rstd, ws, rows, (int64_t)C, 1, nch);
    hipLaunchKernelGGL(colreduce_stage2, dim3((unsigned)((C + 63) / 64)), dim3(256), 0, stream, (const float*)ws, dw, db, (float*)nullptr, (int64_t)C, nch, 2);
    return check_launch("segx_ln_param_grad");
}
extern "C" int segx_rowsum(const float* X, float* out, int64_t R, int64_t S, void* stream_) {
    SEGX_STREAM; SEGX_REQUIRE(X && out && R > 0 && S > 0 && R < 2147483647LL, "segx_rowsum: bad args");
    const int vec = (S % 4 == 0) && ((reinterpret_cast<uintptr_t>(X) & 15) == 0);
    hipLaunchKernelGGL(rowsum_kernel, dim3((unsigned)R), dim3(256), 0, stream, X, out, S, vec);
    return check_launch("segx_rowsum");
}
extern "C" int segx_sum(const float* x, int64_t n, float* out, float* ws /* >= 1024 floats */, float scale, void* stream_) {
    SEGX_STREAM; SEGX_REQUIRE(x && out && ws && n > 0, "segx_sum: bad args");
    const int nb = (int)i64min(1024, (n + 255) / 256);
    hipLaunchKernelGGL(sum_stage1, dim3(nb), dim3(256), 0, stream, x, n, ws);
    hipLaunchKernelGGL(sum_stage2, dim3(1), dim3(256), 0, stream, (const float*)ws, nb, out, scale);
    return check_launch("segx_sum");
}
extern "C" int segx_prenorm_fwd(const float* X, const float* w1, const float* b1, const float* pos, int64_t pos_ld, float pos_weight,
                                const float* mask, float* Y, float* stats, int64_t B, int N, int C, float eps,
                                float p, uint64_t seed, uint64_t offset, void* stream_) {
    SEGX_STREAM; SEGX_REQUIRE(X && w1 && b1 && mask && Y && stats && B > 0 && N > 0, "segx_prenorm_fwd: bad args"); SEGX_ROWCHK(C);
    SEGX_REQUIRE(!pos || (pos_ld >= C && pos_ld % 4 == 0), "segx_prenorm_fwd: pos_ld %lld", (long long)pos_ld);
    const int64_t rows = B * N;
    SEGX_DISPATCH_NV4(C, hipLaunchKernelGGL((prenorm_fwd_kernel<NV4>), row_grid(rows), dim3(256), 0, stream, X, w1, b1, pos, pos_ld, pos_weight, mask, Y, stats, rows, N, C, eps, p, seed, offset, rng_base()));
    return check_launch("segx_prenorm_fwd");
}
extern "C" int segx_prenorm_bwd(const float* dY, const float* X, const float* w1, const float* b1, const float* pos, int64_t pos_ld, float pos_weight,
                                const float* mask, const float* stats, float* dX, float* dU, int64_t B, int N, int C,
                                float p, uint64_t seed, uint64_t offset, void* stream_) {
    SEGX_STREAM; SEGX_REQUIRE(dY && X && w1 && b1 && mask && stats && dX && dU && B > 0 && N > 0, "segx_prenorm_bwd: bad args"); SEGX_ROWCHK(C);
    const int64_t rows = B * N;
    SEGX_DISPATCH_NV4(C, hipLaunchKernelGGL((prenorm_bwd_kernel<NV4>), row_grid(rows), dim3(256), 0, stream, dY, X, w1, b1, pos, pos_ld, pos_weight, mask, stats, dX, dU, rows, N, C, p, seed, offset, rng_base()));
    return check_launch("segx_prenorm_bwd");
}
extern "C" int64_t segx_prenorm_bwd_all_ws_floats(int N, int C) { return 2 * (int64_t)((N + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK) * C; }
extern "C" int segx_prenorm_bwd_all(const float* dY, const float* X, const float* w1, const float* b1, const float* pos, int64_t pos_ld, float pos_weight,
                                    const float* mask, const float* stats, float* dX, float* dsum, float* dw, float* db, float* ws, int B, int N, int C,
                                    float p, uint64_t seed, uint64_t offset, void* stream_) {
    SEGX_STREAM; SEGX_REQUIRE(dY && X && w1 && b1 && mask && stats && dX && dw && db && ws && B > 0 && N > 0 && (!pos || dsum), "segx_prenorm_bwd_all: bad args"); SEGX_ROWCHK(C);
    SEGX_REQUIRE(C <= 2048 && (!pos || (pos_ld >= C && pos_ld % 4 == 0)), "segx_prenorm_bwd_all: C=%d (<= 2048) pos_ld=%lld", C, (long long)pos_ld);
    const int nch = (N + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK;
    SEGX_DISPATCH_NV4(C, hipLaunchKernelGGL((prenorm_bwd_all_kernel<(NV4 > 8 ? 8 : NV4)>), dim3((unsigned)nch), dim3(256), 0, stream, dY, X, w1, b1, pos, pos_ld, pos_weight, mask, stats, dX,
                                            pos ? dsum : (float*)nullptr, ws, B, N, C, p, seed, offset, rng_base()));
    launch_colreduce2(stream, ws, dw, db, nullptr, (int64_t)C, nch, 2);
    return check_launch("segx_prenorm_bwd_all");
}
extern "C" int segx_posembed_fwd(const float* posn, const float* Wp, const float* bp, float* out, float* stats, int64_t N, int C, int pd,
                                 float eps, void* stream_) {
    SEGX_STREAM; SEGX_REQUIRE(posn && Wp && bp && out && stats && N > 0 && pd >= 1 && pd <= 4, "segx_posembed_fwd: bad args"); SEGX_ROWCHK(C);
    SEGX_DISPATCH_NV4(C, hipLaunchKernelGGL((posembed_fwd_kernel<NV4>), row_grid(N), dim3(256), 0, stream, posn, Wp, bp, out, stats, N, C, pd, eps));
    return check_launch("segx_posembed_fwd");
}
extern "C" int segx_posembed_bwd(const float* dOut, const float* posn, const float* Wp, const float* bp, const float* stats, float* dZ,
                                 int64_t N, int C, int pd, void* stream_) {
    SEGX_STREAM; SEGX_REQUIRE(dOut && posn && Wp && bp && stats && dZ && N > 0 && pd >= 1 && pd <= 4, "segx_posembed_bwd: bad args"); SEGX_ROWCHK(C);
    SEGX_DISPATCH_NV4(C, hipLaunchKernelGGL((posembed_bwd_kernel<NV4>), row_grid(N), dim3(256), 0, stream, dOut, posn, Wp, bp, stats, dZ, N, C, pd));
    return check_launch("segx_posembed_bwd");
}
#define SEGX_DISPATCH_MO(Mo, CALL4, CALL1) if ((Mo) == 4) { constexpr int MO = 4; CALL4; } else if ((Mo) == 1) { constexpr int MO = 1; CALL1; } \
    else return segx::fail(-1, "num_modes %d unsupported (1 or 4)", (int)(Mo));
extern "C" int segx_modes_aggr_fwd(const float* Z, const float* lnw, const float* lnb, const float* wa, const float* ba, float* Y, float* stats,
                                   int Mo, int64_t R, int F, float eps, float p, uint64_t seed, uint64_t offset, void* stream_) {
    SEGX_STREAM; SEGX_REQUIRE(Z && (!lnw == !lnb) && wa && ba && Y && stats && R > 0, "segx_modes_aggr_fwd: bad args"); SEGX_ROWCHK(F);
    SEGX_REQUIRE(F <= 2048 || Mo == 1, "segx_modes_aggr_fwd: F=%d too wide for the register-resident 4-mode kernel", F);
    SEGX_DISPATCH_NV4(F, SEGX_DISPATCH_MO(Mo,
        hipLaunchKernelGGL((modes_aggr_fwd_kernel<(NV4 > 8 ? 8 : NV4), 4>), row_grid(R), dim3(256), 0, stream, Z, lnw, lnb, wa, ba, Y, stats, R, F, eps, p, seed, offset, rng_base()),
        hipLaunchKernelGGL((modes_aggr_fwd_kernel<NV4, 1>), row_grid(R), dim3(256), 0, stream, Z, lnw, lnb, wa, ba, Y, stats, R, F, eps, p, seed, offset, rng_base())));
    return check_launch("segx_modes_aggr_fwd");
}
extern "C" int segx_modes_aggr_bwd(const float* dY, const float* Z, const float* lnw, const float* lnb, const float* wa, const float* stats,
                                   float* dZ, float* dscore, int Mo, int64_t R, int F, float p, uint64_t seed, uint64_t offset, void* stream_) {
    SEGX_STREAM; SEGX_REQUIRE(dY && Z && (!lnw == !lnb) && wa && stats && dZ && dscore && R > 0, "segx_modes_aggr_bwd: bad args"); SEGX_ROWCHK(F);
    SEGX_REQUIRE(F <= 2048 || Mo == 1, "segx_modes_aggr_bwd: F=%d too wide for the register-resident 4-mode kernel", F);
    SEGX_DISPATCH_NV4(F, SEGX_DISPATCH_MO(Mo,
        hipLaunchKernelGGL((modes_aggr_bwd_kernel<(NV4 > 8 ? 8 : NV4), 4, false>), row_grid(R), dim3(256), 0, stream, dY, Z, lnw, lnb, wa, stats, dZ, dscore, R, F, p, seed, offset, rng_base(), (float*)nullptr, 1),
        hipLaunchKernelGGL((modes_aggr_bwd_kernel<NV4, 1, false>), row_grid(R), dim3(256), 0, stream, dY, Z, lnw, lnb, wa, stats, dZ, dscore, R, F, p, seed, offset, rng_base(), (float*)nullptr, 1)));
    return check_launch("segx_modes_aggr_bwd");
}
// tokens a wave walks in the fused form: four where that still leaves two rounds of workgroups on the chip, fewer for short token lists
static inline int modes_aggr_tpw(int64_t R) { return R >= 4096 * 4 ? 4 : R >= 2048 * 2 ? 2 : 1; }
static inline bool modes_aggr_fused_ok(int Mo, int F) { return Mo == 4 && F % 4 == 0 && F <= 2048; }
extern "C" int64_t segx_modes_aggr_bwd_all_ws_floats(int Mo, int64_t R, int F) {
    if (!modes_aggr_fused_ok(Mo, F)) return segx_colreduce_ws_floats(R, F, 3);
    const int tpw = modes_aggr_tpw(R);
    return 3 * ((R + 4 * tpw - 1) / (4 * tpw)) * (int64_t)F;
}
extern "C" int segx_modes_aggr_bwd_all(const float* dY, const float* Z, const float* lnw, const float* lnb, const float* wa, const float* stats,
                                       float* dZ, float* dscore, float* dlnw, float* dlnb, float* dwa, float* ws, int Mo, int64_t R, int F,
                                       float p, uint64_t seed, uint64_t offset, void* stream_) {
    SEGX_STREAM; SEGX_REQUIRE(dY && Z && (!lnw == !lnb) && wa && stats && dZ && dscore && dlnw && dlnb && dwa && ws && R > 0, "segx_modes_aggr_bwd_all: bad args"); SEGX_ROWCHK(F);
    if (!modes_aggr_fused_ok(Mo, F)) {                      // one mode / wide rows: the two passes of before
        int rc = segx_modes_aggr_bwd(dY, Z, lnw, lnb, wa, stats, dZ, dscore, Mo, R, F, p, seed, offset, stream_);
        if (rc) return rc;
        return segx_modes_aggr_param_grad(dY, Z, lnw, lnb, wa, stats, dscore, dlnw, dlnb, dwa, ws, Mo, R, F, p, seed, offset, stream_);
    }
    const int tpw = modes_aggr_tpw(R);
    const int64_t nch = (R + 4 * tpw - 1) / (4 * tpw);
    SEGX_REQUIRE(nch < 2147483647LL, "segx_modes_aggr_bwd_all: too many rows");
    SEGX_DISPATCH_NV4(F, hipLaunchKernelGGL((modes_aggr_bwd_kernel<(NV4 > 8 ? 8 : NV4), 4, true>), dim3((unsigned)nch), dim3(256), 0, stream, dY, Z, lnw, lnb, wa, stats, dZ, dscore, R, F,
                                            p, seed, offset, rng_base(), ws, tpw));
    launch_colreduce2(stream, ws, dlnw, dlnb, dwa, (int64_t)F, (int)nch, 3);
    return check_launch("segx_modes_aggr_bwd_all");
}
extern "C" int segx_modes_aggr_param_grad(const float* dY, const float* Z, const float* lnw, const float* lnb, const float* wa, const float* stats,
                                          const float* dscore, float* dlnw, float* dlnb, float* dwa, float* ws, int Mo, int64_t R, int F,
                                          float p, uint64_t seed, uint64_t offset, void* stream_) {
    SEGX_STREAM; SEGX_REQUIRE(dY && Z && (!lnw == !lnb) && wa && stats && dscore && dlnw && dlnb && dwa && ws && R > 0, "segx_modes_aggr_param_grad: bad args");
    const int nch = chunks_for(R);
    const bool v4 = F % 4 == 0 && offset % 4 == 0 && ((reinterpret_cast<uintptr_t>(dY) | reinterpret_cast<uintptr_t>(Z) | reinterpret_cast<uintptr_t>(wa) | reinterpret_cast<uintptr_t>(ws) |
                                                       reinterpret_cast<uintptr_t>(lnw) | reinterpret_cast<uintptr_t>(lnb)) & 15) == 0;
    if (v4) hipLaunchKernelGGL(modes_aggr_pgrad_stage1_v4, dim3((F / 4 + 255) / 256, nch), dim3(256), 0, stream, dY, Z, lnw, lnb, wa, stats, dscore, ws, Mo, R, F, nch, p, seed, offset, rng_base());
    else hipLaunchKernelGGL(modes_aggr_pgrad_stage1, dim3((F + 255) / 256, nch), dim3(256), 0, stream, dY, Z, lnw, lnb, wa, stats, dscore, ws, Mo, R, F, nch, p, seed, offset, rng_base());
    hipLaunchKernelGGL(colreduce_stage2, dim3((F + 63) / 64), dim3(256), 0, stream, (const float*)ws, dlnw, dlnb, dwa, (int64_t)F, nch, 3);
    return check_launch("segx_modes_aggr_param_grad");
}
extern "C" int segx_gelu_bwd(const float* dH, const float* T, float* dT, int64_t n, float p, uint64_t seed, uint64_t offset, void* stream_) {
    SEGX_STREAM; SEGX_REQUIRE(dH && T && dT && n > 0 && n % 4 == 0, "segx_gelu_bwd: n=%lld must be a positive multiple of 4", (long long)n);
    const int nb = (int)i64min(4096, (n / 4 + 255) / 256);
    hipLaunchKernelGGL(gelu_bwd_kernel, dim3(nb), dim3(256), 0, stream, dH, T, dT, n / 4, p, seed, offset, rng_base());
    return check_launch("segx_gelu_bwd");
}
static inline int gelu_colsum_chunks(int64_t rows) { return (int)i64max(1, i64min(2048, rows / 16)); }
extern "C" int64_t segx_gelu_bwd_colsum_ws_floats(int64_t rows, int N) { return (int64_t)gelu_colsum_chunks(rows) * N; }
extern "C" int segx_gelu_bwd_colsum(const float* dH, const float* T, float* dT, float* colsum, float* ws, int64_t rows, int N, float p, uint64_t seed, uint64_t offset,
                                    void* stream_) {
    SEGX_STREAM; SEGX_REQUIRE(dH && T && dT && colsum && ws && rows > 0 && N > 0 && N % 4 == 0 && N <= 2048 && offset % 4 == 0,
                              "segx_gelu_bwd_colsum: rows=%lld N=%d offset=%llu (N <= 2048; N, offset: multiples of 4)", (long long)rows, N, (unsigned long long)offset);
    SEGX_REQUIRE(((reinterpret_cast<uintptr_t>(dH) | reinterpret_cast<uintptr_t>(T) | reinterpret_cast<uintptr_t>(dT) | reinterpret_cast<uintptr_t>(ws)) & 15) == 0, "segx_gelu_bwd_colsum: 16-byte alignment");
    const int nch = gelu_colsum_chunks(rows);
    const int threads = ((N / 4 + 63) / 64) * 64;
    hipLaunchKernelGGL(gelu_bwd_colsum_kernel, dim3((unsigned)nch), dim3(threads), 0, stream, dH, T, dT, ws, rows, N, nch, p, seed, offset, rng_base());
    launch_colreduce2(stream, ws, colsum, nullptr, nullptr, (int64_t)N, nch, 1);
    return check_launch("segx_gelu_bwd_colsum");
}
/* geom = {D, H, W, R, nd}: token grid (D = 1 in 2-D), radius, position dims */
extern "C" int segx_posbias_fwd(const float* S, float* out, const float* table, int64_t nmat, int N, const int* geom, float weight, float clip,
                                const float* gmax, void* stream_) {
    SEGX_STREAM; SEGX_REQUIRE(S && out && table && geom && nmat > 0 && N > 0, "segx_posbias_fwd: bad args");
    PosGeom q{geom[0], geom[1], geom[2], geom[3], geom[4]};
    SEGX_REQUIRE(q.D * q.H * q.W == N && (q.nd == 2 || q.nd == 3) && q.R >= 0, "segx_posbias_fwd: geometry does not match N=%d", N);
    const int64_t total = nmat * N * N;
    hipLaunchKernelGGL(posbias_fwd_kernel, dim3((unsigned)i64min(65536, (total + 255) / 256)), dim3(256), 0, stream, S, out, table, nmat * N, N, q, weight, clip, gmax);
    return check_launch("segx_posbias_fwd");
}
/* dS (clamp-masked copy of dOut; pass S = NULL when the clamp is known inactive and reuse dOut) and dtable */
extern "C" int segx_posbias_bwd(const float* dOut, const float* S, float* dS, float* dtable, int64_t nmat, int N, const int* geom, float weight,
                                float clip, void* stream_) {
    SEGX_STREAM; SEGX_REQUIRE(dOut && dtable && geom && nmat > 0 && N > 0 && (!S == !dS), "segx_posbias_bwd: bad args");
    PosGeom q{geom[0], geom[1], geom[2], geom[3], geom[4]};
    SEGX_REQUIRE(q.D * q.H * q.W == N && (q.nd == 2 || q.nd == 3), "segx_posbias_bwd: geometry does not match N=%d", N);
    const int T = 2 * q.R + 1, nt = q.nd == 3 ? T * T * T : T * T;
    const int64_t total = nmat * N * N;
    if (S) hipLaunchKernelGGL(posbias_bwd_clamp_kernel, dim3((unsigned)i64min(65536, (total + 255) / 256)), dim3(256), 0, stream, dOut, S, dS, total, clip);
    hipLaunchKernelGGL(posbias_bwd_table_kernel, dim3(nt), dim3(256), 0, stream, dOut, dtable, nmat, N, q, weight);
    return check_launch("segx_posbias_bwd");
}
