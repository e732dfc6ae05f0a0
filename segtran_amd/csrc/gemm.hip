// gemm.hip -- batched strided fp32 GEMM on the gfx950 f32 matrix core (v_mfma_f32_32x32x2_f32).
//
// Why f32 MFMA: the parity bar is fp32-faithful logits (bf16 MFMA inputs flip hardened
// labels), and gfx950 has no TF32.  v_mfma_f32_32x32x2_f32 is bit-for-bit a k-ordered fmaf chain and
// peaks at 157.3 TFLOP/s -- the roofline this kernel is measured against.
//
// Tiling: workgroup = 256 threads = 4 waves (2x2); block tile 128x128, k-tile 32; each wave owns a
// 64x64 sub-tile = 2x2 MFMA tiles of 32x32 (4 x f32x16 accumulators = 64 VGPRs).  Both operand tiles are
// staged through LDS k-major ([k][m], row stride 132 floats) so that an MFMA operand fetch is one
// conflict-free ds_read_b32 per lane (lane l needs A[m0 + (l&31)][k0 + (l>>5)]).  The next k-tile's
// global loads (4 x float4 per operand per thread) are issued before the 64 MFMAs of the current tile
// and written to LDS after them, so HBM/L2 latency hides under ~4k cycles of matrix work per wave.
// One MFMA occupies its SIMD for 64 cycles, so 4 LDS reads per 4 MFMAs keep LDS traffic at a few %
// of the matrix-pipe time; >= 2 workgroups per CU cover each other's barriers.
//
// Operands are addressed through (batch0, batch1, row, k) element strides, one of (row, k) being 1:
//   K-contiguous operand -> float4 along k, transposing scalar LDS stores;
//   row-contiguous operand -> float4 along rows, float4 LDS stores.
// All four combinations (NT: linear fwd / QK^T, NN: P.V, dX = dY.W; TN: dW = dY^T.X; TT) are
// instantiated, so no operand is ever materialised transposed in HBM.
#include "common.h"

namespace segx {

constexpr int BM = 128, BN = 128, BKT = 32, LDT = 132;

struct GemmArgs {
    const float* A; const float* B; float* C;
    const float* bias; float* aux; float* gmax;
    int M, N, K, nb1;
    int64_t a_b0, a_b1, a_m, a_k;
    int64_t b_b0, b_b1, b_n, b_k;
    int64_t c_b0, c_b1, c_m;
    int64_t bias_b1;
    float alpha; int epilogue, bias_mode;
    int vecA, vecB;                 // float4 global loads legal (alignment + stride checks done on host)
    int tiles_m, tiles_n;
    float dropout_p; uint64_t seed, offset;
    int k_chunk;                    // split-K: this launch covers k in [z_k*k_chunk, min(K, (z_k+1)*k_chunk))
    int splitk; int64_t c_split;    // slab stride in the workspace
};

// Load this thread's 4 float4 pieces of a 128 x 32 operand tile into registers.
//  KC = true : operand is k-contiguous;  piece f -> row f>>3, k-chunk f&7
//  KC = false: operand is row-contiguous; piece f -> k-row f>>5, row-chunk f&31
// VEC = true (16-B aligned base, all strides and extents multiples of 4): every float4 is either wholly inside
// or wholly outside the operand, so the load is issued UNCONDITIONALLY from a clamped address and zeroed by a
// select -- no branches, so the 8 loads of a k-tile stay in flight together (a guarded load costs an exec-mask
// branch plus an s_waitcnt vmcnt(0) each).  VEC = false is the slow scalar path for odd shapes (K = 2, Cin = 6 ...).
// The zeroing select is deferred to store_tile (through the returned validity mask): consuming a loaded value
// right after the load would make the compiler wait for it BEFORE the MFMA block and lose the overlap.
template <bool KC, bool VEC>
__device__ __forceinline__ unsigned load_tile(float4 (&r)[4], const float* __restrict__ base, int64_t s_row, int64_t s_k,
                                              int row0, int rows, int k0, int kend, int tid) {
    unsigned okmask = 0xFu;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int f = tid + 256 * i;
        const int row = KC ? row0 + (f >> 3) : row0 + ((f & 31) << 2);
        const int k = KC ? k0 + ((f & 7) << 2) : k0 + (f >> 5);
        float4 v;
        if (VEC) {
            const int rc = KC ? (row < rows ? row : rows - 1) : (row < rows ? row : rows - 4);
            const int kc = KC ? (k < kend ? k : kend - 4) : (k < kend ? k : kend - 1);
            const float* p = KC ? base + (int64_t)rc * s_row + kc : base + (int64_t)kc * s_k + rc;
            v = *reinterpret_cast<const float4*>(p);
            if (!((row < rows) && (k < kend))) okmask &= ~(1u << i);
        } else {
            v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (KC) {
                if (row < rows && k < kend) {
                    const float* p = base + (int64_t)row * s_row + (int64_t)k * s_k;
                    v.x = p[0]; if (k + 1 < kend) v.y = p[s_k]; if (k + 2 < kend) v.z = p[2 * s_k]; if (k + 3 < kend) v.w = p[3 * s_k];
                }
            } else {
                if (k < kend && row < rows) {
                    const float* p = base + (int64_t)k * s_k + (int64_t)row * s_row;
                    v.x = p[0]; if (row + 1 < rows) v.y = p[s_row]; if (row + 2 < rows) v.z = p[2 * s_row]; if (row + 3 < rows) v.w = p[3 * s_row];
                }
            }
        }
        r[i] = v;
    }
    return okmask;
}

template <bool KC>
__device__ __forceinline__ void store_tile(float4 (&r)[4], unsigned okmask, float (*T)[LDT], int tid) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int f = tid + 256 * i;
        if (!((okmask >> i) & 1u)) r[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (KC) {
            const int row = f >> 3, k = (f & 7) << 2;
            T[k + 0][row] = r[i].x; T[k + 1][row] = r[i].y; T[k + 2][row] = r[i].z; T[k + 3][row] = r[i].w;
        } else {
            const int k = f >> 5, row = (f & 31) << 2;
            *reinterpret_cast<float4*>(&T[k][row]) = r[i];
        }
    }
}

// Workgroup -> tile map.  The dispatcher places workgroup b on XCD b % 8 (observed, used for speed only): remap so
// that each XCD owns a CONTIGUOUS run of tiles (bijective for any tile count), and walk N fastest inside the run, so
// the workgroups sharing an XCD's private 4-MiB L2 also share their A row-panels / B column-panels.
__device__ __forceinline__ int xcd_tile(int wg, int ntiles) {
    const int q = ntiles >> 3, r = ntiles & 7, xcd = wg & 7, idx = wg >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

template <bool AKC, bool BKC, bool VEC, int EPI>
__global__ __launch_bounds__(256) void gemm_f32_kernel(GemmArgs g) {
    __shared__ __attribute__((aligned(16))) float As[BKT][LDT];
    __shared__ __attribute__((aligned(16))) float Bs[BKT][LDT];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int tile = xcd_tile(blockIdx.x, g.tiles_m * g.tiles_n);
    const int tm = tile / g.tiles_n, tn = tile - tm * g.tiles_n;
    const int zb = blockIdx.y;                       // batch index
    const int zk = blockIdx.z;                       // split-K slab
    const int z0 = zb / g.nb1, z1 = zb - z0 * g.nb1;
    const float* A = g.A + z0 * g.a_b0 + z1 * g.a_b1;
    const float* B = g.B + z0 * g.b_b0 + z1 * g.b_b1;
    const int m0 = tm * BM, n0 = tn * BN;
    const int kbeg = zk * g.k_chunk;
    const int kend = (kbeg + g.k_chunk < g.K) ? kbeg + g.k_chunk : g.K;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    if (kbeg < kend) {                               // an empty split-K slab just writes zeros
        float4 ra[4], rb[4];
        unsigned oka = load_tile<AKC, VEC>(ra, A, g.a_m, g.a_k, m0, g.M, kbeg, kend, tid);
        unsigned okb = load_tile<BKC, VEC>(rb, B, g.b_n, g.b_k, n0, g.N, kbeg, kend, tid);
        store_tile<AKC>(ra, oka, As, tid);
        store_tile<BKC>(rb, okb, Bs, tid);
        __syncthreads();

        const int arow = wm * 64 + (lane & 31), brow = wn * 64 + (lane & 31), kl = lane >> 5;
        for (int k0 = kbeg; k0 < kend; k0 += BKT) {
            const bool more = (k0 + BKT) < kend;
            if (more) {                              // next k-tile: global loads in flight under the 64 MFMAs below
                oka = load_tile<AKC, VEC>(ra, A, g.a_m, g.a_k, m0, g.M, k0 + BKT, kend, tid);
                okb = load_tile<BKC, VEC>(rb, B, g.b_n, g.b_k, n0, g.N, k0 + BKT, kend, tid);
            }
            // operand fragments are fetched one k2-step ahead of the MFMAs that consume them
            float a0 = As[kl][arow], a1 = As[kl][arow + 32], b0 = Bs[kl][brow], b1 = Bs[kl][brow + 32];
#pragma unroll
            for (int kk = 0; kk < BKT; kk += 2) {
                float na0 = 0.f, na1 = 0.f, nb0 = 0.f, nb1 = 0.f;
                if (kk + 2 < BKT) {
                    na0 = As[kk + 2 + kl][arow]; na1 = As[kk + 2 + kl][arow + 32];
                    nb0 = Bs[kk + 2 + kl][brow]; nb1 = Bs[kk + 2 + kl][brow + 32];
                }
                __builtin_amdgcn_sched_barrier(0);   // keep the fragment prefetch ahead of this step's MFMAs
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
                a0 = na0; a1 = na1; b0 = nb0; b1 = nb1;
            }
            __syncthreads();
            if (more) {
                store_tile<AKC>(ra, oka, As, tid);
                store_tile<BKC>(rb, okb, Bs, tid);
            }
            __syncthreads();
        }
    }

    // ---- epilogue: MFMA C layout  col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5) ----------
    const bool split = g.splitk > 1;
    float* C = split ? g.C + (int64_t)zk * g.c_split + (int64_t)zb * g.M * g.N : g.C + z0 * g.c_b0 + z1 * g.c_b1;
    const int64_t ldc = split ? g.N : g.c_m;
    const float alpha = split ? 1.0f : g.alpha;
    const float* bias = (g.bias && !split) ? g.bias + z1 * g.bias_b1 : nullptr;
    const bool bias_n = bias && g.bias_mode == SEGX_BIAS_N, bias_m = bias && g.bias_mode == SEGX_BIAS_M;
    float* AUX = (EPI == SEGX_EPI_GELU) ? g.aux + z0 * g.c_b0 + z1 * g.c_b1 : nullptr;
    const float inv_keep = g.dropout_p > 0.f ? 1.0f / (1.0f - g.dropout_p) : 1.0f;
    const bool full = (m0 + BM <= g.M) && (n0 + BN <= g.N);
    float vmax = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = n0 + wn * 64 + j * 32 + (lane & 31);
            const bool col_ok = full || col < g.N;
            const float bn = (bias_n && col_ok) ? bias[col] : 0.f;
            const int rbase = m0 + wm * 64 + i * 32 + 4 * (lane >> 5);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = rbase + (r & 3) + 8 * (r >> 2);
                const bool ok = full || (col_ok && row < g.M);
                float v = acc[i][j][r] * alpha + bn;
                if (bias_m) v += ok ? bias[row] : 0.f;
                if (EPI == SEGX_EPI_GELU) {
                    if (ok) AUX[(int64_t)row * ldc + col] = v;
                    v = gelu_erf(v);
                    if (g.dropout_p > 0.f)
                        v *= dropout_scale(g.seed, g.offset, ((uint64_t)zb * g.M + row) * g.N + col, g.dropout_p, inv_keep);
                }
                if (ok) { vmax = fmaxf(vmax, v); C[(int64_t)row * ldc + col] = v; }
            }
        }
    }
    if (g.gmax && !split) {
        vmax = wave_max(vmax);
        if (lane == 0) atomicMax(reinterpret_cast<int*>(g.gmax), __float_as_int(vmax));   // vmax >= 0: int order == float order
    }
}

// Split-K second stage: C = alpha * sum_s slab[s] (+ bias), deterministic slab order.
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ ws, float* __restrict__ C, const float* __restrict__ bias,
                                                            int M, int N, int nb1, int splitk, int64_t c_split,
                                                            int64_t c_b0, int64_t c_b1, int64_t c_m, float alpha,
                                                            int bias_mode, int64_t bias_b1, int64_t total) {
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int col = (int)(idx % N);
        const int64_t t = idx / N;
        const int row = (int)(t % M);
        const int zb = (int)(t / M);
        float s = 0.f;
        for (int k = 0; k < splitk; ++k) s += ws[(int64_t)k * c_split + idx];
        s *= alpha;
        const int z0 = zb / nb1, z1 = zb - z0 * nb1;
        if (bias_mode == SEGX_BIAS_N) s += bias[z1 * bias_b1 + col];
        else if (bias_mode == SEGX_BIAS_M) s += bias[z1 * bias_b1 + row];
        C[z0 * c_b0 + z1 * c_b1 + (int64_t)row * c_m + col] = s;
    }
}

static bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace segx

extern "C" int segx_gemm_f32(const float* A, const float* B, float* C, const segx_gemm_desc* d, void* stream_) {
    using namespace segx;
    hipStream_t stream = (hipStream_t)stream_;
    SEGX_REQUIRE(A && B && C && d, "segx_gemm_f32: null pointer");
    SEGX_REQUIRE(d->M > 0 && d->N > 0 && d->K > 0 && d->nb0 > 0 && d->nb1 > 0, "segx_gemm_f32: bad sizes M=%d N=%d K=%d nb=%dx%d",
                 d->M, d->N, d->K, d->nb0, d->nb1);
    SEGX_REQUIRE(d->a_m == 1 || d->a_k == 1, "segx_gemm_f32: A needs a unit stride (a_m=%lld a_k=%lld)", (long long)d->a_m, (long long)d->a_k);
    SEGX_REQUIRE(d->b_n == 1 || d->b_k == 1, "segx_gemm_f32: B needs a unit stride (b_n=%lld b_k=%lld)", (long long)d->b_n, (long long)d->b_k);
    SEGX_REQUIRE(d->epilogue == SEGX_EPI_NONE || d->epilogue == SEGX_EPI_GELU, "segx_gemm_f32: bad epilogue %d", d->epilogue);
    SEGX_REQUIRE(d->epilogue != SEGX_EPI_GELU || d->aux, "segx_gemm_f32: GELU epilogue needs aux");
    SEGX_REQUIRE(d->bias_mode == SEGX_BIAS_NONE || d->bias, "segx_gemm_f32: bias_mode set but bias null");
    SEGX_REQUIRE(d->dropout_p >= 0.f && d->dropout_p < 1.f, "segx_gemm_f32: dropout_p out of range");
    const int splitk = d->splitk > 1 ? d->splitk : 1;
    SEGX_REQUIRE(splitk == 1 || (d->workspace && d->epilogue == SEGX_EPI_NONE && !d->gmax), "segx_gemm_f32: split-K needs workspace and a plain epilogue");

    GemmArgs g;
    g.A = A; g.B = B; g.C = C; g.bias = d->bias_mode ? d->bias : nullptr; g.aux = d->epilogue == SEGX_EPI_GELU ? d->aux : nullptr;
    g.gmax = d->gmax;
    g.M = d->M; g.N = d->N; g.K = d->K; g.nb1 = d->nb1;
    g.a_b0 = d->a_b0; g.a_b1 = d->a_b1; g.a_m = d->a_m; g.a_k = d->a_k;
    g.b_b0 = d->b_b0; g.b_b1 = d->b_b1; g.b_n = d->b_n; g.b_k = d->b_k;
    g.c_b0 = d->c_b0; g.c_b1 = d->c_b1; g.c_m = d->c_m; g.bias_b1 = d->bias_b1;
    g.alpha = d->alpha; g.epilogue = d->epilogue; g.bias_mode = d->bias_mode;
    const bool akc = (d->a_k == 1), bkc = (d->b_k == 1);
    // float4 loads need 16-B aligned bases, every non-unit stride and the contiguous extents multiples of 4
    const bool vecA = aligned16(A) && (d->a_b0 % 4 == 0) && (d->a_b1 % 4 == 0) && ((akc ? d->a_m : d->a_k) % 4 == 0) &&
                      ((akc ? d->K : d->M) % 4 == 0);
    const bool vecB = aligned16(B) && (d->b_b0 % 4 == 0) && (d->b_b1 % 4 == 0) && ((bkc ? d->b_n : d->b_k) % 4 == 0) &&
                      ((bkc ? d->K : d->N) % 4 == 0);
    const bool vec = vecA && vecB;
    g.vecA = vecA; g.vecB = vecB;
    g.tiles_m = ceil_div(d->M, BM); g.tiles_n = ceil_div(d->N, BN);
    g.dropout_p = d->dropout_p; g.seed = d->seed; g.offset = d->offset;
    g.splitk = splitk;
    // k_chunk: multiple of the k-tile so slabs start on tile boundaries (and stay float4-aligned)
    g.k_chunk = splitk == 1 ? d->K : ceil_div(ceil_div(d->K, splitk), BKT) * BKT;
    const int nbatch = d->nb0 * d->nb1;
    g.c_split = (int64_t)nbatch * d->M * d->N;
    if (splitk > 1) g.C = d->workspace;

    dim3 grid(g.tiles_m * g.tiles_n, nbatch, splitk), block(256);
#define SEGX_LAUNCH(AK, BK, V, E) hipLaunchKernelGGL((gemm_f32_kernel<AK, BK, V, E>), grid, block, 0, stream, g)
#define SEGX_LAUNCH_LAYOUT(V, E)                                   \
    do {                                                           \
        if (akc && bkc) SEGX_LAUNCH(true, true, V, E);             \
        else if (akc && !bkc) SEGX_LAUNCH(true, false, V, E);      \
        else if (!akc && bkc) SEGX_LAUNCH(false, true, V, E);      \
        else SEGX_LAUNCH(false, false, V, E);                      \
    } while (0)
    if (d->epilogue == SEGX_EPI_GELU) {
        SEGX_REQUIRE(akc && bkc, "segx_gemm_f32: the GELU epilogue is built for k-contiguous operands (nn.Linear)");
        if (vec) SEGX_LAUNCH(true, true, true, SEGX_EPI_GELU); else SEGX_LAUNCH(true, true, false, SEGX_EPI_GELU);
    } else if (vec) {
        SEGX_LAUNCH_LAYOUT(true, SEGX_EPI_NONE);
    } else {
        SEGX_LAUNCH_LAYOUT(false, SEGX_EPI_NONE);
    }
    int rc = check_launch("segx_gemm_f32");
    if (rc) return rc;
    if (splitk > 1) {
        const int64_t total = g.c_split;
        const int blocks = (int)i64min(2048, (total + 255) / 256);
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, stream, (const float*)d->workspace, C, g.bias,
                           d->M, d->N, d->nb1, splitk, g.c_split, d->c_b0, d->c_b1, d->c_m, d->alpha, d->bias_mode, d->bias_b1, total);
        rc = check_launch("segx_gemm_f32/splitk_reduce");
    }
    return rc;
}
