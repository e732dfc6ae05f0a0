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
#include "gemm_core.h"

namespace segx {

template <bool AKC, bool BKC, bool VEC, int EPI>
__global__ __launch_bounds__(256, 2) void gemm_f32_kernel(GemmArgs g) {
    __shared__ __attribute__((aligned(16))) TileLds lds;
    const TileCoord t = tile_coord(g);
    const DenseLoader<AKC, VEC> la{g.A + t.z0 * g.a_b0 + t.z1 * g.a_b1, g.a_m, g.a_k, t.m0, g.M};
    const DenseLoader<BKC, VEC> lb{g.B + t.z0 * g.b_b0 + t.z1 * g.b_b1, g.b_n, g.b_k, t.n0, g.N};
    f32x16 acc[2][2];
    gemm_mainloop(acc, la, lb, t.kbeg, t.kend, lds);
    gemm_epilogue<EPI>(acc, g, t);
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
