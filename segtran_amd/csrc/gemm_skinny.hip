// gemm_skinny.hip -- the batch-reduced weight gradients of the backbone's FIRST pointwise convolutions (efficientnet/model.py:96, 113: expand / project convs of
// stages 1 - 3 at 512 x 512 .. 128 x 128) as ONE streaming pass (round 6).
//
//   dW[m][n] = sum_b sum_k A_b[m][k] B_b[n][k]      A = dY [Cout][H W], B = X [Cin][H W], both k-contiguous, k = the 65 536 .. 262 144 positions of a plane,
//                                                   one side <= 32 rows, the other <= 192 (192 x 32, 144 x 24, 32 x 192, 24 x 48, 24 x 24, 3 x 160, 56 x 32 ...)
//
// These products are HBM-bound (6 - 14 FLOP / byte): what matters is that every byte of A and B is requested once, in long contiguous pieces, with a whole tile
// of loads in flight per compute unit.  The tile kernels of gemm.hip gave each workgroup a 128 x 32 (32 x 128) output tile and a K-slab of ~1 000 positions
// x one batch member: 63 K-slabs x 6 members x 2 M-tiles = 756 workgroups, each paying its pipeline fill and drain on 32 k-tiles, the 32-row operand fetched once
// per M-tile, 378 slabs of the output to reduce (2.9 - 3.3 TB/s on the fp32 engine).  Here ONE persistent workgroup per compute unit owns the WHOLE M x N output and a contiguous
// run of the (member, 64-position tile) stream; the batch is walked inside the kernel (the sum over the members is part of the contraction), so there are
// <= 256 slabs however large the batch:
//   * a tile = all M + N rows x 64 positions = 256 contiguous bytes per row; its float4 pieces are requested one tile AHEAD into registers (<= 7 per thread)
//     and stored into the other half of a double-buffered LDS image while the current half is being multiplied: one barrier per tile;
//   * eight waves split the 64 positions of a tile (8 each): a lane reads ONE ds_read_b128 per 32-row block and side -- row r, positions 8 w + 4 h .. + 3 --
//     and feeds its four values to four v_mfma_f32_32x32x2f32 steps (exact fp32 products: this is the fp32 engine's arithmetic, the matrix pipe is ~40 % busy
//     at the HBM roof, nothing is gained by the bf16 split here).  Row stride 68 words: the 16 lanes of a ds_read_b128 service group land on 16 distinct
//     4-bank groups (MI355X_MICROARCH.md, LDS);
//   * every wave keeps the whole M x N output (<= 6 blocks x 16 registers); at the end the eight partial outputs are added in wave order through LDS and the
//     workgroup writes ONE slab; the slabs are summed in a fixed order by the batch_reduce stage of segx_gemm_f32 (slab_sum_parts_kernel): deterministic.
#include "common.h"
#include "gemm_skinny.h"

namespace segx {

constexpr int SK_KT = 64, SK_LDW = SK_KT + 4, SK_T = 512;

struct SkinnyArgs {
    const float* A; const float* B; float* ws;
    int M, N, nb1, tpb;                       // tpb: 64-position tiles per batch member
    int64_t a_b0, a_b1, a_m, b_b0, b_b1, b_n;
    int64_t total;                            // nbatch * tpb tiles
};

template <int MB, int NB>
__global__ __launch_bounds__(SK_T) void gemm_skinny_nt_kernel(SkinnyArgs g) {
    constexpr int ROWS = (MB + NB) * 32, BUF = ROWS * SK_LDW;             // floats per LDS buffer
    constexpr int NLD = (ROWS * 16 + SK_T - 1) / SK_T;                    // float4 pieces per thread and tile
    __shared__ __attribute__((aligned(16))) float lds[2 * BUF];
    static_assert(2 * BUF * 4 <= 160 * 1024 && 8 * 1024 <= 2 * BUF, "LDS image");
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, r = lane & 31, h = lane >> 5;
    for (int i = tid; i < 2 * BUF / 4; i += SK_T) reinterpret_cast<f32x4*>(lds)[i] = f32x4{0.f, 0.f, 0.f, 0.f};       // rows beyond M / N stay zero
    const int nA = g.M * 16, nAll = (g.M + g.N) * 16;
    int goff[NLD], loff[NLD]; bool isA[NLD], valid[NLD];
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
        const int f = tid + i * SK_T;
        valid[i] = f < nAll;
        isA[i] = !valid[i] || f < nA;
        const int row = !valid[i] ? 0 : (isA[i] ? f >> 4 : (f - nA) >> 4), c4 = f & 15;
        goff[i] = (int)((int64_t)row * (isA[i] ? g.a_m : g.b_n)) + (valid[i] ? c4 * 4 : 0);
        loff[i] = ((isA[i] ? row : MB * 32 + row) * SK_LDW + c4 * 4);
    }
    const int64_t t0 = g.total * blockIdx.x / gridDim.x, t1 = g.total * (blockIdx.x + 1) / gridDim.x;
    f32x16 acc[MB][NB];
#pragma unroll
    for (int i = 0; i < MB; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;
    f32x4 pre[NLD];
    auto issue = [&](int64_t t) {
        const int64_t b = t / g.tpb; const int64_t k0 = (t - b * g.tpb) * SK_KT;
        const int64_t z0 = b / g.nb1, z1 = b - z0 * g.nb1;
        const float* pa = g.A + z0 * g.a_b0 + z1 * g.a_b1 + k0; const float* pb = g.B + z0 * g.b_b0 + z1 * g.b_b1 + k0;
#pragma unroll
        for (int i = 0; i < NLD; ++i) pre[i] = *reinterpret_cast<const f32x4*>((isA[i] ? pa : pb) + goff[i]);
    };
    auto stash = [&](int buf) {
#pragma unroll
        for (int i = 0; i < NLD; ++i) if (valid[i]) *reinterpret_cast<f32x4*>(lds + buf * BUF + loff[i]) = pre[i];
    };
    __syncthreads();
    if (t0 < t1) { issue(t0); stash(0); }
    __syncthreads();
    for (int64_t t = t0; t < t1; ++t) {
        const int cur = (int)((t - t0) & 1);
        const bool more = t + 1 < t1;
        if (more) issue(t + 1);                                            // next tile's loads in flight under this tile's matrix work
        const float* la = lds + cur * BUF + r * SK_LDW + 8 * w + 4 * h;
        f32x4 a[MB], bq[NB];
#pragma unroll
        for (int i = 0; i < MB; ++i) a[i] = *reinterpret_cast<const f32x4*>(la + i * 32 * SK_LDW);
#pragma unroll
        for (int j = 0; j < NB; ++j) bq[j] = *reinterpret_cast<const f32x4*>(la + (MB + j) * 32 * SK_LDW);
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int i = 0; i < MB; ++i)
#pragma unroll
                for (int j = 0; j < NB; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][e], bq[j][e], acc[i][j], 0, 0, 0);
        if (more) stash(cur ^ 1);
        __syncthreads();
    }
    // the eight waves' partial outputs, added in wave order; one slab per workgroup
    float* red = lds;
    float* slab = g.ws + (int64_t)blockIdx.x * g.M * g.N;
#pragma unroll
    for (int i = 0; i < MB; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j) {
#pragma unroll
            for (int q = 0; q < 16; ++q) red[w * 1024 + (8 * (q >> 2) + 4 * h + (q & 3)) * 32 + r] = acc[i][j][q];
            __syncthreads();
#pragma unroll
            for (int e = tid; e < 1024; e += SK_T) {
                float s = red[e];
#pragma unroll
                for (int v = 1; v < 8; ++v) s += red[v * 1024 + e];
                const int gm = i * 32 + (e >> 5), gn = j * 32 + (e & 31);
                if (gm < g.M && gn < g.N) slab[(int64_t)gm * g.N + gn] = s;
            }
            __syncthreads();
        }
}

// few rows = few bytes per tile: two / three workgroups per compute unit keep >= 32 KB of loads in flight there (their LDS images are 35 - 52 KB)
int skinny_nt_wgs_per_cu(int M, int N) { const int rows = ((M + 31) / 32 + (N + 31) / 32) * 32; return rows <= 64 ? 3 : rows <= 96 ? 2 : 1; }

bool skinny_nt_shape_ok(int M, int N, int64_t K, int nbatch, int grid) {
    const int mb = (M + 31) / 32, nb = (N + 31) / 32;
    if (!(mb >= 1 && nb >= 1 && (mb == 1 || nb == 1) && mb <= 6 && nb <= 6)) return false;
    if (K % SK_KT != 0) return false;
    return (int64_t)nbatch * (K / SK_KT) >= 8LL * grid;                   // a run of at least eight tiles per workgroup
}

int launch_skinny_nt(const float* A, const float* B, float* ws, int M, int N, int64_t K, int nb0, int nb1, int64_t a_b0, int64_t a_b1, int64_t a_m,
                     int64_t b_b0, int64_t b_b1, int64_t b_n, int grid, hipStream_t stream) {
    SkinnyArgs g;
    g.A = A; g.B = B; g.ws = ws; g.M = M; g.N = N; g.nb1 = nb1; g.tpb = (int)(K / SK_KT);
    g.a_b0 = a_b0; g.a_b1 = a_b1; g.a_m = a_m; g.b_b0 = b_b0; g.b_b1 = b_b1; g.b_n = b_n;
    g.total = (int64_t)nb0 * nb1 * g.tpb;
    const int mb = (M + 31) / 32, nb = (N + 31) / 32;
#define SEGX_SKINNY(MB_, NB_) hipLaunchKernelGGL((gemm_skinny_nt_kernel<MB_, NB_>), dim3(grid), dim3(SK_T), 0, stream, g)
    if (nb == 1) switch (mb) { case 1: SEGX_SKINNY(1, 1); break; case 2: SEGX_SKINNY(2, 1); break; case 3: SEGX_SKINNY(3, 1); break;
                               case 4: SEGX_SKINNY(4, 1); break; case 5: SEGX_SKINNY(5, 1); break; default: SEGX_SKINNY(6, 1); break; }
    else switch (nb) { case 2: SEGX_SKINNY(1, 2); break; case 3: SEGX_SKINNY(1, 3); break; case 4: SEGX_SKINNY(1, 4); break;
                       case 5: SEGX_SKINNY(1, 5); break; default: SEGX_SKINNY(1, 6); break; }
#undef SEGX_SKINNY
    return check_launch("segx_gemm_f32/skinny_nt");
}

}  // namespace segx
