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
#include "gemm_x6ws.h"
#include "gemm_skinny.h"
#include "gemm_tuned.h"

namespace segx {

template <class Cfg, bool AKC, bool BKC, bool VEC, int EPI>
__global__ __launch_bounds__(256, 2) void gemm_f32_kernel(GemmArgs g) {
    __shared__ __attribute__((aligned(16))) TileLdsT<Cfg> lds;
    const TileCoord t = tile_coord<Cfg>(g);
    const DenseLoader<AKC, VEC, Cfg::BM> la{g.A + t.z0 * g.a_b0 + t.z1 * g.a_b1, g.a_m, g.a_k, t.m0, g.M};
    const DenseLoader<BKC, VEC, Cfg::BN> lb{g.B + t.z0 * g.b_b0 + t.z1 * g.b_b1, g.b_n, g.b_k, t.n0, g.N};
    f32x16 acc[Cfg::MI][Cfg::NJ];
    gemm_mainloop<Cfg>(acc, la, lb, t.kbeg, t.kend, lds);
    gemm_epilogue<EPI, Cfg>(acc, g, t);
}

// The same GEMM on the bf16x6 engine (gemm_x6.h): fp32 operands split into three bf16 planes on their way into LDS, six bf16 MFMAs per
// block and 16 k.  WPE = resident waves per SIMD the register allocation must allow (LDS: 48 / 36 / 24 KB per workgroup).
template <class Cfg, bool AKC, bool BKC, int EPI, int WPE, int VAR = 0>
__global__ __launch_bounds__(256) SEGX_MIN_WAVES_PER_SIMD(WPE) void gemm_x6_kernel(GemmArgs g) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[X6Lds<Cfg>::BYTES];
    const TileCoord t = tile_coord<Cfg>(g);
    const DenseLoader6<AKC, Cfg::BM> la{g.A + t.z0 * g.a_b0 + t.z1 * g.a_b1, g.a_m, g.a_k, t.m0, g.M};
    const DenseLoader6<BKC, Cfg::BN> lb{g.B + t.z0 * g.b_b0 + t.z1 * g.b_b1, g.b_n, g.b_k, t.n0, g.N};
    f32x16 acc[Cfg::MI][Cfg::NJ];
    gemm_mainloop_x6<Cfg, DenseLoader6<AKC, Cfg::BM>, DenseLoader6<BKC, Cfg::BN>, VAR>(acc, la, lb, t.kbeg, t.kend, lds);
    gemm_epilogue<EPI, Cfg>(acc, g, t);
}

// The 4-wave kernel with the LEAN operand loaders of gemm_x6ws.h (one uniform base per k-tile + a 32-bit offset per piece computed once per tile:
// no per-k-tile address arithmetic on the vector pipe -- ~80 of the ~280 vector instructions a thread issued per k-tile).  Whole 32-k tiles and
// 32-bit operand offsets only (gemm_ws_ok): the host keeps gemm_x6_kernel for everything else.
template <class Cfg, bool AKC, bool BKC, int EPI, int WPE>
__global__ __launch_bounds__(256) SEGX_MIN_WAVES_PER_SIMD(WPE) void gemm_x6_lean_kernel(GemmArgs g) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[X6Lds<Cfg>::BYTES];
    const TileCoord t = tile_coord<Cfg>(g);
    WsDense6<AKC, Cfg::BM> la; WsDense6<BKC, Cfg::BN> lb;
    la.begin(g.A + t.z0 * g.a_b0 + t.z1 * g.a_b1, g.a_m, g.a_k, t.m0, g.M, threadIdx.x);
    lb.begin(g.B + t.z0 * g.b_b0 + t.z1 * g.b_b1, g.b_n, g.b_k, t.n0, g.N, threadIdx.x);
    f32x16 acc[Cfg::MI][Cfg::NJ];
    gemm_mainloop_x6<Cfg, WsDense6<AKC, Cfg::BM>, WsDense6<BKC, Cfg::BN>, 0>(acc, la, lb, t.kbeg, t.kend, lds);
    gemm_epilogue<EPI, Cfg>(acc, g, t);
}

// The wave-specialised persistent form (gemm_x6ws.h): 512 threads, one workgroup per CU (144 / 96 KB of LDS), grid = min(items, 256).
template <class Cfg, bool AKC, bool BKC, int EPI, int PRIO = 0>
__global__ __launch_bounds__(512) void gemm_x6ws_kernel(GemmArgs g) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[X6WsLds<Cfg>::BYTES];
    x6ws_body<Cfg, DenseMk6<Cfg, AKC, BKC>, EPI, PRIO>(g, DenseMk6<Cfg, AKC, BKC>{}, lds);
}

// ... with the B operand split ahead of time (segx_x6_presplit; WsPre6): plain epilogue, whole 32-k stages, 256- or 128-row B tiles
template <class Cfg, bool AKC>
__global__ __launch_bounds__(512) void gemm_x6ws_pre_kernel(GemmArgs g) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[X6WsLds<Cfg>::BYTES];
    x6ws_body<Cfg, PreMk6<Cfg, AKC>, SEGX_EPI_NONE, 0>(g, PreMk6<Cfg, AKC>{}, lds);
}

// fp32 operand [nb][rows][K] (any of the two unit-stride layouts) -> three bf16 planes [nb][plane][rows][K], k contiguous: x = hi + mid + lo with the
// rounding of split3_pair.  A thread owns 8 consecutive k of one row.  Threads run along k for k-contiguous input (coalesced loads and stores), along the
// rows for row-contiguous input (coalesced loads, 16-byte stores one row apart: weights only, a few MB).
__global__ __launch_bounds__(256) void x6_presplit_kernel(const float* __restrict__ W, unsigned short* __restrict__ P, int rows, int K, int64_t s_row, int64_t s_k,
                                                          int nb1, int64_t s_b0, int64_t s_b1, int64_t per_batch) {
    const int kch = K >> 3;
    const int z = blockIdx.y, z0 = z / nb1, z1 = z - z0 * nb1;
    const float* __restrict__ w = W + z0 * s_b0 + z1 * s_b1;
    unsigned short* __restrict__ o = P + (int64_t)z * 3 * per_batch;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < per_batch / 8; idx += (int64_t)gridDim.x * blockDim.x) {
        int row, kc;
        if (s_k == 1) { row = (int)(idx / kch); kc = (int)(idx - (int64_t)row * kch); }
        else { kc = (int)(idx / rows); row = (int)(idx - (int64_t)kc * rows); }
        float v[8];
        const float* src = w + (int64_t)row * s_row + (int64_t)(kc * 8) * s_k;
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = src[(int64_t)j * s_k];
        const Split2 a = split3_pair(v[0], v[1]), b = split3_pair(v[2], v[3]), c = split3_pair(v[4], v[5]), d = split3_pair(v[6], v[7]);
        const int64_t e = (int64_t)row * K + kc * 8;
        *reinterpret_cast<uint4*>(o + e) = make_uint4(a.h, b.h, c.h, d.h);
        *reinterpret_cast<uint4*>(o + per_batch + e) = make_uint4(a.m, b.m, c.m, d.m);
        *reinterpret_cast<uint4*>(o + 2 * per_batch + e) = make_uint4(a.l, b.l, c.l, d.l);
    }
}

// The same reduction for MANY slabs over a SMALL output (batch_reduce of the skinny weight gradients: 6 x 63 slabs of 24 x 144 floats): the slab
// loop is dealt out over PARTS threads per output element (slab s -> part s % PARTS, each part in slab order) and the PARTS partial sums are
// added in part order through LDS -- a fixed summation tree, so still deterministic; 256 / PARTS outputs per workgroup.
template <int PARTS>
__global__ __launch_bounds__(256) void slab_sum_parts_kernel(const float* __restrict__ ws, float* __restrict__ C, const float* __restrict__ bias,
                                                             int M, int N, int nslabs, int64_t total, int64_t c_m, float alpha, int bias_mode) {
    __shared__ float part[256];
    constexpr int EPB = 256 / PARTS;
    const int e = threadIdx.x % EPB, pt = threadIdx.x / EPB;
    const int64_t idx = (int64_t)blockIdx.x * EPB + e;
    float s = 0.f;
    if (idx < total) for (int k = pt; k < nslabs; k += PARTS) s += ws[(int64_t)k * total + idx];
    part[threadIdx.x] = s;
    __syncthreads();
    if (pt == 0 && idx < total) {
        float t = 0.f;
#pragma unroll
        for (int q = 0; q < PARTS; ++q) t += part[q * EPB + e];
        t *= alpha;
        const int col = (int)(idx % N), row = (int)(idx / N);
        if (bias_mode == SEGX_BIAS_N) t += bias[col]; else if (bias_mode == SEGX_BIAS_M) t += bias[row];
        C[(int64_t)row * c_m + col] = t;
    }
}

static bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// Tile / split-K planning.  Skinny operands are a rule (measured, tools/gemm_bench.py tiles, r01-j): an operand with <= 48 rows
// is streamed with the 32-row tile on that side -- the 128-row tile spends 3/4 of its LDS traffic and MFMA issue slots on
// clamped duplicate rows (192x32x65536 weight gradient: 0.34 -> 0.14 ms).  Everything else is priced with a small model:
//   time = rounds of workgroups over the resident slots x (k-tiles x step time + prologue/epilogue) + split-K slab reduction
// Whole-GEMM quantisation matters (784 workgroups on 768 slots take two rounds, not 1.02), and so do padded edge tiles, which
// the workgroup count already contains.  Step times are the measured ~112 TFLOP/s of the engine expressed per k-tile.
// splitk_fixed > 0: the caller has already chosen the split factor; 0: choose it too.  vec = false: only the default tile is built.
// bf16x6 engine: float4-legal operands, neither side skinny (those GEMMs are HBM-bound and stream through the 32-row fp32 tiles)
// the engine of ONE call: segx_gemm_desc.engine (SEGX_ENGINE_SEL_F32 / _BF16X6) or, at SEGX_ENGINE_SEL_DEFAULT, the process default (segx_tune knob 4)
static int call_engine(const segx_gemm_desc* d) {
    return d->engine == SEGX_ENGINE_SEL_F32 ? SEGX_ENGINE_F32 : d->engine == SEGX_ENGINE_SEL_BF16X6 ? SEGX_ENGINE_BF16X6 : kget(knobs().engine);
}
static bool x6_eligible(int engine, int M, int N, bool vec) { return engine == SEGX_ENGINE_BF16X6 && vec && M > 48 && N > 48; }
// ws_ok: the wave-specialised persistent kernels may run this GEMM (whole 32-k stages, 32-bit operand offsets; no fused GELU: its epilogue
// runs on four of the eight waves there and measured 63 against 91 TFLOP/s)
// nrc = number of row-contiguous operands (0..2): their loaders cost the 4-wave kernels 9 % / 34 % per k-tile (r03_f: 201 / 185 / 150 TFLOP/s for
// NT / NN / TN at 24576 x 1792 x 1792), the wave-specialised ones 1 % / 5 % (220 / 218 / 198 incl. the slab reduction)
static void plan6(int M, int N, int K, int nbatch, bool gelu, bool may_split, int splitk_fixed, bool ws_ok, int nrc, int* tile, int* splitk) {
    double best_t = -1.0;
    const float f4 = nrc == 0 ? 1.0f : nrc == 1 ? 1.09f : 1.34f, fws = nrc == 0 ? 1.0f : nrc == 1 ? 1.01f : 1.05f;
    for (const TileInfo6& c6 : kTiles6) {
        if (gelu && c6.id != SEGX_TILE_128x128) continue;                  // the fused GELU epilogue is built for the default tile
        const TileInfo c{c6.id, c6.bm, c6.bn, c6.wg_per_cu, c6.ktile_us * f4, c6.fixed_us};
        double t; int sk;
        if (splitk_fixed > 0 || !may_split) { sk = splitk_fixed > 0 ? splitk_fixed : 1; t = model_us(c, M, N, K, nbatch, sk); }
        else sk = best_splitk(c, M, N, K, nbatch, &t);
        if (best_t < 0.0 || t < best_t * 0.97) { best_t = t; *tile = c.id; *splitk = sk; }
    }
    if (!ws_ok || gelu) return;
    for (const TileInfo6& w6 : kTilesWs) {
        const TileInfo6 c6{w6.id, w6.bm, w6.bn, w6.wg_per_cu, w6.ktile_us * fws, w6.fixed_us};
        double t; int sk;
        if (splitk_fixed > 0 || !may_split) { sk = splitk_fixed > 0 ? splitk_fixed : 1; t = model_us_ws(c6, M, N, K, nbatch, sk, kget(knobs().ws_grid)); }
        else sk = best_splitk_ws(c6, M, N, K, nbatch, kget(knobs().ws_grid), &t);
        if (t < best_t * 0.97) { best_t = t; *tile = c6.id; *splitk = sk; }
    }
}
static bool gemm_lean_ok(const segx_gemm_desc* d);
// a residual operand keeps the call on the 4-wave kernels: their epilogue reads it as 16-byte quads next to the 16-byte stores, the wave-specialised
// kernels' 4-byte epilogue would read it scalar (r04_c: the 160 x 4096 x 960 x 6 dX GEMM 0.47 -> 0.76 ms/step with the residual on the 256 x 128 tile)
static bool gemm_ws_ok(const segx_gemm_desc* d) { return !d->resid && gemm_lean_ok(d); }
// whole 32-k stages and 32-bit operand offsets: what the lean loaders (4-wave lean kernels, wave-specialised kernels) need
static bool gemm_lean_ok(const segx_gemm_desc* d) {
    const bool akc = (d->a_k == 1), bkc = (d->b_k == 1);
    const int64_t a_span = akc ? (int64_t)d->M * d->a_m : (int64_t)d->K * d->a_k, b_span = bkc ? (int64_t)d->N * d->b_n : (int64_t)d->K * d->b_k;
    return d->K % BKT == 0 && a_span < (1LL << 29) && b_span < (1LL << 29);
}
static void plan(int M, int N, int K, int nbatch, bool vec, bool may_split, int splitk_fixed, int* tile, int* splitk) {
    const TileInfo* cand[3]; int nc = 0;
    if (!vec) cand[nc++] = &kTiles[0];
    else if (N <= 48 && M > N) cand[nc++] = &tile_info(SEGX_TILE_128x32);
    else if (M <= 48) cand[nc++] = &tile_info(SEGX_TILE_32x128);
    else { cand[nc++] = &kTiles[0]; cand[nc++] = &kTiles[1]; cand[nc++] = &kTiles[2]; }
    double best_t = -1.0;
    for (int i = 0; i < nc; ++i) {
        double t; int sk;
        if (splitk_fixed > 0 || !may_split) { sk = splitk_fixed > 0 ? splitk_fixed : 1; t = model_us(*cand[i], M, N, K, nbatch, sk); }
        else sk = best_splitk(*cand[i], M, N, K, nbatch, &t);
        if (best_t < 0.0 || t < best_t * 0.97) { best_t = t; *tile = cand[i]->id; *splitk = sk; }   // larger tiles win ties
    }
}
static bool gemm_vec_ok(const float* A, const float* B, const segx_gemm_desc* d) {
    const bool akc = (d->a_k == 1), bkc = (d->b_k == 1);
    // float4 loads need 16-B aligned bases, every non-unit stride and the contiguous extents multiples of 4
    const bool vecA = aligned16(A) && (d->a_b0 % 4 == 0) && (d->a_b1 % 4 == 0) && ((akc ? d->a_m : d->a_k) % 4 == 0) &&
                      ((akc ? d->K : d->M) % 4 == 0);
    const bool vecB = aligned16(B) && (d->b_b0 % 4 == 0) && (d->b_b1 % 4 == 0) && ((bkc ? d->b_n : d->b_k) % 4 == 0) &&
                      ((bkc ? d->K : d->N) % 4 == 0);
    return vecA && vecB;
}
// the streaming skinny weight gradient (gemm_skinny.hip): batch-reduced, both operands k-contiguous and float4-legal, plain epilogue, one side <= 32 rows;
// slabs = the largest multiple of the batch size under the persistent grid (the caller's workspace is splitk x nbatch slabs)
static int skinny_nt_splitk(const float* A, const float* B, const segx_gemm_desc* d) {
    if (!kget(knobs().skinny_nt) || !d->batch_reduce || d->epilogue != SEGX_EPI_NONE || d->gmax || d->resid || d->a_k != 1 || d->b_k != 1) return 0;
    if (!gemm_vec_ok(A, B, d)) return 0;
    const int nbatch = d->nb0 * d->nb1;
    const int sk = (int)i64max(1, (int64_t)kget(knobs().ws_grid) * skinny_nt_wgs_per_cu(d->M, d->N) / nbatch);
    if (!skinny_nt_shape_ok(d->M, d->N, d->K, nbatch, sk * nbatch)) return 0;
    if ((int64_t)d->M * d->a_m >= (1LL << 31) || (int64_t)d->N * d->b_n >= (1LL << 31)) return 0;          // 32-bit row offsets inside a member
    return sk;
}
}  // namespace segx

namespace segx {
static int gemm_plan_impl(const float* A, const float* B, const segx_gemm_desc* d, int* tile, int* splitk, bool use_table) {
    SEGX_REQUIRE(A && B && d && tile && splitk && d->M > 0 && d->N > 0 && d->K > 0 && d->nb0 > 0 && d->nb1 > 0, "segx_gemm_plan: bad args");
    const bool plain = d->epilogue == SEGX_EPI_NONE;
    int t = SEGX_TILE_128x128, sk = 1;
    const bool vec = gemm_vec_ok(A, B, d);
    if (const int ssk = skinny_nt_splitk(A, B, d)) { *tile = SEGX_TILE_SKINNY_NT; *splitk = ssk; return 0; }
    if (use_table && plain && !d->gmax && x6_eligible(call_engine(d), d->M, d->N, vec)) {
        // measured choices first (gemm_tuned.h); a wave-specialised entry still needs its preconditions (they hold for the shapes it was measured on)
        const int nbt = d->nb0 * d->nb1; const bool akc_ = d->a_k == 1, bkc_ = d->b_k == 1;
        for (const TunedGemm& e : kTunedGemm)
            if (e.M == d->M && e.N == d->N && e.K == d->K && e.nb == nbt && (e.akc != 0) == akc_ && (e.bkc != 0) == bkc_) {
                if (e.tile >= SEGX_TILE_256x128 && !gemm_ws_ok(d)) break;
                *tile = e.tile; *splitk = e.splitk;
                return 0;
            }
    }
    if (x6_eligible(call_engine(d), d->M, d->N, vec) && (plain || d->a_k == 1)) plan6(d->M, d->N, d->K, d->nb0 * d->nb1, !plain, plain && !d->gmax, 0, gemm_ws_ok(d), (d->a_k != 1) + (d->b_k != 1), &t, &sk);
    else plan(d->M, d->N, d->K, d->nb0 * d->nb1, vec && plain, plain && !d->gmax, 0, &t, &sk);
    *tile = t; *splitk = sk;
    return 0;
}
}  // namespace segx

extern "C" int segx_gemm_plan(const float* A, const float* B, const segx_gemm_desc* d, int* tile, int* splitk) { return segx::gemm_plan_impl(A, B, d, tile, splitk, true); }
// the cost model's own pick, without the measured table (tools/tune_gemm.py compares every candidate with it to decide which shapes need a table entry)
extern "C" int segx_gemm_plan_model(const float* A, const float* B, const segx_gemm_desc* d, int* tile, int* splitk) { return segx::gemm_plan_impl(A, B, d, tile, splitk, false); }

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
    const bool breduce = d->batch_reduce != 0;
    SEGX_REQUIRE(!breduce || (d->workspace && d->epilogue == SEGX_EPI_NONE && !d->gmax), "segx_gemm_f32: batch_reduce needs workspace and a plain epilogue");

    GemmArgs g;
    g.A = A; g.B = B; g.C = C; g.bias = d->bias_mode ? d->bias : nullptr; g.aux = d->epilogue == SEGX_EPI_GELU ? d->aux : nullptr;
    g.gmax = d->gmax;
    g.M = d->M; g.N = d->N; g.K = d->K; g.nb1 = d->nb1; g.nbatch = d->nb0 * d->nb1;
    g.a_b0 = d->a_b0; g.a_b1 = d->a_b1; g.a_m = d->a_m; g.a_k = d->a_k;
    g.b_b0 = d->b_b0; g.b_b1 = d->b_b1; g.b_n = d->b_n; g.b_k = d->b_k;
    g.c_b0 = d->c_b0; g.c_b1 = d->c_b1; g.c_m = d->c_m; g.bias_b1 = d->bias_b1; g.bias_b0 = d->bias_b0;
    g.alpha = d->alpha; g.epilogue = d->epilogue; g.bias_mode = d->bias_mode;
    const bool akc = (d->a_k == 1), bkc = (d->b_k == 1);
    const bool vec = gemm_vec_ok(A, B, d);
    g.vecA = vec; g.vecB = vec;
    g.dropout_p = d->dropout_p; g.seed = d->seed; g.offset = d->offset; g.rbase = rng_base();
    g.splitk = splitk;
    // k_chunk: multiple of the k-tile so slabs start on tile boundaries (and stay float4-aligned)
    g.k_chunk = splitk == 1 ? d->K : ceil_div(ceil_div(d->K, splitk), BKT) * BKT;
    const int nbatch = d->nb0 * d->nb1;
    g.c_split = (int64_t)nbatch * d->M * d->N;
    g.slab = breduce ? 1 : 0;
    g.Bp = nullptr; g.bp_plane = g.bp_b0 = g.bp_b1 = 0;
    g.resid = d->resid;
    SEGX_REQUIRE(!d->resid || (d->epilogue == SEGX_EPI_NONE && !breduce && !d->gmax), "segx_gemm_f32: resid needs a plain epilogue (no GELU, no batch_reduce, no gmax)");
    if (splitk > 1 || breduce) g.C = d->workspace;
    SEGX_REQUIRE(d->tile >= SEGX_TILE_AUTO && d->tile <= SEGX_TILE_SKINNY_NT, "segx_gemm_f32: bad tile %d", d->tile);
    const bool ws_tile = d->tile >= SEGX_TILE_256x128 && d->tile <= SEGX_TILE_WS256x96;
    SEGX_REQUIRE(d->engine >= SEGX_ENGINE_SEL_DEFAULT && d->engine <= SEGX_ENGINE_SEL_BF16X6, "segx_gemm_f32: bad engine selector %d", d->engine);
    const int engine = call_engine(d);
    SEGX_REQUIRE(!ws_tile || engine == SEGX_ENGINE_BF16X6, "segx_gemm_f32: tile %d exists on the bf16x6 engine only", d->tile);
    const int x6_variant = kget(knobs().x6_variant);
    int tile = d->tile;
    if (tile == SEGX_TILE_SKINNY_NT) {
        // the plan's slab count travels as splitk; anything the streaming kernel does not serve quietly takes the planner's tile (like the 96-row tiles)
        const int ssk = skinny_nt_splitk(A, B, d);
        if (ssk > 0 && splitk <= ssk) {
            const int nslabs = splitk * d->nb0 * d->nb1;
            int rc = launch_skinny_nt(A, B, static_cast<float*>(d->workspace), d->M, d->N, d->K, d->nb0, d->nb1, d->a_b0, d->a_b1, d->a_m, d->b_b0, d->b_b1, d->b_n,
                                      nslabs, stream);
            if (rc) return rc;
            const int64_t total = (int64_t)d->M * d->N;
            if (total * 16 <= 512 * 1024 && nslabs >= 16)
                hipLaunchKernelGGL((slab_sum_parts_kernel<16>), dim3((unsigned)((total + 15) / 16)), dim3(256), 0, stream, (const float*)d->workspace, C, g.bias,
                                   d->M, d->N, nslabs, total, d->c_m, d->alpha, d->bias_mode);
            else
                hipLaunchKernelGGL((slab_sum_parts_kernel<4>), dim3((unsigned)((total + 63) / 64)), dim3(256), 0, stream, (const float*)d->workspace, C, g.bias,
                                   d->M, d->N, nslabs, total, d->c_m, d->alpha, d->bias_mode);
            return check_launch("segx_gemm_f32/skinny_nt_reduce");
        }
        tile = SEGX_TILE_AUTO;
    }
    const bool gelu = d->epilogue == SEGX_EPI_GELU;
    // the wave-specialised kernels address an operand through 32-bit byte offsets from a per-item base and take whole 32-k stages only
    const bool ws_ok = gemm_ws_ok(d), lean_ok = gemm_lean_ok(d);
    bool x6 = x6_eligible(engine, d->M, d->N, vec) && (!gelu || akc) &&
              (tile == SEGX_TILE_AUTO || tile == SEGX_TILE_128x128 || tile == SEGX_TILE_64x128 || tile == SEGX_TILE_64x64 || ws_tile);
    if (tile == SEGX_TILE_AUTO) {
        int sk_unused = 1;
        if (x6) plan6(d->M, d->N, d->K, nbatch, gelu, false, splitk, ws_ok, (!akc) + (!bkc), &tile, &sk_unused);
        else plan(d->M, d->N, d->K, nbatch, vec && !gelu, false, splitk, &tile, &sk_unused);
    }
    if (ws_tile && !ws_ok) tile = SEGX_TILE_128x128;
    // the 96-row tiles (channel counts 272 / 160 / 192 / 672 / 960 of the backbone: 3 x 96 = 288 rows cover 272 where 3 x 128 compute 384) stage their 96-row
    // side with the k-contiguous loader only (the row-contiguous one deals 64 / 128 / 256 rows over a workgroup); no fused GELU
    if ((tile == SEGX_TILE_WS96x256 && (!akc || gelu)) || (tile == SEGX_TILE_WS256x96 && (!bkc || gelu))) tile = SEGX_TILE_128x128;
    const bool ws = x6 && ws_ok && tile >= SEGX_TILE_256x128 && tile <= SEGX_TILE_WS256x96;
    if (!vec || (gelu && !ws) || (ws_tile && !x6)) tile = SEGX_TILE_128x128;       // odd shapes / fused GELU: only the default tile (and the wave-specialised ones) are built

    dim3 block(256);
    using Cfg64 = TileCfg<2, 2, 1, 1>; using Cfg128x32 = TileCfg<4, 1, 1, 1>; using Cfg32x128 = TileCfg<1, 4, 1, 1>;
    using Cfg64x128 = TileCfg<2, 2, 1, 2>;
    if (x6) {
        knobs().x6_launches.fetch_add(1, std::memory_order_relaxed);
#ifdef SEGX_NO_LEAN                                     // bench-only A/B build (tools/build_variant.py)
#define SEGX_LEAN4 false
#else
#define SEGX_LEAN4 true
#endif
#define SEGX_LAUNCH6(CFG, AK, BK, E, W)                                                                    \
    do {                                                                                                   \
        g.tiles_m = ceil_div(d->M, CFG::BM); g.tiles_n = ceil_div(d->N, CFG::BN); g.mfast = kget(knobs().tile_walk) ? tile_walk(d->M, d->N, d->K, g.tiles_m) : 0;                          \
        if (SEGX_LEAN4 && lean_ok) hipLaunchKernelGGL((gemm_x6_lean_kernel<CFG, AK, BK, E, W>), dim3(g.tiles_m * g.tiles_n, nbatch, splitk), block, 0, stream, g); \
        else hipLaunchKernelGGL((gemm_x6_kernel<CFG, AK, BK, E, W>), dim3(g.tiles_m * g.tiles_n, nbatch, splitk), block, 0, stream, g); \
    } while (0)
#define SEGX_LAUNCH6_LAYOUT(CFG, W)                                          \
    do {                                                                     \
        if (akc && bkc) SEGX_LAUNCH6(CFG, true, true, SEGX_EPI_NONE, W);      \
        else if (akc && !bkc) SEGX_LAUNCH6(CFG, true, false, SEGX_EPI_NONE, W); \
        else if (!akc && bkc) SEGX_LAUNCH6(CFG, false, true, SEGX_EPI_NONE, W); \
        else SEGX_LAUNCH6(CFG, false, false, SEGX_EPI_NONE, W);               \
    } while (0)
        using Cfg256x128 = TileCfg<2, 2, 4, 2>;
        // persistent launch: one workgroup per CU, a multiple of eight (one run of items per XCD and round).  Variant 1 = consumers at raised wave
        // priority (same results); the ablation variants 2..5 (results are NOT the GEMM) exist in -DSEGX_BENCH builds only (tools/build_variant.py)
#ifdef SEGX_BENCH
#define SEGX_WS_VARIANTS(CFG, AK, BK, E)                                                                   \
        switch (E == SEGX_EPI_NONE && AK && BK ? x6_variant : (x6_variant == 1 ? 1 : 0)) {                 \
        case 1: hipLaunchKernelGGL((gemm_x6ws_kernel<CFG, AK, BK, E, 1>), dim3(G), dim3(512), 0, stream, g); break; \
        case 2: hipLaunchKernelGGL((gemm_x6ws_kernel<CFG, AK, BK, E, (E == SEGX_EPI_NONE && AK && BK) ? 2 : 0>), dim3(G), dim3(512), 0, stream, g); break; \
        case 3: hipLaunchKernelGGL((gemm_x6ws_kernel<CFG, AK, BK, E, (E == SEGX_EPI_NONE && AK && BK) ? 3 : 0>), dim3(G), dim3(512), 0, stream, g); break; \
        case 4: hipLaunchKernelGGL((gemm_x6ws_kernel<CFG, AK, BK, E, (E == SEGX_EPI_NONE && AK && BK) ? 4 : 0>), dim3(G), dim3(512), 0, stream, g); break; \
        case 5: hipLaunchKernelGGL((gemm_x6ws_kernel<CFG, AK, BK, E, (E == SEGX_EPI_NONE && AK && BK) ? 5 : 0>), dim3(G), dim3(512), 0, stream, g); break; \
        default: hipLaunchKernelGGL((gemm_x6ws_kernel<CFG, AK, BK, E, 0>), dim3(G), dim3(512), 0, stream, g); }
#else
#define SEGX_WS_VARIANTS(CFG, AK, BK, E)                                                                   \
        if (x6_variant == 1) hipLaunchKernelGGL((gemm_x6ws_kernel<CFG, AK, BK, E, 1>), dim3(G), dim3(512), 0, stream, g); \
        else hipLaunchKernelGGL((gemm_x6ws_kernel<CFG, AK, BK, E, 0>), dim3(G), dim3(512), 0, stream, g);
#endif
#define SEGX_LAUNCHWS(CFG, AK, BK, E)                                                                      \
    do {                                                                                                   \
        g.tiles_m = ceil_div(d->M, CFG::BM); g.tiles_n = ceil_div(d->N, CFG::BN); g.mfast = kget(knobs().tile_walk) ? tile_walk(d->M, d->N, d->K, g.tiles_m) : 0;                          \
        const int64_t items = (int64_t)g.tiles_m * g.tiles_n * nbatch * splitk;                            \
        SEGX_REQUIRE(items < 2147483647LL - 512, "segx_gemm_f32: too many tiles");                         \
        const int G = (int)i64min(kget(knobs().ws_grid), (items + 7) / 8 * 8);                                              \
        SEGX_WS_VARIANTS(CFG, AK, BK, E)                                                                    \
    } while (0)
#define SEGX_LAUNCHWS_LAYOUT(CFG)                                                          \
    do {                                                                                   \
        if (gelu) { if (bkc) SEGX_LAUNCHWS(CFG, true, true, SEGX_EPI_GELU); else SEGX_LAUNCHWS(CFG, true, false, SEGX_EPI_GELU); } \
        else if (akc && bkc) SEGX_LAUNCHWS(CFG, true, true, SEGX_EPI_NONE);                \
        else if (akc && !bkc) SEGX_LAUNCHWS(CFG, true, false, SEGX_EPI_NONE);              \
        else if (!akc && bkc) SEGX_LAUNCHWS(CFG, false, true, SEGX_EPI_NONE);              \
        else SEGX_LAUNCHWS(CFG, false, false, SEGX_EPI_NONE);                              \
    } while (0)
#define SEGX_LAUNCH6V(V, W)                                                                                \
    do {                                                                                                   \
        g.tiles_m = ceil_div(d->M, Cfg128::BM); g.tiles_n = ceil_div(d->N, Cfg128::BN); g.mfast = kget(knobs().tile_walk) ? tile_walk(d->M, d->N, d->K, g.tiles_m) : 0;                    \
        hipLaunchKernelGGL((gemm_x6_kernel<Cfg128, true, true, SEGX_EPI_NONE, W, V>), dim3(g.tiles_m * g.tiles_n, nbatch, splitk), block, 0, stream, g); \
    } while (0)
        using Cfg128x256 = TileCfg<2, 2, 2, 4>; using Cfg64x256 = TileCfg<2, 2, 1, 4>; using Cfg96x256 = TileCfg<1, 4, 3, 2>; using Cfg256x96 = TileCfg<4, 1, 2, 3>;      // few output channels x many positions (backbone pointwise convolutions)
        // pre-split B operand (segx_x6_presplit): the wave-specialised 256 x 128 / 128 x 256 kernels with a copy-only B loader; anything else ignores the planes
        const bool pre = d->b_planes && ws && !gelu && (tile == SEGX_TILE_256x128 || tile == SEGX_TILE_WS128x256);
        if (pre) {
            g.Bp = static_cast<const unsigned short*>(d->b_planes); g.bp_plane = (int64_t)d->N * d->K; g.bp_b0 = d->bp_b0; g.bp_b1 = d->bp_b1;
#define SEGX_LAUNCHWS_PRE(CFG)                                                                             \
    do {                                                                                                   \
        g.tiles_m = ceil_div(d->M, CFG::BM); g.tiles_n = ceil_div(d->N, CFG::BN); g.mfast = kget(knobs().tile_walk) ? tile_walk(d->M, d->N, d->K, g.tiles_m) : 0;                          \
        const int64_t items = (int64_t)g.tiles_m * g.tiles_n * nbatch * splitk;                            \
        SEGX_REQUIRE(items < 2147483647LL - 512, "segx_gemm_f32: too many tiles");                         \
        const int G = (int)i64min(kget(knobs().ws_grid), (items + 7) / 8 * 8);                             \
        if (akc) hipLaunchKernelGGL((gemm_x6ws_pre_kernel<CFG, true>), dim3(G), dim3(512), 0, stream, g);  \
        else hipLaunchKernelGGL((gemm_x6ws_pre_kernel<CFG, false>), dim3(G), dim3(512), 0, stream, g);     \
    } while (0)
            if (tile == SEGX_TILE_256x128) SEGX_LAUNCHWS_PRE(Cfg256x128); else SEGX_LAUNCHWS_PRE(Cfg128x256);
#undef SEGX_LAUNCHWS_PRE
        }
        else if (tile == SEGX_TILE_256x128) SEGX_LAUNCHWS_LAYOUT(Cfg256x128);
        else if (tile == SEGX_TILE_WS128x128) SEGX_LAUNCHWS_LAYOUT(Cfg128);
        else if (tile == SEGX_TILE_WS128x256 && !gelu) SEGX_LAUNCHWS_LAYOUT(Cfg128x256);
        else if (tile == SEGX_TILE_WS64x256 && !gelu) SEGX_LAUNCHWS_LAYOUT(Cfg64x256);
        else if (tile == SEGX_TILE_WS96x256) { if (bkc) SEGX_LAUNCHWS(Cfg96x256, true, true, SEGX_EPI_NONE); else SEGX_LAUNCHWS(Cfg96x256, true, false, SEGX_EPI_NONE); }
        else if (tile == SEGX_TILE_WS256x96) { if (akc) SEGX_LAUNCHWS(Cfg256x96, true, true, SEGX_EPI_NONE); else SEGX_LAUNCHWS(Cfg256x96, false, true, SEGX_EPI_NONE); }
        else if (gelu) { if (bkc) SEGX_LAUNCH6(Cfg128, true, true, SEGX_EPI_GELU, 3); else SEGX_LAUNCH6(Cfg128, true, false, SEGX_EPI_GELU, 3); }
        else if (x6_variant > 0 && akc && bkc && (tile == SEGX_TILE_128x128 || tile == SEGX_TILE_AUTO)) {
            switch (x6_variant) { case 1: SEGX_LAUNCH6V(1, 3); break; case 6: SEGX_LAUNCH6V(6, 2); break;
#ifdef SEGX_BENCH
                                  case 2: SEGX_LAUNCH6V(2, 3); break; case 3: SEGX_LAUNCH6V(3, 3); break; case 4: SEGX_LAUNCH6V(4, 3); break; case 5: SEGX_LAUNCH6V(5, 3); break;
#endif
                                  default: SEGX_LAUNCH6V(0, 2); break; }      // 7: the product schedule at two waves per SIMD (what the split-early schedule is compared with)
        }
        else if (tile == SEGX_TILE_64x64) SEGX_LAUNCH6_LAYOUT(Cfg64, 5);
        else if (tile == SEGX_TILE_64x128) SEGX_LAUNCH6_LAYOUT(Cfg64x128, 4);
        else SEGX_LAUNCH6_LAYOUT(Cfg128, 3);
#undef SEGX_LAUNCH6
#undef SEGX_LAUNCH6V
#undef SEGX_LAUNCH6_LAYOUT
#undef SEGX_LAUNCHWS
#undef SEGX_WS_VARIANTS
#undef SEGX_LAUNCHWS_LAYOUT
    } else
#define SEGX_LAUNCH(CFG, AK, BK, V, E)                                                                     \
    do {                                                                                                   \
        g.tiles_m = ceil_div(d->M, CFG::BM); g.tiles_n = ceil_div(d->N, CFG::BN); g.mfast = kget(knobs().tile_walk) ? tile_walk(d->M, d->N, d->K, g.tiles_m) : 0;                          \
        hipLaunchKernelGGL((gemm_f32_kernel<CFG, AK, BK, V, E>), dim3(g.tiles_m * g.tiles_n, nbatch, splitk), block, 0, stream, g); \
    } while (0)
#define SEGX_LAUNCH_LAYOUT(CFG, V, E)                                   \
    do {                                                                \
        if (akc && bkc) SEGX_LAUNCH(CFG, true, true, V, E);             \
        else if (akc && !bkc) SEGX_LAUNCH(CFG, true, false, V, E);      \
        else if (!akc && bkc) SEGX_LAUNCH(CFG, false, true, V, E);      \
        else SEGX_LAUNCH(CFG, false, false, V, E);                      \
    } while (0)
    if (d->epilogue == SEGX_EPI_GELU) {
        SEGX_REQUIRE(akc, "segx_gemm_f32: the GELU epilogue is built for a k-contiguous A operand (nn.Linear, attention fusion)");
        if (bkc) { if (vec) SEGX_LAUNCH(Cfg128, true, true, true, SEGX_EPI_GELU); else SEGX_LAUNCH(Cfg128, true, true, false, SEGX_EPI_GELU); }
        else { if (vec) SEGX_LAUNCH(Cfg128, true, false, true, SEGX_EPI_GELU); else SEGX_LAUNCH(Cfg128, true, false, false, SEGX_EPI_GELU); }
    } else if (!vec) {
        SEGX_LAUNCH_LAYOUT(Cfg128, false, SEGX_EPI_NONE);
    } else if (tile == SEGX_TILE_64x64) {
        SEGX_LAUNCH_LAYOUT(Cfg64, true, SEGX_EPI_NONE);
    } else if (tile == SEGX_TILE_128x32) {
        SEGX_LAUNCH_LAYOUT(Cfg128x32, true, SEGX_EPI_NONE);
    } else if (tile == SEGX_TILE_32x128) {
        SEGX_LAUNCH_LAYOUT(Cfg32x128, true, SEGX_EPI_NONE);
    } else if (tile == SEGX_TILE_64x128) {
        SEGX_LAUNCH_LAYOUT(Cfg64x128, true, SEGX_EPI_NONE);
    } else {
        SEGX_LAUNCH_LAYOUT(Cfg128, true, SEGX_EPI_NONE);
    }
    int rc = check_launch("segx_gemm_f32");
    if (rc) return rc;
    if (breduce) {
        // the workspace holds splitk * nbatch slabs of M x N (slab (zk, zb) at (zk * nbatch + zb) * M * N): one deterministic sum over all of them
        const int64_t total = (int64_t)d->M * d->N;
        SEGX_REQUIRE((int64_t)splitk * nbatch < 2147483647LL, "segx_gemm_f32: too many slabs");
        const int nslabs = splitk * nbatch;
        if (total * 16 <= 512 * 1024 && nslabs >= 16)          // few outputs, many slabs: 16 threads per output
            hipLaunchKernelGGL((slab_sum_parts_kernel<16>), dim3((unsigned)((total + 15) / 16)), dim3(256), 0, stream, (const float*)d->workspace, C, g.bias,
                               d->M, d->N, nslabs, total, d->c_m, d->alpha, d->bias_mode);
        else if (total * 4 <= 1024 * 1024 && nslabs >= 4)
            hipLaunchKernelGGL((slab_sum_parts_kernel<4>), dim3((unsigned)((total + 63) / 64)), dim3(256), 0, stream, (const float*)d->workspace, C, g.bias,
                               d->M, d->N, nslabs, total, d->c_m, d->alpha, d->bias_mode);
        else
            SEGX_SPLITK_REDUCE((unsigned)i64min(2048, (total + 255) / 256), stream, (const float*)d->workspace, C, g.bias, d->M, d->N, 1, nslabs, total, (int64_t)0, (int64_t)0, d->c_m, d->alpha,
                               d->bias_mode, (int64_t)0, (int64_t)0, total, (const float*)nullptr);
        return check_launch("segx_gemm_f32/batch_reduce");
    }
    if (splitk > 1) {
        const int64_t total = g.c_split;
        const int blocks = (int)i64min(2048, (total + 255) / 256);
        SEGX_SPLITK_REDUCE(blocks, stream, (const float*)d->workspace, C, g.bias, d->M, d->N, d->nb1, splitk, g.c_split, d->c_b0, d->c_b1, d->c_m, d->alpha, d->bias_mode, d->bias_b1, d->bias_b0,
                           total, (const float*)d->resid);
        rc = check_launch("segx_gemm_f32/splitk_reduce");
    }
    return rc;
}

extern "C" int64_t segx_x6_presplit_elems(int rows, int K, int nb0, int nb1) { return (int64_t)nb0 * nb1 * 3 * rows * K; }

extern "C" int segx_x6_presplit(const float* W, int rows, int K, int64_t s_row, int64_t s_k, int nb0, int nb1, int64_t s_b0, int64_t s_b1, void* planes,
                                void* stream_) {
    using namespace segx;
    SEGX_REQUIRE(W && planes, "segx_x6_presplit: null pointer");
    SEGX_REQUIRE(rows > 0 && K > 0 && K % 8 == 0 && nb0 > 0 && nb1 > 0, "segx_x6_presplit: rows=%d K=%d (a multiple of 8) nb=%dx%d", rows, K, nb0, nb1);
    SEGX_REQUIRE(s_row == 1 || s_k == 1, "segx_x6_presplit: the operand needs a unit stride (s_row=%lld s_k=%lld)", (long long)s_row, (long long)s_k);
    SEGX_REQUIRE((reinterpret_cast<uintptr_t>(planes) & 15) == 0, "segx_x6_presplit: planes must be 16-byte aligned");
    const int64_t per_batch = (int64_t)rows * K;
    const int64_t blocks = (per_batch / 8 + 255) / 256;
    hipLaunchKernelGGL(x6_presplit_kernel, dim3((unsigned)i64min(blocks, 65536), nb0 * nb1), dim3(256), 0, (hipStream_t)stream_, W,
                       static_cast<unsigned short*>(planes), rows, K, s_row, s_k, nb1, s_b0, s_b1, per_batch);
    return check_launch("segx_x6_presplit");
}
