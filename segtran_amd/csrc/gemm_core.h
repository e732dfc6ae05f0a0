// gemm_core.h -- the shared fp32-MFMA tile engine of libsegx: LDS staging, the 64-MFMA k-tile loop and the epilogue.
// gemm.hip instantiates it with dense strided operand loaders; conv3d.hip with implicit-GEMM (im2col-on-the-fly)
// loaders for the B operand.  See gemm.hip for the design notes.
#pragma once
#include <type_traits>
#include "common.h"

namespace segx {

constexpr int BKT = 32;                // k-tile (BKT = 64 measured 5-7 % slower, r01 microbench)
constexpr int KCH = BKT / 4;           // float4 chunks along k of a k-contiguous operand row

// Workgroup tile = 4 waves arranged WM x WN, each wave owning MI x NJ MFMA blocks of 32 x 32:
//   TileCfg<2,2,2,2> 128 x 128  the compute-bound default (one LDS fragment read per MFMA)
//   TileCfg<2,2,1,2>  64 x 128  Cout <= 64 convolutions / M <= 64 GEMMs (half the A-side waste of a 128-row tile)
//   TileCfg<2,2,1,1>  64 x  64  mid-sized pointwise convolutions: more, smaller workgroups when 128 x 128 tiles leave CUs idle
//   TileCfg<4,1,1,1> 128 x  32 / TileCfg<1,4,1,1> 32 x 128  skinny operands (channel counts 24..56): HBM-bound, stream them
template <int WM_, int WN_, int MI_, int NJ_>
struct TileCfg {
    static_assert(WM_ * WN_ == 4, "four waves per workgroup");
    static constexpr int WM = WM_, WN = WN_, MI = MI_, NJ = NJ_;
    static constexpr int BM = WM * MI * 32, BN = WN * NJ * 32;
    static constexpr int NPA = BM * BKT / 4 / 256, NPB = BN * BKT / 4 / 256;     // float4 pieces per thread per operand tile
};
using Cfg128 = TileCfg<2, 2, 2, 2>;
constexpr int BM = Cfg128::BM, BN = Cfg128::BN, NP = Cfg128::NPA;   // the default tile (conv3d.hip loaders)

struct GemmArgs {
    const float* A; const float* B; float* C;
    const float* bias; float* aux; float* gmax;
    int M, N, K, nb1;
    int nbatch;                     // nb0 * nb1 (the persistent kernels of gemm_x6ws.h walk the batch themselves)
    int64_t a_b0, a_b1, a_m, a_k;
    int64_t b_b0, b_b1, b_n, b_k;
    int64_t c_b0, c_b1, c_m;
    int64_t bias_b1, bias_b0;
    float alpha; int epilogue, bias_mode;
    int vecA, vecB;                 // float4 global loads legal (alignment + stride checks done on host)
    int tiles_m, tiles_n;
    float dropout_p; uint64_t seed, offset; const uint64_t* rbase;     // rbase: device-side base added to offset (segx_set_rng_base)
    int k_chunk;                    // split-K: this launch covers k in [z_k*k_chunk, min(K, (z_k+1)*k_chunk))
    int splitk; int64_t c_split;    // slab stride in the workspace
    const unsigned short* Bp; int64_t bp_plane, bp_b0, bp_b1;   // B operand pre-split into three bf16 planes (segx_x6_presplit), element strides; NULL = none
    int slab;                       // 1: write raw slabs to the workspace even when splitk == 1 (batch_reduce: the batch members are slabs too)
    const float* resid;             // C = alpha * A B^T (+ bias) + resid, resid laid out like C (plain epilogue; split-K adds it in the slab reduction)
    int mfast = 0;                  // tile walk inside an XCD's run: 0 = N fastest (neighbours share their A row-panel), 1 = M fastest (they share their B column-panel): tile_walk()
};

// Which way the tiles of one batch member are walked (a speed choice only: every tile is computed the same way whatever its place in the order).
// N fastest keeps ONE A row-panel in the XCD's L2 and streams the B panels past it -- every B panel is then fetched once per ROW of tiles (tiles_m times over
// the launch, from the fabric: the other fetches happen on other XCDs or rounds later).  That is right while B is the small operand.  When A is small enough to live
// in L2 whole (<= 2 MiB of the 4-MiB L2 an XCD owns) and B is the big one -- the pointwise convolutions over 150 528 voxels / 65 536 .. 262 144 pixels, whose
// 'A' is a few hundred filters -- M fastest lets the tiles_m tiles that need a B panel run next to each other on one XCD: B crosses the fabric once
// (r06: 832 x 150 528 x 480 fetched 3.5x its operands with N fastest).
inline int tile_walk(int64_t M, int64_t N, int64_t K, int tiles_m) {
    const int64_t a_bytes = M * K * 4, b_bytes = N * K * 4;
    return (tiles_m > 1 && a_bytes <= (2 << 20) && b_bytes > 2 * a_bytes) ? 1 : 0;
}

// Load this thread's ROWS*BKT/1024 float4 pieces of a ROWS x BKT operand tile into registers.
//  KC = true : operand is k-contiguous;  piece f -> row f / KCH, k-chunk f % KCH
//  KC = false: operand is row-contiguous; piece f -> k-row f / (ROWS/4), row-chunk f % (ROWS/4)
// VEC = true (16-B aligned base, all strides and extents multiples of 4): every float4 is either wholly inside
// or wholly outside the operand, so the load is issued UNCONDITIONALLY from a clamped address and zeroed by a
// select -- no branches, so the loads of a k-tile stay in flight together (a guarded load costs an exec-mask
// branch plus an s_waitcnt vmcnt(0) each).  VEC = false is the slow scalar path for odd shapes (K = 2, Cin = 6 ...).
// The zeroing select is deferred to store_tile (through the returned validity mask): consuming a loaded value
// right after the load would make the compiler wait for it BEFORE the MFMA block and lose the overlap.
template <bool KC, bool VEC, int ROWS>
__device__ __forceinline__ unsigned load_tile(float4 (&r)[ROWS * BKT / 1024], const float* __restrict__ base, int64_t s_row, int64_t s_k,
                                              int row0, int rows, int k0, int kend, int tid) {
    constexpr int NPT = ROWS * BKT / 1024, RC = ROWS / 4;
    unsigned okmask = NPT == 8 ? 0xFFFFFFFFu : ((1u << (4 * NPT)) - 1u);   // bit 4*i+j: element j of piece i is inside the operand
#pragma unroll
    for (int i = 0; i < NPT; ++i) {
        const int f = tid + 256 * i;
        const int row = KC ? row0 + f / KCH : row0 + ((f % RC) << 2);
        const int k = KC ? k0 + ((f % KCH) << 2) : k0 + f / RC;
        float4 v;
        if (VEC) {
            const int rc = KC ? (row < rows ? row : rows - 1) : (row < rows ? row : rows - 4);
            const int kc = KC ? (k < kend ? k : kend - 4) : (k < kend ? k : kend - 1);
            const float* p = KC ? base + (int64_t)rc * s_row + kc : base + (int64_t)kc * s_k + rc;
            v = *reinterpret_cast<const float4*>(p);
            if (!((row < rows) && (k < kend))) okmask &= ~(0xFu << (4 * i));
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

// LDS tile layouts (per operand, chosen by its loader so that the global->LDS copy is always a float4 store):
//   ROWK  (k-contiguous operands): T[row][k], 32 floats per row, UNPADDED; the eight 16-byte chunks of a row are stored at
//         chunk ^ ((row >> 1) & 7).  An MFMA operand fragment for FOUR consecutive k2-steps is ONE ds_read_b128 per lane, and both
//         that read (16 consecutive rows, one chunk index) and the store (2 rows x 8 chunks) touch 64 distinct banks.
//   KROW  (row-contiguous operands): T[k][row], row stride ROWS + 4.  Fragments are four ds_read_b32.
// Within a group of 8 consecutive k the lanes 0..31 hold k = 0..3 and the lanes 32..63 hold k = 4..7 of the fragment registers
// (x, y, z, w); MFMA step j consumes register j of both operands, i.e. the k pairs (j, 4 + j).  Any pairing is exact as long
// as A and B use the same one; it only fixes the (deterministic) order of the fp32 accumulation.
template <int ROWS> struct TileFloats { static constexpr int value = BKT * (ROWS + 4); };      // >= ROWS * BKT (ROWK)
__device__ __forceinline__ int rowk_off(int row, int chunk) { return row * BKT + ((chunk ^ ((row >> 1) & 7)) << 2); }

template <bool KC, int ROWS>
__device__ __forceinline__ void store_tile(float4 (&r)[ROWS * BKT / 1024], unsigned okmask, float* __restrict__ T, int tid) {
    constexpr int NPT = ROWS * BKT / 1024, RC = ROWS / 4;
    if (okmask != (NPT == 8 ? 0xFFFFFFFFu : ((1u << (4 * NPT)) - 1u))) {     // only tiles on an operand edge pay for the selects
#pragma unroll
        for (int i = 0; i < NPT; ++i) {
            const unsigned mk = okmask >> (4 * i);
            r[i].x = (mk & 1u) ? r[i].x : 0.f; r[i].y = (mk & 2u) ? r[i].y : 0.f;
            r[i].z = (mk & 4u) ? r[i].z : 0.f; r[i].w = (mk & 8u) ? r[i].w : 0.f;
        }
    }
#pragma unroll
    for (int i = 0; i < NPT; ++i) {
        const int f = tid + 256 * i;
        if (KC) *reinterpret_cast<float4*>(T + rowk_off(f / KCH, f % KCH)) = r[i];                     // ROWK: [row][k], swizzled chunks
        else *reinterpret_cast<float4*>(T + (f / RC) * (ROWS + 4) + ((f % RC) << 2)) = r[i];          // KROW: [k][row]
    }
}
// this lane's fragment registers (4 consecutive k2-steps) of row `row` for the k-group g (8 consecutive k)
template <bool ROWK, int ROWS>
__device__ __forceinline__ float4 load_frag(const float* __restrict__ T, int row, int g, int khalf) {
    if (ROWK) return *reinterpret_cast<const float4*>(T + rowk_off(row, g * 2 + khalf));
    const float* p = T + (g * 8 + khalf * 4) * (ROWS + 4) + row;
    return make_float4(p[0], p[ROWS + 4], p[2 * (ROWS + 4)], p[3 * (ROWS + 4)]);
}

// Workgroup -> tile map.  The dispatcher places workgroup b on XCD b % 8 (observed, used for speed only): remap so
// that each XCD owns a CONTIGUOUS run of tiles (bijective for any tile count), and walk N fastest inside the run, so
// the workgroups sharing an XCD's private 4-MiB L2 also share their A row-panels / B column-panels.
__device__ __forceinline__ int xcd_tile(int wg, int ntiles) {
    const int q = ntiles >> 3, r = ntiles & 7, xcd = wg & 7, idx = wg >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

// Operand loader concept (ROWS = the tile's BM for the A side, BN for the B side, NPT = ROWS*BKT/1024):
// `unsigned load(float4 (&r)[NPT], int k0, int kend, int tid) const` fetches this thread's 4*NPT floats of the ROWS x BKT tile
// starting at k0 and returns their validity mask; `store(r, mask, T, tid)` writes them (zeroing the invalid ones) into the
// LDS tile T (layout ROWK / KROW as the loader declares).
template <bool KC, bool VEC, int ROWS = 128>
struct DenseLoader {
    static constexpr bool ROWK = KC;                 // LDS layout of this operand's tile (see store_tile)
    const float* base; int64_t s_row, s_k; int row0, rows;
    __device__ __forceinline__ unsigned load(float4 (&r)[ROWS * BKT / 1024], int k0, int kend, int tid) const {
        return load_tile<KC, VEC, ROWS>(r, base, s_row, s_k, row0, rows, k0, kend, tid);
    }
    __device__ __forceinline__ void store(float4 (&r)[ROWS * BKT / 1024], unsigned okmask, float* T, int tid) const {
        store_tile<KC, ROWS>(r, okmask, T, tid);
    }
};

struct TileCoord { int m0, n0, zb, zk, z0, z1, kbeg, kend; };
template <class Cfg = Cfg128>
__device__ __forceinline__ TileCoord tile_coord(const GemmArgs& g) {
    TileCoord t;
    const int tile = xcd_tile(blockIdx.x, g.tiles_m * g.tiles_n);
    int tm, tn;
    if (g.mfast) { tn = tile / g.tiles_m; tm = tile - tn * g.tiles_m; }
    else { tm = tile / g.tiles_n; tn = tile - tm * g.tiles_n; }
    t.zb = blockIdx.y; t.zk = blockIdx.z;
    t.z0 = t.zb / g.nb1; t.z1 = t.zb - t.z0 * g.nb1;
    t.m0 = tm * Cfg::BM; t.n0 = tn * Cfg::BN;
    t.kbeg = t.zk * g.k_chunk;
    t.kend = (t.kbeg + g.k_chunk < g.K) ? t.kbeg + g.k_chunk : g.K;
    return t;
}

// acc += A_tile . B_tile^T over k in [kbeg, kend): the k-tile pipeline shared by every MFMA kernel of the library
// LDS: two buffers per operand (128 x 128: 2 x 2 x BKT x 132 x 4 B = 67.6 KB per workgroup -> two workgroups per CU)
template <class Cfg>
struct TileLdsT { float A[2][TileFloats<Cfg::BM>::value]; float B[2][TileFloats<Cfg::BN>::value]; };
using TileLds = TileLdsT<Cfg128>;

template <class Cfg = Cfg128, class LA, class LB>
__device__ __forceinline__ void gemm_mainloop(f32x16 (&acc)[Cfg::MI][Cfg::NJ], const LA& la, const LB& lb, int kbeg, int kend, TileLdsT<Cfg>& S) {
    constexpr int MI = Cfg::MI, NJ = Cfg::NJ, NG = BKT / 8;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / Cfg::WN, wn = wave % Cfg::WN;
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    if (kbeg >= kend) return;                        // an empty split-K slab just writes zeros
    // Pipeline (per k-tile t, h = t & 1 static after unrolling by two):
    //   global loads of tile t+2 -> register set h        (two tiles ahead: rides out lock-step fetch latency)
    //   LDS stores of tile t+1 (register set h^1) -> LDS buffer h^1   (issued BEFORE the MFMAs, so they drain under them)
    //   MI*NJ*16 MFMAs on LDS buffer h; the fragments of k-group g+1 (four k2-steps) are fetched under the MFMAs of group g
    //   ONE barrier (buffer h may be overwritten / buffer h^1 is complete)
    float4 ra[2][Cfg::NPA], rb[2][Cfg::NPB];
    unsigned oka[2] = {0u, 0u}, okb[2] = {0u, 0u};
    oka[0] = la.load(ra[0], kbeg, kend, tid);
    okb[0] = lb.load(rb[0], kbeg, kend, tid);
    if (kbeg + BKT < kend) { oka[1] = la.load(ra[1], kbeg + BKT, kend, tid); okb[1] = lb.load(rb[1], kbeg + BKT, kend, tid); }
    la.store(ra[0], oka[0], S.A[0], tid);
    lb.store(rb[0], okb[0], S.B[0], tid);
    __syncthreads();
    const int arow = wm * (32 * MI) + (lane & 31), brow = wn * (32 * NJ) + (lane & 31), kl = lane >> 5;
    for (int k0 = kbeg; k0 < kend; k0 += 2 * BKT) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int kt = k0 + h * BKT;
            if (kt < kend) {
                const float* As = S.A[h]; const float* Bs = S.B[h];
                if (kt + 2 * BKT < kend) { oka[h] = la.load(ra[h], kt + 2 * BKT, kend, tid); okb[h] = lb.load(rb[h], kt + 2 * BKT, kend, tid); }
                if (kt + BKT < kend) { la.store(ra[h ^ 1], oka[h ^ 1], S.A[h ^ 1], tid); lb.store(rb[h ^ 1], okb[h ^ 1], S.B[h ^ 1], tid); }
                float4 a[MI], b[NJ];
#pragma unroll
                for (int i = 0; i < MI; ++i) a[i] = load_frag<LA::ROWK, Cfg::BM>(As, arow + 32 * i, 0, kl);
#pragma unroll
                for (int j = 0; j < NJ; ++j) b[j] = load_frag<LB::ROWK, Cfg::BN>(Bs, brow + 32 * j, 0, kl);
#pragma unroll
                for (int g = 0; g < NG; ++g) {
                    float4 na[MI], nb[NJ];
#pragma unroll
                    for (int i = 0; i < MI; ++i) na[i] = (g + 1 < NG) ? load_frag<LA::ROWK, Cfg::BM>(As, arow + 32 * i, g + 1, kl) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                    for (int j = 0; j < NJ; ++j) nb[j] = (g + 1 < NG) ? load_frag<LB::ROWK, Cfg::BN>(Bs, brow + 32 * j, g + 1, kl) : make_float4(0.f, 0.f, 0.f, 0.f);
                    __builtin_amdgcn_sched_barrier(0);   // keep the fragment prefetch ahead of this group's MFMAs
#pragma unroll
                    for (int i = 0; i < MI; ++i)
#pragma unroll
                        for (int j = 0; j < NJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].x, b[j].x, acc[i][j], 0, 0, 0);
#pragma unroll
                    for (int i = 0; i < MI; ++i)
#pragma unroll
                        for (int j = 0; j < NJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].y, b[j].y, acc[i][j], 0, 0, 0);
#pragma unroll
                    for (int i = 0; i < MI; ++i)
#pragma unroll
                        for (int j = 0; j < NJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].z, b[j].z, acc[i][j], 0, 0, 0);
#pragma unroll
                    for (int i = 0; i < MI; ++i)
#pragma unroll
                        for (int j = 0; j < NJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].w, b[j].w, acc[i][j], 0, 0, 0);
#pragma unroll
                    for (int i = 0; i < MI; ++i) a[i] = na[i];
#pragma unroll
                    for (int j = 0; j < NJ; ++j) b[j] = nb[j];
                }
                __syncthreads();
            }
        }
    }
}

// MFMA C layout: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
// VST: 16-byte stores through quad_transpose4 where the output allows (the 4-wave kernels: +3..4 % on short contractions; the wave-specialised
// kernels keep 4-byte stores -- their one consumer wave per SIMD pays the transposes' dependent DPP chains in full, r03_z: -4..9 %)
template <int EPI, class Cfg = Cfg128, bool VST = true>
__device__ __forceinline__ void gemm_epilogue(const f32x16 (&acc)[Cfg::MI][Cfg::NJ], const GemmArgs& g, const TileCoord& t) {
    constexpr int MI = Cfg::MI, NJ = Cfg::NJ;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wm = wave / Cfg::WN, wn = wave % Cfg::WN;
    const int m0 = t.m0, n0 = t.n0, zb = t.zb, zk = t.zk, z0 = t.z0, z1 = t.z1;
    const bool split = g.splitk > 1 || g.slab;
    float* C = split ? g.C + (int64_t)zk * g.c_split + (int64_t)zb * g.M * g.N : g.C + z0 * g.c_b0 + z1 * g.c_b1;
    const int64_t ldc = split ? g.N : g.c_m;
    const float alpha = split ? 1.0f : g.alpha;
    const float* bias = (g.bias && !split) ? g.bias + z0 * g.bias_b0 + z1 * g.bias_b1 : nullptr;
    const bool bias_n = bias && g.bias_mode == SEGX_BIAS_N, bias_m = bias && g.bias_mode == SEGX_BIAS_M;
    float* AUX = (EPI == SEGX_EPI_GELU) ? g.aux + z0 * g.c_b0 + z1 * g.c_b1 : nullptr;
    const float* RES = (EPI == SEGX_EPI_NONE && g.resid && !split) ? g.resid + z0 * g.c_b0 + z1 * g.c_b1 : nullptr;     // wave-uniform
    const float inv_keep = g.dropout_p > 0.f ? 1.0f / (1.0f - g.dropout_p) : 1.0f;
    const uint64_t roff = g.offset + ((EPI == SEGX_EPI_GELU && g.dropout_p > 0.f && g.rbase) ? *g.rbase : 0);
    const bool full = (m0 + Cfg::BM <= g.M) && (n0 + Cfg::BN <= g.N);
    const uint64_t ebase = roff + (uint64_t)(z0 * g.c_b0 + z1 * g.c_b1);         // stream position of this matrix's element (0, 0)
    const bool quad_rng = ((ebase | (uint64_t)ldc) & 3) == 0;                     // a quad's 4 columns share one Philox counter (wave-uniform)
    // 16-byte stores (quad_transpose4): row length and extent multiples of 4 (a quad of columns is wholly inside or outside), 16-byte aligned bases
    const bool vec_st = VST && (((uint64_t)ldc | (uint64_t)g.N) & 3) == 0 && ((reinterpret_cast<uintptr_t>(C) | (EPI == SEGX_EPI_GELU ? reinterpret_cast<uintptr_t>(AUX) : 0) | reinterpret_cast<uintptr_t>(RES)) & 15) == 0;     // RES is read as float4 too (null: 0)
    float vmax = 0.f;
#pragma unroll
    for (int i = 0; i < MI; ++i) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int col = n0 + wn * (32 * NJ) + j * 32 + (lane & 31);
            const bool col_ok = full || col < g.N;
            const float bn = (bias_n && col_ok) ? bias[col] : 0.f;

            const int rbase = m0 + wm * (32 * MI) + i * 32 + 4 * (lane >> 5);
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                float keep[4] = {1.f, 1.f, 1.f, 1.f};
                if (EPI == SEGX_EPI_GELU && g.dropout_p > 0.f) {                      // element offset in C: what segx_gelu_bwd regenerates
                    if (quad_rng) {
                        const u32x4 pr = philox4(g.seed, (ebase + (uint64_t)(rbase + 8 * rg + (lane & 3)) * ldc + (col & ~3)) >> 2);
                        keep[0] = keep_of(quad_pick<0>(pr, lane & 3), g.dropout_p, inv_keep); keep[1] = keep_of(quad_pick<1>(pr, lane & 3), g.dropout_p, inv_keep);
                        keep[2] = keep_of(quad_pick<2>(pr, lane & 3), g.dropout_p, inv_keep); keep[3] = keep_of(quad_pick<3>(pr, lane & 3), g.dropout_p, inv_keep);
                    } else {
#pragma unroll
                        for (int q = 0; q < 4; ++q) keep[q] = dropout_scale(g.seed, 0, ebase + (uint64_t)(rbase + 8 * rg + q) * ldc + col, g.dropout_p, inv_keep);
                    }
                }
                float pre[4], out[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int r = rg * 4 + q, row = rbase + q + 8 * rg;
                    const bool ok = full || (col_ok && row < g.M);
                    float v = acc[i][j][r] * alpha + bn;
                    if (bias_m) v += ok ? bias[row] : 0.f;
                    pre[q] = v;
                    if (EPI == SEGX_EPI_GELU) v = gelu_erf(v) * keep[q];
                    out[q] = v;
                    if (ok) vmax = fmaxf(vmax, v);
                    if (!vec_st && ok) {
                        if (EPI == SEGX_EPI_GELU) AUX[(int64_t)row * ldc + col] = pre[q];
                        C[(int64_t)row * ldc + col] = RES ? v + RES[(int64_t)row * ldc + col] : v;
                    }
                }
                if (vec_st) {                              // lane (quad, i): row rbase + 8 rg + i, columns (col & ~3) .. + 3
                    const int row = rbase + 8 * rg + (lane & 3), col0 = col & ~3;
                    const bool ok = full || (col0 < g.N && row < g.M);
                    quad_transpose4(out, lane & 1, lane & 2);
                    if (EPI == SEGX_EPI_GELU) {
                        quad_transpose4(pre, lane & 1, lane & 2);
                        if (ok) *reinterpret_cast<float4*>(AUX + (int64_t)row * ldc + col0) = make_float4(pre[0], pre[1], pre[2], pre[3]);
                    }
                    if (ok) {
                        if (RES) { const float4 rv = *reinterpret_cast<const float4*>(RES + (int64_t)row * ldc + col0); out[0] += rv.x; out[1] += rv.y; out[2] += rv.z; out[3] += rv.w; }
                        *reinterpret_cast<float4*>(C + (int64_t)row * ldc + col0) = make_float4(out[0], out[1], out[2], out[3]);
                    }
                }
            }
        }
    }
    if (g.gmax && !split) {
        vmax = wave_max(vmax);
        if (lane == 0) atomicMax(reinterpret_cast<int*>(g.gmax), __float_as_int(vmax));   // vmax >= 0: int order == float order
    }
}

// ---- host-side cost model shared by the GEMM and convolution planners (see gemm.hip) ---------------------------------
struct TileInfo { int id, bm, bn, wg_per_cu; float ktile_us, fixed_us; };   // resident workgroups per CU (LDS bound)
static const TileInfo kTiles[] = {{SEGX_TILE_128x128, 128, 128, 2, 4.6f, 4.0f},
                                  {SEGX_TILE_64x128, 64, 128, 3, 3.4f, 2.5f},
                                  {SEGX_TILE_64x64, 64, 64, 4, 2.4f, 1.5f},
                                  {SEGX_TILE_128x32, 128, 32, 3, 2.4f, 1.5f},
                                  {SEGX_TILE_32x128, 32, 128, 3, 2.4f, 1.5f}};
inline const TileInfo& tile_info(int tile) {
    for (const TileInfo& t : kTiles) if (t.id == tile) return t;
    return kTiles[0];
}
inline double model_us(const TileInfo& ti, int M, int N, int K, int nbatch, int sk) {
    const int64_t tiles = (int64_t)ceil_div(M, ti.bm) * ceil_div(N, ti.bn) * nbatch * sk, slots = 256 * ti.wg_per_cu;
    const int kt = ceil_div(ceil_div(K, sk), BKT);
    // rounds of workgroups: a partial last round costs at least a third of a round (its workgroups still run start to end, but
    // on emptier CUs); a grid below one round runs as long as its busiest CU, whose workgroups gain little from the free slots
    const int64_t full = tiles / slots, rem = tiles % slots;
    double rounds;
    if (full == 0) rounds = 0.6 + 0.4 * (double)((rem + 255) / 256) / ti.wg_per_cu;   // a workgroup alone on its CU is only ~1.25x faster
    else rounds = (double)full + (rem ? 0.35 + 0.65 * (double)rem / (double)slots : 0.0);
    return rounds * (kt * ti.ktile_us + ti.fixed_us) + (sk == 1 ? 0.0 : (double)sk * M * N * nbatch * 8.0 / 3.0e6);
}
inline int best_splitk(const TileInfo& ti, int M, int N, int K, int nbatch, double* t_out) {
    int best = 1; double best_t = model_us(ti, M, N, K, nbatch, 1);
    const int64_t tiles = (int64_t)ceil_div(M, ti.bm) * ceil_div(N, ti.bn) * nbatch;
    if (tiles < 4 * 256 * ti.wg_per_cu && K >= 1024)
        for (int sk = 2; sk <= 128 && K / sk >= 256; ++sk) {
            const double t = model_us(ti, M, N, K, nbatch, sk);
            if (t < best_t * 0.97) { best = sk; best_t = t; }
        }
    *t_out = best_t;
    return best;
}

// Split-K second stage: C = alpha * sum_s slab[s] (+ bias), deterministic slab order.
// r04-g: the slabs of an output element are requested four at a time and added in slab order (the loop used to wait for every slab before asking for
// the next: a 24-slab reduction of a small output took 32 us), and where the layout allows it a thread owns four consecutive columns (16-byte accesses).
// VEC: N % 4 == 0, every base 16-byte aligned, every stride a multiple of 4 (checked on the host).  The sums are the same numbers in the same order.
template <bool VEC>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ ws, float* __restrict__ C, const float* __restrict__ bias,
                                                            int M, int N, int nb1, int splitk, int64_t c_split,
                                                            int64_t c_b0, int64_t c_b1, int64_t c_m, float alpha,
                                                            int bias_mode, int64_t bias_b1, int64_t bias_b0, int64_t total, const float* __restrict__ resid) {
    constexpr int W = VEC ? 4 : 1;
    using V = typename std::conditional<VEC, f32x4, float>::type;
    const int64_t units = total / W;
    for (int64_t u = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; u < units; u += (int64_t)gridDim.x * blockDim.x) {
        const int64_t idx = u * W;
        const int col = (int)(idx % N);
        const int64_t t = idx / N;
        const int row = (int)(t % M);
        const int zb = (int)(t / M);
        V s = V(0.f);
        const float* w = ws + idx;
        int k = 0;
        for (; k + 4 <= splitk; k += 4) {
            const V a0 = *reinterpret_cast<const V*>(w + (int64_t)k * c_split), a1 = *reinterpret_cast<const V*>(w + (int64_t)(k + 1) * c_split);
            const V a2 = *reinterpret_cast<const V*>(w + (int64_t)(k + 2) * c_split), a3 = *reinterpret_cast<const V*>(w + (int64_t)(k + 3) * c_split);
            s += a0; s += a1; s += a2; s += a3;
        }
        if (k + 2 <= splitk) {
            const V a0 = *reinterpret_cast<const V*>(w + (int64_t)k * c_split), a1 = *reinterpret_cast<const V*>(w + (int64_t)(k + 1) * c_split);
            s += a0; s += a1; k += 2;
        }
        if (k < splitk) s += *reinterpret_cast<const V*>(w + (int64_t)k * c_split);
        s *= alpha;
        const int z0 = zb / nb1, z1 = zb - z0 * nb1;
        const float* bp = bias + z0 * bias_b0 + z1 * bias_b1;
        if (bias_mode == SEGX_BIAS_N) s += *reinterpret_cast<const V*>(bp + col);
        else if (bias_mode == SEGX_BIAS_M) s += V(bp[row]);
        const int64_t o = z0 * c_b0 + z1 * c_b1 + (int64_t)row * c_m + col;
        if (resid) s += *reinterpret_cast<const V*>(resid + o);
        *reinterpret_cast<V*>(C + o) = s;
    }
}
// host side: may the reduction use 16-byte accesses?
static bool splitk_vec_ok(const void* ws, const void* C, const void* bias, const void* resid, int N, int64_t c_split, int64_t c_b0, int64_t c_b1, int64_t c_m,
                          int bias_mode, int64_t bias_b1, int64_t bias_b0) {
    auto al = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    if ((N & 3) || (c_split & 3) || (c_b0 & 3) || (c_b1 & 3) || (c_m & 3) || !al(ws) || !al(C) || (resid && !al(resid))) return false;
    if (bias_mode == SEGX_BIAS_N && (!al(bias) || (bias_b1 & 3) || (bias_b0 & 3))) return false;
    return true;
}
#define SEGX_SPLITK_REDUCE(blocks, stream, ws, C, bias, M, N, nb1, splitk, c_split, c_b0, c_b1, c_m, alpha, bias_mode, bias_b1, bias_b0, total, resid)               \
    do {                                                                                                                                                            \
        if (splitk_vec_ok(ws, C, bias, resid, N, c_split, c_b0, c_b1, c_m, bias_mode, bias_b1, bias_b0))                                                            \
            hipLaunchKernelGGL((splitk_reduce_kernel<true>), dim3((unsigned)i64min(2048, ((total) / 4 + 255) / 256)), dim3(256), 0, stream, ws, C, bias, M, N, nb1,  \
                               splitk, c_split, c_b0, c_b1, c_m, alpha, bias_mode, bias_b1, bias_b0, total, resid);                                                 \
        else hipLaunchKernelGGL((splitk_reduce_kernel<false>), dim3(blocks), dim3(256), 0, stream, ws, C, bias, M, N, nb1, splitk, c_split, c_b0, c_b1, c_m, alpha, \
                                bias_mode, bias_b1, bias_b0, total, resid);                                                                                         \
    } while (0)

}  // namespace segx
