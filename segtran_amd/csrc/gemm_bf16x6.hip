// gemm_bf16x6.hip -- EXPERIMENTAL, not on the default path (segx.SegxLib.use_bf16x6 = False): the batched strided GEMM of gemm.hip
// evaluated on the bf16 matrix core with fp32-equivalent accuracy.
//
// Both fp32 operands are split ONCE per call into three bf16 planes, x = hi + mid + lo exactly (8 + 8 + 8 significant bits), stored
// k-contiguous [plane][batch][row][Kp] with rows padded to the 128-row tile and K to the 32-k tile (zeros), so the tile kernel needs
// no edge handling and no layout variants: every operand layout of gemm.hip (k-contiguous or row-contiguous, any batch strides,
// batch-broadcast operands) is resolved by the split pass.  The tile kernel issues six v_mfma_f32_32x32x16_bf16 per 32 x 32 block
// and 16 k -- hi.lo, lo.hi, mid.mid, hi.mid, mid.hi, hi.hi (small terms first); the three dropped terms (mid.lo, lo.mid, lo.lo) are
// below the rounding of the fp32 accumulation.  Measured on the box with the standalone prototype of the same structure
// (tools/gemm_bf16x6_proto.hip, profiles/r01_l_bf16x6_proto.txt): 24576 x 1792 x 1792 at 146 TFLOP/s fp32-equivalent against
// 115-124 for the fp32-MFMA engine, max error / max |C| 1.2e-6 against fp64 (the fp32 MFMA itself: 1.0e-6).
//
// Structure (the prototype's v1): 128 x 128 x 32 tile, 4 waves x (2 x 2) blocks, ONE 48-KB LDS buffer with register prefetch of the
// next k-tile (152 VGPRs -> three workgroups per CU; a double-buffered 16-k variant was 27 % slower), [row][32 k] bf16 rows of 64 B
// whose 16-B chunk index is XOR-swizzled by (row >> 1) & 3 (conflict-free ds_read_b128), XCD-contiguous tile map and the epilogue of
// gemm_core.h (alpha, bias per column / row / (z0, z1), GELU + dropout, running max); split-K over 32-k tile boundaries with the slab
// layout and the deterministic reduction of gemm.hip (splitk_reduce_kernel).
#include "gemm_core.h"

namespace segx {

typedef short bf16x8 __attribute__((ext_vector_type(8)));

union FloatBits { float f; unsigned u; };
__device__ __forceinline__ unsigned short f32_to_bf16_rne(float f) { FloatBits x; x.f = f; x.u += 0x7FFFu + ((x.u >> 16) & 1u); return (unsigned short)(x.u >> 16); }
__device__ __forceinline__ float bf16_to_f32(unsigned short h) { FloatBits x; x.u = (unsigned)h << 16; return x.f; }

// ---- split pass ------------------------------------------------------------------------------------------------------------
// X(z0, z1, row, k) = X[z0*s_b0 + z1*s_b1 + row*s_row + k*s_k]  ->  P[plane][z][row][k8..k8+7], z = z0*nz1 + z1 over the operand's OWN
// batch extents (a broadcast batch dimension has extent 1 here), rows < RP, k < Kp; outside (rows, K): zeros.
// ROWFAST: consecutive threads take consecutive rows (row-contiguous operands) instead of consecutive k-chunks.
struct SplitArgs {
    const float* X; unsigned short* P;
    int rows, K, RP, Kp, nz1; int64_t nz;
    int64_t s_b0, s_b1, s_row, s_k, plane;
};
template <bool ROWFAST>
__global__ __launch_bounds__(256) void split3_kernel(SplitArgs a) {
    const int kc = a.Kp >> 3;
    const int64_t per_z = (int64_t)a.RP * kc, total = a.nz * per_z;
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
        const int64_t z = t / per_z; const int64_t e = t - z * per_z;
        const int row = ROWFAST ? (int)(e % a.RP) : (int)(e / kc), c = ROWFAST ? (int)(e / a.RP) : (int)(e % kc);
        const int z0 = (int)(z / a.nz1), z1 = (int)(z - (int64_t)z0 * a.nz1);
        const float* src = a.X + z0 * a.s_b0 + z1 * a.s_b1 + (int64_t)row * a.s_row + (int64_t)(c * 8) * a.s_k;
        union { unsigned short s[8]; uint4 q; } h, m, l;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const bool ok = row < a.rows && c * 8 + j < a.K;
            const float x = ok ? src[j * a.s_k] : 0.f;
            h.s[j] = f32_to_bf16_rne(x); const float r1 = x - bf16_to_f32(h.s[j]);
            m.s[j] = f32_to_bf16_rne(r1); const float r2 = r1 - bf16_to_f32(m.s[j]);
            l.s[j] = f32_to_bf16_rne(r2);
        }
        const int64_t o = (z * a.RP + row) * (int64_t)a.Kp + c * 8;
        *reinterpret_cast<uint4*>(a.P + o) = h.q;
        *reinterpret_cast<uint4*>(a.P + a.plane + o) = m.q;
        *reinterpret_cast<uint4*>(a.P + 2 * a.plane + o) = l.q;
    }
}

// ---- tile kernel -------------------------------------------------------------------------------------------------------------
struct PlaneGeom { const unsigned short* PA; const unsigned short* PB; int Kp; int64_t planeA, planeB, pa_b0, pa_b1, pb_b0, pb_b1, zstrideA, zstrideB; };
__device__ __forceinline__ int bf_lds_off(int row, int chunk) { return row * 64 + ((chunk ^ ((row >> 1) & 3)) << 4); }   // bytes in one plane tile

template <int EPI>
__global__ __launch_bounds__(256) SEGX_MIN_WAVES_PER_SIMD(2) void gemm_bf16x6_kernel(GemmArgs g, PlaneGeom pg) {
    constexpr int TILE_BYTES = 128 * 64;
    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * 3 * TILE_BYTES];                      // [operand][plane][row][64 B] = 48 KB
    const TileCoord t = tile_coord<Cfg128>(g);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
    const int K = pg.Kp;
    const unsigned short* PA = pg.PA + (t.z0 * pg.pa_b0 + t.z1 * pg.pa_b1) + (int64_t)t.m0 * K;
    const unsigned short* PB = pg.PB + (t.z0 * pg.pb_b0 + t.z1 * pg.pb_b1) + (int64_t)t.n0 * K;
    // global -> register staging of the next k-tile: 3 planes x 2 uint4 per operand per thread, as twelve NAMED registers (arrays of
    // uint4 ended up in scratch in the prototype)
    uint4 a00, a01, a10, a11, a20, a21, b00, b01, b10, b11, b20, b21;
    const int row0 = tid >> 2, row1 = (tid + 256) >> 2, chk = tid & 3;
    const int so0 = bf_lds_off(row0, chk), so1 = bf_lds_off(row1, chk);
#define SEGX_BF_GL1(p, row, RA, RB, k0)                                                                    \
    RA = *reinterpret_cast<const uint4*>(PA + (p) * pg.planeA + (int64_t)(row) * K + (k0) + chk * 8);       \
    RB = *reinterpret_cast<const uint4*>(PB + (p) * pg.planeB + (int64_t)(row) * K + (k0) + chk * 8);
#define SEGX_BF_GLOAD(k0)                                                                                              \
    SEGX_BF_GL1(0, row0, a00, b00, k0) SEGX_BF_GL1(0, row1, a01, b01, k0) SEGX_BF_GL1(1, row0, a10, b10, k0)             \
    SEGX_BF_GL1(1, row1, a11, b11, k0) SEGX_BF_GL1(2, row0, a20, b20, k0) SEGX_BF_GL1(2, row1, a21, b21, k0)
#define SEGX_BF_LS1(p, so, RA, RB)                                                          \
    *reinterpret_cast<uint4*>(lds + (0 * 3 + (p)) * TILE_BYTES + (so)) = RA;                 \
    *reinterpret_cast<uint4*>(lds + (1 * 3 + (p)) * TILE_BYTES + (so)) = RB;
#define SEGX_BF_LSTORE()                                                                                    \
    SEGX_BF_LS1(0, so0, a00, b00) SEGX_BF_LS1(0, so1, a01, b01) SEGX_BF_LS1(1, so0, a10, b10)                \
    SEGX_BF_LS1(1, so1, a11, b11) SEGX_BF_LS1(2, so0, a20, b20) SEGX_BF_LS1(2, so1, a21, b21)
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    // split-K: this workgroup covers the 32-k tiles [kt0, KT) of its slab (k_chunk is a multiple of 32; the padded planes make the last
    // partial tile of the operand a full one)
    const int kt0 = t.kbeg / 32, KT = (t.kend + 31) / 32;
    SEGX_BF_GLOAD(kt0 * 32)
    for (int kt = kt0; kt < KT; ++kt) {
        __syncthreads();
        SEGX_BF_LSTORE()
        __syncthreads();
        const int kn = (kt + 1 < KT ? kt + 1 : kt) * 32;             // branch-free: the last iteration re-reads its own tile
        SEGX_BF_GLOAD(kn)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int chunk = 2 * s + (lane >> 5);                  // lane -> (row lane & 31, the 8 k of half lane >> 5 of this 16-k step)
            bf16x8 a[2][3], b[2][3];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int p = 0; p < 3; ++p) {
                    a[i][p] = *reinterpret_cast<const bf16x8*>(lds + (0 * 3 + p) * TILE_BYTES + bf_lds_off(wm * 64 + i * 32 + (lane & 31), chunk));
                    b[i][p] = *reinterpret_cast<const bf16x8*>(lds + (1 * 3 + p) * TILE_BYTES + bf_lds_off(wn * 64 + i * 32 + (lane & 31), chunk));
                }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    f32x16 c = acc[i][j];
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][2], c, 0, 0, 0);     // hi . lo
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][2], b[j][0], c, 0, 0, 0);     // lo . hi
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], b[j][1], c, 0, 0, 0);     // mid . mid
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][1], c, 0, 0, 0);     // hi . mid
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], b[j][0], c, 0, 0, 0);     // mid . hi
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][0], c, 0, 0, 0);     // hi . hi
                    acc[i][j] = c;
                }
        }
    }
#undef SEGX_BF_GL1
#undef SEGX_BF_GLOAD
#undef SEGX_BF_LS1
#undef SEGX_BF_LSTORE
    gemm_epilogue<EPI, Cfg128>(acc, g, t);
}

// ---- wide variant (variant 2): 128 x 256 x 16 tile, a wave owns 2 x 4 blocks -------------------------------------------------------------
// Motivation: the 128 x 128 kernel is LDS-port bound (a 2 x 2-block wave re-uses each fragment twice).  Here a wave
// owns 64 x 128: 18 fragment reads per 48 MFMAs instead of 24; 16-k tiles keep the staging at 9 uint4 per thread; LDS is double-buffered
// (2 x 36 KB) with ONE barrier per k-tile.  Not yet timed on the device.
using CfgWide = TileCfg<2, 2, 2, 4>;
__device__ __forceinline__ int bf_lds_off16(int row, int chunk) { return row * 32 + ((chunk ^ ((row >> 2) & 1)) << 4); }   // 32-B rows, 2 chunks

template <int EPI>
__global__ __launch_bounds__(256) SEGX_MIN_WAVES_PER_SIMD(2) void gemm_bf16x6_wide_kernel(GemmArgs g, PlaneGeom pg) {
    constexpr int A_BYTES = 128 * 32, B_BYTES = 256 * 32, STAGE = 3 * (A_BYTES + B_BYTES);                   // 36 KB per stage
    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * STAGE];
    const TileCoord t = tile_coord<CfgWide>(g);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
    const int K = pg.Kp;
    const int row = tid >> 1, chk = tid & 1, so = bf_lds_off16(row, chk), so2 = bf_lds_off16(row + 128, chk);
    const unsigned short* ga = pg.PA + (t.z0 * pg.pa_b0 + t.z1 * pg.pa_b1) + (int64_t)(t.m0 + row) * K + chk * 8;
    const unsigned short* gb = pg.PB + (t.z0 * pg.pb_b0 + t.z1 * pg.pb_b1) + (int64_t)(t.n0 + row) * K + chk * 8;
    const int64_t gb2 = (int64_t)128 * K;
    uint4 a0, a1, a2, b00, b01, b10, b11, b20, b21;
#define SEGX_BFW_GLOAD(k0)                                                                                                          \
    a0 = *reinterpret_cast<const uint4*>(ga + (k0)); a1 = *reinterpret_cast<const uint4*>(ga + pg.planeA + (k0));                   \
    a2 = *reinterpret_cast<const uint4*>(ga + 2 * pg.planeA + (k0));                                                                \
    b00 = *reinterpret_cast<const uint4*>(gb + (k0)); b01 = *reinterpret_cast<const uint4*>(gb + gb2 + (k0));                       \
    b10 = *reinterpret_cast<const uint4*>(gb + pg.planeB + (k0)); b11 = *reinterpret_cast<const uint4*>(gb + pg.planeB + gb2 + (k0)); \
    b20 = *reinterpret_cast<const uint4*>(gb + 2 * pg.planeB + (k0)); b21 = *reinterpret_cast<const uint4*>(gb + 2 * pg.planeB + gb2 + (k0));
#define SEGX_BFW_A(st, p) (lds + (st) * STAGE + (p) * A_BYTES)
#define SEGX_BFW_B(st, p) (lds + (st) * STAGE + 3 * A_BYTES + (p) * B_BYTES)
#define SEGX_BFW_LSTORE(st)                                                                                                         \
    *reinterpret_cast<uint4*>(SEGX_BFW_A(st, 0) + so) = a0; *reinterpret_cast<uint4*>(SEGX_BFW_A(st, 1) + so) = a1;                 \
    *reinterpret_cast<uint4*>(SEGX_BFW_A(st, 2) + so) = a2;                                                                         \
    *reinterpret_cast<uint4*>(SEGX_BFW_B(st, 0) + so) = b00; *reinterpret_cast<uint4*>(SEGX_BFW_B(st, 0) + so2) = b01;              \
    *reinterpret_cast<uint4*>(SEGX_BFW_B(st, 1) + so) = b10; *reinterpret_cast<uint4*>(SEGX_BFW_B(st, 1) + so2) = b11;              \
    *reinterpret_cast<uint4*>(SEGX_BFW_B(st, 2) + so) = b20; *reinterpret_cast<uint4*>(SEGX_BFW_B(st, 2) + so2) = b21;
    f32x16 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int kt0 = t.kbeg / 16, KT = (t.kend + 15) / 16;            // 16-k tiles of this slab (k_chunk is a multiple of 32)
    SEGX_BFW_GLOAD(kt0 * 16)
    SEGX_BFW_LSTORE(0)
    __syncthreads();
    { const int k1 = (kt0 + 1 < KT ? kt0 + 1 : kt0) * 16; SEGX_BFW_GLOAD(k1) }
    const int half = lane >> 5;
    const int fa0 = bf_lds_off16(wm * 64 + (lane & 31), half), fa1 = bf_lds_off16(wm * 64 + 32 + (lane & 31), half);
    for (int kt = kt0; kt < KT; ++kt) {
        const int cur = (kt - kt0) & 1;
        bf16x8 a[2][3];
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            a[0][p] = *reinterpret_cast<const bf16x8*>(SEGX_BFW_A(cur, p) + fa0);
            a[1][p] = *reinterpret_cast<const bf16x8*>(SEGX_BFW_A(cur, p) + fa1);
        }
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {                             // two column blocks at a time: 6 B fragments live
            bf16x8 b[2][3];
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int p = 0; p < 3; ++p)
                    b[q][p] = *reinterpret_cast<const bf16x8*>(SEGX_BFW_B(cur, p) + bf_lds_off16(wn * 128 + (2 * jj + q) * 32 + (lane & 31), half));
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    f32x16 c = acc[i][2 * jj + q];
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[q][2], c, 0, 0, 0);     // hi . lo
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][2], b[q][0], c, 0, 0, 0);     // lo . hi
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], b[q][1], c, 0, 0, 0);     // mid . mid
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[q][1], c, 0, 0, 0);     // hi . mid
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], b[q][0], c, 0, 0, 0);     // mid . hi
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[q][0], c, 0, 0, 0);     // hi . hi
                    acc[i][2 * jj + q] = c;
                }
        }
        if (cur) { SEGX_BFW_LSTORE(0) } else { SEGX_BFW_LSTORE(1) }     // tile kt + 1 into the other stage (last iteration: a harmless re-store)
        __syncthreads();
        const int kn = (kt + 2 < KT ? kt + 2 : KT - 1) * 16;
        SEGX_BFW_GLOAD(kn)
    }
#undef SEGX_BFW_GLOAD
#undef SEGX_BFW_A
#undef SEGX_BFW_B
#undef SEGX_BFW_LSTORE
    gemm_epilogue<EPI, CfgWide>(acc, g, t);
}

static int g_bf16x6_variant = 1;      // segx_tune(3, v): 1 = 128 x 128 x 32 (timed on the device), 2 = 128 x 256 x 16 wide waves (not yet timed)
int bf16x6_set_variant(int v) { if (v != 1 && v != 2) return -1; g_bf16x6_variant = v; return 0; }

// ---- implicit-GEMM 3-D convolution forward on the same tile (packed contraction order of conv3d.hip) -----------------------------------
// Y[b][co][p] = sum_k Wp[co][k] * im2col(X[b])[k][p],  k = (channel block of 8, tap, channel in block).  Eight consecutive k are the
// eight channels of one block at ONE tap, so with the activations split into bf16 planes stored channels-last-8,
// P[plane][b][cb][s][8], the B fragment of a lane (position, 8 k) is ONE 16-byte vector at (cb, window origin + tap offset): the
// im2col gather costs one address and one validity test per 8 k.  Weights: the packed fp32 filters split like any k-contiguous A.
struct ConvG { int Cin, ID, IH, IW, OD, OH, OW, KD, KH, KW, sd, sh, sw, pd, ph, pw; };

__global__ __launch_bounds__(256) void split3_cl8_kernel(const float* __restrict__ X, unsigned short* __restrict__ P, int64_t nblk, int S, int64_t plane) {
    const int64_t total = nblk * S;
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
        const int64_t blk = t / S; const int sidx = (int)(t - blk * S);
        const float* src = X + blk * 8 * S + sidx;
        union { unsigned short s[8]; uint4 q; } h, m, l;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float x = src[(int64_t)j * S];
            h.s[j] = f32_to_bf16_rne(x); const float r1 = x - bf16_to_f32(h.s[j]);
            m.s[j] = f32_to_bf16_rne(r1); const float r2 = r1 - bf16_to_f32(m.s[j]);
            l.s[j] = f32_to_bf16_rne(r2);
        }
        *reinterpret_cast<uint4*>(P + t * 8) = h.q;
        *reinterpret_cast<uint4*>(P + plane + t * 8) = m.q;
        *reinterpret_cast<uint4*>(P + 2 * plane + t * 8) = l.q;
    }
}

__global__ __launch_bounds__(256) SEGX_MIN_WAVES_PER_SIMD(2) void conv3d_fwd_bf16x6_kernel(GemmArgs g, PlaneGeom pg, ConvG q) {
    constexpr int TILE_BYTES = 128 * 64;
    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * 3 * TILE_BYTES];
    const TileCoord t = tile_coord<Cfg128>(g);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
    const int K = pg.Kp, KV = q.KD * q.KH * q.KW, KHW = q.KH * q.KW, CB = q.Cin >> 3, nchunks = CB * KV;
    const int S_in = q.ID * q.IH * q.IW, OHW = q.OH * q.OW;
    const unsigned short* PA = pg.PA + (int64_t)t.m0 * K;
    const unsigned short* PB = pg.PB + (int64_t)t.z0 * CB * S_in * 8;                // this sample's activation planes
    uint4 a00, a01, a10, a11, a20, a21, b00, b01, b10, b11, b20, b21;
    const int row0 = tid >> 2, row1 = (tid + 256) >> 2, chk = tid & 3;
    const int so0 = bf_lds_off(row0, chk), so1 = bf_lds_off(row1, chk);
    // window origins of this thread's two output positions
    const int n0_ = t.n0 + row0, n1_ = t.n0 + row1;
    const bool ok0 = n0_ < g.N, ok1 = n1_ < g.N;
    const int p0 = ok0 ? n0_ : 0, p1 = ok1 ? n1_ : 0;
    const int d0 = (p0 / OHW) * q.sd - q.pd, h0 = ((p0 / q.OW) % q.OH) * q.sh - q.ph, w0 = (p0 % q.OW) * q.sw - q.pw;
    const int d1 = (p1 / OHW) * q.sd - q.pd, h1 = ((p1 / q.OW) % q.OH) * q.sh - q.ph, w1 = (p1 % q.OW) * q.sw - q.pw;
    const uint4 zero4 = make_uint4(0u, 0u, 0u, 0u);
#define SEGX_CV_A1(p, row, RA, k0) RA = *reinterpret_cast<const uint4*>(PA + (p) * pg.planeA + (int64_t)(row) * K + (k0) + chk * 8);
#define SEGX_CV_B1(RB0, RB1, RB2, dd, hh, ww, okr)                                                                                          \
    {                                                                                                                                      \
        const int id = (dd) + kd, ih = (hh) + kh, iw = (ww) + kw;                                                                          \
        const bool v = (okr) && qv && (unsigned)id < (unsigned)q.ID && (unsigned)ih < (unsigned)q.IH && (unsigned)iw < (unsigned)q.IW;      \
        const int64_t off = v ? ((int64_t)cb * S_in + ((int64_t)id * q.IH + ih) * q.IW + iw) * 8 : 0;                                      \
        const uint4 x0 = *reinterpret_cast<const uint4*>(PB + off), x1 = *reinterpret_cast<const uint4*>(PB + pg.planeB + off),            \
                    x2 = *reinterpret_cast<const uint4*>(PB + 2 * pg.planeB + off);                                                        \
        RB0 = v ? x0 : zero4; RB1 = v ? x1 : zero4; RB2 = v ? x2 : zero4;                                                                  \
    }
#define SEGX_CV_GLOAD(k0)                                                                                                                   \
    {                                                                                                                                      \
        SEGX_CV_A1(0, row0, a00, k0) SEGX_CV_A1(0, row1, a01, k0) SEGX_CV_A1(1, row0, a10, k0) SEGX_CV_A1(1, row1, a11, k0)                 \
        SEGX_CV_A1(2, row0, a20, k0) SEGX_CV_A1(2, row1, a21, k0)                                                                          \
        const int qi = ((k0) >> 3) + chk; const bool qv = qi < nchunks;                                                                    \
        const int qc = qv ? qi : 0, cb = qc / KV, tap = qc - cb * KV, kd = tap / KHW, tr = tap - kd * KHW, kh = tr / q.KW, kw = tr - kh * q.KW; \
        SEGX_CV_B1(b00, b10, b20, d0, h0, w0, ok0)                                                                                         \
        SEGX_CV_B1(b01, b11, b21, d1, h1, w1, ok1)                                                                                         \
    }
#define SEGX_CV_LS1(p, so, RA, RB)                                                          \
    *reinterpret_cast<uint4*>(lds + (0 * 3 + (p)) * TILE_BYTES + (so)) = RA;                 \
    *reinterpret_cast<uint4*>(lds + (1 * 3 + (p)) * TILE_BYTES + (so)) = RB;
#define SEGX_CV_LSTORE()                                                                                    \
    SEGX_CV_LS1(0, so0, a00, b00) SEGX_CV_LS1(0, so1, a01, b01) SEGX_CV_LS1(1, so0, a10, b10)                \
    SEGX_CV_LS1(1, so1, a11, b11) SEGX_CV_LS1(2, so0, a20, b20) SEGX_CV_LS1(2, so1, a21, b21)
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int KT = K / 32;
    SEGX_CV_GLOAD(0)
    for (int kt = 0; kt < KT; ++kt) {
        __syncthreads();
        SEGX_CV_LSTORE()
        __syncthreads();
        const int kn = (kt + 1 < KT ? kt + 1 : kt) * 32;
        SEGX_CV_GLOAD(kn)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int chunk = 2 * s + (lane >> 5);
            bf16x8 a[2][3], b[2][3];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int p = 0; p < 3; ++p) {
                    a[i][p] = *reinterpret_cast<const bf16x8*>(lds + (0 * 3 + p) * TILE_BYTES + bf_lds_off(wm * 64 + i * 32 + (lane & 31), chunk));
                    b[i][p] = *reinterpret_cast<const bf16x8*>(lds + (1 * 3 + p) * TILE_BYTES + bf_lds_off(wn * 64 + i * 32 + (lane & 31), chunk));
                }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    f32x16 c = acc[i][j];
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][2], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][2], b[j][0], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], b[j][1], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][1], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], b[j][0], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][0], c, 0, 0, 0);
                    acc[i][j] = c;
                }
        }
    }
#undef SEGX_CV_A1
#undef SEGX_CV_B1
#undef SEGX_CV_GLOAD
#undef SEGX_CV_LS1
#undef SEGX_CV_LSTORE
    gemm_epilogue<SEGX_EPI_NONE, Cfg128>(acc, g, t);
}

static inline int round_up(int v, int m) { return (v + m - 1) / m * m; }
struct Bf16x6Plan { int RPA, RPB, Kp; int64_t nzA, nzB, planeA, planeB; int nzA1, nzB1; };
static Bf16x6Plan bf16x6_plan(const segx_gemm_desc* d) {
    Bf16x6Plan p;
    p.RPA = round_up(d->M, 128); p.RPB = round_up(d->N, 256); p.Kp = round_up(d->K, 32);      // N padded for the widest tile (128 x 256)
    const int a0 = d->a_b0 != 0 ? d->nb0 : 1, a1 = d->a_b1 != 0 ? d->nb1 : 1, b0 = d->b_b0 != 0 ? d->nb0 : 1, b1 = d->b_b1 != 0 ? d->nb1 : 1;
    p.nzA = (int64_t)a0 * a1; p.nzB = (int64_t)b0 * b1; p.nzA1 = a1; p.nzB1 = b1;
    p.planeA = p.nzA * p.RPA * p.Kp; p.planeB = p.nzB * p.RPB * p.Kp;
    return p;
}

}  // namespace segx

using namespace segx;

/* bytes of the bf16 plane workspace segx_gemm_f32_bf16x6 needs for this problem */
extern "C" int64_t segx_gemm_bf16x6_ws_bytes(const segx_gemm_desc* d) {
    if (!d || d->M <= 0 || d->N <= 0 || d->K <= 0 || d->nb0 <= 0 || d->nb1 <= 0) return 0;
    const Bf16x6Plan p = bf16x6_plan(d);
    return 2 * 3 * (p.planeA + p.planeB) + 256;
}

extern "C" int segx_gemm_f32_bf16x6(const float* A, const float* B, float* C, const segx_gemm_desc* d, void* ws, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    SEGX_REQUIRE(A && B && C && d && ws, "segx_gemm_f32_bf16x6: null argument");
    SEGX_REQUIRE(d->M > 0 && d->N > 0 && d->K > 0 && d->nb0 > 0 && d->nb1 > 0, "segx_gemm_f32_bf16x6: bad sizes");
    SEGX_REQUIRE(d->epilogue == SEGX_EPI_NONE || d->epilogue == SEGX_EPI_GELU, "segx_gemm_f32_bf16x6: bad epilogue %d", d->epilogue);
    SEGX_REQUIRE(d->epilogue != SEGX_EPI_GELU || d->aux, "segx_gemm_f32_bf16x6: GELU epilogue needs aux");
    const int splitk = d->splitk > 1 ? d->splitk : 1;
    SEGX_REQUIRE(splitk == 1 || (d->workspace && d->epilogue == SEGX_EPI_NONE && !d->gmax), "segx_gemm_f32_bf16x6: split-K needs workspace and a plain epilogue");
    SEGX_REQUIRE((reinterpret_cast<uintptr_t>(ws) & 15) == 0, "segx_gemm_f32_bf16x6: workspace must be 16-byte aligned");
    const Bf16x6Plan p = bf16x6_plan(d);
    unsigned short* PA = reinterpret_cast<unsigned short*>(ws);
    unsigned short* PB = PA + 3 * p.planeA;
    SplitArgs sa{A, PA, d->M, d->K, p.RPA, p.Kp, p.nzA1, p.nzA, d->a_b0, d->a_b1, d->a_m, d->a_k, p.planeA};
    SplitArgs sb{B, PB, d->N, d->K, p.RPB, p.Kp, p.nzB1, p.nzB, d->b_b0, d->b_b1, d->b_n, d->b_k, p.planeB};
    const int64_t ta = p.nzA * p.RPA * (p.Kp / 8), tb = p.nzB * p.RPB * (p.Kp / 8);
    const dim3 ga((unsigned)i64min(1 << 16, (ta + 255) / 256)), gb((unsigned)i64min(1 << 16, (tb + 255) / 256));
    if (d->a_m == 1 && d->a_k != 1) hipLaunchKernelGGL((split3_kernel<true>), ga, dim3(256), 0, stream, sa);
    else hipLaunchKernelGGL((split3_kernel<false>), ga, dim3(256), 0, stream, sa);
    if (d->b_n == 1 && d->b_k != 1) hipLaunchKernelGGL((split3_kernel<true>), gb, dim3(256), 0, stream, sb);
    else hipLaunchKernelGGL((split3_kernel<false>), gb, dim3(256), 0, stream, sb);
    int rc = check_launch("segx_gemm_f32_bf16x6/split");
    if (rc) return rc;

    GemmArgs g;
    g.A = A; g.B = B; g.C = C; g.bias = d->bias_mode ? d->bias : nullptr; g.aux = d->epilogue == SEGX_EPI_GELU ? d->aux : nullptr; g.gmax = d->gmax;
    g.M = d->M; g.N = d->N; g.K = d->K; g.nb1 = d->nb1;
    g.a_b0 = d->a_b0; g.a_b1 = d->a_b1; g.a_m = d->a_m; g.a_k = d->a_k;
    g.b_b0 = d->b_b0; g.b_b1 = d->b_b1; g.b_n = d->b_n; g.b_k = d->b_k;
    g.c_b0 = d->c_b0; g.c_b1 = d->c_b1; g.c_m = d->c_m; g.bias_b1 = d->bias_b1; g.bias_b0 = d->bias_b0;
    g.alpha = d->alpha; g.epilogue = d->epilogue; g.bias_mode = d->bias ? d->bias_mode : SEGX_BIAS_NONE;
    const bool wide = g_bf16x6_variant == 2;
    g.vecA = g.vecB = 1; g.tiles_m = p.RPA / 128; g.tiles_n = wide ? p.RPB / 256 : ceil_div(d->N, 128);
    g.dropout_p = d->dropout_p; g.seed = d->seed; g.offset = d->offset;
    g.splitk = splitk;
    g.k_chunk = splitk == 1 ? d->K : ceil_div(ceil_div(d->K, splitk), 32) * 32;
    g.c_split = (int64_t)d->nb0 * d->nb1 * d->M * d->N;
    if (splitk > 1) g.C = d->workspace;
    PlaneGeom pg;
    pg.PA = PA; pg.PB = PB; pg.Kp = p.Kp; pg.planeA = p.planeA; pg.planeB = p.planeB;
    const int64_t zA = (int64_t)p.RPA * p.Kp, zB = (int64_t)p.RPB * p.Kp;
    pg.pa_b1 = d->a_b1 != 0 ? zA : 0; pg.pa_b0 = d->a_b0 != 0 ? (int64_t)p.nzA1 * zA : 0;
    pg.pb_b1 = d->b_b1 != 0 ? zB : 0; pg.pb_b0 = d->b_b0 != 0 ? (int64_t)p.nzB1 * zB : 0;
    pg.zstrideA = zA; pg.zstrideB = zB;
    const dim3 grid(g.tiles_m * g.tiles_n, d->nb0 * d->nb1, splitk);
    if (wide) {
        if (d->epilogue == SEGX_EPI_GELU) hipLaunchKernelGGL((gemm_bf16x6_wide_kernel<SEGX_EPI_GELU>), grid, dim3(256), 0, stream, g, pg);
        else hipLaunchKernelGGL((gemm_bf16x6_wide_kernel<SEGX_EPI_NONE>), grid, dim3(256), 0, stream, g, pg);
    } else if (d->epilogue == SEGX_EPI_GELU) hipLaunchKernelGGL((gemm_bf16x6_kernel<SEGX_EPI_GELU>), grid, dim3(256), 0, stream, g, pg);
    else hipLaunchKernelGGL((gemm_bf16x6_kernel<SEGX_EPI_NONE>), grid, dim3(256), 0, stream, g, pg);
    rc = check_launch("segx_gemm_f32_bf16x6");
    if (rc || splitk == 1) return rc;
    const int64_t total = g.c_split;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)i64min(2048, (total + 255) / 256)), dim3(256), 0, stream, (const float*)d->workspace, C,
                       g.bias, d->M, d->N, d->nb1, splitk, g.c_split, d->c_b0, d->c_b1, d->c_m, d->alpha, g.bias_mode, d->bias_b1, d->bias_b0, total);
    return check_launch("segx_gemm_f32_bf16x6/splitk_reduce");
}

/* EXPERIMENTAL: segx_conv3d_fwd_packed semantics (packed filters Wp, stride/pad geometry as segx_conv3d_fwd) on the bf16x6 tile.
 * ws: segx_conv3d_bf16x6_ws_bytes(B, Cout, geom) bytes, 16-byte aligned. */
extern "C" int64_t segx_conv3d_bf16x6_ws_bytes(int B, int Cout, const int* geom) {
    if (!geom || B <= 0 || Cout <= 0) return 0;
    const int64_t K = (int64_t)geom[0] * geom[7] * geom[8] * geom[9], S_in = (int64_t)geom[1] * geom[2] * geom[3];
    return 2 * 3 * ((int64_t)round_up(Cout, 128) * round_up((int)K, 32) + (int64_t)B * geom[0] * S_in) + 256;
}
extern "C" int segx_conv3d_fwd_bf16x6(const float* X, const float* Wp, float* Y, int B, int Cout, const int* geom, void* ws, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    SEGX_REQUIRE(X && Wp && Y && geom && ws && B > 0 && Cout > 0 && B <= 65535, "segx_conv3d_fwd_bf16x6: bad args");
    ConvG q{geom[0], geom[1], geom[2], geom[3], geom[4], geom[5], geom[6], geom[7], geom[8], geom[9], geom[10], geom[11], geom[12], geom[13], geom[14], geom[15]};
    SEGX_REQUIRE(q.Cin > 0 && q.Cin % 8 == 0, "segx_conv3d_fwd_bf16x6: Cin = %d is not a multiple of 8", q.Cin);
    const int64_t P = (int64_t)q.OD * q.OH * q.OW, S_in = (int64_t)q.ID * q.IH * q.IW; const int K = q.Cin * q.KD * q.KH * q.KW;
    SEGX_REQUIRE(P > 0 && P < 2147483647LL && K > 0 && (int64_t)q.Cin * S_in < 2147483647LL, "segx_conv3d_fwd_bf16x6: bad geometry");
    SEGX_REQUIRE((reinterpret_cast<uintptr_t>(ws) & 15) == 0, "segx_conv3d_fwd_bf16x6: workspace must be 16-byte aligned");
    const int RPA = round_up(Cout, 128), Kp = round_up(K, 32);
    const int64_t planeA = (int64_t)RPA * Kp, planeB = (int64_t)B * q.Cin * S_in;
    unsigned short* PA = reinterpret_cast<unsigned short*>(ws);
    unsigned short* PB = PA + 3 * planeA;
    SplitArgs sa{Wp, PA, Cout, K, RPA, Kp, 1, 1, 0, 0, K, 1, planeA};
    const int64_t ta = (int64_t)RPA * (Kp / 8), tb = (int64_t)B * (q.Cin / 8) * S_in;
    hipLaunchKernelGGL((split3_kernel<false>), dim3((unsigned)i64min(1 << 16, (ta + 255) / 256)), dim3(256), 0, stream, sa);
    hipLaunchKernelGGL(split3_cl8_kernel, dim3((unsigned)i64min(1 << 16, (tb + 255) / 256)), dim3(256), 0, stream, X, PB, (int64_t)B * (q.Cin / 8), (int)S_in, planeB);
    int rc = check_launch("segx_conv3d_fwd_bf16x6/split");
    if (rc) return rc;
    GemmArgs g;
    g.A = Wp; g.B = X; g.C = Y; g.bias = nullptr; g.aux = nullptr; g.gmax = nullptr;
    g.M = Cout; g.N = (int)P; g.K = K; g.nb1 = 1;
    g.a_b0 = g.a_b1 = 0; g.a_m = K; g.a_k = 1; g.b_b0 = g.b_b1 = g.b_n = g.b_k = 0;
    g.c_b0 = (int64_t)Cout * P; g.c_b1 = 0; g.c_m = P; g.bias_b1 = g.bias_b0 = 0;
    g.alpha = 1.0f; g.epilogue = SEGX_EPI_NONE; g.bias_mode = SEGX_BIAS_NONE; g.vecA = g.vecB = 1;
    g.tiles_m = RPA / 128; g.tiles_n = ceil_div(P, 128);
    g.dropout_p = 0.f; g.seed = g.offset = 0; g.k_chunk = K; g.splitk = 1; g.c_split = 0;
    PlaneGeom pg;
    pg.PA = PA; pg.PB = PB; pg.Kp = Kp; pg.planeA = planeA; pg.planeB = planeB;
    pg.pa_b0 = pg.pa_b1 = pg.pb_b0 = pg.pb_b1 = 0; pg.zstrideA = pg.zstrideB = 0;
    hipLaunchKernelGGL(conv3d_fwd_bf16x6_kernel, dim3(g.tiles_m * g.tiles_n, B, 1), dim3(256), 0, stream, g, pg, q);
    return check_launch("segx_conv3d_fwd_bf16x6");
}
