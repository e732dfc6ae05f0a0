// gemm_h3.h -- "f16x3": the wave-specialised persistent GEMM of gemm_x6ws.h with a TWO-plane fp16 image and THREE matrix instructions per block
// product instead of three bf16 planes and six.
//
// Arithmetic.  Every operand row (the index that is NOT contracted) gets a power-of-two scale s (h3_rowmax / h3_scales kernels: the row's largest
// magnitude lands in [2^14, 2^15), inside fp16's range).  The scaled value splits into two fp16 numbers x s = h + l: h = fp16(x s) (11 significand bits,
// round to nearest even), l = fp16(x s - h) (the next 11 bits; exact subtraction).  |x s - h - l| <= 2^-24 |x s| as long as l is a normal fp16, i.e.
// for every element within 2^-17 of its row's maximum; smaller elements keep an ABSOLUTE error of 2^-25 in scaled units = 2^-40 of the row maximum.
// A block product is v_mfma_f32_32x32x16_f16 on (h.l, l.h, h.h), small terms first; every fp16 x fp16 product is exact in fp32 (22 bits), the
// accumulation is fp32, the dropped l.l term is 2^-24 relative.  The epilogue multiplies by the inverse scales (exact).  So the result carries the
// same error bound as the bf16x6 scheme, 2^-23 sum |a||b|, plus a block-floating-point term 2^-39 (max_row|a| sum|b| + max_row|b| sum|a|) that only
// matters when an operand row spans more than 2^15 in magnitude AND the other operand is large exactly where it is small.
// Why: the bf16x6 kernels are bound by the clock the chip sustains under dense matrix-pipe load (the same kernel runs 223 TFLOP/s
// on random operands and 298 on zeros); half the matrix instructions per product is the one lever left.  Roof: 2.5 PFLOP/s / 3 = 833 TFLOP/s.
//
// LDS image per operand: two planes [plane][row][32 k] of fp16, the row / chunk swizzle of gemm_x6.h.  Loaders: the thread -> element maps of
// WsDense6 plus the row scales of the thread's pieces (see H3Dense).
#pragma once
#include "gemm_x6ws.h"

namespace segx {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16v2 __attribute__((ext_vector_type(2)));

// (x0, x1), already scaled -> two packed fp16 pairs (element 0 in the low half)
struct SplitH { unsigned h, l; };
__device__ __forceinline__ SplitH h3_split_pair(float x0, float x1) {
    const f32v2 v0 = {x0, x1};
    const f16v2 hv = __builtin_convertvector(v0, f16v2);
    float a0 = x0 - (float)hv.x, a1 = x1 - (float)hv.y;
    SEGX_PIN(a0);
    const f32v2 v1 = {a0, a1};
    SplitH o; o.h = __builtin_bit_cast(unsigned, hv); o.l = __builtin_bit_cast(unsigned, __builtin_convertvector(v1, f16v2));
    return o;
}
template <int PLANE_BYTES>
__device__ __forceinline__ void h3_store8(unsigned char* __restrict__ P, int off, const float (&v)[8]) {
    const SplitH a = h3_split_pair(v[0], v[1]), b = h3_split_pair(v[2], v[3]), c = h3_split_pair(v[4], v[5]), d = h3_split_pair(v[6], v[7]);
    *reinterpret_cast<uint4*>(P + off) = make_uint4(a.h, b.h, c.h, d.h);
    *reinterpret_cast<uint4*>(P + PLANE_BYTES + off) = make_uint4(a.l, b.l, c.l, d.l);
}
template <int PLANE_BYTES>
__device__ __forceinline__ void h3_store4(unsigned char* __restrict__ P, int off, float v0, float v1, float v2, float v3) {
    const SplitH a = h3_split_pair(v0, v1), b = h3_split_pair(v2, v3);
    uint2 h, l; h.x = a.h; h.y = b.h; l.x = a.l; l.y = b.l;
    *reinterpret_cast<uint2*>(P + off) = h;
    *reinterpret_cast<uint2*>(P + PLANE_BYTES + off) = l;
}

template <class Cfg> struct H3Lds { static constexpr int A_BYTES = 2 * X6Plane<Cfg::BM>::bytes, B_BYTES = 2 * X6Plane<Cfg::BN>::bytes, BYTES = A_BYTES + B_BYTES; };

// one 32-k stage of a consumer wave: three products per accumulator, small terms first; consecutive matrix instructions go to different accumulators
template <class Cfg>
__device__ __forceinline__ void h3_stage_mfma(f32x16 (&acc)[Cfg::MI][Cfg::NJ], const unsigned char* __restrict__ LA_, const unsigned char* __restrict__ LB_,
                                              int arow, int brow, int kh) {
    constexpr int MI = Cfg::MI, NJ = Cfg::NJ, PA = X6Plane<Cfg::BM>::bytes, PB = X6Plane<Cfg::BN>::bytes;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const int chunk = 2 * s + kh;
        f16x8 b[NJ][2];
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int p = 0; p < 2; ++p) b[j][p] = *reinterpret_cast<const f16x8*>(LB_ + p * PB + x6_off(brow + 32 * j, chunk));
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            f16x8 a[2];
#pragma unroll
            for (int p = 0; p < 2; ++p) a[p] = *reinterpret_cast<const f16x8*>(LA_ + p * PA + x6_off(arow + 32 * i, chunk));
#define SEGX_H3_P(PA_, PB_)                                                                                                \
    _Pragma("unroll") for (int j = 0; j < NJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[PA_], b[j][PB_], acc[i][j], 0, 0, 0);
            SEGX_H3_P(0, 1) SEGX_H3_P(1, 0) SEGX_H3_P(0, 0)
#undef SEGX_H3_P
        }
    }
}

struct H3WsEngine {
    static constexpr bool SCALED = true;
    static constexpr int NSETS = 3;                          // three register sets, loads interleaved with the split (x6ws_body)
    template <class Cfg> struct Lds { static constexpr int A_BYTES = H3Lds<Cfg>::A_BYTES, STAGE = H3Lds<Cfg>::BYTES; };
    template <class Cfg> static __device__ __forceinline__ void mfma(f32x16 (&acc)[Cfg::MI][Cfg::NJ], const unsigned char* __restrict__ LA_,
                                                                     const unsigned char* __restrict__ LB_, int arow, int brow, int kh) {
        h3_stage_mfma<Cfg>(acc, LA_, LB_, arow, brow, kh);
    }
    // one producer stage: loads of (ax, bx) at k0 dealt out between the split-and-store groups of (ay, by) -> LDS images PA / PB
    template <class LA, class LB>
    static __device__ __forceinline__ void stage(const LA& la, const LB& lb, float (&ax)[LA::NREG], float (&bx)[LB::NREG], float (&ay)[LA::NREG],
                                                 float (&by)[LB::NREG], int k0, unsigned char* __restrict__ PA, unsigned char* __restrict__ PB, int ptid);
};

// ---- loaders: WsDense6's maps + row scales in the register set ------------------------------------------------------------------------------
template <bool KC, int ROWS> struct H3Dense;

// Row scales travel in the REGISTER SET (store6 runs on the stream's loader, which may already describe a later work item, so it must not read
// loader state).  One scale per piece and stage would double the load instructions (r03_ae: the producers became the bottleneck), so the scale array
// is stored PERMUTED -- index(row) = (row % 32) * R32 + row / 32, R32 = 8 * ceil(rows / 256) -- which puts the scales of a thread's pieces (rows
// 32 apart) side by side: two 16-byte loads for a 256-row tile, one for 128 rows.  Rows past the edge hold scale 0 (h3_scales_kernel), which zeroes
// the clamped duplicates without a mask.
__host__ __device__ inline int h3_r32(int rows) { return 8 * ((rows + 255) / 256); }

template <int ROWS>
struct H3Dense<true, ROWS> {                                  // k-contiguous: piece f = ptid + 256 i -> row f / 8, four consecutive k at 4 (f % 8)
    static constexpr int NPT = ROWS * BKT / 1024, NREG = 5 * NPT;
    const float* base; const float* scale;
    unsigned off[NPT], soff;
    __device__ __forceinline__ void begin(const float* b, int64_t s_row, int64_t, int row0, int rows, const float* scale_, int ptid) {
        base = b; scale = scale_;
        soff = (unsigned)(((ptid >> 3) * h3_r32(rows) + (row0 >> 5)) << 2);
#pragma unroll
        for (int i = 0; i < NPT; ++i) {
            const int row = row0 + (ptid >> 3) + 32 * i;
            off[i] = (unsigned)(((int64_t)(row < rows ? row : rows - 1) * s_row + ((ptid & 7) << 2)) << 2);
        }
    }
    __device__ __forceinline__ unsigned load6(float (&r)[NREG], int k0, int, int) const {
#ifdef SEGX_PROBE_SAMEK                                      // bench-only probe (results are NOT the GEMM): every stage re-reads k-tile 0 or 1 (cache-hot operands)
        k0 &= 32;
#endif
        const ws_gptr b = ws_uniform_base(base + k0), sb = ws_uniform_base(scale);
#pragma unroll
        for (int i = 0; i < NPT; ++i) {
            const f32x4 v = ws_load<f32x4>(b, off[i]);
            r[4 * i] = v.x; r[4 * i + 1] = v.y; r[4 * i + 2] = v.z; r[4 * i + 3] = v.w;
        }
#pragma unroll
        for (int i = 0; i < NPT / 4; ++i) {
            const f32x4 v = ws_load<f32x4>(sb, soff + 16 * i);
            r[4 * NPT + 4 * i] = v.x; r[4 * NPT + 4 * i + 1] = v.y; r[4 * NPT + 4 * i + 2] = v.z; r[4 * NPT + 4 * i + 3] = v.w;
        }
        return 0u;
    }
    __device__ __forceinline__ void store6(float (&r)[NREG], unsigned, unsigned char* __restrict__ P, int ptid) const {
#pragma unroll
        for (int i = 0; i < NPT; ++i) store_step(r, i, P, ptid);
    }
    // step form (h3_interleave): NL load instructions, NS split-and-store groups
    static constexpr int NL = NPT + NPT / 4, NS = NPT;
    __device__ __forceinline__ void bases(int k0, ws_gptr& b, ws_gptr& sb) const { b = ws_uniform_base(base + k0); sb = ws_uniform_base(scale); }
    __device__ __forceinline__ void load_step(float (&r)[NREG], int i, ws_gptr b, ws_gptr sb) const {
        const f32x4 v = i < NPT ? ws_load<f32x4>(b, off[i < NPT ? i : 0]) : ws_load<f32x4>(sb, soff + 16 * (i - NPT));
        r[4 * i] = v.x; r[4 * i + 1] = v.y; r[4 * i + 2] = v.z; r[4 * i + 3] = v.w;       // data pieces 0 .. NPT-1, then the scale quads: the same register order
    }
    __device__ __forceinline__ void store_step(float (&r)[NREG], int i, unsigned char* __restrict__ P, int ptid) const {
        const int f = ptid + 256 * i, row = f >> 3, kc = f & 7;
        const float s_ = r[4 * NPT + i];
        h3_store4<X6Plane<ROWS>::bytes>(P, x6_off(row, kc >> 1) + ((kc & 1) << 3), r[4 * i] * s_, r[4 * i + 1] * s_, r[4 * i + 2] * s_, r[4 * i + 3] * s_);
    }
};

template <int ROWS>
struct H3Dense<false, ROWS> {                                 // row-contiguous: rows 2 rp, 2 rp + 1 (one 8-byte load per k), KQ = ROWS / 16 consecutive k
    static_assert(ROWS == 256 || ROWS == 128, "row-contiguous f16x3 loader: 128 or 256 rows");
    static constexpr int KQ = ROWS / 16, RP = ROWS / 2, NREG = 2 * KQ + 2;
    const float* base; const float* scale; int64_t s_k;
    unsigned off[KQ], soff0, soff1;
    __device__ __forceinline__ void begin(const float* b, int64_t, int64_t s_k_, int row0, int rows, const float* scale_, int ptid) {
        base = b; scale = scale_; s_k = s_k_;
        const int row = row0 + 2 * (ptid % RP);
        const bool rok = row < rows;                          // rows % 4 == 0 and row even: the pair is inside or outside together
        soff0 = (unsigned)(((row & 31) * h3_r32(rows) + (row >> 5)) << 2);
        soff1 = (unsigned)((((row + 1) & 31) * h3_r32(rows) + ((row + 1) >> 5)) << 2);
#pragma unroll
        for (int j = 0; j < KQ; ++j) off[j] = (unsigned)(((int64_t)(KQ * (ptid / RP) + j) * s_k_ + (rok ? row : rows - 2)) << 2);
    }
    __device__ __forceinline__ unsigned load6(float (&r)[NREG], int k0, int, int) const {
#ifdef SEGX_PROBE_SAMEK
        k0 &= 32;
#endif
        const ws_gptr bk = ws_uniform_base(base + (int64_t)k0 * s_k), sb = ws_uniform_base(scale);
#pragma unroll
        for (int j = 0; j < KQ; ++j) {
            const f32v2 v = ws_load<f32v2>(bk, off[j]);
            r[2 * j] = v.x; r[2 * j + 1] = v.y;
        }
        r[2 * KQ] = ws_load<float>(sb, soff0); r[2 * KQ + 1] = ws_load<float>(sb, soff1);
        return 0u;
    }
    __device__ __forceinline__ void store6(float (&r)[NREG], unsigned, unsigned char* __restrict__ P, int ptid) const {
#pragma unroll
        for (int i = 0; i < NS; ++i) store_step(r, i, P, ptid);
    }
    static constexpr int NL = KQ + 2, NS = KQ / 4;             // loads: KQ row pairs + 2 scales; stores: (row e, octet h) groups, e = i / (KQ / 8)
    __device__ __forceinline__ void bases(int k0, ws_gptr& b, ws_gptr& sb) const { b = ws_uniform_base(base + (int64_t)k0 * s_k); sb = ws_uniform_base(scale); }
    __device__ __forceinline__ void load_step(float (&r)[NREG], int i, ws_gptr b, ws_gptr sb) const {
        if (i < KQ) { const f32v2 v = ws_load<f32v2>(b, off[i < KQ ? i : 0]); r[2 * i] = v.x; r[2 * i + 1] = v.y; }
        else r[2 * KQ + (i - KQ)] = ws_load<float>(sb, i == KQ ? soff0 : soff1);
    }
    __device__ __forceinline__ void store_step(float (&r)[NREG], int i, unsigned char* __restrict__ P, int ptid) const {
        const int row = 2 * (ptid % RP), kg = ptid / RP, e = i / (KQ / 8), h = i % (KQ / 8);
        const float s_ = r[2 * KQ + e];
        const float v[8] = {r[16 * h + e] * s_, r[16 * h + 2 + e] * s_, r[16 * h + 4 + e] * s_, r[16 * h + 6 + e] * s_,
                            r[16 * h + 8 + e] * s_, r[16 * h + 10 + e] * s_, r[16 * h + 12 + e] * s_, r[16 * h + 14 + e] * s_};
        h3_store8<X6Plane<ROWS>::bytes>(P, x6_off(row + e, (KQ / 8) * kg + h), v);
    }
};

// EXPERIMENT (-DSEGX_H3_INTERLEAVE; r03_ag): one producer stage of one operand with the load instructions of register set `rx` (a later stage) dealt out
// between the split-and-store groups of set `ry`.  Cycle stamps (tools/ws_timing.py) had shown a producer stage = 1060 cycles issuing 15 loads + 1884
// splitting; interleaved it is 2850: the ~70 cycles a wave spends per 1-KB load instruction are ISSUE time, not queueing, and do not overlap with the same
// wave's arithmetic.  Kept for the record; the product uses load6 + store6.
template <class LD>
__device__ __forceinline__ void h3_interleave(const LD& ld, float (&rx)[LD::NREG], float (&ry)[LD::NREG], int k0, unsigned char* __restrict__ P, int ptid) {
    ws_gptr b, sb;
    ld.bases(k0, b, sb);
    constexpr int LPS = (LD::NL + LD::NS - 1) / LD::NS;
#pragma unroll
    for (int st = 0; st < LD::NS; ++st) {
#pragma unroll
        for (int l = 0; l < LPS; ++l)
            if (st * LPS + l < LD::NL) ld.load_step(rx, st * LPS + l, b, sb);
        ld.store_step(ry, st, P, ptid);
        // keeps the next group's loads behind this group's LDS stores.  NOT __builtin_amdgcn_sched_barrier(0): with it this pattern (loads into one register
        // set dealt out between the consumers of another) produced wrong results on the device for the 256-row k-contiguous loader (r03_ag: correct
        // without the builtin, correct with all loads first, wrong with both) -- a compiler-only memory fence gives the order without it
        SEGX_LOAD_FENCE();
    }
}

template <class LA, class LB>
__device__ __forceinline__ void H3WsEngine::stage(const LA& la, const LB& lb, float (&ax)[LA::NREG], float (&bx)[LB::NREG], float (&ay)[LA::NREG],
                                                  float (&by)[LB::NREG], int k0, unsigned char* __restrict__ PA, unsigned char* __restrict__ PB, int ptid) {
#if defined(SEGX_H3_INTERLEAVE)                               // experiment (r03_ag): not faster -- see h3_interleave
    h3_interleave(la, ax, ay, k0, PA, ptid);
    h3_interleave(lb, bx, by, k0, PB, ptid);
#else
    la.load6(ax, k0, 0, ptid); lb.load6(bx, k0, 0, ptid);
    la.store6(ay, 0u, PA, ptid); lb.store6(by, 0u, PB, ptid);
#endif
}

template <class Cfg, bool AKC, bool BKC>
struct H3Mk {
    using LA = H3Dense<AKC, Cfg::BM>; using LB = H3Dense<BKC, Cfg::BN>;
    __device__ __forceinline__ void make(const GemmArgs& g, const TileCoord& t, LA& la, LB& lb, int ptid) const {
        la.begin(g.A + t.z0 * g.a_b0 + t.z1 * g.a_b1, g.a_m, g.a_k, t.m0, g.M, g.sa + (int64_t)t.zb * 32 * h3_r32(g.M), ptid);
        lb.begin(g.B + t.z0 * g.b_b0 + t.z1 * g.b_b1, g.b_n, g.b_k, t.n0, g.N, g.sb + (int64_t)t.zb * 32 * h3_r32(g.N), ptid);
    }
};

// =================================================================================================================================================
// The 16-wave form (r03 cycle stamps: the 8-wave f16x3 kernel is bound by its four producer waves, whose ~70-cycle load issues and split arithmetic are
// serial within a wave): the same 256 x 128 tile over EIGHT consumer waves (64 x 64 each: 64 accumulator registers) and EIGHT producer waves (half the
// loads and half the split each), 1024 threads = 4 waves per SIMD = 128 VGPRs per wave; three producer register sets; two 48-KB LDS stages.
// Selected with segx_tune knob 10 = 16.  Measured once at the end of round 3: correct, and NOT faster than the 8-wave kernel (2.58 vs 2.54 ms on
// 24576 x 1792 x 1792 x 4) -- the limit is the CU's vector-load rate (48 KB of fp32 operands per stage at ~16 B/clk), not the number of waves issuing loads
// (DESIGN.md 5c-r3).  Kept, off by default, as the measured counter-example; the default stays the 8-wave kernel.
// =================================================================================================================================================
// TileCfg (gemm_core.h) is pinned to four waves per workgroup; the 16-wave kernel's tile: 8 consumer waves 4 x 2, each 2 x 2 blocks of 32 x 32
struct Cfg16w {
    static constexpr int WM = 4, WN = 2, MI = 2, NJ = 2;
    static constexpr int BM = WM * MI * 32, BN = WN * NJ * 32;
};
__host__ __device__ inline int h3_rg(int rows, int G) { return ((rows + 255) / 256) * (256 / G); }     // granules per permuted row of the scale array

template <bool KC, int ROWS, int PT> struct H3DenseP;        // PT producer threads; scale permutation granule G = PT / 8 rows

template <int ROWS, int PT>
struct H3DenseP<true, ROWS, PT> {                            // k-contiguous: piece f = ptid + PT i -> row f / 8, four consecutive k at 4 (f % 8)
    static constexpr int NPT = ROWS * BKT / (4 * PT), NREG = 5 * NPT, G = PT / 8;
    static_assert(NPT == 4 || NPT == 2, "H3DenseP<true>: 4 or 2 pieces per thread");
    const float* base; const float* scale;
    unsigned off[NPT], soff;
    __device__ __forceinline__ void begin(const float* b, int64_t s_row, int64_t, int row0, int rows, const float* scale_, int ptid) {
        base = b; scale = scale_;
        soff = (unsigned)(((ptid >> 3) * h3_rg(rows, G) + row0 / G) << 2);
#pragma unroll
        for (int i = 0; i < NPT; ++i) {
            const int row = row0 + (ptid >> 3) + G * i;
            off[i] = (unsigned)(((int64_t)(row < rows ? row : rows - 1) * s_row + ((ptid & 7) << 2)) << 2);
        }
    }
    __device__ __forceinline__ unsigned load6(float (&r)[NREG], int k0, int, int) const {
        const ws_gptr b = ws_uniform_base(base + k0), sb = ws_uniform_base(scale);
#pragma unroll
        for (int i = 0; i < NPT; ++i) {
            const f32x4 v = ws_load<f32x4>(b, off[i]);
            r[4 * i] = v.x; r[4 * i + 1] = v.y; r[4 * i + 2] = v.z; r[4 * i + 3] = v.w;
        }
        if (NPT == 4) { const f32x4 v = ws_load<f32x4>(sb, soff); r[16] = v.x; r[17] = v.y; r[18] = v.z; r[19] = v.w; }
        else { const f32v2 v = ws_load<f32v2>(sb, soff); r[4 * NPT] = v.x; r[4 * NPT + 1] = v.y; }
        return 0u;
    }
    __device__ __forceinline__ void store6(float (&r)[NREG], unsigned, unsigned char* __restrict__ P, int ptid) const {
#pragma unroll
        for (int i = 0; i < NPT; ++i) {
            const int f = ptid + PT * i, row = f >> 3, kc = f & 7;
            const float s_ = r[4 * NPT + i];
            h3_store4<X6Plane<ROWS>::bytes>(P, x6_off(row, kc >> 1) + ((kc & 1) << 3), r[4 * i] * s_, r[4 * i + 1] * s_, r[4 * i + 2] * s_, r[4 * i + 3] * s_);
        }
    }
};

template <int ROWS, int PT>
struct H3DenseP<false, ROWS, PT> {                           // row-contiguous: rows 2 rp, 2 rp + 1, KQ consecutive k from KQ (ptid / RP)
    static constexpr int RP = ROWS / 2, KPP = PT / RP, KQ = BKT / KPP, NREG = 2 * KQ + 2, G = PT / 8;
    static_assert(KQ == 8 || KQ == 4, "H3DenseP<false>: 8 or 4 consecutive k per thread");
    const float* base; const float* scale; int64_t s_k;
    unsigned off[KQ], soff0, soff1;
    __device__ __forceinline__ void begin(const float* b, int64_t, int64_t s_k_, int row0, int rows, const float* scale_, int ptid) {
        base = b; scale = scale_; s_k = s_k_;
        const int row = row0 + 2 * (ptid % RP);
        const bool rok = row < rows;                          // rows % 4 == 0 and row even: the pair is inside or outside together
        const int rg = h3_rg(rows, G);
        soff0 = (unsigned)(((row % G) * rg + row / G) << 2);
        soff1 = (unsigned)((((row + 1) % G) * rg + (row + 1) / G) << 2);
#pragma unroll
        for (int j = 0; j < KQ; ++j) off[j] = (unsigned)(((int64_t)(KQ * (ptid / RP) + j) * s_k_ + (rok ? row : rows - 2)) << 2);
    }
    __device__ __forceinline__ unsigned load6(float (&r)[NREG], int k0, int, int) const {
        const ws_gptr bk = ws_uniform_base(base + (int64_t)k0 * s_k), sb = ws_uniform_base(scale);
#pragma unroll
        for (int j = 0; j < KQ; ++j) {
            const f32v2 v = ws_load<f32v2>(bk, off[j]);
            r[2 * j] = v.x; r[2 * j + 1] = v.y;
        }
        r[2 * KQ] = ws_load<float>(sb, soff0); r[2 * KQ + 1] = ws_load<float>(sb, soff1);
        return 0u;
    }
    __device__ __forceinline__ void store6(float (&r)[NREG], unsigned, unsigned char* __restrict__ P, int ptid) const {
        const int row = 2 * (ptid % RP), kg = ptid / RP;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const float s_ = r[2 * KQ + e];
            if (KQ == 8) {
                const float v[8] = {r[e] * s_, r[2 + e] * s_, r[4 + e] * s_, r[6 + e] * s_, r[8 + e] * s_, r[10 + e] * s_, r[12 + e] * s_, r[14 + e] * s_};
                h3_store8<X6Plane<ROWS>::bytes>(P, x6_off(row + e, kg), v);
            } else {
                h3_store4<X6Plane<ROWS>::bytes>(P, x6_off(row + e, kg >> 1) + ((kg & 1) << 3), r[e] * s_, r[2 + e] * s_, r[4 + e] * s_, r[6 + e] * s_);
            }
        }
    }
};

template <class Cfg, bool AKC, bool BKC>
struct H3Mk16 {
    using LA = H3DenseP<AKC, Cfg::BM, 512>; using LB = H3DenseP<BKC, Cfg::BN, 512>;
    __device__ __forceinline__ void make(const GemmArgs& g, const TileCoord& t, LA& la, LB& lb, int ptid) const {
        la.begin(g.A + t.z0 * g.a_b0 + t.z1 * g.a_b1, g.a_m, g.a_k, t.m0, g.M, g.sa + (int64_t)t.zb * 64 * h3_rg(g.M, 64), ptid);
        lb.begin(g.B + t.z0 * g.b_b0 + t.z1 * g.b_b1, g.b_n, g.b_k, t.n0, g.N, g.sb + (int64_t)t.zb * 64 * h3_rg(g.N, 64), ptid);
    }
};

// the persistent body of x6ws_body for 8 consumer + 8 producer waves (Cfg = Cfg16w: 256 x 128, a consumer wave owns 64 x 64)
template <class Cfg, class MK>
__device__ __forceinline__ void h3ws16_body(const GemmArgs& g, const MK& mk, unsigned char* __restrict__ lds) {
    using LA = typename MK::LA; using LB = typename MK::LB;
    static_assert(Cfg::WM * Cfg::WN == 8, "h3ws16_body: eight consumer waves");
    constexpr int STAGE = H3Lds<Cfg>::BYTES, A_BYTES = H3Lds<Cfg>::A_BYTES;
    const int wave = SEGX_WAVE_UNIFORM((int)(threadIdx.x >> 6));
    const int G = gridDim.x, pos = ws_round_pos(blockIdx.x, G);
    const int total = g.tiles_m * g.tiles_n * g.nbatch * g.splitk;
    if (wave >= 8) {
        // producers: the three-set schedule of x6ws_body (loads of stage s+3 issued before stage s+1 is split)
        const int ptid = threadIdx.x - 512;
        X6WsStream<Cfg, MK> st; st.r = 0; st.ptid = ptid; st.open(g, mk, pos, G, total);
        if (!st.valid) return;
        float p0[LA::NREG], q0[LB::NREG], p1[LA::NREG], q1[LB::NREG], p2[LA::NREG], q2[LB::NREG];
        bool v0, v1, v2;
        st.la.load6(p0, st.k, st.kend, ptid); st.lb.load6(q0, st.k, st.kend, ptid); st.next(g, mk, pos, G, total);
        v1 = st.valid; st.la.load6(p1, st.k, st.kend, ptid); st.lb.load6(q1, st.k, st.kend, ptid); st.next(g, mk, pos, G, total);
        v2 = st.valid; st.la.load6(p2, st.k, st.kend, ptid); st.lb.load6(q2, st.k, st.kend, ptid); st.next(g, mk, pos, G, total);
        st.la.store6(p0, 0u, lds, ptid); st.lb.store6(q0, 0u, lds + A_BYTES, ptid);
        int par = 0;
        for (;;) {
            SEGX_LDS_BARRIER(); par ^= 1;
            if (!v1) break;
            v0 = st.valid; st.la.load6(p0, st.k, st.kend, ptid); st.lb.load6(q0, st.k, st.kend, ptid); st.next(g, mk, pos, G, total);
            st.la.store6(p1, 0u, lds + par * STAGE, ptid); st.lb.store6(q1, 0u, lds + par * STAGE + A_BYTES, ptid);
            SEGX_LDS_BARRIER(); par ^= 1;
            if (!v2) break;
            v1 = st.valid; st.la.load6(p1, st.k, st.kend, ptid); st.lb.load6(q1, st.k, st.kend, ptid); st.next(g, mk, pos, G, total);
            st.la.store6(p2, 0u, lds + par * STAGE, ptid); st.lb.store6(q2, 0u, lds + par * STAGE + A_BYTES, ptid);
            SEGX_LDS_BARRIER(); par ^= 1;
            if (!v0) break;
            v2 = st.valid; st.la.load6(p2, st.k, st.kend, ptid); st.lb.load6(q2, st.k, st.kend, ptid); st.next(g, mk, pos, G, total);
            st.la.store6(p0, 0u, lds + par * STAGE, ptid); st.lb.store6(q0, 0u, lds + par * STAGE + A_BYTES, ptid);
        }
        return;
    }
    constexpr int MI = Cfg::MI, NJ = Cfg::NJ;
    const int lane = threadIdx.x & 63, wm = wave / Cfg::WN, wn = wave % Cfg::WN;
    const int arow = wm * (32 * MI) + (lane & 31), brow = wn * (32 * NJ) + (lane & 31), kh = lane >> 5;
    int par = 0;
    for (int r = 0;; ++r) {
        const int item = r * G + pos;
        if (item >= total) break;
        const TileCoord t = ws_item_coord<Cfg>(g, item);
        f32x16 acc[MI][NJ];
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;
        for (int kt = t.kbeg; kt < t.kend; kt += BKT) {
            SEGX_LDS_BARRIER();
            const unsigned char* const P = lds + par * STAGE;
            h3_stage_mfma<Cfg>(acc, P, P + A_BYTES, arow, brow, kh);
            par ^= 1;
        }
        acc_unscale<Cfg>(acc, g, t);
        gemm_epilogue<SEGX_EPI_NONE, Cfg, false>(acc, g, t);
    }
}

}  // namespace segx
