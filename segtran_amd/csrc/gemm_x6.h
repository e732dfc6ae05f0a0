// gemm_x6.h -- the SECOND tile engine of libsegx: fp32 GEMM / implicit-GEMM convolution evaluated on the bf16 matrix core with
// fp32-equivalent accuracy ("bf16x6").  It shares tile coordinates, loaders' global-memory side, split-K and the epilogue with the
// fp32-MFMA engine of gemm_core.h; only the LDS image and the inner product differ.
//
// Arithmetic.  Every fp32 operand value is split, in registers, into three bf16 numbers x = hi + mid + lo (round-to-nearest-even at
// every step: v_cvt_pk_bf16_f32; hi + mid + lo reproduces all 24 significand bits).  A 32 x 32 x 16 block product is six
// v_mfma_f32_32x32x16_bf16 on the pairs (hi.lo, lo.hi, mid.mid, hi.mid, mid.hi, hi.hi), small terms first; every bf16 x bf16 product
// is exact in fp32 and the accumulation is fp32, the three dropped terms (mid.lo, lo.mid, lo.lo) are <= 2^-24 relative.  Measured on
// the device against fp64 (K = 1792): max error / max |C| = 1.2e-6, the fp32 MFMA itself 1.0e-6 (profiles/r01_l_bf16x6_proto.txt).
// The instruction issues at 16x the rate of v_mfma_f32_32x32x2_f32, i.e. six of them do the work of sixteen fp32 MFMAs in the time
// of six: the roof of this engine is 2.5 PFLOP/s / 6 = 417 TFLOP/s of fp32-equivalent work against 157.3 for the fp32 engine.
//
// Why the split happens IN the kernel (round 1 split the operands in a separate pass into bf16 planes in HBM): the pass costs
// 10 B/element of HBM traffic, which is more than the whole GEMM for the 24..272-channel pointwise convolutions of the backbones, needs a
// transposing (uncoalesced) variant for row-contiguous operands and a workspace of 1.5x the operands; here an operand element costs
// 4.5 VALU instructions (v_cvt_pk_bf16_f32 x 3, two expands, v_pk_add_f32 x 2 per PAIR) per workgroup that stages it, issued from waves
// whose partners keep the matrix pipe busy, and every loader of the library (dense in both layouts, im2col gathers) feeds it unchanged.
//
// LDS image (per operand): three planes [plane][row][32 k] of bf16, 64 B per row; the four 16-byte chunks (8 k) of a row are stored at
// chunk ^ swz(row), swz(row) = bit2(row) | (bit1(row) ^ bit3(row)) << 1.  With that swizzle (searched with the bank model of
// MI355X_MICROARCH.md, LDS section: ds_read_b128 is served in the 16-lane groups {0-3,12-15,20-27}, ...) the fragment read of an MFMA block -- lane l
// reads row (l & 31), chunk 2 s + (l >> 5): ONE ds_read_b128 per (block, plane, 16-k step) -- is conflict-free, and so are the stores of the
// k-contiguous loader (ds_write_b64) and of the im2col loader (ds_write_b128); round 1's (row >> 1) & 3 was 2-way on the reads.
#pragma once
#include "gemm_core.h"

namespace segx {

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16v2 __attribute__((ext_vector_type(2)));
typedef float f32v2 __attribute__((ext_vector_type(2)));

// ws_gptr / ws_uniform_base / ws_load: common.h
constexpr int X6_ROWB = 64;                         // bytes of one LDS row of one plane: 32 k of bf16
__device__ __forceinline__ int x6_swz(int row) { return ((row >> 2) & 1) | ((((row >> 1) ^ (row >> 3)) & 1) << 1); }
__device__ __forceinline__ int x6_off(int row, int chunk) { return row * X6_ROWB + ((chunk ^ x6_swz(row)) << 4); }
template <int ROWS> struct X6Plane { static constexpr int bytes = ROWS * X6_ROWB; };

#ifndef SEGX_X6_SPLIT
#define SEGX_X6_SPLIT 1            // 0: v_pk_add_f32 residuals (r02), 1: scalar v_sub_f32 residuals (product), 2: v_dot2c_f32_bf16 residuals (bench)
#endif
// (x0, x1) -> three packed bf16 pairs (element 0 in the low half): x = hi + mid + lo, each step rounded to nearest even
struct Split2 { unsigned h, m, l; };
__device__ __forceinline__ Split2 split3_pair(float x0, float x1) {
#if SEGX_X6_SPLIT == 1
    // the two residual subtractions as scalar v_sub_f32: v_pk_add_f32 issues at a fraction of the plain-VALU rate (r03_a: the wave-specialised
    // kernel's producers went 179 -> 219 TFLOP/s on 24576 x 1792 x 1792 with eleven plain instructions per pair instead of nine with two packed adds)
    const f32v2 v0 = {x0, x1};
    const unsigned hb = __builtin_bit_cast(unsigned, __builtin_convertvector(v0, bf16v2));
    float a0 = x0 - __uint_as_float(hb << 16), a1 = x1 - __uint_as_float(hb & 0xFFFF0000u);
    SEGX_PIN(a0);                                            // keeps the SLP vectoriser from re-packing the pair
    const f32v2 v1 = {a0, a1};
    const unsigned mb = __builtin_bit_cast(unsigned, __builtin_convertvector(v1, bf16v2));
    float b0 = a0 - __uint_as_float(mb << 16), b1 = a1 - __uint_as_float(mb & 0xFFFF0000u);
    SEGX_PIN(b0);
    const f32v2 v2 = {b0, b1};
    Split2 o; o.h = hb; o.m = mb; o.l = __builtin_bit_cast(unsigned, __builtin_convertvector(v2, bf16v2));
    return o;
#elif SEGX_X6_SPLIT == 2
    // bench build: residual = x - bf16 element through v_dot2c_f32_bf16 with the constants (-1, 0) / (0, -1): no expansion of the packed pair
    const f32v2 v0 = {x0, x1};
    const unsigned hb = __builtin_bit_cast(unsigned, __builtin_convertvector(v0, bf16v2));
    float a0 = x0, a1 = x1;
    asm volatile("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(a0) : "v"(hb), "v"(0x0000BF80u));
    asm volatile("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(a1) : "v"(hb), "v"(0xBF800000u));
    const f32v2 v1 = {a0, a1};
    const unsigned mb = __builtin_bit_cast(unsigned, __builtin_convertvector(v1, bf16v2));
    asm volatile("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(a0) : "v"(mb), "v"(0x0000BF80u));
    asm volatile("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(a1) : "v"(mb), "v"(0xBF800000u));
    const f32v2 v2 = {a0, a1};
    Split2 o; o.h = hb; o.m = mb; o.l = __builtin_bit_cast(unsigned, __builtin_convertvector(v2, bf16v2));
    return o;
#endif
    const f32v2 v = {x0, x1};
    const bf16v2 h = __builtin_convertvector(v, bf16v2);
    const f32v2 r1 = v - __builtin_convertvector(h, f32v2);
    const bf16v2 m = __builtin_convertvector(r1, bf16v2);
    const f32v2 r2 = r1 - __builtin_convertvector(m, f32v2);
    const bf16v2 l = __builtin_convertvector(r2, bf16v2);
    Split2 s;
    s.h = __builtin_bit_cast(unsigned, h); s.m = __builtin_bit_cast(unsigned, m); s.l = __builtin_bit_cast(unsigned, l);
    return s;
}
// eight consecutive k of one row -> one 16-byte chunk in each plane
template <int PLANE_BYTES>
__device__ __forceinline__ void x6_store8(unsigned char* __restrict__ P, int off, const float (&v)[8]) {
    const Split2 a = split3_pair(v[0], v[1]), b = split3_pair(v[2], v[3]), c = split3_pair(v[4], v[5]), d = split3_pair(v[6], v[7]);
    *reinterpret_cast<uint4*>(P + off) = make_uint4(a.h, b.h, c.h, d.h);
    *reinterpret_cast<uint4*>(P + PLANE_BYTES + off) = make_uint4(a.m, b.m, c.m, d.m);
    *reinterpret_cast<uint4*>(P + 2 * PLANE_BYTES + off) = make_uint4(a.l, b.l, c.l, d.l);
}
// four consecutive k of one row -> half a chunk (8 bytes) in each plane
template <int PLANE_BYTES>
__device__ __forceinline__ void x6_store4(unsigned char* __restrict__ P, int off, float v0, float v1, float v2, float v3) {
    const Split2 a = split3_pair(v0, v1), b = split3_pair(v2, v3);
    uint2 h, m, l;
    h.x = a.h; h.y = b.h; m.x = a.m; m.y = b.m; l.x = a.l; l.y = b.l;
    *reinterpret_cast<uint2*>(P + off) = h;
    *reinterpret_cast<uint2*>(P + PLANE_BYTES + off) = m;
    *reinterpret_cast<uint2*>(P + 2 * PLANE_BYTES + off) = l;
}

// packed form (split now, store later): 2 fp32 registers -> 3 dwords of bf16 pairs; a group of 8 k -> uint4 per plane, of 4 k -> uint2 per plane
struct Packed8 { uint4 h, m, l; };
struct Packed4 { uint2 h, m, l; };
__device__ __forceinline__ Packed8 x6_split8(const float (&v)[8]) {
    const Split2 a = split3_pair(v[0], v[1]), b = split3_pair(v[2], v[3]), c = split3_pair(v[4], v[5]), d = split3_pair(v[6], v[7]);
    Packed8 o; o.h = make_uint4(a.h, b.h, c.h, d.h); o.m = make_uint4(a.m, b.m, c.m, d.m); o.l = make_uint4(a.l, b.l, c.l, d.l);
    return o;
}
__device__ __forceinline__ Packed4 x6_split4(float v0, float v1, float v2, float v3) {
    const Split2 a = split3_pair(v0, v1), b = split3_pair(v2, v3);
    Packed4 o; o.h.x = a.h; o.h.y = b.h; o.m.x = a.m; o.m.y = b.m; o.l.x = a.l; o.l.y = b.l;
    return o;
}
template <int PLANE_BYTES> __device__ __forceinline__ void x6_put8(unsigned char* __restrict__ P, int off, const Packed8& v) {
    *reinterpret_cast<uint4*>(P + off) = v.h; *reinterpret_cast<uint4*>(P + PLANE_BYTES + off) = v.m; *reinterpret_cast<uint4*>(P + 2 * PLANE_BYTES + off) = v.l;
}
template <int PLANE_BYTES> __device__ __forceinline__ void x6_put4(unsigned char* __restrict__ P, int off, const Packed4& v) {
    *reinterpret_cast<uint2*>(P + off) = v.h; *reinterpret_cast<uint2*>(P + PLANE_BYTES + off) = v.m; *reinterpret_cast<uint2*>(P + 2 * PLANE_BYTES + off) = v.l;
}

// ---- dense operand loaders (float4-legal operands only: the host sends everything else to the fp32 engine) ---------------------------
// Loader concept of this engine: NREG fp32 registers per thread and k-tile;
//   unsigned load6(float (&r)[NREG], int k0, int kend, int tid) const   global -> registers (unconditional clamped loads) + validity mask
//   void store6(float (&r)[NREG], unsigned okmask, unsigned char* P, int tid) const   zero the invalid ones, split, write the three planes
template <bool KC, int ROWS>
struct DenseLoader6;

// k-contiguous operand: the fp32 engine's piece map (piece f = tid + 256 i -> row f / 8, four consecutive k at 4 (f % 8))
template <int ROWS>
struct DenseLoader6<true, ROWS> {
    static constexpr int NPT = ROWS * BKT / 1024, NREG = 4 * NPT;
    const float* base; int64_t s_row, s_k; int row0, rows;
    __device__ __forceinline__ unsigned load6(float (&r)[NREG], int k0, int kend, int tid) const {
        float4 t4[NPT];
        const unsigned ok = load_tile<true, true, ROWS>(t4, base, s_row, s_k, row0, rows, k0, kend, tid);
#pragma unroll
        for (int i = 0; i < NPT; ++i) { r[4 * i] = t4[i].x; r[4 * i + 1] = t4[i].y; r[4 * i + 2] = t4[i].z; r[4 * i + 3] = t4[i].w; }
        return ok;
    }
    __device__ __forceinline__ void store6(float (&r)[NREG], unsigned okmask, unsigned char* __restrict__ P, int tid) const {
        const unsigned full = NPT == 8 ? 0xFFFFFFFFu : ((1u << (4 * NPT)) - 1u);
        if (okmask != full) {
#pragma unroll
            for (int e = 0; e < NREG; ++e) r[e] = ((okmask >> e) & 1u) ? r[e] : 0.f;
        }
#pragma unroll
        for (int i = 0; i < NPT; ++i) {
            const int f = tid + 256 * i, row = f >> 3, kc = f & 7;
            x6_store4<X6Plane<ROWS>::bytes>(P, x6_off(row, kc >> 1) + ((kc & 1) << 3), r[4 * i], r[4 * i + 1], r[4 * i + 2], r[4 * i + 3]);
        }
    }
    // split-early form (the split runs under the MFMAs of the previous k-tile, the packed registers are stored after the barrier)
    struct Packed { Packed4 g[NPT]; };
    __device__ __forceinline__ void split6(float (&r)[NREG], unsigned okmask, Packed& pk) const {
        const unsigned full = NPT == 8 ? 0xFFFFFFFFu : ((1u << (4 * NPT)) - 1u);
        if (okmask != full) {
#pragma unroll
            for (int e = 0; e < NREG; ++e) r[e] = ((okmask >> e) & 1u) ? r[e] : 0.f;
        }
#pragma unroll
        for (int i = 0; i < NPT; ++i) pk.g[i] = x6_split4(r[4 * i], r[4 * i + 1], r[4 * i + 2], r[4 * i + 3]);
    }
    __device__ __forceinline__ void storep6(const Packed& pk, unsigned char* __restrict__ P, int tid) const {
#pragma unroll
        for (int i = 0; i < NPT; ++i) {
            const int f = tid + 256 * i, row = f >> 3, kc = f & 7;
            x6_put4<X6Plane<ROWS>::bytes>(P, x6_off(row, kc >> 1) + ((kc & 1) << 3), pk.g[i]);
        }
    }
};

// row-contiguous operand: a thread owns TWO adjacent rows (one 8-byte load per k) and KQ = ROWS / 16 consecutive k, so that the
// transposition to k-contiguous fragments happens in registers: rows 2 rp, 2 rp + 1 with rp = tid % (ROWS / 2), k = KQ (tid / (ROWS / 2)) + j.
// A wave's load covers ROWS consecutive floats of one k-row (whole 128-byte lines).
template <int ROWS>
struct DenseLoader6<false, ROWS> {
    static_assert(ROWS == 256 || ROWS == 128 || ROWS == 64, "row-contiguous x6 loader: 64, 128 or 256 rows");
    static constexpr int KQ = ROWS / 16, NREG = 2 * KQ, RP = ROWS / 2;
    const float* base; int64_t s_row, s_k; int row0, rows;
    __device__ __forceinline__ unsigned load6(float (&r)[NREG], int k0, int kend, int tid) const {
        const int row = row0 + 2 * (tid % RP), kb = k0 + KQ * (tid / RP);
        const bool rok = row < rows;                                  // rows % 4 == 0 and row even: the pair is inside or outside together
        const float* p = base + (rok ? row : rows - 2);
        unsigned okmask = 0;
#pragma unroll
        for (int j = 0; j < KQ; ++j) {
            const int k = kb + j;
            const bool ok = rok && k < kend;
            const float2 v = *reinterpret_cast<const float2*>(p + (int64_t)(k < kend ? k : kend - 1) * s_k);
            r[2 * j] = v.x; r[2 * j + 1] = v.y;
            okmask |= (ok ? 3u : 0u) << (2 * j);
        }
        return okmask;
    }
    __device__ __forceinline__ void store6(float (&r)[NREG], unsigned okmask, unsigned char* __restrict__ P, int tid) const {
        const unsigned full = NREG >= 32 ? 0xFFFFFFFFu : ((1u << (NREG & 31)) - 1u);
        if (okmask != full) {
#pragma unroll
            for (int e = 0; e < NREG; ++e) r[e] = ((okmask >> e) & 1u) ? r[e] : 0.f;
        }
        const int row = 2 * (tid % RP), kg = tid / RP;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            if (KQ == 16) {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const float v[8] = {r[16 * h + e], r[16 * h + 2 + e], r[16 * h + 4 + e], r[16 * h + 6 + e], r[16 * h + 8 + e], r[16 * h + 10 + e],
                                        r[16 * h + 12 + e], r[16 * h + 14 + e]};
                    x6_store8<X6Plane<ROWS>::bytes>(P, x6_off(row + e, 2 * kg + h), v);
                }
            } else if (KQ == 8) {
                const float v[8] = {r[e], r[2 + e], r[4 + e], r[6 + e], r[8 + e], r[10 + e], r[12 + e], r[14 + e]};
                x6_store8<X6Plane<ROWS>::bytes>(P, x6_off(row + e, kg), v);
            } else {
                x6_store4<X6Plane<ROWS>::bytes>(P, x6_off(row + e, kg >> 1) + ((kg & 1) << 3), r[e], r[2 + e], r[4 + e], r[6 + e]);
            }
        }
    }
    struct Packed { Packed8 g8[KQ >= 8 ? KQ / 4 : 1]; Packed4 g4[2]; };       // KQ 16: 2 rows x 2 octets; 8: 2 rows x 1 octet; 4: 2 rows x 1 quad
    __device__ __forceinline__ void split6(float (&r)[NREG], unsigned okmask, Packed& pk) const {
        const unsigned full = NREG >= 32 ? 0xFFFFFFFFu : ((1u << (NREG & 31)) - 1u);
        if (okmask != full) {
#pragma unroll
            for (int e = 0; e < NREG; ++e) r[e] = ((okmask >> e) & 1u) ? r[e] : 0.f;
        }
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            if (KQ >= 8) {
#pragma unroll
                for (int h = 0; h < KQ / 8; ++h) {
                    const float v[8] = {r[16 * h + e], r[16 * h + 2 + e], r[16 * h + 4 + e], r[16 * h + 6 + e], r[16 * h + 8 + e], r[16 * h + 10 + e],
                                        r[16 * h + 12 + e], r[16 * h + 14 + e]};
                    pk.g8[e * (KQ / 8) + h] = x6_split8(v);
                }
            } else {
                pk.g4[e] = x6_split4(r[e], r[2 + e], r[4 + e], r[6 + e]);
            }
        }
    }
    __device__ __forceinline__ void storep6(const Packed& pk, unsigned char* __restrict__ P, int tid) const {
        const int row = 2 * (tid % RP), kg = tid / RP;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            if (KQ >= 8) {
#pragma unroll
                for (int h = 0; h < KQ / 8; ++h) x6_put8<X6Plane<ROWS>::bytes>(P, x6_off(row + e, (KQ / 8) * kg + h), pk.g8[e * (KQ / 8) + h]);
            } else {
                x6_put4<X6Plane<ROWS>::bytes>(P, x6_off(row + e, kg >> 1) + ((kg & 1) << 3), pk.g4[e]);
            }
        }
    }
};

template <class Cfg> struct X6Lds { static constexpr int A_BYTES = 3 * X6Plane<Cfg::BM>::bytes, B_BYTES = 3 * X6Plane<Cfg::BN>::bytes, BYTES = A_BYTES + B_BYTES; };

// acc += A_tile . B_tile^T over k in [kbeg, kend).  One LDS stage (48 KB at 128 x 128: three workgroups per CU cover each other's barriers --
// measured in round 1: occupancy beats double buffering at this tile size), the next k-tile's global loads in flight under the MFMAs.
// VAR 6 = split-early schedule (a product candidate: same results as 0): the next tile's registers are split into packed bf16 between the two
// 16-k MFMA groups of the current tile, so that the conversion arithmetic issues under matrix instructions of the SAME wave; after the
// barrier only the packed stores remain.
// VAR (bench / bisect only, segx_tune knob 6; results are only defined for 0, 1 and 6): 1 = raised wave priority during the MFMA phase;
// ablations that leave parts of the k-tile loop out to price them: 2 = no split arithmetic, 3 = no LDS stores, 4 = no global loads after the
// first tile, 5 = MFMAs and fragment reads only (no loads, stores or barriers).
template <class Cfg, class LA, class LB, int VAR = 0>
__device__ __forceinline__ void gemm_mainloop_x6(f32x16 (&acc)[Cfg::MI][Cfg::NJ], const LA& la, const LB& lb, int kbeg, int kend,
                                                 unsigned char* __restrict__ lds) {
    constexpr int MI = Cfg::MI, NJ = Cfg::NJ, PA = X6Plane<Cfg::BM>::bytes, PB = X6Plane<Cfg::BN>::bytes;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / Cfg::WN, wn = wave % Cfg::WN;
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    if (kbeg >= kend) return;
    unsigned char* const LA_ = lds;
    unsigned char* const LB_ = lds + X6Lds<Cfg>::A_BYTES;
    float ra[LA::NREG], rb[LB::NREG];
    unsigned oka = la.load6(ra, kbeg, kend, tid), okb = lb.load6(rb, kbeg, kend, tid);
    const int arow = wm * (32 * MI) + (lane & 31), brow = wn * (32 * NJ) + (lane & 31), kh = lane >> 5;
    if constexpr (VAR == 6) {
        typename LA::Packed pa; typename LB::Packed pb;
        la.split6(ra, oka, pa); lb.split6(rb, okb, pb);
        for (int kt = kbeg; kt < kend; kt += BKT) {
            __syncthreads();
            la.storep6(pa, LA_, tid); lb.storep6(pb, LB_, tid);
            __syncthreads();
            const bool more = kt + BKT < kend;
            if (more) { oka = la.load6(ra, kt + BKT, kend, tid); okb = lb.load6(rb, kt + BKT, kend, tid); }
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const int chunk = 2 * s + kh;
                bf16x8 a[MI][3];
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int p = 0; p < 3; ++p) a[i][p] = *reinterpret_cast<const bf16x8*>(LA_ + p * PA + x6_off(arow + 32 * i, chunk));
                // the next tile's conversion arithmetic, UNCONDITIONALLY (after the last tile it converts stale registers, nothing is stored):
                // in one basic block with the second 16-k MFMA group, and dealt out between its matrix instructions by the scheduling hints below
                if (s == 1) { la.split6(ra, oka, pa); lb.split6(rb, okb, pb); }
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    bf16x8 b[3];
#pragma unroll
                    for (int p = 0; p < 3; ++p) b[p] = *reinterpret_cast<const bf16x8*>(LB_ + p * PB + x6_off(brow + 32 * j, chunk));
#pragma unroll
                    for (int i = 0; i < MI; ++i) {
                        f32x16 c = acc[i][j];
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[2], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][2], b[0], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], b[1], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[1], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], b[0], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[0], c, 0, 0, 0);
                        acc[i][j] = c;
                    }
                }
                if (s == 1) {
#pragma unroll
                    for (int q = 0; q < MI * NJ * 6; ++q) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);       // one MFMA
                        __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);       // five VALU (4.5 per element pair x 24 pairs / 24 MFMAs)
                    }
                }
            }
        }
        return;
    }
    for (int kt = kbeg; kt < kend; kt += BKT) {
        if (VAR != 5) __syncthreads();                    // every wave has read the previous tile's fragments
        if (VAR == 2) {                                   // ablation: the stores without the split arithmetic
            unsigned* wa = reinterpret_cast<unsigned*>(LA_) + tid * 4; unsigned* wb = reinterpret_cast<unsigned*>(LB_) + tid * 4;
#pragma unroll
            for (int e = 0; e + 3 < LA::NREG; e += 4) *reinterpret_cast<float4*>(wa + 1024 * (e / 4)) = make_float4(ra[e], ra[e + 1], ra[e + 2], ra[e + 3]);
#pragma unroll
            for (int e = 0; e + 3 < LB::NREG; e += 4) *reinterpret_cast<float4*>(wb + 1024 * (e / 4)) = make_float4(rb[e], rb[e + 1], rb[e + 2], rb[e + 3]);
        } else if (VAR != 3 && VAR != 5) {
            la.store6(ra, oka, LA_, tid);
            lb.store6(rb, okb, LB_, tid);
        } else if (VAR == 3) {                            // keep the loaded values alive without storing them
#pragma unroll
            for (int e = 0; e < LA::NREG; ++e) asm volatile("" :: "v"(ra[e]));
#pragma unroll
            for (int e = 0; e < LB::NREG; ++e) asm volatile("" :: "v"(rb[e]));
        }
        if (VAR != 5) __syncthreads();
        if (VAR < 4 && kt + BKT < kend) { oka = la.load6(ra, kt + BKT, kend, tid); okb = lb.load6(rb, kt + BKT, kend, tid); }
        if (VAR == 1) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int chunk = 2 * s + kh;                 // lane -> (row lane & 31, the 8 k of half lane >> 5 of this 16-k step)
#define SEGX_X6_MFMA6(C, A_, B_)                                                                               \
    C = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A_[0], B_[2], C, 0, 0, 0);     /* hi . lo   */                \
    C = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A_[2], B_[0], C, 0, 0, 0);     /* lo . hi   */                \
    C = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A_[1], B_[1], C, 0, 0, 0);     /* mid . mid */                \
    C = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A_[0], B_[1], C, 0, 0, 0);     /* hi . mid  */                \
    C = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A_[1], B_[0], C, 0, 0, 0);     /* mid . hi  */                \
    C = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A_[0], B_[0], C, 0, 0, 0);     /* hi . hi   */
            if (MI > NJ) {                                // tall wave tile: the NJ x 3 B fragments stay, the A fragments stream one row block at a time
                bf16x8 b[NJ][3];
#pragma unroll
                for (int j = 0; j < NJ; ++j)
#pragma unroll
                    for (int p = 0; p < 3; ++p) b[j][p] = *reinterpret_cast<const bf16x8*>(LB_ + p * PB + x6_off(brow + 32 * j, chunk));
#pragma unroll
                for (int i = 0; i < MI; ++i) {
                    bf16x8 a[3];
#pragma unroll
                    for (int p = 0; p < 3; ++p) a[p] = *reinterpret_cast<const bf16x8*>(LA_ + p * PA + x6_off(arow + 32 * i, chunk));
#pragma unroll
                    for (int j = 0; j < NJ; ++j) { f32x16 c = acc[i][j]; SEGX_X6_MFMA6(c, a, b[j]) acc[i][j] = c; }
                }
            } else {                                      // the MI x 3 A fragments stay, one column block of B at a time
                bf16x8 a[MI][3];
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int p = 0; p < 3; ++p) a[i][p] = *reinterpret_cast<const bf16x8*>(LA_ + p * PA + x6_off(arow + 32 * i, chunk));
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    bf16x8 b[3];
#pragma unroll
                    for (int p = 0; p < 3; ++p) b[p] = *reinterpret_cast<const bf16x8*>(LB_ + p * PB + x6_off(brow + 32 * j, chunk));
#pragma unroll
                    for (int i = 0; i < MI; ++i) { f32x16 c = acc[i][j]; SEGX_X6_MFMA6(c, a[i], b) acc[i][j] = c; }
                }
            }
#undef SEGX_X6_MFMA6
        }
        if (VAR == 1) __builtin_amdgcn_s_setprio(0);
    }
}

// host-side cost model of this engine (same form as kTiles / model_us of gemm_core.h; constants from the r02 device sweeps)
struct TileInfo6 { int id, bm, bn, wg_per_cu; float ktile_us, fixed_us; };
// r03_f sweep with the scalar-subtraction split (24576 x 1792 x 1792 x 4: 3.14 ms on 14 rounds of 56 k-tiles; 8192^3: 5.55 ms on 5.6 rounds of 256):
// 3.9 us per k-tile and round + 8 us per workgroup for 128 x 128; the smaller tiles scaled from the r02_a sweep (3.4 / 2.9 at 4.5)
static const TileInfo6 kTiles6[] = {{SEGX_TILE_128x128, 128, 128, 3, 3.9f, 8.0f},
                                    {SEGX_TILE_64x128, 64, 128, 4, 3.0f, 3.0f},
                                    {SEGX_TILE_64x64, 64, 64, 6, 2.6f, 2.0f}};
inline double model_us6(const TileInfo6& ti, int M, int N, int K, int nbatch, int sk) {
    const TileInfo t{ti.id, ti.bm, ti.bn, ti.wg_per_cu, ti.ktile_us, ti.fixed_us};
    return model_us(t, M, N, K, nbatch, sk);
}

}  // namespace segx
