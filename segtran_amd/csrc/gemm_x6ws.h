// gemm_x6ws.h -- the wave-specialised form of the bf16x6 tile engine (gemm_x6.h): a PERSISTENT 512-thread workgroup per CU whose waves 0-3
// (one per SIMD) only read fragments and issue matrix instructions, while waves 4-7 (their SIMD partners) only load fp32 operand tiles from
// global memory, split them into the three bf16 planes and store them to LDS.  Arithmetic, LDS image, loaders, epilogue and split-K slabs are
// those of gemm_x6.h -- every accumulator sees the same six products per 16 k in the same order, so results are bit-identical to that engine.
//
// Why (r02 profile of the 4-wave kernel: matrix pipe busy 0.36-0.52, waves 55-60 % issue-stalled): there every wave alternates between a
// VALU phase (split + ds_write, ~170 issues) and an MFMA phase between two barriers per k-tile, three workgroups per CU covering each other
// by luck.  Here the two kinds of work never share a wave:
//   * a consumer wave's instruction stream is ds_read_b128 + v_mfma only (MI x NJ x 12 matrix instructions per 32-k stage: 3072 cycles of
//     the SIMD's matrix pipe at 256 x 128), so the pipe idles only at the ONE barrier per stage;
//   * the producer wave on the same SIMD owns the VALU issue slots the consumer does not use (MI355X_MICROARCH.md, "two waves per SIMD":
//     matrix and vector pipes are separate, VALU issue goes to whoever asks) -- 48 elements = 216 conversion instructions + 36 ds_write_b64
//     per thread and stage against 96 matrix instructions of its partner;
//   * LDS is double-buffered (2 x 72 KB at 256 x 128): producers fill stage s+1 while consumers read stage s; the global loads of stage s+2
//     are issued right after the stores of s+1 into the same registers (a whole stage time, ~1.3 us, of latency cover);
//   * the workgroup is persistent and its stage stream runs ACROSS output tiles: while the consumers write a finished tile (epilogue), the
//     producers have already stored the first stage of the next tile and have its second in flight -- no per-tile prologue bubble, which is what
//     the K = 256 attention GEMMs (8 stages per tile) were losing.
// One barrier per stage, `s_waitcnt lgkmcnt(0); s_barrier` by hand: __syncthreads() would also drain vmcnt, i.e. the producers' prefetch.
#pragma once
#include "gemm_x6.h"

#ifndef SEGX_LDS_BARRIER
#define SEGX_LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#endif

namespace segx {

template <class Cfg> struct X6WsLds { static constexpr int STAGE = X6Lds<Cfg>::BYTES, BYTES = 2 * STAGE; };

// work item `item` of a persistent launch -> tile coordinates (tile fastest, N fastest inside: neighbours share operand panels), then batch, then k-slab
template <class Cfg>
__device__ __forceinline__ TileCoord ws_item_coord(const GemmArgs& g, int item) {
    TileCoord t;
    const int ntile = g.tiles_m * g.tiles_n;
    const int tile = item % ntile, rest = item / ntile;
    const int inner = g.mfast ? g.tiles_m : g.tiles_n;                          // tile_walk(): M fastest puts the tiles sharing a B column-panel next to each other
    const int q = tile / inner, r = tile - q * inner;
    const int tm = g.mfast ? r : q, tn = g.mfast ? q : r;
    t.zb = rest % g.nbatch; t.zk = rest / g.nbatch;
    t.z0 = t.zb / g.nb1; t.z1 = t.zb - t.z0 * g.nb1;
    t.m0 = tm * Cfg::BM; t.n0 = tn * Cfg::BN;
    t.kbeg = t.zk * g.k_chunk;
    t.kend = (t.kbeg + g.k_chunk < g.K) ? t.kbeg + g.k_chunk : g.K;
    return t;
}
// workgroup b of G (G % 8 == 0) -> its position inside a round of G items: the dispatcher places workgroup b on XCD b % 8 (observed, speed only),
// so the G / 8 workgroups of one XCD take a CONTIGUOUS run of items and share their panels in that XCD's L2
__device__ __forceinline__ int ws_round_pos(int b, int G) { return (b & 7) * (G >> 3) + (b >> 3); }

// one 32-k stage of matrix work of a consumer wave: acc[i][j] += A(rows arow + 32 i) . B(rows brow + 32 j)^T from the three-plane LDS images
template <class Cfg>
__device__ __forceinline__ void x6ws_stage_mfma(f32x16 (&acc)[Cfg::MI][Cfg::NJ], const unsigned char* __restrict__ LA_, const unsigned char* __restrict__ LB_,
                                                int arow, int brow, int kh) {
    constexpr int MI = Cfg::MI, NJ = Cfg::NJ, PA = X6Plane<Cfg::BM>::bytes, PB = X6Plane<Cfg::BN>::bytes;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const int chunk = 2 * s + kh;                     // lane -> (row lane & 31, the 8 k of half lane >> 5 of this 16-k step)
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
            // six products per accumulator, small terms first (the order of gemm_x6.h); consecutive matrix instructions go to DIFFERENT
            // accumulators so that none waits for its predecessor's result
#define SEGX_X6WS_P(PA_, PB_)                                                                                              \
    _Pragma("unroll") for (int j = 0; j < NJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[PA_], b[j][PB_], acc[i][j], 0, 0, 0);
            SEGX_X6WS_P(0, 2) SEGX_X6WS_P(2, 0) SEGX_X6WS_P(1, 1) SEGX_X6WS_P(0, 1) SEGX_X6WS_P(1, 0) SEGX_X6WS_P(0, 0)
#undef SEGX_X6WS_P
        }
    }
}

// Loader factory concept: `void make(const GemmArgs&, const TileCoord&, LA&, LB&, int ptid) const` builds the two operand loaders of a work item.
// The stream of stages (item, k-tile) a producer walks, one ahead of the stores and two ahead of the consumers.
template <class Cfg, class MK>
struct X6WsStream {
    typename MK::LA la; typename MK::LB lb;
    int r, k, kend; bool valid;
    int ptid;
    __device__ __forceinline__ void open(const GemmArgs& g, const MK& mk, int pos, int G, int total) {   // first non-empty item at or after round r
        for (;; ++r) {
            const int item = r * G + pos;
            if (item >= total) { valid = false; return; }
            const TileCoord t = ws_item_coord<Cfg>(g, item);
            if (t.kbeg < t.kend) { mk.make(g, t, la, lb, ptid); k = t.kbeg; kend = t.kend; valid = true; return; }
        }
    }
    // past the end of the stream (valid = false) the loaders and (k, kend) keep the last stage's values: loads issued from them are legal
    __device__ __forceinline__ void next(const GemmArgs& g, const MK& mk, int pos, int G, int total) {
        if (!valid) return;
        if (k + BKT >= kend) { ++r; open(g, mk, pos, G, total); }
        else k += BKT;
    }
};

// a producer's register set -> the three-plane LDS images of one stage (VAR: the bench ablations of x6ws_body)
template <int VAR, class Cfg, class LA, class LB, int A_BYTES_ = X6Lds<Cfg>::A_BYTES>
__device__ __forceinline__ void x6ws_put(const LA& la, const LB& lb, float (&ra)[LA::NREG], float (&rb)[LB::NREG], unsigned oka, unsigned okb,
                                         unsigned char* __restrict__ P, int ptid) {
    if (VAR == 2) {                                        // stores of the same width and count without the conversion arithmetic
        unsigned* wa = reinterpret_cast<unsigned*>(P) + ptid * 2;
#pragma unroll
        for (int e = 0; e + 1 < LA::NREG; e += 2) {
            uint2 w; w.x = __float_as_uint(ra[e]); w.y = __float_as_uint(ra[e + 1]);
            *reinterpret_cast<uint2*>(wa + 512 * (e / 2)) = w;
            if ((e & 2) == 0) *reinterpret_cast<uint2*>(wa + 512 * (e / 2) + 256) = w;
        }
        unsigned* wb = reinterpret_cast<unsigned*>(P + A_BYTES_) + ptid * 2;
#pragma unroll
        for (int e = 0; e + 1 < LB::NREG; e += 2) {
            uint2 w; w.x = __float_as_uint(rb[e]); w.y = __float_as_uint(rb[e + 1]);
            *reinterpret_cast<uint2*>(wb + 512 * (e / 2)) = w;
            if ((e & 2) == 0) *reinterpret_cast<uint2*>(wb + 512 * (e / 2) + 256) = w;
        }
    } else if (VAR == 4 || VAR == 5) {                     // keep the loaded values alive without storing them
#pragma unroll
        for (int e = 0; e < LA::NREG; ++e) asm volatile("" :: "v"(ra[e]));
#pragma unroll
        for (int e = 0; e < LB::NREG; ++e) asm volatile("" :: "v"(rb[e]));
    } else { la.store6(ra, oka, P, ptid); lb.store6(rb, okb, P + A_BYTES_, ptid); }
}

// EPI as gemm_epilogue.  PRIO (segx_tune knob 6; results are only defined for 0 and 1): 1 = consumers run at raised wave priority; ablations that
// price the producers' parts: 2 = no split arithmetic (raw bits stored), 3 = no global loads after a work item's first stage, 4 = no LDS stores,
// 5 = producers only keep the barrier count (what the consumers reach alone).
template <class Cfg, class MK, int EPI, int PRIO = 0>
__device__ __forceinline__ void x6ws_body(const GemmArgs& g, const MK& mk, unsigned char* __restrict__ lds) {
    using LA = typename MK::LA; using LB = typename MK::LB;
    constexpr int STAGE = X6Lds<Cfg>::BYTES, A_BYTES = X6Lds<Cfg>::A_BYTES;
    const int wave = SEGX_WAVE_UNIFORM((int)(threadIdx.x >> 6));
    const int G = gridDim.x, pos = ws_round_pos(blockIdx.x, G);
    const int total = g.tiles_m * g.tiles_n * g.nbatch * g.splitk;
    if (wave >= 4) {
        // ---- producers: global -> registers -> split -> LDS ----------------------------------------------------------------------------
        // Two register sets.  After barrier s (consumers start on stage s) a producer FIRST issues the global loads of stage s+2 into the
        // free set and THEN splits / stores stage s+1 from the other one: a load has a whole stage time to land.  Loads and stores are
        // UNCONDITIONAL (past the end of the stream they re-read / re-write the last stage: harmless): a load inside a branch makes hipcc's
        // s_waitcnt bookkeeping merge "issued" with "not issued" at the join and wait for the NEWEST loads before every split (r03_c: 167
        // TFLOP/s, slower than one register set); straight-line issue gives the counted `vmcnt(loads of one stage)` this schedule needs.
        // The stream's step to the next work item (X6WsStream::next) is scalar-only control flow without memory instructions.
        // The loaders' store6 must not depend on loader state other than the thread index (true for every loader of the library).
        const int ptid = threadIdx.x - 256;
        X6WsStream<Cfg, MK> st; st.r = 0; st.ptid = ptid; st.open(g, mk, pos, G, total);
        if (!st.valid) return;
        float a0[LA::NREG], b0[LB::NREG], a1[LA::NREG], b1[LB::NREG];
        unsigned oka0, okb0, oka1, okb1;
        oka0 = st.la.load6(a0, st.k, st.kend, ptid); okb0 = st.lb.load6(b0, st.k, st.kend, ptid);
        int r_seen = st.r;
        st.next(g, mk, pos, G, total);
        bool have1 = st.valid, have0;
        oka1 = st.la.load6(a1, st.k, st.kend, ptid); okb1 = st.lb.load6(b1, st.k, st.kend, ptid);
        r_seen = st.r; st.next(g, mk, pos, G, total);
        x6ws_put<PRIO, Cfg, typename MK::LA, typename MK::LB, A_BYTES>(st.la, st.lb, a0, b0, oka0, okb0, lds, ptid);        // stage 0 -> buffer 0
        int par = 0;
        for (;;) {
            SEGX_LDS_BARRIER();                            // the stage in buffer `par` is complete; buffer par ^ 1 has been read to the end
            par ^= 1;
            if (!have1) break;
            have0 = st.valid;
            if (!((PRIO == 3 || PRIO == 5) && st.r == r_seen)) { oka0 = st.la.load6(a0, st.k, st.kend, ptid); okb0 = st.lb.load6(b0, st.k, st.kend, ptid); }
            r_seen = st.r; st.next(g, mk, pos, G, total);
            if (PRIO != 5) x6ws_put<PRIO, Cfg, typename MK::LA, typename MK::LB, A_BYTES>(st.la, st.lb, a1, b1, oka1, okb1, lds + par * STAGE, ptid);
            SEGX_LDS_BARRIER();
            par ^= 1;
            if (!have0) break;
            have1 = st.valid;
            if (!((PRIO == 3 || PRIO == 5) && st.r == r_seen)) { oka1 = st.la.load6(a1, st.k, st.kend, ptid); okb1 = st.lb.load6(b1, st.k, st.kend, ptid); }
            r_seen = st.r; st.next(g, mk, pos, G, total);
            if (PRIO != 5) x6ws_put<PRIO, Cfg, typename MK::LA, typename MK::LB, A_BYTES>(st.la, st.lb, a0, b0, oka0, okb0, lds + par * STAGE, ptid);
        }
        return;
    }
    // ---- consumers: fragments + matrix instructions + the epilogue of every finished tile ----------------------------------------------
    if (PRIO == 1) __builtin_amdgcn_s_setprio(1);
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
            x6ws_stage_mfma<Cfg>(acc, P, P + A_BYTES, arow, brow, kh);
            par ^= 1;
        }
        gemm_epilogue<EPI, Cfg, false>(acc, g, t);        // an empty split-K slab writes zeros
    }
}

// ---- dense strided operands (segx_gemm_f32), stream form --------------------------------------------------------------------------------
// Same thread -> element maps and LDS stores as DenseLoader6 (gemm_x6.h), but everything that does not change along k is computed ONCE per
// work item: a wave-uniform operand base (SGPRs) plus one 32-bit byte offset per piece (VGPRs) -- a stage's loads are `global_load ... v_off, s[base]`
// with no address arithmetic on the vector pipe, which belongs to the split.  Only the last stage of a contraction whose length is not a
// multiple of 32 would need a clamped path: the host sends those shapes, and operands whose offsets do not fit 31 bits, to the 4-wave kernels.
template <bool KC, int ROWS> struct WsDense6;
// The contraction range of every work item is a multiple of the 32-k stage (the host sends other shapes to the 4-wave kernels), so a stage
// has no k edge: no clamped path whose loads hipcc would merge with these (and give both VGPR addresses).
template <int ROWS>
struct WsDense6<true, ROWS> : DenseLoader6<true, ROWS> {                     // k-contiguous: piece i -> row (ptid >> 3) + 32 i, floats 4 (ptid & 7) .. + 3
    using Base = DenseLoader6<true, ROWS>;
    static constexpr int NPT = Base::NPT, NREG = Base::NREG;
    unsigned off[NPT]; unsigned rowmask;
    __device__ __forceinline__ void begin(const float* b, int64_t s_row_, int64_t, int row0_, int rows_, int ptid) {
        this->base = b; this->s_row = s_row_; this->s_k = 1; this->row0 = row0_; this->rows = rows_;
        rowmask = 0u;
#pragma unroll
        for (int i = 0; i < NPT; ++i) {
            const int row = row0_ + (ptid >> 3) + 32 * i;
            const bool ok = row < rows_;
            off[i] = (unsigned)(((int64_t)(ok ? row : rows_ - 1) * s_row_ + ((ptid & 7) << 2)) << 2);
            rowmask |= ok ? (0xFu << (4 * i)) : 0u;
        }
    }
    __device__ __forceinline__ unsigned load6(float (&r)[NREG], int k0, int, int) const {
        const ws_gptr b = ws_uniform_base(this->base + k0);
#pragma unroll
        for (int i = 0; i < NPT; ++i) {
            const f32x4 v = ws_load<f32x4>(b, off[i]);
            r[4 * i] = v.x; r[4 * i + 1] = v.y; r[4 * i + 2] = v.z; r[4 * i + 3] = v.w;
        }
        return rowmask;
    }
};

template <int ROWS>
struct WsDense6<false, ROWS> : DenseLoader6<false, ROWS> {                    // row-contiguous: rows 2 rp, 2 rp + 1 (one 8-byte load per k), KQ consecutive k
    using Base = DenseLoader6<false, ROWS>;
    static constexpr int KQ = Base::KQ, NREG = Base::NREG, RP = Base::RP;
    unsigned off[KQ]; bool rok;                                               // byte offset of (row pair, k = k0 + KQ (ptid / RP) + j) from row k0 of the operand
    __device__ __forceinline__ void begin(const float* b, int64_t, int64_t s_k_, int row0_, int rows_, int ptid) {
        this->base = b; this->s_row = 1; this->s_k = s_k_; this->row0 = row0_; this->rows = rows_;
        const int row = row0_ + 2 * (ptid % RP);
        rok = row < rows_;                                                   // rows % 4 == 0 and row even: the pair is inside or outside together
#pragma unroll
        for (int j = 0; j < KQ; ++j) off[j] = (unsigned)(((int64_t)(KQ * (ptid / RP) + j) * s_k_ + (rok ? row : rows_ - 2)) << 2);
    }
    __device__ __forceinline__ unsigned load6(float (&r)[NREG], int k0, int, int) const {
        const ws_gptr bk = ws_uniform_base(this->base + (int64_t)k0 * this->s_k);
#pragma unroll
        for (int j = 0; j < KQ; ++j) {
            const f32v2 v = ws_load<f32v2>(bk, off[j]);
            r[2 * j] = v.x; r[2 * j + 1] = v.y;
        }
        return rok ? ((KQ == 16) ? 0xFFFFFFFFu : ((1u << (NREG & 31)) - 1u)) : 0u;
    }
};

template <class Cfg, bool AKC, bool BKC>
struct DenseMk6 {
    using LA = WsDense6<AKC, Cfg::BM>; using LB = WsDense6<BKC, Cfg::BN>;
    __device__ __forceinline__ void make(const GemmArgs& g, const TileCoord& t, LA& la, LB& lb, int ptid) const {
        la.begin(g.A + t.z0 * g.a_b0 + t.z1 * g.a_b1, g.a_m, g.a_k, t.m0, g.M, ptid);
        lb.begin(g.B + t.z0 * g.b_b0 + t.z1 * g.b_b1, g.b_n, g.b_k, t.n0, g.N, ptid);
    }
};

// B operand split ahead of time (segx_x6_presplit: three bf16 planes [plane][row][K], k contiguous, the same rounding as split3_pair): a producer moves
// 16-byte chunks (8 k of one row and plane) global -> registers -> LDS with NO arithmetic.  The split is what the producers cannot hide (§5c-r3 of
// without it the kernel runs at the consumers-alone rate); a weight matrix is otherwise re-split by every workgroup that stages it
// (the 1792 x 1792 weights of a 24576-row projection: 96 to 192 times per launch).  Same LDS image as the in-kernel split, hence bit-identical results.
template <int ROWS>
struct WsPre6 {
    static constexpr int NPP = ROWS * 4 / 256;              // 16-byte pieces per thread and plane (piece f = ptid + 256 i: row f / 4, chunk f % 4)
    static constexpr int NREG = 3 * NPP * 4;
    const unsigned short* base; int64_t plane;               // plane stride in elements
    unsigned off[NPP]; unsigned rowmask;
    __device__ __forceinline__ void begin(const unsigned short* b, int64_t plane_, int K, int row0, int rows, int ptid) {
        base = b; plane = plane_; rowmask = 0u;
#pragma unroll
        for (int i = 0; i < NPP; ++i) {
            const int f = ptid + 256 * i, row = row0 + (f >> 2);
            const bool ok = row < rows;
            off[i] = (unsigned)((((int64_t)(ok ? row : rows - 1)) * K + ((f & 3) << 3)) << 1);
            rowmask |= ok ? (1u << i) : 0u;
        }
    }
    __device__ __forceinline__ unsigned load6(float (&r)[NREG], int k0, int, int) const {
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            const ws_gptr b = ws_uniform_base(base + p * plane + k0);
#pragma unroll
            for (int i = 0; i < NPP; ++i) {
                const f32x4 v = ws_load<f32x4>(b, off[i]);
                r[4 * (p * NPP + i)] = v.x; r[4 * (p * NPP + i) + 1] = v.y; r[4 * (p * NPP + i) + 2] = v.z; r[4 * (p * NPP + i) + 3] = v.w;
            }
        }
        return rowmask;
    }
    __device__ __forceinline__ void store6(float (&r)[NREG], unsigned okmask, unsigned char* __restrict__ P, int ptid) const {
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int i = 0; i < NPP; ++i) {
                const int f = ptid + 256 * i, e = 4 * (p * NPP + i);
                const bool ok = (okmask >> i) & 1u;
                uint4 w;
                w.x = ok ? __float_as_uint(r[e]) : 0u; w.y = ok ? __float_as_uint(r[e + 1]) : 0u;
                w.z = ok ? __float_as_uint(r[e + 2]) : 0u; w.w = ok ? __float_as_uint(r[e + 3]) : 0u;
                *reinterpret_cast<uint4*>(P + p * X6Plane<ROWS>::bytes + x6_off(f >> 2, f & 3)) = w;
            }
    }
};

template <class Cfg, bool AKC>
struct PreMk6 {
    using LA = WsDense6<AKC, Cfg::BM>; using LB = WsPre6<Cfg::BN>;
    __device__ __forceinline__ void make(const GemmArgs& g, const TileCoord& t, LA& la, LB& lb, int ptid) const {
        la.begin(g.A + t.z0 * g.a_b0 + t.z1 * g.a_b1, g.a_m, g.a_k, t.m0, g.M, ptid);
        lb.begin(g.Bp + t.z0 * g.bp_b0 + t.z1 * g.bp_b1, g.bp_plane, g.K, t.n0, g.N, ptid);
    }
};

// host-side cost model: a persistent launch of one workgroup per CU walks ceil(items / 256) rounds of work items, each costing
// k-tiles x stage time + a per-tile epilogue (consumers only: the matrix pipe idles while a finished tile is written).  Constants from the
// r03_f device sweep: 256 x 128: 24576 x 1792 x 1792 x 4 = 21 rounds x 56 stages in 2.87 ms, 8192^3 = 8 x 256 in 4.78 ms, K = 256 tiles 29 us;
// 128 x 128: 42 x 56 in 3.33 ms, 16 x 256 in 5.77 ms.
static const TileInfo6 kTilesWs[] = {{SEGX_TILE_256x128, 256, 128, 1, 2.3f, 12.0f},
                                     {SEGX_TILE_WS128x128, 128, 128, 1, 1.4f, 4.0f}};
inline double model_us_ws(const TileInfo6& ti, int M, int N, int K, int nbatch, int sk, int grid) {
    const int64_t items = (int64_t)ceil_div(M, ti.bm) * ceil_div(N, ti.bn) * nbatch * sk;
    const int kt = ceil_div(ceil_div(K, sk), BKT);
    const int64_t rounds = (items + grid - 1) / grid;
    return (double)rounds * (kt * ti.ktile_us + ti.fixed_us) + 3.0 + (sk == 1 ? 0.0 : (double)sk * M * N * nbatch * 8.0 / 3.0e6);
}
inline int best_splitk_ws(const TileInfo6& ti, int M, int N, int K, int nbatch, int grid, double* t_out) {
    int best = 1; double best_t = model_us_ws(ti, M, N, K, nbatch, 1, grid);
    if (K >= 1024)
        for (int sk = 2; sk <= 128 && K / sk >= 256; ++sk) {
            const double t = model_us_ws(ti, M, N, K, nbatch, sk, grid);
            if (t < best_t * 0.97) { best = sk; best_t = t; }
        }
    *t_out = best_t;
    return best;
}

}  // namespace segx
