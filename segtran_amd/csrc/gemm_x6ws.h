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
#ifndef SEGX_WAVE_UNIFORM
#define SEGX_WAVE_UNIFORM(x) __builtin_amdgcn_readfirstlane(x)
#endif

namespace segx {

template <class Cfg> struct X6WsLds { static constexpr int STAGE = X6Lds<Cfg>::BYTES, BYTES = 2 * STAGE; };

// work item `item` of a persistent launch -> tile coordinates (tile fastest, N fastest inside: neighbours share operand panels), then batch, then k-slab
template <class Cfg>
__device__ __forceinline__ TileCoord ws_item_coord(const GemmArgs& g, int item) {
    TileCoord t;
    const int ntile = g.tiles_m * g.tiles_n;
    const int tile = item % ntile, rest = item / ntile;
    const int tm = tile / g.tiles_n, tn = tile - tm * g.tiles_n;
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
    __device__ __forceinline__ void next(const GemmArgs& g, const MK& mk, int pos, int G, int total) {
        k += BKT;
        if (k >= kend) { ++r; open(g, mk, pos, G, total); }
    }
};

// EPI as gemm_epilogue.  PRIO: 1 = consumers run at raised wave priority (bench knob).
template <class Cfg, class MK, int EPI, int PRIO = 0>
__device__ __forceinline__ void x6ws_body(const GemmArgs& g, const MK& mk, unsigned char* __restrict__ lds) {
    using LA = typename MK::LA; using LB = typename MK::LB;
    constexpr int STAGE = X6WsLds<Cfg>::STAGE, A_BYTES = X6Lds<Cfg>::A_BYTES;
    const int wave = SEGX_WAVE_UNIFORM((int)(threadIdx.x >> 6));
    const int G = gridDim.x, pos = ws_round_pos(blockIdx.x, G);
    const int total = g.tiles_m * g.tiles_n * g.nbatch * g.splitk;
    if (wave >= 4) {
        // ---- producers: global -> registers -> split -> LDS, one stage ahead of the consumers ------------------------------------------
        const int ptid = threadIdx.x - 256;
        X6WsStream<Cfg, MK> st; st.r = 0; st.ptid = ptid; st.open(g, mk, pos, G, total);
        if (!st.valid) return;
        float ra[LA::NREG], rb[LB::NREG];
        unsigned oka = st.la.load6(ra, st.k, st.kend, ptid), okb = st.lb.load6(rb, st.k, st.kend, ptid);
        st.la.store6(ra, oka, lds, ptid); st.lb.store6(rb, okb, lds + A_BYTES, ptid);                   // stage 0 -> buffer 0
        st.next(g, mk, pos, G, total);
        if (st.valid) { oka = st.la.load6(ra, st.k, st.kend, ptid); okb = st.lb.load6(rb, st.k, st.kend, ptid); }
        int par = 0;
        bool pending = true;                               // a stored stage the consumers have not been released onto yet
        while (pending) {
            SEGX_LDS_BARRIER();                            // stage in buffer `par` is complete; buffer par ^ 1 has been read to the end
            par ^= 1; pending = false;
            if (st.valid) {
                unsigned char* const P = lds + par * STAGE;
                st.la.store6(ra, oka, P, ptid); st.lb.store6(rb, okb, P + A_BYTES, ptid);
                pending = true;
                st.next(g, mk, pos, G, total);
                if (st.valid) { oka = st.la.load6(ra, st.k, st.kend, ptid); okb = st.lb.load6(rb, st.k, st.kend, ptid); }
            }
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
        gemm_epilogue<EPI, Cfg>(acc, g, t);               // an empty split-K slab writes zeros
    }
}

// ---- dense strided operands (segx_gemm_f32), stream form --------------------------------------------------------------------------------
// Same thread -> element maps and LDS stores as DenseLoader6 (gemm_x6.h), but everything that does not change along k is computed ONCE per
// work item: a wave-uniform operand base (SGPRs) plus one 32-bit byte offset per piece (VGPRs) -- a stage's loads are `global_load ... v_off, s[base]`
// with no address arithmetic on the vector pipe, which belongs to the split.  Only the last stage of a contraction whose length is not a
// multiple of 32 takes the clamped path.  (The host sends operands whose offsets do not fit 31 bits to the 4-wave kernels.)
template <bool KC, int ROWS> struct WsDense6;

template <int ROWS>
struct WsDense6<true, ROWS> : DenseLoader6<true, ROWS> {                     // k-contiguous: piece i -> row (ptid >> 3) + 32 i, floats 4 (ptid & 7) .. + 3
    using Base = DenseLoader6<true, ROWS>;
    static constexpr int NPT = Base::NPT, NREG = Base::NREG;
    unsigned off[NPT]; unsigned rowmask;
    __device__ __forceinline__ void begin(const float* b, int64_t s_row_, int row0_, int rows_, int ptid) {
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
    __device__ __forceinline__ unsigned load6(float (&r)[NREG], int k0, int kend, int ptid) const {
        if (kend - k0 >= BKT) {
            const char* b = reinterpret_cast<const char*>(this->base + k0);
#pragma unroll
            for (int i = 0; i < NPT; ++i) {
                const float4 v = *reinterpret_cast<const float4*>(b + off[i]);
                r[4 * i] = v.x; r[4 * i + 1] = v.y; r[4 * i + 2] = v.z; r[4 * i + 3] = v.w;
            }
            return rowmask;
        }
        const int kk = k0 + ((ptid & 7) << 2);
        const bool kok = kk < kend;                                          // K % 4 == 0: a float4 is inside or outside as a whole
        const char* b = reinterpret_cast<const char*>(this->base + (kok ? kk : kend - 4) - ((ptid & 7) << 2));
#pragma unroll
        for (int i = 0; i < NPT; ++i) {
            const float4 v = *reinterpret_cast<const float4*>(b + off[i]);
            r[4 * i] = v.x; r[4 * i + 1] = v.y; r[4 * i + 2] = v.z; r[4 * i + 3] = v.w;
        }
        return kok ? rowmask : 0u;
    }
};

template <int ROWS>
struct WsDense6<false, ROWS> : DenseLoader6<false, ROWS> {                    // row-contiguous: rows 2 rp, 2 rp + 1 (one 8-byte load per k), KQ consecutive k
    using Base = DenseLoader6<false, ROWS>;
    static constexpr int KQ = Base::KQ, NREG = Base::NREG, RP = Base::RP;
    unsigned off; bool rok;
    __device__ __forceinline__ void begin(const float* b, int64_t s_k_, int row0_, int rows_, int ptid) {
        this->base = b; this->s_row = 1; this->s_k = s_k_; this->row0 = row0_; this->rows = rows_;
        const int row = row0_ + 2 * (ptid % RP);
        rok = row < rows_;                                                   // rows % 4 == 0 and row even: the pair is inside or outside together
        off = (unsigned)(((int64_t)(KQ * (ptid / RP)) * s_k_ + (rok ? row : rows_ - 2)) << 2);
    }
    __device__ __forceinline__ unsigned load6(float (&r)[NREG], int k0, int kend, int ptid) const {
        if (kend - k0 >= BKT) {
            const float* bk = this->base + (int64_t)k0 * this->s_k;          // wave-uniform; one scalar add per k below
#pragma unroll
            for (int j = 0; j < KQ; ++j) {
                const float2 v = *reinterpret_cast<const float2*>(reinterpret_cast<const char*>(bk + (int64_t)j * this->s_k) + off);
                r[2 * j] = v.x; r[2 * j + 1] = v.y;
            }
            return rok ? ((KQ == 16) ? 0xFFFFFFFFu : ((1u << NREG) - 1u)) : 0u;
        }
        const int kb = k0 + KQ * (ptid / RP);
        const char* b = reinterpret_cast<const char*>(this->base) + (off - (unsigned)(((int64_t)(KQ * (ptid / RP)) * this->s_k) << 2));   // row part only
        unsigned okmask = 0;
#pragma unroll
        for (int j = 0; j < KQ; ++j) {
            const int k = kb + j;
            const bool ok = rok && k < kend;
            const float2 v = *reinterpret_cast<const float2*>(b + (((int64_t)(k < kend ? k : kend - 1) * this->s_k) << 2));
            r[2 * j] = v.x; r[2 * j + 1] = v.y;
            okmask |= (ok ? 3u : 0u) << (2 * j);
        }
        return okmask;
    }
};

template <class Cfg, bool AKC, bool BKC>
struct DenseMk6 {
    using LA = WsDense6<AKC, Cfg::BM>; using LB = WsDense6<BKC, Cfg::BN>;
    __device__ __forceinline__ void make(const GemmArgs& g, const TileCoord& t, LA& la, LB& lb, int ptid) const {
        la.begin(g.A + t.z0 * g.a_b0 + t.z1 * g.a_b1, AKC ? g.a_m : g.a_k, t.m0, g.M, ptid);
        lb.begin(g.B + t.z0 * g.b_b0 + t.z1 * g.b_b1, BKC ? g.b_n : g.b_k, t.n0, g.N, ptid);
    }
};

// host-side cost model (same form as kTiles6; ONE workgroup per CU): constants from the r03 device sweep
static const TileInfo6 kTilesWs[] = {{SEGX_TILE_256x128, 256, 128, 1, 1.45f, 3.0f},
                                     {SEGX_TILE_WS128x128, 128, 128, 1, 0.80f, 2.0f}};

}  // namespace segx
