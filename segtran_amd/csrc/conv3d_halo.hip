// conv3d_halo.hip -- the 3 x 3 x 3, stride-1, 'same' convolutions of Inception-I3D (aj_i3d.py:75-97 Unit3D, :198-273 the layer table) on the bf16x6 tile engine
// with an LDS-RESIDENT INPUT HALO (round 6).
//
// The im2col kernels of conv3d.hip gather every input voxel once per tap that touches it: 16 scattered dword loads and ~90 conversion instructions per thread and
// 32-k tile, 27 requests per voxel through L1 / L2 (engine traffic 2.5 - 3 x the algorithmic bytes, 40 - 45 % of wave cycles issue-stalled, r05_zb).  Here a
// workgroup owns a SPATIAL BLOCK of 128 outputs (4 x 4 x 8 or 8 x 4 x 4, depth x height x width) times a tile of 64 / 128 / 192 output channels and walks
// the contraction channel block by channel block (8 input channels):
//   * the block's halo -- (TD + 2)(TH + 2)(TW + 2) = 360 voxels x 8 channels -- is loaded ONCE per channel block, split into the three bf16 planes ONCE (the
//     split arithmetic is paid per voxel, not per tap) and stored channel-minor: one 16-byte slot per voxel and plane = the 8 k-values of one MFMA lane;
//   * the 27 taps are then 27 SHIFTED fragment reads of that image: lane (position p, k-half h) of a 32 x 32 x 16 step reads slot(p + tap_h) -- the two
//     k-halves of a step are two different taps (pairs chosen so that the slot distance inside a pair takes three values only) -- straight into the
//     matrix instruction; no im2col tile is ever formed;
//   * the filters come PRE-SPLIT from segx_conv3d_halo_pack (three bf16 planes in exactly the tile order and LDS row image of gemm_x6.h: [channel block]
//     [tap quad][plane][row][32 k]), so staging a 32-k weight tile is a 16-byte copy per chunk: no conversion arithmetic in the main loop at all.
// Per 32-k tile a thread issues 6 global loads + 6 LDS stores (weights) and 24 fragment reads beside the 48 matrix instructions of its wave; the halo costs
// 16 loads + ~90 conversions + 6 stores per SEVEN tiles.  backward-data = the same kernel on the flipped, transposed filter bank (pack mode 1).
//
// LDS bank rule (MI355X_MICROARCH.md, LDS: ds_read_b128 is served in the 16-lane groups {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, + 32): the 32 positions of
// a fragment block are dealt to lanes so that one group reads the x-rows with EVEN y and the other the rows with odd y; with a halo row stride of 12 slots
// (TW = 8: ten used) resp. 6 slots and a plane stride of 40 (TW = 4: 36 used) the sixteen 16-byte slots of a group are distinct mod 16 for every tap shift
// -- conflict-free reads without a swizzle (tools/halo_banks.py enumerates it).
#include "gemm_x6.h"

namespace segx {

typedef unsigned uvec4 __attribute__((ext_vector_type(4)));

// ---- geometry of one workgroup tile ------------------------------------------------------------------------------------------------------------
template <int MI_, int TD_, int TH_, int TW_>
struct HaloCfg {
    static constexpr int MI = MI_, NJ = 2, WM = 2, WN = 2, TD = TD_, TH = TH_, TW = TW_;
    static_assert(TD * TH * TW == 128 && (TW == 8 || TW == 4) && TH == 4, "halo tile: 128 outputs, rows of 8 or 4, four rows per plane");
    static constexpr int BM = WM * 32 * MI, BN = 128;
    static constexpr int HD = TD + 2, HH = TH + 2, HW = TW + 2;
    static constexpr int S = TW == 8 ? 12 : 6;                       // slots per halo row
    static constexpr int SD = TW == 8 ? HH * S : 40;                 // slots per halo plane (TW = 4: 36 used, 40 for the bank rule)
    static constexpr int NSLOT = HD * SD, NVOX = HD * HH * HW;
    static constexpr int PH = NSLOT * 16;                            // bytes of one bf16 plane of the halo
    static constexpr int PA = BM * X6_ROWB;                          // bytes of one plane of the weight tile ([row][32 k], gemm_x6.h image)
    static constexpr int LDS_BYTES = 3 * PH + 3 * PA;
    static constexpr int A_CHUNKS = BM * 4 / 256;                    // 16-byte chunks per thread, plane and weight tile
    static constexpr int H_ITEMS = (NVOX + 255) / 256;               // (voxel, 8 channels) items per thread and channel block
};

// The contraction order inside a channel block: 28 "taps" (27 + one phantom with zero weights) as 14 pairs; the two k-halves of an MFMA step take the two taps
// of a pair.  Pairs: (kd, kh, 0 | 1) x 9, (kd, 0 | 1, 2) x 3, (0 | 1, 2, 2), (2, 2, 2 | phantom).  Returns kd * 9 + kh * 3 + kw, or 27 for the phantom.
__host__ __device__ constexpr int halo_tap(int pair, int half) {
    if (pair < 9) return (pair / 3) * 9 + (pair % 3) * 3 + half;
    if (pair < 12) return (pair - 9) * 9 + half * 3 + 2;
    if (pair == 12) return half * 9 + 2 * 3 + 2;
    return half == 0 ? 26 : 27;
}
// slot offset of a tap inside the halo (the phantom re-reads tap 26's slot: its weights are zero)
template <class Cfg> __host__ __device__ constexpr int halo_tap_slot(int tap) {
    const int t = tap > 26 ? 26 : tap;
    return (t / 9) * Cfg::SD + ((t / 3) % 3) * Cfg::S + (t % 3);
}

struct HaloArgs {
    const float* X; const unsigned char* Wq; float* Y;
    int Cin, Cout, D, H, W;                  // stride 1, 'same': input extent == output extent
    int64_t x_bs, y_bs;                      // sample strides in floats
    int ntd, nth, ntw, nmt, ncb;             // tiles per axis, output-channel tiles, channel blocks of 8
};

// lane (fragment row r = lane & 31) -> (x-row q of the block, x): the even-y rows go to the first ds_read_b128 lane group, the odd-y rows to the second
template <int TW> __device__ __forceinline__ void halo_row_of(int r, int& q, int& x) {
    const int g = r >> 2;
    if (TW == 8) {                           // rows q = y (0..3) of one plane:  g: 0 -> (0, 0..3)  1 -> (1, 0..3)  2 -> (1, 4..7)  3 -> (0, 4..7)  4 -> (3, 0..3)  5 -> (2, 0..3)  6 -> (2, 4..7)  7 -> (3, 4..7)
        q = (0x32230110u >> (4 * g)) & 15;
        x = (r & 3) + (((0xCCu >> g) & 1) << 2);
    } else {                                 // rows q = 4 * plane-in-block + y:  g -> q = 0 1 3 2 5 4 6 7
        q = (0x76452310u >> (4 * g)) & 15;
        x = r & 3;
    }
}

template <class Cfg, int WPE>
__global__ __launch_bounds__(256) SEGX_MIN_WAVES_PER_SIMD(WPE) void conv3d_halo_fwd_x6_kernel(HaloArgs g) {
    constexpr int MI = Cfg::MI, NJ = Cfg::NJ, PA = Cfg::PA, PH = Cfg::PH, TW = Cfg::TW, TH = Cfg::TH, TD = Cfg::TD;
    __shared__ __attribute__((aligned(16))) unsigned char lds[Cfg::LDS_BYTES];
    unsigned char* const LH = lds;                                   // halo: [plane][slot][8 channels] bf16
    unsigned char* const LA_ = lds + 3 * PH;                         // weights: [plane][row][32 k] bf16, chunk-swizzled (x6_off)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1, kh = lane >> 5;
    // ---- which tile: consecutive logical blocks share an XCD (L2): output-channel tiles of one spatial block first, then neighbours along W, H, D
    unsigned l = xcd_block(blockIdx.x, gridDim.x);
    const int mt = (int)(l % (unsigned)g.nmt); l /= (unsigned)g.nmt;
    const int tw = (int)(l % (unsigned)g.ntw); l /= (unsigned)g.ntw;
    const int th = (int)(l % (unsigned)g.nth); l /= (unsigned)g.nth;
    const int td = (int)(l % (unsigned)g.ntd); const int b = (int)(l / (unsigned)g.ntd);
    const int d0 = td * TD, h0 = th * TH, w0 = tw * TW, m0 = mt * Cfg::BM;
    const int plane = g.H * g.W;
    const int64_t chan = (int64_t)g.D * plane;
    const float* const Xb = g.X + (int64_t)b * g.x_bs;
    // ---- halo items of this thread: voxel v = tid + 256 i -> (hz, hy, hx); global offset inside a channel (or invalid), LDS slot
    int hoff[Cfg::H_ITEMS], hslot[Cfg::H_ITEMS];
#pragma unroll
    for (int i = 0; i < Cfg::H_ITEMS; ++i) {
        const int v = tid + 256 * i;
        const int hz = v / (Cfg::HH * Cfg::HW), r2 = v - hz * (Cfg::HH * Cfg::HW), hy = r2 / Cfg::HW, hx = r2 - hy * Cfg::HW;
        const int zd = d0 - 1 + hz, zh = h0 - 1 + hy, zw = w0 - 1 + hx;
        const bool ok = v < Cfg::NVOX && (unsigned)zd < (unsigned)g.D && (unsigned)zh < (unsigned)g.H && (unsigned)zw < (unsigned)g.W;
        hoff[i] = ok ? (zd * g.H + zh) * g.W + zw : -1;
        hslot[i] = v < Cfg::NVOX ? (hz * Cfg::SD + hy * Cfg::S + hx) * 16 : -1;
    }
    // ---- fragment rows of this lane: weight rows (A) and the halo slot of its position in each of the NJ position blocks (B)
    const int arow = wm * (32 * MI) + (lane & 31);
    int bslot[NJ];                                                   // byte offset of (position + tap (0, 0, 0)) in a halo plane
    int q, x;
    halo_row_of<TW>(lane & 31, q, x);
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int blk = wn * NJ + j;                                 // 32 positions: TW = 8 -> plane blk, rows y = q;  TW = 4 -> planes 2 blk + (q >> 2), rows y = q & 3
        const int dz = TW == 8 ? blk : 2 * blk + (q >> 2), dy = TW == 8 ? q : (q & 3);
        bslot[j] = (dz * Cfg::SD + dy * Cfg::S + x) * 16;
    }
    // ---- weight tile: chunk f = tid + 256 i of a plane -> row f >> 2, 16-byte chunk f & 3
    const unsigned char* wsrc[Cfg::A_CHUNKS]; int wdst[Cfg::A_CHUNKS];
#pragma unroll
    for (int i = 0; i < Cfg::A_CHUNKS; ++i) {
        const int f = tid + 256 * i, row = f >> 2, c = f & 3;
        const int co = m0 + row < g.Cout ? m0 + row : g.Cout - 1;   // rows past Cout re-read the last filter: their results are not stored
        wsrc[i] = g.Wq + ((int64_t)co * 4 + c) * 16;
        wdst[i] = x6_off(row, c);
    }
    const int64_t wplane = (int64_t)g.Cout * X6_ROWB, wtile = 3 * wplane;       // bytes per plane / per (channel block, tap quad) of Wq

    f32x16 acc[MI][NJ];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    float hreg[Cfg::H_ITEMS][8];
    uvec4 wreg[3][Cfg::A_CHUNKS];                                   // (a native vector type: an array of HIP's uint4 structs stayed in scratch memory)
    auto load_halo = [&](int cb) {
        const float* const Xc = Xb + (int64_t)cb * 8 * chan;
#pragma unroll
        for (int i = 0; i < Cfg::H_ITEMS; ++i) {
            const float* p = Xc + (hoff[i] >= 0 ? hoff[i] : 0);
#pragma unroll
            for (int j = 0; j < 8; ++j) hreg[i][j] = p[(int64_t)j * chan];
        }
    };
    auto store_halo = [&]() {
#pragma unroll
        for (int i = 0; i < Cfg::H_ITEMS; ++i) {
            if (hslot[i] >= 0) {
                float v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = hoff[i] >= 0 ? hreg[i][j] : 0.f;
                x6_store8<PH>(LH, hslot[i], v);
            }
        }
    };
    auto load_w = [&](int tile) {                                    // tile = cb * 7 + tap quad
        const int64_t base = (int64_t)tile * wtile;
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int i = 0; i < Cfg::A_CHUNKS; ++i) wreg[p][i] = *reinterpret_cast<const uvec4*>(wsrc[i] + base + p * wplane);
    };
    auto store_w = [&]() {
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int i = 0; i < Cfg::A_CHUNKS; ++i) *reinterpret_cast<uvec4*>(LA_ + p * PA + wdst[i]) = wreg[p][i];
    };

    load_halo(0);
    load_w(0);
    for (int cb = 0; cb < g.ncb; ++cb) {
#pragma unroll
        for (int tq = 0; tq < 7; ++tq) {
            __syncthreads();                                         // every wave has read the previous weight tile (and, at tq == 0, the previous halo)
            if (tq == 0) store_halo();
            store_w();
            __syncthreads();
            // UNCONDITIONAL prefetches (the last iteration re-reads the last tile / channel block and drops it): a load under a run-time condition makes hipcc keep
            // the destination registers in scratch around the branch (seen in the ISA of the first version: 12 - 40 dwords spilled inside this loop)
            const int next = cb * 7 + tq + 1;
            load_w(next < g.ncb * 7 ? next : g.ncb * 7 - 1);
            if (tq == 0) load_halo(cb + 1 < g.ncb ? cb + 1 : cb);    // in flight (16 registers) under the seven tiles of this channel block
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const int pair = 2 * tq + s;
                const int chunk = 2 * s + kh;
                const int so0 = halo_tap_slot<Cfg>(halo_tap(pair, 0)) * 16, so1 = halo_tap_slot<Cfg>(halo_tap(pair, 1)) * 16;
                const int so = so0 + kh * (so1 - so0);               // this lane's tap of the pair
                bf16x8 a[MI][3];
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int p = 0; p < 3; ++p) a[i][p] = *reinterpret_cast<const bf16x8*>(LA_ + p * PA + x6_off(arow + 32 * i, chunk));
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    bf16x8 bb[3];
#pragma unroll
                    for (int p = 0; p < 3; ++p) bb[p] = *reinterpret_cast<const bf16x8*>(LH + p * PH + bslot[j] + so);
#pragma unroll
                    for (int i = 0; i < MI; ++i) {
                        f32x16 c = acc[i][j];
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], bb[2], c, 0, 0, 0);     // hi . lo
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][2], bb[0], c, 0, 0, 0);     // lo . hi
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], bb[1], c, 0, 0, 0);     // mid . mid
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], bb[1], c, 0, 0, 0);     // hi . mid
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], bb[0], c, 0, 0, 0);     // mid . hi
                        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], bb[0], c, 0, 0, 0);     // hi . hi
                        acc[i][j] = c;
                    }
                }
            }
        }
    }
    // ---- epilogue: lane = one position per block, 16 output channels per MFMA block (rows 8 (r >> 2) + 4 kh + (r & 3)); Y[b][co][d][h][w]
    float* const Yb = g.Y + (int64_t)b * g.y_bs;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int blk = wn * NJ + j;
        const int dz = TW == 8 ? blk : 2 * blk + (q >> 2), dy = TW == 8 ? q : (q & 3);
        const int od = d0 + dz, oh = h0 + dy, ow = w0 + x;
        if (od < g.D && oh < g.H && ow < g.W) {
            const int64_t pos = ((int64_t)od * g.H + oh) * g.W + ow;
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int co = m0 + wm * (32 * MI) + 32 * i + 8 * (r >> 2) + 4 * kh + (r & 3);
                    if (co < g.Cout) Yb[(int64_t)co * chan + pos] = acc[i][j][r];
                }
        }
    }
}

// Wq[cb][tq][plane][row][32 k] (bf16): k = 8 c + e  <->  tap halo_tap(2 tq + (c >> 1), c & 1), contracted channel 8 cb + e; the three planes of x = hi + mid + lo.
// mode 0: row = output channel o, value W[o][ci][t];  mode 1 (backward-data): row = INPUT channel of the layer, contracted = its output channels, value
// W[c][o][26 - t].  W is the layer's [Cout][Cin][27] tensor in both modes; rows = O, contracted channels = C (zero beyond C: a last half-empty block).
__global__ __launch_bounds__(256) void conv3d_halo_pack_kernel(const float* __restrict__ W, unsigned short* __restrict__ Wq, int O, int C, int mode) {
    const int ncb = (C + 7) / 8;
    const int64_t total = (int64_t)ncb * 7 * O * 16;                // one thread per (cb, tq, row, pair of k)
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        const int kp = (int)(idx & 15); int64_t r = idx >> 4;
        const int row = (int)(r % O); r /= O;
        const int tq = (int)(r % 7), cb = (int)(r / 7);
        float v[2];
#pragma unroll
        for (int e2 = 0; e2 < 2; ++e2) {
            const int k = 2 * kp + e2, c = k >> 3, e = k & 7;
            const int tap = halo_tap(2 * tq + (c >> 1), c & 1), ch = 8 * cb + e;
            float w = 0.f;
            if (tap < 27 && ch < C) w = mode == 0 ? W[((int64_t)row * C + ch) * 27 + tap] : W[((int64_t)ch * O + row) * 27 + (26 - tap)];
            v[e2] = w;
        }
        const Split2 s = split3_pair(v[0], v[1]);
        const int64_t tile = (int64_t)cb * 7 + tq;
        unsigned* out = reinterpret_cast<unsigned*>(Wq) + ((tile * 3) * O + row) * 16 + kp;         // dwords: 16 per row and plane
        out[0] = s.h; out[(int64_t)O * 16] = s.m; out[(int64_t)2 * O * 16] = s.l;
    }
}


// ---- backward-weight with a resident halo -----------------------------------------------------------------------------------------------------------
// dW[co][ci][tap] = sum over (sample, position p) dY[co][p] X[ci][p + tap - 1]: M = output channels, N = (tap, ci), K = positions.  A workgroup owns 32 NW output
// channels x (28 taps x 8 input channels = 224 columns: seven 32-column MFMA blocks, one of the 28 taps a zero phantom) and STREAMS over its share of the 128-output
// spatial blocks (4 x 4 x 8; K-split over workgroups, deterministic slab reduction afterwards), accumulating in registers (7 x 16 per lane).  Per spatial block
//   * the halo of its 8 input channels is staged ONCE as three x-SHIFTED WINDOWS per halo row (x - 1 .. x + 6, x .. x + 7, x + 1 .. x + 8: the eight k-values of a
//     lane are eight consecutive output positions along W, i.e. 16 contiguous bytes of bf16 -- a tap's kw shift would otherwise land a fragment on a 2-byte
//     boundary), split into the three planes once per voxel: [channel][kw][halo row (d, h)] slots of 16 B;
//   * dY arrives in tiles of OCT position octets (rows (d, h) of the block) x 32 NW channels in the gemm_x6.h row image;
//   * the fragment of (tap, channel) for octet (dz, dy) is ONE ds_read_b128 at slot  ci * CS + kw * KS + (dz + kd) * 6 + (dy + kh).
// The same im2col-free structure as the forward kernel; the im2col weight gradient gathered 16 scalars or two unaligned 16-byte rows per thread and k-tile and
// split every element once per TAP (27 x) -- 114 - 133 TFLOP/s on the large layers (r05_zb).
// Bank rule: with CS = 115 (odd) the eight channels of a tap cover eight slots that are distinct mod 8, and the taps are PAIRED so that the slot offsets of a pair
// differ by 8 mod 16 (kw * 39 + kd * 6 + kh; table below, found by tools/halo_banks.py --search): the 16 lanes of a ds_read_b128 group -- eight channels of
// each tap of a pair, dealt to the lane groups by halo_row_of<8> -- hit sixteen distinct 16-byte slots.
struct HaloW {
    static constexpr int RS = 6, KS = 39, CS = 115;
    static constexpr int PX = 8 * CS * 16;                           // bytes of one plane of the window image
    static constexpr int ROWS = 36, ITEMS = 8 * ROWS;                // (channel, halo row) staging items per spatial block
};
// column block j (0..6), tap-in-block i (0..3) -> tap kd * 9 + kh * 3 + kw (27 = phantom): lanes with i = 0 / 2 share one ds_read_b128 group, i = 1 / 3 the other
__host__ __device__ constexpr int halo_wtap(int j, int i) {
    constexpr int P[14][2] = {{0, 4}, {1, 5}, {2, 9}, {3, 7}, {6, 20}, {8, 15}, {10, 14}, {11, 18}, {12, 16}, {13, 17}, {19, 23}, {21, 25}, {22, 26}, {24, 27}};
    return P[2 * j + (i & 1)][i >> 1];
}
__host__ __device__ constexpr int halo_wtap_slot(int tap) {        // slot offset of a tap inside a channel's window image (phantom: tap 24's partner position)
    return tap > 26 ? 2 * HaloW::RS + 2 + 8 : (tap % 3) * HaloW::KS + (tap / 9) * HaloW::RS + (tap / 3) % 3;
}
struct HaloWArgs {
    const float* dY; const float* X; float* ws;
    int Cin, Cout, D, H, W, B;
    int64_t dy_bs, x_bs;
    int ntd, nth, ntw, nmt, ncb, nsplit, nsb;                        // spatial tiles per axis, output-channel tiles, channel blocks, K-splits, spatial blocks in all (B * ntd * nth * ntw)
};

// Waves: NS row slices of 32 output channels x 2 column halves (blocks 0-3 | 4-6): 2 NS waves, 128 NS threads.  Wave w = slice w % NS, half w / NS, so that waves w
// and w + 4 -- which the hardware places on the same SIMD -- are a four-block and a three-block wave wherever NS is a multiple of 4 (NS = 6: 11 vs 10 blocks per SIMD).
// The first version (one wave per slice, all seven blocks: 112 accumulator registers, two waves per SIMD) spent 27 % of its wave cycles parked at barriers / vmcnt and
// kept the matrix pipe 37 % busy (r06_pw1 counters; the forward kernel: 12 % and 72 %); twice the waves halve every thread's staging work and double the cover.
template <int NS, int OCT, int WPE, bool V4>
__global__ __launch_bounds__(128 * NS) SEGX_MIN_WAVES_PER_SIMD(WPE) void conv3d_halo_wgrad_x6_kernel(HaloWArgs g) {
    constexpr int T = 128 * NS, BM = 32 * NS, PA = BM * 16 * OCT, PX = HaloW::PX, NPIECE = (BM * 2 * OCT + T - 1) / T, XI = (HaloW::ITEMS + T - 1) / T, STEPS = OCT / 2;
    static_assert(BM * 2 * OCT % T == 0 || BM * 2 * OCT < T, "dY pieces divide over the threads");
    static_assert(OCT == 2 || OCT == 4, "dY tiles of two or four position octets");
    __shared__ __attribute__((aligned(16))) unsigned char lds[3 * PX + 3 * PA];
    unsigned char* const LX = lds;                                   // windows: [plane][channel][kw][halo row] x 16 B
    unsigned char* const LA_ = lds + 3 * PX;                         // dY tile: [plane][row][8 OCT k]: OCT = 4 the gemm_x6.h image; OCT = 2 rows of 32 B, chunk ^ bit 4 of the row
    auto aoff = [](int row, int chunk) { return OCT == 4 ? x6_off(row, chunk) : row * 32 + ((chunk ^ ((row >> 4) & 1)) << 4); };
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, kh = lane >> 5;
    const int slice = SEGX_WAVE_UNIFORM(wave % NS), half = SEGX_WAVE_UNIFORM(wave / NS);
    unsigned l = xcd_block(blockIdx.x, gridDim.x);                   // channel blocks of one (split, channel tile) are neighbours: they stream the same dY through one L2
    const int cb = (int)(l % (unsigned)g.ncb); l /= (unsigned)g.ncb;
    const int mt = (int)(l % (unsigned)g.nmt); const int split = (int)(l / (unsigned)g.nmt);
    const int m0 = mt * BM;
    const int sb_lo = (int)((int64_t)g.nsb * split / g.nsplit), sb_hi = (int)((int64_t)g.nsb * (split + 1) / g.nsplit);
    const int plane = g.H * g.W;
    const int64_t chan = (int64_t)g.D * plane;
    // ---- per-lane fragment addresses: A row; B: (tap-in-block, channel) of this lane in each of the seven column blocks
    const int arow = slice * 32 + (lane & 31);
    int ti, ci;
    halo_row_of<8>(lane & 31, ti, ci);
    const int lbase = (ci * HaloW::CS + kh) * 16;                    // + kh: the odd octet of a step is the next row (dy + 1)
    // slot offset of this lane's tap in column block j (recomputed where used: seven more live registers would not fit the 168 of three waves per SIMD at NW = 6)
    const unsigned tsh = 8u * (unsigned)ti;
    // slot offsets of the four taps of column block j, packed into one constant and picked by a per-lane shift (a ?: chain became branches in the loop); the block of
    // a wave is 4 half + jj: the constant is chosen on the scalar unit
    auto bpack = [](int j) {
        return (unsigned)halo_wtap_slot(halo_wtap(j, 0)) | ((unsigned)halo_wtap_slot(halo_wtap(j, 1)) << 8) | ((unsigned)halo_wtap_slot(halo_wtap(j, 2)) << 16) |
               ((unsigned)halo_wtap_slot(halo_wtap(j, 3)) << 24);
    };
    auto btap = [&](int jj) {
        const unsigned packed = half ? bpack(jj < 3 ? 4 + jj : 6) : bpack(jj);
        return (int)(((packed >> tsh) & 255u) << 4);
    };
    // ---- staging maps.  dY piece f = tid + T i -> row f / (2 OCT), quarter q = f % (2 OCT): octet-in-tile q >> 1, floats 4 (q & 1) .. + 3 of the octet.
    // Element offset of a piece = [clamped channel * chan + (q >> 1) * W + 4 (q & 1)] (ainv: per lane, loop-invariant) + a wave-uniform part per (block, tile);
    // rows past Cout re-read the last channel (their accumulators are never stored).
    int arow_s[NPIECE], aq[NPIECE], ainv[NPIECE];
#pragma unroll
    for (int i = 0; i < NPIECE; ++i) {
        const int f0 = tid + T * i, f = f0 < BM * 2 * OCT ? f0 : f0 - BM * 2 * OCT;      // (fewer pieces than threads: the surplus threads re-stage the first pieces with the same values)
        arow_s[i] = f / (2 * OCT); aq[i] = f % (2 * OCT);
        const int co = m0 + arow_s[i] < g.Cout ? m0 + arow_s[i] : g.Cout - 1;
        ainv[i] = co * (int)chan + (aq[i] >> 1) * g.W + 4 * (aq[i] & 1);
    }
    // X item f = tid + T i -> channel f / 36, halo row f % 36 = hz * 6 + hy.  Its offsets are RECOMPUTED from an opaque copy of the thread index where they are used
    // (once per spatial block): kept live across the matrix phase they were spilled, and the reload's s_waitcnt vmcnt(0) sat right behind the dY prefetch (r06_d ISA)
    auto xitem = [&](int i, int& c, int& hz, int& hy) {
        int t2 = tid;
        SEGX_PIN(t2);
        const int f = t2 + T * i;
        c = f / HaloW::ROWS; const int hr = f - c * HaloW::ROWS; hz = hr / 6; hy = hr - hz * 6;
        return f < HaloW::ITEMS;
    };
    f32x16 acc[4];                                                  // column blocks 4 half + jj (the second half uses three)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

    // (validity masks are RECOMPUTED at store time from the block coordinates, which still describe the data in flight: two fewer live registers per staged piece)
    float areg[NPIECE][4];
    float xreg[XI][10];
    int d0 = 0, h0 = 0, w0 = 0, bb = 0;                              // spatial block being LOADED (the loads run one tile ahead of the matrix work)
    auto set_block = [&](int sb) {
        int r = sb;
        const int tw = r % g.ntw; r /= g.ntw;
        const int th = r % g.nth; r /= g.nth;
        const int td = r % g.ntd; bb = r / g.ntd;
        d0 = td * 4; h0 = th * 4; w0 = tw * 8;
    };
    auto load_x = [&]() {                                            // the halo rows of block (bb, d0, h0, w0): ten floats x = w0 - 1 .. w0 + 8 each
        const float* const Xc = g.X + (int64_t)bb * g.x_bs + (int64_t)cb * 8 * chan;     // wave-uniform; everything per lane is a 32-bit element offset (< 2^31: host check)
        const int ubase = ((d0 - 1) * g.H + (h0 - 1)) * g.W + (w0 - 1);                   // wave-uniform (may be negative: added to xinv before use)
#pragma unroll
        for (int i = 0; i < XI; ++i) {
            int c, hz, hy;
            const bool has = xitem(i, c, hz, hy);
            const bool rok = has && (unsigned)(d0 - 1 + hz) < (unsigned)g.D && (unsigned)(h0 - 1 + hy) < (unsigned)g.H;
            const int roff = c * (int)chan + (hz * g.H + hy) * g.W + ubase;
#pragma unroll
            for (int e = 0; e < 10; ++e) {
                const bool ok = rok && (unsigned)(w0 - 1 + e) < (unsigned)g.W;
                xreg[i][e] = Xc[ok ? roff + e : 0];
            }
        }
    };
    auto store_x = [&]() {
#pragma unroll
        for (int i = 0; i < XI; ++i) {
            int c, hz, hy;
            const bool has = xitem(i, c, hz, hy);
            const int hr = hz * 6 + hy;
            if (has) {
                const bool rok = (unsigned)(d0 - 1 + hz) < (unsigned)g.D && (unsigned)(h0 - 1 + hy) < (unsigned)g.H;
                float v[10];
#pragma unroll
                for (int e = 0; e < 10; ++e) v[e] = (rok && (unsigned)(w0 - 1 + e) < (unsigned)g.W) ? xreg[i][e] : 0.f;
                Split2 s[5];
#pragma unroll
                for (int e = 0; e < 5; ++e) s[e] = split3_pair(v[2 * e], v[2 * e + 1]);
                unsigned char* const dst = LX + (c * HaloW::CS + hr) * 16;
#define SEGX_HALO_WIN(PL, FIELD)                                                                                                                         \
                {                                                                                                                                        \
                    const unsigned p0 = s[0].FIELD, p1 = s[1].FIELD, p2 = s[2].FIELD, p3 = s[3].FIELD, p4 = s[4].FIELD;                                  \
                    *reinterpret_cast<uvec4*>(dst + (PL) * PX) = uvec4{p0, p1, p2, p3};                                               /* x - 1 .. x + 6 */ \
                    *reinterpret_cast<uvec4*>(dst + (PL) * PX + HaloW::KS * 16) =                                                                        \
                        uvec4{(p0 >> 16) | (p1 << 16), (p1 >> 16) | (p2 << 16), (p2 >> 16) | (p3 << 16), (p3 >> 16) | (p4 << 16)};    /* x .. x + 7 */     \
                    *reinterpret_cast<uvec4*>(dst + (PL) * PX + 2 * HaloW::KS * 16) = uvec4{p1, p2, p3, p4};                          /* x + 1 .. x + 8 */ \
                }
                SEGX_HALO_WIN(0, h) SEGX_HALO_WIN(1, m) SEGX_HALO_WIN(2, l)
#undef SEGX_HALO_WIN
            }
        }
    };
    auto load_a = [&](int kt) {                                      // dY tile kt of block (bb, d0, h0, w0): octet kt * OCT + (q >> 1) = (dz, dy) with dz = (kt * OCT) / 4 (wave-uniform)
        const float* const Yc = g.dY + (int64_t)bb * g.dy_bs;
        const int zd = d0 + ((kt * OCT) >> 2), zh0 = h0 + ((kt * OCT) & 3);
        const int ubase = (zd * g.H + zh0) * g.W + w0;
        const bool dok = zd < g.D;
#pragma unroll
        for (int i = 0; i < NPIECE; ++i) {
            const bool rok = dok && zh0 + (aq[i] >> 1) < g.H;
            const int xq = w0 + 4 * (aq[i] & 1), roff = ainv[i] + ubase;
            if (V4) {                                                // W % 4 == 0 and 16-byte aligned samples: the four floats are inside the row together
                const bool ok = rok && xq < g.W;
                const f32x4 v = *reinterpret_cast<const f32x4*>(Yc + (ok ? roff : 0));
                areg[i][0] = v.x; areg[i][1] = v.y; areg[i][2] = v.z; areg[i][3] = v.w;
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const bool ok = rok && xq + e < g.W;
                    areg[i][e] = Yc[ok ? roff + e : 0];
                }
            }
        }
    };
    auto store_a = [&](int kt) {                                     // kt: the tile these registers were loaded for (the caller's current tile)
        const int zd = d0 + ((kt * OCT) >> 2), zh0 = h0 + ((kt * OCT) & 3);
        const bool dok = zd < g.D;
#pragma unroll
        for (int i = 0; i < NPIECE; ++i) {
            const bool rok = dok && zh0 + (aq[i] >> 1) < g.H;
            const int xq = w0 + 4 * (aq[i] & 1);
            x6_store4<PA>(LA_, aoff(arow_s[i], aq[i] >> 1) + ((aq[i] & 1) << 3), (rok && xq < g.W) ? areg[i][0] : 0.f, (rok && xq + 1 < g.W) ? areg[i][1] : 0.f,
                          (rok && xq + 2 < g.W) ? areg[i][2] : 0.f, (rok && xq + 3 < g.W) ? areg[i][3] : 0.f);
        }
    };

    constexpr int KT = 16 / OCT;                                     // dY tiles per spatial block
    if (sb_lo < sb_hi) {
        set_block(sb_lo);
        load_x();
        load_a(0);
    }
    for (int sb = sb_lo; sb < sb_hi; ++sb) {
        for (int kt = 0; kt < KT; ++kt) {
            __syncthreads();                                         // the previous tile's fragments (and at kt == 0 the previous block's windows) have been read
            if (kt == 0) store_x();
            store_a(kt);
            __syncthreads();
            // prefetch, UNCONDITIONALLY (the last iteration re-reads its own tile): the next dY tile; after the first tile of a block the next block's halo rows
            const bool last_kt = kt == KT - 1;
            if (last_kt) { set_block(sb + 1 < sb_hi ? sb + 1 : sb); load_x(); }
            load_a(last_kt ? 0 : kt + 1);
#pragma unroll
            for (int s = 0; s < STEPS; ++s) {
                // octets of this step: 2 s + kh of tile kt  ->  (dz, dy) = ((kt * OCT + 2 s) / 4, (kt * OCT + 2 s) % 4 + kh): the row offset of the even octet + kh (in bbase)
                const int oc = kt * OCT + 2 * s, rowoff = ((oc >> 2) * HaloW::RS + (oc & 3)) * 16;
                bf16x8 a[3];
#pragma unroll
                for (int p = 0; p < 3; ++p) a[p] = *reinterpret_cast<const bf16x8*>(LA_ + p * PA + aoff(arow, 2 * s + kh));
#define SEGX_HALO_WBLOCK(JJ, J)                                                                                                                 \
                {                                                                                                                               \
                    bf16x8 b[3];                                                                                                                \
                    _Pragma("unroll") for (int p = 0; p < 3; ++p) b[p] = *reinterpret_cast<const bf16x8*>(LX + p * PX + lbase + btap(J) + rowoff); \
                    f32x16 c = acc[JJ];                                                                                                         \
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[2], c, 0, 0, 0);                                                        \
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[0], c, 0, 0, 0);                                                        \
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1], c, 0, 0, 0);                                                        \
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], c, 0, 0, 0);                                                        \
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], c, 0, 0, 0);                                                        \
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], c, 0, 0, 0);                                                        \
                    acc[JJ] = c;                                                                                                                \
                }
                SEGX_HALO_WBLOCK(0, 0) SEGX_HALO_WBLOCK(1, 1) SEGX_HALO_WBLOCK(2, 2)
                if (half == 0) {                                     // the three-block half has no fourth block (the empty asm keeps hipcc from speculating the six MFMAs into both halves)
                    asm volatile("" ::: "memory");
                    SEGX_HALO_WBLOCK(3, 3)
                }
#undef SEGX_HALO_WBLOCK
            }
        }
    }
    // ---- this split's partial sums: ws[split][co][cb][224]; lane = column 32 j + (lane & 31), rows 8 (r >> 2) + 4 kh + (r & 3) of the wave's 32 channels
    float* const out = g.ws + (((int64_t)split * g.Cout) * g.ncb + cb) * 224;
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
        const int j = 4 * half + jj;
        if (j < 7) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = m0 + slice * 32 + 8 * (r >> 2) + 4 * kh + (r & 3);
                if (co < g.Cout) out[(int64_t)co * g.ncb * 224 + 32 * j + (lane & 31)] = acc[jj][r];
            }
        }
    }
}

// dW[co][ci][t] = sum over splits of ws[split][co][ci / 8][column of (t, ci % 8)] -- in split order (deterministic)
__global__ __launch_bounds__(256) void conv3d_halo_wgrad_reduce_kernel(const float* __restrict__ ws, float* __restrict__ dW, int Cout, int Cin, int nsplit) {
    const int ncb = Cin / 8;
    const int64_t total = (int64_t)Cout * ncb * 224;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        const int col = (int)(idx % 224); const int64_t rc = idx / 224;              // (co, cb)
        const int cb = (int)(rc % ncb), co = (int)(rc / ncb);
        const int j = col >> 5, r = col & 31, g4 = r >> 2;
        const int ti = (0x32230110u >> (4 * g4)) & 15, ci = (r & 3) + (((0xCCu >> g4) & 1) << 2);       // halo_row_of<8>
        int tap = 27;
#pragma unroll
        for (int jj = 0; jj < 7; ++jj)
#pragma unroll
            for (int ii = 0; ii < 4; ++ii) if (jj == j && ii == ti) tap = halo_wtap(jj, ii);
        if (tap > 26) continue;
        float sum = 0.f;
        for (int s = 0; s < nsplit; ++s) sum += ws[(int64_t)s * total + idx];
        dW[((int64_t)co * Cin + cb * 8 + ci) * 27 + tap] = sum;
    }
}

}  // namespace segx

using namespace segx;
#define SEGX_STREAM hipStream_t stream = (hipStream_t)stream_

static bool halo_geom_ok(const int* geom) {
    if (!geom) return false;
    const int Cin = geom[0], ID = geom[1], IH = geom[2], IW = geom[3];
    return Cin > 0 && Cin % 8 == 0 && geom[4] == ID && geom[5] == IH && geom[6] == IW && geom[7] == 3 && geom[8] == 3 && geom[9] == 3 &&
           geom[10] == 1 && geom[11] == 1 && geom[12] == 1 && geom[13] == 1 && geom[14] == 1 && geom[15] == 1;
}
// the tile a layer gets (8: 4 x 4 x 8 outputs, 4: 8 x 4 x 4): the one that pads the extent least (edge tiles are masked: outputs beyond the extent are computed and
// dropped), rows of 8 on a tie; *tiles = spatial tiles per sample
static int halo_tile(int D, int H, int W, int64_t* tiles) {
    const int64_t t8 = (int64_t)ceil_div(D, 4) * ceil_div(H, 4) * ceil_div(W, 8), t4 = (int64_t)ceil_div(D, 8) * ceil_div(H, 4) * ceil_div(W, 4);
    if (tiles) *tiles = t8 <= t4 ? t8 : t4;
    return t8 <= t4 ? 8 : 4;
}
// output channels per workgroup: the tile with the fewest padded rows per unit of measured efficiency (r06_a, tools/conv_bench.py: 128- and 192-row tiles run 7 - 8 %
// more rows per unit time than the 64-row tile, except a 128-row grid of fewer than two workgroups per CU)
static int halo_mtile(int Cout, int64_t tiles) {
    int best = 64; double bc = 1e30;
    const int mts[3] = {64, 128, 192};
    for (int i = 0; i < 3; ++i) {
        const int mt = mts[i], nmt = ceil_div(Cout, mt);
        const double eff = mt == 64 ? 1.0 : mt == 128 ? (tiles * nmt >= 512 ? 1.07 : 0.98) : 1.08;
        const double c = (double)nmt * mt / eff;
        if (c < bc * 0.999) { bc = c; best = mt; }
    }
    return best;
}
/* 1 when segx_conv3d_halo_fwd serves this convolution on the current default engine: 3 x 3 x 3, stride 1, pads 1 ('same'), Cin % 8 == 0, at most a third of the
 * 4 x 4 x 8 / 8 x 4 x 4 output tiles' positions beyond the extent, at least knob 17 spatial tiles over the batch (default 256: one per CU) -- the low-resolution
 * stages keep the split-K im2col kernels (r06_a: 16 x 8 x 8 at batch 4 = 128 tiles runs 0.5 - 0.9 x there, 32 x 16 x 16 = 256 tiles 1.2 - 1.7 x) */
extern "C" int segx_conv3d_halo_ok(int B, int Cout, const int* geom) {
    if (!halo_geom_ok(geom) || B <= 0 || Cout <= 0 || kget(knobs().engine) != SEGX_ENGINE_BF16X6 || kget(knobs().conv_halo) == 0) return 0;
    const int D = geom[1], H = geom[2], W = geom[3];
    if ((int64_t)geom[0] * D * H * W >= 2147483647LL) return 0;
    int64_t tiles = 0;
    halo_tile(D, H, W, &tiles);
    if (tiles * 128 > (int64_t)D * H * W * 3 / 2) return 0;                   // more than half of the tiles' outputs would be padding
    const int64_t all = tiles * B, full = kget(knobs().conv_halo_min_tiles);
    if (all >= full) return 1;
    // between 5/8 of a tile per CU and one (cfg4's 24 x 14 x 14 stage: 192 tiles at batch 4): only where 64-row workgroups still fill two rounds of the chip or the
    // contraction is too short for the im2col kernels' split-K to pay (r06_b, tools/conv_bench.py: 1.1 - 2 x there, 0.8 - 0.95 x on the long-K data gradients)
    return all * 8 >= full * 5 && ((all * ceil_div(Cout, 64) >= 2 * full && geom[0] <= 192) || geom[0] <= 64) ? 1 : 0;      // (Cin > 192: the 288 / 320 -> 144 / 160 data gradients, 0.9 x in the step, r06_i)
}
/* size of the pre-split filter bank in FLOATS (4-byte units of a float tensor the caller allocates): ceil(C / 8) channel blocks x 7 tap quads x 3 planes x O rows x 64 B */
extern "C" int64_t segx_conv3d_halo_wq_floats(int O, int C) { return O > 0 && C > 0 ? (int64_t)((C + 7) / 8) * 7 * 3 * O * 16 : 0; }
extern "C" int segx_conv3d_halo_pack(const float* W, void* Wq, int O, int C, int mode, void* stream_) {
    SEGX_STREAM; SEGX_REQUIRE(W && Wq && O > 0 && C > 0 && (mode == 0 || mode == 1) && (reinterpret_cast<uintptr_t>(Wq) & 15) == 0, "segx_conv3d_halo_pack: bad args");
    const int64_t total = (int64_t)((C + 7) / 8) * 7 * O * 16;
    hipLaunchKernelGGL(conv3d_halo_pack_kernel, dim3((unsigned)i64min(4096, (total + 255) / 256)), dim3(256), 0, stream, W, (unsigned short*)Wq, O, C, mode);
    return check_launch("segx_conv3d_halo_pack");
}
/* Y[b][Cout][D][H][W] = conv3d(X[b][Cin][D][H][W], filters) for the geometries segx_conv3d_halo_ok accepts; Wq from segx_conv3d_halo_pack (O = Cout, C = Cin);
 * x_bs / y_bs: sample strides in floats (0 = dense).  mtile: 0 = chosen here, else 64 / 128 / 192 output channels per workgroup (measurements) */
extern "C" int segx_conv3d_halo_fwd(const float* X, const void* Wq, float* Y, int B, int Cout, const int* geom, int64_t x_bs, int64_t y_bs, int mtile, void* stream_) {
    SEGX_STREAM; SEGX_REQUIRE(X && Wq && Y && B > 0 && Cout > 0 && halo_geom_ok(geom), "segx_conv3d_halo_fwd: bad args (3 x 3 x 3, stride 1, 'same', Cin %% 8 == 0 only)");
    const int Cin = geom[0], D = geom[1], H = geom[2], W = geom[3];
    int64_t tiles = 0;
    const int twd = halo_tile(D, H, W, &tiles);
    SEGX_REQUIRE((int64_t)Cin * D * H * W < 2147483647LL && (int64_t)Cout * D * H * W < (1LL << 40), "segx_conv3d_halo_fwd: sample too large");
    SEGX_REQUIRE((reinterpret_cast<uintptr_t>(Wq) & 15) == 0, "segx_conv3d_halo_fwd: unaligned filter bank");
    if (mtile == 0) mtile = halo_mtile(Cout, tiles * B);
    SEGX_REQUIRE(mtile == 64 || mtile == 128 || mtile == 192, "segx_conv3d_halo_fwd: mtile %d", mtile);
    HaloArgs g; g.X = X; g.Wq = (const unsigned char*)Wq; g.Y = Y; g.Cin = Cin; g.Cout = Cout; g.D = D; g.H = H; g.W = W;
    g.x_bs = x_bs ? x_bs : (int64_t)Cin * D * H * W; g.y_bs = y_bs ? y_bs : (int64_t)Cout * D * H * W;
    g.ntd = ceil_div(D, twd == 8 ? 4 : 8); g.nth = ceil_div(H, 4); g.ntw = ceil_div(W, twd); g.nmt = ceil_div(Cout, mtile); g.ncb = Cin / 8;
    const int64_t wgs = (int64_t)B * g.ntd * g.nth * g.ntw * g.nmt;
    SEGX_REQUIRE(wgs < 2147483647LL, "segx_conv3d_halo_fwd: grid too large");
    knobs().x6_launches.fetch_add(1, std::memory_order_relaxed);
    const dim3 grid((unsigned)wgs);
#define SEGX_HALO_LAUNCH(MI, WPE) do { \
        if (twd == 8) hipLaunchKernelGGL((conv3d_halo_fwd_x6_kernel<HaloCfg<MI, 4, 4, 8>, WPE>), grid, dim3(256), 0, stream, g); \
        else hipLaunchKernelGGL((conv3d_halo_fwd_x6_kernel<HaloCfg<MI, 8, 4, 4>, WPE>), grid, dim3(256), 0, stream, g); } while (0)
    if (mtile == 64) SEGX_HALO_LAUNCH(1, 4);
    else if (mtile == 128) SEGX_HALO_LAUNCH(2, 3);
    else SEGX_HALO_LAUNCH(3, 2);
#undef SEGX_HALO_LAUNCH
    return check_launch("segx_conv3d_halo_fwd");
}

/* r06 -- weight gradient of the same convolutions with a resident halo: dW [Cout][Cin][27] (summed over the batch) from dY [B][Cout][D][H][W] and X [B][Cin][D][H][W];
 * ws: segx_conv3d_halo_wgrad_ws_floats(B, Cout, geom) floats of scratch (K-split slabs, reduced in split order); dy_bs / x_bs: sample strides in floats (0 = dense) */
static int halo_wgrad_plan(int B, int Cout, int Cin, int D, int H, int W, int* nw, int* nmt, int* nsplit, int* nsb) {
    const int64_t tiles = (int64_t)ceil_div(D, 4) * ceil_div(H, 4) * ceil_div(W, 8) * B;
    if (tiles <= 0 || tiles >= 2147483647LL) return -1;
    const int r128 = ceil_div(Cout, 128) * 128, r192 = ceil_div(Cout, 192) * 192;
    *nw = r192 < r128 || (r192 == r128 && Cout > 128) ? 6 : 4;
    *nmt = ceil_div(Cout, 32 * *nw);                                   // nw = row slices of 32 channels per workgroup (4: 512 threads, 6: 768)
    const int units = *nmt * (Cin / 8);                               // workgroups per K-split
    // resident workgroups: two per CU for the 512-thread form, one for the 768-thread form.  The grid must not spill into a part-filled extra round (r06_e: 516
    // workgroups on 512 slots ran 4 of them alone, at twice the time): 512-thread form one round, 768-thread form two full rounds
    int sp = 512 / units;
    if (sp > tiles) sp = (int)tiles;
    if (sp < 1) sp = 1;
    *nsplit = sp; *nsb = (int)tiles;
    return 0;
}
extern "C" int64_t segx_conv3d_halo_wgrad_ws_floats(int B, int Cout, const int* geom) {
    if (!halo_geom_ok(geom) || B <= 0 || Cout <= 0) return 0;
    int nw, nmt, nsplit, nsb;
    if (halo_wgrad_plan(B, Cout, geom[0], geom[1], geom[2], geom[3], &nw, &nmt, &nsplit, &nsb)) return 0;
    return (int64_t)nsplit * Cout * (geom[0] / 8) * 224;
}
/* 1 when segx_conv3d_halo_wgrad serves the layer (the conditions of segx_conv3d_halo_ok with rows of 8 outputs: W padded to a multiple of 8 by at most half) */
extern "C" int segx_conv3d_halo_wgrad_ok(int B, int Cout, const int* geom) {
    if (!halo_geom_ok(geom) || B <= 0 || Cout <= 0 || kget(knobs().engine) != SEGX_ENGINE_BF16X6 || kget(knobs().conv_halo) == 0) return 0;
    const int D = geom[1], H = geom[2], W = geom[3];
    if ((int64_t)geom[0] * D * H * W >= 2147483647LL || (int64_t)Cout * D * H * W >= 2147483647LL || W % 4 != 0) return 0;      // rows of whole float4 (16-byte dY loads)
    const int64_t tiles = (int64_t)ceil_div(D, 4) * ceil_div(H, 4) * ceil_div(W, 8);
    if (tiles * 128 > (int64_t)D * H * W * 3 / 2) return 0;
    // r06_f (tools/conv_bench.py wgrad): 1.15 - 1.5 x the im2col form on every layer with Cin * Cout >= 3072 at >= 256 tiles; the 16 -> 32 ... 32 -> 64 channel
    // convolutions of the second Inception branch (one or two workgroups per K-split, hundreds of slabs) run 0.45 - 0.65 x and keep the im2col form
    if ((int64_t)geom[0] * Cout < (kget(knobs().conv_halo_min_tiles) <= 1 ? 1 : 3072)) return 0;
    return tiles * B >= kget(knobs().conv_halo_min_tiles) ? 1 : 0;
}
extern "C" int segx_conv3d_halo_wgrad(const float* dY, const float* X, float* dW, float* ws, int B, int Cout, const int* geom, int64_t dy_bs, int64_t x_bs, void* stream_) {
    SEGX_STREAM; SEGX_REQUIRE(dY && X && dW && ws && B > 0 && Cout > 0 && halo_geom_ok(geom), "segx_conv3d_halo_wgrad: bad args (3 x 3 x 3, stride 1, 'same', Cin %% 8 == 0 only)");
    const int Cin = geom[0], D = geom[1], H = geom[2], W = geom[3];
    SEGX_REQUIRE((int64_t)Cin * D * H * W < 2147483647LL && (int64_t)Cout * D * H * W < 2147483647LL, "segx_conv3d_halo_wgrad: sample too large");
    HaloWArgs g; g.dY = dY; g.X = X; g.ws = ws; g.Cin = Cin; g.Cout = Cout; g.D = D; g.H = H; g.W = W; g.B = B;
    g.dy_bs = dy_bs ? dy_bs : (int64_t)Cout * D * H * W; g.x_bs = x_bs ? x_bs : (int64_t)Cin * D * H * W;
    g.ntd = ceil_div(D, 4); g.nth = ceil_div(H, 4); g.ntw = ceil_div(W, 8); g.ncb = Cin / 8;
    int nw;
    SEGX_REQUIRE(halo_wgrad_plan(B, Cout, Cin, D, H, W, &nw, &g.nmt, &g.nsplit, &g.nsb) == 0, "segx_conv3d_halo_wgrad: grid too large");
    const int64_t wgs = (int64_t)g.nsplit * g.nmt * g.ncb;
    SEGX_REQUIRE(wgs < 2147483647LL, "segx_conv3d_halo_wgrad: grid too large");
    knobs().x6_launches.fetch_add(1, std::memory_order_relaxed);
    const bool v4 = W % 4 == 0 && g.dy_bs % 4 == 0 && (reinterpret_cast<uintptr_t>(dY) & 15) == 0;
    SEGX_REQUIRE(v4, "segx_conv3d_halo_wgrad: W %% 4 == 0 and 16-byte aligned dY samples only (segx_conv3d_halo_wgrad_ok)");
    if (nw == 6) hipLaunchKernelGGL((conv3d_halo_wgrad_x6_kernel<6, 4, 3, true>), dim3((unsigned)wgs), dim3(768), 0, stream, g);
    else hipLaunchKernelGGL((conv3d_halo_wgrad_x6_kernel<4, 4, 4, true>), dim3((unsigned)wgs), dim3(512), 0, stream, g);
    int rc = check_launch("segx_conv3d_halo_wgrad");
    if (rc) return rc;
    const int64_t total = (int64_t)Cout * g.ncb * 224;
    hipLaunchKernelGGL(conv3d_halo_wgrad_reduce_kernel, dim3((unsigned)i64min(2048, (total + 255) / 256)), dim3(256), 0, stream, (const float*)ws, dW, Cout, Cin, g.nsplit);
    return check_launch("segx_conv3d_halo_wgrad/reduce");
}
