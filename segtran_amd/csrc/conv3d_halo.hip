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
    return all * 8 >= full * 5 && (all * ceil_div(Cout, 64) >= 2 * full || geom[0] <= 64) ? 1 : 0;
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
