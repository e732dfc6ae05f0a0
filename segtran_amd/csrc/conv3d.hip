// conv3d.hip -- Inception-I3D spatial convolutions (K20, aj_i3d.py:57-62,92) as IMPLICIT GEMM on the fp32 MFMA tile
// engine of gemm_core.h, plus the TF-'same' zero-padded max-pool (aj_i3d.py:6-30).
//
//   forward      Y[b][co][p]      = sum_{k=(ci,kd,kh,kw)} W[co][k] * X[b][ci][pos(p) + tap(k)]         M=Cout, N=P,       K=Cin*KV
//   backward-data (stride 1)      = the same kernel on (dY, W flipped & transposed), pad' = K-1-pad
//   backward-weight dW[co][n=(ci,tap)] = sum_{b,p} dY[b][co][p] * X[b][ci][pos(p) + tap]                M=Cout, N=Cin*KV, K=P (split-K, per-sample slabs)
//
// The im2col matrix is never materialised: the B-operand loader gathers straight from the NCDHW activation with
// zero padding (dynamic 'same' padding N7: front = pad // 2) — unconditional loads from clamped addresses plus a
// validity mask applied at LDS-store time, exactly like the dense loader.  The A operand (weights, or dY) is dense.
#include "gemm_x6.h"

namespace segx {


struct ConvGeom {
    int Cin, ID, IH, IW, OD, OH, OW, KD, KH, KW, sd, sh, sw, pd, ph, pw;
};

// ---- forward / backward-data: B(n = output position, k = (ci, tap)) ------------------------------------------------
// Thread map: ONE output position per thread (n0 + (tid & 127): the 64 lanes of a wave read 64 consecutive positions,
// i.e. whole 128-B lines for every tap), BKT/2 k-rows (tid>>7)*(BKT/2) + i.  The (ci, tap) decode of a k-row is wave-uniform.
// Per-thread state is ONE element offset of the window origin and the window's validity as bit masks (the whole window when
// it has <= 32 taps, else one mask per axis), so a gathered element costs ~5 VALU ops (offset add, bit extract, select, mask
// insert) instead of three coordinate adds + three range compares + the offset arithmetic.
// PACK8: the contraction index runs (channel block of 8, tap, channel in block) instead of (channel, tap) -- the weights are packed to
// match (segx_conv3d_pack_weights).  Eight consecutive k then share ONE tap, so a thread's 16 gathers need two tap decodes, two mask
// tests and a constant channel stride instead of sixteen of each (the decode-free loader experiment ran the big 3x3x3 convolutions
// at 98 instead of 60 TFLOP/s), while the 27 taps of a channel block stay within 14 k-tiles of each other (L1/L2 reuse).
template <bool PACK8>
struct ConvFwdLoaderB {
    const float* X; ConvGeom q; FastDiv dKV, dKHW, dKW;
    int pos_off;                    // ((bd * IH) + bh) * IW + bw of the window origin (may be negative: padding)
    unsigned mk0, mk1, mk2;         // KV <= 32: mk0 bit t = tap t inside the input;  else per-axis masks (bits kd / kh / kw)
    bool single;
    __device__ __forceinline__ ConvFwdLoaderB(const float* X_, const ConvGeom& q_, int n0, int P) : X(X_), q(q_) {
        dKV = make_fastdiv(q.KD * q.KH * q.KW); dKHW = make_fastdiv(q.KH * q.KW); dKW = make_fastdiv(q.KW);
        const int n = n0 + (threadIdx.x & 127);
        const bool nvalid = n < P;
        const int nn = nvalid ? n : 0;
        const int od = nn / (q.OH * q.OW), r = nn - od * q.OH * q.OW, oh = r / q.OW, ow = r - oh * q.OW;
        const int bd = od * q.sd - q.pd, bh = oh * q.sh - q.ph, bw = ow * q.sw - q.pw;
        pos_off = (bd * q.IH + bh) * q.IW + bw;
        unsigned md = 0, mh = 0, mw = 0;
        for (int i = 0; i < q.KD; ++i) md |= ((unsigned)(bd + i) < (unsigned)q.ID ? 1u : 0u) << i;
        for (int i = 0; i < q.KH; ++i) mh |= ((unsigned)(bh + i) < (unsigned)q.IH ? 1u : 0u) << i;
        for (int i = 0; i < q.KW; ++i) mw |= ((unsigned)(bw + i) < (unsigned)q.IW ? 1u : 0u) << i;
        single = q.KD * q.KH * q.KW <= 32;
        if (single) {
            unsigned m = 0; int t = 0;
            for (int a = 0; a < q.KD; ++a) for (int b2 = 0; b2 < q.KH; ++b2) for (int c = 0; c < q.KW; ++c, ++t)
                m |= (((md >> a) & (mh >> b2) & (mw >> c)) & 1u) << t;
            md = m;
        }
        if (!nvalid) md = 0;
        mk0 = md; mk1 = mh; mk2 = mw;
    }
    __device__ __forceinline__ unsigned load(float4 (&r)[NP], int k0, int kend, int tid) const {
        unsigned okmask = 0;
        const int KV = q.KD * q.KH * q.KW, KHW = q.KH * q.KW, plane = q.IH * q.IW;
        const int kbase = k0 + (tid >> 7) * (BKT / 2);               // wave-uniform
        float* v = reinterpret_cast<float*>(&r[0]);
        if (PACK8) {
            // r03: the (channel block, tap) decode runs on the SCALAR unit (kbase is wave-uniform: readfirstlane), and a gather is
            // `global_load_dword v, v_off, s[base]` -- SGPR base = X + (8 cb + j) channels, ONE per-thread byte offset (window origin + tap, or 0
            // where the tap falls into the padding) for the eight loads of a (block, tap) pair.  The 64-bit per-gather address chain and the
            // per-lane decode were ~75 of the ~380 vector instructions a thread issued per k-tile against 48 matrix instructions (VALU-bound).
            const int chan = q.ID * plane;                           // channel stride
            const int kb0 = SEGX_WAVE_UNIFORM(kbase);
#pragma unroll
            for (int sub = 0; sub < BKT / 16; ++sub) {
                const int kb = kb0 + 8 * sub;                        // multiple of 8: one (channel block, tap) pair
                const bool kok = kb < kend;
                const int blk = (kok ? kb : 0) >> 3;
                const int cb = fdiv(blk, dKV), t = blk - cb * KV, kd = fdiv(t, dKHW), t2 = t - kd * KHW, kh = fdiv(t2, dKW), kw = t2 - kh * q.KW;
                const unsigned bit = single ? (mk0 >> t) & 1u : ((mk0 >> kd) & (mk1 >> kh) & (mk2 >> kw)) & 1u;
                const bool ok = kok && bit != 0u;
                const unsigned voff = ok ? (unsigned)(pos_off + (kd * q.IH + kh) * q.IW + kw) << 2 : 0u;
                const ws_gptr b = ws_uniform_base(X + (int64_t)(cb * 8) * chan);
#pragma unroll
                for (int j = 0; j < 8; ++j) v[8 * sub + j] = ws_load<float>(b + (int64_t)j * chan * 4, voff);
                okmask |= (ok ? 0xFFu : 0u) << (8 * sub);
            }
            return okmask;
        }
#pragma unroll
        for (int i = 0; i < BKT / 2; ++i) {
            const int k = kbase + i;
            const bool kok = k < kend;
            const int kk = kok ? k : 0;
            // wave-uniform decode of (ci, kd, kh, kw) and of the tap's element offset
            const int ci = fdiv(kk, dKV), t = kk - ci * KV, kd = fdiv(t, dKHW), t2 = t - kd * KHW, kh = fdiv(t2, dKW), kw = t2 - kh * q.KW;
            const int tap_off = (ci * q.ID + kd) * plane + kh * q.IW + kw;
            const unsigned bit = single ? (mk0 >> t) & 1u : ((mk0 >> kd) & (mk1 >> kh) & (mk2 >> kw)) & 1u;
            const bool ok = kok && bit != 0u;
            v[i] = X[ok ? (int64_t)pos_off + tap_off : 0];
            okmask |= (ok ? 1u : 0u) << i;
        }
        return okmask;
    }
    static constexpr bool ROWK = true;              // LDS tile [position][k] (swizzled chunks): this thread's 16 k-values of one position are 4 float4 stores
    __device__ __forceinline__ void store(float4 (&r)[NP], unsigned okmask, float* T, int tid) const {
        const int n = tid & 127, kr = (tid >> 7) * (BKT / 2);
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const unsigned mk = okmask >> (4 * i);
            *reinterpret_cast<float4*>(T + rowk_off(n, (kr >> 2) + i)) =
                make_float4((mk & 1u) ? r[i].x : 0.f, (mk & 2u) ? r[i].y : 0.f, (mk & 4u) ? r[i].z : 0.f, (mk & 8u) ? r[i].w : 0.f);
        }
    }
};

// ---- backward-weight: B(n = (ci, tap), k = output position) ------------------------------------------------------
// Thread map: ONE output position per thread and k-tile (k0 + (tid & (BKT-1)): a wave reads BKT consecutive positions of
// one row = whole 128-B lines), 128*BKT/256 rows (tid / BKT) + (256/BKT)*i whose (channel offset, tap) sit in an LDS table.
// PACK8: the rows run (channel block of 8, tap, channel in block); a thread then owns TWO groups of eight consecutive rows that share a
// tap (rows 8 g + j and 64 + 8 g + j, g = tid / BKT), so the tap lookup and the three range tests happen twice per k-tile instead
// of sixteen times, the gathers of a group differ by the channel stride only, and the LDS copy is four float4 stores.  The result is
// written in the packed row order (segx_conv3d_unpack_wgrad restores [Cout][Cin][KV]).
template <bool PACK8>
struct ConvWgradLoaderB {
    const float* X; ConvGeom q; FastDiv dOHW, dOW;
    const int* rowinfo;                                        // LDS: per row (plain) or per 8-row group (packed): {channel offset (or -1), kd | kh<<10 | kw<<20}
    // fills the table; the caller must __syncthreads() before the first load
    __device__ __forceinline__ ConvWgradLoaderB(const float* X_, const ConvGeom& q_, int n0, int N, int* rowinfo_lds) : X(X_), q(q_), rowinfo(rowinfo_lds) {
        dOHW = make_fastdiv(q.OH * q.OW); dOW = make_fastdiv(q.OW);
        const int KV = q.KD * q.KH * q.KW, KHW = q.KH * q.KW, chan = q.ID * q.IH * q.IW;
        if (PACK8) {
            if (threadIdx.x < 16) {
                const int n = n0 + 8 * threadIdx.x;                                       // first row of the group; N % 8 == 0
                const int blk = (n < N ? n : 0) >> 3, cb = blk / KV, t = blk - cb * KV, kd = t / KHW, t2 = t - kd * KHW, kh = t2 / q.KW, kw = t2 - kh * q.KW;
                rowinfo_lds[2 * threadIdx.x] = n < N ? cb * 8 * chan : -1;
                rowinfo_lds[2 * threadIdx.x + 1] = kd | (kh << 10) | (kw << 20);
            }
        } else if (threadIdx.x < 128) {
            const int n = n0 + threadIdx.x;
            const int nn = n < N ? n : 0;
            const int ci = nn / KV, t = nn - ci * KV, kd = t / KHW, t2 = t - kd * KHW, kh = t2 / q.KW, kw = t2 - kh * q.KW;
            rowinfo_lds[2 * threadIdx.x] = n < N ? ci * chan : -1;                        // < 2^31: checked on the host
            rowinfo_lds[2 * threadIdx.x + 1] = kd | (kh << 10) | (kw << 20);
        }
    }
    __device__ __forceinline__ unsigned load(float4 (&r)[NP], int k0, int kend, int tid) const {
        unsigned okmask = 0;
        const int p = k0 + (tid & (BKT - 1));
        const bool pok = p < kend;
        const int pp = pok ? p : 0;
        const int od = fdiv(pp, dOHW), rr = pp - od * q.OH * q.OW, oh = fdiv(rr, dOW), ow = rr - oh * q.OW;
        const int bd = od * q.sd - q.pd, bh = oh * q.sh - q.ph, bw = ow * q.sw - q.pw;
        float* v = reinterpret_cast<float*>(&r[0]);
        if (PACK8) {
            const int64_t chan = (int64_t)q.ID * q.IH * q.IW;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int grp = tid / BKT + 8 * h;
                const int cb = rowinfo[2 * grp], tp = rowinfo[2 * grp + 1];
                const int id = bd + (tp & 1023), ih = bh + ((tp >> 10) & 1023), iw = bw + (tp >> 20);
                const bool ok = pok && cb >= 0 && (unsigned)id < (unsigned)q.ID && (unsigned)ih < (unsigned)q.IH && (unsigned)iw < (unsigned)q.IW;
                const float* src = X + (ok ? (int64_t)cb + ((int64_t)id * q.IH + ih) * q.IW + iw : 0);
#pragma unroll
                for (int j = 0; j < 8; ++j) v[8 * h + j] = src[j * chan];
                okmask |= (ok ? 0xFFu : 0u) << (8 * h);
            }
            return okmask;
        }
#pragma unroll
        for (int i = 0; i < 4 * NP; ++i) {
            const int row = tid / BKT + (256 / BKT) * i;
            const int cb = rowinfo[2 * row], tp = rowinfo[2 * row + 1];
            const int id = bd + (tp & 1023), ih = bh + ((tp >> 10) & 1023), iw = bw + (tp >> 20);
            const bool ok = pok && cb >= 0 && (unsigned)id < (unsigned)q.ID && (unsigned)ih < (unsigned)q.IH && (unsigned)iw < (unsigned)q.IW;
            const int64_t off = ok ? (int64_t)cb + ((int64_t)id * q.IH + ih) * q.IW + iw : 0;
            v[i] = X[off];
            if (ok) okmask |= 1u << i;
        }
        return okmask;
    }
    static constexpr bool ROWK = false;             // LDS tile [k][row]: one position (k) per thread, 16 rows
    __device__ __forceinline__ void store(float4 (&r)[NP], unsigned okmask, float* T, int tid) const {
        const int k = tid & (BKT - 1), r0 = tid / BKT;
        if (PACK8) {
#pragma unroll
            for (int i = 0; i < NP; ++i) {                        // pieces 0,1 = rows 8 r0 .. +7; pieces 2,3 = rows 64 + 8 r0 .. +7
                const bool ok = (okmask >> (4 * i)) & 1u;
                *reinterpret_cast<float4*>(T + k * (BN + 4) + 64 * (i >> 1) + 8 * r0 + 4 * (i & 1)) = ok ? r[i] : make_float4(0.f, 0.f, 0.f, 0.f);
            }
            return;
        }
        const float* v = reinterpret_cast<const float*>(&r[0]);
#pragma unroll
        for (int i = 0; i < 4 * NP; ++i) T[k * (BN + 4) + r0 + (256 / BKT) * i] = ((okmask >> i) & 1u) ? v[i] : 0.f;
    }
};

// ---- the same B operands for the bf16x6 engine (gemm_x6.h) -------------------------------------------------------------------------------
// forward / backward-data: the PACK8 gather already hands a thread two groups of EIGHT consecutive k of ONE position -- exactly one 16-byte
// bf16 chunk per plane each: the global side is ConvFwdLoaderB<true>::load unchanged, the LDS side is two split-and-store calls.
struct ConvFwdLoaderB6 {
    static constexpr int NREG = 4 * NP;
    ConvFwdLoaderB<true> inner;
    __device__ __forceinline__ ConvFwdLoaderB6(const float* X_, const ConvGeom& q_, int n0, int P) : inner(X_, q_, n0, P) {}
    __device__ __forceinline__ unsigned load6(float (&r)[NREG], int k0, int kend, int tid) const {
        float4 t4[NP];
        const unsigned ok = inner.load(t4, k0, kend, tid);
#pragma unroll
        for (int i = 0; i < NP; ++i) { r[4 * i] = t4[i].x; r[4 * i + 1] = t4[i].y; r[4 * i + 2] = t4[i].z; r[4 * i + 3] = t4[i].w; }
        return ok;
    }
    __device__ __forceinline__ void store6(float (&r)[NREG], unsigned okmask, unsigned char* __restrict__ P, int tid) const {
        const int n = tid & 127, c0 = (tid >> 7) * 2;
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
            float v[8];
            const bool ok = ((okmask >> (8 * sub)) & 1u) != 0u;        // the eight k of a (channel block, tap) pair are inside or outside together
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = ok ? r[8 * sub + j] : 0.f;
            x6_store8<X6Plane<128>::bytes>(P, x6_off(n, c0 + sub), v);
        }
    }
};

// backward-weight: B(n = packed row (channel block, tap, channel in block), k = output position).  The fp32 loader gives a thread ONE position
// and 16 rows; the bf16 fragments need eight consecutive k (positions) of one row, so here a thread owns the position octet k0 + 8 (tid & 3) ..
// + 7 and the TWO rows (tid >> 2) and 64 + (tid >> 2): the eight positions are decoded once (incrementally: ow, carry into oh, od) and serve
// both rows; 16 lanes x 4 octets of a wave read 16 channels x 32 consecutive positions (128-byte runs wherever the octets stay in one image row).
struct __attribute__((aligned(4))) F4u { float x, y, z, w; };       // 16 bytes at dword alignment (global_load_dwordx4 needs no more)
template <int FASTW>                                          // 0: per-position decode; 1: rows of eight / two quads (OW % 8 == 0, OW % 4 == 0); 2: contiguous octet (stride-1 'same', r05)
struct ConvWgradLoaderB6 {
    static constexpr int NREG = 16;
    const float* X; ConvGeom q; FastDiv dOHW, dOW, dSW;
    const int* rowinfo;                                        // LDS, per 8-row group: {channel-block offset (or -1), kd | kh<<10 | kw<<20}
    __device__ __forceinline__ ConvWgradLoaderB6(const float* X_, const ConvGeom& q_, int n0, int N, int* rowinfo_lds) : X(X_), q(q_), rowinfo(rowinfo_lds) {
        dOHW = make_fastdiv(q.OH * q.OW); dOW = make_fastdiv(q.OW); dSW = make_fastdiv(q.sw);
        const int KV = q.KD * q.KH * q.KW, KHW = q.KH * q.KW, chan = q.ID * q.IH * q.IW;
        if (threadIdx.x < 16) {
            const int n = n0 + 8 * threadIdx.x;                                       // first row of the group; N % 8 == 0
            const int blk = (n < N ? n : 0) >> 3, cb = blk / KV, t = blk - cb * KV, kd = t / KHW, t2 = t - kd * KHW, kh = t2 / q.KW, kw = t2 - kh * q.KW;
            rowinfo_lds[2 * threadIdx.x] = n < N ? cb * 8 * chan : -1;
            rowinfo_lds[2 * threadIdx.x + 1] = kd | (kh << 10) | (kw << 20);
        }
    }
    __device__ __forceinline__ unsigned load6(float (&r)[NREG], int k0, int kend, int tid) const {
        const int p0 = k0 + 8 * (tid & 3), row = tid >> 2;
        const int chan = q.ID * q.IH * q.IW, OHW = q.OH * q.OW;
        const int pp = p0 < kend ? p0 : 0;
        int od = fdiv(pp, dOHW), rr = pp - od * OHW, oh = fdiv(rr, dOW), ow = rr - oh * q.OW;
        int cb[2], kd[2], kh[2], kw[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int grp = (row >> 3) + 8 * h, tp = rowinfo[2 * grp + 1];
            const int c = rowinfo[2 * grp];
            cb[h] = c < 0 ? -1 : c + (row & 7) * chan;
            kd[h] = tp & 1023; kh[h] = (tp >> 10) & 1023; kw[h] = tp >> 20;
        }
        unsigned okmask = 0;
        if (FASTW == 2) {
            {
                // r05 -- stride 1 with 'same' geometry (I == O in every axis: the 3 x 3 x 3 Inception convolutions), any row length >= 7 (the 14- and 7-wide stages of
                // cfg4, which used to stay on the fp32 engine): output position p reads input element p + tap offset, so the octet's eight gathers are eight
                // CONSECUTIVE floats even where the octet wraps onto the next output row -- two dword-aligned 16-byte loads -- and only the validity differs: one
                // interval of valid columns per output row the octet touches (at most two rows: OW >= 7), each with its own (id, ih) test.
                const int npos = kend - p0;
                const int len1 = q.OW - ow < 8 ? q.OW - ow : 8;                  // positions of the octet on its first output row
                int od2 = od, oh2 = oh + 1;
                if (oh2 == q.OH) { oh2 = 0; ++od2; }
                const unsigned live = npos >= 8 ? 0xFFu : npos > 0 ? (1u << npos) - 1u : 0u;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int dk = kd[h] - q.pd, hk = kh[h] - q.ph, wk = kw[h] - q.pw;
                    const bool r1 = (unsigned)(od + dk) < (unsigned)q.ID && (unsigned)(oh + hk) < (unsigned)q.IH;
                    const bool r2 = (unsigned)(od2 + dk) < (unsigned)q.ID && (unsigned)(oh2 + hk) < (unsigned)q.IH;
                    int lo1 = -(ow + wk), hi1 = q.IW - (ow + wk);                // valid j on the first row: 0 <= ow + j + wk < IW
                    lo1 = lo1 < 0 ? 0 : lo1; hi1 = hi1 > len1 ? len1 : hi1;
                    int lo2 = len1 - wk, hi2 = len1 + q.IW - wk;                  // on the second row: 0 <= (j - len1) + wk < IW
                    lo2 = lo2 < len1 ? len1 : lo2; hi2 = hi2 > 8 ? 8 : hi2;
                    unsigned m = 0;
                    if (r1 && hi1 > lo1) m |= ((1u << hi1) - 1u) & ~((1u << lo1) - 1u);
                    if (r2 && hi2 > lo2) m |= ((1u << hi2) - 1u) & ~((1u << lo2) - 1u);
                    m &= live;
                    if (cb[h] < 0) m = 0;
                    const int64_t off = m ? (int64_t)cb[h] + pp + ((int64_t)dk * q.IH + hk) * q.IW + wk : 0;
                    if (off >= 0 && off + 8 <= (int64_t)q.Cin * chan) {
                        const F4u u0 = *reinterpret_cast<const F4u*>(X + off), u1 = *reinterpret_cast<const F4u*>(X + off + 4);
                        r[8 * h] = u0.x; r[8 * h + 1] = u0.y; r[8 * h + 2] = u0.z; r[8 * h + 3] = u0.w;
                        r[8 * h + 4] = u1.x; r[8 * h + 5] = u1.y; r[8 * h + 6] = u1.z; r[8 * h + 7] = u1.w;
                    } else {                                                      // the first / last floats of the whole sample: only the valid elements are touched
                        const int first = m ? __builtin_ctz(m) : 0;
#pragma unroll
                        for (int j = 0; j < 8; ++j) r[8 * h + j] = X[off + (((m >> j) & 1u) ? j : first)];
                    }
                    okmask |= m << (8 * h);
                }
                return okmask;
            }
        }
        if (FASTW == 1) {
            if (q.OW % 8 != 0) {
                // OW % 4 == 0 only (the 28-wide stages): the octet is TWO quads of four consecutive ow, the second possibly on the next output row --
                // each quad is one 16-byte load per (channel, tap) row with its own (id, ih) test and interval of valid j (unit stride: host check)
                const int npos = kend - p0;
                int odq[2] = {od, od}, ohq[2] = {oh, oh}, owq[2] = {ow, ow + 4};
                if (owq[1] >= q.OW) { owq[1] -= q.OW; if (++ohq[1] == q.OH) { ohq[1] = 0; ++odq[1]; } }
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int qd = 0; qd < 2; ++qd) {
                        const int id = odq[qd] * q.sd - q.pd + kd[h], ih = ohq[qd] * q.sh - q.ph + kh[h], iw0 = owq[qd] - q.pw + kw[h];
                        const int left = npos - 4 * qd;
                        const bool rok = left > 0 && cb[h] >= 0 && (unsigned)id < (unsigned)q.ID && (unsigned)ih < (unsigned)q.IH;
                        int lo = iw0 < 0 ? -iw0 : 0, hi = q.IW - iw0;
                        hi = hi > 4 ? 4 : hi; hi = hi > left ? left : hi;
                        if (!rok || hi < lo || lo > 4) { lo = 0; hi = 0; }
                        const unsigned m = ((1u << hi) - 1u) & ~((1u << lo) - 1u);
                        const int64_t off = m ? (int64_t)cb[h] + ((int64_t)id * q.IH + ih) * q.IW + iw0 : 0;
                        if (off >= 0 && off + 4 <= (int64_t)q.Cin * chan) {
                            const F4u u = *reinterpret_cast<const F4u*>(X + off);
                            r[8 * h + 4 * qd] = u.x; r[8 * h + 4 * qd + 1] = u.y; r[8 * h + 4 * qd + 2] = u.z; r[8 * h + 4 * qd + 3] = u.w;
                        } else {
#pragma unroll
                            for (int j = 0; j < 4; ++j) r[8 * h + 4 * qd + j] = X[off + (((m >> j) & 1u) ? j : (m ? lo : 0))];
                        }
                        okmask |= m << (8 * h + 4 * qd);
                    }
                return okmask;
            }
            // OW % 8 == 0, unit stride along W, octets aligned to 8: the eight positions are eight consecutive ow of ONE output row, so the eight
            // gathers of a row are eight consecutive floats of one input row -- one address, one (id, ih) test and an interval of valid j per row
            const int npos = kend - p0;                                // < 8 only in the last k-tile (P % 8 != 0 cannot happen: OW % 8 == 0)
            const int bd = od * q.sd - q.pd, bh = oh * q.sh - q.ph, bw = ow * q.sw - q.pw;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int id = bd + kd[h], ih = bh + kh[h], iw0 = bw + kw[h];                 // position j reads iw0 + j * sw
                const bool rok = npos > 0 && cb[h] >= 0 && (unsigned)id < (unsigned)q.ID && (unsigned)ih < (unsigned)q.IH;
                int lo = iw0 < 0 ? fdiv(-iw0 + q.sw - 1, dSW) : 0;
                int hi = iw0 >= q.IW ? 0 : fdiv(q.IW - 1 - iw0, dSW) + 1;                    // first j with iw0 + j * sw >= IW
                hi = hi > 8 ? 8 : hi; hi = hi > npos ? npos : hi;
                if (!rok || hi < lo) { lo = 0; hi = 0; }
                const unsigned m = ((1u << hi) - 1u) & ~((1u << lo) - 1u);
                // Unit stride: the row's eight gathers as TWO dword-aligned 16-byte loads starting AT iw0; where the window sticks out of the row the extra
                // floats are the neighbouring rows' (inside the sample: read, then masked), so only the first / last floats of the whole sample need the
                // scalar form.  Eight scalar gathers per row made this loader load-unit-bound: 64 scattered dwords per instruction, 16 instructions per
                // k-tile (64-row tile: 67 TFLOP/s against 117 for the 128-row tile with the same loader; now 119 / 153).  Stride 2 keeps the gathers:
                // reading 16 floats to use 8 doubles the cache traffic of the 343-tap stem (measured: no gain, 66 TFLOP/s either way).
                const int64_t off = m ? (int64_t)cb[h] + ((int64_t)id * q.IH + ih) * q.IW + iw0 : 0;
                if (q.sw == 1 && off >= 0 && off + 8 <= (int64_t)q.Cin * chan) {
                    const float* rowp = X + off;
                    const F4u u0 = *reinterpret_cast<const F4u*>(rowp), u1 = *reinterpret_cast<const F4u*>(rowp + 4);
                    r[8 * h] = u0.x; r[8 * h + 1] = u0.y; r[8 * h + 2] = u0.z; r[8 * h + 3] = u0.w;
                    r[8 * h + 4] = u1.x; r[8 * h + 5] = u1.y; r[8 * h + 6] = u1.z; r[8 * h + 7] = u1.w;
                } else {
                    const float* src = X + off;
#pragma unroll
                    for (int j = 0; j < 8; ++j) r[8 * h + j] = src[(((m >> j) & 1u) ? j : (m ? lo : 0)) * q.sw];
                }
                okmask |= m << (8 * h);
            }
            return okmask;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const bool pok = p0 + j < kend;
            const int bd = od * q.sd - q.pd, bh = oh * q.sh - q.ph, bw = ow * q.sw - q.pw;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int id = bd + kd[h], ih = bh + kh[h], iw = bw + kw[h];
                const bool ok = pok && cb[h] >= 0 && (unsigned)id < (unsigned)q.ID && (unsigned)ih < (unsigned)q.IH && (unsigned)iw < (unsigned)q.IW;
                r[8 * h + j] = X[ok ? (int64_t)cb[h] + ((int64_t)id * q.IH + ih) * q.IW + iw : 0];
                okmask |= (ok ? 1u : 0u) << (8 * h + j);
            }
            if (++ow == q.OW) { ow = 0; if (++oh == q.OH) { oh = 0; ++od; } }
        }
        return okmask;
    }
    __device__ __forceinline__ void store6(float (&r)[NREG], unsigned okmask, unsigned char* __restrict__ P, int tid) const {
        const int c = tid & 3, row = tid >> 2;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = ((okmask >> (8 * h + j)) & 1u) ? r[8 * h + j] : 0.f;
            x6_store8<X6Plane<128>::bytes>(P, x6_off(row + 64 * h, c), v);
        }
    }
};

// Cfg: 128 x 128, or 64 x 128 when Cout <= 64 (the I3D stem and the 64-channel branches: half of a 128-row A tile would be
// clamped duplicates).  The B-side (im2col) loaders above are written for 128 columns.
using CfgCout64 = TileCfg<2, 2, 1, 2>;
template <bool VEC, class Cfg, bool PACK8>
__global__ __launch_bounds__(256, 2) void conv3d_fwd_kernel(GemmArgs g, ConvGeom q) {
    static_assert(Cfg::BN == 128, "conv loaders fill 128 columns");
    __shared__ __attribute__((aligned(16))) TileLdsT<Cfg> lds;
    const TileCoord t = tile_coord<Cfg>(g);
    const DenseLoader<true, VEC, Cfg::BM> la{g.A, g.a_m, 1, t.m0, g.M};             // weights [Cout][Cin*KV]
    const ConvFwdLoaderB<PACK8> lb(g.B + (int64_t)t.zb * g.b_b0, q, t.n0, g.N);      // X[b]
    f32x16 acc[Cfg::MI][Cfg::NJ];
    gemm_mainloop<Cfg>(acc, la, lb, t.kbeg, t.kend, lds);
    gemm_epilogue<SEGX_EPI_NONE, Cfg>(acc, g, t);
}
template <bool VEC, class Cfg, bool PACK8>
__global__ __launch_bounds__(256, 2) void conv3d_wgrad_kernel(GemmArgs g, ConvGeom q) {
    static_assert(Cfg::BN == 128, "conv loaders fill 128 columns");
    __shared__ __attribute__((aligned(16))) TileLdsT<Cfg> lds;
    const TileCoord t = tile_coord<Cfg>(g);
    const DenseLoader<true, VEC, Cfg::BM> la{g.A + (int64_t)t.zb * g.a_b0, g.a_m, 1, t.m0, g.M};   // dY[b] [Cout][P]
    __shared__ int rowinfo[256];
    const ConvWgradLoaderB<PACK8> lb(g.B + (int64_t)t.zb * g.b_b0, q, t.n0, g.N, rowinfo);   // X[b]
    __syncthreads();
    f32x16 acc[Cfg::MI][Cfg::NJ];
    gemm_mainloop<Cfg>(acc, la, lb, t.kbeg, t.kend, lds);
    gemm_epilogue<SEGX_EPI_NONE, Cfg>(acc, g, t);
}

// the packed convolutions on the bf16x6 engine (float4-legal weights / dY: the host checks)
template <class Cfg, int WPE>
__global__ __launch_bounds__(256) SEGX_MIN_WAVES_PER_SIMD(WPE) void conv3d_fwd_x6_kernel(GemmArgs g, ConvGeom q) {
    static_assert(Cfg::BN == 128, "conv loaders fill 128 columns");
    __shared__ __attribute__((aligned(16))) unsigned char lds[X6Lds<Cfg>::BYTES];
    const TileCoord t = tile_coord<Cfg>(g);
    const DenseLoader6<true, Cfg::BM> la{g.A, g.a_m, 1, t.m0, g.M};                  // packed weights [Cout][Cin*KV]
    const ConvFwdLoaderB6 lb(g.B + (int64_t)t.zb * g.b_b0, q, t.n0, g.N);            // X[b]
    f32x16 acc[Cfg::MI][Cfg::NJ];
    gemm_mainloop_x6<Cfg>(acc, la, lb, t.kbeg, t.kend, lds);
    gemm_epilogue<SEGX_EPI_NONE, Cfg>(acc, g, t);
}
template <class Cfg, int WPE, int FASTW>
__global__ __launch_bounds__(256) SEGX_MIN_WAVES_PER_SIMD(WPE) void conv3d_wgrad_x6_kernel(GemmArgs g, ConvGeom q) {
    static_assert(Cfg::BN == 128, "conv loaders fill 128 columns");
    __shared__ __attribute__((aligned(16))) unsigned char lds[X6Lds<Cfg>::BYTES + 128];
    const TileCoord t = tile_coord<Cfg>(g);
    const DenseLoader6<true, Cfg::BM> la{g.A + (int64_t)t.zb * g.a_b0, g.a_m, 1, t.m0, g.M};       // dY[b] [Cout][P]
    const ConvWgradLoaderB6<FASTW> lb(g.B + (int64_t)t.zb * g.b_b0, q, t.n0, g.N, reinterpret_cast<int*>(lds + X6Lds<Cfg>::BYTES));   // X[b]
    __syncthreads();
    f32x16 acc[Cfg::MI][Cfg::NJ];
    gemm_mainloop_x6<Cfg>(acc, la, lb, t.kbeg, t.kend, lds);
    gemm_epilogue<SEGX_EPI_NONE, Cfg>(acc, g, t);
}

// Wt[ci][co][t] = W[co][ci][KV-1-t]: the transposed, spatially flipped filter bank of the backward-data convolution
__global__ __launch_bounds__(256) void flip_weights_kernel(const float* __restrict__ W, float* __restrict__ Wt, int Cout, int Cin, int KV) {
    const int64_t total = (int64_t)Cout * Cin * KV;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int t = (int)(i % KV); const int64_t r = i / KV; const int co = (int)(r % Cout), ci = (int)(r / Cout);
        Wt[i] = W[((int64_t)co * Cin + ci) * KV + (KV - 1 - t)];
    }
}

// Weights in the PACK8 contraction order: Wp[o][cb][t][cj] (c = 8 cb + cj).  mode 0 (forward): o = co, c = ci, value W[co][ci][t];
// mode 1 (backward-data): o = ci, c = co, value W[co][ci][KV-1-t] (transposed, spatially flipped).  C = channels contracted over.
__global__ __launch_bounds__(256) void pack_weights_kernel(const float* __restrict__ W, float* __restrict__ Wp, int O, int C, int KV, int mode) {
    const int64_t total = (int64_t)O * C * KV;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int cj = (int)(i & 7); int64_t r = i >> 3; const int t = (int)(r % KV); r /= KV; const int cb = (int)(r % (C / 8)), o = (int)(r / (C / 8));
        const int c = cb * 8 + cj;
        Wp[i] = mode == 0 ? W[((int64_t)o * C + c) * KV + t] : W[((int64_t)c * O + o) * KV + (KV - 1 - t)];
    }
}

// Register-tiled variant: a thread owns FOUR consecutive positions of its class along W.  With at most 4 taps per axis (K <= 4 s) the
// 4 positions x 4 w-taps read a window of 7 dY values per (kd, kh, co) instead of 16 -- the scalar kernel above is bound by exactly
// those L1 loads (4096 per position).
template <int CIN>
__global__ __launch_bounds__(256) void conv3d_bwd_data_direct4_kernel(const float* __restrict__ dY, const float* __restrict__ Wt, float* __restrict__ dX,
                                                                      int B, int Cout, ConvGeom q) {
    const int nph = q.sd * q.sh * q.sw, b = blockIdx.y / nph, ph = blockIdx.y - b * nph;
    const int rd = ph / (q.sh * q.sw), rh = (ph / q.sw) % q.sh, rw = ph % q.sw;
    const int td0 = rd >= q.pd ? 0 : (q.pd - rd + q.sd - 1) / q.sd, td1 = (q.ID - 1 + q.pd - rd) >= 0 ? (q.ID - 1 + q.pd - rd) / q.sd : -1;
    const int th0 = rh >= q.ph ? 0 : (q.ph - rh + q.sh - 1) / q.sh, th1 = (q.IH - 1 + q.ph - rh) >= 0 ? (q.IH - 1 + q.ph - rh) / q.sh : -1;
    const int tw0 = rw >= q.pw ? 0 : (q.pw - rw + q.sw - 1) / q.sw, tw1 = (q.IW - 1 + q.pw - rw) >= 0 ? (q.IW - 1 + q.pw - rw) / q.sw : -1;
    const int nD = td1 - td0 + 1, nH = th1 - th0 + 1, nW = tw1 - tw0 + 1;
    if (nD <= 0 || nH <= 0 || nW <= 0) return;
    const int nW4 = (nW + 3) >> 2;
    const int l = blockIdx.x * 256 + threadIdx.x;
    if (l >= nD * nH * nW4) return;
    const int td = td0 + l / (nH * nW4), r2 = l % (nH * nW4), th = th0 + r2 / nW4, twb = tw0 + 4 * (r2 % nW4);
    const int nkw = (q.KW - rw + q.sw - 1) / q.sw;                               // w-taps of this class: kw = rw + sw * m, m < nkw <= 4
    const int64_t isz = (int64_t)q.ID * q.IH * q.IW, osz = (int64_t)q.OD * q.OH * q.OW;
    const float* gb = dY + (int64_t)b * Cout * osz;
    float acc[4][CIN];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int c = 0; c < CIN; ++c) acc[j][c] = 0.f;
    // window columns ow = twb - 3 + i (i = 0..6): position j with w-tap m reads ow = twb + j - m = column i = j - m + 3
    int colo[7]; bool colok[7];
#pragma unroll
    for (int i = 0; i < 7; ++i) { const int ow = twb - 3 + i; colok[i] = ow >= 0 && ow < q.OW; colo[i] = colok[i] ? ow : 0; }
    for (int kd = rd, od = td; kd < q.KD; kd += q.sd, --od) {
        if (od < 0 || od >= q.OD) continue;
        for (int kh = rh, oh = th; kh < q.KH; kh += q.sh, --oh) {
            if (oh < 0 || oh >= q.OH) continue;
            const float* grow = gb + ((int64_t)od * q.OH + oh) * q.OW;
            const float* wrow = Wt + (int64_t)((kd * q.KH + kh) * q.KW + rw) * Cout * CIN;      // tap (kd, kh, kw = rw); next w-tap: + sw * Cout * CIN
            const int64_t wstep = (int64_t)q.sw * Cout * CIN;
            for (int co = 0; co < Cout; ++co) {
                const float* g = grow + (int64_t)co * osz;
                float win[7];
#pragma unroll
                for (int i = 0; i < 7; ++i) { const float v = g[colo[i]]; win[i] = colok[i] ? v : 0.f; }
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    if (m >= nkw) break;
                    const float* w = wrow + m * wstep + co * CIN;                              // uniform
#pragma unroll
                    for (int c = 0; c < CIN; ++c) {
                        const float wv = w[c];
#pragma unroll
                        for (int j = 0; j < 4; ++j) acc[j][c] += wv * win[j - m + 3];
                    }
                }
            }
        }
    }
    const int id = rd - q.pd + q.sd * td, ih = rh - q.ph + q.sh * th;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int iw = rw - q.pw + q.sw * (twb + j);
        if (twb + j > tw1) continue;
        const int64_t pos = ((int64_t)id * q.IH + ih) * q.IW + iw;
#pragma unroll
        for (int c = 0; c < CIN; ++c) dX[((int64_t)b * CIN + c) * isz + pos] = acc[j][c];
    }
}

// dW[co][ci][t] = dWp[co][ci/8][t][ci%8]: the packed-row weight gradient back in the layer's layout
__global__ __launch_bounds__(256) void unpack_wgrad_kernel(const float* __restrict__ dWp, float* __restrict__ dW, int Cout, int Cin, int KV) {
    const int64_t total = (int64_t)Cout * Cin * KV;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int t = (int)(i % KV); int64_t r = i / KV; const int ci = (int)(r % Cin), co = (int)(r / Cin);
        dW[i] = dWp[(((int64_t)co * (Cin / 8) + (ci >> 3)) * KV + t) * 8 + (ci & 7)];
    }
}

// Backward-data for STRIDED convolutions (only the 7x7x7 stride-2 stem, 3 input channels): a direct gather on the
// vector ALU -- dX[b][ci][i] = sum_{co} sum_{taps t with (i + p - t) % s == 0} W[co][ci][t] * dY[b][co][(i + p - t) / s].
// M = Cin = 3 would leave 97 % of an MFMA tile empty, so this one stays off the matrix cores.
// One thread per input POSITION accumulating all CIN channels (each dY value is loaded once for the CIN outputs).
// Backward-data of a STRIDED convolution onto a few input channels (the 7x7x7 stride-2 I3D stem, 3 channels), by residue class:
// a workgroup serves input positions whose (i + pad) has ONE residue (rd, rh, rw) modulo the strides, so the set of contributing
// taps kd = rd + sd*m ... and every weight are UNIFORM across the workgroup (scalar loads), and neighbouring lanes read
// neighbouring dY elements (coalesced).  The earlier position-major kernel re-read the filter bank per lane with two different
// tap sets per wave: 11.9 ms and 22 GB fetched for a 58 MB result (r01-g PMC).
// grid: (position blocks of the class, B * sd*sh*sw)
// filters for the residue-class kernel: Wt[tap][co][ci] = W[co][ci][tap], so a workgroup's uniform (tap, co) walk reads consecutive
// floats (wide scalar loads)
__global__ __launch_bounds__(256) void tapmajor_weights_kernel(const float* __restrict__ W, float* __restrict__ Wt, int Cout, int Cin, int KV) {
    const int64_t total = (int64_t)Cout * Cin * KV;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int c = (int)(i % Cin); const int64_t r = i / Cin; const int co = (int)(r % Cout), t = (int)(r / Cout);
        Wt[i] = W[((int64_t)co * Cin + c) * KV + t];
    }
}
template <int CIN>
__global__ __launch_bounds__(256) void conv3d_bwd_data_direct_kernel(const float* __restrict__ dY, const float* __restrict__ Wt, float* __restrict__ dX,
                                                                     int B, int Cout, ConvGeom q) {
    const int nph = q.sd * q.sh * q.sw, b = blockIdx.y / nph, ph = blockIdx.y - b * nph;
    const int rd = ph / (q.sh * q.sw), rh = (ph / q.sw) % q.sh, rw = ph % q.sw;
    // positions of this class along an axis: i = r - pad + s*t for t in [t0, t1]
    const int td0 = rd >= q.pd ? 0 : (q.pd - rd + q.sd - 1) / q.sd, td1 = (q.ID - 1 + q.pd - rd) >= 0 ? (q.ID - 1 + q.pd - rd) / q.sd : -1;
    const int th0 = rh >= q.ph ? 0 : (q.ph - rh + q.sh - 1) / q.sh, th1 = (q.IH - 1 + q.ph - rh) >= 0 ? (q.IH - 1 + q.ph - rh) / q.sh : -1;
    const int tw0 = rw >= q.pw ? 0 : (q.pw - rw + q.sw - 1) / q.sw, tw1 = (q.IW - 1 + q.pw - rw) >= 0 ? (q.IW - 1 + q.pw - rw) / q.sw : -1;
    const int nD = td1 - td0 + 1, nH = th1 - th0 + 1, nW = tw1 - tw0 + 1;
    if (nD <= 0 || nH <= 0 || nW <= 0) return;
    const int l = blockIdx.x * 256 + threadIdx.x;
    if (l >= nD * nH * nW) return;
    const int td = td0 + l / (nH * nW), r2 = l % (nH * nW), th = th0 + r2 / nW, tw = tw0 + r2 % nW;
    const int id = rd - q.pd + q.sd * td, ih = rh - q.ph + q.sh * th, iw = rw - q.pw + q.sw * tw;
    const int64_t isz = (int64_t)q.ID * q.IH * q.IW, osz = (int64_t)q.OD * q.OH * q.OW;
    const float* gb = dY + (int64_t)b * Cout * osz;
    float acc[CIN];
#pragma unroll
    for (int c = 0; c < CIN; ++c) acc[c] = 0.f;
    for (int kd = rd, od = td; kd < q.KD; kd += q.sd, --od) {                 // od = (id + pd - kd) / sd = td - m
        const bool okd = od >= 0 && od < q.OD;
        for (int kh = rh, oh = th; kh < q.KH; kh += q.sh, --oh) {
            const bool okh = okd && oh >= 0 && oh < q.OH;
            for (int kw = rw, ow = tw; kw < q.KW; kw += q.sw, --ow) {
                if (!(okh && ow >= 0 && ow < q.OW)) continue;               // lanes outside dY sit this tap out (exec mask, no selects)
                const float* g = gb + ((int64_t)od * q.OH + oh) * q.OW + ow;
                const float* w = Wt + (int64_t)((kd * q.KH + kh) * q.KW + kw) * Cout * CIN;      // uniform, contiguous over (co, ci)
#pragma unroll 4
                for (int co = 0; co < Cout; ++co) {
                    const float gv = g[(int64_t)co * osz];
#pragma unroll
                    for (int c = 0; c < CIN; ++c) acc[c] += w[co * CIN + c] * gv;
                }
            }
        }
    }
    const int64_t pos = ((int64_t)id * q.IH + ih) * q.IW + iw;
#pragma unroll
    for (int c = 0; c < CIN; ++c) dX[((int64_t)b * CIN + c) * isz + pos] = acc[c];
}

// out[b][cell] = 1 if the average-pooled |x| summed over channels is > 0 (get_mask: segtran2d.py:229-233, segtran3d.py:266-270)
__global__ __launch_bounds__(256) void nonzero_mask_kernel(const float* __restrict__ X, float* __restrict__ out, int B, int C, int D, int H, int W,
                                                           int kd, int kh, int kw) {
    const int OD = D / kd, OH = H / kh, OW = W / kw;
    const int64_t total = (int64_t)B * OD * OH * OW;
    const float inv = 1.0f / (float)(kd * kh * kw);
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        int64_t r = idx; const int ow = (int)(r % OW); r /= OW; const int oh = (int)(r % OH); r /= OH; const int od = (int)(r % OD); const int b = (int)(r / OD);
        float tot = 0.f;
        for (int c = 0; c < C; ++c) {
            const float* x = X + (((int64_t)b * C + c) * D + od * kd) * H * W;
            float s = 0.f;
            for (int z = 0; z < kd; ++z) for (int y = 0; y < kh; ++y) for (int xx = 0; xx < kw; ++xx)
                s += fabsf(x[((int64_t)z * H + oh * kh + y) * W + ow * kw + xx]);
            tot += s * inv;
        }
        out[idx] = tot > 0.f ? 1.0f : 0.f;
    }
}

// r05: the foreground mask of the BRIDGED image (segtran3d.py:420-425: get_mask(in_bridge_to3(batch))) straight from the raw batch [B][Cb][H][W][D]: one wave per
// pooled cell (kd x kh x kw voxels, pooled over the permuted (D, H, W) order), every voxel's C3 bridge outputs formed in the GEMM's own order (an ascending-k
// fmaf chain, then + bias) and tested for |y| > 0.  The pooled average is positive iff one of its non-negative terms is, so the mask equals the one computed from
// the materialised bridge output -- which cost a K = 4 GEMM at 0.5 TFLOP/s, a permuting copy and a 768-load-per-thread pooling kernel (0.57 ms of the cfg5 step).
__global__ __launch_bounds__(256) void bridge_mask_kernel(const float* __restrict__ X, const float* __restrict__ Wb, const float* __restrict__ bb, float* __restrict__ out,
                                                          int B, int Cb, int C3, int H, int W, int D, int kd, int kh, int kw) {
    const int OD = D / kd, OH = H / kh, OW = W / kw;
    const int64_t cells = (int64_t)B * OD * OH * OW;
    const int lane = threadIdx.x & 63;
    const int64_t chan = (int64_t)H * W * D;
    for (int64_t cell = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); cell < cells; cell += (int64_t)gridDim.x * 4) {
        int64_t r = cell; const int ow = (int)(r % OW); r /= OW; const int oh = (int)(r % OH); r /= OH; const int od = (int)(r % OD); const int b = (int)(r / OD);
        const float* xb = X + (int64_t)b * Cb * chan;
        float any = 0.f, nan = 0.f;
        const int win = kd * kh * kw;
        for (int t = lane; t < win; t += 64) {
            const int z = t % kd, q = t / kd, xx = q % kw, y = q / kw;                   // z (the raw layout's contiguous axis) fastest
            const int64_t off = ((int64_t)(oh * kh + y) * W + (ow * kw + xx)) * D + od * kd + z;
            for (int c = 0; c < C3; ++c) {
                float acc = 0.f;
                for (int k = 0; k < Cb; ++k) acc = fmaf(Wb[c * Cb + k], xb[(int64_t)k * chan + off], acc);
                const float yv = acc + (bb ? bb[c] : 0.f);
                any = fmaxf(any, fabsf(yv) > 0.f ? 1.f : 0.f);
                nan = fmaxf(nan, yv != yv ? 1.f : 0.f);                                    // a NaN makes the pooled average NaN, and NaN > 0 is false
            }
        }
        any = wave_max(any); nan = wave_max(nan);
        if (lane == 0) out[cell] = nan > 0.f ? 0.f : any;
    }
}

// ---- input bridge composed into the I3D stem (segtran3d.py:420-423 `in_bridge_to3` = Conv3d(4 -> 3, 1x1x1, bias) feeding Conv3d_1a_7x7) ------------
// Two consecutive LINEAR maps: stem(pad0(Wb x + b)) = conv(pad0([x, 1, 0...]), Wc) with Wc[o][d][t] = sum_c Ws[o][c][t] Wb[c][d] for d < Cb,
// Wc[o][Cb][t] = sum_c Ws[o][c][t] b[c] (the bias rides on a constant-one channel, which the zero 'same' padding switches off outside the
// volume exactly as it switches off the padded bridge output), remaining channels zero (Cc = 8: the packed / bf16x6 contraction order).
// With the composition the stride-2 7x7x7 transposed convolution onto 3 channels (5.4 ms of cfg4's step) is never needed: the input carries
// no gradient, and dWs / dWb / db follow from dWc by the chain rule (stem_compose_bwd).
__global__ __launch_bounds__(256) void stem_compose_fwd_kernel(const float* __restrict__ Ws, const float* __restrict__ Wb, const float* __restrict__ bb,
                                                               float* __restrict__ Wc, int O, int C3, int Cb, int Cc, int T) {
    const int64_t total = (int64_t)O * Cc * T;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int t = (int)(i % T); const int64_t r = i / T; const int d = (int)(r % Cc), o = (int)(r / Cc);
        float v = 0.f;
        if (d <= Cb) for (int c = 0; c < C3; ++c) v += Ws[((int64_t)o * C3 + c) * T + t] * (d < Cb ? Wb[c * Cb + d] : (bb ? bb[c] : 0.f));
        Wc[i] = v;
    }
}
// dWs[o][c][t] = sum_{d < Cb} dWc[o][d][t] Wb[c][d] + dWc[o][Cb][t] b[c]
__global__ __launch_bounds__(256) void stem_compose_bwd_ws_kernel(const float* __restrict__ dWc, const float* __restrict__ Wb, const float* __restrict__ bb,
                                                                  float* __restrict__ dWs, int O, int C3, int Cb, int Cc, int T) {
    const int64_t total = (int64_t)O * C3 * T;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int t = (int)(i % T); const int64_t r = i / T; const int c = (int)(r % C3), o = (int)(r / C3);
        const float* g = dWc + (int64_t)o * Cc * T + t;
        float v = bb ? g[(int64_t)Cb * T] * bb[c] : 0.f;
        for (int d = 0; d < Cb; ++d) v += g[(int64_t)d * T] * Wb[c * Cb + d];
        dWs[i] = v;
    }
}
// one workgroup per (c, d <= Cb): dWb[c][d] (d < Cb) / db[c] (d == Cb) = sum_{o, t} Ws[o][c][t] dWc[o][d][t], fixed summation order
__global__ __launch_bounds__(256) void stem_compose_bwd_wb_kernel(const float* __restrict__ dWc, const float* __restrict__ Ws, float* __restrict__ dWb,
                                                                  float* __restrict__ dbb, int O, int C3, int Cb, int Cc, int T) {
    __shared__ float red[4];
    const int c = blockIdx.x / (Cb + 1), d = blockIdx.x % (Cb + 1);
    float acc = 0.f;
    for (int64_t i = threadIdx.x; i < (int64_t)O * T; i += 256) {
        const int o = (int)(i / T), t = (int)(i % T);
        acc += Ws[((int64_t)o * C3 + c) * T + t] * dWc[((int64_t)o * Cc + d) * T + t];
    }
    const float tot = block_sum<4>(acc, red);
    if (threadIdx.x == 0) { if (d < Cb) dWb[c * Cb + d] = tot; else if (dbb) dbb[c] = tot; }
}
// x [B][Cb][H][W][D] -> x8 [B][Cc][D][H][W]: the volume with depth moved in front (the reference permutes after the bridge, :422), a constant-one
// channel at index Cb and zero channels up to Cc.  Reads run along D (the input's contiguous axis), writes along W: tiles through LDS.
__global__ __launch_bounds__(256) void bridge_input_kernel(const float* __restrict__ X, float* __restrict__ Y, int Cb, int Cc, int H, int W, int D) {
    __shared__ float tile[32][33];
    const int bc = blockIdx.z, b = bc / Cc, c = bc % Cc, h = blockIdx.y;
    const int nwt = (W + 31) / 32, w0 = (blockIdx.x % nwt) * 32, d0 = (blockIdx.x / nwt) * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    if (c < Cb) {
        const float* x = X + (((int64_t)b * Cb + c) * H + h) * (int64_t)W * D;
        for (int r = ty; r < 32; r += 8) { const int w = w0 + r, d = d0 + tx; tile[r][tx] = (w < W && d < D) ? x[(int64_t)w * D + d] : 0.f; }
    }
    __syncthreads();
    float* y = Y + ((int64_t)b * Cc + c) * D * (int64_t)H * W;
    for (int r = ty; r < 32; r += 8) {
        const int d = d0 + r, w = w0 + tx;
        if (d < D && w < W) y[((int64_t)d * H + h) * W + w] = c < Cb ? tile[tx][r] : (c == Cb ? 1.0f : 0.f);
    }
}

// x [B][Cb][H][W][D] -> x2 [B][2 Cb][D][H][U]: depth in front AND space-to-depth along W for the stride-(2, 2, 1) form of the 3-D stem (SF.stem_bridge_conv_s2d):
//   x2[b][2 c + j][d][h][u] = x[b][c][h][2 u + j - 2][d]   (0 where 2 u + j - 2 falls outside [0, W): the 'same' front pad of 2 and the window's tail)
// one pass (pad + permute + reshape were three ATen copies of the batch, 0.13 - 0.25 ms per step); tiles through LDS as in bridge_input_kernel.
__global__ __launch_bounds__(256) void stem_s2d_input_kernel(const float* __restrict__ X, float* __restrict__ Y, int Cb, int H, int W, int D, int U) {
    __shared__ float tile[32][33];
    const int bc = blockIdx.z, b = bc / (2 * Cb), c2 = bc % (2 * Cb), c = c2 >> 1, j = c2 & 1, h = blockIdx.y;
    const int nut = (U + 31) / 32, u0 = (blockIdx.x % nut) * 32, d0 = (blockIdx.x / nut) * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const float* x = X + (((int64_t)b * Cb + c) * H + h) * (int64_t)W * D;
    for (int r = ty; r < 32; r += 8) {
        const int w = 2 * (u0 + r) + j - 2, d = d0 + tx;
        tile[r][tx] = (u0 + r < U && w >= 0 && w < W && d < D) ? x[(int64_t)w * D + d] : 0.f;
    }
    __syncthreads();
    float* y = Y + ((int64_t)b * 2 * Cb + c2) * D * (int64_t)H * U;
    for (int r = ty; r < 32; r += 8) {
        const int d = d0 + r, u = u0 + tx;
        if (d < D && u < U) y[((int64_t)d * H + h) * U + u] = tile[tx][r];
    }
}

// n-hot label maps of the train step (datasets2d.py:90-139,200-223; datasets3d.py:16-40), uint8 / int32 labels -> float planes
//   mode 0 fundus (exclusive=False): in [B,Cin>=2,S] uint8 -> [B,3,S]: (ch0==0, ch0>=1, ch1>=1)
//   mode 1 polyp: [B,Cin>=1,S] uint8 -> [B,2,S]: (ch0==0, ch0>0)
//   mode 3 fundus, exclusive=True (datasets2d.py:110-111): (ch0==0, ch0>=1 && ch1==0, ch1>=1)
//   mode 2 brats: [B,S] int32 -> [B,4,S]: (l==0, l==3, l in {1,2,3}, l in {1,3})
__global__ __launch_bounds__(256) void label_nhot_kernel(const void* __restrict__ lab, float* __restrict__ out, int B, int Cin, int64_t S, int mode) {
    const int64_t total = (int64_t)B * S;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        const int64_t b = idx / S, s = idx - b * S;
        if (mode == 2) {
            const int l = reinterpret_cast<const int*>(lab)[idx];
            float* o = out + b * 4 * S + s;
            o[0] = l == 0; o[S] = l == 3; o[2 * S] = (l == 1 || l == 2 || l == 3); o[3 * S] = (l == 1 || l == 3);
        } else {
            const unsigned char* m = reinterpret_cast<const unsigned char*>(lab) + b * Cin * S + s;
            if (mode == 0 || mode == 3) { float* o = out + b * 3 * S + s; o[0] = m[0] == 0; o[S] = (m[0] >= 1) && (mode == 0 || m[S] == 0); o[2 * S] = m[S] >= 1; }
            else { float* o = out + b * 2 * S + s; o[0] = m[0] == 0; o[S] = m[0] > 0; }
        }
    }
}

// =================================================================================================
// Max-pool with TF-'same' ZERO padding (aj_i3d.py:28-30 pads with F.pad, i.e. zeros, then pools; inputs are
// post-ReLU so this equals -inf padding, N7).  The scan order and "first maximum wins" rule of ATen are kept so the
// gradient routing is identical; a padded zero that wins receives (and drops) the gradient, as in the reference.
// =================================================================================================
struct PoolGeom { int ID, IH, IW, OD, OH, OW, KD, KH, KW, sd, sh, sw, pd, ph, pw; };

// TKD/TKH/TKW > 0: compile-time window (fully unrolled scan); 0: runtime window.  One thread per output; grid (chunks of a plane, planes):
// no 64-bit division, and the (od, oh, ow) decomposition is two multiply-high divisions (the generic integer divisions of the first version
// made these kernels ALU-bound: 0.9 ms for a 268 MB plane set that streams in 0.1 ms).
template <int TKD, int TKH, int TKW>
__global__ __launch_bounds__(256) void maxpool3d_fwd_kernel(const float* __restrict__ X, float* __restrict__ Y, int* __restrict__ arg,
                                                            PoolGeom q, int64_t planes, FastDiv dOHW, FastDiv dOW) {
    const int KD = TKD ? TKD : q.KD, KH = TKH ? TKH : q.KH, KW = TKW ? TKW : q.KW;
    const int osz = q.OD * q.OH * q.OW, isz = q.ID * q.IH * q.IW, ohw = q.OH * q.OW;
    for (int64_t p = blockIdx.y; p < planes; p += gridDim.y) {
        const float* x = X + p * isz;
        for (int r = blockIdx.x * 256 + threadIdx.x; r < osz; r += gridDim.x * 256) {
            const int od = fdiv(r, dOHW), r2 = r - od * ohw, oh = fdiv(r2, dOW), ow = r2 - oh * q.OW;
            const int d0 = od * q.sd - q.pd, h0 = oh * q.sh - q.ph, w0 = ow * q.sw - q.pw;
            float best = -INFINITY; int bi = -1;
#pragma unroll
            for (int kd = 0; kd < KD; ++kd) {
                const int id = d0 + kd; const bool okd = (unsigned)id < (unsigned)q.ID;
#pragma unroll
                for (int kh = 0; kh < KH; ++kh) {
                    const int ih = h0 + kh; const bool okh = okd && (unsigned)ih < (unsigned)q.IH;
                    const int rowbase = (id * q.IH + ih) * q.IW;
#pragma unroll
                    for (int kw = 0; kw < KW; ++kw) {
                        const int iw = w0 + kw; const bool in = okh && (unsigned)iw < (unsigned)q.IW;
                        const int li = rowbase + iw;
                        const float v = in ? x[li] : 0.f;                      // zero padding
                        if (v > best || v != v) { best = v; bi = in ? li : -1; }
                    }
                }
            }
            Y[p * osz + r] = best; arg[p * osz + r] = bi;
        }
    }
}
// Stride-1 pools with a 3 x 3 in-plane window (the Inception branch pools, TKD = 3; TKD = 1: a (1,3,3) pool), OW % 4 == 0: a thread owns FOUR adjacent
// outputs.  Per window row it loads the six inputs under them once (one aligned float4 + the two neighbours) instead of 4 x 3, and the bounds
// tests are per row, not per tap; the scan order (d, h, w, first maximum wins) per output is unchanged.
template <int TKD>
__global__ __launch_bounds__(256) void maxpool3d_fwd_s1w4_kernel(const float* __restrict__ X, float* __restrict__ Y, int* __restrict__ arg,
                                                                 PoolGeom q, int64_t planes, FastDiv dOHW4, FastDiv dOW4) {
    const int osz = q.OD * q.OH * q.OW, isz = q.ID * q.IH * q.IW, ow4 = q.OW >> 2, ohw4 = q.OH * ow4, osz4 = q.OD * ohw4;
    const int lane = threadIdx.x & 63;
    for (int64_t p = blockIdx.y; p < planes; p += gridDim.y) {
        const float* x = X + p * isz;
        // whole waves iterate together (the two neighbour columns come from the adjacent lanes' float4 by cross-lane moves -- the load unit, not HBM,
        // was the limit with two extra scalar loads per row); lanes past the end work on the last group and do not store
        for (int e0 = blockIdx.x * 256 + (threadIdx.x & ~63); e0 < osz4; e0 += gridDim.x * 256) {
            const bool live = e0 + lane < osz4;
            const int e = live ? e0 + lane : osz4 - 1;
            const int od = fdiv(e, dOHW4), r = e - od * ohw4, oh = fdiv(r, dOW4), ow0 = (r - oh * ow4) << 2;
            const int d0 = od - q.pd, h0 = oh - q.ph, w0 = ow0 - 1;                      // pw = 1 (host check)
            float best[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
            int bi[4] = {-1, -1, -1, -1};
            const bool inl = w0 >= 0, inr = ow0 + 4 < q.IW;
#pragma unroll
            for (int kd = 0; kd < TKD; ++kd) {
                const int id = d0 + kd; const bool okd = (unsigned)id < (unsigned)q.ID;
#pragma unroll
                for (int kh = 0; kh < 3; ++kh) {
                    const int ih = h0 + kh; const bool ok = okd && (unsigned)ih < (unsigned)q.IH;
                    const int rowbase = ((ok ? id : 0) * q.IH + (ok ? ih : 0)) * q.IW;
                    const float4 m = *reinterpret_cast<const float4*>(x + rowbase + ow0);
                    // left / right neighbour columns: the adjacent quad of the same row lives in the adjacent lane (consecutive lanes = consecutive
                    // quads; a row starts where inl is false) -- except at the wave's first / last lane, which load them
                    float vl = __shfl_up(m.w, 1), vr = __shfl_down(m.x, 1);
                    if (lane == 0 && inl) vl = x[rowbase + w0];
                    if (lane == 63 && inr) vr = x[rowbase + ow0 + 4];
                    float v[6]; bool in[6];
                    in[0] = ok && inl; in[5] = ok && inr; in[1] = in[2] = in[3] = in[4] = ok;
                    v[0] = in[0] ? vl : 0.f; v[5] = in[5] ? vr : 0.f;
                    v[1] = ok ? m.x : 0.f; v[2] = ok ? m.y : 0.f; v[3] = ok ? m.z : 0.f; v[4] = ok ? m.w : 0.f;      // zero padding
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int kw = 0; kw < 3; ++kw) {
                            const float val = v[j + kw];
                            if (val > best[j] || val != val) { best[j] = val; bi[j] = in[j + kw] ? rowbase + w0 + j + kw : -1; }
                        }
                }
            }
            if (live) {
                const int64_t o = p * osz + (od * q.OH + oh) * q.OW + ow0;
                *reinterpret_cast<float4*>(Y + o) = make_float4(best[0], best[1], best[2], best[3]);
                *reinterpret_cast<int4*>(arg + o) = make_int4(bi[0], bi[1], bi[2], bi[3]);
            }
        }
    }
}

// r05: the 3 x 3 x 3 stride-1 pool (pd = ph = pw = 1, O == I) SLIDING ALONG DEPTH.  The four-outputs-per-thread kernel above requests nine float4 rows per output
// quad and is bound by the rate at which the CU accepts load instructions, not by HBM (r05_c: 1.9 TB/s; one more load per row took it to 1.4).  Here a thread owns the
// same quad for TD consecutive output slices and walks the TD + 2 input slices under them once: per input slice three rows are loaded and reduced to the slice's
// (max, arg-max) for the four outputs -- the 3 x 3 in-plane scan in its usual order --, which then enters the three output slices it belongs to in scan order
// (as kd = 0 of slice s + 1, kd = 1 of s, kd = 2 of s - 1; `v > best || v != v` at both levels: the first maximum still wins, a NaN still takes over as in the
// flat scan).  3 (TD + 2) / TD row loads per output quad instead of 9, and 48 instead of 108 compare-selects.  Identical outputs and indices.
__global__ __launch_bounds__(256) void maxpool3d_fwd_s1w4d_kernel(const float* __restrict__ X, float* __restrict__ Y, int* __restrict__ arg,
                                                                  PoolGeom q, int64_t planes, int TD, int nchunk, FastDiv dOHW4, FastDiv dOW4) {
    const int osz = q.OD * q.OH * q.OW, ow4 = q.OW >> 2, ohw4 = q.OH * ow4, per = nchunk * ohw4;
    const int lane = threadIdx.x & 63;
    for (int64_t p = blockIdx.y; p < planes; p += gridDim.y) {
        const float* x = X + p * osz;
        for (int e0 = blockIdx.x * 256 + (threadIdx.x & ~63); e0 < per; e0 += gridDim.x * 256) {        // whole waves (neighbour columns across lanes)
            const bool live = e0 + lane < per;
            const int e = live ? e0 + lane : per - 1;
            const int ck = fdiv(e, dOHW4), r = e - ck * ohw4, oh = fdiv(r, dOW4), ow0 = (r - oh * ow4) << 2;
            const int d0 = ck * TD, d1 = min(d0 + TD, q.OD), w0 = ow0 - 1;
            const bool inl = w0 >= 0, inr = ow0 + 4 < q.IW;
            float b0[4], b1[4]; int i0[4], i1[4];                 // output slices s - 1 (waiting for kd = 2) and s (waiting for kd = 1, 2)
#pragma unroll
            for (int j = 0; j < 4; ++j) { b0[j] = b1[j] = -INFINITY; i0[j] = i1[j] = -1; }
            for (int ks = 0; ks < TD + 2; ++ks) {                 // the SAME trip count in every lane (the neighbour columns travel across lanes): slices past d1 store nothing
                const int sl = d0 - 1 + ks;
                const bool okd = (unsigned)sl < (unsigned)q.ID;
                float sv[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY}; int si[4] = {-1, -1, -1, -1};
#pragma unroll
                for (int kh = 0; kh < 3; ++kh) {
                    const int ih = oh - 1 + kh; const bool ok = okd && (unsigned)ih < (unsigned)q.IH;
                    const int rowbase = ((ok ? sl : 0) * q.IH + (ok ? ih : 0)) * q.IW;
                    const float4 m = *reinterpret_cast<const float4*>(x + rowbase + ow0);
                    float vl = __shfl_up(m.w, 1), vr = __shfl_down(m.x, 1);
                    if (lane == 0 && inl) vl = x[rowbase + w0];
                    if (lane == 63 && inr) vr = x[rowbase + ow0 + 4];
                    float v[6]; bool in[6];
                    in[0] = ok && inl; in[5] = ok && inr; in[1] = in[2] = in[3] = in[4] = ok;
                    v[0] = in[0] ? vl : 0.f; v[5] = in[5] ? vr : 0.f;
                    v[1] = ok ? m.x : 0.f; v[2] = ok ? m.y : 0.f; v[3] = ok ? m.z : 0.f; v[4] = ok ? m.w : 0.f;      // zero padding
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int kw = 0; kw < 3; ++kw) {
                            const float val = v[j + kw];
                            if (val > sv[j] || val != val) { sv[j] = val; si[j] = in[j + kw] ? rowbase + w0 + j + kw : -1; }
                        }
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (sv[j] > b0[j] || sv[j] != sv[j]) { b0[j] = sv[j]; i0[j] = si[j]; }                 // kd = 2 of output slice sl - 1: complete
                    if (sv[j] > b1[j] || sv[j] != sv[j]) { b1[j] = sv[j]; i1[j] = si[j]; }                 // kd = 1 of output slice sl
                }
                if (live && sl - 1 >= d0 && sl - 1 < d1) {
                    const int64_t o = p * osz + ((sl - 1) * q.OH + oh) * q.OW + ow0;
                    *reinterpret_cast<float4*>(Y + o) = make_float4(b0[0], b0[1], b0[2], b0[3]);
                    *reinterpret_cast<int4*>(arg + o) = make_int4(i0[0], i0[1], i0[2], i0[3]);
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {                      // rotate: slice sl becomes "sl - 1"; slice sl + 1 starts with this input slice as its kd = 0
                    b0[j] = b1[j]; i0[j] = i1[j];
                    const bool take = sv[j] > -INFINITY || sv[j] != sv[j];
                    b1[j] = take ? sv[j] : -INFINITY; i1[j] = take ? si[j] : -1;
                }
            }
        }
    }
}
// ... and its gather: a thread owns four adjacent input cells for TD consecutive slices and walks the TD + 2 window slices that can cover them once (three rows of
// (arg, gradient) per slice, six window columns each); a window slice s feeds the cells of slices s + 1, s, s - 1 -- so a cell still receives its windows in
// (od, oh, ow) order.  6 (TD + 2) / TD row loads per cell quad instead of 18.
__global__ __launch_bounds__(256) void maxpool3d_bwd_s1w4d_kernel(const float* __restrict__ dY, const int* __restrict__ arg, float* __restrict__ dX,
                                                                  PoolGeom q, int64_t planes, int TD, int nchunk, FastDiv dIHW4, FastDiv dIW4) {
    const int isz = q.ID * q.IH * q.IW, iw4 = q.IW >> 2, ihw4 = q.IH * iw4, per = nchunk * ihw4, hw = q.IH * q.IW;
    const int lane = threadIdx.x & 63;
    for (int64_t p = blockIdx.y; p < planes; p += gridDim.y) {
        const float* g = dY + p * isz; const int* a = arg + p * isz;
        for (int e0 = blockIdx.x * 256 + (threadIdx.x & ~63); e0 < per; e0 += gridDim.x * 256) {
            const bool live = e0 + lane < per;
            const int e = live ? e0 + lane : per - 1;
            const int ck = fdiv(e, dIHW4), r = e - ck * ihw4, ih = fdiv(r, dIW4), iw0 = (r - ih * iw4) << 2;
            const int d0 = ck * TD, d1 = min(d0 + TD, q.ID);
            const int lrow = ih * q.IW + iw0;                       // li0 of cell slice d = d * hw + lrow
            const bool inl = iw0 >= 1, inr = iw0 + 4 < q.OW;
            float a0[4] = {0.f, 0.f, 0.f, 0.f}, a1[4] = {0.f, 0.f, 0.f, 0.f};       // cell slices s - 1 (waiting for its last window slice) and s
            for (int ks = 0; ks < TD + 2; ++ks) {                 // the SAME trip count in every lane (the neighbour columns travel across lanes): slices past d1 store nothing
                const int sl = d0 - 1 + ks;
                const bool okd = (unsigned)sl < (unsigned)q.OD;
                float a2[4] = {0.f, 0.f, 0.f, 0.f};                 // cell slice sl + 1: this window slice is its first
                const int lm = (sl - 1) * hw + lrow, lc = lm + hw, lp = lc + hw;        // li0 of the cell slices sl - 1, sl, sl + 1
#pragma unroll
                for (int y = 0; y < 3; ++y) {
                    const int oh = ih - 1 + y; const bool ok = okd && (unsigned)oh < (unsigned)q.OH;
                    const int rowo = ((ok ? sl : 0) * q.OH + (ok ? oh : 0)) * q.OW;
                    const int4 am = *reinterpret_cast<const int4*>(a + rowo + iw0);
                    const float4 gm = *reinterpret_cast<const float4*>(g + rowo + iw0);
                    int al = __shfl_up(am.w, 1), ar = __shfl_down(am.x, 1);
                    float gl = __shfl_up(gm.w, 1), gr = __shfl_down(gm.x, 1);
                    if (lane == 0 && inl) { al = a[rowo + iw0 - 1]; gl = g[rowo + iw0 - 1]; }
                    if (lane == 63 && inr) { ar = a[rowo + iw0 + 4]; gr = g[rowo + iw0 + 4]; }
                    int av[6]; float gv[6];
                    av[0] = al; gv[0] = (ok && inl) ? gl : 0.f;
                    av[5] = ar; gv[5] = (ok && inr) ? gr : 0.f;
                    av[1] = am.x; av[2] = am.y; av[3] = am.z; av[4] = am.w;
                    gv[1] = ok ? gm.x : 0.f; gv[2] = ok ? gm.y : 0.f; gv[3] = ok ? gm.z : 0.f; gv[4] = ok ? gm.w : 0.f;
#pragma unroll
                    for (int t = 0; t < 6; ++t) {
                        const int jm = av[t] - lm, jc = av[t] - lc, jp = av[t] - lp;
#pragma unroll
                        for (int c = (t >= 2 ? t - 2 : 0); c <= (t <= 3 ? t : 3); ++c) {
                            a0[c] += jm == c ? gv[t] : 0.f; a1[c] += jc == c ? gv[t] : 0.f; a2[c] += jp == c ? gv[t] : 0.f;
                        }
                    }
                }
                if (live && sl - 1 >= d0 && sl - 1 < d1) *reinterpret_cast<float4*>(dX + p * isz + lm) = make_float4(a0[0], a0[1], a0[2], a0[3]);
#pragma unroll
                for (int c = 0; c < 4; ++c) { a0[c] = a1[c]; a1[c] = a2[c]; }
            }
        }
    }
}

// gather form: an input cell sums the gradients of the windows whose arg-max it is.  S2 = true: strides (1 or 2, 2, 2) known at compile time
// (the down-sampling pools of I3D: the window-range divisions become shifts); grid (chunks of a plane, planes), multiply-high decomposition.
template <bool S2>
__global__ __launch_bounds__(256) void maxpool3d_bwd_kernel(const float* __restrict__ dY, const int* __restrict__ arg, float* __restrict__ dX,
                                                            PoolGeom q, int64_t planes, FastDiv dIHW, FastDiv dIW, const float* __restrict__ addend) {
    const int osz = q.OD * q.OH * q.OW, isz = q.ID * q.IH * q.IW, ihw = q.IH * q.IW;
    const int sd = q.sd, sh = S2 ? 2 : q.sh, sw = S2 ? 2 : q.sw;
    for (int64_t p = blockIdx.y; p < planes; p += gridDim.y) {
        const float* g = dY + p * osz; const int* a = arg + p * osz;
        for (int li = blockIdx.x * 256 + threadIdx.x; li < isz; li += gridDim.x * 256) {
            const int id = fdiv(li, dIHW), r = li - id * ihw, ih = fdiv(r, dIW), iw = r - ih * q.IW;
            float acc = 0.f;
            // windows covering (id,ih,iw): od in [ceil((id+pd-KD+1)/sd), floor((id+pd)/sd)]
            const int d1 = min(sd == 1 ? id + q.pd : sd == 2 ? (id + q.pd) >> 1 : (id + q.pd) / sd, q.OD - 1);
            const int h1 = min((ih + q.ph) / sh, q.OH - 1), w1 = min((iw + q.pw) / sw, q.OW - 1);
            const int dn = id + q.pd - q.KD + 1, hn = ih + q.ph - q.KH + 1, wn = iw + q.pw - q.KW + 1;
            const int d0 = dn > 0 ? (sd == 1 ? dn : sd == 2 ? (dn + 1) >> 1 : (dn + sd - 1) / sd) : 0;
            const int h0 = hn > 0 ? (hn + sh - 1) / sh : 0, w0 = wn > 0 ? (wn + sw - 1) / sw : 0;
            for (int od = d0; od <= d1; ++od) for (int oh = h0; oh <= h1; ++oh) {
                const int rowo = (od * q.OH + oh) * q.OW;
                for (int ow = w0; ow <= w1; ++ow) if (a[rowo + ow] == li) acc += g[rowo + ow];
            }
            dX[p * isz + li] = addend ? addend[p * isz + li] + acc : acc;
        }
    }
}

// In-plane stride 2 (the down-sampling pools: windows 3 or 2 wide), IW % 4 == 0: a thread owns FOUR adjacent input cells and one float4 store.
// Their covering windows are <= 3 columns x 2 rows x 2 slices; a window's arg-max can only be one of the cells it covers, so "arg - li0 in [0, 4)"
// is the whole test -- 6 .. 12 probes for four cells instead of 16 .. 32, no per-cell range arithmetic.  Accumulation order (od, oh, ow) as in the
// generic gather.
__global__ __launch_bounds__(256) void maxpool3d_bwd_s2w4_kernel(const float* __restrict__ dY, const int* __restrict__ arg, float* __restrict__ dX,
                                                                 PoolGeom q, int64_t planes, FastDiv dIHW4, FastDiv dIW4, const float* __restrict__ addend) {
    const int osz = q.OD * q.OH * q.OW, isz = q.ID * q.IH * q.IW, iw4 = q.IW >> 2, ihw4 = q.IH * iw4, isz4 = q.ID * ihw4;
    for (int64_t p = blockIdx.y; p < planes; p += gridDim.y) {
        const float* g = dY + p * osz; const int* a = arg + p * osz;
        for (int e = blockIdx.x * 256 + threadIdx.x; e < isz4; e += gridDim.x * 256) {
            const int id = fdiv(e, dIHW4), r = e - id * ihw4, ih = fdiv(r, dIW4), iw0 = (r - ih * iw4) << 2;
            const int li0 = (id * q.IH + ih) * q.IW + iw0;
            const int d1 = min(q.sd == 1 ? id + q.pd : (id + q.pd) >> 1, q.OD - 1), h1 = min((ih + q.ph) >> 1, q.OH - 1), w1 = min((iw0 + 3 + q.pw) >> 1, q.OW - 1);
            const int dn = id + q.pd - q.KD + 1, hn = ih + q.ph - q.KH + 1, wn = iw0 + q.pw - q.KW + 1;
            const int d0 = dn > 0 ? (q.sd == 1 ? dn : (dn + 1) >> 1) : 0, h0 = hn > 0 ? (hn + 1) >> 1 : 0, w0 = wn > 0 ? (wn + 1) >> 1 : 0;
            float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
            for (int od = d0; od <= d1; ++od) for (int oh = h0; oh <= h1; ++oh) {        // (a fixed 2 x 2 x 3 probe set with clamped addresses measured slower: 323 against 259 us, r05_d)
                const int rowo = (od * q.OH + oh) * q.OW;
                for (int ow = w0; ow <= w1; ++ow) {
                    const int j = a[rowo + ow] - li0;
                    const float gv = g[rowo + ow];
                    a0 += j == 0 ? gv : 0.f; a1 += j == 1 ? gv : 0.f; a2 += j == 2 ? gv : 0.f; a3 += j == 3 ? gv : 0.f;
                }
            }
            if (addend) {                                        // r05: the gradient of the input's OTHER consumers (the pooled tensor is an FPN endpoint), added here
                const float4 ad = *reinterpret_cast<const float4*>(addend + p * isz + li0);          // instead of by an accumulation kernel of autograd's (2.4 GB of traffic at cfg5)
                a0 += ad.x; a1 += ad.y; a2 += ad.z; a3 += ad.w;
            }
            *reinterpret_cast<float4*>(dX + p * isz + li0) = make_float4(a0, a1, a2, a3);
        }
    }
}

// Stride-1 pools with a 3 x 3 in-plane window, IW % 4 == 0: FOUR adjacent input cells per thread.  The windows that can cover them are six columns
// (ow = iw0 - 1 .. iw0 + 4: one aligned int4 / float4 + the two neighbours per window row) x 3 rows x TKD slices; column t can only be the arg-max
// of cells t - 2 .. t, so each probe is "arg - li0 == j" for at most three j.  Same (od, oh, ow) accumulation order as the other gathers; no LDS.
template <int TKD>
__global__ __launch_bounds__(256) void maxpool3d_bwd_s1w4_kernel(const float* __restrict__ dY, const int* __restrict__ arg, float* __restrict__ dX,
                                                                 PoolGeom q, int64_t planes, FastDiv dIHW4, FastDiv dIW4) {
    const int osz = q.OD * q.OH * q.OW, isz = q.ID * q.IH * q.IW, iw4 = q.IW >> 2, ihw4 = q.IH * iw4, isz4 = q.ID * ihw4;
    const int lane = threadIdx.x & 63;
    for (int64_t p = blockIdx.y; p < planes; p += gridDim.y) {
        const float* g = dY + p * osz; const int* a = arg + p * osz;
        for (int e0 = blockIdx.x * 256 + (threadIdx.x & ~63); e0 < isz4; e0 += gridDim.x * 256) {     // whole waves (cross-lane neighbour columns, as in the forward)
            const bool live = e0 + lane < isz4;
            const int e = live ? e0 + lane : isz4 - 1;
            const int id = fdiv(e, dIHW4), r = e - id * ihw4, ih = fdiv(r, dIW4), iw0 = (r - ih * iw4) << 2;
            const int li0 = (id * q.IH + ih) * q.IW + iw0;
            const bool inl = iw0 >= 1, inr = iw0 + 4 < q.OW;                 // OW == IW, pw == 1 (host check): window column ow covers cells ow - 1 .. ow + 1
            float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int z = 0; z < TKD; ++z) {
                const int od = id + q.pd - (TKD - 1) + z; const bool okd = (unsigned)od < (unsigned)q.OD;
#pragma unroll
                for (int y = 0; y < 3; ++y) {
                    const int oh = ih + q.ph - 2 + y; const bool ok = okd && (unsigned)oh < (unsigned)q.OH;
                    const int rowo = ((ok ? od : 0) * q.OH + (ok ? oh : 0)) * q.OW;
                    const int4 am = *reinterpret_cast<const int4*>(a + rowo + iw0);
                    const float4 gm = *reinterpret_cast<const float4*>(g + rowo + iw0);
                    int al = __shfl_up(am.w, 1), ar = __shfl_down(am.x, 1);
                    float gl = __shfl_up(gm.w, 1), gr = __shfl_down(gm.x, 1);
                    if (lane == 0 && inl) { al = a[rowo + iw0 - 1]; gl = g[rowo + iw0 - 1]; }
                    if (lane == 63 && inr) { ar = a[rowo + iw0 + 4]; gr = g[rowo + iw0 + 4]; }
                    int av[6]; float gv[6];
                    av[0] = al; gv[0] = (ok && inl) ? gl : 0.f;
                    av[5] = ar; gv[5] = (ok && inr) ? gr : 0.f;
                    av[1] = am.x; av[2] = am.y; av[3] = am.z; av[4] = am.w;
                    gv[1] = ok ? gm.x : 0.f; gv[2] = ok ? gm.y : 0.f; gv[3] = ok ? gm.z : 0.f; gv[4] = ok ? gm.w : 0.f;
#pragma unroll
                    for (int t = 0; t < 6; ++t) {
                        const int j = av[t] - li0;
#pragma unroll
                        for (int c = (t >= 2 ? t - 2 : 0); c <= (t <= 3 ? t : 3); ++c) acc[c] += j == c ? gv[t] : 0.f;
                    }
                }
            }
            if (live) *reinterpret_cast<float4*>(dX + p * isz + li0) = make_float4(acc[0], acc[1], acc[2], acc[3]);
        }
    }
}

// Stride-1 3x3x3 pools (the Inception branch pools, 10 of the 12 per step): a workgroup owns a 4 x 8 x 32 tile of input cells and stages
// the arg-max indices and gradients of the 6 x 10 x 34 windows that can cover it in LDS once; each cell then probes its 27 windows
// there (same od, oh, ow order as the generic gather).  The generic kernel issued those 27 probes against L1/L2: 1 TB/s (r01-j PMC).
constexpr int MP_TD = 4, MP_TH = 8, MP_TW = 32;
__global__ __launch_bounds__(256) void maxpool3d_bwd_s1k3_kernel(const float* __restrict__ dY, const int* __restrict__ arg, float* __restrict__ dX,
                                                                 PoolGeom q, int tiles_h, int tiles_w) {
    __shared__ int sa[MP_TD + 2][MP_TH + 2][MP_TW + 2];
    __shared__ float sg[MP_TD + 2][MP_TH + 2][MP_TW + 2];
    const int64_t p = blockIdx.y;
    int t = blockIdx.x; const int tw = t % tiles_w; t /= tiles_w; const int th = t % tiles_h, td = t / tiles_h;
    const int d0 = td * MP_TD, h0 = th * MP_TH, w0 = tw * MP_TW;
    const int osz = q.OD * q.OH * q.OW, isz = q.ID * q.IH * q.IW;
    const float* g = dY + p * osz; const int* a = arg + p * osz;
    const int od0 = d0 + q.pd - 2, oh0 = h0 + q.ph - 2, ow0 = w0 + q.pw - 2;           // first window that can cover the tile's first cell
    constexpr int HN = (MP_TD + 2) * (MP_TH + 2) * (MP_TW + 2);
    for (int i = threadIdx.x; i < HN; i += 256) {
        const int hx = i % (MP_TW + 2), r = i / (MP_TW + 2), hy = r % (MP_TH + 2), hz = r / (MP_TH + 2);
        const int od = od0 + hz, oh = oh0 + hy, ow = ow0 + hx;
        const bool in = (unsigned)od < (unsigned)q.OD && (unsigned)oh < (unsigned)q.OH && (unsigned)ow < (unsigned)q.OW;
        const int o = in ? (od * q.OH + oh) * q.OW + ow : 0;
        sa[hz][hy][hx] = in ? a[o] : -2;
        sg[hz][hy][hx] = in ? g[o] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < MP_TD * MP_TH * MP_TW / 256; ++j) {
        const int c = threadIdx.x + 256 * j, cx = c % MP_TW, cy = (c / MP_TW) % MP_TH, cz = c / (MP_TW * MP_TH);
        const int id = d0 + cz, ih = h0 + cy, iw = w0 + cx;
        if (id >= q.ID || ih >= q.IH || iw >= q.IW) continue;
        const int li = (id * q.IH + ih) * q.IW + iw;
        float acc = 0.f;
#pragma unroll
        for (int z = 0; z < 3; ++z)
#pragma unroll
            for (int y = 0; y < 3; ++y)
#pragma unroll
                for (int x = 0; x < 3; ++x) if (sa[cz + z][cy + y][cx + x] == li) acc += sg[cz + z][cy + y][cx + x];
        dX[p * isz + li] = acc;
    }
}

// r05: stride-1 3 x 3 x 3 'same' pools (the Inception branch pools: 10 of the 12 pools of a step) with a SLAB of the plane in LDS.  A workgroup owns
// TD output slices of one (sample, channel) plane and stages the TD + 2 input slices under them (forward: x; backward: the window gradients and
// arg-max indices) with whole-line float4 loads -- the plane is contiguous, so the copy is a linear one whatever the row length (rows of 14 or 7
// floats defeat the four-outputs-per-thread forms above: r04_w measured 1.08 TB/s for the tile-with-halo gather and 2.5 TB/s for the per-output
// scan on the 24 x 14 x 14 planes of cfg4).  The 27 taps / probes then run against LDS with consecutive lanes on consecutive cells.  Halo
// re-reads: (TD + 2) / TD of ONE of the three streams (the neighbouring slab's lines are in the XCD's L2 when its workgroup runs beside this one).
// Same scan order, "first maximum wins" rule, zero padding and (od, oh, ow) accumulation order as the kernels above: identical results.
// CAP = floats of LDS per staged array (2048: planes of <= 2048 floats, one slab; 8192: everything else)
template <int CAP>
__global__ __launch_bounds__(256) void maxpool3d_fwd_slab_kernel(const float* __restrict__ X, float* __restrict__ Y, int* __restrict__ arg, PoolGeom q,
                                                                 int TD, int nslab, FastDiv dHW, FastDiv dW) {
    __shared__ __attribute__((aligned(16))) float sx[CAP];
    const int64_t p = blockIdx.x / nslab;
    const int slab = blockIdx.x - (int)p * nslab;
    const int hw = q.IH * q.IW, isz = q.ID * hw;
    const int d0 = slab * TD, d1 = min(d0 + TD, q.ID);               // output slices [d0, d1)
    const int s0 = max(d0 - 1, 0), s1 = min(d1 + 1, q.ID);           // staged input slices [s0, s1)
    const int base = s0 * hw, n = (s1 - s0) * hw;
    const float* x = X + p * isz + base;
    if ((((int64_t)p * isz + base) & 3) == 0 && n >= 4) {            // 16-byte aligned start (the tensor base is: host check)
        // every load of the slab is in flight before the first LDS store (a load + store loop is one HBM round trip per iteration: r05_b measured the
        // first version of this kernel at 0.6 - 1.6 TB/s); addresses past the slab are clamped, their stores predicated
        const int n4 = n >> 2;
        constexpr int NL = CAP / 1024;
        float4 v[NL];
#pragma unroll
        for (int k = 0; k < NL; ++k) { const int i = threadIdx.x + 256 * k; v[k] = reinterpret_cast<const float4*>(x)[i < n4 ? i : (n4 > 0 ? n4 - 1 : 0)]; }
#pragma unroll
        for (int k = 0; k < NL; ++k) { const int i = threadIdx.x + 256 * k; if (i < n4) reinterpret_cast<float4*>(sx)[i] = v[k]; }
        for (int i = (n4 << 2) + threadIdx.x; i < n; i += 256) sx[i] = x[i];
    } else for (int i = threadIdx.x; i < n; i += 256) sx[i] = x[i];
    __syncthreads();
    const int cells = (d1 - d0) * hw;
    for (int c = threadIdx.x; c < cells; c += 256) {
        const int zd = fdiv(c, dHW), r2 = c - zd * hw, oh = fdiv(r2, dW), ow = r2 - oh * q.IW, od = d0 + zd;
        float best = -INFINITY; int bi = -1;                  // (requesting all 27 taps before the compare chain measured SLOWER here: 96 against 70 us on 24 x 14 x 14, r05_d)
#pragma unroll
        for (int kd = 0; kd < 3; ++kd) {
            const int id = od - 1 + kd; const bool okd = (unsigned)id < (unsigned)q.ID;
#pragma unroll
            for (int kh = 0; kh < 3; ++kh) {
                const int ih = oh - 1 + kh; const bool okh = okd && (unsigned)ih < (unsigned)q.IH;
                const int rowbase = (id * q.IH + ih) * q.IW;
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) {
                    const int iw = ow - 1 + kw; const bool in = okh && (unsigned)iw < (unsigned)q.IW;
                    const int li = rowbase + iw;
                    const float v = in ? sx[li - base] : 0.f;            // zero padding
                    if (v > best || v != v) { best = v; bi = in ? li : -1; }
                }
            }
        }
        const int64_t o = p * isz + (int64_t)d0 * hw + c;
        Y[o] = best; arg[o] = bi;
    }
}
template <int CAP>
__global__ __launch_bounds__(256) void maxpool3d_bwd_slab_kernel(const float* __restrict__ dY, const int* __restrict__ arg, float* __restrict__ dX, PoolGeom q,
                                                                 int TD, int nslab, FastDiv dHW, FastDiv dW) {
    __shared__ __attribute__((aligned(16))) float sg[CAP];
    __shared__ __attribute__((aligned(16))) int sa[CAP];
    const int64_t p = blockIdx.x / nslab;
    const int slab = blockIdx.x - (int)p * nslab;
    const int hw = q.IH * q.IW, isz = q.ID * hw;
    const int d0 = slab * TD, d1 = min(d0 + TD, q.ID);               // input slices [d0, d1) of this workgroup
    const int s0 = max(d0 - 1, 0), s1 = min(d1 + 1, q.ID);           // window slices that can cover them
    const int base = s0 * hw, n = (s1 - s0) * hw;
    const float* g = dY + p * isz + base; const int* a = arg + p * isz + base;
    if ((((int64_t)p * isz + base) & 3) == 0 && n >= 4) {
        const int n4 = n >> 2;                                       // all loads in flight before the first LDS store (see the forward kernel)
        constexpr int NL = CAP / 1024;
        float4 vg[NL]; int4 va[NL];
#pragma unroll
        for (int k = 0; k < NL; ++k) {
            const int i = threadIdx.x + 256 * k, ic = i < n4 ? i : (n4 > 0 ? n4 - 1 : 0);
            vg[k] = reinterpret_cast<const float4*>(g)[ic]; va[k] = reinterpret_cast<const int4*>(a)[ic];
        }
#pragma unroll
        for (int k = 0; k < NL; ++k) {
            const int i = threadIdx.x + 256 * k;
            if (i < n4) { reinterpret_cast<float4*>(sg)[i] = vg[k]; reinterpret_cast<int4*>(sa)[i] = va[k]; }
        }
        for (int i = (n4 << 2) + threadIdx.x; i < n; i += 256) { sg[i] = g[i]; sa[i] = a[i]; }
    } else for (int i = threadIdx.x; i < n; i += 256) { sg[i] = g[i]; sa[i] = a[i]; }
    __syncthreads();
    const int cells = (d1 - d0) * hw;
    for (int c = threadIdx.x; c < cells; c += 256) {
        const int zd = fdiv(c, dHW), r2 = c - zd * hw, ih = fdiv(r2, dW), iw = r2 - ih * q.IW, id = d0 + zd;
        const int li = d0 * hw + c;
        // branch-free: the 27 index words, then the 27 gradients (clamped addresses), are requested back to back and selected afterwards.  (r05_b / r05_c: the
        // first form -- `if (in range && sa[o] == li) acc += sg[o]` -- compiled to 27 x 2 DEPENDENT LDS round trips under branches: 0.6 TB/s.)
        int off[27]; bool ok[27];
#pragma unroll
        for (int z = 0; z < 3; ++z) {
            const int od = id - 1 + z; const bool okd = (unsigned)od < (unsigned)q.ID;
#pragma unroll
            for (int y = 0; y < 3; ++y) {
                const int oh = ih - 1 + y; const bool okh = okd && (unsigned)oh < (unsigned)q.IH;
                const int rowo = (od * q.IH + oh) * q.IW - base;
#pragma unroll
                for (int xx = 0; xx < 3; ++xx) {
                    const int ow = iw - 1 + xx, t = (z * 3 + y) * 3 + xx;
                    ok[t] = okh && (unsigned)ow < (unsigned)q.IW;
                    off[t] = ok[t] ? rowo + ow : 0;
                }
            }
        }
        int av[27]; float gv[27];
#pragma unroll
        for (int t = 0; t < 27; ++t) av[t] = sa[off[t]];
#pragma unroll
        for (int t = 0; t < 27; ++t) gv[t] = sg[off[t]];
        float acc = 0.f;
#pragma unroll
        for (int t = 0; t < 27; ++t) acc += (ok[t] && av[t] == li) ? gv[t] : 0.f;      // (od, oh, ow) order, as every other gather
        dX[p * isz + li] = acc;
    }
}
// slices per slab for the slab pools: the largest TD with (TD + 2) slices <= cap floats, 0 = a slice does not fit
static int pool_slab_td(const PoolGeom& q, int cap) { const int fit = cap / (q.IH * q.IW); return fit >= q.ID ? q.ID : fit >= 3 ? fit - 2 : 0; }
static bool pool_is_s1k3_same(const PoolGeom& q) {
    return q.KD == 3 && q.KH == 3 && q.KW == 3 && q.sd == 1 && q.sh == 1 && q.sw == 1 && q.pd == 1 && q.ph == 1 && q.pw == 1 && q.OD == q.ID && q.OH == q.IH && q.OW == q.IW;
}

static bool aligned16c(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
static bool conv_small(int Cout) { return kget(knobs().conv_small_policy) == 1 ? Cout <= 64 : (Cout % 128 >= 1 && Cout % 128 <= 64); }

}  // namespace segx

using namespace segx;
#define SEGX_STREAM hipStream_t stream = (hipStream_t)stream_

static ConvGeom make_geom(const int* g) {
    ConvGeom q; q.Cin = g[0]; q.ID = g[1]; q.IH = g[2]; q.IW = g[3]; q.OD = g[4]; q.OH = g[5]; q.OW = g[6];
    q.KD = g[7]; q.KH = g[8]; q.KW = g[9]; q.sd = g[10]; q.sh = g[11]; q.sw = g[12]; q.pd = g[13]; q.ph = g[14]; q.pw = g[15];
    return q;
}
static void fill_common(GemmArgs& g, int M, int N, int K, int nbatch, int splitk, float* workspace) {
    g.bias = nullptr; g.aux = nullptr; g.gmax = nullptr; g.nb1 = 1; g.bias_b1 = 0; g.bias_b0 = 0; g.alpha = 1.0f; g.epilogue = SEGX_EPI_NONE; g.bias_mode = SEGX_BIAS_NONE;
    g.a_b1 = g.b_b1 = g.c_b1 = 0; g.b_n = g.b_k = 0; g.a_k = 1; g.vecA = g.vecB = 0;
    g.M = M; g.N = N; g.K = K; g.tiles_m = ceil_div(M, BM); g.tiles_n = ceil_div(N, BN);
    g.dropout_p = 0.f; g.seed = g.offset = 0; g.rbase = nullptr; g.splitk = splitk; g.slab = 0; g.resid = nullptr;
    g.k_chunk = splitk == 1 ? K : ceil_div(ceil_div(K, splitk), BKT) * BKT;
    g.c_split = (int64_t)nbatch * M * N;
    if (splitk > 1) g.C = workspace;
}

/* geom = {Cin, ID, IH, IW, OD, OH, OW, KD, KH, KW, sd, sh, sw, pd, ph, pw} (front pads) */
// split factor of the implicit GEMM (M = Cout, N, K, B samples) on the tile the convolution kernels use for this Cout
// x6: the launch will run on the bf16x6 engine (3 / 4 resident workgroups per CU and its own k-tile times: a split factor chosen for the
// fp32 engine's 512 slots left 1008 workgroups on 768 slots -- 1.3 rounds, 64.8 TFLOP/s on the 192 x 1728 x 150528 weight gradient, r02_c)
static int conv_splitk(int M, int N, int K, int B, bool x6) {
    const bool small = conv_small(M);
    double t;
    if (x6) {
        const TileInfo6& c6 = small ? kTiles6[1] : kTiles6[0];
        const TileInfo c{c6.id, c6.bm, c6.bn, c6.wg_per_cu, c6.ktile_us, c6.fixed_us};
        return best_splitk(c, M, N, K, B, &t);
    }
    return best_splitk(tile_info(small ? SEGX_TILE_64x128 : SEGX_TILE_128x128), M, N, K, B, &t);
}
/* the library's split-K factor for segx_conv3d_fwd (wgrad = 0) / segx_conv3d_bwd_weight (wgrad = 1); workspace = splitk * output floats */
extern "C" int64_t segx_conv3d_splitk(int B, int Cout, const int* geom, int wgrad) {
    if (!geom || B <= 0 || Cout <= 0) return 1;
    const ConvGeom q = make_geom(geom);
    const int64_t P = (int64_t)q.OD * q.OH * q.OW, CK = (int64_t)q.Cin * q.KD * q.KH * q.KW;
    if (P <= 0 || P >= 2147483647LL || CK <= 0 || CK >= 2147483647LL) return 1;
    // which engine the launch will take (same tests as conv3d_fwd_impl / conv3d_wgrad_impl; the pointer alignment is the allocator's 256 B)
    const bool packed = q.Cin % 8 == 0, x6 = kget(knobs().engine) == SEGX_ENGINE_BF16X6 && packed;
    const bool same1 = q.sd == 1 && q.sh == 1 && q.sw == 1 && q.ID == q.OD && q.IH == q.OH && q.IW == q.OW && q.OW >= 7 && q.OW % 4 != 0 &&
                       2 * q.pd == q.KD - 1 && 2 * q.ph == q.KH - 1 && 2 * q.pw == q.KW - 1;
    if (wgrad) return conv_splitk(Cout, (int)CK, (int)P, B, x6 && P % 4 == 0 && (((q.OW % 8 == 0 || (q.OW % 4 == 0 && q.sw == 1) || same1) && (!conv_small(Cout) || q.sw == 1 || kget(knobs().conv_x6_wgrad_all) == 2)) || kget(knobs().conv_x6_wgrad_all) == 1));
    return conv_splitk(Cout, (int)P, (int)CK, B, x6 && CK % 4 == 0);
}
/* geom = {Cin, ID, IH, IW, OD, OH, OW, KD, KH, KW, sd, sh, sw, pd, ph, pw} (front pads); splitk > 1: K = Cin*KV split over slabs in
 * workspace (splitk*B*Cout*P floats), reduced deterministically -- for the low-resolution Inception stages whose position grid alone
 * cannot fill the GPU (192 x 588 x 10368: 40 workgroups un-split) */
// x_bs / y_bs: batch strides (floats) of X / Y when they are channel slices of wider NC... tensors; 0 = dense
static int conv3d_fwd_impl(const float* X, const float* W, float* Y, int B, int Cout, const int* geom, int splitk, float* workspace, bool packed,
                           hipStream_t stream, int64_t x_bs = 0, int64_t y_bs = 0) {
    SEGX_REQUIRE(X && W && Y && geom && B > 0 && Cout > 0 && B <= 65535, "segx_conv3d_fwd: bad args");
    const ConvGeom q = make_geom(geom);
    const int64_t P = (int64_t)q.OD * q.OH * q.OW; const int K = q.Cin * q.KD * q.KH * q.KW;
    SEGX_REQUIRE(P > 0 && P < 2147483647LL && K > 0, "segx_conv3d_fwd: bad geometry");
    SEGX_REQUIRE((int64_t)q.Cin * q.ID * q.IH * q.IW < 2147483647LL && q.KD <= 32 && q.KH <= 32 && q.KW <= 32, "segx_conv3d_fwd: sample or window too large");
    SEGX_REQUIRE(!packed || q.Cin % 8 == 0, "segx_conv3d_fwd_packed: Cin = %d is not a multiple of 8", q.Cin);
    if (splitk < 1) splitk = 1;
    SEGX_REQUIRE(splitk == 1 || workspace, "segx_conv3d_fwd: split-K needs a workspace");
    GemmArgs g; g.A = W; g.B = X; g.C = Y;
    g.a_b0 = 0; g.a_m = K; g.b_b0 = x_bs ? x_bs : (int64_t)q.Cin * q.ID * q.IH * q.IW; g.c_b0 = y_bs ? y_bs : (int64_t)Cout * P; g.c_m = P;
    fill_common(g, Cout, (int)P, K, B, splitk, workspace);
    const bool vec = aligned16c(W) && K % 4 == 0, small = conv_small(Cout);
    if (small) g.tiles_m = ceil_div(Cout, CfgCout64::BM);
    dim3 grid(g.tiles_m * g.tiles_n, B, splitk);
#define SEGX_CONV_FWD(V, CFG) do { if (packed) hipLaunchKernelGGL((conv3d_fwd_kernel<V, CFG, true>), grid, dim3(256), 0, stream, g, q); \
                                   else hipLaunchKernelGGL((conv3d_fwd_kernel<V, CFG, false>), grid, dim3(256), 0, stream, g, q); } while (0)
    if (packed && vec && kget(knobs().engine) == SEGX_ENGINE_BF16X6) {          // bf16x6 engine (gemm_x6.h): same tiles, same grid, same split-K slabs
        knobs().x6_launches.fetch_add(1, std::memory_order_relaxed);
        if (small) hipLaunchKernelGGL((conv3d_fwd_x6_kernel<CfgCout64, 4>), grid, dim3(256), 0, stream, g, q);
        else hipLaunchKernelGGL((conv3d_fwd_x6_kernel<Cfg128, 3>), grid, dim3(256), 0, stream, g, q);
    } else if (small && vec) SEGX_CONV_FWD(true, CfgCout64);
    else if (small) SEGX_CONV_FWD(false, CfgCout64);
    else if (vec) SEGX_CONV_FWD(true, Cfg128);
    else SEGX_CONV_FWD(false, Cfg128);
#undef SEGX_CONV_FWD
    int rc = check_launch("segx_conv3d_fwd");
    if (rc || splitk == 1) return rc;
    const int64_t total = g.c_split;
    SEGX_SPLITK_REDUCE((unsigned)i64min(2048, (total + 255) / 256), stream, (const float*)workspace, Y, (const float*)nullptr, Cout, (int)P, 1, splitk, g.c_split,
                       (y_bs ? y_bs : (int64_t)Cout * P), (int64_t)0, (int64_t)P, 1.0f, (int)SEGX_BIAS_NONE, (int64_t)0, (int64_t)0, total, (const float*)nullptr);
    return check_launch("segx_conv3d_fwd/reduce");
}
extern "C" int segx_conv3d_fwd(const float* X, const float* W, float* Y, int B, int Cout, const int* geom, int splitk, float* workspace,
                               void* stream_) {
    return conv3d_fwd_impl(X, W, Y, B, Cout, geom, splitk, workspace, false, (hipStream_t)stream_);
}
/* the same convolution with the filter bank in the packed contraction order of segx_conv3d_pack_weights (Cin % 8 == 0) */
extern "C" int segx_conv3d_fwd_packed(const float* X, const float* Wp, float* Y, int B, int Cout, const int* geom, int splitk, float* workspace,
                                      void* stream_) {
    return conv3d_fwd_impl(X, Wp, Y, B, Cout, geom, splitk, workspace, true, (hipStream_t)stream_);
}
/* Wp[o][c/8][t][c%8]: mode 0 = forward filters (o = Cout index, c = Cin index, value W[o][c][t]); mode 1 = backward-data filters
 * (o = Cin index, c = Cout index, value W[c][o][KV-1-t]); W is always the layer's [Cout][Cin][KV] tensor, C = contracted channels */
extern "C" int segx_conv3d_pack_weights(const float* W, float* Wp, int O, int C, int KV, int mode, void* stream_) {
    SEGX_STREAM; SEGX_REQUIRE(W && Wp && O > 0 && C > 0 && C % 8 == 0 && KV > 0 && (mode == 0 || mode == 1), "segx_conv3d_pack_weights: bad args");
    const int64_t total = (int64_t)O * C * KV;
    hipLaunchKernelGGL(pack_weights_kernel, dim3((unsigned)i64min(4096, (total + 255) / 256)), dim3(256), 0, stream, W, Wp, O, C, KV, mode);
    return check_launch("segx_conv3d_pack_weights");
}
extern "C" int segx_conv3d_flip_weights(const float* W, float* Wt, int Cout, int Cin, int KV, void* stream_) {
    SEGX_STREAM; SEGX_REQUIRE(W && Wt && Cout > 0 && Cin > 0 && KV > 0, "segx_conv3d_flip_weights: bad args");
    const int64_t total = (int64_t)Cout * Cin * KV;
    hipLaunchKernelGGL(flip_weights_kernel, dim3((unsigned)i64min(4096, (total + 255) / 256)), dim3(256), 0, stream, W, Wt, Cout, Cin, KV);
    return check_launch("segx_conv3d_flip_weights");
}
/* dWb[b][Cout][Cin*KV] per-sample weight gradients (sum over b with segx_colsum); workspace: splitk*B*Cout*Cin*KV floats when splitk > 1 */
static int conv3d_wgrad_impl(const float* dY, const float* X, float* dWb, int B, int Cout, const int* geom, int splitk, float* workspace,
                             bool packed, hipStream_t stream, int64_t dy_bs = 0, int64_t x_bs = 0) {
    SEGX_REQUIRE(dY && X && dWb && geom && B > 0 && Cout > 0 && B <= 65535, "segx_conv3d_bwd_weight: bad args");
    const ConvGeom q = make_geom(geom);
    const int64_t P = (int64_t)q.OD * q.OH * q.OW; const int N = q.Cin * q.KD * q.KH * q.KW;
    SEGX_REQUIRE(P > 0 && P < 2147483647LL && N > 0, "segx_conv3d_bwd_weight: bad geometry");
    SEGX_REQUIRE((int64_t)q.Cin * q.ID * q.IH * q.IW < 2147483647LL && q.KD < 1024 && q.KH < 1024 && q.KW < 1024, "segx_conv3d_bwd_weight: sample too large");
    SEGX_REQUIRE(!packed || q.Cin % 8 == 0, "segx_conv3d_bwd_weight_packed: Cin = %d is not a multiple of 8", q.Cin);
    if (splitk < 1) splitk = 1;
    SEGX_REQUIRE(splitk == 1 || workspace, "segx_conv3d_bwd_weight: split-K needs a workspace");
    GemmArgs g; g.A = dY; g.B = X; g.C = dWb;
    g.a_b0 = dy_bs ? dy_bs : (int64_t)Cout * P; g.a_m = P; g.b_b0 = x_bs ? x_bs : (int64_t)q.Cin * q.ID * q.IH * q.IW; g.c_b0 = (int64_t)Cout * N; g.c_m = N;
    fill_common(g, Cout, N, (int)P, B, splitk, workspace);
    const bool vec = aligned16c(dY) && P % 4 == 0, small = conv_small(Cout);
    if (small) g.tiles_m = ceil_div(Cout, CfgCout64::BM);
    dim3 grid(g.tiles_m * g.tiles_n, B, splitk);
#define SEGX_CONV_WG(V, CFG) do { if (packed) hipLaunchKernelGGL((conv3d_wgrad_kernel<V, CFG, true>), grid, dim3(256), 0, stream, g, q); \
                                  else hipLaunchKernelGGL((conv3d_wgrad_kernel<V, CFG, false>), grid, dim3(256), 0, stream, g, q); } while (0)
    // bf16x6 engine: where the eight positions of a thread are eight floats of one input row (OW % 8 == 0: the 56 x 56 stages, 2/3 of the
    // weight-gradient FLOPs of I3D); elsewhere its per-position gather decode costs more VALU time than the six-fold faster matrix instruction
    // saves (r02_a: 63 against 96 TFLOP/s), and the fp32 engine's position-per-thread loader stays.  With unit stride along W the row's eight floats
    // are two 16-byte loads (r02_l: 128-row tile 117 -> 153 TFLOP/s, 64-row tile 65 -> 119 against 92 on the fp32 engine); the strided case
    // (the stride-2 composed stem, 64 filters) keeps eight gathers per row and, on the 64-row tile, stays on the fp32 engine (66 against 83).
    // r05: stride-1 'same' geometry with rows of >= 7 floats that are not a multiple of 4 (the 14- / 7-wide Inception stages of cfg4): the contiguous-octet form
    const bool same1 = q.sd == 1 && q.sh == 1 && q.sw == 1 && q.ID == q.OD && q.IH == q.OH && q.IW == q.OW && q.OW >= 7 && q.OW % 4 != 0 &&
                       2 * q.pd == q.KD - 1 && 2 * q.ph == q.KH - 1 && 2 * q.pw == q.KW - 1;
    const bool fastw = (q.OW % 8 == 0 || (q.OW % 4 == 0 && q.sw == 1) || same1) && g.k_chunk % 8 == 0;          // geometry: the row-of-eight (two-quads, contiguous-octet) loader applies
    if (packed && vec && kget(knobs().engine) == SEGX_ENGINE_BF16X6 && ((fastw && (!small || q.sw == 1 || kget(knobs().conv_x6_wgrad_all) == 2)) || kget(knobs().conv_x6_wgrad_all) == 1)) {
        knobs().x6_launches.fetch_add(1, std::memory_order_relaxed);
        if (fastw && same1) {                                   // the contiguous-octet loader has its own instantiation (in one kernel with the row forms it spilled 12 bytes)
            if (small) hipLaunchKernelGGL((conv3d_wgrad_x6_kernel<CfgCout64, 4, 2>), grid, dim3(256), 0, stream, g, q);
            else hipLaunchKernelGGL((conv3d_wgrad_x6_kernel<Cfg128, 3, 2>), grid, dim3(256), 0, stream, g, q);
        } else if (fastw) {
            if (small) hipLaunchKernelGGL((conv3d_wgrad_x6_kernel<CfgCout64, 4, 1>), grid, dim3(256), 0, stream, g, q);
            else hipLaunchKernelGGL((conv3d_wgrad_x6_kernel<Cfg128, 3, 1>), grid, dim3(256), 0, stream, g, q);
        } else if (small) hipLaunchKernelGGL((conv3d_wgrad_x6_kernel<CfgCout64, 4, 0>), grid, dim3(256), 0, stream, g, q);
        else hipLaunchKernelGGL((conv3d_wgrad_x6_kernel<Cfg128, 3, 0>), grid, dim3(256), 0, stream, g, q);
    } else if (small && vec) SEGX_CONV_WG(true, CfgCout64);
    else if (small) SEGX_CONV_WG(false, CfgCout64);
    else if (vec) SEGX_CONV_WG(true, Cfg128);
    else SEGX_CONV_WG(false, Cfg128);
#undef SEGX_CONV_WG
    int rc = check_launch("segx_conv3d_bwd_weight");
    if (rc || splitk == 1) return rc;
    const int64_t total = g.c_split;
    SEGX_SPLITK_REDUCE((unsigned)i64min(2048, (total + 255) / 256), stream, (const float*)workspace, dWb, (const float*)nullptr, Cout, N, 1, splitk, g.c_split,
                       (int64_t)Cout * N, (int64_t)0, (int64_t)N, 1.0f, (int)SEGX_BIAS_NONE, (int64_t)0, (int64_t)0, total, (const float*)nullptr);
    return check_launch("segx_conv3d_bwd_weight/reduce");
}
extern "C" int segx_conv3d_bwd_weight(const float* dY, const float* X, float* dWb, int B, int Cout, const int* geom, int splitk,
                                      float* workspace, void* stream_) {
    return conv3d_wgrad_impl(dY, X, dWb, B, Cout, geom, splitk, workspace, false, (hipStream_t)stream_);
}
/* the same with the gradient rows in the packed order [Cout][Cin/8][KV][8] (Cin % 8 == 0); segx_conv3d_unpack_wgrad restores [Cout][Cin][KV] */
extern "C" int segx_conv3d_bwd_weight_packed(const float* dY, const float* X, float* dWb, int B, int Cout, const int* geom, int splitk,
                                             float* workspace, void* stream_) {
    return conv3d_wgrad_impl(dY, X, dWb, B, Cout, geom, splitk, workspace, true, (hipStream_t)stream_);
}
/* Channel-slice forms (Inception branches reading / writing slices of a wider NCDHW tensor without a copy): X is the first channel of the slice
 * inside a tensor whose samples lie x_bstride floats apart, likewise Y / dY; every pointer 16-byte aligned; 0 = dense */
extern "C" int segx_conv3d_fwd_packed_bs(const float* X, const float* Wp, float* Y, int B, int Cout, const int* geom, int splitk, float* workspace,
                                         int64_t x_bstride, int64_t y_bstride, void* stream_) {
    SEGX_REQUIRE(x_bstride >= 0 && y_bstride >= 0 && aligned16c(X) && aligned16c(Y), "segx_conv3d_fwd_packed_bs: bad strides / alignment");
    return conv3d_fwd_impl(X, Wp, Y, B, Cout, geom, splitk, workspace, true, (hipStream_t)stream_, x_bstride, y_bstride);
}
extern "C" int segx_conv3d_bwd_weight_packed_bs(const float* dY, const float* X, float* dWb, int B, int Cout, const int* geom, int splitk,
                                                float* workspace, int64_t dy_bstride, int64_t x_bstride, void* stream_) {
    SEGX_REQUIRE(dy_bstride >= 0 && x_bstride >= 0 && aligned16c(X) && aligned16c(dY), "segx_conv3d_bwd_weight_packed_bs: bad strides / alignment");
    return conv3d_wgrad_impl(dY, X, dWb, B, Cout, geom, splitk, workspace, true, (hipStream_t)stream_, dy_bstride, x_bstride);
}
extern "C" int segx_conv3d_unpack_wgrad(const float* dWp, float* dW, int Cout, int Cin, int KV, void* stream_) {
    SEGX_STREAM; SEGX_REQUIRE(dWp && dW && Cout > 0 && Cin > 0 && Cin % 8 == 0 && KV > 0, "segx_conv3d_unpack_wgrad: bad args");
    const int64_t total = (int64_t)Cout * Cin * KV;
    hipLaunchKernelGGL(unpack_wgrad_kernel, dim3((unsigned)i64min(4096, (total + 255) / 256)), dim3(256), 0, stream, dWp, dW, Cout, Cin, KV);
    return check_launch("segx_conv3d_unpack_wgrad");
}

static PoolGeom make_pool(const int* g) {
    PoolGeom q; q.ID = g[0]; q.IH = g[1]; q.IW = g[2]; q.OD = g[3]; q.OH = g[4]; q.OW = g[5]; q.KD = g[6]; q.KH = g[7]; q.KW = g[8];
    q.sd = g[9]; q.sh = g[10]; q.sw = g[11]; q.pd = g[12]; q.ph = g[13]; q.pw = g[14];
    return q;
}
/* geom = {ID, IH, IW, OD, OH, OW, KD, KH, KW, sd, sh, sw, pd, ph, pw}; arg = int32 arg-max index per output (or -1 = padding) */
extern "C" int segx_maxpool3d_fwd(const float* X, float* Y, int* arg, int64_t planes, const int* geom, void* stream_) {
    SEGX_STREAM; SEGX_REQUIRE(X && Y && arg && geom && planes > 0, "segx_maxpool3d_fwd: bad args");
    const PoolGeom q = make_pool(geom);
    const int64_t total = planes * q.OD * q.OH * q.OW;
    SEGX_REQUIRE((int64_t)q.ID * q.IH * q.IW < 2147483647LL && (int64_t)q.OD * q.OH * q.OW < 2147483647LL, "segx_maxpool3d_fwd: plane too large");
    (void)total;
    const int osz = q.OD * q.OH * q.OW;
    const dim3 grid((unsigned)i64min(4096, (osz + 255) / 256), (unsigned)i64min(65535, planes));
    const FastDiv dOHW = make_fastdiv(q.OH * q.OW), dOW = make_fastdiv(q.OW);
    const bool s1w4 = q.sd == 1 && q.sh == 1 && q.sw == 1 && q.KH == 3 && q.KW == 3 && q.pw == 1 && q.OW == q.IW && q.OW % 4 == 0 && q.OW >= 4 &&
                      (int64_t)q.ID * q.IH * q.IW % 4 == 0 && osz % 4 == 0 && aligned16c(X) && aligned16c(Y) && aligned16c(arg);
    // slab-in-LDS form (knob 14: 0 = where the four-outputs-per-thread form does not apply, 1 = wherever it fits, 2 = never)
    const int slab_policy = kget(knobs().pool_slab);
    if (pool_is_s1k3_same(q) && slab_policy != 2 && (slab_policy == 1 || !s1w4) && aligned16c(X)) {
        const int isz = q.ID * q.IH * q.IW, cap = isz <= 2048 ? 2048 : 8192, td = pool_slab_td(q, cap);
        if (td > 0 && planes * ceil_div(q.ID, td) < 2147483647LL) {
            const int nslab = ceil_div(q.ID, td);
            const dim3 sgrid((unsigned)(planes * nslab));
            if (cap == 2048) hipLaunchKernelGGL((maxpool3d_fwd_slab_kernel<2048>), sgrid, dim3(256), 0, stream, X, Y, arg, q, td, nslab, make_fastdiv(q.IH * q.IW), make_fastdiv(q.IW));
            else hipLaunchKernelGGL((maxpool3d_fwd_slab_kernel<8192>), sgrid, dim3(256), 0, stream, X, Y, arg, q, td, nslab, make_fastdiv(q.IH * q.IW), make_fastdiv(q.IW));
            return check_launch("segx_maxpool3d_fwd/slab");
        }
    }
    if (s1w4 && q.KD == 3 && pool_is_s1k3_same(q) && kget(knobs().pool_dslide) != 0 && q.ID >= 4) {           // r05: sliding along depth (knob 15 = 0 keeps the per-slice form)
        const int ow4 = q.OW / 4, td = q.ID >= 16 ? 8 : q.ID >= 8 ? 4 : 2, nchunk = ceil_div(q.ID, td);
        const int64_t per = (int64_t)nchunk * q.OH * ow4;
        const dim3 gridd((unsigned)i64min(4096, (per + 255) / 256), (unsigned)i64min(65535, planes));
        hipLaunchKernelGGL(maxpool3d_fwd_s1w4d_kernel, gridd, dim3(256), 0, stream, X, Y, arg, q, planes, td, nchunk, make_fastdiv(q.OH * ow4), make_fastdiv(ow4));
        return check_launch("segx_maxpool3d_fwd/s1w4d");
    }
    if (s1w4 && (q.KD == 3 || q.KD == 1)) {
        const int ow4 = q.OW / 4;
        const dim3 grid4((unsigned)i64min(4096, (osz / 4 + 255) / 256), (unsigned)i64min(65535, planes));
        if (q.KD == 3) hipLaunchKernelGGL((maxpool3d_fwd_s1w4_kernel<3>), grid4, dim3(256), 0, stream, X, Y, arg, q, planes, make_fastdiv(q.OH * ow4), make_fastdiv(ow4));
        else hipLaunchKernelGGL((maxpool3d_fwd_s1w4_kernel<1>), grid4, dim3(256), 0, stream, X, Y, arg, q, planes, make_fastdiv(q.OH * ow4), make_fastdiv(ow4));
    } else if (q.KD == 3 && q.KH == 3 && q.KW == 3) hipLaunchKernelGGL((maxpool3d_fwd_kernel<3, 3, 3>), grid, dim3(256), 0, stream, X, Y, arg, q, planes, dOHW, dOW);
    else if (q.KD == 1 && q.KH == 3 && q.KW == 3) hipLaunchKernelGGL((maxpool3d_fwd_kernel<1, 3, 3>), grid, dim3(256), 0, stream, X, Y, arg, q, planes, dOHW, dOW);
    else if (q.KD == 2 && q.KH == 2 && q.KW == 2) hipLaunchKernelGGL((maxpool3d_fwd_kernel<2, 2, 2>), grid, dim3(256), 0, stream, X, Y, arg, q, planes, dOHW, dOW);
    else hipLaunchKernelGGL((maxpool3d_fwd_kernel<0, 0, 0>), grid, dim3(256), 0, stream, X, Y, arg, q, planes, dOHW, dOW);
    return check_launch("segx_maxpool3d_fwd");
}
extern "C" int segx_maxpool3d_bwd(const float* dY, const int* arg, float* dX, int64_t planes, const int* geom, const float* addend, void* stream_) {
    SEGX_STREAM; SEGX_REQUIRE(dY && arg && dX && geom && planes > 0, "segx_maxpool3d_bwd: bad args");
    SEGX_REQUIRE(!addend || (reinterpret_cast<uintptr_t>(addend) & 15) == 0, "segx_maxpool3d_bwd: unaligned addend");
    const PoolGeom q = make_pool(geom);
    const int64_t total = planes * q.ID * q.IH * q.IW;
    SEGX_REQUIRE((int64_t)q.ID * q.IH * q.IW < 2147483647LL && (int64_t)q.OD * q.OH * q.OW < 2147483647LL, "segx_maxpool3d_bwd: plane too large");
    const bool bs1w4 = (q.KD == 3 || q.KD == 1) && q.KH == 3 && q.KW == 3 && q.sd == 1 && q.sh == 1 && q.sw == 1 && q.pw == 1 && q.OW == q.IW && q.OH == q.IH && q.IW % 4 == 0 &&
        q.IW >= 4 && (int64_t)q.ID * q.IH * q.IW % 4 == 0 && (int64_t)q.OD * q.OH * q.OW % 4 == 0 && aligned16c(dX) && aligned16c(dY) && aligned16c(arg);
    const bool plain = addend == nullptr;            // the stride-1 forms below do not read `addend`: with one, the generic gather at the end (which honours it) serves
    const int slab_policy = kget(knobs().pool_slab);
    // default: the slab gather where neither the four-cells-per-thread form applies ... (measured, tools/pool_bench.py, profiles/r05_*_pool_bench.txt)
    if (plain && pool_is_s1k3_same(q) && slab_policy != 2 && (slab_policy == 1 || !bs1w4) && aligned16c(dY) && aligned16c(arg)) {
        const int isz = q.ID * q.IH * q.IW, cap = isz <= 2048 ? 2048 : 8192, td = pool_slab_td(q, cap);
        if (td > 0 && planes * ceil_div(q.ID, td) < 2147483647LL) {
            const int nslab = ceil_div(q.ID, td);
            const dim3 sgrid((unsigned)(planes * nslab));
            if (cap == 2048) hipLaunchKernelGGL((maxpool3d_bwd_slab_kernel<2048>), sgrid, dim3(256), 0, stream, dY, arg, dX, q, td, nslab, make_fastdiv(q.IH * q.IW), make_fastdiv(q.IW));
            else hipLaunchKernelGGL((maxpool3d_bwd_slab_kernel<8192>), sgrid, dim3(256), 0, stream, dY, arg, dX, q, td, nslab, make_fastdiv(q.IH * q.IW), make_fastdiv(q.IW));
            return check_launch("segx_maxpool3d_bwd/slab");
        }
    }
    if (plain && bs1w4 && q.KD == 3 && pool_is_s1k3_same(q) && kget(knobs().pool_dslide) != 0 && q.ID >= 4) {
        const int iw4 = q.IW / 4, td = q.ID >= 16 ? 8 : q.ID >= 8 ? 4 : 2, nchunk = ceil_div(q.ID, td);
        const int64_t per = (int64_t)nchunk * q.IH * iw4;
        const dim3 gridd((unsigned)i64min(4096, (per + 255) / 256), (unsigned)i64min(65535, planes));
        hipLaunchKernelGGL(maxpool3d_bwd_s1w4d_kernel, gridd, dim3(256), 0, stream, dY, arg, dX, q, planes, td, nchunk, make_fastdiv(q.IH * iw4), make_fastdiv(iw4));
        return check_launch("segx_maxpool3d_bwd/s1w4d");
    }
    if (plain && bs1w4) {
        const int iw4 = q.IW / 4; const int64_t isz4 = (int64_t)q.ID * q.IH * iw4;
        const dim3 grid4((unsigned)i64min(4096, (isz4 + 255) / 256), (unsigned)i64min(65535, planes));
        if (q.KD == 3) hipLaunchKernelGGL((maxpool3d_bwd_s1w4_kernel<3>), grid4, dim3(256), 0, stream, dY, arg, dX, q, planes, make_fastdiv(q.IH * iw4), make_fastdiv(iw4));
        else hipLaunchKernelGGL((maxpool3d_bwd_s1w4_kernel<1>), grid4, dim3(256), 0, stream, dY, arg, dX, q, planes, make_fastdiv(q.IH * iw4), make_fastdiv(iw4));
        return check_launch("segx_maxpool3d_bwd");
    }
    if (plain && q.KD == 3 && q.KH == 3 && q.KW == 3 && q.sd == 1 && q.sh == 1 && q.sw == 1 && planes <= 65535) {
        const int td = ceil_div(q.ID, MP_TD), th = ceil_div(q.IH, MP_TH), tw = ceil_div(q.IW, MP_TW);
        hipLaunchKernelGGL(maxpool3d_bwd_s1k3_kernel, dim3((unsigned)(td * th * tw), (unsigned)planes), dim3(256), 0, stream, dY, arg, dX, q, th, tw);
        return check_launch("segx_maxpool3d_bwd");
    }
    (void)total;
    const int isz = q.ID * q.IH * q.IW;
    const dim3 grid((unsigned)i64min(4096, (isz + 255) / 256), (unsigned)i64min(65535, planes));
    const FastDiv dIHW = make_fastdiv(q.IH * q.IW), dIW = make_fastdiv(q.IW);
    if (q.sh == 2 && q.sw == 2 && (q.sd == 1 || q.sd == 2) && q.IW % 4 == 0 && isz % 4 == 0 && aligned16c(dX) && q.KW <= 3 && q.KH <= 3 && q.KD <= 3 ) {
        const int iw4 = q.IW / 4;
        const dim3 grid4((unsigned)i64min(4096, (isz / 4 + 255) / 256), (unsigned)i64min(65535, planes));
        hipLaunchKernelGGL(maxpool3d_bwd_s2w4_kernel, grid4, dim3(256), 0, stream, dY, arg, dX, q, planes, make_fastdiv(q.IH * iw4), make_fastdiv(iw4), addend);
    } else if (q.sh == 2 && q.sw == 2 && (q.sd == 1 || q.sd == 2)) hipLaunchKernelGGL((maxpool3d_bwd_kernel<true>), grid, dim3(256), 0, stream, dY, arg, dX, q, planes, dIHW, dIW, addend);
    else hipLaunchKernelGGL((maxpool3d_bwd_kernel<false>), grid, dim3(256), 0, stream, dY, arg, dX, q, planes, dIHW, dIW, addend);
    return check_launch("segx_maxpool3d_bwd");
}
/* wt_ws: Cout*Cin*KV floats of scratch (the filters are re-laid out tap-major once per call) */
extern "C" int segx_conv3d_bwd_data_direct(const float* dY, const float* W, float* dX, float* wt_ws, int B, int Cout, const int* geom, void* stream_) {
    SEGX_STREAM; SEGX_REQUIRE(dY && W && dX && wt_ws && geom && B > 0 && Cout > 0, "segx_conv3d_bwd_data_direct: bad args");
    const ConvGeom q = make_geom(geom);
    SEGX_REQUIRE(q.sd >= 1 && q.sh >= 1 && q.sw >= 1 && (int64_t)B * q.sd * q.sh * q.sw <= 65535 && (int64_t)q.ID * q.IH * q.IW < 2147483647LL,
                 "segx_conv3d_bwd_data_direct: bad strides / sample too large");
    const int KV = q.KD * q.KH * q.KW;
    const int64_t wtot = (int64_t)Cout * q.Cin * KV;
    hipLaunchKernelGGL(tapmajor_weights_kernel, dim3((unsigned)i64min(1024, (wtot + 255) / 256)), dim3(256), 0, stream, W, wt_ws, Cout, q.Cin, KV);
    const int64_t per_class = (int64_t)ceil_div(q.ID, q.sd) * ceil_div(q.IH, q.sh) * ceil_div(q.IW, q.sw);      // upper bound of a class's positions
    dim3 grid((unsigned)((per_class + 255) / 256), (unsigned)(B * q.sd * q.sh * q.sw));
    const float* Wt = wt_ws;
    if (q.KW <= 4 * q.sw) {                       // register-tiled: four positions per thread along W
        const int64_t per4 = (int64_t)ceil_div(q.ID, q.sd) * ceil_div(q.IH, q.sh) * ceil_div(ceil_div(q.IW, q.sw), 4) + ceil_div(q.ID, q.sd) * ceil_div(q.IH, q.sh);
        dim3 grid4((unsigned)((per4 + 255) / 256), (unsigned)(B * q.sd * q.sh * q.sw));
        switch (q.Cin) {
            case 1: hipLaunchKernelGGL((conv3d_bwd_data_direct4_kernel<1>), grid4, dim3(256), 0, stream, dY, Wt, dX, B, Cout, q); break;
            case 2: hipLaunchKernelGGL((conv3d_bwd_data_direct4_kernel<2>), grid4, dim3(256), 0, stream, dY, Wt, dX, B, Cout, q); break;
            case 3: hipLaunchKernelGGL((conv3d_bwd_data_direct4_kernel<3>), grid4, dim3(256), 0, stream, dY, Wt, dX, B, Cout, q); break;
            case 4: hipLaunchKernelGGL((conv3d_bwd_data_direct4_kernel<4>), grid4, dim3(256), 0, stream, dY, Wt, dX, B, Cout, q); break;
            default: return segx::fail(-1, "segx_conv3d_bwd_data_direct: built for Cin <= 4 (the I3D stem), got %d", q.Cin);
        }
        return check_launch("segx_conv3d_bwd_data_direct");
    }
    switch (q.Cin) {
        case 1: hipLaunchKernelGGL((conv3d_bwd_data_direct_kernel<1>), grid, dim3(256), 0, stream, dY, Wt, dX, B, Cout, q); break;
        case 2: hipLaunchKernelGGL((conv3d_bwd_data_direct_kernel<2>), grid, dim3(256), 0, stream, dY, Wt, dX, B, Cout, q); break;
        case 3: hipLaunchKernelGGL((conv3d_bwd_data_direct_kernel<3>), grid, dim3(256), 0, stream, dY, Wt, dX, B, Cout, q); break;
        case 4: hipLaunchKernelGGL((conv3d_bwd_data_direct_kernel<4>), grid, dim3(256), 0, stream, dY, Wt, dX, B, Cout, q); break;
        default: return segx::fail(-1, "segx_conv3d_bwd_data_direct: built for Cin <= 4 (the I3D stem), got %d", q.Cin);
    }
    return check_launch("segx_conv3d_bwd_data_direct");
}
/* in_bridge_to3 composed into the I3D stem (see stem_compose_fwd_kernel): Wc [O][Cc][T] from Ws [O][C3][T], Wb [C3][Cb], bb [C3] (or NULL) */
extern "C" int segx_stem_compose_fwd(const float* Ws, const float* Wb, const float* bb, float* Wc, int O, int C3, int Cb, int Cc, int T, void* stream_) {
    SEGX_STREAM; SEGX_REQUIRE(Ws && Wb && Wc && O > 0 && C3 > 0 && Cb > 0 && Cc > Cb && T > 0, "segx_stem_compose_fwd: bad args");
    const int64_t total = (int64_t)O * Cc * T;
    hipLaunchKernelGGL(stem_compose_fwd_kernel, dim3((unsigned)i64min(4096, (total + 255) / 256)), dim3(256), 0, stream, Ws, Wb, bb, Wc, O, C3, Cb, Cc, T);
    return check_launch("segx_stem_compose_fwd");
}
extern "C" int segx_stem_compose_bwd(const float* dWc, const float* Ws, const float* Wb, const float* bb, float* dWs, float* dWb, float* dbb,
                                     int O, int C3, int Cb, int Cc, int T, void* stream_) {
    SEGX_STREAM; SEGX_REQUIRE(dWc && Ws && Wb && dWs && dWb && O > 0 && C3 > 0 && Cb > 0 && Cc > Cb && T > 0 && (!bb == !dbb), "segx_stem_compose_bwd: bad args");
    const int64_t total = (int64_t)O * C3 * T;
    hipLaunchKernelGGL(stem_compose_bwd_ws_kernel, dim3((unsigned)i64min(4096, (total + 255) / 256)), dim3(256), 0, stream, dWc, Wb, bb, dWs, O, C3, Cb, Cc, T);
    hipLaunchKernelGGL(stem_compose_bwd_wb_kernel, dim3((unsigned)(C3 * (Cb + 1))), dim3(256), 0, stream, dWc, Ws, dWb, dbb, O, C3, Cb, Cc, T);
    return check_launch("segx_stem_compose_bwd");
}
/* x [B][Cb][H][W][D] -> y [B][Cc][D][H][W]: depth first, a constant-one channel at index Cb, zero channels above it */
extern "C" int segx_stem_s2d_input(const float* X, float* Y, int B, int Cb, int H, int W, int D, int U, void* stream_) {
    SEGX_STREAM; SEGX_REQUIRE(X && Y && B > 0 && Cb > 0 && H > 0 && W > 0 && D > 0 && U > 0 && (int64_t)B * 2 * Cb <= 65535 && H <= 65535, "segx_stem_s2d_input: bad args");
    hipLaunchKernelGGL(stem_s2d_input_kernel, dim3((unsigned)(((U + 31) / 32) * ((D + 31) / 32)), (unsigned)H, (unsigned)(B * 2 * Cb)), dim3(256), 0, stream, X, Y, Cb, H, W, D, U);
    return check_launch("segx_stem_s2d_input");
}
extern "C" int segx_bridge_input(const float* X, float* Y, int B, int Cb, int Cc, int H, int W, int D, void* stream_) {
    SEGX_STREAM; SEGX_REQUIRE(X && Y && B > 0 && Cb > 0 && Cc > Cb && H > 0 && W > 0 && D > 0 && (int64_t)B * Cc <= 65535 && H <= 65535, "segx_bridge_input: bad args");
    hipLaunchKernelGGL(bridge_input_kernel, dim3((unsigned)(((W + 31) / 32) * ((D + 31) / 32)), (unsigned)H, (unsigned)(B * Cc)), dim3(256), 0, stream, X, Y, Cb, Cc, H, W, D);
    return check_launch("segx_bridge_input");
}
// The same mask with the batch read in whole D rows (the raw layout's contiguous axis) instead of kd-float segments: one workgroup per (sample, oh, ow) column of
// cells, wave w takes the window rows y = w, w + 4, ..; lane l holds depth positions l, l + 64, .. (D <= 256) of every (y, x) it visits, so a wave's load is one
// contiguous row of D floats (cfg4: 384 bytes; the cell-per-wave form fetched sixteen 16-byte segments 384 bytes apart per instruction: 0.3 TB/s).  The kd depth
// positions of a cell are neighbouring lanes (kd a power of two <= 64): a butterfly folds them.  Four modalities onto three channels (BraTS), else the form above.
__global__ __launch_bounds__(256) void bridge_mask_rows_kernel(const float* __restrict__ X, const float* __restrict__ Wb, const float* __restrict__ bb, float* __restrict__ out,
                                                               int B, int H, int W, int D, int kd, int kh, int kw) {
    constexpr int CB = 4, C3 = 3, DI = 4;
    __shared__ float s_any[4][64 * DI], s_nan[4][64 * DI];
    const int OD = D / kd, OH = H / kh, OW = W / kw;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    int r = blockIdx.x; const int ow = r % OW; r /= OW; const int oh = r % OH; const int b = r / OH;
    float w[C3][CB], bias[C3];
#pragma unroll
    for (int c = 0; c < C3; ++c) {
        bias[c] = bb ? bb[c] : 0.f;
#pragma unroll
        for (int k = 0; k < CB; ++k) w[c][k] = Wb[c * CB + k];
    }
    const int64_t chan = (int64_t)H * W * D;
    const float* xb = X + (int64_t)b * CB * chan;
    float any[DI], nan[DI];
#pragma unroll
    for (int i = 0; i < DI; ++i) { any[i] = 0.f; nan[i] = 0.f; }
    for (int t = wv; t < kh * kw; t += 4) {
        const int y = t / kw, xx = t - y * kw;
        const float* row = xb + ((int64_t)(oh * kh + y) * W + (ow * kw + xx)) * D;
        float v[CB][DI];
#pragma unroll
        for (int k = 0; k < CB; ++k)
#pragma unroll
            for (int i = 0; i < DI; ++i) { const int d = lane + 64 * i; v[k][i] = d < D ? row[(int64_t)k * chan + d] : 0.f; }
#pragma unroll
        for (int i = 0; i < DI; ++i) {
            if (lane + 64 * i >= D) continue;
#pragma unroll
            for (int c = 0; c < C3; ++c) {
                float acc = 0.f;
#pragma unroll
                for (int k = 0; k < CB; ++k) acc = fmaf(w[c][k], v[k][i], acc);
                const float yv = acc + bias[c];
                any[i] = fmaxf(any[i], fabsf(yv) > 0.f ? 1.f : 0.f);
                nan[i] = fmaxf(nan[i], yv != yv ? 1.f : 0.f);
            }
        }
    }
#pragma unroll
    for (int i = 0; i < DI; ++i) { s_any[wv][lane + 64 * i] = any[i]; s_nan[wv][lane + 64 * i] = nan[i]; }
    __syncthreads();
    const int d = threadIdx.x;                                                 // 256 threads = the 256 depth positions this form serves
    float a = fmaxf(fmaxf(s_any[0][d], s_any[1][d]), fmaxf(s_any[2][d], s_any[3][d])), n = fmaxf(fmaxf(s_nan[0][d], s_nan[1][d]), fmaxf(s_nan[2][d], s_nan[3][d]));
    if (d >= OD * kd) { a = 0.f; n = 0.f; }                                    // depth positions behind the last whole cell
    for (int o = 1; o < kd; o <<= 1) { a = fmaxf(a, __shfl_xor(a, o)); n = fmaxf(n, __shfl_xor(n, o)); }
    if (d < OD * kd && (d & (kd - 1)) == 0) out[(((int64_t)b * OD + d / kd) * OH + oh) * OW + ow] = n > 0.f ? 0.f : a;
}

/* X: the raw batch [B][Cb][H][W][D]; Wb [C3][Cb], bb [C3] (or NULL): the input bridge; out [B][D/kd][H/kh][W/kw] = 1 where the bridged image is not identically 0
 * inside the (kd, kh, kw) cell of the permuted (D, H, W) volume */
extern "C" int segx_bridge_mask(const float* X, const float* Wb, const float* bb, float* out, int B, int Cb, int C3, int H, int W, int D, int kd, int kh, int kw, void* stream_) {
    SEGX_STREAM; SEGX_REQUIRE(X && Wb && out && B > 0 && Cb > 0 && C3 > 0 && kd > 0 && kh > 0 && kw > 0 && D >= kd && H >= kh && W >= kw, "segx_bridge_mask: bad args");
    const int64_t cells = (int64_t)B * (D / kd) * (H / kh) * (W / kw);
    if (Cb == 4 && C3 == 3 && D <= 256 && kd <= 64 && (kd & (kd - 1)) == 0 && (int64_t)B * (H / kh) * (W / kw) < 2147483647LL) {
        hipLaunchKernelGGL(bridge_mask_rows_kernel, dim3((unsigned)((int64_t)B * (H / kh) * (W / kw))), dim3(256), 0, stream, X, Wb, bb, out, B, H, W, D, kd, kh, kw);
        return check_launch("segx_bridge_mask/rows");
    }
    hipLaunchKernelGGL(bridge_mask_kernel, dim3((unsigned)i64min(1 << 20, (cells + 3) / 4)), dim3(256), 0, stream, X, Wb, bb, out, B, Cb, C3, H, W, D, kd, kh, kw);
    return check_launch("segx_bridge_mask");
}
extern "C" int segx_nonzero_mask(const float* X, float* out, int B, int C, int D, int H, int W, int kd, int kh, int kw, void* stream_) {
    SEGX_STREAM; SEGX_REQUIRE(X && out && B > 0 && C > 0 && kd > 0 && kh > 0 && kw > 0 && D >= kd && H >= kh && W >= kw, "segx_nonzero_mask: bad args");
    const int64_t total = (int64_t)B * (D / kd) * (H / kh) * (W / kw);
    hipLaunchKernelGGL(nonzero_mask_kernel, dim3((unsigned)i64min(65536, (total + 255) / 256)), dim3(256), 0, stream, X, out, B, C, D, H, W, kd, kh, kw);
    return check_launch("segx_nonzero_mask");
}
extern "C" int segx_label_nhot(const void* labels, float* out, int B, int Cin, int64_t S, int mode, void* stream_) {
    SEGX_STREAM; SEGX_REQUIRE(labels && out && B > 0 && S > 0 && mode >= 0 && mode <= 3 && (mode == 2 || Cin >= (mode == 1 ? 1 : 2)), "segx_label_nhot: bad args");
    const int64_t total = (int64_t)B * S;
    hipLaunchKernelGGL(label_nhot_kernel, dim3((unsigned)i64min(65536, (total + 255) / 256)), dim3(256), 0, stream, labels, out, B, Cin, S, mode);
    return check_launch("segx_label_nhot");
}
