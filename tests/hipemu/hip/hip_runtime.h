// tests/hipemu/hip/hip_runtime.h -- TEST INFRASTRUCTURE ONLY.
//
// A tiny single-threaded SIMT emulator so that the product's HIP sources (segtran_amd/csrc/*.hip)
// can be compiled, UNMODIFIED, by the host clang++ and exercised on CPU tensors in the `-m "not gpu"`
// test-suite (there is no GPU in the build container and only 90 GPU-minutes per round).
// Every GPU thread is a ucontext fiber; __syncthreads(), wave shuffles and the f32 MFMA builtins are
// rendezvous points between the fibers of a block / of a 64-lane wave.  Blocks run one after the other.
// It models *semantics* (index maps, MFMA lane layouts, barriers, LDS sharing), not timing, caches or
// memory ordering.  It is never linked into, nor loaded by, the product library.
#pragma once
#include <ucontext.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cstdint>
#include <functional>
#include <vector>
#include <algorithm>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __shared__ static
#define __launch_bounds__(...)
#define SEGX_MIN_WAVES_PER_SIMD(n)      /* register-budget hint: meaningless on the host */
#define SEGX_PIN(x) ((void)0)            /* code-motion fence on a register value: meaningless on the host */
#define SEGX_LDS_BARRIER() __syncthreads()   /* s_waitcnt lgkmcnt(0); s_barrier of the wave-specialised kernels (gemm_x6ws.h) */
#define SEGX_GLOBAL                      /* address_space(1) of the device build */
#define SEGX_WAVE_UNIFORM(x) (x)         /* v_readfirstlane of a value that is uniform over the wave */
#define SEGX_TEAM_SPIN() hipemu::grid_spin_yield()      /* one poll of a team barrier: later blocks of the grid run meanwhile */
#define SEGX_TEAM_SPIN_DONE() hipemu::grid_spin_done()
#define SEGX_TEAM_LOAD(p) (*(p))
#define SEGX_TEAM_STORE(p, v) (*(p) = (v))
#define SEGX_TEAM_ORDER() ((void)0)
#define SEGX_TEAM_RAISE(p) (++*(p))                     /* the system-scope atomic add on the process's error word */
#define SEGX_QUAD_BCAST(v, Q) ((unsigned)__shfl((int)(v), (Q), 4))   /* DPP quad_perm broadcast of quad lane Q */
#define SEGX_QUAD_XOR(v, X) __shfl_xor((v), (X))                      /* DPP quad_perm exchange with lane ^ X (X = 1, 2) */
#define SEGX_LOAD_FENCE() ((void)0)                       /* compiler-only fence of the device build */
#define __constant__ static

struct dim3 { unsigned x, y, z; dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {} };
struct float2 { float x, y; };
struct float4 { float x, y, z, w; };
struct int2 { int x, y; };
struct int4 { int x, y, z, w; };
struct uint2 { unsigned x, y; };
struct uint4 { unsigned x, y, z, w; };
static inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }
static inline float2 make_float2(float x, float y) { return float2{x, y}; }
static inline uint4 make_uint4(unsigned x, unsigned y, unsigned z, unsigned w) { return uint4{x, y, z, w}; }
static inline int4 make_int4(int x, int y, int z, int w) { return int4{x, y, z, w}; }

typedef void* hipStream_t;
typedef int hipError_t;
typedef void* hipEvent_t;
#define hipSuccess 0
static inline hipError_t hipGetLastError() { return 0; }
static inline hipError_t hipPeekAtLastError() { return 0; }
static inline const char* hipGetErrorString(hipError_t) { return "hipemu"; }
static inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) { memset(p, v, n); return 0; }
static inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, int, hipStream_t) { memcpy(d, s, n); return 0; }
#define hipMemcpyDeviceToDevice 3
static inline hipError_t hipStreamSynchronize(hipStream_t) { return 0; }
/* the device queries of the team launches (backbone.hip: team_host / team_occupancy_ok): a 256-CU device, two workgroups per CU, "pinned" = plain memory */
#define hipDeviceAttributeMultiprocessorCount 0
#define hipHostMallocMapped 1
#define hipHostMallocCoherent 2
static inline hipError_t hipGetDevice(int* d) { *d = 0; return 0; }
static inline hipError_t hipDeviceGetAttribute(int* v, int, int) { *v = 256; return 0; }
static inline hipError_t hipHostMalloc(void** p, size_t n, unsigned) { *p = malloc(n); return *p ? 0 : 1; }
static inline hipError_t hipHostGetDevicePointer(void** d, void* h, unsigned) { *d = h; return 0; }
static inline unsigned long long wall_clock64() { static unsigned long long t = 0; return t += 1000; }
#define __builtin_amdgcn_s_sleep(x) ((void)0)
template <class K> static inline hipError_t hipOccupancyMaxActiveBlocksPerMultiprocessor(int* n, K, int, size_t) { *n = 2; return 0; }
static inline hipError_t hipDeviceSynchronize() { return 0; }

namespace hipemu {
struct U3 { unsigned x, y, z; };
struct Fiber { ucontext_t ctx; char* stack; bool done; U3 tid; int lane, wave; };
struct Wave { int nlanes, arrived; unsigned gen; float fa[64], fb[64]; uint64_t ub[64]; unsigned short sa[64][8], sb[64][8]; };
struct State {
    dim3 grid, block; U3 bid; Fiber* cur; ucontext_t sched;
    int nthreads, bar_arrived; unsigned bar_gen;
    std::vector<Fiber> fibers; std::vector<Wave> waves; std::function<void()>* body;
    unsigned long events = 0;        // barrier / wave releases, finished fibers, satisfied cross-block waits: "this block made progress"
    bool spin_seen = false;          // a fiber of the running block is waiting for ANOTHER block (grid_spin_yield)
};
// A block that waits for other blocks (team barrier of the cooperative BatchNorm kernels): its fibers are kept aside while later blocks run.
// __shared__ is static storage here, so such a kernel must not keep shared-memory state across the wait (the product kernels do not).
struct Parked { std::vector<Fiber> fibers; std::vector<Wave> waves; U3 bid; int bar_arrived; unsigned bar_gen; int left; };
inline State S;
constexpr size_t STACK = 256 * 1024;

inline void yield() { Fiber* f = S.cur; swapcontext(&f->ctx, &S.sched); }
inline void trampoline() { (*S.body)(); S.cur->done = true; swapcontext(&S.cur->ctx, &S.sched); }

inline void block_sync() {
    unsigned g = S.bar_gen;
    if (++S.bar_arrived == S.nthreads) { S.bar_arrived = 0; S.bar_gen++; S.events++; }
    else while (S.bar_gen == g) yield();
}
inline Wave& wave() { return S.waves[S.cur->wave]; }
inline void wave_sync() {
    Wave& w = wave(); unsigned g = w.gen;
    if (++w.arrived == w.nlanes) { w.arrived = 0; w.gen++; S.events++; }
    else while (w.gen == g) yield();
}
// one poll of a condition another BLOCK will make true: the scheduler runs later blocks of the grid while this one waits
inline void grid_spin_yield() { S.spin_seen = true; yield(); }
inline void grid_spin_done() { S.events++; }

inline void init_fibers(dim3 block) {
    int nw = (S.nthreads + 63) / 64; S.waves.assign(nw, Wave());
    if ((int)S.fibers.size() < S.nthreads) {
        size_t old = S.fibers.size(); S.fibers.resize(S.nthreads);
        for (size_t i = old; i < S.fibers.size(); ++i) S.fibers[i].stack = (char*)malloc(STACK);
    }
    S.bar_arrived = 0; S.bar_gen = 0;
    for (int w = 0; w < nw; ++w) { S.waves[w].nlanes = std::min(64, S.nthreads - 64 * w); S.waves[w].arrived = 0; S.waves[w].gen = 0; }
    for (int t = 0; t < S.nthreads; ++t) {
        Fiber& f = S.fibers[t]; f.done = false;
        f.tid = U3{t % block.x, (t / block.x) % block.y, t / (block.x * block.y)};
        f.lane = t & 63; f.wave = t >> 6;
        getcontext(&f.ctx); f.ctx.uc_stack.ss_sp = f.stack; f.ctx.uc_stack.ss_size = STACK; f.ctx.uc_link = nullptr;
        makecontext(&f.ctx, (void (*)())trampoline, 0);
    }
}
// run the block whose state is in S until it finishes (true) or every live fiber is stuck behind a wait for another block (false)
inline bool run_current(int& left) {
    long guard = 0;
    while (left > 0) {
        const unsigned long ev0 = S.events; S.spin_seen = false;
        for (int t = 0; t < S.nthreads; ++t) {
            Fiber& f = S.fibers[t]; if (f.done) continue;
            S.cur = &f; swapcontext(&S.sched, &f.ctx);
            if (f.done) { --left; S.events++; }
        }
        if (S.events == ev0) {
            if (S.spin_seen) return false;
            if (++guard > 100000000) { fprintf(stderr, "hipemu: deadlock (barrier divergence?)\n"); abort(); }
        }
    }
    return true;
}
inline void park(std::vector<Parked>& parked, int left) {
    Parked p; p.fibers.swap(S.fibers); p.waves.swap(S.waves); p.bid = S.bid; p.bar_arrived = S.bar_arrived; p.bar_gen = S.bar_gen; p.left = left;
    parked.push_back(std::move(p));
}
// give every parked block a turn; finished ones hand their fiber stacks back to the pool
inline bool retry_parked(std::vector<Parked>& parked, std::vector<std::vector<Fiber>>& pool) {
    bool any = false;
    for (size_t i = 0; i < parked.size();) {
        Parked& p = parked[i];
        std::vector<Fiber> keepf; std::vector<Wave> keepw; keepf.swap(S.fibers); keepw.swap(S.waves);
        S.fibers.swap(p.fibers); S.waves.swap(p.waves); S.bid = p.bid; S.bar_arrived = p.bar_arrived; S.bar_gen = p.bar_gen;
        int left = p.left;
        const unsigned long ev0 = S.events;
        const bool fin = run_current(left);
        if (S.events != ev0) any = true;
        if (fin) { pool.push_back(std::move(S.fibers)); S.fibers.clear(); S.fibers.swap(keepf); S.waves.swap(keepw); parked.erase(parked.begin() + i); }
        else { p.fibers.swap(S.fibers); p.waves.swap(S.waves); p.bar_arrived = S.bar_arrived; p.bar_gen = S.bar_gen; p.left = left; S.fibers.swap(keepf); S.waves.swap(keepw); ++i; }
    }
    return any;
}
template <class F> void launch(dim3 grid, dim3 block, F&& fn) {
    std::function<void()> body = fn;
    S.grid = grid; S.block = block; S.body = &body;
    S.nthreads = block.x * block.y * block.z;
    std::vector<Parked> parked; std::vector<std::vector<Fiber>> pool;
    for (unsigned bz = 0; bz < grid.z; ++bz) for (unsigned by = 0; by < grid.y; ++by) for (unsigned bx = 0; bx < grid.x; ++bx) {
        if (S.fibers.empty() && !pool.empty()) { S.fibers.swap(pool.back()); pool.pop_back(); }
        S.bid = U3{bx, by, bz};
        init_fibers(block);
        int left = S.nthreads;
        if (!run_current(left)) { park(parked, left); retry_parked(parked, pool); }
        else if (!parked.empty()) retry_parked(parked, pool);
    }
    long guard = 0;
    while (!parked.empty()) {
        if (!retry_parked(parked, pool) && ++guard > 1000000) { fprintf(stderr, "hipemu: deadlock (a block waits for another block that never arrives)\n"); abort(); }
    }
    for (auto& v : pool) for (auto& f : v) free(f.stack);     // the stacks of parked blocks; S.fibers keeps its own for the next launch
}

typedef float f32x16_ __attribute__((ext_vector_type(16)));
typedef float f32x4_ __attribute__((ext_vector_type(4)));
// v_mfma_f32_32x32x2_f32: lane l holds A[i=l&31][k=l>>5], B[k=l>>5][j=l&31];
// D reg r of lane l = D[i=(r&3)+8*(r>>2)+4*(l>>5)][j=l&31]; exact k-ordered fmaf chain.
inline f32x16_ mfma_32x32x2f32(float a, float b, f32x16_ c, int, int, int) {
    Wave& w = wave(); int l = S.cur->lane;
    w.fa[l] = a; w.fb[l] = b; wave_sync();
    int j = l & 31;
    for (int r = 0; r < 16; ++r) { int i = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
        c[r] = fmaf(w.fa[i + 32], w.fb[j + 32], fmaf(w.fa[i], w.fb[j], c[r])); }
    wave_sync(); return c;
}
// v_mfma_f32_32x32x16_bf16 (gfx950): lane l holds A[i=l&31][k=8*(l>>5)+e], B[j=l&31][k=8*(l>>5)+e], e = 0..7 (8 bf16 per lane; layout
// confirmed on the device by tools/mfma_bf16_probe.hip); D as the 32x32x2 form.  Products are exact in fp32; k-ordered fp32 accumulation.
typedef short bf16x8_ __attribute__((ext_vector_type(8)));
inline float bf16_bits_to_float(unsigned short h) { unsigned u = (unsigned)h << 16; float f; memcpy(&f, &u, 4); return f; }
inline f32x16_ mfma_32x32x16bf16(bf16x8_ a, bf16x8_ b, f32x16_ c, int, int, int) {
    Wave& w = wave(); int l = S.cur->lane;
    for (int e = 0; e < 8; ++e) { w.sa[l][e] = (unsigned short)a[e]; w.sb[l][e] = (unsigned short)b[e]; }
    wave_sync();
    int j = l & 31;
    for (int r = 0; r < 16; ++r) { int i = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5); float acc = c[r];
        for (int k = 0; k < 16; ++k)
            acc = fmaf(bf16_bits_to_float(w.sa[i + 32 * (k >> 3)][k & 7]), bf16_bits_to_float(w.sb[j + 32 * (k >> 3)][k & 7]), acc);
        c[r] = acc; }
    wave_sync(); return c;
}
// v_mfma_f32_32x32x16_f16 (gfx950): the operand layout of the bf16 form with fp16 elements (products exact in fp32, k-ordered fp32 accumulation)
typedef _Float16 f16x8_ __attribute__((ext_vector_type(8)));
inline f32x16_ mfma_32x32x16f16(f16x8_ a, f16x8_ b, f32x16_ c, int, int, int) {
    Wave& w = wave(); int l = S.cur->lane;
    for (int e = 0; e < 8; ++e) { _Float16 x = a[e], y = b[e]; unsigned short ux, uy; memcpy(&ux, &x, 2); memcpy(&uy, &y, 2); w.sa[l][e] = ux; w.sb[l][e] = uy; }
    wave_sync();
    int j = l & 31;
    auto h2f = [](unsigned short u) { _Float16 h; memcpy(&h, &u, 2); return (float)h; };
    for (int r = 0; r < 16; ++r) { int i = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5); float acc = c[r];
        for (int k = 0; k < 16; ++k) acc = fmaf(h2f(w.sa[i + 32 * (k >> 3)][k & 7]), h2f(w.sb[j + 32 * (k >> 3)][k & 7]), acc);
        c[r] = acc; }
    wave_sync(); return c;
}
// v_mfma_f32_16x16x4_f32: A[i=l&15][k=l>>4], B[k=l>>4][j=l&15]; D reg r = D[i=4*(l>>4)+r][j=l&15].
inline f32x4_ mfma_16x16x4f32(float a, float b, f32x4_ c, int, int, int) {
    Wave& w = wave(); int l = S.cur->lane;
    w.fa[l] = a; w.fb[l] = b; wave_sync();
    int j = l & 15;
    for (int r = 0; r < 4; ++r) { int i = 4 * (l >> 4) + r; float acc = c[r];
        for (int k = 0; k < 4; ++k) acc = fmaf(w.fa[i + 16 * k], w.fb[j + 16 * k], acc);
        c[r] = acc; }
    wave_sync(); return c;
}
template <class T> inline T shfl_idx(T v, int src) {
    static_assert(sizeof(T) <= 8, "shfl width");
    Wave& w = wave(); int l = S.cur->lane; uint64_t u = 0; memcpy(&u, &v, sizeof(T));
    w.ub[l] = u; wave_sync();
    uint64_t r = w.ub[(src >= 0 && src < w.nlanes) ? src : l]; wave_sync();
    T out; memcpy(&out, &r, sizeof(T)); return out;
}
}  // namespace hipemu

#define threadIdx (hipemu::S.cur->tid)
#define blockIdx (hipemu::S.bid)
#define blockDim (hipemu::S.block)
#define gridDim (hipemu::S.grid)
#define warpSize 64
#define hipLaunchKernelGGL(kern, grid, block, shmem, stream, ...) \
    hipemu::launch(dim3(grid), dim3(block), [&]() { kern(__VA_ARGS__); })

static inline void __syncthreads() { hipemu::block_sync(); }
#define __builtin_amdgcn_mfma_f32_32x32x2f32 hipemu::mfma_32x32x2f32
#define __builtin_amdgcn_mfma_f32_16x16x4f32 hipemu::mfma_16x16x4f32
#define __builtin_amdgcn_mfma_f32_32x32x16_bf16 hipemu::mfma_32x32x16bf16
#define __builtin_amdgcn_mfma_f32_32x32x16_f16 hipemu::mfma_32x32x16f16
#define __builtin_amdgcn_s_setprio(x) ((void)0)
#define __builtin_amdgcn_sched_barrier(x) ((void)0)
#define __builtin_amdgcn_sched_group_barrier(mask, n, id) ((void)0)

template <class T> static inline T __shfl(T v, int src, int width = 64) {
    int l = hipemu::S.cur->lane; int base = l - (l % width); return hipemu::shfl_idx(v, base + (src % width)); }
template <class T> static inline T __shfl_xor(T v, int mask, int width = 64) {
    int l = hipemu::S.cur->lane; return hipemu::shfl_idx(v, l ^ mask); }
template <class T> static inline T __shfl_down(T v, unsigned d, int width = 64) {
    int l = hipemu::S.cur->lane; int s = l + (int)d; if ((s / width) != (l / width)) s = l; return hipemu::shfl_idx(v, s); }
template <class T> static inline T __shfl_up(T v, unsigned d, int width = 64) {
    int l = hipemu::S.cur->lane; int s = l - (int)d; if (s < 0 || (s / width) != (l / width)) s = l; return hipemu::shfl_idx(v, s); }

static inline float atomicAdd(float* p, float v) { float o = *p; *p = o + v; return o; }
static inline int atomicAdd(int* p, int v) { int o = *p; *p = o + v; return o; }
static inline unsigned atomicAdd(unsigned* p, unsigned v) { unsigned o = *p; *p = o + v; return o; }
static inline int atomicMax(int* p, int v) { int o = *p; *p = std::max(o, v); return o; }
static inline unsigned atomicMax(unsigned* p, unsigned v) { unsigned o = *p; *p = std::max(o, v); return o; }
static inline int atomicMin(int* p, int v) { int o = *p; *p = std::min(o, v); return o; }
static inline unsigned atomicMin(unsigned* p, unsigned v) { unsigned o = *p; *p = std::min(o, v); return o; }
static inline void __threadfence() {}

#define __expf(x) expf(x)
#define __logf(x) logf(x)
static inline float rsqrtf(float x) { return 1.0f / sqrtf(x); }
static inline float __frcp_rn(float x) { return 1.0f / x; }
static inline float __fdividef(float a, float b) { return a / b; }
static inline int __float_as_int(float f) { int i; memcpy(&i, &f, 4); return i; }
static inline unsigned __float_as_uint(float f) { unsigned i; memcpy(&i, &f, 4); return i; }
static inline float __int_as_float(int i) { float f; memcpy(&f, &i, 4); return f; }
static inline float __uint_as_float(unsigned i) { float f; memcpy(&f, &i, 4); return f; }
static inline unsigned __umulhi(unsigned a, unsigned b) { return (unsigned)(((uint64_t)a * b) >> 32); }
static inline float __ldg(const float* p) { return *p; }
using std::min; using std::max;
