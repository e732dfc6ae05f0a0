// tools/gemm_bf16x6_proto.hip -- standalone prototype for DESIGN.md section 7, item 1 (NOT part of libsegx, not built by build()).
//
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/gemm_bf16x6_proto tools/gemm_bf16x6_proto.hip && /tmp/gemm_bf16x6_proto
//
// C[M][N] = A[M][K] B[N][K]^T in fp32-equivalent arithmetic on the bf16 MFMA: both operands are split ONCE into three bf16 planes
// (x = hi + mid + lo exactly), the tile kernel issues six v_mfma_f32_32x32x16_bf16 per 32x32 block and 16 k (hi.lo, lo.hi, mid.mid,
// hi.mid, mid.hi, hi.hi; the three dropped terms are below the fp32 accumulation error -- tools/mfma_bf16_probe.hip).
// Structure: 128 x 128 x 32 tile, 4 waves (2 x 2 blocks of 32 x 32 each), single LDS buffer (48 KB) with register prefetch of the
// next k-tile, [row][32 k] bf16 rows of 64 B with the 16-B chunk index XOR-swizzled by (row >> 1) & 3 (conflict-free ds_read_b128).
// This is the *first working shape* to measure, not a tuned kernel: the point is the number it prints next to the 115-124 TFLOP/s of
// the fp32-MFMA engine on the same problem sizes.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

union FU { float f; unsigned u; };
__device__ inline unsigned short f2bf(float f) { FU x; x.f = f; x.u += 0x7FFFu + ((x.u >> 16) & 1u); return (unsigned short)(x.u >> 16); }
__device__ inline float bf2f(unsigned short h) { FU x; x.u = (unsigned)h << 16; return x.f; }

// ---- operand split: X [rows][K] fp32 -> planes [3][rows][K] bf16 (hi, mid, lo), 8 consecutive k per thread ---------------
__global__ __launch_bounds__(256) void split3_kernel(const float* __restrict__ X, unsigned short* __restrict__ P, int64_t rows, int K) {
    const int64_t total = rows * (K / 8), plane = rows * (int64_t)K;
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
        const float4 v0 = reinterpret_cast<const float4*>(X)[2 * t], v1 = reinterpret_cast<const float4*>(X)[2 * t + 1];
        const float x[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
        union { unsigned short s[8]; uint4 q; } h, m, l;
        for (int j = 0; j < 8; ++j) {
            h.s[j] = f2bf(x[j]); const float r1 = x[j] - bf2f(h.s[j]);
            m.s[j] = f2bf(r1); const float r2 = r1 - bf2f(m.s[j]);
            l.s[j] = f2bf(r2);
        }
        reinterpret_cast<uint4*>(P)[t] = h.q;
        reinterpret_cast<uint4*>(P + plane)[t] = m.q;
        reinterpret_cast<uint4*>(P + 2 * plane)[t] = l.q;
    }
}

// ---- tile kernel ---------------------------------------------------------------------------------------------------------
constexpr int BM = 128, BN = 128, BK = 32;
__device__ inline int lds_off(int row, int chunk) { return row * 64 + ((chunk ^ ((row >> 1) & 3)) << 4); }     // bytes within one plane tile

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 2))) void gemm_bf16x6_kernel(const unsigned short* __restrict__ PA, const unsigned short* __restrict__ PB,
                                                            float* __restrict__ C, int M, int N, int K) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * 3 * BM * 64];        // [operand][plane][row][64 B] = 48 KB
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
    const int tiles_n = N / BN;
    const int m0 = (blockIdx.x / tiles_n) * BM, n0 = (blockIdx.x % tiles_n) * BN;
    const int64_t planeA = (int64_t)M * K, planeB = (int64_t)N * K;
    // global -> register staging: per operand 3 planes x 2 uint4 per thread, as twelve named registers (arrays ended up in scratch)
    uint4 a00, a01, a10, a11, a20, a21, b00, b01, b10, b11, b20, b21;
    const int row0 = tid >> 2, row1 = (tid + 256) >> 2, chk = tid & 3;
    const int so0 = lds_off(row0, chk), so1 = lds_off(row1, chk);
#define GL1(p, row, RA, RB, k0)                                                                          \
    RA = *reinterpret_cast<const uint4*>(PA + (p) * planeA + (int64_t)(m0 + (row)) * K + (k0) + chk * 8); \
    RB = *reinterpret_cast<const uint4*>(PB + (p) * planeB + (int64_t)(n0 + (row)) * K + (k0) + chk * 8);
#define GLOAD(k0) GL1(0, row0, a00, b00, k0) GL1(0, row1, a01, b01, k0) GL1(1, row0, a10, b10, k0) GL1(1, row1, a11, b11, k0) GL1(2, row0, a20, b20, k0) GL1(2, row1, a21, b21, k0)
#define LS1(p, so, RA, RB)                                                            \
    *reinterpret_cast<uint4*>(lds + (0 * 3 + (p)) * (BM * 64) + (so)) = RA;           \
    *reinterpret_cast<uint4*>(lds + (1 * 3 + (p)) * (BM * 64) + (so)) = RB;
#define LSTORE() LS1(0, so0, a00, b00) LS1(0, so1, a01, b01) LS1(1, so0, a10, b10) LS1(1, so1, a11, b11) LS1(2, so0, a20, b20) LS1(2, so1, a21, b21)
    f32x16 acc[2][2] = {{{0}, {0}}, {{0}, {0}}};
    const int KT = K / BK;
    GLOAD(0)
    for (int kt = 0; kt < KT; ++kt) {
        __syncthreads();
        LSTORE()
        __syncthreads();
        const int kn = (kt + 1 < KT ? kt + 1 : kt) * BK;       // branch-free: the last iteration re-reads its own tile
        GLOAD(kn)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int chunk = 2 * s + (lane >> 5);
            bf16x8 a[2][3], b[2][3];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int p = 0; p < 3; ++p) {
                    a[i][p] = *reinterpret_cast<const bf16x8*>(lds + (0 * 3 + p) * (BM * 64) + lds_off(wm * 64 + i * 32 + (lane & 31), chunk));
                    b[i][p] = *reinterpret_cast<const bf16x8*>(lds + (1 * 3 + p) * (BM * 64) + lds_off(wn * 64 + i * 32 + (lane & 31), chunk));
                }
            // term-major order: consecutive MFMAs update DIFFERENT accumulators (no back-to-back dependent issue); small terms first
#define TERM(PA_, PB_)                                                                                              \
    _Pragma("unroll") for (int i = 0; i < 2; ++i) _Pragma("unroll") for (int j = 0; j < 2; ++j)                      \
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][PA_], b[j][PB_], acc[i][j], 0, 0, 0);
            TERM(0, 2) TERM(2, 0) TERM(1, 1) TERM(0, 1) TERM(1, 0) TERM(0, 0)
        }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int row = m0 + wm * 64 + i * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5), col = n0 + wn * 64 + j * 32 + (lane & 31);
                C[(int64_t)row * N + col] = acc[i][j][reg];
            }
}


// ---- variant 2: BK = 16, double-buffered LDS (2 x 24 KB), ONE barrier per k-tile -----------------------------------------------
// iteration kt: fragments from stage kt & 1 -> 24 MFMAs -> registers (tile kt + 1) into stage (kt + 1) & 1 -> barrier -> global loads of tile kt + 2
constexpr int BK2 = 16;
__device__ inline int lds_off16(int row, int chunk) { return row * 32 + ((chunk ^ ((row >> 2) & 1)) << 4); }   // 32-B rows, 2 chunks

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 2))) void gemm_bf16x6_db_kernel(const unsigned short* __restrict__ PA,
                                                            const unsigned short* __restrict__ PB, float* __restrict__ C, int M, int N, int K) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * 2 * 3 * BM * 32];     // [stage][operand][plane][row][32 B] = 48 KB
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
    const int tiles_n = N / BN;
    const int m0 = (blockIdx.x / tiles_n) * BM, n0 = (blockIdx.x % tiles_n) * BN;
    const int64_t planeA = (int64_t)M * K, planeB = (int64_t)N * K;
    uint4 a0, a1, a2, b0, b1, b2;
    const int row = tid >> 1, chk = tid & 1, so = lds_off16(row, chk);
    const unsigned short* ga = PA + (int64_t)(m0 + row) * K + chk * 8;
    const unsigned short* gb = PB + (int64_t)(n0 + row) * K + chk * 8;
#define GLOAD2(k0)                                                                                                    \
    a0 = *reinterpret_cast<const uint4*>(ga + (k0)); a1 = *reinterpret_cast<const uint4*>(ga + planeA + (k0));        \
    a2 = *reinterpret_cast<const uint4*>(ga + 2 * planeA + (k0)); b0 = *reinterpret_cast<const uint4*>(gb + (k0));    \
    b1 = *reinterpret_cast<const uint4*>(gb + planeB + (k0)); b2 = *reinterpret_cast<const uint4*>(gb + 2 * planeB + (k0));
#define STAGE(st, op, p) (lds + (((st) * 2 + (op)) * 3 + (p)) * (BM * 32))
#define LSTORE2(st)                                                                                                   \
    *reinterpret_cast<uint4*>(STAGE(st, 0, 0) + so) = a0; *reinterpret_cast<uint4*>(STAGE(st, 0, 1) + so) = a1;       \
    *reinterpret_cast<uint4*>(STAGE(st, 0, 2) + so) = a2; *reinterpret_cast<uint4*>(STAGE(st, 1, 0) + so) = b0;       \
    *reinterpret_cast<uint4*>(STAGE(st, 1, 1) + so) = b1; *reinterpret_cast<uint4*>(STAGE(st, 1, 2) + so) = b2;
    f32x16 acc[2][2] = {{{0}, {0}}, {{0}, {0}}};
    const int KT = K / BK2;
    GLOAD2(0)
    LSTORE2(0)
    __syncthreads();
    { const int k1 = (KT > 1 ? 1 : 0) * BK2; GLOAD2(k1) }
    const int half = lane >> 5;
    const int fa0 = lds_off16(wm * 64 + (lane & 31), half), fa1 = lds_off16(wm * 64 + 32 + (lane & 31), half);
    const int fb0 = lds_off16(wn * 64 + (lane & 31), half), fb1 = lds_off16(wn * 64 + 32 + (lane & 31), half);
    for (int kt = 0; kt < KT; ++kt) {
        const int cur = kt & 1;
        bf16x8 a[2][3], b[2][3];
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            a[0][p] = *reinterpret_cast<const bf16x8*>(STAGE(cur, 0, p) + fa0); a[1][p] = *reinterpret_cast<const bf16x8*>(STAGE(cur, 0, p) + fa1);
            b[0][p] = *reinterpret_cast<const bf16x8*>(STAGE(cur, 1, p) + fb0); b[1][p] = *reinterpret_cast<const bf16x8*>(STAGE(cur, 1, p) + fb1);
        }
#define TERM2(PA_, PB_)                                                                                              \
    _Pragma("unroll") for (int i = 0; i < 2; ++i) _Pragma("unroll") for (int j = 0; j < 2; ++j)                       \
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][PA_], b[j][PB_], acc[i][j], 0, 0, 0);
        TERM2(0, 2) TERM2(2, 0) TERM2(1, 1) TERM2(0, 1) TERM2(1, 0) TERM2(0, 0)
        if (cur) { LSTORE2(0) } else { LSTORE2(1) }              // tile kt + 1 (on the last iteration: a harmless re-store)
        __syncthreads();
        const int kn = (kt + 2 < KT ? kt + 2 : KT - 1) * BK2;
        GLOAD2(kn)
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int r = m0 + wm * 64 + i * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5), c = n0 + wn * 64 + j * 32 + (lane & 31);
                C[(int64_t)r * N + c] = acc[i][j][reg];
            }
}

static int g_variant = 1;
static void run_case(int M, int N, int K) {
    if (M % BM || N % BN || K % BK) { printf("[bf16x6] %d x %d x %d skipped: the prototype has no edge handling (multiples of %d x %d x %d only)\n", M, N, K, BM, BN, BK); return; }
    std::vector<float> A((size_t)M * K), B((size_t)N * K);
    srand(11);
    for (auto& v : A) v = (float)rand() / (float)RAND_MAX * 2.f - 1.f;
    for (auto& v : B) v = ((float)rand() / (float)RAND_MAX * 2.f - 1.f) * 0.05f;
    float *dA, *dB, *dC; unsigned short *pA, *pB;
    CHECK(hipMalloc(&dA, A.size() * 4)); CHECK(hipMalloc(&dB, B.size() * 4)); CHECK(hipMalloc(&dC, (size_t)M * N * 4));
    CHECK(hipMalloc(&pA, A.size() * 6)); CHECK(hipMalloc(&pB, B.size() * 6));
    CHECK(hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice)); CHECK(hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1, e2; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1)); CHECK(hipEventCreate(&e2));
    const int reps = 10;
    float ms_split = 0, ms_gemm = 0;
    for (int r = 0; r < reps + 1; ++r) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(split3_kernel, dim3(4096), dim3(256), 0, 0, dA, pA, (int64_t)M, K);
        hipLaunchKernelGGL(split3_kernel, dim3(4096), dim3(256), 0, 0, dB, pB, (int64_t)N, K);
        CHECK(hipEventRecord(e1));
        if (g_variant == 1) hipLaunchKernelGGL(gemm_bf16x6_kernel, dim3((M / BM) * (N / BN)), dim3(256), 0, 0, pA, pB, dC, M, N, K);
        else hipLaunchKernelGGL(gemm_bf16x6_db_kernel, dim3((M / BM) * (N / BN)), dim3(256), 0, 0, pA, pB, dC, M, N, K);
        CHECK(hipEventRecord(e2)); CHECK(hipEventSynchronize(e2));
        float a, b; CHECK(hipEventElapsedTime(&a, e0, e1)); CHECK(hipEventElapsedTime(&b, e1, e2));
        if (r) { ms_split += a; ms_gemm += b; }
    }
    ms_split /= reps; ms_gemm /= reps;
    std::vector<float> C((size_t)M * N);
    CHECK(hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost));
    double err = 0, scale = 0;
    for (int s = 0; s < 512; ++s) {
        const int i = rand() % M, j = rand() % N;
        double ref = 0; for (int k = 0; k < K; ++k) ref += (double)A[(size_t)i * K + k] * (double)B[(size_t)j * K + k];
        err = fmax(err, fabs(C[(size_t)i * N + j] - ref)); scale = fmax(scale, fabs(ref));
    }
    const double fl = 2.0 * M * N * K;
    printf("[bf16x6 v%d] %6d x %5d x %5d: max err / max |C| = %.3e | tile kernel %.3f ms = %.1f TFLOP/s fp32-equivalent | split %.3f ms | together %.1f TFLOP/s\n",
           g_variant, M, N, K, err / scale, ms_gemm, fl / (ms_gemm * 1e-3) / 1e12, ms_split, fl / ((ms_gemm + ms_split) * 1e-3) / 1e12);
    CHECK(hipFree(dA)); CHECK(hipFree(dB)); CHECK(hipFree(dC)); CHECK(hipFree(pA)); CHECK(hipFree(pB));
}

int main() {
    for (g_variant = 1; g_variant <= 2; ++g_variant) {      // 1: single LDS buffer, BK 32, two barriers; 2: double buffer, BK 16, one barrier
        run_case(4096, 4096, 4096);
        run_case(24576, 1792, 1792);       // cfg2: attention projections / per-mode group_linear slab
        run_case(24576, 896, 896);         // cfg2: layer 3
        run_case(8192, 1792, 256);         // short K: prologue / epilogue dominated
    }
    return 0;
}
