// tools/mfma_bf16_probe.hip -- standalone probe for DESIGN.md section 7, item 1 (NOT part of libsegx, not built by build()).
//
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_bf16_probe tools/mfma_bf16_probe.hip && /tmp/mfma_bf16_probe
//
// Answers, on the device, the three questions a bf16-split fp32 GEMM depends on:
//   1. operand layout of v_mfma_f32_32x32x16_bf16 (assumed: lane l holds row/col l & 31 and the 8 consecutive k = 8 * (l >> 5) + j;
//      C/D as the f32 32x32x2 form) -- checked with asymmetric small-integer matrices against a host product;
//   2. issue rate of the bf16 instruction against v_mfma_f32_32x32x2_f32 (expected 32 vs 64 cycles for 8x the FLOPs);
//   3. accuracy of the 6-term and 9-term bf16 splits of an fp32 dot product of length 1792 against fp64, next to the fp32 MFMA.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef __attribute__((ext_vector_type(8))) short bf16x8;     // 8 bf16 in 4 VGPRs
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

union FU { float f; unsigned u; };
__host__ __device__ inline unsigned short f2bf(float f) {     // round to nearest even
    FU x; x.f = f;
    x.u += 0x7FFFu + ((x.u >> 16) & 1u);
    return (unsigned short)(x.u >> 16);
}
__host__ __device__ inline float bf2f(unsigned short h) { FU x; x.u = (unsigned)h << 16; return x.f; }

// ---- 1. layout ------------------------------------------------------------------------------------------------------
// A [32][16], B [32][16] (row-major, k contiguous) -> C[i][j] = sum_k A[i][k] B[j][k]
__global__ void layout_kernel(const unsigned short* A, const unsigned short* B, float* C) {
    const int lane = threadIdx.x, r = lane & 31, kb = 8 * (lane >> 5);
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (short)A[r * 16 + kb + j]; b[j] = (short)B[r * 16 + kb + j]; }
    f32x16 acc = {0};
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
    for (int reg = 0; reg < 16; ++reg) {
        const int row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5), col = lane & 31;
        C[row * 32 + col] = acc[reg];
    }
}

// ---- 2. rate --------------------------------------------------------------------------------------------------------
template <bool BF16>
__global__ __launch_bounds__(256) void rate_kernel(float* out, int iters) {
    f32x16 acc[4] = {{0}, {0}, {0}, {0}};
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (short)(0x3F80 + threadIdx.x + j); b[j] = (short)(0x3F80 + j); }
    const float fa = 1.0f + threadIdx.x * 1e-3f, fb = 0.5f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (BF16) acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[q], 0, 0, 0);
            else acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, acc[q], 0, 0, 0);
        }
    }
    float s = 0.f;
    for (int q = 0; q < 4; ++q) for (int r = 0; r < 16; ++r) s += acc[q][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

// ---- 3. accuracy ------------------------------------------------------------------------------------------------------
// One wave computes a 32x32 block of C = A B^T (A, B [32][K] fp32) three ways: fp32 MFMA, 6-term and 9-term bf16 splits.
__device__ inline void split3(float x, unsigned short& h, unsigned short& m, unsigned short& l) {
    h = f2bf(x); const float r1 = x - bf2f(h);
    m = f2bf(r1); const float r2 = r1 - bf2f(m);
    l = f2bf(r2);
}
__global__ void accuracy_kernel(const float* A, const float* B, int K, float* C32, float* C6, float* C9) {
    const int lane = threadIdx.x, r = lane & 31;
    f32x16 c32 = {0}, c6 = {0}, c9 = {0};
    for (int k0 = 0; k0 < K; k0 += 16) {
        for (int s = 0; s < 8; ++s) {                                   // fp32: 8 steps of k = 2
            const int k = k0 + 2 * s + (lane >> 5);
            c32 = __builtin_amdgcn_mfma_f32_32x32x2f32(A[r * K + k], B[r * K + k], c32, 0, 0, 0);
        }
        bf16x8 ah, am, al, bh, bm, bl;
        const int kb = k0 + 8 * (lane >> 5);
        for (int j = 0; j < 8; ++j) {
            unsigned short h, m, l;
            split3(A[r * K + kb + j], h, m, l); ah[j] = (short)h; am[j] = (short)m; al[j] = (short)l;
            split3(B[r * K + kb + j], h, m, l); bh[j] = (short)h; bm[j] = (short)m; bl[j] = (short)l;
        }
        // small terms first
        c9 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bl, c9, 0, 0, 0);
        c9 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bl, c9, 0, 0, 0);
        c9 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bm, c9, 0, 0, 0);
        f32x16* both[2] = {&c6, &c9};
        for (int w = 0; w < 2; ++w) {
            f32x16 c = *both[w];
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, c, 0, 0, 0);
            *both[w] = c;
        }
    }
    for (int reg = 0; reg < 16; ++reg) {
        const int row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5), col = lane & 31;
        C32[row * 32 + col] = c32[reg]; C6[row * 32 + col] = c6[reg]; C9[row * 32 + col] = c9[reg];
    }
}

int main() {
    // 1. layout
    std::vector<unsigned short> hA(32 * 16), hB(32 * 16);
    std::vector<float> fA(32 * 16), fB(32 * 16), hC(32 * 32);
    srand(7);
    for (int i = 0; i < 32 * 16; ++i) {
        fA[i] = (float)(rand() % 9 - 4); fB[i] = (float)(rand() % 7 - 3) + (i % 16 == 3 ? 0.5f : 0.f);      // asymmetric, exact in bf16
        hA[i] = f2bf(fA[i]); hB[i] = f2bf(fB[i]);
    }
    unsigned short *dA, *dB; float* dC;
    CHECK(hipMalloc(&dA, hA.size() * 2)); CHECK(hipMalloc(&dB, hB.size() * 2)); CHECK(hipMalloc(&dC, hC.size() * 4));
    CHECK(hipMemcpy(dA, hA.data(), hA.size() * 2, hipMemcpyHostToDevice)); CHECK(hipMemcpy(dB, hB.data(), hB.size() * 2, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(layout_kernel, dim3(1), dim3(64), 0, 0, dA, dB, dC);
    CHECK(hipMemcpy(hC.data(), dC, hC.size() * 4, hipMemcpyDeviceToHost));
    int bad = 0;
    for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
        float ref = 0.f; for (int k = 0; k < 16; ++k) ref += fA[i * 16 + k] * fB[j * 16 + k];
        bad += hC[i * 32 + j] != ref;
    }
    printf("[layout] v_mfma_f32_32x32x16_bf16 with lane -> (row l&31, k = 8*(l>>5)+j): %s (%d of 1024 cells differ)\n", bad ? "MISMATCH" : "confirmed", bad);

    // 2. rate
    int cus = 0; CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
    const int blocks = cus * 2, iters = 20000;
    float* dO; CHECK(hipMalloc(&dO, (size_t)blocks * 256 * 4));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int bf = 0; bf < 2; ++bf) {
        for (int rep = 0; rep < 2; ++rep) {
            CHECK(hipEventRecord(e0));
            if (bf) hipLaunchKernelGGL(rate_kernel<true>, dim3(blocks), dim3(256), 0, 0, dO, iters);
            else hipLaunchKernelGGL(rate_kernel<false>, dim3(blocks), dim3(256), 0, 0, dO, iters);
            CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        }
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        const double flops = (double)blocks * 4 /*waves*/ * iters * 4 * 2.0 * 32 * 32 * (bf ? 16 : 2);
        printf("[rate] %s: %.1f TFLOP/s (%d CUs, 8 waves/CU, 4 independent accumulators)\n", bf ? "v_mfma_f32_32x32x16_bf16" : "v_mfma_f32_32x32x2_f32 ",
               flops / (ms * 1e-3) / 1e12, cus);
    }

    // 3. accuracy
    const int K = 1792;
    std::vector<float> A(32 * K), B(32 * K), c32(1024), c6(1024), c9(1024);
    for (auto& v : A) v = (float)rand() / RAND_MAX * 2.f - 1.f;
    for (auto& v : B) v = ((float)rand() / RAND_MAX * 2.f - 1.f) * 0.05f;
    float *gA, *gB, *g32, *g6, *g9;
    CHECK(hipMalloc(&gA, A.size() * 4)); CHECK(hipMalloc(&gB, B.size() * 4)); CHECK(hipMalloc(&g32, 4096)); CHECK(hipMalloc(&g6, 4096)); CHECK(hipMalloc(&g9, 4096));
    CHECK(hipMemcpy(gA, A.data(), A.size() * 4, hipMemcpyHostToDevice)); CHECK(hipMemcpy(gB, B.data(), B.size() * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(accuracy_kernel, dim3(1), dim3(64), 0, 0, gA, gB, K, g32, g6, g9);
    CHECK(hipMemcpy(c32.data(), g32, 4096, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(c6.data(), g6, 4096, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(c9.data(), g9, 4096, hipMemcpyDeviceToHost));
    double e32 = 0, e6 = 0, e9 = 0, scale = 0;
    for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
        double ref = 0; for (int k = 0; k < K; ++k) ref += (double)A[i * K + k] * (double)B[j * K + k];
        scale = fmax(scale, fabs(ref));
        e32 = fmax(e32, fabs(c32[i * 32 + j] - ref)); e6 = fmax(e6, fabs(c6[i * 32 + j] - ref)); e9 = fmax(e9, fabs(c9[i * 32 + j] - ref));
    }
    printf("[accuracy] K = %d, max |error| / max |C| vs fp64:  fp32 MFMA %.3e   bf16 x6 %.3e   bf16 x9 %.3e\n", K, e32 / scale, e6 / scale, e9 / scale);
    return bad != 0;
}
