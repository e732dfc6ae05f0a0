// gemm_skinny.h -- host entry of the streaming skinny weight-gradient kernel (gemm_skinny.hip); called from segx_gemm_f32 (tile SEGX_TILE_SKINNY_NT)
#pragma once
#include <cstdint>
#include <hip/hip_runtime.h>
namespace segx {
// one side <= 32 rows, the other <= 192, K a multiple of 64 and long enough that each of `grid` workgroups streams >= 8 tiles
int skinny_nt_wgs_per_cu(int M, int N);      // resident workgroups per compute unit the launch is sized for (1..3, by the bytes of a tile)
bool skinny_nt_shape_ok(int M, int N, int64_t K, int nbatch, int grid);
// writes `grid` slabs of M x N floats to ws (slab g = the products of workgroup g's run of the (member, 64-position tile) stream); the caller sums them
int launch_skinny_nt(const float* A, const float* B, float* ws, int M, int N, int64_t K, int nb0, int nb1, int64_t a_b0, int64_t a_b1, int64_t a_m,
                     int64_t b_b0, int64_t b_b1, int64_t b_n, int grid, hipStream_t stream);
}  // namespace segx
