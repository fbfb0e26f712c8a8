// Geometry of the 2-byte training panels (nsr_f16x3_core.h "Training panels"), shared by the chain kernels that write them,
// the weight-gradient kernel that reads them and the host code that carves them (nsr_train.hip).
#pragma once
#include <stdint.h>
#include <hip/hip_runtime.h>

__device__ __host__ __forceinline__ int panel_rows(int panel) { return panel < 9 ? 256 : (panel == 9 ? 128 : 64); }
__device__ __host__ __forceinline__ int64_t panel_rows_before(int panel) {
  return panel <= 9 ? 256 * (int64_t)panel : (panel == 10 ? 2304 + 128 : (panel == 11 ? 2304 + 192 : 2304 + 256));
}
constexpr int kPanelRowBytes = 64;   // 32 points x fp16
__device__ __host__ __forceinline__ int64_t panel_offset_bytes(int64_t n_groups, int panel) {
  return panel_rows_before(panel) * kPanelRowBytes * n_groups;
}
__device__ __host__ __forceinline__ int64_t panel_set_bytes(int64_t n_groups) { return panel_offset_bytes(n_groups, 12); }
// row (within its 64-row panel) of encoding register t (pe: 0..31, de: 0..15) of lane half h: register t is half t & 7 of the
// u32x4 number t >> 3, i.e. unit (t >> 3) & 1 of block t >> 4, bytes 8 ((t >> 2) & 1) .. of the lane's 16
__device__ __host__ __forceinline__ int enc_row(int t, int h) {
  return 32 * (t >> 4) + 16 * ((t >> 3) & 1) + 8 * ((t >> 2) & 1) + 4 * h + (t & 3);
}
