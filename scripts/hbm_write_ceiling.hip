// Development aid (not part of libnsr): what HBM WRITE rate does this part sustain, and does the access pattern of the
// training panels (nsr_f16x3_core.h) reach it?
//
//   hipcc --offload-arch=gfx950 -O3 scripts/hbm_write_ceiling.hip -o /tmp/hbm_write && /tmp/hbm_write
//
// Pure store kernels over a 4 GiB buffer (far beyond L2 + Infinity Cache), 1,024 workgroups of 256 threads, a few passes:
//   stream x4     every thread stores float4, consecutive threads consecutive addresses (the textbook stream)
//   panel dword   the chain kernels' pattern: a wave owns a contiguous 32 KiB "group" (256 rows x 128 B); one store
//                 instruction writes one dword per lane = rows r (lanes 0..31) and r + 4 (lanes 32..63), two whole 128 B
//                 lines; 16 stores per 4 KiB block, blocks in order
// each with default, "nt" and "sc0 sc1" cache policies.  Also a read+write mix is not attempted: the chain kernels only write.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

template <int POLICY>
__device__ __forceinline__ void store_dword(unsigned voff, float v, const float* base, int imm_sel) {
  // imm offsets 0 .. 27 * 128 as in panel_store_r; POLICY 0 default, 1 nt, 2 sc0 sc1
#define ST(OFF)                                                                                                        \
  if (POLICY == 0) asm volatile("global_store_dword %0, %1, %2 offset:" #OFF : : "v"(voff), "v"(v), "s"(base) : "memory");        \
  else if (POLICY == 1) asm volatile("global_store_dword %0, %1, %2 offset:" #OFF " nt" : : "v"(voff), "v"(v), "s"(base) : "memory"); \
  else asm volatile("global_store_dword %0, %1, %2 offset:" #OFF " sc0 sc1" : : "v"(voff), "v"(v), "s"(base) : "memory");
  switch (imm_sel) {
    case 0: ST(0) break;      case 1: ST(128) break;    case 2: ST(256) break;    case 3: ST(384) break;
    case 4: ST(1024) break;   case 5: ST(1152) break;   case 6: ST(1280) break;   case 7: ST(1408) break;
    case 8: ST(2048) break;   case 9: ST(2176) break;   case 10: ST(2304) break;  case 11: ST(2432) break;
    case 12: ST(3072) break;  case 13: ST(3200) break;  case 14: ST(3328) break;  default: ST(3456) break;
  }
#undef ST
}

template <int POLICY>
__global__ void __launch_bounds__(256) panel_writer(float* __restrict__ buf, long n_groups) {
  const int lane = threadIdx.x & 63;
  const long wave = (long)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), n_waves = (long)gridDim.x * 4;
  const unsigned voff = 4u * (unsigned)((lane & 31) + 128 * (lane >> 5));
  for (long g = wave; g < n_groups; g += n_waves) {
    const float* grp = buf + g * (256 * 32);
#pragma unroll 1
    for (int blk = 0; blk < 8; ++blk) {
      const float* b = grp + blk * 32 * 32;
#pragma unroll
      for (int r = 0; r < 16; ++r) store_dword<POLICY>(voff, (float)r, b, r);
    }
  }
}

typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int POLICY>
__global__ void __launch_bounds__(256) stream_writer(f32x4* __restrict__ buf, long n) {
  const long stride = (long)gridDim.x * blockDim.x;
  const f32x4 v = {1.f, 2.f, 3.f, 4.f};
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    if (POLICY == 1) __builtin_nontemporal_store(v, buf + i);
    else buf[i] = v;
  }
}

int main() {
  const size_t bytes = 4ull << 30;
  float* buf;
  CHECK(hipMalloc(&buf, bytes));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  const long n_groups = (long)(bytes / (256 * 32 * 4));
  auto timeit = [&](const char* name, auto launch) {
    launch();
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    const int reps = 5;
    for (int i = 0; i < reps; ++i) launch();
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-28s %8.1f GB/s\n", name, (double)bytes * reps / (ms * 1e-3) / 1e9);
  };
  for (int grid : {1024, 2048, 8192}) {
    printf("grid %d workgroups of 256 threads\n", grid);
    timeit("stream float4", [&] { hipLaunchKernelGGL(stream_writer<0>, dim3(grid), dim3(256), 0, 0, (f32x4*)buf, (long)(bytes / 16)); });
    timeit("stream float4 nt", [&] { hipLaunchKernelGGL(stream_writer<1>, dim3(grid), dim3(256), 0, 0, (f32x4*)buf, (long)(bytes / 16)); });
    timeit("panel dword", [&] { hipLaunchKernelGGL(panel_writer<0>, dim3(grid), dim3(256), 0, 0, buf, n_groups); });
    timeit("panel dword nt", [&] { hipLaunchKernelGGL(panel_writer<1>, dim3(grid), dim3(256), 0, 0, buf, n_groups); });
    timeit("panel dword sc0 sc1", [&] { hipLaunchKernelGGL(panel_writer<2>, dim3(grid), dim3(256), 0, 0, buf, n_groups); });
  }
  CHECK(hipFree(buf));
  return 0;
}
