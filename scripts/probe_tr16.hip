// Development probe (GPU box): (1) the lane map of ds_read_b64_tr_b16 on gfx950, (2) the addressing nsr_wgrad_f16.hip uses on
// a training-panel block stored with the chain kernels' slot rule (nsr_f16x3_core.h: unit_voff), against a plain matrix.
//   hipcc --offload-arch=gfx950 -O2 scripts/probe_tr16.hip -o scripts/bin/probe_tr16 && scripts/bin/probe_tr16
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef __fp16 fp4 __attribute__((__vector_size__(4 * sizeof(__fp16))));
__device__ __forceinline__ h4 tr16(unsigned a) {
  return __builtin_bit_cast(h4, __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp4*)(size_t)a));
}
// (1) lane-linear addresses: out[lane][j]
__global__ void k_map(const unsigned short* in, unsigned short* out) {
  __shared__ __attribute__((aligned(1024))) unsigned short lds[256];
  for (int i = threadIdx.x; i < 256; i += 64) lds[i] = in[i];
  __syncthreads();
  const unsigned base = (unsigned)(size_t)((__attribute__((address_space(3))) char*)lds);
  const h4 v = tr16(base + threadIdx.x * 8);
  // (through a scalar: __builtin_bit_cast of a vector ELEMENT expression reads element 0 whatever the index, hipcc 7.2)
  for (int j = 0; j < 4; ++j) { const _Float16 e = v[j]; out[threadIdx.x * 4 + j] = __builtin_bit_cast(unsigned short, e); }
}
// (2) one 32-feature block of 32 points: X[point][feature] as ids; stored like a chain kernel stores it, read like wgrad reads it
__global__ void k_block(unsigned short* out /* [64 lanes][2 s][2 t][4] */) {
  __shared__ __attribute__((aligned(1024))) unsigned short lds[1024];   // 2 units x 1 KiB
  const int lane = threadIdx.x, m = lane & 31, h = lane >> 5;
  // chain-side: lane (m, h), unit u: halves e = 0..7 = features 16u + 8 (e >> 2) + 4h + (e & 3) of point m -> slot (2m + h) ^ 8u
  for (int u = 0; u < 2; ++u)
    for (int e = 0; e < 8; ++e) {
      const int f = 16 * u + 8 * (e >> 2) + 4 * h + (e & 3);
      lds[u * 512 + (((2 * m + h) ^ (8 * u)) * 8) + e] = (unsigned short)(m * 32 + f);   // id = point * 32 + feature
    }
  __syncthreads();
  const unsigned base = (unsigned)(size_t)((__attribute__((address_space(3))) char*)lds);
  const int g4 = lane >> 4, u = g4 & 1, hh = g4 >> 1, i16 = lane & 15;
  const unsigned c0 = (unsigned)(u * 1024 + 256 * hh + 32 * (i16 >> 2) + 16 * (i16 & 1) + 8 * ((i16 >> 1) & 1));
  const unsigned rd_t0 = c0 + 128u * (unsigned)u, rd_t1 = c0 + 128u * (unsigned)(1 - u);
  for (int s = 0; s < 2; ++s)
    for (int t = 0; t < 2; ++t) {
      const h4 v = tr16(base + (t ? rd_t1 : rd_t0) + 512 * s);
      for (int j = 0; j < 4; ++j) { const _Float16 e = v[j]; out[((lane * 2 + s) * 2 + t) * 4 + j] = __builtin_bit_cast(unsigned short, e); }
    }
}
int main() {
  unsigned short *din, *dout;
  hipMalloc(&din, 512);
  hipMalloc(&dout, 64 * 16 * 2);
  std::vector<unsigned short> in(256), out(1024);
  for (int i = 0; i < 256; ++i) in[i] = (unsigned short)i;
  hipMemcpy(din, in.data(), 512, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k_map, dim3(1), dim3(64), 0, 0, din, dout);
  hipMemcpy(out.data(), dout, 512, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int l = 0; l < 64; ++l)
    for (int j = 0; j < 4; ++j) {
      const int want = (l & 15) + j * 16 + (l >> 4) * 64;   // guide: column l & 15 of the group's 4 x 16 row-major block
      if (out[l * 4 + j] != want) { if (bad < 8) printf("map: lane %d elem %d got %d want %d\n", l, j, out[l * 4 + j], want); ++bad; }
    }
  printf("tr16 lane map: %s (%d mismatches)\n", bad ? "DIFFERENT FROM THE GUIDE" : "as the guide states", bad);
  hipLaunchKernelGGL(k_block, dim3(1), dim3(64), 0, 0, dout);
  hipMemcpy(out.data(), dout, 2048, hipMemcpyDeviceToHost);
  int bad2 = 0;
  for (int l = 0; l < 64; ++l)
    for (int s = 0; s < 2; ++s)
      for (int t = 0; t < 2; ++t)
        for (int j = 0; j < 4; ++j) {
          const int feat = l & 31, hh = l >> 5, pt = 16 * s + 8 * hh + 4 * t + j;
          const int got = out[((l * 2 + s) * 2 + t) * 4 + j], want = pt * 32 + feat;
          if (got != want) { if (bad2 < 8) printf("block: lane %d s %d t %d j %d got (pt %d, f %d) want (pt %d, f %d)\n", l, s, t, j, got >> 5, got & 31, pt, feat); ++bad2; }
        }
  printf("wgrad block addressing: %s (%d mismatches)\n", bad2 ? "WRONG" : "ok", bad2);
  return (bad || bad2) ? 1 : 0;
}
