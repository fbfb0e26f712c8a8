// Development probe (run through gpurun): does the wave's TRAPSTS.EXCP field accumulate IEEE exceptions of VALU
// conversions on gfx950 without any trap handler / EXCP_EN?  If so, "some fp16 conversion overflowed in this wave" costs
// one s_getreg at the end of a kernel instead of a VALU max per converted pair.
// build + run: hipcc --offload-arch=gfx950 -O2 scripts/trapsts_probe.hip -o /tmp/trapsts_probe && /tmp/trapsts_probe
#include <hip/hip_runtime.h>
#include <cstdio>

__device__ __forceinline__ unsigned read_trapsts() {
  unsigned v;
  asm volatile("s_nop 7\n\ts_getreg_b32 %0, hwreg(HW_REG_TRAPSTS)" : "=s"(v));
  return v;
}
__device__ __forceinline__ void clear_excp() { asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_TRAPSTS, 0, 9), 0" ::: "memory"); }

__global__ void probe(const float* in, unsigned* out, float* sink) {
  const float big = in[0], one = in[1], inf = in[2], small = in[3];
  unsigned r = 0;
  out[r++] = read_trapsts();                       // 0: at wave start
  clear_excp();
  out[r++] = read_trapsts();                       // 1: after clearing
  unsigned pk;
  asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(pk) : "v"(big), "v"(one));
  out[r++] = read_trapsts();                       // 2: cvt_pk overflow
  sink[0] = __uint_as_float(pk);
  clear_excp();
  asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(pk) : "v"(one), "v"(one));
  out[r++] = read_trapsts();                       // 3: cvt_pk exact (nothing expected)
  sink[1] = __uint_as_float(pk);
  clear_excp();
  unsigned lo = 0;
  asm volatile("v_fma_mixlo_f16 %0, %1, %2, 0 op_sel_hi:[0,0,0]" : "+v"(lo) : "v"(big), "v"(one));
  out[r++] = read_trapsts();                       // 4: fma_mix -> f16 overflow
  sink[2] = __uint_as_float(lo);
  clear_excp();
  float d;
  asm volatile("v_sub_f32 %0, %1, %1" : "=v"(d) : "v"(inf));
  out[r++] = read_trapsts();                       // 5: inf - inf (invalid)
  sink[3] = d;
  clear_excp();
  float e;
  asm volatile("v_mul_f32 %0, %1, %1" : "=v"(e) : "v"(small));
  out[r++] = read_trapsts();                       // 6: underflow / inexact
  sink[4] = e;
  clear_excp();
  float m;
  asm volatile("v_max_f32 %0, %1, 0" : "=v"(m) : "v"(d));   // max(NaN, 0)
  out[r++] = read_trapsts();                       // 7
  out[r++] = __float_as_uint(m);                   // 8: what relu makes of NaN
  clear_excp();
  // lanes disagree: only lane 5 overflows
  const float mine = (threadIdx.x == 5) ? big : one;
  asm volatile("v_cvt_pk_f16_f32 %0, %1, %1" : "=v"(pk) : "v"(mine));
  out[r++] = read_trapsts();                       // 9: one lane of 64
  sink[5] = __uint_as_float(pk);
  unsigned mode;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_MODE)" : "=s"(mode));
  out[r++] = mode;                                 // 10
}

int main() {
  float h_in[4] = {70000.0f, 1.0f, __builtin_inff(), 1e-30f};
  float *d_in, *d_sink;
  unsigned* d_out;
  hipMalloc(&d_in, sizeof(h_in));
  hipMalloc(&d_out, 64 * sizeof(unsigned));
  hipMalloc(&d_sink, 64 * sizeof(float));
  hipMemcpy(d_in, h_in, sizeof(h_in), hipMemcpyHostToDevice);
  hipMemset(d_out, 0, 64 * sizeof(unsigned));
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d_in, d_out, d_sink);
  unsigned h_out[16];
  hipMemcpy(h_out, d_out, sizeof(h_out), hipMemcpyDeviceToHost);
  const char* what[] = {"wave start", "after clear", "cvt_pk_f16 overflow", "cvt_pk_f16 exact", "fma_mixlo_f16 overflow",
                        "inf - inf", "1e-30^2", "max(NaN, 0)", "relu(NaN) bits", "one lane overflows", "MODE"};
  for (int i = 0; i < 11; ++i) printf("%-24s 0x%08x  EXCP[8:0] = 0x%03x\n", what[i], h_out[i], h_out[i] & 0x1ff);
  return 0;
}
