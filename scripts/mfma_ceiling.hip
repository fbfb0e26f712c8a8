// Development aid (not part of libnsr): what does this part sustain on v_mfma_f32_32x32x16_f16?
//
//   hipcc --offload-arch=gfx950 -O3 scripts/mfma_ceiling.hip -o /tmp/mfma_ceiling && /tmp/mfma_ceiling
//
// Bare MFMA loops, no memory traffic: every wave holds its A / B operands in registers and issues MFMAs for a
// few milliseconds.  Variants: operand DATA (zeros / normal-distributed fp16 "hi" values / the (hi, lo) mix of
// the split-fp16 kernels: two thirds of the MFMAs take one operand at 2^-11 of the other's magnitude),
// DEPENDENCY (1, 2 or 4 accumulators used round-robin) and OCCUPANCY (one or two waves per SIMD).  Reported per
// variant: issued PFLOP/s over all CUs, and the effective shader clock = s_memtime ticks / s_memrealtime ticks x
// 100 MHz, sampled by wave 0 of every workgroup (the chip clocks to its power budget: busier pipes and busier
// operand bits run at a lower clock).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include <string>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

// operands: (nthreads, 4 sets, 8 halves) for A and B
// ORDER: which operand registers consecutive MFMAs name (round 6: does operand REUSE between neighbours change what the power
// limit lets through?)  0: A changes every MFMA, B every second (the default);  1: both change every MFMA;  2: A held for four
// MFMAs, B changes;  3: A held for the whole trip, B changes;  4: both held (one A, one B)
template <int ORDER>
__device__ __forceinline__ int a_set(int k) { return ORDER == 2 ? (k >> 2) & 3 : (ORDER >= 3 ? 0 : k & 3); }
template <int ORDER>
__device__ __forceinline__ int b_set(int k) { return ORDER == 0 ? (k >> 1) & 3 : (ORDER == 1 ? (k + (k >> 2)) & 3 : (ORDER == 4 ? 0 : k & 3)); }
template <int NACC, int WAVES_PER_SIMD, int ORDER = 0>
__global__ void __launch_bounds__(256 * WAVES_PER_SIMD) __attribute__((amdgpu_waves_per_eu(WAVES_PER_SIMD, WAVES_PER_SIMD)))
mfma_loop(const h8* __restrict__ a_in, const h8* __restrict__ b_in, int iters, float* __restrict__ sink,
          unsigned long long* __restrict__ clocks) {
  const int tid = blockIdx.x * blockDim.x + threadIdx.x;
  h8 a[4], b[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    a[i] = a_in[tid * 4 + i];
    b[i] = b_in[tid * 4 + i];
  }
  f32x16 acc[NACC];
#pragma unroll
  for (int j = 0; j < NACC; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.0f;
  unsigned long long t0 = 0, r0 = 0;
  if (threadIdx.x == 0) {
    r0 = __builtin_amdgcn_s_memrealtime();
    t0 = __builtin_amdgcn_s_memtime();
  }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < 24; ++k)   // 24 MFMAs per trip, operand sets and accumulators round-robin
      acc[k % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[a_set<ORDER>(k)], b[b_set<ORDER>(k)], acc[k % NACC], 0, 0, 0);
  }
  if (threadIdx.x == 0) {
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    clocks[2 * blockIdx.x] = t1 - t0;
    clocks[2 * blockIdx.x + 1] = r1 - r0;
  }
  float s = 0.0f;
#pragma unroll
  for (int j = 0; j < NACC; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc[j][r];
  if (s == 12345.678f) sink[tid] = s;   // keep the accumulators alive
}

static float gauss() {
  const float u1 = (rand() + 1.0f) / (RAND_MAX + 2.0f), u2 = rand() / (float)RAND_MAX;
  return sqrtf(-2.0f * logf(u1)) * cosf(6.2831853f * u2);
}

template <int NACC, int WPS, int ORDER = 0>
static void run(const char* data_name, int data, int n_cu) {
  const int threads = 256 * WPS, blocks = n_cu, n = threads * blocks;
  std::vector<_Float16> ha((size_t)n * 32), hb((size_t)n * 32);
  for (size_t i = 0; i < ha.size(); ++i) {
    const int set = (int)((i / 8) % 4);
    float va = 0.0f, vb = 0.0f;
    if (data >= 1) {
      va = 0.08f * gauss();          // weight-like
      vb = fmaxf(gauss(), 0.0f);     // relu-activation-like (half of them zero)
    }
    if (data == 2) {                 // split-fp16 mix: sets 1 / 2 play the "lo" parts on the A / B side
      if (set == 1) va *= 4.8828125e-4f;
      if (set == 2) vb *= 4.8828125e-4f;
    }
    ha[i] = (_Float16)va;
    hb[i] = (_Float16)vb;
  }
  h8 *da, *db;
  float* sink;
  unsigned long long* clk;
  CHECK(hipMalloc(&da, ha.size() * 2));
  CHECK(hipMalloc(&db, hb.size() * 2));
  CHECK(hipMalloc(&sink, (size_t)n * 4));
  CHECK(hipMalloc(&clk, (size_t)blocks * 16));
  CHECK(hipMemcpy(da, ha.data(), ha.size() * 2, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(db, hb.data(), hb.size() * 2, hipMemcpyHostToDevice));
  const int iters = 40000 / WPS;
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  float best = 1e30f;
  double clock_ghz = 0.0;
  for (int rep = 0; rep < 3; ++rep) {
    CHECK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL((mfma_loop<NACC, WPS, ORDER>), dim3(blocks), dim3(threads), 0, 0, da, db, iters, sink, clk);
    CHECK(hipEventRecord(e1, 0));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) {
      best = ms;
      std::vector<unsigned long long> hc((size_t)blocks * 2);
      CHECK(hipMemcpy(hc.data(), clk, hc.size() * 8, hipMemcpyDeviceToHost));
      double sum = 0.0;
      for (int bk = 0; bk < blocks; ++bk) sum += (double)hc[2 * bk] / (double)hc[2 * bk + 1];
      clock_ghz = sum / blocks * 0.1;   // s_memrealtime ticks at 100 MHz
    }
  }
  const double flops = (double)blocks * 4 * WPS * iters * 24.0 * 2.0 * 32 * 32 * 16;
  if (ORDER) printf("order %d ", ORDER);
  printf("%-22s acc=%d waves/SIMD=%d : %7.2f ms  %6.3f PFLOP/s issued  clock %.2f GHz  (%.1f cyc/MFMA/SIMD)\n", data_name, NACC, WPS,
         best, flops / (best * 1e-3) / 1e15, clock_ghz, clock_ghz * 1e9 * best * 1e-3 / ((double)iters * 24.0 * WPS));
  CHECK(hipFree(da)); CHECK(hipFree(db)); CHECK(hipFree(sink)); CHECK(hipFree(clk));
}

// `mfma_ceiling hold <data 0|1|2> <seconds>`: keep launching the one-wave-per-SIMD loop on that operand data for that long, so
// that an outside sampler (rocm-smi: power, clocks -- scripts/power_clock_probe.sh) sees a steady state
static void hold(int data, double seconds, int n_cu) {
  const int threads = 256, blocks = n_cu, n = threads * blocks;
  std::vector<_Float16> ha((size_t)n * 32), hb((size_t)n * 32);
  for (size_t i = 0; i < ha.size(); ++i) {
    const int set = (int)((i / 8) % 4);
    float va = 0.0f, vb = 0.0f;
    if (data >= 1) { va = 0.08f * gauss(); vb = fmaxf(gauss(), 0.0f); }
    if (data == 2) { if (set == 1) va *= 4.8828125e-4f; if (set == 2) vb *= 4.8828125e-4f; }
    ha[i] = (_Float16)va;
    hb[i] = (_Float16)vb;
  }
  h8 *da, *db; float* sink; unsigned long long* clk;
  CHECK(hipMalloc(&da, ha.size() * 2)); CHECK(hipMalloc(&db, hb.size() * 2));
  CHECK(hipMalloc(&sink, (size_t)n * 4)); CHECK(hipMalloc(&clk, (size_t)blocks * 16));
  CHECK(hipMemcpy(da, ha.data(), ha.size() * 2, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(db, hb.data(), hb.size() * 2, hipMemcpyHostToDevice));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  double total_ms = 0.0; long launches = 0;
  while (total_ms < seconds * 1e3) {
    CHECK(hipEventRecord(e0, 0));
    for (int k = 0; k < 8; ++k) hipLaunchKernelGGL((mfma_loop<1, 1>), dim3(blocks), dim3(threads), 0, 0, da, db, 40000, sink, clk);
    CHECK(hipEventRecord(e1, 0));
    CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    total_ms += ms; launches += 8;
  }
  const double flops = (double)launches * blocks * 4 * 40000 * 24.0 * 2.0 * 32 * 32 * 16;
  printf("hold data=%d: %.1f s, %.3f PFLOP/s issued\n", data, total_ms * 1e-3, flops / (total_ms * 1e-3) / 1e15);
}

int main(int argc, char** argv) {
  if (argc >= 4 && std::string(argv[1]) == "hold") {
    hipDeviceProp_t q;
    CHECK(hipGetDeviceProperties(&q, 0));
    hold(atoi(argv[2]), atof(argv[3]), q.multiProcessorCount);
    return 0;
  }
  hipDeviceProp_t p;
  CHECK(hipGetDeviceProperties(&p, 0));
  const int n_cu = p.multiProcessorCount;
  printf("%s, %d CUs, clockRate %.0f MHz\n", p.name, n_cu, p.clockRate / 1e3);
  if (argc >= 2 && std::string(argv[1]) == "reuse") {     // operand reuse between neighbouring MFMAs, on realistic data
    for (int rep = 0; rep < 2; ++rep)
      for (int d = 1; d < 3; ++d) {
        const char* nm = d == 1 ? "random hi x hi" : "split-fp16 (hi, lo) mix";
        run<4, 1, 0>(nm, d, n_cu);
        run<4, 1, 1>(nm, d, n_cu);
        run<4, 1, 2>(nm, d, n_cu);
        run<4, 1, 3>(nm, d, n_cu);
        run<4, 1, 4>(nm, d, n_cu);
      }
    return 0;
  }
  const char* names[3] = {"zeros", "random hi x hi", "split-fp16 (hi, lo) mix"};
  for (int d = 0; d < 3; ++d) {
    run<1, 1>(names[d], d, n_cu);
    run<2, 1>(names[d], d, n_cu);
    run<4, 1>(names[d], d, n_cu);
    run<2, 2>(names[d], d, n_cu);
  }
  return 0;
}
