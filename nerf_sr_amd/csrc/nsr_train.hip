// Training step through the render path: include/nsr_train.h.
//
// Replaces NeRFDownXModel.optimize_parameters (models/nerf_downX_model.py:398-408): train-mode forward_rays
// (:280-313), the s^2 means (:326-348), the two MSE losses (:355-362), autograd's backward through the
// compositing and both MLPs, and torch.optim.Adam (:201-204).
//
// Shape of the computation on MI355X.  A training batch is small next to a rendered frame (thousands of rays), and
// its activations have to be kept for the backward pass anyway, so the MLP runs layer by layer on one fp32-MFMA GEMM
// kernel (nsr_gemm.hip) whose epilogue fuses bias / ReLU / sigmoid / the ReLU mask of the backward pass.  The kernel
// takes either memory orientation of each operand, so all three products of a linear layer (forward, input gradient,
// weight gradient) read the row-major (P, C) activations and the nn.Linear weights as they lie: nothing is
// transposed and nothing is stored twice.  The weight gradient is a split-K GEMM over the sample points with a
// deterministic second-pass reduction (no atomics: results are run-to-run identical).
// Layers are padded to MFMA-friendly shapes once per step (63 -> 64 input channels, the skip concat as
// [pe64 | h4], the density head stacked under xyz_encoding_final as one 288-row layer, the colour head as 32 rows);
// the gradients are scattered back to the nn.Linear shapes by the reduction kernel.
//
// CHAIN path (precision NSR_F16X3, the default there): the network is a per-point chain, so its forward pass and its
// input gradients do not need per-layer GEMMs at all.  The forward pass is the inference kernel (nsr_mlp_f16.hip, TRAIN)
// that additionally keeps every layer's activation as the fp16 operand it makes anyway; the input gradients are one launch
// of the backward chain (nsr_train_chain.hip), which keeps its fp16 operands likewise.  Both write 2-byte "panels" in
// 1 KiB units (nsr_f16x3_core.h) that the weight-gradient kernel (nsr_wgrad_f16.hip) copies into LDS as they are and
// multiplies on ONE fp16 MFMA per product; it also sums the bias gradients.  precision NSR_F16X3_GEMM selects the
// layer-by-layer path above with split-fp16 forward products (the A/B partner of profiles/; NSR_FP32 always takes it).
// Per-ray stages (sampling, compositing, resampling) are the inference kernels (nsr_rays.hip / nsr_render.hip);
// the compositing backward is a one-wave-per-ray kernel like its forward.
#include "nsr_common.h"
#include "nsr_gemm.h"
#include "nsr_train_chain.h"
#include "nsr_panels.h"
#include "../../include/nsr_train.h"

using namespace nsr;

namespace {

constexpr int kW = 256, kPe = 64, kX5 = 320, kGs = 288, kDirOut = 128, kRgbPad = 32;
constexpr int kSigmaCol = 256, kDeCol = 260;     // columns of the [g | sigma | 0 0 0 | de27 | 0] buffer
#ifndef NSR_MAX_SPLITS
#define NSR_MAX_SPLITS 256   // one workgroup per CU.  Same box, 2,048-ray step: 128 -> 6.15 ms, 256 -> 5.30 ms, 512 -> 5.60 ms
#endif
constexpr int kMaxSplits = NSR_MAX_SPLITS;
constexpr int64_t kPartialFloats = (int64_t)kGs * kX5;   // >= every padded weight-gradient shape
constexpr int kChainSlots = 14, kChainRowSlots = 12;     // chain path: partial sums of a network's 14 weight-gradient
                                                        // products and of its bias row sums, all alive until ONE finishing launch

// state_dict indices (nsr.h): layer i (1..8) weight = 2 (i - 1), bias = 2 (i - 1) + 1
constexpr int kFinalW = 16, kFinalB = 17, kDirW = 18, kDirB = 19, kSigmaW = 20, kSigmaB = 21, kRgbW = 22, kRgbB = 23;
__host__ __device__ constexpr int64_t tensor_numel(int t) {
  switch (t) {
    case 0: return 256 * 63;
    case 8: return 256 * 319;
    case kFinalW: return 256 * 256;
    case kDirW: return 128 * 283;
    case kDirB: return 128;
    case kSigmaW: return 256;
    case kSigmaB: return 1;
    case kRgbW: return 3 * 128;
    case kRgbB: return 3;
    default: return (t & 1) ? 256 : 256 * 256;
  }
}

inline int64_t align64(int64_t n) { return (n + 63) & ~(int64_t)63; }   // floats -> 256-byte granules

// ---------------------------------------------------------------------------------------------------------
// small kernels
// ---------------------------------------------------------------------------------------------------------
// dst[(r0 + i) * ld + c0 + j] = src[i][col0 + j]  (or the transpose: dst[(r0 + j) * ld + c0 + i])
__global__ void place_kernel(float* __restrict__ dst, int dst_ld, int r0, int c0, const float* __restrict__ src,
                             int src_ld, int rows, int cols, int col0, int transpose) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= rows * cols) return;
  const int i = idx / cols, j = idx % cols;
  const float v = src[(int64_t)i * src_ld + col0 + j];
  if (transpose) dst[(int64_t)(r0 + j) * dst_ld + c0 + i] = v;
  else dst[(int64_t)(r0 + i) * dst_ld + c0 + j] = v;
}

// E1 + cast_rays for the training layout: one thread per sample point.
//   x5 (P, 320) columns 0..63  = [pe63, 0]
//   gs (P, 288) columns 257..287 = [0 0 0, de27, 0]
__global__ void __launch_bounds__(256) encode_train_kernel(const float* __restrict__ rays, int stride,
                                                           const float* __restrict__ z, int64_t P, int N,
                                                           float* __restrict__ x5, float* __restrict__ gs) {
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= P) return;
  const NsrRay q = nsr_load_ray(rays, p / N, stride);
  const float zk = z[p];
  float pe[64];
#pragma unroll
  for (int c = 0; c < 3; ++c) pe[c] = __fadd_rn(q.o[c], __fmul_rn(zk, q.d[c]));   // cast_rays, models/utils.py:5-14
#pragma unroll
  for (int f = 0; f < 10; ++f)
#pragma unroll
    for (int c = 0; c < 3; ++c) nsr_sincos(ldexpf(pe[c], f), pe[3 + 6 * f + c], pe[3 + 6 * f + 3 + c]);
  pe[63] = 0.0f;
  float4* row = reinterpret_cast<float4*>(x5 + p * kX5);
#pragma unroll
  for (int i = 0; i < 16; ++i) row[i] = make_float4(pe[4 * i], pe[4 * i + 1], pe[4 * i + 2], pe[4 * i + 3]);
  float de[31];   // columns 257..287
  de[0] = de[1] = de[2] = 0.0f;
#pragma unroll
  for (int c = 0; c < 3; ++c) de[3 + c] = q.v[c];
#pragma unroll
  for (int f = 0; f < 4; ++f)
#pragma unroll
    for (int c = 0; c < 3; ++c) nsr_sincos(ldexpf(q.v[c], f), de[6 + 6 * f + c], de[6 + 6 * f + 3 + c]);
  de[30] = 0.0f;
#pragma unroll
  for (int c = 0; c < 31; ++c) gs[p * kGs + 257 + c] = de[c];
}

// N1: sigma + noise * std (models/utils.py:199-212); noise == nullptr copies
__global__ void sigma_noise_kernel(const float* __restrict__ sigma, int sigma_stride, const float* __restrict__ noise,
                                   float std_, int64_t P, float* __restrict__ out) {
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= P) return;
  const float s = sigma[p * sigma_stride];
  out[p] = noise ? __fadd_rn(s, __fmul_rn(noise[p], std_)) : s;
}

// A1 + MSE: one thread per LR pixel.  lr = mean over the s2 sub-rays; loss partial per block; gC = dL/d(comp rgb)
// of every sub-ray = 2 lambda (lr - target) / (3 N_lr_total) / s2.
// Optional variance losses of the same function (comp_low_res_output, models/nerf_downX_model.py:332-336, 349-353; added to
// loss_tot at :374-378): lambda_var * SUM over the LR pixels and channels of torch.var (unbiased: / (s2 - 1)) of the s2 sub-ray
// colours, and lambda_dvar * SUM over the LR pixels of torch.var of the s2 sub-ray depths / far.  Their gradients join gC
// (2 lambda_var (x_k - mean) / (s2 - 1)) and form a per-ray depth gradient g_depth (2 lambda_dvar (d_k / far - mean) /
// ((s2 - 1) far)) that the compositing backward adds through depth = sum_k w_k z_k.
// block_sums: [0, nblk) squared errors, [nblk, 2 nblk) colour variances, [2 nblk, 3 nblk) depth variances.
__global__ void __launch_bounds__(256) lr_loss_kernel(const float* __restrict__ comp, const float* __restrict__ target,
                                                      int64_t n_lr, int s2, double scale, float lambda,
                                                      float* __restrict__ lr_out, float* __restrict__ g_comp,
                                                      double* __restrict__ block_sums, float lambda_var, float lambda_dvar,
                                                      float far, const float* __restrict__ depth, float* __restrict__ g_depth) {
  __shared__ double red[3][256];
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  double sq = 0.0, var_rgb = 0.0, var_d = 0.0;
  if (i < n_lr) {
    for (int c = 0; c < 3; ++c) {
      float acc = 0.0f;
      for (int k = 0; k < s2; ++k) acc = __fadd_rn(acc, comp[(i * s2 + k) * 3 + c]);
      const float lr = __fdiv_rn(acc, (float)s2);
      lr_out[i * 3 + c] = lr;
      const float d = __fsub_rn(lr, target[i * 3 + c]);
      sq += (double)d * (double)d;
      const float gc = (float)(2.0 * (double)lambda * (double)d * scale / (double)s2);
      if (lambda_var != 0.0f) {       // wave-uniform
        double ss = 0.0;
        for (int k = 0; k < s2; ++k) {
          const double e = (double)comp[(i * s2 + k) * 3 + c] - (double)lr;
          ss += e * e;
          g_comp[(i * s2 + k) * 3 + c] = gc + (float)(2.0 * (double)lambda_var * e / (double)(s2 - 1));
        }
        var_rgb += ss / (double)(s2 - 1);
      } else {
        for (int k = 0; k < s2; ++k) g_comp[(i * s2 + k) * 3 + c] = gc;
      }
    }
    if (g_depth) {
      if (lambda_dvar != 0.0f) {
        double m = 0.0;
        for (int k = 0; k < s2; ++k) m += (double)__fdiv_rn(depth[i * s2 + k], far);
        m /= (double)s2;
        double ss = 0.0;
        for (int k = 0; k < s2; ++k) {
          const double e = (double)__fdiv_rn(depth[i * s2 + k], far) - m;
          ss += e * e;
          g_depth[i * s2 + k] = (float)(2.0 * (double)lambda_dvar * e / ((double)(s2 - 1) * (double)far));
        }
        var_d = ss / (double)(s2 - 1);
      } else {
        for (int k = 0; k < s2; ++k) g_depth[i * s2 + k] = 0.0f;
      }
    }
  }
  red[0][threadIdx.x] = sq;
  red[1][threadIdx.x] = var_rgb;
  red[2][threadIdx.x] = var_d;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o)
      for (int q = 0; q < 3; ++q) red[q][threadIdx.x] += red[q][threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x < 3) block_sums[(int64_t)threadIdx.x * gridDim.x + blockIdx.x] = red[threadIdx.x][0];
}
// losses[slot] = lambda * scale * (sum of squared errors over the passes so far)   (single thread block: deterministic order);
// var_losses (optional): [slot] = lambda_var * sum of the colour variances, [2 + slot] = lambda_dvar * sum of the depth variances
__global__ void __launch_bounds__(256) loss_finish_kernel(const double* __restrict__ block_sums, int n, double scale,
                                                          float lambda, float* __restrict__ losses, int slot,
                                                          double* __restrict__ carry, float lambda_var, float lambda_dvar,
                                                          float* __restrict__ var_losses) {
  __shared__ double red[3][256];
  for (int q = 0; q < 3; ++q) {
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) s += block_sums[(int64_t)q * n + i];
    red[q][threadIdx.x] = s;
  }
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o)
      for (int q = 0; q < 3; ++q) red[q][threadIdx.x] += red[q][threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    carry[slot] += red[0][0];                                       // sum of squared errors over the passes so far
    losses[slot] = (float)((double)lambda * carry[slot] * scale);   // scale = 1 / (3 N_lr)
    carry[2 + slot] += red[1][0];
    carry[4 + slot] += red[2][0];
    if (var_losses) {
      var_losses[slot] = (float)((double)lambda_var * carry[2 + slot]);
      var_losses[2 + slot] = (float)((double)lambda_dvar * carry[4 + slot]);
    }
  }
}

// --gamma_correct while training: the per-sample colours of a pass (rgb4: (P, 4), columns 0..2) become pow(rgb, 1 / 2.2)
__global__ void __launch_bounds__(256) gamma_points_kernel(float* __restrict__ rgb4, int64_t P) {
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= P) return;
  float4 v = reinterpret_cast<float4*>(rgb4)[p];
  v.x = nsr_gamma(v.x); v.y = nsr_gamma(v.y); v.z = nsr_gamma(v.z);
  reinterpret_cast<float4*>(rgb4)[p] = v;
}
// g d(colour) / d(pre-activation) of the colour head from the colour y it produced: the slope is y (1 - y) for the sigmoid
// and y (1 - y^2.2) / 2.2 for the gamma-corrected sigmoid y = s^(1 / 2.2).  (The plain form keeps rounds 2-4's operation
// order, (g y) (1 - y): training trajectories are compared bit for bit across builds.)
// head: 0 = sigmoid, 1 = gamma-corrected sigmoid, 2 = none (--color_activation none: slope 1)
__device__ __forceinline__ float colour_head_bwd(float g, float y, int head) {
  if (head == 2) return g;
  return head == 1 ? g * y * (1.0f - powf(y, 2.2f)) * (1.0f / 2.2f) : g * y * (1.0f - y);
}

// Backward of V1 (models/rendering.py:75-111) w.r.t. the point colours and densities, given dL/d(comp rgb).
//   w_k = alpha_k T_k, T_k = prod_{j<k} f_j, f_j = 1 - alpha_j + 1e-10, alpha_k = 1 - exp(-delta_k relu(sigma_k))
//   gw_k = gC . rgb_k (- sum(gC) with a white background: comp += 1 - sum_k w_k) (+ g_depth z_k: depth = sum_k w_k z_k, when a
//   depth-variance loss is on)
//   dL/dalpha_k = gw_k T_k - (sum_{i>k} gw_i w_i) / f_k
//   dL/dsigma_k = dL/dalpha_k * delta_k exp(-delta_k relu(sigma_k)) * [sigma_k > 0];   dL/drgb_k = gC w_k
// and through the sigmoid of the colour head: d(rgb_pre) = d(rgb) * rgb (1 - rgb); under --gamma_correct the colours held
// are y = s^(1/2.2), s = sigmoid(pre), and dy/d(pre) = s^(1/2.2 - 1) s (1 - s) / 2.2 = y (1 - y^2.2) / 2.2.
// `white`: the training entry points' option word (include/nsr_train.h).  NSR_SIGMA_SOFTPLUS: density log(1 + exp(sigma - 1)),
// whose slope sigmoid(sigma - 1) replaces [sigma > 0]; NSR_TRAIN_COLOR_NONE: no colour activation, slope 1.
// Outputs in the GEMM path's training layout: d_rgb (P, 32) columns 0..2 (3..31 zeroed); d_sigma into column 256 of
// g1 (P, 288) (257..287 zeroed).  COMPACT (chain path): one float4 per point, d4[p] = (d_rgb_pre 0..2, d_sigma) -- 16 bytes
// instead of 256 written per point, and one 16-byte read per point for the backward chain instead of two strided ones.
template <int K, bool COMPACT>
__global__ void __launch_bounds__(256) composite_bwd_kernel(const float* __restrict__ rgb4, const float* __restrict__ sigma,
                                                            const float* __restrict__ z, const float* __restrict__ g_comp,
                                                            int64_t R, int N, int white,
                                                            float* __restrict__ d_rgb, float* __restrict__ g1,
                                                            const float* __restrict__ g_depth, float* __restrict__ bias_part) {
  const int lane = threadIdx.x & 63;
  // COMPACT (chain path): `g1` carries the backward chain's ten gmax words, cleared here -- the kernel that runs right in front
  // of the chain -- instead of by a memset launch of their own (round 6)
  if (COMPACT && g1 && blockIdx.x == 0 && threadIdx.x < 10) reinterpret_cast<unsigned*>(g1)[threadIdx.x] = 0u;
  const int64_t r = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (r >= R) return;
  const int64_t base = r * N;
  const float gc0 = g_comp[r * 3 + 0], gc1 = g_comp[r * 3 + 1], gc2 = g_comp[r * 3 + 2];
  const float gd = g_depth ? g_depth[r] : 0.0f;
  const int head = (white & NSR_TRAIN_GAMMA_CORRECT) ? 1 : ((white & NSR_TRAIN_COLOR_NONE) ? 2 : 0);
  const bool softplus = (white & NSR_SIGMA_SOFTPLUS) != 0;
  float zk[K], sg[K], c0[K], c1[K], c2[K];
#pragma unroll
  for (int i = 0; i < K; ++i) {
    const int k = lane * K + i;
    const int64_t p = base + (k < N ? k : N - 1);
    zk[i] = z[p];
    sg[i] = sigma[p];
    c0[i] = rgb4[p * 4 + 0];
    c1[i] = rgb4[p * 4 + 1];
    c2[i] = rgb4[p * 4 + 2];
  }
  const float z_next_lane = __shfl_down(zk[0], 1, 64);
  float alpha[K], ex[K], dl[K], ff[K];
  double pl[K];
  double run = 1.0;
#pragma unroll
  for (int i = 0; i < K; ++i) {
    const int k = lane * K + i;
    const float zn = (i + 1 < K) ? zk[(i + 1 < K) ? i + 1 : i] : z_next_lane;
    const float delta = (k >= N - 1) ? 1e10f : __fsub_rn(zn, zk[i]);
    const float e = expf(__fmul_rn(-delta, softplus ? nsr_softplus_density(sg[i]) : fmaxf(sg[i], 0.0f)));
    float a = __fsub_rn(1.0f, e);
    if (k >= N) a = 0.0f;
    alpha[i] = a;
    ex[i] = e;
    dl[i] = delta;
    const float f = (k < N) ? __fadd_rn(__fsub_rn(1.0f, a), 1e-10f) : 1.0f;
    ff[i] = f;
    run *= (double)f;
    pl[i] = run;
  }
  const double incl = wave_scan_mul_d(run, lane);
  double excl = __shfl_up(incl, 1, 64);
  if (lane == 0) excl = 1.0;
  // weights, gw, and the inclusive prefix of gw_i w_i (suffix = total - prefix)
  float T[K], w[K], gw[K];
  double pre[K];
  double acc = 0.0;
  const float white_term = (white & NSR_WHITE_BKGD) ? __fadd_rn(__fadd_rn(gc0, gc1), gc2) : 0.0f;

#pragma unroll
  for (int i = 0; i < K; ++i) {
    const int k = lane * K + i;
    const double t_d = (i == 0) ? excl : excl * pl[(i > 0) ? i - 1 : 0];
    T[i] = (k == 0) ? 1.0f : (float)t_d;
    w[i] = __fmul_rn(alpha[i], T[i]);
    gw[i] = gc0 * c0[i] + gc1 * c1[i] + gc2 * c2[i] - white_term;
    if (g_depth) gw[i] += gd * zk[i];
    if (k < N) acc += (double)gw[i] * (double)w[i];
    pre[i] = acc;
  }
  const double lane_incl = wave_scan_add_d(acc, lane);
  const double total = __shfl(lane_incl, 63, 64);
  const double lane_excl = lane_incl - acc;
  float bs0 = 0.0f, bs1 = 0.0f, bs2 = 0.0f, bs3 = 0.0f;     // COMPACT: this ray's sums of the four values = its share of the
                                                            // colour head's and the density head's bias gradients
#pragma unroll
  for (int i = 0; i < K; ++i) {
    const int k = lane * K + i;
    if (k >= N) continue;
    const int64_t p = base + k;
    const double suffix = total - (lane_excl + pre[i]);             // sum_{i' > k} gw w
    const float d_alpha = (float)((double)gw[i] * (double)T[i] - suffix / (double)ff[i]);
    float d_sigma;
    if (softplus) d_sigma = d_alpha * dl[i] * ex[i] * (1.0f / (1.0f + expf(1.0f - sg[i])));      // d log(1 + e^(x - 1)) / dx = sigmoid(x - 1)
    else d_sigma = (sg[i] > 0.0f) ? d_alpha * dl[i] * ex[i] : 0.0f;
    const float dr0 = colour_head_bwd(gc0 * w[i], c0[i], head);
    const float dr1 = colour_head_bwd(gc1 * w[i], c1[i], head);
    const float dr2 = colour_head_bwd(gc2 * w[i], c2[i], head);
    if (COMPACT) {
      reinterpret_cast<float4*>(d_rgb)[p] = make_float4(dr0, dr1, dr2, d_sigma);
      bs0 += dr0; bs1 += dr1; bs2 += dr2; bs3 += d_sigma;
      continue;
    }
    float4* dr = reinterpret_cast<float4*>(d_rgb + p * kRgbPad);
    dr[0] = make_float4(dr0, dr1, dr2, 0.0f);
#pragma unroll
    for (int j = 1; j < 8; ++j) dr[j] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    float* gp = g1 + p * kGs + kSigmaCol;
    gp[0] = d_sigma;
#pragma unroll
    for (int j = 1; j < 32; ++j) gp[j] = 0.0f;
  }
  if (COMPACT && bias_part) {      // one float4 per ray; finish_jobs_kernel (kind 2) sums the rays in double
    bs0 = wave_sum(bs0); bs1 = wave_sum(bs1); bs2 = wave_sum(bs2); bs3 = wave_sum(bs3);
    if (lane == 0) reinterpret_cast<float4*>(bias_part)[r] = make_float4(bs0, bs1, bs2, bs3);
  }
}

// bias gradients.  Two deterministic passes each (double accumulation, then one finishing block):
//   colsum_few:  sums of <= 4 columns of a row-major buffer over all P rows (colour-head and density-head biases,
//                whose pre-activation gradients come from the compositing backward, not from a GEMM)
//   tilesum:     sums over the per-row-tile column sums a dgrad GEMM's epilogue left behind (every other bias)
constexpr int kSumBlocks = 256;
__global__ void __launch_bounds__(256) colsum_few_kernel(const float* __restrict__ src, int64_t ld, int64_t P, int col0,
                                                         int cols, double* __restrict__ partial) {
  __shared__ double red[4][256];
  double s[4] = {0.0, 0.0, 0.0, 0.0};
  for (int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x; r < P; r += (int64_t)kSumBlocks * 256)
    for (int c = 0; c < cols; ++c) s[c] += (double)src[r * ld + col0 + c];
  for (int c = 0; c < 4; ++c) red[c][threadIdx.x] = s[c];
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o)
      for (int c = 0; c < 4; ++c) red[c][threadIdx.x] += red[c][threadIdx.x + o];
    __syncthreads();
  }
  if ((int)threadIdx.x < cols) partial[blockIdx.x * 4 + threadIdx.x] = red[threadIdx.x][0];
}
// block (x: 64-column group, y: slice of the row tiles): 64 columns x 4 row phases
__global__ void __launch_bounds__(256) tilesum_partial_kernel(const float* __restrict__ tiles, int64_t n_tiles, int ld,
                                                              int cols, double* __restrict__ partial) {
  __shared__ double red[4][64];
  const int cl = threadIdx.x & 63, c = blockIdx.x * 64 + cl, phase = threadIdx.x >> 6;
  const int64_t per = (n_tiles + gridDim.y - 1) / gridDim.y;
  const int64_t lo = per * blockIdx.y, hi = (lo + per < n_tiles) ? lo + per : n_tiles;
  double s = 0.0;
  if (c < cols)
    for (int64_t t = lo + phase; t < hi; t += 4) s += (double)tiles[t * ld + c];
  red[phase][cl] = s;
  __syncthreads();
  if (phase == 0 && c < cols) partial[(int64_t)blockIdx.y * ld + c] = (red[0][cl] + red[1][cl]) + (red[2][cl] + red[3][cl]);
}
// dst[c] (+)= sum_j partial[j * ld + c]: one wavefront per column (4 columns per block), so the n partials of a
// column are loaded in parallel and combined by shuffles in a fixed order
__global__ void __launch_bounds__(256) sum_finish_kernel(const double* __restrict__ partial, int n, int ld, int cols,
                                                         float* __restrict__ dst, int accumulate) {
  const int lane = threadIdx.x & 63, c = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (c >= cols) return;   // wave-uniform
  double s = 0.0;
  for (int j = lane; j < n; j += 64) s += partial[(int64_t)j * ld + c];
  s = wave_sum_d(s);
  if (lane == 0) dst[c] = (accumulate ? dst[c] : 0.0f) + (float)s;
}

// second pass of the split-K weight gradient + scatter into the nn.Linear shape:
// dst[i * dst_ld + dc0 + j] (+)= sum_z partial[z * stride + (pr0 + i) * p_ld + pc0 + j]
__global__ void __launch_bounds__(256) reduce_place_kernel(float* __restrict__ dst, int dst_ld, int dc0, int rows, int cols,
                                                           const float* __restrict__ partial, int splits, int64_t stride,
                                                           int p_ld, int pr0, int pc0, int accumulate, float scale) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= rows * cols) return;
  const int i = idx / cols, j = idx % cols;
  const float* src = partial + (int64_t)(pr0 + i) * p_ld + pc0 + j;
  double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;   // four independent chains: the loads of a round are all in flight
  int zc = 0;
  for (; zc + 4 <= splits; zc += 4) {
    s0 += (double)src[(zc + 0) * stride];
    s1 += (double)src[(zc + 1) * stride];
    s2 += (double)src[(zc + 2) * stride];
    s3 += (double)src[(zc + 3) * stride];
  }
  for (; zc < splits; ++zc) s0 += (double)src[zc * stride];
  const double s = (s0 + s1) + (s2 + s3);
  float* d = dst + (int64_t)i * dst_ld + dc0 + j;
  *d = (accumulate ? *d : 0.0f) + (float)(s * (double)scale);   // scale: a power of two (pre-scaled operands)
}
// sums over the points of  w[p][c] * panel[p][row]  for c < NW weights per point -- the weight gradients of the 1- and
// 3-row heads (sigma over h8 = forward panel 7, rgb over dir_encoding's output = forward panel 9), whose "GEMM" is a stream
// over one 2-byte panel:  partial[z][c * R + row] = sum over slice z.  The run of a point group is R / 16 units of 1 KiB
// (nsr_f16x3_core.h): 16-byte slot sl of unit U holds, for point m = ((sl ^ 8 (U & 1)) >> 1) and lane half h = sl & 1,
// the features 32 (U >> 1) + 16 (U & 1) + 4 h + {0..3} and + 8 + {0..3}.  Thread t owns slot t & 63 of the units
// (t >> 6) + 4 i: its eight features are the same for every point group, its point is m.
template <int R, int NW>
__global__ void __launch_bounds__(256) panel_wsums_kernel(const char* __restrict__ panel, int64_t P, const float* __restrict__ w,
                                                          int w_stride, int64_t groups_per_slice, float* __restrict__ partial) {
  constexpr int NI = R / 64;               // units per thread and point group
  typedef _Float16 h8v __attribute__((ext_vector_type(8)));
  const int tid = threadIdx.x, sl = tid & 63, u0 = tid >> 6, z = blockIdx.x;
  const int64_t n_groups = P / 32;
  const int64_t g0 = (int64_t)z * groups_per_slice;
  const int64_t g1 = (g0 + groups_per_slice < n_groups) ? g0 + groups_per_slice : n_groups;
  float acc[NI][8][NW];
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int e = 0; e < 8; ++e)
#pragma unroll
      for (int c = 0; c < NW; ++c) acc[i][e][c] = 0.0f;
  for (int64_t g = g0; g < g1; ++g) {
    const char* run = panel + g * (int64_t)(R * 64);
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int U = u0 + 4 * i;
      const int m = (sl ^ (8 * (U & 1))) >> 1;
      const h8v v = __builtin_nontemporal_load(reinterpret_cast<const h8v*>(run + U * 1024 + sl * 16));
      float wk[NW];
#pragma unroll
      for (int c = 0; c < NW; ++c) wk[c] = w[(g * 32 + m) * w_stride + c];
#pragma unroll
      for (int e = 0; e < 8; ++e)
#pragma unroll
        for (int c = 0; c < NW; ++c) acc[i][e][c] = fmaf((float)v[e], wk[c], acc[i][e][c]);
    }
  }
  // the 32 points of a (unit, lane half) are the slots of one parity: sum over slot bits 1..5
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int U = u0 + 4 * i, h = sl & 1;
#pragma unroll
    for (int e = 0; e < 8; ++e)
#pragma unroll
      for (int c = 0; c < NW; ++c) {
        float s = acc[i][e][c];
#pragma unroll
        for (int o = 2; o < 64; o <<= 1) s += __shfl_xor(s, o, 64);
        const int row = 32 * (U >> 1) + 16 * (U & 1) + 8 * (e >> 2) + 4 * h + (e & 3);
        if ((sl >> 1) == 0) partial[(int64_t)z * (NW * R) + c * R + row] = s;
      }
  }
}

// enc_rows: the partial's columns are rows of an encoding panel (nsr_f16x3_core.h, enc_row): 1 = the encoded position,
// column j of the 63 is register t of lane half h with pecol(t, h) == j; 2 = the encoded direction, dircol(t, h) == j
__device__ __forceinline__ int enc_panel_row(int enc_rows, int j) {
  if (enc_rows == 0) return j;
  const int per = enc_rows == 1 ? 30 : 12;           // columns per lane half behind the three raw coordinates
  const int t = j < 2 ? j : (j == 2 ? 0 : (j - 3) % per + 2), h = j < 2 ? 0 : (j == 2 ? 1 : (j - 3) / per);
  return enc_row(t, h);
}
// All second passes of one network's weight / bias gradients in ONE launch (chain path): blockIdx.y = job.
//   kind 0: dst[i * dst_ld + dc0 + j] (+)= scale * sum_z partial[z * stride + i * p_ld + col(j)]   (reduce_place_kernel)
//   kind 1: dst[i] (+)= sum_z partial[z * rows + i]                                                 (rowsum_finish_kernel)
//   kind 2: dst[i] (+)= sum_z partial[z * stride + i], i < rows <= 4, `splits` up to thousands (one partial per ray: the
//           bias gradients of the two heads, composite_bwd_kernel): one wavefront per element, lanes stride over z
// Twenty-odd launches of a few microseconds of work each (one wave of latency-bound workgroups) became the tail of
// the step once the GEMMs before them had shrunk; together they keep the memory system busy.
struct FinishJob {
  float* dst;
  const float* partial;
  int64_t stride;
  int kind, dst_ld, dc0, rows, cols, splits, p_ld, accumulate, enc_rows;
  float scale;
};
constexpr int kMaxFinishJobs = 32;
struct FinishJobs {
  FinishJob j[kMaxFinishJobs];
  int n;
};
__global__ void __launch_bounds__(256) finish_jobs_kernel(FinishJobs jobs) {
  const FinishJob& q = jobs.j[blockIdx.y];
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (q.kind == 2) {
    const int e = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (blockIdx.x != 0 || e >= q.rows) return;       // wave-uniform
    double s = 0.0;
    for (int z = lane; z < q.splits; z += 64) s += (double)q.partial[(int64_t)z * q.stride + e];
    s = wave_sum_d(s);
    if (lane == 0) q.dst[e] = (q.accumulate ? q.dst[e] : 0.0f) + (float)s;
    return;
  }
  if (q.kind == 1 && q.splits > 64) {
    // many slices (the head streams: 1,024, round 6): one wavefront per element, lanes stride over the slices -- a thread
    // that walks them alone is a chain of a hundred dependent round trips
    const int e = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    for (int i = e; i < q.rows; i += 4 * gridDim.x) {      // wave-uniform
      double s = 0.0;
      for (int z = lane; z < q.splits; z += 64) s += (double)q.partial[(int64_t)z * q.rows + i];
      s = wave_sum_d(s);
      if (lane == 0) q.dst[i] = (q.accumulate ? q.dst[i] : 0.0f) + (float)(s * (double)q.scale);
    }
    return;
  }
  if (idx >= q.rows * q.cols) return;
  const float* src;
  int64_t stride;
  float* d;
  if (q.kind == 0) {
    const int i = idx / q.cols, j = idx % q.cols;
    const int js = enc_panel_row(q.enc_rows, j);
    src = q.partial + (int64_t)i * q.p_ld + js;
    stride = q.stride;
    d = q.dst + (int64_t)i * q.dst_ld + q.dc0 + j;
  } else {
    src = q.partial + idx;
    stride = q.rows;
    d = q.dst + idx;
  }
  // Eight loads in flight per thread (round 6; four until then): the kernel is latency-bound -- ~21 partial tiles per
  // product, 26 blocks per CU of which 8 are resident, every round a trip to L2 / HBM (38 us per call for 29 MB).  The
  // association of the sum is fixed (eight chains, then a tree): bit-reproducible run to run.
  double a[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
  int zc = 0;
  for (; zc + 8 <= q.splits; zc += 8) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = src[(int64_t)(zc + u) * stride];
#pragma unroll
    for (int u = 0; u < 8; ++u) a[u] += (double)v[u];
  }
  {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = (zc + u < q.splits) ? src[(int64_t)(zc + u) * stride] : 0.0f;
#pragma unroll
    for (int u = 0; u < 8; ++u) a[u] += (double)v[u];
  }
  const double sum = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
  *d = (q.accumulate ? *d : 0.0f) + (float)(sum * (double)q.scale);
}

struct AdamPtrs {
  float* w[NSR_N_STATE_TENSORS];
  const float* g[NSR_N_STATE_TENSORS];
  float* m[NSR_N_STATE_TENSORS];
  float* v[NSR_N_STATE_TENSORS];
};
// torch/optim/adam.py (_single_tensor_adam) operation order, fp32
__global__ void __launch_bounds__(256) adam_kernel(AdamPtrs a, float beta1, float beta2, float eps, float step_size,
                                                   float bc2_sqrt) {
  const int t = blockIdx.y;
  const int64_t n = tensor_numel(t);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float g = a.g[t][i];
    const float m = __fadd_rn(__fmul_rn(a.m[t][i], beta1), __fmul_rn(g, 1.0f - beta1));
    const float v = __fadd_rn(__fmul_rn(a.v[t][i], beta2), __fmul_rn(__fmul_rn(g, g), 1.0f - beta2));   // addcmul: (g*g)*value
    a.m[t][i] = m;
    a.v[t][i] = v;
    const float denom = __fadd_rn(__fdiv_rn(sqrtf(v), bc2_sqrt), eps);
    a.w[t][i] = __fsub_rn(a.w[t][i], __fmul_rn(step_size, __fdiv_rn(m, denom)));
  }
}

// ---------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------
struct WeightPack {   // zero-padded copies of the weights whose shapes are not MFMA friendly (floats, one block)
  float *w1p, *w5p, *w9p, *wdirp, *wrgbp, *b9p, *brgbp;
  unsigned short* split;   // NSR_F16X3: (hi, lo) fp16 halves of the twelve forward weight matrices (kSplit* below)
};
// forward weight matrices in split-fp16 form: index, rows, K
constexpr int kSplitRows[12] = {256, 256, 256, 256, 256, 256, 256, 256, 256, 32, 128, 32};
constexpr int kSplitK[12] = {64, 256, 256, 256, 320, 256, 256, 256, 256, 256, 288, 128};
constexpr int64_t split_offset(int e) { return e == 0 ? 0 : split_offset(e - 1) + 2 * (int64_t)kSplitRows[e - 1] * kSplitK[e - 1]; }
constexpr int64_t kSplitHalves = split_offset(11) + 2 * (int64_t)kSplitRows[11] * kSplitK[11];

struct Work {   // per-pass buffers, sized for P_max = chunk * (Nc + Ni) sample points; all row-major
  float *x5, *h[9], *gs, *cc, *rgb, *sig;
  float *g0, *g1, *drgb, *col_tiles;
  float* d4;        // chain path: (P, 4) = d(rgb_pre) 0..2, d(sigma) of every sample point (composite_bwd_kernel COMPACT)
  float* bias_part; // chain path: (rays, 4) per-ray sums of d4: the bias gradients of the colour and density heads before their finish
  float *z_c, *z_f, *w_c, *comp, *g_comp, *partial, *scratch_out;
  double *block_sums, *carry;
  float* g_depth;   // per ray: d(loss) / d(depth) of the depth-variance loss (zeros when it is off)
  WeightPack pack[2];
  // chain path (NSR_F16X3): activation panels of the forward pass, gradient panels of the backward chain (2 bytes per
  // value, nsr_f16x3_core.h), the two weight streams per network, per-slice row sums of the weight-gradient products
  char *zpan, *dpan;
  float *row_part, *slots;
  float *stream_f[2], *stream_b[2];
  unsigned* sgn;    // sign panels of the forward pass (nsr_f16x3_core.h)
  float* pscale;    // per gradient panel and point: stored value x pscale = true gradient (written by the backward chain)
  unsigned* gmax;   // float bits of the largest magnitude in each gradient panel (written by the backward chain)
  unsigned* status; // sticky NSR_FLAG_* word of the training step (include/nsr_train.h): ALWAYS the first bytes of the workspace
};

// mode 0: every buffer of both paths (nsr_train_workspace_bytes: sufficient whatever runs); 1: the layer-by-layer GEMM path
// only; 2: the chain path only (no per-layer activation / gradient matrices: 11 KB per sample point less)
int64_t work_floats(int64_t chunk, int nc, int ni, Work* w, float* base, int mode = 0) {
  const int64_t nf = nc + ni, P = chunk * nf;
  int64_t off = 0;
  auto take_if = [&](bool on, int64_t n) {
    if (!on) return static_cast<float*>(nullptr);
    float* p = base ? base + off : nullptr;
    off += align64(n);
    return p;
  };
  auto take = [&](int64_t n) { return take_if(true, n); };
  const bool gemm_path = mode != 2, chain_path = mode != 1;
  Work tmp;
  Work& k = w ? *w : tmp;
  k.status = reinterpret_cast<unsigned*>(take(16));        // offset 0 whatever the path: nsr_train_status reads it blind
  k.x5 = take_if(gemm_path, P * kX5);
  for (int L = 1; L <= 8; ++L) k.h[L] = (L == 4) ? nullptr : take_if(gemm_path, P * kW);   // h4 lives in x5[:, 64:]
  k.gs = take_if(gemm_path, P * kGs);
  k.cc = take_if(gemm_path, P * kDirOut);
  k.rgb = take(P * 4);   k.sig = take(P);
  k.g0 = take_if(gemm_path, P * kGs);   k.g1 = take_if(gemm_path, P * kGs);
  k.drgb = take_if(gemm_path, P * kRgbPad);
  k.d4 = take_if(chain_path, P * 4);
  k.bias_part = take_if(chain_path, chunk * 4);
  k.col_tiles = take((P / 128 + 1) * kW + 2 * 64 * kW + 64);   // per-tile column sums + 64 slices of doubles
  k.z_c = take(chunk * nc);   k.z_f = take(chunk * nf);   k.w_c = take(chunk * nc);
  k.comp = take(chunk * 3);   k.g_comp = take(chunk * 3);
  k.scratch_out = take(chunk * (nf + 8));
  k.partial = take(kMaxSplits * kPartialFloats);           // also scratch of the small bias sums
  // chain path: every second pass of a network waits for one launch; sized by the split-K factor of the larger pass
  const int64_t sp_max = (P + 511) / 512 < 1 ? 1 : ((P + 511) / 512 > kMaxSplits ? kMaxSplits : (P + 511) / 512);
  k.slots = take_if(chain_path, kChainSlots * sp_max * 256 * 256);
  k.block_sums = reinterpret_cast<double*>(take(2 * 3 * (chunk / 256 + 2)));
  k.carry = reinterpret_cast<double*>(take(16));
  k.g_depth = take(chunk);
  for (int n = 0; n < 2; ++n) {
    WeightPack& q = k.pack[n];
    q.w1p = take(256 * 64);   q.w5p = take(256 * 320);   q.w9p = take(288 * 256);   q.wdirp = take(128 * 288);
    q.wrgbp = take(32 * 128);   q.b9p = take(320);   q.brgbp = take(64);
    q.split = reinterpret_cast<unsigned short*>(take((kSplitHalves + 1) / 2));
  }
  const int64_t pan = nsr_f16x3_train_panel_bytes(P) / 4;
  k.zpan = reinterpret_cast<char*>(take_if(chain_path, pan));   k.dpan = reinterpret_cast<char*>(take_if(chain_path, pan));
  k.row_part = take_if(chain_path, kChainRowSlots * sp_max * 256);
  k.gmax = reinterpret_cast<unsigned*>(take_if(chain_path, 64));
  k.pscale = take_if(chain_path, 10 * ((P + 127) / 128) * 128);
  k.sgn = reinterpret_cast<unsigned*>(take_if(chain_path, nsr_f16x3_train_sign_words(P)));
  for (int n = 0; n < 2; ++n) {
    k.stream_f[n] = take_if(chain_path, (int64_t)(nsr_f16x3_packed_bytes() / 4));
    k.stream_b[n] = take_if(chain_path, (int64_t)(nsr_chain_bwd_packed_bytes() / 4));
  }
  return off;
}

#define NSR_TRY(expr)            \
  do {                           \
    const int rc_ = (expr);      \
    if (rc_ != NSR_OK) return rc_; \
  } while (0)

int place(hipStream_t st, float* dst, int dst_ld, int r0, int c0, const float* src, int src_ld, int rows, int cols,
          int col0, int transpose) {
  const int n = rows * cols;
  hipLaunchKernelGGL(place_kernel, dim3((n + 255) / 256), dim3(256), 0, st, dst, dst_ld, r0, c0, src, src_ld, rows, cols,
                     col0, transpose);
  NSR_CHECK_LAUNCH();
  return NSR_OK;
}

int prepare_weights(hipStream_t st, const float* const* w, const WeightPack& q, int precision) {
  // zero the whole pack first (padding rows / columns), it is one contiguous block starting at w1p
  if (hipMemsetAsync(q.w1p, 0, (size_t)((q.brgbp + align64(64)) - q.w1p) * sizeof(float), st) != hipSuccess)
    return NSR_ERR_LAUNCH;
  NSR_TRY(place(st, q.w1p, 64, 0, 0, w[0], 63, 256, 63, 0, 0));
  NSR_TRY(place(st, q.w5p, 320, 0, 0, w[8], 319, 256, 63, 0, 0));
  NSR_TRY(place(st, q.w5p, 320, 0, 64, w[8], 319, 256, 256, 63, 0));
  NSR_TRY(place(st, q.w9p, 256, 0, 0, w[kFinalW], 256, 256, 256, 0, 0));
  NSR_TRY(place(st, q.w9p, 256, 256, 0, w[kSigmaW], 256, 1, 256, 0, 0));
  NSR_TRY(place(st, q.wdirp, 288, 0, 0, w[kDirW], 283, 128, 256, 0, 0));
  NSR_TRY(place(st, q.wdirp, 288, 0, kDeCol, w[kDirW], 283, 128, 27, 256, 0));
  NSR_TRY(place(st, q.wrgbp, 128, 0, 0, w[kRgbW], 128, 3, 128, 0, 0));
  NSR_TRY(place(st, q.b9p, 320, 0, 0, w[kFinalB], 256, 1, 256, 0, 0));
  NSR_TRY(place(st, q.b9p, 320, 0, 256, w[kSigmaB], 1, 1, 1, 0, 0));
  NSR_TRY(place(st, q.brgbp, 64, 0, 0, w[kRgbB], 3, 1, 3, 0, 0));
  if (precision == NSR_F16X3) {
    const float* src[12] = {q.w1p, w[2], w[4], w[6], q.w5p, w[10], w[12], w[14], q.w9p, q.w9p + 256 * 256, q.wdirp, q.wrgbp};
    for (int e = 0; e < 12; ++e) {
      const int64_t n = (int64_t)kSplitRows[e] * kSplitK[e];
      NSR_TRY(split_f16(src[e], n, q.split + split_offset(e), q.split + split_offset(e) + n, st));
    }
  }
  return NSR_OK;
}

// y (P, N) = act(x (P, K) w (N, K)^T + b); `split` (entry e of the pack's split block) selects the split-fp16 product
int lin_fwd(hipStream_t st, const float* x, int64_t ldx, int K, const float* w, int ldw, const float* b, int act,
            float* y, int64_t ldy, int64_t P, int N, int n_valid, const unsigned short* split = nullptr, int e = 0) {
  GemmArgs g{};
  g.A = x; g.lda = ldx; g.B = w; g.ldb = ldw; g.C = y; g.ldc = ldy; g.bias = b;
  g.M = P; g.N = N; g.K = K; g.n_valid = n_valid; g.act = act; g.splits = 1;
  if (!split) return gemm(g, st);
  GemmF16Args a{};
  a.g = g;
  a.g.acc_scale = kSplitInvScale;
  a.Bh = split + split_offset(e);
  a.Bl = a.Bh + (int64_t)kSplitRows[e] * kSplitK[e];
  a.ldbh = kSplitK[e];
  return gemm_f16x3(a, st);
}
// dx (P, N) = (dy (P, K) w[:, 0 : N]) * [mask > 0], w (K, ldw) in the nn.Linear layout (mask may be null);
// bias_grad (N) (+)= column sums of dx = the bias gradient of the layer that produced the masked activation
int lin_dgrad(hipStream_t st, const Work& k, const float* dy, int64_t lddy, int K, const float* w, int ldw,
              const float* mask, int64_t ldm, float* dx, int64_t lddx, int64_t P, int N, float* bias_grad, int acc) {
  GemmArgs g{};
  g.A = dy; g.lda = lddy; g.B = w; g.ldb = ldw; g.b_kmajor = 1; g.C = dx; g.ldc = lddx;
  g.mask = mask; g.ldm = ldm; g.M = P; g.N = N; g.K = K; g.n_valid = N; g.act = kActNone; g.splits = 1;
  g.col_sums = bias_grad ? k.col_tiles : nullptr;
  const int rc = gemm(g, st);
  if (rc != NSR_OK || !bias_grad) return rc;
  double* part = reinterpret_cast<double*>(k.col_tiles + ((P + 127) / 128) * N);   // behind the tile sums
  const int slices = 64;
  hipLaunchKernelGGL(tilesum_partial_kernel, dim3((N + 63) / 64, slices), dim3(256), 0, st, k.col_tiles, (P + 127) / 128, N,
                     N, part);
  NSR_CHECK_LAUNCH();
  hipLaunchKernelGGL(sum_finish_kernel, dim3((N + 3) / 4), dim3(256), 0, st, part, slices, N, N, bias_grad, acc);
  NSR_CHECK_LAUNCH();
  return NSR_OK;
}
// split-K factor of the weight gradients: 2 row tiles x splits workgroups should cover the 256 CUs at least once
int n_splits(int64_t P) {
  int64_t s = (P + 511) / 512;
  return (int)(s < 1 ? 1 : (s > kMaxSplits ? kMaxSplits : s));
}
// partial[z] (M x N) = sum over the z-th slice of the points of dy[p][0..M) x[p][0..N)^T
int lin_wgrad(hipStream_t st, const float* dy, int64_t lddy, int M, const float* x, int64_t ldx, int N, int64_t P,
              float* partial, int splits) {
  GemmArgs g{};
  g.A = dy; g.lda = lddy; g.a_kmajor = 1; g.B = x; g.ldb = ldx; g.b_kmajor = 1; g.C = partial; g.ldc = N;
  g.M = M; g.N = N; g.K = P; g.n_valid = N; g.act = kActNone; g.splits = splits; g.split_stride = kPartialFloats;
  return gemm(g, st);
}
int reduce_place(hipStream_t st, float* dst, int dst_ld, int dc0, int rows, int cols, const float* partial, int splits,
                 int p_ld, int pr0, int pc0, int accumulate, float scale = 1.0f) {
  const int n = rows * cols;
  hipLaunchKernelGGL(reduce_place_kernel, dim3((n + 255) / 256), dim3(256), 0, st, dst, dst_ld, dc0, rows, cols, partial,
                     splits, kPartialFloats, p_ld, pr0, pc0, accumulate, scale);
  NSR_CHECK_LAUNCH();
  return NSR_OK;
}
// `scratch`: >= kSumBlocks * 4 doubles (the split-K partial buffer is free between two weight gradients)
int colsum(hipStream_t st, const float* src, int64_t ld, int64_t P, int col0, int cols, float* dst, int accumulate,
           float* scratch) {
  if (cols > 4) return NSR_ERR_INVALID_ARG;
  double* part = reinterpret_cast<double*>(scratch);
  hipLaunchKernelGGL(colsum_few_kernel, dim3(kSumBlocks), dim3(256), 0, st, src, ld, P, col0, cols, part);
  NSR_CHECK_LAUNCH();
  hipLaunchKernelGGL(sum_finish_kernel, dim3((cols + 3) / 4), dim3(256), 0, st, part, kSumBlocks, 4, cols, dst, accumulate);
  NSR_CHECK_LAUNCH();
  return NSR_OK;
}

// M1 forward with everything kept for the backward pass
int net_forward(hipStream_t st, const float* const* w, const WeightPack& q, const Work& k, int64_t P, int precision, int color_none) {
  const unsigned short* sp = precision == NSR_F16X3 ? q.split : nullptr;
  NSR_TRY(lin_fwd(st, k.x5, kX5, kPe, q.w1p, 64, w[1], kActRelu, k.h[1], kW, P, kW, kW, sp, 0));
  NSR_TRY(lin_fwd(st, k.h[1], kW, kW, w[2], 256, w[3], kActRelu, k.h[2], kW, P, kW, kW, sp, 1));
  NSR_TRY(lin_fwd(st, k.h[2], kW, kW, w[4], 256, w[5], kActRelu, k.h[3], kW, P, kW, kW, sp, 2));
  NSR_TRY(lin_fwd(st, k.h[3], kW, kW, w[6], 256, w[7], kActRelu, k.x5 + kPe, kX5, P, kW, kW, sp, 3));
  NSR_TRY(lin_fwd(st, k.x5, kX5, kX5, q.w5p, 320, w[9], kActRelu, k.h[5], kW, P, kW, kW, sp, 4));
  NSR_TRY(lin_fwd(st, k.h[5], kW, kW, w[10], 256, w[11], kActRelu, k.h[6], kW, P, kW, kW, sp, 5));
  NSR_TRY(lin_fwd(st, k.h[6], kW, kW, w[12], 256, w[13], kActRelu, k.h[7], kW, P, kW, kW, sp, 6));
  NSR_TRY(lin_fwd(st, k.h[7], kW, kW, w[14], 256, w[15], kActRelu, k.h[8], kW, P, kW, kW, sp, 7));
  // xyz_encoding_final stacked over the density head: [g | sigma] into columns 0..256 of the dir layer's input
  // (two launches: the 256 wide columns on the 8-wave tile, the density row on the narrow one, instead of a second
  // 256-wide column tile that would be 7/8 padding)
  NSR_TRY(lin_fwd(st, k.h[8], kW, kW, q.w9p, 256, q.b9p, kActNone, k.gs, kGs, P, kW, kW, sp, 8));
  NSR_TRY(lin_fwd(st, k.h[8], kW, kW, q.w9p + 256 * 256, 256, q.b9p + 256, kActNone, k.gs + kSigmaCol, kGs, P, 32, 1, sp, 9));
  NSR_TRY(lin_fwd(st, k.gs, kGs, kGs, q.wdirp, 288, w[kDirB], kActRelu, k.cc, kDirOut, P, kDirOut, kDirOut, sp, 10));
  NSR_TRY(lin_fwd(st, k.cc, kDirOut, kDirOut, q.wrgbp, 128, q.brgbp, color_none ? kActNone : kActSigmoid, k.rgb, 4, P, kRgbPad, 3, sp, 11));
  return NSR_OK;
}

// backward of M1: d_rgb_pre in k.drgb (P, 32), d_sigma in column 256 of k.g1 (P, 288)
int net_backward(hipStream_t st, const float* const* w, const WeightPack& q, const Work& k, int64_t P, float* const* g,
                 int acc, int stop_grad) {
  const int sp = n_splits(P);
  float* part = k.partial;
  // rgb head
  NSR_TRY(lin_wgrad(st, k.drgb, kRgbPad, kRgbPad, k.cc, kDirOut, kDirOut, P, part, sp));
  NSR_TRY(reduce_place(st, g[kRgbW], 128, 0, 3, 128, part, sp, kDirOut, 0, 0, acc));
  NSR_TRY(colsum(st, k.drgb, kRgbPad, P, 0, 3, g[kRgbB], acc, part));
  NSR_TRY(lin_dgrad(st, k, k.drgb, kRgbPad, kRgbPad, q.wrgbp, 128, k.cc, kDirOut, k.g0, kDirOut, P, kDirOut, g[kDirB], acc));
  // dir_encoding
  NSR_TRY(lin_wgrad(st, k.g0, kDirOut, kDirOut, k.gs, kGs, kGs, P, part, sp));
  NSR_TRY(reduce_place(st, g[kDirW], 283, 0, 128, 256, part, sp, kGs, 0, 0, acc));
  NSR_TRY(reduce_place(st, g[kDirW], 283, 256, 128, 27, part, sp, kGs, 0, kDeCol, acc));
  // d g (its column sums are xyz_encoding_final's bias gradient); column 256 keeps d sigma
  if (stop_grad) {   // --stop_grad (models/networks.py:218-219): dir_encoding's input is detached, d g = 0
    if (hipMemset2DAsync(k.g1, (size_t)kGs * sizeof(float), 0, (size_t)kW * sizeof(float), (size_t)P, st) != hipSuccess) return NSR_ERR_LAUNCH;
    if (!acc && hipMemsetAsync(g[kFinalB], 0, (size_t)kW * sizeof(float), st) != hipSuccess) return NSR_ERR_LAUNCH;
  } else {
    NSR_TRY(lin_dgrad(st, k, k.g0, kDirOut, kDirOut, q.wdirp, 288, nullptr, 0, k.g1, kGs, P, kW, g[kFinalB], acc));
  }
  // xyz_encoding_final + sigma (288-row layer over h8)
  NSR_TRY(lin_wgrad(st, k.g1, kGs, kW, k.h[8], kW, kW, P, part, sp));                 // rows 0..255: xyz_encoding_final
  NSR_TRY(reduce_place(st, g[kFinalW], 256, 0, 256, 256, part, sp, kW, 0, 0, acc));
  NSR_TRY(lin_wgrad(st, k.g1 + kSigmaCol, kGs, 32, k.h[8], kW, kW, P, part, sp));     // row 256 (+ 31 zero rows): sigma
  NSR_TRY(reduce_place(st, g[kSigmaW], 256, 0, 1, 256, part, sp, kW, 0, 0, acc));
  NSR_TRY(colsum(st, k.g1, kGs, P, 256, 1, g[kSigmaB], acc, part));
  NSR_TRY(lin_dgrad(st, k, k.g1, kGs, kGs, q.w9p, 256, k.h[8], kW, k.g0, kW, P, kW, g[15], acc));   // + bias of layer 8
  // xyz_encoding_8 .. 1; the gradient of layer L's pre-activation alternates between the two buffers
  const float* dy = k.g0;
  float* nx = k.g1;
  for (int L = 8; L >= 1; --L) {
    const float* xin = (L == 1 || L == 5) ? k.x5 : k.h[L - 1];
    const int64_t ldx = (L == 1 || L == 5) ? kX5 : kW;
    const int kin = (L == 1) ? kPe : (L == 5 ? kX5 : kW);
    NSR_TRY(lin_wgrad(st, dy, kW, kW, xin, ldx, kin, P, part, sp));
    float* gw = g[2 * (L - 1)];
    if (L == 1) NSR_TRY(reduce_place(st, gw, 63, 0, 256, 63, part, sp, kPe, 0, 0, acc));
    else if (L == 5) {
      NSR_TRY(reduce_place(st, gw, 319, 0, 256, 63, part, sp, kX5, 0, 0, acc));
      NSR_TRY(reduce_place(st, gw, 319, 63, 256, 256, part, sp, kX5, 0, kPe, acc));
    } else NSR_TRY(reduce_place(st, gw, 256, 0, 256, 256, part, sp, kW, 0, 0, acc));
    if (L == 1) break;
    // input of layer L is the output of layer L - 1 (relu'd): h4 sits in x5[:, 64:]
    const float* mask = (L - 1 == 4) ? k.x5 + kPe : k.h[L - 1];
    const int64_t ldm = (L - 1 == 4) ? kX5 : kW;
    // weights in the nn.Linear layout (out, in) ARE the K-major B operand of the input gradient
    const float* wl = (L == 5) ? q.w5p + kPe : w[2 * (L - 1)];
    const int ldw = (L == 5) ? kX5 : kW;
    NSR_TRY(lin_dgrad(st, k, dy, kW, kW, wl, ldw, mask, ldm, nx, kW, P, kW, g[2 * (L - 2) + 1], acc));   // + bias of layer L - 1
    const float* t0 = dy; dy = nx; nx = const_cast<float*>(t0);
  }
  return NSR_OK;
}

// ---- chain path ------------------------------------------------------------------------------------------
int64_t n_groups_of(int64_t P) { return ((P + 127) / 128) * 4; }
char* panel_of(char* set, int64_t P, int panel) { return set + panel_offset_bytes(n_groups_of(P), panel); }

// weight and bias gradients from the panels: zpan = the forward activations, dpan = the input gradients at each point's
// power-of-two scale (k.pscale), both fp16 (nsr_f16x3_core.h); d_rgb_pre and d_sigma in k.d4 (P, 4).  The 12 panel x panel
// products of the network are ONE launch (wgrad_jobs_kernel, nsr_wgrad_f16.hip): ~256
// workgroups share the products' point groups by bytes, a 256 x 256 product ends up with ~21 partial tiles instead of the
// 256 a launch of its own needed to fill the chip -- 12 x fewer partial sums to write, and for finish_jobs_kernel to read
// back.
int chain_weight_grads(hipStream_t st, const Work& k, int64_t P, int64_t n_rays, float* const* g, int acc) {
  const int sp = n_splits(P);
  const int64_t sp_max = sp;   // the workspace's slots are sized for the largest pass (work_floats): at least this one's
  FinishJobs jobs{};
  WgradJobs wj{};
  struct Placed { int job, fin; };     // finish job `fin` reduces the partial tiles (fin >= 0) / row sums (~fin) of product `job`
  Placed placed[2 * kMaxFinishJobs];
  int n_placed = 0;
  int n_big = 0;
  auto big_slot = [&]() { return k.slots + (int64_t)(n_big++) * sp * 256 * 256; };     // sp <= the workspace's sp_max
  // second passes, executed by finish_jobs_kernel at the end
  auto place = [&](float* dst, int dst_ld, int dc0, int rows, int cols, const float* partial, int p_ld, int enc_rows) {
    FinishJob& q = jobs.j[jobs.n++];
    q.kind = 0; q.dst = dst; q.dst_ld = dst_ld; q.dc0 = dc0; q.rows = rows; q.cols = cols; q.partial = partial;
    q.stride = (int64_t)256 * 256; q.splits = sp; q.p_ld = p_ld; q.accumulate = acc; q.enc_rows = enc_rows; q.scale = 1.0f;
    return jobs.n - 1;
  };
  auto sum_rows = [&](float* dst, int rows, const float* partial) {
    FinishJob& q = jobs.j[jobs.n++];
    q.kind = 1; q.dst = dst; q.rows = rows; q.cols = 1; q.partial = partial; q.splits = sp; q.accumulate = acc; q.scale = 1.0f;
    return jobs.n - 1;
  };
  // product of gradient panel a with forward panel b (+ the bias row sums of a): an entry of the job table (slots are
  // handed out after the plan); returns the product's index
  auto product = [&](int a_panel, int b_panel, bool row_sums) -> int {
    if (wj.n >= kMaxWgradJobs) return -1;
    WgradArgs w{};
    w.A = panel_of(k.dpan, P, a_panel); w.M = panel_rows(a_panel); w.a_gbytes = (int64_t)kPanelRowBytes * w.M;
    w.B = panel_of(k.zpan, P, b_panel); w.N = panel_rows(b_panel); w.b_gbytes = (int64_t)kPanelRowBytes * w.N;
    w.a_max_bits = k.gmax + a_panel;
    w.a_pscale = k.pscale + (int64_t)a_panel * n_groups_of(P) * 32;
    w.split_stride = (int64_t)256 * 256;
    w.partial = k.slots;                       // placeholders (validated non-null); real slots after the plan
    w.row_sums = row_sums ? k.row_part : nullptr;
    wj.j[wj.n].w = w;
    return wj.n++;
  };
  auto note = [&](int job, int fin) { placed[n_placed++] = Placed{job, fin}; };
  float* part;
  int pj;
  // the two head streams: 4 slices per CU (round 6).  With one 256-thread workgroup per CU (rounds 3-5: `sp` slices) a CU had
  // 4-16 KiB of loads in flight and the streams ran at 1.5-2.3 TB/s (28 + 21 us coarse, 44 + 31 us fine: 4 % of the step);
  // the slot the partials go to holds sp x 256 x 256 floats, a slice writes 384 or 256
  const int64_t n_pg = P / 32;
  const int hs = (int)(n_pg < 1024 ? (n_pg < 1 ? 1 : n_pg) : 1024);
  const int64_t per = (n_pg + hs - 1) / hs;
  auto sum_rows_n = [&](float* dst, int rows, const float* partial, int n) {
    FinishJob& q = jobs.j[jobs.n++];
    q.kind = 1; q.dst = dst; q.rows = rows; q.cols = 1; q.partial = partial; q.splits = n; q.accumulate = acc; q.scale = 1.0f;
    return jobs.n - 1;
  };
  // rgb head: d_rgb_pre^T relu(zcc), a stream over the panel
  part = big_slot();
  hipLaunchKernelGGL((panel_wsums_kernel<128, 3>), dim3(hs), dim3(256), 0, st, panel_of(k.zpan, P, 9), P, k.d4, 4, per, part);
  NSR_CHECK_LAUNCH();
  sum_rows_n(g[kRgbW], 3 * 128, part, hs);
  auto sum_rays = [&](float* dst, int rows, const float* partial) {     // kind 2: the per-ray partials of composite_bwd_kernel
    FinishJob& q = jobs.j[jobs.n++];
    q.kind = 2; q.dst = dst; q.rows = rows; q.cols = 1; q.partial = partial; q.stride = 4; q.splits = (int)n_rays; q.accumulate = acc; q.scale = 1.0f;
  };
  sum_rays(g[kRgbB], 3, k.bias_part);
  // dir_encoding: dzc^T [g | de]
  if ((pj = product(9, 8, true)) < 0) return NSR_ERR_LAUNCH;
  note(pj, place(g[kDirW], 283, 0, 128, 256, nullptr, kW, 0));
  note(pj, ~sum_rows(g[kDirB], kDirOut, nullptr));
  if ((pj = product(9, 11, false)) < 0) return NSR_ERR_LAUNCH;
  note(pj, place(g[kDirW], 283, 256, 128, 27, nullptr, kPe, 2));
  // xyz_encoding_final: dg^T relu(z8); sigma: d_sigma^T relu(z8)
  if ((pj = product(8, 7, true)) < 0) return NSR_ERR_LAUNCH;
  note(pj, place(g[kFinalW], 256, 0, 256, 256, nullptr, kW, 0));
  note(pj, ~sum_rows(g[kFinalB], kW, nullptr));
  part = big_slot();
  hipLaunchKernelGGL((panel_wsums_kernel<256, 1>), dim3(hs), dim3(256), 0, st, panel_of(k.zpan, P, 7), P, k.d4 + 3, 4, per, part);
  NSR_CHECK_LAUNCH();
  sum_rows_n(g[kSigmaW], 256, part, hs);
  sum_rays(g[kSigmaB], 1, k.bias_part + 3);
  // trunk layers 8..1: dz_L^T (input of layer L)
  for (int L = 8; L >= 1; --L) {
    float* gw = g[2 * (L - 1)];
    int pj_rs = -1;
    if (L > 1) {
      if ((pj = product(L - 1, L - 2, true)) < 0) return NSR_ERR_LAUNCH;
      note(pj, place(gw, L == 5 ? 319 : 256, L == 5 ? 63 : 0, 256, 256, nullptr, kW, 0));
      pj_rs = pj;
    }
    if (L == 1 || L == 5) {   // over the encoded position (panel 10, 64 rows in register order)
      if ((pj = product(L - 1, 10, L == 1)) < 0) return NSR_ERR_LAUNCH;
      note(pj, place(gw, L == 1 ? 63 : 319, 0, 256, 63, nullptr, kPe, 1));
      if (L == 1) pj_rs = pj;
    }
    note(pj_rs, ~sum_rows(g[2 * (L - 1) + 1], kW, nullptr));
  }
  if (n_big > kChainSlots || jobs.n > kMaxFinishJobs) return NSR_ERR_UNSUPPORTED;   // cannot happen
  // as many workgroups as there are CUs -- fewer for a small pass, so that the partial tiles fit the slots the
  // workspace holds (sized by sp_max) and a workgroup always has a few point groups to sweep
  int64_t want = 10 * (sp_max - 1);
  want = want < 1 ? 1 : (want > 256 ? 256 : want);
  const int n_wg = wgrad_jobs_plan(wj, P, (int)want);
  float* next_big = k.slots + (int64_t)n_big * sp * 256 * 256;          // behind the two head slots taken above
  float* next_row = k.row_part;
  const float* big_end = k.slots + (int64_t)kChainSlots * sp_max * 256 * 256;
  const float* row_end = k.row_part + (int64_t)kChainRowSlots * sp_max * 256;
  for (int p = 0; p < wj.n; ++p) {
    WgradJob& q = wj.j[p];
    q.w.partial = next_big;
    next_big += (int64_t)q.n_slots * 256 * 256;
    if (q.w.row_sums) {
      q.w.row_sums = next_row;
      next_row += (int64_t)q.n_slots * q.w.M;
    }
  }
  if (next_big > big_end || next_row > row_end) return NSR_ERR_WORKSPACE;   // cannot happen (see `want`)
  for (int i = 0; i < n_placed; ++i) {
    const WgradJob& q = wj.j[placed[i].job];
    const bool rows = placed[i].fin < 0;
    FinishJob& f = jobs.j[rows ? ~placed[i].fin : placed[i].fin];
    f.partial = rows ? q.w.row_sums : q.w.partial;
    f.splits = q.n_slots;
  }
  NSR_TRY(wgrad_jobs_f16(wj, n_wg, st));
  hipLaunchKernelGGL(finish_jobs_kernel, dim3(256, jobs.n), dim3(256), 0, st, jobs);
  NSR_CHECK_LAUNCH();
  return NSR_OK;
}

// which implementation a precision value selects (include/nsr_train.h): the chain kernels or the layer-by-layer GEMMs
bool chain_selected(int precision) {
  return precision == NSR_F16X3 || precision == NSR_F16X3_BWD3 || precision == NSR_F16X3_BWD2 || precision == NSR_F16X3_BWD1 ||
         precision == NSR_F16X3_BWDM;
}
// MFMAs per product of the backward chain (include/nsr_train.h; NSR_F16X3 = the default, kDefaultBwdTerms)
constexpr int kDefaultBwdTerms = 12;   // NSR_F16X3_BWDM
int chain_bwd_terms(int precision) {
  return precision == NSR_F16X3_BWD3 ? 3 : (precision == NSR_F16X3_BWD2 ? 2 : (precision == NSR_F16X3_BWD1 ? 1 : (precision == NSR_F16X3_BWDM ? 12 : kDefaultBwdTerms)));
}
bool train_precision_ok(int precision) { return precision == NSR_FP32 || chain_selected(precision) || precision == NSR_F16X3_GEMM; }
int gemm_precision(int precision) { return precision == NSR_F16X3_GEMM ? NSR_F16X3 : precision; }   // what the GEMM path's helpers expect

int composite_bwd(hipStream_t st, const Work& k, const float* z, int64_t R, int N, int white, bool compact, const float* g_depth) {
  const dim3 block(256), grid((unsigned)((R + 3) / 4));
  const int K = (N + 63) / 64;
#define NSR_LAUNCH_CB(KK)                                                                                                          \
  do {                                                                                                                             \
    if (compact) hipLaunchKernelGGL((composite_bwd_kernel<KK, true>), grid, block, 0, st, k.rgb, k.sig, z, k.g_comp, R, N, white, k.d4, reinterpret_cast<float*>(k.gmax), g_depth, k.bias_part); \
    else hipLaunchKernelGGL((composite_bwd_kernel<KK, false>), grid, block, 0, st, k.rgb, k.sig, z, k.g_comp, R, N, white, k.drgb, k.g1, g_depth, nullptr);   \
  } while (0)
  switch (K) {
    case 1: NSR_LAUNCH_CB(1); break;
    case 2: NSR_LAUNCH_CB(2); break;
    case 3: NSR_LAUNCH_CB(3); break;
    case 4: NSR_LAUNCH_CB(4); break;
    default: return NSR_ERR_UNSUPPORTED;
  }
#undef NSR_LAUNCH_CB
  NSR_CHECK_LAUNCH();
  return NSR_OK;
}

}  // namespace

extern "C" size_t nsr_train_workspace_bytes_for(int precision, int64_t ray_chunk, int n_coarse, int n_importance) {
  if (ray_chunk <= 0 || n_coarse < 2 || n_importance < 1 || n_coarse + n_importance > 256) return 0;
  if (!train_precision_ok(precision)) return 0;
  return (size_t)work_floats(ray_chunk, n_coarse, n_importance, nullptr, nullptr, chain_selected(precision) ? 2 : 1) * sizeof(float);
}

extern "C" size_t nsr_train_workspace_bytes(int64_t ray_chunk, int n_coarse, int n_importance) {
  if (ray_chunk <= 0 || n_coarse < 2 || n_importance < 1 || n_coarse + n_importance > 256) return 0;
  return (size_t)work_floats(ray_chunk, n_coarse, n_importance, nullptr, nullptr) * sizeof(float);
}

namespace {
int train_impl(const float* const* w_coarse, const float* const* w_fine, float* const* g_coarse,
               float* const* g_fine, const float* rays, int ray_stride, int64_t R, int s2,
               const float* target_lr, int n_coarse, int n_importance, int white_bkgd,
               int lindisp, const float* u_coarse, const float* u_fine,
               const float* noise_coarse, const float* noise_fine, float noise_std,
               float lambda_coarse, float lambda_fine, int precision, int64_t ray_chunk, float* const* outs,
               float* lr_coarse, float* lr_fine, float* losses, void* workspace,
               size_t workspace_bytes, void* stream, const nsr_train_var_losses* var, float* var_losses) {
  if (!w_coarse || !w_fine || !g_coarse || !g_fine || !outs || R < 0 || s2 <= 0 || !nsr_ray_stride_ok(ray_stride))
    return NSR_ERR_INVALID_ARG;
  const bool rgb_var = var && (var->lambda_coarse_var != 0.0f || var->lambda_fine_var != 0.0f);
  const bool depth_var = var && (var->lambda_coarse_depth_var != 0.0f || var->lambda_fine_depth_var != 0.0f);
  // torch.var over ONE sub-ray is 0 / 0 (the reference would train on NaN); a depth variance needs the divisor
  if ((rgb_var || depth_var) && s2 < 2) return NSR_ERR_INVALID_ARG;
  if (depth_var && !(var->far > 0.0f)) return NSR_ERR_INVALID_ARG;
  if (n_coarse < 2 || n_importance < 1 || n_coarse + n_importance > 256) return NSR_ERR_UNSUPPORTED;
  if (!train_precision_ok(precision)) return NSR_ERR_UNSUPPORTED;
  if (R % s2 != 0) return NSR_ERR_INVALID_ARG;
  if (ray_chunk <= 0 || ray_chunk > R) ray_chunk = R;
  if (ray_chunk % s2 != 0) return NSR_ERR_INVALID_ARG;
  // the split-K weight-gradient GEMM contracts over the chunk's sample points in K tiles of 32: every chunk,
  // the shorter last one included, must hold a multiple of 32 points in both passes -- checked BEFORE anything
  // is enqueued, so a rejected call leaves outputs and gradients untouched
  for (int64_t r0 = 0; r0 < R; r0 += ray_chunk) {
    const int64_t rc = (R - r0 < ray_chunk) ? R - r0 : ray_chunk;
    if ((rc * n_coarse) % 32 != 0 || (rc * (n_coarse + n_importance)) % 32 != 0) return NSR_ERR_UNSUPPORTED;
  }
  for (int i = 0; i < NSR_N_STATE_TENSORS; ++i)
    if (!w_coarse[i] || !w_fine[i] || !g_coarse[i] || !g_fine[i]) return NSR_ERR_INVALID_ARG;
  if (R == 0) return NSR_OK;
  if (!rays || !target_lr || !outs[0] || !outs[4] || !lr_coarse || !lr_fine || !losses || !workspace)
    return NSR_ERR_INVALID_ARG;
  if ((reinterpret_cast<uintptr_t>(workspace) & 255) != 0) return NSR_ERR_INVALID_ARG;
  // the step's option word (include/nsr_train.h): the renderer's two bits + the colour head's two
  constexpr int kTrainOpts = NSR_WHITE_BKGD | NSR_SIGMA_SOFTPLUS | NSR_TRAIN_GAMMA_CORRECT | NSR_TRAIN_COLOR_NONE | NSR_TRAIN_STOP_GRAD;
  if ((white_bkgd & ~kTrainOpts) != 0) return NSR_ERR_INVALID_ARG;
  const int gamma = (white_bkgd & NSR_TRAIN_GAMMA_CORRECT) != 0, color_none = (white_bkgd & NSR_TRAIN_COLOR_NONE) != 0;
  const int stop_grad = (white_bkgd & NSR_TRAIN_STOP_GRAD) != 0;
  if (gamma && color_none) return NSR_ERR_UNSUPPORTED;   // pow(x, 1 / 2.2) of an unbounded head: NaN for every negative value
  if (workspace_bytes < nsr_train_workspace_bytes_for(precision, ray_chunk, n_coarse, n_importance)) return NSR_ERR_WORKSPACE;
  const bool noisy = noise_std > 0.0f;
  hipStream_t st = nsr_stream(stream);
  Work k;
  work_floats(ray_chunk, n_coarse, n_importance, &k, static_cast<float*>(workspace), chain_selected(precision) ? 2 : 1);
  const int nc = n_coarse, nf = n_coarse + n_importance;
  const int64_t n_lr_total = R / s2;
  const double mse_scale = 1.0 / (3.0 * (double)n_lr_total);

  const bool chain = chain_selected(precision);
  if (chain) {
    // the weights are re-packed every iteration: a run whose weights drift beyond what the split-fp16 stream carries
    // (|w| >= 1023.75, or NaN) raises NSR_FLAG_WEIGHT_RANGE in the step's status word, like nsr_pack_weights does.
    // Round 6: both networks per launch (6 launches -> 3), and the backward pack also clears the loss carries and writes
    // word 1 of the status block = the colour-head option word the TRAIN instantiation of the forward kernel reads (the
    // blob tail's layout, nsr_common.h; written every call, so a workspace that was never reset cannot switch an option on)
    NSR_TRY(nsr_check_weights_range2(w_coarse, w_fine, NSR_F16X3, k.status, stream));
    NSR_TRY(nsr_f16x3_pack2(w_coarse, k.stream_f[0], w_fine, k.stream_f[1], stream));
    NSR_TRY(nsr_chain_bwd_pack2(w_coarse, k.stream_b[0], w_fine, k.stream_b[1], stop_grad, chain_bwd_terms(precision), k.carry, 8,
                                k.status + 1, color_none ? kOptColorNone : 0u, stream));
  } else {
    NSR_TRY(prepare_weights(st, w_coarse, k.pack[0], gemm_precision(precision)));
    NSR_TRY(prepare_weights(st, w_fine, k.pack[1], gemm_precision(precision)));
    if (hipMemsetAsync(k.carry, 0, 8 * sizeof(double), st) != hipSuccess) return NSR_ERR_LAUNCH;
  }

  for (int64_t r0 = 0; r0 < R; r0 += ray_chunk) {
    const int64_t rc = (R - r0 < ray_chunk) ? R - r0 : ray_chunk;
    const int acc = r0 > 0;
    const float* rays_c = rays + r0 * ray_stride;
    const int64_t lr0 = r0 / s2, n_lr = rc / s2;
    for (int net = 0; net < 2; ++net) {
      const int N = net ? nf : nc;
      const int64_t P = rc * N;
      const float* const* w = net ? w_fine : w_coarse;
      float* const* g = net ? g_fine : g_coarse;
      float* z = net ? k.z_f : k.z_c;
      if (net == 0) {
        NSR_TRY(nsr_sample_along_rays(rays_c, ray_stride, rc, nc, lindisp, u_coarse ? u_coarse + r0 * nc : nullptr, z,
                                      nullptr, stream));
      } else {
        const float* wc = outs[3] ? outs[3] + r0 * nc : k.w_c;
        NSR_TRY(nsr_resample_along_rays(rays_c, ray_stride, k.z_c, wc, rc, nc, n_importance,
                                        u_fine ? u_fine + r0 * n_importance : nullptr, z, nullptr, stream));
      }
      if (!chain) {   // the chain path encodes inside its forward kernel
        hipLaunchKernelGGL(encode_train_kernel, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, st, rays_c, ray_stride, z,
                           P, N, k.x5, k.gs);
        NSR_CHECK_LAUNCH();
      }
      if (chain) NSR_TRY(nsr_f16x3_train_forward(k.stream_f[net], rays_c, ray_stride, z, rc, N, k.rgb, k.zpan, k.sgn, k.status, stream));
      else NSR_TRY(net_forward(st, w, k.pack[net], k, P, gemm_precision(precision), color_none));
      const float* noise = net ? noise_fine : noise_coarse;
      hipLaunchKernelGGL(sigma_noise_kernel, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, st,
                         chain ? k.rgb + 3 : k.gs + kSigmaCol, chain ? 4 : kGs, (noisy && noise) ? noise + r0 * N : nullptr,
                         noise_std, P, k.sig);
      NSR_CHECK_LAUNCH();
      float* comp = outs[4 * net + 0] + r0 * 3;
      float* depth = outs[4 * net + 1] ? outs[4 * net + 1] + r0 : (depth_var ? k.scratch_out : nullptr);   // the depth-variance loss reads it
      float* opac = outs[4 * net + 2] ? outs[4 * net + 2] + r0 : nullptr;
      float* wts = outs[4 * net + 3] ? outs[4 * net + 3] + r0 * N : (net ? nullptr : k.w_c);
      if (gamma) {   // render_rays: out_rgbs = pow(out_rgbs, 1 / 2.2) per sample, in training too (nerf_downX_model.py:271-276)
        hipLaunchKernelGGL(gamma_points_kernel, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, st, k.rgb, P);
        NSR_CHECK_LAUNCH();
      }
      NSR_TRY(nsr_composite(k.rgb, 4, k.sig, 1, z, rc, N, white_bkgd & (NSR_WHITE_BKGD | NSR_SIGMA_SOFTPLUS), comp, depth, opac, wts, stream));
      // s^2 mean, loss, dL/d(comp)
      const float lambda = net ? lambda_fine : lambda_coarse;
      const int nblk = (int)((n_lr + 255) / 256);
      const float l_var = var ? (net ? var->lambda_fine_var : var->lambda_coarse_var) : 0.0f;
      const float l_dvar = var ? (net ? var->lambda_fine_depth_var : var->lambda_coarse_depth_var) : 0.0f;
      hipLaunchKernelGGL(lr_loss_kernel, dim3(nblk), dim3(256), 0, st, comp, target_lr + lr0 * 3, n_lr, s2, mse_scale,
                         lambda, (net ? lr_fine : lr_coarse) + lr0 * 3, k.g_comp, k.block_sums, l_var, l_dvar,
                         var ? var->far : 1.0f, depth_var ? depth : nullptr, depth_var ? k.g_depth : nullptr);
      NSR_CHECK_LAUNCH();
      hipLaunchKernelGGL(loss_finish_kernel, dim3(1), dim3(256), 0, st, k.block_sums, nblk, mse_scale, lambda, losses, net,
                         k.carry, l_var, l_dvar, var_losses);
      NSR_CHECK_LAUNCH();
      NSR_TRY(composite_bwd(st, k, z, rc, N, white_bkgd, chain, depth_var ? k.g_depth : nullptr));
      if (chain) {
        NSR_TRY(nsr_chain_bwd(k.stream_b[net], k.sgn, k.dpan, k.d4, 4, k.d4 + 3, 4, P, k.gmax, k.pscale, chain_bwd_terms(precision), 1, stream));
        NSR_TRY(chain_weight_grads(st, k, P, rc, g, acc));
      } else {
        NSR_TRY(net_backward(st, w, k.pack[net], k, P, g, acc, stop_grad));
      }
    }
  }
  return NSR_OK;
}
}  // namespace

extern "C" int nsr_train_loss_and_grads(const float* const* w_coarse, const float* const* w_fine, float* const* g_coarse,
                                        float* const* g_fine, const float* rays, int ray_stride, int64_t R, int s2,
                                        const float* target_lr, int n_coarse, int n_importance, int white_bkgd,
                                        int lindisp, const float* u_coarse, const float* u_fine,
                                        const float* noise_coarse, const float* noise_fine, float noise_std,
                                        float lambda_coarse, float lambda_fine, int precision, int64_t ray_chunk, float* const* outs,
                                        float* lr_coarse, float* lr_fine, float* losses, void* workspace,
                                        size_t workspace_bytes, void* stream) {
  return train_impl(w_coarse, w_fine, g_coarse, g_fine, rays, ray_stride, R, s2, target_lr, n_coarse, n_importance, white_bkgd, lindisp,
                    u_coarse, u_fine, noise_coarse, noise_fine, noise_std, lambda_coarse, lambda_fine, precision, ray_chunk, outs,
                    lr_coarse, lr_fine, losses, workspace, workspace_bytes, stream, nullptr, nullptr);
}

extern "C" int nsr_train_loss_and_grads_var(const float* const* w_coarse, const float* const* w_fine, float* const* g_coarse,
                                            float* const* g_fine, const float* rays, int ray_stride, int64_t R, int s2,
                                            const float* target_lr, int n_coarse, int n_importance, int white_bkgd,
                                            int lindisp, const float* u_coarse, const float* u_fine,
                                            const float* noise_coarse, const float* noise_fine, float noise_std,
                                            float lambda_coarse, float lambda_fine, int precision, int64_t ray_chunk, float* const* outs,
                                            float* lr_coarse, float* lr_fine, float* losses, void* workspace,
                                            size_t workspace_bytes, void* stream, const nsr_train_var_losses* var, float* var_losses) {
  if (!var) return NSR_ERR_INVALID_ARG;
  return train_impl(w_coarse, w_fine, g_coarse, g_fine, rays, ray_stride, R, s2, target_lr, n_coarse, n_importance, white_bkgd, lindisp,
                    u_coarse, u_fine, noise_coarse, noise_fine, noise_std, lambda_coarse, lambda_fine, precision, ray_chunk, outs,
                    lr_coarse, lr_fine, losses, workspace, workspace_bytes, stream, var, var_losses);
}

extern "C" int nsr_train_status_reset(void* workspace, void* stream) {
  if (!workspace) return NSR_ERR_INVALID_ARG;
  if (hipMemsetAsync(workspace, 0, 64, nsr_stream(stream)) != hipSuccess) return NSR_ERR_LAUNCH;
  return NSR_OK;
}

extern "C" int nsr_train_status(void* workspace, int clear, unsigned* flags_out, void* stream) {
  if (!workspace || !flags_out) return NSR_ERR_INVALID_ARG;
  hipStream_t st = nsr_stream(stream);
  unsigned host = 0;
  if (hipMemcpyAsync(&host, workspace, sizeof(unsigned), hipMemcpyDeviceToHost, st) != hipSuccess) return NSR_ERR_LAUNCH;
  if (clear && hipMemsetAsync(workspace, 0, sizeof(unsigned), st) != hipSuccess) return NSR_ERR_LAUNCH;
  if (hipStreamSynchronize(st) != hipSuccess) return NSR_ERR_LAUNCH;
  *flags_out = host;
  return NSR_OK;
}

extern "C" int nsr_adam_step(float* const* w, const float* const* g, float* const* m, float* const* v, int step, float lr,
                             float beta1, float beta2, float eps, void* stream) {
  if (!w || !g || !m || !v || step < 1) return NSR_ERR_INVALID_ARG;
  AdamPtrs a;
  for (int i = 0; i < NSR_N_STATE_TENSORS; ++i) {
    if (!w[i] || !g[i] || !m[i] || !v[i]) return NSR_ERR_INVALID_ARG;
    a.w[i] = w[i]; a.g[i] = g[i]; a.m[i] = m[i]; a.v[i] = v[i];
  }
  // bias corrections in double like Python's floats, then one rounding to fp32
  const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
  const float step_size = (float)((double)lr / bc1), bc2_sqrt = (float)sqrt(bc2);
  hipLaunchKernelGGL(adam_kernel, dim3(64, NSR_N_STATE_TENSORS), dim3(256), 0, nsr_stream(stream), a, beta1, beta2, eps,
                     step_size, bc2_sqrt);
  NSR_CHECK_LAUNCH();
  return NSR_OK;
}

extern "C" int nsr_split_weights(const float* w, int64_t n, void* w_hi, void* w_lo, void* stream) {
  return split_f16(w, n, static_cast<unsigned short*>(w_hi), static_cast<unsigned short*>(w_lo), nsr_stream(stream));
}

extern "C" int nsr_linear_f16x3(const float* x, int64_t ldx, const void* w_hi, const void* w_lo, int64_t ldw, const float* b, int act,
                                float* y, int64_t ldy, int64_t P, int K, int N, void* stream) {
  if (P < 0 || K <= 0 || N <= 0 || act < 0 || act > 2 || !w_hi || !w_lo) return NSR_ERR_INVALID_ARG;
  if (P == 0) return NSR_OK;
  GemmF16Args a{};
  a.g.A = x; a.g.lda = ldx; a.g.C = y; a.g.ldc = ldy; a.g.bias = b;
  a.g.M = P; a.g.N = N; a.g.K = K; a.g.n_valid = N; a.g.act = act; a.g.splits = 1;
  a.g.acc_scale = kSplitInvScale;
  a.Bh = static_cast<const unsigned short*>(w_hi);
  a.Bl = static_cast<const unsigned short*>(w_lo);
  a.ldbh = ldw;
  return gemm_f16x3(a, nsr_stream(stream));
}

extern "C" int nsr_linear(const float* x, int64_t ldx, const float* w, int64_t ldw, const float* b, int act, float* y,
                          int64_t ldy, float* y_t, int64_t ldyt, int64_t P, int K, int N, void* stream) {
  if (P < 0 || K <= 0 || N <= 0 || act < 0 || act > 2) return NSR_ERR_INVALID_ARG;
  if (P == 0) return NSR_OK;
  GemmArgs g{};
  g.A = x; g.lda = ldx; g.B = w; g.ldb = ldw; g.C = y; g.ldc = ldy; g.Ct = y_t; g.ldct = ldyt; g.bias = b;
  g.M = P; g.N = N; g.K = K; g.n_valid = N; g.act = act; g.splits = 1;
  return gemm(g, nsr_stream(stream));
}
