// D3: the fused forward_rays driver (models/nerf_downX_model.py:280-324, eval
// mode) — one enqueue sequence for the WHOLE ray batch: no ray_chunk /
// point_chunk slicing (288 GB of HBM hold every intermediate of a full image),
// no host synchronisation (the reference syncs twice per 4096-ray chunk,
// nerf_downX_model.py:273,284).
#include "nsr_common.h"

extern "C" int nsr_version(void) { return NSR_VERSION; }

extern "C" const char* nsr_status_string(int status) {
  switch (status) {
    case NSR_OK: return "ok";
    case NSR_ERR_INVALID_ARG: return "invalid argument (null / negative size / misaligned pointer)";
    case NSR_ERR_UNSUPPORTED: return "configuration outside the built path (sample count, degree or precision)";
    case NSR_ERR_LAUNCH: return "HIP kernel launch failed";
    case NSR_ERR_WORKSPACE: return "workspace too small";
    case NSR_ERR_RANGE: return "a weight is non-finite or outside the operand range of the precision (use NSR_FP32)";
    default: return "unknown nsr status";
  }
}

static inline size_t align256(size_t n) { return (n + 255) & ~(size_t)255; }

// the fused route (D2 + V1 in one launch, nsr_render_rays_composited) exists for these shapes; everything else takes the
// network launch followed by the stand-alone compositor and needs the (R, N, 4) raw tensors
static inline bool fused_route(int precision, int n_samples) {
  return (precision == NSR_FP32 || precision == NSR_F16X3) && (n_samples == 64 || n_samples == 128);
}

// workspace carve: z_coarse (R,Nc) | w_coarse (R,Nc) | z_fine (R,Nf) | [unfused route only: raw_coarse (R,Nc,4) | raw_fine (R,Nf,4)]
extern "C" size_t nsr_forward_rays_workspace_bytes_for(int precision, int64_t R, int n_coarse, int n_importance) {
  if (R < 0 || n_coarse <= 0 || n_importance < 0) return 0;
  const size_t r = (size_t)R, nc = (size_t)n_coarse, nf = (size_t)(n_coarse + n_importance);
  size_t total = align256(r * nc * 4) + align256(r * nc * 4);
  if (!fused_route(precision, n_coarse)) total += align256(r * nc * 16);
  if (n_importance > 0) {
    total += align256(r * nf * 4);
    if (!fused_route(precision, n_coarse + n_importance)) total += align256(r * nf * 16);
  }
  return total;
}
// precision-agnostic upper bound (the unfused route of any precision)
extern "C" size_t nsr_forward_rays_workspace_bytes(int64_t R, int n_coarse, int n_importance) {
  return nsr_forward_rays_workspace_bytes_for(-1, R, n_coarse, n_importance);
}

extern "C" int nsr_event_create(void** event_out) {
  if (!event_out) return NSR_ERR_INVALID_ARG;
  hipEvent_t ev;
  if (hipEventCreate(&ev) != hipSuccess) return NSR_ERR_LAUNCH;
  *event_out = ev;
  return NSR_OK;
}
extern "C" int nsr_event_destroy(void* event) {
  if (!event) return NSR_ERR_INVALID_ARG;
  return hipEventDestroy(static_cast<hipEvent_t>(event)) == hipSuccess ? NSR_OK : NSR_ERR_LAUNCH;
}
extern "C" int nsr_event_elapsed_ms(void* start, void* stop, float* ms_out) {
  if (!start || !stop || !ms_out) return NSR_ERR_INVALID_ARG;
  if (hipEventSynchronize(static_cast<hipEvent_t>(stop)) != hipSuccess) return NSR_ERR_LAUNCH;
  return hipEventElapsedTime(ms_out, static_cast<hipEvent_t>(start), static_cast<hipEvent_t>(stop)) == hipSuccess
             ? NSR_OK : NSR_ERR_LAUNCH;
}

extern "C" int nsr_forward_rays(const void* packed_coarse, const void* packed_fine, int precision, const float* rays,
                                int ray_stride, int64_t R, int n_coarse, int n_importance, int white_bkgd, int lindisp,
                                float* const* outs, void* workspace, size_t workspace_bytes, void* stream) {
  return nsr_forward_rays_profiled(packed_coarse, packed_fine, precision, rays, ray_stride, R, n_coarse, n_importance,
                                   white_bkgd, lindisp, outs, workspace, workspace_bytes, stream, nullptr);
}

extern "C" int nsr_forward_rays_profiled(const void* packed_coarse, const void* packed_fine, int precision,
                                         const float* rays, int ray_stride, int64_t R, int n_coarse, int n_importance,
                                         int white_bkgd, int lindisp, float* const* outs, void* workspace,
                                         size_t workspace_bytes, void* stream, void* const* events) {
  hipStream_t st = nsr_stream(stream);
  auto mark = [&](int i) {
    if (events && events[i]) (void)hipEventRecord(static_cast<hipEvent_t>(events[i]), st);
  };
  if (!packed_coarse || !outs || R < 0 || n_coarse <= 0 || n_importance < 0 || (white_bkgd & ~(NSR_WHITE_BKGD | NSR_SIGMA_SOFTPLUS)) != 0)
    return NSR_ERR_INVALID_ARG;
  if (n_importance > 0 && !packed_fine) return NSR_ERR_INVALID_ARG;
  if (workspace_bytes < nsr_forward_rays_workspace_bytes_for(precision, R, n_coarse, n_importance)) return NSR_ERR_WORKSPACE;
  if (R == 0) return NSR_OK;
  if (!rays || !workspace || (reinterpret_cast<uintptr_t>(workspace) & 255) != 0) return NSR_ERR_INVALID_ARG;
  const size_t r = (size_t)R, nc = (size_t)n_coarse, nf = (size_t)(n_coarse + n_importance);
  char* ws = static_cast<char*>(workspace);
  float* z_c = reinterpret_cast<float*>(ws);   ws += align256(r * nc * 4);
  float* w_c_ws = reinterpret_cast<float*>(ws); ws += align256(r * nc * 4);
  float* raw_c = nullptr;
  float* z_f = nullptr;
  float* raw_f = nullptr;
  if (!fused_route(precision, n_coarse)) { raw_c = reinterpret_cast<float*>(ws); ws += align256(r * nc * 16); }
  if (n_importance > 0) {
    z_f = reinterpret_cast<float*>(ws);   ws += align256(r * nf * 4);
    if (!fused_route(precision, n_coarse + n_importance)) { raw_f = reinterpret_cast<float*>(ws); ws += align256(r * nf * 16); }
  }
  int rc;
  // S1: coarse depths
  rc = nsr_sample_along_rays(rays, ray_stride, R, n_coarse, lindisp, nullptr, z_c, nullptr, stream);
  if (rc != NSR_OK) return rc;
  // D2+M1+V1: coarse network at every sample, composited by the same launch when the tile shape allows it (64 or 128
  // samples per ray, contract-grade precisions): the (R, N, 4) network output then never goes to HBM.  The coarse weights
  // are needed by the resampler even if the caller does not want them.
  float* w_c = outs[3] ? outs[3] : w_c_ws;
  mark(0);
  rc = nsr_render_rays_composited(packed_coarse, precision, rays, ray_stride, z_c, R, n_coarse, white_bkgd, nullptr, outs[0],
                                  outs[1], outs[2], w_c, stream);
  if (rc == NSR_ERR_UNSUPPORTED) {   // other sample counts / the single-operand fast paths: network, then compositor
    rc = nsr_render_rays(packed_coarse, precision, rays, ray_stride, z_c, R, n_coarse, raw_c, stream);
    mark(1);
    if (rc != NSR_OK) return rc;
    rc = nsr_composite(raw_c, 4, raw_c + 3, 4, z_c, R, n_coarse, white_bkgd, outs[0], outs[1], outs[2], w_c, stream);
  } else {
    mark(1);
  }
  if (rc != NSR_OK) return rc;
  if (n_importance == 0) return NSR_OK;
  // S2: importance resampling + merge
  rc = nsr_resample_along_rays(rays, ray_stride, z_c, w_c, R, n_coarse, n_importance, nullptr, z_f, nullptr, stream);
  if (rc != NSR_OK) return rc;
  // fine network + compositing
  mark(2);
  rc = nsr_render_rays_composited(packed_fine, precision, rays, ray_stride, z_f, R, n_coarse + n_importance, white_bkgd, nullptr,
                                  outs[4], outs[5], outs[6], outs[7], stream);
  if (rc != NSR_ERR_UNSUPPORTED) {
    mark(3);
    return rc;
  }
  rc = nsr_render_rays(packed_fine, precision, rays, ray_stride, z_f, R, n_coarse + n_importance, raw_f, stream);
  mark(3);
  if (rc != NSR_OK) return rc;
  rc = nsr_composite(raw_f, 4, raw_f + 3, 4, z_f, R, n_coarse + n_importance, white_bkgd, outs[4], outs[5], outs[6],
                     outs[7], stream);
  return rc;
}
