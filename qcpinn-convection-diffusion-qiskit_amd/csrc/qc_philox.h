// Counter-based Philox4x32-10 and the collocation-point draw shared by k_sample (qc_sample.hip) and the merged
// pre-forward stage of the fused step (qc_mlp.hip), so both produce bit-identical points for the same
// (seed, step, segment, global index).
#pragma once
#include <stdint.h>

#include <hip/hip_runtime.h>

namespace {

struct U4 {
  uint32_t x, y, z, w;
};

__device__ __forceinline__ U4 philox4x32_10(U4 c, uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint32_t hi0 = __umulhi(0xD2511F53u, c.x), lo0 = 0xD2511F53u * c.x;
    const uint32_t hi1 = __umulhi(0xCD9E8D57u, c.z), lo1 = 0xCD9E8D57u * c.z;
    c = {hi1 ^ c.y ^ k0, lo1, hi0 ^ c.w ^ k1, lo0};
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  return c;
}

__device__ __forceinline__ float u01(uint32_t v) { return (float)(v >> 8) * (1.0f / 16777216.0f); }  // [0,1)

// segment 0: residual points in [0,1]^3; 1: IC points (t = 0); 2: boundary points: the x = 0 face
// (trainer/diffusion_train.py:13-16), or with face_pts > 0 the four faces x=0, x=1, y=0, y=1 of the second
// workload (train_hybrid_qpinn.py:166-176), face = global index / face_pts
__device__ __forceinline__ void qc_draw_point(int seg, int64_t gidx, int64_t face_pts, uint64_t seed, uint64_t step, float& t,
                                              float& x, float& y) {
  const U4 ctr = {(uint32_t)gidx, (uint32_t)(gidx >> 32), (uint32_t)step, (uint32_t)(step >> 32) ^ ((uint32_t)seg << 30)};
  const U4 r = philox4x32_10(ctr, (uint32_t)seed, (uint32_t)(seed >> 32));
  t = seg == 1 ? 0.f : u01(r.x);
  x = u01(r.y);
  y = u01(r.z);
  if (seg == 2) {
    const int64_t face = face_pts > 0 ? gidx / face_pts : 0;
    if (face == 0) x = 0.f;
    else if (face == 1) x = 1.f;
    else if (face == 2) y = 0.f;
    else y = 1.f;
  }
}

struct QcDraw {      // on-device sampling of one launch's points (enabled = 0: read X instead)
  int enabled;
  int64_t n_ic;      // value tiles: leading IC points
  int64_t off_res, off_ic, off_bc, face_pts;
  uint64_t seed, step;
};

}  // namespace
