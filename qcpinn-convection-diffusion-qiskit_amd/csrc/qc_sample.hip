// Collocation-point sampler: the three uniform batches of one training step in one launch.
//
// Replaces the three torch.rand draws + affine maps of the reference step
// (trainer/diffusion_train.py:9-20 boxes, :34-36 order IC -> BC1 -> residual; data/diffusion_dataset.py:12-19
// x = lo + (hi - lo) * rand).  Counter-based Philox4x32-10 keyed by (seed, step, batch id) and indexed
// by the GLOBAL point index, so a data-parallel run draws exactly the points the single-GPU run draws
// (each rank fills its shard of the same global batch) and no generator state lives on the host.
#include "qc_internal.h"
#include "qc_philox.h"

namespace {

// segment 0: residual points in [0,1]^3; 1: IC points (t = 0); 2: boundary points: the x = 0 face
// (trainer/diffusion_train.py:13-16), or with face_pts > 0 the four faces x=0, x=1, y=0, y=1 of the second
// workload (train_hybrid_qpinn.py:166-176), face = global index / face_pts
__global__ void __launch_bounds__(256) k_sample(float* __restrict__ X_res, int64_t n_res, int64_t off_res,
                                                float* __restrict__ X_val, int64_t n_ic, int64_t off_ic,
                                                int64_t n_bc, int64_t off_bc, int64_t face_pts, uint64_t seed,
                                                uint64_t step) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  int seg;
  int64_t local, gidx;
  float* dst;
  if (i < n_res) {
    seg = 0; local = i; gidx = off_res + local; dst = X_res + local * 3;
  } else if (i < n_res + n_ic) {
    seg = 1; local = i - n_res; gidx = off_ic + local; dst = X_val + local * 3;
  } else if (i < n_res + n_ic + n_bc) {
    seg = 2; local = i - n_res - n_ic; gidx = off_bc + local; dst = X_val + (n_ic + local) * 3;
  } else {
    return;
  }
  float t, x, y;
  qc_draw_point(seg, gidx, face_pts, seed, step, t, x, y);
  dst[0] = t;
  dst[1] = x;
  dst[2] = y;
}

}  // namespace

int qc_sample_launch(float* X_res, int64_t n_res, int64_t off_res, float* X_val, int64_t n_ic, int64_t off_ic,
                     int64_t n_bc, int64_t off_bc, int64_t bc_face_points, uint64_t seed, uint64_t step, hipStream_t st) {
  const int64_t total = n_res + n_ic + n_bc;
  if (total <= 0) return QC_OK;
  hipLaunchKernelGGL(k_sample, dim3(qc_ceil_div(total, 256)), dim3(256), 0, st, X_res, n_res, off_res, X_val, n_ic,
                     off_ic, n_bc, off_bc, bc_face_points, seed, step);
  return QC_OK;
}
