// Data-parallel collective of the training step behind the C ABI: one RCCL all-reduce (sum, fp32, in place) of the
// flat [gradient | L_r, L_bc, L_ic] vector per step over xGMI (the reference itself is single-process
// and has no counterpart).  A caller that is not PyTorch binds these four entry points instead of torch.distributed;
// with a communicator in qc_step_desc the whole step (gradients -> all-reduce -> optimiser) is ONE library call on ONE
// stream: no second host call, no cross-stream events around the 3 KB payload.
//
// librccl is loaded on first use (dlopen), not linked: single-GPU users of libqcpinn_hip.so do not depend on it.
#include "qc_internal.h"
#include "../../include/qcpinn_hip.h"

#include <dlfcn.h>
#include <string.h>

#include <mutex>

namespace {

// the slice of the RCCL API used here (rccl.h: ncclResult_t = int, ncclFloat32 = 7, ncclSum = 0)
struct UniqueId {
  char internal[128];
};
typedef int (*fn_get_unique_id)(UniqueId*);
typedef int (*fn_comm_init_rank)(void** comm, int nranks, UniqueId id, int rank);
typedef int (*fn_comm_destroy)(void* comm);
typedef int (*fn_all_reduce)(const void* send, void* recv, size_t count, int dtype, int op, void* comm, hipStream_t st);

struct Rccl {
  void* handle = nullptr;
  fn_get_unique_id get_unique_id = nullptr;
  fn_comm_init_rank comm_init_rank = nullptr;
  fn_comm_destroy comm_destroy = nullptr;
  fn_all_reduce all_reduce = nullptr;
  bool ok = false, tried = false;
};

Rccl* rccl() {
  static Rccl r;
  static std::mutex mu;
  std::lock_guard<std::mutex> lk(mu);
  if (!r.tried) {
    r.tried = true;
    // a copy already mapped into the process (e.g. PyTorch's) first, then the ROCm installation
    const char* names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so.1"};
    for (const char* nm : names) {
      r.handle = dlopen(nm, RTLD_NOW | RTLD_NOLOAD);
      if (r.handle) break;
    }
    for (int i = 0; i < 3 && !r.handle; ++i) r.handle = dlopen(names[i], RTLD_NOW | RTLD_LOCAL);
    if (r.handle) {
      r.get_unique_id = (fn_get_unique_id)dlsym(r.handle, "ncclGetUniqueId");
      r.comm_init_rank = (fn_comm_init_rank)dlsym(r.handle, "ncclCommInitRank");
      r.comm_destroy = (fn_comm_destroy)dlsym(r.handle, "ncclCommDestroy");
      r.all_reduce = (fn_all_reduce)dlsym(r.handle, "ncclAllReduce");
      r.ok = r.get_unique_id && r.comm_init_rank && r.comm_destroy && r.all_reduce;
    }
  }
  return r.ok ? &r : nullptr;
}

}  // namespace

int qc_comm_allreduce(float* buf, int64_t count, void* comm, hipStream_t st) {
  Rccl* r = rccl();
  if (!r) return QC_ERR_UNSUPPORTED;
  return r->all_reduce(buf, buf, (size_t)count, 7 /* ncclFloat32 */, 0 /* ncclSum */, comm, st) == 0 ? QC_OK : QC_ERR_HIP;
}

extern "C" {

int qc_comm_unique_id(void* id_out) {
  if (!id_out) return QC_ERR_ARG;
  Rccl* r = rccl();
  if (!r) return QC_ERR_UNSUPPORTED;
  UniqueId id;
  if (r->get_unique_id(&id) != 0) return QC_ERR_HIP;
  memcpy(id_out, &id, sizeof(id));
  return QC_OK;
}

int qc_comm_create(const void* id_in, int world, int rank, void** comm_out) {
  if (!id_in || !comm_out || world < 1 || rank < 0 || rank >= world) return QC_ERR_ARG;
  Rccl* r = rccl();
  if (!r) return QC_ERR_UNSUPPORTED;
  UniqueId id;
  memcpy(&id, id_in, sizeof(id));
  void* c = nullptr;
  if (r->comm_init_rank(&c, world, id, rank) != 0 || !c) return QC_ERR_HIP;
  *comm_out = c;
  return QC_OK;
}

int qc_comm_destroy(void* comm) {
  if (!comm) return QC_ERR_ARG;
  Rccl* r = rccl();
  if (!r) return QC_ERR_UNSUPPORTED;
  return r->comm_destroy(comm) == 0 ? QC_OK : QC_ERR_HIP;
}

int qc_allreduce_grads(float* buf_dev, int64_t count, void* comm, void* stream) {
  if (!buf_dev || count <= 0 || !comm) return QC_ERR_ARG;
  return qc_comm_allreduce(buf_dev, count, comm, (hipStream_t)stream);
}

}  // extern "C"
