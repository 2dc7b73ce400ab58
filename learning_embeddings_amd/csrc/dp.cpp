// Data-parallel gradient exchange behind the C ABI: a thin layer over RCCL (xGMI inside a node).
//
// Replaces the reference's single-process nn.DataParallel (oe_h.py:301,1434,1439; ethec_experiments.py:240): broadcast of all
// parameters every forward, gather on device 0, reduce-add of the replica gradients there.  Here every rank (one process per
// GPU) owns a full replica whose gradients live in one flat arena, and the exchange is ONE sum all-reduce per bucket of that arena
// on a caller-given stream: lec_dp_allreduce_sum(comm, arena_ptr + offset, count, dtype, stream).  Reduction is a SUM (the
// reference loss is a plain sum over pairs, oe_h.py:843-846).
//
// RCCL is resolved at run time with dlopen/dlsym: the process that loads liblecone.so normally has PyTorch's own librccl.so mapped
// already, and a second copy linked against /opt/rocm's would own a second set of communicators and IPC state.  RTLD_NOLOAD picks
// up the mapped copy; only a process without one loads the system library.
#include <dlfcn.h>
#include <cstdint>
#include <cstring>
#include <new>
#include <hip/hip_runtime_api.h>
#include "../../include/lecone.h"

namespace lec {
void set_error(const char* fmt, ...);
int hip_fail(hipError_t e, const char* what);

// the slice of rccl.h this file needs (RCCL keeps NCCL's ABI: ncclUniqueId is 128 bytes, results are ints, 0 = success)
struct NcclUniqueId { char internal[128]; };
typedef void* NcclComm;
enum { kNcclFloat32 = 7, kNcclBfloat16 = 9, kNcclSum = 0 };
struct Rccl {
  int (*GetUniqueId)(NcclUniqueId*) = nullptr;
  int (*CommInitRank)(NcclComm*, int, NcclUniqueId, int) = nullptr;
  int (*CommDestroy)(NcclComm) = nullptr;
  int (*AllReduce)(const void*, void*, size_t, int, int, NcclComm, hipStream_t) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
  bool ok = false;
};
static Rccl load_rccl() {
  Rccl r;
  void* h = nullptr;
  const char* names[] = {"librccl.so", "librccl.so.1"};
  for (const char* n : names) if (!h) h = dlopen(n, RTLD_NOW | RTLD_NOLOAD);      // the copy torch mapped
  for (const char* n : names) if (!h) h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
  if (!h) return r;
  r.GetUniqueId = (int (*)(NcclUniqueId*))dlsym(h, "ncclGetUniqueId");
  r.CommInitRank = (int (*)(NcclComm*, int, NcclUniqueId, int))dlsym(h, "ncclCommInitRank");
  r.CommDestroy = (int (*)(NcclComm))dlsym(h, "ncclCommDestroy");
  r.AllReduce = (int (*)(const void*, void*, size_t, int, int, NcclComm, hipStream_t))dlsym(h, "ncclAllReduce");
  r.GetErrorString = (const char* (*)(int))dlsym(h, "ncclGetErrorString");
  r.ok = r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.AllReduce;
  return r;
}
static Rccl& rccl() {
  static Rccl r = load_rccl();            // function-local static: initialised exactly once, thread-safe (C++11)
  return r;
}
static int rccl_fail(int rc, const char* what) {
  Rccl& r = rccl();
  set_error("%s: RCCL error %d (%s)", what, rc, r.GetErrorString ? r.GetErrorString(rc) : "?");
  return LEC_E_HIP;
}
}  // namespace lec

struct lec_dp { lec::NcclComm comm = nullptr; int rank = 0, world = 1, device = 0; };

extern "C" int lec_dp_unique_id(void* id128) {
  using namespace lec;
  if (!id128) { set_error("dp_unique_id: null pointer"); return LEC_E_ARG; }
  Rccl& r = rccl();
  if (!r.ok) { set_error("dp_unique_id: librccl.so not found"); return LEC_E_STATE; }
  NcclUniqueId id;
  if (int rc = r.GetUniqueId(&id)) return rccl_fail(rc, "ncclGetUniqueId");
  std::memcpy(id128, &id, sizeof(id));
  return LEC_OK;
}

extern "C" int lec_dp_init(lec_dp** out, int rank, int world, const void* unique_id, int device) {
  using namespace lec;
  if (!out || !unique_id || world < 1 || rank < 0 || rank >= world || device < 0) { set_error("dp_init: bad arguments (rank %d of %d, device %d)", rank, world, device); return LEC_E_ARG; }
  Rccl& r = rccl();
  if (!r.ok) { set_error("dp_init: librccl.so not found"); return LEC_E_STATE; }
  // ncclCommInitRank binds the communicator to the CURRENT device: switch to `device` for the call and put the caller's device back
  int prev = -1;
  if (hipError_t e = hipGetDevice(&prev)) return hip_fail(e, "dp_init: hipGetDevice");
  if (hipError_t e = hipSetDevice(device)) return hip_fail(e, "dp_init: hipSetDevice");
  lec_dp* d = new (std::nothrow) lec_dp();
  if (!d) { (void)hipSetDevice(prev); set_error("dp_init: out of memory"); return LEC_E_STATE; }
  NcclUniqueId id; std::memcpy(&id, unique_id, sizeof(id));
  const int rc = r.CommInitRank(&d->comm, world, id, rank);
  (void)hipSetDevice(prev);
  if (rc) { delete d; return rccl_fail(rc, "ncclCommInitRank"); }
  d->rank = rank; d->world = world; d->device = device;
  *out = d;
  return LEC_OK;
}

extern "C" int lec_dp_allreduce_sum(lec_dp* d, void* buf, int64_t count, int dtype, lec_stream_t stream) {
  using namespace lec;
  if (!d || !d->comm) { set_error("dp_allreduce_sum: communicator not initialised"); return LEC_E_STATE; }
  if (!buf || count < 0 || (dtype != 0 && dtype != 1)) { set_error("dp_allreduce_sum: bad arguments (count %lld, dtype %d)", (long long)count, dtype); return LEC_E_ARG; }
  if (count == 0) return LEC_OK;
  if (int rc = rccl().AllReduce(buf, buf, (size_t)count, dtype == 0 ? kNcclFloat32 : kNcclBfloat16, kNcclSum, d->comm, (hipStream_t)stream))
    return rccl_fail(rc, "ncclAllReduce");
  return LEC_OK;
}

extern "C" void lec_dp_destroy(lec_dp* d) {
  if (!d) return;
  if (d->comm && lec::rccl().ok) lec::rccl().CommDestroy(d->comm);
  delete d;
}
