// comm.cpp -- RCCL transport of the public-pose exchange (C ABI dpgo_comm_*, include/dpgo_hip.h).
//
// The reference exchanges public poses by pointer calls inside one process (examples/MultiRobotExample.cpp:183-204:
// getSharedPoseDict / setNeighborStatus / updateNeighborPoses) and reduces cost / gradient norm on a central problem
// (:220-254).  With one agent block per GPU / process the same two steps are carried by RCCL over xGMI:
//   * dpgo_comm_exchange : ONE grouped batch (ncclGroupStart .. ncclGroupEnd) of ncclSend / ncclRecv of packed pose
//     tiles per exchange, enqueued on the caller's HIP stream -- the stream the pack kernel (k_gather_tiles) and the
//     coupling SpMM that consumes the tiles run on, so the host never waits;
//   * dpgo_comm_allreduce / dpgo_comm_broadcast : the few scalars of the reductions and the global anchor pose.
// RCCL is bound at run time (dlopen of librccl.so.1; a copy already loaded by the process, e.g. PyTorch-ROCm's, is
// reused), so the solver library itself has no link-time dependency on it.
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>

#include "../../include/dpgo_hip.h"

extern "C" void dpgo_set_last_error(const char* msg);  // dpgo_hip.hip (thread-local message of dpgo_last_error)

namespace {

// the slice of rccl.h this file uses (NCCL 2.x ABI: stable since 2.7)
typedef struct ncclComm* ncclComm_t;
typedef struct {
  char internal[128];
} ncclUniqueId;
typedef enum { ncclSuccess = 0 } ncclResult_t;
typedef enum { ncclFloat64 = 8 } ncclDataType_t;
typedef enum { ncclSum = 0, ncclProd = 1, ncclMax = 2, ncclMin = 3 } ncclRedOp_t;

struct Rccl {
  void* so = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  std::string error;
};

Rccl& rccl() {
  static Rccl r;
  static std::once_flag once;
  std::call_once(once, [&] {
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    // DPGO_RCCL_LIBRARY: load this file instead of the default names (also how the tests hide RCCL)
    const char* forced = std::getenv("DPGO_RCCL_LIBRARY");
    std::string why;
    if (forced && *forced) {
      r.so = dlopen(forced, RTLD_NOW | RTLD_GLOBAL);
      if (!r.so) {
        const char* e = dlerror();  // ONE call: dlerror() clears the message it returns
        why = e ? e : "not found";
      }
    } else {
      for (const char* nm : names) {  // a copy the process already holds first
        r.so = dlopen(nm, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);
        if (r.so) break;
      }
      for (const char* nm : names) {
        if (r.so) break;
        r.so = dlopen(nm, RTLD_NOW | RTLD_GLOBAL);
        if (!r.so) {
          const char* e = dlerror();
          why = e ? e : "not found";
        }
      }
    }
    if (!r.so) {
      r.error = std::string("cannot load RCCL (") + (forced && *forced ? forced : "librccl.so.1") + "): " + why;
      return;
    }
    bool ok = true;
    auto sym = [&](const char* nm) -> void* {
      void* p = dlsym(r.so, nm);
      if (!p) {
        ok = false;
        r.error = std::string("RCCL symbol missing: ") + nm;
      }
      return p;
    };
    r.GetUniqueId = (decltype(r.GetUniqueId))sym("ncclGetUniqueId");
    r.CommInitRank = (decltype(r.CommInitRank))sym("ncclCommInitRank");
    r.CommDestroy = (decltype(r.CommDestroy))sym("ncclCommDestroy");
    r.GroupStart = (decltype(r.GroupStart))sym("ncclGroupStart");
    r.GroupEnd = (decltype(r.GroupEnd))sym("ncclGroupEnd");
    r.Send = (decltype(r.Send))sym("ncclSend");
    r.Recv = (decltype(r.Recv))sym("ncclRecv");
    r.AllReduce = (decltype(r.AllReduce))sym("ncclAllReduce");
    r.Broadcast = (decltype(r.Broadcast))sym("ncclBroadcast");
    r.GetErrorString = (decltype(r.GetErrorString))sym("ncclGetErrorString");
    if (!ok) {
      r.so = nullptr;
    }
  });
  return r;
}

int fail(int code, const std::string& msg) {
  dpgo_set_last_error(msg.c_str());
  return code;
}

#define NCCLC(expr)                                                                                   \
  do {                                                                                                \
    ncclResult_t e_ = (expr);                                                                         \
    if (e_ != ncclSuccess)                                                                            \
      return fail(DPGO_ERR_HIP, std::string(#expr) + ": " + (R.GetErrorString ? R.GetErrorString(e_) : "RCCL error")); \
  } while (0)

}  // namespace

struct dpgo_comm_s {
  ncclComm_t comm = nullptr;
  int nranks = 0, rank = 0, device = 0;
};

extern "C" {

int dpgo_comm_unique_id(char id[DPGO_COMM_ID_BYTES]) {
  if (!id) return fail(DPGO_ERR_INVALID, "null id");
  Rccl& R = rccl();
  if (!R.so) return fail(DPGO_ERR_HIP, R.error);
  ncclUniqueId u;
  NCCLC(R.GetUniqueId(&u));
  static_assert(sizeof(u) == DPGO_COMM_ID_BYTES, "ncclUniqueId is 128 bytes");
  std::memcpy(id, &u, sizeof(u));
  return DPGO_OK;
}

int dpgo_comm_create(dpgo_comm_t* out, int nranks, int rank, const char id[DPGO_COMM_ID_BYTES], int device) {
  if (!out || !id || nranks < 1 || rank < 0 || rank >= nranks) return fail(DPGO_ERR_INVALID, "bad communicator arguments");
  *out = nullptr;
  Rccl& R = rccl();
  if (!R.so) return fail(DPGO_ERR_HIP, R.error);
  int cnt = 0;
  if (hipGetDeviceCount(&cnt) != hipSuccess || cnt <= 0)
    return fail(DPGO_ERR_HIP, "no HIP device (this library has no CPU fallback)");
  if (device < 0 || device >= cnt) return fail(DPGO_ERR_INVALID, "device index out of range");
  if (hipSetDevice(device) != hipSuccess) return fail(DPGO_ERR_HIP, "hipSetDevice failed");
  ncclUniqueId u;
  std::memcpy(&u, id, sizeof(u));
  auto* c = new dpgo_comm_s();
  c->nranks = nranks;
  c->rank = rank;
  c->device = device;
  ncclResult_t e = R.CommInitRank(&c->comm, nranks, u, rank);
  if (e != ncclSuccess) {
    delete c;
    return fail(DPGO_ERR_HIP, std::string("ncclCommInitRank: ") + (R.GetErrorString ? R.GetErrorString(e) : "RCCL error"));
  }
  *out = c;
  return DPGO_OK;
}

int dpgo_comm_destroy(dpgo_comm_t c) {
  if (!c) return DPGO_OK;
  Rccl& R = rccl();
  if (R.so && c->comm) (void)R.CommDestroy(c->comm);
  delete c;
  return DPGO_OK;
}

int dpgo_comm_info(dpgo_comm_t c, int* nranks, int* rank) {
  if (!c) return fail(DPGO_ERR_INVALID, "null communicator");
  if (nranks) *nranks = c->nranks;
  if (rank) *rank = c->rank;
  return DPGO_OK;
}

int dpgo_comm_exchange(dpgo_comm_t c, int nsend, const int* send_peer, const double* const* send_dev,
                       const int* send_count, int nrecv, const int* recv_peer, double* const* recv_dev,
                       const int* recv_count, void* stream) {
  if (!c) return fail(DPGO_ERR_INVALID, "null communicator");
  if (nsend < 0 || nrecv < 0 || (nsend > 0 && (!send_peer || !send_dev || !send_count)) ||
      (nrecv > 0 && (!recv_peer || !recv_dev || !recv_count)))
    return fail(DPGO_ERR_INVALID, "null message arrays");
  for (int k = 0; k < nsend; ++k)
    if (send_peer[k] < 0 || send_peer[k] >= c->nranks || send_count[k] < 0 || (send_count[k] > 0 && !send_dev[k]))
      return fail(DPGO_ERR_INVALID, "bad send message");
  for (int k = 0; k < nrecv; ++k)
    if (recv_peer[k] < 0 || recv_peer[k] >= c->nranks || recv_count[k] < 0 || (recv_count[k] > 0 && !recv_dev[k]))
      return fail(DPGO_ERR_INVALID, "bad receive message");
  if (nsend + nrecv == 0) return DPGO_OK;
  Rccl& R = rccl();
  if (hipSetDevice(c->device) != hipSuccess) return fail(DPGO_ERR_HIP, "hipSetDevice failed");
  hipStream_t s = (hipStream_t)stream;
  // one group: all sends and receives of the exchange progress together, no ordering between ranks can deadlock;
  // messages between one pair of ranks match in issue order (the plan lists them in the same order on both sides)
  NCCLC(R.GroupStart());
  // an error inside the bracket must not leave the communicator with an open group (later calls would queue forever):
  // remember the first failure, close the group, then report
  ncclResult_t first = ncclSuccess;
  const char* what = "";
  for (int k = 0; k < nsend && first == ncclSuccess; ++k)
    if (send_count[k] > 0) {
      first = R.Send(send_dev[k], (size_t)send_count[k], ncclFloat64, send_peer[k], c->comm, s);
      what = "ncclSend";
    }
  for (int k = 0; k < nrecv && first == ncclSuccess; ++k)
    if (recv_count[k] > 0) {
      first = R.Recv(recv_dev[k], (size_t)recv_count[k], ncclFloat64, recv_peer[k], c->comm, s);
      what = "ncclRecv";
    }
  const ncclResult_t end = R.GroupEnd();
  if (first != ncclSuccess)
    return fail(DPGO_ERR_HIP, std::string(what) + ": " + (R.GetErrorString ? R.GetErrorString(first) : "RCCL error"));
  if (end != ncclSuccess)
    return fail(DPGO_ERR_HIP, std::string("ncclGroupEnd: ") + (R.GetErrorString ? R.GetErrorString(end) : "RCCL error"));
  return DPGO_OK;
}

int dpgo_comm_allreduce(dpgo_comm_t c, double* buf_dev, int count, int op, void* stream) {
  if (!c || !buf_dev || count <= 0 || (op != DPGO_COMM_SUM && op != DPGO_COMM_MAX))
    return fail(DPGO_ERR_INVALID, "bad all-reduce arguments");
  Rccl& R = rccl();
  if (hipSetDevice(c->device) != hipSuccess) return fail(DPGO_ERR_HIP, "hipSetDevice failed");
  NCCLC(R.AllReduce(buf_dev, buf_dev, (size_t)count, ncclFloat64, op == DPGO_COMM_SUM ? ncclSum : ncclMax, c->comm,
                    (hipStream_t)stream));
  return DPGO_OK;
}

int dpgo_comm_broadcast(dpgo_comm_t c, double* buf_dev, int count, int root, void* stream) {
  if (!c || !buf_dev || count <= 0 || root < 0 || root >= c->nranks)
    return fail(DPGO_ERR_INVALID, "bad broadcast arguments");
  Rccl& R = rccl();
  if (hipSetDevice(c->device) != hipSuccess) return fail(DPGO_ERR_HIP, "hipSetDevice failed");
  NCCLC(R.Broadcast(buf_dev, buf_dev, (size_t)count, ncclFloat64, root, c->comm, (hipStream_t)stream));
  return DPGO_OK;
}

}  // extern "C"
