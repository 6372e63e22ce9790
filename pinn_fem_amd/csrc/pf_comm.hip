// pf_comm.hip — the sharded GD iteration driven from C with an own RCCL communicator.
//
// No reference analogue (the reference is single-process).  pinn_fem_amd/dist.py holds the
// same schedule in Python over torch.distributed (used with gloo on CPU and for several ranks sharing one
// GPU in tests, 44-46 us of host time per iteration); the product path issues the ~11 kernels and the ONE
// ncclAllReduce of an iteration from here (27-29 us of host time per iteration, measured on MI355X): no Python
// in the loop.
//
// librccl is dlopen'ed from the path the host passes (the one PyTorch already loaded: same library
// instance, second communicator); there is no link-time dependency on it.
#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <rccl/rccl.h>
#include "pf_common.h"

struct pf_comm {
  void* dl;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*);
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int);
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t);
  ncclResult_t (*CommDestroy)(ncclComm_t);
  ncclResult_t (*CommAbort)(ncclComm_t);
  ncclResult_t (*CommCount)(const ncclComm_t, int*);
  ncclResult_t (*CommUserRank)(const ncclComm_t, int*);
  const char* (*GetErrorString)(ncclResult_t);
  ncclComm_t comm;
  int rank, world;
};

static int comm_fail(int code, const char* what, const char* detail) {
  char buf[400];
  snprintf(buf, sizeof(buf), "%s: %s", what, detail ? detail : "");
  pf_set_error(buf);
  return code;
}

static int load_rccl(pf_comm* c, const char* path) {
  c->dl = dlopen(path && path[0] ? path : "librccl.so", RTLD_NOW | RTLD_GLOBAL);
  if (!c->dl) return comm_fail(PF_ERR_HIP, "dlopen(librccl)", dlerror());
#define PF_SYM(field, name)                                                     \
  *(void**)(&c->field) = dlsym(c->dl, name);                                    \
  if (!c->field) return comm_fail(PF_ERR_HIP, "dlsym", name);
  PF_SYM(GetUniqueId, "ncclGetUniqueId")
  PF_SYM(CommInitRank, "ncclCommInitRank")
  PF_SYM(AllReduce, "ncclAllReduce")
  PF_SYM(CommDestroy, "ncclCommDestroy")
  PF_SYM(CommAbort, "ncclCommAbort")
  PF_SYM(CommCount, "ncclCommCount")
  PF_SYM(CommUserRank, "ncclCommUserRank")
  PF_SYM(GetErrorString, "ncclGetErrorString")
#undef PF_SYM
  return PF_OK;
}

extern "C" {

int pf_comm_unique_id(const char* librccl_path, void* id_out) {
  if (!id_out) return comm_fail(PF_ERR_ARG, "pf_comm_unique_id", "null id_out");
  pf_comm c;
  memset(&c, 0, sizeof(c));
  int rc = load_rccl(&c, librccl_path);
  if (rc) return rc;
  ncclUniqueId id;
  const ncclResult_t r = c.GetUniqueId(&id);
  if (r != ncclSuccess) return comm_fail(PF_ERR_HIP, "ncclGetUniqueId", c.GetErrorString(r));
  memcpy(id_out, &id, sizeof(id));
  return PF_OK;     // the dlopen handle stays: the library is resident for the process lifetime anyway
}

int pf_comm_create(const char* librccl_path, const void* id, int rank, int world, void** comm_out) {
  if (!id || !comm_out || world < 1 || rank < 0 || rank >= world)
    return comm_fail(PF_ERR_ARG, "pf_comm_create", "bad argument");
  pf_comm* c = new pf_comm;
  memset(c, 0, sizeof(*c));
  int rc = load_rccl(c, librccl_path);
  if (rc) { delete c; return rc; }
  ncclUniqueId uid;
  memcpy(&uid, id, sizeof(uid));
  const ncclResult_t r = c->CommInitRank(&c->comm, world, uid, rank);   // collective: every rank calls it
  if (r != ncclSuccess) {
    rc = comm_fail(PF_ERR_HIP, "ncclCommInitRank", c->GetErrorString(r));
    delete c;
    return rc;
  }
  c->rank = rank;
  c->world = world;
  *comm_out = c;
  return PF_OK;
}

// what the communicator itself reports (ncclCommUserRank / ncclCommCount), for the bench line
int pf_comm_info(void* comm, int* rank_out, int* nranks_out) {
  if (!comm || !rank_out || !nranks_out) return comm_fail(PF_ERR_ARG, "pf_comm_info", "bad argument");
  pf_comm* c = (pf_comm*)comm;
  ncclResult_t r = c->CommUserRank(c->comm, rank_out);
  if (r == ncclSuccess) r = c->CommCount(c->comm, nranks_out);
  if (r != ncclSuccess) return comm_fail(PF_ERR_HIP, "ncclCommCount", c->GetErrorString(r));
  return PF_OK;
}

int pf_comm_destroy(void* comm) {
  if (!comm) return PF_OK;
  pf_comm* c = (pf_comm*)comm;
  c->CommDestroy(c->comm);
  delete c;
  return PF_OK;
}

// After a rank-local failure: tear the communicator down WITHOUT waiting for outstanding collectives (ncclCommAbort);
// peers blocked in a collective this rank will never join then fail instead of hanging.  Never synchronise the device
// before this call: an unmatched collective already on the stream would never complete.
int pf_comm_abort(void* comm) {
  if (!comm) return PF_OK;
  pf_comm* c = (pf_comm*)comm;
  c->CommAbort(c->comm);
  delete c;
  return PF_OK;
}

// sum over ranks of buf[0..n), in place, on the compute stream.  The interface vectors are a few hundred
// bytes to ~4 kB: the collective is latency bound, and hiding it on a second stream costs more than it
// saves on this runtime (cross-stream event pairs between eager launches: ~9 us each, four per iteration;
// measured with PF_COMM_MODE-style variants on MI355X: 0.261 ms overlapped vs 0.228 ms in stream order).
static int all_reduce(pf_comm* c, float* buf, size_t n, hipStream_t s) {
  const ncclResult_t r = c->AllReduce(buf, buf, n, ncclFloat, ncclSum, c->comm, s);
  if (r != ncclSuccess) return comm_fail(PF_ERR_HIP, "ncclAllReduce", c->GetErrorString(r));
  return PF_OK;
}

}  // extern "C" (re-opened below)
extern "C" {
// sum over ranks of buf[0..n), in place, on `stream`: the driver's collective on its own (the host checks it
// against torch.distributed's result before trusting the communicator)
int pf_comm_all_reduce(void* comm, float* buf, int n, void* stream) {
  if (!comm || !buf || n < 0) return comm_fail(PF_ERR_ARG, "pf_comm_all_reduce", "bad argument");
  return all_reduce((pf_comm*)comm, buf, (size_t)n, (hipStream_t)stream);
}
}
extern "C" {

#define PF_RUN(expr)          \
  do {                        \
    int rc__ = (expr);        \
    if (rc__ != PF_OK) return rc__; \
  } while (0)

}  // extern "C"
// flush: the last iteration's sum u^2, reduced in the (free) last slot of buf — u2_local keeps the LOCAL sum, the next
// chunk's first all-reduce carries it again
static int shard_flush(const pf_problem* p, pf_comm* c, float* buf, size_t n, const float* u2_local, hipStream_t s) {
  float* slot = buf + n - 1;
  if (hipMemcpyAsync(slot, u2_local, sizeof(float), hipMemcpyDeviceToDevice, s) != hipSuccess)
    return comm_fail(PF_ERR_HIP, "pf_shard_iterations", "hipMemcpyAsync failed");
  PF_RUN(all_reduce(c, slot, 1, s));
  PF_RUN(pf_shard_flush(p, slot, (void*)s));
  return PF_OK;
}
extern "C" {

// n_iter sharded iterations and the closing flush, everything in stream order on `stream`:
//   forward, backward (+ pack), interior update  ->  all-reduce(buf)  ->  interface update + bookkeeping
// buf = [iface grad_u | grad_theta | r2, d2, u2 of the previous iteration], u2_local = this rank's last sum u_free^2.
// ONE collective per iteration (a few hundred bytes to ~4 kB: latency bound); nothing here waits for the device.
int pf_shard_iterations(const pf_problem* p, void* comm, int n_iter, float* buf, float* u2_local, void* stream) {
  if (!p || !comm || !buf || !u2_local || n_iter < 0) return comm_fail(PF_ERR_ARG, "pf_shard_iterations", "bad argument");
  pf_comm* c = (pf_comm*)comm;
  hipStream_t s = (hipStream_t)stream;
  const size_t n = (size_t)p->n_iface + (size_t)p->n_theta_active + 3;
  for (int it = 0; it < n_iter; ++it) {
    PF_RUN(pf_shard_forward(p, stream));
    PF_RUN(pf_shard_backward(p, buf, u2_local, stream));
    PF_RUN(pf_shard_update_interior(p, stream));
    PF_RUN(all_reduce(c, buf, n, s));
    PF_RUN(pf_shard_update_shared(p, buf, u2_local, stream));
  }
  return n_iter > 0 ? shard_flush(p, c, buf, n, u2_local, s) : PF_OK;
}

// The same iterations with whole multiples of `iters_per_graph` replayed from a graph that holds the kernels AND the
// collective (pf_shard_graph_create), the remainder launched as above, then the flush.
int pf_shard_iterations_graph(const pf_problem* p, void* comm, void* graph, int iters_per_graph, int n_iter, float* buf,
                              float* u2_local, void* stream) {
  if (!graph || iters_per_graph < 1) return comm_fail(PF_ERR_ARG, "pf_shard_iterations_graph", "bad argument");
  if (!p || !comm || !buf || !u2_local || n_iter < 0) return comm_fail(PF_ERR_ARG, "pf_shard_iterations_graph", "bad argument");
  pf_comm* c = (pf_comm*)comm;
  hipStream_t s = (hipStream_t)stream;
  const size_t n = (size_t)p->n_iface + (size_t)p->n_theta_active + 3;
  int left = n_iter;
  for (; left >= iters_per_graph; left -= iters_per_graph) PF_RUN(pf_graph_launch(graph, stream));
  for (; left > 0; --left) {
    PF_RUN(pf_shard_forward(p, stream));
    PF_RUN(pf_shard_backward(p, buf, u2_local, stream));
    PF_RUN(pf_shard_update_interior(p, stream));
    PF_RUN(all_reduce(c, buf, n, s));
    PF_RUN(pf_shard_update_shared(p, buf, u2_local, stream));
  }
  return n_iter > 0 ? shard_flush(p, c, buf, n, u2_local, s) : PF_OK;
}

// iters_per_graph sharded iterations, collective included, as one hipGraph (handle for pf_graph_destroy).  It bakes in
// *p, buf, u2_local and the communicator.  Every rank must create and replay it alike (the collective is inside).
int pf_shard_graph_create(const pf_problem* p, void* comm, int iters_per_graph, float* buf, float* u2_local, void* stream,
                          void** graph_out) {
  if (!comm) return comm_fail(PF_ERR_ARG, "pf_shard_graph_create", "null communicator");
  return pf_shard_graph_capture(p, iters_per_graph, buf, u2_local, (hipStream_t)stream,
                                [](void* ctx, float* b, size_t n, hipStream_t s) { return all_reduce((pf_comm*)ctx, b, n, s); },
                                comm, graph_out);
}

}  // extern "C"
