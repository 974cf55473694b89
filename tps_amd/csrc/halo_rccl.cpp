// libtpsrhs_rccl.so -- include/tpsrhs_rccl.h: the halo exchange and the scalar reductions of tpsrhs.h over RCCL
// (ncclGroupStart / ncclSend / ncclRecv / ncclGroupEnd on the operator's communication stream).  Replaces, for one
// process per GPU, the MPI_Isend / MPI_Irecv / MPI_Waitall of src/rhs_operator.cpp:775-831 and the MPI_Allreduce
// calls of src/outletBC.cpp:533-540 and src/M2ulPhyS.cpp:2013-2016.
#include "../../include/tpsrhs_rccl.h"

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdlib>
#include <cstring>
#include <set>
#include <string>

static_assert(sizeof(ncclUniqueId) == TPSRHS_RCCL_ID_BYTES, "ncclUniqueId size");

struct tpsrhs_rccl_ctx {
  ncclComm_t comm = nullptr;         // neighbour exchange (grouped send / recv on the operator's communication stream)
  ncclComm_t comm_reduce = nullptr;  // scalar reductions on the compute stream: a communicator of their own, so that RCCL
                                     // does not order them with the exchange in flight on the other stream
  bool own_reduce_comm = false;
  int nranks = 0, rank = 0, device = 0;
  int64_t halo_calls = 0, bytes_sent = 0;
  std::set<int> peers;
  int skip = 0;
};

static thread_local std::string g_err;

static int fail(const std::string &what, ncclResult_t r) {
  g_err = what + ": " + ncclGetErrorString(r);
  return 1;
}

extern "C" {

const char *tpsrhs_rccl_last_error(void) { return g_err.c_str(); }

int tpsrhs_rccl_unique_id(void *id) {
  ncclUniqueId u;
  const ncclResult_t r = ncclGetUniqueId(&u);
  if (r != ncclSuccess) return fail("ncclGetUniqueId", r);
  std::memcpy(id, &u, sizeof(u));
  return 0;
}

int tpsrhs_rccl_create(const void *id, int nranks, int rank, int device, tpsrhs_rccl_ctx **out) {
  if (!id || !out || nranks < 1 || rank < 0 || rank >= nranks) {
    g_err = "tpsrhs_rccl_create: invalid argument";
    return 1;
  }
  if (hipSetDevice(device) != hipSuccess) {
    g_err = "tpsrhs_rccl_create: hipSetDevice failed";
    return 1;
  }
  tpsrhs_rccl_ctx *c = new tpsrhs_rccl_ctx;
  c->nranks = nranks;
  c->rank = rank;
  c->device = device;
  ncclUniqueId u;
  std::memcpy(&u, id, sizeof(u));
  const ncclResult_t r = ncclCommInitRank(&c->comm, nranks, u, rank);
  if (r != ncclSuccess) {
    delete c;
    return fail("ncclCommInitRank", r);
  }
  // Operations on ONE communicator are serialised by RCCL whatever stream they are enqueued on; the boundary-mean /
  // dt reductions run on the compute stream while the trace exchange runs on the communication stream.  A communicator
  // of their own (ncclCommSplit: same ranks, same order) lifts that ordering -- but two communicators in flight on one
  // device are only deadlock-free if every rank enqueues their operations in the same relative order and both kernels
  // can be resident, and this path has not run on more than one GPU yet.  So the split is OPT-IN, TPSRHS_RCCL_SPLIT=1
  // (set identically on every rank: ncclCommSplit is collective), and its outcome is made collective: the ranks agree
  // (a MIN all-reduce of the success flag on the exchange communicator) that EVERY rank has its second communicator,
  // or every rank falls back to the shared one -- never a mixture, which would send one all-reduce to two communicators
  // and hang.  tpsrhs_rccl_comm_info reports which.
  c->comm_reduce = c->comm;
  const char *want = std::getenv("TPSRHS_RCCL_SPLIT");
  if (want && want[0] == '1') {
    ncclComm_t second = nullptr;
    int ok = (ncclCommSplit(c->comm, 0, rank, &second, nullptr) == ncclSuccess && second) ? 1 : 0;
    int *d_ok = nullptr;
    bool agreed = false;
    if (hipMalloc(reinterpret_cast<void **>(&d_ok), sizeof(int)) == hipSuccess) {
      if (hipMemcpy(d_ok, &ok, sizeof(int), hipMemcpyHostToDevice) == hipSuccess &&
          ncclAllReduce(d_ok, d_ok, 1, ncclInt, ncclMin, c->comm, nullptr) == ncclSuccess && hipStreamSynchronize(nullptr) == hipSuccess &&
          hipMemcpy(&ok, d_ok, sizeof(int), hipMemcpyDeviceToHost) == hipSuccess)
        agreed = true;
      hipFree(d_ok);
    }
    if (!agreed) {  // the agreement itself failed: the job cannot know what the other ranks do
      if (second) ncclCommDestroy(second);
      ncclCommDestroy(c->comm);
      delete c;
      g_err = "tpsrhs_rccl_create: the ranks could not agree on the outcome of ncclCommSplit";
      return 1;
    }
    if (ok) {
      c->comm_reduce = second;
      c->own_reduce_comm = true;
    } else if (second) {
      ncclCommDestroy(second);  // another rank was refused: every rank shares the exchange communicator
    }
  }
  *out = c;
  return 0;
}

int tpsrhs_rccl_comm_info(const tpsrhs_rccl_ctx *c, int *nranks, int *reduce_comm_is_separate) {
  if (!c || !c->comm) return 1;
  int n = 0;
  const ncclResult_t r = ncclCommCount(c->comm, &n);
  if (r != ncclSuccess) return fail("ncclCommCount", r);
  if (nranks) *nranks = n;
  if (reduce_comm_is_separate) *reduce_comm_is_separate = c->own_reduce_comm ? 1 : 0;
  return 0;
}

int tpsrhs_rccl_destroy(tpsrhs_rccl_ctx *c) {
  if (!c) return 0;
  if (c->own_reduce_comm && c->comm_reduce) ncclCommDestroy(c->comm_reduce);
  if (c->comm) ncclCommDestroy(c->comm);
  delete c;
  return 0;
}

int tpsrhs_rccl_halo(void *ctx, int /*phase*/, const double *send, double *recv, int num_neighbors,
                     const int *neighbor_ranks, const int64_t *send_offsets, const int64_t *recv_offsets, void *stream) {
  tpsrhs_rccl_ctx *c = static_cast<tpsrhs_rccl_ctx *>(ctx);
  if (!c || !c->comm) {
    g_err = "tpsrhs_rccl_halo: no communicator";
    return 1;
  }
  if (c->skip) return 0;
  hipStream_t s = static_cast<hipStream_t>(stream);
  ncclResult_t r = ncclGroupStart();
  if (r != ncclSuccess) return fail("ncclGroupStart", r);
  for (int i = 0; i < num_neighbors; i++) {
    const int peer = neighbor_ranks[i];
    const size_t ns = static_cast<size_t>(send_offsets[i + 1] - send_offsets[i]);
    const size_t nr = static_cast<size_t>(recv_offsets[i + 1] - recv_offsets[i]);
    if (ns) {
      r = ncclSend(send + send_offsets[i], ns, ncclDouble, peer, c->comm, s);
      if (r != ncclSuccess) {
        ncclGroupEnd();
        return fail("ncclSend", r);
      }
    }
    if (nr) {
      r = ncclRecv(recv + recv_offsets[i], nr, ncclDouble, peer, c->comm, s);
      if (r != ncclSuccess) {
        ncclGroupEnd();
        return fail("ncclRecv", r);
      }
    }
    c->bytes_sent += static_cast<int64_t>(8 * ns);
    c->peers.insert(peer);
  }
  r = ncclGroupEnd();
  if (r != ncclSuccess) return fail("ncclGroupEnd", r);
  c->halo_calls++;
  return 0;
}

int tpsrhs_rccl_reduce(void *ctx, double *values, int count, int op, void *stream) {
  tpsrhs_rccl_ctx *c = static_cast<tpsrhs_rccl_ctx *>(ctx);
  if (!c || !c->comm) {
    g_err = "tpsrhs_rccl_reduce: no communicator";
    return 1;
  }
  const ncclResult_t r = ncclAllReduce(values, values, static_cast<size_t>(count), ncclDouble, op == 1 ? ncclMin : ncclSum,
                                       c->comm_reduce, static_cast<hipStream_t>(stream));
  if (r != ncclSuccess) return fail("ncclAllReduce", r);
  return 0;
}

int tpsrhs_rccl_stats(const tpsrhs_rccl_ctx *c, int64_t *halo_calls, int64_t *bytes_sent, int *peers_seen) {
  if (!c) return 1;
  if (halo_calls) *halo_calls = c->halo_calls;
  if (bytes_sent) *bytes_sent = c->bytes_sent;
  if (peers_seen) *peers_seen = static_cast<int>(c->peers.size());
  return 0;
}

int tpsrhs_rccl_set_skip(tpsrhs_rccl_ctx *c, int skip) {
  if (!c) return 1;
  c->skip = skip;
  return 0;
}

}  // extern "C"
