/* tpsrhs_rccl.h -- the neighbour exchange of tpsrhs.h over RCCL, as a C library of its own.
 *
 * libtpsrhs.so links neither MPI nor RCCL: on a partitioned mesh it hands the packed face traces to the
 * tpsrhs_halo_fn of tpsrhs_runtime.  This library (libtpsrhs_rccl.so, links librccl) is the implementation
 * for one process per GPU on one node: where the reference posts MPI_Isend / MPI_Irecv per face-neighbour rank
 * and MPI_Waitall (initNBlockDataTransfer / waitAllDataTransfer, src/rhs_operator.cpp:775-831) it enqueues ONE
 * group of ncclSend / ncclRecv per phase on the operator's communication stream -- a neighbour all-to-all-v in
 * which every pair of GPUs uses its own xGMI link -- and returns without touching the host again.  The scalar
 * reductions (MPI_Allreduce of the boundary means, src/outletBC.cpp:533-540, and of the time step,
 * src/M2ulPhyS.cpp:2013-2016) map to ncclAllReduce on the same stream.
 *
 * Bootstrap (the role of MPI_Init + MPI_Comm_dup): rank 0 calls tpsrhs_rccl_unique_id, the caller broadcasts the
 * 128 bytes with whatever it already has (MPI_Bcast in TPS, a torch.distributed broadcast in bench.py), every
 * rank calls tpsrhs_rccl_create.
 */
#ifndef TPSRHS_RCCL_H_
#define TPSRHS_RCCL_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TPSRHS_RCCL_ID_BYTES 128

typedef struct tpsrhs_rccl_ctx tpsrhs_rccl_ctx;

/* ncclGetUniqueId; `id` receives TPSRHS_RCCL_ID_BYTES bytes.  0 on success. */
int tpsrhs_rccl_unique_id(void *id);

/* ncclCommInitRank on `device` (hipSetDevice first).  Collective over the `nranks` processes.  0 on success. */
int tpsrhs_rccl_create(const void *id, int nranks, int rank, int device, tpsrhs_rccl_ctx **out);
int tpsrhs_rccl_destroy(tpsrhs_rccl_ctx *ctx);

/* tpsrhs_halo_fn / tpsrhs_reduce_fn of tpsrhs.h; `ctx` is the tpsrhs_rccl_ctx.  They only enqueue on `stream`. */
int tpsrhs_rccl_halo(void *ctx, int phase, const double *send, double *recv, int num_neighbors,
                     const int *neighbor_ranks, const int64_t *send_offsets, const int64_t *recv_offsets, void *stream);
int tpsrhs_rccl_reduce(void *ctx, double *values, int count, int op, void *stream);

/* Counters for honest reporting: calls of tpsrhs_rccl_halo, bytes sent by this rank, distinct peer ranks seen;
 * `skip` != 0 makes tpsrhs_rccl_halo return at once (timing experiments: exposed-communication measurement). */
int tpsrhs_rccl_stats(const tpsrhs_rccl_ctx *ctx, int64_t *halo_calls, int64_t *bytes_sent, int *peers_seen);
int tpsrhs_rccl_set_skip(tpsrhs_rccl_ctx *ctx, int skip);
/* ncclCommCount of the exchange communicator (the number of ranks RCCL itself sees), and whether the scalar
 * reductions run on a communicator of their own or share the exchange's (the default).  The second communicator is
 * opt-in: TPSRHS_RCCL_SPLIT=1 in the environment of EVERY rank (ncclCommSplit is collective); the ranks then agree on the
 * outcome -- all split, or all shared -- before tpsrhs_rccl_create returns. */
int tpsrhs_rccl_comm_info(const tpsrhs_rccl_ctx *ctx, int *nranks, int *reduce_comm_is_separate);
const char *tpsrhs_rccl_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
