/*
 * accblas_comm.h — multi-GPU context of the MI355X accessor-BLAS path
 * (libaccblas_rccl.so = libaccblas_hip.so + RCCL).
 *
 * The reference is single-GPU (cuda/gemv_benchmark.cu:83-99 picks device 0);
 * this header is the boundary SURVEY.md §8b B2 / §8e asks for on top of it:
 *   - GEMV shards by contiguous ROW BLOCKS: every rank multiplies its own rows
 *     with a replicated x; no data-path collective. An all-gather of the result
 *     slices is offered for callers that want a replicated result.
 *   - dot shards by contiguous CHUNKS: every rank reduces its chunk into a
 *     device scalar of the arithmetic type, ONE all-reduce of that single
 *     element combines them, and the cast to the result type happens after the
 *     exchange (as `cast_result` does after the kernel, cuda/dot_kernels.cuh:168-173).
 *   - TRSV does not shard (a strict chain of dependent row blocks): no entry
 *     point here, run replicas.
 * Two ways to form the communicator, both plain RCCL underneath:
 *   one process per GPU   rank 0 calls accblas_comm_unique_id(), ships the 128
 *                         bytes to the other ranks by its own means (MPI, a
 *                         file, torch.distributed …), every rank calls
 *                         accblas_comm_create() with the device it has current;
 *   one process, G GPUs   accblas_comm_create_all(); bracket the per-device
 *                         calls of one collective with accblas_comm_group_begin
 *                         / _end.
 * Conventions are those of accblas.h (int status, int64_t element counts,
 * opaque device pointers, stream as void *, asynchronous launches). RCCL
 * failures return ACCBLAS_ERR_COMM; accblas_comm_last_error_string() has the
 * text.
 */
#ifndef ACCBLAS_COMM_H_
#define ACCBLAS_COMM_H_

#include "accblas.h"

#ifdef __cplusplus
extern "C" {
#endif

#define ACCBLAS_ERR_COMM (-5)
#define ACCBLAS_COMM_ID_BYTES 128

typedef struct accblas_comm accblas_comm;

const char *accblas_comm_last_error_string(void);

/* rank 0: fills `id` (ACCBLAS_COMM_ID_BYTES bytes) for accblas_comm_create */
int accblas_comm_unique_id(void *id);

/* Collective over all ranks. Uses the calling thread's current device; the
 * communicator remembers it. */
int accblas_comm_create(accblas_comm **comm, int world_size, int rank,
                        const void *id);

/* One process driving `n_gpus` devices (`devices` = NULL means 0..n_gpus-1):
 * fills comms[0..n_gpus-1], rank g bound to devices[g]. */
int accblas_comm_create_all(accblas_comm **comms, int n_gpus,
                            const int *devices);

int accblas_comm_destroy(accblas_comm *comm);
int accblas_comm_rank(const accblas_comm *comm, int *rank);
int accblas_comm_world_size(const accblas_comm *comm, int *world_size);
int accblas_comm_device(const accblas_comm *comm, int *device);

/* Single-process use: groups the collectives issued for several
 * communicators between begin and end (ncclGroupStart / ncclGroupEnd). RCCL
 * enqueues a grouped collective only at group end; work that must follow it
 * on the stream (the cast of accblas_comm_acc_dot to a result type other than
 * the arithmetic type) is enqueued by accblas_comm_group_end, after the
 * collectives, so the results are valid once the streams have passed the
 * group end. Begin and end must be called by the same host thread. */
int accblas_comm_group_begin(void);
int accblas_comm_group_end(void);

/* Balanced contiguous partition of [0, n): rank r owns [*begin, *end); the
 * first n % world ranks own one element more. `align` > 1 rounds interior
 * boundaries down to a multiple of `align` elements (e.g. 16-byte vectors). */
int accblas_comm_shard(const accblas_comm *comm, int64_t n, int64_t align,
                       int64_t *begin, int64_t *end);
/* the same partition without a communicator (host arithmetic only) */
int accblas_shard_bounds(int64_t n, int world_size, int rank, int64_t align,
                         int64_t *begin, int64_t *end);

/* dot over the union of the ranks' chunks: x, y are THIS rank's n_local
 * elements; on return (stream-ordered) every rank's device scalar `res`
 * (type res_t) holds the global sum. `handle` as in accblas_acc_dot. Stream
 * order: chunk dot -> all-reduce of ONE element of type `ar` -> cast to res_t
 * (none when res_t == ar: the all-reduce writes `res`). May be called inside a
 * group (see accblas_comm_group_begin), several times per communicator (e.g.
 * x.y and x.x batched for CG): every call takes its own 16-byte scalar slot
 * from a ring of 64, so at most 64 dots per communicator may be recorded in
 * one group (the 65th returns ACCBLAS_ERR_INVALID_ARGUMENT) and at most 64 may
 * be in flight on different streams outside a group. Each dot in flight needs
 * its own `handle` (a handle serialises its calls, accblas.h) unless the calls
 * share a stream. ACCBLAS_COMM_FORCE_COLLECTIVE=1 in the environment makes a
 * world of one rank run the collective as well (tests). */
int accblas_comm_acc_dot(accblas_comm *comm, accblas_handle *handle,
                         accblas_dtype ar, accblas_dtype st,
                         accblas_dtype res_t, int64_t n_local, const void *x,
                         int64_t incx, const void *y, int64_t incy, void *res,
                         void *stream);

/* Row-block GEMV: A is this rank's rows_local x cols block, x is replicated,
 * res_local (rows_local entries, stride incres) receives this rank's slice.
 * If `res_full` is not NULL it must hold world_size * rows_local contiguous
 * storage-type elements and receives every rank's slice (all-gather; needs
 * incres == 1 and equal rows_local on all ranks). */
int accblas_comm_acc_gemv(accblas_comm *comm, accblas_dtype ar,
                          accblas_dtype st, int64_t rows_local, int64_t cols,
                          double alpha, const void *A, int64_t lda,
                          const void *x, int64_t incx, double beta,
                          void *res_local, int64_t incres, void *res_full,
                          void *stream);

#ifdef __cplusplus
} /* extern "C" */
#endif

#endif /* ACCBLAS_COMM_H_ */
