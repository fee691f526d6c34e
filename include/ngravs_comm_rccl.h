/* ngravs_comm_rccl.h -- the communicator vtable of ngravs_host.h filled with RCCL (collectives over xGMI), in plain C.
 *
 * Replaces, for a C host, the MPI calls of the reference's gravity path one for one:
 *     MPI_Allreduce  domain.c:906-907 (extent), :844-877 (per-leaf work/count)      -> ncclAllReduce
 *     MPI_Allgather  domain.c:965-988 (top-leaf lists), pm_periodic.c:285-291       -> ncclAllGather
 *     MPI_Sendrecv   domain.c:695-747 (particle exchange), gravtree.c:195-257 (export/import), forcetree.c:811-816
 *                    (top-leaf moments), pm_periodic.c:385-389, 655-660 (mesh patches), :433, :525 (FFT transposes)
 *                                                                                     -> grouped ncclSend / ncclRecv
 * The exchanged blocks are the library's own DEVICE buffers (ngravs_comm.device_buffers = 1): nothing is staged through host
 * memory; the small host-side reductions (a few doubles ... a few MB of per-leaf sums) go through a device scratch buffer
 * owned by the communicator.
 *
 * Bootstrap: one task calls ngravs_rccl_unique_id() and the host broadcasts the 128 bytes with whatever it has
 * (MPI_Bcast in gadget_glue.c, a torch.distributed broadcast in distributed.py); then every task calls ngravs_rccl_create().
 * One task per GPU (RCCL refuses two ranks of one communicator on the same device).
 *
 * libngravs_rccl.so = host/ngravs_comm_rccl.c + librccl + libamdhip64; it does not depend on libngravs_hip.so.
 */
#ifndef NGRAVS_COMM_RCCL_H
#define NGRAVS_COMM_RCCL_H

#include "ngravs_host.h"

#ifdef __cplusplus
extern "C" {
#endif

#define NGRAVS_RCCL_ID_BYTES 128

typedef struct ngravs_rccl ngravs_rccl;

/* ncclGetUniqueId(): 128 opaque bytes, to be broadcast by the host */
int ngravs_rccl_unique_id(char id[NGRAVS_RCCL_ID_BYTES]);
/* ncclCommInitRank() on HIP device `device` + a private stream + the scratch buffer.  Collective over all `size` tasks. */
int ngravs_rccl_create(const char id[NGRAVS_RCCL_ID_BYTES], int rank, int size, int device, ngravs_rccl **out);
/* fill `cm` (rank, size, device_buffers = 1, user, the callbacks incl. allreduce_dev) */
void ngravs_rccl_fill(ngravs_rccl *r, ngravs_comm *cm);
void ngravs_rccl_destroy(ngravs_rccl *r);
/* collectives issued since creation (or the last reset), host wall-clock seconds inside them, payload bytes handed to RCCL */
void ngravs_rccl_stats(ngravs_rccl *r, int64_t *calls, double *seconds, double *bytes, int reset);
/* text of the last RCCL / HIP error ("" if none) */
const char *ngravs_rccl_last_error(ngravs_rccl *r);
/* the world size RCCL itself reports (ncclCommCount) */
int ngravs_rccl_world(ngravs_rccl *r);
/* every callback of the vtable once with known answers (reductions, a gather, an all-to-all-v with unequal and empty blocks);
 * collective over all tasks and collective-safe (include/ngravs_comm_selftest.h: every task runs every stage, all tasks return the
 * same status); 0 = all as expected on every task, else ngravs_rccl_last_error() says what this task found.  Call it once after
 * ngravs_rccl_create(): a bootstrap or fabric problem then shows before the first step instead of inside it. */
int ngravs_rccl_selftest(ngravs_rccl *r);
/* Every wait of the communicator is bounded: after `seconds` (default 300) without completion a callback writes the task, the
 * collective and the byte counts per peer to stderr and ends the process with exit code 86 (exit_on_timeout != 0, the default: a
 * collective that did not complete cannot be retried in the same process) or returns an error (exit_on_timeout == 0). */
void ngravs_rccl_set_timeout(ngravs_rccl *r, double seconds, int exit_on_timeout);
/* barrier (an all-reduce of one int) -- convenience for hosts without another communicator */
int ngravs_rccl_barrier(ngravs_rccl *r);

#ifdef __cplusplus
}
#endif
#endif
