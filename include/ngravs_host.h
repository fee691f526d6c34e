/* ngravs_host.h -- the multi-task choreography of the gravity path in plain C, over a communicator vtable.
 *
 * libngravs_hip.so only packs and unpacks device buffers (include/ngravs_hip.h); WHO moves the bytes is the host's
 * business: MPI in the reference (gadget_glue.c fills the vtable with MPI_Allreduce / MPI_Allgather / MPI_Alltoallv),
 * RCCL over xGMI in bench.py (torch.distributed behind the same three callbacks), a shared-memory stand-in in
 * host/host_shim_test.c.  The functions below are the reference's multi-task drivers restated once, in C, for all of them:
 *
 *   ngravs_host_domain_decomposition()   domain_Decomposition() -> domain_decompose() (domain.c:62-330): global extent,
 *        per-cell count + work histograms (domain_sumCost :823-877), the split of the Peano curve over the tasks
 *        (domain_findSplit :347-456 by count under a memory bound, domain_shiftSplit :468-544 by work), particle migration
 *        (domain_exchangeParticles :695-795), then the short-range halo that replaces the target export / force import of
 *        gravity_tree() (gravtree.c:112-285) for TreePM runs, then the local Peano order.
 *   ngravs_host_pmforce_periodic()       pmforce_periodic() on the x-slab decomposed mesh (pm_periodic.c:204-790): the four
 *        exchanges of ngravs_pm_slab_*.
 *   ngravs_host_compute_accelerations()  compute_accelerations(0) for gravity (accel.c:24-58).
 *
 * All functions are collective: every task of the communicator calls them in the same order.  Return 0 or a negative
 * ngravs_status (a callback's non-zero return is passed through as NGRAVS_ERR_STATE).
 */
#ifndef NGRAVS_HOST_H
#define NGRAVS_HOST_H

#include "ngravs_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

enum { NGRAVS_OP_SUM = 0, NGRAVS_OP_MIN = 1, NGRAVS_OP_MAX = 2 };
enum { NGRAVS_T_F64 = 0, NGRAVS_T_I64 = 1 };

typedef struct ngravs_comm
{
  int32_t rank, size;         /* ThisTask, NTask (size <= 64) */
  int32_t device_buffers;     /* 1: alltoallv is handed DEVICE pointers (GPU-aware MPI, RCCL); 0: host pointers -- ngravs_host
                                 stages the exchange buffers through host memory with ngravs_memcpy() */
  int32_t reserved;
  void *user;                 /* passed back to the callbacks (MPI_Comm *, a Python object, ...) */
  /* in-place reduction of `count` elements of `dtype` in HOST memory over all tasks */
  int (*allreduce)(void *user, void *buf, int64_t count, int dtype, int op);
  /* every task contributes `bytes` bytes, recv = size * bytes in task order; HOST memory */
  int (*allgather)(void *user, const void *send, void *recv, int64_t bytes);
  /* counts and displacements in BYTES per peer; device or host memory as device_buffers says */
  int (*alltoallv)(void *user, const void *send, const int64_t *send_bytes, const int64_t *send_displ, void *recv,
                   const int64_t *recv_bytes, const int64_t *recv_displ);
} ngravs_comm;

typedef struct ngravs_dd_info
{
  int32_t level;              /* decomposition cells = Peano cells of this level (8^level of them)            */
  int32_t reserved;
  int64_t n_local, n_halo;    /* own particles, halo copies after this decomposition                            */
  int64_t n_migrated_in;      /* particles received in the migration                                            */
  double work_balance;        /* max over tasks of the work sum / mean (the reference's "work-load balance")     */
  double memory_balance;      /* max over tasks of the particle count / mean ("memory-balance")                  */
  double bytes_migration, bytes_halo;   /* payload this task sent to other tasks                                 */
  /* host wall-clock seconds of the stages of the last decomposition (collectives inside them included): 0 extent + histogram +
   * split, 1 migration, 2 top-cell sums + all-reduce, 3 need test (host), 4 request all-gather + pack of the requested cells,
   * 5 import exchange + unpack, 6 global top, 7 local decomposition (keys, sort, gather) */
  double seconds[8];
} ngravs_dd_info;

/* The cut of the curve: owner of every decomposition cell, in Peano-cell order and in [x][y][z] order */
typedef struct ngravs_dd_plan
{
  int32_t level, reserved;
  int64_t ncell;              /* 8^level */
  int32_t *owner_ph, *owner_xyz;   /* malloc'ed by ngravs_host_domain_owners, released by ngravs_host_plan_free */
} ngravs_dd_plan;

/* level 0 => the coarsest level whose cells are still at least as wide as the short-range cut (TreePM), at most 5.
 * part_alloc_factor: the memory bound of domain_findSplit, particles per task <= part_alloc_factor * N/NTask
 * (All.PartAllocFactor; <= 0 => 1.5).  info may be NULL.
 * The library migrates its device-resident particle columns itself.  This is the whole domain_Decomposition(). */
int ngravs_host_domain_decomposition(ngravs_ctx *ctx, const ngravs_comm *comm, int level, double part_alloc_factor,
                                     ngravs_dd_info *info);
/* The same in three steps, for a host whose own particle structures have to move with the particles (the reference's P[]
 * carries velocities, IDs and timestep data the library never sees):
 *   ngravs_host_domain_owners()   extent + histograms + split: fills `plan`
 *   ngravs_dd_get_dest()          (ngravs_hip.h) destination task of every local particle -> the HOST exchanges its records
 *                                 (domain_exchangeParticles) and hands the new local set over with ngravs_set_particles()
 *   ngravs_host_domain_halo()     short-range halo exchange + local Peano order */
int ngravs_host_domain_owners(ngravs_ctx *ctx, const ngravs_comm *comm, int level, double part_alloc_factor, ngravs_dd_plan *plan,
                              ngravs_dd_info *info);
int ngravs_host_domain_halo(ngravs_ctx *ctx, const ngravs_comm *comm, const ngravs_dd_plan *plan, ngravs_dd_info *info);
void ngravs_host_plan_free(ngravs_dd_plan *plan);
int ngravs_host_pmforce_periodic(ngravs_ctx *ctx, const ngravs_comm *comm);
/* host wall-clock seconds of the last ngravs_host_pmforce_periodic() of this process: [0] brick deposit + bounding-box all-gather,
 * then for the four exchanges s = 0..3: [1+3s] pack (incl. the FFTs and the Green's function that precede it), [2+3s] the
 * all-to-all-v, [3+3s] unpack (stage 3: + gradient and gather) */
void ngravs_host_pm_seconds(double out[13]);
int ngravs_host_compute_accelerations(ngravs_ctx *ctx, const ngravs_comm *comm, int pm_step, ngravs_dd_info *info);

/* The split alone (host arrays, no communication): owner[cell] for the 8^level cells in Peano order, from the global count
 * and work histograms; domain_findSplit + domain_shiftSplit.  Returns 0, or -1 if no split respects max_load. */
int ngravs_host_split(const int64_t *count, const double *work, int64_t ncell, int ntask, double max_load, int32_t *owner);

#ifdef __cplusplus
}
#endif
#endif /* NGRAVS_HOST_H */
