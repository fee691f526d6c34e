/* ngravs_host.h -- the multi-task choreography of the gravity path in plain C, over a communicator vtable.
 *
 * libngravs_hip.so only packs and unpacks device buffers (include/ngravs_hip.h); WHO moves the bytes is the host's
 * business: RCCL over xGMI (include/ngravs_comm_rccl.h fills the vtable in C; gadget_glue.c and distributed.py use it), MPI
 * (gadget_glue.c's fallback for hosts without RCCL), torch.distributed "gloo" for the CPU-side rehearsal, a shared-memory
 * stand-in in host/host_shim_test.c.  The functions below are the reference's multi-task drivers restated once, in C, for
 * all of them:
 *
 *   ngravs_host_domain_decomposition()   domain_Decomposition() -> domain_decompose() (domain.c:62-330): global extent, the
 *        adaptive top tree in key space (domain_determineTopTree / domain_topsplit :933-1138: a cell is split while it holds
 *        more than a threshold of particles), per-leaf count + work (domain_sumCost :823-877), the cut of the leaf sequence
 *        over the tasks under a memory bound (the job of domain_findSplit / domain_shiftSplit :347-544), particle migration
 *        (domain_exchangeParticles :695-795), then the top-leaf moments of all tasks and the import of the top leaves a task's
 *        targets may open (force_exchange_pseudodata / force_treeupdate_pseudos, forcetree.c:766-996; replaces the target
 *        export / force import of gravity_tree(), gravtree.c:112-285), then the local Peano order.
 *   ngravs_host_pmforce_periodic()       pmforce_periodic() on the x-slab decomposed mesh (pm_periodic.c:204-790): the four
 *        exchanges of ngravs_pm_slab_*.
 *   ngravs_host_compute_accelerations()  compute_accelerations(0) for gravity (accel.c:24-58).
 *
 * All functions are collective: every task of the communicator calls them in the same order.  Return 0 or a negative
 * ngravs_status (a callback's non-zero return is passed through as NGRAVS_ERR_STATE).  A task that fails locally (out of
 * memory, a failing library call) still takes part in the following collective and reports its status through it: all tasks
 * return an error together instead of some of them waiting for ever.
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
  int32_t device_buffers;     /* 1: alltoallv is handed DEVICE pointers (RCCL, GPU-aware MPI); 0: host pointers -- ngravs_host
                                 stages the exchange buffers through host memory with ngravs_memcpy() */
  int32_t reserved;
  void *user;                 /* passed back to the callbacks (an ngravs_rccl *, an MPI_Comm *, a Python object, ...) */
  /* in-place reduction of `count` elements of `dtype` in HOST memory over all tasks */
  int (*allreduce)(void *user, void *buf, int64_t count, int dtype, int op);
  /* every task contributes `bytes` bytes, recv = size * bytes in task order; HOST memory */
  int (*allgather)(void *user, const void *send, void *recv, int64_t bytes);
  /* counts and displacements in BYTES per peer; device or host memory as device_buffers says */
  int (*alltoallv)(void *user, const void *send, const int64_t *send_bytes, const int64_t *send_displ, void *recv,
                   const int64_t *recv_bytes, const int64_t *recv_displ);
  /* optional (NULL: not provided): in-place reduction of `count` elements in DEVICE memory.  With it the per-leaf sums of
   * the decomposition (a few MB) are reduced where the kernel left them and cross to the host once, already summed */
  int (*allreduce_dev)(void *user, void *dev_buf, int64_t count, int dtype, int op);
} ngravs_comm;

/* The top tree: the reference's TopNodes[] (allvars.h:252-262, domain.c:933-1138) -- an oct-tree in Peano-Hilbert key space,
 * the same on every task.  Node 0 is the root (the domain cube); the 8 children of a split node are consecutive, in key
 * order, so that the leaves in depth-first order are the segments of the space-filling curve.  Leaves are what the domain
 * cut assigns to tasks and what a task imports from another. */
typedef struct ngravs_toptree
{
  int32_t nnode, nleaf, depth, reserved;
  int32_t *child;             /* [nnode] index of the first of the 8 children, or -1: a leaf                         */
  int32_t *level;             /* [nnode] 0 = root                                                                     */
  int32_t *xyz;               /* [3 * nnode] integer coordinates of the cell at its level                              */
  int32_t *leaf;              /* [nnode] number of the leaf along the curve, or -1                                     */
  int32_t *node_of_leaf;      /* [nleaf]                                                                               */
} ngravs_toptree;

typedef struct ngravs_dd_info
{
  int32_t n_topnodes, n_topleaves;   /* the top tree of this decomposition (NTopnodes, NTopleaves)                     */
  int64_t n_local, n_halo;    /* own particles, imported copies after this decomposition                               */
  int64_t n_migrated_in;      /* particles received in the migration                                                    */
  double work_balance;        /* max over tasks of the work sum / mean (the reference's "work-load balance")            */
  double memory_balance;      /* max over tasks of the particle count / mean ("memory-balance")                         */
  double bytes_migration, bytes_halo;   /* payload this task sent to other tasks                                        */
  /* host wall-clock seconds of the stages of the last decomposition (collectives inside them included): 0 extent + leaf sums +
   * their all-reduce + top-tree update + cut, 1 migration, 2 of [0]: the leaf-sum passes and their all-reduces alone,
   * 3 import decision (host), 4 request / count all-gather + pack of the requested leaves, 5 import exchange + unpack,
   * 6 global top, 7 local decomposition (keys, sort, gather) */
  double seconds[8];
  int32_t toptree_rounds;     /* leaf-sum passes this decomposition needed (1 in steady state; more while the tree adapts) */
  int32_t collectives;        /* collective calls of this decomposition                                                   */
} ngravs_dd_info;

/* The cut of the curve: the top tree and the owner of every leaf */
typedef struct ngravs_dd_plan
{
  ngravs_toptree tree;
  int32_t *leaf_owner;        /* [tree.nleaf] */
  double *node_sums;          /* [tree.nnode * NGRAVS_TOP_CW(n_gravs)] global sums of every top node (DomainMoment[] and what
                                 force_treeupdate_pseudos adds up the ancestor chain, forcetree.c:766-947)                  */
  double bounds[2];           /* global minima of ErrTolForceAcc * OldAcc and of the softening length over active particles */
} ngravs_dd_plan;

/* leaf_max: a top node is split while it holds more than leaf_max particles; <= 0 => the reference's TotNumPart /
 * (TOPNODEFACTOR * NTask) with TOPNODEFACTOR = 20 (domain.c:1060, allvars.h:72), at most NGRAVS_TOPLEAF_MAX.
 * part_alloc_factor: the memory bound of the cut, particles per task <= part_alloc_factor * N/NTask (All.PartAllocFactor;
 * <= 0 => 1.5).  info may be NULL.
 * The library migrates its device-resident particle columns itself.  This is the whole domain_Decomposition(). */
#define NGRAVS_TOPLEAF_MAX 3000.0   /* not a power of two: uniform boxes of 2^k particles put 8^-level of them into a cell */
#define NGRAVS_TOPLEVEL_MAX 18   /* BITS_PER_DIMENSION: the reference's keys resolve no finer cell (domain.c:1004) */
#define NGRAVS_TOPNODES_MAX 2400000   /* MAXTOPNODES is 200000 in the reference (allvars.h:70); here the table is cheap */
int ngravs_host_domain_decomposition(ngravs_ctx *ctx, const ngravs_comm *comm, double leaf_max, double part_alloc_factor,
                                     ngravs_dd_info *info);
/* The same in three steps, for a host whose own particle structures have to move with the particles (the reference's P[]
 * carries velocities, IDs and timestep data the library never sees):
 *   ngravs_host_domain_owners()   extent + top tree + per-leaf sums + cut: fills `plan`
 *   ngravs_dd_get_dest()          (ngravs_hip.h) destination task of every local particle -> the HOST exchanges its records
 *                                 (domain_exchangeParticles) and hands the new local set over with ngravs_set_particles()
 *   ngravs_host_domain_halo()     import of the top leaves this task may open + local Peano order */
int ngravs_host_domain_owners(ngravs_ctx *ctx, const ngravs_comm *comm, double leaf_max, double part_alloc_factor, ngravs_dd_plan *plan,
                              ngravs_dd_info *info);
int ngravs_host_domain_halo(ngravs_ctx *ctx, const ngravs_comm *comm, const ngravs_dd_plan *plan, ngravs_dd_info *info);
void ngravs_host_plan_free(ngravs_dd_plan *plan);
/* A step on which the decomposition is kept (domain.c:76, All.TreeDomainUpdateFrequency > 0): after ngravs_update_particles() with
 * the own rows' drifted positions -- the imported copies are refreshed by their owners (one all-to-all-v with the requests of the
 * decomposition), the tree is refit, and the global moments and cell sides of the top nodes are renewed from one all-reduce of
 * per-leaf sums (force_update_pseudoparticles forcetree.c:753, force_update_node_len_toptree :1096-1122).  The refit tree is the
 * single task's refit tree wherever this task's targets look.  Collective.  NGRAVS_ERR_STATE without a kept decomposition. */
int ngravs_host_kept_step(ngravs_ctx *ctx, const ngravs_comm *comm, ngravs_dd_info *info);
/* Every callback of the communicator once, with known answers (include/ngravs_comm_selftest.h: collective-safe -- every task runs
 * every stage and all tasks return the same status: 0, or bits 0-2 for the stage that gave a wrong answer on some task, bit 3 for a
 * missing buffer).  Collective.  ctx: the context whose device holds the exchange buffers of a device_buffers communicator (may be
 * NULL for a host-buffer communicator).  fail_stage (tests): this task pretends stage 1-3 failed.  why (may be NULL): this task's
 * own finding.  A host calls it once after it has filled its vtable, so that a fabric problem shows before the first step. */
int ngravs_host_comm_selftest(ngravs_ctx *ctx, const ngravs_comm *comm, int fail_stage, char *why, int why_len);
int ngravs_host_pmforce_periodic(ngravs_ctx *ctx, const ngravs_comm *comm);
/* host wall-clock seconds of the last ngravs_host_pmforce_periodic() of this thread: [0] brick deposit + bounding-box all-gather,
 * then for the four exchanges s = 0..3: [1+3s] pack (incl. the FFTs and the Green's function that precede it), [2+3s] the
 * all-to-all-v, [3+3s] unpack (stage 3: + gradient and gather) */
void ngravs_host_pm_seconds(double out[13]);
int ngravs_host_compute_accelerations(ngravs_ctx *ctx, const ngravs_comm *comm, int pm_step, ngravs_dd_info *info);

/* ---- pure host pieces (no GPU, no communication): tested on the CPU ------------------------------------------------------ */
/* the complete tree down to `level` (0: the root alone) */
int ngravs_host_toptree_init(ngravs_toptree *t, int level);
/* a tree from its child[] table alone (what ngravs_dd_get_toptree() hands back): levels, coordinates, leaf numbering */
int ngravs_host_toptree_from_children(ngravs_toptree *t, const int32_t *child, int32_t nnode);
/* One round of domain_topsplit (domain.c:1060-1138) on a tree whose leaf counts are known: a leaf with more than `thresh`
 * particles is split (below level max_level; the root always is), a split node that holds <= thresh particles becomes a leaf
 * again.  leaf_count: global particle count per leaf of `t`.  out: the new tree.  Returns the number of leaves of `out` whose
 * counts are not known from `t`, or a negative status.  If nothing violates the rule the tree stays: returns 0 with
 * out->nnode == 0 (nothing is built -- the steady state costs one pass over the node counts). */
int ngravs_host_toptree_adapt(const ngravs_toptree *t, const double *leaf_count, double thresh, int max_level, ngravs_toptree *out);
void ngravs_host_toptree_free(ngravs_toptree *t);
/* the tree the context holds (ngravs_dd_set_toptree), with all its tables, as a VIEW: the pointers belong to the library and stay
 * valid until the next ngravs_dd_set_toptree(); nnode 0: none yet.  Do not free. */
int ngravs_host_toptree_borrow(ngravs_ctx *ctx, ngravs_toptree *view);
/* The cut alone: owner[leaf] for the leaves in curve order from the global count and work per leaf: contiguous segments,
 * the largest work sum of a task as small as the memory bound count <= max_load allows.  Returns 0, or -1 if no cut respects
 * max_load. */
int ngravs_host_split(const double *count, const double *work, int64_t nleaf, int ntask, double max_load, int32_t *owner);
/* The import decision alone: need[leaf] = 1 for every foreign top leaf whose particles task `me` has to hold so that its
 * tree is the single-task tree wherever one of its targets may look (the walk's own conservative tests against boxes around
 * the task's leaves).  node_sums: NGRAVS_TOP_CW(n_gravs) doubles per top NODE; dom: DomainCorner[3], DomainCenter[3], DomainLen,
 * DomainFac; bounds: see ngravs_dd_plan. */
int ngravs_host_import_request(const ngravs_config_t *cfg, const double dom[8], const ngravs_toptree *t, const double *node_sums,
                               const int32_t *leaf_owner, int me, const double bounds[2], uint8_t *need);
/* ... with the boxes around the own leaves grown by `margin` on every side (a kept decomposition: ngravs_dd_keep_margin) */
int ngravs_host_import_request_margin(const ngravs_config_t *cfg, const double dom[8], const ngravs_toptree *t, const double *node_sums,
                               const int32_t *leaf_owner, int me, const double bounds[2], double margin, uint8_t *need);

#ifdef __cplusplus
}
#endif
#endif /* NGRAVS_HOST_H */
