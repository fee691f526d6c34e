/* ngravs_hip.h -- C ABI of libngravs_hip.so, the MI355X (gfx950) gravitational force engine for
 * Gadget-2.0.7-ngravs.
 *
 * This is the drop-in boundary for ONE hot path of the reference: everything under
 * compute_accelerations() (accel.c:24-96) that computes gravity, i.e.
 *     domain_Decomposition()  (domain.c:62)      -> ngravs_domain_decomposition()
 *     force_treebuild()       (forcetree.c:61)   -> ngravs_force_treebuild()
 *     gravity_tree()          (gravtree.c:27)    -> ngravs_gravity_tree()
 *     pmforce_periodic()      (pm_periodic.c:204)-> ngravs_pmforce_periodic()
 *     init_grav_maps()/wire_grav_maps() (ngravs_core.c:201, ngravs.c:64) -> ngravs_config_t.law_*
 *     force_treeallocate() short-range tabulation (forcetree.c:3246-3403) -> ngravs_shortrange_table()
 *     gravity_forcetest() direct sum (gravtree_forcetest.c:28, forcetree.c:3428) -> ngravs_direct_sum()
 *
 * Plain C: pointers and sizes only, no torch / HIP types in any signature.  The reference keeps
 * its state in globals (P[], All, TypeToGrav[], AccelFxns[][] ...); the glue a maintainer adds to
 * the reference (gadget-2.0.7-ngravs_amd/host/gadget_glue.c) copies the fields
 * listed in SURVEY.md 8(b) into ngravs_config_t / ngravs_particles_t and calls the functions below
 * from the reference's own entry points, which keep their `void f(void)` signatures.
 *
 * Error convention: every call returns 0 on success or a negative ngravs_status; in addition a
 * fatal condition invokes the registered on_fatal(code, msg) callback first (default: print
 * "endrun called with an error level of C" in the reference's wording and return), mirroring
 * endrun(code) (endrun.c:25-43).  Codes reuse the reference's numbers where one exists
 * (1 = out of tree nodes, forcetree.c:247-253; 986/987 = massive empty node, forcetree.c:1482,1906).
 *
 * Threading: entry points are called serially from one host thread per process; one process per
 * GPU (SURVEY.md 8(b) "Threading").
 */
#ifndef NGRAVS_HIP_H
#define NGRAVS_HIP_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NGRAVS_ABI_VERSION 3
#define NGRAVS_MAX_GRAVS 3      /* N_GRAVS upper bound compiled in (allvars.h:130-152)            */
#define NGRAVS_NTYPES 6         /* Gadget particle types                                           */
#define NGRAVS_NTAB 2048        /* NTAB: short-range table length (Makefile.reference, forcetree.c:33) */
#define NGRAVS_ASMTH 1.25       /* allvars.h:83                                                     */
#define NGRAVS_RCUT 4.5         /* allvars.h:87                                                     */
#define NGRAVS_BITS_PER_DIMENSION 18 /* allvars.h:34: bits of the reference Peano-Hilbert key      */
#define NGRAVS_GROUP_REACH 4.5  /* default ngravs_config_t.group_reach = RCUT (DESIGN.md "Group walk cut")  */
#define NGRAVS_TREE_BITS 21     /* bits/dim of the engine's internal tree key; key21>>9 == key18    */

/* Force-law identifiers: the device cannot call through the reference's `gravity` function
 * pointers (allvars.h:131-150), so each wired pointer is named by an id.  The same id selects the
 * r-space law (AccelFxns), its k-space Green's function (GreensFxns) and the normalised Green's
 * function (NormedGreensFxns) of that law. */
typedef enum {
  NGRAVS_LAW_NONE = 0,        /* none            ngravs.c:344                         */
  NGRAVS_LAW_NEWTON = 1,      /* newtonian :351, pgdelta :390, normed_pgdelta :400    */
  NGRAVS_LAW_NEG_NEWTON = 2,  /* neg_newtonian :357, neg_pgdelta :406                 */
  NGRAVS_LAW_YUKAWA = 3,      /* yukawa :856, pgyukawa :869, normed_pgyukawa :880     */
  NGRAVS_LAW_COLOYUK = 4,     /* coloyuk :826, pgcoloyuk :830, normed_pgcoloyuk :834  */
  /* The BAM family (ngravs.c:495-668, NGRAVS_ACCUMULATOR_TESTING wiring :163-210): laws that depend on the TARGET mass and
   * on the number N of particles of the source species a node holds (allvars.h:645-648).  Tree-only (their Green's functions
   * are `none`); evaluated by both walks and the direct sum. */
  NGRAVS_LAW_BAMBAM = 5,      /* bambam :495             BAM target, BAM source              */
  NGRAVS_LAW_SOURCEBAM = 6,   /* sourcebambaryon :590    baryon target, BAM source            */
  NGRAVS_LAW_TARGETBAM = 7,   /* sourcebaryonbam :646    BAM target, baryon source            */
  NGRAVS_LAW_COUNT
} ngravs_law;

typedef enum {
  NGRAVS_SPLINE_NONE = 0,        /* none                 */
  NGRAVS_SPLINE_PLUMMER = 1,     /* plummer      ngravs.c:420-434 */
  NGRAVS_SPLINE_NEG_PLUMMER = 2, /* neg_plummer  ngravs.c:438-455 */
  NGRAVS_SPLINE_BAMBAM = 3,      /* bambam_spline :531 */
  NGRAVS_SPLINE_SOURCEBAM = 4,   /* sourcebambaryon_spline :562 */
  NGRAVS_SPLINE_TARGETBAM = 5,   /* sourcebaryonbam_spline :616 */
  NGRAVS_SPLINE_COUNT
} ngravs_spline;

typedef enum {
  NGRAVS_OK = 0,
  NGRAVS_ERR_ARG = -1,        /* bad argument / unsupported configuration                 */
  NGRAVS_ERR_NO_DEVICE = -2,  /* no HIP device or HIP runtime error                       */
  NGRAVS_ERR_NOMEM = -3,      /* device allocation failed                                 */
  NGRAVS_ERR_STATE = -4,      /* call order violated (e.g. walk before tree build)        */
  NGRAVS_ERR_TREE = -5,       /* tree build failed (out of nodes: reference endrun(1))    */
  NGRAVS_ERR_WIRING = -6      /* law table not wired / asymmetric (ngravs_core.c:321-424) */
} ngravs_status;

/* Walk variants (what gravity_tree() does per particle). */
typedef enum {
  NGRAVS_WALK_STRICT = 0, /* per-target opening decisions, exactly force_treeevaluate[_shortrange]
                             (forcetree.c:1244-1610, 1623-2052): same interaction set and count     */
  NGRAVS_WALK_GROUP = 1   /* wavefront-cooperative walk: 64 Peano-contiguous targets share one
                             breadth-first traversal; a node is used only if EVERY target of the
                             group would use it (conservative => at least the reference accuracy)   */
} ngravs_walk_mode;

/* The subset of `All`, TypeToGrav[] and the law tables the path reads. */
typedef struct {
  int32_t abi_version;        /* NGRAVS_ABI_VERSION                                            */
  int32_t n_gravs;            /* N_GRAVS, 1..NGRAVS_MAX_GRAVS                                   */
  int32_t periodic;           /* PERIODIC                                                       */
  int32_t pmgrid;             /* PMGRID, 0 = tree-only                                          */
  double box_size;            /* All.BoxSize                                                    */
  double G;                   /* All.G                                                          */
  double err_tol_theta;       /* All.ErrTolTheta (0 => relative criterion, gravtree.c:334-335)  */
  double err_tol_force_acc;   /* All.ErrTolForceAcc                                             */
  double force_softening[NGRAVS_NTYPES]; /* All.ForceSoftening[] = 2.8*SofteningTable (gravtree.c:514) */
  int32_t type_to_grav[NGRAVS_NTYPES];   /* TypeToGrav[] (ngravs_core.c:201-260)                */
  /* [TARGET][SOURCE] tables exactly as wire_grav_maps() fills them (ngravs.c:73-76) */
  int32_t law_accel[NGRAVS_MAX_GRAVS][NGRAVS_MAX_GRAVS];    /* AccelFxns        -> ngravs_law    */
  int32_t law_spline[NGRAVS_MAX_GRAVS][NGRAVS_MAX_GRAVS];   /* AccelSplines     -> ngravs_spline */
  int32_t law_greens[NGRAVS_MAX_GRAVS][NGRAVS_MAX_GRAVS];   /* GreensFxns       -> ngravs_law    */
  int32_t law_normed[NGRAVS_MAX_GRAVS][NGRAVS_MAX_GRAVS];   /* NormedGreensFxns -> ngravs_law    */
  double yukawa_imass;        /* YUKAWA_IMASS (ngravs.c:41-43), default 60.0                     */
  double asmth;               /* All.Asmth[0] = ASMTH*BoxSize/PMGRID (pm_periodic.c:59); 0 => derive */
  double rcut;                /* All.Rcut[0]  = RCUT*Asmth (pm_periodic.c:60); 0 => derive       */
  double tree_alloc_factor;   /* All.TreeAllocFactor (nodes per particle), 0 => 0.8              */
  double group_reach;         /* group walk only: radius, in units of Asmth, out to which short-range
                                 forces are evaluated.  0 => NGRAVS_GROUP_REACH.  RCUT (4.5) is the
                                 reference's nominal cut (allvars.h:87), 6.0 the end of its table
                                 (tabindex < NTAB, forcetree.c:1962-1967).  The strict walk ignores it. */
  int32_t walk_mode;          /* ngravs_walk_mode                                                */
  int32_t device;             /* HIP device ordinal                                              */
  int32_t rank, world_size;   /* ThisTask, NTask: target shard = Peano segment `rank` of `world_size` */
  double bam_epsilon;         /* BAM_EPSILON (ngravs.c:45-47); 0 => 1.31e-6                                */
  int32_t reserved[6];
} ngravs_config_t;

/* Host- or device-resident particle columns (the fields of struct particle_data the path reads,
 * allvars.h:546-581).  AoS callers pass strides in BYTES (e.g. stride 144 for P[] built with
 * PMGRID), SoA callers pass sizeof(element).  `on_device` != 0 means the pointers are HIP device
 * pointers (zero-copy hand-over; no PCIe traffic inside the call). */
typedef struct {
  int64_t n;                 /* NumPart; 0 is legal (a task without particles in a multi-task run: pointers may be NULL) */
  const double *pos;         /* Pos[3]          */  int64_t pos_stride;
  const double *mass;        /* Mass            */  int64_t mass_stride;
  const int32_t *type;       /* Type            */  int64_t type_stride;
  const double *old_acc;     /* OldAcc (may be NULL => 0) */ int64_t old_acc_stride;
  const uint8_t *active;     /* NULL => all active; else bit 0 set where Ti_endstep==All.Ti_Current (gravtree.c:113) */
  int64_t active_stride;
  const double *grav_pm;     /* GravPM[3] of the last PM step (may be NULL => none).  P[].GravPM lives in the host's P[] between
                              * PM steps; on non-PM steps OldAcc = |GravAccel + GravPM/G| needs it (gravtree.c:318-330) */
  int64_t grav_pm_stride;
  const float *grav_cost;    /* GravCost of the last walk (may be NULL => 0): the work weight 1 + GravCost of the domain cut
                              * (domain.c:859-862).  A host with individual timesteps passes (1 + GravCost)/(Ti_endstep -
                              * Ti_begstep) - 1 to reproduce the reference's weighting by step frequency. */
  int64_t grav_cost_stride;
  int32_t on_device;
  int32_t reserved;
} ngravs_particles_t;

/* Per-call statistics, the numbers gravity_tree() logs to timings.txt/cpu.txt
 * (gravtree.c:408-447, run.c:394-402). */
typedef struct {
  int64_t n_active;          /* Nf                                                       */
  int64_t n_nodes;           /* Numnodestree                                             */
  double interactions;       /* sum of ninteractions over active targets (ia/part * Nf)  */
  double t_domain, t_peano, t_treebuild, t_treewalk, t_pm; /* seconds, device time       */
  double walk_kernel_ms;     /* all walk kernels of the call (traversal + evaluation), HIP events */
  double reserved[7];        /* group walk, per group: [0] list entries [1] nodes tested [2] traversal batches
                              * [3] force-loop slots; split walk: [4] ms in the evaluation kernel (summed over its
                              * launches) [5] number of launches (batches) [6] ms in the traversal kernel */
} ngravs_stats_t;

typedef struct ngravs_ctx ngravs_ctx;
typedef void (*ngravs_fatal_fn)(int code, const char *msg);

/* ---- lifecycle ------------------------------------------------------------------------- */
int ngravs_abi_version(void);
const char *ngravs_build_info(void);                    /* arch, N_GRAVS instantiations, NTAB ... */
void ngravs_config_default(ngravs_config_t *cfg);       /* N_GRAVS=1 Newton/plummer, tree-only    */
/* init_grav_maps()+wire_grav_maps() checks (ngravs_core.c:321-424) happen here. */
int ngravs_create(const ngravs_config_t *cfg, ngravs_ctx **out);
void ngravs_destroy(ngravs_ctx *ctx);
void ngravs_set_fatal_handler(ngravs_ctx *ctx, ngravs_fatal_fn fn);
/* Change the walk parameters between calls (All.ErrTolTheta latch, gravtree.c:334-335). */
int ngravs_set_opening(ngravs_ctx *ctx, double err_tol_theta, double err_tol_force_acc);
int ngravs_set_walk_mode(ngravs_ctx *ctx, int walk_mode);
/* New All.ForceSoftening[6] (= 2.8 * All.SofteningTable[]): set_softenings() (gravtree.c:468-518) recomputes it from the scale
 * factor at the top of every gravity_tree() of a comoving run (gravtree.c:50-51); the glue's set_softenings() forwards it here.
 * Takes effect from the next walk / direct sum / import decision on. */
int ngravs_set_softening(ngravs_ctx *ctx, const double force_softening[NGRAVS_NTYPES]);
/* the configuration the context was created with (opening parameters as last set) */
int ngravs_get_config(ngravs_ctx *ctx, ngravs_config_t *out);
/* Performance / test parameters of the engine, by name (none changes WHAT is computed beyond rounding; there is no
 * environment variable and no way to make the library skip work):
 *   "walk_fused" 1: one fused traversal+evaluation kernel instead of the pair     "walk_batch" n: groups per launch pair
 *   "walk_waves" n: waves per evaluation workgroup (<= 16)                         "walk_lcap" n: initial item-list capacity (>= 1024)
 *   "walk_root" 1: TreePM group walks start at the root, not at the start table   "walk_compact" 0: do not compact sparse active sets
 *   "walk_spread" S: lanes per target for compacted active sets (1..64, 0 = auto)  "walk_exact_reach" 1: fp64 reach test, no fp32 pre-test
 *   "walk_sg" n: groups of 64 targets per traversal unit = per shared item list (0 = auto: with TreePM 4, or 2 / 1 when the last walk evaluated more than 1.4 / 2 x the pairs per target of a uniform box -- a clustered set; else 1)
 *   "walk_nleaf" k: an opened node with <= k particles hands its particles over instead of its children (0..8, -1 = default 8)
 *   "pm_notile" 1: per-particle CIC deposit     "pm_fused_gather" 1: one-pass gradient+gather    "pm_tile_gather" 1: LDS-tiled gather
 *   "pm_tile8" 1: deposit tiles of 8 instead of 16 mesh cells    "tree_levelwise" 1: level-by-level tree build for single-task trees too
 *   "sort_full" 1: Peano order by one radix sort on all key bits (default: top 28 to 42 bits + fix-up of the ties, the same order)
 *   "dd_keep" f: decompositions are kept over several steps (ngravs_host_kept_step): leaves are imported for ALL own particles as
 *       targets, whose cells may have grown by f x the domain's side (0 <= f <= 0.25; default 0: import for this step's active targets)
 *   "moments_octet" 1: node moments with eight lanes per node instead of one thread per node (another summation order; slower)
 * Returns NGRAVS_ERR_ARG for an unknown name or a value out of range. */
int ngravs_set_tuning(ngravs_ctx *ctx, const char *name, double value);
/* Plain copies for hosts that do not link HIP themselves (a C/MPI host staging exchange buffers through host memory):
 * kind 1 = host->device, 2 = device->host, 3 = device->device.  Synchronous. */
int ngravs_memcpy(ngravs_ctx *ctx, void *dst, const void *src, int64_t bytes, int kind);
/* ... and device memory for such a host's own exchange buffers (hipMalloc / hipFree on the context's device) */
int ngravs_device_alloc(ngravs_ctx *ctx, void **ptr, int64_t bytes);
int ngravs_device_free(ngravs_ctx *ctx, void *ptr);

/* ---- data hand-over ---------------------------------------------------------------------- */
/* Replace the engine's particle set (the role of P[] + NumPart).  PERIODIC runs: positions must lie in [0, BoxSize], as they do
 * in the reference after do_box_wrapping() (domain.c:81); the next decomposition fails with NGRAVS_ERR_ARG otherwise. */
int ngravs_set_particles(ngravs_ctx *ctx, const ngravs_particles_t *p);
/* Drifted tree (TreeDomainUpdateFrequency > 0: domain.c:76, predict.c:79-91): the same particles (same n, same order
 * in the caller's arrays) with new positions / OldAcc / active flags.  Keeps the last decomposition and tree topology;
 * the sorted columns and the nodes are refreshed by ngravs_force_update_tree(), which ngravs_gravity_tree() and
 * ngravs_pmforce_periodic() call by themselves when needed.  NGRAVS_ERR_STATE without a built tree. */
int ngravs_update_particles(ngravs_ctx *ctx, const ngravs_particles_t *p);
/* Update OldAcc only (second pass of accel.c:48-52 without re-uploading positions): the caller's OWN rows (NumPart =
 * ngravs_dd_num_local() of them; the imported copies of a multi-task working set are sources only and keep what they came with). */
int ngravs_set_old_acc(ngravs_ctx *ctx, const double *old_acc, int64_t stride, int on_device);

/* ---- the path ---------------------------------------------------------------------------- */
/* domain_findExtent + keys + Peano-Hilbert order (domain.c:882-944, peano.c:36-185).  Computes
 * DomainCorner/Center/Len/Fac, the 18-bit reference keys and the device-side Peano order. */
int ngravs_domain_decomposition(ngravs_ctx *ctx);
/* Tell the library that the coming step is a PM step: the stored GravPM will be recomputed before anything reads it, so it need
 * not be carried through the decomposition (ngravs_compute_accelerations(ctx, 1) does this by itself; a multi-task host that
 * calls the decomposition and the PM force separately saves two passes over the particles) */
int ngravs_discard_grav_pm(ngravs_ctx *ctx);
/* force_treebuild(): returns the number of tree nodes (>0) or a negative status. */
int64_t ngravs_force_treebuild(ngravs_ctx *ctx);
/* The dynamic tree update between rebuilds (predict.c:79-91 node drift + force_update_len(), forcetree.c:1005-1122,
 * and the node kicks of timestep.c:331-344), done as a refit: every node's per-species mass and centre of mass are
 * recomputed bottom-up from the current particle positions (exact, where the reference extrapolates them with node
 * velocities; for N_GRAVS > 1 the reference's kick of a node adds a particle's dv to the node velocities of every species,
 * timestep.c:337-340 -- the refit does not follow that), softening flags likewise, and a cell's side grows to
 * enclose what its particles now reach. */
int ngravs_force_update_tree(ngravs_ctx *ctx);
/* gravity_tree(): walk for all active targets, OldAcc update, xG (gravtree.c:102-341). */
int ngravs_gravity_tree(ngravs_ctx *ctx);
/* pmforce_periodic(): GravPM for all particles (pm_periodic.c:204-790). */
int ngravs_pmforce_periodic(ngravs_ctx *ctx);
/* compute_accelerations(0) for gravity: [PM if pm_step] + domain + build + tree (accel.c:24-58). */
int ngravs_compute_accelerations(ngravs_ctx *ctx, int pm_step);

/* ---- results, in the caller's ORIGINAL particle order -------------------------------------- */
/* Any pointer may be NULL.  stride in bytes as above.  grav_cost = ninteractions (gravtree.c).
 * Rows: exactly the caller's own particles, i.e. the n rows of the last ngravs_set_particles() (ngravs_dd_num_local() rows after
 * a library-side migration) -- the copies of other tasks' particles a multi-task step imports are never delivered, so an array
 * of NumPart rows is enough (ABI 3; ABI 2 wrote NumPart + imported rows).
 * only_active != 0: GravAccel / OldAcc / GravCost are written ONLY for the rows the last hand-over marked active, as the
 * reference does (gravtree.c:318-341 touch only Ti_endstep == Ti_Current) -- inactive rows of the caller's arrays keep
 * their values.  only_active == 0: every row is written; rows that were not walked read GravAccel = 0, GravCost = 0 and
 * their input OldAcc.  GravPM is written for all rows either way (pm_periodic.c:716-763 updates every particle). */
int ngravs_get_accel(ngravs_ctx *ctx, double *grav_accel, int64_t accel_stride,
                     double *grav_pm, int64_t pm_stride, double *old_acc, int64_t old_acc_stride,
                     float *grav_cost, int64_t cost_stride, int on_device, int only_active);
int ngravs_get_stats(ngravs_ctx *ctx, ngravs_stats_t *out);
/* Multi-task trees: top-tree leaves the last GROUP walk wanted to open although their particles were not imported (the box of a
 * group that spans several top leaves can come closer to a node than any of the leaves' boxes the import decision tested, or the
 * host changed the opening criterion after the decomposition).  Such a leaf is used as a monopole and counted here (0 in a single
 * task).  The reference walk (NGRAVS_WALK_STRICT) does not substitute: ngravs_gravity_tree() returns NGRAVS_ERR_STATE instead.
 * Replaces nothing in the reference: its export / import loop (gravtree.c:112-285) follows the walk wherever it goes. */
int ngravs_walk_unopened(ngravs_ctx *ctx, int64_t *count);
/* DomainCorner[3], DomainCenter[3], DomainLen, DomainFac (domain.c:916-923) -> out[8] */
int ngravs_get_domain(ngravs_ctx *ctx, double out[8]);
/* 18-bit reference Peano-Hilbert keys in original particle order (domain.c:938-944). */
int ngravs_get_keys(ngravs_ctx *ctx, int64_t *keys, int on_device);
/* Device-side Peano order: order[i] = original index of the i-th particle along the curve. */
int ngravs_get_order(ngravs_ctx *ctx, int32_t *order, int on_device);
/* Target shard of this rank: positions [first, first+count) of the Peano order (the role of
 * DomainMyStart/DomainMyLast, domain.c:347-456).  Results outside the shard are zero. */
int ngravs_get_shard(ngravs_ctx *ctx, int64_t *first, int64_t *count);
/* Text of the last error reported through the fatal handler. */
const char *ngravs_last_error(ngravs_ctx *ctx);

/* ---- stand-alone pieces of the path ------------------------------------------------------- */
/* peano_hilbert_key(x,y,z,bits) (peano.c:356-398), host. */
int64_t ngravs_peano_hilbert_key(int x, int y, int z, int bits);
/* Keys for n positions on the device given corner/fac (domain.c:938-944); host buffers. */
int ngravs_peano_keys(ngravs_ctx *ctx, const double *pos, int64_t n, const double corner[3],
                      double fac, int bits, int64_t *keys);
/* shortrange_fourier_force[target][source][NTAB] (forcetree.c:3246-3403): out has
 * n_gravs*n_gravs*NTAB doubles; pot_out (may be NULL) the matching shortrange_fourier_pot. */
int ngravs_shortrange_table(const ngravs_config_t *cfg, double *force_out, double *pot_out);
/* force_treeevaluate_direct for targets idx[0..nt) against all particles; with PERIODIC the nearest-image
 * sum plus lattice_corr (Ewald / lattice-sum tables, forcetree.c:3515-3529, 3803-3885), i.e. the truth
 * gravity_forcetest() compares the tree / TreePM force against; result xG into acc[3*nt]. */
int ngravs_direct_sum(ngravs_ctx *ctx, const int32_t *idx, int64_t nt, double *acc);
/* The same truth with several tasks (gravity_forcetest's export of the test particles, gravtree_forcetest.c:100-260): the direct
 * sum for nt EXPLICIT targets (position, mass -- only the BAM laws read it, may be NULL --, type; host arrays) over the particles
 * this task OWNS (not its imported copies).  Every task calls it with the test particles of ALL tasks and the host adds the
 * partial sums up (MPI_Allreduce / ncclAllReduce).  xG, lattice correction included when PERIODIC. */
int ngravs_direct_sum_targets(ngravs_ctx *ctx, const double *pos, const double *mass, const int32_t *type, int64_t nt, double *acc);

/* ---- multi-task domain decomposition ------------------------------------------------------------------
 * The role of domain_decompose()/domain_exchangeParticles() (domain.c:164-330, 554-760) and of the target export of
 * gravity_tree() (gravtree.c:112-285).  The Peano curve is cut at the leaves of an adaptive TOP TREE in key space (the
 * reference's TopNodes[], domain.c:933-1138: a cell is split while it holds more than a threshold of particles); a task owns a
 * run of leaves, holds its own particles plus copies of the particles of every foreign leaf one of its targets may open, and
 * then needs no communication for tree build and walk.  The library packs/unpacks; the HOST performs the collectives (RCCL /
 * MPI behind the vtable of ngravs_host.h, where this sequence lives as plain C):
 *   lo,hi   = ngravs_dd_local_extent()                 -> all-reduce min/max -> ngravs_dd_set_extent()
 *   ngravs_dd_set_toptree(child[])  the current top tree (every task the same)
 *   sums    = ngravs_dd_leaf_sums()                    -> all-reduce sum (device or host) -> host adapts the tree, cuts the curve
 *   records = ngravs_dd_pack(0, leaf_owner, ...)       -> all-to-all-v          -> ngravs_dd_apply_migration()
 *   records = ngravs_dd_pack_leaves(reqmask, ...)      -> all-to-all-v          -> ngravs_dd_set_halo()
 *   ngravs_dd_set_top(node sums, present leaves); ngravs_domain_decomposition(); slab PM (below); ngravs_gravity_tree();
 *   results: ngravs_get_accel() (own rows), ids from ngravs_dd_get_ids().
 * Records are 56 bytes (NGRAVS_DD_RECORD_BYTES): x,y,z,mass,old_acc,grav_cost (f64), meta (i64: type | active<<8 | id<<16).  world_size <= 64. */
#define NGRAVS_DD_RECORD_BYTES 56
#define NGRAVS_DD_MAX_RECORD_BYTES 80
/* bytes per record of ngravs_dd_pack(what, ...): what = 0 (migration) of a TreePM run appends P[].GravPM[3] (the particle's
 * long-range force of the last PM step travels with it, as the whole particle_data does in domain_exchangeParticles,
 * domain.c:695-795; OldAcc on non-PM steps needs it, gravtree.c:318-330): 80 bytes; everything else NGRAVS_DD_RECORD_BYTES */
int64_t ngravs_dd_record_bytes(ngravs_ctx *ctx, int what);
int64_t ngravs_dd_num_local(ngravs_ctx *ctx);
int ngravs_dd_local_extent(ngravs_ctx *ctx, double lo[3], double hi[3]);             /* domain.c:894-905 */
/* peano_hilbert_order() of P[] (domain.c:146, peano.c:36-90, reorder_particles :261-312) for the library's OWN copy of the
 * columns: the rows of the own particles are put into the Peano order of the last local decomposition, so that the passes that
 * visit particles in tree order (per-leaf sums, keys, gather, result scatter) read memory in order.  Only for callers that name
 * rows by ID (ngravs_dd_get_ids) -- the host layer's whole-decomposition driver (ngravs_host.h) calls it; a host that owns the row order (gadget_glue.c:
 * P[] is the reference's, already in this order) never does.  force = 0: only when more than 1 in 8 reads of the last gather
 * left its neighbourhood (64 rows).  Call it at the start of a step: imported copies, order, tree and results of the last
 * step are gone afterwards.  Returns 1 if rows moved, 0 if they were left, < 0: status. */
int ngravs_dd_peano_order(ngravs_ctx *ctx, int force);
int ngravs_dd_set_extent(ngravs_ctx *ctx, const double lo[3], const double hi[3]);   /* result of domain.c:906-907 */
/* DomainCorner[3], DomainCenter[3], DomainLen, DomainFac as set by ngravs_dd_set_extent (before the local Peano order exists) */
int ngravs_get_domain_extent(ngravs_ctx *ctx, double out[8]);
/* The top tree as its child table: child[t] = index of the first of the 8 consecutive children (key order) of node t, or -1 for a
 * leaf; node 0 = the root.  The library numbers the leaves in depth-first (= curve) order.  It keeps the table between steps
 * (ngravs_dd_get_toptree: *child points into the context, valid until the next set; nnode 0: none yet). */
int ngravs_dd_set_toptree(ngravs_ctx *ctx, int32_t nnode, const int32_t *child);
int ngravs_dd_get_toptree(ngravs_ctx *ctx, int32_t *nnode, const int32_t **child);
/* Per top LEAF, curve order, NGRAVS_TOP_CW(n_gravs) doubles: [0] the work sum(1 + GravCost) (domain_sumCost, domain.c:859-862),
 * [1..6] particles per type, then per species mass and mass-weighted position (the local part of DomainMoment[],
 * forcetree.c:766-850) -- of the own particles.  DEVICE buffer owned by the library (valid until the next call);
 * *count = nleaf * NGRAVS_TOP_CW doubles.  The host all-reduces it (in place on the device if it can) and reads it back. */
#define NGRAVS_TOP_CW(ng) (7 + 4 * (ng))
int ngravs_dd_leaf_sums(ngravs_ctx *ctx, void **dev_sums, int64_t *count);
/* min over the own active particles of ErrTolForceAcc * OldAcc and of the softening length: out[2].  With the tuning "dd_keep" > 0
 * (a decomposition that will be kept over steps with other active sets, All.TreeDomainUpdateFrequency > 0): over ALL own particles */
int ngravs_dd_target_bounds(ngravs_ctx *ctx, double out[2]);
/* "dd_keep" x the side of the domain cube: how far the import decision lets a target drift out of its top leaf's cell while the
 * decomposition is kept (0: the decomposition serves this step only) */
int ngravs_dd_keep_margin(ngravs_ctx *ctx, double *margin);
/* what = 0: the records of the own particles whose leaf belongs to another task (leaf_owner[leaf], host array), grouped by
 * destination; counts[r] records go to task r */
int ngravs_dd_pack(ngravs_ctx *ctx, int what, const int32_t *leaf_owner, int nranks, int my_rank, int64_t *counts, void **dev_records,
                   int64_t *nrec);
/* destination task of every local particle under the owner map (its own rank if it stays), host array of ngravs_dd_num_local() ints */
int ngravs_dd_get_dest(ngravs_ctx *ctx, const int32_t *leaf_owner, int32_t *dest);
/* ---- the global top of the tree: top-leaf moments and tree-node import (force_exchange_pseudodata / force_treeupdate_pseudos,
 * forcetree.c:766-996; replaces the target export / partial-force import of gravtree.c:112-285) ---------------------------------
 * The host decides which foreign leaves the task's targets may have to open (ngravs_host_import_request: the walk's own opening
 * tests against the task's domain), the owners ship ALL particles of the requested leaves (ngravs_dd_pack_leaves: reqmask[leaf]
 * bit r = task r asked for it), and ngravs_dd_set_top() hands the global sums of every top node over: the next tree build
 * forces the topology of the top tree from the global counts -- it is the single-task tree's --, gives the top nodes global
 * monopoles, and turns every leaf whose particles are elsewhere into a pseudo node.  Forces are then independent of the
 * number of tasks (domain.c:18-21). */
int ngravs_dd_pack_leaves(ngravs_ctx *ctx, const uint64_t *reqmask, int nranks, int my_rank, int64_t *counts, void **dev_records,
                          int64_t *nrec);
/* node_sums: NGRAVS_TOP_CW doubles per top NODE ([0] = global particle count); present[leaf] != 0: the leaf's particles are on
 * this task (own or imported).  NULL, NULL: single-task trees again. */
int ngravs_dd_set_top(ngravs_ctx *ctx, const double *node_sums, const uint8_t *present);
/* a library-owned device buffer for nrec incoming records (valid until the next ngravs_dd_recv_buffer call) */
int ngravs_dd_recv_buffer(ngravs_ctx *ctx, int64_t nrec, void **dev_records);
int ngravs_dd_apply_migration(ngravs_ctx *ctx, const void *dev_records, int64_t nrec);
int ngravs_dd_set_halo(ngravs_ctx *ctx, const void *dev_records, int64_t nrec);
/* ---- kept decomposition: the steps on which domain.c:76 keeps domain and tree (All.TreeDomainUpdateFrequency > 0) -------------
 * The reference drifts its nodes with their velocities, refreshes the other tasks' top-leaf moments (force_update_pseudoparticles,
 * forcetree.c:753) and the sides of the top nodes (force_update_node_len_toptree, :1096-1122), and goes on exporting targets.  Here
 * the cut, the top tree and the import requests of the last decomposition stay; the caller hands its own rows over again
 * (ngravs_update_particles: same rows, drifted positions) and ngravs_host_kept_step in ngravs_host.h does the rest with two
 * collectives: the owners ship the drifted particles of the leaves that were asked for at the decomposition -- the same records in
 * the same order (ngravs_dd_pack_leaves_kept -> all-to-all-v -> ngravs_dd_refresh_halo); the tree is refit
 * (ngravs_force_update_tree); the per-leaf sums BY THE MEMBERSHIP OF THE DECOMPOSITION plus the grown side of every leaf's cell in
 * its owner's tree are all-reduced (ngravs_dd_leaf_sums_kept: NGRAVS_TOP_CW + 1 doubles per leaf) and set
 * (ngravs_dd_update_top: node sums as for ngravs_dd_set_top, one side per leaf).  The refit tree is then the single task's refit
 * tree wherever this task's targets look.  NGRAVS_ERR_STATE without a kept decomposition (none yet, rows changed, single task). */
int ngravs_dd_leaf_sums_kept(ngravs_ctx *ctx, void **dev_sums, int64_t *count);
int ngravs_dd_pack_leaves_kept(ngravs_ctx *ctx, int64_t *counts, void **dev_records, int64_t *nrec);
int ngravs_dd_refresh_halo(ngravs_ctx *ctx, const void *dev_records, int64_t nrec);
int ngravs_dd_update_top(ngravs_ctx *ctx, const double *node_sums, const double *leaf_len);
/* the plan of the kept decomposition (host memory of the context, valid until the next decomposition): this task's rank and the
 * task count, owner of every leaf, present[leaf] (own or imported here), global sums of every top node */
int ngravs_dd_get_kept(ngravs_ctx *ctx, int32_t *rank, int32_t *world, const int32_t **leaf_owner, const uint8_t **present,
                       const double **node_sums);
int ngravs_dd_set_ids(ngravs_ctx *ctx, const int64_t *ids, int on_device);
int ngravs_dd_get_ids(ngravs_ctx *ctx, int64_t *ids, int on_device);
/* ---- pmforce_periodic() for many tasks: x-slab decomposed mesh -----------------------------------------------------------
 * The reference's scheme (pm_periodic.c:74-123 slab tables; :336-427 density patches -> slab owners; :433,:525 distributed
 * FFT in transposed order; :436-520 Green's function on the transposed layout; :529-670 potential bricks with ghost planes
 * back; :681-763 finite differences + CIC gather on the brick).  A task deposits its own particles (not its halo copies)
 * into a brick -- the box of mesh cells (lo[j] + i) mod PMGRID, 0 <= i < ext[j], its CIC clouds touch -- and four
 * all-to-all-v exchanges move planes of that brick / of the slabs:
 *     ngravs_pm_slab_begin(ctx, rank, world, bbox);              bbox = {lo[3], ext[3]};   host: all-gather -> all_bbox[world][6]
 *     for stage = 0..3:
 *         ngravs_pm_slab_pack(ctx, stage, all_bbox, send_counts, recv_counts, &send, &recv);
 *         host: all-to-all-v of doubles, send_counts[r] to / recv_counts[r] from task r, blocks in task order, device buffers
 *         ngravs_pm_slab_unpack(ctx, stage);
 *   stage 0 density planes -> slab owners (+ 2-D r2c FFTs)   1 transpose x<->y (+ 1-D FFTs, Green, inverse 1-D FFTs)
 *   stage 2 transpose back (+ 2-D c2r FFTs)                  3 potential planes with 2 ghost cells per side -> bricks
 *                                                               (+ gradient and CIC gather: GravPM of the own particles)
 * Task r owns the x planes [r*PMGRID/world, (r+1)*PMGRID/world) (integer division), and the same y range of k-space.
 * Mesh memory per task: NG x (brick + two slabs + exchange buffers).  world <= PMGRID/2. */
int ngravs_pm_slab_begin(ngravs_ctx *ctx, int rank, int world, int32_t bbox[6]);
int ngravs_pm_slab_pack(ngravs_ctx *ctx, int stage, const int32_t *all_bbox, int64_t *send_counts, int64_t *recv_counts,
                        void **send, void **recv);
int ngravs_pm_slab_unpack(ngravs_ctx *ctx, int stage);
/* payload this task sent to OTHER tasks in each of the four exchanges of the last step, bytes */
int ngravs_pm_slab_bytes(ngravs_ctx *ctx, double bytes[4]);

#ifdef __cplusplus
}
#endif
#endif /* NGRAVS_HIP_H */
