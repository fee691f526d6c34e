/* gadget_glue.c -- the reference-side binding: what a Gadget-2.0.7-ngravs maintainer adds to the reference tree so that
 * its own entry points run on libngravs_hip.so.
 *
 * It is compiled INSIDE the reference tree (it includes the reference's allvars.h / proto.h and therefore sees the real
 * struct particle_data, All, TypeToGrav[], AccelFxns[][] ...), replacing gravtree.o, forcetree.o (walks + build),
 * pm_periodic.o, domain.o, peano.o and gravtree_forcetest.o in Makefile.reference's OBJS.  The
 * reference's signatures are kept exactly (proto.h:36,77,78,86,113,114,149,150,155,161), so accel.c, run.c, init.c,
 * timestep.c, predict.c are untouched.
 *
 * This repository cannot build the reference (its headers need GSL and FFTW-2); tests/test_host_glue.py compiles this file
 * with -Wall -Wextra -Werror against tests/glue_stub/ (declarations of exactly the globals and prototypes used here, layouts
 * from SURVEY.md 8(a')), checks with nm that it defines every symbol the link recipe needs, and RUNS it on the GPU with 1, 2
 * and 3 tasks (tests/glue_stub/glue_driver.c: the reference's globals, MPI as forked processes over shared memory and the call
 * sequence of a first step and of a short-range step with a sparse active set; P[]'s results equal the library's,
 * forcetest.txt is written);
 * host/host_shim_test.c drives the same library calls and the same ngravs_host_* helpers with two tasks.
 *
 * One task (NTask == 1): P[] is handed over with byte strides, results come back in P[]'s order.
 * Several tasks: the cut of the Peano curve (work-weighted, domain.c:347-544) is found by ngravs_host_domain_owners();
 * P[] itself is migrated HERE with MPI (it carries Vel, ID, timestep data the library never sees -- domain.c:695-795), the
 * migrated P[] is handed over, and ngravs_host_domain_halo() / ngravs_host_pmforce_periodic() run the device-side exchanges
 * through the MPI vtable below.  Results of the first NumPart rows are P[]'s.
 *
 * TreeDomainUpdateFrequency: with 0.0 every step re-decomposes and rebuilds.  With a value > 0 the
 * reference keeps decomposition and tree while fewer than TreeDomainUpdateFrequency*TotNumPart forces have been computed
 * (domain.c:76) and drifts/kicks the node moments instead (predict.c:79-91, timestep.c:331-344); here those steps hand the
 * drifted positions over with ngravs_update_particles() and the library refits the nodes (ngravs_force_update_tree), so
 * force_update_len() and the node drift/kick loops become no-ops (single task only; several tasks always re-decompose).
 */
#ifdef NGRAVS_BUILD_INSIDE_REFERENCE

#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <mpi.h>
#include "allvars.h"
#include "proto.h"
#include "ngravs.h"
#include "ngravs_hip.h"
#include "ngravs_host.h"
#ifdef NGRAVS_WITH_RCCL
#include "ngravs_comm_rccl.h"	/* the exchanges of the path on RCCL over xGMI instead of MPI (link libngravs_rccl.so) */
static ngravs_rccl *Rccl = NULL;
#endif

#ifndef DOUBLEPRECISION
#error "libngravs_hip reads P[].Pos / Mass / OldAcc as double: build the reference with -DDOUBLEPRECISION (FLOAT = double, allvars.h:93-97)"
#endif
#ifndef UNEQUALSOFTENINGS
#error "libngravs_hip applies per-type softening lengths: build the reference with -DUNEQUALSOFTENINGS"
#endif

static ngravs_ctx *Ctx = NULL;
static ngravs_comm Comm;
static unsigned char *ActiveFlag = NULL;

/* wired function pointer -> law id (the device cannot call through `gravity` pointers) */
static int law_id(gravity f)
{
  if(f == none)
    return NGRAVS_LAW_NONE;
  if(f == newtonian || f == pgdelta || f == normed_pgdelta)
    return NGRAVS_LAW_NEWTON;
  if(f == neg_newtonian || f == neg_pgdelta)
    return NGRAVS_LAW_NEG_NEWTON;
  if(f == yukawa || f == pgyukawa || f == normed_pgyukawa)
    return NGRAVS_LAW_YUKAWA;
  if(f == coloyuk || f == pgcoloyuk || f == normed_pgcoloyuk)
    return NGRAVS_LAW_COLOYUK;
  if(f == bambam)
    return NGRAVS_LAW_BAMBAM;
  if(f == sourcebambaryon)
    return NGRAVS_LAW_SOURCEBAM;
  if(f == sourcebaryonbam)
    return NGRAVS_LAW_TARGETBAM;
  printf("ngravs-hip: a wired force law has no device implementation\n");
  endrun(1050);
  return -1;
}
static int spline_id(gravity f)
{
  if(f == none)
    return NGRAVS_SPLINE_NONE;
  if(f == plummer)
    return NGRAVS_SPLINE_PLUMMER;
  if(f == neg_plummer)
    return NGRAVS_SPLINE_NEG_PLUMMER;
  if(f == bambam_spline)
    return NGRAVS_SPLINE_BAMBAM;
  if(f == sourcebambaryon_spline)
    return NGRAVS_SPLINE_SOURCEBAM;
  if(f == sourcebaryonbam_spline)
    return NGRAVS_SPLINE_TARGETBAM;
  endrun(1051);
  return -1;
}

static void on_fatal(int code, const char *msg)
{
  printf("ngravs-hip: %s\n", msg);
  endrun(code > 0 ? code : 1052);
}

static void must(int rc, int code)
{
  if(rc != 0)
    {
      printf("ngravs-hip: call failed with status %d: %s\n", rc, Ctx ? ngravs_last_error(Ctx) : "");
      endrun(code);
    }
}

static double CommSeconds = 0;	/* wall time inside the communicator's callbacks (All.CPU_CommSum, gravtree.c:452) */
#ifndef NGRAVS_WITH_RCCL
/* ---- the communicator vtable over MPI (include/ngravs_host.h); with -DNGRAVS_WITH_RCCL: include/ngravs_comm_rccl.h instead ---- */
static int mpi_allreduce(void *user, void *buf, int64_t count, int dtype, int op)
{
  MPI_Op o = op == NGRAVS_OP_SUM ? MPI_SUM : (op == NGRAVS_OP_MIN ? MPI_MIN : MPI_MAX);
  const double t0 = second();
  int rc;
  (void)user;
  rc = MPI_Allreduce(MPI_IN_PLACE, buf, (int)count, dtype == NGRAVS_T_F64 ? MPI_DOUBLE : MPI_LONG_LONG, o, MPI_COMM_WORLD) != MPI_SUCCESS;
  CommSeconds += timediff(t0, second());
  return rc;
}
static int mpi_allgather(void *user, const void *send, void *recv, int64_t bytes)
{
  const double t0 = second();
  int rc;
  (void)user;
  rc = MPI_Allgather((void *)send, (int)bytes, MPI_BYTE, recv, (int)bytes, MPI_BYTE, MPI_COMM_WORLD) != MPI_SUCCESS;
  CommSeconds += timediff(t0, second());
  return rc;
}
/* blocks beyond 2 GB per peer are cut into rounds of <= 1 GB (MPI counts are int) */
static int mpi_alltoallv(void *user, const void *send, const int64_t *sbytes, const int64_t *sdispl, void *recv,
                         const int64_t *rbytes, const int64_t *rdispl)
{
  const int64_t chunk = 1 << 30;
  const double t0 = second();
  int64_t off = 0, more = 1;
  int *sc = malloc(sizeof(int) * 4 * NTask), *sd = sc + NTask, *rc = sd + NTask, *rd = rc + NTask, r, err = 0;
  (void)user;
  /* displacements are prefix sums of the byte counts: exchange round by round with per-peer offsets */
  while(more && !err)
    {
      MPI_Request *req = malloc(sizeof(MPI_Request) * 2 * NTask);
      int nreq = 0;
      more = 0;
      for(r = 0; r < NTask; r++)
        {
          int64_t s = sbytes[r] - off, q = rbytes[r] - off;
          s = s < 0 ? 0 : (s > chunk ? chunk : s);
          q = q < 0 ? 0 : (q > chunk ? chunk : q);
          sc[r] = (int)s;
          rc[r] = (int)q;
          if(sbytes[r] - off > chunk || rbytes[r] - off > chunk)
            more = 1;
          if(q > 0)
            err |= MPI_Irecv((char *)recv + rdispl[r] + off, rc[r], MPI_BYTE, r, 7711, MPI_COMM_WORLD, &req[nreq++]) != MPI_SUCCESS;
        }
      for(r = 0; r < NTask; r++)
        if(sc[r] > 0)
          err |= MPI_Isend((char *)send + sdispl[r] + off, sc[r], MPI_BYTE, r, 7711, MPI_COMM_WORLD, &req[nreq++]) != MPI_SUCCESS;
      err |= MPI_Waitall(nreq, req, MPI_STATUSES_IGNORE) != MPI_SUCCESS;
      free(req);
      off += chunk;
    }
  (void)sd;
  (void)rd;
  free(sc);
  CommSeconds += timediff(t0, second());
  return err;
}
#endif

static void ensure_ctx(void)
{
  ngravs_config_t cfg;
  int i, j;
  if(Ctx)
    return;
  ngravs_config_default(&cfg);
  cfg.n_gravs = N_GRAVS;
#ifdef PERIODIC
  cfg.periodic = 1;
#endif
#ifdef PMGRID
  cfg.pmgrid = PMGRID;
  cfg.asmth = All.Asmth[0];
  cfg.rcut = All.Rcut[0];
#endif
  cfg.box_size = All.BoxSize;
  cfg.G = All.G;
  cfg.err_tol_theta = All.ErrTolTheta;
  cfg.err_tol_force_acc = All.ErrTolForceAcc;
  cfg.tree_alloc_factor = All.TreeAllocFactor;
  for(i = 0; i < 6; i++)
    {
      cfg.force_softening[i] = All.ForceSoftening[i];
      cfg.type_to_grav[i] = TypeToGrav[i];
    }
  for(i = 0; i < N_GRAVS; i++)
    for(j = 0; j < N_GRAVS; j++)
      {
        cfg.law_accel[i][j] = law_id(AccelFxns[i][j]);
        cfg.law_spline[i][j] = spline_id(AccelSplines[i][j]);
#ifdef PMGRID
        cfg.law_greens[i][j] = law_id(GreensFxns[i][j]);
        cfg.law_normed[i][j] = law_id(NormedGreensFxns[i][j]);
#endif
      }
#ifdef YUKAWA_IMASS
  cfg.yukawa_imass = YUKAWA_IMASS;
#endif
#ifdef BAM_EPSILON
  cfg.bam_epsilon = BAM_EPSILON;
#endif
#ifdef NGRAVS_GLUE_WALK_STRICT
  cfg.walk_mode = NGRAVS_WALK_STRICT;	/* the reference walk, target by target (tests: several tasks must reproduce one task's forces and counts) */
#else
  cfg.walk_mode = NGRAVS_WALK_GROUP;
#endif
  /* the library always sees its working set (own particles + halo copies) as one task; the tasks are joined by Comm */
  cfg.rank = 0;
  cfg.world_size = 1;
  cfg.device = ThisTask;	/* one rank per GPU of the node */
#ifdef NGRAVS_GLUE_DEVICE
  cfg.device = NGRAVS_GLUE_DEVICE;	/* rehearsals: several tasks on one GPU */
#endif
  if(ngravs_create(&cfg, &Ctx) != NGRAVS_OK)
    endrun(1053);
  ngravs_set_fatal_handler(Ctx, on_fatal);
  memset(&Comm, 0, sizeof(Comm));
#ifdef NGRAVS_WITH_RCCL
  /* RCCL over xGMI: every exchange of the path (domain.c:695-747, gravtree.c:195-257, forcetree.c:811-816, pm_periodic.c:385-389,
   * 433, 525, 655-660) on the library's device buffers.  MPI only hands the ncclUniqueId round, once. */
  {
    char id[NGRAVS_RCCL_ID_BYTES];
    memset(id, 0, sizeof(id));
    if(ThisTask == 0 && ngravs_rccl_unique_id(id))
      endrun(1071);
    MPI_Bcast(id, NGRAVS_RCCL_ID_BYTES, MPI_BYTE, 0, MPI_COMM_WORLD);
    if(ngravs_rccl_create(id, ThisTask, NTask, cfg.device, &Rccl))
      {
	printf("ngravs-hip: ncclCommInitRank failed on task %d (one task per GPU)\n", ThisTask);
	endrun(1072);
      }
    if(ngravs_rccl_selftest(Rccl))   /* every collective once with known answers: a fabric problem ends the run here, with a reason */
      {
        printf("ngravs glue: task %d: RCCL self test: %s\n", ThisTask, ngravs_rccl_last_error(Rccl));
        endrun(1056);
      }
    ngravs_rccl_fill(Rccl, &Comm);
  }
#else
  Comm.rank = ThisTask;
  Comm.size = NTask;
  Comm.device_buffers = 0;	/* plain MPI: ngravs_host stages the exchange buffers through host memory (1 with a GPU-aware MPI) */
  Comm.allreduce = mpi_allreduce;
  Comm.allgather = mpi_allgather;
  Comm.alltoallv = mpi_alltoallv;
  Comm.allreduce_dev = NULL;
#endif
  if(NTask > 1 && All.TotN_gas > 0)
    {
      printf("ngravs-hip: gas particles (SphP[]) are not migrated by this glue\n");
      endrun(1055);
    }
}

/* hand P[0..NumPart) over: the fields of SURVEY.md 8(b), with the byte strides of struct particle_data */
static int PushGravPM = 0;	/* hand P[].GravPM over even on a PM step (decompose_several_tasks(1): it has been computed already) */
static void push_particles(int keep_tree)
{
  ngravs_particles_t p;
  int i;
  ActiveFlag = realloc(ActiveFlag, NumPart > 0 ? NumPart : 1);
  for(i = 0; i < NumPart; i++)
    ActiveFlag[i] = (P[i].Ti_endstep == All.Ti_Current);	/* gravtree.c:113 */
  memset(&p, 0, sizeof(p));
  p.n = NumPart;
  p.pos = &P[0].Pos[0];
  p.pos_stride = sizeof(struct particle_data);
  p.mass = &P[0].Mass;
  p.mass_stride = sizeof(struct particle_data);
  p.type = &P[0].Type;
  p.type_stride = sizeof(struct particle_data);
  p.old_acc = &P[0].OldAcc;
  p.old_acc_stride = sizeof(struct particle_data);
  p.active = ActiveFlag;
  p.active_stride = 1;
  p.grav_cost = &P[0].GravCost;	/* the work weight of the next domain cut (domain.c:859-862) */
  p.grav_cost_stride = sizeof(struct particle_data);
#ifdef PMGRID
  if(All.PM_Ti_endstep != All.Ti_Current || PushGravPM)	/* P[].GravPM of the last PM step enters OldAcc on non-PM steps (gravtree.c:318-330) */
    {
      p.grav_pm = &P[0].GravPM[0];
      p.grav_pm_stride = sizeof(struct particle_data);
    }
#endif
  must(keep_tree ? ngravs_update_particles(Ctx, &p) : ngravs_set_particles(Ctx, &p), 1056);
}

/* domain_decompose() bookkeeping (domain.c:173-195) */
static void count_types(void)
{
  int i, j, *temp;
  for(i = 0; i < 6; i++)
    NtypeLocal[i] = 0;
  for(i = 0; i < N_GRAVS; i++)
    NgravLocal[i] = 0;
  for(i = 0; i < NumPart; i++)
    {
      NtypeLocal[P[i].Type]++;
      NgravLocal[TypeToGrav[P[i].Type]]++;
    }
  temp = malloc(NTask * 6 * sizeof(int));
  MPI_Allgather(NtypeLocal, 6, MPI_INT, temp, 6, MPI_INT, MPI_COMM_WORLD);
  for(i = 0; i < 6; i++)
    {
      Ntype[i] = 0;
      for(j = 0; j < NTask; j++)
	Ntype[i] += temp[j * 6 + i];
    }
  free(temp);
}

/* domain_exchangeParticles (domain.c:695-795) for P[]: every particle goes to dest[i]; one all-to-all-v of whole records */
static void exchange_particles(const int32_t *dest)
{
  int *scount = calloc(4 * NTask, sizeof(int)), *sdispl = scount + NTask, *rcount = sdispl + NTask, *rdispl = rcount + NTask;
  int i, r, nsend = 0, nrecv = 0, nkeep = 0;
  struct particle_data *sendbuf;
  int *cursor;
  for(i = 0; i < NumPart; i++)
    if(dest[i] != ThisTask)
      scount[dest[i]]++;
  MPI_Alltoall(scount, 1, MPI_INT, rcount, 1, MPI_INT, MPI_COMM_WORLD);
  for(r = 0; r < NTask; r++)
    {
      sdispl[r] = nsend;
      rdispl[r] = nrecv;
      nsend += scount[r];
      nrecv += rcount[r];
    }
  if(NumPart - nsend + nrecv > All.MaxPart)
    {
      printf("task %d: domain decomposition needs %d particles, MaxPart = %d\n", ThisTask, NumPart - nsend + nrecv, All.MaxPart);
      endrun(1313);		/* the reference's own code for this condition, domain.c */
    }
  sendbuf = malloc(sizeof(struct particle_data) * (nsend > 0 ? nsend : 1));
  cursor = malloc(sizeof(int) * NTask);
  for(r = 0; r < NTask; r++)
    cursor[r] = sdispl[r];
  for(i = 0; i < NumPart; i++)
    if(dest[i] != ThisTask)
      sendbuf[cursor[dest[i]]++] = P[i];
    else
      P[nkeep++] = P[i];
  free(cursor);
  for(r = 0; r < NTask; r++)	/* counts in bytes */
    {
      scount[r] *= sizeof(struct particle_data);
      sdispl[r] *= sizeof(struct particle_data);
      rcount[r] *= sizeof(struct particle_data);
      rdispl[r] *= sizeof(struct particle_data);
    }
  MPI_Alltoallv(sendbuf, scount, sdispl, MPI_BYTE, &P[nkeep], rcount, rdispl, MPI_BYTE, MPI_COMM_WORLD);
  NumPart = nkeep + nrecv;
  free(sendbuf);
  free(scount);
}

/* Several tasks: top tree + cut from all-reduced leaf sums, migration of whole particle_data records, import of the top leaves
 * this task's targets may open (domain_decompose domain.c:164-330; force_exchange_pseudodata forcetree.c:766-850; what replaces
 * the export / import loop of gravtree.c:112-285).  The import is decided for the opening criterion and the OldAcc in force NOW:
 * DdUseTheta remembers which.  again != 0: P[] has already been handed over for this step and GravPM computed (the second
 * gravity_tree() of a first step, see there). */
#ifndef NGRAVS_GLUE_KEEP_MARGIN
#define NGRAVS_GLUE_KEEP_MARGIN 0.002	/* of the domain cube's side: the drift a kept decomposition allows for before it is cut again */
#endif
static int DdUseTheta = -1;
static int KeptStep = 0;		/* this step walks the kept decomposition of several tasks (ngravs_host_kept_step) */
static long long NumImported = 0;	/* copies of other tasks' particles in this task's tree (the last decomposition) */
static void decompose_several_tasks(int again)
{
  ngravs_dd_plan plan;
  ngravs_dd_info info;
  int32_t *dest;
  if(again)
    {
      PushGravPM = 1;
      push_particles(0);
    }
  /* a decomposition that will serve several steps (domain.c:76) imports for all own particles, with room to drift */
  must(ngravs_set_tuning(Ctx, "dd_keep", All.TreeDomainUpdateFrequency > 0 ? NGRAVS_GLUE_KEEP_MARGIN : 0.0), 1076);
  must(ngravs_set_opening(Ctx, All.ErrTolTheta, All.ErrTolForceAcc), 1062);
  must(ngravs_host_domain_owners(Ctx, &Comm, 0, All.PartAllocFactor, &plan, &info), 1058);
  dest = malloc(sizeof(int32_t) * (NumPart > 0 ? NumPart : 1));
  must(ngravs_dd_get_dest(Ctx, plan.leaf_owner, dest), 1059);
  exchange_particles(dest);
  free(dest);
  push_particles(0);		/* the migrated P[]: its order is the order of the library's results */
  PushGravPM = 0;
  must(ngravs_host_domain_halo(Ctx, &Comm, &plan, &info), 1060);
  ngravs_host_plan_free(&plan);
  DdUseTheta = All.ErrTolTheta != 0;
  NumImported = (long long)info.n_halo;
  if(ThisTask == 0)
    printf("work-load balance=%g   memory-balance=%g\n", info.work_balance, info.memory_balance);	/* domain.c:257-258 */
}

/* proto.h:36 */
void domain_Decomposition(void)
{
  double d[8], t0, t1;
  ensure_ctx();
#ifdef PMGRID
  if(All.PM_Ti_endstep == All.Ti_Current)	/* domain.c:66-73: PM steps always re-decompose (particles get wrapped) */
    All.NumForcesSinceLastDomainDecomp = 1 + All.TotNumPart * All.TreeDomainUpdateFrequency;
#endif
  KeptStep = 0;
  /* domain.c:76.  Several tasks keep a decomposition the same way one task keeps its tree: same cut, same top tree, same imported
   * leaves (DdUseTheta >= 0: there is one) -- the copies follow their originals and the top nodes get their moments and sides
   * again, two collectives (ngravs_host_kept_step; force_update_pseudoparticles + force_update_node_len_toptree there). */
  if((NTask > 1 && DdUseTheta < 0) || All.NumForcesSinceLastDomainDecomp > All.TotNumPart * All.TreeDomainUpdateFrequency)
    {
      t0 = second();
#ifdef PERIODIC
      do_box_wrapping();
#endif
      push_particles(0);
#ifdef PMGRID
      if(All.PM_Ti_endstep == All.Ti_Current)
	must(ngravs_discard_grav_pm(Ctx), 1073);	/* a PM step: long_range_force() recomputes GravPM before gravity_tree() reads it */
#endif
      if(NTask == 1)
	must(ngravs_domain_decomposition(Ctx), 1057);
      else
	decompose_several_tasks(0);
      /* DomainCorner[3], DomainCenter[3], DomainLen, DomainFac are four separate globals (allvars.c:44-47) */
      must(ngravs_get_domain(Ctx, d), 1061);
      DomainCorner[0] = d[0];
      DomainCorner[1] = d[1];
      DomainCorner[2] = d[2];
      DomainCenter[0] = d[3];
      DomainCenter[1] = d[4];
      DomainCenter[2] = d[5];
      DomainLen = d[6];
      DomainFac = d[7];
      count_types();
      All.NumForcesSinceLastDomainDecomp = 0;
      TreeReconstructFlag = 1;
      t1 = second();
      {
	ngravs_stats_t st;
	ngravs_get_stats(Ctx, &st);
	All.CPU_Peano += st.t_peano;	/* device time of keys + sort + gather (peano_hilbert_order) */
	All.CPU_Domain += timediff(t0, t1) - st.t_peano;
      }
    }
  else if(NTask == 1)
    push_particles(1);		/* drifted tree: gravity_tree() refits it (ngravs_force_update_tree) */
  else
    {
      ngravs_dd_info info;
      t0 = second();
      push_particles(1);
      must(ngravs_host_kept_step(Ctx, &Comm, &info), 1075);
      KeptStep = 1;
      All.CPU_Domain += timediff(t0, second());
    }
}

/* proto.h:89, :96 -- the library recomputes node moments and cell sides from the particles (ngravs_force_update_tree);
 * the host-side node loops of predict.c:83-86 and timestep.c:331-344 run empty because Numnodestree stays 0 and
 * Father[] stays -1 (force_treeallocate below) */
void force_update_len(void)
{
}
void force_update_pseudoparticles(void)
{
}

/* proto.h:78 */
int force_treebuild(int npart)
{
  (void)npart;
  return (int)ngravs_force_treebuild(Ctx);	/* the node count is only reported; Numnodestree (host-side nodes) stays 0 */
}

/* proto.h:77, :86 -- the device library owns the tree memory */
void force_treeallocate(int maxnodes, int maxpart)
{
  int i;
  (void)maxnodes;
  Father = malloc(sizeof(int) * (maxpart > 0 ? maxpart : 1));	/* timestep.c:333 reads Father[i]: no host-side parents */
  for(i = 0; i < maxpart; i++)
    Father[i] = -1;
  Numnodestree = 0;		/* predict.c:83: no host-side nodes to drift */
}
void force_treefree(void)
{
  free(Father);
  Father = NULL;
}

/* proto.h:184 -- gravtree.c:468-518: the softening table of the six particle types, comoving lengths capped at their
 * physical maxima; All.ForceSoftening = 2.8 x the Plummer-equivalent length.  Called by init() (init.c:60), by
 * compute_potential() (potential.c:40) and at the top of every gravity_tree() of a comoving run (gravtree.c:50-51): the
 * device library is told every time (ngravs_set_softening), it keeps no copy of its own beyond the last one handed over. */
void set_softenings(void)
{
  const double comoving[6] = { All.SofteningGas, All.SofteningHalo, All.SofteningDisk, All.SofteningBulge, All.SofteningStars,
    All.SofteningBndry
  };
  const double maxphys[6] = { All.SofteningGasMaxPhys, All.SofteningHaloMaxPhys, All.SofteningDiskMaxPhys,
    All.SofteningBulgeMaxPhys, All.SofteningStarsMaxPhys, All.SofteningBndryMaxPhys
  };
  int t;
  for(t = 0; t < 6; t++)
    {
      double eps = comoving[t];
      if(All.ComovingIntegrationOn && eps * All.Time > maxphys[t])
	eps = maxphys[t] / All.Time;
      All.SofteningTable[t] = eps;
      All.ForceSoftening[t] = 2.8 * eps;
    }
  All.MinGasHsml = All.MinGasHsmlFractional * All.ForceSoftening[0];
  if(Ctx)
    must(ngravs_set_softening(Ctx, All.ForceSoftening), 1069);
}

/* proto.h:88 -- forcetree.c:1134: hmax of tree nodes that hold SPH particles, for the hydro neighbour search (accel.c:74).  The
 * device tree holds no gas (ensure_ctx refuses SphP[] with several tasks; density()/hydro_force() walk the host-side ngb tree,
 * which this glue does not provide): nothing to update. */
void force_update_hmax(void)
{
}

#ifdef PERIODIC
/* proto.h:61 -- forcetree.c:3611: the Ewald / lattice-sum correction tables (begrun.c:48, under PERIODIC && (!PMGRID ||
 * FORCETEST)).  The library tabulates them on the device on first use (k_lattice_table) and needs no file cache. */
void lattice_init(void)
{
}
#endif

/* proto.h:83-84, :163 -- potentials are outside this path (the reference's own potential walk does not
 * compile, forcetree.c:2748-2756): compute_potential() (potential.c:94-96, 177-179, 271) ends the run with a clear message
 * instead of a link error. */
static void no_potentials(void)
{
  printf("ngravs-hip: gravitational potentials are not provided by libngravs_hip (compute_potential(): set the energy statistics off)\n");
  endrun(1070);
}
void force_treeevaluate_potential(int target, int mode)
{
  (void)target;
  (void)mode;
  no_potentials();
}
#ifdef PMGRID
void force_treeevaluate_potential_shortrange(int target, int mode)
{
  (void)target;
  (void)mode;
  no_potentials();
}
#ifdef PERIODIC
void pmpotential_periodic(void)
{
  no_potentials();
}
#endif
#endif

/* gravtree.c:384-456: the per-step block of timings.txt and the CPU_* sums it feeds, line for line in the reference's format.
 * What the numbers mean here: Nf / total-Nf as there; ex-frac -- there the exported targets per force computation -- is the
 * imported particles (copies of other tasks' top leaves in this task's tree) per force computation; iter is 1 (the import is
 * decided before the walk, nothing is iterated); work-load balance from the tasks' device times of the walk; particle-load
 * balance max(NumPart) * NTask / TotNumPart; max. nodes of the device trees; part/sec | ia/part as there; the count in brackets
 * (node-level Ewald corrections per force) is 0: the lattice correction is walked on the force walk's own interaction list. */
static void write_timings(const ngravs_stats_t *st)
{
  double mine[7], *all = NULL, ntot = 0, sumt = 0, maxt = 0, sumcomm = 0, costtotal = 0, plb_max = 0, nimp = 0, sumimb = 0;
  long long nt;
  int i, maxnumnodes = 0;
#ifdef NGRAVS_WITH_RCCL
  {
    int64_t calls;
    double sec = 0, bytes;
    ngravs_rccl_stats(Rccl, &calls, &sec, &bytes, 1);
    CommSeconds += sec;
  }
#endif
  mine[0] = (double)st->n_active;
  mine[1] = st->interactions;
  mine[2] = st->t_treewalk;
  mine[3] = CommSeconds;	/* since the last block: decomposition, mesh exchanges and imports of this step */
  mine[4] = (double)st->n_nodes;
  mine[5] = (double)NumImported;
  mine[6] = (double)NumPart;
  CommSeconds = 0;
  all = malloc(sizeof(double) * 7 * NTask);
  MPI_Allgather(mine, 7, MPI_DOUBLE, all, 7, MPI_DOUBLE, MPI_COMM_WORLD);
  for(i = 0; i < NTask; i++)
    {
      const double *a = all + 7 * i;
      ntot += a[0];
      costtotal += a[1];
      sumt += a[2];
      maxt = a[2] > maxt ? a[2] : maxt;
      sumcomm += a[3];
      maxnumnodes = (int)a[4] > maxnumnodes ? (int)a[4] : maxnumnodes;
      nimp += a[5];
      plb_max = a[6] * NTask / (double)All.TotNumPart > plb_max ? a[6] * NTask / (double)All.TotNumPart : plb_max;
    }
  for(i = 0; i < NTask; i++)
    sumimb += maxt - all[7 * i + 2];
  nt = (long long)ntot;
  All.NumForcesSinceLastDomainDecomp += nt;	/* gravtree.c:74-78 */
  All.CPU_TreeConstruction += st->t_treebuild;
  if(ThisTask == 0)
    {
      All.TotNumOfForces += nt;
      fprintf(FdTimings, "Step= %d  t= %g  dt= %g \n", All.NumCurrentTiStep, All.Time, All.TimeStep);
      fprintf(FdTimings, "Nf= %d%09d  total-Nf= %d%09d  ex-frac= %g  iter= %d\n", (int)(nt / 1000000000), (int)(nt % 1000000000),
	      (int)(All.TotNumOfForces / 1000000000), (int)(All.TotNumOfForces % 1000000000), nimp / (ntot + 1.0e-20), 1);
      fprintf(FdTimings, "work-load balance: %g  max=%g avg=%g PE0=%g\n", maxt / (sumt / NTask + 1.0e-20), maxt, sumt / NTask, all[2]);
      fprintf(FdTimings, "particle-load balance: %g\n", plb_max);
      fprintf(FdTimings, "max. nodes: %d, filled: %g\n", maxnumnodes, maxnumnodes / (All.TreeAllocFactor * All.MaxPart + 1.0e-20));
      fprintf(FdTimings, "part/sec=%g | %g  ia/part=%g (%g)\n", ntot / (sumt + 1.0e-20), ntot / (maxt * NTask + 1.0e-20),
	      costtotal / (ntot + 1.0e-20), 0.0);
      fprintf(FdTimings, "\n");
      fflush(FdTimings);
      All.CPU_TreeWalk += sumt / NTask;
      All.CPU_Imbalance += sumimb / NTask;
      All.CPU_CommSum += sumcomm / NTask;
    }
  free(all);
}

/* proto.h:114 */
void gravity_tree(void)
{
  ngravs_stats_t st;
  int64_t unopened = 0;
  ensure_ctx();
  if(All.ComovingIntegrationOn)	/* gravtree.c:50-51: new softening lengths for the new scale factor */
    set_softenings();
  /* The first step calls gravity_tree() twice (accel.c:44-52): with the Barnes-Hut criterion and OldAcc = 0, then with the relative
   * criterion and the OldAcc the first call wrote.  The reference's export / import follows whatever the walk opens; here the
   * top leaves a task imports were chosen at the decomposition, for the first criterion.  Under the second one a walk can open
   * leaves that were not asked for (their mass would be missing from the force; the library refuses such a walk in its reference
   * mode and counts the leaves in the production mode): the import is decided again, for the criterion this call walks with. */
  if(NTask > 1 && DdUseTheta >= 0 && DdUseTheta != (All.ErrTolTheta != 0))
    decompose_several_tasks(1);
  must(ngravs_set_opening(Ctx, All.ErrTolTheta, All.ErrTolForceAcc), 1062);
  /* the walk reads P[].OldAcc as it is NOW (gravtree.c:334-335): the first step calls gravity_tree() twice (accel.c:44-52), the
   * second time with the OldAcc the first call has just written -- the hand-over of the decomposition does not have it yet */
  if(NumPart > 0)
    must(ngravs_set_old_acc(Ctx, &P[0].OldAcc, sizeof(struct particle_data), 0), 1074);
  must(ngravs_gravity_tree(Ctx), 1063);
  if(KeptStep)
    {
      /* the imported leaves were chosen at the decomposition; the particles have moved since.  A walk of ANY task that wanted a
       * leaf it does not hold (the library walked it as a monopole and counted it) sends all tasks through a new decomposition
       * and this walk again -- the reference's export follows whatever the walk opens, so it has no such case. */
      int64_t want = 0;
      long long mine, any = 0;
      (void)ngravs_walk_unopened(Ctx, &want);
      mine = (long long)want;
      MPI_Allreduce(&mine, &any, 1, MPI_LONG_LONG, MPI_MAX, MPI_COMM_WORLD);
#ifdef NGRAVS_GLUE_TEST_KEPT_FALLBACK
      any = 1;			/* tests: take the way out below although nothing was missing */
#endif
      KeptStep = 0;
      if(any > 0)
	{
	  if(ThisTask == 0)
	    printf("ngravs-hip: a kept decomposition no longer holds the leaves the walk opens: decomposing again\n");
#ifdef PERIODIC
	  do_box_wrapping();	/* as every domain_Decomposition() starts (domain.c:81): the particles have drifted since the last one */
#endif
	  decompose_several_tasks(1);
	  All.NumForcesSinceLastDomainDecomp = 0;
	  if(NumPart > 0)
	    must(ngravs_set_old_acc(Ctx, &P[0].OldAcc, sizeof(struct particle_data), 0), 1074);
	  must(ngravs_gravity_tree(Ctx), 1063);
	}
    }
  /* only particles with Ti_endstep == Ti_Current are written (gravtree.c:318-341); inactive rows of P[] keep their values */
  must(ngravs_get_accel(Ctx, &P[0].GravAccel[0], sizeof(struct particle_data), NULL, 0, &P[0].OldAcc,
			sizeof(struct particle_data), &P[0].GravCost, sizeof(struct particle_data), 0, 1), 1064);
  TreeReconstructFlag = 0;
  if(All.TypeOfOpeningCriterion == 1)
    All.ErrTolTheta = 0;	/* gravtree.c:334-335 */
  ngravs_get_stats(Ctx, &st);
  write_timings(&st);
  (void)ngravs_walk_unopened(Ctx, &unopened);
  if(unopened > 0)
    printf("ngravs-hip: task %d: %ld top leaves were used as monopoles where a group of targets wanted them opened (not imported)\n",
	   ThisTask, (long)unopened);
}

#ifdef PMGRID
/* proto.h:155, :161 */
void pm_init_periodic(void)
{
  All.Asmth[0] = ASMTH * All.BoxSize / PMGRID;
  All.Rcut[0] = RCUT * All.Asmth[0];
}
void pmforce_periodic(void)
{
  ngravs_stats_t st;
  ensure_ctx();
  if(NTask == 1)
    must(ngravs_pmforce_periodic(Ctx), 1065);
  else
    must(ngravs_host_pmforce_periodic(Ctx, &Comm), 1066);	/* x-slab decomposed mesh, four exchanges */
  /* GravPM of all NumPart own particles (pm_periodic.c:716-763); the library delivers own rows only -- the imported copies of a
   * multi-task working set never reach P[] (ABI 3) */
  must(ngravs_get_accel(Ctx, NULL, 0, &P[0].GravPM[0], sizeof(struct particle_data), NULL, 0, NULL, 0, 0, 0), 1067);
  ngravs_get_stats(Ctx, &st);
  All.CPU_PM += st.t_pm;
  All.NumForcesSinceLastDomainDecomp = 1 + All.TotNumPart * All.TreeDomainUpdateFrequency;	/* pm_periodic.c:783 */
}
#endif

#ifdef FORCETEST
/* proto.h:113 -- gravtree_forcetest.c:28-356: direct sums for a random FORCETEST fraction of the active particles, one line
 * per tested particle in forcetest.txt.  One task: ngravs_direct_sum() over the library's particles.  Several tasks
 * (gravtree_forcetest.c:100-260 exports the test particles to every task and imports the partial sums): the test particles of
 * all tasks are all-gathered, every task sums over the particles it OWNS (ngravs_direct_sum_targets), and the partial sums
 * are added up; the tasks append their lines to the file one after the other (:283-315). */
void gravity_forcetest(void)
{
  int i, k, nt = 0, *idx, nthis;
  double *acc = NULL;
  char buf[200];
#ifdef PMGRID
  if(All.PM_Ti_endstep != All.Ti_Current)
    return;
#endif
  idx = malloc(sizeof(int) * (NumPart > 0 ? NumPart : 1));
  for(i = 0; i < NumPart; i++)
    if(P[i].Ti_endstep == All.Ti_Current && get_random_number(P[i].ID) < FORCETEST)	/* :54-63 */
      idx[nt++] = i;
  if(NTask == 1)
    {
      if(nt > 0)
	{
	  acc = malloc(sizeof(double) * 3 * nt);
	  must(ngravs_direct_sum(Ctx, idx, nt, acc), 1068);
	}
    }
  else
    {
      int *cnt = malloc(sizeof(int) * 4 * NTask), *off = cnt + NTask, *cnt3 = off + NTask, *off3 = cnt3 + NTask, tot = 0;
      double *mypos = malloc(sizeof(double) * 4 * (nt > 0 ? nt : 1)), *allpos, *allacc;
      int *mytype = malloc(sizeof(int) * (nt > 0 ? nt : 1)), *alltype;
      MPI_Allgather(&nt, 1, MPI_INT, cnt, 1, MPI_INT, MPI_COMM_WORLD);
      for(i = 0; i < NTask; i++)
	{
	  off[i] = tot;
	  off3[i] = 4 * tot;
	  cnt3[i] = 4 * cnt[i];
	  tot += cnt[i];
	}
      for(k = 0; k < nt; k++)
	{
	  for(i = 0; i < 3; i++)
	    mypos[4 * k + i] = P[idx[k]].Pos[i];
	  mypos[4 * k + 3] = P[idx[k]].Mass;
	  mytype[k] = P[idx[k]].Type;
	}
      allpos = malloc(sizeof(double) * 4 * (tot > 0 ? tot : 1));
      alltype = malloc(sizeof(int) * (tot > 0 ? tot : 1));
      allacc = malloc(sizeof(double) * 3 * (tot > 0 ? tot : 1));
      MPI_Allgatherv(mypos, 4 * nt, MPI_DOUBLE, allpos, cnt3, off3, MPI_DOUBLE, MPI_COMM_WORLD);
      MPI_Allgatherv(mytype, nt, MPI_INT, alltype, cnt, off, MPI_INT, MPI_COMM_WORLD);
      if(tot > 0)
	{
	  double *p3 = malloc(sizeof(double) * 3 * tot), *m1 = malloc(sizeof(double) * tot);
	  for(k = 0; k < tot; k++)
	    {
	      for(i = 0; i < 3; i++)
		p3[3 * k + i] = allpos[4 * k + i];
	      m1[k] = allpos[4 * k + 3];
	    }
	  must(ngravs_direct_sum_targets(Ctx, p3, m1, alltype, tot, allacc), 1068);
	  MPI_Allreduce(MPI_IN_PLACE, allacc, 3 * tot, MPI_DOUBLE, MPI_SUM, MPI_COMM_WORLD);
	  free(p3);
	  free(m1);
	}
      if(nt > 0)
	{
	  acc = malloc(sizeof(double) * 3 * nt);
	  memcpy(acc, allacc + 3 * off[ThisTask], sizeof(double) * 3 * nt);
	}
      free(cnt);
      free(mypos);
      free(mytype);
      free(allpos);
      free(alltype);
      free(allacc);
    }
  for(k = 0; k < nt; k++)
    for(i = 0; i < 3; i++)
      P[idx[k]].GravAccelDirect[i] = acc[3 * k + i];
  free(acc);
  for(nthis = 0; nthis < NTask; nthis++)
    {
      if(nthis == ThisTask && nt > 0)
	{
	  sprintf(buf, "%s%s", All.OutputDir, "forcetest.txt");
	  if(!(FdForceTest = fopen(buf, "a")))
	    {
	      printf("error in opening file '%s'\n", buf);
	      endrun(17);
	    }
	  for(k = 0; k < nt; k++)
	    {
	      i = idx[k];
#ifndef PMGRID
	      fprintf(FdForceTest, "%d %g %g %g %g %g %g %g %g %g %g %g %d\n", P[i].Type, All.Time,
		      All.Time - TimeOfLastTreeConstruction, P[i].Pos[0], P[i].Pos[1], P[i].Pos[2], P[i].GravAccelDirect[0],
		      P[i].GravAccelDirect[1], P[i].GravAccelDirect[2], P[i].GravAccel[0], P[i].GravAccel[1], P[i].GravAccel[2],
		      (int)P[i].ID);	/* :297-303 */
#else
	      fprintf(FdForceTest, "%d %f %f %f %f %f %.15e %.15e %.15e %.15e %.15e %.15e %.15e %.15e %.15e %d\n", P[i].Type,
		      All.Time, All.Time - TimeOfLastTreeConstruction, P[i].Pos[0], P[i].Pos[1], P[i].Pos[2],
		      P[i].GravAccelDirect[0], P[i].GravAccelDirect[1], P[i].GravAccelDirect[2], P[i].GravAccel[0],
		      P[i].GravAccel[1], P[i].GravAccel[2], P[i].GravPM[0] + P[i].GravAccel[0], P[i].GravPM[1] + P[i].GravAccel[1],
		      P[i].GravPM[2] + P[i].GravAccel[2], (int)P[i].ID);	/* :305-311 */
#endif
	    }
	  fclose(FdForceTest);
	}
      if(NTask > 1)
	MPI_Barrier(MPI_COMM_WORLD);
    }
  free(idx);
}
#endif

/* proto.h:149, :150 */
peanokey peano_hilbert_key(int x, int y, int z, int bits)
{
  return ngravs_peano_hilbert_key(x, y, z, bits);
}
void peano_hilbert_order(void)
{
  /* The device keeps its own Peano order; P[] stays in the host's order.  The reference sorts P[] by (species, key) so that
   * pm_periodic.c:251-254 can address the particles of one species as a contiguous block through NgravLocal[]; that reader is
   * gone with pm_periodic.o (the device PM selects species by TypeToGrav[P[].Type]), and count_types() above still maintains
   * NgravLocal[] / NtypeLocal[] / Ntype[] for the kept units that print them. */
}

#endif /* NGRAVS_BUILD_INSIDE_REFERENCE */
