/* gadget_glue.c -- the reference-side binding: what a Gadget-2.0.7-ngravs maintainer adds to the
 * reference tree so that its own entry points run on libngravs_hip.so.
 *
 * It is compiled INSIDE the reference tree (it includes the reference's allvars.h / proto.h and
 * therefore sees the real struct particle_data, All, TypeToGrav[], AccelFxns[][] ...), replacing
 * gravtree.o, forcetree.o (walks + build), pm_periodic.o, domain.o's key/sort part and peano.o in
 * Makefile.reference's OBJS.  The reference's signatures are kept exactly (proto.h:36,77,78,86,114,
 * 150,161; ngravs.h:86-87), so accel.c, run.c, init.c, timestep.c are untouched.
 *
 * This file is NOT built in this repository (the reference headers need GSL and FFTW-2, which the
 * image lacks); gadget-2.0.7-ngravs_amd/host/host_shim_test.c exercises the same call sequence
 * against a stand-alone copy of the fields used here, and tests/test_host_glue.py runs it.
 *
 * TreeDomainUpdateFrequency: with 0.0 every step re-decomposes and rebuilds.  With a value > 0 the
 * reference keeps decomposition and tree while fewer than TreeDomainUpdateFrequency*TotNumPart forces have been computed
 * (domain.c:76) and drifts/kicks the node moments instead (predict.c:79-91, timestep.c:331-344); here those steps hand the
 * drifted positions over with ngravs_update_particles() and the library refits the nodes (ngravs_force_update_tree),
 * so force_kick_node()/the node drift loop become no-ops (provided below).
 */
#ifdef NGRAVS_BUILD_INSIDE_REFERENCE

#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "allvars.h"
#include "proto.h"
#include "ngravs.h"
#include "ngravs_hip.h"

static ngravs_ctx *Ctx = NULL;

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
  printf("ngravs-hip: force law %p has no device implementation\n", (void *)f);
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
  endrun(1051);
  return -1;
}

static void on_fatal(int code, const char *msg)
{
  printf("ngravs-hip: %s\n", msg);
  endrun(code > 0 ? code : 1052);
}

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
  cfg.walk_mode = NGRAVS_WALK_GROUP;
  cfg.rank = ThisTask;
  cfg.world_size = NTask;
  cfg.device = ThisTask;	/* one rank per GPU of the node */
  if(ngravs_create(&cfg, &Ctx) != NGRAVS_OK)
    endrun(1053);
  ngravs_set_fatal_handler(Ctx, on_fatal);
}

static unsigned char *ActiveFlag;

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
  if(keep_tree)
    ngravs_update_particles(Ctx, &p);
  else
    ngravs_set_particles(Ctx, &p);
}

/* proto.h:36 */
void domain_Decomposition(void)
{
  ensure_ctx();
#ifdef PMGRID
  if(All.PM_Ti_endstep == All.Ti_Current)	/* domain.c:66-73: PM steps always re-decompose (particles get wrapped) */
    All.NumForcesSinceLastDomainDecomp = 1 + All.TotNumPart * All.TreeDomainUpdateFrequency;
#endif
  if(All.NumForcesSinceLastDomainDecomp > All.TotNumPart * All.TreeDomainUpdateFrequency)	/* domain.c:76 */
    {
#ifdef PERIODIC
      do_box_wrapping();
#endif
      push_particles(0);
      ngravs_domain_decomposition(Ctx);
      ngravs_get_domain(Ctx, &DomainCorner[0]);	/* DomainCorner[3],DomainCenter[3],DomainLen,DomainFac are contiguous in allvars.c */
      All.NumForcesSinceLastDomainDecomp = 0;
      TreeReconstructFlag = 1;
    }
  else
    push_particles(1);		/* drifted tree: gravity_tree() refits it (ngravs_force_update_tree) */
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

/* proto.h:114 */
void gravity_tree(void)
{
  ngravs_stats_t st;
  double t0 = second();
  ensure_ctx();
  ngravs_set_opening(Ctx, All.ErrTolTheta, All.ErrTolForceAcc);
  ngravs_gravity_tree(Ctx);
  ngravs_get_accel(Ctx, &P[0].GravAccel[0], sizeof(struct particle_data), NULL, 0, &P[0].OldAcc,
		   sizeof(struct particle_data), &P[0].GravCost, sizeof(struct particle_data), 0);
  TreeReconstructFlag = 0;
  if(All.TypeOfOpeningCriterion == 1)
    All.ErrTolTheta = 0;	/* gravtree.c:334-335 */
  ngravs_get_stats(Ctx, &st);
  All.TotNumOfForces += st.n_active;
  All.CPU_TreeConstruction += st.t_treebuild;
  All.CPU_TreeWalk += st.t_treewalk;
  if(ThisTask == 0)
    fprintf(FdTimings, "Step= %d  t= %g  Nf= %ld  part/sec=%g  ia/part=%g  (MI355X, %g s)\n", All.NumCurrentTiStep,
	    All.Time, (long) st.n_active, st.n_active / (st.t_treewalk + 1e-30), st.interactions / (st.n_active + 1e-30),
	    timediff(t0, second()));
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
  ngravs_pmforce_periodic(Ctx);
  ngravs_get_accel(Ctx, NULL, 0, &P[0].GravPM[0], sizeof(struct particle_data), NULL, 0, NULL, 0, 0);
  ngravs_get_stats(Ctx, &st);
  All.CPU_PM += st.t_pm;
  All.NumForcesSinceLastDomainDecomp = 1 + All.TotNumPart * All.TreeDomainUpdateFrequency;	/* pm_periodic.c:783 */
}
#endif

/* proto.h:149, :150 */
peanokey peano_hilbert_key(int x, int y, int z, int bits)
{
  return ngravs_peano_hilbert_key(x, y, z, bits);
}
void peano_hilbert_order(void)
{
  /* the device keeps its own Peano order; P[] stays in the host's order */
}

#endif /* NGRAVS_BUILD_INSIDE_REFERENCE */
