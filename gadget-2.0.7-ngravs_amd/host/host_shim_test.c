/* host_shim_test.c -- a plain-C host driving libngravs_hip.so exactly as gadget_glue.c does: an AoS
 * P[] with the reference's struct particle_data layout (offsets measured in SURVEY.md 8(a'):
 * Pos@0 Mass@24 Vel@32 GravAccel@56 GravPM@80 Potential@104 OldAcc@112 ID@120 Type@124 Ti_endstep@128
 * Ti_begstep@132 GravCost@136, size 144), handed over with byte strides.  Part 1 (one task): the tree force against an
 * O(N^2) direct sum done here in C, only active rows of P[] written, a kept (refit) tree.  Part 2 (two tasks): two threads
 * = two "MPI ranks" with one context each on the same GPU, joined by a shared-memory communicator; they run exactly the
 * calls gadget_glue.c makes for NTask > 1 -- ngravs_host_domain_owners, ngravs_dd_get_dest, a host-side exchange of whole
 * particle_data records, ngravs_set_particles, ngravs_host_domain_halo, ngravs_host_pmforce_periodic, ngravs_gravity_tree,
 * ngravs_get_accel -- and the merged forces must reproduce the single-task run.
 * Built and run by tests/test_host_glue.py (-m gpu).
 *   gcc -O2 host_shim_test.c -I../../include -L.. -lngravs_hip -lm -lpthread -Wl,-rpath,$PWD/.. -o host_shim_test
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <pthread.h>
#include "ngravs_hip.h"
#include "ngravs_host.h"

struct particle_data
{
  double Pos[3], Mass, Vel[3], GravAccel[3], GravPM[3], Potential, OldAcc;
  unsigned int ID;
  int Type, Ti_endstep, Ti_begstep;
  float GravCost;
};

static double plummer(double m, double h, double r)
{
  double hi = 1 / h, u = r * hi;
  if(u < 0.5)
    return m * hi * hi * hi * (10.666666666667 + u * u * (32.0 * u - 38.4));
  return m * hi * hi * hi * (21.333333333333 - 48.0 * u + 38.4 * u * u - 10.666666666667 * u * u * u - 0.066666666667 / (u * u * u));
}

static int fatal_seen = 0; /* set if the library reports a fatal condition */
static void on_fatal(int code, const char *msg)
{
  fprintf(stderr, "endrun(%d): %s\n", code, msg);
  fatal_seen = code;
}


/* =====================================================================================================================
 *  Part 2: two tasks.  A shared-memory communicator for threads (the stand-in for MPI_COMM_WORLD).
 * ===================================================================================================================== */
#define NT 2
typedef struct
{
  pthread_barrier_t bar;
  const void *ptr[NT];
  const int64_t *cnt[NT], *dsp[NT];
} shm_world;
typedef struct
{
  shm_world *w;
  int rank;
} shm_rank;

static int shm_allreduce(void *user, void *buf, int64_t count, int dtype, int op)
{
  shm_rank *u = user;
  int64_t i;
  int r;
  void *tmp = malloc((size_t)(8 * (count > 0 ? count : 1)));
  u->w->ptr[u->rank] = buf;
  pthread_barrier_wait(&u->w->bar);
  for(i = 0; i < count; i++)
    if(dtype == NGRAVS_T_F64)
      {
        double v = ((const double *)u->w->ptr[0])[i];
        for(r = 1; r < NT; r++)
          {
            double x = ((const double *)u->w->ptr[r])[i];
            v = op == NGRAVS_OP_SUM ? v + x : (op == NGRAVS_OP_MIN ? (x < v ? x : v) : (x > v ? x : v));
          }
        ((double *)tmp)[i] = v;
      }
    else
      {
        int64_t v = ((const int64_t *)u->w->ptr[0])[i];
        for(r = 1; r < NT; r++)
          {
            int64_t x = ((const int64_t *)u->w->ptr[r])[i];
            v = op == NGRAVS_OP_SUM ? v + x : (op == NGRAVS_OP_MIN ? (x < v ? x : v) : (x > v ? x : v));
          }
        ((int64_t *)tmp)[i] = v;
      }
  pthread_barrier_wait(&u->w->bar);
  memcpy(buf, tmp, (size_t)(8 * count));
  free(tmp);
  pthread_barrier_wait(&u->w->bar);
  return 0;
}
static int shm_allgather(void *user, const void *send, void *recv, int64_t bytes)
{
  shm_rank *u = user;
  int r;
  u->w->ptr[u->rank] = send;
  pthread_barrier_wait(&u->w->bar);
  for(r = 0; r < NT; r++)
    memcpy((char *)recv + r * bytes, u->w->ptr[r], (size_t)bytes);
  pthread_barrier_wait(&u->w->bar);
  return 0;
}
static int shm_alltoallv(void *user, const void *send, const int64_t *sb, const int64_t *sd, void *recv, const int64_t *rb, const int64_t *rd)
{
  shm_rank *u = user;
  int s, bad = 0;
  u->w->ptr[u->rank] = send;
  u->w->cnt[u->rank] = sb;
  u->w->dsp[u->rank] = sd;
  pthread_barrier_wait(&u->w->bar);
  for(s = 0; s < NT; s++)
    {
      if(u->w->cnt[s][u->rank] != rb[s])
        bad = 1;   /* the two sides disagree about a block size */
      else
        memcpy((char *)recv + rd[s], (const char *)u->w->ptr[s] + u->w->dsp[s][u->rank], (size_t)rb[s]);
    }
  pthread_barrier_wait(&u->w->bar);
  return bad;
}

#define N2 24000
#define PMG 32
static struct particle_data *Pall;          /* the whole box: the single-task reference and the initial scatter */
static double Acc1[N2][3], Pm1[N2][3];      /* single-task results by ID */
static double Acc2[N2][3], Pm2[N2][3];      /* two-task results by ID    */
static int Seen2[N2];
static shm_world World;

static void box_config(ngravs_config_t *cfg)
{
  int k;
  ngravs_config_default(cfg);
  cfg->n_gravs = 2;
  cfg->periodic = 1;
  cfg->pmgrid = PMG;
  cfg->box_size = 1.0;
  cfg->G = 1.0;
  cfg->err_tol_theta = 0.5;
  for(k = 0; k < 6; k++)
    cfg->force_softening[k] = 2.8 / (40.0 * cbrt((double)N2));
  cfg->type_to_grav[2] = 1;
  cfg->law_accel[0][1] = cfg->law_accel[1][0] = NGRAVS_LAW_COLOYUK;   /* Newton on the diagonal, Newton+Yukawa between species */
  cfg->law_greens[0][1] = cfg->law_greens[1][0] = cfg->law_normed[0][1] = cfg->law_normed[1][0] = NGRAVS_LAW_COLOYUK;
  cfg->walk_mode = NGRAVS_WALK_STRICT;   /* the reference's per-target walk: its forces must not depend on the number of tasks */
}

static void hand_over(ngravs_ctx *ctx, struct particle_data *P, int n, ngravs_particles_t *pp, int with_pm)
{
  memset(pp, 0, sizeof(*pp));
  pp->n = n;
  pp->pos = &P[0].Pos[0];
  pp->mass = &P[0].Mass;
  pp->type = &P[0].Type;
  pp->old_acc = &P[0].OldAcc;
  pp->grav_cost = &P[0].GravCost;
  pp->pos_stride = pp->mass_stride = pp->type_stride = pp->old_acc_stride = pp->grav_cost_stride = sizeof(struct particle_data);
  if(with_pm)   /* non-PM step: P[].GravPM of the last PM step enters OldAcc (gravtree.c:318-330), as push_particles() in the glue */
    {
      pp->grav_pm = &P[0].GravPM[0];
      pp->grav_pm_stride = sizeof(struct particle_data);
    }
  (void)ctx;
}

/* domain_Decomposition() as in gadget_glue.c for NTask > 1: owners, destinations, host-side exchange of whole particle_data
 * records through the same communicator, the migrated P[] handed over, top-leaf moments + import */
static int glue_decomposition(ngravs_ctx *ctx, ngravs_comm *cm, shm_rank *u, struct particle_data *P, int *np, int with_pm,
                              ngravs_dd_info *info)
{
  const int me = u->rank;
  int i, r, n = *np, nkeep = 0, err = 0, nout[NT] = {0, 0};
  struct particle_data *out[NT];
  ngravs_particles_t pp;
  ngravs_dd_plan plan;
  int32_t *dest;
  hand_over(ctx, P, n, &pp, with_pm);
  err |= ngravs_set_particles(ctx, &pp);
  err |= ngravs_host_domain_owners(ctx, cm, 0, 1.5, &plan, info);
  dest = malloc(sizeof(int32_t) * (n > 0 ? n : 1));
  err |= ngravs_dd_get_dest(ctx, plan.leaf_owner, dest);
  if(err)
    return 21;
  for(r = 0; r < NT; r++)
    out[r] = malloc(sizeof(*P) * (n > 0 ? n : 1));
  for(i = 0; i < n; i++)
    if(dest[i] == me)
      P[nkeep++] = P[i];
    else
      out[dest[i]][nout[dest[i]]++] = P[i];
  {
    int64_t sb[NT], sd[NT], rb[NT], rd[NT], cnt[NT], mat[NT * NT], tot = 0;
    struct particle_data *sendbuf = malloc(sizeof(*P) * (n > 0 ? n : 1));
    for(r = 0; r < NT; r++)
      {
        cnt[r] = nout[r];
        sd[r] = tot * (int64_t)sizeof(*P);
        sb[r] = nout[r] * (int64_t)sizeof(*P);
        memcpy(sendbuf + tot, out[r], sizeof(*P) * nout[r]);
        tot += nout[r];
      }
    shm_allgather(u, cnt, mat, sizeof(cnt));
    tot = 0;
    for(r = 0; r < NT; r++)
      {
        rb[r] = mat[r * NT + me] * (int64_t)sizeof(*P);
        rd[r] = tot * (int64_t)sizeof(*P);
        tot += mat[r * NT + me];
      }
    if(shm_alltoallv(u, sendbuf, sb, sd, P + nkeep, rb, rd))
      return 22;
    n = nkeep + (int)tot;
    free(sendbuf);
  }
  hand_over(ctx, P, n, &pp, with_pm);
  err |= ngravs_set_particles(ctx, &pp);            /* the migrated P[]: its order is the order of the results */
  err |= ngravs_host_domain_halo(ctx, cm, &plan, info);
  ngravs_host_plan_free(&plan);
  for(r = 0; r < NT; r++)
    free(out[r]);
  free(dest);
  *np = n;
  return err ? 23 : 0;
}

static double Old1b[N2], Old2b[N2];         /* OldAcc of the following non-PM step, single task / two tasks, by ID */

static void *task_main(void *arg)
{
  shm_rank *u = arg;
  const int me = u->rank;
  int i, r, n = 0, cap = N2, err = 0;
  struct particle_data *P = malloc(sizeof(*P) * (cap + 1));   /* + one canary row */
  ngravs_config_t cfg;
  ngravs_ctx *ctx = NULL;
  ngravs_comm cm;
  ngravs_dd_info info;
  intptr_t rc = 0;
  for(i = me; i < N2; i += NT)   /* an arbitrary initial distribution: every task has particles everywhere */
    P[n++] = Pall[i];
  box_config(&cfg);
  if(ngravs_create(&cfg, &ctx))
    return (void *)(intptr_t)20;
  memset(&cm, 0, sizeof(cm));
  cm.rank = me;
  cm.size = NT;
  cm.device_buffers = 0;   /* like a plain MPI: exchange buffers are staged through host memory */
  cm.user = u;
  cm.allreduce = shm_allreduce;
  cm.allgather = shm_allgather;
  cm.alltoallv = shm_alltoallv;
  /* ---- step 1 (a PM step): domain_Decomposition(), long_range_force(), gravity_tree() ---- */
  if((err = glue_decomposition(ctx, &cm, u, P, &n, 0, &info)))
    return (void *)(intptr_t)err;
  err |= ngravs_host_pmforce_periodic(ctx, &cm);
  err |= ngravs_gravity_tree(ctx);
  if(err)
    return (void *)(intptr_t)23;
  {
    /* results straight into P[], which holds exactly NumPart = n rows (the library's working set is own + imported rows; it
     * must deliver the own rows only): the row behind the last one is a canary */
    struct particle_data canary;
    memset(&P[n], 0x5a, sizeof(P[n]));
    canary = P[n];
    if(info.n_local != n || info.n_halo <= 0 ||
       ngravs_get_accel(ctx, &P[0].GravAccel[0], sizeof(*P), &P[0].GravPM[0], sizeof(*P), &P[0].OldAcc, sizeof(*P), &P[0].GravCost,
                        sizeof(*P), 0, 0))
      return (void *)(intptr_t)24;
    if(memcmp(&canary, &P[n], sizeof(canary)))
      return (void *)(intptr_t)25;   /* wrote past NumPart rows */
    for(i = 0; i < n; i++)
      {
        const unsigned id = P[i].ID;
        for(r = 0; r < 3; r++)
          {
            Acc2[id][r] = P[i].GravAccel[r];
            Pm2[id][r] = P[i].GravPM[r];
          }
        __sync_fetch_and_add(&Seen2[id], 1);
      }
  }
  if(me == 0)
    printf("two tasks: top tree of %d nodes / %d leaves (%d counting rounds), work balance %.3f, memory balance %.3f; task 0 holds %ld own + %ld imported particles\n",
           info.n_topnodes, info.n_topleaves, info.toptree_rounds, info.work_balance, info.memory_balance, (long)info.n_local, (long)info.n_halo);
  /* ---- step 2 (no PM): P[].GravPM of step 1 is handed over with the particles and must enter OldAcc although the task's
   * working set holds imported rows (gravtree.c:318-330) ---- */
  if((err = glue_decomposition(ctx, &cm, u, P, &n, 1, &info)))
    return (void *)(intptr_t)(30 + err);
  if(ngravs_gravity_tree(ctx) || ngravs_get_accel(ctx, NULL, 0, NULL, 0, &P[0].OldAcc, sizeof(*P), NULL, 0, 0, 0))
    return (void *)(intptr_t)26;
  for(i = 0; i < n; i++)
    Old2b[P[i].ID] = P[i].OldAcc;
  free(P);
  ngravs_destroy(ctx);
  return (void *)rc;
}

static int two_tasks(void)
{
  int i, k, r, bad = 0;
  ngravs_config_t cfg;
  ngravs_ctx *ctx = NULL;
  ngravs_particles_t pp;
  pthread_t th[NT];
  shm_rank ranks[NT];
  double pmax = 0, dpm = 0, *a, *pm;
  Pall = calloc(N2, sizeof(*Pall));
  srand(4711);
  for(i = 0; i < N2; i++)
    {
      for(k = 0; k < 3; k++)
        Pall[i].Pos[k] = (float)(rand() / (RAND_MAX + 1.0));
      Pall[i].Mass = 1.0 / N2;
      Pall[i].Type = 1 + (i & 1);
      Pall[i].ID = i;
    }
  /* single task */
  box_config(&cfg);
  if(ngravs_create(&cfg, &ctx))
    return 10;
  hand_over(ctx, Pall, N2, &pp, 0);
  a = malloc(sizeof(double) * 3 * N2);
  pm = malloc(sizeof(double) * 3 * N2);
  if(ngravs_set_particles(ctx, &pp) || ngravs_compute_accelerations(ctx, 1) || ngravs_get_accel(ctx, a, 24, pm, 24, NULL, 0, NULL, 0, 0, 0))
    return 11;
  memcpy(Acc1, a, sizeof(Acc1));
  memcpy(Pm1, pm, sizeof(Pm1));
  /* the following non-PM step: OldAcc = |GravAccel + GravPM/G| (gravtree.c:318-330) */
  if(ngravs_get_accel(ctx, NULL, 0, NULL, 0, Old1b, 8, NULL, 0, 0, 0) || ngravs_set_old_acc(ctx, Old1b, 8, 0) ||
     ngravs_compute_accelerations(ctx, 0) || ngravs_get_accel(ctx, NULL, 0, NULL, 0, Old1b, 8, NULL, 0, 0, 0))
    return 14;
  ngravs_destroy(ctx);
  /* two tasks */
  pthread_barrier_init(&World.bar, NULL, NT);
  for(r = 0; r < NT; r++)
    {
      ranks[r].w = &World;
      ranks[r].rank = r;
      pthread_create(&th[r], NULL, task_main, &ranks[r]);
    }
  for(r = 0; r < NT; r++)
    {
      void *ret = NULL;
      pthread_join(th[r], &ret);
      if(ret)
        {
          fprintf(stderr, "task %d failed at step %ld\n", r, (long)(intptr_t)ret);
          bad++;
        }
    }
  if(bad)
    return 12;
  {
    /* every particle owned exactly once; GravPM of the slab-decomposed mesh == single mesh; with the global top of the tree and
     * the imported top cells the tree force is the single-task force to summation-order noise (domain.c:18-21) */
    double worst = 0, sum = 0;
    int nbig = 0;
    for(i = 0; i < N2; i++)
      {
        double tot2 = 0, d2 = 0, a2 = 0;
        if(Seen2[i] != 1)
          bad++;
        for(k = 0; k < 3; k++)
          {
            pmax = fmax(pmax, fabs(Pm1[i][k]));
            dpm = fmax(dpm, fabs(Pm2[i][k] - Pm1[i][k]));
            tot2 += (Acc1[i][k] + Pm1[i][k]) * (Acc1[i][k] + Pm1[i][k]);
            d2 += (Acc2[i][k] - Acc1[i][k]) * (Acc2[i][k] - Acc1[i][k]);
            a2 += Acc1[i][k] * Acc1[i][k];
          }
        (void)tot2;
        const double e = sqrt(d2 / a2);
        sum += e;
        worst = fmax(worst, e);
        nbig += e > 1e-10;
      }
    printf("two tasks vs one: GravPM max diff %.2e of max; tree force |da|/|a|: mean %.2e worst %.2e, %d of %d above 1e-10; bad=%d\n",
           dpm / pmax, sum / N2, worst, nbig, N2, bad);
    if(bad || dpm / pmax > 1e-10 || nbig > 0)
      return 13;
    /* the non-PM step after it: same OldAcc as the single task's, i.e. the imported rows did not cost the own rows their GravPM */
    worst = 0;
    for(i = 0; i < N2; i++)
      worst = fmax(worst, fabs(Old2b[i] - Old1b[i]) / Old1b[i]);
    printf("non-PM step after it: OldAcc two tasks vs one, worst relative difference %.2e\n", worst);
    if(!(worst < 1e-10))
      return 15;
  }
  free(a);
  free(pm);
  free(Pall);
  return 0;
}

#ifdef WITH_RCCL
#include "ngravs_comm_rccl.h"
/* =====================================================================================================================
 *  Part 3: one task over RCCL, in C.  The communicator of include/ngravs_comm_rccl.h (host/ngravs_comm_rccl.c) with world size 1 --
 *  one GPU on the box means one task; RCCL refuses two ranks of a communicator on one device -- carries EVERY collective of a
 *  step: the two all-reduces of the decomposition (the per-leaf sums in place on the device), the count/request all-gather, the
 *  import all-to-all-v, the bounding-box all-gather and the four plane exchanges of the slab PM, all on the library's device
 *  buffers.  Forces and GravPM must be the single-task engine's.
 * ===================================================================================================================== */
static int rccl_one_task(void)
{
  char id[NGRAVS_RCCL_ID_BYTES];
  ngravs_rccl *r = NULL;
  ngravs_comm cm;
  ngravs_config_t cfg;
  ngravs_ctx *ctx = NULL;
  ngravs_particles_t pp;
  ngravs_dd_info info;
  int64_t calls = 0;
  double secs = 0, bytes = 0, worst = 0, dpm = 0, pmax = 0;
  double *a = malloc(sizeof(double) * 3 * N2), *pm = malloc(sizeof(double) * 3 * N2);
  int i, k;
  if(ngravs_rccl_unique_id(id) || ngravs_rccl_create(id, 0, 1, 0, &r))
    return 40;
  if(ngravs_rccl_selftest(r))
    {
      printf("RCCL self test: %s\n", ngravs_rccl_last_error(r));
      return 48;
    }
  ngravs_rccl_fill(r, &cm);
  if(ngravs_rccl_world(r) != 1 || !cm.device_buffers || !cm.allreduce_dev)
    return 41;
  box_config(&cfg);
  if(ngravs_create(&cfg, &ctx))
    return 42;
  Pall = calloc(N2, sizeof(*Pall));
  srand(4711);   /* the particles of part 2: Acc1 / Pm1 hold the single-task answer */
  for(i = 0; i < N2; i++)
    {
      for(k = 0; k < 3; k++)
        Pall[i].Pos[k] = (float)(rand() / (RAND_MAX + 1.0));
      Pall[i].Mass = 1.0 / N2;
      Pall[i].Type = 1 + (i & 1);
      Pall[i].ID = i;
    }
  hand_over(ctx, Pall, N2, &pp, 0);
  if(ngravs_set_particles(ctx, &pp) || ngravs_host_compute_accelerations(ctx, &cm, 1, &info) ||
     ngravs_get_accel(ctx, a, 24, pm, 24, NULL, 0, NULL, 0, 0, 0))
    {
      fprintf(stderr, "rccl task: %s / %s\n", ngravs_last_error(ctx), ngravs_rccl_last_error(r));
      return 43;
    }
  ngravs_rccl_stats(r, &calls, &secs, &bytes, 0);
  for(i = 0; i < N2; i++)
    {
      double d2 = 0, a2 = 0;
      for(k = 0; k < 3; k++)
        {
          d2 += (a[3 * i + k] - Acc1[i][k]) * (a[3 * i + k] - Acc1[i][k]);
          a2 += Acc1[i][k] * Acc1[i][k];
          pmax = fmax(pmax, fabs(Pm1[i][k]));
          dpm = fmax(dpm, fabs(pm[3 * i + k] - Pm1[i][k]));
        }
      worst = fmax(worst, sqrt(d2 / a2));
    }
  printf("one task over RCCL (C): %ld collectives, %.2f ms inside them, %.0f bytes; %d in the decomposition (%d counting rounds); "
         "tree force worst |da|/|a| %.2e, GravPM max diff %.2e of max\n", (long)calls, 1e3 * secs, bytes, info.collectives,
         info.toptree_rounds, worst, dpm / pmax);
  ngravs_destroy(ctx);
  ngravs_rccl_destroy(r);
  free(a);
  free(pm);
  free(Pall);
  if(!(worst < 1e-10 && dpm / pmax < 1e-10 && calls >= 8))
    return 44;
  return 0;
}
#endif

int main(void)
{
  const int N = 6000, Ti_Current = 7;
  int i, j, k, bad = 0;
  struct particle_data *P = calloc(N, sizeof(*P));
  if(sizeof(struct particle_data) != 144)
    {
      fprintf(stderr, "layout mismatch: %zu\n", sizeof(struct particle_data));
      return 2;
    }
  srand(12345);
  for(i = 0; i < N; i++)
    {
      for(k = 0; k < 3; k++)
        P[i].Pos[k] = (rand() / (double)RAND_MAX - 0.5) * (i % 3 == 0 ? 0.2 : 2.0);
      P[i].Mass = 1.0 / N;
      P[i].Type = 1 + (i & 1);
      P[i].ID = i;
      P[i].Ti_endstep = (i % 5 == 0) ? Ti_Current + 1 : Ti_Current;   /* 20 % inactive */
      P[i].GravCost = 1;
    }
  ngravs_config_t cfg;
  ngravs_config_default(&cfg);
  cfg.n_gravs = 2;
  cfg.G = 43007.1;
  cfg.err_tol_theta = 0.4;
  for(k = 0; k < 6; k++)
    cfg.force_softening[k] = 2.8 * 0.01;
  cfg.type_to_grav[2] = 1;
  cfg.walk_mode = NGRAVS_WALK_GROUP;
  ngravs_ctx *ctx = NULL;
  if(ngravs_create(&cfg, &ctx) != NGRAVS_OK)
    {
      fprintf(stderr, "ngravs_create failed (no GPU?)\n");
      return 3;
    }
  ngravs_set_fatal_handler(ctx, on_fatal);
  /* error convention: walking before any particle is handed over is a state error, no crash */
  if(ngravs_gravity_tree(ctx) != NGRAVS_ERR_STATE)
    bad++;
  unsigned char *active = malloc(N);
  for(i = 0; i < N; i++)
    active[i] = P[i].Ti_endstep == Ti_Current;
  ngravs_particles_t pp;
  memset(&pp, 0, sizeof(pp));
  pp.n = N;
  pp.pos = &P[0].Pos[0];
  pp.mass = &P[0].Mass;
  pp.type = &P[0].Type;
  pp.old_acc = &P[0].OldAcc;
  pp.pos_stride = pp.mass_stride = pp.type_stride = pp.old_acc_stride = sizeof(struct particle_data);
  pp.active = active;
  pp.active_stride = 1;
  if(ngravs_set_particles(ctx, &pp) || ngravs_compute_accelerations(ctx, 0))
    return 4;
  for(i = 0; i < N; i++)   /* sentinels: inactive rows of P[] must not be touched (gravtree.c:318-341) */
    {
      P[i].GravAccel[0] = P[i].GravAccel[1] = P[i].GravAccel[2] = -77.0;
      P[i].OldAcc = -88.0;
      P[i].GravCost = -99.0f;
    }
  if(ngravs_get_accel(ctx, &P[0].GravAccel[0], sizeof(*P), NULL, 0, &P[0].OldAcc, sizeof(*P), &P[0].GravCost, sizeof(*P), 0, 1))
    return 5;
  double worst = 0, sum = 0;
  int nact = 0;
  for(i = 0; i < N; i++)
    {
      if(!active[i])
        {
          if(P[i].GravAccel[0] != -77.0 || P[i].GravAccel[2] != -77.0 || P[i].OldAcc != -88.0 || P[i].GravCost != -99.0f)
            bad++;
          continue;
        }
      double a[3] = {0, 0, 0};
      for(j = 0; j < N; j++)
        {
          double d[3], r2 = 0, h = 2.8 * 0.01, fac;
          for(k = 0; k < 3; k++)
            {
              d[k] = P[j].Pos[k] - P[i].Pos[k];
              r2 += d[k] * d[k];
            }
          double r = sqrt(r2);
          fac = r >= h ? P[j].Mass / r2 / r : plummer(P[j].Mass, h, r);
          for(k = 0; k < 3; k++)
            a[k] += d[k] * fac;
        }
      double e2 = 0, n2 = 0;
      for(k = 0; k < 3; k++)
        {
          a[k] *= cfg.G;
          e2 += (a[k] - P[i].GravAccel[k]) * (a[k] - P[i].GravAccel[k]);
          n2 += a[k] * a[k];
        }
      double e = sqrt(e2 / n2);
      sum += e;
      nact++;
      if(e > worst)
        worst = e;
      if(fabs(P[i].OldAcc * cfg.G - sqrt(P[i].GravAccel[0] * P[i].GravAccel[0] + P[i].GravAccel[1] * P[i].GravAccel[1] +
                                         P[i].GravAccel[2] * P[i].GravAccel[2])) > 1e-9 * sqrt(n2))
        bad++;
    }
  /* a step that keeps the tree (TreeDomainUpdateFrequency > 0, domain.c:76 false): same P[], unchanged positions ->
   * ngravs_update_particles + gravity_tree (which refits the kept tree) must reproduce the forces bit for bit */
  {
    double (*keep)[3] = malloc(sizeof(double[3]) * N);
    for(i = 0; i < N; i++)
      for(k = 0; k < 3; k++)
        keep[i][k] = P[i].GravAccel[k];
    if(ngravs_update_particles(ctx, &pp) || ngravs_gravity_tree(ctx))
      return 6;
    if(ngravs_get_accel(ctx, &P[0].GravAccel[0], sizeof(*P), NULL, 0, NULL, 0, NULL, 0, 0, 1))
      return 7;
    for(i = 0; i < N; i++)
      for(k = 0; k < 3; k++)
        if(keep[i][k] != P[i].GravAccel[k])
          bad++;
    free(keep);
  }
  /* a comoving run changes All.ForceSoftening[] every step (set_softenings(), gravtree.c:50-51, 468-518): the glue forwards it
   * with ngravs_set_softening(); pairs closer than the new length must now feel the spline, nothing else may change */
  {
    double fs[6], worst2 = 0, changed = 0;
    const double h2 = 2.8 * 0.05;
    for(k = 0; k < 6; k++)
      fs[k] = h2;
    if(ngravs_set_softening(ctx, fs) || ngravs_compute_accelerations(ctx, 0) ||
       ngravs_get_accel(ctx, &P[0].Vel[0], sizeof(*P), NULL, 0, NULL, 0, NULL, 0, 0, 0))   /* second force into the Vel slot */
      return 8;
    for(i = 0; i < N; i += 7)
      {
        double a[3] = {0, 0, 0}, e2 = 0, n2 = 0, c2 = 0;
        if(!active[i])
          continue;
        for(j = 0; j < N; j++)
          {
            double d[3], r2 = 0, r, fac;
            for(k = 0; k < 3; k++)
              {
                d[k] = P[j].Pos[k] - P[i].Pos[k];
                r2 += d[k] * d[k];
              }
            r = sqrt(r2);
            fac = r >= h2 ? P[j].Mass / r2 / r : plummer(P[j].Mass, h2, r);
            for(k = 0; k < 3; k++)
              a[k] += d[k] * fac * cfg.G;
          }
        for(k = 0; k < 3; k++)
          {
            e2 += (a[k] - P[i].Vel[k]) * (a[k] - P[i].Vel[k]);
            c2 += (P[i].GravAccel[k] - P[i].Vel[k]) * (P[i].GravAccel[k] - P[i].Vel[k]);
            n2 += a[k] * a[k];
          }
        worst2 = fmax(worst2, sqrt(e2 / n2));
        changed = fmax(changed, sqrt(c2 / n2));
      }
    printf("after ngravs_set_softening(5 x): worst error against the direct sum with the NEW length %.3e; forces moved by up to %.2e\n",
           worst2, changed);
    if(!(worst2 < 0.1 && changed > 0.05))
      bad++;
  }
  ngravs_stats_t st;
  ngravs_get_stats(ctx, &st);
  printf("host_shim_test: N=%d active=%d (engine says %ld) mean err %.3e worst %.3e ia/part %.1f nodes %ld bad=%d\n", N, nact,
         (long)st.n_active, sum / nact, worst, st.interactions / st.n_active, (long)st.n_nodes, bad);
  ngravs_destroy(ctx);
  if(!(bad == 0 && st.n_active == nact && sum / nact < 5e-3 && worst < 0.1))
    return 1;
  {
    int rc = two_tasks();
#ifdef WITH_RCCL
    if(!rc)
      rc = rccl_one_task();
#endif
    return rc;
  }
}
