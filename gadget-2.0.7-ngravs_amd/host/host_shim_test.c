/* host_shim_test.c -- a plain-C host driving libngravs_hip.so exactly as gadget_glue.c does: an AoS
 * P[] with the reference's struct particle_data layout (offsets measured in SURVEY.md 8(a'):
 * Pos@0 Mass@24 Vel@32 GravAccel@56 GravPM@80 Potential@104 OldAcc@112 ID@120 Type@124 Ti_endstep@128
 * Ti_begstep@132 GravCost@136, size 144), handed over with byte strides.  Checks the tree force
 * against an O(N^2) direct sum done here in C.  Built and run by tests/test_host_glue.py (-m gpu).
 *   gcc -O2 host_shim_test.c -I../../include -L.. -lngravs_hip -lm -Wl,-rpath,$PWD/.. -o host_shim_test
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "ngravs_hip.h"

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
  if(ngravs_get_accel(ctx, &P[0].GravAccel[0], sizeof(*P), NULL, 0, &P[0].OldAcc, sizeof(*P), &P[0].GravCost, sizeof(*P), 0))
    return 5;
  double worst = 0, sum = 0;
  int nact = 0;
  for(i = 0; i < N; i++)
    {
      if(!active[i])
        {
          if(P[i].GravAccel[0] != 0 || P[i].GravCost != 0)
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
    if(ngravs_get_accel(ctx, &P[0].GravAccel[0], sizeof(*P), NULL, 0, NULL, 0, NULL, 0, 0))
      return 7;
    for(i = 0; i < N; i++)
      for(k = 0; k < 3; k++)
        if(keep[i][k] != P[i].GravAccel[k])
          bad++;
    free(keep);
  }
  ngravs_stats_t st;
  ngravs_get_stats(ctx, &st);
  printf("host_shim_test: N=%d active=%d (engine says %ld) mean err %.3e worst %.3e ia/part %.1f nodes %ld bad=%d\n", N, nact,
         (long)st.n_active, sum / nact, worst, st.interactions / st.n_active, (long)st.n_nodes, bad);
  ngravs_destroy(ctx);
  return (bad == 0 && st.n_active == nact && sum / nact < 5e-3 && worst < 0.1) ? 0 : 1;
}
