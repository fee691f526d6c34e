/* tests/glue_stub/glue_driver.c -- TEST-ONLY: a one-task stand-in for the parts of the reference that surround gadget_glue.c, so
 * that the glue is not only compiled but RUN on the GPU: the globals of allvars.c it touches, one-task MPI, the handful of
 * reference helpers it calls (second, timediff, endrun, do_box_wrapping, get_random_number) and the force-law symbols whose
 * ADDRESSES init_grav_maps() wires (ngravs_core.c:201-425; the bodies are never called on this path).
 *
 * main: reads a particle set + parameters written by tests/test_host_glue.py, fills P[] / All the way begrun()/init() would,
 * then runs the reference's own call sequence of one force computation (accel.c:24-58 via run.c): pm_init_periodic(),
 * domain_Decomposition(), pmforce_periodic() [PM step], gravity_tree(), gravity_forcetest() [FORCETEST]; a second, short-range
 * only step with a sparse active set follows (only Ti_endstep == Ti_Current rows may change).  P[]'s results go to a file.
 *
 *   gcc -DNGRAVS_BUILD_INSIDE_REFERENCE -DDOUBLEPRECISION -DUNEQUALSOFTENINGS [-DPERIODIC -DPMGRID=32 -DFORCETEST=0.02]
 *       -DYUKAWA_IMASS=60 -Itests/glue_stub -Iinclude host/gadget_glue.c tests/glue_stub/glue_driver.c -lngravs_hip -lm
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <time.h>
#include <mpi.h>
#include "allvars.h"
#include "proto.h"
#include "ngravs.h"

/* ---- allvars.c ----------------------------------------------------------------------------------------------------------- */
gravity AccelFxns[N_GRAVS][N_GRAVS], AccelSplines[N_GRAVS][N_GRAVS], GreensFxns[N_GRAVS][N_GRAVS], NormedGreensFxns[N_GRAVS][N_GRAVS];
int TypeToGrav[6];
int NgravLocal[N_GRAVS];
int ThisTask = 0, NTask = 1, NumPart = 0;
long long Ntype[6];
int NtypeLocal[6];
int TreeReconstructFlag;
double DomainCorner[3], DomainCenter[3], DomainLen, DomainFac;
double TimeOfLastTreeConstruction;
FILE *FdTimings, *FdForceTest;
int Numnodestree;
int *Father;
struct global_data_all_processes All;
struct particle_data *P;

/* ---- force laws: only their addresses are used here ------------------------------------------------------------------------ */
#define LAWBODY(f) double f(double a, double b, double c, double d, long n) { (void)a; (void)b; (void)c; (void)d; (void)n; return 0.0; }
LAWBODY(none) LAWBODY(newtonian) LAWBODY(neg_newtonian) LAWBODY(plummer) LAWBODY(neg_plummer) LAWBODY(pgdelta) LAWBODY(neg_pgdelta)
LAWBODY(normed_pgdelta) LAWBODY(bambam) LAWBODY(sourcebambaryon) LAWBODY(sourcebaryonbam) LAWBODY(bambam_spline)
LAWBODY(sourcebambaryon_spline) LAWBODY(sourcebaryonbam_spline) LAWBODY(yukawa) LAWBODY(pgyukawa) LAWBODY(normed_pgyukawa)
LAWBODY(coloyuk) LAWBODY(pgcoloyuk) LAWBODY(normed_pgcoloyuk)

/* ---- system.c / run.c helpers ------------------------------------------------------------------------------------------------ */
double second(void)
{
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}
double timediff(double t0, double t1) { return t1 - t0; }
void endrun(int code)
{
  printf("endrun(%d)\n", code);
  fflush(stdout);
  exit(code ? (code & 127) | 1 : 0);
}
double get_random_number(int id)   /* a fixed pseudo-random number per particle ID (system.c:26-40 draws from a table of 1000) */
{
  unsigned int x = (unsigned int)id * 2654435761u;
  x ^= x >> 15;
  return (double)(x % 1000u) / 1000.0;
}
#ifdef PERIODIC
void do_box_wrapping(void)   /* predict.c:107-133 */
{
  int i, j;
  for(i = 0; i < NumPart; i++)
    for(j = 0; j < 3; j++)
      {
        while(P[i].Pos[j] < 0)
          P[i].Pos[j] += All.BoxSize;
        while(P[i].Pos[j] >= All.BoxSize)
          P[i].Pos[j] -= All.BoxSize;
      }
}
#endif

/* ---- MPI with one task ----------------------------------------------------------------------------------------------------- */
static size_t tsize(MPI_Datatype t) { return t == MPI_BYTE ? 1 : (t == MPI_INT ? 4 : 8); }
int MPI_Allreduce(const void *s, void *r, int n, MPI_Datatype t, MPI_Op o, MPI_Comm c)
{
  (void)o;
  (void)c;
  if(s != MPI_IN_PLACE)
    memcpy(r, s, tsize(t) * (size_t)n);
  return MPI_SUCCESS;
}
int MPI_Barrier(MPI_Comm c) { (void)c; return MPI_SUCCESS; }
int MPI_Allgatherv(const void *s, int n, MPI_Datatype t, void *r, const int *cnt, const int *dsp, MPI_Datatype rt, MPI_Comm c)
{
  (void)cnt;
  (void)rt;
  (void)c;
  memcpy((char *)r + tsize(t) * (size_t)dsp[0], s, tsize(t) * (size_t)n);
  return MPI_SUCCESS;
}
int MPI_Bcast(void *b, int n, MPI_Datatype t, int root, MPI_Comm c) { (void)b; (void)n; (void)t; (void)root; (void)c; return MPI_SUCCESS; }
int MPI_Allgather(const void *s, int n, MPI_Datatype t, void *r, int rn, MPI_Datatype rt, MPI_Comm c)
{
  (void)rn;
  (void)rt;
  (void)c;
  memcpy(r, s, tsize(t) * (size_t)n);
  return MPI_SUCCESS;
}
int MPI_Alltoall(const void *s, int n, MPI_Datatype t, void *r, int rn, MPI_Datatype rt, MPI_Comm c)
{
  return MPI_Allgather(s, n, t, r, rn, rt, c);
}
int MPI_Alltoallv(const void *s, const int *sc, const int *sd, MPI_Datatype t, void *r, const int *rc, const int *rd, MPI_Datatype rt, MPI_Comm c)
{
  (void)rc;
  (void)rt;
  (void)c;
  memcpy((char *)r + tsize(t) * (size_t)rd[0], (const char *)s + tsize(t) * (size_t)sd[0], tsize(t) * (size_t)sc[0]);
  return MPI_SUCCESS;
}
/* (never reached with one task: the glue's all-to-all-v over Isend/Irecv only runs inside ngravs_host_* with NTask > 1) */
int MPI_Irecv(void *b, int n, MPI_Datatype t, int src, int tag, MPI_Comm c, MPI_Request *q) { (void)b; (void)n; (void)t; (void)src; (void)tag; (void)c; (void)q; return 1; }
int MPI_Isend(const void *b, int n, MPI_Datatype t, int dst, int tag, MPI_Comm c, MPI_Request *q) { (void)b; (void)n; (void)t; (void)dst; (void)tag; (void)c; (void)q; return 1; }
int MPI_Waitall(int n, MPI_Request *q, MPI_Status *s) { (void)n; (void)q; (void)s; return 1; }

/* ---- the run ----------------------------------------------------------------------------------------------------------------- */
static void dump(FILE *f)
{
  int i;
  for(i = 0; i < NumPart; i++)
    {
      double row[8] = {P[i].GravAccel[0], P[i].GravAccel[1], P[i].GravAccel[2], 0, 0, 0, P[i].OldAcc, (double)P[i].GravCost};
#ifdef PMGRID
      row[3] = P[i].GravPM[0];
      row[4] = P[i].GravPM[1];
      row[5] = P[i].GravPM[2];
#endif
      fwrite(row, sizeof(double), 8, f);
    }
}

int main(int argc, char **argv)
{
  FILE *f;
  double hd[16];
  int i, j, n;
  if(argc < 3 || !(f = fopen(argv[1], "rb")))
    return 2;
  /* header: n, G, BoxSize, ErrTolTheta, ErrTolForceAcc, softening[6] (ForceSoftening / 2.8 = the Plummer-equivalent lengths) */
  if(fread(hd, sizeof(double), 11, f) != 11)
    return 3;
  n = (int)hd[0];
  memset(&All, 0, sizeof(All));
  All.G = hd[1];
  All.BoxSize = hd[2];
  All.ErrTolTheta = hd[3];
  All.ErrTolForceAcc = hd[4];
  All.SofteningGas = hd[5];
  All.SofteningHalo = hd[6];
  All.SofteningDisk = hd[7];
  All.SofteningBulge = hd[8];
  All.SofteningStars = hd[9];
  All.SofteningBndry = hd[10];
  All.TypeOfOpeningCriterion = 1;
  All.TotNumPart = n;
  All.MaxPart = n + 16;
  All.PartAllocFactor = 1.5;
  All.TreeAllocFactor = 0.8;
  All.TreeDomainUpdateFrequency = 0.0;
  All.Time = 1.0;
  strcpy(All.OutputDir, argc > 3 ? argv[3] : "./");
  P = calloc((size_t)All.MaxPart, sizeof(*P));
  NumPart = n;
  for(i = 0; i < n; i++)
    {
      double row[5];
      if(fread(row, sizeof(double), 5, f) != 5)
        return 4;
      P[i].Pos[0] = row[0];
      P[i].Pos[1] = row[1];
      P[i].Pos[2] = row[2];
      P[i].Mass = row[3];
      P[i].Type = (int)row[4];
      P[i].ID = (unsigned int)(i + 1);
      P[i].Ti_endstep = 0;
    }
  fclose(f);
  /* init_grav_maps(): types 1..N_GRAVS -> species 0..N_GRAVS-1, the rest species 0, and the wiring of the bench's C4 case (ngravs_core.c:201-425): Newton inside a
   * species, Newton + Yukawa ("coloyuk") across; for N_GRAVS = 1 plain Newton */
  for(i = 0; i < 6; i++)
    TypeToGrav[i] = (i >= 1 && i <= N_GRAVS) ? i - 1 : 0;
  for(i = 0; i < N_GRAVS; i++)
    for(j = 0; j < N_GRAVS; j++)
      {
        const int cross = i != j;
        AccelFxns[i][j] = cross ? coloyuk : newtonian;
        AccelSplines[i][j] = plummer;
        GreensFxns[i][j] = cross ? pgcoloyuk : pgdelta;
        NormedGreensFxns[i][j] = cross ? normed_pgcoloyuk : normed_pgdelta;
      }
  FdTimings = fopen("/dev/null", "w");
  set_softenings();              /* init.c:60 */
  force_treeallocate((int)(All.TreeAllocFactor * All.MaxPart), All.MaxPart);
#ifdef PMGRID
  pm_init_periodic();            /* init.c:80-84 */
#endif
#ifdef PERIODIC
  lattice_init();                /* begrun.c:48 (a no-op here: the tables are built on the device on first use) */
#endif
  f = fopen(argv[2], "wb");
  if(!f)
    return 5;
  /* step 1: every particle active, a PM step: compute_accelerations(0), accel.c:24-58 */
  All.Ti_Current = 0;
  All.PM_Ti_endstep = 0;
  All.NumForcesSinceLastDomainDecomp = 1 + All.TotNumPart;
  domain_Decomposition();        /* run.c:68 */
#ifdef PMGRID
  pmforce_periodic();            /* long_range_force(), accel.c:39 */
#endif
  gravity_tree();                /* accel.c:44: Barnes-Hut criterion with OldAcc = 0 ... */
  gravity_tree();                /* accel.c:48-52: ... and again with the relative criterion on the first step */
#ifdef FORCETEST
  gravity_forcetest();           /* accel.c:56 */
#endif
  dump(f);
  /* step 2: a short-range step (no PM force), one particle in five active; the others must keep what they have */
  All.Ti_Current = 8;
  All.PM_Ti_endstep = 16;
  for(i = 0; i < NumPart; i++)
    P[i].Ti_endstep = (i % 5 == 2) ? 8 : 16;
  All.NumForcesSinceLastDomainDecomp = 1 + All.TotNumPart;   /* TreeDomainUpdateFrequency = 0: every step re-decomposes */
  domain_Decomposition();
  gravity_tree();
  dump(f);
  fclose(f);
  printf("glue driver: %d particles, N_GRAVS %d, two steps done; TotNumOfForces %lld\n", NumPart, N_GRAVS, All.TotNumOfForces);
  return 0;
}
