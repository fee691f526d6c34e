/* tests/glue_stub/glue_driver.c -- TEST-ONLY: a one-task stand-in for the parts of the reference that surround gadget_glue.c, so
 * that the glue is not only compiled but RUN on the GPU: the globals of allvars.c it touches, one-task MPI, the handful of
 * reference helpers it calls (second, timediff, endrun, do_box_wrapping, get_random_number) and the force-law symbols whose
 * ADDRESSES init_grav_maps() wires (ngravs_core.c:201-425; the bodies are never called on this path).
 *
 * main: reads a particle set + parameters written by tests/test_host_glue.py, fills P[] / All the way begrun()/init() would,
 * then runs the reference's own call sequence of one force computation (accel.c:24-58 via run.c): pm_init_periodic(),
 * domain_Decomposition(), pmforce_periodic() [PM step], gravity_tree(), gravity_forcetest() [FORCETEST]; a second, short-range
 * only step with a sparse active set follows (only Ti_endstep == Ti_Current rows may change).  P[]'s results go to one file per task
 * (rows: GravAccel, GravPM, OldAcc, GravCost, ID; the dump of step 1 and, after it, of step 2 -- the particle number of a task
 * may differ between the two).  -DGLUE_NTASK=2: two tasks (forked processes, MPI through shared memory).
 *
 *   gcc -DNGRAVS_BUILD_INSIDE_REFERENCE -DDOUBLEPRECISION -DUNEQUALSOFTENINGS [-DPERIODIC -DPMGRID=32 -DFORCETEST=0.02]
 *       -DYUKAWA_IMASS=60 -Itests/glue_stub -Iinclude host/gadget_glue.c tests/glue_stub/glue_driver.c -lngravs_hip -lm
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <time.h>
#include <unistd.h>
#include <signal.h>
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
  printf("task %d: endrun(%d)\n", ThisTask, code);
  fflush(stdout);
  /* a task that ends the run would leave the others in their barrier for ever (and on the GPU): the run's tasks are one process
   * group (mpi_start), which ends here as a whole -- MPI_Abort */
  if(code && NTask > 1)
    {
      usleep(300000);   /* (the other tasks may be on their way to say why) */
      kill(0, SIGTERM);
    }
  _exit(code ? (code & 127) | 1 : 0);
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

/* ---- MPI: GLUE_NTASK tasks as forked processes of this program, collectives through one shared mapping ---------------------------
 * (every task copies what it contributes into its slot, a process-shared barrier, every task copies out what it is owed, a second
 * barrier).  Point-to-point: the glue's all-to-all-v posts at most one Isend and one Irecv per peer and then waits for all -- the
 * exchange happens in MPI_Waitall, which every task enters once per round.  With one task everything degenerates to copies. */
#include <pthread.h>
#include <sys/mman.h>
#include <sys/wait.h>
#include <sys/prctl.h>
#include <signal.h>
#include <unistd.h>
#ifndef GLUE_NTASK
#define GLUE_NTASK 1
#endif
#define SLOT ((size_t)96 << 20)
static struct shared
{
  pthread_barrier_t bar;
  int failed;
} *Sh;
static char *Slots;
static char *slot(int r) { return Slots + (size_t)r * SLOT; }
static void sync_tasks(void)
{
  if(NTask > 1)
    pthread_barrier_wait(&Sh->bar);
}
static size_t tsize(MPI_Datatype t) { return t == MPI_BYTE ? 1 : (t == MPI_INT ? 4 : 8); }
static void need(size_t bytes)
{
  if(bytes > SLOT)
    {
      printf("glue driver: a message of %zu bytes does not fit the shared slot\n", bytes);
      exit(3);
    }
}
static void mpi_start(void)
{
  pthread_barrierattr_t at;
  int r;
  NTask = GLUE_NTASK;
  if(NTask > 1)
    setpgid(0, 0);   /* the tasks of this run: one process group (endrun) */
  Sh = mmap(NULL, sizeof(*Sh), PROT_READ | PROT_WRITE, MAP_SHARED | MAP_ANONYMOUS, -1, 0);
  Slots = mmap(NULL, SLOT * (size_t)NTask, PROT_READ | PROT_WRITE, MAP_SHARED | MAP_ANONYMOUS, -1, 0);
  if(Sh == MAP_FAILED || Slots == MAP_FAILED)
    exit(3);
  pthread_barrierattr_init(&at);
  pthread_barrierattr_setpshared(&at, PTHREAD_PROCESS_SHARED);
  pthread_barrier_init(&Sh->bar, &at, (unsigned)NTask);
  fflush(stdout);
  for(r = 1; r < NTask; r++)   /* before anything has touched the GPU */
    if(fork() == 0)
      {
        ThisTask = r;
        prctl(PR_SET_PDEATHSIG, SIGKILL);   /* task 0 killed (a test's timeout): no task is left behind on the GPU */
        break;
      }
}
static int mpi_finish(int rc)
{
  int r, st, bad = rc;
  fflush(stdout);
  if(ThisTask != 0)
    _exit(rc);
  for(r = 1; r < NTask; r++)
    if(wait(&st) < 0 || !WIFEXITED(st) || WEXITSTATUS(st) != 0)
      bad = bad ? bad : 9;
  return bad;
}
int MPI_Allreduce(const void *s, void *r, int n, MPI_Datatype t, MPI_Op o, MPI_Comm c)
{
  const size_t b = tsize(t) * (size_t)n;
  int q, k;
  (void)c;
  need(b);
  memcpy(slot(ThisTask), s == MPI_IN_PLACE ? r : s, b);
  sync_tasks();
  for(k = 0; k < n; k++)
    {
      if(t == MPI_DOUBLE)
        {
          double v = ((double *)slot(0))[k];
          for(q = 1; q < NTask; q++)
            {
              const double w = ((double *)slot(q))[k];
              v = o == MPI_SUM ? v + w : (o == MPI_MIN ? (w < v ? w : v) : (w > v ? w : v));
            }
          ((double *)r)[k] = v;
        }
      else if(t == MPI_LONG_LONG)
        {
          long long v = ((long long *)slot(0))[k];
          for(q = 1; q < NTask; q++)
            {
              const long long w = ((long long *)slot(q))[k];
              v = o == MPI_SUM ? v + w : (o == MPI_MIN ? (w < v ? w : v) : (w > v ? w : v));
            }
          ((long long *)r)[k] = v;
        }
      else
        {
          int v = ((int *)slot(0))[k];
          for(q = 1; q < NTask; q++)
            {
              const int w = ((int *)slot(q))[k];
              v = o == MPI_SUM ? v + w : (o == MPI_MIN ? (w < v ? w : v) : (w > v ? w : v));
            }
          ((int *)r)[k] = v;
        }
    }
  sync_tasks();
  return MPI_SUCCESS;
}
int MPI_Barrier(MPI_Comm c)
{
  (void)c;
  sync_tasks();
  return MPI_SUCCESS;
}
int MPI_Allgatherv(const void *s, int n, MPI_Datatype t, void *r, const int *cnt, const int *dsp, MPI_Datatype rt, MPI_Comm c)
{
  int q;
  (void)rt;
  (void)c;
  need(tsize(t) * (size_t)n);
  memcpy(slot(ThisTask), s, tsize(t) * (size_t)n);
  sync_tasks();
  for(q = 0; q < NTask; q++)
    memcpy((char *)r + tsize(t) * (size_t)dsp[q], slot(q), tsize(t) * (size_t)cnt[q]);
  sync_tasks();
  return MPI_SUCCESS;
}
int MPI_Bcast(void *b, int n, MPI_Datatype t, int root, MPI_Comm c)
{
  (void)c;
  need(tsize(t) * (size_t)n);
  if(ThisTask == root)
    memcpy(slot(root), b, tsize(t) * (size_t)n);
  sync_tasks();
  if(ThisTask != root)
    memcpy(b, slot(root), tsize(t) * (size_t)n);
  sync_tasks();
  return MPI_SUCCESS;
}
int MPI_Allgather(const void *s, int n, MPI_Datatype t, void *r, int rn, MPI_Datatype rt, MPI_Comm c)
{
  int q;
  (void)rn;
  (void)rt;
  (void)c;
  need(tsize(t) * (size_t)n);
  memcpy(slot(ThisTask), s, tsize(t) * (size_t)n);
  sync_tasks();
  for(q = 0; q < NTask; q++)
    memcpy((char *)r + tsize(t) * (size_t)n * (size_t)q, slot(q), tsize(t) * (size_t)n);
  sync_tasks();
  return MPI_SUCCESS;
}
int MPI_Alltoall(const void *s, int n, MPI_Datatype t, void *r, int rn, MPI_Datatype rt, MPI_Comm c)
{
  const size_t b = tsize(t) * (size_t)n;
  int q;
  (void)rn;
  (void)rt;
  (void)c;
  need(b * (size_t)NTask);
  memcpy(slot(ThisTask), s, b * (size_t)NTask);
  sync_tasks();
  for(q = 0; q < NTask; q++)
    memcpy((char *)r + b * (size_t)q, slot(q) + b * (size_t)ThisTask, b);
  sync_tasks();
  return MPI_SUCCESS;
}
int MPI_Alltoallv(const void *s, const int *sc, const int *sd, MPI_Datatype t, void *r, const int *rc, const int *rd, MPI_Datatype rt, MPI_Comm c)
{
  const size_t e = tsize(t), hdr = sizeof(int) * 64;
  size_t end = 0;
  int q;
  (void)rt;
  (void)c;
  for(q = 0; q < NTask; q++)
    if((size_t)(sd[q] + sc[q]) > end)
      end = (size_t)(sd[q] + sc[q]);
  need(hdr + e * end);
  memcpy(slot(ThisTask), sd, sizeof(int) * (size_t)NTask);   /* where each peer's block starts in my send buffer */
  if(end)
    memcpy(slot(ThisTask) + hdr, s, e * end);
  sync_tasks();
  for(q = 0; q < NTask; q++)
    if(rc[q] > 0)
      memcpy((char *)r + e * (size_t)rd[q], slot(q) + hdr + e * (size_t)((int *)slot(q))[ThisTask], e * (size_t)rc[q]);
  sync_tasks();
  return MPI_SUCCESS;
}
/* point-to-point of one round: at most one send and one receive per peer, completed together in MPI_Waitall */
static struct pend
{
  const void *sbuf[64];
  void *rbuf[64];
  size_t sb[64], rb[64];
} Pend;
int MPI_Irecv(void *b, int n, MPI_Datatype t, int src, int tag, MPI_Comm c, MPI_Request *q)
{
  (void)tag;
  (void)c;
  Pend.rbuf[src] = b;
  Pend.rb[src] = tsize(t) * (size_t)n;
  *q = 0;
  return MPI_SUCCESS;
}
int MPI_Isend(const void *b, int n, MPI_Datatype t, int dst, int tag, MPI_Comm c, MPI_Request *q)
{
  (void)tag;
  (void)c;
  Pend.sbuf[dst] = b;
  Pend.sb[dst] = tsize(t) * (size_t)n;
  *q = 0;
  return MPI_SUCCESS;
}
int MPI_Waitall(int n, MPI_Request *q, MPI_Status *s)
{
  const size_t hdr = sizeof(size_t) * 128;
  size_t *h = (size_t *)slot(ThisTask), off = 0;
  int p, rc = MPI_SUCCESS;
  (void)n;
  (void)q;
  (void)s;
  for(p = 0; p < NTask; p++)   /* header: offset and length of the block for every peer */
    {
      h[2 * p] = off;
      h[2 * p + 1] = Pend.sb[p];
      off += Pend.sb[p];
    }
  need(hdr + off);
  for(p = 0; p < NTask; p++)
    if(Pend.sb[p])
      memcpy(slot(ThisTask) + hdr + h[2 * p], Pend.sbuf[p], Pend.sb[p]);
  sync_tasks();
  for(p = 0; p < NTask; p++)
    if(Pend.rb[p])
      {
        const size_t *hp = (const size_t *)slot(p);
        if(hp[2 * ThisTask + 1] != Pend.rb[p])
          rc = 1;   /* the peer sends another size than this task expects */
        else
          memcpy(Pend.rbuf[p], slot(p) + hdr + hp[2 * ThisTask], Pend.rb[p]);
      }
  sync_tasks();
  memset(&Pend, 0, sizeof(Pend));
  return rc;
}

/* ---- the run ----------------------------------------------------------------------------------------------------------------- */
static void dump(FILE *f)
{
  const double np = (double)NumPart;
  int i;
  fwrite(&np, sizeof(double), 1, f);
  for(i = 0; i < NumPart; i++)
    {
      double row[9] = {P[i].GravAccel[0], P[i].GravAccel[1], P[i].GravAccel[2], 0, 0, 0, P[i].OldAcc, (double)P[i].GravCost, (double)P[i].ID};
#ifdef PMGRID
      row[3] = P[i].GravPM[0];
      row[4] = P[i].GravPM[1];
      row[5] = P[i].GravPM[2];
#endif
      fwrite(row, sizeof(double), 9, f);
    }
}

int main(int argc, char **argv)
{
  FILE *f;
  double hd[16];
  char name[600];
  int i, j, n, k = 0;
  if(argc < 3)
    return 2;
  mpi_start();                   /* GLUE_NTASK tasks from here on (forked before anything touches the GPU) */
  if(!(f = fopen(argv[1], "rb")))
    return mpi_finish(2);
  /* header: n, G, BoxSize, ErrTolTheta, ErrTolForceAcc, softening[6] (the Plummer-equivalent lengths of the six types) */
  if(fread(hd, sizeof(double), 11, f) != 11)
    return mpi_finish(3);
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
  All.PartAllocFactor = 1.6;
  All.MaxPart = (int)(All.PartAllocFactor * n / NTask) + 16;
  All.TreeAllocFactor = 0.8;
  All.TreeDomainUpdateFrequency = 0.0;
  All.Time = 1.0;
  strcpy(All.OutputDir, argc > 3 ? argv[3] : "./");
  P = calloc((size_t)All.MaxPart, sizeof(*P));
  for(i = 0; i < n; i++)         /* task r starts with every NTask-th particle: an arbitrary distribution, as after read_ic() */
    {
      double row[5];
      if(fread(row, sizeof(double), 5, f) != 5)
        return mpi_finish(4);
      if(i % NTask != ThisTask)
        continue;
      P[k].Pos[0] = row[0];
      P[k].Pos[1] = row[1];
      P[k].Pos[2] = row[2];
      P[k].Mass = row[3];
      P[k].Type = (int)row[4];
      P[k].ID = (unsigned int)(i + 1);
      P[k].Ti_endstep = 0;
      k++;
    }
  NumPart = k;
  fclose(f);
  /* init_grav_maps(): types 1..N_GRAVS -> species 0..N_GRAVS-1, the rest species 0, and the wiring of the bench's C4 case
   * (ngravs_core.c:201-425): Newton inside a species, Newton + Yukawa ("coloyuk") across; for N_GRAVS = 1 plain Newton */
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
  if(ThisTask == 0)              /* begrun.c:223-230 opens timings.txt on task 0 */
    {
      char tn[600];
      snprintf(tn, sizeof(tn), "%stimings.txt", argc > 3 ? argv[3] : "/tmp/");
      FdTimings = fopen(tn, "w");
    }
  else
    FdTimings = fopen("/dev/null", "w");
  set_softenings();              /* init.c:60 */
  force_treeallocate((int)(All.TreeAllocFactor * All.MaxPart), All.MaxPart);
#ifdef PMGRID
  pm_init_periodic();            /* init.c:80-84 */
#endif
#ifdef PERIODIC
  lattice_init();                /* begrun.c:48 (a no-op here: the tables are built on the device on first use) */
#endif
  snprintf(name, sizeof(name), "%s.%d", argv[2], ThisTask);
  f = fopen(name, "wb");
  if(!f)
    return mpi_finish(5);
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
    P[i].Ti_endstep = ((P[i].ID - 1) % 5 == 2) ? 8 : 16;
  All.TreeDomainUpdateFrequency = 1.0;
  All.NumForcesSinceLastDomainDecomp = 1 + 2 * All.TotNumPart;   /* more forces than the frequency allows: this step re-decomposes */
  domain_Decomposition();
  gravity_tree();
  dump(f);
    {
      /* step 3: TreeDomainUpdateFrequency > 0 and few forces since the last decomposition -- domain.c:76 keeps decomposition and
       * tree, the particles have drifted (move_particles, predict.c:36-104: no box wrapping between decompositions): the glue
       * hands the drifted positions over and the library refits the tree; with several tasks the imported copies follow their
       * originals and the top of the tree is summed again (ngravs_host_kept_step) */
      All.NumForcesSinceLastDomainDecomp = 0;
      All.Ti_Current = 12;
      for(i = 0; i < NumPart; i++)
        {
          P[i].Ti_endstep = ((P[i].ID - 1) % 3 == 1) ? 12 : 16;
          for(j = 0; j < 3; j++)
            P[i].Pos[j] += 1e-3 * (All.BoxSize > 0 ? All.BoxSize : 1.0) * sin(0.37 * (double)P[i].ID + 1.3 * j);
        }
      domain_Decomposition();
      gravity_tree();
      dump(f);
    }
  fclose(f);
  printf("glue driver: task %d of %d holds %d particles, N_GRAVS %d, three steps done; TotNumOfForces %lld\n", ThisTask, NTask, NumPart, N_GRAVS,
         All.TotNumOfForces);
  return mpi_finish(0);
}
