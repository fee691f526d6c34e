/* tests/glue_stub/allvars.h -- TEST-ONLY declarations of exactly the reference globals gadget_glue.c touches, so that the glue can
 * be compiled with -fsyntax-only -Wall -Werror in a tree without the reference (whose own allvars.h needs GSL and FFTW-2).
 * Names, types and the struct particle_data layout follow SURVEY.md 8(a') / 8(b) (reference allvars.h:32-38, 93-97, 131-161,
 * 265-279, 299-450, 546-581, 616, 664).  Nothing here is used by the product. */
#ifndef ALLVARS_H
#define ALLVARS_H
#include <stdio.h>
#ifndef N_GRAVS
#define N_GRAVS 2
#endif
#define ASMTH 1.25
#define RCUT 4.5
#define MAXLEN_FILENAME 100
#ifdef DOUBLEPRECISION
#define FLOAT double
#else
#define FLOAT float
#endif
typedef long long peanokey;
typedef double (*gravity)(double, double, double, double, long);

extern gravity AccelFxns[N_GRAVS][N_GRAVS], AccelSplines[N_GRAVS][N_GRAVS], GreensFxns[N_GRAVS][N_GRAVS], NormedGreensFxns[N_GRAVS][N_GRAVS];
extern int TypeToGrav[6];
extern int NgravLocal[N_GRAVS];
extern int ThisTask, NTask, NumPart;
extern long long Ntype[6];
extern int NtypeLocal[6];
extern int TreeReconstructFlag;
extern double DomainCorner[3], DomainCenter[3], DomainLen, DomainFac;
extern double TimeOfLastTreeConstruction;
extern FILE *FdTimings, *FdForceTest;
extern int Numnodestree;
extern int *Father;

extern struct global_data_all_processes
{
  long long TotNumPart, TotN_gas;
  int MaxPart;
  double PartAllocFactor, TreeAllocFactor;
  double ErrTolTheta, ErrTolForceAcc;
  int TypeOfOpeningCriterion;
  long long TotNumOfForces, NumForcesSinceLastDomainDecomp;
  double G, BoxSize, Time, TimeStep;
  int NumCurrentTiStep, Ti_Current, PM_Ti_endstep;
  double Asmth[2], Rcut[2];
  double ForceSoftening[6], SofteningTable[6];
  double SofteningGas, SofteningHalo, SofteningDisk, SofteningBulge, SofteningStars, SofteningBndry;
  double SofteningGasMaxPhys, SofteningHaloMaxPhys, SofteningDiskMaxPhys, SofteningBulgeMaxPhys, SofteningStarsMaxPhys, SofteningBndryMaxPhys;
  double MinGasHsml, MinGasHsmlFractional;
  int ComovingIntegrationOn;
  double TreeDomainUpdateFrequency;
  double CPU_TreeConstruction, CPU_TreeWalk, CPU_Imbalance, CPU_CommSum, CPU_PM, CPU_Domain, CPU_Peano;
  char OutputDir[MAXLEN_FILENAME];
} All;

extern struct particle_data
{
  FLOAT Pos[3], Mass, Vel[3], GravAccel[3];
#ifdef PMGRID
  FLOAT GravPM[3];
#endif
#ifdef FORCETEST
  FLOAT GravAccelDirect[3];
#endif
  FLOAT Potential, OldAcc;
  unsigned int ID;
  int Type, Ti_endstep, Ti_begstep;
  float GravCost;
} *P;
#endif
