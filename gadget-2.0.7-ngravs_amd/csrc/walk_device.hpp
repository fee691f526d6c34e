// walk_device.hpp -- device helpers shared by the walk kernels (kernels_walk.hip, kernels_eval.hip)
#pragma once
#include "engine.hpp"

#ifndef WAVE
#define WAVE 64
#endif

__device__ __forceinline__ double nearest(double x, double box, double boxhalf)
{
  return (x > boxhalf) ? (x - box) : ((x < -boxhalf) ? (x + box) : x);   // NEAREST, forcetree.c:43
}
// the minimum-image wrap in 3 instructions instead of 8 (mul, rndne, fma) for the group traversal's conservative box tests,
// which only use |x|: identical to NEAREST for |x| < 1.5 box except within an ulp of |x| = box/2, where both images are
// equally far
__device__ __forceinline__ double nearest_abs(double x, double box, double invbox)
{
  return __builtin_fma(-__builtin_rint(x * invbox), box, x);
}
// exp(-x) for x >= 0:  x = (32 n + j) ln2/32 + f, |f| <= ln2/64;  exp(-x) = 2^-n * T[j] * P6(-f), T[j] = 2^(-j/32)
// (32-entry table in LDS: one 256-byte bank row, so distinct entries never conflict).  ~1 ulp.
__device__ __forceinline__ double exp_neg_fast(double x, const double *__restrict__ T)
{
  const double inv = 46.16624130844683;                                   // 32/ln2
  const double hi = 0.02166084938653512, lo = 5.9631716539705866e-12;      // ln2/32 = hi + lo, hi has 21 trailing zero bits
  double m = __builtin_rint(x * inv);
  double f = __builtin_fma(-m, hi, x);                                     // exact for m < 2^21
  f = __builtin_fma(-m, lo, f);
  int mi = (int)m;
  double t = T[mi & 31];
  double y = -f;                                                           // |y| <= ln2/64
  double pz = 1.0 / 720.0;
  pz = __builtin_fma(pz, y, 1.0 / 120.0);
  pz = __builtin_fma(pz, y, 1.0 / 24.0);
  pz = __builtin_fma(pz, y, 1.0 / 6.0);
  pz = __builtin_fma(pz, y, 0.5);
  pz = __builtin_fma(pz, y, 1.0);
  pz = __builtin_fma(pz, y, 1.0);
  return ldexp(t * pz, -(mi >> 5));
}

// wave-level "any lane": the condition's lane mask is compared on the scalar unit (no vector select / compare round trip)
__device__ __forceinline__ bool wave_any(bool p) { return __builtin_amdgcn_ballot_w64(p) != 0ull; }
__device__ __forceinline__ void wave_sync()
{
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  __builtin_amdgcn_wave_barrier();
}
__device__ __forceinline__ int lane_prefix(unsigned long long mask)   // # set bits below this lane
{
  return __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0));
}
__device__ __forceinline__ double wave_min(double v)
{
  for(int off = 32; off > 0; off >>= 1)
    {
      double o = __shfl_xor(v, off);
      v = o < v ? o : v;
    }
  return v;
}
// a wave-uniform double (every lane holds the same value) moved into SGPRs: frees two VGPRs per value that lives as long as the group
__device__ __forceinline__ double wave_uniform(double v)
{
  const int lo = __builtin_amdgcn_readfirstlane(__double2loint(v)), hi = __builtin_amdgcn_readfirstlane(__double2hiint(v));
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_max(double v)
{
  for(int off = 32; off > 0; off >>= 1)
    {
      double o = __shfl_xor(v, off);
      v = o > v ? o : v;
    }
  return v;
}

