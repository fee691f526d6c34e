// pm_common.hpp -- pieces shared by the single-task mesh (kernels_pm.hip) and the slab-decomposed mesh (kernels_pmslab.hip)
#pragma once
#include "engine.hpp"
#include <hipfft/hipfft.h>
#include <cmath>

__device__ __forceinline__ int cell_of(double x, double to_slab, int N, double *frac)
{
  int s = (int)(to_slab * x);                                         // pm_periodic.c:299-301
  if(s >= N)
    s = N - 1;
  *frac = to_slab * x - s;
  return s;
}

__device__ __forceinline__ int wrapN(int a, int N) { return a < 0 ? a + N : (a >= N ? a - N : a); }

// Where a mesh cell lives: in the full padded mesh [N][N][N+2] of the single-task path, or in a task's brick [ext0][ext1][ext2]
// of the cells (lo + i) mod N (slab-decomposed path, kernels_pmslab.hip).  The deposit kernels are written once over this.
struct MeshAddr
{
  int N, brick;
  int lo[3], ext[3];
  __device__ __forceinline__ long long cell(int x, int y, int z) const   // periodic mesh coordinates, each within [-N, 2N)
  {
    if(brick)
      return ((long long)wrapN(wrapN(x, N) - lo[0], N) * ext[1] + wrapN(wrapN(y, N) - lo[1], N)) * ext[2] + wrapN(wrapN(z, N) - lo[2], N);
    return ((long long)wrapN(x, N) * N + wrapN(y, N)) * (N + 2) + wrapN(z, N);
  }
  __host__ __device__ long long species_stride() const
  {
    return brick ? (long long)ext[0] * ext[1] * ext[2] : (long long)N * N * (N + 2);
  }
};

struct GreenParams
{
  int ng, N;
  double asmth2;       // (2 pi asmth / L)^2                                     pm_periodic.c:234-235
  double ym2;          // (YUKAWA_IMASS / 2 pi)^2                                 ngravs.c:871
  double yfac;         // exp(-ym^2 asmth2)                                       ngravs.c:877
  double cN[NG_MAX][NG_MAX], cY[NG_MAX][NG_MAX];   // [source][target], as pm_periodic.c:490 indexes GreensFxns
};


// the k-space factor of pm_periodic.c:436-520 for mode (x, y, z) (z < N/2+1): phi_b(k) = sum_a G_ab(k) rho_a(k) * (-exp(-k^2 asmth2)) / sinc^4.
// rho holds the NG source transforms `sstride` complex numbers apart, this mode at index `idx` of each.
template <int NG>
__device__ __forceinline__ void green_mode(const GreenParams &gp, int x, int y, int z, const double2 *__restrict__ rho, size_t sstride,
                                           size_t idx, double2 (&out)[NG])
{
  const int N = gp.N;
  double kx = x > N / 2 ? x - N : x, ky = y > N / 2 ? y - N : y, kz = z;   // pm_periodic.c:440-451
  double k2 = kx * kx + ky * ky + kz * kz;
#pragma unroll
  for(int b = 0; b < NG; b++)
    out[b].x = out[b].y = 0;
  if(k2 > 0)
    {
      double fx = 1, fy = 1, fz = 1;
      if(kx != 0)
        {
          fx = (M_PI * kx) / N;
          fx = sin(fx) / fx;
        }
      if(ky != 0)
        {
          fy = (M_PI * ky) / N;
          fy = sin(fy) / fy;
        }
      if(kz != 0)
        {
          fz = (M_PI * kz) / N;
          fz = sin(fz) / fz;
        }
      double ff = 1 / (fx * fy * fz);
      double common = -exp(-k2 * gp.asmth2) * ff * ff * ff * ff;       // pm_periodic.c:491
      double gN = 1.0 / k2, gY = 1.0 / (k2 + gp.ym2) * gp.yfac;         // pgdelta / pgyukawa
#pragma unroll
      for(int a = 0; a < NG; a++)
        {
          double2 r = rho[(size_t)a * sstride + idx];
#pragma unroll
          for(int b = 0; b < NG; b++)
            {
              double smth = (gp.cN[a][b] * gN + gp.cY[a][b] * gY) * common;
              out[b].x += r.x * smth;
              out[b].y += r.y * smth;
            }
        }
    }
}

static inline void make_green_params(const ngravs_ctx *c, GreenParams *gpp)
{
  GreenParams &gp = *gpp;
  const int N = c->cfg.pmgrid, ng = c->cfg.n_gravs;
  const double L = c->cfg.box_size;
  memset(&gp, 0, sizeof(gp));
  gp.ng = ng;
  gp.N = N;
  gp.asmth2 = (2 * M_PI) * c->asmth / L;
  gp.asmth2 *= gp.asmth2;
  double ym = c->cfg.yukawa_imass / (2 * M_PI);
  gp.ym2 = ym * ym;
  gp.yfac = exp(-ym * ym * gp.asmth2);
  for(int a = 0; a < ng; a++)
    for(int b = 0; b < ng; b++)
      {
        int law = c->cfg.law_greens[a][b];   // [source][target] (pm_periodic.c:490)
        gp.cN[a][b] = law == NGRAVS_LAW_NEWTON || law == NGRAVS_LAW_COLOYUK ? 1.0 : (law == NGRAVS_LAW_NEG_NEWTON ? -1.0 : 0.0);
        gp.cY[a][b] = law == NGRAVS_LAW_YUKAWA || law == NGRAVS_LAW_COLOYUK ? 1.0 : 0.0;
      }
}

#define FFT_TRY(ctx, expr)                                                                  \
  do                                                                                        \
    {                                                                                       \
      hipfftResult r__ = (expr);                                                            \
      if(r__ != HIPFFT_SUCCESS)                                                             \
        {                                                                                   \
          ngravs_report(ctx, NGRAVS_ERR_NO_DEVICE, std::string(#expr) + ": hipfft error " + std::to_string((int)r__)); \
          return NGRAVS_ERR_NO_DEVICE;                                                      \
        }                                                                                   \
    }                                                                                       \
  while(0)

// kernels_pm.hip: tiled CIC deposit into a full mesh or a brick (see there)
int pm_deposit_tiles(ngravs_ctx *c, const MeshAddr &ma, double *dst);
