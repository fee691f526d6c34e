// kernels_pm.hip -- long-range periodic particle-mesh force.
//
// Replaces (reference): pmforce_periodic (pm_periodic.c:204-790).  The reference runs, for every
// (source, target) species pair, CIC deposit -> forward FFT -> Green multiply -> inverse FFT ->
// three finite-difference sweeps -> CIC gather: 2*N_GRAVS^2 transforms and 3*N_GRAVS^2 mesh sweeps.
// Here: deposit + forward rocFFT once per SOURCE species, one fused k-space kernel that forms, per
// TARGET species b,  phi_b(k) = sum_a G_ab(k) rho_a(k) * (-exp(-k^2 asmth2)) / sinc^4  (linear, so
// identical to summing the reference's per-pair potentials), inverse rocFFT once per target species,
// and one fused kernel per target species that takes the 4-point gradient at the 8 CIC corners on
// the fly and gathers GravPM: 2*N_GRAVS transforms, no force mesh at all.
//
// Mesh layout: real [N][N][N+2] fp64 (in-place r2c padding), complex [N][N][N/2+1].  rocFFT is an
// unnormalised DFT with forward sign -1, like FFTW-2.
#include "engine.hpp"
#include "pm_common.hpp"
#include <hipfft/hipfft.h>

__global__ void k_cic_deposit(const double4 *__restrict__ s_pm, const unsigned char *__restrict__ s_type,
                              const unsigned char *__restrict__ s_flag, long long n,
                              double to_slab, int N, int ng, const int *__restrict__ t2g_tab, double *__restrict__ rho)
{
  long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if(i >= n || (s_flag[i] & 2))   // bit 1: halo copy of another task's particle -- its owner deposits it
    return;
  double4 p = s_pm[i];
  int g = t2g_tab[s_type[i]];
  const long long NZ = N + 2;
  double *grid = rho + (size_t)g * N * N * NZ;
  double dx, dy, dz;
  int sx = cell_of(p.x, to_slab, N, &dx), sy = cell_of(p.y, to_slab, N, &dy), sz = cell_of(p.z, to_slab, N, &dz);
  int sxx = sx + 1 == N ? 0 : sx + 1, syy = sy + 1 == N ? 0 : sy + 1, szz = sz + 1 == N ? 0 : sz + 1;
  double m = p.w;
  // pm_periodic.c:322-329 (same weights, same products)
  atomicAdd(&grid[((long long)sx * N + sy) * NZ + sz], m * (1.0 - dx) * (1.0 - dy) * (1.0 - dz));
  atomicAdd(&grid[((long long)sx * N + syy) * NZ + sz], m * (1.0 - dx) * dy * (1.0 - dz));
  atomicAdd(&grid[((long long)sx * N + sy) * NZ + szz], m * (1.0 - dx) * (1.0 - dy) * dz);
  atomicAdd(&grid[((long long)sx * N + syy) * NZ + szz], m * (1.0 - dx) * dy * dz);
  atomicAdd(&grid[((long long)sxx * N + sy) * NZ + sz], m * (dx) * (1.0 - dy) * (1.0 - dz));
  atomicAdd(&grid[((long long)sxx * N + syy) * NZ + sz], m * (dx)*dy * (1.0 - dz));
  atomicAdd(&grid[((long long)sxx * N + sy) * NZ + szz], m * (dx) * (1.0 - dy) * dz);
  atomicAdd(&grid[((long long)sxx * N + syy) * NZ + szz], m * (dx)*dy * dz);
}

// one thread per complex mode; rho[a] complex in, phi[b] complex out
template <int NG>
__global__ void k_green(const double2 *__restrict__ rho, double2 *__restrict__ phi, GreenParams gp)
{
  const int N = gp.N, NH = N / 2 + 1;
  long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  long long total = (long long)N * N * NH;
  if(idx >= total)
    return;
  int z = (int)(idx % NH);
  int y = (int)((idx / NH) % N);
  int x = (int)(idx / ((long long)NH * N));
  double2 out[NG];
  green_mode<NG>(gp, x, y, z, rho, (size_t)total, (size_t)idx, out);
#pragma unroll
  for(int b = 0; b < NG; b++)
    phi[(size_t)b * total + idx] = out[b];                             // k = 0 -> 0 (pm_periodic.c:519-520)
}

// fused 4-point gradient at the 8 CIC corners + CIC gather (pm_periodic.c:681-763)
__global__ void k_gradient_gather(const double4 *__restrict__ s_pm, const unsigned char *__restrict__ s_type,
                                  const unsigned char *__restrict__ s_flag,
                                  long long first, long long n, double to_slab, int N, const int *__restrict__ t2g_tab,
                                  const double *__restrict__ phi, double fac, double *__restrict__ r_pm)
{
  long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if(i >= n)
    return;
  i += first;
  if(s_flag[i] & 2)
    {
      r_pm[3 * i + 0] = r_pm[3 * i + 1] = r_pm[3 * i + 2] = 0.0;
      return;
    }
  double4 p = s_pm[i];
  int g = t2g_tab[s_type[i]];
  const long long NZ = N + 2;
  const double *grid = phi + (size_t)g * N * N * NZ;
  double dx, dy, dz;
  int sx = cell_of(p.x, to_slab, N, &dx), sy = cell_of(p.y, to_slab, N, &dy), sz = cell_of(p.z, to_slab, N, &dz);
  double wx[2] = {1.0 - dx, dx}, wy[2] = {1.0 - dy, dy}, wz[2] = {1.0 - dz, dz};
  double acc[3] = {0, 0, 0};
  auto wrap = [N](int a) { return a < 0 ? a + N : (a >= N ? a - N : a); };
  auto at = [&](int x, int y, int z) { return grid[((long long)x * N + y) * NZ + z]; };
  // corner order of the reference's gather (x outer, then y/z as written at pm_periodic.c:749-757)
  const int ox[8] = {0, 0, 0, 0, 1, 1, 1, 1}, oy[8] = {0, 1, 0, 1, 0, 1, 0, 1}, oz[8] = {0, 0, 1, 1, 0, 0, 1, 1};
  for(int c = 0; c < 8; c++)
    {
      int x = wrap(sx + ox[c]), y = wrap(sy + oy[c]), z = wrap(sz + oz[c]);
      double w = wx[ox[c]] * wy[oy[c]] * wz[oz[c]];
      double fxv = fac * ((4.0 / 3) * (at(wrap(x - 1), y, z) - at(wrap(x + 1), y, z)) -
                          (1.0 / 6) * (at(wrap(x - 2), y, z) - at(wrap(x + 2), y, z)));
      double fyv = fac * ((4.0 / 3) * (at(x, wrap(y - 1), z) - at(x, wrap(y + 1), z)) -
                          (1.0 / 6) * (at(x, wrap(y - 2), z) - at(x, wrap(y + 2), z)));
      double fzv = fac * ((4.0 / 3) * (at(x, y, wrap(z - 1)) - at(x, y, wrap(z + 1))) -
                          (1.0 / 6) * (at(x, y, wrap(z - 2)) - at(x, y, wrap(z + 2))));
      acc[0] += fxv * w;
      acc[1] += fyv * w;
      acc[2] += fzv * w;
    }
  r_pm[3 * i + 0] = acc[0];
  r_pm[3 * i + 1] = acc[1];
  r_pm[3 * i + 2] = acc[2];
}

// ---------------------------------------------------------------------------------------------------
// Two-pass gather, per target species: (1) the 4-point finite-difference force of every mesh cell, 3 doubles per cell,
// streamed; (2) the CIC gather of 8 cell forces per particle (8 x 24 contiguous bytes instead of 96 scattered potential
// reads).  Same expressions and the same corner order as k_gradient_gather, hence identical results.
// ---------------------------------------------------------------------------------------------------
// The force mesh by marching along x: a block of 8 (y) x 32 (z) threads owns a column of FM_XB planes, every thread keeps
// the five potentials x-2 .. x+2 of its (y, z) in registers, so each potential is read from memory once as the new "x+2" value
// (a one-thread-per-cell kernel re-reads every plane five times from far apart: measured 3.2 GB fetched per species for a
// 1.07 GB mesh); the y and z neighbours are neighbours' lines of the plane just loaded (cache hits).  Same expressions as
// k_gradient_gather.
#define FM_XB 64
__global__ __launch_bounds__(256) void k_force_mesh_march(const double *__restrict__ grid, int N, double fac, double *__restrict__ fm)
{
  const long long NZ = N + 2;
  const int z = blockIdx.x * 32 + (threadIdx.x & 31), y = blockIdx.y * 8 + (threadIdx.x >> 5);
  const int x0 = blockIdx.z * FM_XB;
  if(z >= N || y >= N)
    return;
  auto wrap = [N](int a) { return a < 0 ? a + N : (a >= N ? a - N : a); };
  const int ym1 = wrap(y - 1), yp1 = wrap(y + 1), ym2 = wrap(y - 2), yp2 = wrap(y + 2);
  const int zm1 = wrap(z - 1), zp1 = wrap(z + 1), zm2 = wrap(z - 2), zp2 = wrap(z + 2);
  auto at = [&](int xx, int yy, int zz) { return grid[((long long)xx * N + yy) * NZ + zz]; };
  double pm2 = at(wrap(x0 - 2), y, z), pm1 = at(wrap(x0 - 1), y, z), p0 = at(wrap(x0), y, z), pp1 = at(wrap(x0 + 1), y, z);
  const int x1 = x0 + FM_XB < N ? x0 + FM_XB : N;
  // the three components of a cell are 24 contiguous bytes: a wave (2 rows x 32 cells) turns its 2 x 768 bytes through LDS
  // and stores them as 16-byte pieces, lane after lane (three 8-byte stores at stride 24 touch every line three times)
  __shared__ double stage[4][2][96];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, zl = threadIdx.x & 31, rw = (threadIdx.x >> 5) & 1;
  const bool full = blockIdx.x * 32 + 32 <= N && blockIdx.y * 8 + 8 <= N && (N % 2) == 0;   // whole tile inside the mesh
  for(int x = x0; x < x1; x++)
    {
      const double pp2 = at(wrap(x + 2), y, z);
      const double fy = fac * ((4.0 / 3) * (at(x, ym1, z) - at(x, yp1, z)) - (1.0 / 6) * (at(x, ym2, z) - at(x, yp2, z)));
      const double fz = fac * ((4.0 / 3) * (at(x, y, zm1) - at(x, y, zp1)) - (1.0 / 6) * (at(x, y, zm2) - at(x, y, zp2)));
      const double fx = fac * ((4.0 / 3) * (pm1 - pp1) - (1.0 / 6) * (pm2 - pp2));
      const long long idx = ((long long)x * N + y) * N + z;
      if(full)
        {
          stage[wave][rw][3 * zl + 0] = fx;
          stage[wave][rw][3 * zl + 1] = fy;
          stage[wave][rw][3 * zl + 2] = fz;
          __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
          __builtin_amdgcn_wave_barrier();
          const int yw = blockIdx.y * 8 + 2 * wave;   // first of this wave's two rows
#pragma unroll
          for(int q = lane; q < 96; q += 64)
            {
              const int r = q / 48, o = q - 48 * r;
              const double2 v = *reinterpret_cast<const double2 *>(&stage[wave][r][2 * o]);
              *reinterpret_cast<double2 *>(fm + 3 * (((long long)x * N + yw + r) * N + blockIdx.x * 32) + 2 * o) = v;
            }
          __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
          __builtin_amdgcn_wave_barrier();
        }
      else
        {
          fm[3 * idx + 0] = fx;
          fm[3 * idx + 1] = fy;
          fm[3 * idx + 2] = fz;
        }
      pm2 = pm1;
      pm1 = p0;
      p0 = pp1;
      pp1 = pp2;
    }
}

__global__ void k_gather_force(const double4 *__restrict__ s_pm, const unsigned char *__restrict__ s_type,
                               const unsigned char *__restrict__ s_flag, long long first, long long n, double to_slab, int N,
                               const int *__restrict__ t2g_tab, int species, const double *__restrict__ fm,
                               double *__restrict__ r_pm)
{
  long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if(i >= n)
    return;
  i += first;
  if(s_flag[i] & 2)
    {
      if(species <= 0)
        r_pm[3 * i + 0] = r_pm[3 * i + 1] = r_pm[3 * i + 2] = 0.0;
      return;
    }
  const int g = t2g_tab[s_type[i]];
  if(species >= 0 && g != species)
    return;
  if(species < 0)   // all species' force meshes are resident: [g][N][N][N][3]
    fm += (size_t)g * 3 * N * N * N;
  const double4 p = s_pm[i];
  double dx, dy, dz;
  int sx = cell_of(p.x, to_slab, N, &dx), sy = cell_of(p.y, to_slab, N, &dy), sz = cell_of(p.z, to_slab, N, &dz);
  double wx[2] = {1.0 - dx, dx}, wy[2] = {1.0 - dy, dy}, wz[2] = {1.0 - dz, dz};
  double acc[3] = {0, 0, 0};
  auto wrap = [N](int a) { return a < 0 ? a + N : (a >= N ? a - N : a); };
  // the corners (x, y, z) and (x, y, z + 1) are 48 contiguous bytes (except across the periodic seam): three 16-byte loads per
  // pair instead of six 8-byte ones -- the kernel is bound by the number of scattered load instructions.  Corner order and
  // expressions as in k_gradient_gather: (0,0,0) (0,1,0) (0,0,1) (0,1,1) (1,0,0) ...
  typedef double d2u __attribute__((ext_vector_type(2), aligned(8)));
  const int z1 = wrap(sz + 1);
  const bool zpair = z1 == sz + 1;
  double fv[8][3];
#pragma unroll
  for(int cx = 0; cx < 2; cx++)
#pragma unroll
    for(int cy = 0; cy < 2; cy++)
      {
        const int x = wrap(sx + cx), y = wrap(sy + cy);
        const double *f0 = fm + 3 * (((long long)x * N + y) * N + sz);
        const int c0 = 4 * cx + cy, c1 = 4 * cx + cy + 2;   // the two corners of this (x, y): z offset 0 and 1
        if(zpair)
          {
            const d2u a = *reinterpret_cast<const d2u *>(f0), b_ = *reinterpret_cast<const d2u *>(f0 + 2),
                      c_ = *reinterpret_cast<const d2u *>(f0 + 4);
            fv[c0][0] = a.x;
            fv[c0][1] = a.y;
            fv[c0][2] = b_.x;
            fv[c1][0] = b_.y;
            fv[c1][1] = c_.x;
            fv[c1][2] = c_.y;
          }
        else
          {
            const double *f1 = fm + 3 * (((long long)x * N + y) * N + z1);
#pragma unroll
            for(int d = 0; d < 3; d++)
              {
                fv[c0][d] = f0[d];
                fv[c1][d] = f1[d];
              }
          }
      }
  const int ox[8] = {0, 0, 0, 0, 1, 1, 1, 1}, oy[8] = {0, 1, 0, 1, 0, 1, 0, 1}, oz[8] = {0, 0, 1, 1, 0, 0, 1, 1};
#pragma unroll
  for(int c = 0; c < 8; c++)
    {
      const double w = wx[ox[c]] * wy[oy[c]] * wz[oz[c]];
      acc[0] += fv[c][0] * w;
      acc[1] += fv[c][1] * w;
      acc[2] += fv[c][2] * w;
    }
  r_pm[3 * i + 0] = acc[0];
  r_pm[3 * i + 1] = acc[1];
  r_pm[3 * i + 2] = acc[2];
}

// ---------------------------------------------------------------------------------------------------
// Tiled variants.  The particles are Peano-sorted and the tree node of a level-Lt cell owns one contiguous
// particle range, so one workgroup per node can deposit into / gather from an LDS copy of the mesh patch
// the cell touches: the 8 scattered fp64 global atomics per particle (64 lanes -> 64 rows, the slow atomic
// shape) become LDS atomics plus one coalesced atomic flush per patch row, and the gather's 72 scattered L2
// reads per particle become LDS reads.  Particles that are direct children of nodes above level Lt (sparse
// regions) go through the per-particle kernels.
// ---------------------------------------------------------------------------------------------------
#define PM_DT 18                     // deposit patch edge: 16-cell node + CIC neighbour + 1 slack
#define PM_GT (PM_DT + 4)            // gather patch edge: +-2 for the 4-point gradient
#define PM_TILE_THREADS 1024


template <int NG, int DT>
__global__ __launch_bounds__(DT > 10 ? PM_TILE_THREADS : 256) void k_cic_deposit_tiled(
    const double4 *__restrict__ s_pm, const unsigned char *__restrict__ s_type, const unsigned char *__restrict__ s_flag,
    const int *__restrict__ n_first, const int *__restrict__ n_count, const double4 *__restrict__ n_geo, int node0,
    double to_slab, int N, const int *__restrict__ t2g_tab, double *__restrict__ rho, MeshAddr ma)
{
  extern __shared__ double tile[];   // [NG][DT][DT][DT]
  const int node = node0 + blockIdx.x;
  const double4 geo = n_geo[node];
  const int first = n_first[node], count = n_count[node];
  int o[3];
  {
    const double h = 0.5 * geo.w;
    o[0] = (int)floor((geo.x - h) * to_slab);
    o[1] = (int)floor((geo.y - h) * to_slab);
    o[2] = (int)floor((geo.z - h) * to_slab);
  }
  for(int t = threadIdx.x; t < NG * DT * DT * DT; t += blockDim.x)
    tile[t] = 0.0;
  __syncthreads();
  for(int k = threadIdx.x; k < count; k += blockDim.x)
    {
      const int i = first + k;
      if(s_flag[i] & 2)
        continue;
      const double4 p = s_pm[i];
      const int g = t2g_tab[s_type[i]];
      double dx, dy, dz;
      const int sx = cell_of(p.x, to_slab, N, &dx), sy = cell_of(p.y, to_slab, N, &dy), sz = cell_of(p.z, to_slab, N, &dz);
      const int lx = sx - o[0], ly = sy - o[1], lz = sz - o[2];
      const double m = p.w;
      const double w[8] = {m * (1.0 - dx) * (1.0 - dy) * (1.0 - dz), m * (1.0 - dx) * dy * (1.0 - dz),
                           m * (1.0 - dx) * (1.0 - dy) * dz,         m * (1.0 - dx) * dy * dz,
                           m * (dx) * (1.0 - dy) * (1.0 - dz),       m * (dx)*dy * (1.0 - dz),
                           m * (dx) * (1.0 - dy) * dz,               m * (dx)*dy * dz};   // pm_periodic.c:322-329
      const int ox[8] = {0, 0, 0, 0, 1, 1, 1, 1}, oy[8] = {0, 1, 0, 1, 0, 1, 0, 1}, oz[8] = {0, 0, 1, 1, 0, 0, 1, 1};
      if(lx >= 0 && ly >= 0 && lz >= 0 && lx < DT - 1 && ly < DT - 1 && lz < DT - 1)
        {
          double *tg = tile + (size_t)g * DT * DT * DT;
#pragma unroll
          for(int c = 0; c < 8; c++)
            atomicAdd(&tg[((lx + ox[c]) * DT + (ly + oy[c])) * DT + (lz + oz[c])], w[c]);
        }
      else
        {
          double *grid = rho + (size_t)g * ma.species_stride();   // outside the patch (a cell wider than the patch allows for): direct
#pragma unroll
          for(int c = 0; c < 8; c++)
            atomicAdd(&grid[ma.cell(sx + ox[c], sy + oy[c], sz + oz[c])], w[c]);
        }
    }
  __syncthreads();
  for(int t = threadIdx.x; t < NG * DT * DT * DT; t += blockDim.x)
    {
      const double v = tile[t];
      if(v != 0.0)
        {
          const int g = t / (DT * DT * DT), r = t % (DT * DT * DT);
          const int lx = r / (DT * DT), ly = (r / DT) % DT, lz = r % DT;
          atomicAdd(&rho[(size_t)g * ma.species_stride() + ma.cell(o[0] + lx, o[1] + ly, o[2] + lz)], v);
        }
    }
}

// particles that hang directly off nodes above the tile level
__global__ void k_cic_deposit_loose(const double4 *__restrict__ s_pm, const unsigned char *__restrict__ s_type,
                                    const unsigned char *__restrict__ s_flag,
                                    const int *__restrict__ n_child, int nnodes_above, double to_slab, int N,
                                    const int *__restrict__ t2g_tab, double *__restrict__ rho, MeshAddr ma)
{
  long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if(t >= 8ll * nnodes_above)
    return;
  const int c = n_child[t];
  if(c > -2)
    return;
  const int i = -2 - c;
  if(s_flag[i] & 2)
    return;
  double4 p = s_pm[i];
  int g = t2g_tab[s_type[i]];
  double *grid = rho + (size_t)g * ma.species_stride();
  double dx, dy, dz;
  int sx = cell_of(p.x, to_slab, N, &dx), sy = cell_of(p.y, to_slab, N, &dy), sz = cell_of(p.z, to_slab, N, &dz);
  double m = p.w;
  atomicAdd(&grid[ma.cell(sx, sy, sz)], m * (1.0 - dx) * (1.0 - dy) * (1.0 - dz));
  atomicAdd(&grid[ma.cell(sx, sy + 1, sz)], m * (1.0 - dx) * dy * (1.0 - dz));
  atomicAdd(&grid[ma.cell(sx, sy, sz + 1)], m * (1.0 - dx) * (1.0 - dy) * dz);
  atomicAdd(&grid[ma.cell(sx, sy + 1, sz + 1)], m * (1.0 - dx) * dy * dz);
  atomicAdd(&grid[ma.cell(sx + 1, sy, sz)], m * (dx) * (1.0 - dy) * (1.0 - dz));
  atomicAdd(&grid[ma.cell(sx + 1, sy + 1, sz)], m * (dx)*dy * (1.0 - dz));
  atomicAdd(&grid[ma.cell(sx + 1, sy, sz + 1)], m * (dx) * (1.0 - dy) * dz);
  atomicAdd(&grid[ma.cell(sx + 1, sy + 1, sz + 1)], m * (dx)*dy * dz);
}

// gather: one workgroup per node of the level whose cells are at most 8 mesh cells wide; the potential patches of
// ALL species (edge 8 + misalignment + CIC neighbour + 2x2 gradient halo = 15) sit in LDS, one thread per particle
#define PM_G8 15
template <int NG>
__global__ __launch_bounds__(256) void k_gradient_gather_tiled(
    const double4 *__restrict__ s_pm, const unsigned char *__restrict__ s_type, const unsigned char *__restrict__ s_flag,
    const int *__restrict__ n_first, const int *__restrict__ n_count, const double4 *__restrict__ n_geo, int node0,
    long long shard_first, long long shard_count, double to_slab, int N, const int *__restrict__ t2g_tab,
    const double *__restrict__ phi, double fac, double *__restrict__ r_pm)
{
  extern __shared__ double tile[];   // [NG][G8][G8][G8]
  const int node = node0 + blockIdx.x;
  const int first = n_first[node], count = n_count[node];
  if((long long)first + count <= shard_first || (long long)first >= shard_first + shard_count)
    return;                                     // no target of this task in the cell (uniform per block)
  const double4 geo = n_geo[node];
  const long long NZ = N + 2;
  int o[3];
  {
    const double h = 0.5 * geo.w;
    o[0] = (int)floor((geo.x - h) * to_slab) - 2;
    o[1] = (int)floor((geo.y - h) * to_slab) - 2;
    o[2] = (int)floor((geo.z - h) * to_slab) - 2;
  }
  constexpr int T3 = PM_G8 * PM_G8 * PM_G8;
  for(int t = threadIdx.x; t < NG * T3; t += blockDim.x)
    {
      const int g = t / T3, r = t - g * T3;
      const int lx = r / (PM_G8 * PM_G8), ly = (r / PM_G8) % PM_G8, lz = r % PM_G8;
      tile[t] = phi[(size_t)g * N * N * NZ + ((long long)wrapN(o[0] + lx, N) * N + wrapN(o[1] + ly, N)) * NZ + wrapN(o[2] + lz, N)];
    }
  __syncthreads();
  for(int k = threadIdx.x; k < count; k += blockDim.x)
    {
      const long long i = (long long)first + k;
      if(i < shard_first || i >= shard_first + shard_count)
        continue;
      if(s_flag[i] & 2)
        {
          r_pm[3 * i + 0] = r_pm[3 * i + 1] = r_pm[3 * i + 2] = 0.0;
          continue;
        }
      const int g = t2g_tab[s_type[i]];
      const double *grid = phi + (size_t)g * N * N * NZ;
      const double *tg = tile + (size_t)g * T3;
      const double4 p = s_pm[i];
      double dx, dy, dz;
      const int sx = cell_of(p.x, to_slab, N, &dx), sy = cell_of(p.y, to_slab, N, &dy), sz = cell_of(p.z, to_slab, N, &dz);
      const int lx = sx - o[0], ly = sy - o[1], lz = sz - o[2];
      const bool inside = lx >= 2 && ly >= 2 && lz >= 2 && lx < PM_G8 - 3 && ly < PM_G8 - 3 && lz < PM_G8 - 3;
      const double wx[2] = {1.0 - dx, dx}, wy[2] = {1.0 - dy, dy}, wz[2] = {1.0 - dz, dz};
      double acc[3] = {0, 0, 0};
      const int ox[8] = {0, 0, 0, 0, 1, 1, 1, 1}, oy[8] = {0, 1, 0, 1, 0, 1, 0, 1}, oz[8] = {0, 0, 1, 1, 0, 0, 1, 1};
      auto at = [&](int x, int y, int z) -> double {
        if(inside)
          return tg[(x * PM_G8 + y) * PM_G8 + z];
        return grid[((long long)wrapN(o[0] + x, N) * N + wrapN(o[1] + y, N)) * NZ + wrapN(o[2] + z, N)];
      };
      for(int c = 0; c < 8; c++)
        {
          const int x = lx + ox[c], y = ly + oy[c], z = lz + oz[c];
          const double w = wx[ox[c]] * wy[oy[c]] * wz[oz[c]];
          const double fxv = fac * ((4.0 / 3) * (at(x - 1, y, z) - at(x + 1, y, z)) - (1.0 / 6) * (at(x - 2, y, z) - at(x + 2, y, z)));
          const double fyv = fac * ((4.0 / 3) * (at(x, y - 1, z) - at(x, y + 1, z)) - (1.0 / 6) * (at(x, y - 2, z) - at(x, y + 2, z)));
          const double fzv = fac * ((4.0 / 3) * (at(x, y, z - 1) - at(x, y, z + 1)) - (1.0 / 6) * (at(x, y, z - 2) - at(x, y, z + 2)));
          acc[0] += fxv * w;
          acc[1] += fyv * w;
          acc[2] += fzv * w;
        }
      r_pm[3 * i + 0] = acc[0];
      r_pm[3 * i + 1] = acc[1];
      r_pm[3 * i + 2] = acc[2];
    }
}

__global__ void k_gradient_gather_loose(const double4 *__restrict__ s_pm, const unsigned char *__restrict__ s_type,
                                        const unsigned char *__restrict__ s_flag,
                                        const int *__restrict__ n_child, int nnodes_above, long long shard_first,
                                        long long shard_count, double to_slab, int N, const int *__restrict__ t2g_tab,
                                        const double *__restrict__ phi, double fac, double *__restrict__ r_pm)
{
  long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if(t >= 8ll * nnodes_above)
    return;
  const int c = n_child[t];
  if(c > -2)
    return;
  const long long i = -2 - c;
  if(i < shard_first || i >= shard_first + shard_count)
    return;
  if(s_flag[i] & 2)
    {
      r_pm[3 * i + 0] = r_pm[3 * i + 1] = r_pm[3 * i + 2] = 0.0;
      return;
    }
  double4 p = s_pm[i];
  int g = t2g_tab[s_type[i]];
  const long long NZ = N + 2;
  const double *grid = phi + (size_t)g * N * N * NZ;
  double dx, dy, dz;
  int sx = cell_of(p.x, to_slab, N, &dx), sy = cell_of(p.y, to_slab, N, &dy), sz = cell_of(p.z, to_slab, N, &dz);
  double wx[2] = {1.0 - dx, dx}, wy[2] = {1.0 - dy, dy}, wz[2] = {1.0 - dz, dz};
  double acc[3] = {0, 0, 0};
  auto at = [&](int x, int y, int z) { return grid[((long long)wrapN(x, N) * N + wrapN(y, N)) * NZ + wrapN(z, N)]; };
  const int ox[8] = {0, 0, 0, 0, 1, 1, 1, 1}, oy[8] = {0, 1, 0, 1, 0, 1, 0, 1}, oz[8] = {0, 0, 1, 1, 0, 0, 1, 1};
  for(int cc = 0; cc < 8; cc++)
    {
      int x = sx + ox[cc], y = sy + oy[cc], z = sz + oz[cc];
      double w = wx[ox[cc]] * wy[oy[cc]] * wz[oz[cc]];
      double fxv = fac * ((4.0 / 3) * (at(x - 1, y, z) - at(x + 1, y, z)) - (1.0 / 6) * (at(x - 2, y, z) - at(x + 2, y, z)));
      double fyv = fac * ((4.0 / 3) * (at(x, y - 1, z) - at(x, y + 1, z)) - (1.0 / 6) * (at(x, y - 2, z) - at(x, y + 2, z)));
      double fzv = fac * ((4.0 / 3) * (at(x, y, z - 1) - at(x, y, z + 1)) - (1.0 / 6) * (at(x, y, z - 2) - at(x, y, z + 2)));
      acc[0] += fxv * w;
      acc[1] += fyv * w;
      acc[2] += fzv * w;
    }
  r_pm[3 * i + 0] = acc[0];
  r_pm[3 * i + 1] = acc[1];
  r_pm[3 * i + 2] = acc[2];
}

// the level whose cells are at most 16 mesh cells wide (-1: no such level in this tree -> per-particle kernels)
static int pm_tile_level(const ngravs_ctx *c, double to_slab, double max_cells = 16.0)
{
  if(!c->have_tree || c->tune.pm_notile)
    return -1;
  // (the tree's root cube is 1.001 x the particle extent, domain.c:909-923: a "16-cell" node of a full box is 16.016 cells wide.
  // The patches have one point of slack for that; the rare particle beyond it takes the direct path.)
  double len = c->dom[6];
  for(int l = 0; l < c->nlevels && l < TREE_BITS; l++, len *= 0.5)
    if(len * to_slab <= max_cells * 1.002)
      return (c->level_start[l + 1] - c->level_start[l] > 0) ? l : -1;
  return -1;
}

void pm_release(ngravs_ctx *c)
{
  if(c->fft_fwd)
    {
      hipfftDestroy(*(hipfftHandle *)c->fft_fwd);
      delete(hipfftHandle *)c->fft_fwd;
      c->fft_fwd = nullptr;
    }
  if(c->fft_inv)
    {
      hipfftDestroy(*(hipfftHandle *)c->fft_inv);
      delete(hipfftHandle *)c->fft_inv;
      c->fft_inv = nullptr;
    }
  c->pm_plan_n = 0;
}

// CIC deposit of the working set's own particles into `dst` (zeroed by the caller), a full mesh or a brick (MeshAddr), by tiles:
// tree cells at most 16 mesh cells wide (18^3 patch per species, one 1024-thread workgroup per CU), or -- tuning "pm_tile8" -- at
// most 8 (10^3 patches, 256 threads).  Returns 1 if the tree has no such level (caller deposits per particle), 0, or an error.
// Expects the type -> species table in d_counters[8..13].
int pm_deposit_tiles(ngravs_ctx *c, const MeshAddr &ma, double *dst)
{
  const int N = c->cfg.pmgrid, ng = c->cfg.n_gravs;
  const double to_slab = N / c->cfg.box_size;
  const int bs = 256;
  const bool t8 = c->tune.pm_tile8 != 0;
  const int tl = pm_tile_level(c, to_slab, t8 ? 8.0 : 16.0);
  if(tl < 0)
    return 1;
  const long long tl0 = c->level_start[tl], tln = c->level_start[tl + 1] - tl0;
  const int dt = t8 ? 10 : PM_DT;
  const size_t lds = sizeof(double) * ng * dt * dt * dt;
  auto launch_dep = [&](auto kern) -> int {
    HIP_TRY(c, hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3((unsigned)tln), dim3(t8 ? 256 : PM_TILE_THREADS), lds, c->stream, c->s_pm.p, c->s_type.p, c->s_active.p,
                       c->n_first.p, c->n_count.p, c->n_geo.p, (int)tl0, to_slab, N, c->d_counters.p + 8, dst, ma);
    return NGRAVS_OK;
  };
  int rc;
  if(t8)
    rc = ng == 1 ? launch_dep(k_cic_deposit_tiled<1, 10>) : (ng == 2 ? launch_dep(k_cic_deposit_tiled<2, 10>) : launch_dep(k_cic_deposit_tiled<3, 10>));
  else
    rc = ng == 1 ? launch_dep(k_cic_deposit_tiled<1, PM_DT>)
                 : (ng == 2 ? launch_dep(k_cic_deposit_tiled<2, PM_DT>) : launch_dep(k_cic_deposit_tiled<3, PM_DT>));
  if(rc)
    return rc;
  if(tl0 > 0)
    hipLaunchKernelGGL(k_cic_deposit_loose, dim3((unsigned)((8 * tl0 + bs - 1) / bs)), dim3(bs), 0, c->stream, c->s_pm.p, c->s_type.p,
                       c->s_active.p, c->n_child.p, (int)tl0, to_slab, N, c->d_counters.p + 8, dst, ma);
  HIP_TRY(c, hipGetLastError());
  return NGRAVS_OK;
}

int pm_deposit(ngravs_ctx *c)
{
  const int N = c->cfg.pmgrid, ng = c->cfg.n_gravs;
  const long long n = c->n;
  if(N <= 0 || !c->cfg.periodic)
    {
      ngravs_report(c, NGRAVS_ERR_ARG, "pmforce_periodic needs PERIODIC and PMGRID");
      return NGRAVS_ERR_ARG;
    }
  const size_t real_elems = (size_t)N * N * (N + 2);
  if(c->pm_rho.ensure(real_elems * ng) || c->pm_phi.ensure(real_elems * ng) || c->r_pm.ensure(3 * n) || c->d_counters.ensure(16))
    return NGRAVS_ERR_NOMEM;
  if(c->pm_plan_n != N)
    {
      pm_release(c);
      c->fft_fwd = new hipfftHandle;
      c->fft_inv = new hipfftHandle;
      FFT_TRY(c, hipfftPlan3d((hipfftHandle *)c->fft_fwd, N, N, N, HIPFFT_D2Z));
      FFT_TRY(c, hipfftPlan3d((hipfftHandle *)c->fft_inv, N, N, N, HIPFFT_Z2D));
      FFT_TRY(c, hipfftSetStream(*(hipfftHandle *)c->fft_fwd, c->stream));
      FFT_TRY(c, hipfftSetStream(*(hipfftHandle *)c->fft_inv, c->stream));
      c->pm_plan_n = N;
    }
  // type -> species table on the device (6 ints, lives in d_counters[8..13])
  HIP_TRY(c, hipMemcpyAsync(c->d_counters.p + 8, c->cfg.type_to_grav, sizeof(int) * 6, hipMemcpyHostToDevice, c->stream));
  const double L = c->cfg.box_size, to_slab = N / L;
  HIP_TRY(c, hipMemsetAsync(c->pm_rho.p, 0, sizeof(double) * real_elems * ng, c->stream));
  const int bs = 256;
  unsigned nbp = (unsigned)((n + bs - 1) / bs);
  MeshAddr ma;
  ma.N = N;
  ma.brick = 0;
  for(int j = 0; j < 3; j++)
    {
      ma.lo[j] = 0;
      ma.ext[j] = N;
    }
  int rct = pm_deposit_tiles(c, ma, c->pm_rho.p);
  if(rct == 1)   // no tree level to tile by: one thread per particle
    hipLaunchKernelGGL(k_cic_deposit, dim3(nbp), dim3(bs), 0, c->stream, c->s_pm.p, c->s_type.p, c->s_active.p, n, to_slab, N, ng,
                       c->d_counters.p + 8, c->pm_rho.p);
  else if(rct)
    return rct;
  HIP_TRY(c, hipGetLastError());
  return NGRAVS_OK;
}

// forward FFTs, Green multiply, inverse FFTs, gradient + gather.  Between pm_deposit() and pm_finish() a multi-task
// host all-reduces the density mesh (pm_periodic.c:333-427 ships patches to slab owners instead).
int pm_finish(ngravs_ctx *c)
{
  const int N = c->cfg.pmgrid, ng = c->cfg.n_gravs;
  const long long n = c->n;
  const size_t real_elems = (size_t)N * N * (N + 2);
  const double L = c->cfg.box_size, to_slab = N / L;
  const int bs = 256;
  const int tl = pm_tile_level(c, to_slab);
  const long long tl0 = tl >= 0 ? c->level_start[tl] : 0, tln = tl >= 0 ? c->level_start[tl + 1] - tl0 : 0;
  (void)tln;
  for(int a = 0; a < ng; a++)
    FFT_TRY(c, hipfftExecD2Z(*(hipfftHandle *)c->fft_fwd, c->pm_rho.p + real_elems * a,
                             (hipfftDoubleComplex *)(c->pm_rho.p + real_elems * a)));
  GreenParams gp;
  make_green_params(c, &gp);
  long long modes = (long long)N * N * (N / 2 + 1);
  unsigned nbm = (unsigned)((modes + bs - 1) / bs);
  switch(ng)
    {
    case 1:
      hipLaunchKernelGGL(k_green<1>, dim3(nbm), dim3(bs), 0, c->stream, (const double2 *)c->pm_rho.p, (double2 *)c->pm_phi.p, gp);
      break;
    case 2:
      hipLaunchKernelGGL(k_green<2>, dim3(nbm), dim3(bs), 0, c->stream, (const double2 *)c->pm_rho.p, (double2 *)c->pm_phi.p, gp);
      break;
    default:
      hipLaunchKernelGGL(k_green<3>, dim3(nbm), dim3(bs), 0, c->stream, (const double2 *)c->pm_rho.p, (double2 *)c->pm_phi.p, gp);
      break;
    }
  for(int b = 0; b < ng; b++)
    FFT_TRY(c, hipfftExecZ2D(*(hipfftHandle *)c->fft_inv, (hipfftDoubleComplex *)(c->pm_phi.p + real_elems * b),
                             c->pm_phi.p + real_elems * b));
  double fac = c->cfg.G / (M_PI * L);      // pm_periodic.c:237-238
  fac *= 1 / (2 * L / N);
  // GravPM is needed for this task's own particles only (its target shard); the rest stays zero
  if(c->cfg.world_size > 1)
    HIP_TRY(c, hipMemsetAsync(c->r_pm.p, 0, sizeof(double) * 3 * n, c->stream));
  unsigned nbg = (unsigned)((c->shard_count + bs - 1) / bs);
  // tiled gather (cells at most 8 mesh cells wide, all species' potential patches in LDS): measured at C4 it LOSES to the
  // two-pass gather below (13 ms against 3.0 + 3.3 ms: 15^3 patches for 8^3 cells re-read the mesh 6.6 times in short rows),
  // so it is opt-in for tuning only
  const int gl = c->tune.pm_tile_gather ? pm_tile_level(c, to_slab, 8.0) : -1;
  if(gl >= 0)
    {
      const long long gl0 = c->level_start[gl], gln = c->level_start[gl + 1] - gl0;
      const size_t lds = sizeof(double) * ng * PM_G8 * PM_G8 * PM_G8;
      auto launch_gat = [&](auto kern) -> int {
        HIP_TRY(c, hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(kern, dim3((unsigned)gln), dim3(256), lds, c->stream, c->s_pm.p, c->s_type.p, c->s_active.p, c->n_first.p,
                           c->n_count.p, c->n_geo.p, (int)gl0, (long long)c->shard_first, (long long)c->shard_count, to_slab, N,
                           c->d_counters.p + 8, c->pm_phi.p, fac, c->r_pm.p);
        return NGRAVS_OK;
      };
      int rc = ng == 1 ? launch_gat(k_gradient_gather_tiled<1>) : (ng == 2 ? launch_gat(k_gradient_gather_tiled<2>) : launch_gat(k_gradient_gather_tiled<3>));
      if(rc)
        return rc;
      if(gl0 > 0)
        hipLaunchKernelGGL(k_gradient_gather_loose, dim3((unsigned)((8 * gl0 + bs - 1) / bs)), dim3(bs), 0, c->stream, c->s_pm.p,
                           c->s_type.p, c->s_active.p, c->n_child.p, (int)gl0, (long long)c->shard_first, (long long)c->shard_count,
                           to_slab, N, c->d_counters.p + 8, c->pm_phi.p, fac, c->r_pm.p);
    }
  else if(nbg > 0 && c->cfg.world_size <= 2 && !c->tune.pm_fused_gather)
    {
      // two passes per target species (see k_force_mesh_march); with many tasks the (replicated) force-mesh pass would cost more
      // than the sharded fused gather below
      const long long NN = (long long)N * N * N;
      // all species' force meshes at once (one gather pass) when the device has the room beside what the walk will ask for:
      // C4 6.4 GB; C5 77 GB, which a 288 GB device holds
      const size_t fm_all = (size_t)3 * NN * ng * sizeof(double), fm_have = c->pm_force.cap * sizeof(double);
      size_t free_b = 0, total_b = 0;
      bool all_resident = fm_all <= ((size_t)16 << 30) || fm_have >= fm_all;
      if(!all_resident && hipMemGetInfo(&free_b, &total_b) == hipSuccess)
        all_resident = free_b + fm_have >= fm_all + ((size_t)48 << 30);
      if(c->pm_force.ensure((size_t)(3 * NN) * (all_resident ? ng : 1)))
        return NGRAVS_ERR_NOMEM;
      if(all_resident)
        {
          for(int b = 0; b < ng; b++)
            hipLaunchKernelGGL(k_force_mesh_march, dim3((N + 31) / 32, (N + 7) / 8, (N + FM_XB - 1) / FM_XB), dim3(256), 0, c->stream,
                               c->pm_phi.p + real_elems * b, N, fac, c->pm_force.p + (size_t)3 * NN * b);
          hipLaunchKernelGGL(k_gather_force, dim3(nbg), dim3(bs), 0, c->stream, c->s_pm.p, c->s_type.p, c->s_active.p,
                             (long long)c->shard_first, (long long)c->shard_count, to_slab, N, c->d_counters.p + 8, -1, c->pm_force.p,
                             c->r_pm.p);
        }
      else
      for(int b = 0; b < ng; b++)
        {
          hipLaunchKernelGGL(k_force_mesh_march, dim3((N + 31) / 32, (N + 7) / 8, (N + FM_XB - 1) / FM_XB), dim3(256), 0, c->stream,
                             c->pm_phi.p + real_elems * b, N, fac, c->pm_force.p);
          hipLaunchKernelGGL(k_gather_force, dim3(nbg), dim3(bs), 0, c->stream, c->s_pm.p, c->s_type.p, c->s_active.p,
                             (long long)c->shard_first, (long long)c->shard_count, to_slab, N, c->d_counters.p + 8, b, c->pm_force.p,
                             c->r_pm.p);
        }
    }
  else if(nbg > 0)
    hipLaunchKernelGGL(k_gradient_gather, dim3(nbg), dim3(bs), 0, c->stream, c->s_pm.p, c->s_type.p, c->s_active.p, (long long)c->shard_first,
                       (long long)c->shard_count, to_slab, N, c->d_counters.p + 8, c->pm_phi.p, fac, c->r_pm.p);
  HIP_TRY(c, hipGetLastError());
  c->have_pm = true;
  return NGRAVS_OK;
}

int pm_run(ngravs_ctx *c)
{
  int rc = pm_deposit(c);
  if(rc)
    return rc;
  return pm_finish(c);
}
