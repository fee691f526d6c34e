// kernels_pmslab.hip -- pmforce_periodic for more than one task: the x-slab decomposed mesh.
//
// Replaces (reference): the slab tables of pm_init_periodic (pm_periodic.c:74-123), the shipping of every task's density
// patch to the slab owners (:336-427), rfftwnd_mpi in FFTW_TRANSPOSED_ORDER (:433, :525), the Green's function on the
// transposed layout (:436-520) and the potential bricks with ghost planes sent back to the particle owners (:529-670),
// followed by finite differences and the CIC gather on the local brick (:681-763).
//
// MI355X design: nothing is replicated and no mesh is all-reduced.  A task deposits its OWN particles into a brick (the
// bounding box of their CIC clouds in mesh cells), and four all-to-all-v exchanges move exactly the planes that are needed:
//   stage 0  density planes            brick -> owner of the x-slab (accumulated)      | then 2-D r2c FFTs of the own planes
//   stage 1  k-space transpose x<->y   [x-slab][y][kz] -> [y-slab][kz][x]              | then 1-D FFTs along x, Green, inverse 1-D
//   stage 2  transpose back                                                            | then 2-D c2r FFTs
//   stage 3  potential planes + 2 ghost cells per side  slab owner -> brick            | then 4-point gradient + CIC gather
// The library packs and unpacks (device buffers it owns); the HOST moves the bytes (ngravs_host.c: MPI or RCCL through a
// vtable), one hipStream, no host copy of mesh data.  Per task the mesh memory is NG x (brick + 2 slabs + exchange buffers),
// i.e. proportional to 1/world for compact domains.  As in the single-task path all species are transformed once
// (2 NG transforms instead of the reference's 2 NG^2).
#include "engine.hpp"
#include "pm_common.hpp"
#include <algorithm>

static inline int slab_start(int r, int W, int N) { return (int)(((long long)r * N) / W); }

__device__ __forceinline__ int wrapd(int d, int N)
{
  d += d < 0 ? N : 0;
  d -= d >= N ? N : 0;
  return d;
}

// ---- brick of the own particles --------------------------------------------------------------------------------------
__global__ void k_cell_minmax(const double4 *__restrict__ s_pm, const unsigned char *__restrict__ s_flag, long long n, double to_slab,
                              int N, int *__restrict__ out)
{
  int lo[3] = {1 << 30, 1 << 30, 1 << 30}, hi[3] = {-1, -1, -1};
  for(long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
    {
      if(s_flag[i] & 2)
        continue;
      const double4 p = s_pm[i];
      double f;
      const int c[3] = {cell_of(p.x, to_slab, N, &f), cell_of(p.y, to_slab, N, &f), cell_of(p.z, to_slab, N, &f)};
      for(int j = 0; j < 3; j++)
        {
          lo[j] = c[j] < lo[j] ? c[j] : lo[j];
          hi[j] = c[j] > hi[j] ? c[j] : hi[j];
        }
    }
  for(int j = 0; j < 3; j++)
    {
      for(int off = 32; off > 0; off >>= 1)
        {
          const int a = __shfl_down(lo[j], off), b = __shfl_down(hi[j], off);
          lo[j] = a < lo[j] ? a : lo[j];
          hi[j] = b > hi[j] ? b : hi[j];
        }
    }
  // one value per block and bound, and only where it moves the bound: the six words are shared by the whole grid
  __shared__ int sl[4][3], sh[4][3];
  const int w = threadIdx.x >> 6;
  if((threadIdx.x & 63) == 0)
    for(int j = 0; j < 3; j++)
      {
        sl[w][j] = lo[j];
        sh[w][j] = hi[j];
      }
  __syncthreads();
  if(threadIdx.x < 3)
    {
      const int j = threadIdx.x;
      int a = sl[0][j], b = sh[0][j];
      for(int q = 1; q < (int)(blockDim.x >> 6); q++)
        {
          a = sl[q][j] < a ? sl[q][j] : a;
          b = sh[q][j] > b ? sh[q][j] : b;
        }
      if(a < __atomic_load_n(&out[j], __ATOMIC_RELAXED))
        atomicMin(&out[j], a);
      if(b > __atomic_load_n(&out[3 + j], __ATOMIC_RELAXED))
        atomicMax(&out[3 + j], b);
    }
}

struct Brick
{
  int lo[3], ext[3], N;
  __device__ __forceinline__ long long at(int x, int y, int z) const   // mesh cell -> offset inside one species' brick
  {
    return ((long long)wrapd(x - lo[0], N) * ext[1] + wrapd(y - lo[1], N)) * ext[2] + wrapd(z - lo[2], N);
  }
  __host__ __device__ long long cells() const { return (long long)ext[0] * ext[1] * ext[2]; }
};

__global__ void k_cic_deposit_brick(const double4 *__restrict__ s_pm, const unsigned char *__restrict__ s_type,
                                    const unsigned char *__restrict__ s_flag, long long n, double to_slab, WalkParams wp, Brick B,
                                    double *__restrict__ brick)
{
  long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if(i >= n || (s_flag[i] & 2))   // halo copies are deposited by their owner
    return;
  const double4 p = s_pm[i];
  double *grid = brick + (size_t)wp.t2g[s_type[i]] * B.cells();
  double dx, dy, dz;
  const int sx = cell_of(p.x, to_slab, B.N, &dx), sy = cell_of(p.y, to_slab, B.N, &dy), sz = cell_of(p.z, to_slab, B.N, &dz);
  const double m = p.w;
  // pm_periodic.c:322-329 (same weights, same products)
  atomicAdd(&grid[B.at(sx, sy, sz)], m * (1.0 - dx) * (1.0 - dy) * (1.0 - dz));
  atomicAdd(&grid[B.at(sx, sy + 1, sz)], m * (1.0 - dx) * dy * (1.0 - dz));
  atomicAdd(&grid[B.at(sx, sy, sz + 1)], m * (1.0 - dx) * (1.0 - dy) * dz);
  atomicAdd(&grid[B.at(sx, sy + 1, sz + 1)], m * (1.0 - dx) * dy * dz);
  atomicAdd(&grid[B.at(sx + 1, sy, sz)], m * (dx) * (1.0 - dy) * (1.0 - dz));
  atomicAdd(&grid[B.at(sx + 1, sy + 1, sz)], m * (dx)*dy * (1.0 - dz));
  atomicAdd(&grid[B.at(sx + 1, sy, sz + 1)], m * (dx) * (1.0 - dy) * dz);
  atomicAdd(&grid[B.at(sx + 1, sy + 1, sz + 1)], m * (dx)*dy * dz);
}

// ---- stage 0: density planes brick -> slab ------------------------------------------------------------------------------
// desc (4 long long per brick plane t, ordered by destination): plane index in the brick, position q among the planes of its
// destination, number of planes np of that destination, offset (doubles) of the destination's block in the send buffer
__global__ void k_pack_planes(const double *__restrict__ brick, const long long *__restrict__ desc, int nplanes, int ng, int ey, int ez,
                              long long bcells, double *__restrict__ send)
{
  const long long per = (long long)ey * ez;
  const long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if(idx >= per * nplanes * ng)
    return;
  const long long r = idx % per;
  const int t = (int)((idx / per) % nplanes), g = (int)(idx / (per * nplanes));
  const long long pi = desc[4 * t], q = desc[4 * t + 1], np = desc[4 * t + 2], off = desc[4 * t + 3];
  send[off + (g * np + q) * per + r] = brick[(size_t)g * bcells + pi * per + r];
}

// one source task per launch (no two of its elements hit the same slab cell): xl[q] = local slab plane of the q-th received plane
__global__ void k_add_planes(const double *__restrict__ recv, const long long *__restrict__ xl, int np, int ng, int ey, int ez, int loy,
                             int loz, int N, long long slab_cells, double *__restrict__ slab)
{
  const long long per = (long long)ey * ez;
  const long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if(idx >= per * np * ng)
    return;
  const int k = (int)(idx % ez), j = (int)((idx / ez) % ey);
  const int q = (int)((idx / per) % np), g = (int)(idx / (per * np));
  int y = loy + j, z = loz + k;
  y -= y >= N ? N : 0;
  z -= z >= N ? N : 0;
  slab[(size_t)g * slab_cells + ((long long)xl[q] * N + y) * (N + 2) + z] += recv[idx];
}

// ---- stages 1 / 2: k-space transposes -------------------------------------------------------------------------------------
// tab: [0..N) owner of y, [N..2N) owner of x, then per task: ys, ny, xs, nx, send offset, recv offset (complex units)
#define T_YOWN(tab, y) ((int)(tab)[(y)])
#define T_XOWN(tab, x) ((int)(tab)[N + (x)])
#define T_R(tab, r, k) ((tab)[2 * N + 6 * (r) + (k)])

// C[g][ix][y][z] (own x-slab, z < NH) -> blocks [g][ix][y - ys_d][z] per destination d = owner of y
__global__ void k_tr_pack_fwd(const double2 *__restrict__ C, const long long *__restrict__ tab, int N, int NH, int nx, int ng,
                              double2 *__restrict__ send)
{
  const long long tot = (long long)ng * nx * N * NH;
  const long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if(idx >= tot)
    return;
  const int z = (int)(idx % NH), y = (int)((idx / NH) % N), ix = (int)((idx / ((long long)NH * N)) % nx);
  const int g = (int)(idx / ((long long)NH * N * nx));
  const int d = T_YOWN(tab, y);
  const long long ysd = T_R(tab, d, 0), nyd = T_R(tab, d, 1), off = T_R(tab, d, 4);
  send[off + (((long long)g * nx + ix) * nyd + (y - ysd)) * NH + z] = C[idx];
}
// The two strided halves of the transposes go through a 32 x 32 tile of LDS: both the (x slow, z fast) blocks of the exchange
// and the (z slow, x fast) layout of the 1-D transforms are then read and written in runs of 32 consecutive elements (512
// bytes).  Block = 32 x 8 threads, one tile of one (species, y) plane; grid (N / 32, ceil(NH / 32), ng * ny).
#define TR_TILE 32
// received blocks [g][ix_s][yl][z] of every source s -> T[g][yl][z][x] (x fastest: contiguous 1-D transforms along x)
__global__ __launch_bounds__(TR_TILE * 8) void k_tr_unpack_fwd(const double2 *__restrict__ recv, const long long *__restrict__ tab, int N,
                                                               int NH, int ny, int ng, double2 *__restrict__ T)
{
  __shared__ double2 tile[TR_TILE][TR_TILE + 1];
  const int tx = threadIdx.x & (TR_TILE - 1), ty = threadIdx.x / TR_TILE;
  const int x0 = blockIdx.x * TR_TILE, z0 = blockIdx.y * TR_TILE;
  const int yl = blockIdx.z % ny, g = blockIdx.z / ny;
  for(int xx = ty; xx < TR_TILE; xx += 8)
    {
      const int x = x0 + xx, z = z0 + tx;
      if(x < N && z < NH)
        {
          const int s = T_XOWN(tab, x);
          const long long xss = T_R(tab, s, 2), nxs = T_R(tab, s, 3), off = T_R(tab, s, 5);
          tile[xx][tx] = recv[off + (((long long)g * nxs + (x - xss)) * ny + yl) * NH + z];
        }
    }
  __syncthreads();
  for(int zz = ty; zz < TR_TILE; zz += 8)
    {
      const int x = x0 + tx, z = z0 + zz;
      if(x < N && z < NH)
        T[(((long long)g * ny + yl) * NH + z) * N + x] = tile[tx][zz];
    }
}
// T[g][yl][z][x] -> blocks [g][x - xs_d][yl][z] per destination d = owner of x
__global__ __launch_bounds__(TR_TILE * 8) void k_tr_pack_bwd(const double2 *__restrict__ T, const long long *__restrict__ tab, int N, int NH,
                                                             int ny, int ng, double2 *__restrict__ send)
{
  __shared__ double2 tile[TR_TILE][TR_TILE + 1];
  const int tx = threadIdx.x & (TR_TILE - 1), ty = threadIdx.x / TR_TILE;
  const int x0 = blockIdx.x * TR_TILE, z0 = blockIdx.y * TR_TILE;
  const int yl = blockIdx.z % ny, g = blockIdx.z / ny;
  for(int zz = ty; zz < TR_TILE; zz += 8)
    {
      const int x = x0 + tx, z = z0 + zz;
      if(x < N && z < NH)
        tile[zz][tx] = T[(((long long)g * ny + yl) * NH + z) * N + x];
    }
  __syncthreads();
  for(int xx = ty; xx < TR_TILE; xx += 8)
    {
      const int x = x0 + xx, z = z0 + tx;
      if(x < N && z < NH)
        {
          const int d = T_XOWN(tab, x);
          const long long xsd = T_R(tab, d, 2), nxd = T_R(tab, d, 3), off = T_R(tab, d, 4);
          send[off + (((long long)g * nxd + (x - xsd)) * ny + yl) * NH + z] = tile[tx][xx];
        }
    }
}
// received blocks [g][ix][y - ys_s][z] of every source s -> C[g][ix][y][z]
__global__ void k_tr_unpack_bwd(const double2 *__restrict__ recv, const long long *__restrict__ tab, int N, int NH, int nx, int ng,
                                double2 *__restrict__ C)
{
  const long long tot = (long long)ng * nx * N * NH;
  const long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if(idx >= tot)
    return;
  const int z = (int)(idx % NH), y = (int)((idx / NH) % N), ix = (int)((idx / ((long long)NH * N)) % nx);
  const int g = (int)(idx / ((long long)NH * N * nx));
  const int s = T_YOWN(tab, y);
  const long long yss = T_R(tab, s, 0), nys = T_R(tab, s, 1), off = T_R(tab, s, 5);
  C[idx] = recv[off + (((long long)g * nx + ix) * nys + (y - yss)) * NH + z];
}

// Green's function on the transposed layout T[g][yl][z][x] (pm_periodic.c:436-520 loops y over the own slab, then x, z)
template <int NG>
__global__ void k_green_t(const double2 *__restrict__ rho, double2 *__restrict__ phi, GreenParams gp, int ys, int ny)
{
  const int N = gp.N, NH = N / 2 + 1;
  const long long total = (long long)ny * NH * N;
  const long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if(idx >= total)
    return;
  const int x = (int)(idx % N), z = (int)((idx / N) % NH), yl = (int)(idx / ((long long)N * NH));
  double2 out[NG];
  green_mode<NG>(gp, x, ys + yl, z, rho, (size_t)total, (size_t)idx, out);
#pragma unroll
  for(int b = 0; b < NG; b++)
    phi[(size_t)b * total + idx] = out[b];
}

// ---- stage 3: potential planes slab -> extended brick -----------------------------------------------------------------------
// one requester per launch: its planes xl[q] of the own slab, the (y, z) window of its extended brick
__global__ void k_pack_window(const double *__restrict__ slab, const long long *__restrict__ xl, int np, int ng, int ey, int ez, int loy,
                              int loz, int N, long long slab_cells, double *__restrict__ send)
{
  const long long per = (long long)ey * ez;
  const long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if(idx >= per * np * ng)
    return;
  const int k = (int)(idx % ez), j = (int)((idx / ez) % ey);
  const int q = (int)((idx / per) % np), g = (int)(idx / (per * np));
  int y = loy + j, z = loz + k;
  y -= y >= N ? N : 0;
  z -= z >= N ? N : 0;
  send[idx] = slab[(size_t)g * slab_cells + ((long long)xl[q] * N + y) * (N + 2) + z];
}
// desc (3 long long per plane i of the extended brick): offset of its source's block in recv, position q among that source's
// planes, number of planes np of that source
__global__ void k_unpack_planes(const double *__restrict__ recv, const long long *__restrict__ desc, int nplanes, int ng, int ey, int ez,
                                long long bcells, double *__restrict__ eb)
{
  const long long per = (long long)ey * ez;
  const long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if(idx >= per * nplanes * ng)
    return;
  const long long r = idx % per;
  const int i = (int)((idx / per) % nplanes), g = (int)(idx / (per * nplanes));
  const long long off = desc[3 * i], q = desc[3 * i + 1], np = desc[3 * i + 2];
  eb[(size_t)g * bcells + (long long)i * per + r] = recv[off + (g * np + q) * per + r];
}

// 4-point gradient at the 8 CIC corners + CIC gather on the extended brick (pm_periodic.c:681-763)
__global__ void k_gradient_gather_brick(const double4 *__restrict__ s_pm, const unsigned char *__restrict__ s_type,
                                        const unsigned char *__restrict__ s_flag, long long n, double to_slab, WalkParams wp, Brick E,
                                        const double *__restrict__ eb, double fac, double *__restrict__ r_pm)
{
  long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if(i >= n)
    return;
  if(s_flag[i] & 2)
    {
      r_pm[3 * i + 0] = r_pm[3 * i + 1] = r_pm[3 * i + 2] = 0.0;
      return;
    }
  const double4 p = s_pm[i];
  const double *grid = eb + (size_t)wp.t2g[s_type[i]] * E.cells();
  double dx, dy, dz;
  const int sx = cell_of(p.x, to_slab, E.N, &dx), sy = cell_of(p.y, to_slab, E.N, &dy), sz = cell_of(p.z, to_slab, E.N, &dz);
  const double wx[2] = {1.0 - dx, dx}, wy[2] = {1.0 - dy, dy}, wz[2] = {1.0 - dz, dz};
  double acc[3] = {0, 0, 0};
  auto at = [&](int x, int y, int z) { return grid[E.at(x, y, z)]; };
  // corner order of the reference's gather (x outer, then y/z as written at pm_periodic.c:749-757)
  const int ox[8] = {0, 0, 0, 0, 1, 1, 1, 1}, oy[8] = {0, 1, 0, 1, 0, 1, 0, 1}, oz[8] = {0, 0, 1, 1, 0, 0, 1, 1};
  for(int c = 0; c < 8; c++)
    {
      const int x = sx + ox[c], y = sy + oy[c], z = sz + oz[c];
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

// The same gather in two passes, as on the single mesh (kernels_pm.hip: k_force_mesh_march + k_gather_force): (1) the 4-point force of
// every cell of the extended brick whose +-2 neighbours it holds, 3 doubles per cell; (2) the CIC gather of 8 cell forces per
// particle, the two z-neighbour corners as 48 contiguous bytes.  Same expressions and the same corner order as the fused kernel
// above, hence bitwise the same GravPM; 8 + 12 loads per particle instead of 96, each with three index wraps.
__global__ void k_force_mesh_brick(Brick E, int ng, const double *__restrict__ eb, double fac, double *__restrict__ fm)
{
  const long long cells = E.cells();
  const long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if(idx >= cells * ng)
    return;
  const int g = (int)(idx / cells);
  const long long cell = idx - (long long)g * cells;
  const int lz = (int)(cell % E.ext[2]), ly = (int)((cell / E.ext[2]) % E.ext[1]), lx = (int)(cell / ((long long)E.ext[1] * E.ext[2]));
  const double *grid = eb + (size_t)g * cells;
  // neighbour in local coordinates: a brick that spans the box in a direction (ext = N) wraps, any other must hold it
  auto nb = [&](int l, int d, int ext) -> int {
    int v = l + d;
    if(ext == E.N)
      v = v < 0 ? v + E.N : (v >= E.N ? v - E.N : v);
    return (v >= 0 && v < ext) ? v : -1;
  };
  const int xm1 = nb(lx, -1, E.ext[0]), xp1 = nb(lx, 1, E.ext[0]), xm2 = nb(lx, -2, E.ext[0]), xp2 = nb(lx, 2, E.ext[0]);
  const int ym1 = nb(ly, -1, E.ext[1]), yp1 = nb(ly, 1, E.ext[1]), ym2 = nb(ly, -2, E.ext[1]), yp2 = nb(ly, 2, E.ext[1]);
  const int zm1 = nb(lz, -1, E.ext[2]), zp1 = nb(lz, 1, E.ext[2]), zm2 = nb(lz, -2, E.ext[2]), zp2 = nb(lz, 2, E.ext[2]);
  if((xm1 | xp1 | xm2 | xp2 | ym1 | yp1 | ym2 | yp2 | zm1 | zp1 | zm2 | zp2) < 0)
    return;   // a ghost cell: no particle's CIC cloud reaches it
  auto at = [&](int x, int y, int z) { return grid[((long long)x * E.ext[1] + y) * E.ext[2] + z]; };
  double *f = fm + 3 * idx;
  f[0] = fac * ((4.0 / 3) * (at(xm1, ly, lz) - at(xp1, ly, lz)) - (1.0 / 6) * (at(xm2, ly, lz) - at(xp2, ly, lz)));
  f[1] = fac * ((4.0 / 3) * (at(lx, ym1, lz) - at(lx, yp1, lz)) - (1.0 / 6) * (at(lx, ym2, lz) - at(lx, yp2, lz)));
  f[2] = fac * ((4.0 / 3) * (at(lx, ly, zm1) - at(lx, ly, zp1)) - (1.0 / 6) * (at(lx, ly, zm2) - at(lx, ly, zp2)));
}

__global__ void k_gather_force_brick(const double4 *__restrict__ s_pm, const unsigned char *__restrict__ s_type,
                                     const unsigned char *__restrict__ s_flag, long long n, double to_slab, WalkParams wp, Brick E,
                                     const double *__restrict__ fm, double *__restrict__ r_pm)
{
  long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if(i >= n)
    return;
  if(s_flag[i] & 2)
    {
      r_pm[3 * i + 0] = r_pm[3 * i + 1] = r_pm[3 * i + 2] = 0.0;
      return;
    }
  const double4 p = s_pm[i];
  const double *f = fm + (size_t)wp.t2g[s_type[i]] * 3 * E.cells();
  double dx, dy, dz;
  const int sx = cell_of(p.x, to_slab, E.N, &dx), sy = cell_of(p.y, to_slab, E.N, &dy), sz = cell_of(p.z, to_slab, E.N, &dz);
  const double wx[2] = {1.0 - dx, dx}, wy[2] = {1.0 - dy, dy}, wz[2] = {1.0 - dz, dz};
  const int lx[2] = {wrapd(sx - E.lo[0], E.N), wrapd(sx + 1 - E.lo[0], E.N)}, ly[2] = {wrapd(sy - E.lo[1], E.N), wrapd(sy + 1 - E.lo[1], E.N)};
  const int lz0 = wrapd(sz - E.lo[2], E.N), lz1 = wrapd(sz + 1 - E.lo[2], E.N);
  typedef double d2u __attribute__((ext_vector_type(2), aligned(8)));
  const bool zpair = lz1 == lz0 + 1;
  double fv[8][3];
#pragma unroll
  for(int cx = 0; cx < 2; cx++)
#pragma unroll
    for(int cy = 0; cy < 2; cy++)
      {
        const double *f0 = f + 3 * (((long long)lx[cx] * E.ext[1] + ly[cy]) * E.ext[2] + lz0);
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
            const double *f1 = f + 3 * (((long long)lx[cx] * E.ext[1] + ly[cy]) * E.ext[2] + lz1);
#pragma unroll
            for(int d = 0; d < 3; d++)
              {
                fv[c0][d] = f0[d];
                fv[c1][d] = f1[d];
              }
          }
      }
  // corner order of the reference's gather (x outer, then y/z as written at pm_periodic.c:749-757)
  const int ox[8] = {0, 0, 0, 0, 1, 1, 1, 1}, oy[8] = {0, 1, 0, 1, 0, 1, 0, 1}, oz[8] = {0, 0, 1, 1, 0, 0, 1, 1};
  double acc[3] = {0, 0, 0};
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

// =============================================================================================================================
//  host side
// =============================================================================================================================
#define GRIDN(n) dim3((unsigned)((((long long)(n)) + 255) / 256)), dim3(256)

static void destroy_plan(void *&p)
{
  if(p)
    {
      hipfftDestroy(*(hipfftHandle *)p);
      delete(hipfftHandle *)p;
      p = nullptr;
    }
}

void pmslab_release(ngravs_ctx *c)
{
  PmSlab &s = c->pms;
  destroy_plan(s.plan2f);
  destroy_plan(s.plan2i);
  destroy_plan(s.plan1);
  s.plan_N = s.plan_nx = s.plan_ny = 0;
  s.brick.release();
  s.slab.release();
  s.tbuf.release();
  s.ebrick.release();
  s.fmesh.release();
  s.send.release();
  s.recv.release();
  s.desc.release();
}

static int ensure_plans(ngravs_ctx *c)
{
  PmSlab &s = c->pms;
  const int N = s.N;
  if(s.plan_N == N && s.plan_nx == s.nx && s.plan_ny == s.ny)
    return NGRAVS_OK;
  destroy_plan(s.plan2f);
  destroy_plan(s.plan2i);
  destroy_plan(s.plan1);
  const int NH = N / 2 + 1;
  if(s.nx > 0)
    {
      // 2-D transforms of the own planes, in place on the padded real layout [nx][N][N+2] <-> [nx][N][NH] complex
      int n2[2] = {N, N}, rembed[2] = {N, N + 2}, cembed[2] = {N, NH};
      s.plan2f = new hipfftHandle;
      s.plan2i = new hipfftHandle;
      FFT_TRY(c, hipfftPlanMany((hipfftHandle *)s.plan2f, 2, n2, rembed, 1, N * (N + 2), cembed, 1, N * NH, HIPFFT_D2Z, s.nx));
      FFT_TRY(c, hipfftPlanMany((hipfftHandle *)s.plan2i, 2, n2, cembed, 1, N * NH, rembed, 1, N * (N + 2), HIPFFT_Z2D, s.nx));
      FFT_TRY(c, hipfftSetStream(*(hipfftHandle *)s.plan2f, c->stream));
      FFT_TRY(c, hipfftSetStream(*(hipfftHandle *)s.plan2i, c->stream));
    }
  if(s.ny > 0)
    {
      int n1[1] = {N};
      s.plan1 = new hipfftHandle;
      FFT_TRY(c, hipfftPlanMany((hipfftHandle *)s.plan1, 1, n1, nullptr, 1, N, nullptr, 1, N, HIPFFT_Z2Z, s.ny * NH));
      FFT_TRY(c, hipfftSetStream(*(hipfftHandle *)s.plan1, c->stream));
    }
  s.plan_N = N;
  s.plan_nx = s.nx;
  s.plan_ny = s.ny;
  return NGRAVS_OK;
}

// the planes i (0 <= i < ext) of a brick starting at mesh plane lo that fall into [s0, s0 + sn): appended to `out` in increasing i
static void planes_in_slab(int lo, int ext, int N, int s0, int sn, std::vector<int> &out)
{
  for(int i = 0; i < ext; i++)
    {
      const int x = (lo + i) % N;
      if(x >= s0 && x < s0 + sn)
        out.push_back(i);
    }
}

static void extended(const int *bb, int N, int elo[3], int eext[3])
{
  for(int j = 0; j < 3; j++)
    {
      const int ext = bb[3 + j];
      if(ext <= 0)
        {
          elo[j] = 0;
          eext[j] = 0;
        }
      else if(ext + 4 >= N)
        {
          elo[j] = 0;
          eext[j] = N;
        }
      else
        {
          elo[j] = (bb[j] - 2 + N) % N;
          eext[j] = ext + 4;
        }
    }
  if(eext[0] == 0 || eext[1] == 0 || eext[2] == 0)
    eext[0] = eext[1] = eext[2] = 0;
}

int pmslab_begin(ngravs_ctx *c, int rank, int world, int bbox[6])
{
  PmSlab &s = c->pms;
  const int N = c->cfg.pmgrid, ng = c->cfg.n_gravs;
  if(N <= 0 || !c->cfg.periodic || world < 1 || rank < 0 || rank >= world || 2 * world > N)
    {
      ngravs_report(c, NGRAVS_ERR_ARG, "slab PM needs PERIODIC, PMGRID and at most PMGRID/2 tasks");
      return NGRAVS_ERR_ARG;
    }
  s.world = world;
  s.rank = rank;
  s.N = N;
  s.xs = slab_start(rank, world, N);
  s.nx = slab_start(rank + 1, world, N) - s.xs;
  s.ys = s.xs;
  s.ny = s.nx;
  const long long n = c->n;
  const double to_slab = N / c->cfg.box_size;
  if(c->d_counters.ensure(16) || c->r_pm.ensure(3 * (n > 0 ? n : 1)))
    return NGRAVS_ERR_NOMEM;
  int h[6] = {1 << 30, 1 << 30, 1 << 30, -1, -1, -1};
  HIP_TRY(c, hipMemcpyAsync(c->d_counters.p, h, sizeof(h), hipMemcpyHostToDevice, c->stream));
  if(n > 0)
    hipLaunchKernelGGL(k_cell_minmax, dim3(2048), dim3(256), 0, c->stream, c->s_pm.p, c->s_active.p, n, to_slab, N, c->d_counters.p);
  HIP_TRY(c, hipMemcpyAsync(h, c->d_counters.p, sizeof(h), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  for(int j = 0; j < 3; j++)
    {
      if(h[3 + j] < 0)   // no own particle
        {
          s.lo[j] = 0;
          s.ext[j] = 0;
        }
      else if(h[3 + j] - h[j] + 2 >= N)   // spans the box: wrap inside the brick
        {
          s.lo[j] = 0;
          s.ext[j] = N;
        }
      else
        {
          s.lo[j] = h[j];
          s.ext[j] = h[3 + j] - h[j] + 2;   // + the CIC neighbour
        }
    }
  if(s.ext[0] == 0 || s.ext[1] == 0 || s.ext[2] == 0)
    s.ext[0] = s.ext[1] = s.ext[2] = 0;
  for(int j = 0; j < 3; j++)
    {
      bbox[j] = s.lo[j];
      bbox[3 + j] = s.ext[j];
    }
  Brick B;
  for(int j = 0; j < 3; j++)
    {
      B.lo[j] = s.lo[j];
      B.ext[j] = s.ext[j];
    }
  B.N = N;
  const size_t bc = (size_t)B.cells();
  if(s.brick.ensure(bc * ng + 1))
    return NGRAVS_ERR_NOMEM;
  if(bc > 0)
    {
      HIP_TRY(c, hipMemsetAsync(s.brick.p, 0, sizeof(double) * bc * ng, c->stream));
      // by tiles (LDS patches of one tree level, as on the single mesh) when the tree is there; else 8 global atomics per particle
      MeshAddr ma;
      ma.N = N;
      ma.brick = 1;
      for(int j = 0; j < 3; j++)
        {
          ma.lo[j] = B.lo[j];
          ma.ext[j] = B.ext[j];
        }
      HIP_TRY(c, hipMemcpyAsync(c->d_counters.p + 8, c->cfg.type_to_grav, sizeof(int) * 6, hipMemcpyHostToDevice, c->stream));
      const int rct = pm_deposit_tiles(c, ma, s.brick.p);
      if(rct == 1)
        {
          WalkParams wp;
          make_walk_params(c, &wp);
          hipLaunchKernelGGL(k_cic_deposit_brick, GRIDN(n), 0, c->stream, c->s_pm.p, c->s_type.p, c->s_active.p, n, to_slab, wp, B, s.brick.p);
        }
      else if(rct)
        return rct;
    }
  HIP_TRY(c, hipGetLastError());
  s.stage = 0;
  for(int k = 0; k < 4; k++)
    s.bytes_sent[k] = 0;
  return NGRAVS_OK;
}

static int upload_desc(ngravs_ctx *c, const std::vector<long long> &h)
{
  PmSlab &s = c->pms;
  if(s.desc.ensure(h.size() + 1))
    return NGRAVS_ERR_NOMEM;
  if(!h.empty())
    {
      // h is a local of the caller: the copy reads one of two host vectors that live in the context, used in turn -- every pack
      // stage ends in a stream synchronisation, so a vector's last copy has completed when its turn comes again (no wait here:
      // the table follows the kernels already queued, the kernels that read it follow the table)
      std::vector<long long> &keep = s.hdesc[s.hdesc_turn ^= 1];
      keep.assign(h.begin(), h.end());
      HIP_TRY(c, hipMemcpyAsync(s.desc.p, keep.data(), sizeof(long long) * keep.size(), hipMemcpyHostToDevice, c->stream));
    }
  return NGRAVS_OK;
}

// owner tables of the transposes (see T_YOWN / T_XOWN / T_R); soff / roff in complex numbers
static void transpose_tab(const PmSlab &s, const std::vector<long long> &soff, const std::vector<long long> &roff, std::vector<long long> &tab)
{
  const int N = s.N, W = s.world;
  tab.assign(2 * (size_t)N + 6 * (size_t)W, 0);
  for(int r = 0; r < W; r++)
    {
      const int a = slab_start(r, W, N), b = slab_start(r + 1, W, N);
      for(int v = a; v < b; v++)
        tab[v] = tab[N + v] = r;
      tab[2 * N + 6 * r + 0] = a;
      tab[2 * N + 6 * r + 1] = b - a;
      tab[2 * N + 6 * r + 2] = a;
      tab[2 * N + 6 * r + 3] = b - a;
      tab[2 * N + 6 * r + 4] = soff[r];
      tab[2 * N + 6 * r + 5] = roff[r];
    }
}

int pmslab_pack(ngravs_ctx *c, int stage, const int *all_bbox, int64_t *send_counts, int64_t *recv_counts, void **send, void **recv)
{
  PmSlab &s = c->pms;
  if(stage != s.stage || stage < 0 || stage > 3)
    {
      ngravs_report(c, NGRAVS_ERR_STATE, "slab PM: stages must run in order 0..3 after ngravs_pm_slab_begin");
      return NGRAVS_ERR_STATE;
    }
  const int N = s.N, NH = N / 2 + 1, W = s.world, ng = c->cfg.n_gravs, me = s.rank;
  if(stage == 0)
    {
      if(!all_bbox)
        return NGRAVS_ERR_ARG;
      s.bbox.assign(all_bbox, all_bbox + 6 * (size_t)W);
      for(int j = 0; j < 6; j++)
        if(s.bbox[6 * me + j] != (j < 3 ? s.lo[j] : s.ext[j - 3]))
          {
            ngravs_report(c, NGRAVS_ERR_ARG, "slab PM: all_bbox[rank] is not this task's brick");
            return NGRAVS_ERR_ARG;
          }
      int rcp = ensure_plans(c);
      if(rcp)
        return rcp;
    }
  s.scount.assign(W, 0);
  s.rcount.assign(W, 0);
  std::vector<long long> soff(W + 1, 0), roff(W + 1, 0);
  const size_t slab_cells = (size_t)s.nx * N * (N + 2);
  int rc;
  if(stage == 0)
    {
      // planes of my brick by destination; planes of every task's brick that land in my slab
      const long long per = (long long)s.ext[1] * s.ext[2];
      std::vector<long long> desc;
      for(int d = 0; d < W; d++)
        {
          std::vector<int> pl;
          planes_in_slab(s.lo[0], s.ext[0], N, slab_start(d, W, N), slab_start(d + 1, W, N) - slab_start(d, W, N), pl);
          s.scount[d] = (int64_t)ng * (long long)pl.size() * per;
          soff[d + 1] = soff[d] + s.scount[d];
          for(size_t q = 0; q < pl.size(); q++)
            {
              desc.push_back(pl[q]);
              desc.push_back((long long)q);
              desc.push_back((long long)pl.size());
              desc.push_back(soff[d]);
            }
        }
      for(int r = 0; r < W; r++)
        {
          const int *bb = &s.bbox[6 * r];
          std::vector<int> pl;
          planes_in_slab(bb[0], bb[3], N, s.xs, s.nx, pl);
          s.rcount[r] = (int64_t)ng * (long long)pl.size() * bb[4] * bb[5];
          roff[r + 1] = roff[r] + s.rcount[r];
        }
      if(s.send.ensure((size_t)soff[W] + 1) || s.recv.ensure((size_t)roff[W] + 1) || s.slab.ensure(slab_cells * ng + 1))
        return NGRAVS_ERR_NOMEM;
      if((rc = upload_desc(c, desc)))
        return rc;
      if(soff[W] > 0)
        hipLaunchKernelGGL(k_pack_planes, GRIDN(soff[W]), 0, c->stream, s.brick.p, s.desc.p, s.ext[0], ng, s.ext[1], s.ext[2],
                           (long long)s.ext[0] * per, s.send.p);
    }
  else if(stage == 1 || stage == 2)
    {
      // counts in doubles (complex = 2)
      for(int r = 0; r < W; r++)
        {
          const long long nr = slab_start(r + 1, W, N) - slab_start(r, W, N);
          const long long blk = (long long)ng * s.nx * nr * NH;   // forward: my x planes x its y rows; backward: its x planes x my y rows
          s.scount[r] = 2 * blk;
          s.rcount[r] = 2 * blk;                                  // symmetric because xs == ys, nx == ny per task
          soff[r + 1] = soff[r] + blk;
          roff[r + 1] = roff[r] + blk;
        }
      const size_t tcells = (size_t)s.ny * NH * N;
      if(s.send.ensure(2 * (size_t)soff[W] + 2) || s.recv.ensure(2 * (size_t)roff[W] + 2) || s.tbuf.ensure(2 * tcells * ng * 2 + 2))
        return NGRAVS_ERR_NOMEM;
      std::vector<long long> tab;
      transpose_tab(s, soff, roff, tab);
      if((rc = upload_desc(c, tab)))
        return rc;
      if(stage == 1)
        {
          const long long tot = (long long)ng * s.nx * N * NH;
          if(tot > 0)
            hipLaunchKernelGGL(k_tr_pack_fwd, GRIDN(tot), 0, c->stream, (const double2 *)s.slab.p, s.desc.p, N, NH, s.nx, ng, (double2 *)s.send.p);
        }
      else
        {
          const long long tot = (long long)ng * tcells;
          const double2 *phi_t = (const double2 *)s.tbuf.p + (size_t)ng * tcells;   // second half of tbuf: the potentials
          if(tot > 0)
            hipLaunchKernelGGL(k_tr_pack_bwd, dim3((N + TR_TILE - 1) / TR_TILE, (NH + TR_TILE - 1) / TR_TILE, (unsigned)(ng * s.ny)), dim3(TR_TILE * 8), 0, c->stream,
                               phi_t, s.desc.p, N, NH, s.ny, ng, (double2 *)s.send.p);
        }
    }
  else
    {
      // potential: every requester r gets the planes of its extended brick that I own, cut to its (y, z) window
      std::vector<std::vector<int>> pls(W);
      std::vector<long long> xl;
      std::vector<long long> xoff(W + 1, 0);
      for(int r = 0; r < W; r++)
        {
          int elo[3], eext[3];
          extended(&s.bbox[6 * r], N, elo, eext);
          planes_in_slab(elo[0], eext[0], N, s.xs, s.nx, pls[r]);
          s.scount[r] = (int64_t)ng * (long long)pls[r].size() * eext[1] * eext[2];
          soff[r + 1] = soff[r] + s.scount[r];
          for(int i : pls[r])
            xl.push_back((elo[0] + i) % N - s.xs);
          xoff[r + 1] = (long long)xl.size();
        }
      extended(&s.bbox[6 * me], N, s.elo, s.eext);
      const long long eper = (long long)s.eext[1] * s.eext[2];
      std::vector<long long> edesc(3 * (size_t)s.eext[0], 0);
      for(int r = 0; r < W; r++)
        {
          std::vector<int> pl;
          planes_in_slab(s.elo[0], s.eext[0], N, slab_start(r, W, N), slab_start(r + 1, W, N) - slab_start(r, W, N), pl);
          s.rcount[r] = (int64_t)ng * (long long)pl.size() * eper;
          roff[r + 1] = roff[r] + s.rcount[r];
          for(size_t q = 0; q < pl.size(); q++)
            {
              edesc[3 * (size_t)pl[q] + 0] = roff[r];
              edesc[3 * (size_t)pl[q] + 1] = (long long)q;
              edesc[3 * (size_t)pl[q] + 2] = (long long)pl.size();
            }
        }
      if(s.send.ensure((size_t)soff[W] + 1) || s.recv.ensure((size_t)roff[W] + 1) || s.ebrick.ensure((size_t)s.eext[0] * eper * ng + 1))
        return NGRAVS_ERR_NOMEM;
      // desc: [xl of all requesters][edesc]
      std::vector<long long> desc(xl);
      desc.insert(desc.end(), edesc.begin(), edesc.end());
      if((rc = upload_desc(c, desc)))
        return rc;
      for(int r = 0; r < W; r++)
        {
          if(s.scount[r] == 0)
            continue;
          int elo[3], eext[3];
          extended(&s.bbox[6 * r], N, elo, eext);
          hipLaunchKernelGGL(k_pack_window, GRIDN(s.scount[r]), 0, c->stream, s.slab.p, s.desc.p + xoff[r], (int)pls[r].size(), ng, eext[1],
                             eext[2], elo[1], elo[2], N, (long long)slab_cells, s.send.p + soff[r]);
        }
      s.edesc_off = (long long)xl.size();
    }
  HIP_TRY(c, hipGetLastError());
  HIP_TRY(c, hipStreamSynchronize(c->stream));   // the host's collective reads the send buffer next
  for(int r = 0; r < W; r++)
    {
      send_counts[r] = s.scount[r];
      recv_counts[r] = s.rcount[r];
      if(r != me)
        s.bytes_sent[stage] += 8.0 * (double)s.scount[r];
    }
  *send = s.send.p;
  *recv = s.recv.p;
  return NGRAVS_OK;
}

int pmslab_unpack(ngravs_ctx *c, int stage)
{
  PmSlab &s = c->pms;
  if(stage != s.stage)
    return NGRAVS_ERR_STATE;
  const int N = s.N, NH = N / 2 + 1, W = s.world, ng = c->cfg.n_gravs;
  const size_t slab_cells = (size_t)s.nx * N * (N + 2);
  const size_t tcells = (size_t)s.ny * NH * N;
  int rc;
  if(stage == 0)
    {
      if(slab_cells > 0)
        HIP_TRY(c, hipMemsetAsync(s.slab.p, 0, sizeof(double) * slab_cells * ng, c->stream));
      std::vector<long long> xl;
      std::vector<long long> xoff(W + 1, 0);
      for(int r = 0; r < W; r++)
        {
          const int *bb = &s.bbox[6 * r];
          std::vector<int> pl;
          planes_in_slab(bb[0], bb[3], N, s.xs, s.nx, pl);
          for(int i : pl)
            xl.push_back((bb[0] + i) % N - s.xs);
          xoff[r + 1] = (long long)xl.size();
        }
      if((rc = upload_desc(c, xl)))
        return rc;
      long long roff = 0;
      for(int r = 0; r < W; r++)
        {
          const int *bb = &s.bbox[6 * r];
          const int np = (int)(xoff[r + 1] - xoff[r]);
          if(s.rcount[r] > 0)
            hipLaunchKernelGGL(k_add_planes, GRIDN(s.rcount[r]), 0, c->stream, s.recv.p + roff, s.desc.p + xoff[r], np, ng, bb[4], bb[5],
                               bb[1], bb[2], N, (long long)slab_cells, s.slab.p);
          roff += s.rcount[r];
        }
      for(int a = 0; a < ng && s.nx > 0; a++)
        FFT_TRY(c, hipfftExecD2Z(*(hipfftHandle *)s.plan2f, s.slab.p + slab_cells * a, (hipfftDoubleComplex *)(s.slab.p + slab_cells * a)));
    }
  else if(stage == 1)
    {
      double2 *rho_t = (double2 *)s.tbuf.p, *phi_t = rho_t + (size_t)ng * tcells;
      const long long tot = (long long)ng * tcells;
      if(tot > 0)
        {
          hipLaunchKernelGGL(k_tr_unpack_fwd, dim3((N + TR_TILE - 1) / TR_TILE, (NH + TR_TILE - 1) / TR_TILE, (unsigned)(ng * s.ny)), dim3(TR_TILE * 8), 0,
                             c->stream, (const double2 *)s.recv.p, s.desc.p, N, NH, s.ny, ng, rho_t);
          for(int a = 0; a < ng; a++)
            FFT_TRY(c, hipfftExecZ2Z(*(hipfftHandle *)s.plan1, (hipfftDoubleComplex *)(rho_t + tcells * a),
                                     (hipfftDoubleComplex *)(rho_t + tcells * a), HIPFFT_FORWARD));
          GreenParams gp;
          make_green_params(c, &gp);
          switch(ng)
            {
            case 1:
              hipLaunchKernelGGL(k_green_t<1>, GRIDN(tcells), 0, c->stream, rho_t, phi_t, gp, s.ys, s.ny);
              break;
            case 2:
              hipLaunchKernelGGL(k_green_t<2>, GRIDN(tcells), 0, c->stream, rho_t, phi_t, gp, s.ys, s.ny);
              break;
            default:
              hipLaunchKernelGGL(k_green_t<3>, GRIDN(tcells), 0, c->stream, rho_t, phi_t, gp, s.ys, s.ny);
              break;
            }
          for(int b = 0; b < ng; b++)
            FFT_TRY(c, hipfftExecZ2Z(*(hipfftHandle *)s.plan1, (hipfftDoubleComplex *)(phi_t + tcells * b),
                                     (hipfftDoubleComplex *)(phi_t + tcells * b), HIPFFT_BACKWARD));
        }
    }
  else if(stage == 2)
    {
      const long long tot = (long long)ng * s.nx * N * NH;
      if(tot > 0)
        {
          hipLaunchKernelGGL(k_tr_unpack_bwd, GRIDN(tot), 0, c->stream, (const double2 *)s.recv.p, s.desc.p, N, NH, s.nx, ng, (double2 *)s.slab.p);
          for(int b = 0; b < ng; b++)
            FFT_TRY(c, hipfftExecZ2D(*(hipfftHandle *)s.plan2i, (hipfftDoubleComplex *)(s.slab.p + slab_cells * b), s.slab.p + slab_cells * b));
        }
    }
  else
    {
      const long long nxl = s.edesc_off;   // where the extended-brick descriptors start (set by pmslab_pack)
      Brick E;
      for(int j = 0; j < 3; j++)
        {
          E.lo[j] = s.elo[j];
          E.ext[j] = s.eext[j];
        }
      E.N = N;
      const long long ecells = E.cells();
      if(ecells > 0)
        hipLaunchKernelGGL(k_unpack_planes, GRIDN(ecells * ng), 0, c->stream, s.recv.p, s.desc.p + nxl, s.eext[0], ng, s.eext[1], s.eext[2],
                           ecells, s.ebrick.p);
      const double L = c->cfg.box_size;
      double fac = c->cfg.G / (M_PI * L);      // pm_periodic.c:237-238
      fac *= 1 / (2 * L / N);
      WalkParams wp;
      make_walk_params(c, &wp);
      const long long n = c->n;
      if(n > 0 && ecells > 0 && c->tune.pm_fused_gather)
        hipLaunchKernelGGL(k_gradient_gather_brick, GRIDN(n), 0, c->stream, c->s_pm.p, c->s_type.p, c->s_active.p, n, N / L, wp, E, s.ebrick.p,
                           fac, c->r_pm.p);
      else if(n > 0 && ecells > 0)
        {
          if(s.fmesh.ensure((size_t)3 * ecells * ng + 1))
            return NGRAVS_ERR_NOMEM;
          hipLaunchKernelGGL(k_force_mesh_brick, GRIDN(ecells * ng), 0, c->stream, E, ng, s.ebrick.p, fac, s.fmesh.p);
          hipLaunchKernelGGL(k_gather_force_brick, GRIDN(n), 0, c->stream, c->s_pm.p, c->s_type.p, c->s_active.p, n, N / L, wp, E, s.fmesh.p,
                             c->r_pm.p);
        }
      else if(n > 0)
        HIP_TRY(c, hipMemsetAsync(c->r_pm.p, 0, sizeof(double) * 3 * n, c->stream));
      c->have_pm = true;
    }
  HIP_TRY(c, hipGetLastError());
  s.stage = stage == 3 ? -1 : stage + 1;
  return NGRAVS_OK;
}
