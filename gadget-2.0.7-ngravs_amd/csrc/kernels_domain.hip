// kernels_domain.hip -- domain extent, Peano-Hilbert keys, Peano order.
//
// Replaces (reference): domain_findExtent (domain.c:882-924), the key loop of
// domain_determineTopTree (domain.c:938-944) + peano_hilbert_key (peano.c:356-398), the qsort of
// keys (domain.c:946) and peano_hilbert_order/reorder_particles (peano.c:36-185, 261-295).
//
// HBM-bound integer/byte work: one thread per particle, coalesced SoA columns, keys bit-exact
// (fp64 sub, mul, truncate -- explicit _rn intrinsics so that no FMA contraction can change the
// truncated integer).  The engine sorts on a 63-bit key (21 bits/dim); its top 54 bits are the
// reference's 18-bit/dim key, because scaling by 8 commutes with fp64 rounding.
#include "engine.hpp"
#include "../../include/ngravs_peano.h"
#include <hipcub/hipcub.hpp>

__constant__ unsigned short c_ph_step[48][8] = NGRAVS_PH_STEP_INIT;

__global__ void k_minmax(const double *__restrict__ pos, long long n, double *__restrict__ out)
{
  double lo[3] = {1e37, 1e37, 1e37}, hi[3] = {-1e37, -1e37, -1e37};   // MAX_REAL_NUMBER, allvars.h:44
  for(long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
    for(int j = 0; j < 3; j++)
      {
        double v = pos[3 * i + j];
        lo[j] = v < lo[j] ? v : lo[j];
        hi[j] = v > hi[j] ? v : hi[j];
      }
  __shared__ double sh[6][4];
  for(int j = 0; j < 3; j++)
    for(int off = 32; off > 0; off >>= 1)
      {
        double a = __shfl_down(lo[j], off), b = __shfl_down(hi[j], off);
        lo[j] = a < lo[j] ? a : lo[j];
        hi[j] = b > hi[j] ? b : hi[j];
      }
  int w = threadIdx.x >> 6;
  if((threadIdx.x & 63) == 0)
    for(int j = 0; j < 3; j++)
      {
        sh[j][w] = lo[j];
        sh[3 + j][w] = hi[j];
      }
  __syncthreads();
  if(threadIdx.x == 0)
    {
      int nw = blockDim.x >> 6;
      for(int j = 0; j < 3; j++)
        {
          double a = sh[j][0], b = sh[3 + j][0];
          for(int k = 1; k < nw; k++)
            {
              a = sh[j][k] < a ? sh[j][k] : a;
              b = sh[3 + j][k] > b ? sh[3 + j][k] : b;
            }
          out[blockIdx.x * 6 + j] = a;
          out[blockIdx.x * 6 + 3 + j] = b;
        }
    }
}

int dom_find_extent(ngravs_ctx *c)
{
  const int nb = 1024, bs = 256;
  if(c->red_tmp.ensure(nb * 6))
    return NGRAVS_ERR_NOMEM;
  hipLaunchKernelGGL(k_minmax, dim3(nb), dim3(bs), 0, c->stream, c->in_pos.p, (long long)c->n, c->red_tmp.p);
  std::vector<double> h(nb * 6);
  HIP_TRY(c, hipMemcpyAsync(h.data(), c->red_tmp.p, sizeof(double) * nb * 6, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  double lo[3] = {1e37, 1e37, 1e37}, hi[3] = {-1e37, -1e37, -1e37};
  for(int b = 0; b < nb; b++)
    for(int j = 0; j < 3; j++)
      {
        if(h[b * 6 + j] < lo[j])
          lo[j] = h[b * 6 + j];
        if(h[b * 6 + 3 + j] > hi[j])
          hi[j] = h[b * 6 + 3 + j];
      }
  // domain.c:909-923, same operation order
  double len = 0;
  for(int j = 0; j < 3; j++)
    if(hi[j] - lo[j] > len)
      len = hi[j] - lo[j];
  len *= 1.001;
  for(int j = 0; j < 3; j++)
    {
      c->dom[3 + j] = 0.5 * (lo[j] + hi[j]);
      c->dom[j] = 0.5 * (lo[j] + hi[j]) - 0.5 * len;
    }
  c->dom[6] = len;
  c->dom[7] = 1.0 / len * (double)(((long long)1) << NGRAVS_BITS_PER_DIMENSION);
  return NGRAVS_OK;
}

__device__ __forceinline__ long long ph_key_dev(const unsigned short (*step)[8], int x, int y, int z, int bits)
{
  return ngravs_ph_key_tab(step, x, y, z, bits);
}

// one thread per particle: key at `bits` bits per dimension.  fac_scaled = DomainFac * 2^(bits-18)
__global__ void k_keys(const double *__restrict__ pos, long long n, double cx, double cy, double cz,
                       double fac_scaled, int bits, unsigned long long *__restrict__ keys,
                       unsigned int *__restrict__ iota)
{
  __shared__ unsigned short step[48][8];
  for(int t = threadIdx.x; t < 48 * 8; t += blockDim.x)
    step[t >> 3][t & 7] = c_ph_step[t >> 3][t & 7];
  __syncthreads();
  long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if(i >= n)
    return;
  // (P.Pos - DomainCorner) * DomainFac, truncated (domain.c:940-943); _rn: never fused
  int x = (int)__dmul_rn(__dsub_rn(pos[3 * i + 0], cx), fac_scaled);
  int y = (int)__dmul_rn(__dsub_rn(pos[3 * i + 1], cy), fac_scaled);
  int z = (int)__dmul_rn(__dsub_rn(pos[3 * i + 2], cz), fac_scaled);
  keys[i] = (unsigned long long)ph_key_dev(step, x, y, z, bits);
  if(iota)
    iota[i] = (unsigned int)i;
}

// gather the caller-order columns into Peano order
__global__ void k_gather(const unsigned int *__restrict__ idx, long long n, const double *__restrict__ pos,
                         const double *__restrict__ mass, const int *__restrict__ type,
                         const double *__restrict__ oldacc, const unsigned char *__restrict__ active,
                         double4 *__restrict__ s_pm, unsigned char *__restrict__ s_type,
                         double *__restrict__ s_oldacc, unsigned char *__restrict__ s_active)
{
  long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if(i >= n)
    return;
  unsigned int j = idx[i];
  double4 v;
  v.x = pos[3 * (long long)j + 0];
  v.y = pos[3 * (long long)j + 1];
  v.z = pos[3 * (long long)j + 2];
  v.w = mass[j];
  s_pm[i] = v;
  s_type[i] = (unsigned char)type[j];
  s_oldacc[i] = oldacc[j];
  s_active[i] = active[j];
}

int dom_keys_and_sort(ngravs_ctx *c)
{
  const long long n = c->n;
  const int bs = 256;
  const unsigned nb = (unsigned)((n + bs - 1) / bs);
  if(c->in_key.ensure(n) || c->idx_iota.ensure(n) || c->s_key.ensure(n) || c->s_idx.ensure(n) ||
     c->s_pm.ensure(n) || c->s_type.ensure(n) || c->s_oldacc.ensure(n) || c->s_active.ensure(n))
    return NGRAVS_ERR_NOMEM;
  double fac21 = c->dom[7] * (double)(1 << (TREE_BITS - NGRAVS_BITS_PER_DIMENSION));   // exact power-of-2 scaling
  hipLaunchKernelGGL(k_keys, dim3(nb), dim3(bs), 0, c->stream, c->in_pos.p, n, c->dom[0], c->dom[1], c->dom[2],
                     fac21, TREE_BITS, c->in_key.p, c->idx_iota.p);
  size_t tmp_bytes = 0;
  hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, c->in_key.p, c->s_key.p, c->idx_iota.p, c->s_idx.p, (int)n, 0,
                                     3 * TREE_BITS, c->stream);
  if(c->sort_tmp.ensure(tmp_bytes))
    return NGRAVS_ERR_NOMEM;
  HIP_TRY(c, hipcub::DeviceRadixSort::SortPairs(c->sort_tmp.p, tmp_bytes, c->in_key.p, c->s_key.p, c->idx_iota.p,
                                                c->s_idx.p, (int)n, 0, 3 * TREE_BITS, c->stream));
  hipLaunchKernelGGL(k_gather, dim3(nb), dim3(bs), 0, c->stream, c->s_idx.p, n, c->in_pos.p, c->in_mass.p,
                     c->in_type.p, c->in_oldacc.p, c->in_active.p, c->s_pm.p, c->s_type.p, c->s_oldacc.p,
                     c->s_active.p);
  HIP_TRY(c, hipGetLastError());
  return NGRAVS_OK;
}

int dom_keys_only(ngravs_ctx *c, const double *d_pos, int64_t n, const double corner[3], double fac, int bits,
                  long long *d_keys)
{
  const int bs = 256;
  const unsigned nb = (unsigned)((n + bs - 1) / bs);
  hipLaunchKernelGGL(k_keys, dim3(nb), dim3(bs), 0, c->stream, d_pos, (long long)n, corner[0], corner[1], corner[2],
                     fac, bits, (unsigned long long *)d_keys, (unsigned int *)nullptr);
  HIP_TRY(c, hipGetLastError());
  return NGRAVS_OK;
}
