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

// PERIODIC: the reference maps the particles back onto the box before it decomposes (do_box_wrapping, domain.c:81), so every
// position it hands to this path lies in [0, BoxSize]; the walk's start table and the wrap-free shortcut of the evaluation
// kernel rely on it.  Anything else is refused loudly rather than computed wrongly.
static int dom_check_periodic_extent(ngravs_ctx *c, const double lo[3], const double hi[3])
{
  if(!c->cfg.periodic || c->cfg.box_size <= 0)
    return NGRAVS_OK;
  for(int j = 0; j < 3; j++)
    if(!(lo[j] >= 0.0 && hi[j] <= c->cfg.box_size) && lo[j] <= hi[j])
      {
        ngravs_report(c, NGRAVS_ERR_ARG, "PERIODIC: a position lies outside [0, BoxSize] (wrap the particles first: do_box_wrapping(), domain.c:81)");
        return NGRAVS_ERR_ARG;
      }
  return NGRAVS_OK;
}

int dom_find_extent(ngravs_ctx *c)
{
  if(c->extent_override)   // multi-task: the all-reduced extent (domain.c:906-907) was handed in by the host
    {
      dd_apply_extent(c, c->ext_lo, c->ext_hi);
      return dom_check_periodic_extent(c, c->ext_lo, c->ext_hi);
    }
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
  if(int rc = dom_check_periodic_extent(c, lo, hi))
    return rc;
  // domain.c:909-923, same operation order
  double len = 0;
  for(int j = 0; j < 3; j++)
    if(hi[j] - lo[j] > len)
      len = hi[j] - lo[j];
  len *= 1.001;
  for(int j = 0; j < 3; j++)
    {
      c->pos_lo[j] = lo[j];
      c->pos_hi[j] = hi[j];
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
// With `rec` it also packs the particle's columns into one 48-byte record {x, y, z, mass, oldacc, type | active << 8}
// (streamed, coalesced), so that the permutation into Peano order gathers ONE record per particle instead of five
// scattered column reads.
__global__ void k_keys(const double *__restrict__ pos, long long n, double cx, double cy, double cz,
                       double fac_scaled, int bits, unsigned long long *__restrict__ keys,
                       unsigned int *__restrict__ iota, const double *__restrict__ mass = nullptr,
                       const int *__restrict__ type = nullptr, const double *__restrict__ oldacc = nullptr,
                       const unsigned char *__restrict__ active = nullptr, double2 *__restrict__ rec = nullptr)
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
  if(rec)
    {
      double2 a, b, d;
      a.x = pos[3 * i + 0];
      a.y = pos[3 * i + 1];
      b.x = pos[3 * i + 2];
      b.y = mass[i];
      d.x = oldacc[i];
      d.y = __longlong_as_double((long long)(type[i] & 255) | ((long long)active[i] << 8));
      rec[3 * i + 0] = a;
      rec[3 * i + 1] = b;
      rec[3 * i + 2] = d;
    }
}

// gather the caller-order columns into Peano order
// jumps (optional): how many rows (of every 64th block) are NOT next to the row before them in caller order (more than 64 rows apart): the measure of how
// far the caller-order columns are from Peano order -- every such row is a scattered read here and in every pass that visits the
// rows in Peano order (dd_peano_order_own puts the columns in order when most reads have become jumps)
__global__ void k_gather(const unsigned int *__restrict__ idx, long long n, const double2 *__restrict__ rec,
                         double4 *__restrict__ s_pm, unsigned char *__restrict__ s_type,
                         double *__restrict__ s_oldacc, unsigned char *__restrict__ s_active, int *__restrict__ jumps = nullptr)
{
  long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if(i >= n)
    return;
  const long long j = idx[i];
  if(jumps && (blockIdx.x & 63) == 0)   // a sample: every 64th block (the counter is one word for the whole grid)
    {
      const long long jp = i > 0 ? (long long)idx[i - 1] : j, d = j - jp;
      const unsigned long long jm = __builtin_amdgcn_ballot_w64(d > 64 || d < -64);
      if(jm && (threadIdx.x & 63) == (unsigned)__builtin_ctzll(__builtin_amdgcn_ballot_w64(true)))
        atomicAdd(jumps, (int)__popcll(jm));
    }
  const double2 a = rec[3 * j + 0], b = rec[3 * j + 1], d = rec[3 * j + 2];
  double4 v;
  v.x = a.x;
  v.y = a.y;
  v.z = b.x;
  v.w = b.y;
  s_pm[i] = v;
  const long long meta = __double_as_longlong(d.y);
  s_type[i] = (unsigned char)(meta & 255);
  s_oldacc[i] = d.x;
  s_active[i] = (unsigned char)((meta >> 8) & 255);
}

// Two-stage sort: a stable radix sort on the TOP bits of the key only (28 of the 63 to begin with: 4 passes instead of 9), then every run of
// equal top bits -- short: 2^28 cells for the particles to share -- is put in order of its low bits by the thread that
// finds its head (stable insertion sort of the (key, index) pairs in place).  The result is exactly that of a stable sort on all
// 63 bits.  A run longer than SORT_RUN_MAX (a pathological clump below 1/16384 of the domain) raises a flag and the caller
// sorts again on all bits.
// How many low bits are left to the fix-up adapts to the particle set: 35 (4 passes) to begin with; a run that is too long
// makes this step sort on all bits and the next steps leave 28, then 21, then none (ctx sort_low; new particles start over).
#define SORT_RUN_MAX 64
__global__ void k_sort_fixup(unsigned long long *__restrict__ key, unsigned int *__restrict__ idx, long long n, int *__restrict__ flag,
                             const int SORT_LOW_BITS)
{
  const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if(i >= n)
    return;
  const unsigned long long top = key[i] >> SORT_LOW_BITS;
  if(i > 0 && (key[i - 1] >> SORT_LOW_BITS) == top)
    return;   // not the head of a run
  long long e = i + 1;
  while(e < n && e - i <= SORT_RUN_MAX && (key[e] >> SORT_LOW_BITS) == top)
    e++;
  const int r = (int)(e - i);
  if(r < 2)
    return;
  if(r > SORT_RUN_MAX)
    {
      *flag = 1;
      return;
    }
  for(int a = 1; a < r; a++)   // stable insertion sort by the full key
    {
      const unsigned long long k = key[i + a];
      const unsigned int v = idx[i + a];
      int b = a - 1;
      while(b >= 0 && key[i + b] > k)
        {
          key[i + b + 1] = key[i + b];
          idx[i + b + 1] = idx[i + b];
          b--;
        }
      key[i + b + 1] = k;
      idx[i + b + 1] = v;
    }
}

int dom_keys_and_sort(ngravs_ctx *c)
{
  const long long n = c->n;
  const int bs = 256;
  const unsigned nb = (unsigned)((n + bs - 1) / bs);
  if(c->in_key.ensure(n) || c->idx_iota.ensure(n) || c->s_key.ensure(n) || c->s_idx.ensure(n) || c->in_rec.ensure(3 * n) ||
     c->s_pm.ensure(n) || c->s_type.ensure(n) || c->s_oldacc.ensure(n) || c->s_active.ensure(n))
    return NGRAVS_ERR_NOMEM;
  double fac21 = c->dom[7] * (double)(1 << (TREE_BITS - NGRAVS_BITS_PER_DIMENSION));   // exact power-of-2 scaling
  hipLaunchKernelGGL(k_keys, dim3(nb), dim3(bs), 0, c->stream, c->in_pos.p, n, c->dom[0], c->dom[1], c->dom[2],
                     fac21, TREE_BITS, c->in_key.p, c->idx_iota.p, c->in_mass.p, c->in_type.p, c->in_oldacc.p, c->in_active.p,
                     c->in_rec.p);
  size_t tmp_bytes = 0;
  (void)hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, c->in_key.p, c->s_key.p, c->idx_iota.p, c->s_idx.p, (int)n, 0,
                                     3 * TREE_BITS, c->stream);
  if(c->sort_tmp.ensure(tmp_bytes) || c->d_counters.ensure(16))
    return NGRAVS_ERR_NOMEM;
  bool full = c->tune.sort_full != 0 || n < 4096 || c->sort_low <= 0;
  if(!full)
    {
      int h_flag = 0;
      const int low = c->sort_low;
      HIP_TRY(c, hipMemsetAsync(c->d_counters.p + 15, 0, sizeof(int), c->stream));
      HIP_TRY(c, hipcub::DeviceRadixSort::SortPairs(c->sort_tmp.p, tmp_bytes, c->in_key.p, c->s_key.p, c->idx_iota.p,
                                                    c->s_idx.p, (int)n, low, 3 * TREE_BITS, c->stream));
      hipLaunchKernelGGL(k_sort_fixup, dim3(nb), dim3(bs), 0, c->stream, c->s_key.p, c->s_idx.p, n, c->d_counters.p + 15, low);
      HIP_TRY(c, hipMemcpyAsync(&h_flag, c->d_counters.p + 15, sizeof(int), hipMemcpyDeviceToHost, c->stream));
      HIP_TRY(c, hipStreamSynchronize(c->stream));
      full = h_flag != 0;   // a clump too dense for the fix-up: sort on all bits, and leave it fewer bits from now on
      if(full)
        c->sort_low = low > 28 ? 28 : (low > 21 ? 21 : 0);
    }
  if(full)
    HIP_TRY(c, hipcub::DeviceRadixSort::SortPairs(c->sort_tmp.p, tmp_bytes, c->in_key.p, c->s_key.p, c->idx_iota.p,
                                                  c->s_idx.p, (int)n, 0, 3 * TREE_BITS, c->stream));
  HIP_TRY(c, hipMemsetAsync(c->d_counters.p + 14, 0, sizeof(int), c->stream));
  hipLaunchKernelGGL(k_gather, dim3(nb), dim3(bs), 0, c->stream, c->s_idx.p, n, c->in_rec.p, c->s_pm.p, c->s_type.p,
                     c->s_oldacc.p, c->s_active.p, c->d_counters.p + 14);
  HIP_TRY(c, hipGetLastError());
  return NGRAVS_OK;
}

// drifted tree: same Peano order, fresh columns (positions, OldAcc, active flags) -- pack + gather without the sort
int dom_regather(ngravs_ctx *c)
{
  const long long n = c->n;
  const int bs = 256;
  const unsigned nb = (unsigned)((n + bs - 1) / bs);
  if(c->in_rec.ensure(3 * n))
    return NGRAVS_ERR_NOMEM;
  double fac21 = c->dom[7] * (double)(1 << (TREE_BITS - NGRAVS_BITS_PER_DIMENSION));
  hipLaunchKernelGGL(k_keys, dim3(nb), dim3(bs), 0, c->stream, c->in_pos.p, n, c->dom[0], c->dom[1], c->dom[2], fac21, TREE_BITS,
                     c->in_key.p, (unsigned int *)nullptr, c->in_mass.p, c->in_type.p, c->in_oldacc.p, c->in_active.p, c->in_rec.p);
  hipLaunchKernelGGL(k_gather, dim3(nb), dim3(bs), 0, c->stream, c->s_idx.p, n, c->in_rec.p, c->s_pm.p, c->s_type.p,
                     c->s_oldacc.p, c->s_active.p);
  HIP_TRY(c, hipGetLastError());
  return NGRAVS_OK;
}

int dom_keys_only(ngravs_ctx *c, const double *d_pos, int64_t n, const double corner[3], double fac, int bits,
                  long long *d_keys)
{
  const int bs = 256;
  const unsigned nb = (unsigned)((n + bs - 1) / bs);
  hipLaunchKernelGGL(k_keys, dim3(nb), dim3(bs), 0, c->stream, d_pos, (long long)n, corner[0], corner[1], corner[2],
                     fac, bits, (unsigned long long *)d_keys, (unsigned int *)nullptr, (const double *)nullptr, (const int *)nullptr,
                     (const double *)nullptr, (const unsigned char *)nullptr, (double2 *)nullptr);
  HIP_TRY(c, hipGetLastError());
  return NGRAVS_OK;
}

// =====================================================================================================
//  Multi-task domain decomposition (the role of domain_decompose / domain_exchangeParticles,
//  domain.c:164-330, 554-760, and of the pseudo-particle export of gravtree.c:112-285).
//
//  MI355X design: the global Peano curve is cut at the leaves of an adaptive top tree in key space (the reference's
//  TopNodes[], domain.c:933-1138); a task owns a contiguous run of leaves.  A particle finds its leaf by walking the top
//  tree's child table with the digits of its own 63-bit key (a handful of dependent loads from a table that lives in L2).
//  Two all-to-all-v exchanges per step, both built here as packed records grouped by destination: (0) migration of
//  particles whose leaf changed owner, (2) import: ALL particles of the leaves another task asked for (its targets may
//  open them), so that the task can build the tree it needs and walk its own targets with no further communication --
//  instead of exporting targets and importing partial forces (gravtree.c:170-280).  The host drives the collectives
//  (host/ngravs_host.c over RCCL / MPI); this file only packs and unpacks.
// =====================================================================================================
struct DDRecord
{
  double x, y, z, m, oldacc, cost;
  long long meta;   // type | active << 8 | id << 16
};

__global__ void k_dd_iota(long long *p, long long n)
{
  long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if(i < n)
    p[i] = i;
}

// the top leaf a particle lies in: its key's digits select the child, level by level, until a leaf is reached
__device__ __forceinline__ int dd_leaf(const unsigned short (*step)[8], const double *pos, long long i, double cx, double cy,
                                       double cz, double fac21, const int *__restrict__ t_child, const int *__restrict__ t_leaf)
{
  int x = (int)__dmul_rn(__dsub_rn(pos[3 * i + 0], cx), fac21);
  int y = (int)__dmul_rn(__dsub_rn(pos[3 * i + 1], cy), fac21);
  int z = (int)__dmul_rn(__dsub_rn(pos[3 * i + 2], cz), fac21);
  const unsigned long long key = (unsigned long long)ngravs_ph_key_tab(step, x, y, z, TREE_BITS);
  int node = 0, c, sh = 3 * (TREE_BITS - 1);
  while((c = t_child[node]) >= 0 && sh >= 0)
    {
      node = c + (int)((key >> sh) & 7ull);
      sh -= 3;
    }
  return t_leaf[node];
}

// Per-leaf sums without one atomic per particle and value: visited in the Peano order of the last local decomposition
// (`order`: rows of the working set then, own rows are the ones below n; any permutation is correct, this one puts the
// particles of a leaf next to each other), a wave whose particles all lie in ONE leaf adds up across its lanes and issues one
// atomic per value.  Waves that straddle leaves, and runs without an order, fall back to one atomic per particle.
// (A third level -- the waves of a 1024-thread block adding up in LDS -- measured slower: 1.88 against 1.55 ms for the stage.)
__device__ __forceinline__ double wave_sum_f64(double v)
{
  for(int off = 32; off > 0; off >>= 1)
    v += __shfl_xor(v, off);
  return v;
}
__device__ __forceinline__ bool dd_row(const unsigned int *__restrict__ order, long long norder, long long n, long long j, long long *i)
{
  if(order)
    {
      *i = j < norder ? (long long)order[j] : n;
      return *i < n;
    }
  *i = j;
  return j < n;
}

// per top leaf: [0] work sum(1 + GravCost) (domain_sumCost, domain.c:859-862), [1..6] particles per type, per species mass and
// first moments (the local part of DomainMoment[], forcetree.c:766-850) -- ONE pass for the count/work histogram of the cut
// and the moments of the global top
__global__ void k_dd_leafsums(const double *__restrict__ pos, const double *__restrict__ mass, const int *__restrict__ type,
                              const double *__restrict__ cost, long long n, const unsigned int *__restrict__ order, long long norder,
                              double cx, double cy, double cz, double fac21, const int *__restrict__ t_child,
                              const int *__restrict__ t_leaf, int ng, unsigned t2g_packed, double *__restrict__ sums,
                              const int *__restrict__ leaf_in = nullptr, int stride = 0)
{
  // leaf_in (kept decomposition): the leaf a row was in when the domains were cut -- the particle has drifted since, its
  // membership has not (the reference's particles stay in their nodes, the nodes grow); stride: doubles per leaf (default TOP_CW)
  const int cwl = stride > 0 ? stride : TOP_CW(ng);
  __shared__ unsigned short step[48][8];
  for(int t = threadIdx.x; t < 48 * 8; t += blockDim.x)
    step[t >> 3][t & 7] = c_ph_step[t >> 3][t & 7];
  __syncthreads();
  long long i;
  const bool valid = dd_row(order, norder, n, blockIdx.x * (long long)blockDim.x + threadIdx.x, &i);
  const unsigned long long vm = __builtin_amdgcn_ballot_w64(valid);
  if(vm == 0ull)
    return;
  int ty = -1, g = -1, leaf = -1;
  double m = 0, mx = 0, my = 0, mz = 0, w = 0;
  if(valid)
    {
      leaf = leaf_in ? leaf_in[i] : dd_leaf(step, pos, i, cx, cy, cz, fac21, t_child, t_leaf);
      ty = type[i];
      g = (int)((t2g_packed >> (2 * ty)) & 3u);
      m = mass[i];
      mx = m * pos[3 * i + 0];
      my = m * pos[3 * i + 1];
      mz = m * pos[3 * i + 2];
      w = 1.0 + cost[i];
    }
  const int first = __builtin_ctzll(vm);
  const int leaf0 = __shfl(leaf, first);
  if(__builtin_amdgcn_ballot_w64(valid && leaf == leaf0) == vm)
    {
      double *c = sums + (size_t)leaf0 * cwl;
      const bool lead = (threadIdx.x & 63) == first;
      const double ws = wave_sum_f64(w);
      if(lead)
        atomicAdd(&c[0], ws);
      for(int t = 0; t < NGRAVS_NTYPES; t++)
        {
          const unsigned long long tm = __builtin_amdgcn_ballot_w64(ty == t);
          if(tm && lead)
            atomicAdd(&c[1 + t], (double)__popcll(tm));
        }
      for(int s = 0; s < ng; s++)
        {
          const unsigned long long sm = __builtin_amdgcn_ballot_w64(g == s);
          if(sm == 0ull)
            continue;
          const bool in = g == s;
          const double a0 = wave_sum_f64(in ? m : 0.0), a1 = wave_sum_f64(in ? mx : 0.0), a2 = wave_sum_f64(in ? my : 0.0),
                       a3 = wave_sum_f64(in ? mz : 0.0);
          if(lead)
            {
              atomicAdd(&c[7 + 4 * s + 0], a0);
              atomicAdd(&c[7 + 4 * s + 1], a1);
              atomicAdd(&c[7 + 4 * s + 2], a2);
              atomicAdd(&c[7 + 4 * s + 3], a3);
            }
        }
    }
  else if(valid)
    {
      double *c = sums + (size_t)leaf * cwl;
      atomicAdd(&c[0], w);
      atomicAdd(&c[1 + ty], 1.0);
      atomicAdd(&c[7 + 4 * g + 0], m);
      atomicAdd(&c[7 + 4 * g + 1], mx);
      atomicAdd(&c[7 + 4 * g + 2], my);
      atomicAdd(&c[7 + 4 * g + 3], mz);
    }
}

// which other tasks receive particle i?  One bit per task (world_size <= 64).  leaf_owner: its new owner (migration);
// reqmask: every task that asked for its leaf (import)
__global__ void k_dd_dest(const double *__restrict__ pos, long long n, double cx, double cy, double cz, double fac21,
                          const int *__restrict__ t_child, const int *__restrict__ t_leaf, const int *__restrict__ leaf_owner,
                          const unsigned long long *__restrict__ reqmask, int me, unsigned long long *__restrict__ mask, int *__restrict__ dest,
                          int *__restrict__ leaf_out = nullptr, const int *__restrict__ leaf_in = nullptr)
{
  __shared__ unsigned short step[48][8];
  for(int t = threadIdx.x; t < 48 * 8; t += blockDim.x)
    step[t >> 3][t & 7] = c_ph_step[t >> 3][t & 7];
  __syncthreads();
  long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if(i >= n)
    return;
  const int leaf = leaf_in ? leaf_in[i] : dd_leaf(step, pos, i, cx, cy, cz, fac21, t_child, t_leaf);
  if(leaf_out)
    leaf_out[i] = leaf;
  if(dest)
    dest[i] = leaf_owner[leaf];
  else if(leaf_owner)
    {
      const int o = leaf_owner[leaf];
      mask[i] = o != me ? 1ull << o : 0ull;
    }
  else
    mask[i] = reqmask[leaf] & ~(1ull << me);
}

// One atomic per wave and destination, not per particle: 8 M particles adding to the same counter serialise in the memory
// controller (measured 30 ms for the "stays here" counter alone).  The bits any lane of the wave holds are visited on the
// scalar unit; a ballot gives the lanes of one destination, its population count their number, mbcnt a lane's rank.
__device__ __forceinline__ unsigned long long wave_or64(unsigned long long v)
{
  for(int off = 32; off > 0; off >>= 1)
    v |= __shfl_xor(v, off);
  return v;
}
// Counters that every particle bumps are bumped once per BLOCK: atomics on one address are served one after the other (≈ 12 ns
// each on the MI355X -- one per wave of 8.4 M particles was 1.6 ms), so the waves of a block first add up in LDS.
#define DD_CNT_BLOCKS 1024
__global__ __launch_bounds__(256) void k_dd_count(const unsigned long long *__restrict__ mask, long long n, int nranks,
                                                  unsigned long long *__restrict__ counts)
{
  __shared__ unsigned int cnt[65];
  for(int t = threadIdx.x; t < 65; t += blockDim.x)
    cnt[t] = 0;
  __syncthreads();
  const int lane = threadIdx.x & 63;
  unsigned int nstay = 0;
  for(long long base = blockIdx.x * (long long)blockDim.x; base < n; base += gridDim.x * (long long)blockDim.x)
    {
      const long long i = base + threadIdx.x;
      const unsigned long long m = i < n ? mask[i] : 0ull;
      unsigned long long any = wave_or64(m);
      while(any)
        {
          const int r = __builtin_ctzll(any);
          any &= any - 1;
          const unsigned long long b = __builtin_amdgcn_ballot_w64(((m >> r) & 1ull) != 0);
          if(lane == 0)
            atomicAdd(&cnt[r], (unsigned int)__popcll(b));
        }
      nstay += (unsigned int)__popcll(__builtin_amdgcn_ballot_w64(i < n && m == 0));   // stays / not exported
    }
  if(lane == 0 && nstay)
    atomicAdd(&cnt[nranks], nstay);
  __syncthreads();
  for(int t = threadIdx.x; t <= nranks; t += blockDim.x)
    if(cnt[t])
      atomicAdd(&counts[t], (unsigned long long)cnt[t]);
}

// records of the exported particles, grouped by destination: a block reserves its share of every destination's range with one
// atomic (waves take their places inside it through LDS)
#define DD_FILL_THREADS 1024
__global__ __launch_bounds__(DD_FILL_THREADS) void k_dd_fill(const unsigned long long *__restrict__ mask, long long n,
                                                              const double *__restrict__ pos, const double *__restrict__ mass,
                                                              const int *__restrict__ type, const double *__restrict__ oldacc,
                                                              const unsigned char *__restrict__ active, const long long *__restrict__ id,
                                                              const double *__restrict__ cost, const unsigned long long *__restrict__ offs,
                                                              unsigned long long *__restrict__ cursor, double *__restrict__ out,
                                                              const double *__restrict__ pm, int rdbl, int *__restrict__ slot_row)
{
  __shared__ unsigned int lcnt[64];                                 // per destination: records of this block
  __shared__ unsigned long long lbase[64];                          // ... and where they start
  __shared__ unsigned int woff[DD_FILL_THREADS / 64][64];           // ... and where each wave's start inside
  if(threadIdx.x < 64)
    lcnt[threadIdx.x] = 0;
  __syncthreads();
  const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const unsigned long long m = i < n ? mask[i] : 0ull;
  const unsigned long long all = wave_or64(m);
  for(unsigned long long any = all; any;)
    {
      const int r = __builtin_ctzll(any);
      any &= any - 1;
      const unsigned long long b = __builtin_amdgcn_ballot_w64(((m >> r) & 1ull) != 0);
      if(lane == 0)
        woff[wave][r] = atomicAdd(&lcnt[r], (unsigned int)__popcll(b));
    }
  __syncthreads();
  if(threadIdx.x < 64 && lcnt[threadIdx.x])
    lbase[threadIdx.x] = offs[threadIdx.x] + atomicAdd(&cursor[threadIdx.x], (unsigned long long)lcnt[threadIdx.x]);
  __syncthreads();
  if(!all)
    return;
  DDRecord rec;
  rec.x = rec.y = rec.z = rec.m = rec.oldacc = rec.cost = 0;
  rec.meta = 0;
  double g0 = 0, g1 = 0, g2 = 0;   // migration records of TreePM runs carry P[].GravPM behind the 7 base words (rdbl = 10)
  if(m)
    {
      rec.x = pos[3 * i + 0];
      rec.y = pos[3 * i + 1];
      rec.z = pos[3 * i + 2];
      rec.m = mass[i];
      rec.oldacc = oldacc[i];
      rec.cost = cost[i];
      rec.meta = (long long)type[i] | ((long long)(active[i] & 1) << 8) | (id[i] << 16);
      if(pm)
        {
          g0 = pm[3 * i + 0];
          g1 = pm[3 * i + 1];
          g2 = pm[3 * i + 2];
        }
    }
  for(unsigned long long any = all; any;)
    {
      const int r = __builtin_ctzll(any);
      any &= any - 1;
      const bool mine = ((m >> r) & 1ull) != 0;
      const unsigned long long b = __builtin_amdgcn_ballot_w64(mine);
      if(mine)
        {
          const unsigned long long slot = lbase[r] + woff[wave][r] + __builtin_amdgcn_mbcnt_hi((unsigned)(b >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)b, 0));
          double *o = out + (size_t)slot * rdbl;
          if(slot_row)   // (the order of the records is the order the blocks arrived in: a later pack of the SAME rows reads it here)
            slot_row[slot] = (int)i;
          o[0] = rec.x;
          o[1] = rec.y;
          o[2] = rec.z;
          o[3] = rec.m;
          o[4] = rec.oldacc;
          o[5] = rec.cost;
          o[6] = __longlong_as_double(rec.meta);
          if(rdbl > 7)
            {
              o[7] = g0;
              o[8] = g1;
              o[9] = g2;
            }
        }
    }
}

// keep the particles that stay (mask == 0), compacted to the front of fresh columns
__global__ __launch_bounds__(DD_FILL_THREADS) void k_dd_keep(const unsigned long long *__restrict__ mask, long long n,
                                                              const double *__restrict__ pos, const double *__restrict__ mass,
                                                              const int *__restrict__ type, const double *__restrict__ oldacc,
                                                              const unsigned char *__restrict__ active, const long long *__restrict__ id,
                                                              const double *__restrict__ cost, unsigned long long *__restrict__ cursor,
                                                              double *__restrict__ pos2, double *__restrict__ mass2,
                                                              int *__restrict__ type2, double *__restrict__ oldacc2,
                                                              unsigned char *__restrict__ active2, long long *__restrict__ id2,
                                                              double *__restrict__ cost2, const double *__restrict__ pm, double *__restrict__ pm2)
{
  __shared__ unsigned int lcnt;
  __shared__ unsigned long long lbase;
  if(threadIdx.x == 0)
    lcnt = 0;
  __syncthreads();
  long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  const bool keep = i < n && mask[i] == 0;
  const unsigned long long kb = __builtin_amdgcn_ballot_w64(keep);   // one LDS atomic per wave, one global atomic per block
  unsigned int wo = 0;
  if((threadIdx.x & 63) == 0 && kb)
    wo = atomicAdd(&lcnt, (unsigned int)__popcll(kb));
  wo = __shfl(wo, 0);
  __syncthreads();
  if(threadIdx.x == 0 && lcnt)
    lbase = atomicAdd(cursor, (unsigned long long)lcnt);
  __syncthreads();
  if(!keep)
    return;
  unsigned long long k = lbase + wo + __builtin_amdgcn_mbcnt_hi((unsigned)(kb >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)kb, 0));
  pos2[3 * k + 0] = pos[3 * i + 0];
  pos2[3 * k + 1] = pos[3 * i + 1];
  pos2[3 * k + 2] = pos[3 * i + 2];
  mass2[k] = mass[i];
  type2[k] = type[i];
  oldacc2[k] = oldacc[i];
  active2[k] = active[i];
  id2[k] = id[i];
  cost2[k] = cost[i];
  if(pm2)
    for(int j = 0; j < 3; j++)
      pm2[3 * k + j] = pm ? pm[3 * i + j] : 0.0;
}

__global__ void k_dd_unpack(const double *__restrict__ rec, int rdbl, long long nrec, long long at, int halo, double *__restrict__ pos,
                            double *__restrict__ mass, int *__restrict__ type, double *__restrict__ oldacc,
                            unsigned char *__restrict__ active, long long *__restrict__ id, double *__restrict__ cost,
                            double *__restrict__ pm)
{
  long long k = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if(k >= nrec)
    return;
  const double *q = rec + (size_t)k * rdbl;
  DDRecord r;
  r.x = q[0];
  r.y = q[1];
  r.z = q[2];
  r.m = q[3];
  r.oldacc = q[4];
  r.cost = q[5];
  r.meta = __double_as_longlong(q[6]);
  long long i = at + k;
  if(pm)
    for(int j = 0; j < 3; j++)
      pm[3 * i + j] = rdbl > 7 ? q[7 + j] : 0.0;
  pos[3 * i + 0] = r.x;
  pos[3 * i + 1] = r.y;
  pos[3 * i + 2] = r.z;
  mass[i] = r.m;
  oldacc[i] = r.oldacc;
  type[i] = (int)(r.meta & 255);
  active[i] = halo ? (unsigned char)2 : (unsigned char)((r.meta >> 8) & 1);   // bit 1 = halo copy: source only
  id[i] = r.meta >> 16;
  cost[i] = r.cost;
}

template <typename T> static int grow_keep(ngravs_ctx *c, DevBuf<T> &b, size_t keep, size_t want)
{
  if(want <= b.cap)
    return 0;
  DevBuf<T> nb;
  if(nb.ensure(want + want / 4))
    return -1;
  if(keep && b.p)
    if(hipMemcpyAsync(nb.p, b.p, keep * sizeof(T), hipMemcpyDeviceToDevice, c->stream) != hipSuccess)
      return -1;
  (void)hipStreamSynchronize(c->stream);
  b.release();
  b = nb;
  return 0;
}

// doubles per exchanged record: 7 (DDRecord), + P[].GravPM for the migration of a TreePM run (ngravs_dd_record_bytes)
int dd_record_doubles(const ngravs_ctx *c, int what) { return (what == 0 && c->cfg.pmgrid) ? 10 : 7; }

static void dd_fac(const ngravs_ctx *c, double *fac21) { *fac21 = c->dom[7] * (double)(1 << (TREE_BITS - NGRAVS_BITS_PER_DIMENSION)); }

int dd_local_extent(ngravs_ctx *c, double lo[3], double hi[3])
{
  const int nb = 1024, bs = 256;
  if(c->red_tmp.ensure(nb * 6))
    return NGRAVS_ERR_NOMEM;
  hipLaunchKernelGGL(k_minmax, dim3(nb), dim3(bs), 0, c->stream, c->in_pos.p, (long long)c->n_local, c->red_tmp.p);
  std::vector<double> h(nb * 6);
  HIP_TRY(c, hipMemcpyAsync(h.data(), c->red_tmp.p, sizeof(double) * nb * 6, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  for(int j = 0; j < 3; j++)
    {
      lo[j] = 1e37;
      hi[j] = -1e37;
    }
  for(int b = 0; b < nb; b++)
    for(int j = 0; j < 3; j++)
      {
        if(h[b * 6 + j] < lo[j])
          lo[j] = h[b * 6 + j];
        if(h[b * 6 + 3 + j] > hi[j])
          hi[j] = h[b * 6 + 3 + j];
      }
  return NGRAVS_OK;
}

void dd_apply_extent(ngravs_ctx *c, const double lo[3], const double hi[3])
{
  double len = 0;   // domain.c:909-923
  for(int j = 0; j < 3; j++)
    if(hi[j] - lo[j] > len)
      len = hi[j] - lo[j];
  len *= 1.001;
  for(int j = 0; j < 3; j++)
    {
      c->pos_lo[j] = lo[j];
      c->pos_hi[j] = hi[j];
      c->dom[3 + j] = 0.5 * (lo[j] + hi[j]);
      c->dom[j] = 0.5 * (lo[j] + hi[j]) - 0.5 * len;
    }
  c->dom[6] = len;
  c->dom[7] = 1.0 / len * (double)(((long long)1) << NGRAVS_BITS_PER_DIMENSION);
}

// The top tree as the host hands it over: child table -> host copy with levels / coordinates / leaf numbers
// (ngravs_host_toptree_from_children, host/ngravs_host.c) and the two device tables the kernels descend with.  Setting the tree
// that is already set costs a comparison.
int dd_set_toptree(ngravs_ctx *c, int nnode, const int *child)
{
  TopTree &t = c->top;
  if(nnode < 1 || !child)
    return NGRAVS_ERR_ARG;
  if(t.h.nnode == nnode && t.h.child && memcmp(t.h.child, child, sizeof(int) * (size_t)nnode) == 0)
    return NGRAVS_OK;
  ngravs_toptree fresh;
  int rc = ngravs_host_toptree_from_children(&fresh, child, nnode);
  if(rc)
    {
      ngravs_report(c, rc, "ngravs_dd_set_toptree: not a tree (child[t] = first of 8 consecutive children, or -1)");
      return rc;
    }
  ngravs_host_toptree_free(&t.h);
  t.h = fresh;
  t.on = false;   /* sums and presence of the old tree say nothing about this one */
  c->have_tree = false;
  if(t.child.ensure((size_t)nnode) || t.leaf.ensure((size_t)nnode))
    return NGRAVS_ERR_NOMEM;
  HIP_TRY(c, hipMemcpyAsync(t.child.p, t.h.child, sizeof(int) * (size_t)nnode, hipMemcpyHostToDevice, c->stream));
  HIP_TRY(c, hipMemcpyAsync(t.leaf.p, t.h.leaf, sizeof(int) * (size_t)nnode, hipMemcpyHostToDevice, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  return NGRAVS_OK;
}

// per-leaf sums of the own particles on the device (+ one spare word, zero: the host's status slot in the all-reduce)
int dd_leaf_sums(ngravs_ctx *c, void **dev_sums, int64_t *count)
{
  TopTree &t = c->top;
  if(t.h.nnode < 1)
    return NGRAVS_ERR_STATE;
  const int cw = TOP_CW(c->cfg.n_gravs);
  const long long n = c->n_local, words = (long long)t.h.nleaf * cw + 1;
  if(t.leaf_sums.ensure((size_t)words))
    return NGRAVS_ERR_NOMEM;
  HIP_TRY(c, hipMemsetAsync(t.leaf_sums.p, 0, sizeof(double) * words, c->stream));
  double fac21;
  dd_fac(c, &fac21);
  unsigned t2g_packed = 0;
  for(int ty = 0; ty < NGRAVS_NTYPES; ty++)
    t2g_packed |= ((unsigned)(c->cfg.type_to_grav[ty] & 3)) << (2 * ty);
  const bool ordered = c->own_order_nlocal == n && c->own_order_len >= n && c->s_idx.p;
  const long long nthr = ordered ? c->own_order_len : n;
  if(n > 0)
    hipLaunchKernelGGL(k_dd_leafsums, dim3((unsigned)((nthr + 255) / 256)), dim3(256), 0, c->stream, c->in_pos.p, c->in_mass.p, c->in_type.p,
                       c->in_cost.p, n, ordered ? c->s_idx.p : (const unsigned int *)nullptr, nthr, c->dom[0], c->dom[1], c->dom[2], fac21,
                       t.child.p, t.leaf.p, c->cfg.n_gravs, t2g_packed, t.leaf_sums.p);
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  HIP_TRY(c, hipGetLastError());
  *dev_sums = t.leaf_sums.p;
  *count = words;
  return NGRAVS_OK;
}

// records of the own particles grouped by destination; mask[i] = bit per receiving task (set by the caller's kernel)
static int dd_pack_masked(ngravs_ctx *c, int what, int nranks, int64_t *counts, void **dev_records, int64_t *nrec, DevBuf<int> *slot_row = nullptr)
{
  const long long n = c->n_local;
  unsigned nb = (unsigned)((n + 255) / 256);
  std::vector<unsigned long long> h(65, 0), offs(65, 0);
  if(n > 0)
    {
      hipLaunchKernelGGL(k_dd_count, dim3(nb < DD_CNT_BLOCKS ? nb : DD_CNT_BLOCKS), dim3(256), 0, c->stream, c->dd_mask.p, n, nranks, c->dd_counts.p);
      HIP_TRY(c, hipMemcpyAsync(h.data(), c->dd_counts.p, sizeof(unsigned long long) * 65, hipMemcpyDeviceToHost, c->stream));
      HIP_TRY(c, hipStreamSynchronize(c->stream));
    }
  long long tot = 0;
  for(int r = 0; r < nranks; r++)
    {
      offs[r] = tot;
      counts[r] = (int64_t)h[r];
      tot += h[r];
    }
  const int rdbl = dd_record_doubles(c, what);
  if(c->dd_send.ensure((size_t)(tot > 0 ? tot : 1) * sizeof(double) * rdbl) || (slot_row && slot_row->ensure((size_t)(tot > 0 ? tot : 1))))
    return NGRAVS_ERR_NOMEM;
  if(tot > 0)
    {
      HIP_TRY(c, hipMemcpyAsync(c->dd_counts.p + 65, offs.data(), sizeof(unsigned long long) * 65, hipMemcpyHostToDevice, c->stream));
      hipLaunchKernelGGL(k_dd_fill, dim3((unsigned)((n + DD_FILL_THREADS - 1) / DD_FILL_THREADS)), dim3(DD_FILL_THREADS), 0, c->stream, c->dd_mask.p, n, c->in_pos.p, c->in_mass.p, c->in_type.p,
                         c->in_oldacc.p, c->in_active.p, c->in_id.p, c->in_cost.p, c->dd_counts.p + 65, c->dd_counts.p + 130,
                         (double *)c->dd_send.p, (what == 0 && c->pm_parked) ? c->pm_orig.p : (const double *)nullptr, rdbl,
                         slot_row ? slot_row->p : (int *)nullptr);
    }
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  HIP_TRY(c, hipGetLastError());
  *dev_records = c->dd_send.p;
  *nrec = tot;
  c->dd_last_what = what;
  c->dd_last_sent = tot;
  return NGRAVS_OK;
}

// migration (what = 0): every own particle whose leaf belongs to another task goes there
int dd_pack(ngravs_ctx *c, int what, const int *leaf_owner, int nranks, int me, int64_t *counts, void **dev_records, int64_t *nrec)
{
  TopTree &t = c->top;
  if(nranks > 64 || what != 0 || t.h.nnode < 1)
    return NGRAVS_ERR_ARG;
  const long long n = c->n_local;
  if(c->dd_mask.ensure(n > 0 ? n : 1) || t.leaf_owner.ensure((size_t)t.h.nleaf) || c->dd_counts.ensure(3 * 65 + 2))
    return NGRAVS_ERR_NOMEM;
  HIP_TRY(c, hipMemcpyAsync(t.leaf_owner.p, leaf_owner, sizeof(int) * (size_t)t.h.nleaf, hipMemcpyHostToDevice, c->stream));
  t.h_leaf_owner.assign(leaf_owner, leaf_owner + t.h.nleaf);
  t.own_leaf_n = -1;   // a new cut: nothing is kept until its leaves have been packed
  HIP_TRY(c, hipMemsetAsync(c->dd_counts.p, 0, sizeof(unsigned long long) * (3 * 65 + 2), c->stream));
  double fac21;
  dd_fac(c, &fac21);
  if(n > 0)
    hipLaunchKernelGGL(k_dd_dest, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, c->in_pos.p, n, c->dom[0], c->dom[1], c->dom[2], fac21,
                       t.child.p, t.leaf.p, t.leaf_owner.p, (const unsigned long long *)nullptr, me, c->dd_mask.p, (int *)nullptr);
  return dd_pack_masked(c, 0, nranks, counts, dev_records, nrec);
}

// ---- the global top of the tree (multi-task) ---------------------------------------------------------------------------
// smallest ErrTolForceAcc * OldAcc and smallest softening length over the own ACTIVE particles: what the conservative
// opening tests of a whole domain need (the group walk uses the same two minima per group)
__global__ void k_dd_bounds(const double *__restrict__ oldacc, const int *__restrict__ type, const unsigned char *__restrict__ active,
                            long long n, WalkParams wp, double *__restrict__ out, int all)
{
  double a = 1e300, h = 1e300;
  for(long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
    if((active[i] & 1) || all)   // (all: a decomposition that will be kept -- every own particle is a target of one of its steps)
      {
        a = fmin(a, wp.errtol_acc * oldacc[i]);
        h = fmin(h, wp.fsoft[type[i]]);
      }
  for(int off = 32; off > 0; off >>= 1)
    {
      a = fmin(a, __shfl_down(a, off));
      h = fmin(h, __shfl_down(h, off));
    }
  if((threadIdx.x & 63) == 0)
    {
      // doubles >= 0 order like their bit patterns; an atomic only where the wave lowers the bound (the two words are shared by
      // the whole grid: unconditional atomics queue up there)
      const unsigned long long ua = (unsigned long long)__double_as_longlong(a), uh = (unsigned long long)__double_as_longlong(h);
      if(ua < __atomic_load_n((unsigned long long *)&out[0], __ATOMIC_RELAXED))
        atomicMin((unsigned long long *)&out[0], ua);
      if(uh < __atomic_load_n((unsigned long long *)&out[1], __ATOMIC_RELAXED))
        atomicMin((unsigned long long *)&out[1], uh);
    }
}

int dd_target_bounds(ngravs_ctx *c, double out[2])
{
  if(c->red_tmp.ensure(2))
    return NGRAVS_ERR_NOMEM;
  double big[2] = {1e300, 1e300};
  HIP_TRY(c, hipMemcpyAsync(c->red_tmp.p, big, sizeof(big), hipMemcpyHostToDevice, c->stream));
  WalkParams wp;
  make_walk_params(c, &wp);
  if(c->n_local > 0)
    hipLaunchKernelGGL(k_dd_bounds, dim3(2048), dim3(256), 0, c->stream, c->in_oldacc.p, c->in_type.p, c->in_active.p, (long long)c->n_local,
                       wp, c->red_tmp.p, c->tune.dd_keep > 0 ? 1 : 0);
  HIP_TRY(c, hipMemcpyAsync(out, c->red_tmp.p, sizeof(big), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  return NGRAVS_OK;
}

// destination task of every local particle (host array): what the migration pack would do, without packing
int dd_get_dest(ngravs_ctx *c, const int *leaf_owner, int *dest)
{
  TopTree &t = c->top;
  const long long n = c->n_local;
  if(t.h.nnode < 1)
    return NGRAVS_ERR_STATE;
  t.h_leaf_owner.assign(leaf_owner, leaf_owner + t.h.nleaf);   // (also on a task without particles: the cut is what a kept step goes by)
  t.own_leaf_n = -1;
  if(n <= 0)
    return NGRAVS_OK;
  DevBuf<int> d;
  if(t.leaf_owner.ensure((size_t)t.h.nleaf) || d.ensure(n))
    return NGRAVS_ERR_NOMEM;
  HIP_TRY(c, hipMemcpyAsync(t.leaf_owner.p, leaf_owner, sizeof(int) * (size_t)t.h.nleaf, hipMemcpyHostToDevice, c->stream));
  t.h_leaf_owner.assign(leaf_owner, leaf_owner + t.h.nleaf);
  t.own_leaf_n = -1;
  double fac21;
  dd_fac(c, &fac21);
  hipLaunchKernelGGL(k_dd_dest, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, c->in_pos.p, n, c->dom[0], c->dom[1], c->dom[2],
                     fac21, t.child.p, t.leaf.p, t.leaf_owner.p, (const unsigned long long *)nullptr, 0, (unsigned long long *)nullptr, d.p);
  HIP_TRY(c, hipMemcpyAsync(dest, d.p, sizeof(int) * n, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  d.release();
  return NGRAVS_OK;
}

// leaf import: the records of the own particles of every leaf another task asked for (what = 2 of the pack family)
int dd_pack_leaves(ngravs_ctx *c, const unsigned long long *reqmask, int nranks, int me, int64_t *counts, void **dev_records, int64_t *nrec)
{
  TopTree &t = c->top;
  if(nranks > 64 || t.h.nnode < 1)
    return NGRAVS_ERR_ARG;
  const long long n = c->n_local;
  if(c->dd_mask.ensure(n > 0 ? n : 1) || t.reqmask.ensure((size_t)t.h.nleaf) || c->dd_counts.ensure(3 * 65 + 2) || t.own_leaf.ensure(n > 0 ? n : 1))
    return NGRAVS_ERR_NOMEM;
  HIP_TRY(c, hipMemcpyAsync(t.reqmask.p, reqmask, sizeof(unsigned long long) * (size_t)t.h.nleaf, hipMemcpyHostToDevice, c->stream));
  HIP_TRY(c, hipMemsetAsync(c->dd_counts.p, 0, sizeof(unsigned long long) * (3 * 65 + 2), c->stream));
  double fac21;
  dd_fac(c, &fac21);
  // (the leaf of every own row is kept: on the steps that keep this decomposition the rows are the same, their positions are not)
  if(n > 0)
    hipLaunchKernelGGL(k_dd_dest, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, c->in_pos.p, n, c->dom[0], c->dom[1], c->dom[2], fac21,
                       t.child.p, t.leaf.p, (const int *)nullptr, t.reqmask.p, me, c->dd_mask.p, (int *)nullptr, t.own_leaf.p);
  t.own_leaf_n = n;
  t.kept_rank = me;
  t.kept_world = nranks;
  int rc = dd_pack_masked(c, 2, nranks, counts, dev_records, nrec, &t.kept_row);
  t.kept_counts.assign(counts, counts + nranks);
  t.kept_total = rc ? -1 : *nrec;
  return rc;
}

// ---- kept decomposition: the same cut, the same requests, drifted particles (ngravs_host_kept_step) ---------------------------
// force_update_pseudoparticles (forcetree.c:753) + the role of the export / import loop on a step that keeps domain and tree.
static int kept_ok(const ngravs_ctx *c)
{
  const TopTree &t = c->top;
  return t.on && t.h.nnode > 0 && t.own_leaf_n == c->n_local && t.kept_rank >= 0 && (int)t.h_leaf_owner.size() == t.h.nleaf &&
         (int)t.h_present.size() == t.h.nleaf;
}

// the records of the own particles of every leaf another task asked for at the decomposition: SAME rows in the SAME slots (the
// receivers refresh their copies row by row), new positions.  The slots are those k_dd_fill handed out then (t.kept_row).
__global__ void k_dd_fill_kept(const int *__restrict__ slot_row, long long nslot, const double *__restrict__ pos, const double *__restrict__ mass,
                               const int *__restrict__ type, const double *__restrict__ oldacc, const unsigned char *__restrict__ active,
                               const long long *__restrict__ id, const double *__restrict__ cost, double *__restrict__ out)
{
  const long long s = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if(s >= nslot)
    return;
  const long long i = slot_row[s];
  double *o = out + (size_t)s * 7;
  o[0] = pos[3 * i + 0];
  o[1] = pos[3 * i + 1];
  o[2] = pos[3 * i + 2];
  o[3] = mass[i];
  o[4] = oldacc[i];
  o[5] = cost[i];
  o[6] = __longlong_as_double((long long)type[i] | ((long long)(active[i] & 1) << 8) | (id[i] << 16));
}

int dd_pack_leaves_kept(ngravs_ctx *c, int64_t *counts, void **dev_records, int64_t *nrec)
{
  TopTree &t = c->top;
  if(!kept_ok(c) || t.kept_total < 0 || (int)t.kept_counts.size() != t.kept_world)
    return NGRAVS_ERR_STATE;
  const long long tot = t.kept_total;
  if(c->dd_send.ensure((size_t)(tot > 0 ? tot : 1) * sizeof(double) * 7))
    return NGRAVS_ERR_NOMEM;
  if(tot > 0)
    hipLaunchKernelGGL(k_dd_fill_kept, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, c->stream, t.kept_row.p, tot, c->in_pos.p, c->in_mass.p,
                       c->in_type.p, c->in_oldacc.p, c->in_active.p, c->in_id.p, c->in_cost.p, (double *)c->dd_send.p);
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  HIP_TRY(c, hipGetLastError());
  for(int r = 0; r < t.kept_world; r++)
    counts[r] = t.kept_counts[r];
  *dev_records = c->dd_send.p;
  *nrec = tot;
  c->dd_last_what = 2;
  c->dd_last_sent = tot;
  return NGRAVS_OK;
}

// per leaf, by the membership of the decomposition: the sums of ngravs_dd_leaf_sums + one more word, the grown side of the leaf's
// cell in the tree of its OWNER (every other task adds 0: the sum over the tasks is the owner's value)
int dd_leaf_sums_kept(ngravs_ctx *c, void **dev_sums, int64_t *count)
{
  TopTree &t = c->top;
  if(!kept_ok(c))
    return NGRAVS_ERR_STATE;
  const int cwk = TOP_CW(c->cfg.n_gravs) + 1;
  const long long n = c->n_local, words = (long long)t.h.nleaf * cwk + 1;
  if(t.kept_sums.ensure((size_t)words))
    return NGRAVS_ERR_NOMEM;
  HIP_TRY(c, hipMemsetAsync(t.kept_sums.p, 0, sizeof(double) * words, c->stream));
  unsigned t2g_packed = 0;
  for(int ty = 0; ty < NGRAVS_NTYPES; ty++)
    t2g_packed |= ((unsigned)(c->cfg.type_to_grav[ty] & 3)) << (2 * ty);
  if(n > 0)
    hipLaunchKernelGGL(k_dd_leafsums, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, c->in_pos.p, c->in_mass.p, c->in_type.p,
                       c->in_cost.p, n, (const unsigned int *)nullptr, n, c->dom[0], c->dom[1], c->dom[2], 0.0, t.child.p, t.leaf.p,
                       c->cfg.n_gravs, t2g_packed, t.kept_sums.p, t.own_leaf.p, cwk);
  HIP_TRY(c, hipGetLastError());
  int rc = tree_top_leaf_len(c, t.kept_sums.p, cwk);
  if(rc)
    return rc;
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  *dev_sums = t.kept_sums.p;
  *count = words;
  return NGRAVS_OK;
}

// the imported copies take the positions (and masses) their owners hold now: the records arrive in the order of the
// decomposition's import (ngravs_dd_set_halo), rows n_local ... of the input columns
int dd_refresh_halo(ngravs_ctx *c, const void *dev_records, int64_t nrec)
{
  if(!kept_ok(c) || nrec != c->n - c->n_local)
    return NGRAVS_ERR_STATE;
  if(nrec > 0)
    hipLaunchKernelGGL(k_dd_unpack, dim3((unsigned)((nrec + 255) / 256)), dim3(256), 0, c->stream, (const double *)dev_records, 7,
                       (long long)nrec, (long long)c->n_local, 1, c->in_pos.p, c->in_mass.p, c->in_type.p, c->in_oldacc.p, c->in_active.p, c->in_id.p,
                       c->in_cost.p, (double *)nullptr);
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  HIP_TRY(c, hipGetLastError());
  c->tree_stale = true;
  return NGRAVS_OK;
}

// new global sums of every top node (counts and presence are those of the decomposition) and the grown sides of the leaves' cells
int dd_update_top(ngravs_ctx *c, const double *node_sums, const double *leaf_len)
{
  TopTree &t = c->top;
  if(!kept_ok(c) || !node_sums || !leaf_len || !c->have_tree)
    return NGRAVS_ERR_STATE;
  const int cw = TOP_CW(c->cfg.n_gravs), nn = t.h.nnode;
  if(t.leaf_len.ensure((size_t)t.h.nleaf))
    return NGRAVS_ERR_NOMEM;
  HIP_TRY(c, hipMemcpyAsync(t.gsum.p, node_sums, sizeof(double) * (size_t)nn * cw, hipMemcpyHostToDevice, c->stream));
  HIP_TRY(c, hipMemcpyAsync(t.leaf_len.p, leaf_len, sizeof(double) * (size_t)t.h.nleaf, hipMemcpyHostToDevice, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  return tree_top_refit(c);
}

// The global sums of every top node (all-reduced leaf sums, added up the tree by the host) and which leaves are present on
// this task (own or imported): uploaded for the tree build.  A node is PARTIAL if a leaf below it holds particles that are
// not here.
int dd_set_top(ngravs_ctx *c, const double *node_sums, const unsigned char *present)
{
  TopTree &t = c->top;
  if(!node_sums || !present)
    {
      t.on = false;
      return NGRAVS_OK;
    }
  if(t.h.nnode < 1)
    return NGRAVS_ERR_STATE;
  const int cw = TOP_CW(c->cfg.n_gravs), nn = t.h.nnode;
  std::vector<int> cnt((size_t)nn);
  std::vector<unsigned char> info((size_t)nn);
  for(int i = nn - 1; i >= 0; i--)   // children have larger indices than their parent
    {
      cnt[i] = (int)(node_sums[(size_t)i * cw] + 0.5);
      unsigned char part = 0;
      if(t.h.child[i] < 0)
        part = (cnt[i] > 0 && !present[t.h.leaf[i]]) ? 1 : 0;
      else
        for(int k = 0; k < 8; k++)
          part |= (info[t.h.child[i] + k] >> 3) & 1;
      const int *xyz = t.h.xyz + 3 * (size_t)i;
      info[i] = (unsigned char)(((xyz[0] & 1) << 2) | ((xyz[1] & 1) << 1) | (xyz[2] & 1) | (part << 3));
    }
  if(t.gcnt.ensure((size_t)nn) || t.gsum.ensure((size_t)nn * cw) || t.info.ensure((size_t)nn))
    return NGRAVS_ERR_NOMEM;
  HIP_TRY(c, hipMemcpyAsync(t.gcnt.p, cnt.data(), sizeof(int) * (size_t)nn, hipMemcpyHostToDevice, c->stream));
  HIP_TRY(c, hipMemcpyAsync(t.gsum.p, node_sums, sizeof(double) * (size_t)nn * cw, hipMemcpyHostToDevice, c->stream));
  HIP_TRY(c, hipMemcpyAsync(t.info.p, info.data(), (size_t)nn, hipMemcpyHostToDevice, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  t.on = true;
  t.h_present.assign(present, present + t.h.nleaf);
  t.h_node_sums.assign(node_sums, node_sums + (size_t)nn * cw);
  t.total_count = cnt.empty() ? 0.0 : (double)cnt[0];
  t.import_reach = c->cfg.walk_mode == NGRAVS_WALK_GROUP ? fmin(6.0, c->cfg.group_reach > 0 ? c->cfg.group_reach : NGRAVS_GROUP_REACH) : 6.0;
  c->have_tree = false;
  return NGRAVS_OK;
}

// after the migration all-to-all: drop what was sent (mask != 0 from the last dd_pack(what=0)), append what arrived
int dd_apply_migration(ngravs_ctx *c, const void *dev_records, int64_t nrec)
{
  if(c->dd_last_what != 0)
    return NGRAVS_ERR_STATE;
  const long long n = c->n_local;
  if(nrec == 0 && c->dd_last_sent == 0)   // steady state: nothing left, nothing arrived -- the columns stay where they are
    {
      c->n = c->n_local;
      c->have_order = c->have_tree = c->have_pm = c->have_acc = false;
      return NGRAVS_OK;
    }
  DevBuf<double> pos2, mass2, old2, cost2, pm2;
  DevBuf<int> type2;
  DevBuf<unsigned char> act2;
  DevBuf<long long> id2;
  const size_t cap = (size_t)(n + nrec + 64);
  const int rdbl = dd_record_doubles(c, 0);
  const bool with_pm = rdbl > 7;   // TreePM: P[].GravPM moves with the particle (zeros while no PM force exists)
  if(pos2.ensure(3 * cap) || mass2.ensure(cap) || old2.ensure(cap) || type2.ensure(cap) || act2.ensure(cap) || id2.ensure(cap) ||
     cost2.ensure(cap) || (with_pm && pm2.ensure(3 * cap)))
    return NGRAVS_ERR_NOMEM;
  HIP_TRY(c, hipMemsetAsync(c->dd_counts.p + 195, 0, sizeof(unsigned long long), c->stream));
  if(n > 0)
    hipLaunchKernelGGL(k_dd_keep, dim3((unsigned)((n + DD_FILL_THREADS - 1) / DD_FILL_THREADS)), dim3(DD_FILL_THREADS), 0, c->stream, c->dd_mask.p, n, c->in_pos.p,
                       c->in_mass.p, c->in_type.p, c->in_oldacc.p, c->in_active.p, c->in_id.p, c->in_cost.p, c->dd_counts.p + 195, pos2.p,
                       mass2.p, type2.p, old2.p, act2.p, id2.p, cost2.p, c->pm_parked ? c->pm_orig.p : (const double *)nullptr,
                       with_pm ? pm2.p : (double *)nullptr);
  unsigned long long kept = 0;
  HIP_TRY(c, hipMemcpyAsync(&kept, c->dd_counts.p + 195, sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  if(nrec > 0)
    hipLaunchKernelGGL(k_dd_unpack, dim3((unsigned)((nrec + 255) / 256)), dim3(256), 0, c->stream, (const double *)dev_records, rdbl,
                       (long long)nrec, (long long)kept, 0, pos2.p, mass2.p, type2.p, old2.p, act2.p, id2.p, cost2.p,
                       with_pm ? pm2.p : (double *)nullptr);
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  HIP_TRY(c, hipGetLastError());
  c->in_pos.release();
  c->in_mass.release();
  c->in_oldacc.release();
  c->in_type.release();
  c->in_active.release();
  c->in_id.release();
  c->in_cost.release();
  c->in_cost = cost2;
  if(with_pm)
    {
      c->pm_orig.release();
      c->pm_orig = pm2;   // pm_parked stays as it was: zeros are not a PM force
    }
  c->in_pos = pos2;
  c->in_mass = mass2;
  c->in_oldacc = old2;
  c->in_type = type2;
  c->in_active = act2;
  c->in_id = id2;
  c->n_local = (int64_t)kept + nrec;
  c->n = c->n_local;
  c->own_order_nlocal = -1;   // rows moved: the old order no longer names them
  c->have_order = c->have_tree = c->have_pm = c->have_acc = false;
  return NGRAVS_OK;
}

// ---- peano_hilbert_order() of the own rows (peano.c:36-90, reorder_particles :261-312) -------------------------------------------
// The reference sorts P[] itself along the curve at the end of every decomposition, so that whatever visits particles in tree
// order reads memory in order.  The caller-order columns here are the library's own copy in the library-driven decomposition
// (rows are named by ID, ngravs_get_ids): when most of the last gather's reads were jumps they are put into the Peano order of
// that decomposition, once -- afterwards the order only drifts with the particles, and the per-leaf sums, the key pass and the
// gather of the following steps read coalesced.
struct OwnRow
{
  unsigned int nl;
  __device__ bool operator()(unsigned int row) const { return row < nl; }
};

__global__ void k_dd_reorder(const unsigned int *__restrict__ rows, long long nl, const double *__restrict__ pos,
                             const double *__restrict__ mass, const int *__restrict__ type, const double *__restrict__ oldacc,
                             const unsigned char *__restrict__ active, const long long *__restrict__ id, const double *__restrict__ cost,
                             const double *__restrict__ pm, double *__restrict__ pos2, double *__restrict__ mass2, int *__restrict__ type2,
                             double *__restrict__ oldacc2, unsigned char *__restrict__ active2, long long *__restrict__ id2,
                             double *__restrict__ cost2, double *__restrict__ pm2)
{
  const long long k = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if(k >= nl)
    return;
  const long long i = rows[k];
  pos2[3 * k + 0] = pos[3 * i + 0];
  pos2[3 * k + 1] = pos[3 * i + 1];
  pos2[3 * k + 2] = pos[3 * i + 2];
  mass2[k] = mass[i];
  type2[k] = type[i];
  oldacc2[k] = oldacc[i];
  active2[k] = active[i];
  id2[k] = id[i];
  cost2[k] = cost[i];
  if(pm2)
    for(int j = 0; j < 3; j++)
      pm2[3 * k + j] = pm[3 * i + j];
}

// returns 1 if the rows were put in order, 0 if they were left (no order known, or close enough to it), < 0: status
int dd_peano_order_own(ngravs_ctx *c, int force)
{
  const long long nl = c->n_local, len = c->own_order_len;
  if(c->own_order_nlocal != nl || len < nl || !c->s_idx.p || nl < 1 || !c->in_id.p || !c->in_cost.p || !c->d_counters.p)
    return 0;
  int jumps = 0;
  HIP_TRY(c, hipMemcpyAsync(&jumps, c->d_counters.p + 14, sizeof(int), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  if(!force && (long long)jumps * 64 * 8 < len)   // (every 64th block counts) fewer than 1 read in 8 leaves its neighbourhood: in order
    return 0;
  // the own rows in Peano order: s_idx without the imported copies (rows >= n_local)
  if(c->idx_iota.ensure((size_t)len))
    return NGRAVS_ERR_NOMEM;
  size_t tmp_bytes = 0;
  OwnRow own;
  own.nl = (unsigned int)nl;
  (void)hipcub::DeviceSelect::If(nullptr, tmp_bytes, c->s_idx.p, c->idx_iota.p, c->d_counters.p + 14, (int)len, own, c->stream);
  if(c->sort_tmp.ensure(tmp_bytes))
    return NGRAVS_ERR_NOMEM;
  HIP_TRY(c, hipcub::DeviceSelect::If(c->sort_tmp.p, tmp_bytes, c->s_idx.p, c->idx_iota.p, c->d_counters.p + 14, (int)len, own, c->stream));
  int nsel = 0;
  HIP_TRY(c, hipMemcpyAsync(&nsel, c->d_counters.p + 14, sizeof(int), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  if(nsel != nl)
    {
      ngravs_report(c, NGRAVS_ERR_STATE, "peano order of the own rows: the stored order does not name every own row once");
      return NGRAVS_ERR_STATE;
    }
  DevBuf<double> pos2, mass2, old2, cost2, pm2;
  DevBuf<int> type2;
  DevBuf<unsigned char> act2;
  DevBuf<long long> id2;
  const size_t cap = (size_t)nl + 64;
  const bool with_pm = c->pm_parked && c->pm_orig.p;
  if(pos2.ensure(3 * cap) || mass2.ensure(cap) || old2.ensure(cap) || type2.ensure(cap) || act2.ensure(cap) || id2.ensure(cap) ||
     cost2.ensure(cap) || (with_pm && pm2.ensure(3 * cap)))
    {
      pos2.release(), mass2.release(), old2.release(), type2.release(), act2.release(), id2.release(), cost2.release(), pm2.release();
      return NGRAVS_ERR_NOMEM;
    }
  hipLaunchKernelGGL(k_dd_reorder, dim3((unsigned)((nl + 255) / 256)), dim3(256), 0, c->stream, c->idx_iota.p, nl, c->in_pos.p, c->in_mass.p,
                     c->in_type.p, c->in_oldacc.p, c->in_active.p, c->in_id.p, c->in_cost.p, with_pm ? c->pm_orig.p : (const double *)nullptr,
                     pos2.p, mass2.p, type2.p, old2.p, act2.p, id2.p, cost2.p, with_pm ? pm2.p : (double *)nullptr);
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  HIP_TRY(c, hipGetLastError());
  c->in_pos.release(), c->in_mass.release(), c->in_oldacc.release(), c->in_type.release(), c->in_active.release(), c->in_id.release(),
      c->in_cost.release();
  c->in_pos = pos2, c->in_mass = mass2, c->in_oldacc = old2, c->in_type = type2, c->in_active = act2, c->in_id = id2, c->in_cost = cost2;
  if(with_pm)
    {
      c->pm_orig.release();
      c->pm_orig = pm2;
    }
  c->n = nl;                  // the imported copies of the last step are gone with the order that named them
  c->own_order_nlocal = -1;   // the rows ARE in that order now
  c->have_order = c->have_tree = c->have_acc = false;
  return 1;
}

int dd_set_halo(ngravs_ctx *c, const void *dev_records, int64_t nrec)
{
  const long long nl = c->n_local, tot = nl + nrec;
  if(grow_keep(c, c->in_pos, 3 * nl, 3 * tot) || grow_keep(c, c->in_mass, nl, tot) || grow_keep(c, c->in_oldacc, nl, tot) ||
     grow_keep(c, c->in_type, nl, tot) || grow_keep(c, c->in_active, nl, tot) || grow_keep(c, c->in_id, nl, tot) ||
     grow_keep(c, c->in_cost, nl, tot))
    return NGRAVS_ERR_NOMEM;
  if(nrec > 0)
    hipLaunchKernelGGL(k_dd_unpack, dim3((unsigned)((nrec + 255) / 256)), dim3(256), 0, c->stream, (const double *)dev_records, 7,
                       (long long)nrec, nl, 1, c->in_pos.p, c->in_mass.p, c->in_type.p, c->in_oldacc.p, c->in_active.p, c->in_id.p,
                       c->in_cost.p, (double *)nullptr);
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  HIP_TRY(c, hipGetLastError());
  c->n = tot;
  c->have_order = c->have_tree = c->have_pm = c->have_acc = false;
  return NGRAVS_OK;
}

int dd_fill_ids(ngravs_ctx *c)
{
  if(c->in_id.ensure(c->n > 0 ? c->n : 1))
    return NGRAVS_ERR_NOMEM;
  if(c->n > 0)
    hipLaunchKernelGGL(k_dd_iota, dim3((unsigned)((c->n + 255) / 256)), dim3(256), 0, c->stream, c->in_id.p, (long long)c->n);
  return NGRAVS_OK;
}
