// kernels_tree.hip -- oct-tree construction and per-species monopole moments on the GPU.
//
// Replaces (reference): force_treebuild_single (forcetree.c:93-281: sequential insertion, one
// particle per leaf) and force_update_node_recursive (forcetree.c:451-743: per-species mass and
// centre of mass, softening bit-flags).
//
// MI355X design: the reference's root cube is the Peano key cube (forcetree.c:106-108 vs
// domain.c:916-923), so an octree cell at depth d is exactly the set of particles sharing the top
// 3d key bits, and in the key-sorted particle array every cell is one contiguous range.  The tree
// is therefore built top-down, one level per launch, one thread per node: a node splits its range
// at the 7 digit boundaries (linear scan for small ranges, binary search for large ones);
// sub-ranges of >=2 particles become nodes of the next level (indices from a prefix sum, so the
// node array is level-contiguous and deterministic), single particles become leaf children --
// the same topology as the reference's insertion tree (one particle per leaf, chains included)
// below its top-level domain grid.  Cell centres use the reference's own recurrence
// (centre +- len/4 per level), so they are bit-identical.  Cells that are still shared at the
// 21st level (below the reference's 18-bit key resolution) become bucket leaves.
// Moments are accumulated bottom-up, one launch per level, fp64 accumulators as forcetree.c:470.
#include "engine.hpp"
#include <hipcub/hipcub.hpp>


__device__ __forceinline__ int key_digit(unsigned long long k, int level)   // digit deciding the child of a level-`level` node
{
  return (int)((k >> (3 * (TREE_BITS - 1 - level))) & 7ull);
}

// child[] encoding after this kernel: -1 empty, -2-p particle p, >=0 : start particle of a sub-range (fixed up in k_link)
// Multi-task trees: for a node that is a SPLIT node of the global top tree (n_top[node] = t, t_child[t] >= 0) the KIND of every
// child follows from the GLOBAL particle count of the child's top node (t_gcnt[t_child[t] + k]) -- 0: empty, 1: particle leaf,
// >= 2: node -- whatever part of it is present on this task; so the topology of the top is the single-task tree's
// (forcetree.c:292-431 builds the same top-level nodes from the global TopNodes on every task).
#define TB 128   // nodes per (virtual) block of the build kernels

// The build never returns to the host between levels: lv[2 l], lv[2 l + 1] = first node and node count of level l live on the
// device (k_scan_blocks writes the next level's), the kernels loop over virtual blocks of TB consecutive nodes with whatever
// grid they were launched with, and node indices of the next level come from a two-stage scan (per-block sums here, the scan
// of the block sums in k_scan_blocks, the scan inside a block in k_link) -- deterministic, level-contiguous numbering.
__global__ __launch_bounds__(TB) void k_split(const unsigned long long *__restrict__ key, const int *__restrict__ n_first,
                                              const int *__restrict__ n_count, const int *__restrict__ lv, int level,
                                              int *__restrict__ n_child, int *__restrict__ n_nchild, int *__restrict__ blocksum,
                                              const int *__restrict__ n_top = nullptr, const int *__restrict__ t_child = nullptr,
                                              const int *__restrict__ t_gcnt = nullptr)
{
  __shared__ int wsum[TB / 64];
  const int node0 = lv[2 * level], nnodes_level = lv[2 * level + 1];
  const int nvb = (nnodes_level + TB - 1) / TB;
  for(int vb = blockIdx.x; vb < nvb; vb += gridDim.x)
    {
      const int t = vb * TB + threadIdx.x;
      int nn = 0;
      if(t < nnodes_level)
        {
          int node = node0 + t;
          int f = n_first[node], cnt = n_count[node];
          int b[9];
          b[0] = f;
          b[8] = f + cnt;
          if(cnt <= 24)
            {
              int k = 1;
              for(int i = f; i < f + cnt; i++)
                {
                  int d = key_digit(key[i], level);
                  while(k <= d)
                    b[k++] = i;
                }
              while(k < 8)
                b[k++] = f + cnt;
            }
          else
            {
              for(int k = 1; k < 8; k++)
                {
                  int lo = b[k - 1], hi = f + cnt;   // first index with digit >= k
                  while(lo < hi)
                    {
                      int mid = (lo + hi) >> 1;
                      if(key_digit(key[mid], level) < k)
                        lo = mid + 1;
                      else
                        hi = mid;
                    }
                  b[k] = lo;
                }
            }
          int tc = -1;                                               // first child top node if this node is a split node of the top tree
          if(n_top)
            {
              const int t = n_top[node];
              if(t >= 0)
                tc = t_child[t];
            }
          for(int k = 0; k < 8; k++)
            {
              int c = b[k + 1] - b[k], v;
              int kind = c;                                          // 0 empty, 1 particle, >= 2 node
              if(tc >= 0)
                {
                  kind = t_gcnt[tc + k];
                  if(kind == 1 && c != 1)
                    kind = 0;                                        // a single particle that lives elsewhere: its parent is never opened here
                }
              if(kind == 0)
                v = -1;
              else if(kind == 1)
                v = -2 - b[k];
              else
                {
                  v = b[k];
                  nn++;
                }
              n_child[8 * (long long)node + k] = v;
            }
          n_nchild[t] = nn;
        }
      int s = nn;
      for(int off = 32; off > 0; off >>= 1)
        s += __shfl_down(s, off);
      __syncthreads();
      if((threadIdx.x & 63) == 0)
        wsum[threadIdx.x >> 6] = s;
      __syncthreads();
      if(threadIdx.x == 0)
        {
          int tot = 0;
          for(int w = 0; w < TB / 64; w++)
            tot += wsum[w];
          blocksum[vb] = tot;
        }
    }
}

// exclusive scan of the block sums of one level (one workgroup), and the next level's extent; out of nodes -> lv_err
__global__ __launch_bounds__(1024) void k_scan_blocks(int *__restrict__ blocksum, int *__restrict__ lv, int level, int maxn)
{
  __shared__ int part[1024];
  const int node0 = lv[2 * level], cnt = lv[2 * level + 1];
  const int nvb = (cnt + TB - 1) / TB;
  const int per = (nvb + 1023) / 1024;
  const int lo = threadIdx.x * per, hi = lo + per < nvb ? lo + per : nvb;
  int s = 0;
  for(int i = lo; i < hi; i++)
    s += blocksum[i];
  part[threadIdx.x] = s;
  __syncthreads();
  for(int off = 1; off < 1024; off <<= 1)   // Hillis-Steele inclusive scan of the 1024 partial sums
    {
      int v = threadIdx.x >= off ? part[threadIdx.x - off] : 0;
      __syncthreads();
      part[threadIdx.x] += v;
      __syncthreads();
    }
  int run = threadIdx.x ? part[threadIdx.x - 1] : 0;
  for(int i = lo; i < hi; i++)
    {
      const int v = blocksum[i];
      blocksum[i] = run;
      run += v;
    }
  if(threadIdx.x == 0)
    {
      int next_cnt = part[1023];
      if((long long)node0 + cnt + next_cnt > (long long)maxn)
        {
          lv[2 * (MAX_LEVELS + 2)] = 1;   // maximum number of tree nodes reached (forcetree.c:247)
          next_cnt = 0;
        }
      if(level + 1 > TREE_BITS)
        next_cnt = 0;
      lv[2 * (level + 1)] = node0 + cnt;
      lv[2 * (level + 1) + 1] = next_cnt;
    }
}

// give the sub-ranges their node indices and initialise the nodes of the next level
__global__ __launch_bounds__(TB) void k_link(const double4 *__restrict__ s_pm, int *__restrict__ n_first, int *__restrict__ n_count,
                                             int *__restrict__ n_child, double4 *__restrict__ n_geo, int *__restrict__ n_flags,
                                             const int *__restrict__ n_nchild, const int *__restrict__ blocksum, const int *__restrict__ lv,
                                             int level, double cx, double cy, double cz, double fac21, int *__restrict__ n_top = nullptr,
                                             const int *__restrict__ t_child = nullptr, const unsigned char *__restrict__ t_info = nullptr)
{
  typedef hipcub::BlockScan<int, TB> BlockScan;
  __shared__ typename BlockScan::TempStorage tmp;
  const int node0 = lv[2 * level], nnodes_level = lv[2 * level + 1], next0 = lv[2 * (level + 1)];
  const int nvb = (nnodes_level + TB - 1) / TB;
  for(int vb = blockIdx.x; vb < nvb; vb += gridDim.x)
    {
      const int t = vb * TB + threadIdx.x;
      const int mine = t < nnodes_level ? n_nchild[t] : 0;
      int excl;
      BlockScan(tmp).ExclusiveSum(mine, excl);
      __syncthreads();
      if(t >= nnodes_level || mine == 0)
        continue;
      int node = node0 + t;
      int end = n_first[node] + n_count[node];
      double4 g = n_geo[node];
      int nxt = next0 + blocksum[vb] + excl;
      int tc = -1;   // first child top node if this node is a split node of the global top tree
      if(n_top)
        {
          const int t = n_top[node];
          if(t >= 0)
            tc = t_child[t];
        }
      int starts[8], vals[8];
      for(int k = 0; k < 8; k++)
        {
          int v = n_child[8 * (long long)node + k];
          vals[k] = v;
          starts[k] = v >= 0 ? v : (v <= -2 ? -2 - v : -1);
        }
      for(int k = 0; k < 8; k++)
        {
          if(vals[k] < 0)
            continue;
          int e = end;
          for(int kk = k + 1; kk < 8; kk++)
            if(starts[kk] >= 0)
              {
                e = starts[kk];
                break;
              }
          int cn = nxt++;
          n_child[8 * (long long)node + k] = cn;
          n_first[cn] = starts[k];
          n_count[cn] = e - starts[k];
          // geometric octant of the child from its first particle's cell coordinates (or, in the global top of a multi-task
          // tree, where a cell may hold no local particle, from the cell table); centre recurrence of forcetree.c:190-206
          // (centre +- 0.25*len of the parent)
          int ox, oy, oz, fl = (level + 1 >= TREE_BITS) ? FLAG_BUCKET : 0;
          if(n_top)
            n_top[cn] = tc >= 0 ? tc + k : -1;
          if(tc >= 0)
            {
              const int ct = tc + k, inf = t_info[ct];
              ox = (inf >> 2) & 1;
              oy = (inf >> 1) & 1;
              oz = inf & 1;
              if(inf & 8)
                fl |= FLAG_PARTIAL;
              if(t_child[ct] < 0 && e - starts[k] == 0)
                fl |= FLAG_PSEUDO;          // a top leaf all of whose particles live on other tasks (forcetree.c:345-431 pseudo particle)
            }
          else
            {
              double4 p = s_pm[starts[k]];
              int ix = (int)__dmul_rn(__dsub_rn(p.x, cx), fac21);
              int iy = (int)__dmul_rn(__dsub_rn(p.y, cy), fac21);
              int iz = (int)__dmul_rn(__dsub_rn(p.z, cz), fac21);
              int sh = TREE_BITS - 1 - level;
              ox = (ix >> sh) & 1;
              oy = (iy >> sh) & 1;
              oz = (iz >> sh) & 1;
            }
          double q = 0.25 * g.w;
          double4 cg;
          cg.x = ox ? g.x + q : g.x - q;
          cg.y = oy ? g.y + q : g.y - q;
          cg.z = oz ? g.z + q : g.z - q;
          cg.w = 0.5 * g.w;
          n_geo[cn] = cg;
          n_flags[cn] = fl;
        }
    }
}


// =============================================================================================
//  One-pass build (single-task trees).  In the key-sorted array two neighbours i, i+1 share their first d_i key digits; a cell
//  of level l is a node iff it holds at least two particles, i.e. iff it contains a neighbouring pair with d >= l.  Hence
//  particle i is the FIRST particle of exactly the nodes of levels (d_{i-1}, d_i] (a chain when that interval holds more than
//  one level), every node is started by exactly one particle, and the level-contiguous node numbering of the level-by-level
//  build above (nodes of a level in the order of their ranges) is  level_start[l] + #{starts of level l before i}.  So the whole
//  topology follows from ONE pass over the keys plus per-level prefix counts -- 5 launches instead of 63:
//    k_tb_count   per block of 256 particles, the number of nodes each level gets from it (22 packed counters)
//    k_tb_scan    exclusive scan over the blocks, one workgroup per level
//    k_tb_levels  level table (first node, count) and the out-of-nodes check
//    k_tb_fill    empty child slots of all nodes
//    k_tb_nodes   every particle writes the nodes it starts (first particle, cell from the reference's centre recurrence
//                 replayed from the root, flags), hooks each into its parent's child slot, and hooks itself as a particle
//                 leaf into the deepest node that contains it.
//  The parent of a node of level l started at i is the level-(l-1) node containing i: the one i itself starts, or -- when
//  l - 1 = d_{i-1} -- the last level-(l-1) node started before i.  Particle counts of the nodes come from the moments pass
//  (children's counts + particle leaves); buckets (level TREE_BITS: identical keys) count their run of equal keys here.
// =============================================================================================
#define TBN 256   // threads per block of the one-pass kernels
#define TBCH 2    // chunks of TBN consecutive particles per block
__device__ __forceinline__ int common_digits(unsigned long long a, unsigned long long b)
{
  const unsigned long long x = a ^ b;
  if(x == 0)
    return TREE_BITS;
  return (__clzll((long long)x) - (64 - 3 * TREE_BITS)) / 3;
}
__device__ __forceinline__ void tb_neighbours(const unsigned long long *__restrict__ key, long long n, long long i, unsigned long long &k,
                                              int &dprev, int &dcur)
{
  k = 0;
  dprev = dcur = -1;   // beyond the end: starts nothing
  if(i < n)
    {
      k = key[i];
      dprev = i > 0 ? common_digits(key[i - 1], k) : -1;
      dcur = i + 1 < n ? common_digits(k, key[i + 1]) : -1;
    }
}
__device__ __forceinline__ int wave_imin(int v)
{
  for(int off = 32; off > 0; off >>= 1)
    {
      const int o = __shfl_xor(v, off);
      v = o < v ? o : v;
    }
  return __builtin_amdgcn_readfirstlane(v);
}
__device__ __forceinline__ int wave_imax(int v)
{
  for(int off = 32; off > 0; off >>= 1)
    {
      const int o = __shfl_xor(v, off);
      v = o > v ? o : v;
    }
  return __builtin_amdgcn_readfirstlane(v);
}

// per block (TBCH x TBN consecutive particles): how many nodes every level gets from it.  Which lanes start a node of level l
// is one ballot, the count its population count -- no per-lane counters.
__global__ __launch_bounds__(TBN) void k_tb_count(const unsigned long long *__restrict__ key, long long n, int *__restrict__ bcount, int nblk_pad)
{
  __shared__ int tot[TREE_BITS + 1];
  if(threadIdx.x <= TREE_BITS)
    tot[threadIdx.x] = 0;
  __syncthreads();
  const int lane = threadIdx.x & 63;
  for(int ch = 0; ch < TBCH; ch++)
    {
      const long long i = ((long long)blockIdx.x * TBCH + ch) * TBN + threadIdx.x;
      unsigned long long k;
      int dprev, dcur;
      tb_neighbours(key, n, i, k, dprev, dcur);
      const int l0 = wave_imin(dprev) + 1, l1 = wave_imax(dcur);
      for(int l = l0; l <= l1; l++)
        {
          const unsigned long long m = __builtin_amdgcn_ballot_w64(dprev < l && l <= dcur);
          if(lane == 0 && m)
            atomicAdd(&tot[l], __popcll(m));
        }
    }
  __syncthreads();
  if(threadIdx.x <= TREE_BITS)
    bcount[(size_t)threadIdx.x * nblk_pad + blockIdx.x] = tot[threadIdx.x];
}

// one workgroup per level: exclusive scan of the level's block counts in place (tiles of 4096, coalesced), total to lvcnt[level]
__global__ __launch_bounds__(1024) void k_tb_scan(int *__restrict__ bcount, int nblk_pad, int *__restrict__ lvcnt)
{
  typedef hipcub::BlockScan<int, 1024> BlockScan;
  __shared__ typename BlockScan::TempStorage tmp;
  int4 *row = reinterpret_cast<int4 *>(bcount + (size_t)blockIdx.x * nblk_pad);
  const int nq = nblk_pad / 4;
  int carry = 0;
  for(int t0 = 0; t0 < nq; t0 += 1024)
    {
      const int q = t0 + threadIdx.x;
      int4 v = {0, 0, 0, 0};
      if(q < nq)
        v = row[q];
      const int s = v.x + v.y + v.z + v.w;
      int ex, total;
      BlockScan(tmp).ExclusiveSum(s, ex, total);
      __syncthreads();
      if(q < nq)
        {
          int4 o;
          o.x = carry + ex;
          o.y = o.x + v.x;
          o.z = o.y + v.y;
          o.w = o.z + v.z;
          row[q] = o;
        }
      carry += total;
    }
  if(threadIdx.x == 0)
    lvcnt[blockIdx.x] = carry;
}

__global__ void k_tb_levels(const int *__restrict__ lvcnt, int *__restrict__ lv, int maxn)
{
  if(threadIdx.x != 0 || blockIdx.x != 0)
    return;
  long long run = 0;
  bool bad = false;
  for(int l = 0; l <= TREE_BITS; l++)
    run += lvcnt[l];
  if(run > (long long)maxn)
    bad = true;   // maximum number of tree nodes reached (forcetree.c:247)
  run = 0;
  for(int l = 0; l <= MAX_LEVELS + 1; l++)
    {
      const int cnt = (!bad && l <= TREE_BITS) ? lvcnt[l] : 0;
      lv[2 * l] = (int)run;
      lv[2 * l + 1] = cnt;
      run += cnt;
    }
  lv[2 * (MAX_LEVELS + 2)] = bad ? 1 : 0;
}

__global__ void k_tb_fill(int4 *__restrict__ n_child4, const int *__restrict__ lv)
{
  const long long total = (long long)lv[2 * (TREE_BITS + 1)];   // first node beyond the last level = number of nodes
  const int4 e = {-1, -1, -1, -1};
  for(long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < 2 * total; t += (long long)gridDim.x * blockDim.x)
    n_child4[t] = e;
}

__global__ __launch_bounds__(TBN) void k_tb_nodes(const unsigned long long *__restrict__ key, const double4 *__restrict__ s_pm, long long n,
                                                  const int *__restrict__ bbase, int nblk_pad, const int *__restrict__ lv,
                                                  int *__restrict__ n_first, int *__restrict__ n_count, int *__restrict__ n_child,
                                                  double4 *__restrict__ n_geo, int *__restrict__ n_flags, double4 root, double cx,
                                                  double cy, double cz, double fac21)
{
  __shared__ int base[TREE_BITS + 1];              // index of the next node of every level (running over the chunks)
  __shared__ int wtot[TBN / 64][TREE_BITS + 1];    // starts per wave and level, current chunk
  if(lv[2 * (MAX_LEVELS + 2)])
    return;
  if(threadIdx.x <= TREE_BITS)
    base[threadIdx.x] = lv[2 * threadIdx.x] + bbase[(size_t)threadIdx.x * nblk_pad + blockIdx.x];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for(int ch = 0; ch < TBCH; ch++)
    {
      const long long i = ((long long)blockIdx.x * TBCH + ch) * TBN + threadIdx.x;
      unsigned long long k;
      int dprev, dcur;
      tb_neighbours(key, n, i, k, dprev, dcur);
      const bool starter = dcur > dprev;
      // the levels this wave touches: ancestors at level dprev, starts in (dprev, dcur]
      const int l0 = wave_imin(dprev < 0 ? 0 : dprev), l1 = wave_imax(dcur > dprev ? dcur : dprev);
      if(lane <= TREE_BITS)
        wtot[wave][lane] = 0;
      for(int l = l0; l <= l1; l++)
        {
          const unsigned long long m = __builtin_amdgcn_ballot_w64(dprev < l && l <= dcur);
          if(lane == 0)
            wtot[wave][l] = __popcll(m);
        }
      __syncthreads();   // wtot of all waves, and base[] of this chunk
      int ix = 0, iy = 0, iz = 0;
      if(starter)
        {
          const double4 p = s_pm[i];
          ix = (int)__dmul_rn(__dsub_rn(p.x, cx), fac21);
          iy = (int)__dmul_rn(__dsub_rn(p.y, cy), fac21);
          iz = (int)__dmul_rn(__dsub_rn(p.z, cz), fac21);
        }
      // centre recurrence of forcetree.c:190-206 (centre +- 0.25 * len of the parent), replayed from the root: every lane
      // follows the cells of its own particle, one level per step
      double4 g = root;
      auto step = [&](int l) {
        const int sh = TREE_BITS - l;
        const double q = 0.25 * g.w;
        g.x = ((ix >> sh) & 1) ? g.x + q : g.x - q;
        g.y = ((iy >> sh) & 1) ? g.y + q : g.y - q;
        g.z = ((iz >> sh) & 1) ? g.z + q : g.z - q;
        g.w = 0.5 * g.w;
      };
      for(int l = 1; l < l0; l++)
        step(l);
      int anc = -1;      // the last node of level dprev started before this particle: it holds the pair (i - 1, i)
      int parent = -1;
      for(int l = l0; l <= l1; l++)
        {
          if(l > 0 && l >= l0)
            {
              if(l > 0)
                step(l);
            }
          const bool st = dprev < l && l <= dcur;
          const unsigned long long m = __builtin_amdgcn_ballot_w64(st);
          int wb = base[l];
          for(int w = 0; w < TBN / 64; w++)
            wb += w < wave ? wtot[w][l] : 0;
          const int rank = wb + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0));
          if(l == dprev)
            {
              anc = rank - 1;
              parent = anc;
            }
          if(st)
            {
              const int idx = rank;
              n_first[idx] = (int)i;
              n_geo[idx] = g;
              n_flags[idx] = l >= TREE_BITS ? FLAG_BUCKET : 0;
              if(l >= TREE_BITS)
                {
                  int c = 1;
                  while(i + c < n && key[i + c] == k)
                    c++;
                  n_count[idx] = c;
                }
              if(l > 0)
                n_child[8 * (long long)parent + key_digit(k, l - 1)] = idx;
              parent = idx;
            }
        }
      const int dtop = dprev > dcur ? dprev : dcur;   // level of the deepest node that contains this particle
      if(i < n && dtop < TREE_BITS)                     // (members of a bucket are reached through its range)
        n_child[8 * (long long)parent + key_digit(k, dtop)] = -2 - (int)i;
      __syncthreads();
      if(threadIdx.x <= TREE_BITS)
        {
          int t = 0;
          for(int w = 0; w < TBN / 64; w++)
            t += wtot[w][threadIdx.x];
          base[threadIdx.x] += t;
        }
      __syncthreads();
    }
}

struct SoftAcc
{
  int maxsofttype, diff;
};
__device__ __forceinline__ void soft_merge(SoftAcc &a, int t, int tdiff, const double *fsoft)
{
  // forcetree.c:571-596 / :640-658 (same rule for a child node's type and for a particle's type)
  a.diff |= tdiff;
  if(a.maxsofttype == 7)
    a.maxsofttype = t;
  else if(t != 7)
    {
      if(fsoft[t] > fsoft[a.maxsofttype])
        {
          a.maxsofttype = t;
          a.diff = 1;
        }
      else if(fsoft[t] < fsoft[a.maxsofttype])
        a.diff = 1;
    }
}

template <int NG>
__global__ void k_moments(const double4 *__restrict__ s_pm, const unsigned char *__restrict__ s_type,
                          const int *__restrict__ n_first, const int *__restrict__ n_count,
                          const int *__restrict__ n_child, const double4 *__restrict__ n_geo,
                          double4 *__restrict__ n_mom, int *__restrict__ n_flags, int node0, int nnodes_level,
                          WalkParams wp, double4 *__restrict__ geo_rw = nullptr, int *__restrict__ n_npart = nullptr,
                          int *__restrict__ n_count_rw = nullptr)
{
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  if(t >= nnodes_level)
    return;
  int node = node0 + t;
  double m[NG], sx[NG], sy[NG], sz[NG];
  int np[NG];   // Nparticles[] of NGRAVS_ACCUMULATOR (forcetree.c:471-626)
#pragma unroll
  for(int g = 0; g < NG; g++)
    {
      m[g] = sx[g] = sy[g] = sz[g] = 0;
      np[g] = 0;
    }
  SoftAcc sa;
  sa.maxsofttype = 7;
  sa.diff = 0;
  int cnt_acc = 0;   // particles below this node (one-pass build: the counts are not known before this pass)
  int fl = n_flags[node];
  // refit of a drifted tree (geo_rw != 0): the node keeps its centre but its side grows to enclose whatever its particles
  // and (already grown) child cells now reach, the role of force_update_len() (forcetree.c:1005-1122)
  const double4 geo0 = n_geo[node];
  double need = 0;
  auto add_particle_v = [&](const double4 v, const int ty) {
    if(geo_rw)
      need = fmax(need, fmax(fabs(v.x - geo0.x), fmax(fabs(v.y - geo0.y), fabs(v.z - geo0.z))));
    int gg = wp.t2g[ty];
#pragma unroll
    for(int g = 0; g < NG; g++)
      if(g == gg)
        {
          np[g]++;
          m[g] += v.w;
          sx[g] += v.w * v.x;
          sy[g] += v.w * v.y;
          sz[g] += v.w * v.z;
        }
    soft_merge(sa, ty, 0, wp.fsoft);
    cnt_acc++;
  };
  auto add_particle = [&](int p) {
    double4 v = s_pm[p];
    if(geo_rw)
      need = fmax(need, fmax(fabs(v.x - geo0.x), fmax(fabs(v.y - geo0.y), fabs(v.z - geo0.z))));
    int ty = s_type[p];
    int gg = wp.t2g[ty];
#pragma unroll
    for(int g = 0; g < NG; g++)
      if(g == gg)
        {
          np[g]++;
          m[g] += v.w;
          sx[g] += v.w * v.x;
          sy[g] += v.w * v.y;
          sz[g] += v.w * v.z;
        }
    soft_merge(sa, ty, 0, wp.fsoft);
    cnt_acc++;
  };
  if(fl & FLAG_BUCKET)
    {
      int f = n_first[node], cnt = n_count[node];
      for(int p = f; p < f + cnt; p++)
        add_particle(p);
    }
  else
    {
      // The records of a batch of children are requested first and summed afterwards, in child order (the summation order of
      // the level-by-level recursion, forcetree.c:560-640): the kernel is bound by the latency of these scattered loads, and a
      // load inside the branch that consumes it would serialise them.
      const int4 *cp = reinterpret_cast<const int4 *>(n_child + 8 * (long long)node);
      const int4 c_lo = cp[0], c_hi = cp[1];
      const int ch[8] = {c_lo.x, c_lo.y, c_lo.z, c_lo.w, c_hi.x, c_hi.y, c_hi.z, c_hi.w};
      constexpr int B = NG == 1 ? 4 : 2;   // (measured at C4, N_GRAVS = 2: 2 -> see DESIGN; 4 costs occupancy: 125 VGPRs)
#pragma unroll
      for(int k0 = 0; k0 < 8; k0 += B)
        {
          double4 rec[B][NG];   // particle: [0] = position and mass; node: its NG monopoles
          int aux[B], cnt_c[B];  // particle: type; node: flags
#pragma unroll
          for(int q = 0; q < B; q++)
            {
              const int c = ch[k0 + q];
              aux[q] = 0;
              cnt_c[q] = 0;
              if(c <= -2)
                {
                  rec[q][0] = s_pm[-2 - c];
                  aux[q] = s_type[-2 - c];
                }
              else if(c >= 0)
                {
#pragma unroll
                  for(int g = 0; g < NG; g++)
                    rec[q][g] = n_mom[(long long)c * NG + g];
                  aux[q] = n_flags[c];
                  if(n_count_rw)
                    cnt_c[q] = n_count_rw[c];
                }
            }
#pragma unroll
          for(int q = 0; q < B; q++)
            {
              const int c = ch[k0 + q];
              if(c == -1)
                continue;
              if(c <= -2)
                add_particle_v(rec[q][0], aux[q]);
              else
                {
#pragma unroll
                  for(int g = 0; g < NG; g++)
                    {
                      const double4 cm = rec[q][g];
                      if(n_npart)
                        np[g] += n_npart[(long long)c * NG + g];
                      m[g] += cm.w;
                      sx[g] += cm.w * cm.x;
                      sy[g] += cm.w * cm.y;
                      sz[g] += cm.w * cm.z;
                    }
                  const int cf = aux[q];
                  soft_merge(sa, (cf >> 2) & 7, (cf >> 5) & 1, wp.fsoft);
                  cnt_acc += cnt_c[q];
                  if(geo_rw)
                    {
                      const double4 cg = n_geo[c];
                      need = fmax(need, fmax(fabs(cg.x - geo0.x), fmax(fabs(cg.y - geo0.y), fabs(cg.z - geo0.z))) + 0.5 * cg.w);
                    }
                }
            }
        }
    }
  double4 geo = geo0;
  if(geo_rw && 2.0 * need > geo.w)
    {
      geo.w = 2.0 * need;
      geo_rw[node] = geo;
    }
#pragma unroll
  for(int g = 0; g < NG; g++)
    {
      double4 o;
      if(m[g] > 0)
        {
          o.x = sx[g] / m[g];
          o.y = sy[g] / m[g];
          o.z = sz[g] / m[g];
        }
      else
        {
          o.x = geo.x;
          o.y = geo.y;
          o.z = geo.z;
        }
      o.w = m[g];
      n_mom[(long long)node * NG + g] = o;
      if(n_npart)
        n_npart[(long long)node * NG + g] = np[g];
    }
  n_flags[node] = (fl & (FLAG_BUCKET | FLAG_PSEUDO | FLAG_PARTIAL)) | (4 * sa.maxsofttype + 32 * sa.diff);
  if(n_count_rw && !(fl & FLAG_BUCKET))
    n_count_rw[node] = cnt_acc;
}

// The same pass with EIGHT lanes per node, one per child slot (tuning "moments_octet"): a node's child indices are one 32-byte
// line, its particle children are consecutive in the sorted array and its child nodes consecutive in the level-contiguous
// numbering, so the eight record loads of an octet are one contiguous stretch and all in flight at once.  MEASURED SLOWER at C4
// (tree build 5.74 against 4.95 ms): eight times the threads, three of four lanes without a child at the deep levels, and 48
// cross-lane moves per node cost more than the coalescing gains -- the one-thread-per-node pass stays the default.
// The eight partial sums are added in a fixed butterfly order ((0+1)+(2+3))+((4+5)+(6+7)): deterministic, the same on every
// task and in both build variants.  Lane 0 of the octet writes.
__device__ __forceinline__ double oct_sum(double v)
{
  v += __shfl_xor(v, 1);
  v += __shfl_xor(v, 2);
  v += __shfl_xor(v, 4);
  return v;
}
template <int NG>
__global__ __launch_bounds__(256) void k_moments8(const double4 *__restrict__ s_pm, const unsigned char *__restrict__ s_type,
                                                   const int *__restrict__ n_first, const int *__restrict__ n_count,
                                                   const int *__restrict__ n_child, const double4 *__restrict__ n_geo,
                                                   double4 *__restrict__ n_mom, int *__restrict__ n_flags, int node0, int nnodes_level,
                                                   WalkParams wp, double4 *__restrict__ geo_rw = nullptr, int *__restrict__ n_npart = nullptr,
                                                   int *__restrict__ n_count_rw = nullptr)
{
  const long long gt = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  const int t = (int)(gt >> 3), q = threadIdx.x & 7;
  const bool live = t < nnodes_level;   // uniform over the octet
  const int node = node0 + (live ? t : 0);
  double m[NG], sx[NG], sy[NG], sz[NG];
  int np[NG];
#pragma unroll
  for(int g = 0; g < NG; g++)
    {
      m[g] = sx[g] = sy[g] = sz[g] = 0;
      np[g] = 0;
    }
  SoftAcc sa;
  sa.maxsofttype = 7;
  sa.diff = 0;
  int cnt_acc = 0;
  const int fl = n_flags[node];
  const double4 geo0 = n_geo[node];
  double need = 0;
  auto add_particle_v = [&](const double4 v, const int ty) {
    if(geo_rw)
      need = fmax(need, fmax(fabs(v.x - geo0.x), fmax(fabs(v.y - geo0.y), fabs(v.z - geo0.z))));
    const int gg = wp.t2g[ty];
#pragma unroll
    for(int g = 0; g < NG; g++)
      if(g == gg)
        {
          np[g]++;
          m[g] += v.w;
          sx[g] += v.w * v.x;
          sy[g] += v.w * v.y;
          sz[g] += v.w * v.z;
        }
    soft_merge(sa, ty, 0, wp.fsoft);
    cnt_acc++;
  };
  if(live)
    {
      if(fl & FLAG_BUCKET)
        {
          const int f = n_first[node], cnt = n_count[node];
          for(int p = f + q; p < f + cnt; p += 8)
            add_particle_v(s_pm[p], s_type[p]);
        }
      else
        {
          const int c = n_child[8 * (long long)node + q];
          if(c <= -2)
            add_particle_v(s_pm[-2 - c], s_type[-2 - c]);
          else if(c >= 0)
            {
#pragma unroll
              for(int g = 0; g < NG; g++)
                {
                  const double4 cm = n_mom[(long long)c * NG + g];
                  if(n_npart)
                    np[g] += n_npart[(long long)c * NG + g];
                  m[g] += cm.w;
                  sx[g] += cm.w * cm.x;
                  sy[g] += cm.w * cm.y;
                  sz[g] += cm.w * cm.z;
                }
              const int cf = n_flags[c];
              soft_merge(sa, (cf >> 2) & 7, (cf >> 5) & 1, wp.fsoft);
              if(n_count_rw)
                cnt_acc += n_count_rw[c];
              if(geo_rw)
                {
                  const double4 cg = n_geo[c];
                  need = fmax(need, fmax(fabs(cg.x - geo0.x), fmax(fabs(cg.y - geo0.y), fabs(cg.z - geo0.z))) + 0.5 * cg.w);
                }
            }
        }
    }
  // combine the eight lanes of the octet
#pragma unroll
  for(int g = 0; g < NG; g++)
    {
      m[g] = oct_sum(m[g]);
      sx[g] = oct_sum(sx[g]);
      sy[g] = oct_sum(sy[g]);
      sz[g] = oct_sum(sz[g]);
      if(n_npart)
        {
          np[g] += __shfl_xor(np[g], 1);
          np[g] += __shfl_xor(np[g], 2);
          np[g] += __shfl_xor(np[g], 4);
        }
    }
#pragma unroll
  for(int off = 1; off < 8; off <<= 1)
    {
      const int ot = __shfl_xor(sa.maxsofttype, off), od = __shfl_xor(sa.diff, off);
      // the lower lane of a pair merges the higher one's partial (fixed order); the higher lane's copy is never used
      if(!(q & off))
        soft_merge(sa, ot, od, wp.fsoft);
      cnt_acc += __shfl_xor(cnt_acc, off);
      if(geo_rw)
        need = fmax(need, __shfl_xor(need, off));
    }
  if(!live || q != 0)
    return;
  double4 geo = geo0;
  if(geo_rw && 2.0 * need > geo.w)
    {
      geo.w = 2.0 * need;
      geo_rw[node] = geo;
    }
#pragma unroll
  for(int g = 0; g < NG; g++)
    {
      double4 o;
      if(m[g] > 0)
        {
          o.x = sx[g] / m[g];
          o.y = sy[g] / m[g];
          o.z = sz[g] / m[g];
        }
      else
        {
          o.x = geo.x;
          o.y = geo.y;
          o.z = geo.z;
        }
      o.w = m[g];
      n_mom[(long long)node * NG + g] = o;
      if(n_npart)
        n_npart[(long long)node * NG + g] = np[g];
    }
  n_flags[node] = (fl & (FLAG_BUCKET | FLAG_PSEUDO | FLAG_PARTIAL)) | (4 * sa.maxsofttype + 32 * sa.diff);
  if(n_count_rw && !(fl & FLAG_BUCKET))
    n_count_rw[node] = cnt_acc;
}

// Global top of a multi-task tree: monopoles and softening flags of the tree nodes that are top-tree nodes, from the all-reduced
// sums (force_treeupdate_pseudos, forcetree.c:851-947, adds the remote top-leaf moments up the ancestor chain; here every top
// node takes the sum over ALL tasks directly).  Top LEAVES: only the pseudo nodes (the leaves present here keep the bottom-up
// moments of their own subtree).
template <int NG>
__global__ void k_top_moments(const int *__restrict__ n_top, const int *__restrict__ t_child, const double *__restrict__ gsum,
                              const double4 *__restrict__ n_geo, double4 *__restrict__ n_mom, int *__restrict__ n_flags, int node0,
                              int nnodes_level, WalkParams wp)
{
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  if(t >= nnodes_level)
    return;
  const int node = node0 + t, fl = n_flags[node], tn = n_top[node];
  if(tn < 0 || (t_child[tn] < 0 && !(fl & FLAG_PSEUDO)))
    return;
  const double *s = gsum + (size_t)tn * TOP_CW(NG);
  const double4 geo = n_geo[node];
#pragma unroll
  for(int g = 0; g < NG; g++)
    {
      double4 o;
      const double m = s[7 + 4 * g];
      if(m > 0)
        {
          o.x = s[7 + 4 * g + 1] / m;
          o.y = s[7 + 4 * g + 2] / m;
          o.z = s[7 + 4 * g + 3] / m;
        }
      else
        {
          o.x = geo.x;
          o.y = geo.y;
          o.z = geo.z;
        }
      o.w = m;
      n_mom[(long long)node * NG + g] = o;
    }
  SoftAcc sa;
  sa.maxsofttype = 7;
  sa.diff = 0;
  for(int ty = 0; ty < NGRAVS_NTYPES; ty++)
    if(s[1 + ty] > 0)
      soft_merge(sa, ty, 0, wp.fsoft);
  n_flags[node] = (fl & (FLAG_BUCKET | FLAG_PSEUDO | FLAG_PARTIAL)) | (4 * sa.maxsofttype + 32 * sa.diff);
}

static int tree_top_moments(ngravs_ctx *c)
{
  const TopTree &t = c->top;
  WalkParams wp;
  make_walk_params(c, &wp);
  for(int l = (t.h.depth < c->nlevels - 1 ? t.h.depth : c->nlevels - 1); l >= 0; l--)
    {
      const long long l0 = c->level_start[l], lc = c->level_start[l + 1] - l0;
      if(lc <= 0)
        continue;
      const unsigned nb = (unsigned)((lc + 127) / 128);
      switch(c->cfg.n_gravs)
        {
        case 1:
          hipLaunchKernelGGL(k_top_moments<1>, dim3(nb), dim3(128), 0, c->stream, c->n_top.p, t.child.p, t.gsum.p, c->n_geo.p, c->n_mom.p,
                             c->n_flags.p, (int)l0, (int)lc, wp);
          break;
        case 2:
          hipLaunchKernelGGL(k_top_moments<2>, dim3(nb), dim3(128), 0, c->stream, c->n_top.p, t.child.p, t.gsum.p, c->n_geo.p, c->n_mom.p,
                             c->n_flags.p, (int)l0, (int)lc, wp);
          break;
        default:
          hipLaunchKernelGGL(k_top_moments<3>, dim3(nb), dim3(128), 0, c->stream, c->n_top.p, t.child.p, t.gsum.p, c->n_geo.p, c->n_mom.p,
                             c->n_flags.p, (int)l0, (int)lc, wp);
          break;
        }
    }
  HIP_TRY(c, hipGetLastError());
  return NGRAVS_OK;
}

// ---- kept decomposition (ngravs_host_kept_step): the top of a refit tree from the data of all tasks ------------------------------
// force_update_node_len_toptree / force_update_pseudoparticles (forcetree.c:753, 1096-1122): a top leaf whose particles live on
// another task has no particles here to grow its cell from; its owner's refit knows the side the single-task refit would give it.
// Tree nodes that are top LEAVES owned by this task: their grown side -> kept_sums[leaf * stride + stride - 1]
__global__ void k_top_leaf_len(const int *__restrict__ n_top, const int *__restrict__ t_child, const int *__restrict__ t_leaf,
                               const int *__restrict__ leaf_owner, int me, const double4 *__restrict__ n_geo, int node0, int nnodes_level,
                               double *__restrict__ kept_sums, int stride)
{
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  if(t >= nnodes_level)
    return;
  const int node = node0 + t, tn = n_top[node];
  if(tn < 0 || t_child[tn] >= 0)
    return;
  const int leaf = t_leaf[tn];
  if(leaf >= 0 && leaf_owner[leaf] == me)
    kept_sums[(size_t)leaf * stride + stride - 1] = n_geo[node].w;
}

int tree_top_leaf_len(ngravs_ctx *c, double *dev_kept_sums, int stride)
{
  const TopTree &t = c->top;
  if(!c->have_tree || !t.on)
    return NGRAVS_ERR_STATE;
  if(c->n == 0)   // (a task without particles and without imported copies: no tree nodes)
    return NGRAVS_OK;
  for(int l = (t.h.depth < c->nlevels - 1 ? t.h.depth : c->nlevels - 1); l >= 0; l--)
    {
      const long long l0 = c->level_start[l], lc = c->level_start[l + 1] - l0;
      if(lc <= 0)
        continue;
      hipLaunchKernelGGL(k_top_leaf_len, dim3((unsigned)((lc + 127) / 128)), dim3(128), 0, c->stream, c->n_top.p, t.child.p, t.leaf.p,
                         t.leaf_owner.p, t.kept_rank, c->n_geo.p, (int)l0, (int)lc, dev_kept_sums, stride);
    }
  HIP_TRY(c, hipGetLastError());
  return NGRAVS_OK;
}

// sides of the top nodes of a refit tree, bottom-up: an absent leaf (pseudo node) takes the side all tasks agreed on, a split top
// node grows to enclose its (grown) children by the rule of k_moments' refit
__global__ void k_top_len(const int *__restrict__ n_top, const int *__restrict__ t_child, const int *__restrict__ t_leaf,
                          const double *__restrict__ leaf_len, const int *__restrict__ n_child, const int *__restrict__ n_flags,
                          double4 *__restrict__ n_geo, int node0, int nnodes_level)
{
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  if(t >= nnodes_level)
    return;
  const int node = node0 + t, tn = n_top[node];
  if(tn < 0)
    return;
  double4 geo = n_geo[node];
  if(t_child[tn] < 0)
    {
      if(n_flags[node] & FLAG_PSEUDO)
        {
          const double want = leaf_len[t_leaf[tn]];
          if(want > geo.w)
            {
              geo.w = want;
              n_geo[node] = geo;
            }
        }
      return;
    }
  double need = 0;
  for(int k = 0; k < 8; k++)
    {
      const int ch = n_child[8 * (long long)node + k];
      if(ch >= 0)
        {
          const double4 cg = n_geo[ch];
          need = fmax(need, fmax(fabs(cg.x - geo.x), fmax(fabs(cg.y - geo.y), fabs(cg.z - geo.z))) + 0.5 * cg.w);
        }
    }
  if(2.0 * need > geo.w)
    {
      geo.w = 2.0 * need;
      n_geo[node] = geo;
    }
}

int tree_top_refit(ngravs_ctx *c)
{
  const TopTree &t = c->top;
  if(!c->have_tree || !t.on)
    return NGRAVS_ERR_STATE;
  if(c->n == 0)   // (a task without particles and without imported copies: no tree nodes)
    return NGRAVS_OK;
  for(int l = (t.h.depth < c->nlevels - 1 ? t.h.depth : c->nlevels - 1); l >= 0; l--)
    {
      const long long l0 = c->level_start[l], lc = c->level_start[l + 1] - l0;
      if(lc <= 0)
        continue;
      hipLaunchKernelGGL(k_top_len, dim3((unsigned)((lc + 127) / 128)), dim3(128), 0, c->stream, c->n_top.p, t.child.p, t.leaf.p, t.leaf_len.p,
                         c->n_child.p, c->n_flags.p, c->n_geo.p, (int)l0, (int)lc);
    }
  HIP_TRY(c, hipGetLastError());
  return tree_top_moments(c);
}

// multipole moments, softening flags (and, for a refit, grown cell sides) of all nodes, bottom-up, one launch per level
int tree_moments(ngravs_ctx *c, bool refit, bool counts)
{
  const int ng = c->cfg.n_gravs;
  const int bs = 128;
  double4 *grw = refit ? c->n_geo.p : nullptr;
  int *npp = nullptr;
  if(cfg_has_bam(c->cfg))
    {
      if(c->n_npart.ensure((size_t)c->max_nodes * ng))
        return NGRAVS_ERR_NOMEM;
      npp = c->n_npart.p;
    }
  WalkParams wp;
  make_walk_params(c, &wp);
  int *cntp = counts ? c->n_count.p : nullptr;
  for(int l = c->nlevels - 1; l >= 0; l--)
    {
      long long l0 = c->level_start[l], lc = c->level_start[l + 1] - l0;
      if(lc <= 0)
        continue;
      unsigned nb = (unsigned)((lc + bs - 1) / bs);
      if(c->tune.moments_octet)
        {
          const unsigned nb8 = (unsigned)((8 * lc + 255) / 256);
          switch(ng)
            {
            case 1:
              hipLaunchKernelGGL(k_moments8<1>, dim3(nb8), dim3(256), 0, c->stream, c->s_pm.p, c->s_type.p, c->n_first.p, c->n_count.p,
                                 c->n_child.p, c->n_geo.p, c->n_mom.p, c->n_flags.p, (int)l0, (int)lc, wp, grw, npp, cntp);
              break;
            case 2:
              hipLaunchKernelGGL(k_moments8<2>, dim3(nb8), dim3(256), 0, c->stream, c->s_pm.p, c->s_type.p, c->n_first.p, c->n_count.p,
                                 c->n_child.p, c->n_geo.p, c->n_mom.p, c->n_flags.p, (int)l0, (int)lc, wp, grw, npp, cntp);
              break;
            default:
              hipLaunchKernelGGL(k_moments8<3>, dim3(nb8), dim3(256), 0, c->stream, c->s_pm.p, c->s_type.p, c->n_first.p, c->n_count.p,
                                 c->n_child.p, c->n_geo.p, c->n_mom.p, c->n_flags.p, (int)l0, (int)lc, wp, grw, npp, cntp);
              break;
            }
          continue;
        }
      switch(ng)
        {
        case 1:
          hipLaunchKernelGGL(k_moments<1>, dim3(nb), dim3(bs), 0, c->stream, c->s_pm.p, c->s_type.p, c->n_first.p,
                             c->n_count.p, c->n_child.p, c->n_geo.p, c->n_mom.p, c->n_flags.p, (int)l0, (int)lc, wp, grw, npp, cntp);
          break;
        case 2:
          hipLaunchKernelGGL(k_moments<2>, dim3(nb), dim3(bs), 0, c->stream, c->s_pm.p, c->s_type.p, c->n_first.p,
                             c->n_count.p, c->n_child.p, c->n_geo.p, c->n_mom.p, c->n_flags.p, (int)l0, (int)lc, wp, grw, npp, cntp);
          break;
        default:
          hipLaunchKernelGGL(k_moments<3>, dim3(nb), dim3(bs), 0, c->stream, c->s_pm.p, c->s_type.p, c->n_first.p,
                             c->n_count.p, c->n_child.p, c->n_geo.p, c->n_mom.p, c->n_flags.p, (int)l0, (int)lc, wp, grw, npp, cntp);
          break;
        }
    }
  HIP_TRY(c, hipGetLastError());
  if(c->top.on)
    return tree_top_moments(c);
  return NGRAVS_OK;
}

int tree_build(ngravs_ctx *c)
{
  const long long n = c->n;
  double taf = c->cfg.tree_alloc_factor > 0 ? c->cfg.tree_alloc_factor : 0.8;
  long long maxn = (long long)(taf * (double)n) + 1024;
  const TopTree &top = c->top;
  if(top.on)
    maxn += top.h.nnode;   // the global top: every top node may be a tree node
  c->max_nodes = maxn;
  const int ng = c->cfg.n_gravs;
  if(c->n_first.ensure(maxn) || c->n_count.ensure(maxn) || c->n_child.ensure(8 * maxn) || c->n_flags.ensure(maxn) ||
     c->n_geo.ensure(maxn) || c->n_mom.ensure(maxn * ng) || c->n_nchild.ensure(maxn) || c->scan_out.ensure(maxn) ||
     c->d_counters.ensure(16) || (c->top.on && c->n_top.ensure(maxn)))
    return NGRAVS_ERR_NOMEM;
  // root = the domain cube (forcetree.c:103-110)
  int h_first = 0, h_count = (int)n, h_flags = 0;
  double4 h_geo;
  h_geo.x = c->dom[3];
  h_geo.y = c->dom[4];
  h_geo.z = c->dom[5];
  h_geo.w = c->dom[6];
  HIP_TRY(c, hipMemcpyAsync(c->n_first.p, &h_first, sizeof(int), hipMemcpyHostToDevice, c->stream));
  HIP_TRY(c, hipMemcpyAsync(c->n_count.p, &h_count, sizeof(int), hipMemcpyHostToDevice, c->stream));
  HIP_TRY(c, hipMemcpyAsync(c->n_flags.p, &h_flags, sizeof(int), hipMemcpyHostToDevice, c->stream));
  HIP_TRY(c, hipMemcpyAsync(c->n_geo.p, &h_geo, sizeof(double4), hipMemcpyHostToDevice, c->stream));
  if(top.on)
    {
      unsigned char part0 = 0;
      HIP_TRY(c, hipMemsetAsync(c->n_top.p, 0, sizeof(int), c->stream));   // the root is top node 0
      HIP_TRY(c, hipMemcpyAsync(&part0, top.info.p, 1, hipMemcpyDeviceToHost, c->stream));
      HIP_TRY(c, hipStreamSynchronize(c->stream));
      h_flags = (part0 & 8) ? FLAG_PARTIAL : 0;
      HIP_TRY(c, hipMemcpyAsync(c->n_flags.p, &h_flags, sizeof(int), hipMemcpyHostToDevice, c->stream));
    }
  double fac21 = c->dom[7] * (double)(1 << (TREE_BITS - NGRAVS_BITS_PER_DIMENSION));
  // level table on the device: lv[2 l], lv[2 l + 1] = first node, node count of level l; lv[2 (MAX_LEVELS + 2)] = out-of-nodes flag
  const int LVN = 2 * (MAX_LEVELS + 2) + 2;
  if(c->d_levels.ensure(LVN) || c->scan_out.ensure((size_t)(maxn / TB + 2)))
    return NGRAVS_ERR_NOMEM;
  int h_lv[2 * (MAX_LEVELS + 2) + 2];
  memset(h_lv, 0, sizeof(h_lv));
  h_lv[1] = 1;   // the root
  const bool onepass = !top.on && n >= 2 && !c->tune.tree_levelwise;
  if(onepass)
    {
      // the whole topology from one pass over the sorted keys (see k_tb_nodes)
      const int nblk = (int)((n + (long long)TBN * TBCH - 1) / ((long long)TBN * TBCH));
      const int nblk_pad = (nblk + 3) & ~3;   // rows of the count table are scanned as int4
      if(c->tb_count.ensure((size_t)(TREE_BITS + 1) * nblk_pad + 64))
        return NGRAVS_ERR_NOMEM;
      int *lvcnt = c->tb_count.p + (size_t)(TREE_BITS + 1) * nblk_pad;
      if(nblk_pad != nblk)
        HIP_TRY(c, hipMemsetAsync(c->tb_count.p, 0, sizeof(int) * (size_t)(TREE_BITS + 1) * nblk_pad, c->stream));
      hipLaunchKernelGGL(k_tb_count, dim3(nblk), dim3(TBN), 0, c->stream, c->s_key.p, n, c->tb_count.p, nblk_pad);
      hipLaunchKernelGGL(k_tb_scan, dim3(TREE_BITS + 1), dim3(1024), 0, c->stream, c->tb_count.p, nblk_pad, lvcnt);
      hipLaunchKernelGGL(k_tb_levels, dim3(1), dim3(64), 0, c->stream, lvcnt, c->d_levels.p, (int)(maxn > 2147483647ll ? 2147483647ll : maxn));
      {
        long long guess = c->nnodes > 0 ? c->nnodes + c->nnodes / 8 : n / 2;
        unsigned nbf = (unsigned)((2 * guess + 255) / 256);
        nbf = nbf < 1 ? 1 : (nbf > 262144u ? 262144u : nbf);
        hipLaunchKernelGGL(k_tb_fill, dim3(nbf), dim3(256), 0, c->stream, reinterpret_cast<int4 *>(c->n_child.p), c->d_levels.p);
      }
      hipLaunchKernelGGL(k_tb_nodes, dim3(nblk), dim3(TBN), 0, c->stream, c->s_key.p, c->s_pm.p, n, c->tb_count.p, nblk_pad, c->d_levels.p,
                         c->n_first.p, c->n_count.p, c->n_child.p, c->n_geo.p, c->n_flags.p, h_geo, c->dom[0], c->dom[1], c->dom[2], fac21);
    }
  else
    HIP_TRY(c, hipMemcpyAsync(c->d_levels.p, h_lv, sizeof(int) * LVN, hipMemcpyHostToDevice, c->stream));
  // No host round trip between levels: every level's launches are issued blindly, with grids sized from the level populations of
  // the previous build (any grid is correct: the kernels loop over virtual blocks); levels beyond the deepest find count 0.
  for(int level = 0; level < TREE_BITS && !onepass; level++)
    {
      long long guess = c->level_hint[level] > 0 ? c->level_hint[level] + c->level_hint[level] / 8 : ((level < 8) ? (1ll << (3 * level)) : (n + 1) / 2);
      if(guess > (n + 1) / 2 + 1 && level > 0 && !top.on)
        guess = (n + 1) / 2 + 1;
      unsigned nb = (unsigned)((guess + TB - 1) / TB);
      nb = nb < 1 ? 1 : (nb > 262144u ? 262144u : nb);
      const bool in_top = top.on && level < top.h.depth;   // nodes of this level may be split nodes of the global top tree
      hipLaunchKernelGGL(k_split, dim3(nb), dim3(TB), 0, c->stream, c->s_key.p, c->n_first.p, c->n_count.p, c->d_levels.p, level, c->n_child.p,
                         c->n_nchild.p, c->scan_out.p, in_top ? c->n_top.p : (const int *)nullptr, top.child.p, top.gcnt.p);
      hipLaunchKernelGGL(k_scan_blocks, dim3(1), dim3(1024), 0, c->stream, c->scan_out.p, c->d_levels.p, level, (int)maxn);
      hipLaunchKernelGGL(k_link, dim3(nb), dim3(TB), 0, c->stream, c->s_pm.p, c->n_first.p, c->n_count.p, c->n_child.p, c->n_geo.p,
                         c->n_flags.p, c->n_nchild.p, c->scan_out.p, c->d_levels.p, level, c->dom[0], c->dom[1], c->dom[2], fac21,
                         in_top ? c->n_top.p : (int *)nullptr, top.child.p, top.info.p);
    }
  HIP_TRY(c, hipGetLastError());
  HIP_TRY(c, hipMemcpyAsync(h_lv, c->d_levels.p, sizeof(int) * LVN, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));   // the only one of the build: the host needs the level extents from here on
  if(h_lv[2 * (MAX_LEVELS + 2)])
    {
      ngravs_report(c, 1, "maximum number of tree-nodes reached (increase tree_alloc_factor)");   // endrun(1), forcetree.c:247
      return NGRAVS_ERR_TREE;
    }
  int nl = 0;
  for(int l = 0; l <= TREE_BITS; l++)
    {
      c->level_start[l] = h_lv[2 * l];
      c->level_hint[l] = h_lv[2 * l + 1];
      if(h_lv[2 * l + 1] > 0)
        nl = l + 1;
    }
  c->nlevels = nl;
  c->nnodes = (long long)h_lv[2 * (nl - 1)] + h_lv[2 * (nl - 1) + 1];
  for(int l = nl; l <= MAX_LEVELS + 1; l++)
    c->level_start[l] = c->nnodes;
  int rcm = tree_moments(c, false, onepass);
  if(rcm)
    return rcm;
  HIP_TRY(c, hipGetLastError());
  c->stats.n_nodes = c->nnodes;
  return NGRAVS_OK;
}
