// kernels_walk.hip -- the short-range / full tree walk, OldAcc + G post-processing, direct sum.
//
// Replaces (reference): force_treeevaluate (forcetree.c:1244-1610), force_treeevaluate_shortrange
// (forcetree.c:1623-2052), the per-particle loop + post-processing of gravity_tree
// (gravtree.c:112-149, 318-341) and force_treeevaluate_direct (forcetree.c:3428-3548).
//
// Two walk kernels over the same level-ordered tree, one wave64 per 64 Peano-contiguous targets:
//
//  k_walk_strict : the reference semantics per target.  The wave walks the tree depth-first in
//      lock-step (control flow is wave-uniform: node indices live in SGPRs, node records are
//      scalar loads); every lane takes its OWN prune / open / accept decision exactly as the
//      reference does, and a lane that has used or pruned a node sleeps until the walk has left
//      that node's subtree (`lane_skip`: a stack depth).  Interaction set, interaction count and
//      summation order per target equal the reference's (children are visited in Peano instead
//      of Morton order, which only reorders the fp64 sum).
//
//  k_walk_group2 : the MI355X production walk, as a traversal kernel (MODE 1), an evaluation kernel (MODE 2) or both in
//      one (MODE 0).  Traversal: the 64 lanes work cooperatively -- a LIFO of pending nodes is popped 64 at a time, each
//      lane tests ONE node against the group's bounding box with the conservative form of every reference test (a node
//      is used only if every target would use it, dropped only if every target would drop it); accepted monopoles and
//      particle leaves are recorded as item indices per source species (one packed wave prefix sum per round).
//      Evaluation: the items are visited in a golden-ratio stride order, culled against the box, compacted into an LDS
//      pool; every lane builds its own 64-bit hit mask (packed-fp32 reach pre-test) and walks its own bits through the
//      force law.  The TreePM short-range tables (the distinct ones of the symmetric wiring) are staged in LDS once per
//      persistent evaluation workgroup; workgroups pull groups from per-XCD atomic counters (XCD-aware Peano segments,
//      stealing when exhausted).  S lanes can share one target (sparse active sets, leftover groups).
//
// Force laws: the reference calls AccelFxns[tg][sg] through a pointer; here the wired table is
// lowered to coefficients  a(r) = m [ cN/r^2 + cY exp(-r ym)(ym/r + 1/r^2) ]  (none, newtonian,
// neg_newtonian, yukawa, coloyuk) and kernels are compiled per N_GRAVS and per "has Yukawa".
#include "engine.hpp"
#include "walk_device.hpp"
#include "eval_asm.inc"   // ER_DIRECT_ASM: the tree-only force loop (the other blocks belong to kernels_eval.hip)
#include <hipcub/hipcub.hpp>
#include <type_traits>

// ---------------------------------------------------------------------------------------------
// force laws, reference formulation (strict walk, direct sum)
// ---------------------------------------------------------------------------------------------
// the BAM family (ngravs.c:495-668): laws of the TARGET mass and of the particle number N behind the source
__device__ __forceinline__ double bam_eta(int law, double target, double m, double N, double eps)
{
  if(law == NGRAVS_LAW_BAMBAM)
    return 4.0 * M_PI * eps / (target + m / N);                        // ngravs.c:506, :542
  if(law == NGRAVS_LAW_SOURCEBAM)
    return 4.0 * M_PI * eps * N / m;                                   // ngravs.c:569, :597
  return 4.0 * M_PI * eps / target;                                    // ngravs.c:624, :654
}
__device__ __forceinline__ double bam_accel(int law, double target, double m, double r, double N, double eps)
{
  double eta = bam_eta(law, target, m, N, eps), rho = 2 * target * m / M_PI;
  double reta = r * eta, reta2 = reta * reta, eta3 = eta * eta * eta;
  if(reta < 0.1)
    return rho * eta3 * (2.0 * r / 3.0 - 4.0 * reta2 * r / 5.0 + 6.0 * reta2 * reta2 * r / 7.0);
  return rho * eta3 * (atan(reta) / (reta2 * eta) - 1.0 / (reta * eta * (1 + reta2)));
}
__device__ __forceinline__ double bam_spline(int law, double target, double m, double r, double N, double eps)
{
  double eta = bam_eta(law, target, m, N, eps), rho = 2 * target * m / M_PI;
  double reta = r * eta, reta2 = reta * reta, eta3 = eta * eta * eta;
  if(reta < 0.1)
    return rho * eta3 * (2.0 / 3.0 - 4.0 * reta2 / 5.0 + 6.0 * reta2 * reta2 / 7.0);
  return rho * eta3 * (atan(reta) / (reta2 * reta) - 1.0 / (reta2 * (1 + reta2)));
}

__device__ __forceinline__ double law_accel_ref(int law, double m, double r2, double r, double ym, double target = 1.0, double N = 1.0,
                                                double eps = 0.0)
{
  switch(law)
    {
    case NGRAVS_LAW_BAMBAM:
    case NGRAVS_LAW_SOURCEBAM:
    case NGRAVS_LAW_TARGETBAM:
      return bam_accel(law, target, m, r, N, eps);
    case NGRAVS_LAW_NEWTON:
      return m / r2;                                                   // ngravs.c:351
    case NGRAVS_LAW_NEG_NEWTON:
      return -m / r2;                                                  // ngravs.c:357
    case NGRAVS_LAW_YUKAWA:
      return m * exp(-r * ym) * (ym / r + 1.0 / r2);                   // ngravs.c:856-861
    case NGRAVS_LAW_COLOYUK:
      return m * exp(-r * ym) * (ym / r + 1.0 / r2) + m / r2;          // ngravs.c:826
    default:
      return 0.0;
    }
}
__device__ __forceinline__ double law_spline_ref(int id, double m, double h, double r, double target = 1.0, double N = 1.0, double eps = 0.0)
{
  if(id == NGRAVS_SPLINE_NONE)
    return 0.0;
  if(id >= NGRAVS_SPLINE_BAMBAM)
    return bam_spline(id == NGRAVS_SPLINE_BAMBAM ? NGRAVS_LAW_BAMBAM : (id == NGRAVS_SPLINE_SOURCEBAM ? NGRAVS_LAW_SOURCEBAM : NGRAVS_LAW_TARGETBAM),
                      target, m, r, N, eps);
  double h_inv = 1 / h, v;                                             // ngravs.c:420-434, literal constants
  r *= h_inv;
  if(r < 0.5)
    v = m * h_inv * h_inv * h_inv * (10.666666666667 + r * r * (32.0 * r - 38.4));
  else
    v = m * h_inv * h_inv * h_inv *
        (21.333333333333 - 48.0 * r + 38.4 * r * r - 10.666666666667 * r * r * r - 0.066666666667 / (r * r * r));
  return id == NGRAVS_SPLINE_NEG_PLUMMER ? -v : v;
}

struct LawIds
{
  int accel[NG_MAX][NG_MAX], spline[NG_MAX][NG_MAX];
};

// =============================================================================================
//  periodic tree-only path: Ewald / lattice-sum correction (reference lattice_init forcetree.c:3611-3793,
//  ewald_force ngravs.c:1170-1232, yukawa_lattice_force ngravs.c:1019-1090, lattice_corr forcetree.c:3803-3885)
// =============================================================================================
#define LAT_EN 64
#define LAT_E1 (LAT_EN + 1)
#define LAT_SZ ((size_t)3 * LAT_E1 * LAT_E1 * LAT_E1)

#pragma clang fp contract(off)
// one thread per table point of one law; out[3][E1^3], already divided by BoxSize^2
__global__ void k_lattice_table(int law, double ymass, double L2, double *__restrict__ out)
{
  int n = blockIdx.x * blockDim.x + threadIdx.x;
  if(n >= LAT_E1 * LAT_E1 * LAT_E1)
    return;
  const int i = n / (LAT_E1 * LAT_E1), j = (n / LAT_E1) % LAT_E1, k = n % LAT_E1;
  const double x[3] = {0.5 * ((double)i) / LAT_EN, 0.5 * ((double)j) / LAT_EN, 0.5 * ((double)k) / LAT_EN};
  double f[3] = {0, 0, 0};
  if(n != 0 && (law == NGRAVS_LAW_NEWTON || law == NGRAVS_LAW_NEG_NEWTON || law == NGRAVS_LAW_COLOYUK))
    {
      const double alpha = 2.0, sgn = law == NGRAVS_LAW_NEG_NEWTON ? -1.0 : 1.0;
      double g[3] = {0, 0, 0};
      const double r2 = x[0] * x[0] + x[1] * x[1] + x[2] * x[2];
      for(int c = 0; c < 3; c++)
        g[c] += x[c] / (r2 * sqrt(r2));
      for(int n0 = -4; n0 <= 4; n0++)
        for(int n1 = -4; n1 <= 4; n1++)
          for(int n2 = -4; n2 <= 4; n2++)
            {
              const double dx[3] = {x[0] - n0, x[1] - n1, x[2] - n2};
              const double r = sqrt(dx[0] * dx[0] + dx[1] * dx[1] + dx[2] * dx[2]);
              const double val = erfc(alpha * r) + 2 * alpha * r / sqrt(M_PI) * exp(-alpha * alpha * r * r);
              for(int c = 0; c < 3; c++)
                g[c] -= dx[c] / (r * r * r) * val;
            }
      for(int h0 = -4; h0 <= 4; h0++)
        for(int h1 = -4; h1 <= 4; h1++)
          for(int h2_ = -4; h2_ <= 4; h2_++)
            {
              const int h[3] = {h0, h1, h2_};
              const int h2 = h0 * h0 + h1 * h1 + h2_ * h2_;
              if(h2 > 0)
                {
                  const double hdotx = x[0] * h0 + x[1] * h1 + x[2] * h2_;
                  const double val = 2.0 / ((double)h2) * exp(-M_PI * M_PI * h2 / (alpha * alpha)) * sin(2 * M_PI * hdotx);
                  for(int c = 0; c < 3; c++)
                    g[c] -= h[c] * val;
                }
            }
      for(int c = 0; c < 3; c++)
        f[c] += sgn * g[c];
    }
  if(n != 0 && (law == NGRAVS_LAW_YUKAWA || law == NGRAVS_LAW_COLOYUK))
    {
      const double alpha = 5.64;
      double ym = ymass, g[3];
      const double r2 = x[0] * x[0] + x[1] * x[1] + x[2] * x[2];
      double r = sqrt(r2);
      for(int c = 0; c < 3; c++)
        g[c] = exp(-r * ym) * (ym + 1.0 / r) * x[c] / r2;
      for(int n0 = -5; n0 <= 5; n0++)
        for(int n1 = -5; n1 <= 5; n1++)
          for(int n2 = -5; n2 <= 5; n2++)
            {
              const double dx[3] = {x[0] - n0, x[1] - n1, x[2] - n2};
              r = sqrt(dx[0] * dx[0] + dx[1] * dx[1] + dx[2] * dx[2]);
              const double ep = exp(ym * r) * erfc(alpha * r + ym / (2 * alpha)), em = exp(-ym * r) * erfc(alpha * r - ym / (2 * alpha));
              double val = 0.5 * (ep + em);
              for(int c = 0; c < 3; c++)
                g[c] -= dx[c] / (r * r * r) * val;
              val = 0.5 * ym * (-ep + em) + 2 * alpha * exp(-alpha * alpha * r * r - ym * ym / (4 * alpha * alpha)) / sqrt(M_PI);
              for(int c = 0; c < 3; c++)
                g[c] -= dx[c] / (r * r) * val;
            }
      ym /= 2 * M_PI;
      for(int h0 = -5; h0 <= 5; h0++)
        for(int h1 = -5; h1 <= 5; h1++)
          for(int h2_ = -5; h2_ <= 5; h2_++)
            {
              const int h[3] = {h0, h1, h2_};
              const int h2 = h0 * h0 + h1 * h1 + h2_ * h2_;
              if(h2 > 0)
                {
                  const double hdotx = x[0] * h0 + x[1] * h1 + x[2] * h2_;
                  const double val = 2 * exp(-M_PI * M_PI * (h2 + ym * ym) / (alpha * alpha)) * sin(2 * M_PI * hdotx) / (h2 + ym * ym);
                  for(int c = 0; c < 3; c++)
                    g[c] -= h[c] * val;
                }
            }
      for(int c = 0; c < 3; c++)
        f[c] += g[c];
    }
  for(int c = 0; c < 3; c++)
    out[(size_t)c * LAT_E1 * LAT_E1 * LAT_E1 + n] = f[c] / L2;
}

// lattice_corr: trilinear lookup in the octant table of one species pair; result to be multiplied by the source mass
__device__ __forceinline__ void lat_lookup(const double *__restrict__ t3, double fac_intp, double dx, double dy, double dz,
                                           double &fx, double &fy, double &fz)
{
  const double sx = dx < 0 ? 1.0 : -1.0, sy = dy < 0 ? 1.0 : -1.0, sz = dz < 0 ? 1.0 : -1.0;
  double u = fabs(dx) * fac_intp, v = fabs(dy) * fac_intp, w = fabs(dz) * fac_intp;
  int i = (int)u, j = (int)v, k = (int)w;
  i = i >= LAT_EN ? LAT_EN - 1 : i;
  j = j >= LAT_EN ? LAT_EN - 1 : j;
  k = k >= LAT_EN ? LAT_EN - 1 : k;
  u -= i;
  v -= j;
  w -= k;
  const double f1 = (1 - u) * (1 - v) * (1 - w), f2 = (1 - u) * (1 - v) * (w), f3 = (1 - u) * (v) * (1 - w), f4 = (1 - u) * (v) * (w),
               f5 = (u) * (1 - v) * (1 - w), f6 = (u) * (1 - v) * (w), f7 = (u) * (v) * (1 - w), f8 = (u) * (v) * (w);
  const size_t o = ((size_t)i * LAT_E1 + j) * LAT_E1 + k, sj = LAT_E1, si = (size_t)LAT_E1 * LAT_E1, sc = si * LAT_E1;
  const double *a = t3;
  fx = sx * (a[o] * f1 + a[o + 1] * f2 + a[o + sj] * f3 + a[o + sj + 1] * f4 + a[o + si] * f5 + a[o + si + 1] * f6 + a[o + si + sj] * f7 + a[o + si + sj + 1] * f8);
  a = t3 + sc;
  fy = sy * (a[o] * f1 + a[o + 1] * f2 + a[o + sj] * f3 + a[o + sj + 1] * f4 + a[o + si] * f5 + a[o + si + 1] * f6 + a[o + si + sj] * f7 + a[o + si + sj + 1] * f8);
  a = t3 + 2 * sc;
  fz = sz * (a[o] * f1 + a[o + 1] * f2 + a[o + sj] * f3 + a[o + sj + 1] * f4 + a[o + si] * f5 + a[o + si + 1] * f6 + a[o + si + sj] * f7 + a[o + si + sj + 1] * f8);
}
#pragma clang fp contract(fast)

// =============================================================================================
//  strict walk
// =============================================================================================
#pragma clang fp contract(off)

template <int NG, bool PM, bool LATT>
__global__ __launch_bounds__(256) void k_walk_strict(TreeView tv, const double4 *__restrict__ s_pm,
                                                     const unsigned char *__restrict__ s_type,
                                                     const double *__restrict__ s_oldacc,
                                                     const unsigned char *__restrict__ s_active,
                                                     const double *__restrict__ table, WalkParams wp, LawIds li,
                                                     long long t_first, long long t_count,
                                                     double *__restrict__ r_acc, int *__restrict__ r_nint, int *__restrict__ err_flag)
{
  __shared__ int st_node[4][MAX_LEVELS + 2], st_slot[4][MAX_LEVELS + 2];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const long long grp = (long long)blockIdx.x * 4 + wave;
  const long long ti = t_first + grp * WAVE + lane;
  const bool valid = (grp * WAVE + lane) < t_count && (s_active[ti < t_first + t_count ? ti : t_first] & 1) != 0;
  if(grp * WAVE >= t_count)
    return;
  double px = 0, py = 0, pz = 0, aold = 0, hT = 0, pmass = 1.0;
  int ptype = 0, tg = 0;
  if(valid)
    {
      double4 p = s_pm[ti];
      px = p.x;
      py = p.y;
      pz = p.z;
      pmass = p.w;
      ptype = s_type[ti];
      tg = wp.t2g[ptype];
      hT = wp.fsoft[ptype];
      aold = wp.errtol_acc * s_oldacc[ti];
    }
  double ax = 0, ay = 0, az = 0;
  int nint = 0;
  // The wave walks the tree in lock-step; a lane that has dealt with a node (used it, or pruned it) while another lane wants it
  // opened sits out until the walk has left that node's subtree: lane_skip = the stack depth of the node, the lane takes part only
  // while sp < lane_skip.  (Depth, not particle ranges: the pseudo nodes of a multi-task tree have EMPTY ranges, which a test on
  // particle indices cannot tell from "behind the subtree".)
  const int NO_SKIP = 0x7fffffff;
  int lane_skip = valid ? NO_SKIP : -1;
  int *sn = st_node[wave], *ss = st_slot[wave];
  int sp = -1;

  // one source (particle or one species of a node): forcetree.c:1534-1583 / :1953-2032
  auto interact = [&](int g, double dx, double dy, double dz, double r2, double m, double h, double N = 1.0) -> bool {
    double r = sqrt(r2), fac;
    if(PM)
      {
        int tab = (int)(wp.asmthfac * r);
        if(tab >= NTAB)
          return false;
        if(r >= h)
          {
            fac = law_accel_ref(li.accel[tg][g], m, r2, r, wp.ym);
            fac -= m * wp.utor2wpi * table[((size_t)tg * NG + g) * NTAB + tab];
            fac /= r;
          }
        else
          fac = law_spline_ref(li.spline[tg][g], m, h, r);
      }
    else
      {
        if(r >= h)
          fac = law_accel_ref(li.accel[tg][g], m, r2, r, wp.ym, pmass, N, wp.bam_eps) / r;
        else
          fac = law_spline_ref(li.spline[tg][g], m, h, r, pmass, N, wp.bam_eps);
      }
    ax += dx * fac;
    ay += dy * fac;
    az += dz * fac;
    return true;
  };

  auto do_particle = [&](int p) {
    double4 q = s_pm[p];          // uniform address -> scalar load
    int qt = s_type[p];
    if(lane_skip > sp)
      {
        int sg = wp.t2g[qt];
        double dx = q.x - px, dy = q.y - py, dz = q.z - pz;
        if(wp.periodic)
          {
            dx = nearest(dx, wp.box, wp.boxhalf);
            dy = nearest(dy, wp.box, wp.boxhalf);
            dz = nearest(dz, wp.box, wp.boxhalf);
          }
        double r2 = dx * dx + dy * dy + dz * dz;
        double h = hT;
        if(h < wp.fsoft[qt])
          h = wp.fsoft[qt];
        bool added = interact(sg, dx, dy, dz, r2, q.w, h);
        if(added || !PM)
          nint++;
      }
  };

  // returns true if some lane wants the node opened
  auto visit = [&](int c) -> bool {
    int fl = tv.flags[c];
    double4 geo = tv.geo[c];
    double4 mom[NG];
#pragma unroll
    for(int g = 0; g < NG; g++)
      mom[g] = tv.mom[(long long)c * NG + g];
    bool open = false;
    const bool takes_part = lane_skip > sp;
    if(takes_part)
      {
        double dx[NG], dy[NG], dz[NG], r2[NG];
        double r2min = INFINITY, r2max = -INFINITY, summass = 0;
#pragma unroll
        for(int g = 0; g < NG; g++)
          {
            summass += mom[g].w;
            dx[g] = mom[g].x - px;
            dy[g] = mom[g].y - py;
            dz[g] = mom[g].z - pz;
            if(wp.periodic)
              {
                dx[g] = nearest(dx[g], wp.box, wp.boxhalf);
                dy[g] = nearest(dy[g], wp.box, wp.boxhalf);
                dz[g] = nearest(dz[g], wp.box, wp.boxhalf);
              }
            r2[g] = dx[g] * dx[g] + dy[g] * dy[g] + dz[g] * dz[g];
            if(r2[g] < r2min)
              r2min = r2[g];
            if(r2[g] > r2max)
              r2max = r2[g];
          }
        const double len = geo.w;
        bool done = false;   // pruned
        if(PM && r2min > wp.rcut2)
          {
            double eff = wp.rcut + 0.5 * len;                           // forcetree.c:1828-1862
            double d0 = geo.x - px, d1 = geo.y - py, d2 = geo.z - pz;
            if(wp.periodic)
              {
                d0 = nearest(d0, wp.box, wp.boxhalf);
                d1 = nearest(d1, wp.box, wp.boxhalf);
                d2 = nearest(d2, wp.box, wp.boxhalf);
              }
            if(d0 < -eff || d0 > eff || d1 < -eff || d1 > eff || d2 < -eff || d2 > eff)
              done = true;
          }
        if(!done)
          {
            if(wp.use_theta)
              {
                if(len * len > r2min * wp.theta2)                      // forcetree.c:1437-1445 (theta2 = ErrTolTheta^2)
                  open = true;
              }
            else
              {
                if(summass * len * len > r2min * r2min * aold)         // forcetree.c:1454
                  open = true;
                else if(fabs(geo.x - px) < 0.60 * len && fabs(geo.y - py) < 0.60 * len &&
                        fabs(geo.z - pz) < 0.60 * len)                 // forcetree.c:1462-1472 (no NEAREST)
                  open = true;
              }
          }
        if(!done && !open)
          {
            double h = hT;
            int mst = (fl >> 2) & 7;
            if(mst == 7)
              open = true;                                             // empty node: nothing below, harmless
            else
              {
                if(h < wp.fsoft[mst])
                  {
                    h = wp.fsoft[mst];
                    if(r2max < h * h && ((fl >> 5) & 1))               // forcetree.c:1488-1499
                      open = true;
                  }
                if(!open)
                  {
                    bool added = false;
#pragma unroll
                    for(int g = 0; g < NG; g++)
                      if(mom[g].w != 0.0)
                        added |= interact(g, dx[g], dy[g], dz[g], r2[g], mom[g].w, h, tv.npart ? (double)tv.npart[(long long)c * NG + g] : 1.0);
                    if(added || !PM)
                      nint++;
                  }
              }
          }
      }
    const bool wave_open = __any(open ? 1 : 0) != 0;
    if(wave_open && takes_part && !open)
      lane_skip = sp + 1;   // the node goes onto the stack at depth sp + 1: this lane is done with everything below it
    // a top leaf whose particles were not imported has no children: its mass would drop out of the force.  The import decision
    // covers what the criterion in force at the decomposition opens; the walk must not run with another one (walk_run reports it)
    if(wave_open && (fl & FLAG_PSEUDO) && threadIdx.x % WAVE == 0)
      atomicOr(err_flag, 4);
    return wave_open;
  };

  // lock-step depth-first driver, shared by the force walk and the lattice-correction walk
  auto dfs = [&](auto &&visit_fn, auto &&particle_fn) {
    sp = -1;
    if(visit_fn(0))
      {
        sp = 0;
        sn[0] = 0;
        ss[0] = 0;
      }
    while(sp >= 0)
      {
        int node = __builtin_amdgcn_readfirstlane(sn[sp]);
        int slot = __builtin_amdgcn_readfirstlane(ss[sp]);
        int fl = tv.flags[node];
        if(fl & FLAG_BUCKET)
          {
            int f = tv.first[node], cnt = tv.count[node];
            if(slot >= cnt)
              {
                sp--;
                if(lane_skip > sp && lane_skip >= 0)
                  lane_skip = NO_SKIP;
                continue;
              }
            ss[sp] = slot + 1;
            particle_fn(f + slot);
            continue;
          }
        if(slot >= 8)
          {
            sp--;
            if(lane_skip > sp && lane_skip >= 0)
              lane_skip = NO_SKIP;   // the walk has left the subtree this lane sat out
            continue;
          }
        ss[sp] = slot + 1;
        int c = __builtin_amdgcn_readfirstlane(tv.child[8 * (long long)node + slot]);
        if(c == -1)
          continue;
        if(c <= -2)
          {
            particle_fn(-2 - c);
            continue;
          }
        if(visit_fn(c))
          {
            sp++;
            sn[sp] = c;
            ss[sp] = 0;
          }
      }
  };
  dfs(visit, do_particle);

  if(LATT)
    {
      // force_treeevaluate_lattice_correction (forcetree.c:2077-2455): its own walk -- a node the opening criterion
      // rejects may still be used if it is small (<= 0.2 box) and does not straddle the half-box seam
      const double *lat = table;   // [tg][sg][3][E1^3]
      lane_skip = valid ? NO_SKIP : -1;
      auto lat_particle = [&](int p) {
        double4 q = s_pm[p];
        int qt = s_type[p];
        if(lane_skip > sp)
          {
            int sg = wp.t2g[qt];
            double dx = nearest(q.x - px, wp.box, wp.boxhalf), dy = nearest(q.y - py, wp.box, wp.boxhalf),
                   dz = nearest(q.z - pz, wp.box, wp.boxhalf);
            double fx, fy, fz;
            lat_lookup(lat + ((size_t)tg * NG + sg) * LAT_SZ, wp.fac_intp, dx, dy, dz, fx, fy, fz);
            ax += q.w * fx;
            ay += q.w * fy;
            az += q.w * fz;
            nint++;
          }
      };
      auto lat_visit = [&](int c) -> bool {
        double4 geo = tv.geo[c];
        double4 mom[NG];
#pragma unroll
        for(int g = 0; g < NG; g++)
          mom[g] = tv.mom[(long long)c * NG + g];
        bool open = false;
        const bool takes_part = lane_skip > sp;
        if(takes_part)
          {
            double dx[NG], dy[NG], dz[NG];
            double r2min = INFINITY, summass = 0;
#pragma unroll
            for(int g = 0; g < NG; g++)
              {
                summass += mom[g].w;
                dx[g] = nearest(mom[g].x - px, wp.box, wp.boxhalf);
                dy[g] = nearest(mom[g].y - py, wp.box, wp.boxhalf);
                dz[g] = nearest(mom[g].z - pz, wp.box, wp.boxhalf);
                double r2 = dx[g] * dx[g] + dy[g] * dy[g] + dz[g] * dz[g];
                if(r2 < r2min)
                  r2min = r2;
              }
            const double len = geo.w;
            bool openflag = false;
            if(wp.use_theta)
              openflag = len * len > r2min * wp.theta2;
            else
              {
                if(summass * len * len > r2min * r2min * aold)
                  openflag = true;
                else if(fabs(geo.x - px) < 0.60 * len && fabs(geo.y - py) < 0.60 * len && fabs(geo.z - pz) < 0.60 * len)
                  openflag = true;
              }
            if(openflag)
              {
                double u0 = nearest(geo.x - px, wp.box, wp.boxhalf), u1 = nearest(geo.y - py, wp.box, wp.boxhalf),
                       u2 = nearest(geo.z - pz, wp.box, wp.boxhalf);
                const double lim = 0.5 * (wp.box - len);
                if(fabs(u0) > lim || fabs(u1) > lim || fabs(u2) > lim || len > 0.20 * wp.box)   // forcetree.c:2203-2243
                  open = true;
              }
            if(!open)
              {
#pragma unroll
                for(int g = 0; g < NG; g++)
                  if(mom[g].w != 0.0)
                    {
                      double fx, fy, fz;
                      lat_lookup(lat + ((size_t)tg * NG + g) * LAT_SZ, wp.fac_intp, dx[g], dy[g], dz[g], fx, fy, fz);
                      ax += mom[g].w * fx;
                      ay += mom[g].w * fy;
                      az += mom[g].w * fz;
                    }
                nint++;
              }
          }
        const bool wave_open = __any(open ? 1 : 0) != 0;
        if(wave_open && takes_part && !open)
          lane_skip = sp + 1;
        return wave_open;
      };
      dfs(lat_visit, lat_particle);
    }
  if(valid)
    {
      r_acc[3 * ti + 0] = ax;
      r_acc[3 * ti + 1] = ay;
      r_acc[3 * ti + 2] = az;
      r_nint[ti] = nint;
    }
}

#pragma clang fp contract(fast)

// =============================================================================================
//  group walk
// =============================================================================================
#define GW_MAXWAVES 12      // fused kernel: waves per persistent workgroup (3 per SIMD at 168 VGPRs)
#define GW_STACK 8192       // fused kernel: pending-node LIFO per wave (global scratch)
// one pool per wave (the species are evaluated one after the other): 128 x (double4 pos/mass, 4 floats = position relative to
// the group's box centre and its square for the packed-fp32 reach pre-test, 1 byte = softening type)
// per wave: [NULL entry (double4)] [128 x double4 position/mass] [4 x 128 floats] [128 type bytes] -- 16-byte multiple
#define GW2_WAVE_LDS (sizeof(double4) + (sizeof(double4) + 4 * sizeof(float) + 1) * 128)
#define GW_NLEAF 8          // an opened node with <= NLEAF particles hands over its particles directly

// =============================================================================================
//  group walk: traversal and force evaluation are separate phases per group (and, by default, separate kernels).
//
//  Phase 1 is a cooperative traversal (64 pending nodes tested per round, one per lane); accepted monopoles / particle
//  leaves are only recorded as item indices in a global scratch list.  Phase 2 visits that list in a
//  golden-ratio stride order, so every chunk of 64 items is an even sample of the whole neighbourhood of the
//  group instead of one corner of it; each lane tests the 64 chunk entries (LDS broadcast) against ITS OWN
//  target and keeps a 64-bit hit mask per source species; the force loop then lets every lane walk its own
//  bits (ES per trip).  Per-lane lists remove the bounding-box waste of shared lists (a target needs ~375 of
//  the ~2550 entries its group collects) and the stride order balances the lanes: the force loop runs ~545 slots per
//  lane and group instead of ~1100 (eight sub-group lists, an earlier version of this kernel) / ~1900 (one shared list).
// =============================================================================================
#define GW2_ITEMS 16384      // item scratch per wave and source species (global); phase 2 runs early if it would overflow

// MODE 0: fused (traversal + evaluation per group, per-wave scratch).  MODE 1: traversal only -- one wave per group of the
// batch [g_first, g_first+g_cnt), no LDS, few registers, so that many waves hide the dependent node fetches; the item list
// of group k and source species g goes to region_base + k*(NG*lcap+scap) + g*lcap (the pending-node LIFO sits behind
// the lists) and its length to gcount[k*NG+g].  MODE 2: evaluation only -- persistent workgroups with the tables in LDS run phase 2 over those lists.
#define GW3_LIST_MIN 4096     // split walk: item ints per group and source species (lcap; grown by the host after an overflow)
#define GW3_LIST_MAX 65536
#define GW3_STK_MIN 4096      // split walk: pending-node LIFO ints per group (scap; grown likewise)
#define GW3_STK_MAX 65536
#ifndef GW_DIRECT
#define GW_DIRECT 1   // the tree-only force loop in assembly: 1 = ER_DIRECT_ASM (one entry per trip, the next one in flight), 2 = ER_DIRECT2_ASM (two
                      // interleaved entries per trip)
#endif
#ifndef GW2_ES
#define GW2_ES 1   // measured at C4 (round 2, after the table-bin exp and the expanded-form masks): 1 -> 91.5 ms, 2 -> 96.8 ms
#endif
#define GW2_MAXWAVES 16      // evaluation kernel: 4 waves per SIMD (128 VGPRs), 5 KB of LDS each beside the tables
#define GW3_TBLOCK 256       // traversal kernel: 4 groups per workgroup
#define GW3_RING 1024        // traversal kernel: LIFO positions mirrored in LDS per wave (a round pushes at most 512)

template <int NG, bool PM, bool YUK, bool TAB_LDS, bool LATT, int MODE>
__global__ __launch_bounds__(MODE == 1 ? GW3_TBLOCK : (MODE == 2 ? GW2_MAXWAVES : GW_MAXWAVES) * 64) void k_walk_group2(
    TreeView tv, const double4 *__restrict__ s_pm, const unsigned char *__restrict__ s_type,
    const double *__restrict__ s_oldacc, const unsigned char *__restrict__ s_active,
    const double *__restrict__ table, WalkParams wp, long long t_first, long long t_count, int *__restrict__ counter,
    int *__restrict__ stack_base, int *__restrict__ err_flag, double *__restrict__ r_acc, int *__restrict__ r_nint,
    int *__restrict__ region_base, int *__restrict__ gcount, long long g_first, long long g_cnt, int lcap, int scap,
    int *__restrict__ glist, int S, int G0, const int *__restrict__ tlist, int SG)
{
  // SG (split walk): SG consecutive groups of G targets share ONE traversal and one set of item lists -- a traversal unit.
  // The evaluation kernel culls the unit's lists against the box of its own group, so pool, masks and forces are those of a
  // per-group list; the lists (HBM traffic of the hand-over) and the traversal's work shrink by ~SG/1.7.  MODE 1 runs over
  // units (g_first, g_cnt, the group index are unit indices), MODE 2 over groups.
  constexpr int ES = GW2_ES;   // entries per force-loop trip (independent instruction streams)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  __shared__ int ring_s[MODE == 1 ? (GW3_TBLOCK / 64) * GW3_RING : 1];   // traversal kernel: LDS mirror of each wave's LIFO top
  int *const ring = ring_s + (MODE == 1 ? (threadIdx.x >> 6) * GW3_RING : 0);
  // LDS: [tables (if TAB_LDS)] [exp table 32] [per wave: chunk pool 64 x (double4 pos/mass, double h, uchar species)]
  double *tab_s = reinterpret_cast<double *>(smem);
  const int ntabs = wp.ntab_lds + wp.exp_tab;   // distinct short-range tables [+ the exp(-ym r_bin) table]
  const size_t tab_bytes = (PM && TAB_LDS) ? sizeof(double) * ntabs * NTAB : 0;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  double *expT = reinterpret_cast<double *>(smem + tab_bytes);
  unsigned char *wbase = smem + tab_bytes + 40 * sizeof(double) + (size_t)wave * GW2_WAVE_LDS;
  double *fsT = expT + 32;   // softening length per particle type (8 entries; index 7 = empty node)
  double4 *lpos = reinterpret_cast<double4 *>(wbase) + 1;   // lpos[-1]: the NULL entry
  float *lfx = reinterpret_cast<float *>(wbase + sizeof(double4) * (2 * WAVE + 1));
  float *lfy = lfx + 2 * WAVE, *lfz = lfx + 4 * WAVE, *le2 = lfx + 6 * WAVE;
  unsigned char *lty = wbase + sizeof(double4) + (sizeof(double4) + 4 * sizeof(float)) * 2 * WAVE;
  if(MODE != 1)
    {
      if(threadIdx.x < 32)
        expT[threadIdx.x] = exp2(-(double)threadIdx.x / 32.0);
      if(threadIdx.x >= 32 && threadIdx.x < 40)
        fsT[threadIdx.x - 32] = threadIdx.x - 32 < NGRAVS_NTYPES ? wp.fsoft[threadIdx.x - 32] : 0.0;
      if(PM && TAB_LDS)
        for(int t = threadIdx.x; t < ntabs * NTAB; t += blockDim.x)
          {
            const int u = t / NTAB;
            tab_s[t] = table[(size_t)(u < wp.ntab_lds ? wp.slot_src[u] : NG * NG) * NTAB + (t % NTAB)];
          }
      __syncthreads();
    }
  const double *tabp = (PM && TAB_LDS) ? tab_s : table;
  if(MODE != 1 && lane == 0)
    {
      // the pool never holds more than 63 + 64 entries: slot 127 is free for the NULL entry (far away, massless, unsoftened)
      double4 z;
      z.x = z.y = z.z = 1e10;
      z.w = 0.0;
      lpos[127] = z;
      lpos[-1] = z;   // a lane whose mask is exhausted computes index -1 (v_ffbl_b32 of 0), see the force loop
      lty[127] = 7;   // fsT[7] = 0: unsoftened
      lfx[127] = lfy[127] = lfz[127] = 1e10f;
      le2[127] = 3e20f;
    }
  // item lists, one per source species: per-wave scratch (fused) or the group's region (split)
  int *stack = nullptr, *lists[NG];
  const int LIST_CAP = MODE == 0 ? GW2_ITEMS : lcap;
  const int STK_CAP = MODE == 0 ? GW_STACK : scap;
#pragma unroll
  for(int g = 0; g < NG; g++)
    lists[g] = nullptr;
  if(MODE == 0)
    {
      stack = stack_base + ((size_t)blockIdx.x * (blockDim.x >> 6) + wave) * (GW_STACK + NG * GW2_ITEMS);
#pragma unroll
      for(int g = 0; g < NG; g++)
        lists[g] = stack + GW_STACK + g * GW2_ITEMS;
    }
  // MODE 0 walks all groups of the shard; the split kernels one batch of them
  // S lanes share one target (S = 1: 64 targets per wave).  With S > 1 a wave walks G = 64/S targets -- scattered targets
  // (sparse active sets, the outskirts) whose 64-target box would be far wider than the short-range reach -- and the S lanes
  // of a target split the pool entries among them (entry j goes to lane j mod S), so all 64 lanes keep evaluating; their
  // partial sums are added at the end.  S = 64 is one target per wave: the box is the target itself and every group test
  // the reference's own per-target test.
  // glist: MODE 1 appends the groups whose lists or LIFO outgrew their region (counter[2] = how many, counter[3] = how many
  // of them by the LIFO); MODE 0 with a glist walks exactly those groups (of G0 targets each) again in sub-groups of G.
  const int G = WAVE / S;
  const long long ngroups = (MODE == 0 && !glist) ? (t_count + G - 1) / G : (MODE == 0 ? g_cnt * (G0 / G) : g_cnt);
  unsigned long long lane_pat = ~0ull;   // the pool entries this lane evaluates
  if(S > 1)
    {
      lane_pat = 0;
      for(int j = lane & (S - 1); j < WAVE; j += S)
        lane_pat |= 1ull << j;
    }
  const long long gbase = MODE == 0 ? 0 : g_first;
  const double BIG = 1e300;
  const double invbox = wp.box > 0 ? 1.0 / wp.box : 0.0;
  double h2max = 0;   // square of the largest softening length of any particle type (wave-uniform)
  bool usoft = true;  // all types share one softening length
#pragma unroll
  for(int q = 0; q < NGRAVS_NTYPES; q++)
    {
      h2max = fmax(h2max, wp.fsoft[q] * wp.fsoft[q]);
      usoft = usoft && wp.fsoft[q] == wp.fsoft[0];
    }

  // XCD-aware group assignment: the Peano order is cut into 8 contiguous segments, one per XCD (own L2), and a
  // workgroup pulls groups from the segment of the XCD it runs on (neighbouring groups share most of their
  // tree nodes and sources); an exhausted segment steals from the others.  Placement only affects speed.
  unsigned xcc = 0;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  xcc &= 7u;
  const long long seg = (ngroups + 7) / 8;
  int steal = 0;
  bool t_done = false;
  // walk statistics: summed per wave over its groups and flushed ONCE (device-scope atomics on four shared addresses,
  // one set per group, serialise in the memory controller and cost more than the traversal itself)
  unsigned long long acc_st[4] = {0, 0, 0, 0};
  for(;;)
    {
      long long grp = -1;
      if(MODE == 1)
        {
          // plain grid: workgroup b runs on XCD b%8 (round-robin dispatch), so give it groups of the b%8-th segment
          if(t_done)
            break;
          t_done = true;
          const long long nblk8 = gridDim.x / 8;
          const long long gl = ((long long)(blockIdx.x & 7) * nblk8 + (blockIdx.x >> 3)) * (GW3_TBLOCK / 64) + wave;
          if(gl >= ngroups)
            break;
          grp = gl;
        }
      while(MODE != 1 && steal < 8)
        {
          const int sx = (int)((xcc + steal) & 7u);
          int k = 0;
          if(lane == 0)
            k = atomicAdd(&counter[8 + sx], 1);
          k = __builtin_amdgcn_readfirstlane(k);
          const long long g0 = seg * sx + k;
          if(k < seg && g0 < ngroups)
            {
              grp = g0;
              break;
            }
          steal++;
        }
      if(grp < 0)
        break;
      long long tk_base = -1;
      if(MODE == 0 && glist)
        {
          const int R = G0 / G;
          const long long w = grp;
          grp = glist[w / R];
          tk_base = grp * G0 + (w % R) * G;
        }
      const long long ru = MODE == 2 ? grp / SG : grp;   // the unit's region in this batch
      if(MODE != 0)
        {
          int *base = region_base + (size_t)ru * ((size_t)NG * lcap + scap);
#pragma unroll
          for(int g = 0; g < NG; g++)
            lists[g] = base + (size_t)g * lcap;
          stack = base + (size_t)NG * lcap;
        }
      grp += gbase;
      // targets: the shard's Peano-ordered particles, or (individual timesteps: few active particles) the compacted list of
      // its active ones, so that a wave works for 64 active targets instead of the few a 64-particle stretch contains
      const int nsub = MODE == 1 ? SG : 1;
      const long long tk = (tk_base >= 0 ? tk_base : grp * G * nsub) + lane / S;
      const bool in_range = tk < t_count;
      const long long ti = tlist ? (in_range ? (long long)tlist[tk] : 0ll) : t_first + tk;
      const bool valid = in_range && (tlist != nullptr || (s_active[ti] & 1) != 0);
      double px = 0, py = 0, pz = 0, aold = 0, hT = 0;
      int tg = 0;
      if(valid)
        {
          double4 p = s_pm[ti];
          px = p.x;
          py = p.y;
          pz = p.z;
          int ptype = s_type[ti];
          tg = wp.t2g[ptype];
          hT = wp.fsoft[ptype];
          aold = wp.errtol_acc * s_oldacc[ti];
        }
      // per-lane extremes over the targets this lane stands for (one; the traversal of a unit: one per group of the unit)
      double mnx = valid ? px : BIG, mxx = valid ? px : -BIG, mny = valid ? py : BIG, mxy = valid ? py : -BIG;
      double mnz = valid ? pz : BIG, mxz = valid ? pz : -BIG, mna = valid ? aold : BIG, mnh = valid ? hT : BIG;
      bool anyvalid = valid;
      if(MODE == 1)
        for(int j = 1; j < nsub; j++)
          {
            const long long tkj = tk + (long long)j * G;
            const bool inr = tkj < t_count;
            const long long tij = tlist ? (inr ? (long long)tlist[tkj] : 0ll) : t_first + tkj;
            if(inr && (tlist != nullptr || (s_active[tij] & 1) != 0))
              {
                const double4 p = s_pm[tij];
                const int ptype = s_type[tij];
                mnx = fmin(mnx, p.x);
                mxx = fmax(mxx, p.x);
                mny = fmin(mny, p.y);
                mxy = fmax(mxy, p.y);
                mnz = fmin(mnz, p.z);
                mxz = fmax(mxz, p.z);
                mna = fmin(mna, wp.errtol_acc * s_oldacc[tij]);
                mnh = fmin(mnh, wp.fsoft[ptype]);
                anyvalid = true;
              }
          }
      if(!__any(anyvalid ? 1 : 0))
        {
          if(MODE == 1 && lane < NG)
            gcount[(grp - gbase) * NG + lane] = 0;
          continue;
        }
      // law coefficients and table row of this lane against the source species being evaluated (set by phase2)
      double cNg = 0, cYg = 0, cSg = 0;
      // BAM / NGRAVS_ACCUMULATOR laws (ngravs.c:495-668; tree-only, non-periodic wirings: this instantiation only): they depend on the
      // TARGET's mass and on the particle number behind the source, so the lane keeps its mass and the ids of its two laws
      // (tree-only instantiations have no tables: their TAB_LDS slot selects the variant that carries the BAM laws -- compiled in
      // unconditionally they cost the plain Newtonian tree-only walk 15 % of its evaluation kernel: registers, a runtime test per use)
      constexpr bool BAMCAP = !PM && !LATT && !YUK && TAB_LDS;
      double pmassT = 1.0;
      int lawA = 0, lawS = 0;
      if constexpr(BAMCAP)
        if(wp.bam && valid)
          pmassT = s_pm[ti].w;
      const double *trow = tabp;
      const double *const etab = (PM && TAB_LDS) ? tab_s + (size_t)wp.ntab_lds * NTAB : table + (size_t)NG * NG * NTAB;
      // group bounding box and the conservative scalars
      double lox = wave_min(mnx), hix = wave_max(mxx);
      double loy = wave_min(mny), hiy = wave_max(mxy);
      double loz = wave_min(mnz), hiz = wave_max(mxz);
      const double bcx = wave_uniform(0.5 * (lox + hix)), bcy = wave_uniform(0.5 * (loy + hiy)), bcz = wave_uniform(0.5 * (loz + hiz));
      const double bhx = wave_uniform(0.5 * (hix - lox)), bhy = wave_uniform(0.5 * (hiy - loy)), bhz = wave_uniform(0.5 * (hiz - loz));
      const double aold_min = wave_min(mna);
      const double hT_min = wave_min(mnh);
      // may sources be wrapped once per group (relative to the box centre) instead of per pair?
      const double bhmax = fmax(bhx, fmax(bhy, bhz));
      // (wave-uniform by construction; read through an SGPR so that the branches on them are scalar branches)
      const bool prewrap = __builtin_amdgcn_readfirstlane(
                               (int)(wp.periodic && PM && (wp.boxhalf - bhmax) * (wp.boxhalf - bhmax) > wp.reach2 &&
                                     (wp.boxhalf - bhmax) > 0)) != 0;
      const bool lanewrap = wp.periodic && !prewrap;
      // A pre-wrapped pool holds positions RELATIVE to the box centre (what the cull computes anyway; adding the centre back
      // would cost 9 VALU per item and round once more), so the force loop subtracts the target's relative position.
      const double tpx = prewrap ? px - bcx : px, tpy = prewrap ? py - bcy : py, tpz = prewrap ? pz - bcz : pz;
      // (1) A group whose whole region [box - reach, box + reach] lies inside the periodic box needs no image arithmetic
      // at all: with every source inside [0, L] (wp.src_in_box) a plain difference IS the nearest-image difference for
      // everything within reach, and everything else fails the cull either way.
      bool nowrap = !wp.periodic;
      if(PM && wp.periodic && wp.src_in_box)
        {
          const double rl_ = __builtin_sqrt(wp.reach2);
          nowrap = __builtin_amdgcn_readfirstlane((int)(bcx - bhx - rl_ >= 0.0 && bcx + bhx + rl_ <= wp.box && bcy - bhy - rl_ >= 0.0 &&
                                                        bcy + bhy + rl_ <= wp.box && bcz - bhz - rl_ >= 0.0 &&
                                                        bcz + bhz + rl_ <= wp.box)) != 0;
        }
      // packed-fp32 reach pre-test (PM, no per-pair wrapping): positions relative to the box centre in fp32, threshold
      // widened by the worst-case rounding so that no true hit is lost; the force loop re-tests in fp64 (in[k])
      typedef float f16v __attribute__((ext_vector_type(16)));
      const bool fastmask = PM && !lanewrap;
      // r2 - thr = |e|^2 - 2 e.p + |p|^2 - thr with |e|^2 stored per pool entry: three packed FMAs per two entries.
      // Rounding: e and p are fp32 roundings of coordinates relative to the box centre (each component <= bhmax + reach), so
      // the true r2 moves by <= 4 rl dl; the evaluation itself makes <= 8 roundings of magnitudes <= M = 3 (2 bhmax + rl)^2.
      const float tfx = (float)(px - bcx), tfy = (float)(py - bcy), tfz = (float)(pz - bcz);
      const float m2x = -2.0f * tfx, m2y = -2.0f * tfy, m2z = -2.0f * tfz;
      float cth;
      {
        const double rl = __builtin_sqrt(wp.reach2);
        const double dl = 4.76837158203125e-07 * (bhmax + rl);   // 2^-21 x the largest relative coordinate
        const double M = 3.0 * (2.0 * bhmax + rl) * (2.0 * bhmax + rl);
        const double thr = (wp.reach2 + 4.0 * rl * dl + 1.0e-6 * M) * (1.0 + 2e-6);
        cth = (float)((double)tfx * tfx + (double)tfy * tfy + (double)tfz * tfz - thr);
        cth = cth - 1.2e-7f * __builtin_fabsf(cth);   // the cast may have rounded up: one ulp down
      }
      // The masks are a 64 x 64 x 4 matrix product on the matrix cores (v_mfma_f32_32x32x2_f32, exact f32):
      //   D[entry][target] = |e|^2  +  (ex, ey, ez, 1) . (-2 tx, -2 ty, -2 tz, |t|^2 - thr),   hit <=> D < 0.
      // Rows are pool entries, columns targets, one 32 x 32 tile per (entry block, target block), K = 4 as two instructions.
      // B operand of lane l = B[k = l >> 5][column l & 31]: the target of column c of block tb is lane 32 tb + c, so every lane
      // fetches the constants of that lane once per group; mB[tb][0] = (-2 tx | -2 ty), mB[tb][1] = (-2 tz | |t|^2 - thr).
      float mB[2][2];
      const int mh = lane >> 5;   // k index this lane supplies = half of the wave
#pragma unroll
      for(int tb = 0; tb < 2; tb++)
        {
          const int src = 32 * tb + (lane & 31);
          const float x_ = __shfl(m2x, src), y_ = __shfl(m2y, src), z_ = __shfl(m2z, src), c_ = __shfl(cth, src);
          mB[tb][0] = mh ? y_ : x_;
          mB[tb][1] = mh ? c_ : z_;
        }
      // A operand of lane l = A[row l & 31][k = l >> 5].  Row i carries pool entry 32 eb + 16 ((i >> 2) & 1) + 4 (i >> 3) + (i & 3):
      // with that assignment accumulator register r of a lane holds entry 16 (l >> 5) + r of the block, i.e. its 16 sign bits
      // are 16 consecutive mask bits and its C input |e|^2 is 16 consecutive floats.
      // One per-lane LDS address serves all three operand reads: mA0 = (ex | ey) of the lane's row; mA0 + 4 WAVE floats is ez
      // for the lower half (the upper half lands in |e|^2 and is replaced by the constant 1); the C rows follow from mh.
      const int mrow = 16 * ((lane >> 2) & 1) + 4 * ((lane & 31) >> 3) + (lane & 3);
      const float *const mA0 = (mh ? lfy : lfx) + mrow;

      double ax = 0, ay = 0, az = 0;
      int nint = 0;
      int st_entries = 0, st_nodes = 0, st_batches = 0;   // walk statistics (per group, wave-uniform)

      // ES list entries against this lane's target as ES independent straight-line streams (no branch on the common
      // path).  Measured: with 4 waves per SIMD the latencies are hidden by the other waves, and ES = 2 beats 4 (fewer
      // registers -> no spills in the loop, and a trip only rounds the longest lane's hit count up to a multiple of 2).
      // Inactive slots (a lane whose mask is exhausted) point at the pool's NULL entry (index 127: far away, mass 0), so the
      // common path needs no per-slot masking at all; the rare slot that passed the fp32 pre-test but fails the exact
      // r2 < reach2 test is removed under a wave-level branch.
      // actm[k]: the lane mask of act[k], handed in by the caller (who has it from its loop condition; a second ballot of the
      // same condition would cost 2 VALU)
      auto evalN = [&](auto lw_tag, auto et_tag, const int g, const double4 (&e)[ES], const int (&jj)[ES],
                       const unsigned long long (&actm)[ES]) {
        // this lane's bit of actm[k], extracted where it is needed (the rare branches) and not in the common path
        auto is_act = [&](int k) -> bool {
          unsigned long long am = actm[k];
          asm volatile("" : "+s"(am));
          return ((am >> lane) & 1ull) != 0;
        };
        constexpr bool LW = decltype(lw_tag)::value;
        constexpr bool ET = decltype(et_tag)::value;   // Yukawa factor through the table bins (wp.exp_tab, hoisted by the caller)
        double dx[ES], dy[ES], dz[ES], r2[ES], rinv[ES], r[ES], fac[ES], mw[ES];
        unsigned long long fpos = 0;   // lane masks, combined on the scalar unit (a ballot of a compound condition costs 2 VALU)
#pragma unroll
        for(int k = 0; k < ES; k++)
          {
            dx[k] = e[k].x - tpx;
            dy[k] = e[k].y - tpy;
            dz[k] = e[k].z - tpz;
            if(LW)
              {
                dx[k] = nearest(dx[k], wp.box, wp.boxhalf);
                dy[k] = nearest(dy[k], wp.box, wp.boxhalf);
                dz[k] = nearest(dz[k], wp.box, wp.boxhalf);
              }
            r2[k] = dx[k] * dx[k] + dy[k] * dy[k] + dz[k] * dz[k];
            mw[k] = e[k].w;
            if(PM)
              fpos |= actm[k] & __builtin_amdgcn_ballot_w64(!(r2[k] < wp.reach2));
          }
        if(PM && fpos != 0ull)                                            // rare: beyond the exact cut
          {
            asm volatile("; beyond the exact cut" ::: "memory");          // keeps this a branch (if-converted it costs 6 VALU per slot)
#pragma unroll
            for(int k = 0; k < ES; k++)
              {
                double r2o = r2[k];
                asm volatile("" : "+v"(r2o));   // opaque copy: the test is redone HERE, not hoisted into the common path as an integer
                const bool out = is_act(k) && !(r2o < wp.reach2);
                mw[k] = out ? 0.0 : mw[k];
                nint -= out ? 1 : 0;
              }
          }
        bool anysoft = false;
#pragma unroll
        for(int k = 0; k < ES; k++)
          {
            // self / coincident pairs stay finite (d = 0 kills them)
            const double q2 = r2[k] + 1e-290;
            double ri = __builtin_amdgcn_rsq(q2);
            ri = ri * (1.5 - 0.5 * q2 * ri * ri);                         // one Newton step: ~2^-51
            const double rr = q2 * ri;                                    // sqrt(r2) to ~2^-51
            rinv[k] = ri;
            r[k] = rr;
            const double ri2 = ri * ri;
            double f = cNg * ri2;
            int tab = 0;
            double xt = 0;
            if(PM)
              {
                xt = wp.asmthfac * rr;
                tab = (int)xt;                                            // saturating conversion, then clamped
                tab = tab < NTAB - 1 ? tab : NTAB - 1;
              }
            if(YUK)
              {
                double ex;
                if(PM && ET)
                  {
                    // exp(-ym r) = E[tab] exp(-u), u = ub * (position inside the bin), ub = ym/asmthfac < 1e-3: degree-4 Taylor
                    // (u^5/120 < 1e-17) in the bin fraction with pre-scaled coefficients ec[k-1] = ub^k/k!, Estrin form.
                    // (A slot beyond the table, whose fraction belongs to another bin, has been given mass 0 above.)
                    const double fb = __builtin_amdgcn_fract(xt);
                    const double f2 = fb * fb;
                    const double lo = __builtin_fma(fb, -wp.ec[0], 1.0);
                    const double hi = __builtin_fma(fb, -wp.ec[2], wp.ec[1]);
                    const double pz = __builtin_fma(f2, __builtin_fma(f2, wp.ec[3], hi), lo);
                    ex = etab[tab] * pz;
                  }
                else
                  ex = exp_neg_fast(rr * wp.ym, expT);
                f += cYg * ex * (wp.ym * ri + ri2);
              }
            if(PM)
              f -= wp.utor2wpi * trow[tab];
            fac[k] = f * mw[k] * ri;
            anysoft |= r2[k] < h2max;                                     // closer than the largest softening length at all?
          }
        if(wave_any(anysoft))                                             // rare: possibly inside the softening radius
          {
            asm volatile("; softened pair" ::: "memory");
#pragma unroll
            for(int k = 0; k < ES; k++)
              {
                const double h = __builtin_fmax(hT, fsT[lty[is_act(k) ? jj[k] : 127]]);     // the pair's softening (forcetree.c:1415-1417)
                const bool soft = r[k] < h;
                double h_inv = 1 / h, u = r[k] * h_inv;
                double v = (u < 0.5) ? (10.666666666667 + u * u * (32.0 * u - 38.4))
                                     : (21.333333333333 - 48.0 * u + 38.4 * u * u - 10.666666666667 * u * u * u -
                                        0.066666666667 / (u * u * u));
                double fs = cSg * mw[k] * h_inv * h_inv * h_inv * v;
                fac[k] = soft ? fs : fac[k];
              }
          }
        if constexpr(BAMCAP)
          if(wp.bam)
            {
#pragma unroll
              for(int k = 0; k < ES; k++)
                {
                  // forcetree.c:1534-1583 with AccelFxns / AccelSplines of the BAM family: f(target mass, source mass, r, N)
                  const int je = is_act(k) ? jj[k] : 127;
                  const double h = __builtin_fmax(hT, fsT[lty[je]]);
                  const double Nn = (double)__float_as_int(le2[je]);
                  double f = 0.0;
                  if(mw[k] != 0.0)   // (the NULL entry: no mass, and its slot holds no particle number)
                    f = r[k] >= h ? law_accel_ref(lawA, mw[k], r2[k], r[k], wp.ym, pmassT, Nn, wp.bam_eps) * rinv[k]
                                  : law_spline_ref(lawS, mw[k], h, r[k], pmassT, Nn, wp.bam_eps);
                  fac[k] = f;
                }
            }
#pragma unroll
        for(int k = 0; k < ES; k++)
          {
            ax = __builtin_fma(dx[k], fac[k], ax);
            ay = __builtin_fma(dy[k], fac[k], ay);
            az = __builtin_fma(dz[k], fac[k], az);
            if(LATT)
              {
                // periodic tree-only: every source also contributes its infinite lattice of images (forcetree.c:1605-1607);
                // here on the SAME (finer) interaction list as the nearest-image force
                double fx, fy, fz;
                const bool a_ = is_act(k);
                lat_lookup(table + ((size_t)tg * NG + g) * LAT_SZ, wp.fac_intp, a_ ? dx[k] : 0.0, a_ ? dy[k] : 0.0,
                           a_ ? dz[k] : 0.0, fx, fy, fz);
                ax = __builtin_fma(mw[k], fx, ax);
                ay = __builtin_fma(mw[k], fy, ay);
                az = __builtin_fma(mw[k], fz, az);
              }
          }
      };

      int st_iters = 0;
      int n_items[NG], sp = 1;
      bool bad = false;
#pragma unroll
      for(int g = 0; g < NG; g++)
        {
          n_items[g] = MODE == 2 ? __builtin_amdgcn_readfirstlane(gcount[ru * NG + g]) : 0;
          bad |= n_items[g] < 0;
        }
      auto STK = [&](int i) -> int & { return stack[i]; };
      // Traversal kernel: the top of the LIFO is mirrored in LDS (a ring of GW3_RING positions per wave, written through): a round
      // then starts from an LDS read instead of a global round trip for the 64 node indices it pops -- one of the dependent memory
      // stages of a round.  ring_lo: positions >= ring_lo are intact in the ring (a push at q overwrites the slot of q - GW3_RING).
      int ring_lo = 0;
      if(MODE == 2)
        sp = 0;
      else
        {
          bool from_root = true;
          if(PM && tv.ltab_level > 0 && wp.periodic)
            {
              // the cells of the start level that overlap [box - reach, box + reach], per axis (wave-uniform): sample the
              // interval every half cell, wrap into the periodic box, keep each new cell index (at most 4 per axis)
              const double rl = __builtin_sqrt(wp.reach2), cl = tv.ltab_cl, inv_cl = 1.0 / cl;
              const int nc = 1 << tv.ltab_level;
              const double blo[3] = {bcx - bhx - rl, bcy - bhy - rl, bcz - bhz - rl};
              const double bhi[3] = {bcx + bhx + rl, bcy + bhy + rl, bcz + bhz + rl};
              int cidx[3][4], cn[3];
              bool ok = true;
#pragma unroll
              for(int a = 0; a < 3; a++)
                {
                  cn[a] = 0;
#pragma unroll
                  for(int q = 0; q < 4; q++)
                    cidx[a][q] = 0;
                  if(!(bhi[a] - blo[a] < 3.0 * cl) || !(bhi[a] - blo[a] < wp.boxhalf))
                    ok = false;
                  int last = -1;
#pragma unroll
                  for(int sidx = 0; sidx <= 7; sidx++)
                    {
                      double x = blo[a] + 0.5 * cl * sidx;
                      if(sidx == 7 || x > bhi[a])
                        x = bhi[a];
                      x -= wp.box * __builtin_floor(x / wp.box);
                      int ci = (int)((x - tv.ltab_corner[a]) * inv_cl);
                      ci = ci < 0 ? 0 : (ci >= nc ? nc - 1 : ci);
                      if(ci != last)
                        {
                          if(cn[a] < 4)
                            {
#pragma unroll
                              for(int q = 0; q < 4; q++)
                                if(q == cn[a])
                                  cidx[a][q] = ci;
                              cn[a]++;
                            }
                          else
                            ok = false;
                          last = ci;
                        }
                    }
                }
              const int total = cn[0] * cn[1] * cn[2];
              if(ok && total <= WAVE)
                {
                  from_root = false;
                  if(lane < total)
                    {
                      const int i0 = lane / (cn[1] * cn[2]), rem_ = lane - i0 * (cn[1] * cn[2]);
                      const int i1 = rem_ / cn[2], i2 = rem_ - i1 * cn[2];
                      int c0 = cidx[0][0], c1 = cidx[1][0], c2 = cidx[2][0];
#pragma unroll
                      for(int q = 1; q < 4; q++)
                        {
                          c0 = i0 == q ? cidx[0][q] : c0;
                          c1 = i1 == q ? cidx[1][q] : c1;
                          c2 = i2 == q ? cidx[2][q] : c2;
                        }
                      const int v0 = tv.ltab[((size_t)c0 * nc + c1) * nc + c2];
                      STK(lane) = v0;
                      if constexpr(MODE == 1)
                        ring[lane] = v0;
                    }
                  sp = total;
                }
            }
          if(from_root && lane == 0)
            {
              STK(0) = 0;
              if constexpr(MODE == 1)
                ring[0] = 0;
            }
        }
      wave_sync();
      bool overflow = false, stk_overflow = false;   // a list or the LIFO is full
      if(MODE == 2 && bad)   // the traversal kernel overflowed this group's region (error flag already set)
        continue;

      // ---- phase 2: evaluate the recorded items ------------------------------------------------------------
      auto phase2 = [&](const int g, const int *__restrict__ items, const int n_) {
        // the list length is the same in every lane: in an SGPR the stride search below (integer remainders) runs on the scalar unit
        const int n = __builtin_amdgcn_readfirstlane(n_);
        wave_sync();
        st_entries += n;
        if(n == 0)
          return;
        cNg = wp.cN[tg][g];
        cYg = wp.cY[tg][g];
        cSg = wp.cS[tg][g];
        if constexpr(BAMCAP)
          if(wp.bam)
            {
              lawA = wp.law_accel[tg][g];
              lawS = wp.law_spline[tg][g];
            }
        if(PM && TAB_LDS)
          trow = tabp + (size_t)wp.tab_slot[tg * NG + g] * NTAB;
        else
          trow = tabp + ((size_t)tg * NG + g) * NTAB;
        // The list is visited in QUADS of four consecutive items (one 16-byte load per lane and four chunks: neighbouring
        // items are neighbours in space, so element b of 64 quads spread over the list is still an even sample of the whole
        // neighbourhood, and the list is read with a quarter of the cache lines a 4-byte gather per chunk touches), quads in a
        // golden-ratio stride order: quad (i * s) mod nq is visited i-th, s coprime with nq.
        // (The stride runs over M = 64 * nsuper slots, the quads padded to whole super-chunks: an ODD stride is coprime with 64, so only
        // the small factor nsuper is left for Euclid -- integer remainders are long instruction sequences on this hardware, and the
        // golden-ratio neighbourhood is its worst case; the at most 63 empty slots are skipped.)
        const int nq = (n + 3) >> 2;
        const int nsuper = (nq + WAVE - 1) / WAVE, nchunks = 4 * nsuper;
        const int M = nsuper * WAVE;
        int s_ = (int)(0.6180339887498949 * M) | 1;
        if(s_ >= M)
          s_ = 1;
        for(;;)
          {
            int a = nsuper, b = s_ % nsuper;
            while(b)
              {
                int t = a % b;
                a = b;
                b = t;
              }
            if(a == 1)
              break;
            s_ += 2;
            if(s_ >= M)
              {
                s_ = 1;
                break;
              }
          }
        const int step64 = (int)((64ll * s_) % M);
        int slot = (int)(((long long)lane * s_) % M);
        int npool = 0;
        // one extra pass (cc == nchunks) only drains what is left in the pool, so that the force loop exists once.
        // Software pipeline on the memory side: while chunk c is evaluated, the source records of chunk c+1 are in flight, and
        // the quad of item indices of the NEXT four chunks was requested four chunks ago.
        auto fetch_quad = [&](int sc, int4 &v, int &nvalid) {
          v.x = v.y = v.z = v.w = 0;
          nvalid = 0;
          if(sc < nsuper && slot < nq)
            {
              v = reinterpret_cast<const int4 *>(items)[slot];
              nvalid = n - 4 * slot;
              nvalid = nvalid > 4 ? 4 : nvalid;
            }
          slot += step64;
          slot = slot >= M ? slot - M : slot;
        };
        auto fetch_rec = [&](bool hv, int item, double4 &q, int &hs) {   // hs: softening TYPE of the source (fsT index)
          // (without a record q and hs keep their previous contents: the consumer tests `have` first)
          if(hv)
            {
              const int k = -1 - item;   // monopole: node * NG + g
              int nsrc = 0;              // BAM laws: Nparticles[] of the node (allvars.h:645-648), 1 for a particle; carried above the type bits
              if constexpr(BAMCAP)
                if(wp.bam)
                  nsrc = (item >= 0 ? 1 : tv.npart[k]) << 3;
              if(usoft)
                {
                  // one softening length for all types: a single 32-byte gather per item, no type / flag bytes (each of
                  // which would pull another cache line)
                  const double4 *src = item >= 0 ? s_pm + item : tv.mom + k;
                  q = *src;
                  hs = nsrc;
                }
              else
                {
                  // record and type / flag word are fetched together; the softening length comes from the LDS copy of the
                  // table (a lookup in the kernel arguments would be a third, dependent, memory round trip)
                  const bool isp = item >= 0;
                  const double4 *src = isp ? s_pm + item : tv.mom + k;
                  q = *src;
                  int ty;
                  if(isp)
                    ty = s_type[item];
                  else
                    ty = (tv.flags[k / NG] >> 2) & 7;
                  hs = ty | nsrc;
                }
            }
        };
        int4 qd;                   // item quad of the current four chunks (the next one is requested when this one is used up:
        int nv;                    // a second register set for it costs more in spills than the exposed latency, which the
        fetch_quad(0, qd, nv);     // other waves of the SIMD cover)
        double4 q1;
        q1.x = q1.y = q1.z = q1.w = 0;
        int hs1 = 0;
        bool have1 = nv > 0;
        fetch_rec(have1, qd.x, q1, hs1);
        double4 *pp = lpos;
        unsigned char *ph = lty;
        for(int cc = 0; cc <= nchunks; cc++)
          {
            const bool last = cc == nchunks;
            // The record of this chunk is consumed (culled, compacted into the pool) BEFORE the next one is requested into
            // the same registers -- no copies -- and the request is in flight during the mask / force section below.
            double4 q = q1;
            const int hs = hs1;
            bool live = have1 && q.w != 0.0;
            double ex = q.x - bcx, ey = q.y - bcy, ez = q.z - bcz;
            if(!nowrap)
              {
                ex = nearest_abs(ex, wp.box, invbox);   // a tie (|ex| = box/2) is far beyond any reach
                ey = nearest_abs(ey, wp.box, invbox);   // a tie (|ey| = box/2) is far beyond any reach
                ez = nearest_abs(ez, wp.box, invbox);   // a tie (|ez| = box/2) is far beyond any reach
              }
            if(PM)
              {
                // a source farther than the cut from the whole bounding box contributes to no target
                double b0 = fmax(0.0, fabs(ex) - bhx), b1 = fmax(0.0, fabs(ey) - bhy), b2 = fmax(0.0, fabs(ez) - bhz);
                live = live && (b0 * b0 + b1 * b1 + b2 * b2 < wp.reach2);
              }
            if(prewrap)
              {
                q.x = ex;
                q.y = ey;
                q.z = ez;
              }
            // compact the live entries behind the ones already waiting in the pool (capacity 2 x 64)
            const unsigned long long lm = __ballot(live ? 1 : 0);
            if(live)
              {
                const int o = npool + lane_prefix(lm);
                pp[o] = q;
                ph[o] = (unsigned char)(hs & 7);
                const float fx = (float)ex, fy = (float)ey, fz = (float)ez;
                lfx[o] = fx;
                lfy[o] = fy;
                lfz[o] = fz;
                le2[o] = __builtin_fmaf(fz, fz, __builtin_fmaf(fy, fy, fx * fx));
                if constexpr(BAMCAP)
                  if(wp.bam)
                    le2[o] = __int_as_float(hs >> 3);   // tree-only walks build no fp32 masks: the slot carries the particle number
              }
            npool += __popcll(lm);
            {
              // chunk cc+1: element (cc+1) & 3 of its quad
              const int bn = (cc + 1) & 3;
              if(bn == 0)
                fetch_quad((cc + 1) / 4, qd, nv);
              const int itn = bn == 0 ? qd.x : (bn == 1 ? qd.y : (bn == 2 ? qd.z : qd.w));
              have1 = cc + 1 < nchunks && bn < nv;
              fetch_rec(have1, itn, q1, hs1);
            }
            wave_sync();
            if(npool >= WAVE || (last && npool > 0))
              {
                const int nc = npool < WAVE ? npool : WAVE;
                // ---- every lane marks the entries within reach of ITS target: two 32-bit words, constant bit per
                //      unrolled iteration (cndmask + or), LDS reads hoisted by the unroll
                unsigned int mlo = 0, mhi = 0;
                const bool direct = !PM && S == 1;   // tree-only: no masks at all (see the force loop)
                if(direct)
                  {
                  }
                else if(fastmask && !wp.exact_reach)   // tuning "walk_exact_reach": use the exact fp64 test below instead (tests)
                  {
                    // one 32-entry block per mask word: two tiles (target blocks), each two chained MFMAs on the C input
                    // |e|^2; the sign of every result is shifted into a 16-bit word (v_alignbit, highest register first), and one
                    // v_permlane32_swap hands the words to the lanes that own the targets
#pragma unroll
                    for(int w = 0; w < 2; w++)
                      {
                        const float a0 = mA0[32 * w];
                        const float a1r = mA0[4 * WAVE + 32 * w];   // lfz[row] for the lower half (lfz = lfx + 4 WAVE floats)
                        const float a1 = mh ? 1.0f : a1r;
                        const float *const mC = le2 + 16 * mh;
                        unsigned int wt[2];
#pragma unroll
                        for(int tb = 0; tb < 2; tb++)
                          {
                            f16v acc;
#pragma unroll
                            for(int q = 0; q < 4; q++)
                              {
                                const float4 c4 = *reinterpret_cast<const float4 *>(mC + 32 * w + 4 * q);
                                acc[4 * q + 0] = c4.x;
                                acc[4 * q + 1] = c4.y;
                                acc[4 * q + 2] = c4.z;
                                acc[4 * q + 3] = c4.w;
                              }
                            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, mB[tb][0], acc, 0, 0, 0);
                            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, mB[tb][1], acc, 0, 0, 0);
                            unsigned int word = 0;
#pragma unroll
                            for(int r = 15; r >= 0; r--)
                              word = __builtin_amdgcn_alignbit(word, __float_as_uint(acc[r]), 31);
                            wt[tb] = word;
                          }
                        // lanes 32-63 of wt[0] <-> lanes 0-31 of wt[1]: afterwards [0] = this lane's target against entries
                        // 0-15 of the block, [1] = against entries 16-31
                        const auto sw = __builtin_amdgcn_permlane32_swap(wt[0], wt[1], false, false);
                        const unsigned int word = sw[0] | (sw[1] << 16);
                        if(w == 0)
                          mlo = word;
                        else
                          mhi = word;
                      }
                    // entries beyond nc are stale; inactive lanes take nothing
                    const unsigned long long okm = !valid ? 0ull : (nc >= 64 ? ~0ull : ((1ull << nc) - 1ull));
                    mlo &= (unsigned int)okm;
                    mhi &= (unsigned int)(okm >> 32);
                  }
                else
                  {
#pragma unroll
                    for(int w = 0; w < 2; w++)
                      {
                        unsigned int word = 0;
#pragma unroll
                        for(int b = 0; b < 32; b++)
                          {
                            const int j = 32 * w + b;
                            if(j < nc)
                              {
                                const double4 e = pp[j];
                                double dx = e.x - tpx, dy = e.y - tpy, dz = e.z - tpz;
                                if(lanewrap)
                                  {
                                    dx = nearest(dx, wp.box, wp.boxhalf);
                                    dy = nearest(dy, wp.box, wp.boxhalf);
                                    dz = nearest(dz, wp.box, wp.boxhalf);
                                  }
                                const double r2 = dx * dx + dy * dy + dz * dz;
                                const bool hit = valid && (PM ? (r2 < wp.reach2) : true);
                                word |= hit ? (1u << b) : 0u;
                              }
                          }
                        if(w == 0)
                          mlo = word;
                        else
                          mhi = word;
                      }
                  }
                // ---- force loop: every lane walks its own bits, ES per trip
                unsigned long long m = (((unsigned long long)mhi << 32) | mlo) & lane_pat;
                nint += direct ? (valid ? nc : 0) : __popcll(m);   // evalN takes the (rare) slots beyond the exact cut off again
                bool direct_done = false;
                if constexpr(!PM && !LATT && ES == 1)
                  if(direct && !lanewrap && !(BAMCAP && wp.bam))
                    {
                      // the common tree-only case in assembly (eval_asm.inc ER_DIRECT_ASM: unrolled twice, the next entry in flight,
                      // accumulators in place); a lane without a target takes part with zero coefficients
                      typedef __attribute__((address_space(3))) unsigned char *lds_p;
                      unsigned ptr = (unsigned)(unsigned long)(lds_p)(unsigned char *)pp;
                      const unsigned tya = (unsigned)(unsigned long)(lds_p)lty;
                      const unsigned fsa = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(unsigned long)(lds_p)(unsigned char *)fsT);
                      const double cNv = valid ? cNg : 0.0, cSv = valid ? cSg : 0.0;
                      const int ncs = __builtin_amdgcn_readfirstlane(nc);
                      const double h2u = wave_uniform(h2max);
#if GW_DIRECT == 2
                      asm volatile(ER_DIRECT2_ASM
                                   : [ax] "+v"(ax), [ay] "+v"(ay), [az] "+v"(az), [ptr] "+v"(ptr)
                                   : [n] "s"(ncs), [tpx] "v"(tpx), [tpy] "v"(tpy), [tpz] "v"(tpz), [cN] "v"(cNv), [cS] "v"(cSv), [hT] "v"(hT),
                                     [tyb] "v"(tya), [fst] "s"(fsa), [tiny] "s"(1e-290), [h2max] "s"(h2u)
                                   : ER_DIRECT2_CLOBBERS);
#else
                      asm volatile(ER_DIRECT_ASM
                                   : [ax] "+v"(ax), [ay] "+v"(ay), [az] "+v"(az), [ptr] "+v"(ptr)
                                   : [n] "s"(ncs), [tpx] "v"(tpx), [tpy] "v"(tpy), [tpz] "v"(tpz), [cN] "v"(cNv), [cS] "v"(cSv), [hT] "v"(hT),
                                     [tyb] "v"(tya), [fst] "s"(fsa), [tiny] "s"(1e-290), [h2max] "s"(h2u)
                                   : ER_DIRECT_CLOBBERS);
#endif
                      st_iters += nc;
                      direct_done = true;
                      m = 0;
                    }
                if(direct && !direct_done)
                  {
                    // tree-only: every pool entry interacts with every target -- no masks to decode, all lanes read the
                    // same entries (LDS broadcast)
                    for(int j0 = 0; j0 < nc; j0 += ES)
                      {
                        st_iters += ES;
                        bool act[ES];
                        unsigned long long actm[ES];
                        int jj[ES];
                        double4 e[ES];
#pragma unroll
                        for(int k = 0; k < ES; k++)
                          {
                            act[k] = valid && j0 + k < nc;
                            actm[k] = __builtin_amdgcn_ballot_w64(act[k]);
                            jj[k] = j0 + k < nc ? j0 + k : 127;   // 127: the NULL entry
                            e[k] = pp[jj[k]];
                          }
                        if(!valid)   // a lane without a target must not accumulate
                          {
#pragma unroll
                            for(int k = 0; k < ES; k++)
                              e[k].w = 0.0;
                          }
                        if(lanewrap)
                          evalN(std::true_type{}, std::false_type{}, g, e, jj, actm);
                        else
                          evalN(std::false_type{}, std::false_type{}, g, e, jj, actm);
                      }
                    m = 0;
                  }
                // The loop exists once per (per-pair wrap, table-bin exp) combination, chosen by scalar branches OUTSIDE it:
                // inside, the only control flow is the two rare wave-level branches of evalN.
                auto bit_loop = [&](auto lw_tag, auto et_tag) {
                  for(;;)
                    {
                      unsigned long long actm[ES];
                      actm[0] = __builtin_amdgcn_ballot_w64(m != 0);
                      if(actm[0] == 0ull)
                        break;
                      st_iters += ES;
                      int jj[ES];
#pragma unroll
                      for(int k = 0; k < ES; k++)
                        {
                          if(k > 0)
                            actm[k] = __builtin_amdgcn_ballot_w64(m != 0);
                          // lowest set bit; an exhausted mask gives -1 (v_ffbl_b32 of 0 is -1, and -1 | 32 = -1): the NULL
                          // entry sits at pool index -1, so no select is needed
                          int jl, jh;
                          asm("v_ffbl_b32 %0, %1" : "=v"(jl) : "v"((unsigned int)m));
                          asm("v_ffbl_b32 %0, %1" : "=v"(jh) : "v"((unsigned int)(m >> 32)));
                          const unsigned int ju = (unsigned int)jl < ((unsigned int)jh | 32u) ? (unsigned int)jl : ((unsigned int)jh | 32u);
                          jj[k] = (int)ju;
                          m &= m - 1;
                        }
                      double4 e[ES];
#pragma unroll
                      for(int k = 0; k < ES; k++)
                        e[k] = pp[jj[k]];
                      evalN(lw_tag, et_tag, g, e, jj, actm);
                    }
                };
                if(lanewrap)
                  {
                    if(YUK && wp.exp_tab)
                      bit_loop(std::true_type{}, std::true_type{});
                    else
                      bit_loop(std::true_type{}, std::false_type{});
                  }
                else
                  {
                    if(YUK && wp.exp_tab)
                      bit_loop(std::false_type{}, std::true_type{});
                    else
                      bit_loop(std::false_type{}, std::false_type{});
                  }
                wave_sync();
                // move the remainder to the front
                const int rem = npool - nc;
                double4 tq;
                unsigned char th = 0;
                float t0 = 0, t1 = 0, t2 = 0, t3 = 0;
                tq.x = tq.y = tq.z = tq.w = 0;
                if(lane < rem)
                  {
                    tq = pp[WAVE + lane];
                    th = ph[WAVE + lane];
                    t0 = lfx[WAVE + lane];
                    t1 = lfy[WAVE + lane];
                    t2 = lfz[WAVE + lane];
                    t3 = le2[WAVE + lane];
                  }
                wave_sync();
                if(lane < rem)
                  {
                    pp[lane] = tq;
                    ph[lane] = th;
                    lfx[lane] = t0;
                    lfy[lane] = t1;
                    lfz[lane] = t2;
                    le2[lane] = t3;
                  }
                npool = rem;
                wave_sync();
              }
          }
        wave_sync();
      };
      auto phase2_all = [&]() {
        for(int g = 0; g < NG; g++)
          {
            const int *lg = lists[0];
            int ng_ = n_items[0];
#pragma unroll
            for(int q = 1; q < NG; q++)
              if(q == g)
                {
                  lg = lists[q];
                  ng_ = n_items[q];
                }
            phase2(g, lg, ng_);
          }
#pragma unroll
        for(int g = 0; g < NG; g++)
          n_items[g] = 0;
      };

      // ---- phase 1: cooperative traversal; items are only recorded ---------------------------------------------
      while(sp > 0)
        {
          {
            // one batch appends at most 64 monopoles and 8 x 64 particles to a list
            bool full = false;
#pragma unroll
            for(int g = 0; g < NG; g++)
              full |= n_items[g] + 9 * WAVE > LIST_CAP;
            if constexpr(MODE == 1)
              {
                if(full)   // the host falls back to the fused kernel, which evaluates early instead
                  {
                    overflow = true;
                    break;
                  }
              }
            else if(full)
              phase2_all();
          }
          // ---------------- test up to 64 pending nodes against the group's bounding box ----------------
          const int nb = sp < WAVE ? sp : WAVE;
          sp -= nb;
          st_nodes += nb;
          st_batches++;
          int my = -1;
          if(MODE == 1 && sp >= ring_lo)
            my = lane < nb ? ring[(sp + lane) & (GW3_RING - 1)] : -1;
          else
            my = lane < nb ? STK(sp + lane) : -1;
          wave_sync();
          // decision: 0 drop, 1 accept (monopoles), 2 open (children), 3 open as a leaf (all particles of the range)
          int dec = 0;
          int first = 0, count = 0;
          unsigned massmask = 0;
          bool pseudo_hit = false;
          int4 ch_lo = {-1, -1, -1, -1}, ch_hi = {-1, -1, -1, -1};
          if(my >= 0)
            {
              // The kernels are bound by the number of scattered lane-loads (the other waves hide the latency), so the
              // record is fetched in stages: geometry first, the multipole moments only if the cell is within reach, the child
              // links only if the node is opened.
              const double4 geo = tv.geo[my];
              const int fl = tv.flags[my];
              const double len = geo.w, half = 0.5 * len;
              double cx = geo.x - bcx, cy = geo.y - bcy, cz = geo.z - bcz;   // plain (inside-cell test has no NEAREST)
              double wx = cx, wy = cy, wz = cz;
              if(!nowrap)   // (a unit whose whole region lies inside the box needs no images: see `nowrap` above)
                {
                  wx = nearest_abs(cx, wp.box, invbox);
                  wy = nearest_abs(cy, wp.box, invbox);
                  wz = nearest_abs(cz, wp.box, invbox);
                }
              const int mst = (fl >> 2) & 7;
              bool drop = (mst == 7);   // empty
              if(PM && !drop)
                {
                  // (i) nothing inside the cell can be within the cut of any target
                  double q0 = fmax(0.0, fabs(wx) - bhx - half), q1 = fmax(0.0, fabs(wy) - bhy - half),
                         q2 = fmax(0.0, fabs(wz) - bhz - half);
                  if(q0 * q0 + q1 * q1 + q2 * q2 >= wp.reach2)
                    drop = true;
                }
              if(!drop)
                {
                  first = tv.first[my];
                  count = tv.count[my];
                  double r2min = BIG, r2far = 0, summass = 0;   // r2far: the largest of the species' distances, each to the nearest point of the box
#pragma unroll
                  for(int g = 0; g < NG; g++)
                    {
                      const double4 mom = tv.mom[(long long)my * NG + g];
                      summass += mom.w;
                      massmask |= (mom.w != 0.0) ? (1u << g) : 0u;
                      double dx = mom.x - bcx, dy = mom.y - bcy, dz = mom.z - bcz;
                      if(!nowrap)
                        {
                          dx = nearest_abs(dx, wp.box, invbox);
                          dy = nearest_abs(dy, wp.box, invbox);
                          dz = nearest_abs(dz, wp.box, invbox);
                        }
                      double a0 = fmax(0.0, fabs(dx) - bhx), a1 = fmax(0.0, fabs(dy) - bhy), a2 = fmax(0.0, fabs(dz) - bhz);
                      double r2g = a0 * a0 + a1 * a1 + a2 * a2;
                      r2min = r2g < r2min ? r2g : r2min;
                      r2far = r2g > r2far ? r2g : r2far;
                    }
                  // (ii) the reference's own cut (forcetree.c:1828-1862) holds for every target
                  if(PM && r2min > wp.rcut2)
                    {
                      double eff = wp.rcut + half;
                      if(fabs(wx) - bhx > eff || fabs(wy) - bhy > eff || fabs(wz) - bhz > eff)
                        drop = true;
                    }
                  if(!drop)
                    {
                      bool open;
                      if(wp.use_theta)
                        open = len * len > r2min * wp.theta2;
                      else
                        {
                          open = summass * len * len > r2min * r2min * aold_min;
                          if(!open)
                            open = (fabs(cx) - bhx < 0.60 * len) && (fabs(cy) - bhy < 0.60 * len) &&
                                   (fabs(cz) - bhz < 0.60 * len);
                        }
                      const double hs_node = usoft ? wp.fsoft[0] : wp.fsoft[mst];   // no dependent table load when all types share one length
                      // forcetree.c:1488-1499: a target opens a node of mixed softening if r2max -- the LARGEST of its distances to the
                      // species' centres of mass -- is inside the softening length.  Every target's r2max is at least r2far (each
                      // distance to the box is a lower bound), so r2far < h^2 whenever any target would open: conservative, and
                      // for a one-target box exactly the reference's test
                      if(!open && hT_min < hs_node && r2far < hs_node * hs_node && ((fl >> 5) & 1))
                        open = true;
                      // A top leaf whose particles were not imported (FLAG_PSEUDO: global monopoles, no children) cannot be opened.
                      // The import decision covers every node a TARGET may open under the criterion in force at the decomposition; the
                      // box of a group that spans several top leaves can come closer to a node than any of their boxes, and a host may
                      // have changed the criterion since.  Such a node is used as a monopole (it used to drop out of the force) and
                      // counted: ngravs_walk_unopened().
                      if(open && (fl & FLAG_PSEUDO))
                        {
                          open = false;
                          pseudo_hit = true;
                        }
                      if(open)
                        dec = ((fl & FLAG_BUCKET) || (count <= wp.nleaf && !(fl & FLAG_PARTIAL))) ? 3 : 2;
                      else
                        dec = 1;
                    }
                }
            }
          {
            const unsigned long long ph_ = __ballot(pseudo_hit ? 1 : 0);
            if(ph_ != 0ull && lane == 0)
              atomicAdd(&counter[4], __popcll(ph_));
          }
          if(dec == 2)
            {
              const int4 *cp = reinterpret_cast<const int4 *>(tv.child + 8 * (long long)my);
              ch_lo = cp[0];
              ch_hi = cp[1];
            }
          // record the items of this batch (appended in traversal order; phase 2 reads them in stride order)
#pragma unroll
          for(int g = 0; g < NG; g++)
            {
              const bool pg = dec == 1 && ((massmask >> g) & 1u);
              unsigned long long mask = __ballot(pg ? 1 : 0);
              if(pg)
                lists[g][n_items[g] + lane_prefix(mask)] = -1 - (my * NG + g);
              n_items[g] += __popcll(mask);
            }
          const int chv[8] = {ch_lo.x, ch_lo.y, ch_lo.z, ch_lo.w, ch_hi.x, ch_hi.y, ch_hi.z, ch_hi.w};
          // source species of the (up to 8) particles this lane is about to record, 2 bits each: eight independent
          // byte loads in flight instead of one dependent load per ballot round
          unsigned sp8 = 0;
          if(NG > 1)
            {
              // unconditional loads (index 0 when there is nothing to look up): all eight are in flight before the first
              // is used; a predicated load per slot made the compiler wait for each in turn
              unsigned char ty8[8];
#pragma unroll
              for(int q = 0; q < 8; q++)
                {
                  int pi = 0;
                  if(dec == 2 && chv[q] <= -2)
                    pi = -2 - chv[q];
                  if(dec == 3 && q < count)
                    pi = first + q;
                  ty8[q] = s_type[pi];
                }
#pragma unroll
              for(int q = 0; q < 8; q++)
                sp8 |= ((wp.t2g_packed >> (2 * ty8[q])) & 3u) << (2 * q);   // only the fields of real items are read later
            }
          // ---- append: every lane holds up to 8 things to record -- child nodes for the LIFO (dec 2) and particles for the
          //      item lists (children of an opened node, or the first 8 particles of a small node's range, dec 3).  One packed
          //      wave prefix sum (10 bits per field: at most 64 x 8 = 512 per kind) gives every lane its write offsets, and it
          //      then writes its own items back to back: siblings stay adjacent on the LIFO (and in memory, when they are
          //      popped and tested together), and the 8 x (1 + NG) ballot rounds of a slot-by-slot append are gone.
          if(__any(dec >= 2))
            {
              int itemv[8];
              unsigned cnt_pack = 0;   // nodes | species-0 particles << 10 | species-1 particles << 20
              unsigned cnt_g2 = 0;     // species-2 particles (N_GRAVS = 3)
#pragma unroll
              for(int q = 0; q < 8; q++)
                {
                  int it = -1;   // >= 0: node for the LIFO; <= -2: particle (-2 - index); -1: nothing
                  if(dec == 2)
                    it = chv[q];
                  else if(dec == 3 && q < count)
                    it = -2 - (first + q);
                  itemv[q] = it;
                  const int sgq = (int)((sp8 >> (2 * q)) & 3u);
                  const bool isp = it <= -2;
                  cnt_pack += it >= 0 ? 1u : (isp && sgq < 2 ? (1u << 10) << (10 * sgq) : 0u);
                  cnt_g2 += isp && sgq == 2 ? 1u : 0u;
                }
              unsigned inc = cnt_pack, inc2 = cnt_g2;   // inclusive wave scans
#pragma unroll
              for(int off = 1; off < WAVE; off <<= 1)
                {
                  const unsigned y = __shfl_up(inc, off);
                  inc += lane >= off ? y : 0u;
                  if(NG > 2)
                    {
                      const unsigned y2 = __shfl_up(inc2, off);
                      inc2 += lane >= off ? y2 : 0u;
                    }
                }
              const unsigned tot = __shfl(inc, WAVE - 1), tot2 = NG > 2 ? __shfl(inc2, WAVE - 1) : 0u;
              const unsigned exc = inc - cnt_pack, exc2 = inc2 - cnt_g2;
              const int tn = (int)(tot & 1023u);
              if(sp + tn > STK_CAP)
                {
                  overflow = true;
                  stk_overflow = true;
                }
              else
                {
                  // branch-free: the LIFO and the lists lie in one region, so every item is ONE predicated store to
                  // rbase[offset], the offset picked by selects (a per-kind branch cascade costs four times the instructions)
                  int *const rbase = stack;
                  const int o_l0 = (int)(lists[0] - stack), o_l1 = NG > 1 ? (int)(lists[NG > 1 ? 1 : 0] - stack) : 0,
                            o_l2 = NG > 2 ? (int)(lists[NG > 2 ? 2 : 0] - stack) : 0;
                  int on = sp + (int)(exc & 1023u);
                  int op0 = o_l0 + n_items[0] + (int)((exc >> 10) & 1023u);
                  int op1 = NG > 1 ? o_l1 + n_items[NG > 1 ? 1 : 0] + (int)((exc >> 20) & 1023u) : 0;
                  int op2 = NG > 2 ? o_l2 + n_items[NG > 2 ? 2 : 0] + (int)exc2 : 0;
#pragma unroll
                  for(int q = 0; q < 8; q++)
                    {
                      const int it = itemv[q];
                      const int sgq = (int)((sp8 >> (2 * q)) & 3u);
                      const bool isn = it >= 0, isp = it <= -2;
                      const bool p0 = isp && sgq == 0, p1 = NG > 1 && isp && sgq == 1, p2 = NG > 2 && isp && sgq == 2;
                      const int off = isn ? on : (p0 ? op0 : (p1 ? op1 : op2));
                      const int val = isn ? it : -2 - it;
                      if(isn || isp)
                        rbase[off] = val;
                      if constexpr(MODE == 1)
                        if(isn)
                          ring[on & (GW3_RING - 1)] = val;
                      on += isn ? 1 : 0;
                      op0 += p0 ? 1 : 0;
                      op1 += p1 ? 1 : 0;
                      op2 += p2 ? 1 : 0;
                    }
                  sp += tn;
                  ring_lo = sp - GW3_RING > ring_lo ? sp - GW3_RING : ring_lo;
                  n_items[0] += (int)((tot >> 10) & 1023u);
                  if(NG > 1)
                    n_items[NG > 1 ? 1 : 0] += (int)((tot >> 20) & 1023u);
                  if(NG > 2)
                    n_items[NG > 2 ? 2 : 0] += (int)tot2;
                }
            }
          // coincident-key buckets (more than 8 particles in one leaf): the rest of the range, 64 lanes x k-th particle
          if(!overflow && __any(dec == 3 && count > 8))
            {
              int kmax = 0;
              {
                int cmine = (dec == 3) ? count : 0;
                for(int off = 32; off > 0; off >>= 1)
                  {
                    int o = __shfl_xor(cmine, off);
                    cmine = o > cmine ? o : cmine;
                  }
                kmax = cmine;
              }
              for(int k = 8; k < kmax; k++)
                {
                  bool full = false;
#pragma unroll
                  for(int g = 0; g < NG; g++)
                    full |= n_items[g] + WAVE > LIST_CAP;
                  if(full)
                    {
                      overflow = true;   // a bucket larger than the scratch list: not a sane input
                      break;
                    }
                  const bool more = (dec == 3) && k < count;
                  const int sgp = (NG > 1 && more) ? (int)((wp.t2g_packed >> (2 * s_type[first + k])) & 3u) : 0;
#pragma unroll
                  for(int g = 0; g < NG; g++)
                    {
                      const bool pg = more && sgp == g;
                      unsigned long long pmask = __ballot(pg ? 1 : 0);
                      if(pg)
                        lists[g][n_items[g] + lane_prefix(pmask)] = first + k;
                      n_items[g] += __popcll(pmask);
                    }
                }
            }
          if(overflow)
            break;
          wave_sync();
        }
      if constexpr(MODE == 1)
        {
#pragma unroll
          for(int g = 0; g < NG; g++)
            if(lane == g)
              gcount[(grp - gbase) * NG + g] = overflow ? -1 : n_items[g];
        }
      else if(!overflow)
        phase2_all();
      if(MODE == 1)
        {
          if(lane == 0)
            {
              gcount[NG * g_cnt + (grp - gbase)] = st_nodes;
              gcount[(NG + 1) * g_cnt + (grp - gbase)] = st_batches;
            }
        }
      else
        {
          if(MODE == 2 && (grp - gbase) % SG == 0)   // the unit's traversal statistics, once
            {
              const long long u_cnt = (g_cnt + SG - 1) / SG;
              st_nodes = __builtin_amdgcn_readfirstlane(gcount[NG * u_cnt + ru]);          // wave-uniform: keep the statistics in SGPRs
              st_batches = __builtin_amdgcn_readfirstlane(gcount[(NG + 1) * u_cnt + ru]);
            }
          acc_st[0] += (unsigned long long)st_entries;
          acc_st[1] += (unsigned long long)st_nodes;
          acc_st[2] += (unsigned long long)st_batches;
          acc_st[3] += (unsigned long long)st_iters;
        }
      if(overflow)
        {
          if(lane == 0)
            {
              if(MODE == 1)
                {
                  glist[atomicAdd(&counter[2], 1)] = (int)grp;   // finished by the fused kernel in the same step
                  if(stk_overflow)
                    atomicAdd(&counter[3], 1);
                }
              else
                atomicOr(err_flag, stk_overflow ? 2 : 1);
            }
          continue;
        }
      if(MODE != 1 && S > 1)
        {
          for(int off = 1; off < S; off <<= 1)   // the S lanes of a target hold partial sums
            {
              ax += __shfl_xor(ax, off);
              ay += __shfl_xor(ay, off);
              az += __shfl_xor(az, off);
              nint += __shfl_xor(nint, off);
            }
        }
      if(MODE != 1 && valid && (lane & (S - 1)) == 0)
        {
          r_acc[3 * ti + 0] = ax;
          r_acc[3 * ti + 1] = ay;
          r_acc[3 * ti + 2] = az;
          r_nint[ti] = nint;
        }
    }
  if(MODE != 1 && lane == 0)
    {
      unsigned long long *st64 = reinterpret_cast<unsigned long long *>(counter + 16);
#pragma unroll
      for(int q = 0; q < 4; q++)
        if(acc_st[q])
          atomicAdd(&st64[q], acc_st[q]);
    }
}

// =============================================================================================
//  post-processing (gravtree.c:318-341): OldAcc = |GravAccel + GravPM/G|, GravAccel *= G
// =============================================================================================
// (also Nf and the interaction sum of gravtree.c:74-78, 408-447: sums[0] += interactions of the walked particles, sums[1] += their
// number.  Grid-stride over the targets with a few thousand waves, so that the two atomics per wave at the end stay a few
// thousand: one pair per 64 particles -- a million atomics on two addresses -- cost 25 ms.)
__global__ void k_finish(long long t_first, long long t_count, const unsigned char *__restrict__ s_active,
                         double *__restrict__ r_acc, const double *__restrict__ r_pm, double *__restrict__ r_oldacc,
                         double G, int have_pm, const int *__restrict__ r_nint, double *__restrict__ sums)
{
  double s = 0, na = 0;
  for(long long k = blockIdx.x * (long long)blockDim.x + threadIdx.x; k < t_count; k += (long long)gridDim.x * blockDim.x)
    {
      const long long i = t_first + k;
      if(!(s_active[i] & 1))
        continue;
      s += (double)r_nint[i];
      na += 1.0;
      double ax = r_acc[3 * i], ay = r_acc[3 * i + 1], az = r_acc[3 * i + 2];
      double bx = ax, by = ay, bz = az;
      if(have_pm)
        {
          bx += r_pm[3 * i] / G;
          by += r_pm[3 * i + 1] / G;
          bz += r_pm[3 * i + 2] / G;
        }
      r_oldacc[i] = sqrt(bx * bx + by * by + bz * bz);
      r_acc[3 * i] = ax * G;
      r_acc[3 * i + 1] = ay * G;
      r_acc[3 * i + 2] = az * G;
    }
  for(int off = 32; off > 0; off >>= 1)
    {
      s += __shfl_down(s, off);
      na += __shfl_down(na, off);
    }
  // one pair of atomics per BLOCK (the two words are shared by the whole grid: 16 384 waves queueing there cost 0.3 ms)
  __shared__ double bs[4], bn[4];
  const int w = threadIdx.x >> 6;
  if((threadIdx.x & 63) == 0)
    {
      bs[w] = s;
      bn[w] = na;
    }
  __syncthreads();
  if(threadIdx.x == 0)
    {
      for(int q = 1; q < (int)(blockDim.x >> 6); q++)
        {
          s += bs[q];
          na += bn[q];
        }
      if(na > 0)
        {
          atomicAdd(&sums[0], s);
          atomicAdd(&sums[1], na);
        }
    }
}

// =============================================================================================
//  direct summation for a list of targets (forcetree.c:3428-3548, no Ewald term)
// =============================================================================================
#pragma clang fp contract(off)
// targets: particles tidx[k] of the working set, or (tidx == nullptr) explicit records t_pm[k] / t_type[k] -- a test particle of
// ANOTHER task in the distributed gravity_forcetest(); own_only: the imported copies (active bit 1) are not sources, each task
// contributes its own particles and the host adds the partial sums up
__global__ __launch_bounds__(256) void k_direct(const double4 *__restrict__ s_pm, const unsigned char *__restrict__ s_type,
                                                 long long n, const int *__restrict__ tidx, long long nt, WalkParams wp,
                                                 LawIds li, double G, const double *__restrict__ lat, double *__restrict__ acc,
                                                 const double4 *__restrict__ t_pm = nullptr, const int *__restrict__ t_type = nullptr,
                                                 const unsigned char *__restrict__ s_active = nullptr, int own_only = 0)
{
  // one block per target, threads stride over sources, fp64 block reduction
  long long k = blockIdx.x;
  if(k >= nt)
    return;
  double4 p;
  int ptype;
  if(tidx)
    {
      const int t = tidx[k];
      p = s_pm[t];
      ptype = s_type[t];
    }
  else
    {
      p = t_pm[k];
      ptype = t_type[k];
    }
  const int tg = wp.t2g[ptype];
  double ax = 0, ay = 0, az = 0;
  for(long long i = threadIdx.x; i < n; i += blockDim.x)
    {
      if(own_only && (s_active[i] & 2))
        continue;
      double4 q = s_pm[i];
      int qt = s_type[i], sg = wp.t2g[qt];
      double h = wp.fsoft[qt] > wp.fsoft[ptype] ? wp.fsoft[qt] : wp.fsoft[ptype];
      double dx = q.x - p.x, dy = q.y - p.y, dz = q.z - p.z;
      if(wp.periodic)
        {
          dx = nearest(dx, wp.box, wp.boxhalf);
          dy = nearest(dy, wp.box, wp.boxhalf);
          dz = nearest(dz, wp.box, wp.boxhalf);
        }
      double r2 = dx * dx + dy * dy + dz * dz, r = sqrt(r2), u = r * (1 / h), fac;
      if(u >= 1)
        fac = law_accel_ref(li.accel[tg][sg], q.w, r2, r, wp.ym, p.w, 1.0, wp.bam_eps) / r;
      else
        fac = law_spline_ref(li.spline[tg][sg], q.w, h, r, p.w, 1.0, wp.bam_eps);
      ax += dx * fac;
      ay += dy * fac;
      az += dz * fac;
      if(lat && u > 1.0e-5)   // forcetree.c:3519-3528
        {
          double fx, fy, fz;
          lat_lookup(lat + ((size_t)tg * wp.ng + sg) * LAT_SZ, wp.fac_intp, dx, dy, dz, fx, fy, fz);
          ax += q.w * fx;
          ay += q.w * fy;
          az += q.w * fz;
        }
    }
  __shared__ double sh[3][4];
  for(int off = 32; off > 0; off >>= 1)
    {
      ax += __shfl_down(ax, off);
      ay += __shfl_down(ay, off);
      az += __shfl_down(az, off);
    }
  if((threadIdx.x & 63) == 0)
    {
      sh[0][threadIdx.x >> 6] = ax;
      sh[1][threadIdx.x >> 6] = ay;
      sh[2][threadIdx.x >> 6] = az;
    }
  __syncthreads();
  if(threadIdx.x == 0)
    {
      acc[3 * k] = (sh[0][0] + sh[0][1] + sh[0][2] + sh[0][3]) * G;
      acc[3 * k + 1] = (sh[1][0] + sh[1][1] + sh[1][2] + sh[1][3]) * G;
      acc[3 * k + 2] = (sh[2][0] + sh[2][1] + sh[2][2] + sh[2][3]) * G;
    }
}
#pragma clang fp contract(fast)

// =============================================================================================
//  host side
// =============================================================================================
void make_walk_params(const ngravs_ctx *c, WalkParams *wp)
{
  const ngravs_config_t &cfg = c->cfg;
  memset(wp, 0, sizeof(*wp));
  wp->ng = cfg.n_gravs;
  wp->periodic = cfg.periodic;
  wp->pm = cfg.pmgrid != 0;
  wp->use_theta = cfg.err_tol_theta != 0;
  wp->exact_reach = c->tune.walk_exact_reach;
  wp->nleaf = c->tune.walk_nleaf < 0 ? GW_NLEAF : c->tune.walk_nleaf;
  wp->box = cfg.box_size;
  wp->boxhalf = 0.5 * cfg.box_size;
  wp->theta2 = cfg.err_tol_theta * cfg.err_tol_theta;
  wp->errtol_acc = cfg.err_tol_force_acc;
  if(cfg.pmgrid)
    {
      wp->rcut = c->rcut;
      wp->rcut2 = c->rcut * c->rcut;
      wp->asmthfac = 0.5 / c->asmth * (NTAB / 3.0);          // forcetree.c:1708
      wp->utor2wpi = 1.0 / (M_PI * 4 * c->asmth * c->asmth);  // forcetree.c:1711
      double reach = NTAB / wp->asmthfac;                    // tabindex < NTAB  <=>  r < 6*asmth
      if(cfg.walk_mode == NGRAVS_WALK_GROUP)
        {
          double ru = cfg.group_reach > 0 ? cfg.group_reach : NGRAVS_GROUP_REACH;
          if(ru < 6.0)
            reach = ru * c->asmth;
        }
      wp->reach2 = reach * reach;
    }
  wp->ym = cfg.box_size > 0 ? cfg.yukawa_imass / cfg.box_size : 0.0;
  wp->bam_eps = cfg.bam_epsilon > 0 ? cfg.bam_epsilon : 1.31e-6;
  // distinct short-range tables of the wiring (same law pair <=> same table), and the bin-wise Yukawa factor
  {
    const int ng = cfg.n_gravs;
    wp->ntab_lds = 0;
    for(int k = 0; k < ng * ng; k++)
      {
        int found = -1;
        for(int u = 0; u < wp->ntab_lds && found < 0; u++)
          {
            const int q = wp->slot_src[u];
            if(cfg.law_accel[q / ng][q % ng] == cfg.law_accel[k / ng][k % ng] &&
               cfg.law_normed[q / ng][q % ng] == cfg.law_normed[k / ng][k % ng])
              found = u;
          }
        if(found < 0)
          {
            found = wp->ntab_lds++;
            wp->slot_src[found] = k;
          }
        wp->tab_slot[k] = found;
      }
    wp->exp_tab = 0;
    wp->inv_asmthfac = 0;
    if(cfg.pmgrid && wp->asmthfac > 0)
      {
        wp->inv_asmthfac = 1.0 / wp->asmthfac;
        wp->exp_tab = (wp->ym * wp->inv_asmthfac < 1.0e-3) ? 1 : 0;   // u^5/120 < 1e-17
      }
    // (a refit tree holds drifted positions the extent below never saw -- the reference does not wrap between decompositions,
    // predict.c:79-91 --, so the shortcut is off until the next decomposition has checked them)
    wp->src_in_box = (c->tree_refit || c->tree_stale) ? 0 : 1;
    for(int j = 0; j < 3; j++)
      if(!(c->pos_lo[j] >= 0.0 && c->pos_hi[j] <= cfg.box_size))
        wp->src_in_box = 0;
    const double ub = wp->ym * wp->inv_asmthfac;   // exponent across one table bin
    wp->ec[0] = ub;
    wp->ec[1] = ub * ub / 2.0;
    wp->ec[2] = ub * ub * ub / 6.0;
    wp->ec[3] = ub * ub * ub * ub / 24.0;
  }
  wp->bam = cfg_has_bam(cfg) ? 1 : 0;
  for(int i = 0; i < NG_MAX; i++)
    for(int j = 0; j < NG_MAX; j++)
      {
        wp->law_accel[i][j] = (i < cfg.n_gravs && j < cfg.n_gravs) ? cfg.law_accel[i][j] : NGRAVS_LAW_NONE;
        wp->law_spline[i][j] = (i < cfg.n_gravs && j < cfg.n_gravs) ? cfg.law_spline[i][j] : NGRAVS_SPLINE_NONE;
      }
  wp->fac_intp = cfg.box_size > 0 ? 2.0 * LAT_EN / cfg.box_size : 0.0;   // forcetree.c:3737
  for(int t = 0; t < NGRAVS_NTYPES; t++)
    {
      wp->fsoft[t] = cfg.force_softening[t];
      wp->t2g[t] = cfg.type_to_grav[t];
      wp->t2g_packed = (t == 0 ? 0u : wp->t2g_packed) | ((unsigned)(cfg.type_to_grav[t] & 3) << (2 * t));
    }
  for(int i = 0; i < NG_MAX; i++)
    for(int j = 0; j < NG_MAX; j++)
      {
        int law = (i < cfg.n_gravs && j < cfg.n_gravs) ? cfg.law_accel[i][j] : NGRAVS_LAW_NONE;
        int spl = (i < cfg.n_gravs && j < cfg.n_gravs) ? cfg.law_spline[i][j] : NGRAVS_SPLINE_NONE;
        wp->cN[i][j] = law == NGRAVS_LAW_NEWTON || law == NGRAVS_LAW_COLOYUK ? 1.0 : (law == NGRAVS_LAW_NEG_NEWTON ? -1.0 : 0.0);
        wp->cY[i][j] = law == NGRAVS_LAW_YUKAWA || law == NGRAVS_LAW_COLOYUK ? 1.0 : 0.0;
        wp->cS[i][j] = spl == NGRAVS_SPLINE_PLUMMER ? 1.0 : (spl == NGRAVS_SPLINE_NEG_PLUMMER ? -1.0 : 0.0);
      }
}

static void make_law_ids(const ngravs_ctx *c, LawIds *li)
{
  memset(li, 0, sizeof(*li));
  for(int i = 0; i < c->cfg.n_gravs; i++)
    for(int j = 0; j < c->cfg.n_gravs; j++)
      {
        li->accel[i][j] = c->cfg.law_accel[i][j];
        li->spline[i][j] = c->cfg.law_spline[i][j];
      }
}

static bool has_yukawa(const ngravs_ctx *c)
{
  for(int i = 0; i < c->cfg.n_gravs; i++)
    for(int j = 0; j < c->cfg.n_gravs; j++)
      if(c->cfg.law_accel[i][j] == NGRAVS_LAW_YUKAWA || c->cfg.law_accel[i][j] == NGRAVS_LAW_COLOYUK)
        return true;
  return false;
}

static TreeView tree_view(ngravs_ctx *c)
{
  TreeView tv;
  tv.first = c->n_first.p;
  tv.count = c->n_count.p;
  tv.child = c->n_child.p;
  tv.flags = c->n_flags.p;
  tv.geo = c->n_geo.p;
  tv.mom = c->n_mom.p;
  tv.npart = cfg_has_bam(c->cfg) ? c->n_npart.p : nullptr;
  tv.nnodes = (int)c->nnodes;
  tv.ltab = c->lvl_table.p;
  tv.ltab_level = c->lvl_table_level;
  tv.ltab_cl = c->lvl_table_level ? c->dom[6] / (double)(1 << c->lvl_table_level) : 0.0;
  for(int j = 0; j < 3; j++)
    tv.ltab_corner[j] = c->dom[j];
  return tv;
}

// ---- start table: the TreePM group walk only ever uses nodes within the short-range reach of a group's box, so instead of
// re-descending the same top levels for every group (a third of the traversal's batches) it starts from the cells of one
// tree level that overlap that region.  Valid when the level is complete (all 8^L cells are nodes: then no particle hangs
// directly below a shallower node) -- the deepest such level whose cells are still at least two reaches wide.
__global__ void k_level_table(const double4 *__restrict__ geo, int node0, int cnt, double cx, double cy, double cz, double inv_cl,
                              int nc, int *__restrict__ tab)
{
  int k = blockIdx.x * blockDim.x + threadIdx.x;
  if(k >= cnt)
    return;
  const double4 g = geo[node0 + k];
  int ix = (int)((g.x - cx) * inv_cl), iy = (int)((g.y - cy) * inv_cl), iz = (int)((g.z - cz) * inv_cl);
  ix = ix < 0 ? 0 : (ix >= nc ? nc - 1 : ix);
  iy = iy < 0 ? 0 : (iy >= nc ? nc - 1 : iy);
  iz = iz < 0 ? 0 : (iz >= nc ? nc - 1 : iz);
  tab[((size_t)ix * nc + iy) * nc + iz] = node0 + k;
}

static int ensure_level_table(ngravs_ctx *c, double reach)
{
  c->lvl_table_level = 0;
  // a refit (drifted) tree keeps its cell centres but particles may have left their cells (the sides grow to enclose
  // them): the regular cell grid no longer tells which nodes reach into a region, so such walks start at the root
  if(!c->cfg.pmgrid || !c->cfg.periodic || c->tree_refit || c->tune.walk_root)
    return NGRAVS_OK;
  int best = 0;
  for(int l = 2; l <= 6 && l < c->nlevels; l++)
    {
      const long long have = c->level_start[l + 1] - c->level_start[l];
      if(have != (1ll << (3 * l)))
        break;
      if(c->dom[6] / (double)(1 << l) >= 2.0 * reach)
        best = l;
    }
  if(!best)
    return NGRAVS_OK;
  const int nc = 1 << best;
  const long long cnt = 1ll << (3 * best);
  if(c->lvl_table.ensure((size_t)cnt))
    return NGRAVS_ERR_NOMEM;
  hipLaunchKernelGGL(k_level_table, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, c->stream, c->n_geo.p,
                     (int)c->level_start[best], (int)cnt, c->dom[0], c->dom[1], c->dom[2], (double)nc / c->dom[6], nc,
                     c->lvl_table.p);
  HIP_TRY(c, hipGetLastError());
  c->lvl_table_level = best;
  return NGRAVS_OK;
}

// lattice_init: one Ewald / lattice sum per distinct law, replicated into the [target][source] slots
static int ensure_lattice(ngravs_ctx *c)
{
  if(c->lat_ready)
    return NGRAVS_OK;
  const int ng = c->cfg.n_gravs;
  if(c->lat.ensure((size_t)ng * ng * LAT_SZ))
    return NGRAVS_ERR_NOMEM;
  const double L2 = c->cfg.box_size * c->cfg.box_size;
  const int npts = LAT_E1 * LAT_E1 * LAT_E1;
  for(int a = 0; a < ng; a++)
    for(int b = 0; b < ng; b++)
      {
        const int law = c->cfg.law_accel[a][b];
        int src = -1;
        for(int k = 0; k < a * ng + b; k++)
          if(c->cfg.law_accel[k / ng][k % ng] == law)
            {
              src = k;
              break;
            }
        double *dst = c->lat.p + (size_t)(a * ng + b) * LAT_SZ;
        if(src >= 0)
          HIP_TRY(c, hipMemcpyAsync(dst, c->lat.p + (size_t)src * LAT_SZ, sizeof(double) * LAT_SZ, hipMemcpyDeviceToDevice, c->stream));
        else
          hipLaunchKernelGGL(k_lattice_table, dim3((npts + 63) / 64), dim3(64), 0, c->stream, law, c->cfg.yukawa_imass, L2, dst);
      }
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  HIP_TRY(c, hipGetLastError());
  c->lat_ready = true;
  return NGRAVS_OK;
}

template <int NG, bool PM, bool LATT> static void launch_strict(ngravs_ctx *c, const WalkParams &wp, const LawIds &li)
{
  long long ngroups = (c->shard_count + WAVE - 1) / WAVE;
  unsigned nb = (unsigned)((ngroups + 3) / 4);
  hipLaunchKernelGGL((k_walk_strict<NG, PM, LATT>), dim3(nb), dim3(256), 0, c->stream, tree_view(c), c->s_pm.p, c->s_type.p,
                     c->s_oldacc.p, c->s_active.p, LATT ? c->lat.p : c->table.p, wp, li, (long long)c->shard_first,
                     (long long)c->shard_count, c->r_acc.p, c->r_nint.p, c->walk_counters.p + 1);
}

// targets of the group walk: all particles of the shard, or the compacted active ones (walk_select_targets)
static inline long long walk_tcount(const ngravs_ctx *c) { return c->walk_ntargets >= 0 ? c->walk_ntargets : c->shard_count; }
static inline const int *walk_tlist(const ngravs_ctx *c) { return c->walk_ntargets >= 0 ? c->walk_tlist.p : nullptr; }

struct ActiveFlag
{
  __host__ __device__ __forceinline__ bool operator()(const unsigned char &a) const { return (a & 1) != 0; }
};

// individual timesteps (gravtree.c:113: only particles with Ti_endstep == Ti_Current are walked): when the caller marked
// less than 3/4 of the shard active, the active indices are compacted (Peano order is kept) and the walk groups those
static int walk_select_targets(ngravs_ctx *c)
{
  c->walk_ntargets = -1;
  c->walk_dense_tlist = false;
  // a multi-task working set interleaves the own rows with imported copies (sources only): walking stretches of 64 ROWS would
  // leave a lane idle for every imported row (30-40 % at 8 tasks), so the own active rows are always compacted
  const bool imports = c->n_local != c->n;
  if((c->all_active && !imports) || c->shard_count <= 0 || !c->tune.walk_compact)
    return NGRAVS_OK;
  const int n = (int)c->shard_count;
  if(c->walk_tlist.ensure((size_t)n) || c->walk_counters.ensure(32))
    return NGRAVS_ERR_NOMEM;
  hipcub::CountingInputIterator<int> idx((int)c->shard_first);
  hipcub::TransformInputIterator<bool, ActiveFlag, const unsigned char *> flags(c->s_active.p + c->shard_first, ActiveFlag());
  size_t bytes = 0;
  HIP_TRY(c, hipcub::DeviceSelect::Flagged(nullptr, bytes, idx, flags, c->walk_tlist.p, c->walk_counters.p + 24, n, c->stream));
  if(c->walk_tmp.ensure(bytes))
    return NGRAVS_ERR_NOMEM;
  HIP_TRY(c, hipcub::DeviceSelect::Flagged(c->walk_tmp.p, bytes, idx, flags, c->walk_tlist.p, c->walk_counters.p + 24, n, c->stream));
  int cnt = 0;
  HIP_TRY(c, hipMemcpyAsync(&cnt, c->walk_counters.p + 24, sizeof(int), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  if(4ll * cnt < 3ll * n || imports)
    c->walk_ntargets = cnt;
  // most own rows active: the list is the Peano order of the own particles with the imported rows taken out -- as contiguous as
  // a single task's targets, so traversal units of four groups pay off as they do there
  c->walk_dense_tlist = imports && 4ll * cnt >= 3ll * c->n_local;
  return NGRAVS_OK;
}

template <int NG, bool PM, bool YUK, bool TAB_LDS, bool LATT>
static int launch_group2_t(ngravs_ctx *c, const WalkParams &wp, int *glist = nullptr, int nlist = 0)
{
  // lanes per target: the walk's own S, or for the leftover pass (glist) enough to give the few scattered groups many waves
  const int S0 = c->walk_spread > 1 ? c->walk_spread : 1, G0 = (WAVE / S0) * (glist ? c->walk_sg : 1);
  int S = S0;
  if(glist)
    {
      const int want = nlist <= 4096 ? 64 : (nlist <= 65536 ? 8 : 1);
      S = want > S0 ? want : S0;
    }
  const int G = WAVE / S;
  int ncu = 256;
  hipDeviceProp_t prop;
  if(hipGetDeviceProperties(&prop, c->cfg.device) == hipSuccess && prop.multiProcessorCount > 0)
    ncu = prop.multiProcessorCount;
  const size_t fixed = ((PM && TAB_LDS) ? sizeof(double) * (wp.ntab_lds + wp.exp_tab) * NTAB : 0) + 40 * sizeof(double);
  // one persistent workgroup per CU with as many waves as fit beside the tables (or several smaller ones)
  int waves = (int)((160 * 1024 - fixed) / GW2_WAVE_LDS);
  int per_cu = 1;
  if(waves > GW_MAXWAVES)
    waves = GW_MAXWAVES;   // register-limited: 3 waves per SIMD (__launch_bounds__), one workgroup per CU
  if(waves < 1)
    waves = 1;
  size_t lds = fixed + (size_t)waves * GW2_WAVE_LDS;
  long long ngroups = glist ? (long long)nlist * (G0 / G) : (walk_tcount(c) + G - 1) / G;
  long long nblk = (long long)ncu * per_cu;
  if(nblk > (ngroups + waves - 1) / waves)
    nblk = (ngroups + waves - 1) / waves;
  if(nblk < 1)
    nblk = 1;
  if(c->walk_stack.ensure((size_t)nblk * waves * (GW_STACK + (size_t)NG * GW2_ITEMS)) || c->walk_counters.ensure(32))
    return NGRAVS_ERR_NOMEM;
  if(glist)   // second pass of a split walk: keep its statistics, restart the per-XCD group counters
    HIP_TRY(c, hipMemsetAsync(c->walk_counters.p + 8, 0, sizeof(int) * 8, c->stream));
  else
    HIP_TRY(c, hipMemsetAsync(c->walk_counters.p, 0, sizeof(int) * 32, c->stream));
  auto kern = k_walk_group2<NG, PM, YUK, TAB_LDS, LATT, 0>;
  HIP_TRY(c, hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(kern, dim3((unsigned)nblk), dim3(waves * 64), lds, c->stream, tree_view(c), c->s_pm.p,
                     c->s_type.p, c->s_oldacc.p, c->s_active.p, LATT ? c->lat.p : c->table.p, wp, (long long)c->shard_first,
                     walk_tcount(c), c->walk_counters.p, c->walk_stack.p, c->walk_counters.p + 1, c->r_acc.p,
                     c->r_nint.p, (int *)nullptr, (int *)nullptr, 0ll, glist ? (long long)nlist : 0ll, 0, 0, glist, S, G0, walk_tlist(c), 1);
  return NGRAVS_OK;
}

// split walk: per batch of groups a high-occupancy traversal kernel (MODE 1) writes the item lists, then the persistent
// evaluation kernel (MODE 2) consumes them.  Same results as the fused kernel (same lists, same order).
template <int NG, bool PM, bool YUK, bool TAB_LDS, bool LATT> static int launch_group3_t(ngravs_ctx *c, const WalkParams &wp)
{
  int ncu = 256;
  hipDeviceProp_t prop;
  if(hipGetDeviceProperties(&prop, c->cfg.device) == hipSuccess && prop.multiProcessorCount > 0)
    ncu = prop.multiProcessorCount;
  const size_t fixed = ((PM && TAB_LDS) ? sizeof(double) * (wp.ntab_lds + wp.exp_tab) * NTAB : 0) + 40 * sizeof(double);
  int waves = (int)((160 * 1024 - fixed) / GW2_WAVE_LDS);
  if(waves > GW2_MAXWAVES)
    waves = GW2_MAXWAVES;   // register-limited: 4 waves per SIMD (__launch_bounds__), one workgroup per CU
  if(c->tune.walk_waves > 0 && c->tune.walk_waves < waves)
    waves = c->tune.walk_waves;   // tuning: fewer waves per evaluation workgroup
  if(waves < 1)
    waves = 1;
  const size_t lds = fixed + (size_t)waves * GW2_WAVE_LDS;
  const int S = c->walk_spread > 1 ? c->walk_spread : 1, G = WAVE / S;
  const long long ngroups = (walk_tcount(c) + G - 1) / G;
  // groups per traversal unit: 4 for TreePM walks of Peano-contiguous targets (the short-range region of 256 neighbours is
  // 1.7 x that of 64: lists and traversal work per target fall to ~0.4), 1 for tree-only walks (no cut: a wider box opens more
  // of the tree for every target) and for scattered (compacted / spread) targets
  int SG = (PM && (c->walk_ntargets < 0 || c->walk_dense_tlist)) ? 4 : 1;
  // ... in a CLUSTERED set the last walk says so: it evaluated more pairs per target than a uniform box of the same mean density
  // holds in the cut sphere (every source inside the sphere a particle: the uniform regime).  A smaller unit has a smaller box,
  // accepts more cells as monopoles, and wins back more than its longer lists cost (2^20 particles, 60 % of them in one clump:
  // 1857 / 1393 / 1079 pairs per target and 8.0 / 7.2 / 5.2 ms for units of 4 / 2 / 1 groups)
  // (walk_ia_ratio is normalised to units of four groups; the unit only changes when the ratio leaves a band around the threshold, so
  // that a set near one does not flip from step to step: down at 1.4 / 2.0, up again below 1.2 / 1.7)
  if(SG == 4)
    {
      const double r = c->walk_ia_ratio;
      int st = c->walk_unit_state;
      if(r > 0)
        {
          if(st == 4)
            st = r > 2.0 ? 1 : (r > 1.4 ? 2 : 4);
          else if(st == 2)
            st = r > 2.0 ? 1 : (r < 1.2 ? 4 : 2);
          else
            st = r < 1.2 ? 4 : (r < 1.7 ? 2 : 1);
        }
      c->walk_unit_state = st;
      SG = st;
    }
  if(c->tune.walk_sg >= 1)
    SG = c->tune.walk_sg;
  if(SG != c->walk_sg)
    c->walk_lcap = 0;   // list capacities are per unit
  c->walk_sg = SG;
  const long long nunits = (ngroups + SG - 1) / SG;
  // per-unit region: NG item lists of lcap ints + the LIFO.  lcap starts small and is doubled (persistently) by walk_run
  // when a list overflows; the batch shrinks so that the scratch stays within ~8 GB
  if(c->walk_lcap < 1024)
    {
      c->walk_lcap = (NG == 1 ? 2 : 1) * (SG > 1 ? 2 : 1) * GW3_LIST_MIN;
      if(c->tune.walk_lcap >= 1024)   // tests: small lists force the leftover pass
        c->walk_lcap = c->tune.walk_lcap;
    }
  if(c->walk_scap < GW3_STK_MIN)
    c->walk_scap = GW3_STK_MIN;
  c->walk_lcap &= ~3;   // the evaluation kernel reads the lists in 16-byte quads
  const int lcap = c->walk_lcap, scap = c->walk_scap;
  const size_t region_ints = (size_t)NG * lcap + scap;
  // batches as large as memory comfortably allows (every traversal/evaluation launch pair costs ~0.4 ms of ramp and tail:
  // 16 / 8 / 1 launches per step at C4 measured 114 / 107 / 104 ms of evaluation): a quarter of the free device memory, at
  // least 8 GB and at most 64 GB, unless NGRAVS_WALK_BATCH fixes the group count
  long long batch = 1 << 20;   // units per launch pair
  size_t cap_bytes = (size_t)8 << 30;
  {
    size_t free_b = 0, total_b = 0;
    if(hipMemGetInfo(&free_b, &total_b) == hipSuccess)
      {
        free_b += c->walk_stack.cap * sizeof(int);   // what this scratch already holds counts as available
        if(free_b / 4 > cap_bytes)
          cap_bytes = free_b / 4;
        if(cap_bytes > ((size_t)64 << 30))
          cap_bytes = (size_t)64 << 30;
      }
  }
  if(c->tune.walk_batch > 0)
    {
      batch = c->tune.walk_batch;
      cap_bytes = (size_t)64 << 30;
    }
  while(batch > 8192 && (size_t)batch * region_ints * sizeof(int) > cap_bytes)
    batch /= 2;
  if(batch > nunits)
    batch = nunits;
  if(batch < 1)
    batch = 1;
  if(c->walk_stack.ensure((size_t)batch * (region_ints + NG + 2)) || c->walk_counters.ensure(32) ||
     c->walk_ovf.ensure((size_t)nunits))
    return NGRAVS_ERR_NOMEM;
  int *region = c->walk_stack.p, *gcount = c->walk_stack.p + (size_t)batch * region_ints;
  HIP_TRY(c, hipMemsetAsync(c->walk_counters.p, 0, sizeof(int) * 32, c->stream));
  auto kt = k_walk_group2<NG, PM, YUK, TAB_LDS, LATT, 1>;
  auto ke = k_walk_group2<NG, PM, YUK, TAB_LDS, LATT, 2>;
  HIP_TRY(c, hipFuncSetAttribute(reinterpret_cast<const void *>(ke), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  // TreePM lists with the tables in LDS go through the ring-pool evaluation kernel when at least 4 slots per wave fit
  int ringK = 0, ring_waves = waves;
  if constexpr(PM && TAB_LDS && !LATT)
    if(c->tune.walk_ring)
      {
        ring_waves = GW2_MAXWAVES;
        if(c->tune.walk_waves > 0 && c->tune.walk_waves < ring_waves)
          ring_waves = c->tune.walk_waves;
        ringK = eval_ring_slots(wp, YUK, ring_waves);
        if(c->tune.walk_ring_k >= 4 && c->tune.walk_ring_k < ringK)
          ringK = c->tune.walk_ring_k;
        if(ringK)
          waves = ring_waves;
      }
  const size_t nbatches = (size_t)((nunits + batch - 1) / batch);
  while(c->ev_batch.size() < 3 * nbatches)
    {
      hipEvent_t e;
      HIP_TRY(c, hipEventCreate(&e));
      c->ev_batch.push_back(e);
    }
  c->walk_batches = (int)nbatches;
  size_t ib = 0;
  for(long long u0 = 0; u0 < nunits; u0 += batch, ib++)
    {
      const long long nbu = nunits - u0 < batch ? nunits - u0 : batch;   // units of this batch
      const long long g0 = u0 * SG, nb = ngroups - g0 < nbu * SG ? ngroups - g0 : nbu * SG;   // its groups
      HIP_TRY(c, hipEventRecord(c->ev_batch[3 * ib], c->stream));
      const long long tblk = 8 * (((nbu + GW3_TBLOCK / 64 - 1) / (GW3_TBLOCK / 64) + 7) / 8);
      hipLaunchKernelGGL(kt, dim3((unsigned)tblk), dim3(GW3_TBLOCK), 0, c->stream, tree_view(c), c->s_pm.p, c->s_type.p,
                         c->s_oldacc.p, c->s_active.p, LATT ? c->lat.p : c->table.p, wp, (long long)c->shard_first,
                         walk_tcount(c), c->walk_counters.p, (int *)nullptr, c->walk_counters.p + 1, c->r_acc.p,
                         c->r_nint.p, region, gcount, u0, nbu, lcap, scap, c->walk_ovf.p, S, G, walk_tlist(c), SG);
      HIP_TRY(c, hipEventRecord(c->ev_batch[3 * ib + 1], c->stream));
      if(g0 > 0)
        HIP_TRY(c, hipMemsetAsync(c->walk_counters.p + 8, 0, sizeof(int) * 8, c->stream));
      long long nblk = ncu;
      if(nblk > (nb + waves - 1) / waves)
        nblk = (nb + waves - 1) / waves;
      if(ringK)
        {
          // the evaluation kernel with the ring pool (kernels_eval.hip): same lists, same pairs, per-lane cursors
          int rr = launch_eval_ring(c, tree_view(c), wp, YUK, (int)nblk, ring_waves, ringK, region, gcount, g0, nb, lcap, scap, S,
                                    walk_tlist(c), SG, walk_tcount(c));
          if(rr != NGRAVS_OK)
            return rr;
        }
      else
      hipLaunchKernelGGL(ke, dim3((unsigned)nblk), dim3(waves * 64), lds, c->stream, tree_view(c), c->s_pm.p, c->s_type.p,
                         c->s_oldacc.p, c->s_active.p, LATT ? c->lat.p : c->table.p, wp, (long long)c->shard_first,
                         walk_tcount(c), c->walk_counters.p, (int *)nullptr, c->walk_counters.p + 1, c->r_acc.p,
                         c->r_nint.p, region, gcount, g0, nb, lcap, scap, c->walk_ovf.p, S, G, walk_tlist(c), SG);
      HIP_TRY(c, hipEventRecord(c->ev_batch[3 * ib + 2], c->stream));
    }
  return NGRAVS_OK;
}

template <int NG>
static int launch_group(ngravs_ctx *c, const WalkParams &wp, bool allow_split, bool *used_split, int *glist = nullptr, int nlist = 0)
{
  *used_split = false;
  if(!glist)
    c->walk_batches = 0;
  const bool pm = c->cfg.pmgrid != 0, yuk = has_yukawa(c);
  // tables in LDS while they leave room for the pools of 16 waves (three 16 KB tables incl. the exp table; the C5 wiring --
  // Newton on the diagonal, one law off it -- has two distinct ones); otherwise they are read through L1/L2
  constexpr bool TL = (NG <= 2);
  if constexpr(NG == 3)
    if(pm && wp.ntab_lds + wp.exp_tab <= 3)
      {
        if(allow_split && !c->tune.walk_fused && !glist)
          {
            *used_split = true;
            return has_yukawa(c) ? launch_group3_t<NG, true, true, true, false>(c, wp) : launch_group3_t<NG, true, false, true, false>(c, wp);
          }
        return has_yukawa(c) ? launch_group2_t<NG, true, true, true, false>(c, wp, glist, nlist)
                             : launch_group2_t<NG, true, false, true, false>(c, wp, glist, nlist);
      }
  const bool v2 = c->tune.walk_fused != 0;   // fused kernel for the whole walk
  if(allow_split && !v2 && !glist)
    {
      *used_split = true;
      if(pm)
        return yuk ? launch_group3_t<NG, true, true, TL, false>(c, wp) : launch_group3_t<NG, true, false, TL, false>(c, wp);
      if(c->cfg.periodic)
        return yuk ? launch_group3_t<NG, false, true, false, true>(c, wp) : launch_group3_t<NG, false, false, false, true>(c, wp);
      if(!yuk && wp.bam)
        return launch_group3_t<NG, false, false, true, false>(c, wp);   // the variant with the BAM laws
      return yuk ? launch_group3_t<NG, false, true, false, false>(c, wp) : launch_group3_t<NG, false, false, false, false>(c, wp);
    }
  if(pm)
    return yuk ? launch_group2_t<NG, true, true, TL, false>(c, wp, glist, nlist)
               : launch_group2_t<NG, true, false, TL, false>(c, wp, glist, nlist);
  if(c->cfg.periodic)
    return yuk ? launch_group2_t<NG, false, true, false, true>(c, wp, glist, nlist)
               : launch_group2_t<NG, false, false, false, true>(c, wp, glist, nlist);
  if(!yuk && wp.bam)
    return launch_group2_t<NG, false, false, true, false>(c, wp, glist, nlist);
  return yuk ? launch_group2_t<NG, false, true, false, false>(c, wp, glist, nlist)
             : launch_group2_t<NG, false, false, false, false>(c, wp, glist, nlist);
}

int walk_run(ngravs_ctx *c)
{
  const long long n = c->n;
  if(c->r_acc.ensure(3 * n) || c->r_nint.ensure(n) || c->r_oldacc.ensure(n))
    return NGRAVS_ERR_NOMEM;
  const bool latt = c->cfg.periodic && !c->cfg.pmgrid;
  if(latt)
    {
      int rcl = ensure_lattice(c);   // begrun.c:47-49: lattice_init() if PERIODIC && !PMGRID
      if(rcl)
        return rcl;
    }
  WalkParams wp;
  make_walk_params(c, &wp);
  LawIds li;
  make_law_ids(c, &li);
  c->walk_ntargets = -1;
  const bool strict = c->cfg.walk_mode == NGRAVS_WALK_STRICT;   // (the BAM laws run in the group walk too: wp.bam)
  if(c->top.on && c->cfg.pmgrid)
    {
      // multi-task tree: the leaves this task imported were chosen for the reach of the walk mode in force at the decomposition
      const double need = strict ? 6.0 : fmin(6.0, c->cfg.group_reach > 0 ? c->cfg.group_reach : NGRAVS_GROUP_REACH);
      if(need > c->top.import_reach * (1.0 + 1e-12))
        {
          ngravs_report(c, NGRAVS_ERR_STATE, "the walk reaches farther than the multi-task decomposition imported for: set the walk mode "
                                             "(and group_reach) before the decomposition");
          return NGRAVS_ERR_STATE;
        }
    }
  if(!strict)
    {
      int rct = ensure_level_table(c, sqrt(wp.reach2));
      if(rct)
        return rct;
      if((rct = walk_select_targets(c)))
        return rct;
    }
  // rows the walk does not write (inactive particles, other tasks' shards, halo copies) read as zero
  if(!(c->all_active && c->cfg.world_size == 1 && c->n_local == c->n))
    {
      HIP_TRY(c, hipMemsetAsync(c->r_nint.p, 0, sizeof(int) * n, c->stream));
      HIP_TRY(c, hipMemsetAsync(c->r_acc.p, 0, sizeof(double) * 3 * n, c->stream));
      // ... and keep their OldAcc: k_finish only rewrites the rows of walked particles (gravtree.c:318-331)
      HIP_TRY(c, hipMemcpyAsync(c->r_oldacc.p, c->s_oldacc.p, sizeof(double) * n, hipMemcpyDeviceToDevice, c->stream));
    }
  const bool pm = c->cfg.pmgrid != 0;
  HIP_TRY(c, hipEventRecord(c->evk0, c->stream));
  int rc = NGRAVS_OK;
  bool used_split = false;
  auto group_launch = [&](bool allow_split, int *glist, int nlist) -> int {
    bool us = false;
    int r;
    switch(c->cfg.n_gravs)
      {
      case 1:
        r = launch_group<1>(c, wp, allow_split, &us, glist, nlist);
        break;
      case 2:
        r = launch_group<2>(c, wp, allow_split, &us, glist, nlist);
        break;
      default:
        r = launch_group<3>(c, wp, allow_split, &us, glist, nlist);
        break;
      }
    if(!glist)
      used_split = us;
    return r;
  };
  auto strict_launch = [&]() {
    switch(c->cfg.n_gravs)
      {
      case 1:
        pm ? launch_strict<1, true, false>(c, wp, li) : (latt ? launch_strict<1, false, true>(c, wp, li) : launch_strict<1, false, false>(c, wp, li));
        break;
      case 2:
        pm ? launch_strict<2, true, false>(c, wp, li) : (latt ? launch_strict<2, false, true>(c, wp, li) : launch_strict<2, false, false>(c, wp, li));
        break;
      default:
        pm ? launch_strict<3, true, false>(c, wp, li) : (latt ? launch_strict<3, false, true>(c, wp, li) : launch_strict<3, false, false>(c, wp, li));
        break;
      }
  };
  if(strict)
    {
      if(c->walk_counters.ensure(32))
        return NGRAVS_ERR_NOMEM;
      HIP_TRY(c, hipMemsetAsync(c->walk_counters.p, 0, sizeof(int) * 32, c->stream));
      strict_launch();
    }
  else
    {
      // sparse active sets: 64 compacted targets span a box much wider than the short-range reach, so the conservative
      // group tests would collect (and then mostly discard) huge lists; walk them in sub-groups of 64/S targets, S lanes
      // per target (k_walk_group2).
      c->walk_spread = 0;
      if(c->walk_ntargets >= 0 && !c->walk_dense_tlist && pm && c->cfg.box_size > 0)
        {
          const double vol_per_target = pow(c->cfg.box_size, 3) * (double)c->shard_count / ((double)c->n * (double)(c->walk_ntargets > 0 ? c->walk_ntargets : 1));
          // measured (16 M particles, 10 % / 1 % / 0.1 % active): S = 1 wins while 64 targets span less than ~2.5 reaches,
          // beyond that one target per wave does (intermediate S multiply the traversals without shrinking the lists enough)
          const int sp = cbrt(vol_per_target * 64.0) > 2.5 * sqrt(wp.reach2) ? 64 : 1;
          c->walk_spread = sp;
          if(c->tune.walk_spread >= 1)   // tests / tuning: 1, 2, 4 ... 64 lanes per target
            c->walk_spread = c->tune.walk_spread;
          if(c->walk_spread <= 1)
            c->walk_spread = 0;
        }
      else if(c->tune.walk_spread > 1)
        c->walk_spread = c->tune.walk_spread;   // dense target sets: sub-groups of 64/S targets on request (a smaller box accepts more monopoles)
      else if(c->tune.walk_spread == 0)
        {
          // Measured (tools/spread_sweep.sh, profiles/r04_spread_sweep.json): two lanes per target pay in a strongly clustered TreePM
          // set once its units are single groups (2^20 particles, 60 % in one clump: 1079 -> 871 pairs per target, 5.15 -> 4.67 ms);
          // a tree-only set with too few groups to fill the device (GalaxyCollision.IC: 938 groups of 64 for 4096 waves) walks faster
          // with 4 lanes per target (4.2 -> 1.9 ms); a large one does not (4 M Plummer sphere: 12.5 / 14.8 / 19.5 ms for S = 1 / 2 / 4).
          const long long g64 = (walk_tcount(c) + WAVE - 1) / WAVE;
          if(pm && c->walk_unit_state == 1 && c->walk_ia_ratio > 2.0)
            c->walk_spread = 2;
          else if(!pm && !latt && g64 > 0 && g64 < 2048)
            c->walk_spread = g64 < 1024 ? 4 : 2;
        }
      rc = group_launch(true, nullptr, 0);
    }
  if(rc != NGRAVS_OK)
    return rc;
  HIP_TRY(c, hipEventRecord(c->evk1, c->stream));
  HIP_TRY(c, hipGetLastError());
  c->walk_unopened = 0;
  if(strict && c->top.on)
    {
      int flag = 0;
      HIP_TRY(c, hipMemcpyAsync(&flag, c->walk_counters.p + 1, sizeof(int), hipMemcpyDeviceToHost, c->stream));
      HIP_TRY(c, hipStreamSynchronize(c->stream));
      if(flag & 4)
        {
          ngravs_report(c, NGRAVS_ERR_STATE, "the walk opens a top-tree leaf whose particles were not imported: the opening criterion or OldAcc "
                                             "changed since the decomposition (decompose again before this walk)");
          return NGRAVS_ERR_STATE;
        }
    }
  if(!strict)
    {
      int flag = 0;
      unsigned long long st64[4] = {0, 0, 0, 0};
      HIP_TRY(c, hipMemcpyAsync(&flag, c->walk_counters.p + 1, sizeof(int), hipMemcpyDeviceToHost, c->stream));
      HIP_TRY(c, hipMemcpyAsync(st64, c->walk_counters.p + 16, 4 * sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream));
      HIP_TRY(c, hipStreamSynchronize(c->stream));
      if(used_split)
        {
          // groups whose item lists or LIFO outgrew their region of the split walk were skipped by the evaluation kernel.
          // They are the sparse outskirts, where 64 Peano-contiguous targets span a box so large that the conservative
          // group tests open most of the tree: exactly those groups are redone by the fused kernel in sub-groups (one
          // target per wave if they are few: the box is then the target itself and every test the reference's own).
          // If they are many the regions grow for the following steps.
          int ovf[2] = {0, 0};
          HIP_TRY(c, hipMemcpy(ovf, c->walk_counters.p + 2, 2 * sizeof(int), hipMemcpyDeviceToHost));
          if(ovf[0] > 0)
            {
              rc = group_launch(false, c->walk_ovf.p, ovf[0]);
              if(rc != NGRAVS_OK)
                return rc;
              HIP_TRY(c, hipEventRecord(c->evk1, c->stream));
              HIP_TRY(c, hipGetLastError());
              HIP_TRY(c, hipMemcpyAsync(&flag, c->walk_counters.p + 1, sizeof(int), hipMemcpyDeviceToHost, c->stream));
              HIP_TRY(c, hipMemcpyAsync(st64, c->walk_counters.p + 16, 4 * sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream));
              HIP_TRY(c, hipStreamSynchronize(c->stream));
              const int Gs = (WAVE / (c->walk_spread > 1 ? c->walk_spread : 1)) * c->walk_sg;
              const long long ng = (walk_tcount(c) + Gs - 1) / Gs;
              if((long long)ovf[0] * 256 > ng)
                {
                  if(2 * ovf[1] > ovf[0])
                    c->walk_scap = c->walk_scap < GW3_STK_MAX ? 2 * c->walk_scap : c->walk_scap;
                  else
                    c->walk_lcap = c->walk_lcap < GW3_LIST_MAX ? 2 * c->walk_lcap : c->walk_lcap;
                }
            }
        }
      const int Gn = WAVE / (c->walk_spread > 1 ? c->walk_spread : 1);
      double ngroups = (double)((walk_tcount(c) + Gn - 1) / Gn);
      c->stats.reserved[0] = st64[0] / ngroups;   // pool entries per group
      c->stats.reserved[1] = st64[1] / ngroups;   // nodes tested per group
      c->stats.reserved[2] = st64[2] / ngroups;   // traversal batches per group
      c->stats.reserved[3] = st64[3] / ngroups;   // force-loop slots per lane and group
      // split walk: time of the evaluation / traversal kernels summed over the batches, and the batch count
      c->stats.reserved[4] = c->stats.reserved[5] = c->stats.reserved[6] = 0;
      for(int b = 0; b < c->walk_batches; b++)
        {
          float te = 0, tt = 0;
          (void)hipEventElapsedTime(&tt, c->ev_batch[3 * b], c->ev_batch[3 * b + 1]);
          (void)hipEventElapsedTime(&te, c->ev_batch[3 * b + 1], c->ev_batch[3 * b + 2]);
          c->stats.reserved[4] += te;
          c->stats.reserved[6] += tt;
        }
      c->stats.reserved[5] = c->walk_batches;
      {
        int unop = 0;
        HIP_TRY(c, hipMemcpy(&unop, c->walk_counters.p + 4, sizeof(int), hipMemcpyDeviceToHost));
        c->walk_unopened = unop;
      }
      if(flag)
        {
          ngravs_report(c, NGRAVS_ERR_TREE, "group walk: pending-node LIFO overflow");
          return NGRAVS_ERR_TREE;
        }
    }
  return NGRAVS_OK;
}

int walk_finish(ngravs_ctx *c)
{
  const int bs = 256;
  unsigned nb = (unsigned)((c->shard_count + bs - 1) / bs);
  nb = nb > 4096u ? 4096u : nb;   // grid-stride: see k_finish
  if(c->red_tmp.ensure(2))
    return NGRAVS_ERR_NOMEM;
  HIP_TRY(c, hipMemsetAsync(c->red_tmp.p, 0, 2 * sizeof(double), c->stream));   // read back by ngravs_gravity_tree
  if(nb == 0)
    return NGRAVS_OK;
  hipLaunchKernelGGL(k_finish, dim3(nb), dim3(bs), 0, c->stream, (long long)c->shard_first, (long long)c->shard_count,
                     c->s_active.p, c->r_acc.p, c->r_pm.p, c->r_oldacc.p, c->cfg.G, (c->have_pm && c->cfg.pmgrid) ? 1 : 0,
                     c->r_nint.p, c->red_tmp.p);
  HIP_TRY(c, hipGetLastError());
  return NGRAVS_OK;
}

// direct sum for explicit target records over the OWN particles of this task (distributed gravity_forcetest)
int direct_run_targets(ngravs_ctx *c, const double4 *d_tpm, const int *d_ttype, int64_t nt, double *d_acc)
{
  WalkParams wp;
  make_walk_params(c, &wp);
  LawIds li;
  make_law_ids(c, &li);
  const bool latt = c->cfg.periodic != 0;
  if(latt)
    {
      int rcl = ensure_lattice(c);
      if(rcl)
        return rcl;
    }
  hipLaunchKernelGGL(k_direct, dim3((unsigned)nt), dim3(256), 0, c->stream, c->s_pm.p, c->s_type.p, (long long)c->n, (const int *)nullptr,
                     (long long)nt, wp, li, c->cfg.G, latt ? c->lat.p : (const double *)nullptr, d_acc, d_tpm, d_ttype, c->s_active.p, 1);
  HIP_TRY(c, hipGetLastError());
  return NGRAVS_OK;
}

int direct_run(ngravs_ctx *c, const int *d_idx, int64_t nt, double *d_acc)
{
  WalkParams wp;
  make_walk_params(c, &wp);
  LawIds li;
  make_law_ids(c, &li);
  const bool latt = c->cfg.periodic != 0;   // forcetree.c:3515-3529: the PERIODIC direct sum adds lattice_corr
  if(latt)
    {
      int rcl = ensure_lattice(c);
      if(rcl)
        return rcl;
    }
  hipLaunchKernelGGL(k_direct, dim3((unsigned)nt), dim3(256), 0, c->stream, c->s_pm.p, c->s_type.p, (long long)c->n, d_idx,
                     (long long)nt, wp, li, c->cfg.G, latt ? c->lat.p : (const double *)nullptr, d_acc);
  HIP_TRY(c, hipGetLastError());
  return NGRAVS_OK;
}
