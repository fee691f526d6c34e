/* ngravs_host.c -- multi-task drivers of the gravity path in plain C over a communicator vtable (include/ngravs_host.h).
 * Linked into libngravs_hip.so; uses nothing but the public C ABI of include/ngravs_hip.h and the caller's callbacks, so the
 * same code runs under MPI (host/gadget_glue.c), under RCCL through torch.distributed (distributed.py) and under the
 * shared-memory communicator of host/host_shim_test.c.
 */
#define _POSIX_C_SOURCE 199309L
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include "ngravs_host.h"

#define CHECK(expr)            \
  do                           \
    {                          \
      int rc__ = (expr);       \
      if(rc__ != 0)            \
        return rc__ < 0 ? rc__ : NGRAVS_ERR_STATE; \
    }                          \
  while(0)

/* Peano-Hilbert key of every cell (x, y, z) of level d, [x][y][z] order: depends on the level only, so it is computed once per
 * process (three loops over 8^level cells per step cost 3 ms each at level 5).  Lock-free publication: a second thread that
 * raced builds the same table and drops it. */
static int32_t *ph_tables[8];
static const int32_t *ph_table(int d)
{
  int32_t *t;
  int x, y, z, nc;
  if(d < 0 || d > 7)
    return NULL;
  t = __atomic_load_n(&ph_tables[d], __ATOMIC_ACQUIRE);
  if(t)
    return t;
  nc = 1 << d;
  t = malloc(sizeof(int32_t) * ((size_t)nc * nc * nc));
  if(!t)
    return NULL;
  for(x = 0; x < nc; x++)
    for(y = 0; y < nc; y++)
      for(z = 0; z < nc; z++)
        t[((size_t)x * nc + y) * nc + z] = (int32_t)ngravs_peano_hilbert_key(x, y, z, d);
  {
    int32_t *expected = NULL;
    if(!__atomic_compare_exchange_n(&ph_tables[d], &expected, t, 0, __ATOMIC_RELEASE, __ATOMIC_ACQUIRE))
      {
        free(t);
        t = expected;
      }
  }
  return t;
}

static double wall_now(void)
{
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

/* all-to-all-v of library device buffers; counts in bytes.  Without device-capable transport the blocks go through host memory. */
static int exchange(ngravs_ctx *ctx, const ngravs_comm *cm, const void *dsend, const int64_t *sbytes, void *drecv, const int64_t *rbytes)
{
  const int W = cm->size;
  int64_t *sd = malloc(sizeof(int64_t) * 2 * (size_t)(W + 1)), *rd = sd + W + 1;
  int r, rc = 0;
  if(!sd)
    return NGRAVS_ERR_NOMEM;
  sd[0] = rd[0] = 0;
  for(r = 0; r < W; r++)
    {
      sd[r + 1] = sd[r] + sbytes[r];
      rd[r + 1] = rd[r] + rbytes[r];
    }
  if(cm->device_buffers)
    rc = cm->alltoallv(cm->user, dsend, sbytes, sd, drecv, rbytes, rd);
  else
    {
      void *hs = malloc((size_t)(sd[W] > 0 ? sd[W] : 1)), *hr = malloc((size_t)(rd[W] > 0 ? rd[W] : 1));
      if(!hs || !hr)
        rc = NGRAVS_ERR_NOMEM;
      if(!rc)
        rc = ngravs_memcpy(ctx, hs, dsend, sd[W], 2);
      if(!rc)
        rc = cm->alltoallv(cm->user, hs, sbytes, sd, hr, rbytes, rd);
      if(!rc)
        rc = ngravs_memcpy(ctx, drecv, hr, rd[W], 1);
      free(hs);
      free(hr);
    }
  free(sd);
  return rc == 0 ? 0 : (rc < 0 ? rc : NGRAVS_ERR_STATE);
}

/* ---- the split of the Peano-cell sequence over the tasks --------------------------------------------------------------- */
typedef struct
{
  const int64_t *count;
  double maxload;
  int32_t *owner;
  int *start, *end;      /* first / last cell of every task */
  double *load;          /* particles of every task          */
} split_t;

/* domain_findSplit (domain.c:347-456): bisect the cells [first, last] among ncpu tasks so that the larger of the two average
 * particle loads is minimal, recursively; fails if a side would exceed maxload per task */
static int find_split(split_t *S, int cpustart, int ncpu, int first, int last)
{
  const int nleft = ncpu / 2, nright = ncpu - nleft;
  double load = 0, left = 0;
  int i, split = first + nleft;
  for(i = first; i <= last; i++)
    load += (double)S->count[i];
  for(i = first; i < split; i++)
    left += (double)S->count[i];
  while(split < last - (nright - 1) && split > 0)   /* every task keeps at least one cell */
    {
      const double cur = fmax(left / nleft, (load - left) / nright);
      const double nxt = fmax((left + (double)S->count[split]) / nleft, (load - left - (double)S->count[split]) / nright);
      if(nxt > cur)
        break;
      left += (double)S->count[split];
      split++;
    }
  if(left > S->maxload * nleft || load - left > S->maxload * nright)
    return -1;
  if(nleft >= 2 && find_split(S, cpustart, nleft, first, split - 1))
    return -1;
  if(nright >= 2 && find_split(S, cpustart + nleft, nright, split, last))
    return -1;
  if(nleft == 1)
    {
      for(i = first; i < split; i++)
        S->owner[i] = cpustart;
      S->load[cpustart] = left;
      S->start[cpustart] = first;
      S->end[cpustart] = split - 1;
    }
  if(nright == 1)
    {
      for(i = split; i <= last; i++)
        S->owner[i] = cpustart + nleft;
      S->load[cpustart + nleft] = load - left;
      S->start[cpustart + nleft] = split;
      S->end[cpustart + nleft] = last;
    }
  return 0;
}

int ngravs_host_split(const int64_t *count, const double *work, int64_t ncell, int ntask, double max_load, int32_t *owner)
{
  split_t S;
  double *wk;
  int t, moved, rc;
  int64_t iter = 0, i;
  if(!count || !owner || ntask < 1 || ncell < ntask)
    return -1;
  if(ntask == 1)
    {
      for(i = 0; i < ncell; i++)
        owner[i] = 0;
      return 0;
    }
  S.count = count;
  S.maxload = max_load > 0 ? max_load : 1e300;
  S.owner = owner;
  S.start = malloc(sizeof(int) * 2 * (size_t)ntask);
  S.end = S.start + ntask;
  S.load = malloc(sizeof(double) * 2 * (size_t)ntask);
  wk = S.load + ntask;
  rc = find_split(&S, 0, ntask, 0, (int)ncell - 1);
  if(rc == 0 && work)
    {
      /* domain_shiftSplit (domain.c:468-544): move boundary cells between neighbours while that LOWERS the larger of the
       * two work sums (strictly: a move that leaves it unchanged would be undone by the next sweep, for ever) and keeps the
       * particle load within bounds */
      for(t = 0; t < ntask; t++)
        wk[t] = 0;
      for(i = 0; i < ncell; i++)
        wk[owner[i]] += work[i];
      do
        {
          moved = 0;
          for(t = 0; t < ntask - 1; t++)
            {
              const double maxw = fmax(wk[t], wk[t + 1]);
              if(wk[t] < wk[t + 1])
                {
                  const int cell = S.start[t + 1];
                  if(S.end[t + 1] <= cell)   /* the neighbour keeps at least one cell */
                    continue;
                  if(fmax(wk[t] + work[cell], wk[t + 1] - work[cell]) < maxw && S.load[t] + (double)count[cell] <= S.maxload)
                    {
                      wk[t] += work[cell];
                      wk[t + 1] -= work[cell];
                      S.load[t] += (double)count[cell];
                      S.load[t + 1] -= (double)count[cell];
                      owner[cell] = t;
                      S.start[t + 1]++;
                      S.end[t]++;
                      moved++;
                    }
                }
              else
                {
                  const int cell = S.end[t];
                  if(S.start[t] >= cell)
                    continue;
                  if(fmax(wk[t] - work[cell], wk[t + 1] + work[cell]) < maxw && S.load[t + 1] + (double)count[cell] <= S.maxload)
                    {
                      wk[t] -= work[cell];
                      wk[t + 1] += work[cell];
                      S.load[t] -= (double)count[cell];
                      S.load[t + 1] += (double)count[cell];
                      owner[cell] = t + 1;
                      S.end[t]--;
                      S.start[t + 1]--;
                      moved++;
                    }
                }
            }
          iter++;
        }
      while(moved > 0 && iter < 10 * ncell);
    }
  free(S.start);
  free(S.load);
  return rc;
}

/* ---- domain_Decomposition ------------------------------------------------------------------------------------------------ */
static int choose_level(const ngravs_config_t *cfg)
{
  /* the coarsest cells still at least as wide as the short-range cut (the halo looks one cell layer around a task's cells);
   * the curve's cube is 1.001 x the box */
  int lvl = 1;
  if(cfg->pmgrid > 0)
    {
      const double reach = 6.0 * NGRAVS_ASMTH * cfg->box_size / cfg->pmgrid;
      while(lvl < 5 && cfg->box_size / (double)(1 << (lvl + 1)) >= 1.05 * reach)
        lvl++;
      return lvl;
    }
  return 4;
}

void ngravs_host_plan_free(ngravs_dd_plan *plan)
{
  if(plan)
    {
      free(plan->owner_ph);   /* owner_xyz lives in the same block */
      plan->owner_ph = plan->owner_xyz = NULL;
      plan->ncell = 0;
    }
}

/* domain_findExtent + domain_sumCost + domain_findSplit + domain_shiftSplit (domain.c:882-924, 823-877, 347-544) */
int ngravs_host_domain_owners(ngravs_ctx *ctx, const ngravs_comm *cm, int level, double paf, ngravs_dd_plan *plan, ngravs_dd_info *info)
{
  ngravs_config_t cfg;
  double lo[3], hi[3], total = 0, wtot = 0, wmax = 0, cmax = 0, *work = NULL, *twork = NULL;
  int64_t *hist = NULL, ncell, i;
  int32_t *owner_ph = NULL, *owner_xyz;
  int r, x, y, z, nc, rc = 0, W, me;
  ngravs_dd_info local;
  if(!ctx || !cm || !plan || cm->size < 1 || cm->size > 64 || cm->rank < 0 || cm->rank >= cm->size)
    return NGRAVS_ERR_ARG;
  W = cm->size;
  me = cm->rank;
  (void)me;
  if(!info)
    info = &local;
  memset(info, 0, sizeof(*info));
  memset(plan, 0, sizeof(*plan));
  info->seconds[0] = -wall_now();
  CHECK(ngravs_get_config(ctx, &cfg));
  CHECK(ngravs_dd_local_extent(ctx, lo, hi));
  {
    /* one collective for both ends: max(hi) = -min(-hi) */
    double e[6] = {lo[0], lo[1], lo[2], -hi[0], -hi[1], -hi[2]};
    CHECK(cm->allreduce(cm->user, e, 6, NGRAVS_T_F64, NGRAVS_OP_MIN));
    for(r = 0; r < 3; r++)
      {
        lo[r] = e[r];
        hi[r] = -e[3 + r];
      }
  }
  CHECK(ngravs_dd_set_extent(ctx, lo, hi));
  if(level <= 0)
    level = choose_level(&cfg);
  while(level < 7 && (1ll << (3 * level)) < 4ll * W)
    level++;
  info->level = level;
  ncell = 1ll << (3 * level);
  nc = 1 << level;
  hist = malloc(sizeof(int64_t) * (size_t)ncell);
  work = malloc(sizeof(double) * (size_t)ncell);
  owner_ph = malloc(sizeof(int32_t) * 2 * (size_t)ncell);
  twork = malloc(sizeof(double) * 2 * (size_t)W);
  if(!hist || !work || !owner_ph || !twork)
    rc = NGRAVS_ERR_NOMEM;
  owner_xyz = owner_ph ? owner_ph + ncell : NULL;
  if(!rc)
    rc = ngravs_dd_histogram(ctx, level, hist, work);
  if(!rc)
    {
      /* counts and work in ONE collective: whole numbers below 2^53 add up exactly as doubles */
      double *both = malloc(sizeof(double) * 2 * (size_t)ncell);
      if(!both)
        rc = NGRAVS_ERR_NOMEM;
      else
        {
          for(i = 0; i < ncell; i++)
            {
              both[i] = (double)hist[i];
              both[ncell + i] = work[i];
            }
          rc = cm->allreduce(cm->user, both, 2 * ncell, NGRAVS_T_F64, NGRAVS_OP_SUM);
          for(i = 0; i < ncell && !rc; i++)
            {
              hist[i] = (int64_t)(both[i] + 0.5);
              work[i] = both[ncell + i];
            }
          free(both);
        }
    }
  if(!rc)
    {
      for(i = 0; i < ncell; i++)
        total += (double)hist[i];
      if(ngravs_host_split(hist, work, ncell, W, (paf > 0 ? paf : 1.5) * total / W, owner_ph) &&
         ngravs_host_split(hist, work, ncell, W, 0.0, owner_ph))
        rc = NGRAVS_ERR_ARG;
    }
  if(!rc)
    {
      for(r = 0; r < 2 * W; r++)
        twork[r] = 0;
      for(i = 0; i < ncell; i++)
        {
          twork[owner_ph[i]] += work[i];
          twork[W + owner_ph[i]] += (double)hist[i];
        }
      for(r = 0; r < W; r++)
        {
          wtot += twork[r];
          wmax = fmax(wmax, twork[r]);
          cmax = fmax(cmax, twork[W + r]);
        }
      info->work_balance = wtot > 0 ? wmax / (wtot / W) : 1.0;
      info->memory_balance = total > 0 ? cmax / (total / W) : 1.0;
      {
        const int32_t *pht = ph_table(level);
        if(!pht)
          rc = NGRAVS_ERR_NOMEM;
        else
          for(x = 0; x < nc; x++)
            for(y = 0; y < nc; y++)
              for(z = 0; z < nc; z++)
                owner_xyz[((size_t)x * nc + y) * nc + z] = owner_ph[pht[((size_t)x * nc + y) * nc + z]];
      }
      plan->level = level;
      plan->ncell = ncell;
      plan->owner_ph = owner_ph;
      plan->owner_xyz = owner_xyz;
      owner_ph = NULL;
    }
  free(hist);
  free(work);
  free(owner_ph);
  free(twork);
  info->seconds[0] += wall_now();
  return rc == 0 ? 0 : (rc < 0 ? rc : NGRAVS_ERR_STATE);
}

/* one packed-record exchange: what = 0 particle migration (domain_exchangeParticles, domain.c:695-795), what = 1 the
 * short-range halo (replaces the target export / force import of gravtree.c:195-257) */
static int record_exchange(ngravs_ctx *ctx, const ngravs_comm *cm, const ngravs_dd_plan *plan, int what, ngravs_dd_info *info)
{
  const int W = cm->size, me = cm->rank;
  int64_t *counts = malloc(sizeof(int64_t) * (size_t)(3 * 65 + W * W)), *mat, *sb, *rb, nrec = 0, nrecv = 0;
  void *rec = NULL, *recvbuf = NULL;
  int r, rc;
  if(!counts)
    return NGRAVS_ERR_NOMEM;
  mat = counts + 65;
  sb = mat + W * W;
  rb = sb + 65;
  rc = ngravs_dd_pack(ctx, what, plan->level, plan->owner_ph, plan->owner_xyz, W, me, counts, &rec, &nrec);
  if(!rc)
    rc = cm->allgather(cm->user, counts, mat, (int64_t)sizeof(int64_t) * W);
  if(!rc)
    {
      const int64_t rbytes = ngravs_dd_record_bytes(ctx, what);   /* migration records of TreePM runs carry GravPM */
      for(r = 0; r < W; r++)
        {
          sb[r] = counts[r] * rbytes;
          rb[r] = mat[(size_t)r * W + me] * rbytes;
          nrecv += mat[(size_t)r * W + me];
          if(r != me)
            *(what == 0 ? &info->bytes_migration : &info->bytes_halo) += (double)sb[r];
        }
      rc = ngravs_dd_recv_buffer(ctx, nrecv, &recvbuf);
    }
  if(!rc)
    rc = exchange(ctx, cm, rec, sb, recvbuf, rb);
  if(!rc)
    rc = what == 0 ? ngravs_dd_apply_migration(ctx, recvbuf, nrecv) : ngravs_dd_set_halo(ctx, recvbuf, nrecv);
  if(what == 0)
    info->n_migrated_in = nrecv;
  else
    info->n_halo = nrecv;
  free(counts);
  return rc;
}

/* ---- which foreign top cells may this task's targets have to open? ----------------------------------------------------------
 * The walk's own tests (group traversal of kernels_walk.hip = the conservative form of forcetree.c:1364-1518, 1828-1862) against
 * boxes that enclose the task's domain, applied from the root of the global top tree downwards.  A node no target can open is
 * used as a monopole at most: nothing below it is needed.  A node that may be opened: its single-particle children, and -- if it
 * holds <= 8 particles, which the group walk hands over as a leaf -- everything below it, must be on this task. */
typedef struct
{
  int L, ng, periodic, pm, use_theta;
  double box, theta2, aold_min, h_min, rcut, reach6, fsoft[6], corner[3], len;
  const double *sums;    /* all levels, TOP_CW doubles per cell */
  const int64_t *off;    /* first cell of every level */
  const int32_t *xyz;    /* ix | iy << 10 | iz << 20 per cell */
  int nbox;
  double (*bc)[3], (*bh)[3];
  uint8_t *need;         /* per level-L cell */
  const uint8_t *mine;   /* all levels: nothing below this cell could be requested (every top leaf below is this task's own, or empty) */
} need_t;

static double near_abs(double x, double box) { return x - box * rint(x / box); }

static int may_open(const need_t *T, int d, int64_t prefix)
{
  const int cw = NGRAVS_TOP_CW(T->ng);
  const double *s = T->sums + (size_t)(T->off[d] + prefix) * cw;
  const int32_t xyz = T->xyz[T->off[d] + prefix];
  const double len = T->len / (double)(1 << d), half = 0.5 * len;
  double c[3], com[NGRAVS_MAX_GRAVS][3], summass = 0, hs_node = 0;
  int g, j, b, ty, mixed = 0, first = 1;
  c[0] = T->corner[0] + ((xyz & 1023) + 0.5) * len;
  c[1] = T->corner[1] + (((xyz >> 10) & 1023) + 0.5) * len;
  c[2] = T->corner[2] + (((xyz >> 20) & 1023) + 0.5) * len;
  for(g = 0; g < T->ng; g++)
    {
      const double m = s[7 + 4 * g];
      summass += m;
      for(j = 0; j < 3; j++)
        com[g][j] = m > 0 ? s[7 + 4 * g + 1 + j] / m : c[j];
    }
  for(ty = 0; ty < 6; ty++)
    if(s[1 + ty] > 0)
      {
        if(!first && T->fsoft[ty] != hs_node)
          mixed = 1;
        if(first || T->fsoft[ty] > hs_node)
          hs_node = T->fsoft[ty];
        first = 0;
      }
  for(b = 0; b < T->nbox; b++)
    {
      double w[3], pl[3], r2min = 1e300, q2 = 0;
      int drop = 0, open, inside = 1;
      for(j = 0; j < 3; j++)
        {
          pl[j] = c[j] - T->bc[b][j];
          w[j] = T->periodic ? near_abs(pl[j], T->box) : pl[j];
          {
            const double q = fmax(0.0, fabs(w[j]) - T->bh[b][j] - half);
            q2 += q * q;
          }
        }
      if(T->pm && q2 >= T->reach6 * T->reach6)
        continue;      /* nothing inside the cell is within the short-range table of any target of this box */
      for(g = 0; g < T->ng; g++)
        {
          double r2 = 0;
          for(j = 0; j < 3; j++)
            {
              double dd = com[g][j] - T->bc[b][j];
              if(T->periodic)
                dd = near_abs(dd, T->box);
              dd = fmax(0.0, fabs(dd) - T->bh[b][j]);
              r2 += dd * dd;
            }
          if(r2 < r2min)
            r2min = r2;
        }
      if(T->pm && r2min > T->rcut * T->rcut)
        for(j = 0; j < 3; j++)
          if(fabs(w[j]) - T->bh[b][j] > T->rcut + half)
            drop = 1;
      if(drop)
        continue;
      if(T->use_theta)
        open = len * len > r2min * T->theta2;
      else
        {
          open = summass * len * len > r2min * r2min * T->aold_min;
          for(j = 0; j < 3; j++)
            if(!(fabs(pl[j]) - T->bh[b][j] < 0.60 * len))
              inside = 0;
          open = open || inside;
        }
      if(!open && T->h_min < hs_node && r2min < hs_node * hs_node && mixed)
        open = 1;
      if(open)
        return 1;
    }
  return 0;
}

static void need_all_below(const need_t *T, int d, int64_t prefix)
{
  const int sh = 3 * (T->L - d);
  int64_t i;
  for(i = prefix << sh; i < ((prefix + 1) << sh); i++)
    if(T->sums[(size_t)(T->off[T->L] + i) * NGRAVS_TOP_CW(T->ng)] > 0.5)
      T->need[i] = 1;
}

/* called for the children of a node that may be opened */
static void need_visit(const need_t *T, int d, int64_t prefix)
{
  const double cnt = T->sums[(size_t)(T->off[d] + prefix) * NGRAVS_TOP_CW(T->ng)];
  int k;
  if(cnt < 0.5 || T->mine[T->off[d] + prefix])
    return;
  if(cnt < 1.5)   /* a single particle: it hangs directly below the opened parent */
    {
      need_all_below(T, d, prefix);
      return;
    }
  if(!may_open(T, d, prefix))
    return;
  if(d == T->L || cnt < 8.5)   /* a top leaf, or a node the group walk hands over particle by particle (GW_NLEAF) */
    {
      need_all_below(T, d, prefix);
      return;
    }
  for(k = 0; k < 8; k++)
    need_visit(T, d + 1, prefix * 8 + k);
}

/* Top-leaf moments + tree-node import + local Peano order: the second half of domain_Decomposition() for a task whose own
 * particles are in place (force_exchange_pseudodata / force_treeupdate_pseudos, forcetree.c:766-996, and what replaces the
 * export / import loop of gravtree.c:112-285) */
int ngravs_host_domain_halo(ngravs_ctx *ctx, const ngravs_comm *cm, const ngravs_dd_plan *plan, ngravs_dd_info *info)
{
  ngravs_dd_info local;
  ngravs_config_t cfg;
  need_t T;
  double *sums = NULL, dom[8], bounds[2], (*bc)[3] = NULL, (*bh)[3] = NULL, blo[64][3], bhi[64][3];
  int64_t *off = NULL, ncell, tot, i, *counts = NULL, *mat, *sb, *rb, nrec = 0, nrecv = 0;
  int32_t *xyz = NULL;
  uint8_t *need = NULL, *allneed = NULL, *present = NULL, *mine = NULL;
  uint64_t *reqmask = NULL;
  void *rec = NULL, *recvbuf = NULL;
  int L, nc, d, x, y, z, r, j, k, cw, used[64], rc = 0, W, me;
  if(!ctx || !cm || !plan || !plan->owner_ph)
    return NGRAVS_ERR_ARG;
  W = cm->size;
  me = cm->rank;
  if(!info)
    {
      memset(&local, 0, sizeof(local));
      info = &local;
    }
  CHECK(ngravs_get_config(ctx, &cfg));
  CHECK(ngravs_get_domain_extent(ctx, dom));
  L = plan->level;
  ncell = plan->ncell;
  nc = 1 << L;
  cw = NGRAVS_TOP_CW(cfg.n_gravs);
  off = malloc(sizeof(int64_t) * (size_t)(L + 2));
  off[0] = 0;
  for(d = 0; d <= L; d++)
    off[d + 1] = off[d] + (1ll << (3 * d));
  tot = off[L + 1];
  sums = calloc((size_t)tot * cw, sizeof(double));
  xyz = malloc(sizeof(int32_t) * (size_t)tot);
  need = calloc((size_t)ncell, 1);
  mine = malloc((size_t)tot);
  allneed = malloc((size_t)ncell * (size_t)W);
  present = malloc((size_t)ncell);
  reqmask = calloc((size_t)ncell, sizeof(uint64_t));
  counts = malloc(sizeof(int64_t) * (size_t)(3 * 65 + W * W));
  if(!sums || !xyz || !need || !mine || !allneed || !present || !reqmask || !counts)
    rc = NGRAVS_ERR_NOMEM;
  /* top-leaf sums of all tasks (DomainMoment[], forcetree.c:766-850), then every coarser level */
  if(!rc)
    {
      info->seconds[2] = -wall_now();
      rc = ngravs_dd_cell_sums(ctx, L, sums + (size_t)off[L] * cw);
    }
  if(!rc)
    rc = cm->allreduce(cm->user, sums + (size_t)off[L] * cw, ncell * cw, NGRAVS_T_F64, NGRAVS_OP_SUM);
  if(!rc)
    {
      info->seconds[2] += wall_now();
      info->seconds[3] = -wall_now();
      rc = ngravs_dd_target_bounds(ctx, bounds);
    }
  if(!rc)
    {
      for(d = L - 1; d >= 0; d--)
        for(i = 0; i < (1ll << (3 * d)); i++)
          for(k = 0; k < 8; k++)
            for(j = 0; j < cw; j++)
              sums[(size_t)(off[d] + i) * cw + j] += sums[(size_t)(off[d + 1] + i * 8 + k) * cw + j];
      for(d = 0; d <= L && !rc; d++)
        {
          const int32_t *pht = ph_table(d);
          const int ncd = 1 << d;
          if(!pht)
            {
              rc = NGRAVS_ERR_NOMEM;
              break;
            }
          for(x = 0; x < ncd; x++)
            for(y = 0; y < ncd; y++)
              for(z = 0; z < ncd; z++)
                xyz[off[d] + pht[((size_t)x * ncd + y) * ncd + z]] = x | (y << 10) | (z << 20);
        }
      /* boxes around the own cells that hold particles, one per coarse (<= 4^3) block of the domain grid */
      {
        const int csh = L > 2 ? L - 2 : 0;
        const double cl = dom[6] / nc;
        const int32_t *phL = ph_table(L);
        for(k = 0; k < 64; k++)
          used[k] = 0;
        for(x = 0; x < nc; x++)
          for(y = 0; y < nc; y++)
            for(z = 0; z < nc; z++)
              {
                const int64_t cell = phL ? phL[((size_t)x * nc + y) * nc + z] : ngravs_peano_hilbert_key(x, y, z, L);
                const int cxyz[3] = {x, y, z};
                if(plan->owner_ph[cell] != me || sums[(size_t)(off[L] + cell) * cw] < 0.5)
                  continue;
                k = (((x >> csh) & 3) * 4 + ((y >> csh) & 3)) * 4 + ((z >> csh) & 3);
                for(j = 0; j < 3; j++)
                  {
                    const double lo = dom[j] + cxyz[j] * cl, hi = lo + cl;
                    if(!used[k] || lo < blo[k][j])
                      blo[k][j] = lo;
                    if(!used[k] || hi > bhi[k][j])
                      bhi[k][j] = hi;
                  }
                used[k] = 1;
              }
        bc = malloc(sizeof(*bc) * 64);
        bh = malloc(sizeof(*bh) * 64);
        T.nbox = 0;
        for(k = 0; k < 64; k++)
          if(used[k])
            {
              for(j = 0; j < 3; j++)
                {
                  bc[T.nbox][j] = 0.5 * (blo[k][j] + bhi[k][j]);
                  bh[T.nbox][j] = 0.5 * (bhi[k][j] - blo[k][j]) + 1e-9 * dom[6];   /* rounding slack */
                }
              T.nbox++;
            }
      }
      T.L = L;
      T.ng = cfg.n_gravs;
      T.periodic = cfg.periodic;
      T.pm = cfg.pmgrid != 0;
      T.use_theta = cfg.err_tol_theta != 0;
      T.box = cfg.box_size;
      T.theta2 = cfg.err_tol_theta * cfg.err_tol_theta;
      T.aold_min = bounds[0];
      T.h_min = bounds[1];
      T.rcut = cfg.rcut;
      T.reach6 = 6.0 * cfg.asmth;
      for(j = 0; j < 6; j++)
        T.fsoft[j] = cfg.force_softening[j];
      for(j = 0; j < 3; j++)
        T.corner[j] = dom[j];
      T.len = dom[6];
      T.sums = sums;
      T.off = off;
      T.xyz = xyz;
      T.bc = bc;
      T.bh = bh;
      T.need = need;
      /* below a cell whose top leaves are all this task's own (or empty) there is nothing to request */
      for(i = 0; i < ncell; i++)
        mine[off[L] + i] = plan->owner_ph[i] == me || sums[(size_t)(off[L] + i) * cw] < 0.5;
      for(d = L - 1; d >= 0; d--)
        for(i = 0; i < (1ll << (3 * d)); i++)
          {
            uint8_t a = 1;
            for(k = 0; k < 8; k++)
              a &= mine[off[d + 1] + i * 8 + k];
            mine[off[d] + i] = a;
          }
      T.mine = mine;
      if(T.nbox > 0)   /* the root is opened by every target inside it */
        for(k = 0; k < 8; k++)
          need_visit(&T, 1, k);
      for(i = 0; i < ncell; i++)
        if(plan->owner_ph[i] == me)
          need[i] = 0;   /* own cells are here already */
      info->seconds[3] += wall_now();
      info->seconds[4] = -wall_now();
      rc = cm->allgather(cm->user, need, allneed, ncell);
    }
  if(!rc)
    {
      /* the owners ship every particle of the requested cells (replaces the export of targets, gravtree.c:195-257) */
      for(r = 0; r < W; r++)
        for(i = 0; i < ncell; i++)
          if(allneed[(size_t)r * ncell + i] && plan->owner_ph[i] == me)
            reqmask[i] |= 1ull << r;
      mat = counts + 65;
      sb = mat + W * W;
      rb = sb + 65;
      rc = ngravs_dd_pack_cells(ctx, L, reqmask, W, me, counts, &rec, &nrec);
      if(!rc)
        rc = cm->allgather(cm->user, counts, mat, (int64_t)sizeof(int64_t) * W);
      if(!rc)
        {
          for(r = 0; r < W; r++)
            {
              sb[r] = counts[r] * NGRAVS_DD_RECORD_BYTES;
              rb[r] = mat[(size_t)r * W + me] * NGRAVS_DD_RECORD_BYTES;
              nrecv += mat[(size_t)r * W + me];
              if(r != me)
                info->bytes_halo += (double)sb[r];
            }
          rc = ngravs_dd_recv_buffer(ctx, nrecv, &recvbuf);
        }
      info->seconds[4] += wall_now();
      info->seconds[5] = -wall_now();
      if(!rc)
        rc = exchange(ctx, cm, rec, sb, recvbuf, rb);
      if(!rc)
        rc = ngravs_dd_set_halo(ctx, recvbuf, nrecv);
      info->seconds[5] += wall_now();
      info->n_halo = nrecv;
    }
  if(!rc)
    {
      info->seconds[6] = -wall_now();
      for(i = 0; i < ncell; i++)
        present[i] = (plan->owner_ph[i] == me || need[i]) ? 1 : 0;
      rc = ngravs_dd_set_top(ctx, L, sums + (size_t)off[L] * cw, present);
      info->seconds[6] += wall_now();
    }
  if(!rc)
    {
      info->n_local = ngravs_dd_num_local(ctx);
      info->seconds[7] = -wall_now();
      rc = ngravs_domain_decomposition(ctx);
      info->seconds[7] += wall_now();
    }
  free(off);
  free(sums);
  free(xyz);
  free(need);
  free(mine);
  free(allneed);
  free(present);
  free(reqmask);
  free(counts);
  free(bc);
  free(bh);
  return rc == 0 ? 0 : (rc < 0 ? rc : NGRAVS_ERR_STATE);
}

int ngravs_host_domain_decomposition(ngravs_ctx *ctx, const ngravs_comm *cm, int level, double paf, ngravs_dd_info *info)
{
  ngravs_dd_plan plan;
  ngravs_dd_info local;
  int rc;
  if(!info)
    info = &local;
  CHECK(ngravs_host_domain_owners(ctx, cm, level, paf, &plan, info));
  info->seconds[1] = -wall_now();
  rc = record_exchange(ctx, cm, &plan, 0, info);
  info->seconds[1] += wall_now();
  if(!rc)
    rc = ngravs_host_domain_halo(ctx, cm, &plan, info);
  ngravs_host_plan_free(&plan);
  return rc == 0 ? 0 : (rc < 0 ? rc : NGRAVS_ERR_STATE);
}

/* ---- pmforce_periodic on the slab-decomposed mesh -------------------------------------------------------------------------- */
static __thread double pm_seconds[13];   /* per host thread (host_shim_test runs two tasks as two threads); last call: [0] deposit + bounding boxes, then per stage s: [1+3s] pack, [2+3s] exchange, [3+3s] unpack */

void ngravs_host_pm_seconds(double out[13])
{
  int i;
  for(i = 0; i < 13; i++)
    out[i] = pm_seconds[i];
}

int ngravs_host_pmforce_periodic(ngravs_ctx *ctx, const ngravs_comm *cm)
{
  const int W = cm->size;
  int32_t bb[6], *all;
  int64_t *sc, *rc_;
  void *send = NULL, *recv = NULL;
  int stage, r, rc;
  if(!ctx || !cm || W < 1)
    return NGRAVS_ERR_ARG;
  all = malloc(sizeof(int32_t) * 6 * (size_t)W);
  sc = malloc(sizeof(int64_t) * 2 * (size_t)W);
  if(!all || !sc)
    {
      free(all);
      free(sc);
      return NGRAVS_ERR_NOMEM;
    }
  rc_ = sc + W;
  memset(pm_seconds, 0, sizeof(pm_seconds));
  pm_seconds[0] = -wall_now();
  rc = ngravs_pm_slab_begin(ctx, cm->rank, W, bb);
  if(!rc)
    rc = cm->allgather(cm->user, bb, all, (int64_t)sizeof(bb));   /* meshmin/meshmax lists, pm_periodic.c:285-291 */
  pm_seconds[0] += wall_now();
  for(stage = 0; stage < 4 && !rc; stage++)
    {
      double t0 = wall_now(), t1;
      rc = ngravs_pm_slab_pack(ctx, stage, all, sc, rc_, &send, &recv);
      for(r = 0; r < W; r++)
        {
          sc[r] *= (int64_t)sizeof(double);
          rc_[r] *= (int64_t)sizeof(double);
        }
      t1 = wall_now();
      pm_seconds[1 + 3 * stage] = t1 - t0;
      if(!rc)
        rc = exchange(ctx, cm, send, sc, recv, rc_);
      t0 = wall_now();
      pm_seconds[2 + 3 * stage] = t0 - t1;
      if(!rc)
        rc = ngravs_pm_slab_unpack(ctx, stage);
      pm_seconds[3 + 3 * stage] = wall_now() - t0;
    }
  free(all);
  free(sc);
  return rc == 0 ? 0 : (rc < 0 ? rc : NGRAVS_ERR_STATE);
}

/* compute_accelerations(0), gravity part (accel.c:24-58) */
int ngravs_host_compute_accelerations(ngravs_ctx *ctx, const ngravs_comm *cm, int pm_step, ngravs_dd_info *info)
{
  ngravs_config_t cfg;
  CHECK(ngravs_get_config(ctx, &cfg));
  CHECK(ngravs_host_domain_decomposition(ctx, cm, 0, 0.0, info));
  if(pm_step && cfg.pmgrid)
    CHECK(ngravs_host_pmforce_periodic(ctx, cm));
  return ngravs_gravity_tree(ctx);
}
