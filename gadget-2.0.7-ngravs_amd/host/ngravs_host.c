/* ngravs_host.c -- multi-task drivers of the gravity path in plain C over a communicator vtable (include/ngravs_host.h).
 * Linked into libngravs_hip.so; uses nothing but the public C ABI of include/ngravs_hip.h and the caller's callbacks, so the
 * same code runs under RCCL (host/ngravs_comm_rccl.c: gadget_glue.c, distributed.py), under MPI (gadget_glue.c's fallback),
 * under torch.distributed "gloo" (rehearsals) and under the shared-memory communicator of host/host_shim_test.c.
 *
 * Collectives of one decomposition in the steady state: (1) all-reduce MIN of the extent and the target bounds, (2) all-reduce
 * SUM of the per-leaf sums, (3) all-gather of the migration counts and the import requests, (4) all-to-all-v of migrating
 * particles -- skipped by every task when the count matrix says nothing moves --, (5) all-to-all-v of the imported leaves.
 * The receive counts of (5) need no collective: a leaf's owner holds ALL its particles after (4) and every task knows the
 * global count of every leaf.
 */
#define _POSIX_C_SOURCE 199309L
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include "ngravs_host.h"
#include "ngravs_peano.h"

#define CHECK(expr)            \
  do                           \
    {                          \
      int rc__ = (expr);       \
      if(rc__ != 0)            \
        return rc__ < 0 ? rc__ : NGRAVS_ERR_STATE; \
    }                          \
  while(0)

static double wall_now(void)
{
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

static int status_of(int rc) { return rc == 0 ? 0 : (rc < 0 ? rc : NGRAVS_ERR_STATE); }

/* all-to-all-v of library device buffers; counts in bytes.  Without device-capable transport the blocks go through host memory. */
static int exchange(ngravs_ctx *ctx, const ngravs_comm *cm, const void *dsend, const int64_t *sbytes, void *drecv, const int64_t *rbytes)
{
  const int W = cm->size;
  int64_t *sd = malloc(sizeof(int64_t) * 2 * (size_t)(W + 1)), *rd;
  int r, rc = 0;
  if(!sd)
    return NGRAVS_ERR_NOMEM;
  rd = sd + W + 1;
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
  return status_of(rc);
}

/* =========================================================================================================================
 *  The top tree (TopNodes[], domain.c:933-1138)
 * ========================================================================================================================= */
void ngravs_host_toptree_free(ngravs_toptree *t)
{
  if(!t)
    return;
  free(t->child);   /* one block: child, level, leaf, xyz; node_of_leaf separately */
  free(t->node_of_leaf);
  memset(t, 0, sizeof(*t));
}

static int tt_alloc(ngravs_toptree *t, int32_t nnode)
{
  memset(t, 0, sizeof(*t));
  t->child = malloc(sizeof(int32_t) * 6 * (size_t)(nnode > 0 ? nnode : 1));
  if(!t->child)
    return NGRAVS_ERR_NOMEM;
  t->level = t->child + nnode;
  t->leaf = t->level + nnode;
  t->xyz = t->leaf + nnode;
  t->nnode = nnode;
  return 0;
}

/* levels, cell coordinates and the numbering of the leaves along the curve from the child table.  The children of a node are
 * stored in key order; which octant the r-th child is follows from the Peano-Hilbert state of the node (include/ngravs_peano.h:
 * step[state][octant] = digit | next state << 3), carried down from the root. */
static int tt_finish(ngravs_toptree *t)
{
  const int32_t n = t->nnode;
  int32_t *stack, *state, sp = 0, nleaf = 0, i;
  if(n < 1)
    return NGRAVS_ERR_ARG;
  for(i = 0; i < n; i++)
    if(t->child[i] >= 0 && (t->child[i] < 1 || t->child[i] > n - 8))
      return NGRAVS_ERR_ARG;
  stack = malloc(sizeof(int32_t) * 2 * (size_t)n);
  if(!stack)
    return NGRAVS_ERR_NOMEM;
  state = stack + n;
  for(i = 0; i < n; i++)
    t->leaf[i] = -2;   /* not reached yet */
  t->level[0] = 0;
  t->xyz[0] = t->xyz[1] = t->xyz[2] = 0;
  state[0] = 0;
  t->depth = 0;
  stack[sp++] = 0;
  /* depth-first, children in key order: the leaves come out in curve order */
  while(sp > 0)
    {
      const int32_t p = stack[--sp];
      int oct;
      if(t->child[p] < 0)
        {
          t->leaf[p] = nleaf++;
          continue;
        }
      t->leaf[p] = -1;
      if(t->level[p] + 1 > t->depth)
        t->depth = t->level[p] + 1;
      for(oct = 0; oct < 8; oct++)
        {
          const unsigned v = ngravs_ph_step_host[state[p]][oct];
          const int32_t c = t->child[p] + (int32_t)(v & 7u);
          if(t->leaf[c] != -2)
            {
              free(stack);
              return NGRAVS_ERR_ARG;   /* two parents */
            }
          t->leaf[c] = -3;
          t->level[c] = t->level[p] + 1;
          t->xyz[3 * c + 0] = 2 * t->xyz[3 * p + 0] + ((oct >> 2) & 1);
          t->xyz[3 * c + 1] = 2 * t->xyz[3 * p + 1] + ((oct >> 1) & 1);
          t->xyz[3 * c + 2] = 2 * t->xyz[3 * p + 2] + (oct & 1);
          state[c] = (int32_t)(v >> 3);
        }
      for(oct = 7; oct >= 0; oct--)   /* pushed in reverse: child 0 is popped first */
        stack[sp++] = t->child[p] + oct;
    }
  free(stack);
  for(i = 0; i < n; i++)
    if(t->leaf[i] < -1)
      return NGRAVS_ERR_ARG;   /* unreachable node */
  t->nleaf = nleaf;
  free(t->node_of_leaf);
  t->node_of_leaf = malloc(sizeof(int32_t) * (size_t)nleaf);
  if(!t->node_of_leaf)
    return NGRAVS_ERR_NOMEM;
  for(i = 0; i < n; i++)
    if(t->leaf[i] >= 0)
      t->node_of_leaf[t->leaf[i]] = i;
  return 0;
}

/* a copy of a finished tree (all tables), e.g. of the view ngravs_host_toptree_borrow() lends */
static int tt_clone(const ngravs_toptree *src, ngravs_toptree *out)
{
  CHECK(tt_alloc(out, src->nnode));
  memcpy(out->child, src->child, sizeof(int32_t) * (size_t)src->nnode);
  memcpy(out->level, src->level, sizeof(int32_t) * (size_t)src->nnode);
  memcpy(out->leaf, src->leaf, sizeof(int32_t) * (size_t)src->nnode);
  memcpy(out->xyz, src->xyz, sizeof(int32_t) * 3 * (size_t)src->nnode);
  out->nleaf = src->nleaf;
  out->depth = src->depth;
  out->node_of_leaf = malloc(sizeof(int32_t) * (size_t)(src->nleaf > 0 ? src->nleaf : 1));
  if(!out->node_of_leaf)
    {
      ngravs_host_toptree_free(out);
      return NGRAVS_ERR_NOMEM;
    }
  memcpy(out->node_of_leaf, src->node_of_leaf, sizeof(int32_t) * (size_t)src->nleaf);
  return 0;
}

int ngravs_host_toptree_from_children(ngravs_toptree *t, const int32_t *child, int32_t nnode)
{
  int rc;
  if(!t || !child || nnode < 1)
    return NGRAVS_ERR_ARG;
  CHECK(tt_alloc(t, nnode));
  memcpy(t->child, child, sizeof(int32_t) * (size_t)nnode);
  rc = tt_finish(t);
  if(rc)
    ngravs_host_toptree_free(t);
  return rc;
}

int ngravs_host_toptree_init(ngravs_toptree *t, int level)
{
  int64_t nn = 0, first, cnt, i;
  int d, rc;
  if(!t || level < 0 || level > 7)
    return NGRAVS_ERR_ARG;
  for(d = 0; d <= level; d++)
    nn += 1ll << (3 * d);
  CHECK(tt_alloc(t, (int32_t)nn));
  /* breadth-first numbering: the nodes of level d start at (8^d - 1) / 7 */
  for(d = 0, first = 0; d <= level; d++)
    {
      cnt = 1ll << (3 * d);
      for(i = 0; i < cnt; i++)
        t->child[first + i] = d < level ? (int32_t)(first + cnt + 8 * i) : -1;
      first += cnt;
    }
  rc = tt_finish(t);
  if(rc)
    ngravs_host_toptree_free(t);
  return rc;
}

/* global particle count of every node from the leaf counts (children before parents: a child's index is larger than its
 * parent's in every tree this file builds -- checked) */
static int tt_node_counts(const ngravs_toptree *t, const double *leaf_count, double *cnt)
{
  int32_t i, k;
  for(i = t->nnode - 1; i >= 0; i--)
    if(t->child[i] < 0)
      cnt[i] = leaf_count[t->leaf[i]];
    else
      {
        if(t->child[i] <= i)
          return NGRAVS_ERR_ARG;
        cnt[i] = 0;
        for(k = 0; k < 8; k++)
          cnt[i] += cnt[t->child[i] + k];
      }
  return 0;
}

/* One round of the reference's rule (domain_topsplit, domain.c:1060-1138: the root is split; a daughter is split if
 * Count > TotNumPart / (TOPNODEFACTOR * NTask); cells of the deepest key level are never split) applied to a tree whose leaf
 * counts are known.  A heavy leaf is split by as many levels at once as a uniform filling would need, at most three (its new
 * leaves' counts are unknown: the caller counts again, and the next round merges what was split too far), so that a centrally
 * concentrated set needs a few rounds instead of one per level.  The new tree is written breadth-first. */
int ngravs_host_toptree_adapt(const ngravs_toptree *t, const double *leaf_count, double thresh, int max_level, ngravs_toptree *out)
{
  double *cnt;
  int32_t *queue, qh = 0, qt = 0, unknown = 0, i;
  int64_t cap;
  int rc;
  if(!t || !leaf_count || !out || t->nnode < 1 || !(thresh >= 1.0))
    return NGRAVS_ERR_ARG;
  if(max_level > NGRAVS_TOPLEVEL_MAX)
    max_level = NGRAVS_TOPLEVEL_MAX;
  cnt = malloc(sizeof(double) * (size_t)t->nnode);
  if(!cnt)
    return NGRAVS_ERR_NOMEM;
  if((rc = tt_node_counts(t, leaf_count, cnt)))
    {
      free(cnt);
      return rc;
    }
  {
    /* the steady state: no leaf to split, no subtree to merge -- the tree stays (out->nnode = 0 says so; nothing is built) */
    int violated = t->child[0] < 0;
    for(i = 1; i < t->nnode && !violated; i++)
      if(t->child[i] < 0 ? (cnt[i] > thresh && t->level[i] < max_level) : !(cnt[i] > thresh && t->level[i] < max_level))
        violated = 1;
    if(!violated)
      {
        free(cnt);
        memset(out, 0, sizeof(*out));
        return 0;
      }
  }
  cap = (int64_t)t->nnode + 64;
  for(i = 0; i < t->nnode; i++)
    if(t->child[i] < 0 && cnt[i] > thresh)
      cap += 8 + 64 + 512;   /* <= 3 levels per heavy leaf */
  if(cap > NGRAVS_TOPNODES_MAX)
    cap = NGRAVS_TOPNODES_MAX;
  queue = malloc(sizeof(int32_t) * 2 * (size_t)cap);   /* queue[2 q] = old node, or -1 - (levels still to split blindly); queue[2 q + 1] = level */
  if(!queue || tt_alloc(out, (int32_t)cap))
    {
      free(cnt);
      free(queue);
      return NGRAVS_ERR_NOMEM;
    }
  queue[0] = 0;
  queue[1] = 0;
  qt = 1;
  while(qh < qt && !rc)
    {
      const int32_t src = queue[2 * qh], lvl = queue[2 * qh + 1], me = qh;   /* the q-th node of the queue is node q of the new tree */
      int32_t first_src = -1, blind = -1;
      int split;
      qh++;
      if(src >= 0)
        {
          split = lvl == 0 || (cnt[src] > thresh && lvl < max_level);   /* the root is always split (TopNodes[0].Size >= 8) */
          if(split && t->child[src] >= 0)
            first_src = t->child[src];
          else if(split)
            {
              double c = cnt[src] / 8.0;
              blind = 0;
              while(blind < 2 && c > thresh && lvl + 1 + blind < max_level)
                {
                  c /= 8.0;
                  blind++;
                }
            }
        }
      else
        {
          blind = -1 - src - 1;   /* levels left below this one */
          split = blind >= 0;
          if(!split)
            unknown++;
        }
      if(!split)
        {
          out->child[me] = -1;   /* a leaf of the old tree that stays one, a light subtree merged into one leaf, or a new leaf */
          continue;
        }
      if((int64_t)qt + 8 > cap)
        {
          rc = NGRAVS_ERR_NOMEM;
          break;
        }
      out->child[me] = qt;
      for(i = 0; i < 8; i++)
        {
          queue[2 * qt] = first_src >= 0 ? first_src + i : -1 - blind;
          queue[2 * qt + 1] = lvl + 1;
          qt++;
        }
    }
  free(cnt);
  free(queue);
  if(!rc)
    {
      /* the arrays were laid out for `cap` nodes: repack for the qt nodes there are */
      ngravs_toptree packed;
      rc = tt_alloc(&packed, qt);
      if(!rc)
        {
          memcpy(packed.child, out->child, sizeof(int32_t) * (size_t)qt);
          ngravs_host_toptree_free(out);
          *out = packed;
          rc = tt_finish(out);
        }
    }
  if(rc)
    {
      ngravs_host_toptree_free(out);
      return rc;
    }
  return unknown;
}

/* =========================================================================================================================
 *  The cut of the leaf sequence over the tasks.  The reference bisects the sequence by particle count and then shifts the
 *  boundaries by work (domain_findSplit / domain_shiftSplit, domain.c:347-544).  Here the whole problem is solved at once: the
 *  smallest bottleneck B such that the sequence can be cut into NTask contiguous pieces with work <= B and count <= max_load
 *  each (whether a B is feasible is a greedy sweep; B is found by bisection on the real line), then the greedy cut for that B,
 *  every task keeping at least one leaf.
 * ========================================================================================================================= */
/* prefix sums pw[i] = work of leaves [0, i), pc[i] = count of leaves [0, i): a task that starts at leaf `first` takes the longest
 * run with work <= B and count <= max_load (both prefix sums are monotone: one binary search), at least one leaf, and leaves one
 * leaf for every later task */
static int cut_sweep(const double *pw, const double *pc, int64_t n, int ntask, double B, double max_load, int32_t *owner)
{
  int64_t i = 0, k;
  int t;
  for(t = 0; t < ntask; t++)
    {
      const int64_t limit = n - (ntask - 1 - t);   /* exclusive end this task may reach */
      int64_t lo = i, hi = limit;                  /* largest e in [i, limit] with the run [i, e) within both bounds */
      if(i >= limit)
        return -1;
      if(pc[i + 1] - pc[i] > max_load)
        return -1;   /* one leaf alone breaks the memory bound */
      while(lo < hi)
        {
          const int64_t mid = (lo + hi + 1) >> 1;
          if(pw[mid] - pw[i] <= B && pc[mid] - pc[i] <= max_load)
            lo = mid;
          else
            hi = mid - 1;
        }
      if(lo == i)
        lo = i + 1;   /* a task takes at least one leaf (the caller's B is never below the heaviest leaf, up to rounding) */
      if(owner)
        for(k = i; k < lo; k++)
          owner[k] = t;
      i = lo;
    }
  return i == n ? 0 : -1;
}

int ngravs_host_split(const double *count, const double *work, int64_t nleaf, int ntask, double max_load, int32_t *owner)
{
  double lo = 0, hi = 0, wmax = 0, *pw, *pc;
  int64_t i;
  int it, rc;
  if(!count || !owner || ntask < 1 || nleaf < ntask)
    return -1;
  if(!work)
    work = count;
  if(!(max_load > 0))
    max_load = 1e300;
  if(ntask == 1)
    {
      for(i = 0; i < nleaf; i++)
        owner[i] = 0;
      return 0;
    }
  pw = malloc(sizeof(double) * 2 * (size_t)(nleaf + 1));
  if(!pw)
    return -1;
  pc = pw + nleaf + 1;
  pw[0] = pc[0] = 0;
  for(i = 0; i < nleaf; i++)
    {
      pw[i + 1] = pw[i] + work[i];
      pc[i + 1] = pc[i] + count[i];
      if(work[i] > wmax)
        wmax = work[i];
    }
  hi = pw[nleaf] * (1.0 + 1e-12) + 1e-300;
  if(cut_sweep(pw, pc, nleaf, ntask, hi, max_load, NULL))
    {
      free(pw);
      return -1;   /* the memory bound alone cannot be met */
    }
  lo = fmax(wmax, pw[nleaf] / ntask) * (1.0 - 1e-12);
  if(cut_sweep(pw, pc, nleaf, ntask, lo, max_load, NULL) == 0)
    hi = lo;
  else
    for(it = 0; it < 64 && hi - lo > 1e-13 * hi; it++)
      {
        const double mid = 0.5 * (lo + hi);
        if(cut_sweep(pw, pc, nleaf, ntask, mid, max_load, NULL) == 0)
          hi = mid;
        else
          lo = mid;
      }
  rc = cut_sweep(pw, pc, nleaf, ntask, hi, max_load, owner);
  free(pw);
  return rc;
}

/* =========================================================================================================================
 *  Which foreign top leaves may this task's targets have to open?
 *  The walk's own tests (group traversal of kernels_walk.hip = the conservative form of forcetree.c:1364-1518, 1828-1862)
 *  against boxes that enclose the task's leaves, applied from the root of the top tree downwards.  A node no target can open is
 *  used as a monopole at most: nothing below it is needed.  A node that may be opened: its single-particle children, and -- if
 *  it holds <= 8 particles, which the group walk hands over as a leaf -- everything below it, must be on this task.
 * ========================================================================================================================= */
typedef struct
{
  int ng, periodic, pm, use_theta;
  double box, theta2, aold_min, h_min, rcut, reach6, fsoft[6], corner[3], len;
  const ngravs_toptree *t;
  const double *sums;    /* TOP_CW doubles per node */
  int nbox;
  double (*bc)[3], (*bh)[3];
  uint8_t *need;         /* per leaf */
  const uint8_t *mine;   /* per node: nothing below could be requested (every leaf below is this task's own, or empty) */
} need_t;

static double near_abs(double x, double box) { return x - box * rint(x / box); }

static int may_open(const need_t *T, int32_t node)
{
  const int cw = NGRAVS_TOP_CW(T->ng);
  const double *s = T->sums + (size_t)node * cw;
  const int d = T->t->level[node];
  const double len = ldexp(T->len, -d), half = 0.5 * len;
  double c[3], com[NGRAVS_MAX_GRAVS][3], summass = 0, hs_node = 0;
  int g, j, b, ty, mixed = 0, first = 1;
  for(j = 0; j < 3; j++)
    c[j] = T->corner[j] + (T->t->xyz[3 * node + j] + 0.5) * len;
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

static void need_all_below(const need_t *T, int32_t node)
{
  if(T->t->child[node] < 0)
    {
      if(T->sums[(size_t)node * NGRAVS_TOP_CW(T->ng)] > 0.5)
        T->need[T->t->leaf[node]] = 1;
    }
  else
    {
      int k;
      for(k = 0; k < 8; k++)
        need_all_below(T, T->t->child[node] + k);
    }
}

/* called for the children of a node that may be opened */
static void need_visit(const need_t *T, int32_t node)
{
  const double cnt = T->sums[(size_t)node * NGRAVS_TOP_CW(T->ng)];
  int k;
  if(cnt < 0.5 || T->mine[node])
    return;
  if(cnt < 1.5)   /* a single particle: it hangs directly below the opened parent */
    {
      need_all_below(T, node);
      return;
    }
  if(!may_open(T, node))
    return;
  if(T->t->child[node] < 0 || cnt < 8.5)   /* a top leaf, or a node the group walk hands over particle by particle (GW_NLEAF) */
    {
      need_all_below(T, node);
      return;
    }
  for(k = 0; k < 8; k++)
    need_visit(T, T->t->child[node] + k);
}

int ngravs_host_import_request(const ngravs_config_t *cfg, const double dom[8], const ngravs_toptree *t, const double *node_sums,
                               const int32_t *leaf_owner, int me, const double bounds[2], uint8_t *need)
{
  return ngravs_host_import_request_margin(cfg, dom, t, node_sums, leaf_owner, me, bounds, 0.0, need);
}

int ngravs_host_import_request_margin(const ngravs_config_t *cfg, const double dom[8], const ngravs_toptree *t, const double *node_sums,
                                      const int32_t *leaf_owner, int me, const double bounds[2], double margin, uint8_t *need)
{
  need_t T;
  double blo[64][3], bhi[64][3], (*bc)[3], (*bh)[3];
  uint8_t *mine;
  int32_t i;
  int j, k, cw, used[64], rc = 0;
  if(!cfg || !dom || !t || !node_sums || !leaf_owner || !bounds || !need || t->nnode < 1)
    return NGRAVS_ERR_ARG;
  cw = NGRAVS_TOP_CW(cfg->n_gravs);
  memset(need, 0, (size_t)t->nleaf);
  mine = malloc((size_t)t->nnode);
  bc = malloc(sizeof(*bc) * 64);
  bh = malloc(sizeof(*bh) * 64);
  if(!mine || !bc || !bh)
    rc = NGRAVS_ERR_NOMEM;
  if(!rc)
    {
      /* boxes around the own leaves that hold particles, one per block of the level-2 grid (64 blocks) */
      for(k = 0; k < 64; k++)
        used[k] = 0;
      for(i = 0; i < t->nleaf; i++)
        {
          const int32_t node = t->node_of_leaf[i];
          const int d = t->level[node];
          const double cl = ldexp(dom[6], -d);
          int b[3];
          if(leaf_owner[i] != me || node_sums[(size_t)node * cw] < 0.5)
            continue;
          for(j = 0; j < 3; j++)
            b[j] = d >= 2 ? (t->xyz[3 * node + j] >> (d - 2)) : (t->xyz[3 * node + j] << (2 - d));
          k = (b[0] * 4 + b[1]) * 4 + b[2];
          for(j = 0; j < 3; j++)
            {
              const double lo = dom[j] + t->xyz[3 * node + j] * cl, hi = lo + cl;
              if(!used[k] || lo < blo[k][j])
                blo[k][j] = lo;
              if(!used[k] || hi > bhi[k][j])
                bhi[k][j] = hi;
            }
          used[k] = 1;
        }
      T.nbox = 0;
      for(k = 0; k < 64; k++)
        if(used[k])
          {
            for(j = 0; j < 3; j++)
              {
                bc[T.nbox][j] = 0.5 * (blo[k][j] + bhi[k][j]);
                bh[T.nbox][j] = 0.5 * (bhi[k][j] - blo[k][j]) + 1e-9 * dom[6] + margin;   /* rounding slack; drift while the decomposition is kept */
              }
            T.nbox++;
          }
      T.ng = cfg->n_gravs;
      T.periodic = cfg->periodic;
      T.pm = cfg->pmgrid != 0;
      T.use_theta = cfg->err_tol_theta != 0;
      T.box = cfg->box_size;
      T.theta2 = cfg->err_tol_theta * cfg->err_tol_theta;
      T.aold_min = bounds[0];
      T.h_min = bounds[1];
      T.rcut = cfg->rcut;
      /* how far a target's short-range force reaches: the end of the table (tabindex < NTAB: 6 Asmth, forcetree.c:1962-1967) for
       * the reference walk, the sphere of group_reach Asmth (default RCUT) for the group walk -- whichever walk the context is
       * set to; the library refuses a walk that reaches farther than the decomposition assumed (walk_run) */
      T.reach6 = (cfg->walk_mode == NGRAVS_WALK_GROUP ? fmin(6.0, cfg->group_reach > 0 ? cfg->group_reach : NGRAVS_GROUP_REACH) : 6.0) * cfg->asmth;
      for(j = 0; j < 6; j++)
        T.fsoft[j] = cfg->force_softening[j];
      for(j = 0; j < 3; j++)
        T.corner[j] = dom[j];
      T.len = dom[6];
      T.t = t;
      T.sums = node_sums;
      T.bc = bc;
      T.bh = bh;
      T.need = need;
      /* below a node whose leaves are all this task's own (or empty) there is nothing to request */
      for(i = t->nnode - 1; i >= 0; i--)
        if(t->child[i] < 0)
          mine[i] = leaf_owner[t->leaf[i]] == me || node_sums[(size_t)i * cw] < 0.5;
        else
          {
            uint8_t a = 1;
            for(k = 0; k < 8; k++)
              a &= mine[t->child[i] + k];
            mine[i] = a;
          }
      T.mine = mine;
      if(T.nbox > 0 && t->child[0] >= 0)   /* the root is opened by every target inside it */
        for(k = 0; k < 8; k++)
          need_visit(&T, t->child[0] + k);
      for(i = 0; i < t->nleaf; i++)
        if(leaf_owner[i] == me)
          need[i] = 0;   /* own leaves are here already */
    }
  free(mine);
  free(bc);
  free(bh);
  return rc;
}

/* =========================================================================================================================
 *  domain_Decomposition
 * ========================================================================================================================= */
void ngravs_host_plan_free(ngravs_dd_plan *plan)
{
  if(plan)
    {
      ngravs_host_toptree_free(&plan->tree);
      free(plan->leaf_owner);
      free(plan->node_sums);
      plan->leaf_owner = NULL;
      plan->node_sums = NULL;
    }
}

/* The all-reduced per-leaf sums of the top tree `t` -> sums (host, nleaf * cw doubles + 1 status word).  One collective, which
 * a task enters even after a local failure (rc_in != 0, or a failing library call here): with zeros and a raised status word, so
 * that all tasks return the error together. */
static int leaf_sums_round(ngravs_ctx *ctx, const ngravs_comm *cm, const ngravs_toptree *t, int cw, double *sums, int rc_in,
                           ngravs_dd_info *info)
{
  void *dev = NULL;
  int64_t count = 0;
  const int64_t want = (int64_t)t->nleaf * cw;
  int rc = rc_in, rcc;
  if(!rc)
    rc = ngravs_dd_set_toptree(ctx, t->nnode, t->child);
  if(!rc)
    rc = ngravs_dd_leaf_sums(ctx, &dev, &count);   /* the library keeps one spare word (zero) behind the table: the status */
  if(!rc && count != want + 1)
    rc = NGRAVS_ERR_STATE;
  if(cm->allreduce_dev)
    {
      if(rc)
        {
          /* take part with zeros and status 1 from a scratch buffer of the library */
          memset(sums, 0, sizeof(double) * (size_t)want);
          sums[want] = 1.0;
          if(ngravs_dd_recv_buffer(ctx, (want + 1) * (int64_t)sizeof(double) / NGRAVS_DD_MAX_RECORD_BYTES + 1, &dev) ||
             ngravs_memcpy(ctx, dev, sums, (int64_t)sizeof(double) * (want + 1), 1))
            return status_of(rc);   /* not even that: the other tasks are left waiting */
        }
      rcc = cm->allreduce_dev(cm->user, dev, want + 1, NGRAVS_T_F64, NGRAVS_OP_SUM);
      info->collectives++;
      if(!rcc)
        rcc = ngravs_memcpy(ctx, sums, dev, (int64_t)sizeof(double) * (want + 1), 2);
    }
  else
    {
      if(!rc)
        rc = ngravs_memcpy(ctx, sums, dev, (int64_t)sizeof(double) * want, 2);
      if(rc)
        memset(sums, 0, sizeof(double) * (size_t)want);
      sums[want] = rc ? 1.0 : 0.0;
      rcc = cm->allreduce(cm->user, sums, want + 1, NGRAVS_T_F64, NGRAVS_OP_SUM);
      info->collectives++;
    }
  if(rc || rcc)
    return status_of(rc ? rc : rcc);
  return sums[want] > 0.5 ? NGRAVS_ERR_STATE : 0;
}

/* domain_findExtent + domain_determineTopTree + domain_sumCost + the cut (domain.c:882-924, 933-1138, 823-877, 347-544) */
static int domain_owners(ngravs_ctx *ctx, const ngravs_comm *cm, double leaf_max, double paf, ngravs_dd_plan *plan, ngravs_dd_info *info,
                         int own_rows);

int ngravs_host_domain_owners(ngravs_ctx *ctx, const ngravs_comm *cm, double leaf_max, double paf, ngravs_dd_plan *plan, ngravs_dd_info *info)
{
  return domain_owners(ctx, cm, leaf_max, paf, plan, info, 0);
}

/* own_rows: the rows are the library's (ngravs_host_domain_decomposition: named by ID) -- it may put them in Peano order first */
static int domain_owners(ngravs_ctx *ctx, const ngravs_comm *cm, double leaf_max, double paf, ngravs_dd_plan *plan, ngravs_dd_info *info,
                         int own_rows)
{
  ngravs_config_t cfg;
  ngravs_toptree tree, next;
  double lo[3], hi[3], bounds[2] = {1e300, 1e300}, e[10], total = 0, wtot = 0, wmax = 0, cmax = 0, *sums = NULL, *lcount = NULL, *lwork,
         *twork = NULL, thresh, n_own;
  int32_t *owner = NULL, i;
  int r, k, q, cw, rc = 0, W, round, have_tree = 0;
  ngravs_dd_info local;
  if(!ctx || !cm || !plan || cm->size < 1 || cm->size > 64 || cm->rank < 0 || cm->rank >= cm->size)
    return NGRAVS_ERR_ARG;
  W = cm->size;
  if(!info)
    info = &local;
  memset(info, 0, sizeof(*info));
  memset(plan, 0, sizeof(*plan));
  memset(&tree, 0, sizeof(tree));
  memset(&cfg, 0, sizeof(cfg));
  info->seconds[0] = -wall_now();
  rc = ngravs_get_config(ctx, &cfg);
  cw = NGRAVS_TOP_CW(cfg.n_gravs > 0 ? cfg.n_gravs : 1);
  if(!rc && own_rows)
    {
      rc = ngravs_dd_peano_order(ctx, 0);   /* peano_hilbert_order() of P[] (domain.c:146), when the rows have left that order */
      rc = rc > 0 ? 0 : rc;
    }
  if(!rc)
    rc = ngravs_dd_local_extent(ctx, lo, hi);
  if(!rc)
    rc = ngravs_dd_target_bounds(ctx, bounds);
  n_own = (double)ngravs_dd_num_local(ctx);
  /* the top tree of the last decomposition is cloned BEFORE the first collective, so that a task that cannot (out of memory) says
   * so through that collective's status word and all tasks return together */
  if(!rc)
    {
      ngravs_toptree view;
      rc = ngravs_host_toptree_borrow(ctx, &view);   /* the library's finished copy of the last tree: levels, coordinates, leaf numbers */
      if(!rc && view.nnode > 0)
        {
          rc = tt_clone(&view, &tree);
          have_tree = !rc;
        }
    }
  /* one collective for both ends of the extent (max(hi) = -min(-hi)), the bounds of the opening tests and the status */
  for(r = 0; r < 3; r++)
    {
      e[r] = rc ? 1e300 : lo[r];
      e[3 + r] = rc ? 1e300 : -hi[r];
    }
  e[6] = bounds[0];
  e[7] = bounds[1];
  e[8] = rc ? -1.0 : 0.0;
  e[9] = -n_own;   /* -> the largest particle count of a task: what the first guess of the top tree is sized with, THE SAME on every task */
  {
    const int rcc = cm->allreduce(cm->user, e, 10, NGRAVS_T_F64, NGRAVS_OP_MIN);
    info->collectives++;
    if(rcc)
      return status_of(rcc);
  }
  if(e[8] < -0.5)
    {
      if(have_tree)
        ngravs_host_toptree_free(&tree);
      return rc ? status_of(rc) : NGRAVS_ERR_STATE;
    }
  for(r = 0; r < 3; r++)
    {
      lo[r] = e[r];
      hi[r] = -e[3 + r];
    }
  plan->bounds[0] = e[6];
  plan->bounds[1] = e[7];
  rc = ngravs_dd_set_extent(ctx, lo, hi);   /* (the same numbers on every task: it fails everywhere or nowhere) */
  /* no tree yet: a first guess, the complete tree whose leaves a uniform filling would leave with about the threshold (a few
   * hundred bytes: the one allocation of this function whose failure is not reported through a collective) */
  if(!rc && !have_tree)
    {
        {
          /* (sized from all-reduced numbers only: a guess from the own count would give tasks with different counts different
           * trees, and the all-reduce of the leaf sums that follows different lengths) */
          const double guess = -e[9] * W, lm = leaf_max > 0 ? leaf_max : fmin(NGRAVS_TOPLEAF_MAX, guess / (20.0 * W));
          int lvl = 1;
          while(lvl < 6 && guess / (double)(1ll << (3 * lvl)) > lm)
            lvl++;
          rc = ngravs_host_toptree_init(&tree, lvl);
        }
    }
  if(rc)   /* (set_extent fails everywhere or nowhere; the first guess is the one local failure left) */
    {
      if(have_tree)
        ngravs_host_toptree_free(&tree);
      return status_of(rc);
    }
  for(round = 0; round < 32; round++)
    {
      const int64_t want = (int64_t)tree.nleaf * cw;
      const double t0 = wall_now();
      int unknown;
      free(sums);
      sums = malloc(sizeof(double) * (size_t)(want + 1));
      if(!sums)
        {
          ngravs_host_toptree_free(&tree);
          free(lcount);
          return NGRAVS_ERR_NOMEM;
        }
      rc = leaf_sums_round(ctx, cm, &tree, cw, sums, rc, info);
      info->seconds[2] += wall_now() - t0;
      info->toptree_rounds = round + 1;
      if(rc)
        break;
      /* per leaf: [0] carries the work; the particle count is the sum of the per-type counts */
      free(lcount);
      lcount = malloc(sizeof(double) * 2 * (size_t)tree.nleaf);
      if(!lcount)
        {
          rc = NGRAVS_ERR_NOMEM;
          break;
        }
      total = 0;
      for(i = 0; i < tree.nleaf; i++)
        {
          const double *p = sums + (size_t)i * cw;
          lcount[i] = p[1] + p[2] + p[3] + p[4] + p[5] + p[6];
          lcount[tree.nleaf + i] = p[0];
          total += lcount[i];
        }
      thresh = leaf_max > 0 ? leaf_max : fmin(NGRAVS_TOPLEAF_MAX, total / (20.0 * W));   /* domain.c:1127: TotNumPart / (TOPNODEFACTOR * NTask) */
      if(thresh < 1.0)
        thresh = 1.0;
      unknown = ngravs_host_toptree_adapt(&tree, lcount, thresh, NGRAVS_TOPLEVEL_MAX, &next);
      if(unknown < 0)
        {
          rc = unknown;
          break;
        }
      if(unknown == 0 && (next.nnode == 0 || next.nnode == tree.nnode))
        {
          ngravs_host_toptree_free(&next);
          break;   /* the tree obeys the rule and is the one these sums were taken for */
        }
      ngravs_host_toptree_free(&tree);
      tree = next;   /* new leaves (or merged ones): count again with the new tree */
    }
  if(!rc && round >= 32)
    rc = NGRAVS_ERR_STATE;
  if(!rc)
    {
      owner = malloc(sizeof(int32_t) * (size_t)tree.nleaf);
      twork = malloc(sizeof(double) * 2 * (size_t)W);
      plan->node_sums = calloc((size_t)tree.nnode * cw, sizeof(double));
      if(!owner || !twork || !plan->node_sums)
        rc = NGRAVS_ERR_NOMEM;
    }
  if(!rc)
    {
      lwork = lcount + tree.nleaf;
      if(ngravs_host_split(lcount, lwork, tree.nleaf, W, (paf > 0 ? paf : 1.5) * total / W, owner) &&
         ngravs_host_split(lcount, lwork, tree.nleaf, W, 0.0, owner))
        rc = NGRAVS_ERR_ARG;
    }
  if(!rc)
    {
      lwork = lcount + tree.nleaf;
      for(r = 0; r < 2 * W; r++)
        twork[r] = 0;
      for(i = 0; i < tree.nleaf; i++)
        {
          twork[owner[i]] += lwork[i];
          twork[W + owner[i]] += lcount[i];
        }
      for(r = 0; r < W; r++)
        {
          wtot += twork[r];
          wmax = fmax(wmax, twork[r]);
          cmax = fmax(cmax, twork[W + r]);
        }
      info->work_balance = wtot > 0 ? wmax / (wtot / W) : 1.0;
      info->memory_balance = total > 0 ? cmax / (total / W) : 1.0;
      /* sums of every top node, children in fixed order: the same numbers on every task (force_treeupdate_pseudos adds the
       * top-leaf moments up the ancestor chain, forcetree.c:851-947) */
      for(i = tree.nnode - 1; i >= 0; i--)
        {
          double *dst = plan->node_sums + (size_t)i * cw;
          if(tree.child[i] < 0)
            {
              memcpy(dst, sums + (size_t)tree.leaf[i] * cw, sizeof(double) * (size_t)cw);
              dst[0] = lcount[tree.leaf[i]];
            }
          else
            for(k = 0; k < 8; k++)
              {
                const double *src = plan->node_sums + (size_t)(tree.child[i] + k) * cw;
                for(q = 0; q < cw; q++)
                  dst[q] += src[q];
              }
        }
      plan->tree = tree;
      plan->leaf_owner = owner;
      memset(&tree, 0, sizeof(tree));
      owner = NULL;
      info->n_topnodes = plan->tree.nnode;
      info->n_topleaves = plan->tree.nleaf;
    }
  free(sums);
  free(lcount);
  free(owner);
  free(twork);
  ngravs_host_toptree_free(&tree);
  if(rc)
    ngravs_host_plan_free(plan);
  info->seconds[0] += wall_now();
  return status_of(rc);
}

/* particle migration (domain_exchangeParticles, domain.c:695-795) and the import requests of all tasks in ONE all-gather:
 * every task contributes its send counts (W int64), a status word, and its request bits (one byte per leaf) */
static int migrate_and_request(ngravs_ctx *ctx, const ngravs_comm *cm, const ngravs_dd_plan *plan, int do_migration, int rc_in,
                               const uint8_t *need, uint8_t *allneed, ngravs_dd_info *info)
{
  const int W = cm->size, me = cm->rank;
  const int64_t nleaf = plan->tree.nleaf, head_bytes = (int64_t)sizeof(int64_t) * (W + 1), blk = head_bytes + nleaf;
  char *mine = malloc((size_t)blk), *all = malloc((size_t)blk * (size_t)W);
  int64_t counts[65], sb[65], rb[65], head[65], nrec = 0, nrecv = 0, moving = 0;
  void *rec = NULL, *recvbuf = NULL;
  int r, q, rc = rc_in, rcc, bad = 0;
  if(!mine || !all)
    {
      free(mine);
      free(all);
      return NGRAVS_ERR_NOMEM;   /* the buffers of the collective itself are missing: this task cannot take part */
    }
  memset(counts, 0, sizeof(counts));
  if(!rc && do_migration)
    rc = ngravs_dd_pack(ctx, 0, plan->leaf_owner, W, me, counts, &rec, &nrec);
  for(r = 0; r < W; r++)
    head[r] = rc ? 0 : counts[r];
  head[W] = rc ? 1 : 0;
  memcpy(mine, head, (size_t)head_bytes);
  memcpy(mine + head_bytes, need, (size_t)nleaf);
  rcc = cm->allgather(cm->user, mine, all, blk);
  info->collectives++;
  if(rcc && !rc)
    rc = rcc;
  for(r = 0; r < W && !rcc; r++)
    {
      memcpy(head, all + (size_t)blk * r, (size_t)head_bytes);
      if(head[W])
        bad = 1;
      memcpy(allneed + (size_t)nleaf * r, all + (size_t)blk * r + head_bytes, (size_t)nleaf);
      rb[r] = head[me];
      for(q = 0; q < W; q++)
        if(q != r)
          moving += head[q];
    }
  free(mine);
  free(all);
  if(!rc && bad)
    rc = NGRAVS_ERR_STATE;   /* all tasks see the raised status word: all stop here */
  if(!rc && do_migration)
    {
      const int64_t rbytes = ngravs_dd_record_bytes(ctx, 0);   /* migration records of TreePM runs carry GravPM */
      const double t0 = wall_now();
      for(r = 0; r < W; r++)
        {
          nrecv += rb[r];
          sb[r] = counts[r] * rbytes;
          rb[r] *= rbytes;
          if(r != me)
            info->bytes_migration += (double)sb[r];
        }
      info->n_migrated_in = nrecv;
      if(moving > 0)   /* every task sees the same matrix: all skip the exchange together when nothing moves */
        {
          rc = ngravs_dd_recv_buffer(ctx, nrecv, &recvbuf);
          if(!rc)
            {
              rc = exchange(ctx, cm, rec, sb, recvbuf, rb);
              info->collectives++;
            }
        }
      if(!rc)
        rc = ngravs_dd_apply_migration(ctx, recvbuf, nrecv);
      info->seconds[1] += wall_now() - t0;
    }
  return status_of(rc);
}

/* Top-leaf moments + tree-node import + local Peano order: the second half of domain_Decomposition() for a task whose own
 * particles are in place (force_exchange_pseudodata / force_treeupdate_pseudos, forcetree.c:766-996, and what replaces the
 * export / import loop of gravtree.c:112-285).  `migrate`: the library's own columns still have to move (library-side
 * migration); 0 when the host has exchanged its P[] itself and handed the result over. */
static int domain_halo(ngravs_ctx *ctx, const ngravs_comm *cm, const ngravs_dd_plan *plan, int migrate, ngravs_dd_info *info)
{
  ngravs_dd_info local;
  ngravs_config_t cfg;
  double dom[8], margin = 0;
  int64_t nleaf, i, sc[65], rcn[65], nrec = 0, nrecv = 0;
  uint8_t *need = NULL, *allneed = NULL, *present = NULL;
  uint64_t *reqmask = NULL;
  void *rec = NULL, *recvbuf = NULL;
  int r, cw, rc = 0, W, me;
  if(!ctx || !cm || !plan || !plan->leaf_owner || !plan->node_sums)
    return NGRAVS_ERR_ARG;
  W = cm->size;
  me = cm->rank;
  if(!info)
    {
      memset(&local, 0, sizeof(local));
      info = &local;
    }
  nleaf = plan->tree.nleaf;
  memset(&cfg, 0, sizeof(cfg));
  rc = ngravs_get_config(ctx, &cfg);
  if(!rc)
    rc = ngravs_get_domain_extent(ctx, dom);
  cw = NGRAVS_TOP_CW(cfg.n_gravs > 0 ? cfg.n_gravs : 1);
  need = calloc((size_t)nleaf, 1);
  allneed = malloc((size_t)nleaf * (size_t)W);
  present = malloc((size_t)nleaf);
  reqmask = calloc((size_t)nleaf, sizeof(uint64_t));
  if(!need || !allneed || !present || !reqmask)
    {
      free(need);
      free(allneed);
      free(present);
      free(reqmask);
      return NGRAVS_ERR_NOMEM;   /* the buffers of the next collective itself are missing: this task cannot take part */
    }
  info->seconds[3] = -wall_now();
  if(!rc)
    rc = ngravs_dd_keep_margin(ctx, &margin);
  if(!rc)
    rc = ngravs_host_import_request_margin(&cfg, dom, &plan->tree, plan->node_sums, plan->leaf_owner, me, plan->bounds, margin, need);
  info->seconds[3] += wall_now();
  info->seconds[4] = -wall_now();
  if(rc)
    memset(need, 0, (size_t)nleaf);
  rc = migrate_and_request(ctx, cm, plan, migrate, rc, need, allneed, info);   /* a failed task reports through its status word */
  if(!rc)
    {
      /* the owners ship every particle of the requested leaves (replaces the export of targets, gravtree.c:195-257); a leaf's
       * owner holds all its particles now, so everybody knows every count: a requester receives from an owner the global
       * counts of the leaves it asked that owner for */
      for(r = 0; r < W; r++)
        rcn[r] = 0;
      for(r = 0; r < W; r++)
        for(i = 0; i < nleaf; i++)
          if(allneed[(size_t)r * nleaf + i] && plan->leaf_owner[i] != r)
            {
              if(plan->leaf_owner[i] == me)
                reqmask[i] |= 1ull << r;
              if(r == me)
                rcn[plan->leaf_owner[i]] += (int64_t)(plan->node_sums[(size_t)plan->tree.node_of_leaf[i] * cw] + 0.5);
            }
      rc = ngravs_dd_pack_leaves(ctx, reqmask, W, me, sc, &rec, &nrec);
      if(!rc)
        {
          for(r = 0; r < W; r++)
            {
              nrecv += rcn[r];
              sc[r] *= NGRAVS_DD_RECORD_BYTES;
              rcn[r] *= NGRAVS_DD_RECORD_BYTES;
              if(r != me)
                info->bytes_halo += (double)sc[r];
            }
          rc = ngravs_dd_recv_buffer(ctx, nrecv, &recvbuf);
        }
      info->seconds[4] += wall_now();
      info->seconds[5] = -wall_now();
      if(!rc)
        {
          rc = exchange(ctx, cm, rec, sc, recvbuf, rcn);
          info->collectives++;
        }
      if(!rc)
        rc = ngravs_dd_set_halo(ctx, recvbuf, nrecv);
      info->seconds[5] += wall_now();
      info->n_halo = nrecv;
    }
  if(!rc)
    {
      info->seconds[6] = -wall_now();
      for(i = 0; i < nleaf; i++)
        present[i] = (plan->leaf_owner[i] == me || need[i]) ? 1 : 0;
      rc = ngravs_dd_set_top(ctx, plan->node_sums, present);
      info->seconds[6] += wall_now();
    }
  if(!rc)
    {
      info->n_local = ngravs_dd_num_local(ctx);
      info->seconds[7] = -wall_now();
      rc = ngravs_domain_decomposition(ctx);
      info->seconds[7] += wall_now();
    }
  free(need);
  free(allneed);
  free(present);
  free(reqmask);
  return status_of(rc);
}

int ngravs_host_domain_halo(ngravs_ctx *ctx, const ngravs_comm *cm, const ngravs_dd_plan *plan, ngravs_dd_info *info)
{
  return domain_halo(ctx, cm, plan, 0, info);
}

int ngravs_host_domain_decomposition(ngravs_ctx *ctx, const ngravs_comm *cm, double leaf_max, double paf, ngravs_dd_info *info)
{
  ngravs_dd_plan plan;
  ngravs_dd_info local;
  int rc;
  if(!info)
    info = &local;
  CHECK(domain_owners(ctx, cm, leaf_max, paf, &plan, info, 1));
  rc = domain_halo(ctx, cm, &plan, 1, info);
  ngravs_host_plan_free(&plan);
  return status_of(rc);
}

/* ---- a step that keeps the decomposition (domain.c:76 with TreeDomainUpdateFrequency > 0) -----------------------------------------
 * The caller has handed its own rows over again with ngravs_update_particles() (same rows, drifted positions).  Two collectives:
 *   (1) all-to-all-v: the owners ship the drifted particles of the leaves that were requested at the decomposition (same records,
 *       same order) -> the imported copies take their new positions; then the tree is refit (ngravs_force_update_tree)
 *   (2) all-reduce (on the device if the communicator can): per leaf the sums of the own particles by the membership of the
 *       decomposition + the grown side of the leaf's cell in its owner's tree + one status word -> global moments and sides of the
 *       top nodes (force_update_pseudoparticles forcetree.c:753, force_update_node_len_toptree :1096-1122)
 * A task that fails locally before (2) reports through its status word; (1) has none (the counts of the exchange are fixed by the
 * decomposition: a task that cannot pack cannot take part).  info (may be NULL): collectives, seconds[4] pack, [5] exchange +
 * refresh + refit, [2] leaf sums + all-reduce, [6] top update. */
int ngravs_host_kept_step(ngravs_ctx *ctx, const ngravs_comm *cm, ngravs_dd_info *info)
{
  ngravs_dd_info local;
  ngravs_config_t cfg;
  ngravs_toptree tree;
  const int32_t *owner = NULL;
  const uint8_t *present = NULL;
  const double *old_sums = NULL;
  int32_t me = -1, W = 0;
  int64_t sc[65], rcn[65], nrec = 0, nrecv = 0, count = 0, want, i;
  double *sums = NULL, *node_sums = NULL, *leaf_len = NULL;
  void *rec = NULL, *recvbuf = NULL, *dev = NULL;
  int r, k, q, cw, cwk, rc, rcc = 0;
  if(!ctx || !cm)
    return NGRAVS_ERR_ARG;
  if(!info)
    info = &local;
  memset(info, 0, sizeof(*info));
  CHECK(ngravs_dd_get_kept(ctx, &me, &W, &owner, &present, &old_sums));
  if(W != cm->size || me != cm->rank)
    return NGRAVS_ERR_STATE;
  memset(&cfg, 0, sizeof(cfg));
  CHECK(ngravs_get_config(ctx, &cfg));
  CHECK(ngravs_host_toptree_borrow(ctx, &tree));
  cw = NGRAVS_TOP_CW(cfg.n_gravs > 0 ? cfg.n_gravs : 1);
  cwk = cw + 1;
  /* ---- (1) the imported copies follow their originals */
  info->seconds[4] = -wall_now();
  rc = ngravs_dd_pack_leaves_kept(ctx, sc, &rec, &nrec);
  for(r = 0; r < W; r++)
    rcn[r] = 0;
  for(i = 0; i < tree.nleaf; i++)
    if(present[i] && owner[i] != me)   /* requested then, received now: the global count of the leaf (nothing has migrated) */
      rcn[owner[i]] += (int64_t)(old_sums[(size_t)tree.node_of_leaf[i] * cw] + 0.5);
  for(r = 0; r < W; r++)
    {
      nrecv += rcn[r];
      sc[r] = rc ? 0 : sc[r] * NGRAVS_DD_RECORD_BYTES;
      rcn[r] *= NGRAVS_DD_RECORD_BYTES;
    }
  if(!rc)
    rc = ngravs_dd_recv_buffer(ctx, nrecv, &recvbuf);
  info->seconds[4] += wall_now();
  if(rc)
    return status_of(rc);   /* (see above: the exchange has no status word) */
  info->seconds[5] = -wall_now();
  rc = exchange(ctx, cm, rec, sc, recvbuf, rcn);
  info->collectives++;
  if(!rc)
    rc = ngravs_dd_refresh_halo(ctx, recvbuf, nrecv);
  if(!rc)
    rc = ngravs_force_update_tree(ctx);
  info->seconds[5] += wall_now();
  /* ---- (2) the top of the tree from the sums of all tasks */
  info->seconds[2] = -wall_now();
  want = (int64_t)tree.nleaf * cwk;
  sums = malloc(sizeof(double) * (size_t)(want + 1));
  node_sums = calloc((size_t)tree.nnode * (size_t)cw, sizeof(double));
  leaf_len = malloc(sizeof(double) * (size_t)(tree.nleaf > 0 ? tree.nleaf : 1));
  if(!rc && (!sums || !node_sums || !leaf_len))
    rc = NGRAVS_ERR_NOMEM;
  if(!rc)
    rc = ngravs_dd_leaf_sums_kept(ctx, &dev, &count);
  if(!rc && count != want + 1)
    rc = NGRAVS_ERR_STATE;
  if(cm->allreduce_dev && !rc)
    {
      rcc = cm->allreduce_dev(cm->user, dev, want + 1, NGRAVS_T_F64, NGRAVS_OP_SUM);
      info->collectives++;
      if(!rcc)
        rcc = ngravs_memcpy(ctx, sums, dev, (int64_t)sizeof(double) * (want + 1), 2);
    }
  else if(sums)
    {
      /* host all-reduce; a task that failed takes part with zeros and raises the status word */
      if(!rc)
        rc = ngravs_memcpy(ctx, sums, dev, (int64_t)sizeof(double) * want, 2);
      if(rc)
        memset(sums, 0, sizeof(double) * (size_t)want);
      sums[want] = rc ? 1.0 : 0.0;
      rcc = cm->allreduce(cm->user, sums, want + 1, NGRAVS_T_F64, NGRAVS_OP_SUM);
      info->collectives++;
    }
  info->seconds[2] += wall_now();
  if(!rc && !rcc && sums[want] > 0.5)
    rc = NGRAVS_ERR_STATE;   /* another task failed */
  if(!rc && !rcc)
    {
      info->seconds[6] = -wall_now();
      for(i = tree.nnode - 1; i >= 0; i--)
        {
          double *dst = node_sums + (size_t)i * cw;
          if(tree.child[i] < 0)
            {
              const double *p = sums + (size_t)tree.leaf[i] * cwk;
              memcpy(dst, p, sizeof(double) * (size_t)cw);
              dst[0] = p[1] + p[2] + p[3] + p[4] + p[5] + p[6];   /* [0] of a node: its particle count (as ngravs_dd_set_top takes it) */
              leaf_len[tree.leaf[i]] = p[cw];
            }
          else
            for(k = 0; k < 8; k++)
              {
                const double *src = node_sums + (size_t)(tree.child[i] + k) * cw;
                for(q = 0; q < cw; q++)
                  dst[q] += src[q];
              }
        }
      rc = ngravs_dd_update_top(ctx, node_sums, leaf_len);
      info->seconds[6] += wall_now();
    }
  info->n_local = ngravs_dd_num_local(ctx);
  info->n_halo = nrecv;
  info->n_topnodes = tree.nnode;
  info->n_topleaves = tree.nleaf;
  free(sums);
  free(node_sums);
  free(leaf_len);
  return status_of(rc ? rc : rcc);
}

/* ---- the communicator proves itself (include/ngravs_comm_selftest.h) ------------------------------------------------------------ */
#include "ngravs_comm_selftest.h"
static void *st_alloc(void *user, size_t bytes)
{
  void *p = NULL;
  return ngravs_device_alloc(user, &p, (int64_t)bytes) ? NULL : p;
}
static void st_release(void *user, void *p) { (void)ngravs_device_free(user, p); }
static int st_upload(void *user, void *dev, const void *host, size_t bytes) { return ngravs_memcpy(user, dev, host, (int64_t)bytes, 1); }
static int st_download(void *user, void *host, const void *dev, size_t bytes) { return ngravs_memcpy(user, host, dev, (int64_t)bytes, 2); }
static int st_fill(void *user, void *dev, int byte, size_t bytes)
{
  void *h = malloc(bytes ? bytes : 1);
  int rc;
  if(!h)
    return 1;
  memset(h, byte, bytes);
  rc = ngravs_memcpy(user, dev, h, (int64_t)bytes, 1);
  free(h);
  return rc;
}

int ngravs_host_comm_selftest(ngravs_ctx *ctx, const ngravs_comm *cm, int fail_stage, char *why, int why_len)
{
  ngravs_selftest_mem mem = {st_alloc, st_release, st_upload, st_download, st_fill, ctx};
  return ngravs_comm_selftest_run(cm, (cm && cm->device_buffers && ctx) ? &mem : NULL, fail_stage, why, why_len);
}

/* ---- pmforce_periodic on the slab-decomposed mesh -------------------------------------------------------------------------- */
/* per host thread (host_shim_test runs two tasks as two threads); last call: [0] deposit + bounding boxes, then per stage s:
 * [1+3s] pack, [2+3s] exchange, [3+3s] unpack */
static __thread double pm_seconds[13];

void ngravs_host_pm_seconds(double out[13])
{
  int i;
  for(i = 0; i < 13; i++)
    out[i] = pm_seconds[i];
}

int ngravs_host_pmforce_periodic(ngravs_ctx *ctx, const ngravs_comm *cm)
{
  int32_t bb[7], *all;
  int64_t *sc, *rc_;
  void *send = NULL, *recv = NULL;
  int stage, r, rc, rcc, W, bad = 0;
  if(!ctx || !cm || cm->size < 1)
    return NGRAVS_ERR_ARG;
  W = cm->size;
  all = malloc(sizeof(int32_t) * 7 * (size_t)W);
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
  memset(bb, 0, sizeof(bb));
  rc = ngravs_pm_slab_begin(ctx, cm->rank, W, bb);
  bb[6] = rc ? 1 : 0;   /* the status travels with the boxes: all tasks stop together */
  rcc = cm->allgather(cm->user, bb, all, (int64_t)sizeof(bb));   /* meshmin/meshmax lists, pm_periodic.c:285-291 */
  if(rcc && !rc)
    rc = rcc;
  for(r = 0; r < W && !rcc; r++)
    {
      bad |= all[7 * r + 6];
      memmove(all + 6 * r, all + 7 * r, sizeof(int32_t) * 6);
    }
  if(!rc && bad)
    rc = NGRAVS_ERR_STATE;
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
  return status_of(rc);
}

/* compute_accelerations(0), gravity part (accel.c:24-58) */
int ngravs_host_compute_accelerations(ngravs_ctx *ctx, const ngravs_comm *cm, int pm_step, ngravs_dd_info *info)
{
  ngravs_config_t cfg;
  CHECK(ngravs_get_config(ctx, &cfg));
  if(pm_step && cfg.pmgrid)
    CHECK(ngravs_discard_grav_pm(ctx));   /* recomputed below: nothing to carry through the decomposition */
  CHECK(ngravs_host_domain_decomposition(ctx, cm, 0.0, 0.0, info));
  if(pm_step && cfg.pmgrid)
    CHECK(ngravs_host_pmforce_periodic(ctx, cm));
  return ngravs_gravity_tree(ctx);
}
