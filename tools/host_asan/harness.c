/* ASAN/UBSAN harness for the pure-C pieces of host/ngravs_host.c: random particle sets -> top tree rounds (adapt), cut (split),
 * import request; no GPU, no communicator. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include "ngravs_host.h"

static unsigned long long rs = 88172645463325252ull;
static double rnd(void) { rs ^= rs << 13; rs ^= rs >> 7; rs ^= rs << 17; return (double)(rs >> 11) / 9007199254740992.0; }

/* leaf of a point: descend the tree by integer coordinates (the device does it by Peano digits; any consistent child order would
 * do for this harness as long as counts are consistent: use the tree's own xyz tables) */
static int leaf_of(const ngravs_toptree *t, const double p[3])
{
  int node = 0;
  while(t->child[node] >= 0)
    {
      const int lvl = t->level[node] + 1;
      int k, found = -1;
      const int cx = (int)(p[0] * (1 << lvl)), cy = (int)(p[1] * (1 << lvl)), cz = (int)(p[2] * (1 << lvl));
      for(k = 0; k < 8; k++)
        {
          const int c = t->child[node] + k;
          if(t->xyz[3 * c] == cx && t->xyz[3 * c + 1] == cy && t->xyz[3 * c + 2] == cz)
            found = c;
        }
      if(found < 0)
        {
          printf("no child holds the point\n");
          exit(2);
        }
      node = found;
    }
  return t->leaf[node];
}

int main(void)
{
  int trial;
  for(trial = 0; trial < 40; trial++)
    {
      const int n = 200 + (int)(rnd() * 30000), W = 1 + (int)(rnd() * 8), ng = 1 + trial % 3, cw = NGRAVS_TOP_CW(ng);
      const double thresh = 5 + rnd() * n / (4.0 * W);
      double *pos = malloc(sizeof(double) * 3 * n);
      ngravs_toptree t, next;
      int i, rounds = 0, rc;
      const int clump = trial % 4 == 1;
      for(i = 0; i < n; i++)
        {
          int j;
          for(j = 0; j < 3; j++)
            {
              double v = clump && i > n / 10 ? 0.37 + 1e-3 * rnd() : rnd();
              pos[3 * i + j] = v >= 1.0 ? 0.999999 : v;
            }
        }
      if(ngravs_host_toptree_init(&t, trial % 3))
        return 3;
      for(;;)
        {
          double *cnt = calloc((size_t)t.nleaf, sizeof(double));
          int unknown;
          for(i = 0; i < n; i++)
            cnt[leaf_of(&t, pos + 3 * i)] += 1.0;
          unknown = ngravs_host_toptree_adapt(&t, cnt, thresh, trial % 5 == 2 ? 4 : NGRAVS_TOPLEVEL_MAX, &next);
          free(cnt);
          if(unknown < 0)
            return 4;
          rounds++;
          if(unknown == 0 && (next.nnode == 0 || next.nnode == t.nnode))
            {
              ngravs_host_toptree_free(&next);
              break;
            }
          ngravs_host_toptree_free(&t);
          t = next;
          if(rounds > 40)
            return 5;
        }
      {
        /* the cut + the import request of every task */
        double *cnt = calloc((size_t)t.nleaf, sizeof(double)), *work = calloc((size_t)t.nleaf, sizeof(double));
        double *sums = calloc((size_t)t.nnode * cw, sizeof(double)), dom[8] = {0, 0, 0, 0.5, 0.5, 0.5, 1.0, 262144.0}, bounds[2] = {1e-3, 1e-3};
        int32_t *owner = malloc(sizeof(int32_t) * (size_t)t.nleaf);
        uint8_t *need = malloc((size_t)t.nleaf);
        ngravs_config_t cfg;
        ngravs_toptree copy;
        int me;
        for(i = 0; i < n; i++)
          {
            const int l = leaf_of(&t, pos + 3 * i);
            double *q = sums + (size_t)t.node_of_leaf[l] * cw;
            cnt[l] += 1.0;
            work[l] += 1.0 + 300.0 * rnd();
            q[0] += 1.0;
            q[1 + 1] += 1.0;
            q[7] += 1.0 / n;
            q[8] += pos[3 * i] / n;
            q[9] += pos[3 * i + 1] / n;
            q[10] += pos[3 * i + 2] / n;
          }
        for(i = t.nnode - 1; i >= 0; i--)
          if(t.child[i] >= 0)
            {
              int k, q;
              for(k = 0; k < 8; k++)
                for(q = 0; q < cw; q++)
                  sums[(size_t)i * cw + q] += sums[(size_t)(t.child[i] + k) * cw + q];
            }
        rc = ngravs_host_split(cnt, work, t.nleaf, W, 1.5 * n / W, owner);
        if(rc)
          rc = ngravs_host_split(cnt, work, t.nleaf, W, 0.0, owner);
        if(rc && t.nleaf >= W)
          return 6;
        memset(&cfg, 0, sizeof(cfg));
        cfg.n_gravs = ng;
        cfg.periodic = trial % 2;
        cfg.box_size = 1.0;
        cfg.pmgrid = trial % 2 ? 32 : 0;
        cfg.asmth = 1.25 / 32;
        cfg.rcut = 4.5 * cfg.asmth;
        cfg.err_tol_theta = trial % 3 ? 0.5 : 0.0;
        cfg.err_tol_force_acc = 0.005;
        for(i = 0; i < 6; i++)
          cfg.force_softening[i] = 1e-3;
        if(!rc)
          for(me = 0; me < W; me++)
            if(ngravs_host_import_request(&cfg, dom, &t, sums, owner, me, bounds, need))
              return 7;
        /* a tree from its child table alone */
        if(ngravs_host_toptree_from_children(&copy, t.child, t.nnode) || copy.nleaf != t.nleaf || copy.depth != t.depth)
          return 8;
        ngravs_host_toptree_free(&copy);
        printf("trial %2d: n %5d W %d thresh %7.1f: %5d leaves / %5d nodes, depth %d, %d rounds%s\n", trial, n, W, thresh, t.nleaf, t.nnode, t.depth,
               rounds, rc ? " (fewer leaves than tasks)" : "");
        free(cnt);
        free(work);
        free(sums);
        free(owner);
        free(need);
      }
      ngravs_host_toptree_free(&t);
      free(pos);
    }
  printf("OK\n");
  return 0;
}
