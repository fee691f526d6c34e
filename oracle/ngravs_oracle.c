/* ngravs_oracle.c -- CPU restatement of the reference's gravity path.  TEST INFRASTRUCTURE ONLY
 * (see ngravs_oracle.h for who may use it and how it is pinned).  Each function cites the
 * reference file:line it follows; the arithmetic (operation order, comparisons, truncations) is
 * kept as in the reference so that results agree to summation-order noise, but the code is
 * written from the algorithm, over flat arrays instead of the reference's globals.
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif
#include "ngravs_oracle.h"

#define MAXG NGRAVS_MAX_GRAVS
#define NTAB NGRAVS_NTAB
#define PH_BITS NGRAVS_BITS_PER_DIMENSION
#define TOPNODEFACTOR 20.0 /* domain.c:29 */

/* ------------------------------------------------------------------------------------------
 * Peano-Hilbert key, in the reference's own formulation: orientation table + quarter-turn maps
 * (peano.c:300-345 data, :356-398 loop).  Independent of include/ngravs_peano.h on purpose.
 * ------------------------------------------------------------------------------------------ */
static const char *ph_orient[24] = {
    "07163425", "74650312", "43527061", "30214756", "10672354", "03741265", "32450176", "21563047",
    "61705243", "12036574", "25341607", "56472130", "76014532", "65127403", "54236710", "47305621",
    "67541023", "70436152", "01327645", "16250734", "23105467", "34072516", "45763201", "52614370"};
static const int ph_turnx[24] = {4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 0, 1, 2, 3, 17, 18, 19, 16, 23, 20, 21, 22};
static const int ph_turny[24] = {1, 2, 3, 0, 16, 17, 18, 19, 11, 8, 9, 10, 22, 23, 20, 21, 14, 15, 12, 13, 4, 5, 6, 7};
static const int ph_nx[8] = {3, 0, 0, 2, 2, 0, 0, 1};
static const int ph_ny[8] = {0, 1, 1, 2, 2, 3, 3, 0};
static const int ph_sense[8] = {-1, -1, -1, +1, +1, -1, -1, -1};

int64_t orc_peano_hilbert_key(int x, int y, int z, int bits)
{
  int64_t key = 0;
  int rot = 0, sense = 1;
  for(int mask = 1 << (bits - 1); mask > 0; mask >>= 1)
    {
      int oct = ((x & mask) ? 4 : 0) + ((y & mask) ? 2 : 0) + ((z & mask) ? 1 : 0);
      int q = ph_orient[rot][oct] - '0';
      key = (key << 3) + (sense == 1 ? q : 7 - q);
      sense *= ph_sense[q];
      for(int k = 0; k < ph_nx[q]; k++)
        rot = ph_turnx[rot];
      for(int k = 0; k < ph_ny[q]; k++)
        rot = ph_turny[rot];
    }
  return key;
}

/* domain_findExtent (domain.c:882-924): dom = corner[3], center[3], len, fac */
void orc_domain_extent(const double *pos, int64_t n, double dom[8])
{
  double lo[3] = {1e37, 1e37, 1e37}, hi[3] = {-1e37, -1e37, -1e37};
  for(int64_t i = 0; i < n; i++)
    for(int j = 0; j < 3; j++)
      {
        if(lo[j] > pos[3 * i + j])
          lo[j] = pos[3 * i + j];
        if(hi[j] < pos[3 * i + j])
          hi[j] = pos[3 * i + j];
      }
  double len = 0;
  for(int j = 0; j < 3; j++)
    if(hi[j] - lo[j] > len)
      len = hi[j] - lo[j];
  len *= 1.001;
  for(int j = 0; j < 3; j++)
    {
      dom[3 + j] = 0.5 * (lo[j] + hi[j]);
      dom[j] = 0.5 * (lo[j] + hi[j]) - 0.5 * len;
    }
  dom[6] = len;
  dom[7] = 1.0 / len * (double)(((int64_t)1) << PH_BITS);
}

/* domain.c:938-944: the double products are truncated to int by the call */
void orc_keys(const double *pos, int64_t n, const double dom[8], int64_t *keys)
{
  for(int64_t i = 0; i < n; i++)
    keys[i] = orc_peano_hilbert_key((int)((pos[3 * i] - dom[0]) * dom[7]), (int)((pos[3 * i + 1] - dom[1]) * dom[7]),
                                    (int)((pos[3 * i + 2] - dom[2]) * dom[7]), PH_BITS);
}

/* peano_hilbert_order (peano.c:36-185): species-major, then key; ties keep qsort's freedom, we
 * break them by original index (any tie order is valid in the reference) */
typedef struct
{
  int64_t key;
  int32_t idx, grav;
} ord_t;
static int ord_cmp(const void *a, const void *b)
{
  const ord_t *p = a, *q = b;
  if(p->grav != q->grav)
    return p->grav < q->grav ? -1 : 1;
  if(p->key != q->key)
    return p->key < q->key ? -1 : 1;
  return p->idx < q->idx ? -1 : (p->idx > q->idx);
}
void orc_peano_order(const ngravs_config_t *cfg, const int64_t *keys, const int32_t *type, int64_t n, int32_t *order)
{
  ord_t *o = malloc(sizeof(ord_t) * (size_t)n);
  for(int64_t i = 0; i < n; i++)
    {
      o[i].key = keys[i];
      o[i].idx = (int32_t)i;
      o[i].grav = cfg->type_to_grav[type[i]];
    }
  qsort(o, (size_t)n, sizeof(ord_t), ord_cmp);
  for(int64_t i = 0; i < n; i++)
    order[i] = o[i].idx;
  free(o);
}

/* ------------------------------------------------------------------------------------------
 * Top-level (domain) tree in key space for NTask = 1 (domain.c:933-1138).  With one task the
 * "global" split (threshold TotNumPart/(20*NTask)) over the list of local leaves (threshold
 * TotNumPart/(20*NTask^2)) reproduces the local tree, so only the local split is restated.
 * ------------------------------------------------------------------------------------------ */
typedef struct
{
  int daughter, leaf;
  int64_t size, startkey, count, pstart;
} topnode_t;
typedef struct
{
  topnode_t *nd;
  int n, cap, nleaves;
} toptree_t;

static int key_cmp(const void *a, const void *b)
{
  int64_t x = *(const int64_t *)a, y = *(const int64_t *)b;
  return x < y ? -1 : (x > y);
}
static void top_split(toptree_t *tt, const int64_t *ks, int node, int64_t startkey, double limit)
{
  if(tt->nd[node].size < 8)
    return;
  if(tt->n + 8 > tt->cap)
    {
      tt->cap *= 2;
      tt->nd = realloc(tt->nd, sizeof(topnode_t) * (size_t)tt->cap);
    }
  int d0 = tt->n;
  tt->nd[node].daughter = d0;
  for(int i = 0; i < 8; i++)
    {
      topnode_t *s = &tt->nd[d0 + i];
      s->size = tt->nd[node].size / 8;
      s->count = 0;
      s->daughter = -1;
      s->leaf = -1;
      s->startkey = startkey + i * s->size;
      s->pstart = tt->nd[node].pstart;
    }
  tt->n += 8;
  for(int64_t p = tt->nd[node].pstart; p < tt->nd[node].pstart + tt->nd[node].count; p++)
    {
      int bin = (int)((ks[p] - startkey) / (tt->nd[node].size / 8));
      topnode_t *s = &tt->nd[d0 + bin];
      if(s->count == 0)
        s->pstart = p;
      s->count++;
    }
  for(int i = 0; i < 8; i++)
    if((double)tt->nd[d0 + i].count > limit)
      top_split(tt, ks, d0 + i, tt->nd[d0 + i].startkey, limit);
}
/* domain_walktoptree (domain.c): number the leaves depth-first in daughter order */
static void top_number(toptree_t *tt, int no)
{
  if(tt->nd[no].daughter == -1)
    tt->nd[no].leaf = tt->nleaves++;
  else
    for(int i = 0; i < 8; i++)
      top_number(tt, tt->nd[no].daughter + i);
}
static toptree_t *toptree_make(const int64_t *keys, int64_t n)
{
  int64_t *ks = malloc(sizeof(int64_t) * (size_t)(n > 0 ? n : 1));
  memcpy(ks, keys, sizeof(int64_t) * (size_t)n);
  qsort(ks, (size_t)n, sizeof(int64_t), key_cmp);
  toptree_t *tt = malloc(sizeof(toptree_t));
  tt->cap = 1024;
  tt->nd = malloc(sizeof(topnode_t) * (size_t)tt->cap);
  tt->n = 1;
  tt->nleaves = 0;
  tt->nd[0].daughter = -1;
  tt->nd[0].leaf = -1;
  tt->nd[0].size = ((int64_t)1) << (3 * PH_BITS);
  tt->nd[0].startkey = 0;
  tt->nd[0].count = n;
  tt->nd[0].pstart = 0;
  top_split(tt, ks, 0, 0, (double)n / (TOPNODEFACTOR * 1 * 1));
  top_number(tt, 0);
  free(ks);
  return tt;
}
int orc_toptree_count(const int64_t *keys, int64_t n, int *ntopleaves)
{
  toptree_t *tt = toptree_make(keys, n);
  int r = tt->n;
  if(ntopleaves)
    *ntopleaves = tt->nleaves;
  free(tt->nd);
  free(tt);
  return r;
}

/* ------------------------------------------------------------------------------------------
 * Tree.  Index spaces as in the reference: [0,maxpart) particles, [maxpart,maxpart+maxnodes)
 * internal nodes (forcetree.c:3216-3217).  No pseudo-particles (NTask = 1).
 * ------------------------------------------------------------------------------------------ */
struct orc_tree
{
  int64_t n, maxpart, maxnodes, numnodes;
  int ng, ntopleaves;
  const double *pos, *mass;
  const int32_t *type;
  double *len, *center;       /* [maxnodes], [maxnodes][3]            */
  int32_t *suns;              /* [maxnodes][8] (build-time children)  */
  double *s, *nmass;          /* [maxnodes][3][ng], [maxnodes][ng]    */
  int64_t *npart;             /* [maxnodes][ng]: Nparticles[] of NGRAVS_ACCUMULATOR (allvars.h:645-648, forcetree.c:471-626) */
  int32_t *bitflags, *sibling, *nextnode, *father;
  int32_t *pnext, *pfather;   /* Nextnode[], Father[] of particles    */
  int32_t last;
};

static void empty_nodes(orc_tree *t, toptree_t *tt, int *leaf_node, int no, int topnode, int bits, int x, int y,
                        int z, int64_t *nfree)
{
  /* force_create_empty_nodes (forcetree.c:292-336) */
  if(tt->nd[topnode].daughter < 0)
    return;
  for(int i = 0; i < 2; i++)
    for(int j = 0; j < 2; j++)
      for(int k = 0; k < 2; k++)
        {
          int sub = 7 & (int)orc_peano_hilbert_key((x << 1) + i, (y << 1) + j, (z << 1) + k, bits);
          int slot = i + 2 * j + 4 * k;
          int64_t nn = *nfree;
          int64_t a = no - t->maxpart, b = nn - t->maxpart;
          if(b >= t->maxnodes)
            {
              fprintf(stderr, "oracle: out of tree nodes in empty_nodes\n");
              exit(11);
            }
          t->suns[8 * a + slot] = (int32_t)nn;
          t->len[b] = 0.5 * t->len[a];
          t->center[3 * b + 0] = t->center[3 * a + 0] + (2 * i - 1) * 0.25 * t->len[a];
          t->center[3 * b + 1] = t->center[3 * a + 1] + (2 * j - 1) * 0.25 * t->len[a];
          t->center[3 * b + 2] = t->center[3 * a + 2] + (2 * k - 1) * 0.25 * t->len[a];
          for(int q = 0; q < 8; q++)
            t->suns[8 * b + q] = -1;
          int dsub = tt->nd[topnode].daughter + sub;
          if(tt->nd[dsub].daughter == -1)
            leaf_node[tt->nd[dsub].leaf] = (int)nn;
          *nfree = nn + 1;
          t->numnodes++;
          empty_nodes(t, tt, leaf_node, (int)nn, dsub, bits + 1, 2 * x + i, 2 * y + j, 2 * z + k, nfree);
        }
}

static void link_last(orc_tree *t, int no)
{
  /* the "last" threading of forcetree.c:480-491 / :724-741 */
  if(t->last >= 0)
    {
      if(t->last >= t->maxpart)
        t->nextnode[t->last - t->maxpart] = no;
      else
        t->pnext[t->last] = no;
    }
  t->last = no;
}

static void update_node(orc_tree *t, const ngravs_config_t *cfg, int no, int sib, int father)
{
  /* force_update_node_recursive (forcetree.c:451-743), velocities/hmax omitted (not on the path) */
  if(no < t->maxpart)
    {
      link_last(t, no);
      t->pfather[no] = father;
      return;
    }
  int64_t a = no - t->maxpart;
  int ng = t->ng;
  int32_t suns[8];
  for(int j = 0; j < 8; j++)
    suns[j] = t->suns[8 * a + j];
  link_last(t, no);
  double s[3][MAXG], m[MAXG];
  int64_t np[MAXG];
  for(int g = 0; g < ng; g++)
    {
      s[0][g] = s[1][g] = s[2][g] = m[g] = 0;
      np[g] = 0;
    }
  int maxsofttype = 7, diffsoft = 0;
  for(int j = 0; j < 8; j++)
    {
      int p = suns[j];
      if(p < 0)
        continue;
      int nextsib = sib;
      for(int jj = j + 1; jj < 8; jj++)
        if(suns[jj] >= 0)
          {
            nextsib = suns[jj];
            break;
          }
      update_node(t, cfg, p, nextsib, no);
      if(p >= t->maxpart)
        {
          int64_t c = p - t->maxpart;
          for(int g = 0; g < ng; g++)
            {
              np[g] += t->npart[c * ng + g];
              m[g] += t->nmass[c * ng + g];
              s[0][g] += t->nmass[c * ng + g] * t->s[(c * 3 + 0) * ng + g];
              s[1][g] += t->nmass[c * ng + g] * t->s[(c * 3 + 1) * ng + g];
              s[2][g] += t->nmass[c * ng + g] * t->s[(c * 3 + 2) * ng + g];
            }
          int cst = (t->bitflags[c] >> 2) & 7;
          diffsoft |= (t->bitflags[c] >> 5) & 1;
          if(maxsofttype == 7)
            maxsofttype = cst;
          else if(cst != 7)
            {
              if(cfg->force_softening[cst] > cfg->force_softening[maxsofttype])
                {
                  maxsofttype = cst;
                  diffsoft = 1;
                }
              else if(cfg->force_softening[cst] < cfg->force_softening[maxsofttype])
                diffsoft = 1;
            }
        }
      else
        {
          int ty = t->type[p];
          int g = cfg->type_to_grav[ty];
          np[g]++;
          m[g] += t->mass[p];
          s[0][g] += t->mass[p] * t->pos[3 * p + 0];
          s[1][g] += t->mass[p] * t->pos[3 * p + 1];
          s[2][g] += t->mass[p] * t->pos[3 * p + 2];
          if(maxsofttype == 7)
            maxsofttype = ty;
          else
            {
              if(cfg->force_softening[ty] > cfg->force_softening[maxsofttype])
                {
                  maxsofttype = ty;
                  diffsoft = 1;
                }
              else if(cfg->force_softening[ty] < cfg->force_softening[maxsofttype])
                diffsoft = 1;
            }
        }
    }
  for(int g = 0; g < ng; g++)
    {
      if(m[g] > 0)
        {
          s[0][g] /= m[g];
          s[1][g] /= m[g];
          s[2][g] /= m[g];
        }
      else
        {
          s[0][g] = t->center[3 * a + 0];
          s[1][g] = t->center[3 * a + 1];
          s[2][g] = t->center[3 * a + 2];
        }
      t->s[(a * 3 + 0) * ng + g] = s[0][g];
      t->s[(a * 3 + 1) * ng + g] = s[1][g];
      t->s[(a * 3 + 2) * ng + g] = s[2][g];
      t->nmass[a * ng + g] = m[g];
      t->npart[a * ng + g] = np[g];
    }
  t->bitflags[a] = 4 * maxsofttype + 32 * diffsoft;
  t->sibling[a] = sib;
  t->father[a] = father;
}

orc_tree *orc_tree_build(const ngravs_config_t *cfg, const double *pos, const double *mass, const int32_t *type,
                         int64_t n, const double dom[8])
{
  orc_tree *t = calloc(1, sizeof(orc_tree));
  int ng = cfg->n_gravs;
  t->n = n;
  t->ng = ng;
  t->maxpart = n;
  t->pos = pos;
  t->mass = mass;
  t->type = type;
  /* keys + top-level tree exactly as domain_determineTopTree would leave them */
  int64_t *keys = malloc(sizeof(int64_t) * (size_t)n);
  orc_keys(pos, n, dom, keys);
  toptree_t *tt = toptree_make(keys, n);
  t->ntopleaves = tt->nleaves;
  double taf = cfg->tree_alloc_factor > 0 ? cfg->tree_alloc_factor : 0.8;
  t->maxnodes = (int64_t)(taf * n) + tt->n + 64;
  size_t mn = (size_t)t->maxnodes;
  t->len = malloc(sizeof(double) * mn);
  t->center = malloc(sizeof(double) * 3 * mn);
  t->suns = malloc(sizeof(int32_t) * 8 * mn);
  t->s = malloc(sizeof(double) * 3 * ng * mn);
  t->nmass = malloc(sizeof(double) * ng * mn);
  t->npart = malloc(sizeof(int64_t) * ng * mn);
  t->bitflags = malloc(sizeof(int32_t) * mn);
  t->sibling = malloc(sizeof(int32_t) * mn);
  t->nextnode = malloc(sizeof(int32_t) * mn);
  t->father = malloc(sizeof(int32_t) * mn);
  t->pnext = malloc(sizeof(int32_t) * (size_t)(n + 1));
  t->pfather = malloc(sizeof(int32_t) * (size_t)(n + 1));
  int *leaf_node = malloc(sizeof(int) * (size_t)tt->nleaves);

  /* force_treebuild_single (forcetree.c:93-281) */
  int64_t nfree = t->maxpart;
  t->len[0] = dom[6];
  for(int j = 0; j < 3; j++)
    t->center[j] = dom[3 + j];
  for(int j = 0; j < 8; j++)
    t->suns[j] = -1;
  t->numnodes = 1;
  nfree++;
  if(tt->nd[0].daughter < 0)
    leaf_node[tt->nd[0].leaf] = (int)t->maxpart;
  empty_nodes(t, tt, leaf_node, (int)t->maxpart, 0, 1, 0, 0, 0, &nfree);

  int parent = -1, subnode = 0;
  for(int64_t i = 0; i < n; i++)
    {
      double eps = cfg->force_softening[type[i]];
      int no = 0;
      while(tt->nd[no].daughter >= 0)
        no = tt->nd[no].daughter + (int)((keys[i] - tt->nd[no].startkey) / (tt->nd[no].size / 8));
      int th = leaf_node[tt->nd[no].leaf];
      for(;;)
        {
          if(th >= t->maxpart)
            {
              int64_t a = th - t->maxpart;
              subnode = 0;
              if(pos[3 * i + 0] > t->center[3 * a + 0])
                subnode += 1;
              if(pos[3 * i + 1] > t->center[3 * a + 1])
                subnode += 2;
              if(pos[3 * i + 2] > t->center[3 * a + 2])
                subnode += 4;
              int nn = t->suns[8 * a + subnode];
              if(nn >= 0)
                {
                  parent = th;
                  th = nn;
                }
              else
                {
                  t->suns[8 * a + subnode] = (int32_t)i;
                  break;
                }
            }
          else
            {
              /* a leaf holding particle `th`: replace it by a new internal node */
              int64_t pa = parent - t->maxpart, b = nfree - t->maxpart;
              if(b >= t->maxnodes)
                {
                  fprintf(stderr, "oracle: maximum number of tree-nodes reached\n");
                  exit(1);
                }
              t->suns[8 * pa + subnode] = (int32_t)nfree;
              t->len[b] = 0.5 * t->len[pa];
              double lenhalf = 0.25 * t->len[pa];
              t->center[3 * b + 0] = t->center[3 * pa + 0] + ((subnode & 1) ? lenhalf : -lenhalf);
              t->center[3 * b + 1] = t->center[3 * pa + 1] + ((subnode & 2) ? lenhalf : -lenhalf);
              t->center[3 * b + 2] = t->center[3 * pa + 2] + ((subnode & 4) ? lenhalf : -lenhalf);
              for(int q = 0; q < 8; q++)
                t->suns[8 * b + q] = -1;
              subnode = 0;
              if(pos[3 * th + 0] > t->center[3 * b + 0])
                subnode += 1;
              if(pos[3 * th + 1] > t->center[3 * b + 1])
                subnode += 2;
              if(pos[3 * th + 2] > t->center[3 * b + 2])
                subnode += 4;
              if(t->len[b] < 1.0e-3 * eps)
                {
                  /* the reference randomises the subnode here (forcetree.c:225-238, GSL RNG):
                   * unpinned and unreachable for ICs without coincident particles */
                  fprintf(stderr, "oracle: coincident particles (%ld,%d) below 1e-3 softening\n", (long)i, th);
                  exit(2);
                }
              t->suns[8 * b + subnode] = th;
              th = (int)nfree;
              t->numnodes++;
              nfree++;
            }
        }
    }
  /* moments + threading */
  t->last = -1;
  update_node(t, cfg, (int)t->maxpart, -1, -1);
  if(t->last >= t->maxpart)
    t->nextnode[t->last - t->maxpart] = -1;
  else
    t->pnext[t->last] = -1;
  free(leaf_node);
  free(tt->nd);
  free(tt);
  free(keys);
  return t;
}

void orc_tree_free(orc_tree *t)
{
  if(!t)
    return;
  free(t->len);
  free(t->center);
  free(t->suns);
  free(t->s);
  free(t->nmass);
  free(t->npart);
  free(t->bitflags);
  free(t->sibling);
  free(t->nextnode);
  free(t->father);
  free(t->pnext);
  free(t->pfather);
  free(t);
}
int64_t orc_tree_numnodes(const orc_tree *t) { return t->numnodes; }
int orc_tree_ntopleaves(const orc_tree *t) { return t->ntopleaves; }
void orc_tree_get_node(const orc_tree *t, int64_t i, double *out, int32_t *bitflags)
{
  int ng = t->ng;
  out[0] = t->len[i];
  out[1] = t->center[3 * i];
  out[2] = t->center[3 * i + 1];
  out[3] = t->center[3 * i + 2];
  for(int g = 0; g < ng; g++)
    {
      out[4 + 4 * g + 0] = t->s[(i * 3 + 0) * ng + g];
      out[4 + 4 * g + 1] = t->s[(i * 3 + 1) * ng + g];
      out[4 + 4 * g + 2] = t->s[(i * 3 + 2) * ng + g];
      out[4 + 4 * g + 3] = t->nmass[i * ng + g];
    }
  *bitflags = t->bitflags[i];
}

/* ------------------------------------------------------------------------------------------
 * Force laws (ngravs.c:344-886).  Argument conventions as in the reference:
 *   accel(law, source_mass, r2, r)            returns +|a| (caller divides by r)
 *   spline(id, source_mass, h, r)             returns fac with the 1/r folded in
 *   greens(law, k2, k), normed(law, k2, k)    k in mesh units / table units
 * ------------------------------------------------------------------------------------------ */
/* the BAM family (ngravs.c:495-668): `target` is the target particle's mass, N the number of particles of the source species
 * behind `src` (1 for a particle, Nparticles[g] of a node, allvars.h:645-648).  accel form: |a| * r already folded as the
 * reference does ("r put back in because forcetree.c divides it out"); spline form: fac with 1/r folded in. */
static double bam_eps(const ngravs_config_t *c) { return c->bam_epsilon > 0 ? c->bam_epsilon : 1.31e-6; }
static double bam_eta(const ngravs_config_t *c, int law, double target, double src, long N)
{
  switch(law)
    {
    case NGRAVS_LAW_BAMBAM:
      return 4.0 * M_PI * bam_eps(c) / (target + src / N);   /* :506, :542 */
    case NGRAVS_LAW_SOURCEBAM:
      return 4.0 * M_PI * bam_eps(c) * N / src;              /* :569, :597 */
    default:
      return 4.0 * M_PI * bam_eps(c) / target;               /* :624, :654 */
    }
}
static double bam_accel(const ngravs_config_t *c, int law, double target, double src, double r, long N)
{
  double eta = bam_eta(c, law, target, src, N), rho = 2 * target * src / M_PI;
  double reta = r * eta, reta2 = reta * reta, eta3 = eta * eta * eta;
  if(reta < 0.1)
    return rho * eta3 * (2.0 * r / 3.0 - 4.0 * reta2 * r / 5.0 + 6.0 * reta2 * reta2 * r / 7.0);
  return rho * eta3 * (atan(reta) / (reta2 * eta) - 1.0 / (reta * eta * (1 + reta2)));
}
static double bam_spline(const ngravs_config_t *c, int law, double target, double src, double r, long N)
{
  double eta = bam_eta(c, law, target, src, N), rho = 2 * target * src / M_PI;
  double reta = r * eta, reta2 = reta * reta, eta3 = eta * eta * eta;
  if(reta < 0.1)
    return rho * eta3 * (2.0 / 3.0 - 4.0 * reta2 / 5.0 + 6.0 * reta2 * reta2 / 7.0);
  return rho * eta3 * (atan(reta) / (reta2 * reta) - 1.0 / (reta2 * (1 + reta2)));
}

static double law_accel_tn(const ngravs_config_t *c, int law, double target, double src, double r2, double r, long N);
static double law_accel(const ngravs_config_t *c, int law, double src, double r2, double r)
{
  return law_accel_tn(c, law, 1.0, src, r2, r, 1);
}
static double law_accel_tn(const ngravs_config_t *c, int law, double target, double src, double r2, double r, long N)
{
  double ym;
  switch(law)
    {
    case NGRAVS_LAW_BAMBAM:
    case NGRAVS_LAW_SOURCEBAM:
    case NGRAVS_LAW_TARGETBAM:
      return bam_accel(c, law, target, src, r, N);
    case NGRAVS_LAW_NEWTON:
      return src / r2; /* ngravs.c:351 */
    case NGRAVS_LAW_NEG_NEWTON:
      return -src / r2;
    case NGRAVS_LAW_YUKAWA: /* ngravs.c:856-861 */
      ym = c->yukawa_imass / c->box_size;
      return src * exp(-r * ym) * (ym / r + 1.0 / r2);
    case NGRAVS_LAW_COLOYUK: /* ngravs.c:826 : yukawa + newtonian */
      ym = c->yukawa_imass / c->box_size;
      return src * exp(-r * ym) * (ym / r + 1.0 / r2) + src / r2;
    default:
      return 0.0;
    }
}
static double law_spline_tn(const ngravs_config_t *c, int id, double target, double src, double h, double r, long N);
static double law_spline(int id, double src, double h, double r)
{
  return law_spline_tn(NULL, id, 1.0, src, h, r, 1);
}
static double law_spline_tn(const ngravs_config_t *c, int id, double target, double src, double h, double r, long N)
{
  /* plummer (ngravs.c:420-434), literal constants kept */
  if(id == NGRAVS_SPLINE_NONE)
    return 0.0;
  if(id == NGRAVS_SPLINE_BAMBAM)
    return bam_spline(c, NGRAVS_LAW_BAMBAM, target, src, r, N);
  if(id == NGRAVS_SPLINE_SOURCEBAM)
    return bam_spline(c, NGRAVS_LAW_SOURCEBAM, target, src, r, N);
  if(id == NGRAVS_SPLINE_TARGETBAM)
    return bam_spline(c, NGRAVS_LAW_TARGETBAM, target, src, r, N);
  double h_inv = 1 / h, v;
  r *= h_inv;
  if(r < 0.5)
    v = src * h_inv * h_inv * h_inv * (10.666666666667 + r * r * (32.0 * r - 38.4));
  else
    v = src * h_inv * h_inv * h_inv *
        (21.333333333333 - 48.0 * r + 38.4 * r * r - 10.666666666667 * r * r * r - 0.066666666667 / (r * r * r));
  return id == NGRAVS_SPLINE_NEG_PLUMMER ? -v : v;
}
static double law_greens(const ngravs_config_t *c, double asmth, int law, double k2, double k)
{
  (void)k;
  double ym, a2;
  switch(law)
    {
    case NGRAVS_LAW_NEWTON:
      return 1.0 / k2; /* pgdelta :390 */
    case NGRAVS_LAW_NEG_NEWTON:
      return -1.0 / k2;
    case NGRAVS_LAW_YUKAWA: /* pgyukawa :869-878 */
    case NGRAVS_LAW_COLOYUK:
      ym = c->yukawa_imass / (2 * M_PI);
      a2 = (2 * M_PI) * asmth / c->box_size;
      a2 *= a2;
      return 1.0 / (k2 + ym * ym) * exp(-ym * ym * a2) + (law == NGRAVS_LAW_COLOYUK ? 1.0 / k2 : 0.0);
    default:
      return 0.0;
    }
}
static double law_normed(const ngravs_config_t *c, double asmth, int law, double k2, double k)
{
  (void)k;
  double ym;
  switch(law)
    {
    case NGRAVS_LAW_NEWTON:
      return 1.0;
    case NGRAVS_LAW_NEG_NEWTON:
      return -1.0; /* not defined by the reference (no normed_neg_pgdelta); sign by analogy */
    case NGRAVS_LAW_YUKAWA: /* normed_pgyukawa :880-885 with gridKtoNormK (ngravs_core.c:27-35) */
    case NGRAVS_LAW_COLOYUK:
      ym = 4 * M_PI * asmth * (c->yukawa_imass / (2 * M_PI)) / c->box_size;
      return k2 / (k2 + ym * ym) * exp(-ym * ym * 0.25) + (law == NGRAVS_LAW_COLOYUK ? 1.0 : 0.0);
    default:
      return 0.0;
    }
}
static double cfg_asmth(const ngravs_config_t *c)
{
  return c->asmth > 0 ? c->asmth : NGRAVS_ASMTH * c->box_size / c->pmgrid; /* pm_periodic.c:59 */
}
static double cfg_rcut(const ngravs_config_t *c)
{
  return c->rcut > 0 ? c->rcut : NGRAVS_RCUT * cfg_asmth(c); /* pm_periodic.c:60 */
}
double orc_law_eval(const ngravs_config_t *cfg, int which, int id, double a3, double a4)
{
  switch(which)
    {
    case 0:
      return law_accel(cfg, id, 1.0, a3, a4);
    case 1:
      return law_spline_tn(cfg, id, 1.0, 1.0, a3, a4, 1);
    case 2:
      return law_greens(cfg, cfg_asmth(cfg), id, a3, a4);
    default:
      return law_normed(cfg, cfg_asmth(cfg), id, a3, a4);
    }
}

/* ------------------------------------------------------------------------------------------
 * The dynamic tree update between rebuilds (TreeDomainUpdateFrequency > 0), in the reference's semantics:
 *   node velocities vs[3][g] = mass-weighted mean velocity of the node's particles of species g
 *       (force_update_node_recursive, forcetree.c:451-743, accumulates them next to s[][]);
 *   move_particles (predict.c:79-91): s[j][g] += vs[j][g] * dt_drift for every node, then
 *   force_update_len -> force_update_node_len_local (forcetree.c:1005-1085): a particle that left its father's cell
 *       enlarges it to len = 2 max_k |Pos_k - center_k|, and the enlargement is handed up the father chain,
 *       len_p = 2 |center_p.x - center_no.x| + len_no while 0.999999 * that exceeds len_p.
 * Node kicks (timestep.c:331-344): when a particle is kicked by dv while the tree is kept, every ancestor's velocity of EVERY
 * species k with mass gets dv * m / M_k -- for N_GRAVS = 1 exactly what keeps vs the mass-weighted mean velocity; for
 * N_GRAVS > 1 the reference adds the kick of a particle to the node velocities of the species it does not belong to as well
 * (restated as written).  orc_tree_drift_kicked: vs from the velocities the tree was built with, then the kicks dv, then the drift.
 * newpos = the drifted positions (the caller drifts the particles, as move_particles does); vel, dt the velocities and the
 * drift interval used for it.  The tree then refers to newpos.
 * ------------------------------------------------------------------------------------------ */
void orc_tree_drift_kicked(orc_tree *t, const ngravs_config_t *cfg, const double *newpos, const double *vel, const double *dv, double dt);
void orc_tree_drift(orc_tree *t, const ngravs_config_t *cfg, const double *newpos, const double *vel, double dt)
{
  orc_tree_drift_kicked(t, cfg, newpos, vel, NULL, dt);
}

void orc_tree_drift_kicked(orc_tree *t, const ngravs_config_t *cfg, const double *newpos, const double *vel, const double *dv, double dt)
{
  const int ng = t->ng;
  const int64_t nn = t->numnodes;
  double *vs = calloc((size_t)nn * 3 * ng, sizeof(double));
  for(int64_t i = 0; i < t->n; i++)
    {
      const int g = cfg->type_to_grav[t->type[i]];
      for(int no = t->pfather[i]; no >= 0; no = t->father[no - t->maxpart])
        {
          const int64_t a = no - t->maxpart;
          for(int j = 0; j < 3; j++)
            vs[(a * 3 + j) * ng + g] += t->mass[i] * vel[3 * i + j];
        }
    }
  for(int64_t a = 0; a < nn; a++)   /* force_update_node_recursive: vs = sum(m v) / M per species (forcetree.c:617-619, 674-676, 695-697) */
    for(int g = 0; g < ng; g++)
      if(t->nmass[a * ng + g] > 0)
        for(int j = 0; j < 3; j++)
          vs[(a * 3 + j) * ng + g] /= t->nmass[a * ng + g];
  if(dv)
    for(int64_t i = 0; i < t->n; i++)   /* timestep.c:331-344 */
      for(int no = t->pfather[i]; no >= 0; no = t->father[no - t->maxpart])
        {
          const int64_t a = no - t->maxpart;
          for(int j = 0; j < 3; j++)
            for(int k = 0; k < ng; k++)
              if(t->nmass[a * ng + k] > 0)
                vs[(a * 3 + j) * ng + k] += dv[3 * i + j] * t->mass[i] / t->nmass[a * ng + k];
        }
  for(int64_t a = 0; a < nn; a++)
    for(int g = 0; g < ng; g++)
      if(t->nmass[a * ng + g] > 0)
        for(int j = 0; j < 3; j++)
          t->s[(a * 3 + j) * ng + g] += vs[(a * 3 + j) * ng + g] * dt;   /* predict.c:83-86 */
  free(vs);
  t->pos = newpos;
  for(int64_t i = 0; i < t->n; i++)   /* force_update_node_len_local, forcetree.c:1043-1085 */
    {
      int no = t->pfather[i];
      int64_t a = no - t->maxpart;
      double distmax = 0;
      for(int k = 0; k < 3; k++)
        {
          double dist = fabs(newpos[3 * i + k] - t->center[3 * a + k]);
          if(dist > distmax)
            distmax = dist;
        }
      if(distmax + distmax > t->len[a])
        {
          t->len[a] = distmax + distmax;
          int p = t->father[a];
          while(p >= 0)
            {
              const int64_t b = p - t->maxpart;
              distmax = fabs(t->center[3 * b] - t->center[3 * a]);
              distmax = distmax + distmax + t->len[a];
              if(0.999999 * distmax > t->len[b])
                {
                  t->len[b] = distmax;
                  a = b;
                  p = t->father[b];
                }
              else
                break;
            }
        }
    }
}

/* ------------------------------------------------------------------------------------------
 * The walks: force_treeevaluate (forcetree.c:1244-1610) and force_treeevaluate_shortrange
 * (:1623-2052), mode 0.  One function, `pm` selects the TreePM lines.
 * ------------------------------------------------------------------------------------------ */
#define NEAREST(x) (((x) > boxhalf) ? ((x)-boxsize) : (((x) < -boxhalf) ? ((x) + boxsize) : (x)))

/* `reach` (may be NULL; test instrumentation, not part of the reference): reach[p] = the smallest side of a tree node through
 * which particle p contributed to this target -- 0 for a particle-particle interaction, len for every particle below a node
 * that was used as a monopole.  It tells which parts of the tree a task must hold for the walk of its targets. */
static void reach_mark_subtree(const orc_tree *t, int64_t a, double len, double *reach);
static int walk_one_r(const orc_tree *t, const ngravs_config_t *cfg, int64_t target, double aold_in,
                      const double *table, double acc[3], double *reach);
static int walk_one(const orc_tree *t, const ngravs_config_t *cfg, int64_t target, double aold_in,
                    const double *table, double acc[3])
{
  return walk_one_r(t, cfg, target, aold_in, table, acc, NULL);
}
static int walk_one_r(const orc_tree *t, const ngravs_config_t *cfg, int64_t target, double aold_in,
                      const double *table, double acc[3], double *reach)
{
  const int ng = t->ng, pm = cfg->pmgrid != 0, periodic = cfg->periodic != 0;
  const double boxsize = cfg->box_size, boxhalf = 0.5 * cfg->box_size;
  const double px = t->pos[3 * target], py = t->pos[3 * target + 1], pz = t->pos[3 * target + 2];
  const int ptype = t->type[target];
  const int tg = cfg->type_to_grav[ptype];
  const double pmass = t->mass[target];          /* the BAM laws need the target's mass (forcetree.c:1342, pmass) */
  long nn[MAXG];                                 /* ... and the particle number behind every source (Nparticles[], :1563) */
  const double aold = cfg->err_tol_force_acc * aold_in;
  double rcut = 0, rcut2 = 0, asmthfac = 0, utor2wpi = 0;
  if(pm)
    {
      double asmth = cfg_asmth(cfg);
      rcut = cfg_rcut(cfg);
      rcut2 = rcut * rcut;
      asmthfac = 0.5 / asmth * (NTAB / 3.0);
      utor2wpi = 1.0 / (M_PI * 4 * asmth * asmth);
    }
  double ax = 0, ay = 0, az = 0;
  int nint = 0;
  double r2[MAXG], dx[MAXG], dy[MAXG], dz[MAXG], m[MAXG];
  double h = 0;
  int no = (int)t->maxpart;
  while(no >= 0)
    {
      int sg;
      if(no < t->maxpart)
        {
          sg = cfg->type_to_grav[t->type[no]];
          m[sg] = t->mass[no];
          nn[sg] = 1;
          dx[sg] = t->pos[3 * no] - px;
          dy[sg] = t->pos[3 * no + 1] - py;
          dz[sg] = t->pos[3 * no + 2] - pz;
          if(periodic)
            {
              dx[sg] = NEAREST(dx[sg]);
              dy[sg] = NEAREST(dy[sg]);
              dz[sg] = NEAREST(dz[sg]);
            }
          r2[sg] = dx[sg] * dx[sg] + dy[sg] * dy[sg] + dz[sg] * dz[sg];
          h = cfg->force_softening[ptype];
          if(h < cfg->force_softening[t->type[no]])
            h = cfg->force_softening[t->type[no]];
          if(reach)
            reach[no] = 0.0;
          no = t->pnext[no];
        }
      else
        {
          int64_t a = no - t->maxpart;
          double r2min = INFINITY, r2max = -INFINITY, summass = 0;
          for(int g = 0; g < ng; g++)
            {
              m[g] = t->nmass[a * ng + g];
              nn[g] = (long)t->npart[a * ng + g];
              summass += m[g];
              dx[g] = t->s[(a * 3 + 0) * ng + g] - px;
              dy[g] = t->s[(a * 3 + 1) * ng + g] - py;
              dz[g] = t->s[(a * 3 + 2) * ng + g] - pz;
              if(periodic)
                {
                  dx[g] = NEAREST(dx[g]);
                  dy[g] = NEAREST(dy[g]);
                  dz[g] = NEAREST(dz[g]);
                }
              r2[g] = dx[g] * dx[g] + dy[g] * dy[g] + dz[g] * dz[g];
              if(r2[g] < r2min)
                r2min = r2[g];
              if(r2[g] > r2max)
                r2max = r2[g];
            }
          sg = -1;
          const double len = t->len[a];
          const double *ctr = &t->center[3 * a];
          if(pm && r2min > rcut2)
            {
              /* forcetree.c:1828-1862 */
              double eff = rcut + 0.5 * len, d;
              int skip = 0;
              for(int j = 0; j < 3 && !skip; j++)
                {
                  d = ctr[j] - (j == 0 ? px : (j == 1 ? py : pz));
                  if(periodic)
                    d = NEAREST(d);
                  if(d < -eff || d > eff)
                    skip = 1;
                }
              if(skip)
                {
                  no = t->sibling[a];
                  continue;
                }
            }
          if(cfg->err_tol_theta != 0)
            {
              if(len * len > r2min * cfg->err_tol_theta * cfg->err_tol_theta)
                {
                  no = t->nextnode[a];
                  continue;
                }
            }
          else
            {
              if(summass * len * len > r2min * r2min * aold)
                {
                  no = t->nextnode[a];
                  continue;
                }
              /* inside-the-cell test: no NEAREST here, as in the reference (:1462-1472,:1885-1897) */
              if(fabs(ctr[0] - px) < 0.60 * len && fabs(ctr[1] - py) < 0.60 * len && fabs(ctr[2] - pz) < 0.60 * len)
                {
                  no = t->nextnode[a];
                  continue;
                }
            }
          h = cfg->force_softening[ptype];
          int mst = (t->bitflags[a] >> 2) & 7;
          if(mst == 7)
            {
              if(summass > 0)
                {
                  fprintf(stderr, "oracle: endrun(986/987) massive node without softening type\n");
                  exit(3);
                }
              no = t->nextnode[a];
              continue;
            }
          if(h < cfg->force_softening[mst])
            {
              h = cfg->force_softening[mst];
              if(r2max < h * h && ((t->bitflags[a] >> 5) & 1))
                {
                  no = t->nextnode[a];
                  continue;
                }
            }
          if(reach)
            reach_mark_subtree(t, a, len, reach);
          no = t->sibling[a];
        }
      /* interaction(s): forcetree.c:1534-1585 / :1953-2032 */
      int added = 0;
      for(int g = (sg >= 0 ? sg : 0); g < (sg >= 0 ? sg + 1 : ng); g++)
        {
          if(sg < 0 && m[g] == 0.0)
            continue;
          double r = sqrt(r2[g]), fac;
          if(pm)
            {
              int tab = (int)(asmthfac * r);
              if(tab >= NTAB)
                continue;
              if(r >= h)
                {
                  fac = law_accel_tn(cfg, cfg->law_accel[tg][g], pmass, m[g], r2[g], r, nn[g]);
                  fac -= m[g] * utor2wpi * table[((size_t)tg * ng + g) * NTAB + tab];
                  fac /= r;
                }
              else
                fac = law_spline_tn(cfg, cfg->law_spline[tg][g], pmass, m[g], h, r, nn[g]);
            }
          else
            {
              if(r >= h)
                fac = law_accel_tn(cfg, cfg->law_accel[tg][g], pmass, m[g], r2[g], r, nn[g]) / r;
              else
                fac = law_spline_tn(cfg, cfg->law_spline[tg][g], pmass, m[g], h, r, nn[g]);
            }
          ax += dx[g] * fac;
          ay += dy[g] * fac;
          az += dz[g] * fac;
          added = 1;
        }
      if(added || !pm)
        nint++;
    }
  acc[0] = ax;
  acc[1] = ay;
  acc[2] = az;
  return nint;
}

/* every particle below node a, along the thread of the tree (nextnode descends, pnext / sibling move on) */
static void reach_mark_subtree(const orc_tree *t, int64_t a, double len, double *reach)
{
  const int end = t->sibling[a];
  int q = t->nextnode[a];
  while(q != end && q >= 0)
    {
      if(q < t->maxpart)
        {
          if(len < reach[q])
            reach[q] = len;
          q = t->pnext[q];
        }
      else
        q = t->nextnode[q - t->maxpart];
    }
}

/* test instrumentation: the walks of targets idx[0..nt), serially; reach[n] must come in filled with a large value */
int orc_walk_reach(const orc_tree *t, const ngravs_config_t *cfg, const int32_t *idx, int64_t nt, const double *old_acc,
                   const double *table, double *reach)
{
  if(cfg->pmgrid && !table)
    return -1;
  if(!idx)
    nt = t->n;
  for(int64_t k = 0; k < nt; k++)
    {
      int64_t i = idx ? idx[k] : k;
      double a[3];
      walk_one_r(t, cfg, i, old_acc ? old_acc[i] : 0.0, table, a, reach);
    }
  return 0;
}

int orc_walk(const orc_tree *t, const ngravs_config_t *cfg, const int32_t *idx, int64_t nt, const double *old_acc,
             const double *table, double *acc, int32_t *nint, int nthreads)
{
  if(cfg->pmgrid && !table)
    return -1;
  if(!idx)
    nt = t->n;
#ifdef _OPENMP
  if(nthreads > 0)
    omp_set_num_threads(nthreads);
#endif
#pragma omp parallel for schedule(dynamic, 256)
  for(int64_t k = 0; k < nt; k++)
    {
      int64_t i = idx ? idx[k] : k;
      double a[3];
      int ni = walk_one(t, cfg, i, old_acc ? old_acc[i] : 0.0, table, a);
      acc[3 * k] = a[0];
      acc[3 * k + 1] = a[1];
      acc[3 * k + 2] = a[2];
      if(nint)
        nint[k] = ni;
    }
  return 0;
}

void orc_finish(const ngravs_config_t *cfg, int64_t n, double *acc, const double *pm, double *old_acc_out)
{
  /* gravtree.c:318-341 */
  for(int64_t i = 0; i < n; i++)
    {
      double ax = acc[3 * i], ay = acc[3 * i + 1], az = acc[3 * i + 2];
      if(pm)
        {
          ax += pm[3 * i] / cfg->G;
          ay += pm[3 * i + 1] / cfg->G;
          az += pm[3 * i + 2] / cfg->G;
        }
      if(old_acc_out)
        old_acc_out[i] = sqrt(ax * ax + ay * ay + az * az);
      for(int j = 0; j < 3; j++)
        acc[3 * i + j] *= cfg->G;
    }
}

/* ------------------------------------------------------------------------------------------
 * Short-range tables (forcetree.c:3246-3403, ngravs_core.c:45-184).  The reference takes a
 * length-589 682 backward DFT of the symmetric sequence in[j] = Gnorm(k_j) exp(-k_j^2 Z^2); the
 * integrand underflows to exactly 0 beyond k ~ 55, so the same DFT sums are evaluated here
 * directly over the non-zero inputs (identical mathematics, no FFTW).
 * ------------------------------------------------------------------------------------------ */
void orc_shortrange_table(const ngravs_config_t *cfg, double *force, double *pot)
{
  const int ntab = NTAB, len = 3, ol = 8;
  const long long n = 12LL * ntab * ol * len - 6 * ol * len + 2; /* ngravs_core.c:177 */
  const double Z = 0.5;
  const double dk = 2.0 * M_PI * ntab * 6.0 * ol / (3.0 * n); /* jTok(1), ngravs_core.c:45-48 */
  const double asmth = cfg_asmth(cfg);
  const int ng = cfg->n_gravs;
  const long long mmax = (long long)ol * (6 * (ntab - 1) + 3) + 4; /* highest out[] index touched */
  double *fin = malloc(sizeof(double) * (size_t)(n / 2));
  double *out = malloc(sizeof(double) * (size_t)(mmax + 4));
  double *run = malloc(sizeof(double) * (size_t)(mmax / 3 + 4));
  for(int nA = 0; nA < ng; nA++)     /* sources   */
    for(int nB = 0; nB < ng; nB++)   /* receivers */
      {
        int law = cfg->law_normed[nB][nA];
        long long jmax = 0;
        for(long long j = 0; j < n / 2; j++)
          {
            double k = dk * j, k2 = k * k;
            fin[j] = law_normed(cfg, asmth, law, k2, k) * exp(-k2 * Z * Z);
            if(fin[j] != 0.0)
              jmax = j;
            else if(k > 60.0)
              break;
          }
        /* out[m] = sum_j in[j] exp(+2 pi i j m / n), in symmetric => real */
#pragma omp parallel for schedule(static)
        for(long long m = 0; m <= mmax + 2; m++)
          {
            double s = fin[0];
            for(long long j = 1; j <= jmax; j++)
              {
                long long jm = (j * m) % n;
                s += 2.0 * fin[j] * cos(2.0 * M_PI * (double)jm / (double)n);
              }
            out[m] = s;
          }
        const double norm = dk;
        /* running Newton-Cotes 3/8 integral (ngravs_core.c:137-144); mTox(j) = 3j/(6 ntab ol) */
        double sum = 0.0;
        run[0] = 0.0;
        for(long long m = 0; m + 3 <= mmax + 2; m += 3)
          {
            double x0 = 3.0 * m / (6.0 * ntab * ol), x3 = 3.0 * (m + 3) / (6.0 * ntab * ol);
            sum += (x3 - x0) * 0.125 * norm * (out[m] + 3.0 * out[m + 1] + 3.0 * out[m + 2] + out[m + 3]);
            run[m / 3 + 1] = sum;
          }
        for(int i = 0; i < ntab; i++)
          {
            long long gi = (long long)ol * (6 * i + 3); /* gadgetToFourier */
            double temp = out[gi] * norm, tempI = run[gi / 3];
            double u = 3.0 / ntab * (i + 0.5);
            tempI /= u * u;
            temp /= u;
            if(pot)
              pot[((size_t)nB * ng + nA) * ntab + i] = temp;
            tempI -= temp;
            force[((size_t)nB * ng + nA) * ntab + i] = tempI;
          }
      }
  free(fin);
  free(out);
  free(run);
}

/* ------------------------------------------------------------------------------------------
 * Periodic PM (pm_periodic.c:204-790), one rank.  The patch / slab exchange of the reference is
 * communication only; the arithmetic (CIC weights, Green's multiplier, 4-point gradient, CIC
 * gather) is restated on one global mesh.  FFT: own radix-2 complex transform (FFTW-2 is a plain
 * unnormalised DFT, forward sign -1).
 * ------------------------------------------------------------------------------------------ */
static void fft1d(double *re, double *im, int n, int sign)
{
  for(int i = 1, j = 0; i < n; i++)
    {
      int bit = n >> 1;
      for(; j & bit; bit >>= 1)
        j ^= bit;
      j ^= bit;
      if(i < j)
        {
          double t = re[i];
          re[i] = re[j];
          re[j] = t;
          t = im[i];
          im[i] = im[j];
          im[j] = t;
        }
    }
  for(int len = 2; len <= n; len <<= 1)
    {
      double ang = sign * 2.0 * M_PI / len;
      for(int i = 0; i < n; i += len)
        for(int k = 0; k < len / 2; k++)
          {
            double wr = cos(ang * k), wi = sin(ang * k);
            int a = i + k, b = i + k + len / 2;
            double xr = re[b] * wr - im[b] * wi, xi = re[b] * wi + im[b] * wr;
            re[b] = re[a] - xr;
            im[b] = im[a] - xi;
            re[a] += xr;
            im[a] += xi;
          }
    }
}
static void fft3d(double *re, double *im, int n, int sign)
{
#pragma omp parallel
  {
    double *br = malloc(sizeof(double) * n), *bi = malloc(sizeof(double) * n);
    /* z lines */
#pragma omp for schedule(static)
    for(long long l = 0; l < (long long)n * n; l++)
      fft1d(re + l * n, im + l * n, n, sign);
    /* y lines */
#pragma omp for schedule(static)
    for(long long l = 0; l < (long long)n * n; l++)
      {
        long long x = l / n, z = l % n;
        for(int y = 0; y < n; y++)
          {
            br[y] = re[(x * n + y) * n + z];
            bi[y] = im[(x * n + y) * n + z];
          }
        fft1d(br, bi, n, sign);
        for(int y = 0; y < n; y++)
          {
            re[(x * n + y) * n + z] = br[y];
            im[(x * n + y) * n + z] = bi[y];
          }
      }
    /* x lines */
#pragma omp for schedule(static)
    for(long long l = 0; l < (long long)n * n; l++)
      {
        long long y = l / n, z = l % n;
        for(int x = 0; x < n; x++)
          {
            br[x] = re[((long long)x * n + y) * n + z];
            bi[x] = im[((long long)x * n + y) * n + z];
          }
        fft1d(br, bi, n, sign);
        for(int x = 0; x < n; x++)
          {
            re[((long long)x * n + y) * n + z] = br[x];
            im[((long long)x * n + y) * n + z] = bi[x];
          }
      }
    free(br);
    free(bi);
  }
}

int orc_pm_periodic(const ngravs_config_t *cfg, const double *pos, const double *mass, const int32_t *type,
                    int64_t n, double *gravpm)
{
  const int N = cfg->pmgrid, ng = cfg->n_gravs;
  if(N <= 0 || (N & (N - 1)))
    return -1;
  const double L = cfg->box_size, to_slab = N / L;
  const double asmth = cfg_asmth(cfg);
  double asmth2 = (2 * M_PI) * asmth / L;
  asmth2 *= asmth2;
  double fac = cfg->G / (M_PI * L);
  fac *= 1 / (2 * L / N);
  const size_t M = (size_t)N * N * N;
  double *re = malloc(sizeof(double) * M), *im = malloc(sizeof(double) * M), *fg = malloc(sizeof(double) * M);
  if(!re || !im || !fg)
    return -2;
  for(int64_t i = 0; i < 3 * n; i++)
    gravpm[i] = 0;
#define IDX(x, y, z) ((((size_t)(x)) * N + (y)) * N + (z))
#define WRAP(a) (((a) + N) % N)
  for(int nA = 0; nA < ng; nA++)   /* sources   */
    for(int nB = 0; nB < ng; nB++) /* receivers */
      {
        memset(re, 0, sizeof(double) * M);
        memset(im, 0, sizeof(double) * M);
        for(int64_t i = 0; i < n; i++) /* CIC deposit, pm_periodic.c:297-331 */
          {
            if(cfg->type_to_grav[type[i]] != nA)
              continue;
            int sx = (int)(to_slab * pos[3 * i]), sy = (int)(to_slab * pos[3 * i + 1]), sz = (int)(to_slab * pos[3 * i + 2]);
            if(sx >= N)
              sx = N - 1;
            if(sy >= N)
              sy = N - 1;
            if(sz >= N)
              sz = N - 1;
            double dx = to_slab * pos[3 * i] - sx, dy = to_slab * pos[3 * i + 1] - sy, dz = to_slab * pos[3 * i + 2] - sz;
            int sxx = WRAP(sx + 1), syy = WRAP(sy + 1), szz = WRAP(sz + 1);
            double mm = mass[i];
            re[IDX(sx, sy, sz)] += mm * (1.0 - dx) * (1.0 - dy) * (1.0 - dz);
            re[IDX(sx, syy, sz)] += mm * (1.0 - dx) * dy * (1.0 - dz);
            re[IDX(sx, sy, szz)] += mm * (1.0 - dx) * (1.0 - dy) * dz;
            re[IDX(sx, syy, szz)] += mm * (1.0 - dx) * dy * dz;
            re[IDX(sxx, sy, sz)] += mm * (dx) * (1.0 - dy) * (1.0 - dz);
            re[IDX(sxx, syy, sz)] += mm * (dx)*dy * (1.0 - dz);
            re[IDX(sxx, sy, szz)] += mm * (dx) * (1.0 - dy) * dz;
            re[IDX(sxx, syy, szz)] += mm * (dx)*dy * dz;
          }
        fft3d(re, im, N, -1);
        int law = cfg->law_greens[nA][nB]; /* indexed [source][target] as in pm_periodic.c:490 */
#pragma omp parallel for schedule(static)
        for(int x = 0; x < N; x++)
          for(int y = 0; y < N; y++)
            for(int z = 0; z < N; z++)
              {
                double kx = x > N / 2 ? x - N : x, ky = y > N / 2 ? y - N : y, kz = z > N / 2 ? z - N : z;
                double k2 = kx * kx + ky * ky + kz * kz;
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
                    double smth = law_greens(cfg, asmth, law, k2, sqrt(k2));
                    smth *= -exp(-k2 * asmth2) * ff * ff * ff * ff;
                    re[IDX(x, y, z)] *= smth;
                    im[IDX(x, y, z)] *= smth;
                  }
              }
        re[0] = im[0] = 0.0;
        fft3d(re, im, N, +1); /* unnormalised inverse: re = potential */
        for(int dim = 0; dim < 3; dim++)
          {
#pragma omp parallel for schedule(static)
            for(int x = 0; x < N; x++)
              for(int y = 0; y < N; y++)
                for(int z = 0; z < N; z++)
                  {
                    int l[3] = {x, y, z}, r[3] = {x, y, z}, ll[3] = {x, y, z}, rr[3] = {x, y, z};
                    l[dim] = WRAP(l[dim] - 1);
                    r[dim] = WRAP(r[dim] + 1);
                    ll[dim] = WRAP(ll[dim] - 2);
                    rr[dim] = WRAP(rr[dim] + 2);
                    fg[IDX(x, y, z)] = fac * ((4.0 / 3) * (re[IDX(l[0], l[1], l[2])] - re[IDX(r[0], r[1], r[2])]) -
                                              (1.0 / 6) * (re[IDX(ll[0], ll[1], ll[2])] - re[IDX(rr[0], rr[1], rr[2])]));
                  }
#pragma omp parallel for schedule(static)
            for(int64_t i = 0; i < n; i++)
              {
                if(cfg->type_to_grav[type[i]] != nB)
                  continue;
                int sx = (int)(to_slab * pos[3 * i]), sy = (int)(to_slab * pos[3 * i + 1]), sz = (int)(to_slab * pos[3 * i + 2]);
                if(sx >= N)
                  sx = N - 1;
                if(sy >= N)
                  sy = N - 1;
                if(sz >= N)
                  sz = N - 1;
                double dx = to_slab * pos[3 * i] - sx, dy = to_slab * pos[3 * i + 1] - sy, dz = to_slab * pos[3 * i + 2] - sz;
                int sxx = WRAP(sx + 1), syy = WRAP(sy + 1), szz = WRAP(sz + 1);
                double a = fg[IDX(sx, sy, sz)] * (1.0 - dx) * (1.0 - dy) * (1.0 - dz);
                a += fg[IDX(sx, syy, sz)] * (1.0 - dx) * dy * (1.0 - dz);
                a += fg[IDX(sx, sy, szz)] * (1.0 - dx) * (1.0 - dy) * dz;
                a += fg[IDX(sx, syy, szz)] * (1.0 - dx) * dy * dz;
                a += fg[IDX(sxx, sy, sz)] * (dx) * (1.0 - dy) * (1.0 - dz);
                a += fg[IDX(sxx, syy, sz)] * (dx)*dy * (1.0 - dz);
                a += fg[IDX(sxx, sy, szz)] * (dx) * (1.0 - dy) * dz;
                a += fg[IDX(sxx, syy, szz)] * (dx)*dy * dz;
                gravpm[3 * i + dim] += a;
              }
          }
      }
  free(re);
  free(im);
  free(fg);
  return 0;
}


/* ------------------------------------------------------------------------------------------
 * Periodic tree-only path: Ewald / lattice-sum correction tables (lattice_init forcetree.c:3611-3793;
 * ewald_force ngravs.c:1170-1232; yukawa_lattice_force ngravs.c:1019-1090; coloyuk_lattice_force :840-847)
 * and their trilinear lookup (lattice_corr forcetree.c:3803-3885).  EN = NGRAVS_EN = 64.
 * table layout: [3][EN+1][EN+1][EN+1], already divided by BoxSize^2 (forcetree.c:3742-3750).
 * ------------------------------------------------------------------------------------------ */
#define ORC_EN 64
static void lat_newton(int i, int j, int k, const double x[3], double force[3])
{
  const double alpha = 2.0;
  for(int c = 0; c < 3; c++)
    force[c] = 0;
  if(i == 0 && j == 0 && k == 0)
    return;
  double r2 = x[0] * x[0] + x[1] * x[1] + x[2] * x[2];
  for(int c = 0; c < 3; c++)
    force[c] += x[c] / (r2 * sqrt(r2));
  for(int n0 = -4; n0 <= 4; n0++)
    for(int n1 = -4; n1 <= 4; n1++)
      for(int n2 = -4; n2 <= 4; n2++)
        {
          double dx[3] = {x[0] - n0, x[1] - n1, x[2] - n2};
          double r = sqrt(dx[0] * dx[0] + dx[1] * dx[1] + dx[2] * dx[2]);
          double val = erfc(alpha * r) + 2 * alpha * r / sqrt(M_PI) * exp(-alpha * alpha * r * r);
          for(int c = 0; c < 3; c++)
            force[c] -= dx[c] / (r * r * r) * val;
        }
  for(int h0 = -4; h0 <= 4; h0++)
    for(int h1 = -4; h1 <= 4; h1++)
      for(int h2_ = -4; h2_ <= 4; h2_++)
        {
          int h[3] = {h0, h1, h2_};
          double hdotx = x[0] * h0 + x[1] * h1 + x[2] * h2_;
          int h2 = h0 * h0 + h1 * h1 + h2_ * h2_;
          if(h2 > 0)
            {
              double val = 2.0 / ((double)h2) * exp(-M_PI * M_PI * h2 / (alpha * alpha)) * sin(2 * M_PI * hdotx);
              for(int c = 0; c < 3; c++)
                force[c] -= h[c] * val;
            }
        }
}
static void lat_yukawa(double ymass, int i, int j, int k, const double x[3], double force[3])
{
  const double alpha = 5.64;
  for(int c = 0; c < 3; c++)
    force[c] = 0;   /* the reference leaves the origin entry unset (ngravs.c:1029-1030); 0 here */
  if(i == 0 && j == 0 && k == 0)
    return;
  double r2 = x[0] * x[0] + x[1] * x[1] + x[2] * x[2], r = sqrt(r2), ym = ymass;
  for(int c = 0; c < 3; c++)
    force[c] = exp(-r * ym) * (ym + 1.0 / r) * x[c] / r2;
  for(int n0 = -5; n0 <= 5; n0++)
    for(int n1 = -5; n1 <= 5; n1++)
      for(int n2 = -5; n2 <= 5; n2++)
        {
          double dx[3] = {x[0] - n0, x[1] - n1, x[2] - n2};
          r = sqrt(dx[0] * dx[0] + dx[1] * dx[1] + dx[2] * dx[2]);
          double ep = exp(ym * r) * erfc(alpha * r + ym / (2 * alpha)), em = exp(-ym * r) * erfc(alpha * r - ym / (2 * alpha));
          double val = 0.5 * (ep + em);
          for(int c = 0; c < 3; c++)
            force[c] -= dx[c] / (r * r * r) * val;
          val = 0.5 * ym * (-ep + em) + 2 * alpha * exp(-alpha * alpha * r * r - ym * ym / (4 * alpha * alpha)) / sqrt(M_PI);
          for(int c = 0; c < 3; c++)
            force[c] -= dx[c] / (r * r) * val;
        }
  ym /= 2 * M_PI;
  for(int h0 = -5; h0 <= 5; h0++)
    for(int h1 = -5; h1 <= 5; h1++)
      for(int h2_ = -5; h2_ <= 5; h2_++)
        {
          int h[3] = {h0, h1, h2_};
          double hdotx = x[0] * h0 + x[1] * h1 + x[2] * h2_;
          int h2 = h0 * h0 + h1 * h1 + h2_ * h2_;
          if(h2 > 0)
            {
              double val = 2 * exp(-M_PI * M_PI * (h2 + ym * ym) / (alpha * alpha)) * sin(2 * M_PI * hdotx) / (h2 + ym * ym);
              for(int c = 0; c < 3; c++)
                force[c] -= h[c] * val;
            }
        }
}
void orc_lattice_table(const ngravs_config_t *cfg, int law, double *tab)
{
  const int E1 = ORC_EN + 1;
  const double L2 = cfg->box_size * cfg->box_size;
#pragma omp parallel for schedule(dynamic, 64)
  for(int n = 0; n < E1 * E1 * E1; n++)
    {
      int i = n / (E1 * E1), j = (n / E1) % E1, k = n % E1;
      double x[3] = {0.5 * ((double)i) / ORC_EN, 0.5 * ((double)j) / ORC_EN, 0.5 * ((double)k) / ORC_EN};
      double f[3] = {0, 0, 0}, g[3];
      if(law == NGRAVS_LAW_NEWTON || law == NGRAVS_LAW_NEG_NEWTON || law == NGRAVS_LAW_COLOYUK)
        {
          lat_newton(i, j, k, x, g);
          for(int c = 0; c < 3; c++)
            f[c] += (law == NGRAVS_LAW_NEG_NEWTON ? -g[c] : g[c]);
        }
      if(law == NGRAVS_LAW_YUKAWA || law == NGRAVS_LAW_COLOYUK)
        {
          lat_yukawa(cfg->yukawa_imass, i, j, k, x, g);
          for(int c = 0; c < 3; c++)
            f[c] += g[c];
        }
      for(int c = 0; c < 3; c++)
        tab[(size_t)c * E1 * E1 * E1 + n] = f[c] / L2;
    }
}
/* lattice_corr: dx,dy,dz dimensionful nearest-image displacement -> fper[3] (to be multiplied by the source mass) */
static void lat_lookup(const double *tab, double box, double dx, double dy, double dz, double fper[3])
{
  const int E1 = ORC_EN + 1;
  const double fac_intp = 2 * ORC_EN / box;
  int sx, sy, sz;
  if(dx < 0) { dx = -dx; sx = +1; } else sx = -1;
  if(dy < 0) { dy = -dy; sy = +1; } else sy = -1;
  if(dz < 0) { dz = -dz; sz = +1; } else sz = -1;
  double u = dx * fac_intp, v = dy * fac_intp, w = dz * fac_intp;
  int i = (int)u, j = (int)v, k = (int)w;
  if(i >= ORC_EN) i = ORC_EN - 1;
  if(j >= ORC_EN) j = ORC_EN - 1;
  if(k >= ORC_EN) k = ORC_EN - 1;
  u -= i; v -= j; w -= k;
  double f[8] = {(1 - u) * (1 - v) * (1 - w), (1 - u) * (1 - v) * (w), (1 - u) * (v) * (1 - w), (1 - u) * (v) * (w),
                 (u) * (1 - v) * (1 - w),     (u) * (1 - v) * (w),     (u) * (v) * (1 - w),     (u) * (v) * (w)};
  const int sg[3] = {sx, sy, sz};
  for(int c = 0; c < 3; c++)
    {
      const double *t = tab + (size_t)c * E1 * E1 * E1;
#define T3(a, b, d) t[((size_t)(a) * E1 + (b)) * E1 + (d)]
      fper[c] = sg[c] * (T3(i, j, k) * f[0] + T3(i, j, k + 1) * f[1] + T3(i, j + 1, k) * f[2] + T3(i, j + 1, k + 1) * f[3] +
                         T3(i + 1, j, k) * f[4] + T3(i + 1, j, k + 1) * f[5] + T3(i + 1, j + 1, k) * f[6] + T3(i + 1, j + 1, k + 1) * f[7]);
#undef T3
    }
}

/* force_treeevaluate_lattice_correction (forcetree.c:2077-2455), mode 0; lat = [tg][sg][3][E1^3] */
static int lattice_walk_one(const orc_tree *t, const ngravs_config_t *cfg, int64_t target, double aold_in, const double *lat,
                            double acc[3])
{
  const int ng = t->ng, E1 = ORC_EN + 1;
  const size_t tsz = (size_t)3 * E1 * E1 * E1;
  const double boxsize = cfg->box_size, boxhalf = 0.5 * cfg->box_size;
  const double px = t->pos[3 * target], py = t->pos[3 * target + 1], pz = t->pos[3 * target + 2];
  const int tg = cfg->type_to_grav[t->type[target]];
  const double aold = cfg->err_tol_force_acc * aold_in;
  double ax = 0, ay = 0, az = 0;
  int cost = 0;
  double dx[MAXG], dy[MAXG], dz[MAXG], m[MAXG], r2[MAXG];
  int no = (int)t->maxpart;
  while(no >= 0)
    {
      int sg;
      if(no < t->maxpart)
        {
          sg = cfg->type_to_grav[t->type[no]];
          m[sg] = t->mass[no];
          dx[sg] = NEAREST(t->pos[3 * no] - px);
          dy[sg] = NEAREST(t->pos[3 * no + 1] - py);
          dz[sg] = NEAREST(t->pos[3 * no + 2] - pz);
          no = t->pnext[no];
        }
      else
        {
          int64_t a = no - t->maxpart;
          double r2min = INFINITY, summass = 0;
          for(int g = 0; g < ng; g++)
            {
              m[g] = t->nmass[a * ng + g];
              summass += m[g];
              dx[g] = NEAREST(t->s[(a * 3 + 0) * ng + g] - px);
              dy[g] = NEAREST(t->s[(a * 3 + 1) * ng + g] - py);
              dz[g] = NEAREST(t->s[(a * 3 + 2) * ng + g] - pz);
              r2[g] = dx[g] * dx[g] + dy[g] * dy[g] + dz[g] * dz[g];
              if(r2[g] < r2min)
                r2min = r2[g];
            }
          sg = -1;
          const double len = t->len[a];
          const double *ctr = &t->center[3 * a];
          int openflag = 0;
          if(cfg->err_tol_theta != 0)
            {
              if(len * len > r2min * cfg->err_tol_theta * cfg->err_tol_theta)
                openflag = 1;
            }
          else
            {
              if(summass * len * len > r2min * r2min * aold)
                openflag = 1;
              else if(fabs(ctr[0] - px) < 0.60 * len && fabs(ctr[1] - py) < 0.60 * len && fabs(ctr[2] - pz) < 0.60 * len)
                openflag = 1;
            }
          if(openflag)
            {
              int must = 0;
              for(int j = 0; j < 3 && !must; j++)
                {
                  double u = ctr[j] - (j == 0 ? px : (j == 1 ? py : pz));
                  if(u > boxhalf)
                    u -= boxsize;
                  if(u < -boxhalf)
                    u += boxsize;
                  if(fabs(u) > 0.5 * (boxsize - len))
                    must = 1;
                }
              if(!must && len > 0.20 * boxsize)
                must = 1;
              if(must)
                {
                  no = t->nextnode[a];
                  continue;
                }
            }
          no = t->sibling[a];
        }
      for(int g = (sg >= 0 ? sg : 0); g < (sg >= 0 ? sg + 1 : ng); g++)
        {
          if(sg < 0 && m[g] == 0.0)
            continue;
          double f[3];
          lat_lookup(lat + ((size_t)tg * ng + g) * tsz, boxsize, dx[g], dy[g], dz[g], f);
          ax += m[g] * f[0];
          ay += m[g] * f[1];
          az += m[g] * f[2];
        }
      cost++;
    }
  acc[0] = ax;
  acc[1] = ay;
  acc[2] = az;
  return cost;
}

/* adds the lattice correction (and its cost) to acc/nint computed by orc_walk: forcetree.c:1605-1607 */
int orc_lattice_walk(const orc_tree *t, const ngravs_config_t *cfg, const int32_t *idx, int64_t nt, const double *old_acc,
                     const double *lat, double *acc, int32_t *nint, int nthreads)
{
  if(!idx)
    nt = t->n;
#ifdef _OPENMP
  if(nthreads > 0)
    omp_set_num_threads(nthreads);
#endif
#pragma omp parallel for schedule(dynamic, 256)
  for(int64_t k = 0; k < nt; k++)
    {
      int64_t i = idx ? idx[k] : k;
      double a[3];
      int c = lattice_walk_one(t, cfg, i, old_acc ? old_acc[i] : 0.0, lat, a);
      acc[3 * k] += a[0];
      acc[3 * k + 1] += a[1];
      acc[3 * k + 2] += a[2];
      if(nint)
        nint[k] += c;
    }
  return 0;
}

/* force_treeevaluate_direct (forcetree.c:3428-3548); lat != NULL adds lattice_corr (PERIODIC); xG as gravity_forcetest */
static void orc_direct_impl(const ngravs_config_t *cfg, const double *pos, const double *mass, const int32_t *type, int64_t n,
                            const int32_t *idx, int64_t nt, double *acc, int nthreads, const double *lat);
void orc_direct(const ngravs_config_t *cfg, const double *pos, const double *mass, const int32_t *type, int64_t n,
                const int32_t *idx, int64_t nt, double *acc, int nthreads)
{
  orc_direct_impl(cfg, pos, mass, type, n, idx, nt, acc, nthreads, NULL);
}
void orc_direct_lattice(const ngravs_config_t *cfg, const double *pos, const double *mass, const int32_t *type, int64_t n,
                        const int32_t *idx, int64_t nt, double *acc, int nthreads, const double *lat)
{
  orc_direct_impl(cfg, pos, mass, type, n, idx, nt, acc, nthreads, lat);
}
static void orc_direct_impl(const ngravs_config_t *cfg, const double *pos, const double *mass, const int32_t *type, int64_t n,
                            const int32_t *idx, int64_t nt, double *acc, int nthreads, const double *lat)
{
  const double boxsize = cfg->box_size, boxhalf = 0.5 * cfg->box_size;
#ifdef _OPENMP
  if(nthreads > 0)
    omp_set_num_threads(nthreads);
#endif
#pragma omp parallel for schedule(static)
  for(int64_t k = 0; k < nt; k++)
    {
      int64_t t = idx ? idx[k] : k;
      int ptype = type[t], tg = cfg->type_to_grav[ptype];
      double ax = 0, ay = 0, az = 0;
      for(int64_t i = 0; i < n; i++)
        {
          double h = cfg->force_softening[type[i]] > cfg->force_softening[ptype] ? cfg->force_softening[type[i]]
                                                                                   : cfg->force_softening[ptype];
          double dx = pos[3 * i] - pos[3 * t], dy = pos[3 * i + 1] - pos[3 * t + 1], dz = pos[3 * i + 2] - pos[3 * t + 2];
          if(cfg->periodic)
            {
              while(dx > boxhalf)
                dx -= boxsize;
              while(dy > boxhalf)
                dy -= boxsize;
              while(dz > boxhalf)
                dz -= boxsize;
              while(dx < -boxhalf)
                dx += boxsize;
              while(dy < -boxhalf)
                dy += boxsize;
              while(dz < -boxhalf)
                dz += boxsize;
            }
          double r2 = dx * dx + dy * dy + dz * dz, r = sqrt(r2), u = r * (1 / h), fac;
          int sg = cfg->type_to_grav[type[i]];
          if(u >= 1)
            fac = law_accel_tn(cfg, cfg->law_accel[tg][sg], mass[t], mass[i], r2, r, 1) / r;
          else
            fac = law_spline_tn(cfg, cfg->law_spline[tg][sg], mass[t], mass[i], h, r, 1);
          ax += dx * fac;
          ay += dy * fac;
          az += dz * fac;
          if(lat && u > 1.0e-5)   /* forcetree.c:3519-3528 */
            {
              double fc[3];
              const size_t tsz = (size_t)3 * (ORC_EN + 1) * (ORC_EN + 1) * (ORC_EN + 1);
              lat_lookup(lat + ((size_t)tg * cfg->n_gravs + sg) * tsz, cfg->box_size, dx, dy, dz, fc);
              ax += mass[i] * fc[0];
              ay += mass[i] * fc[1];
              az += mass[i] * fc[2];
            }
        }
      acc[3 * k] = ax * cfg->G;
      acc[3 * k + 1] = ay * cfg->G;
      acc[3 * k + 2] = az * cfg->G;
    }
}

/* Test instrumentation (not a reference function): the reference's short-range PAIR interaction (forcetree.c:1953-2032: law minus
 * the tabulated long-range part, spline inside the softening length, nothing beyond the table) applied to EVERY particle within
 * `reach` of a target -- what a TreePM walk computes when every source inside its cut is taken as a particle.  The production
 * group walk of the engine cuts at a sphere of group_reach * Asmth; where its lists hold particles only it must reproduce this sum
 * pair for pair.  acc without G (as the walks leave it); nint = pairs that added something. */
void orc_direct_shortrange(const ngravs_config_t *cfg, const double *pos, const double *mass, const int32_t *type, int64_t n,
                           const int32_t *idx, int64_t nt, const double *table, double reach, double *acc, int32_t *nint, int nthreads)
{
  const int ng = cfg->n_gravs;
  const double boxsize = cfg->box_size, boxhalf = 0.5 * cfg->box_size;
  const double asmth = cfg_asmth(cfg), asmthfac = 0.5 / asmth * (NTAB / 3.0), utor2wpi = 1.0 / (M_PI * 4 * asmth * asmth);
  const double reach2 = reach * reach;
#ifdef _OPENMP
  if(nthreads > 0)
    omp_set_num_threads(nthreads);
#endif
#pragma omp parallel for schedule(dynamic, 4)
  for(int64_t k = 0; k < nt; k++)
    {
      const int64_t t = idx ? idx[k] : k;
      const int ptype = type[t], tg = cfg->type_to_grav[ptype];
      double ax = 0, ay = 0, az = 0;
      int cnt = 0;
      for(int64_t i = 0; i < n; i++)
        {
          double dx = pos[3 * i] - pos[3 * t], dy = pos[3 * i + 1] - pos[3 * t + 1], dz = pos[3 * i + 2] - pos[3 * t + 2];
          if(cfg->periodic)
            {
              dx = NEAREST(dx);
              dy = NEAREST(dy);
              dz = NEAREST(dz);
            }
          const double r2 = dx * dx + dy * dy + dz * dz;
          if(!(r2 < reach2))
            continue;
          const double r = sqrt(r2);
          const int tab = (int)(asmthfac * r);
          if(tab >= NTAB)
            continue;
          const int sg = cfg->type_to_grav[type[i]];
          double h = cfg->force_softening[ptype], fac;
          if(h < cfg->force_softening[type[i]])
            h = cfg->force_softening[type[i]];
          if(r >= h)
            {
              fac = law_accel_tn(cfg, cfg->law_accel[tg][sg], mass[t], mass[i], r2, r, 1);
              fac -= mass[i] * utor2wpi * table[((size_t)tg * ng + sg) * NTAB + tab];
              fac /= r;
            }
          else
            fac = law_spline_tn(cfg, cfg->law_spline[tg][sg], mass[t], mass[i], h, r, 1);
          ax += dx * fac;
          ay += dy * fac;
          az += dz * fac;
          cnt++;
        }
      acc[3 * k] = ax;
      acc[3 * k + 1] = ay;
      acc[3 * k + 2] = az;
      if(nint)
        nint[k] = cnt;
    }
}
