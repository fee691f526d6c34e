/* ngravs_oracle.h -- CPU restatement of the reference's gravity path (TEST INFRASTRUCTURE ONLY).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 * Nothing under gadget-2.0.7-ngravs_amd/ links, imports or calls it.
 *
 * PINNING (see DESIGN.md "Oracle"): the reference itself cannot be built in this image without
 * writing stand-ins for GSL and FFTW-2 headers/libraries (allvars.h:20, ngravs.h:3-11), which
 * this build's rules forbid, so there is no oracle/_ref.  The restatement is pinned by
 *   - the Peano-Hilbert known answers recorded from the reference in SURVEY.md 8(c),
 *   - the reference's recorded tree statistics on its own shipped IC GalaxyCollision.IC
 *     (29 325 nodes, 176 top leaves, 1178.53 / 598.546 interactions per particle)
 *     and on the seeded uniform TreePM boxes (141.313 / 178.133 and 367.5 / 742.7 ia/particle),
 *   - analytic known answers (Newtonian short-range table closed form, plummer spline
 *     continuity, Newton's-third-law probe of ngravs_core.c:371).
 */
#ifndef NGRAVS_ORACLE_H
#define NGRAVS_ORACLE_H
#include "../include/ngravs_hip.h" /* shared plain-C types only: ngravs_config_t, law ids */

#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_tree orc_tree;

int64_t orc_peano_hilbert_key(int x, int y, int z, int bits);            /* peano.c:356-398  */
void orc_domain_extent(const double *pos, int64_t n, double dom[8]);       /* domain.c:882-924 */
void orc_keys(const double *pos, int64_t n, const double dom[8], int64_t *keys); /* domain.c:938-944 */
/* species-major, key-minor order of peano_hilbert_order() (peano.c:36-185) */
void orc_peano_order(const ngravs_config_t *cfg, const int64_t *keys, const int32_t *type, int64_t n,
                     int32_t *order);
/* domain_determineTopTree for NTask=1 (domain.c:933-1138): returns NTopnodes, *ntopleaves */
int orc_toptree_count(const int64_t *keys, int64_t n, int *ntopleaves);

/* force_treebuild (forcetree.c:61-281, 292-336, 451-743) on particles in the given order */
orc_tree *orc_tree_build(const ngravs_config_t *cfg, const double *pos, const double *mass,
                         const int32_t *type, int64_t n, const double dom[8]);
void orc_tree_free(orc_tree *t);
/* dynamic tree update in the reference's semantics: node drift with node velocities (predict.c:79-91) + force_update_len
 * (forcetree.c:1005-1085); newpos must outlive the tree */
void orc_tree_drift(orc_tree *t, const ngravs_config_t *cfg, const double *newpos, const double *vel, double dt);
int64_t orc_tree_numnodes(const orc_tree *t);
int orc_tree_ntopleaves(const orc_tree *t);
/* node record i (0..numnodes): out = len, center[3], then per species g: s[3], mass; flags */
void orc_tree_get_node(const orc_tree *t, int64_t i, double *out, int32_t *bitflags);

/* force_treeevaluate / force_treeevaluate_shortrange for targets idx[0..nt) (NULL => all);
 * acc is in units without G (as the walks leave it), nint = ninteractions.  OpenMP over targets.
 * table = shortrange_fourier_force[tg][sg][NTAB] when cfg->pmgrid != 0. */
int orc_walk(const orc_tree *t, const ngravs_config_t *cfg, const int32_t *idx, int64_t nt,
             const double *old_acc, const double *table, double *acc, int32_t *nint, int nthreads);
/* test instrumentation (not a reference function): reach[p] = min(reach[p], smallest side of a node through which particle p
 * entered the walk of one of the targets; 0 = particle-particle) -- what a task must hold to walk these targets */
int orc_walk_reach(const orc_tree *t, const ngravs_config_t *cfg, const int32_t *idx, int64_t nt, const double *old_acc,
                   const double *table, double *reach);
/* test instrumentation: the reference's short-range pair interaction summed over every particle within `reach` of each target
 * (acc without G, nint = pairs) -- the quantity the production group walk computes where its lists hold particles only */
void orc_direct_shortrange(const ngravs_config_t *cfg, const double *pos, const double *mass, const int32_t *type, int64_t n,
                           const int32_t *idx, int64_t nt, const double *table, double reach, double *acc, int32_t *nint, int nthreads);
/* gravtree.c:318-341: old_acc_out = |acc + pm/G|, acc *= G */
void orc_finish(const ngravs_config_t *cfg, int64_t n, double *acc, const double *pm, double *old_acc_out);

/* forcetree.c:3246-3403 + ngravs_core.c:72-184 */
void orc_shortrange_table(const ngravs_config_t *cfg, double *force, double *pot);
/* pm_periodic.c:204-790 (single rank); gravpm includes G */
int orc_pm_periodic(const ngravs_config_t *cfg, const double *pos, const double *mass,
                    const int32_t *type, int64_t n, double *gravpm);
/* forcetree.c:3428-3548 without the Ewald term; acc xG */
void orc_direct(const ngravs_config_t *cfg, const double *pos, const double *mass, const int32_t *type,
                int64_t n, const int32_t *idx, int64_t nt, double *acc, int nthreads);
/* periodic tree-only path: lattice (Ewald) tables [3][65^3] for one law, the correction walk, direct sum with lattice_corr */
void orc_lattice_table(const ngravs_config_t *cfg, int law, double *tab);
int orc_lattice_walk(const orc_tree *t, const ngravs_config_t *cfg, const int32_t *idx, int64_t nt, const double *old_acc,
                     const double *lat, double *acc, int32_t *nint, int nthreads);
void orc_direct_lattice(const ngravs_config_t *cfg, const double *pos, const double *mass, const int32_t *type, int64_t n,
                        const int32_t *idx, int64_t nt, double *acc, int nthreads, const double *lat);
/* scalar law probes for the KATs: which = 0 accel,1 spline,2 greens,3 normed */
double orc_law_eval(const ngravs_config_t *cfg, int which, int id, double a3, double a4);

#ifdef __cplusplus
}
#endif
#endif
