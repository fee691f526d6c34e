// engine.hpp -- internal state of libngravs_hip.so (not part of the C ABI).
//
// Data layout in HBM (all SoA, fp64 as the reference's -DDOUBLEPRECISION build):
//   input columns, caller's order : in_pos[3n] in_mass[n] in_type[n] in_oldacc[n] in_active[n]
//   Peano-sorted particle columns : s_pm[n] = double4{x,y,z,mass}  s_type[n] u8  s_oldacc[n]
//                                   s_active[n] u8  s_key[n] u64 (21 bits/dim)  s_idx[n] u32
//   tree (breadth-first, level-contiguous): n_first/n_count (particle range), n_child[8*nodes]
//       (>=0 node, -1 empty, <=-2 particle -2-p), n_geo = double4{cx,cy,cz,len},
//       n_mom[nodes*NG] = double4{sx,sy,sz,mass}, n_flags (reference bitflags bits 2-5), n_level
//   results, Peano order          : r_acc[3n] r_nint[n] r_pm[3n] r_oldacc[n]
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include "../../include/ngravs_host.h"

#define NG_MAX NGRAVS_MAX_GRAVS
#define NTAB NGRAVS_NTAB
#define TREE_BITS NGRAVS_TREE_BITS
// n_flags bits beyond the reference's bitflags 2-5 (max-softening type, mixed softening)
#define FLAG_BUCKET 64     // bit 6: the node holds its particles directly (deepest level)
#define FLAG_PSEUDO 128    // bit 7: top-level cell whose particles live on another task: global monopoles, no children
#define FLAG_PARTIAL 256   // bit 8: the cell contains particles that are not on this task (never handed over as a leaf)
#define MAX_LEVELS (TREE_BITS + 1)

struct ngravs_ctx;

#define HIP_TRY(ctx, expr)                                                                        \
  do                                                                                              \
    {                                                                                             \
      hipError_t e__ = (expr);                                                                    \
      if(e__ != hipSuccess)                                                                       \
        {                                                                                         \
          ngravs_report(ctx, NGRAVS_ERR_NO_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e__)); \
          return NGRAVS_ERR_NO_DEVICE;                                                            \
        }                                                                                         \
    }                                                                                             \
  while(0)

void ngravs_report(ngravs_ctx *ctx, int code, const std::string &msg);

template <typename T> struct DevBuf
{
  T *p = nullptr;
  size_t cap = 0;
  int ensure(size_t n)
  {
    if(n <= cap)
      return 0;
    if(p)
      (void)hipFree(p);
    p = nullptr;
    cap = 0;
    size_t want = n + n / 16 + 64;
    if(hipMalloc((void **)&p, want * sizeof(T)) != hipSuccess)
      return -1;
    cap = want;
    return 0;
  }
  void release()
  {
    if(p)
      (void)hipFree(p);
    p = nullptr;
    cap = 0;
  }
};

// constants every kernel needs, passed by value (fits kernarg / SGPRs)
struct WalkParams
{
  int ng, periodic, pm, use_theta;
  int nleaf;                // group walk: an opened node with <= nleaf particles hands over its particles directly (0: only buckets)
  int exact_reach;          // group walk: exact fp64 reach test instead of the packed-fp32 pre-test (ngravs_set_tuning)
  double box, boxhalf;
  double theta2;            // ErrTolTheta^2
  double errtol_acc;        // ErrTolForceAcc
  double rcut, rcut2, asmthfac, utor2wpi, reach2;   // TreePM constants (forcetree.c:1708-1711)
  double ym;                // YUKAWA_IMASS / BoxSize (ngravs.c:859)
  double bam_eps;           // BAM_EPSILON (ngravs.c:45-47)
  double fac_intp;          // 2*NGRAVS_EN/BoxSize: lattice-table lookup scale (forcetree.c:3737)
  double fsoft[NGRAVS_NTYPES];
  int t2g[NGRAVS_NTYPES];
  unsigned t2g_packed;   // the same map, 2 bits per type (register-resident lookups)
  // law coefficients [target][source]: accel = m*(cN/r2 + cY*exp(-r ym)(ym/r + 1/r2)); spline = cS*plummer
  double cN[NG_MAX][NG_MAX], cY[NG_MAX][NG_MAX], cS[NG_MAX][NG_MAX];
  // TreePM short-range tables as the evaluation kernel stages them: identical tables of the symmetric wiring are stored once
  int ntab_lds;                 // distinct tables
  int tab_slot[NG_MAX * NG_MAX];   // [target * ng + source] -> slot
  int slot_src[NG_MAX * NG_MAX];   // slot -> a [target * ng + source] index that holds it
  // Yukawa factor through the table bins: exp(-ym r) = E[tab] * P5(-ym (r - tab/asmthfac)), E[tab] = exp(-ym tab/asmthfac)
  // appended to the table buffer (valid while ym * bin width is small, else 0 and exp() is evaluated in full)
  int exp_tab;
  double inv_asmthfac;
  double ec[4];   // (ym / asmthfac)^k / k!, k = 1..4: the polynomial in the bin fraction
  int src_in_box;   // every particle (and so every node centre of mass) lies inside [0, BoxSize]: groups away from the faces skip the image arithmetic
  // the BAM / NGRAVS_ACCUMULATOR family in the group walk (tree-only wirings): law ids [target][source] and the flag that one is wired
  int bam;
  int law_accel[NG_MAX][NG_MAX], law_spline[NG_MAX][NG_MAX];

};

struct TreeView
{
  const int *first, *count, *child, *flags;
  const int *npart;        // particles per species below a node (NGRAVS_ACCUMULATOR, allvars.h:645-648); null unless a BAM law is wired
  const double4 *geo, *mom;
  int nnodes;
  // start table of the group walk (TreePM only): node index of every cell of one complete tree level, [ix][iy][iz]
  const int *ltab;
  int ltab_level;          // 0: none, walks start at the root
  double ltab_corner[3], ltab_cl;
};

// ngravs_set_tuning(): performance / test parameters, set explicitly by the host (no environment variables)
struct Tuning
{
  int walk_fused = 0, walk_waves = 0, walk_lcap = 0, walk_root = 0, walk_compact = 1, walk_spread = 0, walk_exact_reach = 0, walk_sg = 0, walk_nleaf = -1;
  long long walk_batch = 0;
  int pm_notile = 0, pm_fused_gather = 0, pm_tile_gather = 0, pm_tile8 = 0;
  int sort_full = 0;        // Peano order by one radix sort on all 63 key bits (default: top 42 bits + fix-up of the rare ties)
  int tree_levelwise = 0;   // build the tree level by level (the multi-task path) also for single-task trees
  int moments_octet = 0;
  double dd_keep = 0;       // > 0: the next decompositions will be KEPT for some steps -- import for ALL own particles (not only the active ones), own boxes grown by dd_keep x the domain's side
  int walk_ring = 1;        // TreePM evaluation through the ring-pool kernel (kernels_eval.hip); 0: k_walk_group2<...,2>
  int walk_ring_k = 0;      // ... with at most this many slots per wave (0: as many as fit, at most 8)    // moments pass with eight lanes per node (k_moments8; measured slower: 5.7 against 5.0 ms of build at C4)
};

// The global top of the tree for multi-task runs (force_exchange_pseudodata / force_treeupdate_pseudos, forcetree.c:766-996):
// the reference's adaptive TopNodes[] (domain.c:933-1138), the same on every task.  Every task knows, for every top node, the
// GLOBAL particle count and per-species mass / first moments.  The tree build forces the topology of the top tree from the
// global counts, so that it is the single-task tree's; a top LEAF whose particles are not on this task becomes a pseudo node
// (global monopoles, no children).
#define TOP_CW(ng) (7 + 4 * (ng))   // doubles per node: count (leaf sums: work), particles per type [6], per species m, m x, m y, m z
struct TopTree
{
  double total_count = 0;   // particles of all tasks (the root's count of the last ngravs_dd_set_top)
  bool on = false;                     // sums + presence are set: the next tree build forces the global top (ngravs_dd_set_top)
  double import_reach = 0;             // short-range reach (units of Asmth) the import decision was made for (walk mode at set_top)
  ngravs_toptree h = {0, 0, 0, 0, nullptr, nullptr, nullptr, nullptr, nullptr};   // host copy: child, level, xyz, leaf numbers
  DevBuf<int> child, leaf;             // per node: first child or -1; leaf number (curve order) or -1
  DevBuf<int> gcnt;                    // per node: global particle count
  DevBuf<unsigned char> info;          // per node: bits 0-2 octant in its parent (x << 2 | y << 1 | z), bit 3 PARTIAL
  DevBuf<double> gsum;                 // per node: TOP_CW doubles
  DevBuf<int> leaf_owner;              // per leaf
  DevBuf<unsigned long long> reqmask;  // per leaf: tasks that asked for its particles
  DevBuf<double> leaf_sums;            // per leaf: TOP_CW doubles of the own particles (+ 1 spare word)
  // ---- kept decomposition (the steps on which domain.c:76 keeps domain and tree: ngravs_host_kept_step) ----
  DevBuf<int> own_leaf;                // per own row: its top leaf at the decomposition (ngravs_dd_pack_leaves); rows stay, positions drift
  long long own_leaf_n = -1;           // own rows it covers (-1: none kept)
  int kept_rank = -1, kept_world = 0;
  std::vector<int> h_leaf_owner;       // host copies of the last decomposition's plan
  std::vector<unsigned char> h_present;
  std::vector<double> h_node_sums;
  DevBuf<int> kept_row;                // kept decomposition: the own row in every slot of the leaf-import records (k_dd_fill's order)
  std::vector<int64_t> kept_counts;    // ... records per receiving task
  long long kept_total = -1;
  DevBuf<double> kept_sums;            // per leaf: TOP_CW + 1 doubles (the last one: the grown side of the leaf's cell, from its owner) + 1 status word
  DevBuf<double> leaf_len;             // per leaf: grown side of its cell (all tasks' maximum), for the pseudo nodes of a refit
};

// slab-decomposed particle mesh of the multi-task path (kernels_pmslab.hip)
struct PmSlab
{
  int world = 0, rank = 0, N = 0, stage = -1;
  int xs = 0, nx = 0, ys = 0, ny = 0;      // own x-slab of the real mesh, own y-slab of the transposed k-space
  int lo[3] = {0, 0, 0}, ext[3] = {0, 0, 0};   // brick: the mesh cells (lo + i) mod N the own particles' CIC clouds touch
  int elo[3] = {0, 0, 0}, eext[3] = {0, 0, 0}; // the same +-2 cells (4-point gradient)
  std::vector<int> bbox;                   // lo[3], ext[3] of every task's brick
  std::vector<int64_t> scount, rcount;     // doubles per peer of the stage being exchanged
  long long edesc_off = 0;                 // stage 3: where the extended-brick plane descriptors start in desc
  DevBuf<double> brick, slab, tbuf, ebrick, fmesh, send, recv;
  DevBuf<long long> desc;
  std::vector<long long> hdesc[2];         // host copies of the descriptor tables in flight (upload_desc)
  int hdesc_turn = 0;
  void *plan2f = nullptr, *plan2i = nullptr, *plan1 = nullptr;
  int plan_N = 0, plan_nx = 0, plan_ny = 0;
  double bytes_sent[4] = {0, 0, 0, 0};     // payload of the last step's four exchanges (this task, bytes)
};

struct ngravs_ctx
{
  ngravs_config_t cfg;
  Tuning tune;
  PmSlab pms;
  TopTree top;
  DevBuf<int> n_top;          // top-tree node a tree node is (multi-task trees; -1: none)
  ngravs_fatal_fn on_fatal = nullptr;
  hipStream_t stream = nullptr;
  double asmth = 0, rcut = 0;
  int64_t n = 0;           // particles in the working set (own + halo copies)
  int64_t n_local = 0;     // own particles: the first n_local of the input columns
  bool extent_override = false;
  double ext_lo[3], ext_hi[3];
  int dd_last_what = -1;
  long long dd_last_sent = 0;
  bool have_particles = false, have_order = false, have_tree = false, have_pm = false, have_acc = false;
  bool pm_parked = false;     // pm_orig holds the caller's GravPM (handed over with ngravs_set_particles)
  double dom[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  double pos_lo[3] = {0, 0, 0}, pos_hi[3] = {-1, -1, -1};   // extent of all positions the current decomposition saw (all tasks)
  int64_t shard_first = 0, shard_count = 0;

  // inputs (caller order)
  DevBuf<double> in_pos, in_mass, in_oldacc;
  DevBuf<double> in_cost;      // P[].GravCost (interactions of the particle's last walk): the work weight of the domain cut
  DevBuf<int> in_type;
  DevBuf<unsigned char> in_active;
  DevBuf<double2> in_rec;      // packed 48-byte records of the caller-order columns (dom_keys_and_sort)
  DevBuf<unsigned long long> in_key;
  DevBuf<long long> in_id;
  // multi-task decomposition scratch
  DevBuf<unsigned long long> dd_mask, dd_counts;
  int sort_low = 35;   // key bits the two-stage sort leaves to its fix-up (35 -> 28 -> 21 -> 0 = plain sort, as runs of ties get too long)
  long long own_order_nlocal = -1, own_order_len = 0;   // s_idx still is the Peano order of the last local decomposition (of own_order_len rows, own_order_nlocal of them own)
  DevBuf<unsigned char> dd_send, dd_recv;
  // sorted
  DevBuf<double4> s_pm;
  DevBuf<unsigned char> s_type, s_active;
  DevBuf<double> s_oldacc;
  DevBuf<unsigned long long> s_key;
  DevBuf<unsigned int> s_idx, idx_iota;
  DevBuf<unsigned char> sort_tmp;
  DevBuf<double> red_tmp;
  // tree
  int64_t max_nodes = 0, nnodes = 0;
  bool tree_refit = false;    // the tree was refit since it was built (cells may have grown)
  bool tree_stale = false;    // ngravs_update_particles was called: the columns of the sorted set are out of date
  int nlevels = 0;
  DevBuf<int> lvl_table;      // see TreeView::ltab
  int lvl_table_level = 0;
  int64_t level_start[MAX_LEVELS + 2];
  DevBuf<int> n_first, n_count, n_child, n_flags, n_nchild;
  DevBuf<double4> n_geo, n_mom;
  DevBuf<int> tb_count;   // one-pass build: per-level node counts of every block of particles
  DevBuf<int> n_npart;        // [nodes][NG] particle counts per species (BAM wirings only)
  DevBuf<int> scan_out;
  DevBuf<unsigned char> scan_tmp;
  DevBuf<int> d_counters;
  DevBuf<int> d_levels;        // tree build: first node / node count of every level (device-resident level table)
  long long level_hint[MAX_LEVELS + 2] = {0};   // level populations of the previous build (launch sizing only)
  // walk
  DevBuf<double> table;       // [ng][ng][NTAB] shortrange_fourier_force
  bool table_ready = false;
  DevBuf<double> lat;         // [ng][ng][3][65^3] Ewald / lattice-sum force corrections (periodic tree-only, periodic direct sum)
  bool lat_ready = false;
  DevBuf<int> walk_stack;     // per-wave scratch
  DevBuf<int> walk_tlist;     // compacted active targets of the shard (individual timesteps), Peano order
  DevBuf<unsigned char> walk_tmp;
  long long walk_ntargets = -1;   // >= 0: the group walk runs over walk_tlist[0..walk_ntargets)
  bool walk_dense_tlist = false;  // ... and that list is the own rows of a multi-task working set, nearly all of them active
  int walk_spread = 0;        // > 1: every group of 64 targets is walked as `spread` sub-groups by the fused kernel
  int walk_sg = 1;            // split walk: groups per traversal unit (shared item lists) of the last launch
  long long walk_unopened = 0;   // top leaves the last group walk wanted opened but had to use as monopoles (not imported)
  int walk_unit_state = 4;    // groups per traversal unit the TreePM walk is in (4, 2 or 1): changed with hysteresis on walk_ia_ratio
  double walk_ia_ratio = 0;   // pairs per target of the last TreePM group walk / what a uniform box of the same mean density gives
                              // (0: no such walk yet): > 1 in clustered sets, where smaller traversal units accept more cells
  bool all_active = true;     // the caller passed no active flags
  DevBuf<int> walk_ovf;       // split walk: groups left to the fused kernel (lists or LIFO outgrew their region)
  DevBuf<int> walk_counters;  // [1] overflow flag, [2] groups in walk_ovf, [3] of them by the LIFO, [8..15] per-XCD group counters, [16..23] 64-bit walk statistics
  DevBuf<double> r_acc, r_pm, r_oldacc;
  DevBuf<int> r_nint;
  // pm
  int pm_plan_n = 0;
  void *fft_fwd = nullptr, *fft_inv = nullptr;   // hipfftHandle (int) boxed
  DevBuf<double> pm_rho;      // [ng][N][N][N+2] real / complex in place
  DevBuf<double> pm_force;     // [N][N][N][3]: finite-difference force mesh of one target species (two-pass gather)
  DevBuf<double> pm_phi;      // [ng][N][N][N+2]
  DevBuf<double> pm_orig;     // GravPM in caller order (persists between PM steps)
  // staging for results
  DevBuf<double> out_tmp;
  DevBuf<float> out_tmpf;
  std::vector<unsigned char> host_stage;
  ngravs_stats_t stats;
  hipEvent_t ev0 = nullptr, ev1 = nullptr, evk0 = nullptr, evk1 = nullptr;
  std::vector<hipEvent_t> ev_batch;   // split walk: 3 events per batch (before traversal, between, after evaluation)
  int walk_batches = 0;               // batches of the last split walk (0: fused kernel)
  int walk_lcap = 0;                  // split walk: item-list capacity per group and species (grown on overflow)
  int walk_scap = 0;                  // split walk: LIFO capacity per group (grown on overflow)
  std::string last_error;
};

// ---- kernels_domain.hip
int dom_find_extent(ngravs_ctx *c);
int dom_keys_and_sort(ngravs_ctx *c);
int dom_keys_only(ngravs_ctx *c, const double *d_pos, int64_t n, const double corner[3], double fac, int bits,
                  long long *d_keys);
int dd_local_extent(ngravs_ctx *c, double lo[3], double hi[3]);
void dd_apply_extent(ngravs_ctx *c, const double lo[3], const double hi[3]);
int dd_set_toptree(ngravs_ctx *c, int nnode, const int *child);
int dd_leaf_sums(ngravs_ctx *c, void **dev_sums, int64_t *count);
int dd_pack(ngravs_ctx *c, int what, const int *leaf_owner, int nranks, int me, int64_t *counts, void **dev_records, int64_t *nrec);
int dd_apply_migration(ngravs_ctx *c, const void *dev_records, int64_t nrec);
int dd_get_dest(ngravs_ctx *c, const int *leaf_owner, int *dest);
int dd_target_bounds(ngravs_ctx *c, double out[2]);
int dd_pack_leaves(ngravs_ctx *c, const unsigned long long *reqmask, int nranks, int me, int64_t *counts, void **dev_records, int64_t *nrec);
int dd_set_top(ngravs_ctx *c, const double *node_sums, const unsigned char *present);
int dd_set_halo(ngravs_ctx *c, const void *dev_records, int64_t nrec);
int dd_leaf_sums_kept(ngravs_ctx *c, void **dev_sums, int64_t *count);
int dd_pack_leaves_kept(ngravs_ctx *c, int64_t *counts, void **dev_records, int64_t *nrec);
int dd_refresh_halo(ngravs_ctx *c, const void *dev_records, int64_t nrec);
int dd_update_top(ngravs_ctx *c, const double *node_sums, const double *leaf_len);
int dd_peano_order_own(ngravs_ctx *c, int force);
int dd_fill_ids(ngravs_ctx *c);
int dd_record_doubles(const ngravs_ctx *c, int what);
// ---- kernels_tree.hip
int tree_build(ngravs_ctx *c);
int tree_moments(ngravs_ctx *c, bool refit, bool counts = false);
int tree_top_leaf_len(ngravs_ctx *c, double *dev_kept_sums, int stride);   // kept steps: the grown sides of the own top leaves' cells
int tree_top_refit(ngravs_ctx *c);   // kept steps: moments of the top nodes from the new global sums, their sides from the leaves' up
int dom_regather(ngravs_ctx *c);
static inline bool cfg_has_bam(const ngravs_config_t &cfg)
{
  for(int i = 0; i < cfg.n_gravs; i++)
    for(int j = 0; j < cfg.n_gravs; j++)
      if(cfg.law_accel[i][j] >= NGRAVS_LAW_BAMBAM || cfg.law_spline[i][j] >= NGRAVS_SPLINE_BAMBAM)
        return true;
  return false;
}
// ---- kernels_walk.hip
void make_walk_params(const ngravs_ctx *c, WalkParams *wp);
int walk_run(ngravs_ctx *c);
int walk_finish(ngravs_ctx *c);
int direct_run(ngravs_ctx *c, const int *d_idx, int64_t nt, double *d_acc);
int direct_run_targets(ngravs_ctx *c, const double4 *d_tpm, const int *d_ttype, int64_t nt, double *d_acc);
// ---- kernels_eval.hip
int eval_ring_slots(const WalkParams &wp, bool yuk, int waves);
int launch_eval_ring(ngravs_ctx *c, const TreeView &tv, const WalkParams &wp, bool yuk, int nblk, int waves, int K, const int *region,
                     const int *gcount, long long g0, long long nb, int lcap, int scap, int S, const int *tlist, int SG, long long t_count);
// ---- kernels_pm.hip
int pm_run(ngravs_ctx *c);
int pm_deposit(ngravs_ctx *c);
int pm_finish(ngravs_ctx *c);
void pm_release(ngravs_ctx *c);
// ---- kernels_pmslab.hip
int pmslab_begin(ngravs_ctx *c, int rank, int world, int bbox[6]);
int pmslab_pack(ngravs_ctx *c, int stage, const int *all_bbox, int64_t *send_counts, int64_t *recv_counts, void **send, void **recv);
int pmslab_unpack(ngravs_ctx *c, int stage);
void pmslab_release(ngravs_ctx *c);
// ---- shortrange_table.cpp
void host_shortrange_table(const ngravs_config_t *cfg, double *force, double *pot);
double cfg_asmth(const ngravs_config_t *c);
double cfg_rcut(const ngravs_config_t *c);
