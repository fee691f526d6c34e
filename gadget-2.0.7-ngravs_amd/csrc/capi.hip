// capi.hip -- the C ABI of include/ngravs_hip.h: orchestration only, no arithmetic of the path.
//
// Mirrors the reference's drivers: gravity_tree (gravtree.c:27-460), long_range_force /
// pmforce_periodic (longrange.c:56, pm_periodic.c:204), domain_Decomposition (domain.c:62-154),
// compute_accelerations (accel.c:24-96), init_grav_maps (ngravs_core.c:201-425).
#include "engine.hpp"
#include "../../include/ngravs_peano.h"
#include <chrono>
#include <cmath>

void ngravs_report(ngravs_ctx *ctx, int code, const std::string &msg)
{
  if(ctx)
    ctx->last_error = msg;
  if(ctx && ctx->on_fatal)
    ctx->on_fatal(code, msg.c_str());
  else
    fprintf(stderr, "task %d: endrun called with an error level of %d (%s)\n", ctx ? ctx->cfg.rank : 0, code, msg.c_str());
}

// ---- small device helpers ---------------------------------------------------------------------
__global__ void k_pack_strided_f64(const unsigned char *src, long long stride, int ncomp, long long n, double *dst)
{
  long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if(i >= n)
    return;
  const double *s = reinterpret_cast<const double *>(src + i * stride);
  for(int k = 0; k < ncomp; k++)
    dst[i * ncomp + k] = s[k];
}
// Type column: values outside 0..5 would index fsoft[] / t2g[] out of bounds; they are counted (the host path
// rejects them before the upload) and clamped
__global__ void k_pack_type(const unsigned char *src, long long stride, long long n, int *dst, int *nbad)
{
  long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if(i >= n)
    return;
  int t = *reinterpret_cast<const int *>(src + i * stride);
  if(t < 0 || t >= NGRAVS_NTYPES)
    {
      atomicAdd(nbad, 1);
      t = t < 0 ? 0 : NGRAVS_NTYPES - 1;
    }
  dst[i] = t;
}
// active flag: only bit 0 is the caller's (bit 1 marks halo copies inside the engine)
__global__ void k_pack_active(const unsigned char *src, long long stride, long long n, unsigned char *dst)
{
  long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if(i < n)
    dst[i] = src[i * stride] ? 1 : 0;
}
// rows of the caller-order result that belong to active particles only (gravtree.c:318-341 touch nothing else)
__global__ void k_copy_masked_f64(const unsigned char *__restrict__ act, long long n, int ncomp, const double *__restrict__ src,
                                  double *__restrict__ dst)
{
  long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if(i >= n || !(act[i] & 1))
    return;
  for(int k = 0; k < ncomp; k++)
    dst[i * ncomp + k] = src[i * ncomp + k];
}
__global__ void k_copy_masked_f32(const unsigned char *__restrict__ act, long long n, const float *__restrict__ src, float *__restrict__ dst)
{
  long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if(i < n && (act[i] & 1))
    dst[i] = src[i];
}
__global__ void k_pack_strided_f32_to_f64(const unsigned char *src, long long stride, long long n, double *dst)
{
  long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if(i < n)
    dst[i] = (double)*reinterpret_cast<const float *>(src + i * stride);
}
// GravCost of the walked particles -> the work-weight column (caller order); other rows keep theirs (gravtree.c:387-392 updates
// P[].GravCost of active particles only)
__global__ void k_cost_update(const unsigned int *__restrict__ idx, const unsigned char *__restrict__ act, const int *__restrict__ nint,
                              long long first, long long count, double *__restrict__ cost)
{
  long long k = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if(k >= count)
    return;
  const long long i = first + k;
  if(act[i] & 1)
    cost[idx[i]] = (double)nint[i];
}
__global__ void k_fill_f64(double *p, long long n, double v)
{
  long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if(i < n)
    p[i] = v;
}
__global__ void k_fill_u8(unsigned char *p, long long n, unsigned char v)
{
  long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if(i < n)
    p[i] = v;
}
// Peano order -> caller order
__global__ void k_unpermute_f64(const unsigned int *idx, long long n, int ncomp, const double *src, double *dst)
{
  long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if(i >= n)
    return;
  long long j = idx[i];
  for(int k = 0; k < ncomp; k++)
    dst[j * ncomp + k] = src[i * ncomp + k];
}
__global__ void k_unpermute_i2f(const unsigned int *idx, long long n, const int *src, float *dst)
{
  long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if(i < n)
    dst[idx[i]] = (float)src[i];
}
__global__ void k_permute_f64(const unsigned int *idx, long long n, int ncomp, const double *src, double *dst)
{
  long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if(i >= n)
    return;
  long long j = idx[i];
  for(int k = 0; k < ncomp; k++)
    dst[i * ncomp + k] = src[j * ncomp + k];
}
// Peano order <- caller order for a column that only exists for the first `lim` (own) rows: imported copies read 0
__global__ void k_permute_f64_lim(const unsigned int *idx, long long n, long long lim, int ncomp, const double *src, double *dst)
{
  long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if(i >= n)
    return;
  long long j = idx[i];
  for(int k = 0; k < ncomp; k++)
    dst[i * ncomp + k] = j < lim ? src[j * ncomp + k] : 0.0;
}
// Peano order -> caller order, own rows only
__global__ void k_unpermute_f64_lim(const unsigned int *idx, long long n, long long lim, int ncomp, const double *src, double *dst)
{
  long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if(i >= n)
    return;
  long long j = idx[i];
  if(j < lim)
    for(int k = 0; k < ncomp; k++)
      dst[j * ncomp + k] = src[i * ncomp + k];
}
__global__ void k_key18(const unsigned long long *k21, long long n, long long *out)
{
  long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if(i < n)
    out[i] = (long long)(k21[i] >> (3 * (TREE_BITS - NGRAVS_BITS_PER_DIMENSION)));
}
__global__ void k_inverse_perm(const unsigned int *idx, long long n, int *inv)
{
  long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if(i < n)
    inv[idx[i]] = (int)i;
}
__global__ void k_map_idx(const int *inv, const int *in, long long n, int *out)
{
  long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if(i < n)
    out[i] = inv[in[i]];
}
#define GRID1(n) dim3((unsigned)(((n) + 255) / 256)), dim3(256)

// ---- lifecycle --------------------------------------------------------------------------------
extern "C" int ngravs_abi_version(void) { return NGRAVS_ABI_VERSION; }

extern "C" const char *ngravs_build_info(void)
{
  static char buf[512];
  snprintf(buf, sizeof(buf),
           "libngravs_hip abi=%d arch=gfx950 N_GRAVS<=%d NTAB=%d TREE_BITS=%d sizeof(config)=%zu sizeof(particles)=%zu "
           "sizeof(stats)=%zu walks=strict,group laws=none,newtonian,neg_newtonian,yukawa,coloyuk,bambam,sourcebambaryon,sourcebaryonbam",
           NGRAVS_ABI_VERSION, NGRAVS_MAX_GRAVS, NGRAVS_NTAB, NGRAVS_TREE_BITS, sizeof(ngravs_config_t),
           sizeof(ngravs_particles_t), sizeof(ngravs_stats_t));
  return buf;
}

extern "C" void ngravs_config_default(ngravs_config_t *cfg)
{
  memset(cfg, 0, sizeof(*cfg));
  cfg->abi_version = NGRAVS_ABI_VERSION;
  cfg->n_gravs = 1;
  cfg->G = 1.0;
  cfg->err_tol_theta = 0.5;
  cfg->err_tol_force_acc = 0.005;
  cfg->yukawa_imass = 60.0;
  cfg->tree_alloc_factor = 0.8;
  cfg->world_size = 1;
  for(int i = 0; i < NGRAVS_MAX_GRAVS; i++)
    for(int j = 0; j < NGRAVS_MAX_GRAVS; j++)
      {
        cfg->law_accel[i][j] = cfg->law_greens[i][j] = cfg->law_normed[i][j] = NGRAVS_LAW_NEWTON;
        cfg->law_spline[i][j] = NGRAVS_SPLINE_PLUMMER;
      }
}

// the sanity checks of init_grav_maps (ngravs_core.c:235-261, 321-424)
static int check_config(const ngravs_config_t *cfg, std::string &why)
{
  if(cfg->abi_version != NGRAVS_ABI_VERSION)
    {
      why = "abi_version mismatch";
      return NGRAVS_ERR_ARG;
    }
  if(cfg->n_gravs < 1 || cfg->n_gravs > NGRAVS_MAX_GRAVS)
    {
      why = "n_gravs out of range";
      return NGRAVS_ERR_ARG;
    }
  if(cfg->pmgrid && !cfg->periodic)
    {
      why = "non-periodic PM is disabled by ngravs itself (ngravs_core.c:235-242)";
      return NGRAVS_ERR_ARG;
    }
  if((cfg->periodic || cfg->pmgrid) && !(cfg->box_size > 0))
    {
      why = "box_size must be > 0";
      return NGRAVS_ERR_ARG;
    }
  if(cfg->pmgrid && cfg->type_to_grav[0] != 0)
    {
      why = "gas must be gravitational species 0 with PMGRID (ngravs_core.c:255-261)";
      return NGRAVS_ERR_WIRING;
    }
  if(cfg->world_size < 1 || cfg->rank < 0 || cfg->rank >= cfg->world_size)
    {
      why = "rank/world_size";
      return NGRAVS_ERR_ARG;
    }
  for(int t = 0; t < NGRAVS_NTYPES; t++)
    if(cfg->type_to_grav[t] < 0 || cfg->type_to_grav[t] >= cfg->n_gravs)
      {
        why = "TypeToGrav entry outside [0,N_GRAVS)";
        return NGRAVS_ERR_WIRING;
      }
  for(int i = 0; i < cfg->n_gravs; i++)
    for(int j = 0; j < cfg->n_gravs; j++)
      {
        if(cfg->law_accel[i][j] < 0 || cfg->law_accel[i][j] >= NGRAVS_LAW_COUNT || cfg->law_spline[i][j] < 0 ||
           cfg->law_spline[i][j] >= NGRAVS_SPLINE_COUNT || cfg->law_greens[i][j] < 0 ||
           cfg->law_greens[i][j] >= NGRAVS_LAW_COUNT || cfg->law_normed[i][j] < 0 || cfg->law_normed[i][j] >= NGRAVS_LAW_COUNT)
          {
            why = "force-law table slot not wired (ngravs_core.c:321-360)";
            return NGRAVS_ERR_WIRING;
          }
        // Newton's third law probe F[i][j](1,1,0.5,3,1) == F[j][i](...) (ngravs_core.c:371-403): equal ids, or the two views of
        // the BAM-baryon pair (sourcebambaryon / sourcebaryonbam agree for unit masses and N = 1)
        auto same = [](int a, int b, int p, int q) { return a == b || (a == p && b == q) || (a == q && b == p); };
        if(!same(cfg->law_accel[i][j], cfg->law_accel[j][i], NGRAVS_LAW_SOURCEBAM, NGRAVS_LAW_TARGETBAM) ||
           !same(cfg->law_spline[i][j], cfg->law_spline[j][i], NGRAVS_SPLINE_SOURCEBAM, NGRAVS_SPLINE_TARGETBAM) ||
           cfg->law_normed[i][j] != cfg->law_normed[j][i] || cfg->law_greens[i][j] != cfg->law_greens[j][i])
          {
            why = "force-law table violates Newton's third law (ngravs_core.c:371-403)";
            return NGRAVS_ERR_WIRING;
          }
      }
  if(cfg_has_bam(*cfg) && (cfg->pmgrid || cfg->periodic))
    {
      why = "the BAM laws have no Green's functions: tree-only, non-periodic runs (ngravs.c:189-194)";
      return NGRAVS_ERR_WIRING;
    }
  return NGRAVS_OK;
}

extern "C" int ngravs_create(const ngravs_config_t *cfg, ngravs_ctx **out)
{
  if(!cfg || !out)
    return NGRAVS_ERR_ARG;
  *out = nullptr;
  std::string why;
  int rc = check_config(cfg, why);
  if(rc != NGRAVS_OK)
    {
      ngravs_report(nullptr, rc, why);
      return rc;
    }
  int ndev = 0;
  if(hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    {
      ngravs_report(nullptr, NGRAVS_ERR_NO_DEVICE, "no HIP device: libngravs_hip has no CPU fallback");
      return NGRAVS_ERR_NO_DEVICE;
    }
  if(cfg->device < 0 || cfg->device >= ndev)
    return NGRAVS_ERR_ARG;
  ngravs_ctx *c = new ngravs_ctx;
  c->cfg = *cfg;
  memset(&c->stats, 0, sizeof(c->stats));
  if(hipSetDevice(cfg->device) != hipSuccess || hipStreamCreate(&c->stream) != hipSuccess ||
     hipEventCreate(&c->ev0) != hipSuccess || hipEventCreate(&c->ev1) != hipSuccess ||
     hipEventCreate(&c->evk0) != hipSuccess || hipEventCreate(&c->evk1) != hipSuccess)
    {
      delete c;
      return NGRAVS_ERR_NO_DEVICE;
    }
  if(cfg->pmgrid)
    {
      c->asmth = cfg_asmth(cfg);
      c->rcut = cfg_rcut(cfg);
    }
  *out = c;
  return NGRAVS_OK;
}

extern "C" void ngravs_destroy(ngravs_ctx *c)
{
  if(!c)
    return;
  (void)hipSetDevice(c->cfg.device);
  (void)hipStreamSynchronize(c->stream);
  pm_release(c);
  pmslab_release(c);
  c->in_cost.release();
  c->in_pos.release();
  c->in_mass.release();
  c->in_oldacc.release();
  c->in_type.release();
  c->in_active.release();
  c->in_key.release();
  c->in_rec.release();
  c->in_id.release();
  c->dd_mask.release();
  c->dd_counts.release();
  c->dd_send.release();
  c->dd_recv.release();
  c->top.child.release();
  c->top.leaf.release();
  c->top.gcnt.release();
  c->top.info.release();
  c->top.gsum.release();
  c->top.leaf_owner.release();
  c->top.reqmask.release();
  c->top.leaf_sums.release();
  ngravs_host_toptree_free(&c->top.h);
  c->n_top.release();
  c->s_pm.release();
  c->s_type.release();
  c->s_active.release();
  c->s_oldacc.release();
  c->s_key.release();
  c->s_idx.release();
  c->idx_iota.release();
  c->sort_tmp.release();
  c->red_tmp.release();
  c->n_first.release();
  c->n_count.release();
  c->n_child.release();
  c->n_flags.release();
  c->n_nchild.release();
  c->n_geo.release();
  c->n_mom.release();
  c->n_npart.release();
  c->scan_out.release();
  c->tb_count.release();
  c->scan_tmp.release();
  c->d_counters.release();
  c->d_levels.release();
  c->table.release();
  c->lat.release();
  c->walk_stack.release();
  c->walk_counters.release();
  c->walk_ovf.release();
  c->walk_tlist.release();
  c->walk_tmp.release();
  c->lvl_table.release();
  c->r_acc.release();
  c->r_pm.release();
  c->r_oldacc.release();
  c->r_nint.release();
  c->pm_rho.release();
  c->pm_phi.release();
  c->pm_force.release();
  c->pm_orig.release();
  c->out_tmp.release();
  c->out_tmpf.release();
  (void)hipEventDestroy(c->ev0);
  (void)hipEventDestroy(c->ev1);
  (void)hipEventDestroy(c->evk0);
  (void)hipEventDestroy(c->evk1);
  for(hipEvent_t e : c->ev_batch)
    (void)hipEventDestroy(e);
  (void)hipStreamDestroy(c->stream);
  delete c;
}

extern "C" void ngravs_set_fatal_handler(ngravs_ctx *ctx, ngravs_fatal_fn fn)
{
  if(ctx)
    ctx->on_fatal = fn;
}

extern "C" int ngravs_set_opening(ngravs_ctx *c, double theta, double errtol)
{
  if(!c)
    return NGRAVS_ERR_ARG;
  c->cfg.err_tol_theta = theta;
  c->cfg.err_tol_force_acc = errtol;
  return NGRAVS_OK;
}

// set_softenings() (gravtree.c:468-518) recomputes All.ForceSoftening[] from the scale factor at the top of every gravity_tree()
// of a comoving run (gravtree.c:50-51).  The walk, the direct sum and the import decision read the new lengths from the next
// call on; the max-softening-type flags of the nodes are those of the last tree build or refit, as in the reference
// (force_update_node_recursive, forcetree.c:704-713, runs at build time only).
extern "C" int ngravs_set_softening(ngravs_ctx *c, const double force_softening[NGRAVS_NTYPES])
{
  if(!c || !force_softening)
    return NGRAVS_ERR_ARG;
  for(int t = 0; t < NGRAVS_NTYPES; t++)
    if(!(force_softening[t] >= 0.0))
      {
        ngravs_report(c, NGRAVS_ERR_ARG, "ngravs_set_softening: negative or NaN softening length");
        return NGRAVS_ERR_ARG;
      }
  for(int t = 0; t < NGRAVS_NTYPES; t++)
    c->cfg.force_softening[t] = force_softening[t];
  return NGRAVS_OK;
}

extern "C" int ngravs_set_walk_mode(ngravs_ctx *c, int mode)
{
  if(!c || (mode != NGRAVS_WALK_STRICT && mode != NGRAVS_WALK_GROUP))
    return NGRAVS_ERR_ARG;
  c->cfg.walk_mode = mode;
  return NGRAVS_OK;
}

extern "C" int ngravs_get_config(ngravs_ctx *c, ngravs_config_t *out)
{
  if(!c || !out)
    return NGRAVS_ERR_ARG;
  *out = c->cfg;
  if(c->cfg.pmgrid)
    {
      out->asmth = c->asmth;
      out->rcut = c->rcut;
    }
  return NGRAVS_OK;
}

extern "C" int ngravs_set_tuning(ngravs_ctx *c, const char *name, double v)
{
  if(!c || !name)
    return NGRAVS_ERR_ARG;
  const std::string k(name);
  const long long iv = (long long)v;
  Tuning &t = c->tune;
  if(k == "walk_fused")
    t.walk_fused = iv != 0;
  else if(k == "walk_batch" && iv >= 0)
    t.walk_batch = iv;
  else if(k == "walk_waves" && iv >= 0 && iv <= 16)
    t.walk_waves = (int)iv;
  else if(k == "walk_lcap" && (iv == 0 || (iv >= 1024 && iv <= 65536)))
    {
      t.walk_lcap = (int)iv;
      c->walk_lcap = 0;   // re-initialised by the next split walk
    }
  else if(k == "walk_root")
    t.walk_root = iv != 0;
  else if(k == "walk_compact")
    t.walk_compact = iv != 0;
  else if(k == "walk_spread" && iv >= 0 && iv <= 64 && (iv & (iv - 1)) == 0)
    t.walk_spread = (int)iv;
  else if(k == "walk_sg" && iv >= 0 && iv <= 16)
    t.walk_sg = (int)iv;
  else if(k == "walk_ring" && iv >= 0 && iv <= 1)
    t.walk_ring = (int)iv;
  else if(k == "walk_ring_k" && iv >= 0 && iv <= 8)
    t.walk_ring_k = (int)iv;
  else if(k == "walk_nleaf" && iv >= -1 && iv <= 8)
    t.walk_nleaf = (int)iv;
  else if(k == "walk_exact_reach")
    t.walk_exact_reach = iv != 0;
  else if(k == "pm_notile")
    t.pm_notile = iv != 0;
  else if(k == "pm_fused_gather")
    t.pm_fused_gather = iv != 0;
  else if(k == "pm_tile_gather")
    t.pm_tile_gather = iv != 0;
  else if(k == "sort_full")
    t.sort_full = iv != 0;
  else if(k == "pm_tile8")
    t.pm_tile8 = iv != 0;
  else if(k == "tree_levelwise")
    t.tree_levelwise = iv != 0;
  else if(k == "dd_keep" && v >= 0 && v <= 0.25)
    t.dd_keep = v;
  else if(k == "moments_octet")
    t.moments_octet = iv != 0;
  else
    {
      ngravs_report(c, NGRAVS_ERR_ARG, "ngravs_set_tuning: unknown name or value out of range: " + k);
      return NGRAVS_ERR_ARG;
    }
  return NGRAVS_OK;
}

extern "C" int ngravs_memcpy(ngravs_ctx *c, void *dst, const void *src, int64_t bytes, int kind)
{
  if(!c || bytes < 0 || kind < 1 || kind > 3 || (bytes > 0 && (!dst || !src)))
    return NGRAVS_ERR_ARG;
  if(bytes == 0)
    return NGRAVS_OK;
  (void)hipSetDevice(c->cfg.device);
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  HIP_TRY(c, hipMemcpy(dst, src, (size_t)bytes, kind == 1 ? hipMemcpyHostToDevice : (kind == 2 ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice)));
  return NGRAVS_OK;
}

extern "C" int ngravs_device_alloc(ngravs_ctx *c, void **ptr, int64_t bytes)
{
  if(!c || !ptr || bytes < 0)
    return NGRAVS_ERR_ARG;
  *ptr = nullptr;
  (void)hipSetDevice(c->cfg.device);
  if(hipMalloc(ptr, (size_t)(bytes > 0 ? bytes : 1)) != hipSuccess)
    return NGRAVS_ERR_NOMEM;
  return NGRAVS_OK;
}

extern "C" int ngravs_device_free(ngravs_ctx *c, void *ptr)
{
  if(!c)
    return NGRAVS_ERR_ARG;
  (void)hipSetDevice(c->cfg.device);
  if(ptr)
    (void)hipFree(ptr);
  return NGRAVS_OK;
}

// ---- data hand-over -----------------------------------------------------------------------------
static int upload_column_f64(ngravs_ctx *c, const void *src, int64_t stride, int ncomp, int64_t n, int on_device, double *dst)
{
  if(on_device)
    {
      if(stride == (int64_t)sizeof(double) * ncomp)
        HIP_TRY(c, hipMemcpyAsync(dst, src, sizeof(double) * ncomp * n, hipMemcpyDeviceToDevice, c->stream));
      else
        hipLaunchKernelGGL(k_pack_strided_f64, GRID1(n), 0, c->stream, (const unsigned char *)src, (long long)stride, ncomp,
                           (long long)n, dst);
      return NGRAVS_OK;
    }
  if(stride == (int64_t)sizeof(double) * ncomp)
    {
      HIP_TRY(c, hipMemcpyAsync(dst, src, sizeof(double) * ncomp * n, hipMemcpyHostToDevice, c->stream));
      HIP_TRY(c, hipStreamSynchronize(c->stream));
      return NGRAVS_OK;
    }
  c->host_stage.resize(sizeof(double) * ncomp * (size_t)n);
  double *h = reinterpret_cast<double *>(c->host_stage.data());
  for(int64_t i = 0; i < n; i++)
    memcpy(h + i * ncomp, (const unsigned char *)src + i * stride, sizeof(double) * ncomp);
  HIP_TRY(c, hipMemcpyAsync(dst, h, sizeof(double) * ncomp * n, hipMemcpyHostToDevice, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  return NGRAVS_OK;
}

static double ev_ms(ngravs_ctx *c);

static int set_particles_impl(ngravs_ctx *c, const ngravs_particles_t *p, bool keep_tree)
{
  if(!c || !p || p->n < 0 || (p->n > 0 && (!p->pos || !p->mass || !p->type)))
    return NGRAVS_ERR_ARG;
  if(p->n == 0 && keep_tree)
    {
      // a task without own particles on a kept step of several tasks: its working set is imported copies (or nothing)
      if(!c->have_particles || c->n_local != 0)
        return NGRAVS_ERR_STATE;
      c->tree_stale = c->have_tree;
      return NGRAVS_OK;
    }
  if(p->n == 0)
    {
      // a task without particles (NumPart = 0): legal with several tasks -- it still owns mesh slabs, takes part in every
      // collective and may receive particles in the next migration.  Nothing to upload.
      (void)hipSetDevice(c->cfg.device);
      c->n = c->n_local = 0;
      c->own_order_nlocal = -1;
      c->sort_low = 35;
      c->all_active = true;
      c->walk_ia_ratio = 0;   // (a new particle set: the walk's unit is chosen afresh)
      c->walk_unit_state = 4;
      c->have_particles = true;
      c->have_order = c->have_tree = c->have_pm = c->have_acc = false;
      c->top.on = false;
      c->pm_parked = false;
      return NGRAVS_OK;
    }
  // (a multi-task working set: the caller's rows are its own particles; the imported copies behind them are refreshed by their
  // owners, ngravs_host_kept_step)
  const bool keep_halo = keep_tree && c->n_local != c->n;
  if(keep_tree && (!c->have_order || !c->have_tree || p->n != c->n_local))
    {
      ngravs_report(c, NGRAVS_ERR_STATE, "ngravs_update_particles: needs a built tree over the same own particles");
      return NGRAVS_ERR_STATE;
    }
  if(p->n >= (1ll << 31) - 64)
    {
      ngravs_report(c, NGRAVS_ERR_ARG, "int particle indices (reference All.MaxPart is int): n must be < 2^31");
      return NGRAVS_ERR_ARG;
    }
  (void)hipSetDevice(c->cfg.device);
  const int64_t n = p->n;
  if(c->in_pos.ensure(3 * n) || c->in_mass.ensure(n) || c->in_oldacc.ensure(n) || c->in_type.ensure(n) || c->in_active.ensure(n) ||
     (!keep_tree && c->in_cost.ensure(n)))
    {
      ngravs_report(c, NGRAVS_ERR_NOMEM, "device allocation failed");
      return NGRAVS_ERR_NOMEM;
    }
  if(!keep_halo)
    c->n = n;
  c->n_local = n;
  if(!keep_tree)
    {
      // The Peano order of the last decomposition (s_idx) is only a visiting order for the per-leaf sums -- any permutation is
      // correct, a good one lets a wave add up before it touches memory.  A host that hands the same rows over again with
      // drifted positions (P[] between two migrations) keeps it; a different row count means other rows.
      if(c->own_order_nlocal != n)
        c->own_order_nlocal = -1;
      c->sort_low = 35;
    }
  int rc;
  if(!keep_tree && (rc = dd_fill_ids(c)))
    return rc;
  if((rc = upload_column_f64(c, p->pos, p->pos_stride, 3, n, p->on_device, c->in_pos.p)))
    return rc;
  if((rc = upload_column_f64(c, p->mass, p->mass_stride, 1, n, p->on_device, c->in_mass.p)))
    return rc;
  if(p->old_acc)
    {
      if((rc = upload_column_f64(c, p->old_acc, p->old_acc_stride, 1, n, p->on_device, c->in_oldacc.p)))
        return rc;
    }
  else
    hipLaunchKernelGGL(k_fill_f64, GRID1(n), 0, c->stream, c->in_oldacc.p, (long long)n, 0.0);
  if(!keep_tree)
    {
      if(p->grav_cost && p->on_device)
        hipLaunchKernelGGL(k_pack_strided_f32_to_f64, GRID1(n), 0, c->stream, (const unsigned char *)p->grav_cost,
                           (long long)p->grav_cost_stride, (long long)n, c->in_cost.p);
      else if(p->grav_cost)
        {
          c->host_stage.resize(sizeof(double) * (size_t)n);
          double *h = reinterpret_cast<double *>(c->host_stage.data());
          for(int64_t i = 0; i < n; i++)
            {
              float f;
              memcpy(&f, (const unsigned char *)p->grav_cost + i * p->grav_cost_stride, sizeof(float));
              h[i] = f;
            }
          HIP_TRY(c, hipMemcpyAsync(c->in_cost.p, h, sizeof(double) * n, hipMemcpyHostToDevice, c->stream));
          HIP_TRY(c, hipStreamSynchronize(c->stream));
        }
      else
        hipLaunchKernelGGL(k_fill_f64, GRID1(n), 0, c->stream, c->in_cost.p, (long long)n, 0.0);
    }
  // Type (int32)
  if(p->on_device)
    {
      if(c->d_counters.ensure(16))
        return NGRAVS_ERR_NOMEM;
      int nbad = 0;
      HIP_TRY(c, hipMemsetAsync(c->d_counters.p + 15, 0, sizeof(int), c->stream));
      hipLaunchKernelGGL(k_pack_type, GRID1(n), 0, c->stream, (const unsigned char *)p->type, (long long)p->type_stride,
                         (long long)n, c->in_type.p, c->d_counters.p + 15);
      HIP_TRY(c, hipMemcpyAsync(&nbad, c->d_counters.p + 15, sizeof(int), hipMemcpyDeviceToHost, c->stream));
      HIP_TRY(c, hipStreamSynchronize(c->stream));
      if(nbad)
        {
          ngravs_report(c, NGRAVS_ERR_ARG, "particle Type outside 0..5");
          return NGRAVS_ERR_ARG;
        }
    }
  else
    {
      c->host_stage.resize(sizeof(int) * (size_t)n);
      int *h = reinterpret_cast<int *>(c->host_stage.data());
      for(int64_t i = 0; i < n; i++)
        {
          memcpy(h + i, (const unsigned char *)p->type + i * p->type_stride, sizeof(int));
          if(h[i] < 0 || h[i] >= NGRAVS_NTYPES)
            {
              ngravs_report(c, NGRAVS_ERR_ARG, "particle Type outside 0..5");
              return NGRAVS_ERR_ARG;
            }
        }
      HIP_TRY(c, hipMemcpyAsync(c->in_type.p, h, sizeof(int) * n, hipMemcpyHostToDevice, c->stream));
      HIP_TRY(c, hipStreamSynchronize(c->stream));
    }
  c->all_active = p->active == nullptr;
  if(p->active)
    {
      if(p->on_device)
        hipLaunchKernelGGL(k_pack_active, GRID1(n), 0, c->stream, (const unsigned char *)p->active,
                           (long long)p->active_stride, (long long)n, c->in_active.p);
      else
        {
          c->host_stage.resize((size_t)n);
          for(int64_t i = 0; i < n; i++)
            c->host_stage[i] = ((const unsigned char *)p->active)[i * p->active_stride] ? 1 : 0;
          HIP_TRY(c, hipMemcpyAsync(c->in_active.p, c->host_stage.data(), (size_t)n, hipMemcpyHostToDevice, c->stream));
          HIP_TRY(c, hipStreamSynchronize(c->stream));
        }
    }
  else
    hipLaunchKernelGGL(k_fill_u8, GRID1(n), 0, c->stream, c->in_active.p, (long long)n, (unsigned char)1);
  HIP_TRY(c, hipGetLastError());
  c->walk_ia_ratio = 0;   // (a new particle set: the walk's unit is chosen afresh)
  c->walk_unit_state = 4;
  c->have_particles = true;
  if(keep_tree)
    c->tree_stale = true;    // same order and topology; columns and moments are refreshed by ngravs_force_update_tree
  else
    {
      c->have_order = c->have_tree = c->have_pm = c->have_acc = false;   // new P[]: nothing carries over ...
      c->top.on = false;   // (the top tree itself is kept: it is the first guess of the next decomposition)
      c->pm_parked = false;
      if(p->grav_pm && c->cfg.pmgrid)
        {
          // ... except P[].GravPM, which the host keeps between PM steps: parked in caller order, permuted into the Peano
          // order by the next ngravs_domain_decomposition (OldAcc on non-PM steps needs it, gravtree.c:318-330)
          if(c->pm_orig.ensure(3 * n))
            return NGRAVS_ERR_NOMEM;
          if((rc = upload_column_f64(c, p->grav_pm, p->grav_pm_stride, 3, n, p->on_device, c->pm_orig.p)))
            return rc;
          c->pm_parked = true;
        }
    }
  return NGRAVS_OK;
}

extern "C" int ngravs_set_particles(ngravs_ctx *c, const ngravs_particles_t *p) { return set_particles_impl(c, p, false); }

extern "C" int ngravs_update_particles(ngravs_ctx *c, const ngravs_particles_t *p) { return set_particles_impl(c, p, true); }

// the drifted tree of predict.c:79-91 + force_update_len(): same decomposition and topology, fresh columns, recomputed
// moments and grown cell sides
extern "C" int ngravs_force_update_tree(ngravs_ctx *c)
{
  if(!c || !c->have_order || !c->have_tree)
    return NGRAVS_ERR_STATE;
  (void)hipSetDevice(c->cfg.device);
  if(c->n == 0)   // an empty working set (a task without particles): no nodes to refit
    {
      c->tree_stale = false;
      return NGRAVS_OK;
    }
  HIP_TRY(c, hipEventRecord(c->ev0, c->stream));
  int rc = dom_regather(c);
  if(rc)
    return rc;
  if((rc = tree_moments(c, true)))
    return rc;
  HIP_TRY(c, hipEventRecord(c->ev1, c->stream));
  c->stats.t_treebuild = ev_ms(c) * 1e-3;
  c->stats.t_domain = c->stats.t_peano = 0;
  c->tree_stale = false;
  c->tree_refit = true;
  return NGRAVS_OK;
}

extern "C" int ngravs_set_old_acc(ngravs_ctx *c, const double *old_acc, int64_t stride, int on_device)
{
  if(!c || !c->have_particles || !old_acc)
    return NGRAVS_ERR_ARG;
  (void)hipSetDevice(c->cfg.device);
  if(c->n_local == 0)
    return NGRAVS_OK;
  // the caller's rows are its OWN particles (NumPart = ngravs_dd_num_local()); imported copies of a multi-task working set are
  // sources only, their OldAcc is never read
  int rc = upload_column_f64(c, old_acc, stride, 1, c->n_local, on_device, c->in_oldacc.p);
  if(rc)
    return rc;
  if(c->have_order)
    hipLaunchKernelGGL(k_permute_f64, GRID1(c->n), 0, c->stream, c->s_idx.p, (long long)c->n, 1, c->in_oldacc.p, c->s_oldacc.p);
  HIP_TRY(c, hipGetLastError());
  return NGRAVS_OK;
}

// ---- the path -------------------------------------------------------------------------------------
static double ev_ms(ngravs_ctx *c)
{
  float ms = 0;
  (void)hipEventSynchronize(c->ev1);
  (void)hipEventElapsedTime(&ms, c->ev0, c->ev1);
  return ms;
}

// P[].GravPM survives between PM steps (it is only rewritten by pmforce_periodic): park the OWN rows of r_pm (Peano order of the
// last decomposition) in caller order before the rows are re-sorted, migrated or joined by new imports.  pm_orig then is a
// caller-order column like in_mass: the migration moves it with the particles (kernels_domain.hip).
static int park_grav_pm(ngravs_ctx *c)
{
  if(!c->have_pm || !c->have_order)
    return NGRAVS_OK;
  if(c->pm_orig.ensure(3 * (c->n_local > 0 ? c->n_local : 1)))
    return NGRAVS_ERR_NOMEM;
  hipLaunchKernelGGL(k_unpermute_f64_lim, GRID1(c->n), 0, c->stream, c->s_idx.p, (long long)c->n, (long long)c->n_local, 3, c->r_pm.p,
                     c->pm_orig.p);
  c->pm_parked = true;
  c->have_pm = false;
  return NGRAVS_OK;
}

static int domain_decomposition_impl(ngravs_ctx *c, bool keep_pm)
{
  if(!c || !c->have_particles)
    return NGRAVS_ERR_STATE;
  (void)hipSetDevice(c->cfg.device);
  if(keep_pm)
    {
      if(int rc = park_grav_pm(c))   // unless this step recomputes GravPM anyway
        return rc;
    }
  else
    c->pm_parked = false;
  // parked GravPM (own rows; handed over with the particles, ngravs_particles_t.grav_pm, or parked above / by
  // ngravs_dd_local_extent): permuted into the new order below; imported copies carry none and are never read (k_finish
  // visits active own rows only)
  c->have_pm = c->pm_parked;
  c->pm_parked = false;
  if(c->n == 0)
    {
      // a task without particles and without imported copies: an empty order
      c->own_order_nlocal = -1;
      c->shard_first = c->shard_count = 0;
      c->have_order = true;
      c->have_tree = c->tree_stale = c->have_pm = false;
      c->stats.t_domain = c->stats.t_peano = 0;
      return NGRAVS_OK;
    }
  HIP_TRY(c, hipEventRecord(c->ev0, c->stream));
  int rc = dom_find_extent(c);
  if(rc)
    return rc;
  HIP_TRY(c, hipEventRecord(c->ev1, c->stream));
  c->stats.t_domain = ev_ms(c) * 1e-3;
  HIP_TRY(c, hipEventRecord(c->ev0, c->stream));
  c->own_order_nlocal = -1;
  rc = dom_keys_and_sort(c);
  if(rc)
    return rc;
  c->own_order_nlocal = c->n_local;   // the per-cell sums of the next multi-task decomposition visit the own rows in this order
  c->own_order_len = c->n;
  HIP_TRY(c, hipEventRecord(c->ev1, c->stream));
  c->stats.t_peano = ev_ms(c) * 1e-3;
  // target shard of this rank: a contiguous segment of the Peano order (domain.c:347-456 cuts the
  // curve into NTask segments; here by particle count)
  int64_t ws = c->cfg.world_size, r = c->cfg.rank;
  int64_t lo = (c->n * r) / ws, hi = (c->n * (r + 1)) / ws;
  lo = (lo / 64) * 64;                       // wave-aligned cuts keep groups identical across world sizes
  hi = (r + 1 == ws) ? c->n : (hi / 64) * 64;
  c->shard_first = lo;
  c->shard_count = hi - lo;
  c->have_order = true;
  c->have_tree = false;   // TreeReconstructFlag = 1 (domain.c:84)
  c->tree_stale = false;
  if(c->have_pm)
    {
      if(c->r_pm.ensure(3 * c->n))
        return NGRAVS_ERR_NOMEM;
      hipLaunchKernelGGL(k_permute_f64_lim, GRID1(c->n), 0, c->stream, c->s_idx.p, (long long)c->n, (long long)c->n_local, 3, c->pm_orig.p,
                         c->r_pm.p);
      HIP_TRY(c, hipGetLastError());
    }
  return NGRAVS_OK;
}

extern "C" int ngravs_domain_decomposition(ngravs_ctx *c) { return domain_decomposition_impl(c, true); }

// The step that follows recomputes GravPM (a PM step: long_range_force() comes before gravity_tree(), accel.c:34-46): the stored
// long-range force need not be parked in caller order, migrated and permuted into the new Peano order only to be overwritten
extern "C" int ngravs_discard_grav_pm(ngravs_ctx *c)
{
  if(!c)
    return NGRAVS_ERR_ARG;
  c->have_pm = false;
  c->pm_parked = false;
  return NGRAVS_OK;
}

extern "C" int64_t ngravs_force_treebuild(ngravs_ctx *c)
{
  if(!c || !c->have_order)
    return NGRAVS_ERR_STATE;
  (void)hipSetDevice(c->cfg.device);
  if(c->n == 0)   // nothing to build a tree of (a task without particles): no nodes, nothing will walk it
    {
      c->nnodes = 0;
      c->have_tree = true;
      c->tree_refit = false;
      c->stats.t_treebuild = 0;
      return 0;
    }
  if(hipEventRecord(c->ev0, c->stream) != hipSuccess)
    return NGRAVS_ERR_NO_DEVICE;
  int rc = tree_build(c);
  if(rc)
    return rc;
  if(hipEventRecord(c->ev1, c->stream) != hipSuccess)
    return NGRAVS_ERR_NO_DEVICE;
  c->stats.t_treebuild = ev_ms(c) * 1e-3;
  c->have_tree = true;
  c->tree_refit = false;
  return c->nnodes;
}

static int ensure_table(ngravs_ctx *c)
{
  if(!c->cfg.pmgrid || c->table_ready)
    return NGRAVS_OK;
  const int ng = c->cfg.n_gravs;
  std::vector<double> h((size_t)(ng * ng + 1) * NTAB);
  host_shortrange_table(&c->cfg, h.data(), nullptr);
  {
    // bin-wise Yukawa factor E[tab] = exp(-ym tab/asmthfac) behind the tables (kernels_walk.hip, WalkParams::exp_tab)
    WalkParams wp;
    make_walk_params(c, &wp);
    for(int t = 0; t < NTAB; t++)
      h[(size_t)ng * ng * NTAB + t] = exp(-wp.ym * (double)t * wp.inv_asmthfac);
  }
  if(c->table.ensure(h.size()))
    return NGRAVS_ERR_NOMEM;
  HIP_TRY(c, hipMemcpyAsync(c->table.p, h.data(), sizeof(double) * h.size(), hipMemcpyHostToDevice, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  c->table_ready = true;
  return NGRAVS_OK;
}

extern "C" int ngravs_gravity_tree(ngravs_ctx *c)
{
  if(!c || !c->have_order)
    return NGRAVS_ERR_STATE;
  (void)hipSetDevice(c->cfg.device);
  int rc;
  if(!c->have_tree)   // gravtree.c:56-67
    {
      int64_t nn = ngravs_force_treebuild(c);
      if(nn < 0)
        return (int)nn;
    }
  else if(c->tree_stale && (rc = ngravs_force_update_tree(c)))   // drifted tree: refresh columns and moments first
    return rc;
  if((rc = ensure_table(c)))
    return rc;
  if(c->n_local == 0)   // no targets on this task
    {
      c->stats.t_treewalk = 0;
      c->stats.walk_kernel_ms = 0;
      c->stats.interactions = 0;
      c->stats.n_active = 0;
      c->have_acc = true;
      return NGRAVS_OK;
    }
  HIP_TRY(c, hipEventRecord(c->ev0, c->stream));
  if((rc = walk_run(c)))
    return rc;
  if((rc = walk_finish(c)))
    return rc;
  HIP_TRY(c, hipEventRecord(c->ev1, c->stream));
  c->stats.t_treewalk = ev_ms(c) * 1e-3;
  float kms = 0;
  (void)hipEventSynchronize(c->evk1);
  (void)hipEventElapsedTime(&kms, c->evk0, c->evk1);
  c->stats.walk_kernel_ms = kms;
  // Nf and interaction sum (gravtree.c:74-78, 408-447)
  double h[2] = {0, 0};   // summed by k_finish
  HIP_TRY(c, hipMemcpyAsync(h, c->red_tmp.p, 2 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  c->stats.interactions = h[0];
  c->stats.n_active = (int64_t)h[1];
  c->have_acc = true;
  if(c->cfg.pmgrid && c->cfg.walk_mode == NGRAVS_WALK_GROUP && h[1] > 0 && c->cfg.box_size > 0)
    {
      // pairs per target against the uniform expectation: the cut sphere's volume times the mean density of ALL tasks' particles
      const double reach = (c->cfg.group_reach > 0 ? c->cfg.group_reach : NGRAVS_GROUP_REACH) * c->asmth;
      const double ntot = c->top.on && c->top.total_count > 0 ? c->top.total_count : (double)c->n;
      const double L = c->cfg.box_size, expect = 4.18879020478639 * reach * reach * reach * ntot / (L * L * L);
      // The figure depends on the unit and the spread the walk used (clustered 2^20 probe: 1857 / 1393 / 1079 pairs per target for
      // units of 4 / 2 / 1 groups, 871 for 1 group of 32 targets): it is kept as the equivalent for units of four groups of 64, and
      // only dense walks update it (a sparse active set says nothing about the next dense step).
      const bool dense = c->walk_ntargets < 0 || c->walk_dense_tlist;
      if(expect > 0 && dense)
        {
          const double f_sg = c->walk_sg >= 4 ? 1.0 : (c->walk_sg >= 2 ? 1857.0 / 1393.0 : 1857.0 / 1079.0);
          const double f_s = c->walk_spread >= 2 ? 1079.0 / 871.0 : 1.0;
          c->walk_ia_ratio = (h[0] / h[1]) / expect * f_sg * f_s;
        }
    }
  if(c->shard_count > 0 && c->extent_override)   // the work weights only matter to a multi-task domain cut
    hipLaunchKernelGGL(k_cost_update, GRID1(c->shard_count), 0, c->stream, c->s_idx.p, c->s_active.p, c->r_nint.p,
                       (long long)c->shard_first, (long long)c->shard_count, c->in_cost.p);
  return NGRAVS_OK;
}

extern "C" int ngravs_pmforce_periodic(ngravs_ctx *c)
{
  if(!c || !c->have_particles)
    return NGRAVS_ERR_STATE;
  (void)hipSetDevice(c->cfg.device);
  int rc;
  if(!c->have_order && (rc = ngravs_domain_decomposition(c)))   // domain.c:66-73: PM steps always re-decompose
    return rc;
  if(!c->have_tree)   // the mesh patches of the tiled deposit / gather are the cells of one tree level
    {
      int64_t nn = ngravs_force_treebuild(c);
      if(nn < 0)
        return (int)nn;
    }
  else if(c->tree_stale && (rc = ngravs_force_update_tree(c)))
    return rc;
  HIP_TRY(c, hipEventRecord(c->ev0, c->stream));
  if((rc = pm_run(c)))
    return rc;
  HIP_TRY(c, hipEventRecord(c->ev1, c->stream));
  c->stats.t_pm = ev_ms(c) * 1e-3;
  return NGRAVS_OK;
}

extern "C" int ngravs_compute_accelerations(ngravs_ctx *c, int pm_step)
{
  if(!c || !c->have_particles)
    return NGRAVS_ERR_STATE;
  int rc;
  if((rc = domain_decomposition_impl(c, !(pm_step && c->cfg.pmgrid))))
    return rc;
  if(pm_step && c->cfg.pmgrid && (rc = ngravs_pmforce_periodic(c)))   // accel.c:34-42
    return rc;
  return ngravs_gravity_tree(c);                                        // accel.c:46
}

// ---- results ----------------------------------------------------------------------------------------
static int download_strided(ngravs_ctx *c, const void *dsrc, size_t elem, int ncomp, int64_t n, void *dst, int64_t stride,
                            int on_device)
{
  const size_t row = elem * ncomp;
  if(on_device)
    {
      if((size_t)stride != row)
        {
          ngravs_report(c, NGRAVS_ERR_ARG, "device outputs must be contiguous");
          return NGRAVS_ERR_ARG;
        }
      HIP_TRY(c, hipMemcpyAsync(dst, dsrc, row * n, hipMemcpyDeviceToDevice, c->stream));
      HIP_TRY(c, hipStreamSynchronize(c->stream));
      return NGRAVS_OK;
    }
  if((size_t)stride == row)
    {
      HIP_TRY(c, hipMemcpyAsync(dst, dsrc, row * n, hipMemcpyDeviceToHost, c->stream));
      HIP_TRY(c, hipStreamSynchronize(c->stream));
      return NGRAVS_OK;
    }
  c->host_stage.resize(row * (size_t)n);
  HIP_TRY(c, hipMemcpyAsync(c->host_stage.data(), dsrc, row * n, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  for(int64_t i = 0; i < n; i++)
    memcpy((unsigned char *)dst + i * stride, c->host_stage.data() + i * row, row);
  return NGRAVS_OK;
}

// device result column (caller order, contiguous) -> the caller's array; with a mask only the rows of active particles
static int deliver(ngravs_ctx *c, const void *dsrc, size_t elem, int ncomp, int64_t n, void *dst, int64_t stride, int on_device,
                   const unsigned char *d_mask)
{
  if(!d_mask)
    return download_strided(c, dsrc, elem, ncomp, n, dst, stride, on_device);
  const size_t row = elem * ncomp;
  if(on_device)
    {
      if((size_t)stride != row)
        {
          ngravs_report(c, NGRAVS_ERR_ARG, "device outputs must be contiguous");
          return NGRAVS_ERR_ARG;
        }
      if(elem == sizeof(double))
        hipLaunchKernelGGL(k_copy_masked_f64, GRID1(n), 0, c->stream, d_mask, (long long)n, ncomp, (const double *)dsrc, (double *)dst);
      else
        hipLaunchKernelGGL(k_copy_masked_f32, GRID1(n), 0, c->stream, d_mask, (long long)n, (const float *)dsrc, (float *)dst);
      HIP_TRY(c, hipStreamSynchronize(c->stream));
      return NGRAVS_OK;
    }
  c->host_stage.resize(row * (size_t)n + (size_t)n);
  unsigned char *hm = c->host_stage.data() + row * (size_t)n;
  HIP_TRY(c, hipMemcpyAsync(c->host_stage.data(), dsrc, row * n, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipMemcpyAsync(hm, d_mask, (size_t)n, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  for(int64_t i = 0; i < n; i++)
    if(hm[i] & 1)
      memcpy((unsigned char *)dst + i * stride, c->host_stage.data() + i * row, row);
  return NGRAVS_OK;
}

extern "C" int ngravs_get_accel(ngravs_ctx *c, double *grav_accel, int64_t accel_stride, double *grav_pm, int64_t pm_stride,
                                double *old_acc, int64_t old_acc_stride, float *grav_cost, int64_t cost_stride, int on_device,
                                int only_active)
{
  if(!c || !c->have_order)
    return NGRAVS_ERR_STATE;
  (void)hipSetDevice(c->cfg.device);
  // the working set is own rows [0, n_local) followed by imported copies; ONLY the own rows are the caller's (its arrays hold
  // NumPart = ngravs_dd_num_local() rows -- the imports of a multi-task step never reach them)
  const int64_t n = c->n, nl = c->n_local;
  if(nl == 0)   // a task without own particles: no rows to deliver
    return NGRAVS_OK;
  if(c->out_tmp.ensure(3 * n) || c->out_tmpf.ensure(n))
    return NGRAVS_ERR_NOMEM;
  // in_active is the caller-order flag column of the last hand-over
  const unsigned char *mask = only_active ? c->in_active.p : nullptr;
  int rc;
  if(grav_accel)
    {
      if(!c->have_acc)
        return NGRAVS_ERR_STATE;
      HIP_TRY(c, hipMemsetAsync(c->out_tmp.p, 0, sizeof(double) * 3 * n, c->stream));
      hipLaunchKernelGGL(k_unpermute_f64, GRID1(n), 0, c->stream, c->s_idx.p, (long long)n, 3, c->r_acc.p, c->out_tmp.p);
      if((rc = deliver(c, c->out_tmp.p, sizeof(double), 3, nl, grav_accel, accel_stride, on_device, mask)))
        return rc;
    }
  if(grav_pm)
    {
      if(!c->have_pm)
        return NGRAVS_ERR_STATE;
      hipLaunchKernelGGL(k_unpermute_f64, GRID1(n), 0, c->stream, c->s_idx.p, (long long)n, 3, c->r_pm.p, c->out_tmp.p);
      if((rc = deliver(c, c->out_tmp.p, sizeof(double), 3, nl, grav_pm, pm_stride, on_device, nullptr)))
        return rc;
    }
  if(old_acc)
    {
      if(!c->have_acc)
        return NGRAVS_ERR_STATE;
      hipLaunchKernelGGL(k_unpermute_f64, GRID1(n), 0, c->stream, c->s_idx.p, (long long)n, 1, c->r_oldacc.p, c->out_tmp.p);
      if((rc = deliver(c, c->out_tmp.p, sizeof(double), 1, nl, old_acc, old_acc_stride, on_device, mask)))
        return rc;
    }
  if(grav_cost)
    {
      if(!c->have_acc)
        return NGRAVS_ERR_STATE;
      hipLaunchKernelGGL(k_unpermute_i2f, GRID1(n), 0, c->stream, c->s_idx.p, (long long)n, c->r_nint.p, c->out_tmpf.p);
      if((rc = deliver(c, c->out_tmpf.p, sizeof(float), 1, nl, grav_cost, cost_stride, on_device, mask)))
        return rc;
    }
  return NGRAVS_OK;
}

extern "C" int ngravs_get_stats(ngravs_ctx *c, ngravs_stats_t *out)
{
  if(!c || !out)
    return NGRAVS_ERR_ARG;
  *out = c->stats;
  return NGRAVS_OK;
}

// ---- kept decomposition (include/ngravs_hip.h, ngravs_host_kept_step)
static std::string kept_state(const ngravs_ctx *c, const char *who)
{
  char b[320];
  snprintf(b, sizeof(b), "%s: no kept decomposition to work on (own rows %lld, rows at the cut %lld, working set %lld, tree %d, top %d, task %d of %d)", who,
           (long long)c->n_local, (long long)c->top.own_leaf_n, (long long)c->n, (int)c->have_tree, (int)c->top.on, c->top.kept_rank, c->top.kept_world);
  return std::string(b);
}
extern "C" int ngravs_dd_leaf_sums_kept(ngravs_ctx *c, void **dev_sums, int64_t *count)
{
  if(!c || !dev_sums || !count)
    return NGRAVS_ERR_ARG;
  (void)hipSetDevice(c->cfg.device);
  const int rc = dd_leaf_sums_kept(c, dev_sums, count);
  if(rc == NGRAVS_ERR_STATE)
    ngravs_report(c, rc, kept_state(c, "ngravs_dd_leaf_sums_kept"));
  return rc;
}
extern "C" int ngravs_dd_pack_leaves_kept(ngravs_ctx *c, int64_t *counts, void **dev_records, int64_t *nrec)
{
  if(!c || !counts || !dev_records || !nrec)
    return NGRAVS_ERR_ARG;
  (void)hipSetDevice(c->cfg.device);
  const int rc = dd_pack_leaves_kept(c, counts, dev_records, nrec);
  if(rc == NGRAVS_ERR_STATE)
    ngravs_report(c, rc, kept_state(c, "ngravs_dd_pack_leaves_kept"));
  return rc;
}
extern "C" int ngravs_dd_refresh_halo(ngravs_ctx *c, const void *dev_records, int64_t nrec)
{
  if(!c || nrec < 0 || (nrec > 0 && !dev_records))
    return NGRAVS_ERR_ARG;
  (void)hipSetDevice(c->cfg.device);
  const int rc = dd_refresh_halo(c, dev_records, nrec);
  if(rc == NGRAVS_ERR_STATE)
    ngravs_report(c, rc, kept_state(c, "ngravs_dd_refresh_halo"));
  return rc;
}
extern "C" int ngravs_dd_update_top(ngravs_ctx *c, const double *node_sums, const double *leaf_len)
{
  if(!c)
    return NGRAVS_ERR_ARG;
  (void)hipSetDevice(c->cfg.device);
  const int rc = dd_update_top(c, node_sums, leaf_len);
  if(rc == NGRAVS_ERR_STATE)
    ngravs_report(c, rc, kept_state(c, "ngravs_dd_update_top"));
  return rc;
}
extern "C" int ngravs_dd_get_kept(ngravs_ctx *c, int32_t *rank, int32_t *world, const int32_t **leaf_owner, const uint8_t **present,
                                  const double **node_sums)
{
  if(!c)
    return NGRAVS_ERR_ARG;
  const TopTree &t = c->top;
  if(!(t.on && t.h.nnode > 0 && t.own_leaf_n == c->n_local && t.kept_rank >= 0 && (int)t.h_leaf_owner.size() == t.h.nleaf &&
       (int)t.h_present.size() == t.h.nleaf))
    {
      ngravs_report(c, NGRAVS_ERR_STATE, kept_state(c, "ngravs_dd_get_kept"));
      return NGRAVS_ERR_STATE;
    }
  if(rank)
    *rank = t.kept_rank;
  if(world)
    *world = t.kept_world;
  if(leaf_owner)
    *leaf_owner = t.h_leaf_owner.data();
  if(present)
    *present = t.h_present.data();
  if(node_sums)
    *node_sums = t.h_node_sums.data();
  return NGRAVS_OK;
}

extern "C" int ngravs_walk_unopened(ngravs_ctx *c, int64_t *count)
{
  if(!c || !count)
    return NGRAVS_ERR_ARG;
  *count = (int64_t)c->walk_unopened;
  return NGRAVS_OK;
}

extern "C" int ngravs_get_domain(ngravs_ctx *c, double out[8])
{
  if(!c || !c->have_order)
    return NGRAVS_ERR_STATE;
  memcpy(out, c->dom, sizeof(double) * 8);
  return NGRAVS_OK;
}

extern "C" int ngravs_get_keys(ngravs_ctx *c, int64_t *keys, int on_device)
{
  if(!c || !c->have_order || !keys)
    return NGRAVS_ERR_STATE;
  (void)hipSetDevice(c->cfg.device);
  const int64_t n = c->n;
  if(c->out_tmp.ensure(3 * n))
    return NGRAVS_ERR_NOMEM;
  long long *tmp = reinterpret_cast<long long *>(c->out_tmp.p);
  hipLaunchKernelGGL(k_key18, GRID1(n), 0, c->stream, c->in_key.p, (long long)n, tmp);
  return download_strided(c, tmp, sizeof(long long), 1, c->n_local, keys, sizeof(long long), on_device);   // own rows
}

extern "C" int ngravs_get_order(ngravs_ctx *c, int32_t *order, int on_device)
{
  if(!c || !c->have_order || !order)
    return NGRAVS_ERR_STATE;
  (void)hipSetDevice(c->cfg.device);
  return download_strided(c, c->s_idx.p, sizeof(int), 1, c->n, order, sizeof(int), on_device);
}

extern "C" int ngravs_get_shard(ngravs_ctx *c, int64_t *first, int64_t *count)
{
  if(!c || !c->have_order)
    return NGRAVS_ERR_STATE;
  if(first)
    *first = c->shard_first;
  if(count)
    *count = c->shard_count;
  return NGRAVS_OK;
}

extern "C" const char *ngravs_last_error(ngravs_ctx *c) { return c ? c->last_error.c_str() : ""; }

// ---- stand-alone pieces -------------------------------------------------------------------------------
extern "C" int64_t ngravs_peano_hilbert_key(int x, int y, int z, int bits) { return ngravs_ph_key(x, y, z, bits); }

extern "C" int ngravs_peano_keys(ngravs_ctx *c, const double *pos, int64_t n, const double corner[3], double fac, int bits,
                                 int64_t *keys)
{
  if(!c || !pos || !keys || n <= 0 || bits < 1 || bits > 21)
    return NGRAVS_ERR_ARG;
  (void)hipSetDevice(c->cfg.device);
  DevBuf<double> dpos;
  DevBuf<long long> dkeys;
  if(dpos.ensure(3 * n) || dkeys.ensure(n))
    return NGRAVS_ERR_NOMEM;
  int rc = NGRAVS_OK;
  if(hipMemcpyAsync(dpos.p, pos, sizeof(double) * 3 * n, hipMemcpyHostToDevice, c->stream) != hipSuccess)
    rc = NGRAVS_ERR_NO_DEVICE;
  if(!rc)
    rc = dom_keys_only(c, dpos.p, n, corner, fac, bits, dkeys.p);
  if(!rc && hipMemcpyAsync(keys, dkeys.p, sizeof(long long) * n, hipMemcpyDeviceToHost, c->stream) != hipSuccess)
    rc = NGRAVS_ERR_NO_DEVICE;
  (void)hipStreamSynchronize(c->stream);
  dpos.release();
  dkeys.release();
  return rc;
}

extern "C" int ngravs_shortrange_table(const ngravs_config_t *cfg, double *force_out, double *pot_out)
{
  if(!cfg || !force_out || cfg->n_gravs < 1 || cfg->n_gravs > NGRAVS_MAX_GRAVS)
    return NGRAVS_ERR_ARG;
  host_shortrange_table(cfg, force_out, pot_out);
  return NGRAVS_OK;
}

extern "C" int ngravs_direct_sum(ngravs_ctx *c, const int32_t *idx, int64_t nt, double *acc)
{
  if(!c || !c->have_order || !idx || !acc || nt <= 0)
    return NGRAVS_ERR_STATE;
  (void)hipSetDevice(c->cfg.device);
  DevBuf<int> inv, din, dout;
  DevBuf<double> dacc;
  if(inv.ensure(c->n) || din.ensure(nt) || dout.ensure(nt) || dacc.ensure(3 * nt))
    return NGRAVS_ERR_NOMEM;
  int rc = NGRAVS_OK;
  hipLaunchKernelGGL(k_inverse_perm, GRID1(c->n), 0, c->stream, c->s_idx.p, (long long)c->n, inv.p);
  if(hipMemcpyAsync(din.p, idx, sizeof(int) * nt, hipMemcpyHostToDevice, c->stream) != hipSuccess)
    rc = NGRAVS_ERR_NO_DEVICE;
  hipLaunchKernelGGL(k_map_idx, GRID1(nt), 0, c->stream, inv.p, din.p, (long long)nt, dout.p);
  if(!rc)
    rc = direct_run(c, dout.p, nt, dacc.p);
  if(!rc && hipMemcpyAsync(acc, dacc.p, sizeof(double) * 3 * nt, hipMemcpyDeviceToHost, c->stream) != hipSuccess)
    rc = NGRAVS_ERR_NO_DEVICE;
  (void)hipStreamSynchronize(c->stream);
  inv.release();
  din.release();
  dout.release();
  dacc.release();
  return rc;
}

extern "C" int ngravs_direct_sum_targets(ngravs_ctx *c, const double *pos, const double *mass, const int32_t *type, int64_t nt, double *acc)
{
  if(!c || !c->have_order || !pos || !type || !acc || nt <= 0)
    return NGRAVS_ERR_STATE;
  for(int64_t k = 0; k < nt; k++)
    if(type[k] < 0 || type[k] >= NGRAVS_NTYPES)
      return NGRAVS_ERR_ARG;
  (void)hipSetDevice(c->cfg.device);
  DevBuf<double4> dt;
  DevBuf<int> dty;
  DevBuf<double> dacc;
  if(dt.ensure(nt) || dty.ensure(nt) || dacc.ensure(3 * nt))
    return NGRAVS_ERR_NOMEM;
  std::vector<double4> h((size_t)nt);
  for(int64_t k = 0; k < nt; k++)
    {
      h[k].x = pos[3 * k];
      h[k].y = pos[3 * k + 1];
      h[k].z = pos[3 * k + 2];
      h[k].w = mass ? mass[k] : 1.0;
    }
  int rc = NGRAVS_OK;
  if(hipMemcpyAsync(dt.p, h.data(), sizeof(double4) * nt, hipMemcpyHostToDevice, c->stream) != hipSuccess ||
     hipMemcpyAsync(dty.p, type, sizeof(int) * nt, hipMemcpyHostToDevice, c->stream) != hipSuccess)
    rc = NGRAVS_ERR_NO_DEVICE;
  if(!rc)
    rc = direct_run_targets(c, dt.p, dty.p, nt, dacc.p);
  if(!rc && hipMemcpyAsync(acc, dacc.p, sizeof(double) * 3 * nt, hipMemcpyDeviceToHost, c->stream) != hipSuccess)
    rc = NGRAVS_ERR_NO_DEVICE;
  (void)hipStreamSynchronize(c->stream);
  dt.release();
  dty.release();
  dacc.release();
  return rc;
}

// ---- multi-task domain decomposition (host-driven collectives; see kernels_domain.hip) -------------------------------
extern "C" int64_t ngravs_dd_num_local(ngravs_ctx *c) { return c ? c->n_local : NGRAVS_ERR_ARG; }

extern "C" int ngravs_dd_local_extent(ngravs_ctx *c, double lo[3], double hi[3])
{
  if(!c || !c->have_particles)
    return NGRAVS_ERR_STATE;
  (void)hipSetDevice(c->cfg.device);
  if(int rc = park_grav_pm(c))   // GravPM of the own rows outlives the order that is about to go (non-PM steps: gravtree.c:318-330)
    return rc;
  c->n = c->n_local;   // a new step: forget the previous halo
  c->have_order = c->have_tree = false;
  return dd_local_extent(c, lo, hi);
}

extern "C" int ngravs_dd_peano_order(ngravs_ctx *c, int force)
{
  if(!c || !c->have_particles)
    return NGRAVS_ERR_STATE;
  (void)hipSetDevice(c->cfg.device);
  if(int rc = park_grav_pm(c))   // GravPM leaves the old order first: it moves with the rows as a caller-order column
    return rc;
  return dd_peano_order_own(c, force);
}

extern "C" int ngravs_dd_set_extent(ngravs_ctx *c, const double lo[3], const double hi[3])
{
  if(!c)
    return NGRAVS_ERR_ARG;
  c->extent_override = lo && hi;
  if(c->extent_override)
    {
      for(int j = 0; j < 3; j++)
        {
          c->ext_lo[j] = lo[j];
          c->ext_hi[j] = hi[j];
        }
      dd_apply_extent(c, lo, hi);
    }
  return NGRAVS_OK;
}

extern "C" int ngravs_dd_set_toptree(ngravs_ctx *c, int32_t nnode, const int32_t *child)
{
  if(!c || !child || nnode < 1)
    return NGRAVS_ERR_ARG;
  (void)hipSetDevice(c->cfg.device);
  return dd_set_toptree(c, nnode, child);
}

extern "C" int ngravs_dd_get_toptree(ngravs_ctx *c, int32_t *nnode, const int32_t **child)
{
  if(!c || !nnode || !child)
    return NGRAVS_ERR_ARG;
  *nnode = c->top.h.nnode;
  *child = c->top.h.child;
  return NGRAVS_OK;
}

extern "C" int ngravs_host_toptree_borrow(ngravs_ctx *c, ngravs_toptree *view)
{
  if(!c || !view)
    return NGRAVS_ERR_ARG;
  *view = c->top.h;
  return NGRAVS_OK;
}

extern "C" int ngravs_dd_leaf_sums(ngravs_ctx *c, void **dev_sums, int64_t *count)
{
  if(!c || !c->have_particles || !c->extent_override || !dev_sums || !count)
    return NGRAVS_ERR_STATE;
  (void)hipSetDevice(c->cfg.device);
  return dd_leaf_sums(c, dev_sums, count);
}

extern "C" int ngravs_dd_pack(ngravs_ctx *c, int what, const int32_t *leaf_owner, int nranks, int my_rank, int64_t *counts,
                              void **dev_records, int64_t *nrec)
{
  if(!c || !c->have_particles || !c->extent_override || !leaf_owner || !counts || !dev_records || !nrec)
    return NGRAVS_ERR_STATE;
  (void)hipSetDevice(c->cfg.device);
  return dd_pack(c, what, leaf_owner, nranks, my_rank, counts, dev_records, nrec);
}

extern "C" int ngravs_get_domain_extent(ngravs_ctx *c, double out[8])
{
  if(!c || !c->extent_override || !out)
    return NGRAVS_ERR_STATE;
  memcpy(out, c->dom, sizeof(double) * 8);
  return NGRAVS_OK;
}

extern "C" int ngravs_dd_keep_margin(ngravs_ctx *c, double *margin)
{
  if(!c || !margin)
    return NGRAVS_ERR_ARG;
  *margin = c->tune.dd_keep * c->dom[6];
  return NGRAVS_OK;
}

extern "C" int ngravs_dd_target_bounds(ngravs_ctx *c, double out[2])
{
  if(!c || !c->have_particles || !out)
    return NGRAVS_ERR_STATE;
  (void)hipSetDevice(c->cfg.device);
  return dd_target_bounds(c, out);
}

extern "C" int ngravs_dd_pack_leaves(ngravs_ctx *c, const uint64_t *reqmask, int nranks, int my_rank, int64_t *counts, void **dev_records,
                                     int64_t *nrec)
{
  if(!c || !c->have_particles || !c->extent_override || !reqmask || !counts || !dev_records || !nrec)
    return NGRAVS_ERR_STATE;
  (void)hipSetDevice(c->cfg.device);
  return dd_pack_leaves(c, (const unsigned long long *)reqmask, nranks, my_rank, counts, dev_records, nrec);
}

extern "C" int ngravs_dd_set_top(ngravs_ctx *c, const double *node_sums, const uint8_t *present)
{
  if(!c)
    return NGRAVS_ERR_ARG;
  // (periodic tree-only runs: the lattice-correction walk, forcetree.c:2077-2455, opens a node only if the force walk's criterion
  // opens it AND the node is large or straddles the half-box seam -- a subset of what the force walk opens, so the leaves the
  // import decision brings in for the force walk serve it too)
  (void)hipSetDevice(c->cfg.device);
  return dd_set_top(c, node_sums, present);
}

extern "C" int ngravs_dd_get_dest(ngravs_ctx *c, const int32_t *leaf_owner, int32_t *dest)
{
  if(!c || !c->have_particles || !c->extent_override || !leaf_owner || !dest)
    return NGRAVS_ERR_STATE;
  (void)hipSetDevice(c->cfg.device);
  return dd_get_dest(c, leaf_owner, dest);
}

extern "C" int ngravs_dd_recv_buffer(ngravs_ctx *c, int64_t nrec, void **dev_records)
{
  if(!c || nrec < 0 || !dev_records)
    return NGRAVS_ERR_ARG;
  (void)hipSetDevice(c->cfg.device);
  if(c->dd_recv.ensure((size_t)(nrec > 0 ? nrec : 1) * NGRAVS_DD_MAX_RECORD_BYTES))
    return NGRAVS_ERR_NOMEM;
  *dev_records = c->dd_recv.p;
  return NGRAVS_OK;
}

extern "C" int64_t ngravs_dd_record_bytes(ngravs_ctx *c, int what)
{
  return c ? (int64_t)sizeof(double) * dd_record_doubles(c, what) : NGRAVS_ERR_ARG;
}

extern "C" int ngravs_dd_apply_migration(ngravs_ctx *c, const void *dev_records, int64_t nrec)
{
  if(!c || !c->have_particles || nrec < 0)
    return NGRAVS_ERR_STATE;
  (void)hipSetDevice(c->cfg.device);
  return dd_apply_migration(c, dev_records, nrec);
}

extern "C" int ngravs_dd_set_halo(ngravs_ctx *c, const void *dev_records, int64_t nrec)
{
  if(!c || !c->have_particles || nrec < 0)
    return NGRAVS_ERR_STATE;
  (void)hipSetDevice(c->cfg.device);
  return dd_set_halo(c, dev_records, nrec);
}

extern "C" int ngravs_dd_set_ids(ngravs_ctx *c, const int64_t *ids, int on_device)
{
  if(!c || !c->have_particles || !ids)
    return NGRAVS_ERR_STATE;
  (void)hipSetDevice(c->cfg.device);
  if(c->n_local == 0)
    return NGRAVS_OK;
  HIP_TRY(c, hipMemcpyAsync(c->in_id.p, ids, sizeof(long long) * c->n_local, on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice,
                            c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  return NGRAVS_OK;
}

extern "C" int ngravs_dd_get_ids(ngravs_ctx *c, int64_t *ids, int on_device)
{
  if(!c || !c->have_particles || !ids)
    return NGRAVS_ERR_STATE;
  (void)hipSetDevice(c->cfg.device);
  if(c->n_local == 0)
    return NGRAVS_OK;
  return download_strided(c, c->in_id.p, sizeof(long long), 1, c->n_local, ids, sizeof(long long), on_device);
}

// ---- slab-decomposed pmforce_periodic (kernels_pmslab.hip) ------------------------------------------------------------------
extern "C" int ngravs_pm_slab_begin(ngravs_ctx *c, int rank, int world, int32_t bbox[6])
{
  if(!c || !bbox)
    return NGRAVS_ERR_ARG;
  if(!c->have_order)
    return NGRAVS_ERR_STATE;
  (void)hipSetDevice(c->cfg.device);
  if(!c->have_tree)   // the patches of the tiled brick deposit are the cells of one tree level (as in ngravs_pmforce_periodic)
    {
      int64_t nn = ngravs_force_treebuild(c);
      if(nn < 0)
        return (int)nn;
    }
  else if(c->tree_stale)   // drifted particles: the sorted columns are refreshed by the refit
    {
      int rc = ngravs_force_update_tree(c);
      if(rc)
        return rc;
    }
  HIP_TRY(c, hipEventRecord(c->ev0, c->stream));
  int bb[6];
  int rc = pmslab_begin(c, rank, world, bb);
  for(int j = 0; j < 6; j++)
    bbox[j] = bb[j];
  return rc;
}

extern "C" int ngravs_pm_slab_pack(ngravs_ctx *c, int stage, const int32_t *all_bbox, int64_t *send_counts, int64_t *recv_counts,
                                   void **send, void **recv)
{
  if(!c || !send_counts || !recv_counts || !send || !recv)
    return NGRAVS_ERR_ARG;
  (void)hipSetDevice(c->cfg.device);
  return pmslab_pack(c, stage, all_bbox, send_counts, recv_counts, send, recv);
}

extern "C" int ngravs_pm_slab_unpack(ngravs_ctx *c, int stage)
{
  if(!c)
    return NGRAVS_ERR_ARG;
  (void)hipSetDevice(c->cfg.device);
  int rc = pmslab_unpack(c, stage);
  if(rc == NGRAVS_OK && stage == 3)
    {
      HIP_TRY(c, hipEventRecord(c->ev1, c->stream));
      c->stats.t_pm = ev_ms(c) * 1e-3;
    }
  return rc;
}

extern "C" int ngravs_pm_slab_bytes(ngravs_ctx *c, double bytes[4])
{
  if(!c || !bytes)
    return NGRAVS_ERR_ARG;
  for(int k = 0; k < 4; k++)
    bytes[k] = c->pms.bytes_sent[k];
  return NGRAVS_OK;
}
