/* ngravs_comm_rccl.c -- struct ngravs_comm over RCCL, plain C (include/ngravs_comm_rccl.h).
 *
 * One communicator, one private HIP stream.  Every callback is synchronous: the library's pack kernels have finished when
 * ngravs_host.c calls (its ngravs_dd_pack / ngravs_pm_slab_pack return after their stream has drained), and the callback returns
 * after the RCCL stream has drained, so the unpack kernels the library launches next see the data.
 *
 * all-to-all-v: one ncclGroupStart/End bracket of ncclSend/ncclRecv pairs, one pair per peer with a non-empty block; the
 * block a task addresses to itself is a device-to-device copy on the same stream.  xGMI is point-to-point: each pair travels
 * over the direct link of its two GPUs, all pairs at once.
 * Host-memory reductions / gathers (the vtable's allreduce / allgather): staged through a device scratch buffer that grows
 * on demand -- hipMemcpyAsync in, collective, hipMemcpyAsync out on the one stream, one synchronisation.
 */
#define _POSIX_C_SOURCE 199309L
#ifndef __HIP_PLATFORM_AMD__
#define __HIP_PLATFORM_AMD__ 1
#endif
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>
#include "ngravs_comm_rccl.h"

struct ngravs_rccl
{
  ncclComm_t comm;
  hipStream_t stream;
  int rank, size, device;
  void *scratch;          /* device */
  size_t scratch_bytes;
  int64_t calls;
  double seconds, bytes;
  char err[256];
};

static double now_s(void)
{
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

#define HIPOK(r, expr)                                                                 \
  do                                                                                   \
    {                                                                                  \
      hipError_t e__ = (expr);                                                         \
      if(e__ != hipSuccess)                                                            \
        {                                                                              \
          snprintf((r)->err, sizeof((r)->err), "%s: %s", #expr, hipGetErrorString(e__)); \
          return 1;                                                                    \
        }                                                                              \
    }                                                                                  \
  while(0)
#define NCCLOK(r, expr)                                                                 \
  do                                                                                    \
    {                                                                                   \
      ncclResult_t e__ = (expr);                                                        \
      if(e__ != ncclSuccess)                                                            \
        {                                                                               \
          snprintf((r)->err, sizeof((r)->err), "%s: %s", #expr, ncclGetErrorString(e__)); \
          return 1;                                                                     \
        }                                                                               \
    }                                                                                   \
  while(0)

static int need_scratch(ngravs_rccl *r, size_t bytes)
{
  if(bytes <= r->scratch_bytes)
    return 0;
  if(r->scratch)
    (void)hipFree(r->scratch);
  r->scratch = NULL;
  r->scratch_bytes = 0;
  bytes += bytes / 4 + 4096;
  HIPOK(r, hipMalloc(&r->scratch, bytes));
  r->scratch_bytes = bytes;
  return 0;
}

static void account(ngravs_rccl *r, double t0, double bytes)
{
  r->calls++;
  r->seconds += now_s() - t0;
  r->bytes += bytes;
}

static ncclRedOp_t red_op(int op) { return op == NGRAVS_OP_SUM ? ncclSum : (op == NGRAVS_OP_MIN ? ncclMin : ncclMax); }
static ncclDataType_t red_type(int dtype) { return dtype == NGRAVS_T_F64 ? ncclDouble : ncclInt64; }

/* in-place reduction of a DEVICE buffer over all tasks */
static int rccl_allreduce_dev(void *user, void *dev, int64_t count, int dtype, int op)
{
  ngravs_rccl *r = user;
  const double t0 = now_s();
  if(count <= 0)
    return 0;
  HIPOK(r, hipSetDevice(r->device));
  NCCLOK(r, ncclAllReduce(dev, dev, (size_t)count, red_type(dtype), red_op(op), r->comm, r->stream));
  HIPOK(r, hipStreamSynchronize(r->stream));
  account(r, t0, 8.0 * (double)count);
  return 0;
}

/* in-place reduction of `count` elements in HOST memory: through the device scratch */
static int rccl_allreduce(void *user, void *buf, int64_t count, int dtype, int op)
{
  ngravs_rccl *r = user;
  const double t0 = now_s();
  const size_t bytes = 8 * (size_t)(count > 0 ? count : 0);
  if(count <= 0)
    return 0;
  HIPOK(r, hipSetDevice(r->device));
  if(need_scratch(r, bytes))
    return 1;
  HIPOK(r, hipMemcpyAsync(r->scratch, buf, bytes, hipMemcpyHostToDevice, r->stream));
  NCCLOK(r, ncclAllReduce(r->scratch, r->scratch, (size_t)count, red_type(dtype), red_op(op), r->comm, r->stream));
  HIPOK(r, hipMemcpyAsync(buf, r->scratch, bytes, hipMemcpyDeviceToHost, r->stream));
  HIPOK(r, hipStreamSynchronize(r->stream));
  account(r, t0, (double)bytes);
  return 0;
}

/* every task contributes `bytes` bytes of HOST memory; recv = size * bytes in task order */
static int rccl_allgather(void *user, const void *send, void *recv, int64_t bytes)
{
  ngravs_rccl *r = user;
  const double t0 = now_s();
  const size_t b = (size_t)(bytes > 0 ? bytes : 0), all = b * (size_t)r->size, off = (b + 255) & ~(size_t)255;   /* gathered blocks start aligned */
  char *d;
  if(b == 0)
    return 0;
  HIPOK(r, hipSetDevice(r->device));
  if(need_scratch(r, off + all))
    return 1;
  d = r->scratch;
  HIPOK(r, hipMemcpyAsync(d, send, b, hipMemcpyHostToDevice, r->stream));
  NCCLOK(r, ncclAllGather(d, d + off, b, ncclInt8, r->comm, r->stream));
  HIPOK(r, hipMemcpyAsync(recv, d + off, all, hipMemcpyDeviceToHost, r->stream));
  HIPOK(r, hipStreamSynchronize(r->stream));
  account(r, t0, (double)all);
  return 0;
}

/* device blocks, counts and displacements in bytes per peer */
static int rccl_alltoallv(void *user, const void *send, const int64_t *sbytes, const int64_t *sdispl, void *recv,
                          const int64_t *rbytes, const int64_t *rdispl)
{
  ngravs_rccl *r = user;
  const double t0 = now_s();
  double moved = 0;
  int p, pairs = 0;
  HIPOK(r, hipSetDevice(r->device));
  if(sbytes[r->rank] != rbytes[r->rank])
    {
      snprintf(r->err, sizeof(r->err), "all-to-all-v: the block a task sends to itself has two sizes");
      return 1;
    }
  if(sbytes[r->rank] > 0)
    HIPOK(r, hipMemcpyAsync((char *)recv + rdispl[r->rank], (const char *)send + sdispl[r->rank], (size_t)sbytes[r->rank],
                            hipMemcpyDeviceToDevice, r->stream));
  for(p = 0; p < r->size; p++)
    if(p != r->rank && (sbytes[p] > 0 || rbytes[p] > 0))
      pairs++;
  if(pairs)
    {
      NCCLOK(r, ncclGroupStart());
      for(p = 0; p < r->size; p++)
        {
          if(p == r->rank)
            continue;
          if(rbytes[p] > 0)
            NCCLOK(r, ncclRecv((char *)recv + rdispl[p], (size_t)rbytes[p], ncclInt8, p, r->comm, r->stream));
          if(sbytes[p] > 0)
            {
              NCCLOK(r, ncclSend((const char *)send + sdispl[p], (size_t)sbytes[p], ncclInt8, p, r->comm, r->stream));
              moved += (double)sbytes[p];
            }
        }
      NCCLOK(r, ncclGroupEnd());
    }
  HIPOK(r, hipStreamSynchronize(r->stream));
  account(r, t0, moved);
  return 0;
}

int ngravs_rccl_unique_id(char id[NGRAVS_RCCL_ID_BYTES])
{
  ncclUniqueId u;
  if(sizeof(u) != NGRAVS_RCCL_ID_BYTES || !id)
    return 1;
  if(ncclGetUniqueId(&u) != ncclSuccess)
    return 1;
  memcpy(id, &u, sizeof(u));
  return 0;
}

int ngravs_rccl_create(const char id[NGRAVS_RCCL_ID_BYTES], int rank, int size, int device, ngravs_rccl **out)
{
  ngravs_rccl *r;
  ncclUniqueId u;
  if(!id || !out || size < 1 || rank < 0 || rank >= size)
    return 1;
  *out = NULL;
  r = calloc(1, sizeof(*r));
  if(!r)
    return 1;
  r->rank = rank;
  r->size = size;
  r->device = device;
  memcpy(&u, id, sizeof(u));
  if(hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&r->stream, hipStreamNonBlocking) != hipSuccess)
    {
      free(r);
      return 1;
    }
  if(ncclCommInitRank(&r->comm, size, u, rank) != ncclSuccess)
    {
      (void)hipStreamDestroy(r->stream);
      free(r);
      return 1;
    }
  *out = r;
  return 0;
}

void ngravs_rccl_fill(ngravs_rccl *r, ngravs_comm *cm)
{
  memset(cm, 0, sizeof(*cm));
  cm->rank = r->rank;
  cm->size = r->size;
  cm->device_buffers = 1;
  cm->user = r;
  cm->allreduce = rccl_allreduce;
  cm->allgather = rccl_allgather;
  cm->alltoallv = rccl_alltoallv;
  cm->allreduce_dev = rccl_allreduce_dev;
}

void ngravs_rccl_destroy(ngravs_rccl *r)
{
  if(!r)
    return;
  (void)hipSetDevice(r->device);
  (void)hipStreamSynchronize(r->stream);
  (void)ncclCommDestroy(r->comm);
  if(r->scratch)
    (void)hipFree(r->scratch);
  (void)hipStreamDestroy(r->stream);
  free(r);
}

void ngravs_rccl_stats(ngravs_rccl *r, int64_t *calls, double *seconds, double *bytes, int reset)
{
  if(!r)
    return;
  if(calls)
    *calls = r->calls;
  if(seconds)
    *seconds = r->seconds;
  if(bytes)
    *bytes = r->bytes;
  if(reset)
    {
      r->calls = 0;
      r->seconds = r->bytes = 0;
    }
}

const char *ngravs_rccl_last_error(ngravs_rccl *r) { return r ? r->err : ""; }

int ngravs_rccl_world(ngravs_rccl *r)
{
  int n = -1;
  if(!r || ncclCommCount(r->comm, &n) != ncclSuccess)
    return -1;
  return n;
}

/* Every collective of the vtable once, with known answers: a reduction, a gather and an all-to-all-v with unequal and empty
 * blocks (block r -> p: (r + 2 p) mod 5 words of value 64 r + p; none when (r + p) mod 7 == 3).  Collective; 0 = all as
 * expected on this task.  A host calls it once after ngravs_rccl_create() -- what it costs is three small collectives -- so
 * that a fabric or bootstrap problem shows as an error message before the first step, not as a hang inside one. */
static int64_t st_words(int from, int to) { return (from + to) % 7 == 3 ? 0 : (from + 2 * to) % 5; }

int ngravs_rccl_selftest(ngravs_rccl *r)
{
  int64_t sum[2], *hs = NULL, *hr = NULL, sb[64], sd[64], rb[64], rd[64], ns = 0, nr = 0, k;
  unsigned char *g = NULL, mine[3];
  void *ds = NULL, *dr = NULL;
  int p, rc = 1;
  if(!r || r->size > 64)
    return 1;
  sum[0] = r->rank + 1;
  sum[1] = -(int64_t)r->rank;
  if(rccl_allreduce(r, sum, 1, NGRAVS_T_I64, NGRAVS_OP_SUM) || rccl_allreduce(r, sum + 1, 1, NGRAVS_T_I64, NGRAVS_OP_MIN))
    return 1;
  if(sum[0] != (int64_t)r->size * (r->size + 1) / 2 || sum[1] != -(int64_t)(r->size - 1))
    {
      snprintf(r->err, sizeof(r->err), "self test: all-reduce gave %lld / %lld", (long long)sum[0], (long long)sum[1]);
      return 1;
    }
  g = malloc(3 * (size_t)r->size);
  if(!g)
    return 1;
  mine[0] = (unsigned char)r->rank;
  mine[1] = (unsigned char)(r->rank ^ 0x5a);
  mine[2] = 7;
  if(rccl_allgather(r, mine, g, 3))
    goto done;
  for(p = 0; p < r->size; p++)
    if(g[3 * p] != (unsigned char)p || g[3 * p + 1] != (unsigned char)(p ^ 0x5a) || g[3 * p + 2] != 7)
      {
        snprintf(r->err, sizeof(r->err), "self test: all-gather block %d is wrong", p);
        goto done;
      }
  for(p = 0; p < r->size; p++)
    {
      sb[p] = 8 * st_words(r->rank, p);
      rb[p] = 8 * st_words(p, r->rank);
      sd[p] = 8 * ns;
      rd[p] = 8 * nr;
      ns += st_words(r->rank, p);
      nr += st_words(p, r->rank);
    }
  hs = malloc(8 * (size_t)(ns + 1));
  hr = calloc((size_t)(nr + 1), 8);
  if(!hs || !hr || hipMalloc(&ds, 8 * (size_t)(ns + 1)) != hipSuccess || hipMalloc(&dr, 8 * (size_t)(nr + 1)) != hipSuccess)
    goto done;
  for(p = 0, k = 0; p < r->size; p++)
    {
      int64_t q;
      for(q = 0; q < st_words(r->rank, p); q++)
        hs[k++] = 64 * r->rank + p;
    }
  if(hipMemcpy(ds, hs, 8 * (size_t)ns, hipMemcpyHostToDevice) != hipSuccess || hipMemset(dr, 0xff, 8 * (size_t)(nr + 1)) != hipSuccess)
    goto done;
  if(rccl_alltoallv(r, ds, sb, sd, dr, rb, rd))
    goto done;
  if(hipMemcpy(hr, dr, 8 * (size_t)nr, hipMemcpyDeviceToHost) != hipSuccess)
    goto done;
  for(p = 0, k = 0; p < r->size; p++)
    {
      int64_t q;
      for(q = 0; q < st_words(p, r->rank); q++)
        if(hr[k++] != 64 * p + r->rank)
          {
            snprintf(r->err, sizeof(r->err), "self test: all-to-all-v block from task %d is wrong", p);
            goto done;
          }
    }
  rc = 0;
done:
  free(g);
  free(hs);
  free(hr);
  if(ds)
    (void)hipFree(ds);
  if(dr)
    (void)hipFree(dr);
  return rc;
}

int ngravs_rccl_barrier(ngravs_rccl *r)
{
  int64_t one = 1;
  return rccl_allreduce(r, &one, 1, NGRAVS_T_I64, NGRAVS_OP_SUM);
}
