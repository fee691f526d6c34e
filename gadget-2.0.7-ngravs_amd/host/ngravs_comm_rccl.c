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
 *
 * No wait is unbounded: every callback waits for its stream by polling (wait_stream), looks at ncclCommGetAsyncError() while it
 * does, and gives up after `timeout_s` seconds (default 300; ngravs_rccl_set_timeout): it then writes the task, the collective and
 * the byte counts per peer to stderr and -- unless the host asked for an error return instead -- ends the process with exit code 86.
 * A collective that does not complete cannot be retried in the same process (the communicator is in an unknown state); a job
 * that hangs until an outer time limit kills it tells nobody where.
 */
#define _POSIX_C_SOURCE 199309L
#ifndef __HIP_PLATFORM_AMD__
#define __HIP_PLATFORM_AMD__ 1
#endif
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>
#include "ngravs_comm_rccl.h"
#include "ngravs_comm_selftest.h"

struct ngravs_rccl
{
  ncclComm_t comm;
  hipStream_t stream;
  int rank, size, device;
  void *scratch;          /* device */
  size_t scratch_bytes;
  int64_t calls;
  double seconds, bytes;
  double timeout_s;       /* limit of every wait */
  int exit_on_timeout;    /* 1 (default): a wait that runs out ends the process (exit code 86) after the message; 0: error return */
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

/* ... inside an ncclGroupStart / ncclGroupEnd bracket: the group is closed before the error is returned */
#define NCCLOK_G(r, expr)                                                               \
  do                                                                                    \
    {                                                                                   \
      ncclResult_t e__ = (expr);                                                        \
      if(e__ != ncclSuccess)                                                            \
        {                                                                               \
          snprintf((r)->err, sizeof((r)->err), "%s: %s", #expr, ncclGetErrorString(e__)); \
          (void)ncclGroupEnd();                                                         \
          return 1;                                                                     \
        }                                                                               \
    }                                                                                   \
  while(0)

/* Wait for the communicator's stream without blocking for ever: poll the stream, look at the communicator's asynchronous error
 * state, give up after timeout_s.  `what` and the byte counts per peer (may be NULL) go into the message. */
static int wait_stream(ngravs_rccl *r, const char *what, const int64_t *sbytes, const int64_t *rbytes)
{
  const double t0 = now_s();
  long spins = 0;
  for(;;)
    {
      hipError_t e = hipStreamQuery(r->stream);
      if(e == hipSuccess)
        return 0;
      if(e != hipErrorNotReady)
        {
          snprintf(r->err, sizeof(r->err), "%s: hipStreamQuery: %s", what, hipGetErrorString(e));
          return 1;
        }
      if((++spins & 255) == 0)
        {
          ncclResult_t ae = ncclSuccess;
          double waited = now_s() - t0;
          if(ncclCommGetAsyncError(r->comm, &ae) == ncclSuccess && ae != ncclSuccess && ae != ncclInProgress)
            {
              snprintf(r->err, sizeof(r->err), "%s: RCCL reports an asynchronous error: %s", what, ncclGetErrorString(ae));
              return 1;
            }
          if(waited > r->timeout_s)
            {
              char msg[1024];
              int n = snprintf(msg, sizeof(msg), "ngravs_rccl: task %d of %d: %s did not complete within %.0f s", r->rank, r->size, what, r->timeout_s);
              if(sbytes && rbytes)
                {
                  int p;
                  n += snprintf(msg + n, sizeof(msg) - (size_t)n, "; bytes to / from peer:");
                  for(p = 0; p < r->size && n < (int)sizeof(msg) - 40; p++)
                    n += snprintf(msg + n, sizeof(msg) - (size_t)n, " %d:%lld/%lld", p, (long long)sbytes[p], (long long)rbytes[p]);
                }
              snprintf(r->err, sizeof(r->err), "%s: no completion within %.0f s", what, r->timeout_s);
              fprintf(stderr, "%s\n", msg);
              fflush(stderr);
              if(r->exit_on_timeout)
                _exit(86);
              return 1;
            }
          if(waited > 2e-4)   /* a collective of this path takes tens of microseconds: past that, stop burning the core */
            {
              struct timespec ts = {0, 20000};
              (void)nanosleep(&ts, NULL);
            }
        }
    }
}

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
  if(wait_stream(r, "all-reduce (device buffer)", NULL, NULL))
    return 1;
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
  if(wait_stream(r, "all-reduce", NULL, NULL))
    return 1;
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
  if(wait_stream(r, "all-gather", NULL, NULL))
    return 1;
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
            NCCLOK_G(r, ncclRecv((char *)recv + rdispl[p], (size_t)rbytes[p], ncclInt8, p, r->comm, r->stream));
          if(sbytes[p] > 0)
            {
              NCCLOK_G(r, ncclSend((const char *)send + sdispl[p], (size_t)sbytes[p], ncclInt8, p, r->comm, r->stream));
              moved += (double)sbytes[p];
            }
        }
      NCCLOK(r, ncclGroupEnd());
    }
  if(wait_stream(r, "all-to-all-v", sbytes, rbytes))
    return 1;
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
  r->timeout_s = 300.0;
  r->exit_on_timeout = 1;
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

void ngravs_rccl_set_timeout(ngravs_rccl *r, double seconds, int exit_on_timeout)
{
  if(!r)
    return;
  if(seconds > 0)
    r->timeout_s = seconds;
  r->exit_on_timeout = exit_on_timeout ? 1 : 0;
}

/* Every collective of the vtable once, with known answers (include/ngravs_comm_selftest.h): reductions, a gather and an
 * all-to-all-v with unequal and empty blocks.  Collective, and collective-SAFE: every task runs every stage whatever it finds, the
 * verdicts are combined by a last all-reduce, all tasks return the same status (0 = all as expected everywhere).  The buffers of the
 * all-to-all-v are filled on the communicator's own stream (a memset on the null stream could land after the receives). */
static void *rst_alloc(void *user, size_t bytes)
{
  ngravs_rccl *r = user;
  void *p = NULL;
  if(hipSetDevice(r->device) != hipSuccess || hipMalloc(&p, bytes ? bytes : 1) != hipSuccess)
    return NULL;
  return p;
}
static void rst_release(void *user, void *p)
{
  (void)user;
  (void)hipFree(p);
}
static int rst_upload(void *user, void *dev, const void *host, size_t bytes)
{
  ngravs_rccl *r = user;
  if(bytes && hipMemcpyAsync(dev, host, bytes, hipMemcpyHostToDevice, r->stream) != hipSuccess)
    return 1;
  return wait_stream(r, "self test upload", NULL, NULL);
}
static int rst_download(void *user, void *host, const void *dev, size_t bytes)
{
  ngravs_rccl *r = user;
  if(bytes && hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, r->stream) != hipSuccess)
    return 1;
  return wait_stream(r, "self test download", NULL, NULL);
}
static int rst_fill(void *user, void *dev, int byte, size_t bytes)
{
  ngravs_rccl *r = user;
  if(bytes && hipMemsetAsync(dev, byte, bytes, r->stream) != hipSuccess)
    return 1;
  return wait_stream(r, "self test fill", NULL, NULL);
}

int ngravs_rccl_selftest(ngravs_rccl *r)
{
  ngravs_comm cm;
  ngravs_selftest_mem mem = {rst_alloc, rst_release, rst_upload, rst_download, rst_fill, r};
  char why[200];
  int st;
  if(!r || r->size > 64)
    return 1;
  ngravs_rccl_fill(r, &cm);
  st = ngravs_comm_selftest_run(&cm, &mem, 0, why, (int)sizeof(why));
  if(st && why[0] && !strstr(r->err, "self test"))
    snprintf(r->err, sizeof(r->err), "%s", why);
  return st;
}

int ngravs_rccl_barrier(ngravs_rccl *r)
{
  int64_t one = 1;
  return rccl_allreduce(r, &one, 1, NGRAVS_T_I64, NGRAVS_OP_SUM);
}
