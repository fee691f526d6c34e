/* ngravs_comm_selftest.h -- every callback of a communicator vtable (include/ngravs_host.h) once, with known answers.
 *
 * Header-only (static functions) so that libngravs_hip.so (ngravs_host_comm_selftest), libngravs_rccl.so (ngravs_rccl_selftest)
 * and a host's own MPI vtable can run the same test without depending on each other.
 *
 * What it is for: a host calls it once after it has built its communicator, so that a fabric or bootstrap problem shows as an
 * error message before the first step and not as a hang inside one.  It replaces nothing in the reference (MPI_Init either works
 * or aborts there); it exists because the multi-task path of this library has only ever met its real fabric at world size 1.
 *
 * Collective-safe by construction: EVERY task runs ALL stages whatever it finds on the way -- a task that left after a wrong
 * answer would let its peers wait in the next collective for ever (the hang the test is there to prevent).  Verdicts are kept in a
 * status word and combined by one last all-reduce, so all tasks return the same value:
 *   0                      every stage as expected on every task
 *   bit 0 / 1 / 2          stage 1 (all-reduce SUM + MIN) / 2 (all-gather) / 3 (all-to-all-v with unequal and empty blocks) gave a
 *                          wrong answer or its callback failed on SOME task
 *   bit 3                  a buffer could not be allocated on some task (that task took part with what it had: stage 3 is then
 *                          skipped by everybody, see below)
 * Stage 3 needs buffers on every task; whether all have them is agreed by the all-reduce that closes stage 2, so that either all
 * tasks enter the all-to-all-v or none does.
 * fail_stage (tests): this task pretends stage 1, 2 or 3 found a wrong answer.  `why` receives this task's own finding (or "").
 */
#ifndef NGRAVS_COMM_SELFTEST_H
#define NGRAVS_COMM_SELFTEST_H

#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "ngravs_host.h"

/* where the all-to-all-v buffers live when the communicator wants device pointers (cm->device_buffers); NULL for host buffers */
typedef struct ngravs_selftest_mem
{
  void *(*alloc)(void *user, size_t bytes);
  void (*release)(void *user, void *p);
  int (*upload)(void *user, void *dev, const void *host, size_t bytes);      /* 0 = ok; complete on return */
  int (*download)(void *user, void *host, const void *dev, size_t bytes);
  int (*fill)(void *user, void *dev, int byte, size_t bytes);                /* memset, complete on return */
  void *user;
} ngravs_selftest_mem;

/* block task `from` sends to task `to`: (from + 2 to) mod 5 words of value 64 from + to; none when (from + to) mod 7 == 3 */
static int64_t ngravs_st_words(int from, int to) { return (from + to) % 7 == 3 ? 0 : (from + 2 * to) % 5; }

static int ngravs_comm_selftest_run(const ngravs_comm *cm, const ngravs_selftest_mem *mem, int fail_stage, char *why, int why_len)
{
  int64_t sum[2], agree, *hs = NULL, *hr = NULL, sb[64], sd[64], rb[64], rd[64], ns = 0, nr = 0, k;
  unsigned char *g = NULL, mine[3];
  void *ds = NULL, *dr = NULL;
  int64_t status = 0;
  int p, W, me, have_buffers;
  if(why && why_len > 0)
    why[0] = 0;
  if(!cm || cm->size < 1 || cm->size > 64 || !cm->allreduce || !cm->allgather || !cm->alltoallv)
    return 15;   /* not a usable vtable: nothing collective has been entered */
  W = cm->size;
  me = cm->rank;
#define NGRAVS_ST_NOTE(...)                      \
  do                                             \
    {                                            \
      if(why && why_len > 0 && !why[0])          \
        snprintf(why, (size_t)why_len, __VA_ARGS__); \
    }                                            \
  while(0)
  /* ---- stage 1: reductions */
  sum[0] = me + 1;
  sum[1] = -(int64_t)me;
  if(cm->allreduce(cm->user, sum, 1, NGRAVS_T_I64, NGRAVS_OP_SUM) | cm->allreduce(cm->user, sum + 1, 1, NGRAVS_T_I64, NGRAVS_OP_MIN))
    {
      status |= 1;
      NGRAVS_ST_NOTE("self test: an all-reduce callback failed");
    }
  else if(sum[0] != (int64_t)W * (W + 1) / 2 || sum[1] != -(int64_t)(W - 1) || fail_stage == 1)
    {
      status |= 1;
      NGRAVS_ST_NOTE("self test: all-reduce gave %lld / %lld", (long long)sum[0], (long long)sum[1]);
    }
  /* ---- stage 2: gather (a task without its receive buffer still sends, into a scratch of its own) */
  g = malloc(3 * (size_t)W);
  mine[0] = (unsigned char)me;
  mine[1] = (unsigned char)(me ^ 0x5a);
  mine[2] = 7;
  if(!g)
    {
      static unsigned char fallback[3 * 64];
      status |= 8;
      NGRAVS_ST_NOTE("self test: out of memory");
      (void)cm->allgather(cm->user, mine, fallback, 3);
    }
  else if(cm->allgather(cm->user, mine, g, 3))
    {
      status |= 2;
      NGRAVS_ST_NOTE("self test: the all-gather callback failed");
    }
  else
    {
      for(p = 0; p < W; p++)
        if(g[3 * p] != (unsigned char)p || g[3 * p + 1] != (unsigned char)(p ^ 0x5a) || g[3 * p + 2] != 7 || fail_stage == 2)
          {
            status |= 2;
            NGRAVS_ST_NOTE("self test: all-gather block %d is wrong", p);
            break;
          }
    }
  /* ---- stage 3: buffers first, then agree that everybody has them */
  for(p = 0; p < W; p++)
    {
      sb[p] = 8 * ngravs_st_words(me, p);
      rb[p] = 8 * ngravs_st_words(p, me);
      sd[p] = 8 * ns;
      rd[p] = 8 * nr;
      ns += ngravs_st_words(me, p);
      nr += ngravs_st_words(p, me);
    }
  hs = malloc(8 * (size_t)(ns + 1));
  hr = calloc((size_t)(nr + 1), 8);
  have_buffers = hs && hr;
  if(have_buffers && cm->device_buffers)
    {
      if(!mem)
        have_buffers = 0;
      else
        {
          ds = mem->alloc(mem->user, 8 * (size_t)(ns + 1));
          dr = mem->alloc(mem->user, 8 * (size_t)(nr + 1));
          have_buffers = ds && dr;
        }
    }
  if(have_buffers)
    {
      for(p = 0, k = 0; p < W; p++)
        {
          int64_t q;
          for(q = 0; q < ngravs_st_words(me, p); q++)
            hs[k++] = 64 * me + p;
        }
      if(cm->device_buffers && (mem->upload(mem->user, ds, hs, 8 * (size_t)ns) | mem->fill(mem->user, dr, 0xff, 8 * (size_t)(nr + 1))))
        have_buffers = 0;
    }
  if(!have_buffers)
    {
      status |= 8;
      NGRAVS_ST_NOTE("self test: no buffers for the all-to-all-v");
    }
  agree = status & 8;
  if(cm->allreduce(cm->user, &agree, 1, NGRAVS_T_I64, NGRAVS_OP_MAX))
    {
      status |= 1;
      agree = 8;   /* nothing can be agreed through a failing all-reduce: do not enter a collective that needs agreement */
      NGRAVS_ST_NOTE("self test: an all-reduce callback failed");
    }
  if(!agree)
    {
      const void *sp = cm->device_buffers ? ds : (void *)hs;
      void *rp = cm->device_buffers ? dr : (void *)hr;
      if(cm->alltoallv(cm->user, sp, sb, sd, rp, rb, rd))
        {
          status |= 4;
          NGRAVS_ST_NOTE("self test: the all-to-all-v callback failed");
        }
      else if(cm->device_buffers && mem->download(mem->user, hr, dr, 8 * (size_t)nr))
        {
          status |= 4;
          NGRAVS_ST_NOTE("self test: reading the all-to-all-v result back failed");
        }
      else
        {
          int bad = fail_stage == 3 ? 0 : -1;
          for(p = 0, k = 0; p < W && bad < 0; p++)
            {
              int64_t q;
              for(q = 0; q < ngravs_st_words(p, me); q++)
                if(hr[k++] != 64 * p + me)
                  {
                    bad = p;
                    break;
                  }
            }
          if(bad >= 0)
            {
              status |= 4;
              NGRAVS_ST_NOTE("self test: all-to-all-v block from task %d is wrong", bad);
            }
        }
    }
  /* ---- the verdict of all tasks (bits are OR-ed as the maximum of each bit: four reductions would do; one sum of bit counts does too) */
  {
    int64_t bits[4];
    for(p = 0; p < 4; p++)
      bits[p] = (status >> p) & 1;
    if(cm->allreduce(cm->user, bits, 4, NGRAVS_T_I64, NGRAVS_OP_MAX))
      status |= 1;
    else
      for(p = 0; p < 4; p++)
        status |= bits[p] << p;
  }
  if(status && why && why_len > 0 && !why[0])
    snprintf(why, (size_t)why_len, "self test: failed on another task (status %d)", (int)status);
#undef NGRAVS_ST_NOTE
  free(g);
  free(hs);
  free(hr);
  if(ds)
    mem->release(mem->user, ds);
  if(dr)
    mem->release(mem->user, dr);
  return (int)status;
}

#endif
