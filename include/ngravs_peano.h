/* ngravs_peano.h -- Peano-Hilbert key as a table-driven finite-state machine, usable from C, C++
 * and HIP device code.  Replaces peano_hilbert_key() (reference peano.c:356-398): one fused
 * lookup per level (tools/gen_peano_table.py) instead of quadrant lookup + rotation loops.
 * Keys are bit-exact with the reference for every (x,y,z,bits) (tests/test_peano.py KATs from
 * SURVEY.md 8(c)). */
#ifndef NGRAVS_PEANO_H
#define NGRAVS_PEANO_H
#include <stdint.h>
#include "ngravs_peano_table.h"

#if defined(__HIPCC__)
#define NGRAVS_HD __host__ __device__
#else
#define NGRAVS_HD
#endif

/* table pointer is passed in so that device code can keep it in LDS / constant memory */
static inline NGRAVS_HD int64_t ngravs_ph_key_tab(const unsigned short (*step)[8], int x, int y, int z,
                                                  int bits)
{
  int64_t key = 0;
  int state = 0; /* rotation 0, sense +1 */
  for(int lvl = bits - 1; lvl >= 0; lvl--)
    {
      int oct = (((x >> lvl) & 1) << 2) | (((y >> lvl) & 1) << 1) | ((z >> lvl) & 1);
      unsigned v = step[state][oct];
      key = (key << 3) | (int64_t)(v & 7u);
      state = (int)(v >> 3);
    }
  return key;
}

static const unsigned short ngravs_ph_step_host[48][8] = NGRAVS_PH_STEP_INIT;
static inline int64_t ngravs_ph_key(int x, int y, int z, int bits)
{
  return ngravs_ph_key_tab(ngravs_ph_step_host, x, y, z, bits);
}

#endif
