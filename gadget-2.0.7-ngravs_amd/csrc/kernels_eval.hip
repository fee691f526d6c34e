// kernels_eval.hip -- evaluation kernel of the TreePM group walk with a RING POOL and per-lane cursors (round 4).
//
// Replaces (reference): the force part of force_treeevaluate_shortrange (forcetree.c:1623-2052; pair interaction :1953-2032)
// for the item lists the traversal kernel (kernels_walk.hip, MODE 1) wrote.  Same interaction set, same arithmetic and the
// same operation order per pair as k_walk_group2<..., 2>; what differs is how the 64 lanes of a wave share the pool.
//
// k_walk_group2<...,2> evaluates the pool in synchronous blocks of 64 entries: every lane walks the bits of its own hit mask and
// the wave leaves the block when the LAST lane is done -- 543 trips per group at C4 for a mean of 374 hits per lane (31 % of the
// force-loop slots idle).  Here the pool is a ring of K slots of 32 entries; every lane has its own cursor (slot address q,
// remaining bits m of its 32-bit mask for that slot) and moves on to the next slot as soon as ITS bits are used up, while the
// stragglers finish.  A slot is recycled when the wave minimum has passed it.  Lanes meet only at the end of a list.
//
// LDS per wave: K slots; per workgroup one terminal stub (read-only after the set-up: all waves share it).  Slot layout relative
// to q = address of the slot (all offsets are positive: they fold into the instructions' offset fields):
//     q +   0 .. 255      nextmask[64]  u32: the mask words of the NEXT block (one per lane), 0 until that block exists
//     q + 256             next          u32: q of the slot that holds the next block; q itself until that block exists
//     q + 271             type byte of the NULL entry (7: unsoftened)
//     q + 272 .. 303      softening type of the 32 entries (u8)
//     q + 304 .. 319      (x, y) of the NULL entry (far away): index -1, what v_ffbl_b32 gives for an empty mask
//     q + 320 .. 831      (x, y) of the 32 entries, 16 bytes each (relative to the group's box centre unless images are taken per pair)
//     q + 832 .. 847      (z, mass) of the NULL entry (massless)
//     q + 848 .. 1359     (z, mass) of the 32 entries
// Two arrays of 16-byte halves instead of 32-byte records: a lane reads the two halves of ITS entry with two ds_read_b128; with
// 32-byte records each of the two instructions touched only every other group of four banks (entry j's first half lives in
// banks 8j .. 8j+3 mod 64), i.e. at best half the LDS width; 16-byte strides use all of it.
// A lane whose bits are used up reads { next, nextmask[lane] } of ITS slot (two independent LDS reads, exec-masked): if the next
// block exists it is there, else it stays where it is with an empty mask and asks again on the next trip.  The last block of a
// list points to the terminal stub (empty masks, next = itself), so "wait until no lane is on the oldest slot" is the only
// loop condition there is.  The fp32 copies of the entry positions (MFMA A operand) are gone: the reach masks of a 32-entry block
// are a 32 x 64 x 6 product  D = (ex, ey, ez, |e|^2, 1, 0) . (-2tx, -2ty, -2tz, 1, |t|^2 - thr, 0)  with the A operand converted
// from the fp64 record when the block is complete (three chained v_mfma_f32_32x32x2_f32 per tile, C = 0).
#include "engine.hpp"
#include "walk_device.hpp"
#include <type_traits>

#define ER_SLOT 1360u        // bytes per slot
#define ER_STUB 848u         // the terminal stub: a slot's head + the two NULL halves where a slot has them
#define ER_NM 0u             // nextmask[lane]
#define ER_NEXT 256u         // next
#define ER_TYPE 272u         // type bytes (the NULL entry's at -1)
#define ER_XY 320u           // (x, y) of entry 0; the NULL entry's 16 bytes in front
#define ER_ZM 848u           // (z, mass) of entry 0; the NULL entry's 16 bytes in front
#ifndef ER_MAXWAVES
#define ER_MAXWAVES 16   // waves per workgroup the kernel is compiled for (128 VGPRs at 16; -DER_MAXWAVES=12: 168)
#endif

extern __shared__ __attribute__((aligned(16))) unsigned char er_smem[];

// LDS accesses by absolute 32-bit LDS address (the address of er_smem is part of it: added once per wave, not per access)
#define ER_AS3(T, a) (*(__attribute__((address_space(3))) T *)(unsigned long)(a))
__device__ __forceinline__ unsigned lds_u32(unsigned a) { return ER_AS3(const unsigned, a); }
__device__ __forceinline__ void lds_st_u32(unsigned a, unsigned v) { ER_AS3(unsigned, a) = v; }
__device__ __forceinline__ unsigned char lds_u8(unsigned a) { return ER_AS3(const unsigned char, a); }
__device__ __forceinline__ void lds_st_u8(unsigned a, unsigned char v) { ER_AS3(unsigned char, a) = v; }
// entry j of the slot at q (j = -1: the NULL entry)
__device__ __forceinline__ double4 er_entry(unsigned q, int j)
{
  typedef double d2 __attribute__((ext_vector_type(2)));
  const unsigned a = q + ((unsigned)j << 4);
  const d2 lo = ER_AS3(const d2, a + ER_XY), hi = ER_AS3(const d2, a + ER_ZM);
  double4 r;
  r.x = lo.x;
  r.y = lo.y;
  r.z = hi.x;
  r.w = hi.y;
  return r;
}

// ---- the assembly blocks (eval_asm.inc, generated by tools/gen_eval_asm.py from templates with named registers) -------------------
// ER_TRIP_ASM  the trip loop.  One trip: every lane takes the lowest set bit j of its mask m (none: j = -1, the NULL entry), reads
//              entry j of its slot q, clears the bit, and -- if that was its last bit -- reads { next, nextmask[lane] } of its slot
//              under an exec mask; then the pair interaction of forcetree.c:1953-2032 in the formulation of k_walk_group2's force
//              loop (one Newton step on v_rsq_f64, table bin = (int)(asmthfac r), Yukawa factor E[bin] * P4(bin fraction), softened
//              pairs and slots beyond the exact cut under wave-level branches).  The loop ends when no lane is left on the slot `tail`.
// ER_CULL_ASM  the cull of one chunk: 64 records (one per lane, massless when the lane has none) against the group's box, the
//              survivors compacted behind the wr_pos entries that wait from slot q0 on (q1, q2: the two slots that follow it).
// Left to the compiler, the loop-carried state of these blocks (cursor, accumulators) was copied between registers on every trip
// and around every stage.  The blocks use v104 .. v127 and s90 .. s95 as temporaries (clobbers).
#include "eval_asm.inc"
#ifndef ER_ES
#define ER_ES 4   // the assembly trip loop: 4 (the default): ER_TRIP4_ASM, one entry per trip, unrolled twice, the rare "beyond the exact cut" block out
                  // of line -- one taken branch per two trips where 1: ER_TRIP_ASM takes four (68.7-68.8 against 69.1-69.2 ms on one box).
                  // Built, measured, correct and slower (-DER_ES=...):
                  // 2: ER_TRIP2_*_ASM, two entries per trip as two interleaved streams -- its 48 temporaries push the records in flight into
                  //    scratch: 72.9 against 69.6 ms at C4;  3: ER_TRIP3_*_ASM, the NEXT trip's entry requested a trip ahead (unrolled twice over
                  //    two register sets, 34 temporaries, 32 spills) -- SQ_WAIT_INST_ANY -3.6 %, SQ_INSTS_VALU +2 %, 72.4 against 69.2 ms on one
                  //    box: the LDS round trip at the start of a trip is not what the SIMDs wait for
#endif

// law coefficients of a [target species][source species] pair as the kernel keeps them in LDS (read once per list and lane)
struct ErLaw
{
  double cN, cY, cS;
  unsigned trow;   // LDS address of the pair's short-range table
  unsigned pad;
};

template <int NG, bool YUK, bool ET>
__global__ __launch_bounds__(ER_MAXWAVES * 64) void k_eval_ring(
    const double4 *__restrict__ n_mom, const int *__restrict__ n_flags, const double4 *__restrict__ s_pm,
    const unsigned char *__restrict__ s_type, const unsigned char *__restrict__ s_active, const double *__restrict__ table,
    WalkParams wp, long long t_first,
    long long t_count, int *__restrict__ counter, double *__restrict__ r_acc, int *__restrict__ r_nint,
    const int *__restrict__ region_base, const int *__restrict__ gcount, long long g_first, long long g_cnt, int lcap, int scap,
    int S, const int *__restrict__ tlist, int SG, int K)
{
  // LDS: [tables][exp table 32][softening per type 8][law table NG x NG][terminal stub][per wave: K slots]
  double *tab_s = reinterpret_cast<double *>(er_smem);
  const int ntabs = wp.ntab_lds + (YUK ? wp.exp_tab : 0);   // distinct short-range tables [+ the exp(-ym r_bin) table of a wiring with a Yukawa law]
  const unsigned tab_bytes = (unsigned)(sizeof(double) * ntabs * NTAB);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  double *expT = reinterpret_cast<double *>(er_smem + tab_bytes);
  double *fsT = expT + 32;   // softening length per particle type (8 entries; index 7 = NULL entry)
  ErLaw *lawT = reinterpret_cast<ErLaw *>(fsT + 8);
  const unsigned fixed_bytes = tab_bytes + 40u * (unsigned)sizeof(double) + (unsigned)(NG * NG * sizeof(ErLaw));
  const unsigned wave_bytes = (unsigned)K * ER_SLOT;
  const unsigned lds0 = (unsigned)(unsigned long)(__attribute__((address_space(3))) unsigned char *)er_smem;   // LDS address of er_smem
  const unsigned term_q = lds0 + fixed_bytes;                     // q of the terminal stub (no entries: only its NULL entry is read)
  const unsigned wbase = (unsigned)__builtin_amdgcn_readfirstlane((int)(term_q + ER_STUB + (unsigned)wave * wave_bytes));
  bool usoft = true;  // all types share one softening length
#pragma unroll
  for(int q = 0; q < NGRAVS_NTYPES; q++)
    usoft = usoft && wp.fsoft[q] == wp.fsoft[0];
  {
    if(threadIdx.x < 32)
      expT[threadIdx.x] = exp2(-(double)threadIdx.x / 32.0);
    if(threadIdx.x >= 32 && threadIdx.x < 40)
      fsT[threadIdx.x - 32] = threadIdx.x - 32 < NGRAVS_NTYPES ? wp.fsoft[threadIdx.x - 32] : 0.0;
    if(threadIdx.x < NG * NG)
      {
        const int a = threadIdx.x / NG, b = threadIdx.x % NG;
        ErLaw l;
        l.cN = wp.cN[a][b];
        l.cY = wp.cY[a][b];
        l.cS = wp.cS[a][b];
        l.trow = lds0 + (unsigned)(wp.tab_slot[a * NG + b] * NTAB * sizeof(double));
        l.pad = 0;
        lawT[threadIdx.x] = l;
      }
    for(int t = threadIdx.x; t < ntabs * NTAB; t += blockDim.x)
      {
        const int u = t / NTAB;
        tab_s[t] = table[(size_t)(u < wp.ntab_lds ? wp.slot_src[u] : NG * NG) * NTAB + (t % NTAB)];
      }
    // the constant parts of this wave's slots: NULL entries, their type bytes, the terminal stub; the type bytes of the entries
    // stay 0 when all particle types share one softening length
    if(lane <= K && (lane < K || wave == 0))
      {
        typedef double d2 __attribute__((ext_vector_type(2)));
        const unsigned qs = lane < K ? wbase + (unsigned)lane * ER_SLOT : term_q;   // lane K of wave 0: the terminal stub
        d2 far, zm;
        far.x = far.y = 1e10;
        zm.x = 1e10;
        zm.y = 0.0;
        ER_AS3(d2, qs + ER_XY - 16u) = far;
        ER_AS3(d2, qs + ER_ZM - 16u) = zm;
        lds_st_u8(qs + ER_TYPE - 1, 7);   // fsT[7] = 0: unsoftened
      }
    if(lane < 8)
      for(int k = 0; k < K; k++)
        lds_st_u32(wbase + (unsigned)k * ER_SLOT + ER_TYPE + 4u * (unsigned)lane, 0u);
    if(wave == 0)
      {
        lds_st_u32(term_q + ER_NM + 4 * lane, 0u);
        if(lane == 0)
          lds_st_u32(term_q + ER_NEXT, term_q);
      }
    __syncthreads();
  }
  // LDS address of the exp(-ym r_bin) table -- which the softening lengths follow at ER_FST_OFF_ET; a kernel without Yukawa laws
  // stages no such table and is handed the address the tables end at (the softening lengths are ER_FST_OFF_NOET behind it)
  const unsigned etab_a = lds0 + (YUK ? (unsigned)(wp.ntab_lds * NTAB * sizeof(double)) : tab_bytes);
  const int G = WAVE / S;
  const long long ngroups = g_cnt;
  unsigned lane_pat = ~0u;   // the pool entries of a block this lane evaluates (S lanes share a target: entry j goes to lane j mod S)
  if(S > 1)
    {
      lane_pat = 0;
      for(int j = lane & (S - 1); j < 32; j += S)
        lane_pat |= 1u << j;
    }
  const double invbox = wp.box > 0 ? 1.0 / wp.box : 0.0;
  double h2max = 0;   // square of the largest softening length of any particle type (wave-uniform)
#pragma unroll
  for(int q = 0; q < NGRAVS_NTYPES; q++)
    h2max = fmax(h2max, wp.fsoft[q] * wp.fsoft[q]);
  h2max = wave_uniform(h2max);
  // XCD-aware group assignment (as k_walk_group2): the Peano order is cut into 8 contiguous segments, one per XCD (own L2); an
  // exhausted segment steals from the others.  Placement only affects speed.
  unsigned xcc = 0;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  xcc &= 7u;
  const long long seg = (ngroups + 7) / 8;
  int steal = 0;
  unsigned long long acc_st[4] = {0, 0, 0, 0};
  const unsigned lane4 = 4u * (unsigned)lane;
  const int mh = lane >> 5;   // MFMA: k index this lane supplies = half of the wave
  // A-operand row of this lane -> pool entry (see k_walk_group2), as a byte offset into a slot
  const int mrow = 16 * ((lane >> 2) & 1) + 4 * ((lane & 31) >> 3) + (lane & 3);
  typedef float f16v __attribute__((ext_vector_type(16)));
  const double BIG = 1e300;
  for(;;)
    {
      long long grp = -1;
      while(steal < 8)
        {
          const int sx = (int)((xcc + steal) & 7u);
          int k = 0;
          if(lane == 0)
            k = atomicAdd(&counter[8 + sx], 1);
          k = __builtin_amdgcn_readfirstlane(k);
          const long long g0 = seg * sx + k;
          if(k < seg && g0 < ngroups)
            {
              grp = g0;
              break;
            }
          steal++;
        }
      if(grp < 0)
        break;
      const long long ru = grp / SG;   // the unit's region in this batch
      const int *const lbase = region_base + (size_t)ru * ((size_t)NG * lcap + scap);
      const long long grel = grp;
      grp += g_first;
      const long long tk = grp * G + lane / S;
      const bool in_range = tk < t_count;
      const long long ti = tlist ? (in_range ? (long long)tlist[tk] : 0ll) : t_first + tk;
      const bool valid = in_range && (tlist != nullptr || (s_active[ti] & 1) != 0);
      double px = 0, py = 0, pz = 0, hT = 0;
      int tg = 0;
      if(valid)
        {
          double4 p = s_pm[ti];
          px = p.x;
          py = p.y;
          pz = p.z;
          int ptype = s_type[ti];
          tg = wp.t2g[ptype];
          hT = wp.fsoft[ptype];
        }
      if(!__any(valid ? 1 : 0))
        continue;
      int n_items[NG];
      bool bad = false;
#pragma unroll
      for(int g = 0; g < NG; g++)
        {
          n_items[g] = __builtin_amdgcn_readfirstlane(gcount[ru * NG + g]);
          bad |= n_items[g] < 0;
        }
      if(bad)   // the traversal kernel overflowed this unit's region: the fused kernel redoes it
        continue;
      // group bounding box (wave-uniform, SGPRs)
      const double lox = wave_min(valid ? px : BIG), hix = wave_max(valid ? px : -BIG);
      const double loy = wave_min(valid ? py : BIG), hiy = wave_max(valid ? py : -BIG);
      const double loz = wave_min(valid ? pz : BIG), hiz = wave_max(valid ? pz : -BIG);
      const double bcx = wave_uniform(0.5 * (lox + hix)), bcy = wave_uniform(0.5 * (loy + hiy)), bcz = wave_uniform(0.5 * (loz + hiz));
      const double bhx = wave_uniform(0.5 * (hix - lox)), bhy = wave_uniform(0.5 * (hiy - loy)), bhz = wave_uniform(0.5 * (hiz - loz));
      const double bhmax = fmax(bhx, fmax(bhy, bhz));
      // may sources be wrapped once per group (relative to the box centre) instead of per pair?
      const bool prewrap = __builtin_amdgcn_readfirstlane(
                               (int)(wp.periodic && (wp.boxhalf - bhmax) * (wp.boxhalf - bhmax) > wp.reach2 && (wp.boxhalf - bhmax) > 0)) != 0;
      const bool lanewrap = wp.periodic && !prewrap;
      // a group whose whole region [box - reach, box + reach] lies inside the periodic box needs no image arithmetic at all
      bool nowrap = !wp.periodic;
      if(wp.periodic && wp.src_in_box)
        {
          const double rl_ = __builtin_sqrt(wp.reach2);
          nowrap = __builtin_amdgcn_readfirstlane((int)(bcx - bhx - rl_ >= 0.0 && bcx + bhx + rl_ <= wp.box && bcy - bhy - rl_ >= 0.0 &&
                                                        bcy + bhy + rl_ <= wp.box && bcz - bhz - rl_ >= 0.0 &&
                                                        bcz + bhz + rl_ <= wp.box)) != 0;
        }
      // fp32 reach pre-test on the matrix cores (no per-pair wrapping): threshold widened by the worst-case rounding so that no
      // true hit is lost; the force loop re-tests in fp64.  Bound as in k_walk_group2: e and p are fp32 roundings of coordinates
      // relative to the box centre (<= bhmax + reach each), the evaluation makes <= 8 roundings of magnitudes <= 3 (2 bhmax + rl)^2
      float mB[2][3];
      {
        const float tfx = (float)(px - bcx), tfy = (float)(py - bcy), tfz = (float)(pz - bcz);
        const float m2x = -2.0f * tfx, m2y = -2.0f * tfy, m2z = -2.0f * tfz;
        float cth;
        const double rl = __builtin_sqrt(wp.reach2);
        const double dl = 4.76837158203125e-07 * (bhmax + rl);   // 2^-21 x the largest relative coordinate
        const double M = 3.0 * (2.0 * bhmax + rl) * (2.0 * bhmax + rl);
        const double thr = (wp.reach2 + 4.0 * rl * dl + 1.0e-6 * M) * (1.0 + 2e-6);
        cth = (float)((double)tfx * tfx + (double)tfy * tfy + (double)tfz * tfz - thr);
        cth = cth - 1.2e-7f * __builtin_fabsf(cth);   // the cast may have rounded up: one ulp down
        // B operand of lane l = B[k = l >> 5][column l & 31]; the target of column c of block tb is lane 32 tb + c
#pragma unroll
        for(int tb = 0; tb < 2; tb++)
          {
            const int src = 32 * tb + (lane & 31);
            const float x_ = __shfl(m2x, src), y_ = __shfl(m2y, src), z_ = __shfl(m2z, src), c_ = __shfl(cth, src);
            mB[tb][0] = mh ? y_ : x_;        // k = 0, 1: -2 tx, -2 ty
            mB[tb][1] = mh ? 1.0f : z_;      // k = 2, 3: -2 tz, 1 (times |e|^2)
            mB[tb][2] = mh ? 0.0f : c_;      // k = 4, 5: |t|^2 - thr (times 1), 0
          }
      }

      double ax = 0, ay = 0, az = 0;
      int nint = 0;
      int st_entries = 0, st_iters = 0;

      // ---- all lists of the group; LW: images per pair (the group's box is too wide to wrap the sources once per group)
      auto run_lists = [&](auto lw_tag) {
        constexpr bool LW = decltype(lw_tag)::value;
        const bool fastmask = !LW && !wp.exact_reach;
        // the pool holds positions RELATIVE to the box centre (what the cull computes anyway) unless images are taken per pair
        const double tpx = LW ? px : px - bcx, tpy = LW ? py : py - bcy, tpz = LW ? pz : pz - bcz;
        for(int g = 0; g < NG; g++)
          {
            const int n = n_items[g < NG ? g : 0];
            const int *__restrict__ items = lbase + (size_t)g * lcap;
            wave_sync();
            st_entries += n;
            if(n == 0)
              continue;
            double cNg, cYg, cSg;
            unsigned trow_a;
            {
              const ErLaw l = lawT[tg * NG + g];
              cNg = l.cN;
              cYg = l.cY;
              cSg = l.cS;
              trow_a = l.trow;
            }
            // quads of four consecutive items in a golden-ratio stride order (see k_walk_group2)
            const int nq = (n + 3) >> 2;
            const int nsuper = (nq + WAVE - 1) / WAVE, nchunks = 4 * nsuper;
            const int M = nsuper * WAVE;
            int s_ = (int)(0.6180339887498949 * M) | 1;
            if(s_ >= M)
              s_ = 1;
            for(;;)
              {
                int a = nsuper, b = s_ % nsuper;
                while(b)
                  {
                    int t = a % b;
                    a = b;
                    b = t;
                  }
                if(a == 1)
                  break;
                s_ += 2;
                if(s_ >= M)
                  {
                    s_ = 1;
                    break;
                  }
              }
            const int step64 = (int)((64ll * s_) % M);
            int slot = (int)(((long long)lane * s_) % M);
            int4 qd = {0, 0, 0, 0};   // item quad of the current four chunks
            int nv = 0;
            // ring state (wave-uniform): blocks tail .. tail + nlive - 1 are complete and not yet left by every lane; wr_pos entries
            // (up to 95: a partial block + one chunk) wait from slot wr_slot on
            int wr_slot = 0, wr_pos = 0, tail_slot = 0, nlive = 0;
            bool first = true;
            unsigned newest_q = term_q;
            unsigned m = 0, q = term_q;   // this lane's cursor: parked on the terminal stub
            // One turn per chunk of 64 items: its records are requested, the oldest blocks are evaluated until the chunk fits into
            // the ring (the requests are in flight meanwhile), the chunk is culled into the ring, the blocks it completes get their
            // masks.  After the last chunk the ring is drained.
            double rx = 0, ry = 0, rz = 0, rw = 0;
            int hs1 = 0;
            for(int cc = 0;; cc++)
              {
                const bool have_chunk = cc < nchunks;
                rw = 0.0;   // the record of this lane's item (no item: massless, the cull drops it, whatever its position says)
                if(have_chunk)
                  {
                    const int bn = cc & 3;
                    if(bn == 0)
                      {
                        nv = 0;   // (no quad: no lane uses qd)
                        if(slot < nq)
                          {
                            qd = reinterpret_cast<const int4 *>(items)[slot];
                            nv = n - 4 * slot;
                          }
                        slot += step64;
                        slot = slot >= M ? slot - M : slot;
                      }
                    const int item = bn == 0 ? qd.x : (bn == 1 ? qd.y : (bn == 2 ? qd.z : qd.w));
                    if(bn < nv)
                      {
                        const bool isp = item >= 0;
                        const unsigned idx = isp ? (unsigned)item : ~(unsigned)item;   // monopole: node * NG + g = -1 - item
                        const double4 *src = isp ? s_pm + idx : n_mom + idx;
                        const double4 r = *src;
                        rx = r.x;
                        ry = r.y;
                        rz = r.z;
                        rw = r.w;
                        if(!usoft)   // (one softening length for all types: no type / flag bytes, each of which would pull another cache line)
                          hs1 = isp ? (int)s_type[idx] : ((n_flags[idx / NG] >> 2) & 7);
                      }
                  }
                // ---- consume: the oldest blocks, until the chunk fits (no chunk left: until the ring is empty)
                while(nlive > 0 && (!have_chunk || K * 32 - (nlive * 32 + wr_pos) < WAVE))
                  {
                  const unsigned tail_q = wbase + (unsigned)tail_slot * ER_SLOT;
                  if constexpr(!LW && (ET || !YUK))
                    {
                      // The trip loop is written out in gfx950 assembly (ER_TRIP_ASM): left to the compiler, the loop-carried state
                      // (cursor, accumulators) was copied between registers on every trip and around every call of the loop.
                      int ntr;
#define ER_TRIP_CALL(ASMTEXT, CLOBBERS)                                                                                              \
  asm volatile(ASMTEXT                                                                                         \
               : [m] "+v"(m), [q] "+v"(q), [ax] "+v"(ax), [ay] "+v"(ay), [az] "+v"(az), [nint] "+v"(nint), [ntr] "=&s"(ntr)           \
               : [tpx] "v"(tpx), [tpy] "v"(tpy), [tpz] "v"(tpz), [cN] "v"(cNg), [cY] "v"(cYg), [cS] "v"(cSg), [hT] "v"(hT),          \
                 [trow] "v"(trow_a), [lane4] "v"(lane4), [tail] "s"(tail_q), [reach2] "s"(wp.reach2), [tiny] "s"(1e-290),           \
                 [asmthfac] "s"(wp.asmthfac), [ec0] "s"(wp.ec[0]), [ec1] "s"(wp.ec[1]), [ec2] "s"(wp.ec[2]), [ec3] "s"(wp.ec[3]),   \
                 [utor2wpi] "s"(wp.utor2wpi), [ym] "s"(wp.ym), [h2max] "s"(h2max), [etab] "s"(etab_a)                              \
               : CLOBBERS)
#if ER_ES == 2
                      if constexpr(YUK)
                        ER_TRIP_CALL(ER_TRIP2_YUK_ASM(ER_FST_OFF_ET), ER_TRIP2_CLOBBERS);
                      else
                        ER_TRIP_CALL(ER_TRIP2_NOYUK_ASM(ER_FST_OFF_NOET), ER_TRIP2_CLOBBERS);
#elif ER_ES == 3
                      if constexpr(YUK)
                        ER_TRIP_CALL(ER_TRIP3_YUK_ASM(ER_FST_OFF_ET), ER_TRIP3_CLOBBERS);
                      else
                        ER_TRIP_CALL(ER_TRIP3_NOYUK_ASM(ER_FST_OFF_NOET), ER_TRIP3_CLOBBERS);
#elif ER_ES == 4
                      if constexpr(YUK)
                        ER_TRIP_CALL(ER_TRIP4_ASM(ER_YUK_ET, ER_FST_OFF_ET), ER_TRIP_CLOBBERS);
                      else
                        ER_TRIP_CALL(ER_TRIP4_ASM(ER_NOYUK, ER_FST_OFF_NOET), ER_TRIP_CLOBBERS);
#else
                      if constexpr(YUK)
                        ER_TRIP_CALL(ER_TRIP_ASM(ER_YUK_ET, ER_FST_OFF_ET), ER_TRIP_CLOBBERS);
                      else
                        ER_TRIP_CALL(ER_TRIP_ASM(ER_NOYUK, ER_FST_OFF_NOET), ER_TRIP_CLOBBERS);
#endif
#undef ER_TRIP_CALL
                      st_iters += ntr;
                    }
                  else
                  for(;;)
                    {
                      if(__builtin_amdgcn_ballot_w64(q == tail_q) == 0ull)
                        break;
                      st_iters += (int)(__builtin_popcountll(__builtin_amdgcn_ballot_w64(true)) >> 6);   // (+1, computed on the scalar unit)
                      int j;
                      asm("v_ffbl_b32 %0, %1" : "=v"(j) : "v"(m));   // an exhausted mask gives -1: the NULL entry in front of the slot
                      const double4 e = er_entry(q, j);
                      const unsigned qe = q;   // (the slot of this trip's entry: the cursor may move on below)
                      // m - 1 carries unless m is 0: the carry IS the mask of the lanes that hold a real entry this trip
                      unsigned long long actm_;
                      unsigned mm1;
                      asm("v_add_co_u32_e64 %0, %1, %2, -1" : "=v"(mm1), "=s"(actm_) : "v"(m));
                      // (an asm result counts as divergent; through readfirstlane the mask stays on the scalar unit)
                      const unsigned long long actm = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(actm_ >> 32)) << 32) |
                                                      (unsigned)__builtin_amdgcn_readfirstlane((int)actm_);
                      m &= mm1;
                      if(m == 0u)   // this lane's bits of the block are used up: follow the link (or stay, if the next block is not there yet)
                        {
                          const unsigned qn = lds_u32(q + ER_NEXT);
                          m = lds_u32(q + ER_NM + lane4);
                          q = qn;
                        }
                      // ---- one pool entry against this lane's target (forcetree.c:1953-2032), as k_walk_group2's evalN
                      double dx = e.x - tpx, dy = e.y - tpy, dz = e.z - tpz;
                      if(LW)
                        {
                          dx = nearest(dx, wp.box, wp.boxhalf);
                          dy = nearest(dy, wp.box, wp.boxhalf);
                          dz = nearest(dz, wp.box, wp.boxhalf);
                        }
                      const double r2 = dx * dx + dy * dy + dz * dz;
                      double mw = e.w;
                      const unsigned long long fpos = actm & __builtin_amdgcn_ballot_w64(!(r2 < wp.reach2));
                      if(fpos != 0ull)                                                  // rare: beyond the exact cut
                        {
                          asm volatile("; beyond the exact cut" ::: "memory");          // keeps this a branch
                          double r2o = r2;
                          asm volatile("" : "+v"(r2o));
                          unsigned long long am = actm;
                          asm volatile("" : "+s"(am));
                          const bool out = ((am >> lane) & 1ull) != 0 && !(r2o < wp.reach2);
                          mw = out ? 0.0 : mw;
                          nint -= out ? 1 : 0;
                        }
                      // self / coincident pairs stay finite (d = 0 kills them)
                      const double q2 = r2 + 1e-290;
                      double ri = __builtin_amdgcn_rsq(q2);
                      ri = ri * (1.5 - 0.5 * q2 * ri * ri);                             // one Newton step: ~2^-51
                      const double rr = q2 * ri;                                        // sqrt(r2) to ~2^-51
                      const double ri2 = ri * ri;
                      double f = cNg * ri2;
                      const double xt = wp.asmthfac * rr;
                      int tab = (int)xt;                                                // saturating conversion, then clamped
                      tab = tab < NTAB - 1 ? tab : NTAB - 1;
                      if(YUK)
                        {
                          double ex_;
                          if(ET)
                            {
                              // exp(-ym r) = E[tab] exp(-u), u = ub * (position inside the bin): degree-4 Taylor in the bin fraction
                              // (Horner with one scalar operand per instruction: the constant bus takes one)
                              const double fb = __builtin_amdgcn_fract(xt);
                              double pz_ = fb * wp.ec[3];
                              pz_ = pz_ - wp.ec[2];
                              pz_ = __builtin_fma(pz_, fb, wp.ec[1]);
                              pz_ = __builtin_fma(pz_, fb, -wp.ec[0]);
                              pz_ = __builtin_fma(pz_, fb, 1.0);
                              ex_ = ER_AS3(const double, etab_a + 8u * (unsigned)tab) * pz_;
                            }
                          else
                            ex_ = exp_neg_fast(rr * wp.ym, expT);
                          f += cYg * ex_ * (wp.ym * ri + ri2);
                        }
                      f -= wp.utor2wpi * ER_AS3(const double, trow_a + 8u * (unsigned)tab);
                      double fac = f * mw * ri;
                      if(wave_any(r2 < h2max))                                          // rare: possibly inside the softening radius
                        {
                          asm volatile("; softened pair" ::: "memory");
                          // (the entry's type byte is at [slot] + ER_TYPE + j)
                          const double h = __builtin_fmax(hT, fsT[lds_u8(qe + (unsigned)j + ER_TYPE)]);   // forcetree.c:1415-1417
                          const bool soft = rr < h;
                          double h_inv = 1 / h, u = rr * h_inv;
                          double v = (u < 0.5) ? (10.666666666667 + u * u * (32.0 * u - 38.4))
                                               : (21.333333333333 - 48.0 * u + 38.4 * u * u - 10.666666666667 * u * u * u - 0.066666666667 / (u * u * u));
                          double fs = cSg * mw * h_inv * h_inv * h_inv * v;
                          fac = soft ? fs : fac;
                        }
                      ax = __builtin_fma(dx, fac, ax);
                      ay = __builtin_fma(dy, fac, ay);
                      az = __builtin_fma(dz, fac, az);
                    }
                  tail_slot = tail_slot + 1 >= K ? 0 : tail_slot + 1;
                  nlive--;
                  }
                if(!have_chunk)
                  break;
                // ---- produce: cull the chunk into the ring
                {
                  const int s1 = wr_slot + 1 >= K ? wr_slot + 1 - K : wr_slot + 1, s2 = wr_slot + 2 >= K ? wr_slot + 2 - K : wr_slot + 2;
                  const unsigned q0 = wbase + (unsigned)wr_slot * ER_SLOT, q1 = wbase + (unsigned)s1 * ER_SLOT, q2 = wbase + (unsigned)s2 * ER_SLOT;
                  if(!LW && usoft)
                    {
                      int cnt;
#define ER_CULL_CALL(WRAPSEG)                                                                                                     \
  asm volatile(ER_CULL_ASM(WRAPSEG)                                                                                                \
               : [cnt] "=&s"(cnt)                                                                                                   \
               : [rx] "v"(rx), [ry] "v"(ry), [rz] "v"(rz), [rw] "v"(rw), [bcx] "s"(bcx), [bcy] "s"(bcy), [bcz] "s"(bcz), [bhx] "s"(bhx),  \
                 [bhy] "s"(bhy), [bhz] "s"(bhz), [reach2] "s"(wp.reach2), [box] "s"(wp.box), [invbox] "s"(invbox), [wrpos] "s"(wr_pos),   \
                 [q0] "s"(q0), [q1] "s"(q1), [q2] "s"(q2)                                                                           \
               : ER_CULL_CLOBBERS)
                      if(nowrap)
                        ER_CULL_CALL("");
                      else
                        ER_CULL_CALL(ER_CULL_WRAP);
#undef ER_CULL_CALL
                      wr_pos += cnt;
                    }
                  else
                    {
                      double ex = rx - bcx, ey = ry - bcy, ez = rz - bcz;
                      if(!nowrap)
                        {
                          ex = nearest_abs(ex, wp.box, invbox);   // a tie (|ex| = box/2) is far beyond any reach
                          ey = nearest_abs(ey, wp.box, invbox);
                          ez = nearest_abs(ez, wp.box, invbox);
                        }
                      // a source farther than the cut from the whole bounding box contributes to no target
                      const double b0 = fmax(0.0, fabs(ex) - bhx), b1 = fmax(0.0, fabs(ey) - bhy), b2 = fmax(0.0, fabs(ez) - bhz);
                      const bool live = rw != 0.0 && (b0 * b0 + b1 * b1 + b2 * b2 < wp.reach2);
                      const unsigned long long lm = __ballot(live ? 1 : 0);
                      if(live)
                        {
                          const int L = wr_pos + lane_prefix(lm);   // < 96: at most three slots from wr_slot on
                          const unsigned sq = L < 32 ? q0 : (L < 64 ? q1 : q2);
                          const unsigned ea = sq + 16u * (unsigned)(L & 31);
                          typedef double d2 __attribute__((ext_vector_type(2)));
                          d2 lo, hi;
                          lo.x = LW ? rx : ex;
                          lo.y = LW ? ry : ey;
                          hi.x = LW ? rz : ez;
                          hi.y = rw;
                          ER_AS3(d2, ea + ER_XY) = lo;
                          ER_AS3(d2, ea + ER_ZM) = hi;
                          if(!usoft)
                            lds_st_u8(sq + ER_TYPE + (unsigned)(L & 31), (unsigned char)(hs1 & 7));
                        }
                      wr_pos += __popcll(lm);
                    }
                  wave_sync();
                }
                // ---- the blocks the chunk completed (after the last chunk: also the partial one) get their masks
                const bool last = cc + 1 >= nchunks;
                while(wr_pos >= 32 || (last && wr_pos > 0))
                  {
                    // ---- the reach masks of the block in slot wr_slot (nfill entries) and its publication
                    const int nfill = wr_pos < 32 ? wr_pos : 32;
                    const unsigned sq = wbase + (unsigned)wr_slot * ER_SLOT;
                    unsigned word = 0;
                    if(fastmask)
                      {
                        const double4 er = er_entry(sq, mrow);
                        const float fx = (float)er.x, fy = (float)er.y, fz = (float)er.z;
                        const float e2 = __builtin_fmaf(fz, fz, __builtin_fmaf(fy, fy, fx * fx));
                        const float a0 = mh ? fy : fx, a1 = mh ? e2 : fz, a2 = mh ? 0.0f : 1.0f;
                        unsigned wt[2];
#pragma unroll
                        for(int tb = 0; tb < 2; tb++)
                          {
                            f16v acc;
#pragma unroll
                            for(int r = 0; r < 16; r++)
                              acc[r] = 0.0f;
                            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, mB[tb][0], acc, 0, 0, 0);
                            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, mB[tb][1], acc, 0, 0, 0);
                            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a2, mB[tb][2], acc, 0, 0, 0);
                            unsigned w_ = 0;
#pragma unroll
                            for(int r = 15; r >= 0; r--)
                              w_ = __builtin_amdgcn_alignbit(w_, __float_as_uint(acc[r]), 31);
                            wt[tb] = w_;
                          }
                        // lanes 32-63 of wt[0] <-> lanes 0-31 of wt[1]: afterwards [0] = this lane's target against entries 0-15, [1] = 16-31
                        const auto sw = __builtin_amdgcn_permlane32_swap(wt[0], wt[1], false, false);
                        word = sw[0] | (sw[1] << 16);
                      }
                    else
                      {
                        for(int b = 0; b < nfill; b++)
                          {
                            const double4 e = er_entry(sq, b);
                            double dx = e.x - tpx, dy = e.y - tpy, dz = e.z - tpz;
                            if(LW)
                              {
                                dx = nearest(dx, wp.box, wp.boxhalf);
                                dy = nearest(dy, wp.box, wp.boxhalf);
                                dz = nearest(dz, wp.box, wp.boxhalf);
                              }
                            const double r2 = dx * dx + dy * dy + dz * dz;
                            word |= (r2 < wp.reach2) ? (1u << b) : 0u;
                          }
                      }
                    // entries beyond nfill are stale; lanes without a target take nothing; S lanes of a target share the entries
                    const unsigned okm = !valid ? 0u : (nfill >= 32 ? ~0u : ((1u << nfill) - 1u));
                    word &= okm & lane_pat;
                    nint += __popc(word);   // the force loop takes the (rare) slots beyond the exact cut off again
                    if(first)
                      {
                        m = word;
                        q = sq;
                        first = false;
                      }
                    else
                      {
                        lds_st_u32(newest_q + ER_NM + lane4, word);
                        if(lane == 0)
                          lds_st_u32(newest_q + ER_NEXT, sq);
                      }
                    lds_st_u32(sq + ER_NM + lane4, 0u);
                    if(lane == 0)
                      lds_st_u32(sq + ER_NEXT, sq);
                    newest_q = sq;
                    wave_sync();
                    wr_pos -= nfill;
                    wr_slot = wr_slot + 1 >= K ? 0 : wr_slot + 1;
                    nlive++;
                  }
                if(last)
                  {
                    if(!first && lane == 0)
                      lds_st_u32(newest_q + ER_NEXT, term_q);   // the list ends here
                    wave_sync();
                  }
              }
            wave_sync();
          }
      };
      if(lanewrap)
        run_lists(std::true_type{});
      else
        run_lists(std::false_type{});

      {
        int st_nodes = 0, st_batches = 0;
        if(grel % SG == 0)   // the unit's traversal statistics, once
          {
            const long long u_cnt = (g_cnt + SG - 1) / SG;
            st_nodes = __builtin_amdgcn_readfirstlane(gcount[NG * u_cnt + ru]);
            st_batches = __builtin_amdgcn_readfirstlane(gcount[(NG + 1) * u_cnt + ru]);
          }
        acc_st[0] += (unsigned long long)st_entries;
        acc_st[1] += (unsigned long long)st_nodes;
        acc_st[2] += (unsigned long long)st_batches;
        acc_st[3] += (unsigned long long)st_iters;
      }
      if(S > 1)
        {
          for(int off = 1; off < S; off <<= 1)   // the S lanes of a target hold partial sums
            {
              ax += __shfl_xor(ax, off);
              ay += __shfl_xor(ay, off);
              az += __shfl_xor(az, off);
              nint += __shfl_xor(nint, off);
            }
        }
      if(valid && (lane & (S - 1)) == 0)
        {
          r_acc[3 * ti + 0] = ax;
          r_acc[3 * ti + 1] = ay;
          r_acc[3 * ti + 2] = az;
          r_nint[ti] = nint;
        }
    }
  if(lane == 0)
    {
      unsigned long long *st64 = reinterpret_cast<unsigned long long *>(counter + 16);
#pragma unroll
      for(int qq = 0; qq < 4; qq++)
        if(acc_st[qq])
          atomicAdd(&st64[qq], acc_st[qq]);
    }
}

// ---- host side ---------------------------------------------------------------------------------------------------
static size_t er_fixed_bytes(const WalkParams &wp, bool yuk)
{
  return sizeof(double) * (size_t)(wp.ntab_lds + (yuk ? wp.exp_tab : 0)) * NTAB + 40 * sizeof(double) + (size_t)wp.ng * wp.ng * sizeof(ErLaw) +
         ER_STUB;   // (+ the terminal stub all waves share)
}
// How many ring slots fit beside the tables for `waves` waves (0: the ring kernel cannot run -- fewer than 4 slots)
int eval_ring_slots(const WalkParams &wp, bool yuk, int waves)
{
  const size_t fixed = er_fixed_bytes(wp, yuk);
  const size_t avail = 160 * 1024 - fixed;
  const long long per_wave = (long long)(avail / (size_t)waves);
  long long K = per_wave / (long long)ER_SLOT;
  if(K > 8)
    K = 8;
  return K >= 4 ? (int)K : 0;
}

template <int NG, bool YUK, bool ET>
static int launch_eval_ring_t(ngravs_ctx *c, const TreeView &tv, const WalkParams &wp, int nblk, int waves, int K, const int *region,
                              const int *gcount, long long g0, long long nb, int lcap, int scap, int S, const int *tlist, int SG,
                              long long t_count)
{
  const size_t fixed = er_fixed_bytes(wp, YUK);
  const size_t lds = fixed + (size_t)waves * ((size_t)K * ER_SLOT);
  auto ke = k_eval_ring<NG, YUK, ET>;
  HIP_TRY(c, hipFuncSetAttribute(reinterpret_cast<const void *>(ke), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(ke, dim3((unsigned)nblk), dim3(waves * 64), lds, c->stream, tv.mom, tv.flags, c->s_pm.p, c->s_type.p, c->s_active.p, c->table.p, wp,
                     (long long)c->shard_first, t_count, c->walk_counters.p, c->r_acc.p, c->r_nint.p, region, gcount, g0, nb, lcap, scap,
                     S, tlist, SG, K);
  return NGRAVS_OK;
}

int launch_eval_ring(ngravs_ctx *c, const TreeView &tv, const WalkParams &wp, bool yuk, int nblk, int waves, int K, const int *region,
                     const int *gcount, long long g0, long long nb, int lcap, int scap, int S, const int *tlist, int SG, long long t_count)
{
#define ER_GO(NG_, Y_, E_) launch_eval_ring_t<NG_, Y_, E_>(c, tv, wp, nblk, waves, K, region, gcount, g0, nb, lcap, scap, S, tlist, SG, t_count)
  const bool et = yuk && wp.exp_tab;   // Yukawa factor through the table bins
  switch(c->cfg.n_gravs)
    {
    case 1:
      return yuk ? (et ? ER_GO(1, true, true) : ER_GO(1, true, false)) : ER_GO(1, false, false);
    case 2:
      return yuk ? (et ? ER_GO(2, true, true) : ER_GO(2, true, false)) : ER_GO(2, false, false);
    default:
      return yuk ? (et ? ER_GO(3, true, true) : ER_GO(3, true, false)) : ER_GO(3, false, false);
    }
#undef ER_GO
}
