// What a compacted (target, entry) PAIR LIST would add to every trip of the evaluation kernel's force loop,
// measured in isolation on gfx950 with the evaluation kernel's occupancy (16 waves per CU, 4 per SIMD):
//   T0  the yardstick: a trip of the present loop as issue slots -- 49 full-rate fp64 VALU instructions + one v_rsq_f64
//   T1  T0 + the trip fetches ITS target's parameters from the lane that owns the target (position 3 x f64, softening/type word:
//       7 ds_bpermute_b32) -- in the present loop they are loop-invariant registers
//   T2  T0 + the trip's three force components are summed per target across the lanes (pairs of one target are consecutive lanes:
//       a segmented inclusive scan by key, 6 steps x 3 doubles through ds_bpermute) and the segment's last lane hands the sum to the
//       lane that owns the target (3 x f64 = 6 ds_bpermute_b32) -- in the present loop a lane adds into its own registers
//   T3  T0 + both
// Prints the time of each variant and the cost of a trip in units of T0's 52 issue slots.
//   hipcc -O3 --offload-arch=gfx950 -o pairlist_cost pairlist_cost.hip && ./pairlist_cost
#include <hip/hip_runtime.h>
#include <cstdio>
#define TRIPS 2048

__device__ __forceinline__ double bperm(int src_lane, double v)
{
  const int lo = __builtin_amdgcn_ds_bpermute(src_lane << 2, __double2loint(v));
  const int hi = __builtin_amdgcn_ds_bpermute(src_lane << 2, __double2hiint(v));
  return __hiloint2double(hi, lo);
}

template <int MODE> __global__ __launch_bounds__(1024) void k(double *out, const int *__restrict__ seg, double a, double b)
{
  const int lane = threadIdx.x & 63;
  // this lane's target: position, a type word; its accumulators
  double tx = a + lane * 1e-3, ty = a - lane * 1e-3, tz = a + lane * 2e-3;
  int tw = lane & 7;
  double ax = 0, ay = 0, az = 0;
  double x[4];
  for(int q = 0; q < 4; q++)
    x[q] = a + q * 0.125 + threadIdx.x * 1e-9;
  // the pair list of a trip: lane -> target (key), from a table so that the compiler cannot fold it; segments of ~5 lanes
  for(int t = 0; t < TRIPS; t++)
    {
      const int key = seg[((t & 15) << 6) + lane];   // target of this lane's pair in this trip (non-decreasing across the lanes)
      double px = tx, py = ty, pz = tz;
      int pw = tw;
      if(MODE & 1)
        {
          px = bperm(key, tx);
          py = bperm(key, ty);
          pz = bperm(key, tz);
          pw = __builtin_amdgcn_ds_bpermute(key << 2, tw);
        }
      // the yardstick: 48 fma in four chains + rsq + one more
      x[0] += px * 1e-30 + (double)pw * 1e-30;
#pragma unroll
      for(int r = 0; r < 12; r++)
#pragma unroll
        for(int q = 0; q < 4; q++)
          asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(x[q]) : "v"(b));
      asm volatile("v_rsq_f64 %0, %0" : "+v"(x[3]));
      double fx = x[0] + py * 1e-30, fy = x[1] + pz * 1e-30, fz = x[2];
      if(MODE & 2)
        {
          // segmented inclusive scan (sum of the lanes of the same key at or before this lane)
#pragma unroll
          for(int s = 1; s < 64; s <<= 1)
            {
              const int from = lane - s;
              const int k2 = __builtin_amdgcn_ds_bpermute((from < 0 ? lane : from) << 2, key);
              const double gx = bperm(from < 0 ? lane : from, fx), gy = bperm(from < 0 ? lane : from, fy), gz = bperm(from < 0 ? lane : from, fz);
              const bool same = from >= 0 && k2 == key;
              fx += same ? gx : 0.0;
              fy += same ? gy : 0.0;
              fz += same ? gz : 0.0;
            }
          // the last lane of a segment holds the target's sum of this trip; the owner fetches it (lane of the segment end from the
          // table as well: the prefix sums of the per-target pair counts give it without a search)
          const int endl = seg[1024 + ((t & 15) << 6) + lane];   // lane that ends the segment of target `lane` in this trip, or -1
          const double sx = bperm(endl < 0 ? lane : endl, fx), sy = bperm(endl < 0 ? lane : endl, fy), sz = bperm(endl < 0 ? lane : endl, fz);
          ax += endl >= 0 ? sx : 0.0;
          ay += endl >= 0 ? sy : 0.0;
          az += endl >= 0 ? sz : 0.0;
        }
      else
        {
          ax += fx;
          ay += fy;
          az += fz;
        }
    }
  out[blockIdx.x * blockDim.x + threadIdx.x] = ax + ay + az + x[3];
}

template <int MODE> static double run(double *d, const int *seg, int blocks)
{
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(1024), 0, 0, d, seg, 1.0000001, 0.9999999);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(1024), 0, 0, d, seg, 1.0000001, 0.9999999);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  return ms;
}

int main()
{
  const int blocks = 256 * 4;   // 4 rounds of one 1024-thread workgroup (16 waves) per CU
  double *d;
  int *seg, h[2048];
  hipMalloc(&d, sizeof(double) * blocks * 1024);
  hipMalloc(&seg, sizeof(h));
  // 16 trip patterns: segments of 3..8 lanes (a target has ~12.5 hits per block of 64 entries: its pairs fill ~5 lanes of ~2.5 trips)
  unsigned s = 12345;
  for(int t = 0; t < 16; t++)
    {
      int lane = 0, key = 0;
      for(int l = 0; l < 64; l++)
        h[1024 + 64 * t + l] = -1;
      while(lane < 64)
        {
          s = s * 1664525u + 1013904223u;
          int len = 3 + (int)((s >> 16) % 6);
          if(lane + len > 64)
            len = 64 - lane;
          for(int q = 0; q < len; q++)
            h[64 * t + lane + q] = key;
          h[1024 + 64 * t + key] = lane + len - 1;
          lane += len;
          key++;
        }
    }
  hipMemcpy(seg, h, sizeof(h), hipMemcpyHostToDevice);
  const double t0 = run<0>(d, seg, blocks), t1 = run<1>(d, seg, blocks), t2 = run<2>(d, seg, blocks), t3 = run<3>(d, seg, blocks);
  printf("{\"trips_per_wave\": %d, \"waves\": %d, \"ms\": {\"T0_yardstick_52_slots\": %.3f, \"T1_plus_target_fetch\": %.3f, "
         "\"T2_plus_segmented_sum\": %.3f, \"T3_both\": %.3f},\n \"trip_cost_in_units_of_T0\": {\"T1\": %.3f, \"T2\": %.3f, \"T3\": %.3f},\n"
         " \"extra_issue_slots_per_trip\": {\"target_fetch\": %.1f, \"segmented_sum\": %.1f, \"both\": %.1f}}\n",
         TRIPS, blocks * 16, t0, t1, t2, t3, t1 / t0, t2 / t0, t3 / t0, 52.0 * (t1 / t0 - 1), 52.0 * (t2 / t0 - 1), 52.0 * (t3 / t0 - 1));
  return 0;
}
