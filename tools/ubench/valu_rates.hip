// Issue cost of a few fp64 / conversion instructions on gfx950, relative to v_fma_f64: 16 waves per CU (4 per SIMD, as the
// evaluation kernel), eight independent chains per lane, 4096 instructions of the tested kind per chain and wave.
//   hipcc -O3 --offload-arch=gfx950 -o valu_rates valu_rates.hip && ./valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define REP 512
template <int OP> __global__ __launch_bounds__(1024) void k(double *out, double a, double b)
{
  double x[8];
  for(int q = 0; q < 8; q++)
    x[q] = a + q * 0.125 + threadIdx.x * 1e-9;
  for(int it = 0; it < REP; it++)
    {
#pragma unroll
      for(int q = 0; q < 8; q++)
        {
          if(OP == 0)
            asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(x[q]) : "v"(b));
          if(OP == 1)
            asm volatile("v_rsq_f64 %0, %0" : "+v"(x[q]));
          if(OP == 2)
            asm volatile("v_fract_f64 %0, %0" : "+v"(x[q]));
          if(OP == 3)
            {
              int t;
              asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(t) : "v"(x[q]));
              asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(x[q]) : "v"(t));
            }
          if(OP == 4)
            asm volatile("v_add_f64 %0, %0, %1" : "+v"(x[q]) : "v"(b));
          if(OP == 5)
            asm volatile("v_mul_f64 %0, %0, %1" : "+v"(x[q]) : "v"(b));
          if(OP == 6)
            asm volatile("v_rcp_f64 %0, %0" : "+v"(x[q]));
          if(OP == 7)
            asm volatile("v_max_f64 %0, %0, %1" : "+v"(x[q]) : "v"(b));
        }
    }
  double s = 0;
  for(int q = 0; q < 8; q++)
    s += x[q];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int OP> static double run(double *d, int blocks)
{
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(1024), 0, 0, d, 1.0000001, 0.9999999);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(1024), 0, 0, d, 1.0000001, 0.9999999);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  return ms;
}
int main()
{
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  const int blocks = p.multiProcessorCount * 8;   // 8 rounds of one 16-wave workgroup per CU
  double *d;
  hipMalloc(&d, sizeof(double) * blocks * 1024);
  const char *names[8] = {"v_fma_f64", "v_rsq_f64", "v_fract_f64", "v_cvt_i32_f64 + v_cvt_f64_i32", "v_add_f64", "v_mul_f64",
                          "v_rcp_f64", "v_max_f64"};
  double t[8];
  t[0] = run<0>(d, blocks);
  t[1] = run<1>(d, blocks);
  t[2] = run<2>(d, blocks);
  t[3] = run<3>(d, blocks);
  t[4] = run<4>(d, blocks);
  t[5] = run<5>(d, blocks);
  t[6] = run<6>(d, blocks);
  t[7] = run<7>(d, blocks);
  // instructions of the tested group per SIMD: 8 rounds x 4 waves x REP x 8
  const double groups = 8.0 * 4 * REP * 8;
  for(int i = 0; i < 8; i++)
    printf("%-34s %8.3f ms  %6.2f cycles per group and SIMD at 2.4 GHz  (x %.2f of v_fma_f64)\n", names[i], t[i],
           t[i] * 1e-3 * 2.4e9 / groups, t[i] / t[0]);
  return 0;
}
