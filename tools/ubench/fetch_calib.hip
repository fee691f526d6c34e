// What does FETCH_SIZE (rocprofv3 --pmc) report for the access shapes of the walk's evaluation kernel?  The guide calibrates it
// for wide coalesced streaming reads only (gfx950: 1/2 of the bytes).  Every kernel here reads a KNOWN number of bytes exactly
// once, from buffers far larger than the 256 MiB Infinity Cache, in one of the shapes the evaluation kernel uses:
//   k_stream16   16 B per lane, coalesced, streaming                       (the guide's calibrated case: the control)
//   k_quad_list  16 B per lane, the 64 lanes of a wave at 64 places spread over one item list (golden-ratio stride order over
//                an 8192-entry list, as the evaluation kernel visits its lists); every list is read once by four waves in turn
//                is NOT modelled here -- one wave per list, so each byte is fetched once
//   k_gather32   32 B (double4) per lane at random record indices, every record exactly once (the pool gathers)
//   k_gather32w  the same, but the records of a wave come from one window of 4096 neighbouring records (a group's neighbourhood)
// Prints one JSON line per kernel with the bytes it read; tools/ubench/fetch_calib.sh joins them with the counter values.
//   hipcc -O3 --offload-arch=gfx950 -o fetch_calib fetch_calib.hip && rocprofv3 --pmc FETCH_SIZE ... -- ./fetch_calib
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <numeric>
#include <algorithm>
#include <random>
#define CK(x) do { hipError_t e = (x); if(e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while(0)

__global__ void k_stream16(const int4 *__restrict__ src, long long n, int *__restrict__ out)
{
  int acc = 0;
  for(long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
    {
      const int4 v = src[i];
      acc += v.x ^ v.y ^ v.z ^ v.w;
    }
  if(acc == 0x7fffffff)
    out[0] = acc;
}

// one wave per list of LIST ints (= LIST / 4 quads); the wave reads the quads in chunks of 64, lane l of chunk c at quad
// ((c * 64 + l) * STRIDE) mod NQ with STRIDE coprime to NQ: consecutive lanes are far apart in the list
#define LIST 8192
__global__ void k_quad_list(const int4 *__restrict__ lists, long long nlists, int *__restrict__ out)
{
  const int lane = threadIdx.x & 63;
  const long long wave = (blockIdx.x * (long long)blockDim.x + threadIdx.x) >> 6;
  const int NQ = LIST / 4, STRIDE = 1265;   // ~ NQ / golden ratio, odd: coprime to 2048
  int acc = 0;
  if(wave < nlists)
    {
      const int4 *l = lists + wave * NQ;
      for(int c = 0; c < NQ / 64; c++)
        {
          const int4 v = l[((c * 64 + lane) * STRIDE) & (NQ - 1)];
          acc += v.x ^ v.y ^ v.z ^ v.w;
        }
    }
  if(acc == 0x7fffffff)
    out[0] = acc;
}

__global__ void k_gather32(const double4 *__restrict__ rec, const unsigned int *__restrict__ idx, long long n, double *__restrict__ out)
{
  double acc = 0;
  for(long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
    {
      const double4 v = rec[idx[i]];
      acc += v.x + v.y + v.z + v.w;
    }
  if(acc == 1.2345e300)
    out[0] = acc;
}

int main()
{
  const long long NBYTES = 2ll << 30;   // 2 GiB per test buffer: 8 x the Infinity Cache
  int *d_out;
  double *d_outd;
  CK(hipMalloc(&d_out, 64));
  CK(hipMalloc(&d_outd, 64));
  void *buf;
  CK(hipMalloc(&buf, NBYTES));
  CK(hipMemset(buf, 1, NBYTES));
  // 1. stream
  {
    const long long n = NBYTES / 16;
    hipLaunchKernelGGL(k_stream16, dim3(256 * 16), dim3(256), 0, 0, (const int4 *)buf, n, d_out);
    CK(hipDeviceSynchronize());
    printf("{\"kernel\": \"k_stream16\", \"bytes\": %lld}\n", NBYTES);
  }
  // 2. quad list
  {
    const long long nlists = NBYTES / (LIST * 4);
    hipLaunchKernelGGL(k_quad_list, dim3((unsigned)((nlists * 64 + 255) / 256)), dim3(256), 0, 0, (const int4 *)buf, nlists, d_out);
    CK(hipDeviceSynchronize());
    printf("{\"kernel\": \"k_quad_list\", \"bytes\": %lld}\n", nlists * LIST * 4);
  }
  // 3 / 4. gathers: a random permutation of all records; then permutations inside windows of 4096 records
  {
    const long long n = NBYTES / 32;
    std::vector<unsigned int> h((size_t)n);
    std::iota(h.begin(), h.end(), 0u);
    std::mt19937_64 rng(12345);
    std::shuffle(h.begin(), h.end(), rng);
    unsigned int *d_idx;
    CK(hipMalloc(&d_idx, sizeof(unsigned int) * n));
    CK(hipMemcpy(d_idx, h.data(), sizeof(unsigned int) * n, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_gather32, dim3(256 * 16), dim3(256), 0, 0, (const double4 *)buf, d_idx, n, d_outd);
    CK(hipDeviceSynchronize());
    printf("{\"kernel\": \"k_gather32\", \"bytes\": %lld, \"index_bytes\": %lld, \"order\": \"random over all records\"}\n", n * 32, n * 4);
    std::iota(h.begin(), h.end(), 0u);
    for(long long w = 0; w + 4096 <= n; w += 4096)
      std::shuffle(h.begin() + w, h.begin() + w + 4096, rng);
    CK(hipMemcpy(d_idx, h.data(), sizeof(unsigned int) * n, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_gather32, dim3(256 * 16), dim3(256), 0, 0, (const double4 *)buf, d_idx, n, d_outd);
    CK(hipDeviceSynchronize());
    printf("{\"kernel\": \"k_gather32\", \"bytes\": %lld, \"index_bytes\": %lld, \"order\": \"random inside windows of 4096 records (128 KB)\"}\n", n * 32, n * 4);
    CK(hipFree(d_idx));
  }
  CK(hipFree(buf));
  return 0;
}
